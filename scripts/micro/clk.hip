// micro: shader clock vs 100 MHz realtime counter; dependent-op latencies for f32/f64 FMA, LDS, DPP+readlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out) {
  __shared__ float lds[256];
  const int lane = threadIdx.x;
  lds[lane] = lane;
  __syncthreads();
  unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  float a = lane * 1e-3f;
  for (int i = 0; i < 4096; i++) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  double d = lane * 1e-3;
  for (int i = 0; i < 4096; i++) d = __builtin_fma(d, 1.0001, 0.5);
  unsigned long long c2 = __builtin_readcyclecounter();
  int idx = lane;
  for (int i = 0; i < 1024; i++) idx = (int)lds[idx & 255] & 255;
  unsigned long long c3 = __builtin_readcyclecounter();
  int v = lane;
  for (int i = 0; i < 1024; i++) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v = __builtin_amdgcn_readlane(v, 3) + lane;
  }
  unsigned long long c4 = __builtin_readcyclecounter();
  float q = lane + 1.5f;
  for (int i = 0; i < 1024; i++) q = 1.0f / q + 1.5f;
  unsigned long long c5 = __builtin_readcyclecounter();
  if (lane == 0) {
    out[0] = c1 - c0; out[1] = r1 - r0; out[2] = c2 - c1; out[3] = c3 - c2; out[4] = c4 - c3; out[5] = c5 - c4;
    out[6] = (unsigned long long)(a + d + idx + v + q);
  }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize(); }
  unsigned long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  printf("f32 fma chain: %.1f cyc/op; cycles=%llu realtime(100MHz)=%llu -> clock %.0f MHz\n", h[0] / 4096.0, h[0], h[1], h[0] / (h[1] / 100.0));
  printf("f64 fma chain: %.1f cyc/op\nLDS dependent load: %.1f cyc\nDPP+readlane round: %.1f cyc\nf32 correctly-rounded div+add: %.1f cyc\n", h[2] / 4096.0, h[3] / 1024.0, h[4] / 1024.0, h[5] / 1024.0);
  return 0;
}
