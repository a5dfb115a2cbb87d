// Which kind of kernel does a host -> device upload slow down?  Three probes of 64 workgroups (the shape of the tracker's one-workgroup-
// per-stream kernels), timed alone and beside a continuous stream of 20 MB pinned uploads on a copy stream:
//   chase    dependent loads through a 64 MB table in device memory (HBM / L2 latency)
//   args     a trivial kernel with a 700-byte by-value argument (the tracker's Pipe bundle): launch + kernarg fetch
//   alu      registers and LDS only
//   hostrd   each workgroup reads 4 KB of page-locked HOST memory (what k_frame_head does with the zero-copy input block)
// build: hipcc --offload-arch=gfx950 -O3 -o build_variants/h2d_probe scripts/micro/h2d_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Big { double v[88]; };  // ~700 B

__global__ __launch_bounds__(64) void k_chase(const unsigned* __restrict__ tab, int steps, unsigned* out) {
  unsigned i = (blockIdx.x * 64 + threadIdx.x) * 9973u & ((1u << 24) - 1);
  for (int k = 0; k < steps; k++) i = tab[i];
  out[blockIdx.x * 64 + threadIdx.x] = i;
}
__global__ __launch_bounds__(64) void k_args(Big b, double* out) {
  double s = 0;
  for (int k = threadIdx.x & 7; k < 88; k += 8) s += b.v[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_alu(int steps, double* out) {
  __shared__ double sm[64];
  double x = threadIdx.x * 1e-3 + 1.0;
  for (int k = 0; k < steps; k++) {
    sm[threadIdx.x] = x;
    __syncthreads();
    x = x * 1.0000001 + sm[(threadIdx.x + 1) & 63] * 1e-9;
    __syncthreads();
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void k_hostrd(const double* __restrict__ host, double* out) {
  double s = 0;
  for (int k = 0; k < 8; k++) s += host[(size_t)blockIdx.x * 512 + k * 64 + threadIdx.x];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
  const size_t TAB = 1u << 24;  // 64 MB of unsigned
  std::vector<unsigned> perm(TAB);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  unsigned *d_tab, *d_out;
  double *d_dout, *h_in, *d_hin;
  CK(hipMalloc(&d_tab, TAB * 4));
  CK(hipMalloc(&d_out, 64 * 64 * 4));
  CK(hipMalloc(&d_dout, 64 * 64 * 8));
  CK(hipMemcpy(d_tab, perm.data(), TAB * 4, hipMemcpyHostToDevice));
  CK(hipHostMalloc((void**)&h_in, 64 * 512 * 8, hipHostMallocMapped));
  for (int i = 0; i < 64 * 512; i++) h_in[i] = i;
  CK(hipHostGetDevicePointer((void**)&d_hin, h_in, 0));
  const size_t UP = 20u << 20;
  void *h_up, *d_up;
  CK(hipHostMalloc(&h_up, UP, 0));
  CK(hipMalloc(&d_up, UP));
  hipStream_t cs, ks;
  CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  Big big;
  for (int i = 0; i < 88; i++) big.v[i] = i;
  const char* names[4] = {"chase", "args", "alu", "hostrd"};
  for (int pass = 0; pass < 2; pass++) {
    for (int mode = 0; mode < 2; mode++) {  // 0 alone, 1 beside uploads
      for (int probe = 0; probe < 4; probe++) {
        if (mode) for (int k = 0; k < 400; k++) CK(hipMemcpyAsync(d_up, h_up, UP, hipMemcpyHostToDevice, cs));  // ~140 ms of uploads queued
        std::vector<float> ms;
        for (int rep = 0; rep < 60; rep++) {
          CK(hipEventRecord(e0, ks));
          switch (probe) {
            case 0: hipLaunchKernelGGL(k_chase, dim3(64), dim3(64), 0, ks, d_tab, 40, d_out); break;
            case 1: hipLaunchKernelGGL(k_args, dim3(64), dim3(64), 0, ks, big, d_dout); break;
            case 2: hipLaunchKernelGGL(k_alu, dim3(64), dim3(64), 0, ks, 300, d_dout); break;
            case 3: hipLaunchKernelGGL(k_hostrd, dim3(64), dim3(64), 0, ks, d_hin, d_dout); break;
          }
          CK(hipEventRecord(e1, ks));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          ms.push_back(t * 1000.f);
        }
        const bool still = hipStreamQuery(cs) == hipErrorNotReady;
        CK(hipStreamSynchronize(cs));
        std::sort(ms.begin(), ms.end());
        printf("pass %d  %-7s %-14s p10 %7.1f  p50 %7.1f  p90 %7.1f us%s\n", pass, names[probe], mode ? "beside uploads" : "alone", ms[6], ms[30], ms[54],
               mode && !still ? "  (the uploads ended before the probe did)" : "");
      }
    }
  }
  return 0;
}
