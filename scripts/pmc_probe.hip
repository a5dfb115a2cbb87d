// Known-byte probes for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (measurement aid, not part of the product):
// each kernel streams N bytes (N = 1 GiB by default, well past the 256 MiB Infinity Cache) with ONE access width, so that
// counter value x unit / N is the factor by which the counter under- or over-reports that access pattern.
//   hipcc --offload-arch=gfx950 -O3 scripts/pmc_probe.hip -o build_variants/pmc_probe
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o p -- build_variants/pmc_probe   (and again with WRITE_SIZE)
// scripts/pmc_calibrate.py runs both passes and reduces them.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ __launch_bounds__(256) void probe_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
template <typename T>
__global__ __launch_bounds__(256) void probe_read(const T* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = src[i];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
  }
  if (acc == 0x12345678u) sink[0] = acc;  // (never true for the fill pattern: keeps the loads alive without a store per thread)
}
template <typename T>
__global__ __launch_bounds__(256) void probe_write(T* __restrict__ dst, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
// the LK template-cache pattern: every wave moves six 1 KB rows (one dwordx4 per lane and row) of its own 6 KB record
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe_rows16_write(u4* __restrict__ dst, size_t nrec) {
  for (size_t r = blockIdx.x; r < nrec; r += gridDim.x)
    for (int k = 0; k < 6; k++) __builtin_nontemporal_store(u4{threadIdx.x, (unsigned)k, 3u, 4u}, dst + r * 384 + 64 * k + threadIdx.x);
}
__global__ __launch_bounds__(64) void probe_rows16_read(const u4* __restrict__ src, uint32_t* __restrict__ sink, size_t nrec) {
  uint32_t acc = 0;
  for (size_t r = blockIdx.x; r < nrec; r += gridDim.x)
    for (int k = 0; k < 6; k++) {
      const u4 v = __builtin_nontemporal_load(src + r * 384 + 64 * k + threadIdx.x);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  if (acc == 0x12345678u) sink[0] = acc;
}
// byte loads of an image (the index-reflecting slow paths): one byte per lane
__global__ __launch_bounds__(256) void probe_read1(const uint8_t* __restrict__ src, uint32_t* __restrict__ sink, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));        \
      return 1;                                                      \
    }                                                                \
  } while (0)

int main(int argc, char** argv) {
  const size_t N = argc > 1 ? strtoull(argv[1], nullptr, 0) : (size_t)1 << 30;
  const int reps = 3;
  uint8_t *a = nullptr, *b = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc(&a, N));
  CK(hipMalloc(&b, N));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0x5a, N));
  CK(hipMemset(b, 0, N));
  const int G = 256 * 16;
  for (int r = 0; r < reps; r++) {
    hipLaunchKernelGGL(probe_copy<uint4>, dim3(G), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, N / 16);
    hipLaunchKernelGGL(probe_copy<uint32_t>, dim3(G), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, N / 4);
    hipLaunchKernelGGL(probe_read<uint4>, dim3(G), dim3(256), 0, 0, (const uint4*)a, sink, N / 16);
    hipLaunchKernelGGL(probe_read<uint2>, dim3(G), dim3(256), 0, 0, (const uint2*)a, sink, N / 8);
    hipLaunchKernelGGL(probe_read<uint32_t>, dim3(G), dim3(256), 0, 0, (const uint32_t*)a, sink, N / 4);
    hipLaunchKernelGGL(probe_read1, dim3(G), dim3(256), 0, 0, (const uint8_t*)a, sink, N / 4);   // a quarter of the buffer (slow)
    hipLaunchKernelGGL(probe_write<uint4>, dim3(G), dim3(256), 0, 0, (uint4*)b, N / 16, uint4{1, 2, 3, 4});
    hipLaunchKernelGGL(probe_write<double>, dim3(G), dim3(256), 0, 0, (double*)b, N / 8, 1.5);
    hipLaunchKernelGGL(probe_write<uint32_t>, dim3(G), dim3(256), 0, 0, (uint32_t*)b, N / 4, 7u);
    hipLaunchKernelGGL(probe_rows16_write, dim3(4 * G), dim3(64), 0, 0, (u4*)b, N / 6144);
    hipLaunchKernelGGL(probe_rows16_read, dim3(4 * G), dim3(64), 0, 0, (const u4*)a, sink, N / 6144);
    CK(hipDeviceSynchronize());
  }
  printf("pmc_probe: N = %zu bytes, %d repetitions\n", N, reps);
  return 0;
}
