// flvis_amd device-side common helpers (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FLVIS_WAVE 64

namespace flvis {

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// reflect, then clamp (tiles hanging far over the image edge read defined-but-unused pixels)
__device__ __forceinline__ int reflect101c(int i, int n) {
  i = reflect101(i, n);
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// order-preserving float -> uint32 (larger float => larger uint)
__device__ __forceinline__ uint32_t f32_ordered(float f) {
  uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t u) {
  uint32_t b = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(b);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// The one-workgroup-per-stream kernels of the frame chain run beside full-chip kernels of the other stream (corner response, LK): with
// FLVIS_CHAIN_PRIO = 1 .. 3 (build knob) their waves ask for the SIMD's issue slots first (s_setprio).  Measured in round 5 (session
// s40: 56.9k / 56.7k frames/s at 0, 56.6k at 1, 56.5k / 56.5k at 3; every stage time unchanged to the microsecond): the chain's kernels
// are not held up by their neighbours' instruction issue -- the default stays 0.
#ifndef FLVIS_CHAIN_PRIO
#define FLVIS_CHAIN_PRIO 0
#endif
// folded joins (KJoin, track_kernels.hpp).  kj_wait: thread 0 sleeps until the words have reached their numbers (system-scope loads: never
// a cached line), the workgroup follows it through the barrier; gives up after ~4 s like k_wait_flag.  kj_signal: behind the workgroup's
// last store -- barrier, one agent-scope release by thread 0, one arrival; the last arrival stores the number (release) and resets the counter.
template <typename KJ>
__device__ __forceinline__ void kj_wait(const KJ& kj) {
  if (!kj.wait[0] && !kj.wait[1]) return;
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    const unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (!kj.wait[k]) continue;
      while (__hip_atomic_load(kj.wait[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < kj.wait_seq[k]) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 400000000ull) {
          if (kj.err) __hip_atomic_store(kj.err, kj.wait_seq[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
template <typename KJ>
__device__ __forceinline__ void kj_post_wait(const KJ& kj) {
  if (!kj.post) return;
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(kj.post, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < kj.post_seq) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 400000000ull) {
        if (kj.err) __hip_atomic_store(kj.err, kj.post_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}
template <typename KJ>
__device__ __forceinline__ void kj_signal(const KJ& kj) {
  if (!kj.sig) return;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z;
    if (atomicAdd(kj.sig_cnt, 1u) == nb - 1u) {
      __hip_atomic_store(kj.sig_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the other workgroups' releases, before the word says they are done)
      __hip_atomic_store(kj.sig, kj.sig_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__device__ __forceinline__ void chain_priority() {
  if (FLVIS_CHAIN_PRIO > 0) __builtin_amdgcn_s_setprio(FLVIS_CHAIN_PRIO);
}

// wave64 reductions over all 64 lanes (result valid in every lane)
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// one landmark entry's share of a keyframe payload's checksum (FLVIS_KF_CHECK): position-weighted, summed in any order
__device__ __forceinline__ unsigned kf_word_hash(unsigned long long v) { return (unsigned)v * 2654435761u + (unsigned)(v >> 32) * 40503u; }
__device__ __forceinline__ unsigned kf_entry_hash(int k, long long id, const double* p2, const double* p3) {
  unsigned h = kf_word_hash((unsigned long long)id);
  h = h * 31u + kf_word_hash((unsigned long long)__double_as_longlong(p2[0]));
  h = h * 31u + kf_word_hash((unsigned long long)__double_as_longlong(p2[1]));
  h = h * 31u + kf_word_hash((unsigned long long)__double_as_longlong(p3[0]));
  h = h * 31u + kf_word_hash((unsigned long long)__double_as_longlong(p3[1]));
  h = h * 31u + kf_word_hash((unsigned long long)__double_as_longlong(p3[2]));
  return h * (unsigned)(k + 1);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}
// number of set bits in `mask` strictly below this lane
__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
  return __popcll(mask & ((1ull << lane_id()) - 1ull));
}

// Exclusive rank of `flag` among the threads of the workgroup (thread order) and the workgroup total.
// blockDim.x == 64 * NW; s_cnt: int[NW] in LDS.  Two barriers; every thread of the workgroup must call it.
template <int NW>
__device__ __forceinline__ int block_rank(bool flag, int* s_cnt, int& total) {
  const unsigned long long b = __ballot(flag);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_cnt[wv] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    const int cc = s_cnt[k];
    tot += cc;
    if (k < wv) off += cc;
  }
  __syncthreads();
  total = tot;
  return off + lane_prefix(b);
}

// sum over the 64 lanes of 32 values per lane, scattered: returns the total of value `idx` (idx as returned, < 32) in
// every lane; each exchange step halves the values a lane carries (63 shuffles instead of 32 x 6).  Fixed order.
template <int N, int O>
__device__ __forceinline__ void wave_rs_step(const double (&in)[2 * N], double (&out)[N], int lane, int& base) {
  const bool up = (lane & O) != 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double send = up ? in[i] : in[i + N];
    const double keep = up ? in[i + N] : in[i];
    out[i] = keep + __shfl_xor(send, O, 64);
  }
  base += up ? N : 0;
}
__device__ __forceinline__ double wave_reduce_scatter32(const double (&v)[32], int& idx) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  double a16[16], a8[8], a4[4], a2[2], a1[1];
  wave_rs_step<16, 32>(v, a16, lane, base);
  wave_rs_step<8, 16>(a16, a8, lane, base);
  wave_rs_step<4, 8>(a8, a4, lane, base);
  wave_rs_step<2, 4>(a4, a2, lane, base);
  wave_rs_step<1, 2>(a2, a1, lane, base);
  idx = base;
  return a1[0] + __shfl_xor(a1[0], 1, 64);
}

}  // namespace flvis
