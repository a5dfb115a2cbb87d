// flvis_amd: ORB extraction + Hamming matching for gfx950 -- the keyframe-rate front half of the
// reference's loop closing: cv::ORB::create(1000,1.2f,8,31,0,2,HARRIS_SCORE,31,20)->detectAndCompute
// (src/backend/vo_loopclosing.cpp:242-243) and BFMatcher knn x2 + mutual/ratio test (:603-639).  The arithmetic is the
// OpenCV 3.2/3.3 line's (features2d orb/fast/fast_score, imgproc resize INTER_LINEAR + 8-bit fixed-point separable Gaussian,
// core fastAtan2), integer wherever OpenCV's is, so the stages are bit-exact against the CPU restatement.
//
// Launch structure for a batch of n_img images (all byte/integer work, latency- and L2-bound, no GEMM shape anywhere):
//   k_orb_resize        level l from level l-1 (chained like OpenCV's pyramid), one thread per output pixel
//   k_fast_score        all levels in one launch, 64x16 pixel tiles staged in LDS, 4 pixels per thread
//   k_fast_nms          strict 3x3 maximum + image-border filter -> sparse score map, per-(image,level) histogram
//   k_gauss7            all levels, 64x16 tiles, horizontal pass into LDS, vertical pass out
//   k_orb_select        one 1024-thread workgroup per (image, level): retainBest(2n) threshold from the histogram, ordered
//                       compaction of the survivors (wave-contiguous raster segments, no barriers inside the sweeps),
//                       Harris response, exact radix select of the n-th response in LDS, ordered emission
//   k_orb_describe      one wave per keypoint: intensity-centroid moments by wave reduction, fastAtan2, 4 BRIEF tests per
//                       lane, 4 ballots -> 32 descriptor bytes
//   k_hamming_knn2      one query per thread, train descriptors broadcast from LDS; k_orb_match_filter: ordered compaction
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/flvis_hip.h"
#include "ctx.hpp"
#include "dev_common.hpp"

namespace flvis {

constexpr int ORB_MAX_LEVELS = 12;
constexpr int ORB_EDGE = 31;        // edgeThreshold
constexpr int ORB_HALF = 15;        // patchSize / 2
constexpr int ORB_CAND_CAP = 8192;  // per (image, level): survivors of the FAST-score cut held in LDS
constexpr int SEL_T = 1024;

struct OrbLevels {
  int n;
  int w[ORB_MAX_LEVELS], h[ORB_MAX_LEVELS], pitch[ORB_MAX_LEVELS];
  size_t off[ORB_MAX_LEVELS];  // byte offset of the level inside one image's block of a "full" map (level 0 included)
  size_t stride;               // bytes per image in a full map
  float scale[ORB_MAX_LEVELS];
  int nfeat[ORB_MAX_LEVELS];
  int tile0[ORB_MAX_LEVELS + 1];  // first 64x16 tile of each level in the flattened tile index
  int umax[ORB_HALF + 2];
};

// level l of image `img`: level 0 is the caller's image (pitch w), levels >= 1 live in the pyramid map
__device__ __forceinline__ const uint8_t* lvl_ptr(const OrbLevels& L, const uint8_t* img0, const uint8_t* pyr, int l, int img,
                                                  int& pitch) {
  if (l == 0) {
    pitch = L.w[0];
    return img0 + (size_t)img * L.w[0] * L.h[0];
  }
  pitch = L.pitch[l];
  return pyr + (size_t)img * L.stride + L.off[l];
}

__device__ __forceinline__ int tile_level(const OrbLevels& L, int tile) {
  int l = 0;
  while (l + 1 < L.n && tile >= L.tile0[l + 1]) l++;
  return l;
}

// ------------------------------------------------------------------------------------------------------ resize
// cv::resize INTER_LINEAR, CV_8UC1: 11-bit coefficients, ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2
__global__ void __launch_bounds__(256) k_orb_resize(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                                    size_t sstride, uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                    size_t dstride, double scale_x, double scale_y) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= dw || dy >= dh) return;
  const uint8_t* S = src + (size_t)blockIdx.z * sstride;
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) fx = 0.f, sx = 0;
  const bool two = sx + 1 < sw;
  if (!two) fx = 0.f, sx = sw - 1;
  const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy);
  fy -= (float)sy;
  const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
  const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
  const uint8_t* R0 = S + (size_t)y0 * spitch;
  const uint8_t* R1 = S + (size_t)y1 * spitch;
  int r0, r1;
  if (two) {
    r0 = R0[sx] * a0 + R0[sx + 1] * a1;
    r1 = R1[sx] * a0 + R1[sx + 1] * a1;
  } else {
    r0 = R0[sx] * 2048;
    r1 = R1[sx] * 2048;
  }
  const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
  dst[(size_t)blockIdx.z * dstride + (size_t)dy * dpitch + dx] = (uint8_t)v;
}

// ------------------------------------------------------------------------------------------------------ FAST-9/16
#ifndef FLVIS_ORB_TILE_H
#define FLVIS_ORB_TILE_H 32
#endif
constexpr int FT_W = 64, FT_H = FLVIS_ORB_TILE_H, FT_P = 72;  // LDS tile: (64+6) x (FT_H+6), pitch 72
constexpr int FT_R = FT_H / 4;                                  // rows per thread (256 threads = 4 rows of 64)
static_assert(FT_H % 4 == 0 && FT_H >= 4 && FT_H <= 64, "tile height");

// cooperative load of the (FT_W+6) x (FT_H+6) neighbourhood of a tile into LDS: all global loads of a thread are issued
// before the first LDS store (a load -> store loop serialises on the memory latency, ~1 us per trip under load)
constexpr int TL_TOTAL = (FT_H + 6) * (FT_W + 6);
constexpr int TL_N = (TL_TOTAL + 255) / 256;
template <bool REFLECT>
__device__ __forceinline__ void load_tile_u8(const uint8_t* __restrict__ src, int pitch, int W, int H, int tx0, int ty0,
                                             uint8_t* tile) {
  uint8_t v[TL_N];
#pragma unroll
  for (int k = 0; k < TL_N; k++) {
    const int i = threadIdx.x + k * 256;
    v[k] = 0;
    if (i < TL_TOTAL) {
      const int yy = i / (FT_W + 6), xx = i - yy * (FT_W + 6);
      const int gx = REFLECT ? reflect101c(tx0 + xx - 3, W) : min(max(tx0 + xx - 3, 0), W - 1);
      const int gy = REFLECT ? reflect101c(ty0 + yy - 3, H) : min(max(ty0 + yy - 3, 0), H - 1);
      v[k] = src[(size_t)gy * pitch + gx];
    }
  }
#pragma unroll
  for (int k = 0; k < TL_N; k++) {
    const int i = threadIdx.x + k * 256;
    if (i < TL_TOTAL) {
      const int yy = i / (FT_W + 6), xx = i - yy * (FT_W + 6);
      tile[yy * FT_P + xx] = v[k];
    }
  }
}

__device__ __forceinline__ int fast_score_px(const uint8_t* t /*LDS, centre*/, int thr) {
  // circle of radius 3, clockwise from (0,3) like fast.cpp's makeOffsets; the result does not depend on the start
  const int v = t[0];
  int d[16];
  d[0] = v - t[3 * FT_P + 0];
  d[1] = v - t[3 * FT_P + 1];
  d[2] = v - t[2 * FT_P + 2];
  d[3] = v - t[1 * FT_P + 3];
  d[4] = v - t[3];
  d[5] = v - t[-1 * FT_P + 3];
  d[6] = v - t[-2 * FT_P + 2];
  d[7] = v - t[-3 * FT_P + 1];
  d[8] = v - t[-3 * FT_P + 0];
  d[9] = v - t[-3 * FT_P - 1];
  d[10] = v - t[-2 * FT_P - 2];
  d[11] = v - t[-1 * FT_P - 3];
  d[12] = v - t[-3];
  d[13] = v - t[1 * FT_P - 3];
  d[14] = v - t[2 * FT_P - 2];
  d[15] = v - t[3 * FT_P - 1];
  unsigned dark = 0, bright = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    dark |= (d[k] > thr ? 1u : 0u) << k;
    bright |= (d[k] < -thr ? 1u : 0u) << k;
  }
  // 9 contiguous set bits on the 16-bit circle
  unsigned m = dark | (dark << 16);
  unsigned rd = m & (m >> 1);
  rd &= rd >> 2;
  rd &= rd >> 4;
  rd &= m >> 8;
  m = bright | (bright << 16);
  unsigned rb = m & (m >> 1);
  rb &= rb >> 2;
  rb &= rb >> 4;
  rb &= m >> 8;
  rd &= 0xFFFFu;
  rb &= 0xFFFFu;
  if (!(rd | rb)) return 0;
  // score = max over the 16 arcs of the arc minimum of |difference| (in the polarity that made it a corner) - 1
  // (fast_score.cpp cornerScore<16>: the other polarity cannot exceed the threshold the scan starts from)
  int e[16];
#pragma unroll
  for (int k = 0; k < 16; k++) e[k] = rd ? d[k] : -d[k];
  int m2[16], m4[16], m8[16];
#pragma unroll
  for (int k = 0; k < 16; k++) m2[k] = min(e[k], e[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; k++) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
  for (int k = 0; k < 16; k++) m8[k] = min(m4[k], m4[(k + 4) & 15]);
  int best = thr;
#pragma unroll
  for (int k = 0; k < 16; k++) best = max(best, min(m8[k], e[(k + 8) & 15]));
  return best - 1;
}

// src: level images (level 0 = caller's image); score: full map.  grid.x = flattened tiles, grid.y = image
__global__ void __launch_bounds__(256) k_fast_score(OrbLevels L, const uint8_t* __restrict__ img0,
                                                    const uint8_t* __restrict__ pyr, uint8_t* __restrict__ score, int thr) {
  __shared__ uint8_t tile[(FT_H + 6) * FT_P];
  const int l = tile_level(L, blockIdx.x);
  const int W = L.w[l], H = L.h[l];
  const int tpr = (W + FT_W - 1) / FT_W;
  const int t = blockIdx.x - L.tile0[l];
  const int tx0 = (t % tpr) * FT_W, ty0 = (t / tpr) * FT_H;
  int pitch;
  const uint8_t* src = lvl_ptr(L, img0, pyr, l, blockIdx.y, pitch);
  load_tile_u8<false>(src, pitch, W, H, tx0, ty0, tile);
  __syncthreads();
  uint8_t* out = score + (size_t)blockIdx.y * L.stride + L.off[l];
  const int lx = threadIdx.x & 63;
#pragma unroll 2
  for (int r = 0; r < FT_R; r++) {
    const int ly = (threadIdx.x >> 6) * FT_R + r;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) continue;
    int s = 0;
    if (x >= 3 && x < W - 3 && y >= 3 && y < H - 3) s = fast_score_px(tile + (ly + 3) * FT_P + lx + 3, thr);
    out[(size_t)y * L.pitch[l] + x] = (uint8_t)s;
  }
}

// single-image variant at the cv::FAST call shape (tightly packed in and out)
__global__ void __launch_bounds__(256) k_fast_score_plain(const uint8_t* __restrict__ img, int W, int H,
                                                          uint8_t* __restrict__ score, int thr) {
  __shared__ uint8_t tile[(FT_H + 6) * FT_P];
  const int tx0 = blockIdx.x * FT_W, ty0 = blockIdx.y * FT_H;
  const uint8_t* src = img + (size_t)blockIdx.z * W * H;
  load_tile_u8<false>(src, W, W, H, tx0, ty0, tile);
  __syncthreads();
  uint8_t* out = score + (size_t)blockIdx.z * W * H;
  const int lx = threadIdx.x & 63;
#pragma unroll 2
  for (int r = 0; r < FT_R; r++) {
    const int ly = (threadIdx.x >> 6) * FT_R + r;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) continue;
    int s = 0;
    if (x >= 3 && x < W - 3 && y >= 3 && y < H - 3) s = fast_score_px(tile + (ly + 3) * FT_P + lx + 3, thr);
    out[(size_t)y * W + x] = (uint8_t)s;
  }
}

// strict 3x3 maximum (fast.cpp non-max suppression) + KeyPointsFilter::runByImageBorder(edgeThreshold) -> sparse score
// map `nms` of the border box only (row y-31, column x-31, pitch box_pitch = BW rounded up to 16, pad columns zero; 0 = not a
// keypoint) and the per-(image, level) histogram of the surviving scores
__device__ __forceinline__ int box_pitch(int W) { return (W - 2 * ORB_EDGE + 15) & ~15; }

__global__ void __launch_bounds__(256) k_fast_nms(OrbLevels L, const uint8_t* __restrict__ score, uint8_t* __restrict__ nms,
                                                  unsigned* __restrict__ hist /*[img][level][256]*/) {
  __shared__ unsigned sh[256];
  const int l = tile_level(L, blockIdx.x);
  const int W = L.w[l], H = L.h[l], P = L.pitch[l];
  if (W <= 2 * ORB_EDGE || H <= 2 * ORB_EDGE) return;  // uniform: the level has no border box
  const int BP = box_pitch(W);
  const int tpr = (W + FT_W - 1) / FT_W;
  const int t = blockIdx.x - L.tile0[l];
  const int tx0 = (t % tpr) * FT_W, ty0 = (t / tpr) * FT_H;
  const uint8_t* sc = score + (size_t)blockIdx.y * L.stride + L.off[l];
  uint8_t* out = nms + (size_t)blockIdx.y * L.stride + L.off[l];
  sh[threadIdx.x] = 0;
  __syncthreads();
  const int lx = threadIdx.x & 63;
#pragma unroll 4
  for (int r = 0; r < FT_R; r++) {
    const int x = tx0 + lx, y = ty0 + (threadIdx.x >> 6) * FT_R + r;
    if (x >= W || y >= H) continue;
    if (y < ORB_EDGE || y >= H - ORB_EDGE || x < ORB_EDGE || x - ORB_EDGE >= BP) continue;
    int keep = 0;
    if (x < W - ORB_EDGE) {
      const uint8_t* c = sc + (size_t)y * P + x;
      const int s = c[0];
      if (s > 0 && s > c[-1] && s > c[1] && s > c[-P - 1] && s > c[-P] && s > c[-P + 1] && s > c[P - 1] && s > c[P] &&
          s > c[P + 1])
        keep = s;
    }
    out[(size_t)(y - ORB_EDGE) * BP + (x - ORB_EDGE)] = (uint8_t)keep;
    if (keep) atomicAdd(&sh[keep], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[((size_t)blockIdx.y * ORB_MAX_LEVELS + l) * 256 + threadIdx.x], sh[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------------ Gaussian 7x7
struct GaussK {
  int k[7];
};

__device__ __forceinline__ void gauss_tile(const uint8_t* __restrict__ src, int W, int H, int spitch,
                                           uint8_t* __restrict__ dst, int dpitch, int tx0, int ty0, const GaussK& g,
                                           uint8_t* tile /*[22][72]*/, int* tmp /*[22][64]*/) {
  load_tile_u8<true>(src, spitch, W, H, tx0, ty0, tile);
  __syncthreads();
  for (int i = threadIdx.x; i < (FT_H + 6) * FT_W; i += 256) {
    const int yy = i >> 6, xx = i & 63;
    const uint8_t* p = tile + yy * FT_P + xx;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) s += __mul24(g.k[j], (int)p[j]);  // 24-bit multiplies are full rate, 32-bit ones are not
    tmp[yy * FT_W + xx] = s;
  }
  __syncthreads();
  const int lx = threadIdx.x & 63;
#pragma unroll 2
  for (int r = 0; r < FT_R; r++) {
    const int ly = (threadIdx.x >> 6) * FT_R + r;
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) continue;
    int s = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) s += __mul24(g.k[j], tmp[(ly + j) * FT_W + lx]);  // |tmp| <= 255 * 257 < 2^23
    const int v = (s + (1 << 15)) >> 16;
    dst[(size_t)y * dpitch + x] = (uint8_t)min(255, max(0, v));
  }
}

__global__ void __launch_bounds__(256) k_gauss7(OrbLevels L, const uint8_t* __restrict__ img0, const uint8_t* __restrict__ pyr,
                                                uint8_t* __restrict__ blur, GaussK g) {
  __shared__ uint8_t tile[(FT_H + 6) * FT_P];
  __shared__ int tmp[(FT_H + 6) * FT_W];
  const int l = tile_level(L, blockIdx.x);
  const int W = L.w[l], H = L.h[l];
  const int tpr = (W + FT_W - 1) / FT_W;
  const int t = blockIdx.x - L.tile0[l];
  int pitch;
  const uint8_t* src = lvl_ptr(L, img0, pyr, l, blockIdx.y, pitch);
  gauss_tile(src, W, H, pitch, blur + (size_t)blockIdx.y * L.stride + L.off[l], L.pitch[l], (t % tpr) * FT_W,
             (t / tpr) * FT_H, g, tile, tmp);
}

__global__ void __launch_bounds__(256) k_gauss7_plain(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W,
                                                      int H, GaussK g) {
  __shared__ uint8_t tile[(FT_H + 6) * FT_P];
  __shared__ int tmp[(FT_H + 6) * FT_W];
  gauss_tile(src + (size_t)blockIdx.z * W * H, W, H, W, dst + (size_t)blockIdx.z * W * H, W, blockIdx.x * FT_W,
             blockIdx.y * FT_H, g, tile, tmp);
}

// ------------------------------------------------------------------------------------------------------ selection
// orb.cpp HarrisResponses (blockSize 7, k = 0.04), float evaluation order as written there
__device__ __forceinline__ float harris7(const uint8_t* __restrict__ img, int P, int x0, int y0) {
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < 7; i++) {
    const uint8_t* p = img + (size_t)(y0 - 3 + i) * P + (x0 - 3);
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const int Ix = ((int)p[j + 1] - (int)p[j - 1]) * 2 + ((int)p[j - P + 1] - (int)p[j - P - 1]) +
                     ((int)p[j + P + 1] - (int)p[j + P - 1]);
      const int Iy = ((int)p[j + P] - (int)p[j - P]) * 2 + ((int)p[j + P - 1] - (int)p[j - P - 1]) +
                     ((int)p[j + P + 1] - (int)p[j - P + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  }
  const float scale = 1.f / ((1 << 2) * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

struct LevelKp {
  int x, y;
  float resp;
};

// one workgroup per (level = blockIdx.x, image = blockIdx.y); dynamic LDS: pos[CAP] u32, key[CAP] u32
__global__ void __launch_bounds__(SEL_T) k_orb_select(OrbLevels L, const uint8_t* __restrict__ img0,
                                                      const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ nms,
                                                      const unsigned* __restrict__ hist, LevelKp* __restrict__ lvl_kp,
                                                      int* __restrict__ lvl_n, int lvl_cap, int* __restrict__ overflow) {
  extern __shared__ unsigned sel_smem[];
  unsigned* pos = sel_smem;
  unsigned* key = sel_smem + ORB_CAND_CAP;
  __shared__ unsigned s_hist[256];
  __shared__ int s_wcnt[SEL_T / 64];
  __shared__ int s_cnt[SEL_T / 64];
  __shared__ unsigned s_u[4];
  const int l = blockIdx.x, img = blockIdx.y;
  const int W = L.w[l], H = L.h[l], P = L.pitch[l];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  int* out_n = lvl_n + (size_t)img * ORB_MAX_LEVELS + l;
  const int BW = W - 2 * ORB_EDGE, BH = H - 2 * ORB_EDGE;
  const int nfeat = L.nfeat[l];
  if (BW <= 0 || BH <= 0 || nfeat <= 0) {
    if (tid == 0) *out_n = 0;
    return;
  }
  // retainBest(2 * nfeat) on the FAST score: T1 = the (2n)-th largest score, everything >= T1 stays
  if (tid < 256) s_hist[tid] = hist[((size_t)img * ORB_MAX_LEVELS + l) * 256 + tid];
  __syncthreads();
  if (tid == 0) {
    unsigned tot = 0;
    for (int i = 0; i < 256; i++) tot += s_hist[i];
    unsigned T1 = 1;
    if (tot > (unsigned)(2 * nfeat)) {
      unsigned cum = 0;
      for (int v = 255; v >= 1; v--) {
        cum += s_hist[v];
        if (cum >= (unsigned)(2 * nfeat)) {
          T1 = v;
          break;
        }
      }
    }
    s_u[0] = T1;
  }
  __syncthreads();
  const int T1 = (int)s_u[0];
  const uint8_t* nm = nms + (size_t)img * L.stride + L.off[l];
  // raster sweep over the border-box map in 16-pixel chunks (one aligned 16-byte load per lane), one contiguous run of
  // chunks per wave: sweep A counts, sweep B emits the survivors' positions in raster order
  const int BP = box_pitch(W), CPR = BP >> 4;
  const int NC = CPR * BH;
  const int seg = ((NC + SEL_T / 64 - 1) / (SEL_T / 64) + 63) / 64 * 64;
  const int s0 = wv * seg, s1 = min(NC, s0 + seg);
  const int iters = s1 > s0 ? (s1 - s0 + 63) / 64 : 0;  // uniform per wave
  const unsigned t1 = (unsigned)T1;
  auto chunk_count = [&](const uint4& q) {
    int c = 0;
    const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int b = 0; b < 4; b++) c += ((w4[k] >> (8 * b)) & 255u) >= t1 ? 1 : 0;
    return c;
  };
  int cnt = 0;
  for (int it = 0; it < iters; it++) {
    const int i = s0 + it * 64 + ln;
    int c = 0;
    if (i < s1) c = chunk_count(*reinterpret_cast<const uint4*>(nm + (size_t)i * 16));
    cnt += wave_sum_i32(c);
  }
  if (ln == 0) s_wcnt[wv] = cnt;
  __syncthreads();
  int base = 0, total = 0;
  for (int k = 0; k < SEL_T / 64; k++) {
    const int c = s_wcnt[k];
    if (k < wv) base += c;
    total += c;
  }
  for (int it = 0; it < iters; it++) {
    const int i = s0 + it * 64 + ln;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (i < s1) q = *reinterpret_cast<const uint4*>(nm + (size_t)i * 16);
    const int c = i < s1 ? chunk_count(q) : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (ln >= o) incl += t;
    }
    int o = base + incl - c;
    if (c) {
      const int y = i / CPR, x0 = (i - y * CPR) * 16;
      const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (((w4[k] >> (8 * b)) & 255u) >= t1) {
            if (o < ORB_CAND_CAP) pos[o] = ((unsigned)(y + ORB_EDGE) << 16) | (unsigned)(x0 + 4 * k + b + ORB_EDGE);
            o++;
          }
    }
    base += __shfl(incl, 63, 64);
  }
  if (tid == 0 && total > ORB_CAND_CAP) overflow[img] = 1;
  const int m = min(total, ORB_CAND_CAP);
  __syncthreads();
  {  // Harris response of every survivor, one per thread
    int plv;
    const uint8_t* im = lvl_ptr(L, img0, pyr, l, img, plv);
    for (int i = tid; i < m; i += SEL_T) {
      float r = harris7(im, plv, (int)(pos[i] & 0xFFFFu), (int)(pos[i] >> 16));
      r = r + 0.0f;  // -0 -> +0 so that the ordered key compares like the float
      key[i] = f32_ordered(r);
    }
  }
  __syncthreads();
  // retainBest(nfeat) on the Harris response: exact n-th largest key by MSB-first radix select
  unsigned thr_key = 0;
  if (m > nfeat) {
    unsigned prefix = 0, mask = 0;
    int remaining = nfeat;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < m; i += SEL_T) {
        const unsigned k = key[i];
        if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int cum = 0, bsel = 0;
        for (int v = 255; v >= 0; v--) {
          const int hv = (int)s_hist[v];
          if (cum + hv >= remaining) {
            bsel = v;
            break;
          }
          cum += hv;
        }
        s_u[1] = (unsigned)bsel;
        s_u[2] = (unsigned)(remaining - cum);
      }
      __syncthreads();
      prefix |= s_u[1] << shift;
      mask |= 255u << shift;
      remaining = (int)s_u[2];
      __syncthreads();
    }
    thr_key = prefix;
  }
  LevelKp* out = lvl_kp + ((size_t)img * ORB_MAX_LEVELS + l) * lvl_cap;
  int written = 0;
  for (int i0 = 0; i0 < m; i0 += SEL_T) {
    const int i = i0 + tid;
    const bool p = i < m && key[i] >= thr_key;
    int tot;
    const int o = written + block_rank<SEL_T / 64>(p, s_cnt, tot);
    if (p && o < lvl_cap) {
      LevelKp k;
      k.x = (int)(pos[i] & 0xFFFFu);
      k.y = (int)(pos[i] >> 16);
      k.resp = f32_unordered(key[i]);
      out[o] = k;
    }
    written += tot;
  }
  if (tid == 0) {
    if (written > lvl_cap) overflow[img] = 1;
    *out_n = min(written, lvl_cap);
  }
}

// ------------------------------------------------------------------------------------------------------ describe
// cv::fastAtan2 (degrees), 3.x polynomial, float operation order as written there
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float k = (float)(180 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
              p7 = -0.04432655554792128f * k;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// one wave per keypoint; grid.x = ceil(max keypoints / 4), grid.y = image
__global__ void __launch_bounds__(256) k_orb_describe(OrbLevels L, const uint8_t* __restrict__ img0,
                                                      const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                      const LevelKp* __restrict__ lvl_kp, const int* __restrict__ lvl_n,
                                                      int lvl_cap, const int8_t* __restrict__ pattern /*[256][4]*/,
                                                      float* __restrict__ kps, uint8_t* __restrict__ desc,
                                                      int* __restrict__ count, int cap, int* __restrict__ overflow) {
  const int img = blockIdx.y;
  const int ln = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);  // index into the level-major concatenation
  int l = 0, first = 0, total = 0;
  bool found = false;
  for (int q = 0; q < L.n; q++) {
    const int c = lvl_n[(size_t)img * ORB_MAX_LEVELS + q];
    if (!found && k < total + c) {
      l = q;
      first = total;
      found = true;
    }
    total += c;
  }
  if (k == 0 && ln == 0) {
    count[img] = min(total, cap);
    if (total > cap) overflow[img] = 1;
  }
  if (!found || k >= cap) return;
  const LevelKp kp = lvl_kp[((size_t)img * ORB_MAX_LEVELS + l) * lvl_cap + (k - first)];
  int P;
  const uint8_t* im = lvl_ptr(L, img0, pyr, l, img, P);
  const uint8_t* c = im + (size_t)kp.y * P + kp.x;
  // intensity centroid over the disc of radius 15 (orb.cpp ICAngles): two rows of up to 31 pixels per step
  int m10 = 0, m01 = 0;
  {
    const int u = (ln & 31) - ORB_HALF;
    for (int v0 = -ORB_HALF; v0 <= ORB_HALF; v0 += 2) {
      const int v = v0 + (ln >> 5);
      if (v <= ORB_HALF && (ln & 31) < 31) {
        const int d = L.umax[v < 0 ? -v : v];
        if (u >= -d && u <= d) {
          const int val = c[v * P + u];
          m10 += u * val;
          m01 += v * val;
        }
      }
    }
    m10 = wave_sum_i32(m10);
    m01 = wave_sum_i32(m01);
  }
  const float angle_deg = fast_atan2_deg((float)m01, (float)m10);
  const float sf = L.scale[l];
  float px = (float)kp.x, py = (float)kp.y;
  if (l != 0) {
    px *= sf;
    py *= sf;
  }
  // orb.cpp computeOrbDescriptors: centre and steering exactly as written there
  const float scale = 1.f / sf;
  float angle = angle_deg;
  angle *= (float)(3.1415926535897932384626433832795 / 180.f);
  const float a = (float)cos((double)angle), b = (float)sin((double)angle);
  const int cx = __float2int_rn(px * scale), cy = __float2int_rn(py * scale);
  const uint8_t* bc = blur + (size_t)img * L.stride + L.off[l] + (size_t)cy * L.pitch[l] + cx;
  const int BP = L.pitch[l];
  unsigned long long bits[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int t = r * 64 + ln;
    const char4 q = reinterpret_cast<const char4*>(pattern)[t];
    const float x0 = (float)q.x * a - (float)q.y * b, y0 = (float)q.x * b + (float)q.y * a;
    const float x1 = (float)q.z * a - (float)q.w * b, y1 = (float)q.z * b + (float)q.w * a;
    const int t0 = bc[__float2int_rn(y0) * BP + __float2int_rn(x0)];
    const int t1 = bc[__float2int_rn(y1) * BP + __float2int_rn(x1)];
    bits[r] = __ballot(t0 < t1);
  }
  if (ln < 4) reinterpret_cast<unsigned long long*>(desc + ((size_t)img * cap + k) * 32)[ln] =
      ln == 0 ? bits[0] : (ln == 1 ? bits[1] : (ln == 2 ? bits[2] : bits[3]));
  if (ln == 0) {
    float* o = kps + ((size_t)img * cap + k) * 6;
    o[0] = px;
    o[1] = py;
    o[2] = 31.f * sf;
    o[3] = angle_deg;
    o[4] = kp.resp;
    o[5] = (float)l;
  }
}

// ------------------------------------------------------------------------------------------------------ matching
// BFMatcher(NORM_HAMMING).knnMatch(k = 2): thread = query, train set streamed through LDS in tiles of 256 descriptors
__global__ void __launch_bounds__(256) k_hamming_knn2(const uint8_t* __restrict__ q, const int* __restrict__ nq, int qcap,
                                                      const uint8_t* __restrict__ t, const int* __restrict__ nt, int tcap,
                                                      int* __restrict__ idx, int* __restrict__ dist) {
  __shared__ unsigned long long tl[256 * 4];
  const int pair = blockIdx.y;
  const int NQ = min(nq[pair], qcap), NT = min(nt[pair], tcap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= NQ) return;  // uniform
  const unsigned long long* Q = reinterpret_cast<const unsigned long long*>(q + ((size_t)pair * qcap) * 32);
  const unsigned long long* T = reinterpret_cast<const unsigned long long*>(t + ((size_t)pair * tcap) * 32);
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  if (i < NQ) {
    q0 = Q[(size_t)i * 4];
    q1 = Q[(size_t)i * 4 + 1];
    q2 = Q[(size_t)i * 4 + 2];
    q3 = Q[(size_t)i * 4 + 3];
  }
  int b0 = -1, b1 = -1, d0 = INT_MAX, d1 = INT_MAX;
  for (int j0 = 0; j0 < NT; j0 += 256) {
    const int nj = min(256, NT - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < nj * 4; e += 256) tl[e] = T[(size_t)j0 * 4 + e];
    __syncthreads();
    for (int j = 0; j < nj; j++) {
      const int d = __popcll(q0 ^ tl[j * 4]) + __popcll(q1 ^ tl[j * 4 + 1]) + __popcll(q2 ^ tl[j * 4 + 2]) +
                    __popcll(q3 ^ tl[j * 4 + 3]);
      if (d < d1) {
        if (d < d0) {
          d1 = d0;
          b1 = b0;
          d0 = d;
          b0 = j0 + j;
        } else {
          d1 = d;
          b1 = j0 + j;
        }
      }
    }
  }
  if (i < NQ) {
    int* oi = idx + ((size_t)pair * qcap + i) * 2;
    int* od = dist + ((size_t)pair * qcap + i) * 2;
    oi[0] = b0;
    oi[1] = b1;
    od[0] = d0;
    od[1] = d1;
  }
}

// vo_loopclosing.cpp:621-639: i survives when the best match of its best match is i and d0/d1 < ratio; ordered by i
__global__ void __launch_bounds__(256) k_orb_match_filter(const int* __restrict__ na, int acap, const int* __restrict__ nb,
                                                          int bcap, const int* __restrict__ i12, const int* __restrict__ d12,
                                                          const int* __restrict__ i21, float ratio, int* __restrict__ pairs,
                                                          int* __restrict__ npairs) {
  __shared__ int s_cnt[4];
  const int pair = blockIdx.x;
  const int NA = min(na[pair], acap), NB = min(nb[pair], bcap);
  int written = 0;
  if (NA >= 2 && NB >= 2) {
    for (int i0 = 0; i0 < NA; i0 += 256) {
      const int i = i0 + threadIdx.x;
      bool p = false;
      int t = -1;
      if (i < NA) {
        t = i12[((size_t)pair * acap + i) * 2];
        if (t >= 0 && i21[((size_t)pair * bcap + t) * 2] == i) {
          const float f0 = (float)d12[((size_t)pair * acap + i) * 2], f1 = (float)d12[((size_t)pair * acap + i) * 2 + 1];
          p = (double)f0 * 1.0 / (double)f1 < (double)ratio;
        }
      }
      int tot;
      const int o = written + block_rank<4>(p, s_cnt, tot);
      if (p) {
        pairs[((size_t)pair * acap + o) * 2] = i;
        pairs[((size_t)pair * acap + o) * 2 + 1] = t;
      }
      written += tot;
    }
  }
  if (threadIdx.x == 0) npairs[pair] = written;
}

// ------------------------------------------------------------------------------------------------------ host side
static inline int h_round(double v) { return (int)std::lrint(v); }

static void default_pattern(int8_t* pat) {  // OpenCV's makeRandomPattern(31, pattern, 512): cv::RNG(0x34985739)
  uint64_t state = 0x34985739;
  auto next = [&]() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  };
  for (int i = 0; i < 1024; i++) pat[i] = (int8_t)((int)(next() % 31u) - 15);
}

static GaussK gauss_kernel() {  // getGaussianKernel(7, 2, CV_32F) scaled by 2^8 and rounded (filter.cpp, 8-bit fixed point)
  GaussK g;
  float cf[7];
  double sum = 0, s2 = -0.5 / (2.0 * 2.0);
  for (int i = 0; i < 7; i++) {
    const double x = i - 3.0;
    cf[i] = (float)std::exp(s2 * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) {
    cf[i] = (float)(cf[i] * sum);
    g.k[i] = h_round((double)cf[i] * 256.0);
  }
  return g;
}

static int make_levels(int w, int h, const flvis_orb_params& p, OrbLevels& L) {
  std::memset(&L, 0, sizeof L);
  L.n = p.nlevels;
  size_t off = 0;
  int tiles = 0;
  for (int l = 0; l < L.n; l++) {
    const float s = (float)std::pow((double)p.scale_factor, (double)l);
    L.scale[l] = s;
    L.w[l] = h_round(w / s);
    L.h[l] = h_round(h / s);
    if (L.w[l] < 8 || L.h[l] < 8) return -1;
    L.pitch[l] = l == 0 ? w : align_up(L.w[l], 16);
    L.off[l] = off;
    off += (size_t)L.pitch[l] * L.h[l];
    off = (off + 255) / 256 * 256;
    L.tile0[l] = tiles;
    tiles += ((L.w[l] + FT_W - 1) / FT_W) * ((L.h[l] + FT_H - 1) / FT_H);
  }
  L.tile0[L.n] = tiles;
  L.stride = off;
  // orb.cpp computeKeyPoints: geometric split of nfeatures over the levels
  const float factor = (float)(1.0 / p.scale_factor);
  float ndesired = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L.n));
  int sum = 0;
  for (int l = 0; l < L.n - 1; l++) {
    L.nfeat[l] = h_round(ndesired);
    sum += L.nfeat[l];
    ndesired *= factor;
  }
  L.nfeat[L.n - 1] = std::max(p.nfeatures - sum, 0);
  // the disc of the intensity centroid
  const int hp = ORB_HALF;
  const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
  for (int v = 0; v <= vmax; ++v) L.umax[v] = h_round(std::sqrt((double)hp * hp - v * v));
  for (int v = hp, v0 = 0; v >= vmin; --v) {
    while (L.umax[v0] == L.umax[v0 + 1]) ++v0;
    L.umax[v] = v0;
    ++v0;
  }
  return tiles;
}

}  // namespace flvis

using namespace flvis;

#define CHECK_CTX(ctx) \
  if (!(ctx)) return FLVIS_ERR_INVALID_ARG;
#define CHECK_LAUNCH(ctx, what)                               \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return (ctx)->hip_fail(e__, what); \
  } while (0)

extern "C" {

int flvis_orb_default_pattern(int8_t* h_pattern512x2) {
  if (!h_pattern512x2) return FLVIS_ERR_INVALID_ARG;
  default_pattern(h_pattern512x2);
  return FLVIS_OK;
}

int flvis_hip_resize_linear(flvis_ctx* ctx, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh,
                            int n_img) {
  CHECK_CTX(ctx);
  if (!d_src || !d_dst || sw < 2 || sh < 2 || dw < 1 || dh < 1 || n_img <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "resize_linear: bad args");
  const double sx = 1. / ((double)dw / sw), sy = 1. / ((double)dh / sh);
  k_orb_resize<<<dim3((dw + 63) / 64, (dh + 3) / 4, n_img), 256, 0, ctx->stream>>>(d_src, sw, sh, sw, (size_t)sw * sh, d_dst,
                                                                                   dw, dh, dw, (size_t)dw * dh, sx, sy);
  CHECK_LAUNCH(ctx, "resize_linear");
  return FLVIS_OK;
}

int flvis_hip_fast_score(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img, int threshold, uint8_t* d_score) {
  CHECK_CTX(ctx);
  if (!d_img || !d_score || w < 7 || h < 7 || n_img <= 0 || threshold < 0 || threshold > 254)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "fast_score: bad args");
  k_fast_score_plain<<<dim3((w + FT_W - 1) / FT_W, (h + FT_H - 1) / FT_H, n_img), 256, 0, ctx->stream>>>(d_img, w, h, d_score,
                                                                                                       threshold);
  CHECK_LAUNCH(ctx, "fast_score");
  return FLVIS_OK;
}

int flvis_hip_gaussian_blur7(flvis_ctx* ctx, const uint8_t* d_src, uint8_t* d_dst, int w, int h, int n_img) {
  CHECK_CTX(ctx);
  if (!d_src || !d_dst || d_src == d_dst || w < 4 || h < 4 || n_img <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "gaussian_blur7: bad args (in-place is not supported)");
  k_gauss7_plain<<<dim3((w + FT_W - 1) / FT_W, (h + FT_H - 1) / FT_H, n_img), 256, 0, ctx->stream>>>(d_src, d_dst, w, h,
                                                                                                   gauss_kernel());
  CHECK_LAUNCH(ctx, "gaussian_blur7");
  return FLVIS_OK;
}

int flvis_hip_orb_detect_and_compute(flvis_ctx* ctx, const uint8_t* d_img, int w, int h, int n_img,
                                     const flvis_orb_params* prm, const int8_t* h_pattern, float* d_kps, uint8_t* d_desc,
                                     int* d_count, int cap, int* d_overflow) {
  CHECK_CTX(ctx);
  if (!d_img || !prm || !d_kps || !d_desc || !d_count || cap <= 0 || n_img <= 0 || w < 64 || h < 64 || w > 32767 ||
      h > 32767)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "orb: bad args");
  if (prm->nlevels < 1 || prm->nlevels > ORB_MAX_LEVELS || !(prm->scale_factor > 1.0f) || prm->nfeatures < 1 ||
      prm->fast_threshold < 1 || prm->fast_threshold > 254)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "orb: bad parameters");
  OrbLevels L;
  const int tiles = make_levels(w, h, *prm, L);
  if (tiles < 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "orb: image too small for the number of levels");
  int lvl_cap = 0;
  for (int l = 0; l < L.n; l++) lvl_cap = std::max(lvl_cap, L.nfeat[l]);
  if (2 * lvl_cap > ORB_CAND_CAP * 3 / 4) return ctx->fail(FLVIS_ERR_CAPACITY, "orb: nfeatures too large for the candidate buffer");
  lvl_cap = lvl_cap + lvl_cap / 4 + 64;  // room for ties at the cut
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)k_orb_select, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       ORB_CAND_CAP * 8);
    if (e != hipSuccess) return ctx->hip_fail(e, "orb: hipFuncSetAttribute");
    attr_done = true;
  }
  uint8_t* pyr = (uint8_t*)ctx->scratch("orb_pyr", L.stride * n_img);
  uint8_t* score = (uint8_t*)ctx->scratch("orb_score", L.stride * n_img);
  uint8_t* nms = (uint8_t*)ctx->scratch("orb_nms", L.stride * n_img);
  uint8_t* blur = (uint8_t*)ctx->scratch("orb_blur", L.stride * n_img);
  unsigned* hist = (unsigned*)ctx->scratch("orb_hist", sizeof(unsigned) * 256 * ORB_MAX_LEVELS * n_img);
  LevelKp* lvl_kp = (LevelKp*)ctx->scratch("orb_lvl_kp", sizeof(LevelKp) * (size_t)lvl_cap * ORB_MAX_LEVELS * n_img);
  int* lvl_n = (int*)ctx->scratch("orb_lvl_n", sizeof(int) * ORB_MAX_LEVELS * n_img);
  int* ovf = (int*)ctx->scratch("orb_ovf", sizeof(int) * n_img);
  int8_t* pat = (int8_t*)ctx->scratch("orb_pattern", 1024);
  if (!pyr || !score || !nms || !blur || !hist || !lvl_kp || !lvl_n || !ovf || !pat)
    return ctx->fail(FLVIS_ERR_HIP, "orb: scratch allocation failed");
  int8_t hp[1024];
  if (h_pattern)
    std::memcpy(hp, h_pattern, 1024);
  else
    default_pattern(hp);
  hipStream_t st = ctx->stream;
  if (ctx->orb_pattern.size() != 1024 || std::memcmp(ctx->orb_pattern.data(), hp, 1024) != 0) {
    // (re)upload only when the pattern changes: steady-state calls stay asynchronous.  The source lives in the context.
    hipStreamSynchronize(st);  // earlier launches may still read the buffer
    ctx->orb_pattern.assign(reinterpret_cast<const char*>(hp), 1024);
    if (hipMemcpyAsync(pat, ctx->orb_pattern.data(), 1024, hipMemcpyHostToDevice, st) != hipSuccess) {
      ctx->orb_pattern.clear();
      return ctx->fail(FLVIS_ERR_HIP, "orb: pattern upload");
    }
    hipStreamSynchronize(st);
  }
  int* ovf_out = d_overflow ? d_overflow : ovf;
  hipMemsetAsync(hist, 0, sizeof(unsigned) * 256 * ORB_MAX_LEVELS * n_img, st);
  hipMemsetAsync(ovf_out, 0, sizeof(int) * n_img, st);
  for (int l = 1; l < L.n; l++) {
    const uint8_t* src = l == 1 ? d_img : pyr + L.off[l - 1];
    const size_t sstride = l == 1 ? (size_t)w * h : L.stride;
    const double sx = 1. / ((double)L.w[l] / L.w[l - 1]), sy = 1. / ((double)L.h[l] / L.h[l - 1]);
    k_orb_resize<<<dim3((L.w[l] + 63) / 64, (L.h[l] + 3) / 4, n_img), 256, 0, st>>>(
        src, L.w[l - 1], L.h[l - 1], L.pitch[l - 1], sstride, pyr + L.off[l], L.w[l], L.h[l], L.pitch[l], L.stride, sx, sy);
  }
  k_fast_score<<<dim3(tiles, n_img), 256, 0, st>>>(L, d_img, pyr, score, prm->fast_threshold);
  k_fast_nms<<<dim3(tiles, n_img), 256, 0, st>>>(L, score, nms, hist);
  k_gauss7<<<dim3(tiles, n_img), 256, 0, st>>>(L, d_img, pyr, blur, gauss_kernel());
  k_orb_select<<<dim3(L.n, n_img), SEL_T, ORB_CAND_CAP * 8, st>>>(L, d_img, pyr, nms, hist, lvl_kp, lvl_n, lvl_cap, ovf_out);
  int max_total = 0;
  for (int l = 0; l < L.n; l++) max_total += lvl_cap;
  max_total = std::min(max_total, cap);
  k_orb_describe<<<dim3((max_total + 3) / 4, n_img), 256, 0, st>>>(L, d_img, pyr, blur, lvl_kp, lvl_n, lvl_cap, pat, d_kps,
                                                                   d_desc, d_count, cap, ovf_out);
  CHECK_LAUNCH(ctx, "orb_detect_and_compute");
  return FLVIS_OK;
}

int flvis_hip_hamming_knn2(flvis_ctx* ctx, const uint8_t* d_query, const int* d_nq, int qcap, const uint8_t* d_train,
                           const int* d_nt, int tcap, int n_pairs, int* d_idx, int* d_dist) {
  CHECK_CTX(ctx);
  if (!d_query || !d_nq || !d_train || !d_nt || !d_idx || !d_dist || qcap <= 0 || tcap <= 0 || n_pairs <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "hamming_knn2: bad args");
  k_hamming_knn2<<<dim3((qcap + 255) / 256, n_pairs), 256, 0, ctx->stream>>>(d_query, d_nq, qcap, d_train, d_nt, tcap, d_idx,
                                                                           d_dist);
  CHECK_LAUNCH(ctx, "hamming_knn2");
  return FLVIS_OK;
}

int flvis_hip_orb_match(flvis_ctx* ctx, const uint8_t* d_a, const int* d_na, int acap, const uint8_t* d_b, const int* d_nb,
                        int bcap, int n_pairs, double ratio_max, int* d_pairs, int* d_npairs) {
  CHECK_CTX(ctx);
  if (!d_a || !d_na || !d_b || !d_nb || !d_pairs || !d_npairs || acap <= 0 || bcap <= 0 || n_pairs <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "orb_match: bad args");
  int* i12 = (int*)ctx->scratch("orbm_i12", sizeof(int) * 2 * (size_t)acap * n_pairs);
  int* d12 = (int*)ctx->scratch("orbm_d12", sizeof(int) * 2 * (size_t)acap * n_pairs);
  int* i21 = (int*)ctx->scratch("orbm_i21", sizeof(int) * 2 * (size_t)bcap * n_pairs);
  int* d21 = (int*)ctx->scratch("orbm_d21", sizeof(int) * 2 * (size_t)bcap * n_pairs);
  if (!i12 || !d12 || !i21 || !d21) return ctx->fail(FLVIS_ERR_HIP, "orb_match: scratch allocation failed");
  hipStream_t st = ctx->stream;
  k_hamming_knn2<<<dim3((acap + 255) / 256, n_pairs), 256, 0, st>>>(d_a, d_na, acap, d_b, d_nb, bcap, i12, d12);
  k_hamming_knn2<<<dim3((bcap + 255) / 256, n_pairs), 256, 0, st>>>(d_b, d_nb, bcap, d_a, d_na, acap, i21, d21);
  k_orb_match_filter<<<n_pairs, 256, 0, st>>>(d_na, acap, d_nb, bcap, i12, d12, i21, (float)ratio_max, d_pairs, d_npairs);
  CHECK_LAUNCH(ctx, "orb_match");
  return FLVIS_OK;
}

}  // extern "C"
