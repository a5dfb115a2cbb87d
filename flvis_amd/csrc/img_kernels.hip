// flvis_amd: image-scan kernels of the FLVIS front-end for gfx950 (CDNA4), batched over S independent streams.
//
//   k_hist256 / k_equalize_lut / k_lut_apply   -> cv::equalizeHist        (reference: src/frontend/f2f_tracking.cpp:141-145)
//   k_pyr_down                                 -> cv::pyrDown levels of calcOpticalFlowPyrLK's pyramids
//                                                 (reference: src/processing/lkorb_tracking.cpp:64-73, camera_frame.cpp:124-128)
//   k_eig_cand                                 -> cornerMinEigenVal + minMaxLoc + the 3x3 local maxima in ONE pass (the
//                                                 quality threshold is applied by k_gftt_pick, see k_eig_cand)
//   k_bgr_to_gray / k_copy_image16             -> cv::cvtColor BGR(A)2GRAY (f2f_tracking.cpp:74-111) / plain ingest
//   k_gftt_pick                                -> ranked min-distance selection of cv::goodFeaturesToTrack (tiered top-K)
//                                                 (reference: src/processing/feature_dem.cpp:160,221)
//   k_feature_dem                              -> FeatureDEM::detect / ::redetect (feature_dem.cpp:92-266, quirks kept)
//
// Streaming passes over u8 images (O(1) flop/byte): tiles are staged in LDS with dword global loads, one workgroup per
// (tile, stream).  Measured, they are issue/latency bound rather than HBM bound (DESIGN.md sections 4, 8, 10).
// Built with -ffp-contract=off: float arithmetic is op-for-op the oracle's (tests compare bit-exactly).
#include <cstdlib>

#include "dev_common.hpp"
#include "dem_sort.hpp"
#include "eig_strip.hpp"
#include "img_kernels.hpp"

namespace flvis {

// ------------------------------------------------------------------------------------------------ tile loader
// Loads a (TH x TW) u8 tile whose top-left image coordinate is (x0,y0) (may be negative / beyond the image: REFLECT_101)
// into LDS `tile` with row stride TS.  x0 must be a multiple of 4 (dword path for interior dwords).
template <int TH, int TW, int TS>
__device__ __forceinline__ void load_tile_u8(const uint8_t* __restrict__ img, int w, int h, int pitch, int x0, int y0,
                                             uint8_t* tile, int bx = 0, int by = 0) {
  static_assert(TW % 4 == 0, "tile width must be a multiple of 4");
  constexpr int DW = TW / 4;
  // A tile that lies inside the image (block-uniform test; all but the edge tiles): nothing to reflect, and ALL of a thread's loads are
  // issued before the first LDS store -- one memory round trip for the tile instead of one per trip of the loop below (which waits
  // for its load before it issues the next).  256 threads assumed for the trip count (more trips than needed are predicated off).
  // (bx, by): physical REFLECT_101 border of the source image (a pyramid level whose border is complete): "inside" then includes it.
  if (x0 >= -bx && y0 >= -by && x0 + TW <= w + bx && y0 + TH <= h + by && blockDim.x == 256) {
    constexpr int N = TH * DW, T = (N + 255) / 256;
    const __attribute__((address_space(1))) uint8_t* const base =
        (const __attribute__((address_space(1))) uint8_t*)(img + (ptrdiff_t)y0 * pitch + x0);
    uint32_t v[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int i = (int)threadIdx.x + 256 * t;
      const int r = i / DW, c = i - r * DW;
      const bool on = 256 * (t + 1) <= N || i < N;
      v[t] = *(const __attribute__((address_space(1))) uint32_t*)(base + (on ? (size_t)r * pitch + 4 * c : 0));
    }
#pragma unroll
    for (int t = 0; t < T; t++) asm volatile("" : "+v"(v[t]));
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int i = (int)threadIdx.x + 256 * t;
      const int r = i / DW, c = i - r * DW;
      if (256 * (t + 1) <= N || i < N) *reinterpret_cast<uint32_t*>(tile + r * TS + 4 * c) = v[t];
    }
    return;
  }
  for (int i = threadIdx.x; i < TH * DW; i += blockDim.x) {
    int r = i / DW, c = i - r * DW;
    int gy = reflect101c(y0 + r, h);
    int gx = x0 + 4 * c;
    const uint8_t* row = img + (size_t)gy * pitch;
    uint32_t v;
    if (gx >= 0 && gx + 3 < w) {
      v = *reinterpret_cast<const uint32_t*>(row + gx);
    } else {
      uint32_t b0 = row[reflect101c(gx, w)], b1 = row[reflect101c(gx + 1, w)], b2 = row[reflect101c(gx + 2, w)],
               b3 = row[reflect101c(gx + 3, w)];
      v = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    *reinterpret_cast<uint32_t*>(tile + r * TS + 4 * c) = v;
  }
}

// ------------------------------------------------------------------------------------------------ equalizeHist
__global__ __launch_bounds__(256) void k_hist256(ImgSel src, int w, int h, int pitch, size_t sstride,
                                                 unsigned* __restrict__ hist, const int* __restrict__ active) {
  const int s = blockIdx.y;
  if (active && !active[s]) return;
  __shared__ unsigned lh[4][256];
  for (int i = threadIdx.x; i < 1024; i += 256) (&lh[0][0])[i] = 0;
  __syncthreads();
  const uint8_t* img = src.ptr(s, sstride);
  const int wv = threadIdx.x >> 6;
  const int dwords_per_row = w >> 2;  // w % 4 == 0 (checked on host)
  const int total = dwords_per_row * h;  // 32-bit index math: a 64-bit division per dword dominated this kernel
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int y = i / dwords_per_row, x4 = i - y * dwords_per_row;
    uint32_t v = *reinterpret_cast<const uint32_t*>(img + (size_t)y * pitch + 4 * x4);
    atomicAdd(&lh[wv][v & 255], 1u);
    atomicAdd(&lh[wv][(v >> 8) & 255], 1u);
    atomicAdd(&lh[wv][(v >> 16) & 255], 1u);
    atomicAdd(&lh[wv][v >> 24], 1u);
  }
  __syncthreads();
  unsigned t = lh[0][threadIdx.x] + lh[1][threadIdx.x] + lh[2][threadIdx.x] + lh[3][threadIdx.x];
  if (t) atomicAdd(&hist[s * 256 + threadIdx.x], t);
}

// one workgroup of 256 threads per stream: hist -> LUT (OpenCV: first non-zero bin -> 0, scale = 255/(total-hist[i0]))
__global__ __launch_bounds__(256) void k_equalize_lut(unsigned* __restrict__ hist, uint8_t* __restrict__ lut, int total,
                                                      const int* __restrict__ active) {
  const int s = blockIdx.x;
  if (active && !active[s]) return;
  __shared__ unsigned hs[256];
  __shared__ unsigned cs[256];
  __shared__ int first;
  const int t = threadIdx.x;
  hs[t] = hist[s * 256 + t];
  hist[s * 256 + t] = 0;  // ready for the next frame
  if (t == 0) first = 256;
  __syncthreads();
  if (hs[t]) atomicMin(&first, t);
  cs[t] = hs[t];
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // inclusive scan
    unsigned v = (t >= o) ? cs[t - o] : 0;
    __syncthreads();
    cs[t] += v;
    __syncthreads();
  }
  const int i0 = first;
  uint8_t out;
  if ((int)hs[i0] == total) {
    out = (uint8_t)i0;  // constant image: dst filled with that value (only bin i0 is ever looked up)
  } else if (t <= i0) {
    out = 0;
  } else {
    float scale = (256 - 1.f) / (float)(total - (int)hs[i0]);
    int sum = (int)(cs[t] - cs[i0]);
    int v = __float2int_rn((float)sum * scale);
    out = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
  lut[s * 256 + t] = out;
}

__global__ __launch_bounds__(256) void k_lut_apply(ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch,
                                                   size_t sstride, size_t dstride, const uint8_t* __restrict__ lut,
                                                   const int* __restrict__ active) {
  const int s = blockIdx.y;
  if (active && !active[s]) return;
  __shared__ uint8_t l[256];
  l[threadIdx.x] = lut ? lut[s * 256 + threadIdx.x] : (uint8_t)threadIdx.x;
  __syncthreads();
  const uint8_t* in = src.ptr(s, sstride);
  uint8_t* out = const_cast<uint8_t*>(dst.ptr(s, dstride));
  const int dpr = w >> 2;
  const int total = dpr * h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int y = i / dpr, x4 = i - y * dpr;
    uint32_t v = *reinterpret_cast<const uint32_t*>(in + (size_t)y * spitch + 4 * x4);
    uint32_t o = (uint32_t)l[v & 255] | ((uint32_t)l[(v >> 8) & 255] << 8) | ((uint32_t)l[(v >> 16) & 255] << 16) |
                 ((uint32_t)l[v >> 24] << 24);
    *reinterpret_cast<uint32_t*>(out + (size_t)y * dpitch + 4 * x4) = o;
  }
}

// cv::cvtColor BGR2GRAY / BGRA2GRAY (f2f_tracking.cpp:74-111): 14-bit fixed point, 4 pixels per lane (w % 4 == 0), packed in
// and out as dwords
template <int CH>
__global__ __launch_bounds__(256) void k_bgr_to_gray(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int quads) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= quads) return;
  const uint32_t* in = reinterpret_cast<const uint32_t*>(src) + (size_t)i * CH;
  uint32_t o = 0;
  if (CH == 4) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t v = in[k];
      const uint32_t y = ((v & 255u) * 1868u + ((v >> 8) & 255u) * 9617u + ((v >> 16) & 255u) * 4899u + (1u << 13)) >> 14;
      o |= y << (8 * k);
    }
  } else {
    const uint32_t a = in[0], b = in[1], c = in[2];  // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
    const uint32_t px[4][3] = {{a & 255u, (a >> 8) & 255u, (a >> 16) & 255u},
                               {a >> 24, b & 255u, (b >> 8) & 255u},
                               {(b >> 16) & 255u, b >> 24, c & 255u},
                               {(c >> 8) & 255u, (c >> 16) & 255u, c >> 24}};
#pragma unroll
    for (int k = 0; k < 4; k++) o |= ((px[k][0] * 1868u + px[k][1] * 9617u + px[k][2] * 4899u + (1u << 13)) >> 14) << (8 * k);
  }
  reinterpret_cast<uint32_t*>(dst)[i] = o;
}

// plain image copy into the pyramid's level-0 slot (modes without equalizeHist): 16 bytes per lane, w % 16 == 0
__global__ __launch_bounds__(256) void k_copy_image16(ImgSel src, ImgSel dst, int w16, int h, int spitch, int dpitch,
                                                      size_t sstride, size_t dstride, const int* __restrict__ active) {
  const int s = blockIdx.y;
  if (active && !active[s]) return;
  const uint8_t* in = src.ptr(s, sstride);
  uint8_t* out = const_cast<uint8_t*>(dst.ptr(s, dstride));
  const int total = w16 * h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int y = i / w16, x = i - y * w16;
    *reinterpret_cast<uint4*>(out + (size_t)y * dpitch + 16 * x) =
        *reinterpret_cast<const uint4*>(in + (size_t)y * spitch + 16 * x);
  }
}

// ingest of images whose rows are not dword aligned (width not a multiple of 4, e.g. KITTI's 1241 x 376, tightly packed): one
// destination dword per lane, its bytes read one by one; the padding bytes of the destination row are written as zero
__global__ __launch_bounds__(256) void k_copy_image_any(ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch,
                                                        size_t sstride, size_t dstride, const int* __restrict__ active) {
  const int s = blockIdx.y;
  if (active && !active[s]) return;
  const uint8_t* in = src.ptr(s, sstride);
  uint8_t* out = const_cast<uint8_t*>(dst.ptr(s, dstride));
  const int dpr = dpitch >> 2;
  const int total = dpr * h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int y = i / dpr, x = 4 * (i - y * dpr);
    const uint8_t* r = in + (size_t)y * spitch + x;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (x + k < w) v |= (uint32_t)r[k] << (8 * k);
    *reinterpret_cast<uint32_t*>(out + (size_t)y * dpitch + x) = v;
  }
}

// ------------------------------------------------------------------------------------------------ pyrDown
// One workgroup -> PD_TH x PD_TW destination pixels.  Source tile (2*TH+3) x (2*TW+4(+pad)) staged in LDS.
constexpr int PD_TW = 64, PD_TH = 16;
constexpr int PD_SW = 2 * PD_TW + 8;  // 136: starts at 2*x0-4 (dword aligned), covers 2*x0-2 .. 2*x0+2*TW
constexpr int PD_SH = 2 * PD_TH + 3;  // 35

// Stores the pixels (x .. x + nv - 1, y) of a level (bytes of v, low byte first; x % 4 == 0) and, for a level with a physical
// BORDER_REFLECT_101 border (bx columns, by rows; see k_pyr_border), every border pixel that mirrors one of them: a pixel at column X in
// 1 .. bx is also the border pixel at -X, one in w-1-bx .. w-2 the one at 2 (w - 1) - X, the same for rows, and both together for the
// corners.  The kernel that produces a level thus leaves it with its border complete -- no extra pass over the edges, no extra launch.
// (Needs bx < w and by < h: reflect-then-clamp, which the border is defined by, is then the plain mirror.)
__device__ __forceinline__ void store4_mirrored(uint8_t* img, int pitch, int w, int h, int bx, int by, int x, int y, uint32_t v, int nv) {
  uint8_t* const row = img + (ptrdiff_t)y * pitch;
  if (nv == 4) {
    *reinterpret_cast<uint32_t*>(row + x) = v;
  } else {
    for (int k = 0; k < nv; k++) row[x + k] = (uint8_t)(v >> (8 * k));
  }
  if (bx == 0 && by == 0) return;
  const bool xl = x <= bx && x + nv - 1 >= 1, xr = x + nv - 1 >= w - 1 - bx && x <= w - 2;
  const int ty0 = (y >= 1 && y <= by) ? -y : 0x7fffffff, ty1 = (y >= h - 1 - by && y <= h - 2) ? 2 * (h - 1) - y : 0x7fffffff;
  if (!xl && !xr && ty0 == 0x7fffffff && ty1 == 0x7fffffff) return;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int ty = t == 0 ? y : (t == 1 ? ty0 : ty1);
    if (ty == 0x7fffffff) continue;
    uint8_t* const r2 = img + (ptrdiff_t)ty * pitch;
    if (t > 0) {
      if (nv == 4) {
        *reinterpret_cast<uint32_t*>(r2 + x) = v;
      } else {
        for (int k = 0; k < nv; k++) r2[x + k] = (uint8_t)(v >> (8 * k));
      }
    }
    if (xl || xr) {
      for (int k = 0; k < nv; k++) {
        const int X = x + k;
        const uint8_t b = (uint8_t)(v >> (8 * k));
        if (X >= 1 && X <= bx) r2[-X] = b;
        if (X >= w - 1 - bx && X <= w - 2) r2[2 * (w - 1) - X] = b;
      }
    }
  }
}

template <bool INGEST>
__device__ __forceinline__ void pyr_down_tile(ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst0, int d0pitch,
                                              size_t d0stride, ImgSel dst, int dpitch, size_t dstride,
                                              const int* __restrict__ active, int bx, int by, int sbx, int sby) {
  const int s = blockIdx.z;
  if (active && !active[s]) return;
  const int dw = (sw + 1) >> 1, dh = (sh + 1) >> 1;
  const int x0 = blockIdx.x * PD_TW, y0 = blockIdx.y * PD_TH;
  __shared__ __attribute__((aligned(16))) uint8_t tile[PD_SH * PD_SW];
  __shared__ __attribute__((aligned(16))) uint16_t hrow[PD_SH * PD_TW];
  const uint8_t* img = src.ptr(s, sstride);
  load_tile_u8<PD_SH, PD_SW, PD_SW>(img, sw, sh, spitch, 2 * x0 - 4, 2 * y0 - 2, tile, sbx, sby);
  __syncthreads();
  if (INGEST) {
    // level 0 = a copy of the caller's image (it must outlive the caller's buffer: it is the "previous image" of the next
    // frame): this workgroup's 32 x 128 source block sits in the tile at rows 2..33, bytes 4..131 (dword aligned)
    uint8_t* o0 = const_cast<uint8_t*>(dst0.ptr(s, d0stride));
    for (int i = threadIdx.x; i < 2 * PD_TH * (2 * PD_TW / 4); i += 256) {
      const int r = i / (2 * PD_TW / 4), k = i - r * (2 * PD_TW / 4);
      const int y = 2 * y0 + r, x = 2 * x0 + 4 * k;
      if (y < sh && x + 3 < sw)
        store4_mirrored(o0, d0pitch, sw, sh, bx, by, x, y, *reinterpret_cast<const uint32_t*>(tile + (r + 2) * PD_SW + 4 + 4 * k), 4);
    }
  }
  // horizontal [1 4 6 4 1] at every second column: a thread makes 4 consecutive outputs of a row from four aligned dwords of
  // the tile (bytes 8k+2 .. 8k+12) and stores them as one 8-byte LDS write -- the same integer sums as pixel by pixel
  for (int i = threadIdx.x; i < PD_SH * (PD_TW / 4); i += 256) {
    const int r = i / (PD_TW / 4), k = i - r * (PD_TW / 4);
    const uint2 dlo = *reinterpret_cast<const uint2*>(tile + r * PD_SW + 8 * k);      // (rows are 136 bytes apart: 8-byte aligned)
    const uint2 dhi = *reinterpret_cast<const uint2*>(tile + r * PD_SW + 8 * k + 8);
    const uint4 d{dlo.x, dlo.y, dhi.x, dhi.y};
    const uint32_t b2 = (d.x >> 16) & 255u, b3 = d.x >> 24, b4 = d.y & 255u, b5 = (d.y >> 8) & 255u, b6 = (d.y >> 16) & 255u,
                   b7 = d.y >> 24, b8 = d.z & 255u, b9 = (d.z >> 8) & 255u, b10 = (d.z >> 16) & 255u, b11 = d.z >> 24,
                   b12 = d.w & 255u;
    const uint32_t h0 = b2 + 4 * b3 + 6 * b4 + 4 * b5 + b6, h1 = b4 + 4 * b5 + 6 * b6 + 4 * b7 + b8;
    const uint32_t h2 = b6 + 4 * b7 + 6 * b8 + 4 * b9 + b10, h3 = b8 + 4 * b9 + 6 * b10 + 4 * b11 + b12;
    *reinterpret_cast<uint2*>(hrow + r * PD_TW + 4 * k) = uint2{h0 | (h1 << 16), h2 | (h3 << 16)};
  }
  __syncthreads();
  // vertical pass: 4 consecutive outputs per thread (exactly one item per thread), one dword store
  uint8_t* out = const_cast<uint8_t*>(dst.ptr(s, dstride));
  {
    const int i = threadIdx.x;
    const int r = i / (PD_TW / 4), k = i - r * (PD_TW / 4);
    const int x = x0 + 4 * k, y = y0 + r;
    if (r < PD_TH && x < dw && y < dh) {
      uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const uint2 q = *reinterpret_cast<const uint2*>(hrow + (2 * r + j) * PD_TW + 4 * k);
        const uint32_t wgt = j == 0 || j == 4 ? 1u : (j == 2 ? 6u : 4u);
        v0 += wgt * (q.x & 0xffffu);
        v1 += wgt * (q.x >> 16);
        v2 += wgt * (q.y & 0xffffu);
        v3 += wgt * (q.y >> 16);
      }
      const uint32_t o0b = (v0 + 128) >> 8, o1b = (v1 + 128) >> 8, o2b = (v2 + 128) >> 8, o3b = (v3 + 128) >> 8;
      // (nv < 4: right edge of an image whose width is not a multiple of 4)
      store4_mirrored(out, dpitch, dw, dh, bx, by, x, y, o0b | (o1b << 8) | (o2b << 16) | (o3b << 24), dw - x < 4 ? dw - x : 4);
    }
  }
}

__global__ __launch_bounds__(256) void k_pyr_down(ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst,
                                                  int dpitch, size_t dstride, const int* __restrict__ active, int bx, int by, int sbx, int sby) {
  pyr_down_tile<false>(src, sw, sh, spitch, sstride, dst, 0, 0, dst, dpitch, dstride, active, bx, by, sbx, sby);
}

// The first pyramid level fused with the ingest copy (modes without equalizeHist): the caller's image is read ONCE and
// written out as level 0 and level 1 (the former k_copy_image16 + k_pyr_down pair read it twice).  Needs sw % 4 == 0.
__global__ __launch_bounds__(256) void k_pyr_down_ingest(ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst0,
                                                         int d0pitch, size_t d0stride, ImgSel dst, int dpitch, size_t dstride,
                                                         const int* __restrict__ active, int bx, int by) {
  pyr_down_tile<true>(src, sw, sh, spitch, sstride, dst0, d0pitch, d0stride, dst, dpitch, dstride, active, bx, by, 0, 0);
}

// ------------------------------------------------------------------------------------------------ pyramid border
// The pyramids of the tracker are stored with a physical BORDER_REFLECT_101 border of LK_BORDER_X columns / LK_BORDER_Y rows (what
// cv::buildOpticalFlowPyramid keeps around its levels, lkpyramid.cpp): k_lk_track's patch and region staging then never reflects an
// index for a point inside the image.  One launch fills the border of every level of one pyramid: blockIdx.y = level, blockIdx.z =
// stream; an item is one dword of the padded buffer that is not an image dword.  The values are img(reflect101c(y), reflect101c(x)),
// the very function the kernels' slow paths evaluate per item, so a staged block is the same bytes either way.
__global__ __launch_bounds__(256) void k_pyr_border(PyrSel pyr, const int* __restrict__ active, unsigned level_mask) {
  const int s = blockIdx.z, l = blockIdx.y;
  if (active && !active[s]) return;
  if (!((level_mask >> l) & 1u)) return;
  const int bx = pyr.bx[l], by = pyr.by[l];
  if (bx == 0 && by == 0) return;
  const int w = pyr.w[l], h = pyr.h[l], pitch = pyr.pitch[l];
  uint8_t* img = const_cast<uint8_t*>(pyr.lvl[l].ptr(s, pyr.stride[l]));
  const int rowdw = pitch >> 2;            // dwords of a padded row: columns -bx .. pitch - bx - 1
  const int wdw = w >> 2;                  // image dwords of a row that lie wholly inside the image
  const int sidedw = rowdw - wdw;          // border dwords of an image row (left border, then right border + row padding)
  const int n_tb = 2 * by * rowdw, n_side = h * sidedw;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_tb + n_side; i += gridDim.x * 256) {
    int x, y;
    if (i < n_tb) {
      const int r = i / rowdw, k = i - r * rowdw;
      y = r < by ? r - by : h + (r - by);
      x = 4 * k - bx;
    } else {
      const int j = i - n_tb;
      y = j / sidedw;
      const int k = j - y * sidedw;
      x = 4 * k < bx ? 4 * k - bx : 4 * wdw + (4 * k - bx);
    }
    const uint8_t* row = img + (ptrdiff_t)reflect101c(y, h) * pitch;
    const uint32_t b0 = row[reflect101c(x, w)], b1 = row[reflect101c(x + 1, w)], b2 = row[reflect101c(x + 2, w)], b3 = row[reflect101c(x + 3, w)];
    *reinterpret_cast<uint32_t*>(img + (ptrdiff_t)y * pitch + x) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
  }
}

// ------------------------------------------------------------------------------------------------ min-eigenvalue map
// Tile of EG_TH x EG_TW outputs (+HALO ring when HALO=1).  Needs cov on a +1 ring and image on a +2 ring beyond that.
constexpr int EG_TW = 64, EG_TH = 16;

template <int HALO>
struct EigTile {
  static constexpr int OW = EG_TW + 2 * HALO, OH = EG_TH + 2 * HALO;  // eig region
  static constexpr int CW = OW + 2, CH = OH + 2;                      // cov (sobel) region
  static constexpr int IW_ = CW + 2, IH = CH + 2;                     // image region
  static constexpr int IPAD = 4;                                      // left pad so the tile starts dword aligned
  static constexpr int IW = ((IW_ + IPAD + 3) / 4) * 4;               // LDS tile width (bytes)
};

// Computes the eig values of one tile into LDS `eig` (OH x OW floats).  Float op order is fixed (raster 3x3 sums) so results are reproducible bit-for-bit.
template <int HALO>
__device__ __forceinline__ void eig_tile(const uint8_t* __restrict__ img, int w, int h, int pitch, int x0, int y0,
                                         uint8_t* tile, float* sfx, float* sfy, float* eig) {
  using T = EigTile<HALO>;
  // image tile origin: (x0 - HALO - 2 - IPAD, y0 - HALO - 2); x0 is a multiple of 64, IPAD chosen so origin % 4 == 0
  constexpr int XOFF = HALO + 2 + (4 - ((HALO + 2) & 3)) % 4;  // HALO=0 -> 4 ; HALO=1 -> 4
  static_assert(XOFF % 4 == 0 && XOFF <= T::IPAD + 3, "alignment");
  const int ix0 = x0 - XOFF, iy0 = y0 - HALO - 2;
  load_tile_u8<T::IH, T::IW, T::IW>(img, w, h, pitch, ix0, iy0, tile);
  __syncthreads();
  const float scale = (float)(1.0 / (255.0 * 4.0 * 3.0));
  // sobel on the cov region; position (X,Y) is first reflected into the image (boxFilter REFLECT_101 on cov)
  for (int i = threadIdx.x; i < T::CH * T::CW; i += blockDim.x) {
    int r = i / T::CW, c = i - r * T::CW;
    int X = x0 - HALO - 1 + c, Y = y0 - HALO - 1 + r;
    int Xr = reflect101(X, w), Yr = reflect101(Y, h);
    // clamp for tiles that hang over the right/bottom edge far beyond the image (values unused there)
    int tx = Xr - ix0, ty = Yr - iy0;
    float fx = 0.f, fy = 0.f;
    if (tx >= 1 && tx < T::IW - 1 && ty >= 1 && ty < T::IH - 1) {
      const uint8_t* p = tile + ty * T::IW + tx;
      int a = p[-T::IW - 1], b = p[-T::IW], cc = p[-T::IW + 1];
      int d = p[-1], f = p[1];
      int g = p[T::IW - 1], hh = p[T::IW], k = p[T::IW + 1];
      int dx = (cc + 2 * f + k) - (a + 2 * d + g);
      int dy = (g + 2 * hh + k) - (a + 2 * b + cc);
      fx = (float)dx * scale;
      fy = (float)dy * scale;
    }
    sfx[i] = fx;
    sfy[i] = fy;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T::OH * T::OW; i += blockDim.x) {
    int r = i / T::OW, c = i - r * T::OW;
    float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        float fx = sfx[(r + j) * T::CW + c + k], fy = sfy[(r + j) * T::CW + c + k];
        sa += fx * fx;
        sb += fx * fy;
        sc += fy * fy;
      }
    float a = sa * 0.5f, b = sb, cc = sc * 0.5f;
    eig[i] = (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
  }
  __syncthreads();
}

// One pass over the image: the eig map of the tile (+1 ring), the per-stream maximum (minMaxLoc; maxenc[s] must be zero
// before the launch) and the 3x3 local maxima as 64-bit sort keys, key = ~((ordered(val) << 32) | pixel_offset), so that
// ascending key order == (val desc, offset desc).
// cv::goodFeaturesToTrack thresholds first (THRESH_TOZERO at max*qualityLevel), dilates, and keeps val != 0 && val == dilated.
// For a pixel above the threshold that is the same as "no raw neighbour is greater": a greater neighbour is itself above
// the threshold, a neighbour at or below it becomes 0.  So the local-maximum test does not need the threshold, the map is
// computed ONCE (it used to be computed twice: once for the maximum, once for the thresholded maxima), and the threshold
// is applied by k_gftt_pick, which runs when the maximum is complete.  (Removing the second pass takes 0.15 ms off a
// 64-stream step.)
__global__ __launch_bounds__(256) void k_eig_cand(ImgSel src, int w, int h, int pitch, size_t sstride,
                                                  unsigned* __restrict__ maxenc, unsigned long long* __restrict__ keys,
                                                  int* __restrict__ nkeys, int cap, const int* __restrict__ active) {
  const int s = blockIdx.z;
  if (active && !active[s]) return;
  using T = EigTile<1>;
  __shared__ __attribute__((aligned(16))) uint8_t tile[T::IH * T::IW];
  __shared__ float sfx[T::CH * T::CW], sfy[T::CH * T::CW], eig[T::OH * T::OW];
  __shared__ unsigned wmax[4];
  const int x0 = blockIdx.x * EG_TW, y0 = blockIdx.y * EG_TH;
  eig_tile<1>(src.ptr(s, sstride), w, h, pitch, x0, y0, tile, sfx, sfy, eig);
  // candidates are collected per workgroup in LDS and appended with ONE global atomic per workgroup (the per-stream
  // counter would otherwise serialise ~10^4 atomics per image); their order is irrelevant, the keys are sorted next
  __shared__ unsigned long long lkeys[EG_TH * EG_TW];
  __shared__ int lcount, lbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  unsigned m = 0;
  for (int i = threadIdx.x; i < EG_TH * EG_TW; i += 256) {
    int r = i / EG_TW, c = i - r * EG_TW;
    int x = x0 + c, y = y0 + r;
    if (x >= w || y >= h) continue;
    const float* e = eig + (r + 1) * T::OW + (c + 1);
    const float v = e[0];
    const unsigned ev = f32_ordered(v);
    m = ev > m ? ev : m;
    if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) continue;
    if (!(v > 0.f)) continue;
    bool ismax = true;
#pragma unroll
    for (int j = -1; j <= 1; j++)
#pragma unroll
      for (int k = -1; k <= 1; k++)
        if (e[j * T::OW + k] > v) ismax = false;
    if (ismax) {
      int slot = atomicAdd(&lcount, 1);
      unsigned long long key = ((unsigned long long)ev << 32) | (unsigned)(y * w + x);
      lkeys[slot] = ~key;
    }
  }
  m = wave_max_u32(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  const int n = lcount;
  if (threadIdx.x == 0) {
    unsigned a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    atomicMax(&maxenc[s], a > b ? a : b);
    if (n > 0) lbase = atomicAdd(&nkeys[s], n);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    int slot = lbase + i;
    if (slot < cap) keys[(size_t)s * cap + slot] = lkeys[i];
  }
}

// The same pass with the strip-mined phases of eig_strip.hpp (four Sobel pairs / four responses per thread from one register
// window; bit-identical values -- tests/test_eig_strip.py runs those very functions on the host over whole images).  Opt-in
// (FLVIS_EIG_STRIP=1) until it has been measured on the GPU against k_eig_cand.
__global__ __launch_bounds__(256) void k_eig_cand_strip(ImgSel src, int w, int h, int pitch, size_t sstride, unsigned* __restrict__ maxenc,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ nkeys, int cap,
                                                        const int* __restrict__ active) {
  namespace ES = eigstrip;
  static_assert(ES::TW == EG_TW && ES::TH == EG_TH && ES::IW == EigTile<1>::IW && ES::IH == EigTile<1>::IH && ES::CW == EigTile<1>::CW &&
                    ES::OW == EigTile<1>::OW,
                "eig_strip.hpp describes the tile of k_eig_cand");
  const int s = blockIdx.z;
  if (active && !active[s]) return;
  __shared__ __attribute__((aligned(16))) uint8_t tile[ES::IH * ES::IW];
  __shared__ __attribute__((aligned(16))) float sfx[ES::CH * ES::CW], sfy[ES::CH * ES::CW], eig[ES::OH * ES::OW];
  __shared__ unsigned wmax[4];
  const int x0 = blockIdx.x * EG_TW, y0 = blockIdx.y * EG_TH;
  load_tile_u8<ES::IH, ES::IW, ES::IW>(src.ptr(s, sstride), w, h, pitch, x0 - ES::XOFF, y0 - ES::HALO - 2, tile);
  __syncthreads();
  for (int item = threadIdx.x; item < ES::A_ITEMS; item += 256) ES::sobel_strip(item, w, h, x0, y0, tile, sfx, sfy);
  __syncthreads();
  for (int item = threadIdx.x; item < ES::B_ITEMS; item += 256) ES::box_strip(item, sfx, sfy, eig);
  __syncthreads();
  // 3x3 maxima: one strip of four output pixels per thread; the keys of the workgroup are gathered in LDS (one LDS atomic per
  // thread that found any) and appended to the stream's list with ONE global atomic, as in k_eig_cand
  __shared__ unsigned long long lkeys[EG_TH * EG_TW];
  __shared__ int lcount, lbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  static_assert(ES::C_ITEMS == 256, "one strip per thread");
  unsigned long long k4[4];
  uint32_t m = 0;
  const unsigned found = ES::nms_strip(threadIdx.x, w, h, x0, y0, eig, k4, m);
  if (found) {
    const int slot = atomicAdd(&lcount, __popc(found));
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (found >> i & 1u) lkeys[slot + __popc(found & ((1u << i) - 1u))] = k4[i];
  }
  m = wave_max_u32(m);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  const int n = lcount;
  if (threadIdx.x == 0) {
    unsigned a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    atomicMax(&maxenc[s], a > b ? a : b);
    if (n > 0) lbase = atomicAdd(&nkeys[s], n);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    int slot = lbase + i;
    if (slot < cap) keys[(size_t)s * cap + slot] = lkeys[i];
  }
}

// ------------------------------------------------------------------------------------------------ ranked selection
// cv::goodFeaturesToTrack's tail: sort the candidates by (response desc, offset desc) and walk them greedily, dropping a
// candidate iff an already accepted corner lies at squared distance < minDist^2, until maxCorners are accepted.
// With qualityLevel ~1e-3 an image yields tens of thousands of candidates but the walk stops after a few hundred, so the
// candidates are NOT sorted as a whole: one workgroup per stream builds a 4096-bin histogram of the top 12 bits of the
// (order-preserving) response, takes the highest bins that fit a 4096-key LDS tier, sorts only that tier (bitonic, in
// LDS) and walks it; further tiers follow only while maxCorners is not reached.  Keys of one bin always travel together,
// so the walk order is exactly the fully sorted order.  A single bin that overflows a tier (degenerate images) falls
// back to the full bitonic sort through L2.  The accepted corners live in an LDS bitmap (w*h bits); a candidate tests
// its 2R+1 rows against per-row chord masks of the disc (two 32-bit windows per row), the candidates of one 64-wide
// batch are resolved against each other in rank order by a short fixed-point iteration over ballots.
constexpr int SORT_T = 1024;
constexpr int SORT_LDS = 4096;
constexpr int PICK_BINS = 4096;
constexpr int PICK_MAXR = 64;  // largest minDistance handled by the chord-mask walk (checked by the callers)

// full ascending bitonic sort of K[0..np2) (np2 = n rounded up to a power of two, padded with ~0) using the 4096-key LDS
// window sk; sub-sequences that fit the window are finished in LDS, wider strides go through L2
__device__ void sort_keys_global(unsigned long long* __restrict__ K, int n, int cap, unsigned long long* sk) {
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  if (np2 > cap) np2 = cap;  // cap is a power of two
  for (int i = n + threadIdx.x; i < np2; i += SORT_T) K[i] = ~0ull;  // pad (sorts last)
  __syncthreads();
  if (np2 <= 1) return;
  const int chunk = np2 < SORT_LDS ? np2 : SORT_LDS;
  for (int base = 0; base < np2; base += chunk) {
    for (int i = threadIdx.x; i < chunk; i += SORT_T) sk[i] = K[base + i];
    __syncthreads();
    for (int k = 2; k <= chunk; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < chunk / 2; t += SORT_T) {
          int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          int l = i | j;
          bool up = (((base + i) & k) == 0);
          unsigned long long a = sk[i], b = sk[l];
          if ((a > b) == up) {
            sk[i] = b;
            sk[l] = a;
          }
        }
        __syncthreads();
      }
    for (int i = threadIdx.x; i < chunk; i += SORT_T) K[base + i] = sk[i];
    __syncthreads();
  }
  for (int k = chunk << 1; k <= np2; k <<= 1) {
    int j = k >> 1;
    for (; j >= chunk; j >>= 1) {  // global strides
      for (int t = threadIdx.x; t < np2 / 2; t += SORT_T) {
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int l = i | j;
        bool up = ((i & k) == 0);
        unsigned long long a = K[i], b = K[l];
        if ((a > b) == up) {
          K[i] = b;
          K[l] = a;
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    for (int base = 0; base < np2; base += chunk) {  // strides < chunk in LDS
      for (int i = threadIdx.x; i < chunk; i += SORT_T) sk[i] = K[base + i];
      __syncthreads();
      for (int jj = chunk >> 1; jj > 0; jj >>= 1) {
        for (int t = threadIdx.x; t < chunk / 2; t += SORT_T) {
          int i = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
          int l = i | jj;
          bool up = (((base + i) & k) == 0);
          unsigned long long a = sk[i], b = sk[l];
          if ((a > b) == up) {
            sk[i] = b;
            sk[l] = a;
          }
        }
        __syncthreads();
      }
      for (int i = threadIdx.x; i < chunk; i += SORT_T) K[base + i] = sk[i];
      __syncthreads();
    }
  }
}

// ascending bitonic sort of the first m keys of sk (LDS), m <= SORT_LDS
__device__ void sort_keys_lds(unsigned long long* sk, int m) {
  int np2 = 1;
  while (np2 < m) np2 <<= 1;
  for (int i = m + threadIdx.x; i < np2; i += SORT_T) sk[i] = ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < np2 / 2; t += SORT_T) {
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        int l = i | j;
        bool up = ((i & k) == 0);
        unsigned long long a = sk[i], b = sk[l];
        if ((a > b) == up) {
          sk[i] = b;
          sk[l] = a;
        }
      }
      __syncthreads();
    }
}

struct PickState {
  int accepted;
  int chord[PICK_MAXR + 1];  // largest |dx| inside the disc at |dy|, -1: none
};

// greedy walk of m sorted keys (ascending = best first) by ONE wave; returns with ps.accepted updated
__device__ void pick_walk(const unsigned long long* keys, int m, int w, int h, int wpr, unsigned* bitmap, int maxc, float md2,
                          int R, bool use_dist, float* __restrict__ out, int out_cap, PickState& ps) {
  const int lane = threadIdx.x & 63;
  int accepted = ps.accepted;
  for (int base = 0; base < m && (maxc <= 0 || accepted < maxc); base += 64) {
    const int i = base + lane;
    const bool valid = i < m;
    int x = 0, y = 0;
    if (valid) {
      unsigned off = (unsigned)(~keys[i]);
      y = off / (unsigned)w;
      x = off - y * w;
    }
    bool good = valid;
    if (good && use_dist) {
      for (int dy = -R; dy <= R && good; dy++) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        const int cw = ps.chord[dy < 0 ? -dy : dy];
        if (cw < 0) continue;
        int xa = x - cw, xb = x + cw;  // inclusive column range of the disc in this row
        if (xa < 0) xa = 0;
        if (xb >= w) xb = w - 1;
        const unsigned* row = bitmap + yy * wpr;
        for (int wd = xa >> 5; wd <= (xb >> 5); wd++) {
          const int lo = (wd << 5) > xa ? 0 : (xa & 31), hi = ((wd << 5) + 31) < xb ? 31 : (xb & 31);
          const unsigned mask = (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo);
          if (row[wd] & mask) {
            good = false;
            break;
          }
        }
      }
    }
    // resolve the batch in rank order: cm = earlier lanes of the batch inside my disc; a lane is decided once all lanes
    // of its cm are, and is accepted iff none of them was accepted
    unsigned long long accmask = __ballot(good);
    if (use_dist) {
      unsigned long long cm = 0;
      const unsigned long long cand = accmask;
      for (unsigned long long todo = cand; todo;) {
        const int k = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int kx = __builtin_amdgcn_readlane(x, k), ky = __builtin_amdgcn_readlane(y, k);
        if (lane > k) {
          float fdx = (float)(x - kx), fdy = (float)(y - ky);
          if (fdx * fdx + fdy * fdy < md2) cm |= 1ull << k;
        }
      }
      unsigned long long decided = ~cand;  // lanes without a candidate are decided (rejected)
      unsigned long long acc = 0;
      bool mine = !good;
      while (true) {
        bool now = false, a = false;
        if (!mine && (cm & ~decided) == 0ull) {
          now = true;
          a = (cm & acc) == 0ull;
        }
        const unsigned long long bn = __ballot(now), ba = __ballot(now && a);
        if (bn == 0ull) break;
        decided |= bn;
        acc |= ba;
        if (now) {
          mine = true;
          good = a;
        }
      }
      accmask = acc;
    }
    const int pos = accepted + lane_prefix(accmask);
    if (good && (maxc <= 0 || pos < maxc) && pos < out_cap) {
      out[2 * pos] = (float)x;
      out[2 * pos + 1] = (float)y;
      atomicOr(&bitmap[y * wpr + (x >> 5)], 1u << (x & 31));
    }
    accepted += __popcll(accmask);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
  ps.accepted = accepted;
}

__global__ __launch_bounds__(SORT_T) void k_gftt_pick(unsigned long long* __restrict__ keys, int* __restrict__ nkeys, int cap,
                                                      int w, int h, const int* __restrict__ max_corners_s, int max_corners,
                                                      double min_distance, float* __restrict__ out_xy,
                                                      int* __restrict__ out_n, int out_cap, const int* __restrict__ active,
                                                      unsigned* __restrict__ maxenc, const double* __restrict__ qual_s,
                                                      double quality) {
  const int s = blockIdx.x;
  if (active && !active[s]) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char pick_smem[];
  unsigned long long* sk = reinterpret_cast<unsigned long long*>(pick_smem);          // [SORT_LDS] tier
  int* hist = reinterpret_cast<int*>(sk + SORT_LDS);                                   // [PICK_BINS + 1] suffix counts
  unsigned* bitmap = reinterpret_cast<unsigned*>(hist + PICK_BINS + 8);                // ceil(w/32) words per row
  __shared__ PickState ps;
  __shared__ int s_lo, s_hi, s_m, s_full, s_fill;
  const int tid = threadIdx.x;
  // threshold(eig, eig, maxVal * qualityLevel, 0, THRESH_TOZERO): candidates at or below it do not exist for the selection
  const unsigned thr_ord = f32_ordered((float)((double)f32_unordered(maxenc[s]) * (qual_s ? qual_s[s] : quality)));
  __syncthreads();
  if (tid == 0) maxenc[s] = 0;  // the eig pass is done with it: hand it back zeroed
  int n = nkeys[s];
  if (n > cap) n = cap;
  const int maxc = max_corners_s ? max_corners_s[s] : max_corners;
  unsigned long long* K = keys + (size_t)s * cap;
  float* out = out_xy + (size_t)s * out_cap * 2;
  const bool use_dist = min_distance >= 1.0;
  const float md2 = (float)(min_distance * min_distance);
  int R = use_dist ? (int)__double2int_rn(min_distance) : 0;  // |dx|,|dy| < minDistance <= R (+0.5)
  if (R > PICK_MAXR) R = PICK_MAXR;  // (callers reject minDistance > PICK_MAXR)
  const int wpr = (w + 31) >> 5;
  for (int i = tid; i < wpr * h; i += SORT_T) bitmap[i] = 0;
  for (int i = tid; i <= PICK_BINS; i += SORT_T) hist[i] = 0;
  if (tid <= PICK_MAXR) {
    int cw = -1;
    if (tid <= R)
      for (int dx = 0; dx <= R; dx++) {
        float fdx = (float)dx, fdy = (float)tid;
        if (fdx * fdx + fdy * fdy < md2) cw = dx;
      }
    ps.chord[tid] = cw;
  }
  if (tid == 0) {
    ps.accepted = 0;
    s_full = 0;
  }
  __syncthreads();
  // histogram of the top 12 bits of the ordered response (bin 4095 = best)
  for (int i = tid; i < n; i += SORT_T) {
    const unsigned long long nk = ~K[i];
    if ((unsigned)(nk >> 32) > thr_ord) atomicAdd(&hist[(unsigned)(nk >> 52)], 1);
  }
  __syncthreads();
  // suffix sums: hist[b] <- number of keys with bin >= b (hist[PICK_BINS] = 0); 4 bins per thread + workgroup scan
  {
    const int b0 = PICK_BINS - 4 * (tid + 1);  // this thread's bins b0..b0+3, thread 0 owns the top four
    const int c3 = hist[b0 + 3], c2 = hist[b0 + 2], c1 = hist[b0 + 1], c0 = hist[b0];
    int mysum = c0 + c1 + c2 + c3, inc = mysum;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    __shared__ int wtot[SORT_T / 64];
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int before = inc - mysum;  // keys in better bins
    for (int k = 0; k < wv; k++) before += wtot[k];
    hist[b0 + 3] = before + c3;
    hist[b0 + 2] = before + c3 + c2;
    hist[b0 + 1] = before + c3 + c2 + c1;
    hist[b0] = before + mysum;
  }
  __syncthreads();
  int hi = PICK_BINS;  // bins >= hi are consumed
  bool sorted_all = false;
  while (true) {
    if (tid == 0) {
      int lo = hi, m = 0;
      if (!s_full && hi > 0) {
        // lowest lo with count(lo..hi-1) <= the tier's size: binary search on the monotone suffix counts.  The first tier is sized by what
        // the walk usually looks at before maxCorners are accepted (about three candidates per corner with the minimum distances of the
        // reference's rigs), not by what fits: sorting 2048 keys instead of 4096 is twelve bitonic passes fewer on half the data; a
        // walk that needs more gets the next tier (the order is the fully sorted order either way)
        int tier = SORT_LDS;
        if (hi == PICK_BINS && maxc > 0 && 3 * maxc < SORT_LDS) tier = 3 * maxc > 1024 ? 3 * maxc : 1024;
        const int basec = hist[hi];
        int a = 0, b = hi - 1;  // answer in [a, hi-1] if bin hi-1 alone fits
        if (hist[hi - 1] - basec > SORT_LDS) {
          s_full = 1;
        } else {
          if (hist[hi - 1] - basec > tier) tier = SORT_LDS;  // (the best bin alone is larger than the small tier)
          while (a < b) {
            const int mid = (a + b) >> 1;
            if (hist[mid] - basec <= tier) b = mid;
            else a = mid + 1;
          }
          lo = a;
          m = hist[lo] - basec;
        }
      }
      s_lo = lo;
      s_hi = hi;
      s_m = m;
    }
    __syncthreads();
    if (s_full) break;
    const int lo = s_lo, m = s_m;
    if (lo == hi) break;  // nothing left
    // gather the tier (order irrelevant: sorted next)
    if (tid == 0) s_fill = 0;
    __syncthreads();
    for (int i = tid; i < n; i += SORT_T) {
      const unsigned long long k = K[i];
      const int bin = (int)((~k) >> 52);
      if (bin >= lo && bin < hi && (unsigned)((~k) >> 32) > thr_ord) sk[atomicAdd(&s_fill, 1)] = k;
    }
    __syncthreads();
    sort_keys_lds(sk, m);
    if (tid < 64) pick_walk(sk, m, w, h, wpr, bitmap, maxc, md2, R, use_dist, out, out_cap, ps);
    __syncthreads();
    hi = lo;
    if ((maxc > 0 && ps.accepted >= maxc) || hi == 0) {
      sorted_all = true;
      break;
    }
  }
  if (!sorted_all && s_full) {
    // degenerate response distribution (or a very large minDistance): full sort, then walk the sorted array; corners
    // accepted by earlier tiers stay (their keys come first again and are re-walked against an emptied bitmap)
    for (int i = tid; i < wpr * h; i += SORT_T) bitmap[i] = 0;
    if (tid == 0) ps.accepted = 0;
    __syncthreads();
    sort_keys_global(K, n, cap, sk);
    __syncthreads();
    const int nv = hist[0];  // keys above the threshold: a prefix of the sorted array
    if (tid < 64) {
      for (int base = 0; base < nv && (maxc <= 0 || ps.accepted < maxc); base += SORT_LDS) {
        const int m = (nv - base) < SORT_LDS ? (nv - base) : SORT_LDS;
        pick_walk(K + base, m, w, h, wpr, bitmap, maxc, md2, R, use_dist, out, out_cap, ps);
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    int accepted = ps.accepted;
    if (maxc > 0 && accepted > maxc) accepted = maxc;
    if (accepted > out_cap) accepted = out_cap;
    out_n[s] = accepted;
    nkeys[s] = 0;  // last consumer: hand the candidate counter back zeroed
  }
}

// ------------------------------------------------------------------------------------------------ FeatureDEM
// One wave per stream.  mode[s]==1: detect (init), mode[s]==2: redetect (existing landmark positions seed the regions).
// Reproduces feature_dem.cpp including calHarrisR's quirks, integer cv::Point rounding in redetect, the cross-shaped
// spacing test and the "push then check size" region cap.  std::sort ties are resolved stably (GFTT rank order).
constexpr int DEM_MAXC = 4096;   // max GFTT corners per call (2*gftt_num; KITTI.yaml asks for 2 x 2000); candidate arrays in dynamic LDS
constexpr int DEM_MAXR = 192;    // max entries kept per region (existing + new)

__device__ __forceinline__ float dem_harris(const uint8_t* __restrict__ img, int pitch, float ptx, float pty) {
  int xx = (int)ptx, yy = (int)pty;
  const uint8_t* p = img + (size_t)yy * pitch + xx;
  int p0 = p[-pitch - 1], p1 = p[-pitch], p2 = p[-pitch + 1];
  int p3 = p[-1];
  int p5 = p[pitch + 1];  // quirk (feature_dem.cpp:71)
  int p6 = p[pitch - 1], p7 = p[pitch], p8 = p[pitch + 1];
  float IX = (float)((p0 + p3 + p6 - (p2 + p5 + p8)) / 3);
  float IY = (float)((p0 + p1 + p2 - (p6 + p7 + p8)) / 3);
  float X2 = IX * IX, Y2 = IY * IX, XY = IX * IX;
  return (X2 * Y2) - (XY * XY) - 0.05f * (X2 + Y2) * (X2 + Y2);
}

constexpr int DEM_T = 256;
// Two kernels since round 3.  k_feature_dem_prep needs only the corner list and the image: region and Harris score of every candidate,
// candidates grouped by region and sorted by score inside a region (stable) -> per stream the sorted coordinates (region-major) and the
// region offsets.  In the tracker it runs on the detection stream right after k_gftt_pick.  k_feature_dem is what stays on the frame's
// critical path: the existing landmarks fill their regions, the greedy spacing walk, the output.
constexpr int DEMP_T = 1024;  // k_feature_dem_prep: 16 waves, one per region where a region is sorted once more (ties)
// Accessor of dem_sort.hpp for an array of up to 64 NREG elements that lives in NREG (1 or 2) vector registers ACROSS the lanes of the
// wave (element e in lane e & 63 of register e >> 6): every index is wave-uniform, so get / set are v_readlane / v_writelane and the sequential
// algorithm runs on the scalar unit.  An element is (class << 16) | position, class = number of candidates of the region with a
// strictly greater score: a greater score <=> a smaller class, equal scores <=> equal classes (sortbysecdesc through integers).
#ifndef FLVIS_DEM_SORT_LANES
#define FLVIS_DEM_SORT_LANES 0
#endif
// ... and for the same packed elements as a plain array (LDS)
struct PackedArray {
  typedef int value_type;
  int* v;
  __device__ __forceinline__ int get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, int x) const { v[i] = x; }
  __device__ __forceinline__ bool before(int a, int b) const { return (a >> 16) < (b >> 16); }
};
template <int NREG>
struct LaneArray {
  typedef int value_type;
  int& r0;
  int& r1;  // (NREG == 1: unused)
  // v_writelane_b32 with the value in an SGPR and the lane in M0 (a VOP3 reads one SGPR on gfx9); M0 is saved and restored
  static __device__ __forceinline__ int writelane(int old, int val, int lane) {
    int tmp;
    val = __builtin_amdgcn_readfirstlane(val);    // (wave-uniform by construction; the constraint "s" needs the compiler to know it)
    lane = __builtin_amdgcn_readfirstlane(lane);
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(old), "=&s"(tmp)
                 : "s"(val), "s"(lane));
    return old;
  }
  __device__ __forceinline__ int get(int i) const {
    i = __builtin_amdgcn_readfirstlane(i);
    const int a = __builtin_amdgcn_readlane(r0, i & 63);
    if (NREG == 1) return a;
    const int b = __builtin_amdgcn_readlane(r1, i & 63);
    return i < 64 ? a : b;
  }
  __device__ __forceinline__ void set(int i, int x) const {
    i = __builtin_amdgcn_readfirstlane(i);
    if (NREG == 1) {
      r0 = writelane(r0, x, i);
    } else {  // (both registers are written, one result is kept: no pointer selects, the array stays in registers)
      const int n0 = writelane(r0, x, i & 63), n1 = writelane(r1, x, i & 63);
      r0 = i < 64 ? n0 : r0;
      r1 = i < 64 ? r1 : n1;
    }
  }
  __device__ __forceinline__ bool before(int a, int b) const { return (a >> 16) < (b >> 16); }
};

__device__ __forceinline__ void k_feature_dem_prep_body(ImgSel src, int w, int h, int pitch, size_t sstride, DemParams prm,
                                                            const float* __restrict__ corners, const int* __restrict__ ncorners,
                                                            int corner_cap, const int* __restrict__ active, float* __restrict__ sorted_xy,
                                                            int* __restrict__ region_off) {
  const int s = blockIdx.x;
  if (active && !active[s]) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // candidate arrays, sized by the corner capacity of the call (dynamic LDS: 18 bytes per corner)
  extern __shared__ __attribute__((aligned(16))) unsigned char dem_smem[];
  const int cmax = ((corner_cap < DEM_MAXC ? corner_cap : DEM_MAXC) + 1) & ~1;
  float* cx = reinterpret_cast<float*>(dem_smem);
  float* cy = cx + cmax;
  float* cscore = cy + cmax;
  short* creg = reinterpret_cast<short*>(cscore + cmax);
  short* bucket = creg + cmax;      // candidate indices grouped by region, index order inside a region
  __shared__ int rcount[16], roff[17];
  const uint8_t* img = src.ptr(s, sstride);
  int nc = ncorners[s];
  if (nc > corner_cap) nc = corner_cap;
  if (nc > DEM_MAXC) nc = DEM_MAXC;
  const float* C = corners + (size_t)s * corner_cap * 2;
  if (tid < 16) rcount[tid] = 0;
  __syncthreads();
  // candidates: region + score
  for (int i = tid; i < nc; i += DEMP_T) {
    float px = C[2 * i], py = C[2 * i + 1];
    int r = -1;
    float sc = 0.f;
    if (px >= 3 && px < (w - 3) && py >= 3 && py < (h - 3)) {
      r = (int)(4.f * floorf(py / (float)prm.regionHeight) + px / (float)prm.regionWidth);
      sc = dem_harris(img, pitch, px, py);
      atomicAdd(&rcount[r], 1);
    }
    cx[i] = px;
    cy[i] = py;
    cscore[i] = sc;
    creg[i] = (short)r;
  }
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int r = 0; r < 16; r++) {
      roff[r] = o;
      o += rcount[r];
    }
    roff[16] = o;
  }
  __syncthreads();
  // bucket the candidates by region keeping index order: wave q handles regions q, q+4, ...: ballots over the candidates
  for (int r = wv; r < 16; r += DEMP_T / 64) {
    int o = roff[r];
    for (int base = 0; base < nc; base += 64) {
      const int i = base + lane;
      const bool in = i < nc && creg[i] == r;
      const unsigned long long bq = __ballot(in);
      if (in) bucket[o + lane_prefix(bq)] = (short)i;
      o += __popcll(bq);
    }
  }
  __syncthreads();
  // Order inside a region = what the reference's std::sort(..., sortbysecdesc) leaves (feature_dem.cpp:170,230).  Where all scores of
  // a region differ that is THE descending order: position = #(score greater), one candidate per thread.  std::sort is not stable,
  // though, and the quirky score is built from two small integers, so ties happen: a region with a tie is sorted once more by ONE lane
  // that walks through libstdc++'s introsort (dem_sort.hpp) on the region's candidates in their input order -- the order of equal
  // scores then is the one the reference's binary produces, which decides whom the greedy spacing walk meets first.
  float* const SX = sorted_xy + (size_t)s * corner_cap * 2;
  __shared__ int rtie[16];
  __shared__ int sstack[16][3 * demsort::STACK];
  int* const ccls = reinterpret_cast<int*>(cx + 3 * cmax) + cmax;  // (behind creg / bucket: class of every bucket position; 4 more bytes per corner)
  if (tid < 16) rtie[tid] = 0;
  __syncthreads();
  for (int j = tid; j < roff[16]; j += DEMP_T) {
    const int i = bucket[j];
    const int r = creg[i];
    const float sc = cscore[i];
    int greater = 0, earlier = 0;
    bool tie = false;
    for (int q = roff[r]; q < roff[r + 1]; q++) {
      const float sj = cscore[bucket[q]];
      greater += sj > sc;
      earlier += sj == sc && q < j;
      tie = tie || (sj == sc && q != j);
    }
    if (tie) rtie[r] = 1;
    ccls[j] = (greater << 16) | (j - roff[r]);  // (class, input position in the region): the element the region's sort moves around
    const int pos = greater + earlier;
    SX[2 * (roff[r] + pos)] = cx[i];
    SX[2 * (roff[r] + pos) + 1] = cy[i];
  }
  __syncthreads();
  // one wave per region with a tie: libstdc++'s introsort on the region's candidates in their input order, as packed (class, position)
  // integers.  Up to 16 candidates std::sort IS an insertion sort, i.e. stable: the ranks above are its result already.  Beyond that
  // only std::sort's first phase -- the quicksort levels on ranges longer than 16 -- is walked through sequentially (lane 0 of the
  // region's wave, array in LDS; FLVIS_DEM_SORT_LANES: in registers across the lanes): the insertion sort that follows is stable, so its
  // result is the STABLE order of what the first phase leaves, and that is a rank every lane computes for its own elements (position =
  // class + equal classes earlier in the array): n log2(n / 16) sequential element visits instead of n log2 n + n^2 / 64.
  if (wv < 16 && rtie[wv] && roff[wv + 1] - roff[wv] > demsort::THRESHOLD) {
    const int r0 = roff[wv], n = roff[wv + 1] - r0;
#if FLVIS_DEM_SORT_LANES
    if (n <= 128) {
      int a0 = lane < n ? ccls[r0 + lane] : 0;
      int a1 = 64 + lane < n ? ccls[r0 + 64 + lane] : 0;
      if (n <= 64)
        demsort::quicksort_phase(LaneArray<1>{a0, a1}, n, sstack[wv]);
      else
        demsort::quicksort_phase(LaneArray<2>{a0, a1}, n, sstack[wv]);
      if (lane < n) ccls[r0 + lane] = a0;
      if (64 + lane < n) ccls[r0 + 64 + lane] = a1;
    } else
#endif
    {
      if (lane == 0) demsort::quicksort_phase(PackedArray{ccls + r0}, n, sstack[wv]);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int e = lane; e < n; e += 64) {
      const int key = ccls[r0 + e], cls = key >> 16;
      int earlier = 0;
      for (int q = 0; q < e; q++) earlier += (ccls[r0 + q] >> 16) == cls;
      const int i = bucket[r0 + (key & 0xffff)];
      SX[2 * (r0 + cls + earlier)] = cx[i];
      SX[2 * (r0 + cls + earlier) + 1] = cy[i];
    }
  }
  if (tid < 17) region_off[(size_t)s * 17 + tid] = roff[tid];
}
__global__ __launch_bounds__(DEMP_T) void k_feature_dem_prep(ImgSel src, int w, int h, int pitch, size_t sstride, DemParams prm,
                                                            const float* __restrict__ corners, const int* __restrict__ ncorners,
                                                            int corner_cap, const int* __restrict__ active, float* __restrict__ sorted_xy,
                                                            int* __restrict__ region_off, KJoin kj) {
  kj_wait(kj);
  k_feature_dem_prep_body(src, w, h, pitch, sstride, prm, corners, ncorners, corner_cap, active, sorted_xy, region_off);
  kj_signal(kj);
}

__device__ __forceinline__ void k_feature_dem_body(int w, int h, DemParams prm, const float* __restrict__ sorted_xy,
                                                       const int* __restrict__ region_off, int corner_cap, const int* __restrict__ mode,
                                                       const double* __restrict__ exist_xy, const int* __restrict__ nexist,
                                                       int exist_cap, float* __restrict__ out_xy, int* __restrict__ out_n,
                                                       int out_cap) {
  const int s = blockIdx.x;
  const int md = mode ? mode[s] : 1;
  if (md == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // the sorted candidates, staged once (dynamic LDS: 8 bytes per corner)
  extern __shared__ __attribute__((aligned(16))) unsigned char dem_smem[];
  const int cmax = ((corner_cap < DEM_MAXC ? corner_cap : DEM_MAXC) + 1) & ~1;
  float* cx = reinterpret_cast<float*>(dem_smem);
  float* cy = cx + cmax;
  __shared__ int roff[17];
  __shared__ float kx[16][DEM_MAXR], ky[16][DEM_MAXR];
  __shared__ int kcount[16], knew0[16], ooff[17];
  __shared__ int wcnt[DEM_T / 64][16];
  if (tid < 17) roff[tid] = region_off[(size_t)s * 17 + tid];
  if (tid < 16) kcount[tid] = 0;
  __syncthreads();
  {
    const float* const SX = sorted_xy + (size_t)s * corner_cap * 2;
    const int tot = roff[16] < cmax ? roff[16] : cmax;
    for (int j = tid; j < tot; j += DEM_T) {
      cx[j] = SX[2 * j];
      cy[j] = SX[2 * j + 1];
    }
  }
  // existing features (redetect) fill their regions in landmark order: ordered per-region ranks by ballots, chunk by chunk
  if (md == 2) {
    int ne = nexist[s];
    if (ne > exist_cap) ne = exist_cap;
    const double* E = exist_xy + (size_t)s * exist_cap * 2;
    for (int base = 0; base < ne; base += DEM_T) {
      const int i = base + tid;
      int r = -1;
      float px = 0.f, py = 0.f;
      if (i < ne) {
        px = (float)E[2 * i];
        py = (float)E[2 * i + 1];
        if (px >= 3 && px < (w - 3) && py >= 3 && py < (h - 3))
          r = (int)(4.f * floorf(py / (float)prm.regionHeight) + px / (float)prm.regionWidth);
      }
      int myrank = 0;
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const unsigned long long bq = __ballot(r == q);
        if (lane == 0) wcnt[wv][q] = __popcll(bq);
        if (r == q) myrank = lane_prefix(bq);
      }
      __syncthreads();
      if (r >= 0) {
        int k = kcount[r] + myrank;
        for (int v = 0; v < wv; v++) k += wcnt[v][r];
        if (k < DEM_MAXR) {
          kx[r][k] = px;
          ky[r][k] = py;
        }
      }
      __syncthreads();
      if (tid < 16) {
        int k = kcount[tid];
        for (int v = 0; v < DEM_T / 64; v++) k += wcnt[v][tid];
        kcount[tid] = k < DEM_MAXR ? k : DEM_MAXR;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid < 16) knew0[tid] = kcount[tid];
  __syncthreads();
  // greedy spacing per region: 16 lanes per region, the lanes split the already kept points of the region
  {
    const int r = tid >> 4, sub = tid & 15;  // 256 threads = 16 regions x 16 lanes (a quarter wave each)
    const int bd = prm.boundary_dis;
    int kept = kcount[r];
    unsigned count = 0;
    const int j0 = roff[r], j1 = roff[r + 1];
    for (int j = j0; j < j1; j++) {
      float px = cx[j], py = cy[j];
      if (md == 2) {  // cv::Point pt = Point2f (rounds), feature_dem.cpp:174
        px = (float)__float2int_rn(px);
        py = (float)__float2int_rn(py);
      }
      int bad = 0;
      for (int k = sub; k < kept; k += 16) {
        float dis_x = fabsf(px - kx[r][k]);
        float dis_y = fabsf(py - ky[r][k]);
        if (dis_x <= (float)bd || dis_y <= (float)bd) bad = 1;
      }
      // OR over the 16 lanes of this region (one row of 16 lanes of the wave): one ballot of the whole wave, this row's 16 bits
      // (a row that has left the loop contributes zeros and does not look)
      bad = (int)((__ballot(bad != 0) >> (lane & 48)) & 0xFFFFull);
      if (!bad) {
        if (kept < DEM_MAXR && sub == 0) {
          kx[r][kept] = px;
          ky[r][kept] = py;
        }
        kept++;
        count++;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        if (md == 1) {
          if (count >= prm.max_region_feature_num) break;
        } else {
          if ((unsigned)kept >= prm.max_region_feature_num) break;
        }
      }
    }
    if (sub == 0) kcount[r] = kept < DEM_MAXR ? kept : DEM_MAXR;
  }
  __syncthreads();
  // output: new points, regions in order 0..15, per region in acceptance order
  if (tid == 0) {
    int o = 0;
    for (int r = 0; r < 16; r++) {
      ooff[r] = o;
      o += kcount[r] - knew0[r];
    }
    ooff[16] = o;
    out_n[s] = o < out_cap ? o : out_cap;
  }
  __syncthreads();
  {
    float* O = out_xy + (size_t)s * out_cap * 2;
    const int r = tid >> 4, sub = tid & 15;
    for (int k = knew0[r] + sub; k < kcount[r]; k += 16) {
      const int o = ooff[r] + (k - knew0[r]);
      if (o < out_cap) {
        O[2 * o] = kx[r][k];
        O[2 * o + 1] = ky[r][k];
      }
    }
  }
}
__global__ __launch_bounds__(DEM_T) void k_feature_dem(int w, int h, DemParams prm, const float* __restrict__ sorted_xy,
                                                       const int* __restrict__ region_off, int corner_cap, const int* __restrict__ mode,
                                                       const double* __restrict__ exist_xy, const int* __restrict__ nexist,
                                                       int exist_cap, float* __restrict__ out_xy, int* __restrict__ out_n,
                                                       int out_cap, KJoin kj) {
  kj_wait(kj);
  k_feature_dem_body(w, h, prm, sorted_xy, region_off, corner_cap, mode, exist_xy, nexist, exist_cap, out_xy, out_n, out_cap);
  kj_signal(kj);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline int div_up(int a, int b) { return (a + b - 1) / b; }

void launch_equalize_hist(hipStream_t st, ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch, size_t sstride,
                          size_t dstride, int S, unsigned* hist, uint8_t* lut, const int* active) {
  int blocks = div_up((w / 4) * h, 256 * 8);
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(k_hist256, dim3(blocks, S), dim3(256), 0, st, src, w, h, spitch, sstride, hist, active);
  hipLaunchKernelGGL(k_equalize_lut, dim3(S), dim3(256), 0, st, hist, lut, w * h, active);
  int ablocks = div_up((w / 4) * h, 256 * 4);
  hipLaunchKernelGGL(k_lut_apply, dim3(ablocks, S), dim3(256), 0, st, src, dst, w, h, spitch, dpitch, sstride, dstride,
                     (const uint8_t*)lut, active);
}

void launch_bgr_to_gray(hipStream_t st, const uint8_t* src, int channels, uint8_t* dst, size_t npixels) {
  const int quads = (int)(npixels / 4);
  if (channels == 4)
    hipLaunchKernelGGL(k_bgr_to_gray<4>, dim3(div_up(quads, 256)), dim3(256), 0, st, src, dst, quads);
  else
    hipLaunchKernelGGL(k_bgr_to_gray<3>, dim3(div_up(quads, 256)), dim3(256), 0, st, src, dst, quads);
}

void launch_copy_image(hipStream_t st, ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch, size_t sstride,
                       size_t dstride, int S, const int* active) {
  const bool a16 = !src.ind && (w % 16) == 0 && (spitch % 16) == 0 && (dpitch % 16) == 0 && (sstride % 16) == 0 && (dstride % 16) == 0 &&
                   ((uintptr_t)src.b[0] % 16) == 0 && ((uintptr_t)src.b[1] % 16) == 0 && ((uintptr_t)dst.b[0] % 16) == 0 &&
                   ((uintptr_t)dst.b[1] % 16) == 0;
  if (a16) {
    hipLaunchKernelGGL(k_copy_image16, dim3(div_up((w / 16) * h, 256 * 2), S), dim3(256), 0, st, src, dst, w / 16, h, spitch,
                       dpitch, sstride, dstride, active);
    return;
  }
  int ablocks = div_up((w / 4) * h, 256 * 4);
  hipLaunchKernelGGL(k_lut_apply, dim3(ablocks, S), dim3(256), 0, st, src, dst, w, h, spitch, dpitch, sstride, dstride,
                     (const uint8_t*)nullptr, active);
}

void launch_copy_image_any(hipStream_t st, ImgSel src, ImgSel dst, int w, int h, int spitch, int dpitch, size_t sstride,
                           size_t dstride, int S, const int* active) {
  hipLaunchKernelGGL(k_copy_image_any, dim3(div_up((dpitch / 4) * h, 256 * 4), S), dim3(256), 0, st, src, dst, w, h, spitch, dpitch,
                     sstride, dstride, active);
}

bool pyr_border_fusable(int w, int h, int bx, int by) { return bx < w && by < h; }

void launch_pyr_down(hipStream_t st, ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst, int dpitch,
                     size_t dstride, int S, const int* active, int bx, int by, int sbx, int sby) {
  int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  hipLaunchKernelGGL(k_pyr_down, dim3(div_up(dw, PD_TW), div_up(dh, PD_TH), S), dim3(256), 0, st, src, sw, sh, spitch,
                     sstride, dst, dpitch, dstride, active, bx, by, sbx, sby);
}

void launch_pyr_down_ingest(hipStream_t st, ImgSel src, int sw, int sh, int spitch, size_t sstride, ImgSel dst0, int d0pitch,
                            size_t d0stride, ImgSel dst, int dpitch, size_t dstride, int S, const int* active, int bx, int by) {
  int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
  hipLaunchKernelGGL(k_pyr_down_ingest, dim3(div_up(dw, PD_TW), div_up(dh, PD_TH), S), dim3(256), 0, st, src, sw, sh, spitch,
                     sstride, dst0, d0pitch, d0stride, dst, dpitch, dstride, active, bx, by);
}

// stage_events (optional): 6 events = (begin, end) for eig_max, eig_nms, gftt_pick.
// reset_counters: zero maxenc / nkeys first; a caller that allocated them zeroed and always runs the full chain can pass
// false -- k_gftt_pick hands them back zeroed.
// the corner-response pass on its own: per-stream maximum (maxenc, zeroed by the caller) and the candidate keys (keys / nkeys, zeroed)
// variant 0: k_eig_cand (LDS tiles), 1: k_eig_cand_strip, 2: k_eig_walk with `rows` rows per chunk
void launch_pyr_border(hipStream_t st, const PyrSel& pyr, int S, const int* active, unsigned level_mask) {
  bool any = false;
  for (int l = 0; l <= pyr.levels; l++) any = any || (((level_mask >> l) & 1u) && (pyr.bx[l] > 0 || pyr.by[l] > 0));
  if (!any) return;
  hipLaunchKernelGGL(k_pyr_border, dim3(8, pyr.levels + 1, S), dim3(256), 0, st, pyr, active, level_mask);
}
void launch_corner_response(hipStream_t st, int variant, int rows, ImgSel src, int w, int h, int pitch, size_t sstride, int S, unsigned* maxenc,
                            unsigned long long* keys, int* nkeys, int cap, const int* active) {
  dim3 grid(div_up(w, EG_TW), div_up(h, EG_TH), S);
  if (variant == 2)
    launch_eig_walk(st, src, w, h, pitch, sstride, S, maxenc, keys, nkeys, cap, active, rows);
  else if (variant == 1)
    hipLaunchKernelGGL(k_eig_cand_strip, grid, dim3(256), 0, st, src, w, h, pitch, sstride, maxenc, keys, nkeys, cap, active);
  else
    hipLaunchKernelGGL(k_eig_cand, grid, dim3(256), 0, st, src, w, h, pitch, sstride, maxenc, keys, nkeys, cap, active);
}

void launch_gftt(hipStream_t st, ImgSel src, int w, int h, int pitch, size_t sstride, int S, GfttScratch sc,
                 const double* qual_s, double quality, const int* maxc_s, int max_corners, double min_distance,
                 float* out_xy, int* out_n, int out_cap, const int* active, hipEvent_t* ev, bool reset_counters) {
  if (reset_counters) {
    hipMemsetAsync(sc.maxenc, 0, sizeof(unsigned) * S, st);
    hipMemsetAsync(sc.nkeys, 0, sizeof(int) * S, st);
  }
  if (ev) hipEventRecord(ev[0], st);
  // which kernel computes the corner response: the wave walk (eig_walk.hip) with 60 rows per chunk unless FLVIS_EIG_WALK=<rows> says
  // otherwise; FLVIS_EIG_WALK=0 selects the LDS-tile kernel k_eig_cand, FLVIS_EIG_WALK=0 FLVIS_EIG_STRIP=1 its strip-mined form (both
  // kept for the A/B of profiles/r03_eig_walk_ab.md and as independent implementations in the parity tests)
  static const int variant_rows = [] {
    const char* e = getenv("FLVIS_EIG_WALK");
    if (!e) return 60;
    if (atoi(e) > 0) return atoi(e);
    e = getenv("FLVIS_EIG_STRIP");
    return (e && atoi(e) != 0) ? -2 : -1;
  }();
  launch_corner_response(st, variant_rows > 0 ? 2 : (variant_rows == -2 ? 1 : 0), variant_rows, src, w, h, pitch, sstride, S, sc.maxenc, sc.keys,
                         sc.nkeys, sc.cap, active);
  if (ev) hipEventRecord(ev[1], st), hipEventRecord(ev[2], st), hipEventRecord(ev[3], st), hipEventRecord(ev[4], st);
  const size_t lds = sizeof(unsigned long long) * SORT_LDS + sizeof(int) * (PICK_BINS + 8) +
                     (size_t)((w + 31) / 32) * h * sizeof(unsigned);
  hipLaunchKernelGGL(k_gftt_pick, dim3(S), dim3(SORT_T), lds, st, sc.keys, sc.nkeys, sc.cap, w, h, maxc_s, max_corners,
                     min_distance, out_xy, out_n, out_cap, active, sc.maxenc, qual_s, quality);
  if (ev) hipEventRecord(ev[5], st);
}

void launch_feature_dem_prep(hipStream_t st, ImgSel src, int w, int h, int pitch, size_t sstride, int S, DemParams prm,
                             const float* corners, const int* ncorners, int corner_cap, const int* active, float* sorted_xy,
                             int* region_off, const KJoin* kj) {
  const int cmax = ((corner_cap < DEM_MAXC ? corner_cap : DEM_MAXC) + 1) & ~1;
  hipLaunchKernelGGL(k_feature_dem_prep, dim3(S), dim3(DEMP_T), (size_t)cmax * 20, st, src, w, h, pitch, sstride, prm, corners, ncorners,
                     corner_cap, active, sorted_xy, region_off, kj ? *kj : KJoin{});
}
void launch_feature_dem(hipStream_t st, int w, int h, int S, DemParams prm, const float* sorted_xy, const int* region_off,
                        int corner_cap, const int* mode, const double* exist_xy, const int* nexist, int exist_cap, float* out_xy,
                        int* out_n, int out_cap, const KJoin* kj) {
  const int cmax = ((corner_cap < DEM_MAXC ? corner_cap : DEM_MAXC) + 1) & ~1;
  hipLaunchKernelGGL(k_feature_dem, dim3(S), dim3(DEM_T), (size_t)cmax * 8, st, w, h, prm, sorted_xy, region_off, corner_cap, mode,
                     exist_xy, nexist, exist_cap, out_xy, out_n, out_cap, kj ? *kj : KJoin{});
}

hipError_t img_kernels_init() {
  // the selection bitmap may exceed the default 64 KB dynamic LDS window only for images > 512K pixels
  hipError_t e = hipFuncSetAttribute((const void*)k_feature_dem_prep, hipFuncAttributeMaxDynamicSharedMemorySize, DEM_MAXC * 20);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_gftt_pick, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

}  // namespace flvis
