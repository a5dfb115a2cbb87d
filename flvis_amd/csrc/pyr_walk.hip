// flvis_amd: cv::pyrDown levels by walking waves (gfx950) -- the pyramids of cv::calcOpticalFlowPyrLK (buildOpticalFlowPyramid) as
// F2FTracking / LKORBTracking use them (src/processing/lkorb_tracking.cpp:64-73, src/frontend/camera_frame.cpp:124-128).
//
// The tile kernels (img_kernels.hip: k_pyr_down, k_pyr_down_ingest) stage a 136 x 35 block per workgroup in LDS, wait, filter rows
// into LDS, wait, filter columns: three dependent phases and two barriers for ~5 KB of data, thousands of short-lived workgroups.
// Here a WAVE owns a band of rows over the full image width and walks down it:
//   * a lane owns 16 consecutive pixels of a source row (one 16-byte load; a 640-pixel row is one 640-byte request of 40 lanes),
//     8 of the next level, 4 and 2 of the ones after;
//   * the two / one pixels a lane needs from its neighbours for the horizontal [1 4 6 4 1] come from the neighbouring LANE
//     (ds_bpermute of one dword each way), the image's left / right REFLECT_101 edge from the lane's own bytes;
//   * the horizontal sums of a row are pairs of 16-bit fields in a dword (v_perm_b32 picks the operand bytes), and the vertical
//     [1 4 6 4 1] is kept as TWO running sums per output column -- a source row 2m adds 6x to output row m and 1x to rows m-1 and
//     m+1, a row 2m+1 adds 4x to rows m and m+1 -- so no row is held back: a row is loaded once, used, and gone.  The top / bottom
//     REFLECT_101 edge only changes those weights (row 1 counts 8x for output row 0, ...); every sum is at most 255 * 16 * 16 =
//     65280 and fits its 16-bit field, so the packed dword arithmetic is the same integer arithmetic as pixel by pixel;
//   * a finished output row is stored and -- when the launch produces two or three levels -- handed to the next level's horizontal
//     pass at once, in registers (8 bytes per lane -> the same code at half the width);
//   * the rows a band needs above and below its own (2 source rows per side for one level, 6 for two) are loaded again by the
//     neighbouring band: 19 loads for 16 rows;
//   * a stored row goes to its home position and to every position of the level's physical border (img_kernels.hpp: LK_BORDER_X
//     columns, LK_BORDER_Y rows) that mirrors it, like store4_mirrored does for the tile kernels;
//   * with COPY0 the source row itself is stored as level 0 of the pyramid (the ingest copy of the caller's image).
// The results are bit-identical with the tile kernels and with the checker's cv::pyrDown restatement (tests/test_gpu_image.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "img_kernels.hpp"

namespace flvis {

// (file-local names carry the prefix pw_ / Walk; no anonymous namespace: "(anonymous namespace)" in a kernel's name breaks the profile
// summaries' name parsing)
typedef uint32_t pw_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t pw_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) uint8_t pw_g8;

constexpr int PW_MAXL = 4;  // the source + up to three produced levels
// dwords a lane holds of a row of level J below the source (16, 8, 4 pixels; level 3: 2 pixels in the low half of one dword)
constexpr int pw_nd(int J) { return J < 3 ? 4 >> J : 1; }

struct PwWalkArgs {
  ImgSel src;
  int sw, sh, spitch;
  size_t sstride;
  ImgSel lvl[PW_MAXL];  // [0]: level 0 of the pyramid (the copy of the source, COPY0 only), [1..nout]: the produced levels
  int pitch[PW_MAXL];
  size_t stride[PW_MAXL];
  int bx[PW_MAXL], by[PW_MAXL];
  const int* active;
  int rows_per_band;  // rows of the LAST produced level per wave
};

struct PwWalkCtx {
  uint8_t* ptr[PW_MAXL];
  int pitch[PW_MAXL], W[PW_MAXL], H[PW_MAXL], bx[PW_MAXL], by[PW_MAXL];
  int need_lo[PW_MAXL], need_hi[PW_MAXL];  // rows of a level this band loads / produces (inclusive)
  int own_lo[PW_MAXL], own_hi[PW_MAXL];    // rows of a level this band stores [lo, hi)
  int lane, nl, addr_l, addr_r;
};

struct PwWalkAcc {  // the two running vertical sums of the levels that are filtered: cur = the output row in progress, nxt = the one after
  uint32_t c0[4], n0[4], c1[2], n1[2], c2[1], n2[1];
};
template <int J>
__device__ __forceinline__ uint32_t (&pw_cur(PwWalkAcc& A))[pw_nd(J)] {
  if constexpr (J == 0) return A.c0;
  else if constexpr (J == 1) return A.c1;
  else return A.c2;
}
template <int J>
__device__ __forceinline__ uint32_t (&pw_nxt(PwWalkAcc& A))[pw_nd(J)] {
  if constexpr (J == 0) return A.n0;
  else if constexpr (J == 1) return A.n1;
  else return A.n2;
}

__device__ __forceinline__ uint32_t pw_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// a * w + c on the two 16-bit fields of a dword (v_pk_mad_u16; no field exceeds 65280 here, so nothing wraps)
typedef unsigned short pw_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pw_mad(uint32_t a, uint32_t w, uint32_t c) {
  const pw_h2 wv = {(unsigned short)w, (unsigned short)w};
  return __builtin_bit_cast(uint32_t, (pw_h2)(__builtin_bit_cast(pw_h2, a) * wv + __builtin_bit_cast(pw_h2, c)));
}
__device__ __forceinline__ uint32_t pw_mul(uint32_t a, uint32_t w) {
  const pw_h2 wv = {(unsigned short)w, (unsigned short)w};
  return __builtin_bit_cast(uint32_t, (pw_h2)(__builtin_bit_cast(pw_h2, a) * wv));
}

template <int ND>
__device__ __forceinline__ void pw_store(pw_g8* p, const uint32_t (&d)[ND]) {
  if constexpr (ND == 4) *reinterpret_cast<__attribute__((address_space(1))) pw_u4*>(p) = pw_u4{d[0], d[1], d[2], d[3]};
  else if constexpr (ND == 2) *reinterpret_cast<__attribute__((address_space(1))) pw_u2*>(p) = pw_u2{d[0], d[1]};
  else *reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(p) = d[0];
}

// Stores row Y of a level (4 ND bytes per lane; dl / dr: the left neighbour's last and the right neighbour's first dword) at its home
// and at the border positions that mirror it: columns -X = X (X = 1 .. bx), W - 1 + X = W - 1 - X, rows likewise.
template <int ND>
__device__ __forceinline__ void pw_store_row(const PwWalkCtx& c, int J, int Y, const uint32_t (&d)[ND], uint32_t dl, uint32_t dr) {
  constexpr int NB = 4 * ND;
  const int W = c.W[J], H = c.H[J], bx = c.bx[J], by = c.by[J], pitch = c.pitch[J];
  pw_g8* const base = (pw_g8*)c.ptr[J];
  const int ty0 = (Y >= 1 && Y <= by) ? -Y : 0x7fffffff, ty1 = (Y >= H - 1 - by && Y <= H - 2 && by > 0) ? 2 * (H - 1) - Y : 0x7fffffff;
  const int nbl = bx / NB;  // lanes whose pixels have an image in the left (the last nbl lanes: in the right) border
  uint32_t lb[ND], rb[ND];
  if (bx) {
#pragma unroll
    for (int m = 0; m < ND; m++) {
      const int q = ND - 1 - m;
      // left border dword m of this lane: columns -NB (lane + 1) + 4 m .. + 3 = the lane's pixels NB - 4 m, ... - 1, - 2, - 3 (pixel NB: the neighbour's first)
      lb[m] = pw_perm(q + 1 < ND ? d[q + 1 < ND ? q + 1 : 0] : dr, d[q], 0x01020304u);
      // right border dword m of the lane j = nl - 1 - lane: columns W + NB j + 4 m .. + 3 = the lane's pixels NB - 2 - 4 m, ... - 3 (pixel -1: the neighbour's last)
      rb[m] = pw_perm(d[q], q > 0 ? d[q > 0 ? q - 1 : 0] : dl, 0x03040506u);
    }
  }
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int ty = t == 0 ? Y : (t == 1 ? ty0 : ty1);
    if (ty == 0x7fffffff) continue;
    pw_g8* const row = base + (ptrdiff_t)ty * pitch;
    if (c.lane < c.nl) pw_store<ND>(row + NB * c.lane, d);
    if (bx) {
      if (c.lane < nbl) pw_store<ND>(row - NB * (c.lane + 1), lb);
      if (c.lane >= c.nl - nbl && c.lane < c.nl) pw_store<ND>(row + W + NB * (c.nl - 1 - c.lane), rb);
    }
  }
}

// ... of a level of 2 pixels per lane (the low half of d)
__device__ __forceinline__ void pw_store_row_half(const PwWalkCtx& c, int J, int Y, uint32_t d, uint32_t dl, uint32_t dr) {
  const int W = c.W[J], H = c.H[J], bx = c.bx[J], by = c.by[J], pitch = c.pitch[J];
  pw_g8* const base = (pw_g8*)c.ptr[J];
  const int ty0 = (Y >= 1 && Y <= by) ? -Y : 0x7fffffff, ty1 = (Y >= H - 1 - by && Y <= H - 2 && by > 0) ? 2 * (H - 1) - Y : 0x7fffffff;
  const int nbl = bx / 2;
  // left border: columns -2 (lane + 1), + 1 = the lane's pixels 2 (the neighbour's first), 1; right border of lane j = nl - 1 - lane:
  // columns W + 2 j, + 1 = its pixels 0, -1 (the neighbour's second)
  const unsigned short lb = (unsigned short)pw_perm(dr, d, 0x0c0c0104u), rb = (unsigned short)pw_perm(d, dl, 0x0c0c0104u);
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int ty = t == 0 ? Y : (t == 1 ? ty0 : ty1);
    if (ty == 0x7fffffff) continue;
    pw_g8* const row = base + (ptrdiff_t)ty * pitch;
    typedef __attribute__((address_space(1))) unsigned short gus;
    if (c.lane < c.nl) *reinterpret_cast<gus*>(row + 2 * c.lane) = (unsigned short)d;
    if (bx) {
      if (c.lane < nbl) *reinterpret_cast<gus*>(row - 2 * (c.lane + 1)) = lb;
      if (c.lane >= c.nl - nbl && c.lane < c.nl) *reinterpret_cast<gus*>(row + W + 2 * (c.nl - 1 - c.lane)) = rb;
    }
  }
}

// horizontal [1 4 6 4 1] at the even columns of the lane's 4 ND pixels: h[i] = the sums centred on pixels 4 i (low half) and 4 i + 2
template <int ND>
__device__ __forceinline__ void pw_hsum(const uint32_t (&d)[ND], uint32_t dl, uint32_t dr, uint32_t (&h)[ND]) {
#pragma unroll
  for (int i = 0; i < ND; i++) {
    const uint32_t prev = i ? d[i ? i - 1 : 0] : dl, next = i + 1 < ND ? d[i + 1 < ND ? i + 1 : 0] : dr;
    const uint32_t em2 = pw_perm(d[i], prev, 0x0c040c02u);  // pixels 4 i - 2 | 4 i
    const uint32_t em1 = pw_perm(d[i], prev, 0x0c050c03u);  //        4 i - 1 | 4 i + 1
    const uint32_t e0 = d[i] & 0x00ff00ffu;                 //        4 i     | 4 i + 2
    const uint32_t e1 = pw_perm(0u, d[i], 0x0c030c01u);     //        4 i + 1 | 4 i + 3
    const uint32_t e2 = pw_perm(next, d[i], 0x0c040c02u);   //        4 i + 2 | 4 i + 4
    h[i] = e0 * 6u + (((em1 + e1) << 2) + (em2 + e2));
  }
}

template <int J, int NOUT, bool COPY0>
__device__ __forceinline__ void pw_push(const PwWalkCtx& c, PwWalkAcc& A, int r, const uint32_t (&d)[pw_nd(J)]);

// a finished vertical sum (16 weights) -> the output row's bytes, stored and / or handed on
template <int J, int NOUT, bool COPY0>
__device__ __forceinline__ void pw_emit(const PwWalkCtx& c, PwWalkAcc& A, int Y, const uint32_t (&acc)[pw_nd(J)]) {
  constexpr int ND = pw_nd(J);
  uint32_t o[pw_nd(J + 1)];
  if constexpr (ND >= 2) {
#pragma unroll
    for (int k = 0; k < ND / 2; k++) o[k] = pw_perm(acc[2 * k + 1] + 0x00800080u, acc[2 * k] + 0x00800080u, 0x07050301u);  // (v + 128) >> 8
  } else {
    o[0] = pw_perm(0u, acc[0] + 0x00800080u, 0x0c0c0301u);  // two pixels
  }
  pw_push<J + 1, NOUT, COPY0>(c, A, Y, o);
}

// row r of level J (rows arrive in increasing order, the first one is even)
template <int J, int NOUT, bool COPY0>
__device__ __forceinline__ void pw_push(const PwWalkCtx& c, PwWalkAcc& A, int r, const uint32_t (&d)[pw_nd(J)]) {
  constexpr int ND = pw_nd(J);
  uint32_t dl = (uint32_t)__builtin_amdgcn_ds_bpermute(c.addr_l, (int)d[ND - 1]);
  uint32_t dr = (uint32_t)__builtin_amdgcn_ds_bpermute(c.addr_r, (int)d[0]);
  if ((J > 0 || COPY0) && r >= c.own_lo[J] && r < c.own_hi[J]) {
    if constexpr (J < 3) pw_store_row<ND>(c, J, r, d, dl, dr);
    else pw_store_row_half(c, J, r, d[0], dl, dr);
  }
  if constexpr (J < NOUT) {
    // the image's own edges: columns -2, -1 are columns 2, 1; column W is column W - 2
    if (c.lane == 0) dl = pw_perm(0u, d[0], 0x01020c0cu);
    if (c.lane == c.nl - 1) dr = pw_perm(0u, d[ND - 1], 0x0c0c0c02u);
    uint32_t h[ND];
    pw_hsum<ND>(d, dl, dr, h);
    uint32_t(&cur)[ND] = pw_cur<J>(A);
    uint32_t(&nxt)[ND] = pw_nxt<J>(A);
    const int H = c.H[J], lo = c.need_lo[J + 1], hi = c.need_hi[J + 1];
    uint32_t ev[ND];
    int Ya, Yb;  // output rows this row completes: Ya from ev (even rows), Yb from cur (the last row of the level)
    if (!(r & 1)) {
      // output row r / 2 - 1 is complete with this row; r / 2 takes it 6x, r / 2 + 1 once.  REFLECT_101: row 2 counts twice for output row
      // 0 (it is row -2 as well); the last row but one (H even) is row H as well: 7x; the last row but two (H odd) is row H + 1 as well
      const uint32_t we = r == 2 ? 2u : 1u, wc = r == H - 2 ? 7u : 6u, wn = r == H - 3 ? 2u : 1u;
#pragma unroll
      for (int i = 0; i < ND; i++) {
        ev[i] = pw_mad(h[i], we, cur[i]);
        cur[i] = pw_mad(h[i], wc, nxt[i]);
        nxt[i] = pw_mul(h[i], wn);
      }
      Ya = (r >> 1) - 1;
      Yb = Ya + 1;  // (H odd: the last row completes two output rows)
    } else {
      // 4x for output rows (r - 1) / 2 and (r + 1) / 2.  REFLECT_101: row 1 is row -1 as well, the last row but one (H odd) is row H as well
      const uint32_t wc = r == 1 ? 8u : 4u, wn = r == H - 2 ? 8u : 4u;
#pragma unroll
      for (int i = 0; i < ND; i++) {
        ev[i] = 0;
        cur[i] = pw_mad(h[i], wc, cur[i]);
        nxt[i] = pw_mad(h[i], wn, nxt[i]);
      }
      Ya = -1;
      Yb = (r - 1) >> 1;  // (H even: the last row completes the last output row)
    }
    const bool em_a = Ya >= lo && Ya <= hi, em_b = r == H - 1 && Yb >= lo && Yb <= hi;
    // ONE copy of everything that follows an output row (its stores, the next level's filter): the two cases take turns in a loop
#pragma unroll 1
    for (int e = 0; e < 2; e++) {
      if (!(e == 0 ? em_a : em_b)) continue;
      uint32_t v[ND];
#pragma unroll
      for (int i = 0; i < ND; i++) v[i] = e == 0 ? ev[i] : cur[i];
      pw_emit<J, NOUT, COPY0>(c, A, e == 0 ? Ya : Yb, v);
    }
  }
}

// source rows in flight per wave (the loop body is unrolled as many times)
constexpr int pw_pf(int) { return 4; }

template <int NOUT, bool COPY0>
__global__ __launch_bounds__(256) void k_pyr_walk(PwWalkArgs a) {
  const int s = blockIdx.y;
  if (a.active && !a.active[s]) return;
  PwWalkCtx c;
  c.lane = threadIdx.x & 63;
  c.nl = a.sw >> 4;
  c.addr_l = 4 * (c.lane > 0 ? c.lane - 1 : 0);
  c.addr_r = 4 * (c.lane + 1 < 64 ? c.lane + 1 : 63);
  c.W[0] = a.sw;
  c.H[0] = a.sh;
#pragma unroll
  for (int j = 1; j < PW_MAXL; j++) c.W[j] = (c.W[j - 1] + 1) >> 1, c.H[j] = (c.H[j - 1] + 1) >> 1;
#pragma unroll
  for (int j = 0; j < PW_MAXL; j++) {
    const bool on = j <= NOUT && (j > 0 || COPY0);
    c.ptr[j] = on ? const_cast<uint8_t*>(a.lvl[j].ptr(s, a.stride[j])) : nullptr;
    c.pitch[j] = a.pitch[j], c.bx[j] = a.bx[j], c.by[j] = a.by[j];
    c.need_lo[j] = c.need_hi[j] = c.own_lo[j] = c.own_hi[j] = 0;
  }
  const int band = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Y0 = band * a.rows_per_band;
  if (Y0 >= c.H[NOUT]) return;
  const int Y1 = Y0 + a.rows_per_band < c.H[NOUT] ? Y0 + a.rows_per_band : c.H[NOUT];
  c.need_lo[NOUT] = Y0, c.need_hi[NOUT] = Y1 - 1;
  c.own_lo[NOUT] = Y0, c.own_hi[NOUT] = Y1;
#pragma unroll
  for (int j = NOUT - 1; j >= 0; j--) {
    const int lo = 2 * c.need_lo[j + 1] - 2, hi = 2 * c.need_hi[j + 1] + 2;
    c.need_lo[j] = lo > 0 ? lo : 0;
    c.need_hi[j] = hi < c.H[j] - 1 ? hi : c.H[j] - 1;
    c.own_lo[j] = 2 * c.own_lo[j + 1];
    c.own_hi[j] = 2 * c.own_hi[j + 1] < c.H[j] ? 2 * c.own_hi[j + 1] : c.H[j];
  }
  PwWalkAcc A;
#pragma unroll
  for (int i = 0; i < 4; i++) A.c0[i] = A.n0[i] = 0;
#pragma unroll
  for (int i = 0; i < 2; i++) A.c1[i] = A.n1[i] = 0;
  A.c2[0] = A.n2[0] = 0;
  const pw_g8* const img = (const pw_g8*)a.src.ptr(s, a.sstride) + 16 * (c.lane < c.nl ? c.lane : c.nl - 1);
  const int r0 = c.need_lo[0], r1 = c.need_hi[0];
  constexpr int PW_PF = pw_pf(NOUT);
  pw_u4 buf[PW_PF];
#pragma unroll
  for (int k = 0; k < PW_PF; k++) {
    const int r = r0 + k < r1 ? r0 + k : r1;
    buf[k] = *reinterpret_cast<const __attribute__((address_space(1))) pw_u4*>(img + (size_t)r * a.spitch);
  }
  for (int base = r0; base <= r1; base += PW_PF) {
#pragma unroll
    for (int k = 0; k < PW_PF; k++) {
      const int r = base + k;
      if (r > r1) break;
      const uint32_t d[4] = {buf[k].x, buf[k].y, buf[k].z, buf[k].w};
      if (r + PW_PF <= r1) buf[k] = *reinterpret_cast<const __attribute__((address_space(1))) pw_u4*>(img + (size_t)(r + PW_PF) * a.spitch);
      pw_push<0, NOUT, COPY0>(c, A, r, d);
    }
  }
}

bool pyr_walk_ok(int sw, int sh, int nout, const int* bx, const int* by, bool copy0) {
  static const bool off = getenv("FLVIS_PYR_TILES") && atoi(getenv("FLVIS_PYR_TILES")) != 0;  // (A/B knob: the LDS-tile kernels)
  if (off || nout < 1 || nout > 3 || (sw & 15) || sw < 64 || sw > 1024 || sh < (8 << nout)) return false;
  int w = sw, h = sh;
  for (int j = 0; j <= nout; j++) {
    if ((j > 0 || copy0) && (bx[j] || by[j])) {
      if ((bx[j] & 15) || 2 * bx[j] > w || by[j] >= h - 1) return false;
    }
    w = (w + 1) >> 1, h = (h + 1) >> 1;
  }
  return true;
}

void launch_pyr_walk(hipStream_t st, ImgSel src, int sw, int sh, int spitch, size_t sstride, const PyrSel& pyr, int first, int nout, bool copy0,
                     int S, const int* active) {
  PwWalkArgs a{};
  a.src = src, a.sw = sw, a.sh = sh, a.spitch = spitch, a.sstride = sstride;
  for (int j = 0; j <= nout; j++) {
    a.lvl[j] = pyr.lvl[first + j];
    a.pitch[j] = pyr.pitch[first + j], a.stride[j] = pyr.stride[first + j];
    a.bx[j] = pyr.bx[first + j], a.by[j] = pyr.by[first + j];
  }
  a.active = active;
  // rows of the last produced level per wave (A/B knobs; FLVIS_PYR_BAND: launches of one level, FLVIS_PYR_BAND2: of two)
  static const int rows_env = getenv("FLVIS_PYR_BAND") ? atoi(getenv("FLVIS_PYR_BAND")) : 0;
  static const int rows_env2 = getenv("FLVIS_PYR_BAND2") ? atoi(getenv("FLVIS_PYR_BAND2")) : 0;
  int hl = sh;
  for (int j = 0; j < nout; j++) hl = (hl + 1) >> 1;
  static const int rows_env3 = getenv("FLVIS_PYR_BAND3") ? atoi(getenv("FLVIS_PYR_BAND3")) : 0;
  a.rows_per_band = nout == 1 ? (rows_env > 0 ? rows_env : 4) : nout == 2 ? (rows_env2 > 0 ? rows_env2 : 2) : (rows_env3 > 0 ? rows_env3 : 2);
  const int bands = (hl + a.rows_per_band - 1) / a.rows_per_band;
  const dim3 grid((bands + 3) / 4, S), block(256);
  if (nout == 1) {
    if (copy0) hipLaunchKernelGGL((k_pyr_walk<1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_pyr_walk<1, false>), grid, block, 0, st, a);
  } else if (nout == 2) {
    if (copy0) hipLaunchKernelGGL((k_pyr_walk<2, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_pyr_walk<2, false>), grid, block, 0, st, a);
  } else {
    if (copy0) hipLaunchKernelGGL((k_pyr_walk<3, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_pyr_walk<3, false>), grid, block, 0, st, a);
  }
}

}  // namespace flvis
