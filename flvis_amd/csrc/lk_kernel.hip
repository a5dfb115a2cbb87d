// flvis_amd: batched pyramidal Lucas-Kanade tracker for gfx950 (CDNA4).
//
// Replaces cv::calcOpticalFlowPyrLK as the reference calls it:
//   temporal tracking   src/processing/lkorb_tracking.cpp:64-73   (31x31, maxLevel 10 -> clamped, 30 it, eps 1e-3, USE_INITIAL_FLOW)
//   stereo "matching"   src/processing/camera_frame.cpp:124-128   (same call, img0 -> img1, maxLevel 5 -> clamped)
//
// Mapping: ONE WAVE (64 lanes) per (stream, point); the wave walks the pyramid levels top-down and runs the <=30
// Gauss-Newton iterations in-kernel.  Lane l owns window row (l>>1) and the 16-column half (l&1) of the 31x31 window:
// the interpolated template (I, Ix, Iy as packed int16 pairs) lives in that lane's registers for the whole level.
// Scharr derivatives are computed on the fly from the staged source patch (no derivative image in HBM).
// The search image is cached per level as a 41-row x 52-byte REGION around the current position (window + 4 px margin,
// dword-aligned columns): iterations read their 32x32 window from LDS and only a move of more than the margin re-stages
// it -- no global load and no barrier in a steady-state iteration.  The bilinear interpolation + residual run on packed
// 16-bit pairs (v_perm_b32 to widen two bytes, v_dot2c_i32_i16 for the weights and for the b1/b2 accumulation): ~8
// instructions per pixel.  Sums are exact integers (int32 per lane; the wave total is reduced as 16-bit halves with DPP
// row reductions + 4 readlanes) -> bit-exact vs the oracle.
#include "dev_common.hpp"
#include "img_kernels.hpp"

namespace flvis {

constexpr int LK_WIN = 31;
constexpr int LK_PS = 40;     // template-source patch row stride (bytes); 36 used
constexpr int LK_PROWS = 34;  // rows of the template-source patch
constexpr int LK_RM = 4;      // search-region margin (pixels)
constexpr int LK_RS = 52;     // search-region row stride (bytes): 13 dwords (odd -> conflict-free row walks)
constexpr int LK_RROWS = 33 + 2 * LK_RM;
// template cache (LKParams::tc): one slot per (stream, point); LK_TC_HDR header dwords -- position bits (2), tag (2), mask of the levels
// stored (1), the rest unused -- followed by LK_TC_LVL dwords per level: the 24 template registers (tI, tX, tY: 8 packed pairs each) of
// the 64 lanes as six 1 KB rows (one dwordx4 per lane and row); lane 63, which holds no template (window row 31), carries the level's
// three Hessian sums (int64) in its first six dwords
constexpr int LK_TC_HDR = 64;
constexpr int LK_TC_LVL = 64 * 24;

typedef short lk_s2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) uint32_t lk_gu32;  // a dword in global memory (global_load instead of flat_load)
typedef uint32_t lk_u4 __attribute__((ext_vector_type(4)));  // (a native vector: the non-temporal builtins do not take HIP's uint4 class)
// (by value: __builtin_bit_cast applied directly to a vector-element expression such as q.y reads element 0 with this compiler)
__device__ __forceinline__ lk_s2 lk_as_s2(uint32_t v) { return __builtin_bit_cast(lk_s2, v); }

// patch[r][c] = img(X0 + c, Y0 + r) for r < nrows, c < 36, REFLECT_101 outside the image.  Item i = 9 r + k is dword k of row r; a lane
// takes items lane, lane + 64, ...  When the whole patch lies inside the image (wave-uniform test; nearly every point) there is nothing to
// reflect and nothing to divide: a lane's (row, dword) advances by (7, 1) per trip (64 = 7 * 9 + 1) with one wrap test, its byte offset
// into the image and its LDS address advance by constants, and the address is a uniform base + a 32-bit lane offset.
// (bx, by) = the level's physical REFLECT_101 border: the fast path covers every block that lies inside the bordered storage.
__device__ __forceinline__ bool lk_load_patch(const uint8_t* __restrict__ img, int w, int h, int pitch, int bx, int by, int X0, int Y0,
                                              int nrows, uint8_t* patch) {
  if (X0 >= -bx && Y0 >= -by && X0 + 39 < w + bx && Y0 + nrows <= h + by) {  // (+ 39: the second dword of the last item stays inside the row)
    // ALL loads of the lane are issued before the first one is consumed: one memory round trip for the patch instead of one per trip
    // (the loop form waited for every load before it issued the next -- five dependent round trips of ~1 us under load)
    const lk_gu32* const base = (const lk_gu32*)(img + (ptrdiff_t)Y0 * pitch + (X0 & ~3));
    const int sh = X0 & 3;
    int r = (int)threadIdx.x / 9, k = (int)threadIdx.x - 9 * r;
    unsigned off = (unsigned)(r * pitch + 4 * k);
    static_assert(LK_PROWS * 9 <= 5 * 64, "five trips");
    const int n = nrows * 9;
    uint32_t lo[5], hi[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const bool on = (int)threadIdx.x + 64 * t < n;
      const unsigned o = on ? off : 0u;   // (a lane past the end re-reads item 0 of the patch: in bounds, never stored)
      lo[t] = *(const lk_gu32*)((const __attribute__((address_space(1))) uint8_t*)base + o);
      hi[t] = *(const lk_gu32*)((const __attribute__((address_space(1))) uint8_t*)base + o + 4);
      k += 1;
      const bool wrap = k >= 9;
      k = wrap ? k - 9 : k;
      off += (unsigned)(7 * pitch + 4) + (wrap ? (unsigned)(pitch - 36) : 0u);
    }
    // (every loaded value is "used" here, so the compiler cannot sink the last, predicated trip's loads behind the wait for the others)
#pragma unroll
    for (int t = 0; t < 5; t++) asm volatile("" : "+v"(lo[t]), "+v"(hi[t]));
    r = (int)threadIdx.x / 9;
    k = (int)threadIdx.x - 9 * r;
    unsigned dst = (unsigned)(r * LK_PS + 4 * k);
#pragma unroll
    for (int t = 0; t < 5; t++) {
      if ((int)threadIdx.x + 64 * t < n) *reinterpret_cast<uint32_t*>(patch + dst) = __builtin_amdgcn_alignbyte(hi[t], lo[t], sh);
      k += 1;
      const bool wrap = k >= 9;
      k = wrap ? k - 9 : k;
      dst += (unsigned)(7 * LK_PS + 4) + (wrap ? (unsigned)(LK_PS - 36) : 0u);
    }
    return false;
  }
  for (int i = threadIdx.x; i < nrows * 9; i += 64) {
    int r = i / 9, k = i - r * 9;
    int Y = Y0 + r, X = X0 + 4 * k;
    uint32_t v;
    if (Y >= 0 && Y < h && X >= 0 && X + 3 < w) {
      const uint8_t* row = img + (size_t)Y * pitch;
      int a = X & ~3, sh = X & 3;
      uint32_t lo = *reinterpret_cast<const uint32_t*>(row + a);
      uint32_t hi = sh ? *reinterpret_cast<const uint32_t*>(row + a + 4) : 0u;
      v = __builtin_amdgcn_alignbyte(hi, lo, sh);
    } else {
      const uint8_t* row = img + (size_t)reflect101c(Y, h) * pitch;
      uint32_t b0 = row[reflect101c(X, w)], b1 = row[reflect101c(X + 1, w)], b2 = row[reflect101c(X + 2, w)],
               b3 = row[reflect101c(X + 3, w)];
      v = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    *reinterpret_cast<uint32_t*>(patch + r * LK_PS + 4 * k) = v;
  }
  return true;
}

// region[r][c] = img(X0 + c, Y0 + r), r < LK_RROWS, c < 52, X0 a multiple of 4 (aligned dword loads), REFLECT_101 outside.  Same fast
// path when the region lies inside the image: (row, dword) advances by (4, 12) per trip (64 = 4 * 13 + 12).
__device__ __forceinline__ bool lk_load_region(const uint8_t* __restrict__ img, int w, int h, int pitch, int bx, int by, int X0, int Y0,
                                               uint8_t* region) {
  if (X0 >= -bx && Y0 >= -by && X0 + 51 < w + bx && Y0 + LK_RROWS <= h + by) {
    // nine loads per lane in flight at once, then nine LDS stores (see lk_load_patch)
    const __attribute__((address_space(1))) uint8_t* const base = (const __attribute__((address_space(1))) uint8_t*)(img + (ptrdiff_t)Y0 * pitch + X0);
    int r = (int)threadIdx.x / 13, k = (int)threadIdx.x - 13 * r;
    unsigned off = (unsigned)(r * pitch + 4 * k);
    static_assert(LK_RS == 52, "the region rows are contiguous in LDS: item i sits at byte 4 i");
    constexpr int N = LK_RROWS * 13, T = (N + 63) / 64;
    uint32_t v[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
      const bool on = 64 * (t + 1) <= N || (int)threadIdx.x + 64 * t < N;
      v[t] = *(const lk_gu32*)(base + (on ? off : 0u));
      k += 12;
      const bool wrap = k >= 13;
      k = wrap ? k - 13 : k;
      off += (unsigned)(4 * pitch + 48) + (wrap ? (unsigned)(pitch - 52) : 0u);
    }
#pragma unroll
    for (int t = 0; t < T; t++) asm volatile("" : "+v"(v[t]));
#pragma unroll
    for (int t = 0; t < T; t++)
      if (64 * (t + 1) <= N || (int)threadIdx.x + 64 * t < N) *reinterpret_cast<uint32_t*>(region + 4 * ((int)threadIdx.x + 64 * t)) = v[t];
    return false;
  }
  for (int i = threadIdx.x; i < LK_RROWS * 13; i += 64) {
    int r = i / 13, k = i - r * 13;
    int Y = Y0 + r, X = X0 + 4 * k;
    uint32_t v;
    if (Y >= 0 && Y < h && X >= 0 && X + 3 < w) {
      v = *reinterpret_cast<const uint32_t*>(img + (size_t)Y * pitch + X);
    } else {
      const uint8_t* row = img + (size_t)reflect101c(Y, h) * pitch;
      uint32_t b0 = row[reflect101c(X, w)], b1 = row[reflect101c(X + 1, w)], b2 = row[reflect101c(X + 2, w)],
               b3 = row[reflect101c(X + 3, w)];
      v = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    *reinterpret_cast<uint32_t*>(region + r * LK_RS + 4 * k) = v;
  }
  return true;
}

// bytes K and K+1 of the dword array D, widened to a pair of 16-bit lanes (one v_perm_b32; K is a compile-time constant)
#define LK_PAIR(D, K)                                                                                              \
  ((((K)&3) == 3) ? __builtin_amdgcn_perm((D)[((K) >> 2) + 1], (D)[(K) >> 2], 0x0c040c03u)                         \
                  : __builtin_amdgcn_perm(0u, (D)[(K) >> 2], 0x0c000c00u | (uint32_t)((K)&3) | ((uint32_t)(((K)&3) + 1) << 16)))

// sum of v over the wave as a wave-uniform value: 4 DPP steps give every lane its row-of-16 total, two row broadcasts
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) carry the row totals upwards, lane 63 holds the wave total --
// one readlane instead of four readlanes + three scalar adds (integer sums: any order is exact)
__device__ __forceinline__ int lk_wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15, rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31, rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
// exact 64-bit total of a per-lane int32: 16-bit halves cannot overflow 32 bits over 64 lanes
__device__ __forceinline__ long long lk_wave_sum_wide(int v) {
  const int lo = lk_wave_sum_i32(v & 0xffff), hi = lk_wave_sum_i32(v >> 16);
  return ((long long)hi << 16) + (long long)lo;
}

// Instruction diet of the iteration (round 6).  The LK launches are bound by INSTRUCTION ISSUE -- one instruction of any kind per 4 cycles
// and SIMD, vector, scalar, LDS, wait and no-op alike (profiles/r06_chain_ab.md; SQ counters: 0.44 scalar instructions per vector one) --
// not by latency: what shortens them is fewer instructions per iteration.  -DFLVIS_LK_DIET=0 keeps rounds 3-5's forms.
#ifndef FLVIS_LK_DIET
#define FLVIS_LK_DIET 1
#endif
// a . b + c with c in a SCALAR register (v_dot2_i32_i16, the three-source form): the rounding bias of a bilinear interpolation is a
// constant, and the accumulate-in-place form the compiler picks for the builtin (v_dot2c_i32_i16) needs a v_mov of it per pixel
__device__ __forceinline__ int lk_dot2_bias(lk_s2 a, lk_s2 b, int bias_uniform) {
#if FLVIS_LK_DIET
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(bias_uniform));
  return r;
#else
  return __builtin_amdgcn_sdot2(a, b, bias_uniform, false);
#endif
}
// the exact wave totals of two per-lane int32 as floats, (float)(long long) of each (round to nearest even): the four 16-bit-half chains
// advance together, step by step (every DPP step's operands were written four instructions earlier: no wait states to fill), and the
// totals are put together in double -- hi * 65536 + lo is exact below 2^53, one rounding into float -- on the vector unit in lane 63
// instead of ~13 scalar instructions per value (count leading zeros, shift, sticky bit, convert, scale); two read-lanes instead of four
__device__ __forceinline__ void lk_wave_sum2_f32(int v1, int v2, float& f1, float& f2) {
#if FLVIS_LK_DIET
  int a = v1 & 0xffff, b = v1 >> 16, c = v2 & 0xffff, d = v2 >> 16;
#define LK_STEP4(CTRL, RM)                                           \
  {                                                                  \
    const int ta = __builtin_amdgcn_update_dpp(0, a, CTRL, RM, 0xf, false); \
    const int tb = __builtin_amdgcn_update_dpp(0, b, CTRL, RM, 0xf, false); \
    const int tc = __builtin_amdgcn_update_dpp(0, c, CTRL, RM, 0xf, false); \
    const int td = __builtin_amdgcn_update_dpp(0, d, CTRL, RM, 0xf, false); \
    a += ta, b += tb, c += tc, d += td;                              \
  }
  LK_STEP4(0xB1, 0xf)   // quad_perm [1,0,3,2]
  LK_STEP4(0x4E, 0xf)   // quad_perm [2,3,0,1]
  LK_STEP4(0x141, 0xf)  // row_half_mirror
  LK_STEP4(0x140, 0xf)  // row_mirror
  LK_STEP4(0x142, 0xa)  // row_bcast:15, rows 1 and 3
  LK_STEP4(0x143, 0xc)  // row_bcast:31, rows 2 and 3
#undef LK_STEP4
  const float g1 = (float)__builtin_fma((double)b, 65536.0, (double)a), g2 = (float)__builtin_fma((double)d, 65536.0, (double)c);
  f1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g1), 63));
  f2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g2), 63));
#else
  f1 = (float)lk_wave_sum_wide(v1);
  f2 = (float)lk_wave_sum_wide(v2);
#endif
}

// the exact 64-bit wave totals of three per-lane int32 (the Hessian sums of a template level): six 16-bit-half chains advancing together
// -- a DPP step's operand was written six instructions earlier, so none of the ~30 wait-state no-ops of three reductions one after the
// other (lk_wave_sum_wide x 3)
__device__ __forceinline__ void lk_wave_sum3_wide(int v1, int v2, int v3, long long& s1, long long& s2, long long& s3) {
#if FLVIS_LK_DIET
  int a = v1 & 0xffff, b = v1 >> 16, c = v2 & 0xffff, d = v2 >> 16, e = v3 & 0xffff, f = v3 >> 16;
#define LK_STEP6(CTRL, RM)                                                    \
  {                                                                           \
    const int ta = __builtin_amdgcn_update_dpp(0, a, CTRL, RM, 0xf, false);  \
    const int tb = __builtin_amdgcn_update_dpp(0, b, CTRL, RM, 0xf, false);  \
    const int tc = __builtin_amdgcn_update_dpp(0, c, CTRL, RM, 0xf, false);  \
    const int td = __builtin_amdgcn_update_dpp(0, d, CTRL, RM, 0xf, false);  \
    const int te = __builtin_amdgcn_update_dpp(0, e, CTRL, RM, 0xf, false);  \
    const int tf = __builtin_amdgcn_update_dpp(0, f, CTRL, RM, 0xf, false);  \
    a += ta, b += tb, c += tc, d += td, e += te, f += tf;                     \
  }
  LK_STEP6(0xB1, 0xf)
  LK_STEP6(0x4E, 0xf)
  LK_STEP6(0x141, 0xf)
  LK_STEP6(0x140, 0xf)
  LK_STEP6(0x142, 0xa)
  LK_STEP6(0x143, 0xc)
#undef LK_STEP6
  s1 = ((long long)__builtin_amdgcn_readlane(b, 63) << 16) + (long long)__builtin_amdgcn_readlane(a, 63);
  s2 = ((long long)__builtin_amdgcn_readlane(d, 63) << 16) + (long long)__builtin_amdgcn_readlane(c, 63);
  s3 = ((long long)__builtin_amdgcn_readlane(f, 63) << 16) + (long long)__builtin_amdgcn_readlane(e, 63);
#else
  s1 = lk_wave_sum_wide(v1);
  s2 = lk_wave_sum_wide(v2);
  s3 = lk_wave_sum_wide(v3);
#endif
}

// The interpolated template of one level: I, Ix, Iy of the 31 x 31 window as packed int16 pairs in the lane's registers (lane = window
// row lane >> 1, 16-column half lane & 1) and this lane's share of the three Hessian sums.  Packed 16-bit path: two columns per
// instruction (v_pk_*), the bilinear weights applied with v_dot2c_i32_i16.  Every lane (window rows 0 .. 31: the lanes of row 31 only
// serve their neighbours) computes the Scharr derivatives of ITS row once; the derivatives of row r + 1, which the bilinear
// interpolation also needs, come from the lane two above (same column half) through ds_bpermute -- 18 exchanges on the LDS pipe instead
// of computing every derivative row twice on the VALUs.  INTERIOR: every pixel the stencil is evaluated at lies inside the image.
// Otherwise the derivative image is ZERO outside the image (cv::calcOpticalFlowPyrLK's derivative pyramid has a constant border, the
// intensity pyramid a reflected one): the staged patch holds the reflected intensities, the derivatives of this lane's row are masked
// per column pair before they are used and exchanged (9 masks from one 18-bit column-validity word).
template <bool INTERIOR>
__device__ __forceinline__ void lk_template(const uint8_t* patch, int lane, int ipx, int ipy, int W, int H, lk_s2 wT, lk_s2 wB,
                                            lk_s2 (&tI)[8], lk_s2 (&tX)[8], lk_s2 (&tY)[8], int& a11, int& a12, int& a22) {
  constexpr int W_BITS = 14;
  const int r = lane >> 1, c0 = (lane & 1) * 16;
  uint32_t d[3][5];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < 5; k++) d[j][k] = *reinterpret_cast<const uint32_t*>(patch + (r + j) * LK_PS + c0 + 4 * k);
  uint32_t bits = 0;
  if (!INTERIOR) {
    // window columns c0 .. c0 + 17 of this lane's row: image (ipx + c0 + c, ipy + r)
    const int xs = ipx + c0, Y = ipy + r;
    int lo = -xs, hi = W - xs;
    lo = lo < 0 ? 0 : (lo > 18 ? 18 : lo);
    hi = hi < 0 ? 0 : (hi > 18 ? 18 : hi);
    bits = (Y >= 0 && Y < H && hi > lo) ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
  }
  lk_s2 DX[2][9], DY[2][9];  // (v[2j], v[2j+1]) of window row r + dr
  {
    lk_s2 T0[10], T1[10];
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const lk_s2 A = __builtin_bit_cast(lk_s2, LK_PAIR(d[0], 2 * j)), B = __builtin_bit_cast(lk_s2, LK_PAIR(d[1], 2 * j)),
                  Cc = __builtin_bit_cast(lk_s2, LK_PAIR(d[2], 2 * j));
      T0[j] = (A + Cc) * (short)3 + B * (short)10;
      T1[j] = Cc - A;
    }
    const int up2 = ((lane + 2) & 63) * 4;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      DX[0][j] = T0[j + 1] - T0[j];
      const lk_s2 T1o = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, T1[j + 1]),
                                                                       __builtin_bit_cast(uint32_t, T1[j]), 0x05040302u));
      DY[0][j] = (T1[j + 1] + T1[j]) * (short)3 + T1o * (short)10;
      if (!INTERIOR) {
        const uint32_t m = ((bits >> (2 * j)) & 1u ? 0x0000ffffu : 0u) | ((bits >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u);
        DX[0][j] = __builtin_bit_cast(lk_s2, __builtin_bit_cast(uint32_t, DX[0][j]) & m);
        DY[0][j] = __builtin_bit_cast(lk_s2, __builtin_bit_cast(uint32_t, DY[0][j]) & m);
      }
      DX[1][j] = __builtin_bit_cast(lk_s2, __builtin_amdgcn_ds_bpermute(up2, __builtin_bit_cast(int, DX[0][j])));
      DY[1][j] = __builtin_bit_cast(lk_s2, __builtin_amdgcn_ds_bpermute(up2, __builtin_bit_cast(int, DY[0][j])));
    }
  }
  if (r < LK_WIN) {
    // (intensity rows r, r + 1 of the window are patch rows r + 1, r + 2: d[1], d[2])
#pragma unroll
    for (int c2 = 0; c2 < 8; c2++) {
      int iv[2], ix[2], iy[2];
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int c = 2 * c2 + hh;
        lk_s2 x0, x1, y0, y1;  // (v[c], v[c+1]) of rows r, r+1
        if (hh == 0) {
          x0 = DX[0][c2]; x1 = DX[1][c2]; y0 = DY[0][c2]; y1 = DY[1][c2];
        } else {
          x0 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, DX[0][c2 + 1]), __builtin_bit_cast(uint32_t, DX[0][c2]), 0x05040302u));
          x1 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, DX[1][c2 + 1]), __builtin_bit_cast(uint32_t, DX[1][c2]), 0x05040302u));
          y0 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, DY[0][c2 + 1]), __builtin_bit_cast(uint32_t, DY[0][c2]), 0x05040302u));
          y1 = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, DY[1][c2 + 1]), __builtin_bit_cast(uint32_t, DY[1][c2]), 0x05040302u));
        }
        int ax = lk_dot2_bias(x0, wT, 1 << (W_BITS - 1)), ay = lk_dot2_bias(y0, wT, 1 << (W_BITS - 1));
        int ai = lk_dot2_bias(__builtin_bit_cast(lk_s2, LK_PAIR(d[1], c + 1)), wT, 1 << (W_BITS - 5 - 1));
        ax = __builtin_amdgcn_sdot2(x1, wB, ax, false);
        ay = __builtin_amdgcn_sdot2(y1, wB, ay, false);
        ai = __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_s2, LK_PAIR(d[2], c + 1)), wB, ai, false);
        const bool on = c0 + c < LK_WIN;  // window column 31 of the second half does not exist
        ix[hh] = on ? (ax >> W_BITS) : 0;
        iy[hh] = on ? (ay >> W_BITS) : 0;
        iv[hh] = on ? (ai >> (W_BITS - 5)) : 0;
      }
      tI[c2] = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm((uint32_t)iv[1], (uint32_t)iv[0], 0x05040100u));
      tX[c2] = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm((uint32_t)ix[1], (uint32_t)ix[0], 0x05040100u));
      tY[c2] = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm((uint32_t)iy[1], (uint32_t)iy[0], 0x05040100u));
      a11 = __builtin_amdgcn_sdot2(tX[c2], tX[c2], a11, false);
      a12 = __builtin_amdgcn_sdot2(tX[c2], tY[c2], a12, false);
      a22 = __builtin_amdgcn_sdot2(tY[c2], tY[c2], a22, false);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      tI[c] = lk_s2{0, 0};
      tX[c] = lk_s2{0, 0};
      tY[c] = lk_s2{0, 0};
    }
  }
}

// One level's template from the image: stage the source patch (rows ipy-1 .. ipy+32, columns ipx-1 .. ipx+34), Scharr + bilinear
// interpolation, the three Hessian sums over the wave.  Returns whether the staging took the index-reflecting path.
__device__ __forceinline__ bool lk_template_level(const uint8_t* img, int W, int H, int pitch, int bx, int by, int ipx, int ipy, int iw00,
                                                  int iw01, int iw10, int iw11, uint8_t* patch, int lane, lk_s2 (&tI)[8], lk_s2 (&tX)[8],
                                                  lk_s2 (&tY)[8], long long& iA11, long long& iA12, long long& iA22) {
  __syncthreads();
  const bool slow = lk_load_patch(img, W, H, pitch, bx, by, ipx - 1, ipy - 1, LK_PROWS, patch);
  __syncthreads();
  int a11 = 0, a12 = 0, a22 = 0;
  // every pixel the Scharr stencil is evaluated at lies inside the image -> no border masks (wave-uniform test)
  const bool interior = ipx >= 0 && ipx + 32 <= W - 1 && ipy >= 0 && ipy + 31 <= H - 1;
  const lk_s2 wT = lk_s2{(short)iw00, (short)iw01}, wB = lk_s2{(short)iw10, (short)iw11};
  if (interior)
    lk_template<true>(patch, lane, ipx, ipy, W, H, wT, wB, tI, tX, tY, a11, a12, a22);
  else
    lk_template<false>(patch, lane, ipx, ipy, W, H, wT, wB, tI, tX, tY, a11, a12, a22);
  lk_wave_sum3_wide(a11, a12, a22, iA11, iA12, iA22);
  return slow;
}
// The same as a CALL, for the temporal launch: there nearly every template comes from the cache, and with the computation out of line
// (its own register allocation, results handed over through the caller's stack) the kernel's hot path -- cache loads, region staging,
// iterations -- fits a register budget that lets twice as many waves share a SIMD.
struct LKTmpl {
  uint32_t w[24];
  long long a11, a12, a22;
};
__device__ __noinline__ void lk_template_level_cold(const uint8_t* img, int W, int H, int pitch, int bx, int by, int ipx, int ipy, int iw00,
                                                    int iw01, int iw10, int iw11, uint8_t* patch, int lane, LKTmpl* out) {
  lk_s2 tI[8], tX[8], tY[8];
  long long a11, a12, a22;
  lk_template_level(img, W, H, pitch, bx, by, ipx, ipy, iw00, iw01, iw10, iw11, patch, lane, tI, tX, tY, a11, a12, a22);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    out->w[k] = __builtin_bit_cast(uint32_t, tI[k]);
    out->w[8 + k] = __builtin_bit_cast(uint32_t, tX[k]);
    out->w[16 + k] = __builtin_bit_cast(uint32_t, tY[k]);
  }
  out->a11 = a11;
  out->a12 = a12;
  out->a22 = a22;
}

// one level's templates into a cache slot: the 24 template registers of the 64 lanes as six 1 KB rows; lane 63 (window row 31: no template)
// carries the three Hessian sums in the place of its tI registers
__device__ __forceinline__ void lk_tc_store_level(uint32_t* tc_ptr, int level, int lane, const lk_s2 (&tI)[8], const lk_s2 (&tX)[8],
                                                  const lk_s2 (&tY)[8], long long iA11, long long iA12, long long iA22) {
  lk_u4* dst = reinterpret_cast<lk_u4*>(tc_ptr + LK_TC_HDR + (size_t)level * LK_TC_LVL) + lane;
  uint32_t wI[8];
#pragma unroll
  for (int k = 0; k < 8; k++) wI[k] = __builtin_bit_cast(uint32_t, tI[k]);
  if (lane == 63) {
    wI[0] = (uint32_t)iA11; wI[1] = (uint32_t)((unsigned long long)iA11 >> 32);
    wI[2] = (uint32_t)iA12; wI[3] = (uint32_t)((unsigned long long)iA12 >> 32);
    wI[4] = (uint32_t)iA22; wI[5] = (uint32_t)((unsigned long long)iA22 >> 32);
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    __builtin_nontemporal_store(lk_u4{wI[4 * k], wI[4 * k + 1], wI[4 * k + 2], wI[4 * k + 3]}, dst + 64 * k);
    __builtin_nontemporal_store(lk_u4{__builtin_bit_cast(uint32_t, tX[4 * k]), __builtin_bit_cast(uint32_t, tX[4 * k + 1]),
                                      __builtin_bit_cast(uint32_t, tX[4 * k + 2]), __builtin_bit_cast(uint32_t, tX[4 * k + 3])}, dst + 64 * (2 + k));
    __builtin_nontemporal_store(lk_u4{__builtin_bit_cast(uint32_t, tY[4 * k]), __builtin_bit_cast(uint32_t, tY[4 * k + 1]),
                                      __builtin_bit_cast(uint32_t, tY[4 * k + 2]), __builtin_bit_cast(uint32_t, tY[4 * k + 3])}, dst + 64 * (4 + k));
  }
}

// Base address of one pyramid level for stream s.  The slot choice cur[s] (a global load the compiler may not hoist: memory could have
// changed) is read ONCE per wave and passed in: every level of a pyramid that selects by slot shares one slot array (fill_pyr); ind0 =
// the already loaded base of an indirect level 0.  A level then costs kernel-argument reads only, no dependent global round trip.
__device__ __forceinline__ const uint8_t* lk_level_ptr(const PyrSel& P, int level, int s, int kc, const uint8_t* ind0) {
  const ImgSel& I = P.lvl[level];
  if (I.ind) return (level == 0 ? ind0 : *I.ind) + (size_t)s * P.stride[level];
  return I.b[I.cur ? (kc ^ I.flip) : 0] + (size_t)s * P.stride[level];
}

#ifndef FLVIS_LK_PREFETCH
#define FLVIS_LK_PREFETCH 0  // (build-variant knob; measured: the 24 registers it holds cost more than the round trip it saves)
#endif
#ifndef FLVIS_LK_EARLY_REGION
#define FLVIS_LK_EARLY_REGION 1  // (build-variant knob, round 6: a cached level stages its first search region beside the template loads)
#endif
#ifndef FLVIS_LK_WAVES
#define FLVIS_LK_WAVES 4  // (build-variant knob: waves per SIMD the register allocation aims at)
#endif
#ifndef FLVIS_LK_WAVES_T
#define FLVIS_LK_WAVES_T 4  // ... of the temporal launch (its template computation is out of line)
#endif
// 4 waves per SIMD (<= 128 VGPRs): this kernel is latency-bound (PMC: VALU busy ~20%), occupancy is what pays
// ROLE names the launch in the profiles and fixes what the template cache may do: 0 the stand-alone entry point (no cache), 1 the
// tracker's temporal launch (may take templates from the cache), 2 its stereo launch (may store them), 4 the stereo launch fed by
// k_lk_templates_ahead (tc_mode 3: takes the templates that kernel made for the tracked landmarks, computes and stores the others)
#ifdef FLVIS_LK_UTIL
// (build variant, scripts/lk_util.py: how full the chip is during an LK launch.  Every wave that tracks a point leaves its start and end
// time (wall clock, 100 MHz) in a slot of its own: [launch kind: role % 3][the launch's slot, 8 of them][wave][start, end])
__device__ unsigned long long g_lk_wt[3][8][32768][2];
#define LK_UTIL_BEGIN const unsigned long long lk_t0_ = wall_clock64(); const unsigned lk_widx_ = (blockIdx.y * gridDim.x + blockIdx.x) & 32767u;
#define LK_UTIL_END(ROLE)                                                              \
  if (threadIdx.x == 0) {                                                              \
    unsigned long long* u_ = g_lk_wt[(ROLE) % 3][prm.dbg_slot & 7][lk_widx_];          \
    u_[0] = lk_t0_;                                                                    \
    u_[1] = wall_clock64();                                                            \
  }
#else
#define LK_UTIL_BEGIN
#define LK_UTIL_END(ROLE)
#endif
template <int ROLE>
__device__ __forceinline__ void lk_track_body(const PyrSel& prev, const PyrSel& next, const float* __restrict__ prev_pts,
                                              float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                              const int* __restrict__ count, int nmax, const LKParams& prm,
                                              const int* __restrict__ active) {
  // XCD-aware workgroup -> (stream, point) map: workgroup b is observed to run on XCD b % 8 (each XCD has its own 4 MiB
  // L2), so a stream's workgroups are renumbered onto one XCD and its two pyramids (~0.8 MB) are fetched from HBM once
  // instead of once per XCD.  A bijection whenever the grid size is a multiple of 8; speed only, never correctness.
  int s = blockIdx.y, bx = blockIdx.x;
  {
    const int G = gridDim.x, N = G * gridDim.y;
    if ((N & 7) == 0) {
      const int L = bx + G * s;
      const int Lp = (L & 7) * (N >> 3) + (L >> 3);
      s = Lp / G;
      bx = Lp - s * G;
    }
  }
  if (active && !active[s]) return;
  int n = count[s];
  if (n > nmax) n = nmax;
  if (bx >= n) return;
  LK_UTIL_BEGIN
  __shared__ __attribute__((aligned(16))) uint8_t patch[LK_RROWS * LK_RS + 12];  // template patch, then search region
  const int lane = threadIdx.x;
  int kc_prev = 0, kc_next = 0;
#pragma unroll
  for (int l = LK_MAX_LEVELS - 1; l >= 0; l--) {  // (the slot array of the lowest level that has one; all levels share it)
    if (l <= prev.levels && prev.lvl[l].cur) kc_prev = prev.lvl[l].cur[s];
    if (l <= next.levels && next.lvl[l].cur) kc_next = next.lvl[l].cur[s];
  }
  const uint8_t* const ind_prev0 = prev.lvl[0].ind ? *prev.lvl[0].ind : nullptr;
  const uint8_t* const ind_next0 = next.lvl[0].ind ? *next.lvl[0].ind : nullptr;
  const int r = lane >> 1;
  const int c0 = (lane & 1) * 16;
  const int W_BITS = 14;
  // (epsilon^2 in a VECTOR register pair: the scalar registers are oversubscribed by the two pyramids' argument blocks, and the compiler
  // re-read this kernel argument with an s_load + s_waitcnt in front of the convergence test of EVERY iteration)
  double eps2 = prm.eps2;
#if FLVIS_LK_DIET
  if (ROLE == 1 || ROLE == 4) asm volatile("" : "+v"(eps2));  // (the launches with registers to spare: 121 of 128; the others are at 128)
#endif
  const float FLT_SCALE = 1.f / (1 << 20);
  const float halfWin = (LK_WIN - 1) * 0.5f;

  for (int pb = bx; pb < n; pb += gridDim.x) {
    // (order of dispatch: a launch ends one wave's duration after its last workgroup has started, so the long points should start first.
    //  prm.order 1: the stream's points from the last to the first -- a frame's new landmarks, which the stereo matcher has no depth to
    //  start from, are appended behind the tracked ones)
    const int p = prm.order == 1 ? n - 1 - pb : pb;
    const size_t pi = ((size_t)s * nmax + p) * 2;
    const float ppx0 = prev_pts[pi], ppy0 = prev_pts[pi + 1];
    float nx = next_pts[pi], ny = next_pts[pi + 1];
    int st = 1;
    // template cache: the slot this point stores its templates in (stereo matcher), or the slot it may take them from (temporal
    // tracker: valid if it was written for this very position of this very image)
    uint32_t* tc_ptr = nullptr;
    bool tc_store = false, tc_hit = false;
    uint32_t tc_mask = 0;
    if (ROLE == 2 && prm.tc_mode == 1) {
      if (p < prm.tc_cap) {
        tc_store = true;
        tc_ptr = prm.tc + ((size_t)s * prm.tc_cap + p) * prm.tc_stride;
      }
    } else if ((ROLE == 1 && prm.tc_mode == 2) || (ROLE == 4 && prm.tc_mode == 3)) {
      // the caller has compared the slot's header (position bits, tag) with this point: code = slot | (mask of stored levels << 16), or -1;
      // tc_mode 3: or -(slot + 2), the slot a point without templates stores its own in
      const int code = __builtin_amdgcn_readfirstlane(prm.tc_slot[(size_t)s * nmax + p]);
      if (code >= 0 && (code & 0xffff) < prm.tc_cap) {
        tc_ptr = prm.tc + ((size_t)s * prm.tc_cap + (code & 0xffff)) * prm.tc_stride;
        tc_hit = true;
        tc_mask = (uint32_t)code >> 16;
      } else if (ROLE == 4 && code <= -2 && -code - 2 < prm.tc_cap) {
        tc_ptr = prm.tc + ((size_t)s * prm.tc_cap + (-code - 2)) * prm.tc_stride;
        tc_store = true;
      }
    }
    lk_u4 pq[6] = {};  // templates of level pq_level, in flight or arrived (-1: none)
    int pq_level = -1;
    for (int level = prev.levels; level >= 0; level--) {
      const float sc = (float)(1. / (1 << level));
      float ppx = ppx0 * sc, ppy = ppy0 * sc;
      float npx, npy;
      if (level == prev.levels) {
        if (prm.use_initial) {
          npx = nx * sc;
          npy = ny * sc;
        } else {
          npx = ppx;
          npy = ppy;
        }
      } else {
        npx = nx * 2.f;
        npy = ny * 2.f;
      }
      nx = npx;
      ny = npy;
      const int W = prev.w[level], H = prev.h[level];
      ppx -= halfWin;
      ppy -= halfWin;
      const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
      if (ipx < -LK_WIN || ipx >= W || ipy < -LK_WIN || ipy >= H) {
        if (level == 0) st = 0;
        continue;
      }
      float a = ppx - (float)ipx, b = ppy - (float)ipy;
      int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
      int iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
      int iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

      // ---- template.  Either from the cache (the stereo matcher of the previous frame made this very template: same image, same
      // position), or computed: stage source patch rows ipy-1..ipy+32, cols ipx-1..ipx+34, Scharr + bilinear interpolation
      lk_s2 tI[8], tX[8], tY[8];  // pixel pairs (c, c+1)
      long long iA11, iA12, iA22;
      const bool cached = (ROLE == 1 || ROLE == 4) && tc_hit && ((tc_mask >> level) & 1u);
      // (the launches that may take templates from the cache look at the search image before the template stage: see below)
      constexpr bool kEarlyRegion = (ROLE == 1 || ROLE == 4) && FLVIS_LK_EARLY_REGION;
      int JW = 0, JH = 0;
      const uint8_t* Jimg = nullptr;
      if (kEarlyRegion) {
        JW = next.w[level], JH = next.h[level];
        Jimg = lk_level_ptr(next, level, s, kc_next, ind_next0);
      }
      bool region_ok = false;
      int RX0 = 0, RY0 = 0;
      if (cached) {
        if (prm.stats_tc && lane == 0) atomicAdd(&prm.stats_tc[0], 1ull);
        if (pq_level != level) {  // (the top level, or the level above did not get this far)
          const lk_u4* src = reinterpret_cast<const lk_u4*>(tc_ptr + LK_TC_HDR + (size_t)level * LK_TC_LVL) + lane;
#pragma unroll
          for (int k = 0; k < 6; k++) pq[k] = __builtin_nontemporal_load(src + 64 * k);
        }
        if (kEarlyRegion) {
        // the search region of the level's FIRST iteration, staged while the templates travel: its position is the level's start position,
        // known before the templates are (a cached level needs no source patch, so the LDS buffer is free) -- one memory round trip per
        // level instead of two.  Same region, same bytes as the iteration would stage itself.
        {
          const int einx = (int)floorf(npx - halfWin), einy = (int)floorf(npy - halfWin);
          if (!(einx < -LK_WIN || einx >= JW || einy < -LK_WIN || einy >= JH)) {
            RX0 = (einx - LK_RM) & ~3;
            RY0 = einy - LK_RM;
            __syncthreads();
            const bool slow = lk_load_region(Jimg, JW, JH, next.pitch[level], next.bx[level], next.by[level], RX0, RY0, patch);
            __syncthreads();
            if (prm.stats_tc && slow && lane == 0) atomicAdd(&prm.stats_tc[2], 1ull);
            region_ok = true;
          }
        }
        }
        lk_u4 q[6];
#pragma unroll
        for (int k = 0; k < 6; k++) q[k] = pq[k];
        pq_level = -1;
#if FLVIS_LK_PREFETCH
        // the next level's templates are requested now: they travel while this level iterates, together with the search region's loads
        // (one memory round trip per level instead of two)
        if (level > 0 && ((tc_mask >> (level - 1)) & 1u)) {
          const lk_u4* src = reinterpret_cast<const lk_u4*>(tc_ptr + LK_TC_HDR + (size_t)(level - 1) * LK_TC_LVL) + lane;
#pragma unroll
          for (int k = 0; k < 6; k++) pq[k] = __builtin_nontemporal_load(src + 64 * k);
          pq_level = level - 1;
        }
#endif
        // (the three Hessian sums ride in lane 63's registers: the lanes of window row 31 hold no template)
        iA11 = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)q[0].y, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)q[0].x, 63));
        iA12 = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)q[0].w, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)q[0].z, 63));
        iA22 = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)q[1].y, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)q[1].x, 63));
#pragma unroll
        for (int k = 0; k < 2; k++) {
          tI[4 * k + 0] = lk_as_s2(q[k].x); tI[4 * k + 1] = lk_as_s2(q[k].y);
          tI[4 * k + 2] = lk_as_s2(q[k].z); tI[4 * k + 3] = lk_as_s2(q[k].w);
          tX[4 * k + 0] = lk_as_s2(q[2 + k].x); tX[4 * k + 1] = lk_as_s2(q[2 + k].y);
          tX[4 * k + 2] = lk_as_s2(q[2 + k].z); tX[4 * k + 3] = lk_as_s2(q[2 + k].w);
          tY[4 * k + 0] = lk_as_s2(q[4 + k].x); tY[4 * k + 1] = lk_as_s2(q[4 + k].y);
          tY[4 * k + 2] = lk_as_s2(q[4 + k].z); tY[4 * k + 3] = lk_as_s2(q[4 + k].w);
        }
      } else if (ROLE == 1 || ROLE == 4) {
        LKTmpl T;
        lk_template_level_cold(lk_level_ptr(prev, level, s, kc_prev, ind_prev0), W, H, prev.pitch[level], prev.bx[level], prev.by[level], ipx, ipy, iw00,
                               iw01, iw10, iw11, patch, lane, &T);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          tI[k] = lk_as_s2(T.w[k]);
          tX[k] = lk_as_s2(T.w[8 + k]);
          tY[k] = lk_as_s2(T.w[16 + k]);
        }
        iA11 = T.a11;
        iA12 = T.a12;
        iA22 = T.a22;
        if (ROLE == 4 && tc_store) {
          lk_tc_store_level(tc_ptr, level, lane, tI, tX, tY, iA11, iA12, iA22);
          tc_mask |= 1u << level;
        }
      } else {
        const bool slow = lk_template_level(lk_level_ptr(prev, level, s, kc_prev, ind_prev0), W, H, prev.pitch[level], prev.bx[level], prev.by[level], ipx,
                                            ipy, iw00, iw01, iw10, iw11, patch, lane, tI, tX, tY, iA11, iA12, iA22);
        if (prm.stats_tc && slow && lane == 0) atomicAdd(&prm.stats_tc[1], 1ull);
#ifndef FLVIS_LK_NO_TC_STORE  // (timing-only build variant: is the stereo launch bound by its template-cache stores?  the next frame's temporal
                              //  launch then finds no templates; results unchanged -- profiles/r05_lk_ab.md)
        if (ROLE == 2 && tc_store) {
          lk_tc_store_level(tc_ptr, level, lane, tI, tX, tY, iA11, iA12, iA22);
          tc_mask |= 1u << level;
        }
#endif
      }
      const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      const float minEig = __fdiv_rn(A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12),
                                     (float)(2 * LK_WIN * LK_WIN));
      if (minEig < prm.min_eig || D < 1.1920929e-07f) {
        if (level == 0) st = 0;
        continue;
      }
      D = __fdiv_rn(1.f, D);
      npx -= halfWin;
      npy -= halfWin;
      float pdx = 0.f, pdy = 0.f;
      if (!kEarlyRegion) {
        JW = next.w[level], JH = next.h[level];
        Jimg = lk_level_ptr(next, level, s, kc_next, ind_next0);
      }
      int iters_run = 0;
      for (int j = 0; j < prm.max_iter; j++) {
        iters_run = j + 1;
        const int inx = (int)floorf(npx), iny = (int)floorf(npy);
        if (inx < -LK_WIN || inx >= JW || iny < -LK_WIN || iny >= JH) {
          if (level == 0) st = 0;
          break;
        }
        a = npx - (float)inx;
        b = npy - (float)iny;
        iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
        iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
        iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        if (!region_ok || inx < RX0 || inx - RX0 > 15 || iny < RY0 || iny - RY0 > 2 * LK_RM) {
          RX0 = (inx - LK_RM) & ~3;
          RY0 = iny - LK_RM;
          __syncthreads();
          const bool slow = lk_load_region(Jimg, JW, JH, next.pitch[level], next.bx[level], next.by[level], RX0, RY0, patch);
          __syncthreads();
          if (prm.stats_tc && slow && lane == 0) atomicAdd(&prm.stats_tc[2], 1ull);
          region_ok = true;
        }
        int b1 = 0, b2 = 0;
        if (r < LK_WIN) {
          const int bx = inx - RX0 + c0, sh = bx & 3;
          const uint8_t* rp = patch + (iny - RY0 + r) * LK_RS + (bx & ~3);
          uint32_t e[2][5];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            uint32_t dd[6];
#pragma unroll
            for (int k = 0; k < 6; k++) dd[k] = *reinterpret_cast<const uint32_t*>(rp + q * LK_RS + 4 * k);
#pragma unroll
            for (int k = 0; k < 5; k++) e[q][k] = __builtin_amdgcn_alignbyte(dd[k + 1], dd[k], sh);
          }
          const lk_s2 wT = lk_s2{(short)iw00, (short)iw01}, wB = lk_s2{(short)iw10, (short)iw11};
#pragma unroll
          for (int c2 = 0; c2 < 8; c2++) {
            int iv[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
              const int c = 2 * c2 + hh;
              const uint32_t pt = LK_PAIR(e[0], c), pb = LK_PAIR(e[1], c);
              int acc = lk_dot2_bias(__builtin_bit_cast(lk_s2, pt), wT, 1 << (W_BITS - 5 - 1));
              acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_s2, pb), wB, acc, false);
              iv[hh] = acc >> (W_BITS - 5);
            }
            const lk_s2 ivp = __builtin_bit_cast(lk_s2, __builtin_amdgcn_perm((uint32_t)iv[1], (uint32_t)iv[0], 0x05040100u));
            const lk_s2 diff = ivp - tI[c2];
            // columns >= 31 have tX = tY = 0, so they add nothing
            b1 = __builtin_amdgcn_sdot2(diff, tX[c2], b1, false);
            b2 = __builtin_amdgcn_sdot2(diff, tY[c2], b2, false);
          }
        }
        float sb1, sb2;
        lk_wave_sum2_f32(b1, b2, sb1, sb2);
        const float fb1 = sb1 * FLT_SCALE, fb2 = sb2 * FLT_SCALE;
        const float ddx = (A12 * fb2 - A22 * fb1) * D;
        const float ddy = (A12 * fb1 - A11 * fb2) * D;
        npx += ddx;
        npy += ddy;
        nx = npx + halfWin;
        ny = npy + halfWin;
        if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= eps2) break;
        if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
          nx -= ddx * 0.5f;
          ny -= ddy * 0.5f;
          break;
        }
        pdx = ddx;
        pdy = ddy;
      }
      if (prm.stats && lane == 0) {
        atomicAdd(&prm.stats[2 * level], (unsigned long long)iters_run);
        atomicAdd(&prm.stats[2 * level + 1], 1ull);
      }
      if (st && level == 0) {  // error stage of calcOpticalFlowPyrLK: final window must still start inside
        const float fx = nx - halfWin, fy = ny - halfWin;
        const int inx = (int)floorf(fx), iny = (int)floorf(fy);
        if (inx < -LK_WIN || inx >= JW || iny < -LK_WIN || iny >= JH) st = 0;
      }
    }
    if (lane == 0) {
      next_pts[pi] = nx;
      next_pts[pi + 1] = ny;
      status[(size_t)s * nmax + p] = (uint8_t)st;
      if ((ROLE == 2 || ROLE == 4) && tc_store) {
        tc_ptr[0] = __float_as_uint(ppx0);
        tc_ptr[1] = __float_as_uint(ppy0);
        *reinterpret_cast<long long*>(tc_ptr + 2) = prm.tc_tag[s];
        tc_ptr[4] = tc_mask;
      }
    }
  }
  LK_UTIL_END(ROLE)
}

// the three launches (named apart in the profiles; the register budget is per launch kind)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FLVIS_LK_WAVES, FLVIS_LK_WAVES))) void k_lk_track(
    PyrSel prev, PyrSel next, const float* __restrict__ prev_pts, float* __restrict__ next_pts, uint8_t* __restrict__ status,
    const int* __restrict__ count, int nmax, LKParams prm, const int* __restrict__ active) {
  lk_track_body<0>(prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FLVIS_LK_WAVES_T, FLVIS_LK_WAVES_T))) void k_lk_track_temporal(
    PyrSel prev, PyrSel next, const float* __restrict__ prev_pts, float* __restrict__ next_pts, uint8_t* __restrict__ status,
    const int* __restrict__ count, int nmax, LKParams prm, const int* __restrict__ active) {
  lk_track_body<1>(prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FLVIS_LK_WAVES, FLVIS_LK_WAVES))) void k_lk_track_stereo(
    PyrSel prev, PyrSel next, const float* __restrict__ prev_pts, float* __restrict__ next_pts, uint8_t* __restrict__ status,
    const int* __restrict__ count, int nmax, LKParams prm, const int* __restrict__ active) {
  lk_track_body<2>(prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FLVIS_LK_WAVES_T, FLVIS_LK_WAVES_T))) void k_lk_track_stereo_fed(
    PyrSel prev, PyrSel next, const float* __restrict__ prev_pts, float* __restrict__ next_pts, uint8_t* __restrict__ status,
    const int* __restrict__ count, int nmax, LKParams prm, const int* __restrict__ active) {
  lk_track_body<4>(prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
}

// The templates of a frame's tracked landmarks, made AHEAD of the stereo matcher (round 6).  recover3DPts_c_FromStereo's LK call
// (camera_frame.cpp:124-128) takes its templates at the landmarks' pixels in the frame's left image; for a landmark the temporal tracker
// has just followed into this frame that pixel is known as soon as LKORBTracking's optical flow is (lkorb_tracking.cpp:64-73), ~0.4 ms
// before the stereo matcher runs -- and the kernels in between (the two RANSACs, the pose optimisation, the outlier filter) are one
// workgroup per stream on a quarter of the CUs.  This kernel computes those templates there, on a low-priority stream of its own, into
// the cache slots the stereo launch (ROLE 4) and the next frame's temporal launch read: slot j = the survivor's index in the frame
// (k_track_collect), header = position bits, frame id, mask of the levels stored.  Same code, same arithmetic as the stereo launch's own
// template stage (lk_template_level): the results of the tracker do not change by a bit.  One wave per (stream, point), XCD-aware map.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FLVIS_LK_WAVES, FLVIS_LK_WAVES))) void k_lk_templates_ahead(
    PyrSel img, const float* __restrict__ pts, const int* __restrict__ count, int nmax, uint32_t* __restrict__ tc, int tc_cap, int tc_stride,
    const long long* __restrict__ tag) {
  int s = blockIdx.y, bx = blockIdx.x;
  {
    const int G = gridDim.x, N = G * gridDim.y;
    if ((N & 7) == 0) {
      const int L = bx + G * s;
      const int Lp = (L & 7) * (N >> 3) + (L >> 3);
      s = Lp / G;
      bx = Lp - s * G;
    }
  }
  int n = count[s];
  if (n > nmax) n = nmax;
  if (n > tc_cap) n = tc_cap;
  if (bx >= n) return;
  __shared__ __attribute__((aligned(16))) uint8_t patch[LK_RROWS * LK_RS + 12];
  const int lane = threadIdx.x;
  int kc = 0;
#pragma unroll
  for (int l = LK_MAX_LEVELS - 1; l >= 0; l--)
    if (l <= img.levels && img.lvl[l].cur) kc = img.lvl[l].cur[s];
  const uint8_t* const ind0 = img.lvl[0].ind ? *img.lvl[0].ind : nullptr;
  const int W_BITS = 14;
  const float halfWin = (LK_WIN - 1) * 0.5f;
  for (int p = bx; p < n; p += gridDim.x) {
    const size_t pi = ((size_t)s * nmax + p) * 2;
    const float ppx0 = pts[pi], ppy0 = pts[pi + 1];
    uint32_t* const tc_ptr = tc + ((size_t)s * tc_cap + p) * tc_stride;
    uint32_t mask = 0;
    for (int level = img.levels; level >= 0; level--) {
      // (the position arithmetic of lk_track_body, operation for operation)
      const float sc = (float)(1. / (1 << level));
      float ppx = ppx0 * sc, ppy = ppy0 * sc;
      const int W = img.w[level], H = img.h[level];
      ppx -= halfWin;
      ppy -= halfWin;
      const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
      if (ipx < -LK_WIN || ipx >= W || ipy < -LK_WIN || ipy >= H) continue;
      const float a = ppx - (float)ipx, b = ppy - (float)ipy;
      const int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
      const int iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
      const int iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
      const int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      lk_s2 tI[8], tX[8], tY[8];
      long long iA11, iA12, iA22;
      lk_template_level(lk_level_ptr(img, level, s, kc, ind0), W, H, img.pitch[level], img.bx[level], img.by[level], ipx, ipy, iw00, iw01, iw10,
                        iw11, patch, lane, tI, tX, tY, iA11, iA12, iA22);
      lk_tc_store_level(tc_ptr, level, lane, tI, tX, tY, iA11, iA12, iA22);
      mask |= 1u << level;
    }
    if (lane == 0) {
      tc_ptr[0] = __float_as_uint(ppx0);
      tc_ptr[1] = __float_as_uint(ppy0);
      *reinterpret_cast<long long*>(tc_ptr + 2) = tag[s];
      tc_ptr[4] = mask;
    }
  }
}

void launch_lk_templates_ahead(hipStream_t st, const PyrSel& img, const float* pts, const int* count, int nmax, int S, uint32_t* tc, int tc_cap,
                               int tc_stride, const long long* tag, int max_pts) {
  int gx = nmax < 512 ? nmax : 512;
  if (max_pts > 0 && max_pts < gx) gx = (max_pts + 7) & ~7;
  if (gx > nmax) gx = nmax;
  hipLaunchKernelGGL(k_lk_templates_ahead, dim3(gx, S), dim3(64), 0, st, img, pts, count, nmax, tc, tc_cap, tc_stride, tag);
}

int lk_tc_slot_dwords(int levels) { return LK_TC_HDR + (levels + 1) * LK_TC_LVL; }

void launch_lk_track(hipStream_t st, const PyrSel& prev, const PyrSel& next, const float* prev_pts, float* next_pts,
                     uint8_t* status, const int* count, int nmax, int S, LKParams prm, const int* active, int max_pts, int role) {
  // one wave per point; the kernel strides by the grid width, so any width is correct.  Sized by the caller's bound on
  // the point count (the tracker holds <= 16 regions x max_region_feature_num landmarks), rounded so that the XCD-aware
  // renumbering stays a bijection (grid size a multiple of 8)
  int gx = nmax < 512 ? nmax : 512;
  if (max_pts > 0 && max_pts < gx) gx = (max_pts + 7) & ~7;
  if (gx > nmax) gx = nmax;
  if (role == 1)
    hipLaunchKernelGGL(k_lk_track_temporal, dim3(gx, S), dim3(64), 0, st, prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
  else if (role == 2)
    hipLaunchKernelGGL(k_lk_track_stereo, dim3(gx, S), dim3(64), 0, st, prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
  else if (role == 4)
    hipLaunchKernelGGL(k_lk_track_stereo_fed, dim3(gx, S), dim3(64), 0, st, prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
  else
    hipLaunchKernelGGL(k_lk_track, dim3(gx, S), dim3(64), 0, st, prev, next, prev_pts, next_pts, status, count, nmax, prm, active);
}

}  // namespace flvis

#ifdef FLVIS_LK_UTIL
extern "C" int flvis_debug_lk_util(unsigned long long* out, int reset) {  // out: 3 * 8 * 32768 * 2 values
  const size_t bytes = sizeof(unsigned long long) * 3 * 8 * 32768 * 2;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(flvis::g_lk_wt), bytes) != hipSuccess) return -1;
  if (reset) {
    void* dp = nullptr;
    if (hipGetSymbolAddress(&dp, HIP_SYMBOL(flvis::g_lk_wt)) != hipSuccess || hipMemset(dp, 0, bytes) != hipSuccess) return -1;
  }
  return 0;
}
#endif
