// flvis_amd: corner response of cv::goodFeaturesToTrack (cornerMinEigenVal, blockSize 3, Sobel 3; feature_dem.cpp:160,221) as a
// WAVE WALK: one wave64 owns a strip of 64 image columns (one column per lane) and walks down the rows; everything a pixel needs
// from its left / right neighbour comes from the neighbouring LANE (ds_bpermute / DPP), everything it needs from the rows above
// stays in registers.  No LDS tile, no barrier, no halo recomputation inside a strip:
//   image byte -> horizontal difference / smoothing (packed 16 bit) -> Sobel pair of the row above -> the three products
//   fx*fx, fx*fy, fy*fy ONCE per pixel -> 3x3 sums in the raster order of the one-pixel code (bit-identical floats) ->
//   smaller eigenvalue -> 3x3 local maxima of the row above that -> candidate keys + the stream's maximum.
// Every lane carries TWO independent pixels -- the same column in two row chunks of the strip -- so that all float work of the
// sums is packed (v_pk_add_f32 / v_pk_mul_f32: two pixels per instruction).  Same outputs as k_eig_cand (img_kernels.hip): the
// per-stream maximum (ordered bits) and the 3x3 local maxima as sort keys ~((ordered(value) << 32) | pixel offset); the order of
// the keys in the list is irrelevant (k_gftt_pick sorts them).
//
// Geometry.  Strip k covers image columns X = 60 k - 2 + lane; lanes 2 .. 61 produce outputs (a response needs the Sobel pairs
// of lanes +-1, a local maximum the responses of lanes +-1; lanes 0 and 63 fetch the image byte beyond the strip with a second
// load, so the Sobel pairs are valid on all 64 lanes).  A wave handles two chunks of R rows: rows [2 c R, 2 c R + R) in the low
// half and the R rows below in the high half of every packed register.  Producing row y needs image rows y - 3 .. y + 3, so a
// chunk costs R + 7 steps.
//
// Borders.  Positions outside the image are reflected (REFLECT_101) twice in the reference: the covariance maps by boxFilter, the
// image by Sobel.  A lane (row) outside the image loads the reflected column (row); the Sobel pair it then computes is the pair
// of the reflected position except for the sign of the derivative across the border (fx for a column, fy for a row), which is
// folded into the scale factor of that lane (row).
#include <hip/hip_runtime.h>

#include "dev_common.hpp"
#include "img_kernels.hpp"

namespace flvis {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr int EW_OUT = 60;      // output columns per strip
constexpr int EW_LAG = 7;       // steps before the first output row of a chunk
constexpr int EW_KEYS = 384;    // per-wave key buffer in LDS
constexpr int EW_PF = 3;        // image rows in flight ahead of the step that consumes them

__device__ __forceinline__ float bperm_f(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ f2 bperm_f2(int addr, f2 v) { return f2{bperm_f(addr, v.x), bperm_f(addr, v.y)}; }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int ew_rank(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// v_max_f32 as is: fmaxf() first canonicalises both operands (two more instructions per call) because it cannot know that the
// values are never signalling NaNs; the responses are finite
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}


// Correctly rounded sqrtf (the oracle calls the host's IEEE sqrtf) for the arguments that occur here, x = (a - c)^2 + b^2: x is 0 or
// lies in [2^-96, 4).  (Every product fx*fx, fx*fy, fy*fy is 0 or at least (1/3060)^2 > 2^-24 in magnitude, hence a multiple of 2^-47;
// so are their sums, a and c are multiples of 2^-48, and a non-zero x is at least 2^-96.)  The compiler's sqrtf for
// -fhip-fp32-correctly-rounded-divide-sqrt is this sequence -- v_sqrt_f32 (1 ulp), then the neighbours s -+ 1 ulp tested with exact
// FMA residuals -- wrapped in a rescaling for arguments below 2^-96 and a fix-up for 0 / inf / NaN, seven more instructions that
// cannot trigger here: v_sqrt_f32(0) = 0 survives both tests (the residuals are NaN and -0).  flvis_hip_debug_sqrt_check compares the
// two over every float of the domain (tests/test_gpu_image.py).
__device__ __forceinline__ float ew_sqrt_pos(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  float r = rm <= 0.f ? sm : s;
  r = rp > 0.f ? sp : r;
  return r;
}
__device__ __forceinline__ f2 sqrt2(f2 v) { return f2{ew_sqrt_pos(v.x), ew_sqrt_pos(v.y)}; }

// the two square roots on every float of [first_bits, first_bits + n): number of arguments where they differ (test aid)
__global__ void k_sqrt_check(unsigned first_bits, unsigned n, unsigned long long* __restrict__ mismatches) {
  unsigned long long bad = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float x = __uint_as_float(first_bits + i);
    bad += __float_as_uint(ew_sqrt_pos(x)) != __float_as_uint(sqrtf(x));
  }
  if (bad) atomicAdd(mismatches, bad);
}
void launch_sqrt_check(hipStream_t st, unsigned first_bits, unsigned n, unsigned long long* mismatches) {
  hipLaunchKernelGGL(k_sqrt_check, dim3(2048), dim3(256), 0, st, first_bits, n, mismatches);
}

struct EwState {
  // two previous image rows: horizontal difference I(x+1) - I(x-1) and smoothing I(x-1) + 2 I(x) + I(x+1), both halves packed
  s2 hd[3], hs[3];
  // products of the two newest Sobel rows: [row slot][fx fx, fx fy, fy fy][left lane, own, right lane]; S3: the three-term row
  // sum (left + own) + right of the row before them
  f2 T[2][3][3], S3[3];
  f2 E[3];     // the three newest response rows
  f2 vmax;     // running maximum of the responses of this wave's own output pixels
  unsigned pix[EW_PF][2];  // image bytes in flight: [slot][own column | halo column], low half | high half << 16
};

struct EwConst {
  __amdgpu_buffer_rsrc_t img;  // buffer descriptor of the stream's image: loads are buffer_load_ubyte v, lane offset, rsrc, row offset
  int w, h, pitch;
  int y0[2], y1[2];       // output rows of the two halves
  int col_own, col_halo;  // per lane: byte offsets of the reflected own column and of the halo column lanes 0 / 63 fetch
  int addr_l, addr_r;     // ds_bpermute addresses of the left / right lane
  bool lane_first, lane_last, out_col;
  float nms_floor;        // per lane: the smallest response that can be a local maximum here: the smallest positive float on the
                          // lanes that produce candidates (value > 0), +inf on the others (never)
  float sx;               // per lane: +-scale (minus where the column is a reflected one)
  int x;
};

// image row `iy` (virtual: reflected into the image) of both halves -> the packed bytes of this lane's own and halo column
__device__ __forceinline__ void ew_load(const EwConst& c, int iy0, int iy1, unsigned (&pix)[2]) {
  const int r0 = reflect101c(iy0, c.h) * c.pitch, r1 = reflect101c(iy1, c.h) * c.pitch;  // wave-uniform row offsets (scalar)
  const unsigned a0 = __builtin_amdgcn_raw_buffer_load_b8(c.img, c.col_own, r0, 0), a1 = __builtin_amdgcn_raw_buffer_load_b8(c.img, c.col_own, r1, 0),
                 b0 = __builtin_amdgcn_raw_buffer_load_b8(c.img, c.col_halo, r0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b8(c.img, c.col_halo, r1, 0);
  pix[0] = a0 | (a1 << 16);
  pix[1] = b0 | (b1 << 16);
}

template <int PH>  // step number mod 6: all register slots are compile-time
__device__ __forceinline__ void ew_step(const EwConst& c, EwState& st, int s, float* lval, unsigned* loff, int& nk) {
  constexpr int H0 = PH % 3, H1 = (PH + 1) % 3, H2 = (PH + 2) % 3;  // hd / hs slots: rows i-2, i-1, i (H2 is written now)
  constexpr int TA = PH % 2, TB = (PH + 1) % 2;                      // product slots: Sobel rows Y-1 (TA), Y (TB, written now)
  constexpr int E0 = PH % 3, E1 = (PH + 1) % 3, E2 = (PH + 2) % 3;  // response rows i-5, i-4, i-3; E0 is overwritten by row i-2
  constexpr int PF = PH % EW_PF;
  const int i0 = c.y0[0] - 3 + s, i1 = c.y0[1] - 3 + s;              // image rows of this step (virtual)

  // ---- 3x3 local maxima of response row i-4 (independent of this step's image row: issued first, its lane exchanges overlap)
  const f2 cm = f2{vmax(st.E[E0].x, st.E[E2].x), vmax(st.E[E0].y, st.E[E2].y)};   // column maximum without the centre
  const f2 cf = f2{vmax(cm.x, st.E[E1].x), vmax(cm.y, st.E[E1].y)};               // ... with it
  const f2 cfl = bperm_f2(c.addr_l, cf), cfr = bperm_f2(c.addr_r, cf);

  // ---- image row i: horizontal difference and smoothing, both halves packed in 16-bit lanes
  const unsigned P = st.pix[PF][0], Ph = st.pix[PF][1];
  ew_load(c, i0 + EW_PF, i1 + EW_PF, st.pix[PF]);                     // refill the slot with the row EW_PF steps ahead
  unsigned Lp = (unsigned)__builtin_amdgcn_ds_bpermute(c.addr_l, (int)P), Rp = (unsigned)__builtin_amdgcn_ds_bpermute(c.addr_r, (int)P);
  Lp = c.lane_first ? Ph : Lp;
  Rp = c.lane_last ? Ph : Rp;
  const s2 l = __builtin_bit_cast(s2, Lp), r = __builtin_bit_cast(s2, Rp), m = __builtin_bit_cast(s2, P);
  st.hd[H2] = r - l;
  st.hs[H2] = m * s2{2, 2} + (l + r);

  // ---- Sobel pair of row Y = i-1 and its products
  const s2 dx = st.hd[H1] * s2{2, 2} + (st.hd[H0] + st.hd[H2]);
  const s2 dy = st.hs[H2] - st.hs[H0];
  const int Y0 = i0 - 1, Y1 = i1 - 1;
  const float scale = (float)(1.0 / (255.0 * 4.0 * 3.0));
  // +-scale per half, the sign bit built with scalar integer arithmetic (row Y outside [0, h) <=> Y or h-1-Y negative)
  const f2 sy = f2{__uint_as_float(__float_as_uint(scale) | ((unsigned)(Y0 | (c.h - 1 - Y0)) & 0x80000000u)),
                   __uint_as_float(__float_as_uint(scale) | ((unsigned)(Y1 | (c.h - 1 - Y1)) & 0x80000000u))};
  const f2 fx = f2{(float)(int)dx.x, (float)(int)dx.y} * f2{c.sx, c.sx};
  const f2 fy = f2{(float)(int)dy.x, (float)(int)dy.y} * sy;
  st.T[TB][0][1] = fx * fx;
  st.T[TB][1][1] = fx * fy;
  st.T[TB][2][1] = fy * fy;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    st.T[TB][p][0] = bperm_f2(c.addr_l, st.T[TB][p][1]);
    st.T[TB][p][2] = bperm_f2(c.addr_r, st.T[TB][p][1]);
  }

  // ---- finish the local maxima of row y = i-4
  {
    const int y[2] = {i0 - 4, i1 - 4};
    const float v[2] = {st.E[E1].x, st.E[E1].y};
    // "positive and no neighbour greater" as ONE comparison: value >= max(neighbours, floor of the lane)
    const float nb[2] = {vmax(vmax3(cfl.x, cm.x, cfr.x), c.nms_floor), vmax(vmax3(cfl.y, cm.y, cfr.y), c.nms_floor)};
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const bool row_ok = y[q] >= c.y0[q] && y[q] < c.y1[q] && y[q] >= 1 && y[q] < c.h - 1;  // wave-uniform
      if (row_ok) {
        const bool ismax = v[q] >= nb[q];
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(ismax);
        if (bal) {
          // buffered raw (response bits, pixel offset); ew_flush turns them into sort keys with all lanes busy
          if (ismax) {
            const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, (unsigned)nk));
            lval[slot] = v[q];
            loff[slot] = (unsigned)(y[q] * c.w + c.x);
          }
          nk += __builtin_amdgcn_readfirstlane(__popcll(bal));
        }
      }
    }
  }

  // ---- response row r = i-2 from Sobel rows i-3 (S3), i-2 (TA), i-1 (TB): nine-term sums in raster order
  f2 acc[3];
#pragma unroll
  for (int p = 0; p < 3; p++) {
    f2 a = st.S3[p];
    a = a + st.T[TA][p][0];
    a = a + st.T[TA][p][1];
    a = a + st.T[TA][p][2];
    a = a + st.T[TB][p][0];
    a = a + st.T[TB][p][1];
    a = a + st.T[TB][p][2];
    acc[p] = a;
    st.S3[p] = (st.T[TA][p][0] + st.T[TA][p][1]) + st.T[TA][p][2];  // first row of the next step's window
  }
  const f2 half = f2{0.5f, 0.5f};
  const f2 a = acc[0] * half, b = acc[1], cc = acc[2] * half;
  const f2 e = (a + cc) - sqrt2((a - cc) * (a - cc) + b * b);
  st.E[E0] = e;
  {
    const int rr[2] = {i0 - 2, i1 - 2};
    if (rr[0] >= c.y0[0] && rr[0] < c.y1[0]) st.vmax.x = vmax(st.vmax.x, e.x);
    if (rr[1] >= c.y0[1] && rr[1] < c.y1[1]) st.vmax.y = vmax(st.vmax.y, e.y);
  }
}

// appends the wave's buffered keys to the stream's list (one global atomic per flush)
__device__ __noinline__ void ew_flush(const float* lval, const unsigned* loff, int nk, unsigned long long* __restrict__ keys, int* __restrict__ nkeys, int cap) {
  if (nk == 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int base = 0;
  if (lane_id() == 0) base = atomicAdd(nkeys, nk);
  base = __builtin_amdgcn_readfirstlane(base);
  for (int i = lane_id(); i < nk; i += 64)
    if (base + i < cap) {
      const unsigned long long key = ((unsigned long long)f32_ordered(lval[i]) << 32) | loff[i];
      keys[base + i] = ~key;  // ascending key order == (response descending, pixel offset descending), as k_eig_cand's keys
    }
  __builtin_amdgcn_wave_barrier();
}

// grid: 1-D, one wave per workgroup; workgroup id -> (stream, strip, chunk pair) with a stream's workgroups on one XCD (id % 8)
__global__ __launch_bounds__(64) void k_eig_walk(ImgSel src, int w, int h, int pitch, size_t sstride, int S, int n_strips, int n_pairs, int R,
                                                 unsigned* __restrict__ maxenc, unsigned long long* __restrict__ keys,
                                                 int* __restrict__ nkeys, int cap, const int* __restrict__ active) {
  __shared__ float lval[EW_KEYS];
  __shared__ unsigned loff[EW_KEYS];
  const int per_stream = n_strips * n_pairs;
  const int xcd = blockIdx.x & 7, q8 = blockIdx.x >> 3;
  const int s = (q8 / per_stream) * 8 + xcd;
  if (s >= S) return;
  if (active && !active[s]) return;
  const int tile = q8 % per_stream;
  const int strip = tile % n_strips, pair = tile / n_strips;
  const int lane = threadIdx.x;

  EwConst c;
  {  // the image base is wave-uniform: a buffer descriptor in scalar registers (no per-lane 64-bit address arithmetic)
    const unsigned long long a = (unsigned long long)src.ptr(s, sstride);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    c.img = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  }
  c.w = w, c.h = h, c.pitch = pitch;
  c.y0[0] = 2 * pair * R;
  c.y0[1] = c.y0[0] + R;
  c.y1[0] = min(c.y0[0] + R, h);
  c.y1[1] = min(c.y0[1] + R, h);
  const int xs = strip * EW_OUT - 2;
  const int X = xs + lane;
  c.x = X;
  c.col_own = reflect101c(X, w);
  c.col_halo = lane == 0 ? reflect101c(xs - 1, w) : (lane == 63 ? reflect101c(xs + 64, w) : c.col_own);
  c.addr_l = ((lane + 63) & 63) * 4;
  c.addr_r = ((lane + 1) & 63) * 4;
  c.lane_first = lane == 0;
  c.lane_last = lane == 63;
  c.out_col = lane >= 2 && lane <= 61 && X < w;
  c.nms_floor = (c.out_col && X >= 1 && X < w - 1) ? __uint_as_float(1u) : INFINITY;
  const float scale = (float)(1.0 / (255.0 * 4.0 * 3.0));
  c.sx = (X < 0 || X >= w) ? -scale : scale;

  EwState st;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    st.hd[k] = s2{0, 0};
    st.hs[k] = s2{0, 0};
    st.E[k] = f2{0.f, 0.f};
    st.S3[k] = f2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; j++) st.T[0][k][j] = st.T[1][k][j] = f2{0.f, 0.f};
  }
  st.vmax = f2{-INFINITY, -INFINITY};
#pragma unroll
  for (int k = 0; k < EW_PF; k++) ew_load(c, c.y0[0] - 3 + k, c.y0[1] - 3 + k, st.pix[k]);

  unsigned long long* gkeys = keys + (size_t)s * cap;
  int nk = 0;
  const int nsteps = R + EW_LAG;
  constexpr int FULL = EW_KEYS - 2 * EW_OUT;  // a step appends at most 2 x EW_OUT keys
#define EW_STEP(k)                                                    \
  if (nk > FULL) {                                                    \
    ew_flush(lval, loff, nk, gkeys, nkeys + s, cap);                  \
    nk = 0;                                                           \
  }                                                                   \
  ew_step<k>(c, st, s0 + k, lval, loff, nk);
  for (int s0 = 0; s0 < nsteps; s0 += 6) {
    EW_STEP(0)
    if (s0 + 1 >= nsteps) break;
    EW_STEP(1)
    if (s0 + 2 >= nsteps) break;
    EW_STEP(2)
    if (s0 + 3 >= nsteps) break;
    EW_STEP(3)
    if (s0 + 4 >= nsteps) break;
    EW_STEP(4)
    if (s0 + 5 >= nsteps) break;
    EW_STEP(5)
  }
#undef EW_STEP
  ew_flush(lval, loff, nk, gkeys, nkeys + s, cap);

  // the maximum over this wave's own output pixels (rows of the two halves that lie in the image; a half without rows kept -inf)
  unsigned m = 0;
  if (c.out_col) {
    if (c.y1[0] > c.y0[0]) m = f32_ordered(st.vmax.x);
    if (c.y1[1] > c.y0[1]) {
      const unsigned m1 = f32_ordered(st.vmax.y);
      m = m1 > m ? m1 : m;
    }
  }
  m = wave_max_u32(m);
  if (lane == 0 && m) atomicMax(&maxenc[s], m);
}

void launch_eig_walk(hipStream_t st, ImgSel src, int w, int h, int pitch, size_t sstride, int S, unsigned* maxenc, unsigned long long* keys,
                     int* nkeys, int cap, const int* active, int rows_per_chunk) {
  const int n_strips = (w + EW_OUT - 1) / EW_OUT;
  int R = rows_per_chunk > 0 ? rows_per_chunk : 60;
  int n_pairs = (h + 2 * R - 1) / (2 * R);
  if (n_pairs < 1) n_pairs = 1;
  R = (h + 2 * n_pairs - 1) / (2 * n_pairs);  // even out the chunks
  const int S8 = (S + 7) / 8 * 8;
  const int grid = S8 * n_strips * n_pairs;
  hipLaunchKernelGGL(k_eig_walk, dim3(grid), dim3(64), 0, st, src, w, h, pitch, sstride, S, n_strips, n_pairs, R, maxenc, keys, nkeys, cap, active);
}

}  // namespace flvis
