// flvis_amd: batched pyramidal Lucas-Kanade tracker for gfx950 (CDNA4).
//
// Replaces cv::calcOpticalFlowPyrLK as the reference calls it:
//   temporal tracking   src/processing/lkorb_tracking.cpp:64-73   (31x31, maxLevel 10 -> clamped, 30 it, eps 1e-3, USE_INITIAL_FLOW)
//   stereo "matching"   src/processing/camera_frame.cpp:124-128   (same call, img0 -> img1, maxLevel 5 -> clamped)
//
// Mapping: ONE WAVE (64 lanes) per (stream, point); the wave walks the pyramid levels top-down and runs the <=30
// Gauss-Newton iterations in-kernel.  Lane l owns window row (l>>1) and the 16-column half (l&1) of the 31x31 window:
// the interpolated template (I, Ix, Iy as int16) lives in that lane's registers for the whole level, so LDS only
// holds the 34x36-byte source patch / 32x36-byte search patch, staged with dword loads + v_alignbyte so that each
// lane then reads 5 aligned dwords per row.  Scharr derivatives are computed on the fly from the staged patch (no
// derivative image in HBM).  Sums are exact integers (int32 per lane, int64 across the wave) -> bit-exact vs oracle.
#include "dev_common.hpp"
#include "img_kernels.hpp"

namespace flvis {

constexpr int LK_WIN = 31;
constexpr int LK_PS = 40;     // LDS patch row stride (bytes); 36 used
constexpr int LK_PROWS = 34;  // rows of the template-source patch

// patch[r][c] = img(X0 + c, Y0 + r) for r < nrows, c < 36, REFLECT_101 outside the image
__device__ __forceinline__ void lk_load_patch(const uint8_t* __restrict__ img, int w, int h, int pitch, int X0, int Y0,
                                              int nrows, uint8_t* patch) {
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
}

__device__ __forceinline__ int descale_i(int x, int n) { return (x + (1 << (n - 1))) >> n; }

#define LK_BYTE(D, K) ((int)(((D)[(K) >> 2] >> (((K)&3) * 8)) & 255u))

__global__ __launch_bounds__(64) void k_lk_track(PyrSel prev, PyrSel next, const float* __restrict__ prev_pts,
                                                 float* __restrict__ next_pts, uint8_t* __restrict__ status,
                                                 const int* __restrict__ count, int nmax, LKParams prm,
                                                 const int* __restrict__ active) {
  const int s = blockIdx.y;
  if (active && !active[s]) return;
  int n = count[s];
  if (n > nmax) n = nmax;
  __shared__ __attribute__((aligned(16))) uint8_t patch[LK_PROWS * LK_PS];
  const int lane = threadIdx.x;
  const int r = lane >> 1;
  const int c0 = (lane & 1) * 16;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float halfWin = (LK_WIN - 1) * 0.5f;

  for (int p = blockIdx.x; p < n; p += gridDim.x) {
    const size_t pi = ((size_t)s * nmax + p) * 2;
    const float ppx0 = prev_pts[pi], ppy0 = prev_pts[pi + 1];
    float nx = next_pts[pi], ny = next_pts[pi + 1];
    int st = 1;
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

      // ---- template: stage source patch rows ipy-1..ipy+32, cols ipx-1..ipx+34
      __syncthreads();
      lk_load_patch(prev.lvl[level].ptr(s, prev.stride[level]), W, H, prev.pitch[level], ipx - 1, ipy - 1, LK_PROWS,
                    patch);
      __syncthreads();
      short tI[16], tX[16], tY[16];
      int a11 = 0, a12 = 0, a22 = 0;
      if (r < LK_WIN) {
        uint32_t d[4][5];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int k = 0; k < 5; k++) d[j][k] = *reinterpret_cast<const uint32_t*>(patch + (r + j) * LK_PS + c0 + 4 * k);
        // Scharr derivatives on window rows r (dr=0) and r+1 (dr=1), window cols c0..c0+16
        int dx[2][17], dy[2][17];
#pragma unroll
        for (int dr = 0; dr < 2; dr++) {
          int t0[19], t1[19];
#pragma unroll
          for (int k = 0; k < 19; k++) {
            int va = LK_BYTE(d[dr], k), vb = LK_BYTE(d[dr + 1], k), vc = LK_BYTE(d[dr + 2], k);
            t0[k] = (va + vc) * 3 + vb * 10;
            t1[k] = vc - va;
          }
          const int Y = ipy + r + dr;
          const bool yin = (Y >= 0 && Y < H);
#pragma unroll
          for (int c = 0; c < 17; c++) {
            const int X = ipx + c0 + c;
            const bool in = yin && X >= 0 && X < W;
            dx[dr][c] = in ? (t0[c + 2] - t0[c]) : 0;
            dy[dr][c] = in ? ((t1[c + 2] + t1[c]) * 3 + t1[c + 1] * 10) : 0;
          }
        }
#pragma unroll
        for (int c = 0; c < 16; c++) {
          if (c0 + c < LK_WIN) {
            int i00 = LK_BYTE(d[1], c + 1), i01 = LK_BYTE(d[1], c + 2), i10 = LK_BYTE(d[2], c + 1),
                i11 = LK_BYTE(d[2], c + 2);
            int ival = descale_i(i00 * iw00 + i01 * iw01 + i10 * iw10 + i11 * iw11, W_BITS - 5);
            int ixval = descale_i(dx[0][c] * iw00 + dx[0][c + 1] * iw01 + dx[1][c] * iw10 + dx[1][c + 1] * iw11, W_BITS);
            int iyval = descale_i(dy[0][c] * iw00 + dy[0][c + 1] * iw01 + dy[1][c] * iw10 + dy[1][c + 1] * iw11, W_BITS);
            tI[c] = (short)ival;
            tX[c] = (short)ixval;
            tY[c] = (short)iyval;
            a11 += ixval * ixval;
            a12 += ixval * iyval;
            a22 += iyval * iyval;
          } else {
            tI[c] = 0;
            tX[c] = 0;
            tY[c] = 0;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 16; c++) {
          tI[c] = 0;
          tX[c] = 0;
          tY[c] = 0;
        }
      }
      const long long iA11 = wave_sum_i64((long long)a11), iA12 = wave_sum_i64((long long)a12),
                      iA22 = wave_sum_i64((long long)a22);
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
      const int JW = next.w[level], JH = next.h[level];
      const uint8_t* Jimg = next.lvl[level].ptr(s, next.stride[level]);
      for (int j = 0; j < prm.max_iter; j++) {
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
        __syncthreads();
        lk_load_patch(Jimg, JW, JH, next.pitch[level], inx, iny, 32, patch);
        __syncthreads();
        int b1 = 0, b2 = 0;
        if (r < LK_WIN) {
          uint32_t e[2][5];
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int k = 0; k < 5; k++) e[q][k] = *reinterpret_cast<const uint32_t*>(patch + (r + q) * LK_PS + c0 + 4 * k);
#pragma unroll
          for (int c = 0; c < 16; c++) {
            int j00 = LK_BYTE(e[0], c), j01 = LK_BYTE(e[0], c + 1), j10 = LK_BYTE(e[1], c), j11 = LK_BYTE(e[1], c + 1);
            int diff = descale_i(j00 * iw00 + j01 * iw01 + j10 * iw10 + j11 * iw11, W_BITS - 5) - (int)tI[c];
            // columns >= 31 have tX = tY = 0, so they add nothing
            b1 += diff * (int)tX[c];
            b2 += diff * (int)tY[c];
          }
        }
        const long long ib1 = wave_sum_i64((long long)b1), ib2 = wave_sum_i64((long long)b2);
        const float fb1 = (float)ib1 * FLT_SCALE, fb2 = (float)ib2 * FLT_SCALE;
        const float ddx = (A12 * fb2 - A22 * fb1) * D;
        const float ddy = (A12 * fb1 - A11 * fb2) * D;
        npx += ddx;
        npy += ddy;
        nx = npx + halfWin;
        ny = npy + halfWin;
        if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= prm.eps2) break;
        if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) {
          nx -= ddx * 0.5f;
          ny -= ddy * 0.5f;
          break;
        }
        pdx = ddx;
        pdy = ddy;
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
    }
  }
}

void launch_lk_track(hipStream_t st, const PyrSel& prev, const PyrSel& next, const float* prev_pts, float* next_pts,
                     uint8_t* status, const int* count, int nmax, int S, LKParams prm, const int* active) {
  int gx = nmax < 512 ? nmax : 512;
  hipLaunchKernelGGL(k_lk_track, dim3(gx, S), dim3(64), 0, st, prev, next, prev_pts, next_pts, status, count, nmax, prm,
                     active);
}

}  // namespace flvis
