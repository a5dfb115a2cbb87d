// flvis_amd: strip-mined phases of the corner-response tile (cornerMinEigenVal of cv::goodFeaturesToTrack, blockSize 3, Sobel 3):
// the same arithmetic as eig_tile<1> of img_kernels.hip, pixel for pixel and operation for operation, with every thread producing
// FOUR horizontally adjacent values from one register-resident window instead of one value from nine (eighteen) LDS reads:
//   * 3x3 local maxima: a 3 x 6 window of responses -> the candidate keys of 4 output pixels (phase C).
//   * Sobel: a 3 x 6 byte window -> 4 (fx, fy) pairs.  The integer sums are exact, so they are shared as column sums
//     (dx = col[k+2] - col[k], col = top + 2 mid + bottom) and row differences (dy = d[k] + 2 d[k+1] + d[k+2], d = bottom - top).
//   * covariance box sums + smaller eigenvalue: a 3 x 6 window of fx / fy -> the 18 products fx*fx, fx*fy, fy*fy once, then the
//     nine-term sums of each of the 4 pixels in the SAME raster order as the one-pixel code (the float results are bit-identical).
// Static instruction counts of the gfx950 ISA: 55 per Sobel pair and 84 per response in the one-pixel form, 22 and 59 in the strips
// (the nine-term sums cannot be shared without changing the order of the float additions, so the additions stay).
//
// Plain C++ on purpose (no HIP intrinsics; the only wide loads are memcpy from pointers declared aligned): the same functions are
// compiled for the host by tests/cpp/eig_strip_check.cpp, which runs them tile by tile over whole images and compares the
// response map bit for bit with the CPU restatement -- the arithmetic of this variant is checked without a GPU.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define FLVIS_EIG_HD __host__ __device__ __forceinline__
#else
#define FLVIS_EIG_HD inline
#endif

namespace flvis {
namespace eigstrip {

constexpr int TW = 64, TH = 16, HALO = 1;          // outputs per tile; the response is computed on a +1 ring for the 3x3 maxima
constexpr int OW = TW + 2 * HALO, OH = TH + 2 * HALO;  // response region          66 x 18
constexpr int CW = OW + 2, CH = OH + 2;                // (fx, fy) region           68 x 20
constexpr int IH = CH + 2;                             // image rows of the tile    22
constexpr int IW = 76;                                 // image bytes per tile row (4 bytes of left padding: the tile starts dword aligned)
constexpr int XOFF = 4;                                // tile column 0 is image column x0 - XOFF; tile row 0 is image row y0 - HALO - 2
constexpr int A_STRIPS = CW / 4;                       // 17 Sobel strips per row
constexpr int A_ITEMS = CH * A_STRIPS;                 // 340
constexpr int B_FULL = OW / 4;                         // 16 strips of four per row of the response region, and one of two
constexpr int B_ITEMS = OH * B_FULL + OH;              // 288 strips of four first, then the 18 strips of two (one per row): the
                                                       // two shapes do not share a wave except in the last one

FLVIS_EIG_HD int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// one (fx, fy) value the way eig_tile<1> computes it: the position is first reflected into the image (boxFilter's REFLECT_101
// on the covariance maps), then the Sobel pair is taken from the tile
FLVIS_EIG_HD void sobel_one(int r, int c, int w, int h, int x0, int y0, const uint8_t* tile, float& fx, float& fy) {
  const float scale = (float)(1.0 / (255.0 * 4.0 * 3.0));
  const int X = x0 - HALO - 1 + c, Y = y0 - HALO - 1 + r;
  const int tx = reflect101(X, w) - (x0 - XOFF), ty = reflect101(Y, h) - (y0 - HALO - 2);
  fx = 0.f, fy = 0.f;
  if (tx >= 1 && tx < IW - 1 && ty >= 1 && ty < IH - 1) {
    const uint8_t* p = tile + ty * IW + tx;
    const int a = p[-IW - 1], b = p[-IW], cc = p[-IW + 1], d = p[-1], f = p[1], g = p[IW - 1], hh = p[IW], k = p[IW + 1];
    fx = (float)((cc + 2 * f + k) - (a + 2 * d + g)) * scale;
    fy = (float)((g + 2 * hh + k) - (a + 2 * b + cc)) * scale;
  }
}

// phase A, item in [0, A_ITEMS): (fx, fy) of the four positions (r, 4 q .. 4 q + 3) of the Sobel region
FLVIS_EIG_HD void sobel_strip(int item, int w, int h, int x0, int y0, const uint8_t* tile, float* sfx, float* sfy) {
  const int r = item / A_STRIPS, c0 = 4 * (item - r * A_STRIPS);
  const int X0 = x0 - HALO - 1 + c0, Y = y0 - HALO - 1 + r;
  float* ofx = sfx + r * CW + c0;
  float* ofy = sfy + r * CW + c0;
  if (Y >= 0 && Y < h && X0 >= 0 && X0 + 3 < w) {
    // nothing is reflected: the strip's window is rows r .. r + 2, bytes c0 + 1 .. c0 + 6 of the tile
    const float scale = (float)(1.0 / (255.0 * 4.0 * 3.0));
    // (two aligned dwords per row: bytes c0 .. c0 + 7, of which 1 .. 6 are the window; little endian on both sides)
    uint32_t lo[3], hi[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const uint8_t* q = (const uint8_t*)__builtin_assume_aligned(tile + (r + j) * IW + c0, 4);
      memcpy(&lo[j], q, 4);
      memcpy(&hi[j], q + 4, 4);
    }
    int col[6], dif[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const int sh = 8 * ((j + 1) & 3);
      const int t = (int)(((j < 3 ? lo[0] : hi[0]) >> sh) & 255u), m = (int)(((j < 3 ? lo[1] : hi[1]) >> sh) & 255u),
                b = (int)(((j < 3 ? lo[2] : hi[2]) >> sh) & 255u);
      col[j] = t + 2 * m + b;
      dif[j] = b - t;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      ofx[k] = (float)(col[k + 2] - col[k]) * scale;
      ofy[k] = (float)(dif[k] + 2 * dif[k + 1] + dif[k + 2]) * scale;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) sobel_one(r, c0 + k, w, h, x0, y0, tile, ofx[k], ofy[k]);
  }
}

// the smaller eigenvalue of [[sa/2, sb], [sb, sc/2]] exactly as eig_tile<1> writes it
FLVIS_EIG_HD float min_eig(float sa, float sb, float sc) {
  const float a = sa * 0.5f, b = sb, cc = sc * 0.5f;
  return (a + cc) - sqrtf((a - cc) * (a - cc) + b * b);
}

// N responses (r, c0 .. c0 + N - 1) from the 3 x (N + 2) window of (fx, fy)
template <int N>
FLVIS_EIG_HD void box_window(const float* fxw, const float* fyw, float* out) {
  float pxx[3][N + 2], pxy[3][N + 2], pyy[3][N + 2];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < N + 2; k++) {
      const float fx = fxw[j * CW + k], fy = fyw[j * CW + k];
      pxx[j][k] = fx * fx;
      pxy[j][k] = fx * fy;
      pyy[j][k] = fy * fy;
    }
#pragma unroll
  for (int q = 0; q < N; q++) {
    float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 3; k++) {  // raster order of the 3x3 window, as in the one-pixel form
        sa += pxx[j][q + k];
        sb += pxy[j][q + k];
        sc += pyy[j][q + k];
      }
    out[q] = min_eig(sa, sb, sc);
  }
}

// phase B, item in [0, B_ITEMS): the responses of one strip of the response region
FLVIS_EIG_HD void box_strip(int item, const float* sfx, const float* sfy, float* eig) {
  if (item < OH * B_FULL) {
    const int r = item / B_FULL, c0 = 4 * (item - r * B_FULL);
    box_window<4>(sfx + r * CW + c0, sfy + r * CW + c0, eig + r * OW + c0);
  } else {
    const int r = item - OH * B_FULL, c0 = 4 * B_FULL;  // the last two responses of row r
    box_window<2>(sfx + r * CW + c0, sfy + r * CW + c0, eig + r * OW + c0);
  }
}

// order-preserving float -> uint32 (larger float => larger uint), as dev_common.hpp's f32_ordered
FLVIS_EIG_HD uint32_t ordered_bits(float f) {
  uint32_t b;
  memcpy(&b, &f, 4);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

constexpr int C_ITEMS = TH * (TW / 4);  // 256: one strip of four output pixels per thread

// phase C, item in [0, C_ITEMS): the four output pixels (y0 + r, x0 + 4 q .. + 3).  Returns a 4-bit mask of the 3x3 local maxima among
// them (a pixel strictly inside the image whose response is positive and not exceeded by any neighbour -- see k_eig_cand) with the
// sort key ~((ordered(response) << 32) | pixel offset) of pixel q in keys[q] (defined where the mask bit is set), and in max_ordered
// the largest ordered(response) of the strip's pixels that lie in the image (0 when none does): exactly what the one-pixel loop of
// k_eig_cand contributes for them.  Written without branches over the pixels: "no neighbour is greater" is !(max of the eight > v)
// (fmaxf ignores a NaN operand just as the comparison `neighbour > v` does).
FLVIS_EIG_HD unsigned nms_strip(int item, int w, int h, int x0, int y0, const float* eig, unsigned long long* keys, uint32_t& max_ordered) {
  const int r = item / (TW / 4), c0 = 4 * (item - r * (TW / 4));
  const int y = y0 + r;
  max_ordered = 0;
  if (y >= h || x0 + c0 >= w) return 0u;
  float win[3][6];  // responses of rows r .. r + 2, columns c0 .. c0 + 5 of the response region (output pixel (r, c) sits at (r + 1, c + 1))
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int k = 0; k < 6; k++) win[j][k] = eig[(r + j) * OW + c0 + k];
  float colmax[6];  // max of the top and bottom row per column: shared by the pixels whose windows contain the column
#pragma unroll
  for (int k = 0; k < 6; k++) colmax[k] = fmaxf(win[0][k], win[2][k]);
  const bool row_inside = y >= 1 && y < h - 1;
  unsigned mask = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int x = x0 + c0 + q;
    const float v = win[1][q + 1];
    const uint32_t ev = ordered_bits(v);
    const bool in_image = x < w;
    max_ordered = (in_image && ev > max_ordered) ? ev : max_ordered;
    const float nb = fmaxf(fmaxf(fmaxf(colmax[q], colmax[q + 1]), colmax[q + 2]), fmaxf(win[1][q], win[1][q + 2]));
    const bool is_max = in_image && row_inside && x >= 1 && x < w - 1 && v > 0.f && !(nb > v);
    keys[q] = ~(((unsigned long long)ev << 32) | (unsigned)(y * w + x));
    mask |= is_max ? (1u << q) : 0u;
  }
  return mask;
}

}  // namespace eigstrip
}  // namespace flvis
