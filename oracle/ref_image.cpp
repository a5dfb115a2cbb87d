// ORACLE (test infrastructure, NOT product code) -- image-domain stages of the FLVIS front-end.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// What is restated here and where the reference calls it:
//   equalizeHist            src/frontend/f2f_tracking.cpp:127,143-144          (cv::equalizeHist)
//   pyramid + Scharr + LK   src/processing/lkorb_tracking.cpp:64-73            (cv::calcOpticalFlowPyrLK, 31x31, 30 it, 1e-3)
//                           src/processing/camera_frame.cpp:124-128            (same, img0 -> img1 "stereo matching")
//   goodFeaturesToTrack     src/processing/feature_dem.cpp:160,221             (cv::goodFeaturesToTrack, block 3, min-eig)
//   FeatureDEM              src/processing/feature_dem.cpp:12-266              (first-party, restated with its quirks)
//
// The OpenCV routines live in a dependency that is NOT under /root/reference (OpenCV "3 EXACT else 4",
// CMakeLists.txt:46-49).  They are restated from OpenCV's published algorithms (Bouguet pyramidal LK with 14-bit
// fixed-point bilinear weights and int16 Scharr derivatives; Shi-Tomasi min-eigenvalue map with Sobel-3 + 3x3 box;
// 5-tap [1 4 6 4 1] pyrDown with (x+128)>>8 rounding; BORDER_REFLECT_101 everywhere).
// **parity unpinned**: the reference has no tests / golden vectors for these stages; the oracle is
// validated in tests/ by independent numpy restatements and by closed-form synthetic flow.
//
// Two deliberate, documented choices where OpenCV itself is build-dependent:
//   (1) LK window sums (A11,A12,A22,b1,b2) are accumulated EXACTLY in int64 and converted to float once; OpenCV
//       accumulates in float (scalar path) or in SIMD lanes (order differs per build), so its last bits are not defined.
//   (2) Sort ties: GFTT candidates are ordered by (response desc, pixel offset desc) -- OpenCV >=3.4.2 greaterThanPtr;
//       FeatureDEM's per-region std::sort is the real std::sort of this toolchain (libstdc++, as the reference's GCC build): ties end where its introsort leaves them.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "ref_api.h"

namespace ref {

static inline int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// ------------------------------------------------------------------------------------------ cvtColor BGR(A) -> gray
// cv::cvtColor(img, out, CV_BGR2GRAY / CV_BGRA2GRAY) as F2FTracking::image_feed applies it to 3/4-channel input
// (src/frontend/f2f_tracking.cpp:74-111, mbRGB is constant 0).  OpenCV 3.x RGB2Gray<uchar> (imgproc/src/color.cpp): 14-bit
// fixed point, B2Y = 1868, G2Y = 9617, R2Y = 4899, CV_DESCALE = (x + 2^13) >> 14; the alpha channel is ignored.
void cvt_bgr_to_gray(const uint8_t* src, int channels, uint8_t* dst, int w, int h) {
  for (size_t i = 0; i < (size_t)w * h; i++) {
    const uint8_t* p = src + i * channels;
    dst[i] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
  }
}

// ------------------------------------------------------------------------------------------ equalizeHist
void equalize_hist(const uint8_t* src, uint8_t* dst, int w, int h) {
  int hist[256] = {0};
  const int total = w * h;
  for (int i = 0; i < total; i++) hist[src[i]]++;
  int i = 0;
  while (!hist[i]) ++i;
  if (hist[i] == total) {
    memset(dst, i, (size_t)total);
    return;
  }
  float scale = (256 - 1.f) / (total - hist[i]);
  int sum = 0;
  int lut[256];
  for (lut[i++] = 0; i < 256; ++i) {
    sum += hist[i];
    long v = lrintf(sum * scale);  // saturate_cast<uchar>(float) == cvRound + clamp
    lut[i] = (int)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
  for (int k = 0; k < total; k++) dst[k] = (uint8_t)lut[src[k]];
}

// ------------------------------------------------------------------------------------------ pyrDown
void pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  std::vector<int> rowbuf((size_t)5 * dw);
  static const int k5[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; y++) {
    for (int r = 0; r < 5; r++) {
      const uint8_t* srow = src + (size_t)reflect101(2 * y - 2 + r, h) * w;
      for (int x = 0; x < dw; x++) {
        int s = 0;
        for (int c = 0; c < 5; c++) s += k5[c] * srow[reflect101(2 * x - 2 + c, w)];
        rowbuf[(size_t)r * dw + x] = s;
      }
    }
    for (int x = 0; x < dw; x++) {
      int s = 0;
      for (int r = 0; r < 5; r++) s += k5[r] * rowbuf[(size_t)r * dw + x];
      dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
    }
  }
}

int lk_num_levels(int w, int h, int win, int max_level) {
  // buildOpticalFlowPyramid: stop once the NEXT level would be <= winSize in either dimension.
  int level = 0;
  for (; level <= max_level; ++level) {
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (w <= win || h <= win) return level;
  }
  return max_level;
}

// ------------------------------------------------------------------------------------------ Scharr (calcSharrDeriv)
// dx = [3 10 3]^T (vertical smooth) x [-1 0 1]; dy = [-1 0 1]^T x [3 10 3]; reflect-101 inside the image.
static inline void scharr_at(const uint8_t* img, int w, int h, int x, int y, int& dx, int& dy) {
  int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
  int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const uint8_t *r0 = img + (size_t)ym * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)yp * w;
  int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
  int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
  dx = t0p - t0m;
  dy = (t1p + t1m) * 3 + t1c * 10;
}

struct Level {
  int w, h;
  std::vector<uint8_t> img;
};

static inline int pix(const Level& L, int x, int y) {  // image with its REFLECT_101 border of winSize
  return L.img[(size_t)reflect101(y, L.h) * L.w + reflect101(x, L.w)];
}
static inline void deriv(const Level& L, int x, int y, int& dx, int& dy) {  // BORDER_CONSTANT(0) outside the image
  if (x < 0 || y < 0 || x >= L.w || y >= L.h) {
    dx = dy = 0;
    return;
  }
  scharr_at(L.img.data(), L.w, L.h, x, y, dx, dy);
}

static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

static void build_pyramid(const uint8_t* img, int w, int h, int levels, std::vector<Level>& pyr) {
  pyr.resize(levels + 1);
  pyr[0].w = w;
  pyr[0].h = h;
  pyr[0].img.assign(img, img + (size_t)w * h);
  for (int l = 1; l <= levels; l++) {
    pyr[l].w = (pyr[l - 1].w + 1) / 2;
    pyr[l].h = (pyr[l - 1].h + 1) / 2;
    pyr[l].img.resize((size_t)pyr[l].w * pyr[l].h);
    pyr_down(pyr[l - 1].img.data(), pyr[l - 1].w, pyr[l - 1].h, pyr[l].img.data());
  }
}

// cv::calcOpticalFlowPyrLK restated (LKTrackerInvoker::operator()).  next_pts is in/out (OPTFLOW_USE_INITIAL_FLOW).
void calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
                              float* next_pts, uint8_t* status, int n, int win, int max_level, int max_iter,
                              double eps, int use_initial_flow, float min_eig_thr) {
  const int levels = lk_num_levels(w, h, win, max_level);
  std::vector<Level> P, N;
  build_pyramid(prev, w, h, levels, P);
  build_pyramid(next, w, h, levels, N);
  if (max_iter < 0) max_iter = 0;
  if (max_iter > 100) max_iter = 100;
  if (eps < 0) eps = 0;
  if (eps > 10) eps = 10;
  const double eps2 = eps * eps;
  const float halfWin = (win - 1) * 0.5f;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  std::vector<short> IWin((size_t)win * win), dIWin((size_t)win * win * 2);
  for (int i = 0; i < n; i++) status[i] = 1;

  for (int level = levels; level >= 0; level--) {
    const Level& I = P[level];
    const Level& J = N[level];
    for (int p = 0; p < n; p++) {
      float ppx = prev_pts[2 * p] * (float)(1. / (1 << level));
      float ppy = prev_pts[2 * p + 1] * (float)(1. / (1 << level));
      float npx, npy;
      if (level == levels) {
        if (use_initial_flow) {
          npx = next_pts[2 * p] * (float)(1. / (1 << level));
          npy = next_pts[2 * p + 1] * (float)(1. / (1 << level));
        } else {
          npx = ppx;
          npy = ppy;
        }
      } else {
        npx = next_pts[2 * p] * 2.f;
        npy = next_pts[2 * p + 1] * 2.f;
      }
      next_pts[2 * p] = npx;
      next_pts[2 * p + 1] = npy;

      ppx -= halfWin;
      ppy -= halfWin;
      int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
      if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
        if (level == 0) status[p] = 0;
        continue;
      }
      float a = ppx - ipx, b = ppy - ipy;
      int iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
      int iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
      int iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
      int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      int64_t iA11 = 0, iA12 = 0, iA22 = 0;
      for (int y = 0; y < win; y++) {
        for (int x = 0; x < win; x++) {
          int X = ipx + x, Y = ipy + y;
          int ival = descale(pix(I, X, Y) * iw00 + pix(I, X + 1, Y) * iw01 + pix(I, X, Y + 1) * iw10 +
                                 pix(I, X + 1, Y + 1) * iw11,
                             W_BITS - 5);
          int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
          deriv(I, X, Y, dx00, dy00);
          deriv(I, X + 1, Y, dx01, dy01);
          deriv(I, X, Y + 1, dx10, dy10);
          deriv(I, X + 1, Y + 1, dx11, dy11);
          int ixval = descale(dx00 * iw00 + dx01 * iw01 + dx10 * iw10 + dx11 * iw11, W_BITS);
          int iyval = descale(dy00 * iw00 + dy01 * iw01 + dy10 * iw10 + dy11 * iw11, W_BITS);
          IWin[(size_t)y * win + x] = (short)ival;
          dIWin[((size_t)y * win + x) * 2] = (short)ixval;
          dIWin[((size_t)y * win + x) * 2 + 1] = (short)iyval;
          iA11 += (int64_t)ixval * ixval;
          iA12 += (int64_t)ixval * iyval;
          iA22 += (int64_t)iyval * iyval;
        }
      }
      float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
      if (minEig < min_eig_thr || D < 1.1920929e-07f /*FLT_EPSILON*/) {
        if (level == 0) status[p] = 0;
        continue;
      }
      D = 1.f / D;
      npx -= halfWin;
      npy -= halfWin;
      float pdx = 0, pdy = 0;
      for (int j = 0; j < max_iter; j++) {
        int inx = (int)floorf(npx), iny = (int)floorf(npy);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
          if (level == 0) status[p] = 0;
          break;
        }
        a = npx - inx;
        b = npy - iny;
        iw00 = (int)lrintf((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = (int)lrintf(a * (1.f - b) * (1 << W_BITS));
        iw10 = (int)lrintf((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t ib1 = 0, ib2 = 0;
        for (int y = 0; y < win; y++) {
          for (int x = 0; x < win; x++) {
            int X = inx + x, Y = iny + y;
            int diff = descale(pix(J, X, Y) * iw00 + pix(J, X + 1, Y) * iw01 + pix(J, X, Y + 1) * iw10 +
                                   pix(J, X + 1, Y + 1) * iw11,
                               W_BITS - 5) -
                       IWin[(size_t)y * win + x];
            ib1 += (int64_t)diff * dIWin[((size_t)y * win + x) * 2];
            ib2 += (int64_t)diff * dIWin[((size_t)y * win + x) * 2 + 1];
          }
        }
        float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
        float dx = (A12 * b2 - A22 * b1) * D;
        float dy = (A12 * b1 - A11 * b2) * D;
        npx += dx;
        npy += dy;
        next_pts[2 * p] = npx + halfWin;
        next_pts[2 * p + 1] = npy + halfWin;
        if ((double)dx * dx + (double)dy * dy <= eps2) break;
        if (j > 0 && std::fabs(dx + pdx) < 0.01 && std::fabs(dy + pdy) < 0.01) {
          next_pts[2 * p] -= dx * 0.5f;
          next_pts[2 * p + 1] -= dy * 0.5f;
          break;
        }
        pdx = dx;
        pdy = dy;
      }
      // error stage (the reference passes an err vector, so the final in-bounds test is live)
      if (status[p] && level == 0) {
        float fx = next_pts[2 * p] - halfWin, fy = next_pts[2 * p + 1] - halfWin;
        int inx = (int)floorf(fx), iny = (int)floorf(fy);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) status[p] = 0;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ min-eigenvalue map
// cornerMinEigenVal(block 3, Sobel 3): Dx,Dy = Sobel * 1/(255*4*3); cov = box3x3(dx^2, dxdy, dy^2) (unnormalised);
// lambda_min = (a/2 + c/2) - sqrt((a/2 - c/2)^2 + b^2).  All borders REFLECT_101.
static inline void sobel_at(const uint8_t* img, int w, int h, int x, int y, int& dx, int& dy) {
  int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const uint8_t *r0 = img + (size_t)ym * w, *r1 = img + (size_t)y * w, *r2 = img + (size_t)yp * w;
  dx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
  dy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
}

void min_eigen_map(const uint8_t* img, int w, int h, float* eig) {
  const float scale = (float)(1.0 / (255.0 * 4.0 * 3.0));
  std::vector<float> cxx((size_t)w * h), cxy((size_t)w * h), cyy((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int dx, dy;
      sobel_at(img, w, h, x, y, dx, dy);
      float fx = (float)dx * scale, fy = (float)dy * scale;
      cxx[(size_t)y * w + x] = fx * fx;
      cxy[(size_t)y * w + x] = fx * fy;
      cyy[(size_t)y * w + x] = fy * fy;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float sa = 0, sb = 0, sc = 0;
      for (int j = -1; j <= 1; j++) {
        int yy = reflect101(y + j, h);
        for (int i = -1; i <= 1; i++) {
          int xx = reflect101(x + i, w);
          sa += cxx[(size_t)yy * w + xx];
          sb += cxy[(size_t)yy * w + xx];
          sc += cyy[(size_t)yy * w + xx];
        }
      }
      float a = sa * 0.5f, b = sb, c = sc * 0.5f;
      eig[(size_t)y * w + x] = (a + c) - sqrtf((a - c) * (a - c) + b * b);
    }
}

// cv::goodFeaturesToTrack(img, out, maxCorners, q, minDistance[, mask=all 255]), blockSize 3, min-eig.
int good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality, double min_distance,
                           float* out_xy) {
  std::vector<float> eig((size_t)w * h);
  min_eigen_map(img, w, h, eig.data());
  float maxv = -3.402823466e+38f;
  for (size_t i = 0; i < eig.size(); i++) maxv = std::max(maxv, eig[i]);  // minMaxLoc over the whole map
  const float thr = (float)((double)maxv * quality);
  std::vector<std::pair<float, int>> cand;
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      float v = eig[(size_t)y * w + x];
      if (!(v > thr) || v == 0.f) continue;  // THRESH_TOZERO, then val != 0
      bool ismax = true;                      // val == dilate3x3(thresholded eig)
      for (int j = -1; j <= 1 && ismax; j++)
        for (int i = -1; i <= 1; i++) {
          float nv = eig[(size_t)(y + j) * w + x + i];
          float tv = nv > thr ? nv : 0.f;
          if (tv > v) {
            ismax = false;
            break;
          }
        }
      if (ismax) cand.push_back({v, y * w + x});
    }
  std::sort(cand.begin(), cand.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) {
    return a.first > b.first ? true : a.first < b.first ? false : a.second > b.second;
  });
  int ncorners = 0;
  if (min_distance >= 1) {
    const int cell = (int)lrint(min_distance);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
    const float md2 = (float)(min_distance * min_distance);
    for (size_t k = 0; k < cand.size(); k++) {
      int y = cand[k].second / w, x = cand[k].second % w;
      int xc = x / cell, yc = y / cell;
      int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
      bool good = true;
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++)
          for (auto& m : grid[(size_t)yy * gw + xx]) {
            float dx = x - m.first, dy = y - m.second;
            if (dx * dx + dy * dy < md2) {
              good = false;
              break;
            }
          }
      if (good) {
        grid[(size_t)yc * gw + xc].push_back({(float)x, (float)y});
        out_xy[2 * ncorners] = (float)x;
        out_xy[2 * ncorners + 1] = (float)y;
        ++ncorners;
        if (max_corners > 0 && ncorners == max_corners) break;
      }
    }
  } else {
    for (size_t k = 0; k < cand.size(); k++) {
      out_xy[2 * ncorners] = (float)(cand[k].second % w);
      out_xy[2 * ncorners + 1] = (float)(cand[k].second / w);
      ++ncorners;
      if (max_corners > 0 && ncorners == max_corners) break;
    }
  }
  return ncorners;
}

// ------------------------------------------------------------------------------------------ FeatureDEM
// feature_dem.cpp:12-54 ctor; :59-88 calHarrisR (quirks A6); :92-121 fillIntoRegion; :124-213 redetect; :215-266 detect
FeatureDEM::FeatureDEM(int image_width, int image_height, const double f_para[6]) {
  width = image_width;
  height = image_height;
  regionWidth = (int)floor(width / 4.0);
  regionHeight = (int)floor(height / 4.0);
  boundary_dis = (int)floor(f_para[2] / 2.0);
  max_region_feature_num = (unsigned)f_para[0];
  min_region_feature_num = (unsigned)f_para[1];
  gftt_num = (int)f_para[3];
  gftt_ql = f_para[4];
  gftt_dis = (int)f_para[5];
}

float FeatureDEM::calHarrisR(const uint8_t* img, float ptx, float pty) const {
  int xx = (int)ptx, yy = (int)pty;
  auto at = [&](int x, int y) -> int { return img[(size_t)y * width + x]; };
  int p0 = at(xx - 1, yy - 1), p1 = at(xx, yy - 1), p2 = at(xx + 1, yy - 1);
  int p3 = at(xx - 1, yy);
  int p5 = at(xx + 1, yy + 1);  // quirk: reads (x+1,y+1), feature_dem.cpp:71
  int p6 = at(xx - 1, yy + 1), p7 = at(xx, yy + 1), p8 = at(xx + 1, yy + 1);
  float IX = (float)((p0 + p3 + p6 - (p2 + p5 + p8)) / 3);  // integer division, :78
  float IY = (float)((p0 + p1 + p2 - (p6 + p7 + p8)) / 3);
  float X2 = IX * IX;
  float Y2 = IY * IX;  // quirk :81
  float XY = IX * IX;  // quirk :82
  float R = (X2 * Y2) - (XY * XY) - 0.05f * (X2 + Y2) * (X2 + Y2);
  return R;
}

void FeatureDEM::fillIntoRegion(const uint8_t* img, const std::vector<Pt2f>& pts, std::vector<Scored> (&region)[16],
                                bool existed) const {
  for (size_t i = 0; i < pts.size(); i++) {
    Pt2f pt = pts[i];
    if (pt.x >= 3 && pt.x < (width - 3) && pt.y >= 3 && pt.y < (height - 3)) {
      // float arithmetic: std::floor(float) overload + int*float + float (feature_dem.cpp:102,116)
      int regionNum = (int)(4.f * floorf(pt.y / (float)regionHeight) + pt.x / (float)regionWidth);
      float score = existed ? 99999.0f : calHarrisR(img, pt.x, pt.y);
      region[regionNum].push_back({pt, score});
    }
  }
}

static bool sortbysecdesc(const FeatureDEM::Scored& a, const FeatureDEM::Scored& b) { return a.score > b.score; }

void FeatureDEM::detect(const uint8_t* img, std::vector<Pt2f>& newPts) const {
  newPts.clear();
  std::vector<float> xy((size_t)gftt_num * 2 * 2 + 2);
  int nf = good_features_to_track(img, width, height, gftt_num * 2, gftt_ql, gftt_dis, xy.data());
  std::vector<Pt2f> features(nf);
  for (int i = 0; i < nf; i++) features[i] = {xy[2 * i], xy[2 * i + 1]};
  std::vector<Scored> region[16];
  fillIntoRegion(img, features, region, false);
  for (int i = 0; i < 16; i++) {
    std::sort(region[i].begin(), region[i].end(), sortbysecdesc);  // the reference's very call (feature_dem.cpp:230): ties end where libstdc++'s introsort leaves them
    std::vector<Scored> tmp = region[i];
    region[i].clear();
    unsigned count = 0;
    for (size_t j = 0; j < tmp.size(); j++) {
      int ok = 1;
      for (size_t k = 0; k < region[i].size(); k++) {
        float dis_x = fabsf(tmp[j].pt.x - region[i][k].pt.x);
        float dis_y = fabsf(tmp[j].pt.y - region[i][k].pt.y);
        if (dis_x <= boundary_dis || dis_y <= boundary_dis) ok = 0;
      }
      if (ok) {
        region[i].push_back(tmp[j]);
        count++;
        if (count >= max_region_feature_num) break;
      }
    }
  }
  for (int i = 0; i < 16; i++)
    for (size_t j = 0; j < region[i].size(); j++) newPts.push_back(region[i][j].pt);
}

void FeatureDEM::redetect(const uint8_t* img, const std::vector<Pt2f>& existedPts, std::vector<Pt2f>& newPts) const {
  newPts.clear();
  std::vector<Scored> regionKeyPts[16];
  fillIntoRegion(img, existedPts, regionKeyPts, true);
  std::vector<float> xy((size_t)gftt_num * 2 + 2);
  int nf = good_features_to_track(img, width, height, gftt_num, gftt_ql, gftt_dis, xy.data());
  std::vector<Pt2f> features(nf);
  for (int i = 0; i < nf; i++) features[i] = {xy[2 * i], xy[2 * i + 1]};
  std::vector<Scored> prepare[16];
  fillIntoRegion(img, features, prepare, false);
  for (int i = 0; i < 16; i++) {
    std::sort(prepare[i].begin(), prepare[i].end(), sortbysecdesc);  // (feature_dem.cpp:170)
    for (size_t j = 0; j < prepare[i].size(); j++) {
      int noFeatureNearby = 1;
      // cv::Point pt = Point2f  (rounds; GFTT output is integral already), feature_dem.cpp:174
      int px = (int)lrintf(prepare[i][j].pt.x), py = (int)lrintf(prepare[i][j].pt.y);
      for (size_t k = 0; k < regionKeyPts[i].size(); k++) {
        float dis_x = fabsf((float)px - regionKeyPts[i][k].pt.x);
        float dis_y = fabsf((float)py - regionKeyPts[i][k].pt.y);
        if (dis_x <= boundary_dis || dis_y <= boundary_dis) noFeatureNearby = 0;
      }
      if (noFeatureNearby) {
        regionKeyPts[i].push_back({{(float)px, (float)py}, 999999.0f});
        newPts.push_back({(float)px, (float)py});
        if (regionKeyPts[i].size() >= max_region_feature_num) break;
      }
    }
  }
}

}  // namespace ref

// ------------------------------------------------------------------------------------------ C entry points (ctypes)
extern "C" {
void ref_equalize_hist(const uint8_t* src, uint8_t* dst, int w, int h) { ref::equalize_hist(src, dst, w, h); }
void ref_cvt_bgr_to_gray(const uint8_t* src, int channels, uint8_t* dst, int w, int h) {
  ref::cvt_bgr_to_gray(src, channels, dst, w, h);
}
void ref_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) { ref::pyr_down(src, w, h, dst); }
int ref_lk_num_levels(int w, int h, int win, int max_level) { return ref::lk_num_levels(w, h, win, max_level); }
void ref_calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
                                  float* next_pts, uint8_t* status, int n, int win, int max_level, int max_iter,
                                  double eps, int use_initial_flow, float min_eig_thr) {
  ref::calc_optical_flow_pyr_lk(prev, next, w, h, prev_pts, next_pts, status, n, win, max_level, max_iter, eps,
                                use_initial_flow, min_eig_thr);
}
void ref_min_eigen_map(const uint8_t* img, int w, int h, float* eig) { ref::min_eigen_map(img, w, h, eig); }
int ref_good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double q, double min_dist,
                               float* out_xy) {
  return ref::good_features_to_track(img, w, h, max_corners, q, min_dist, out_xy);
}
int ref_feature_dem_detect(const uint8_t* img, int w, int h, const double* f_para, float* out_xy, int cap) {
  ref::FeatureDEM dem(w, h, f_para);
  std::vector<ref::Pt2f> pts;
  dem.detect(img, pts);
  int n = (int)std::min<size_t>(pts.size(), (size_t)cap);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = pts[i].x;
    out_xy[2 * i + 1] = pts[i].y;
  }
  return (int)pts.size();
}
int ref_feature_dem_redetect(const uint8_t* img, int w, int h, const double* f_para, const double* existed_xy,
                             int n_existed, float* out_xy, int cap) {
  ref::FeatureDEM dem(w, h, f_para);
  std::vector<ref::Pt2f> ex(n_existed), pts;
  for (int i = 0; i < n_existed; i++) ex[i] = {(float)existed_xy[2 * i], (float)existed_xy[2 * i + 1]};
  dem.redetect(img, ex, pts);
  int n = (int)std::min<size_t>(pts.size(), (size_t)cap);
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = pts[i].x;
    out_xy[2 * i + 1] = pts[i].y;
  }
  return (int)pts.size();
}
}
