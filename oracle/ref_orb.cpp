// ORACLE (test infrastructure, NOT product code): CPU restatement of the ORB extraction + Hamming matching that the
// reference runs once per keyframe in its loop-closing nodelet:
//   cv::ORB::create(1000,1.2f,8,31,0,2,cv::ORB::HARRIS_SCORE,31,20)->detectAndCompute   src/backend/vo_loopclosing.cpp:242-243
//   cv::BFMatcher(NORM_HAMMING,false).knnMatch(..,2) both ways + mutual/ratio test      src/backend/vo_loopclosing.cpp:603-639
// The arithmetic lives in OpenCV (absent from /root/reference; pinned only as "3 EXACT, else 4", CMakeLists.txt:46-49).
// This file restates the published algorithm of the OpenCV 3.2/3.3 line (modules/features2d/src/{orb,fast,fast_score}.cpp,
// modules/imgproc/src/{imgwarp,smooth,filter}.cpp, modules/core/src/mathfuncs_core): pyramid by INTER_LINEAR resize of the
// previous level (11-bit fixed-point bilinear), FAST-9/16 with 3x3 non-max suppression, border filter, retainBest(2n) by
// FAST score, Harris response (7x7 block, k = 0.04), retainBest(n), intensity-centroid angle (fastAtan2), 7x7 sigma-2
// Gaussian blur (8-bit fixed-point separable filter), steered BRIEF with WTA_K = 2.
// PARITY UNPINNED: no OpenCV here, no golden vectors in the reference.  Two documented deviations:
//   * keypoint ORDER: OpenCV's order falls out of std::nth_element; here it is level-major, raster within a level
//     (the keypoint SET is the one KeyPointsFilter::retainBest defines: everything >= the n-th best response);
//   * the 256-pair sampling pattern: OpenCV's learned bit_pattern_31_ table is not available offline; the default here
//     is OpenCV's own fallback generator makeRandomPattern(31, ., 512) (RNG 0x34985739); a caller-supplied table
//     (e.g. bit_pattern_31_ for DBoW vocabulary compatibility) is used verbatim.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ref {

static inline int cv_round(double v) { return (int)std::lrint(v); }  // round half to even, like cvRound
static inline int cv_floor(double v) { return (int)std::floor(v); }
static inline int cv_ceil(double v) { return (int)std::ceil(v); }
static inline short sat_short(float v) {
  int i = cv_round(v);
  return (short)std::min(32767, std::max(-32768, i));
}

// ---- cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (imgwarp.cpp: resizeGeneric_ / HResizeLinear / VResizeLinear)
void resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  const int ONE = 2048;  // INTER_RESIZE_COEF_SCALE
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) fx = 0, sx = 0;
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) fx = 0, sx = sw - 1;
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = sat_short((1.f - fx) * ONE);
    ialpha[dx * 2 + 1] = sat_short(fx * ONE);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = sat_short((1.f - fy) * ONE);
    ibeta[dy * 2 + 1] = sat_short(fy * ONE);
  }
  std::vector<int> r0(dw), r1(dw);
  auto hrow = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);
    const uint8_t* S = src + (size_t)sy * sw;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      out[dx] = dx < xmax ? S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1] : S[sx] * ONE;
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    hrow(yofs[dy], r0);
    hrow(yofs[dy] + 1, r1);
    int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    for (int dx = 0; dx < dw; dx++)
      dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---- FAST-9/16 corner score map (fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>): 0 where not a corner, else the
// largest threshold at which the pixel is still a corner (>= threshold).  Rows/cols closer than 3 px to the edge are not tested.
static const int FAST_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int FAST_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

void fast_score_map(const uint8_t* img, int w, int h, int threshold, uint8_t* score) {
  std::memset(score, 0, (size_t)w * h);
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int v = img[(size_t)y * w + x];
      int d[25];
      for (int k = 0; k < 25; k++) d[k] = v - img[(size_t)(y + FAST_DY[k & 15]) * w + x + FAST_DX[k & 15]];
      // corner test: 9 contiguous pixels all darker than v - t (d > t) or all brighter than v + t (d < -t)
      bool corner = false;
      for (int s = 0; s < 16 && !corner; s++) {
        bool dark = true, bright = true;
        for (int k = s; k < s + 9; k++) {
          dark = dark && d[k] > threshold;
          bright = bright && d[k] < -threshold;
        }
        corner = dark || bright;
      }
      if (!corner) continue;
      int a0 = threshold;
      for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = std::min(a, d[k + j]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
      }
      int b0 = -a0;
      for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = std::max(b, d[k + j]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
      }
      score[(size_t)y * w + x] = (uint8_t)(-b0 - 1);
    }
}

struct FastKp {
  int x, y, score;
};
// FAST with nonmaxSuppression=true: strict 3x3 maximum of the score map; raster order (the order fast.cpp emits).
void fast_detect(const uint8_t* img, int w, int h, int threshold, std::vector<FastKp>& out) {
  std::vector<uint8_t> sc((size_t)w * h);
  fast_score_map(img, w, h, threshold, sc.data());
  out.clear();
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      int s = sc[(size_t)y * w + x];
      if (!s) continue;
      bool mx = true;
      for (int dy = -1; dy <= 1 && mx; dy++)
        for (int dx = -1; dx <= 1; dx++)
          if ((dx || dy) && sc[(size_t)(y + dy) * w + x + dx] >= s) {
            mx = false;
            break;
          }
      if (mx) out.push_back({x, y, s});
    }
}

// ---- GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8UC1: getGaussianKernel(7, 2, CV_32F) -> 8-bit fixed
// point (filter.cpp createSeparableLinearFilter: kernels * 256 rounded, column pass (sum + 2^15) >> 16, saturated)
void gauss_kernel7_fixed(int k[7]) {
  float cf[7];
  double sum = 0, scale2X = -0.5 / (2.0 * 2.0);
  for (int i = 0; i < 7; i++) {
    double x = i - 3.0;
    cf[i] = (float)std::exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < 7; i++) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = cv_round((double)cf[i] * 256.0);
  }
}
static inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}
void gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) {
  int k[7];
  gauss_kernel7_fixed(k);
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = -3; i <= 3; i++) s += k[i + 3] * src[(size_t)y * w + reflect101(x + i, w)];
      tmp[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = -3; i <= 3; i++) s += k[i + 3] * tmp[(size_t)reflect101(y + i, h) * w + x];
      int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * w + x] = (uint8_t)std::min(255, std::max(0, v));
    }
}

// ---- cv::fastAtan2 (degrees, 3.x polynomial)
float fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
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

// ---- orb.cpp helpers
void orb_umax(int half_patch, int* umax /*[half_patch+2]*/) {
  int vmax = cv_floor(half_patch * std::sqrt(2.0) / 2 + 1);
  int vmin = cv_ceil(half_patch * std::sqrt(2.0) / 2);
  for (int v = 0; v <= half_patch + 1; v++) umax[v] = 0;
  for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)half_patch * half_patch - v * v));
  for (int v = half_patch, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

float orb_ic_angle(const uint8_t* img, int w, int x, int y, const int* umax, int half_k) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = img + (size_t)y * w + x;
  for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
  for (int v = 1; v <= half_k; ++v) {
    int v_sum = 0, d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * w], val_minus = center[u - v * w];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

float orb_harris(const uint8_t* img, int w, int x0, int y0, int blockSize, float harris_k) {
  int r = blockSize / 2;
  float scale = 1.f / ((1 << 2) * blockSize * 255.f);
  float scale_sq_sq = scale * scale * scale * scale;
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < blockSize; i++)
    for (int j = 0; j < blockSize; j++) {
      const uint8_t* p = img + (size_t)(y0 - r + i) * w + (x0 - r + j);
      int Ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
      int Iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

// makeRandomPattern(patchSize, pattern, npoints) with cv::RNG(0x34985739)
void orb_default_pattern(int8_t* pat /*[512][2]*/) {
  uint64_t state = 0x34985739;
  auto next = [&]() {
    state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
    return (unsigned)state;
  };
  const int patch = 31;
  for (int i = 0; i < 512; i++) {
    pat[2 * i] = (int8_t)(int)(next() % (unsigned)(patch / 2 + 1 + patch / 2) + (-patch / 2));
    pat[2 * i + 1] = (int8_t)(int)(next() % (unsigned)(patch / 2 + 1 + patch / 2) + (-patch / 2));
  }
}

void orb_level_sizes(int w, int h, int nlevels, float scale_factor, int* lw, int* lh, float* lscale) {
  for (int l = 0; l < nlevels; l++) {
    float s = (float)std::pow((double)scale_factor, (double)l);  // getScale(level, 0, scaleFactor)
    lscale[l] = s;
    lw[l] = cv_round(w / s);
    lh[l] = cv_round(h / s);
  }
}

void orb_features_per_level(int nfeatures, int nlevels, float scale_factor, int* n_per_level) {
  float factor = (float)(1.0 / scale_factor);
  float ndesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    n_per_level[l] = cv_round(ndesired);
    sum += n_per_level[l];
    ndesired *= factor;
  }
  n_per_level[nlevels - 1] = std::max(nfeatures - sum, 0);
}

struct OrbKp {
  float x, y, size, angle, response;
  int octave;
};

// KeyPointsFilter::retainBest as a set: everything with response >= the n-th largest response
template <class T, class F>
static void retain_best(std::vector<T>& v, int n, F resp) {
  if (n < 0 || (int)v.size() <= n) return;
  if (n == 0) {
    v.clear();
    return;
  }
  std::vector<float> r;
  for (auto& e : v) r.push_back(resp(e));
  std::nth_element(r.begin(), r.begin() + (n - 1), r.end(), std::greater<float>());
  float thr = r[n - 1];
  std::vector<T> out;
  for (auto& e : v)
    if (resp(e) >= thr) out.push_back(e);
  v.swap(out);
}

int orb_detect_and_compute(const uint8_t* img, int w, int h, int nfeatures, float scale_factor, int nlevels,
                           int fast_threshold, const int8_t* pattern /*[512][2] or null*/, OrbKp* kps, uint8_t* desc,
                           int cap, uint8_t* pyr_out /*optional: all levels concatenated*/,
                           uint8_t* blur_out /*optional*/) {
  const int edge = 31, patch = 31, half = 15;
  std::vector<int> lw(nlevels), lh(nlevels), npl(nlevels);
  std::vector<float> ls(nlevels);
  orb_level_sizes(w, h, nlevels, scale_factor, lw.data(), lh.data(), ls.data());
  orb_features_per_level(nfeatures, nlevels, scale_factor, npl.data());
  int8_t defpat[1024];
  if (!pattern) {
    orb_default_pattern(defpat);
    pattern = defpat;
  }
  int umax[17];
  orb_umax(half, umax);
  std::vector<std::vector<uint8_t>> pyr(nlevels);
  pyr[0].assign(img, img + (size_t)w * h);
  for (int l = 1; l < nlevels; l++) {
    pyr[l].resize((size_t)lw[l] * lh[l]);
    resize_linear_u8(pyr[l - 1].data(), lw[l - 1], lh[l - 1], pyr[l].data(), lw[l], lh[l]);
  }
  int n_out = 0;
  size_t off = 0;
  for (int l = 0; l < nlevels; l++) {
    int W = lw[l], H = lh[l];
    const uint8_t* L = pyr[l].data();
    if (pyr_out) std::memcpy(pyr_out + off, L, (size_t)W * H);
    std::vector<FastKp> fk;
    fast_detect(L, W, H, fast_threshold, fk);
    // KeyPointsFilter::runByImageBorder(keypoints, img.size(), edgeThreshold)
    std::vector<FastKp> in;
    if (H > 2 * edge && W > 2 * edge)
      for (auto& k : fk)
        if (k.x >= edge && k.x < W - edge && k.y >= edge && k.y < H - edge) in.push_back(k);
    retain_best(in, 2 * npl[l], [](const FastKp& k) { return (float)k.score; });
    struct Cand {
      int x, y;
      float r;
    };
    std::vector<Cand> c;
    for (auto& k : in) c.push_back({k.x, k.y, orb_harris(L, W, k.x, k.y, 7, 0.04f)});
    retain_best(c, npl[l], [](const Cand& k) { return k.r; });
    std::vector<uint8_t> blur((size_t)W * H);
    gaussian_blur7(L, W, H, blur.data());
    if (blur_out) std::memcpy(blur_out + off, blur.data(), (size_t)W * H);
    off += (size_t)W * H;
    float sf = ls[l];
    for (auto& k : c) {
      if (n_out >= cap) return -1;
      OrbKp kp;
      kp.angle = orb_ic_angle(L, W, k.x, k.y, umax, half);
      kp.x = (float)k.x;
      kp.y = (float)k.y;
      if (l != 0) {
        kp.x *= sf;
        kp.y *= sf;
      }
      kp.size = patch * sf;
      kp.response = k.r;
      kp.octave = l;
      // computeOrbDescriptors
      float scale = 1.f / sf;
      float angle = kp.angle;
      angle *= (float)(3.1415926535897932384626433832795 / 180.f);
      float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
      int cx = cv_round(kp.x * scale), cy = cv_round(kp.y * scale);
      const uint8_t* center = blur.data() + (size_t)cy * W + cx;
      auto val = [&](int idx) {
        float px = pattern[2 * idx], py = pattern[2 * idx + 1];
        float x = px * a - py * b;
        float y = px * b + py * a;
        int ix = cv_round(x), iy = cv_round(y);
        return (int)center[iy * W + ix];
      };
      uint8_t* d = desc + (size_t)n_out * 32;
      for (int i = 0; i < 32; i++) {
        int byte = 0;
        for (int j = 0; j < 8; j++) {
          int t0 = val(16 * i + 2 * j), t1 = val(16 * i + 2 * j + 1);
          byte |= (t0 < t1) << j;
        }
        d[i] = (uint8_t)byte;
      }
      kps[n_out++] = kp;
    }
  }
  return n_out;
}

// ---- BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2): ascending distance, ties keep the lower train index
void hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx /*[nq][2]*/, int* dist /*[nq][2]*/) {
  for (int i = 0; i < nq; i++) {
    int bi[2] = {-1, -1}, bd[2] = {INT32_MAX, INT32_MAX};
    for (int j = 0; j < nt; j++) {
      int d = 0;
      for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(q[i * 32 + k] ^ t[j * 32 + k]));
      if (d < bd[1]) {
        if (d < bd[0]) {
          bd[1] = bd[0];
          bi[1] = bi[0];
          bd[0] = d;
          bi[0] = j;
        } else {
          bd[1] = d;
          bi[1] = j;
        }
      }
    }
    idx[2 * i] = bi[0];
    idx[2 * i + 1] = bi[1];
    dist[2 * i] = bd[0];
    dist[2 * i + 1] = bd[1];
  }
}

// vo_loopclosing.cpp:603-639: mutual best match + ratio test, in query order -> pairs (query idx in A, train idx in B)
int orb_match_mutual_ratio(const uint8_t* a, int na, const uint8_t* b, int nb, double ratio_max, int* pairs) {
  if (na < 2 || nb < 2) return 0;
  std::vector<int> i12(2 * na), d12(2 * na), i21(2 * nb), d21(2 * nb);
  hamming_knn2(a, na, b, nb, i12.data(), d12.data());
  hamming_knn2(b, nb, a, na, i21.data(), d21.data());
  int n = 0;
  for (int i = 0; i < na; i++) {
    int t = i12[2 * i];
    if (i21[2 * t] != i) continue;
    float d0 = (float)d12[2 * i], d1 = (float)d12[2 * i + 1];
    if (d0 * 1.0 / d1 < (double)(float)ratio_max) {
      pairs[2 * n] = i;
      pairs[2 * n + 1] = t;
      n++;
    }
  }
  return n;
}

}  // namespace ref

extern "C" {
void ref_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  ref::resize_linear_u8(src, sw, sh, dst, dw, dh);
}
void ref_fast_score_map(const uint8_t* img, int w, int h, int thr, uint8_t* score) { ref::fast_score_map(img, w, h, thr, score); }
int ref_fast_detect(const uint8_t* img, int w, int h, int thr, int* out /*[cap][3]*/, int cap) {
  std::vector<ref::FastKp> k;
  ref::fast_detect(img, w, h, thr, k);
  int n = (int)std::min<size_t>(k.size(), cap);
  for (int i = 0; i < n; i++) out[3 * i] = k[i].x, out[3 * i + 1] = k[i].y, out[3 * i + 2] = k[i].score;
  return (int)k.size();
}
void ref_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) { ref::gaussian_blur7(src, w, h, dst); }
void ref_gauss_kernel7_fixed(int* k) { ref::gauss_kernel7_fixed(k); }
float ref_fast_atan2(float y, float x) { return ref::fast_atan2(y, x); }
void ref_orb_umax(int half_patch, int* umax) { ref::orb_umax(half_patch, umax); }
float ref_orb_ic_angle(const uint8_t* img, int w, int x, int y) {
  int umax[17];
  ref::orb_umax(15, umax);
  return ref::orb_ic_angle(img, w, x, y, umax, 15);
}
float ref_orb_harris(const uint8_t* img, int w, int x, int y) { return ref::orb_harris(img, w, x, y, 7, 0.04f); }
void ref_orb_default_pattern(int8_t* pat) { ref::orb_default_pattern(pat); }
void ref_orb_level_sizes(int w, int h, int nlevels, float sf, int* lw, int* lh, float* ls) {
  ref::orb_level_sizes(w, h, nlevels, sf, lw, lh, ls);
}
void ref_orb_features_per_level(int nfeatures, int nlevels, float sf, int* n) { ref::orb_features_per_level(nfeatures, nlevels, sf, n); }
int ref_orb_detect_and_compute(const uint8_t* img, int w, int h, int nfeatures, float scale_factor, int nlevels,
                               int fast_threshold, const int8_t* pattern, float* kps /*[cap][6]*/, uint8_t* desc, int cap,
                               uint8_t* pyr_out, uint8_t* blur_out) {
  std::vector<ref::OrbKp> k(cap);
  int n = ref::orb_detect_and_compute(img, w, h, nfeatures, scale_factor, nlevels, fast_threshold, pattern, k.data(), desc,
                                      cap, pyr_out, blur_out);
  for (int i = 0; i < n; i++) {
    kps[6 * i] = k[i].x, kps[6 * i + 1] = k[i].y, kps[6 * i + 2] = k[i].size, kps[6 * i + 3] = k[i].angle;
    kps[6 * i + 4] = k[i].response, kps[6 * i + 5] = (float)k[i].octave;
  }
  return n;
}
void ref_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist) {
  ref::hamming_knn2(q, nq, t, nt, idx, dist);
}
int ref_orb_match_mutual_ratio(const uint8_t* a, int na, const uint8_t* b, int nb, double ratio_max, int* pairs) {
  return ref::orb_match_mutual_ratio(a, na, b, nb, ratio_max, pairs);
}
}
