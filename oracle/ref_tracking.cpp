// ORACLE (test infrastructure, NOT product code) -- FLVIS front-end state machine, IMU filter, config.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows (file:line in /root/reference):
//   F2FTracking::init/imu_feed/image_feed/init_frame   src/frontend/f2f_tracking.cpp:5-453
//   LKORBTracking::tracking                            src/processing/lkorb_tracking.cpp:9-202 (mirror-index quirk A1 kept)
//   CameraFrame::*                                     src/processing/camera_frame.cpp:18-529
//   LandMarkInFrame ctor / id counter                  src/processing/landmark.cpp:3-39
//   VIMOTION                                           src/processing/vi_motion.cpp:3-464, src/utils/include/kinetic_math.h
//   TrackingNodeletClass::onInit (config -> camera)    src/frontend/vo_tracking.cpp:114-306
//   cv::stereoRectify(CALIB_ZERO_DISPARITY, alpha 0)   OpenCV 3.x cvStereoRectify, restated (parity unpinned)
// parity unpinned: the reference ships no tests; this restatement is exercised end-to-end on synthetic streams in tests/.
#include "ref_tracking.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

namespace ref {

// ------------------------------------------------------------------------------------------ glibc rand()
void GlibcRand::seed(unsigned s) {
  int32_t st[344 + 34];
  std::vector<int32_t> v(344);
  v[0] = (int32_t)s;
  for (int i = 1; i < 31; i++) {
    int64_t w = (16807LL * v[i - 1]) % 2147483647;
    if (w < 0) w += 2147483647;
    v[i] = (int32_t)w;
  }
  for (int i = 31; i < 34; i++) v[i] = v[i - 31];
  for (int i = 34; i < 344; i++) v[i] = (int32_t)((uint32_t)v[i - 31] + (uint32_t)v[i - 3]);
  for (int i = 0; i < 34; i++) r[i] = v[344 - 34 + i];
  pos = 0;
  (void)st;
}
int GlibcRand::next() {
  // r holds the last 34 outputs (ring); new = r[-31] + r[-3]
  int32_t a = r[(pos + 34 - 31) % 34], b = r[(pos + 34 - 3) % 34];
  int32_t n = (int32_t)((uint32_t)a + (uint32_t)b);
  r[pos] = n;
  pos = (pos + 1) % 34;
  return (int)(((uint32_t)n) >> 1);
}

// ------------------------------------------------------------------------------------------ config
static SE3 se3_from_mat44(const double* m) {
  Mat3 R = {{{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}}};
  return se3_from_mat(R, {m[3], m[7], m[11]});
}

// cvRodrigues2 (OpenCV 3.x calibration.cpp) as cvStereoRectify calls it, with the formulas in OpenCV's order.  Matrix -> vector:
// r = (R32 - R23, R13 - R31, R21 - R12), s = sqrt(r.r / 4), c = (trace - 1) / 2, theta = acos(c), r *= theta / (2 s)  (OpenCV first
// re-orthonormalises R by an SVD; the rig rotations are orthonormal to rounding).  Vector -> matrix: R = c I + (1 - c) r r^T + s [r]x.
static Vec3 cv_rodrigues_from_mat(const Mat3& R) {
  double rx = R.m[2][1] - R.m[1][2], ry = R.m[0][2] - R.m[2][0], rz = R.m[1][0] - R.m[0][1];
  const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R.m[0][0] + R.m[1][1] + R.m[2][2] - 1) * 0.5;
  c = c > 1. ? 1. : (c < -1. ? -1. : c);
  const double theta = std::acos(c);
  if (s < 1e-5) {
    if (c > 0) return {0, 0, 0};
    double t;
    t = (R.m[0][0] + 1) * 0.5;
    rx = std::sqrt(std::max(t, 0.));
    t = (R.m[1][1] + 1) * 0.5;
    ry = std::sqrt(std::max(t, 0.)) * (R.m[0][1] < 0 ? -1. : 1.);
    t = (R.m[2][2] + 1) * 0.5;
    rz = std::sqrt(std::max(t, 0.)) * (R.m[0][2] < 0 ? -1. : 1.);
    if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R.m[1][2] > 0) != (ry * rz > 0)) rz = -rz;
    const double k = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
    return {rx * k, ry * k, rz * k};
  }
  double vth = 1 / (2 * s);
  vth *= theta;
  return {rx * vth, ry * vth, rz * vth};
}
static Mat3 cv_rodrigues_to_mat(Vec3 r) {
  const double theta = std::sqrt(r.x * r.x + r.y * r.y + r.z * r.z);
  if (theta < DBL_EPSILON) return mat3_identity();
  const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = 1. / theta;
  const double x = r.x * itheta, y = r.y * itheta, z = r.z * itheta;
  const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
  const double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Mat3 R;
  for (int k = 0; k < 9; k++) R.m[k / 3][k % 3] = c * I[k] + c1 * rrt[k] + s * r_x[k];
  return R;
}

// cvStereoRectify (OpenCV 3.x), flags = CALIB_ZERO_DISPARITY, alpha = 0, newImageSize = imageSize
static void stereo_rectify(const double K1[4], const double D1[4], const double K2[4], const double D2[4], int nx, int ny,
                           const Mat3& R, Vec3 T, Mat3& R1, Mat3& R2, double P1[12], double P2[12]) {
  Vec3 om = cv_rodrigues_from_mat(R);
  om = -0.5 * om;
  Mat3 r_r = cv_rodrigues_to_mat(om);
  Vec3 t = r_r * T;
  int idx = std::fabs(t[0]) > std::fabs(t[1]) ? 0 : 1;
  double c = t[idx], nt = norm(t);
  Vec3 uu{0, 0, 0};
  uu[idx] = c > 0 ? 1 : -1;
  Vec3 ww = cross(t, uu);
  double nw = norm(ww);
  if (nw > 0.0) ww = (std::acos(std::fabs(c) / nt) / nw) * ww;
  Mat3 wR = cv_rodrigues_to_mat(ww);
  R1 = wR * transpose(r_r);
  R2 = wR * r_r;
  t = R2 * T;
  double fc_new = DBL_MAX;
  const double* Ks[2] = {K1, K2};
  const double* Ds[2] = {D1, D2};
  for (int k = 0; k < 2; k++) {
    double dk1 = Ds[k][0];
    double fc = idx == 0 ? Ks[k][1] : Ks[k][0];  // A(idx^1, idx^1)
    if (dk1 < 0) fc *= 1 + dk1 * (nx * nx + ny * ny) / (4 * fc * fc);
    fc_new = std::min(fc_new, fc);
  }
  double ccx[2], ccy[2];
  const Mat3* Rs[2] = {&R1, &R2};
  const Mat3 I = mat3_identity();
  for (int k = 0; k < 2; k++) {
    float pts[8];
    for (int i = 0; i < 4; i++) {
      int j = (i < 2) ? 0 : 1;
      pts[2 * i] = (float)((i % 2) * (nx - 1));
      pts[2 * i + 1] = (float)(j * (ny - 1));
    }
    double Pn[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    float und[8];
    undistort_points(pts, 4, Ks[k], Ds[k], I, Pn, und);  // normalised coordinates
    double ax = 0, ay = 0;
    for (int i = 0; i < 4; i++) {
      Vec3 X = (*Rs[k]) * Vec3{(double)und[2 * i], (double)und[2 * i + 1], 1.0};
      ax += (double)(float)(fc_new * X.x / X.z);
      ay += (double)(float)(fc_new * X.y / X.z);
    }
    ccx[k] = (nx - 1) / 2 - ax / 4;
    ccy[k] = (ny - 1) / 2 - ay / 4;
  }
  ccx[0] = ccx[1] = (ccx[0] + ccx[1]) * 0.5;  // CALIB_ZERO_DISPARITY
  ccy[0] = ccy[1] = (ccy[0] + ccy[1]) * 0.5;
  for (int i = 0; i < 12; i++) P1[i] = P2[i] = 0;
  P1[0] = P1[5] = fc_new;
  P1[2] = ccx[0];
  P1[6] = ccy[0];
  P1[10] = 1;
  P2[0] = P2[5] = fc_new;
  P2[2] = ccx[1];
  P2[6] = ccy[1];
  P2[10] = 1;
  P2[4 * idx + 3] = t[idx] * fc_new;
  // alpha = 0: scale so that only valid pixels remain (inner rectangles of a 9x9 grid)
  double inner[2][4];
  double* Ps[2] = {P1, P2};
  for (int k = 0; k < 2; k++) {
    const int N = 9;
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    for (int y = 0; y < N; y++)
      for (int x = 0; x < N; x++) {
        float p[2] = {(float)x * nx / (N - 1), (float)y * ny / (N - 1)}, q[2];
        undistort_points(p, 1, Ks[k], Ds[k], *Rs[k], Ps[k], q);
        if (x == 0) iX0 = std::max(iX0, q[0]);
        if (x == N - 1) iX1 = std::min(iX1, q[0]);
        if (y == 0) iY0 = std::max(iY0, q[1]);
        if (y == N - 1) iY1 = std::min(iY1, q[1]);
      }
    inner[k][0] = iX0;
    inner[k][1] = iY0;
    inner[k][2] = iX1 - iX0;
    inner[k][3] = iY1 - iY0;
  }
  double cx1_0 = ccx[0], cy1_0 = ccy[0], cx2_0 = ccx[1], cy2_0 = ccy[1];
  double cx1 = cx1_0, cy1 = cy1_0, cx2 = cx2_0, cy2 = cy2_0;  // newImgSize == imageSize
  double s0 = std::max(std::max(std::max(cx1 / (cx1_0 - inner[0][0]), cy1 / (cy1_0 - inner[0][1])),
                                (nx - cx1) / (inner[0][0] + inner[0][2] - cx1_0)),
                       (ny - cy1) / (inner[0][1] + inner[0][3] - cy1_0));
  s0 = std::max(std::max(std::max(std::max(cx2 / (cx2_0 - inner[1][0]), cy2 / (cy2_0 - inner[1][1])),
                                  (nx - cx2) / (inner[1][0] + inner[1][2] - cx2_0)),
                         (ny - cy2) / (inner[1][1] + inner[1][3] - cy2_0)),
                s0);
  double s = s0;  // alpha = 0
  fc_new *= s;
  P1[0] = P1[5] = fc_new;
  P1[2] = cx1;
  P1[6] = cy1;
  P2[0] = P2[5] = fc_new;
  P2[2] = cx2;
  P2[6] = cy2;
  P2[4 * idx + 3] = s * P2[4 * idx + 3];
}

bool config_finalize(Config& c) {
  // vi_type.h: 0 d435i depth, 1 euroc, 2 d435+pixhawk depth, 3 d435i stereo, 4 kitti stereo, 5 d435 stereo + pixhawk
  switch (c.type_of_vi) {
    case 1:
      c.cam_type = STEREO_UNRECT;
      c.has_imu_type = 1;
      c.skip_first_n_imgs = 0;
      c.need_equal_hist = 1;
      break;
    case 3:
    case 5:
      c.cam_type = STEREO_RECT;
      c.has_imu_type = 1;
      c.skip_first_n_imgs = 50;
      c.need_equal_hist = 0;
      break;
    case 0:
    case 2: {  // vo_tracking.cpp:142-170: DEPTH_D435, pinhole K from cam0_intrinsics, no distortion, 50 skipped frames
      c.cam_type = DEPTH_D435;
      c.has_imu_type = 1;
      c.skip_first_n_imgs = 50;
      c.need_equal_hist = 0;
      const double P[12] = {c.cam0_intrinsics[0], 0, c.cam0_intrinsics[2], 0, 0, c.cam0_intrinsics[1], c.cam0_intrinsics[3], 0,
                            0, 0, 1, 0};
      memcpy(c.P0, P, sizeof(P));
      memset(c.P1, 0, sizeof(c.P1));
      for (int i = 0; i < 9; i++) c.R0[i] = c.R1[i] = (i % 4 == 0) ? 1.0 : 0.0;
      return true;
    }
    case 4: {  // vo_tracking.cpp:146,265-306: KITTI, STEREO_RECT, imu NONE, init(dc, SE3(), ..., 0, false)
      c.cam_type = STEREO_RECT;
      c.has_imu_type = 3;
      c.skip_first_n_imgs = 0;
      c.need_equal_hist = 0;
      // K0 = K1 = K0_rect = P0.rowRange(0,3).colRange(0,3) (:284), D = 0 (:283), R = I (:289,:290)
      const double fx = c.P0[0], fy = c.P0[5], cx = c.P0[2], cy = c.P0[6];
      if (!(fx > 0) || !(fy > 0)) return false;
      const double K[4] = {fx, fy, cx, cy};
      for (int i = 0; i < 4; i++) {
        c.cam0_intrinsics[i] = c.cam1_intrinsics[i] = K[i];
        c.cam0_distortion[i] = c.cam1_distortion[i] = 0;
      }
      for (int i = 0; i < 9; i++) c.R0[i] = c.R1[i] = (i % 4 == 0) ? 1.0 : 0.0;
      // mat_T_c0_c1 = K_inverse * P1_, rotation block set to identity (:272-276).  K.inverse() on a fixed 3x3 is Eigen's
      // cofactor formula: inverse(i,j) = cofactor(j,i) * (1 / det), det = sum over column 0 of cofactor * entry
      const double Km[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
      auto cof = [&](int i, int j) {  // cofactor_3x3<i,j>: m(i1,j1) * m(i2,j2) - m(i1,j2) * m(i2,j1), i1=(i+1)%3 ...
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return Km[i1][j1] * Km[i2][j2] - Km[i1][j2] * Km[i2][j1];
      };
      const double c0[3] = {cof(0, 0), cof(1, 0), cof(2, 0)};
      const double det = (c0[0] * Km[0][0] + c0[1] * Km[1][0]) + c0[2] * Km[2][0];
      const double invdet = 1.0 / det;
      double inv[3][3];
      for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) inv[j][i] = cof(i, j) * invdet;
      const double pc[3] = {c.P1[3], c.P1[7], c.P1[11]};
      double t[3];
      for (int r = 0; r < 3; r++) t[r] = (inv[r][0] * pc[0] + inv[r][1] * pc[1]) + inv[r][2] * pc[2];
      const double T01[16] = {1, 0, 0, t[0], 0, 1, 0, t[1], 0, 0, 1, t[2], 0, 0, 0, 1};
      const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      memcpy(c.T_cam0_cam1, T01, sizeof(T01));
      memcpy(c.T_imu_cam0, eye, sizeof(eye));
      return true;
    }
    default:
      return false;
  }
  SE3 T_c0_c1 = se3_from_mat44(c.T_cam0_cam1);
  SE3 T_c1_c0 = se3_inverse(T_c0_c1);
  Mat3 R0, R1;
  stereo_rectify(c.cam0_intrinsics, c.cam0_distortion, c.cam1_intrinsics, c.cam1_distortion, c.image_width,
                 c.image_height, quat_to_mat(T_c1_c0.q), T_c1_c0.t, R0, R1, c.P0, c.P1);
  for (int i = 0; i < 9; i++) {
    c.R0[i] = R0.m[i / 3][i % 3];
    c.R1[i] = R1.m[i / 3][i % 3];
  }
  return true;
}

// flat yaml: "key: scalar" or "key: [a, b, ...]" possibly continued over lines; '#' starts a comment
bool config_load_yaml(const char* path, Config& c, char* err, int errlen) {
  std::ifstream f(path);
  if (!f) {
    snprintf(err, errlen, "cannot open %s", path);
    return false;
  }
  std::map<std::string, std::vector<double>> kv;
  std::map<std::string, std::string> raw;
  std::string line, key, acc;
  bool in_list = false;
  auto flush_list = [&](const std::string& k, std::string s) {
    for (char& ch : s)
      if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
    std::istringstream is(s);
    double v;
    std::vector<double> out;
    while (is >> v) out.push_back(v);
    kv[k] = out;
  };
  while (std::getline(f, line)) {
    size_t h = line.find('#');
    if (h != std::string::npos) line = line.substr(0, h);
    if (in_list) {
      acc += " " + line;
      if (line.find(']') != std::string::npos) {
        flush_list(key, acc);
        in_list = false;
      }
      continue;
    }
    size_t col = line.find(':');
    if (col == std::string::npos) continue;
    key = line.substr(0, col);
    key.erase(0, key.find_first_not_of(" \t"));
    key.erase(key.find_last_not_of(" \t") + 1);
    std::string val = line.substr(col + 1);
    if (val.find('[') != std::string::npos) {
      acc = val;
      if (val.find(']') != std::string::npos)
        flush_list(key, acc);
      else
        in_list = true;
    } else {
      size_t a = val.find_first_not_of(" \t\r");
      if (a == std::string::npos) {  // "key:" followed by a list on the next lines
        acc = "";
        in_list = true;
        continue;
      }
      val = val.substr(a);
      val.erase(val.find_last_not_of(" \t\r") + 1);
      raw[key] = val;
      if (val == "True" || val == "true")
        kv[key] = {1};
      else if (val == "False" || val == "false")
        kv[key] = {0};
      else
        kv[key] = {atof(val.c_str())};
    }
  }
  auto need = [&](const char* k, size_t n, double* dst) -> bool {
    auto it = kv.find(k);
    if (it == kv.end() || it->second.size() < n) {
      snprintf(err, errlen, "yaml key missing or short: %s", k);
      return false;
    }
    for (size_t i = 0; i < n; i++) dst[i] = it->second[i];
    return true;
  };
  memset(&c, 0, sizeof(c));
  double v;
  if (!need("type_of_vi", 1, &v)) return false;
  c.type_of_vi = (int)v;
  if (!need("image_width", 1, &v)) return false;
  c.image_width = (int)v;
  if (!need("image_height", 1, &v)) return false;
  c.image_height = (int)v;
  const bool depth_mode = c.type_of_vi == 0 || c.type_of_vi == 2;
  if (!need("cam0_intrinsics", 4, c.cam0_intrinsics) || !need("cam0_distortion_coeffs", 4, c.cam0_distortion)) return false;
  if (depth_mode) {  // vo_tracking.cpp:149-154 reads only these
    if (!need("depth_factor", 1, &c.depth_factor) || !need("T_imu_cam0", 16, c.T_imu_cam0)) return false;
    double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(c.T_cam0_cam1, eye, sizeof(eye));
  } else if (!need("cam1_intrinsics", 4, c.cam1_intrinsics) || !need("cam1_distortion_coeffs", 4, c.cam1_distortion)) {
    return false;
  }
  if (depth_mode) {
  } else if (c.type_of_vi == 4) {  // KITTI: vo_tracking.cpp:267-270
    double a[16], b[16];
    if (!need("cam0_projection_matrix", 16, a) || !need("cam1_projection_matrix", 16, b)) return false;
    memcpy(c.P0, a, sizeof(double) * 12);
    memcpy(c.P1, b, sizeof(double) * 12);
  } else if (c.type_of_vi == 1) {  // EuRoC: vo_tracking.cpp:218-236
    double a[16], b[16], m[16];
    if (!need("T_mavimu_cam0", 16, a) || !need("T_mavimu_cam1", 16, b) || !need("T_imu_mavimu", 16, m)) return false;
    SE3 T_mavi_c0 = se3_from_mat44(a), T_mavi_c1 = se3_from_mat44(b), T_i_mavi = se3_from_mat44(m);
    SE3 T_c0_c1 = se3_mul(se3_inverse(T_mavi_c0), T_mavi_c1);
    SE3 T_i_c0 = se3_mul(T_i_mavi, T_mavi_c0);
    auto put = [](const SE3& T, double* o) {
      Mat3 R = quat_to_mat(T.q);
      double mm[16] = {R.m[0][0], R.m[0][1], R.m[0][2], T.t.x, R.m[1][0], R.m[1][1], R.m[1][2], T.t.y,
                       R.m[2][0], R.m[2][1], R.m[2][2], T.t.z, 0, 0, 0, 1};
      memcpy(o, mm, sizeof(mm));
    };
    put(T_c0_c1, c.T_cam0_cam1);
    put(T_i_c0, c.T_imu_cam0);
  } else {
    if (!need("T_imu_cam0", 16, c.T_imu_cam0) || !need("T_cam0_cam1", 16, c.T_cam0_cam1)) return false;
  }
  const char* vk[6] = {"vifusion_para1", "vifusion_para2", "vifusion_para3", "vifusion_para4", "vifusion_para5", "vifusion_para6"};
  const char* fk[6] = {"feature_para1", "feature_para2", "feature_para3", "feature_para4", "feature_para5", "feature_para6"};
  const char* dk[3] = {"dr_para1", "dr_para2", "dr_para3"};
  for (int i = 0; i < 6; i++)
    if (!need(vk[i], 1, &c.vifusion_para[i]) || !need(fk[i], 1, &c.feature_para[i])) return false;
  for (int i = 0; i < 3; i++)
    if (!need(dk[i], 1, &c.dr_para[i])) return false;
  if (!need("window_size", 1, &v)) return false;
  c.window_size = (int)v;
  if (c.window_size < 3 || c.window_size > 100) c.window_size = 10;  // vo_localmap.cpp:443-447
  if (!config_finalize(c)) {
    snprintf(err, errlen, "unsupported type_of_vi %d", c.type_of_vi);
    return false;
  }
  return true;
}

// ------------------------------------------------------------------------------------------ VIMOTION
static inline Quat scalar_multi_q(float a, Quat b) { return {a * b.w, a * b.x, a * b.y, a * b.z}; }  // float a! kinetic_math.h:123
static inline Quat q1_multi_q2(Quat q1, Quat q2) {                                                  // kinetic_math.h:113-121
  Quat q;
  q.w = q2.w * q1.w - q2.x * q1.x - q2.y * q1.y - q2.z * q1.z;
  q.x = q2.x * q1.w + q2.w * q1.x + q2.z * q1.y - q2.y * q1.z;
  q.y = q2.y * q1.w - q2.z * q1.x + q2.w * q1.y + q2.x * q1.z;
  q.z = q2.z * q1.w + q2.y * q1.x - q2.x * q1.y + q2.w * q1.z;
  return q;
}
static inline Quat q_plus_q(Quat a, Quat b) { return {a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
static const size_t STATES_QUEUE_SIZE = 400;

VIMOTION::VIMOTION(const SE3& T_i_c_, double g, double p1, double p2, double p3, double p4, double p5, double p6) {
  T_i_c = T_i_c_;
  T_c_i = se3_inverse(T_i_c);
  acc_bias = gyro_bias = {0, 0, 0};
  init_state.pos = init_state.vel = {0, 0, 0};
  init_state.q_w_i = {1, 0, 0, 0};
  init_state.imu_data = {{0, 0, 0}, {0, 0, 0}, 0};
  imu_initialized = false;
  is_first_data = true;
  magnitude_g = g;
  gravity = {0, 0, -g};
  para_1 = p1;
  para_2 = p2;
  para_3 = p3;
  para_4 = p4;
  ba_sat = p5;
  bw_sat = p6;
}

static void madgwick_feedback(Quat q_prev, Vec3 acc, double acc_norm, double gain, Quat& qdot) {
  double ax = acc.x / acc_norm, ay = acc.y / acc_norm, az = acc.z / acc_norm;
  double qw = q_prev.w, qx = q_prev.x, qy = q_prev.y, qz = q_prev.z;
  double s[4];
  s[0] = 2 * qx * (ay + 2 * qw * qx + 2 * qy * qz) - 2 * qy * (ax - 2 * qw * qy + 2 * qx * qz);
  s[1] = 2 * qw * (ay + 2 * qw * qx + 2 * qy * qz) + 2 * qz * (ax - 2 * qw * qy + 2 * qx * qz) -
         4 * qx * (-2 * qx * qx - 2 * qy * qy + az + 1);
  s[2] = 2 * qz * (ay + 2 * qw * qx + 2 * qy * qz) - 2 * qw * (ax - 2 * qw * qy + 2 * qx * qz) -
         4 * qy * (-2 * qx * qx - 2 * qy * qy + az + 1);
  s[3] = 2 * qx * (ax - 2 * qw * qy + 2 * qx * qz) + 2 * qy * (ay + 2 * qw * qx + 2 * qy * qz);
  double sn = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3]);
  for (int i = 0; i < 4; i++) s[i] *= sn;  // quirk A16: s *= s.norm()
  qdot.w -= gain * s[0];
  qdot.x -= gain * s[1];
  qdot.y -= gain * s[2];
  qdot.z -= gain * s[3];
}

void VIMOTION::viIMUinitialization(const IMUSTATE& imu, Quat& q_w_i, Vec3& pos, Vec3& vel) {
  q_w_i = {1, 0, 0, 0};
  pos = vel = {0, 0, 0};
  init_state.imu_data = imu;
  init_state.pos = pos;
  init_state.vel = vel;
  Vec3 acc = imu.acc_raw - acc_bias;
  Vec3 gyro = imu.gyro_raw - gyro_bias;
  if (is_first_data) {
    if ((norm(acc) - magnitude_g) < 0.3) {
      Vec3 rpy{detm::det_atan2(-acc.y, -acc.z), detm::det_atan2(acc.x, -acc.z), 0};
      init_state.q_w_i = rpy2Q(rpy);
      states.push_back(init_state);
      if (states.size() >= STATES_QUEUE_SIZE) states.pop_front();
      is_first_data = false;
      q_w_i = rpy2Q(rpy);
    }
  } else {
    double dt = imu.timestamp - states.back().imu_data.timestamp;
    Quat q_prev = states.back().q_w_i;
    Quat omega{0, gyro.x, gyro.y, gyro.z};
    Quat qdot = scalar_multi_q(0.5f, q1_multi_q2(q_prev, omega));
    double acc_norm = norm(acc);
    if ((acc_norm - magnitude_g) < 0.3) madgwick_feedback(q_prev, acc, acc_norm, 10 * para_1, qdot);
    Quat q_new = quat_normalized(q_plus_q(q_prev, scalar_multi_q((float)dt, qdot)));
    init_state.q_w_i = q_new;
    states.push_back(init_state);
    if (states.size() >= STATES_QUEUE_SIZE) states.pop_front();
    if (states.size() > 30) imu_initialized = true;
  }
}

void VIMOTION::viVisiontrigger(Quat& init_orientation) {
  MOTION_STATE state = states.back();
  state.pos = state.vel = {0, 0, 0};
  Vec3 rpy = Q2rpy(state.q_w_i);
  rpy.z = 0;
  Quat q = quat_normalized(rpy2Q(rpy));
  state.q_w_i = q;
  states.clear();
  states.push_back(state);
  init_orientation = state.q_w_i;
}

void VIMOTION::viIMUPropagation(const IMUSTATE& imu, Quat& q_w_i, Vec3& pos, Vec3& vel) {
  Vec3 acc = imu.acc_raw - acc_bias, gyro = imu.gyro_raw - gyro_bias;
  MOTION_STATE s_prev = states.back(), s_new;
  double dt = imu.timestamp - s_prev.imu_data.timestamp;
  Quat q_prev = s_prev.q_w_i;
  Mat3 R_prev = quat_to_mat(q_prev);
  Quat omega{0, gyro.x, gyro.y, gyro.z};
  Quat qdot = scalar_multi_q(0.5f, q1_multi_q2(q_prev, omega));
  double acc_norm = norm(acc);
  if ((acc_norm - magnitude_g) < 0.3) madgwick_feedback(q_prev, acc, acc_norm, para_1, qdot);
  s_new.q_w_i = quat_normalized(q_plus_q(q_prev, scalar_multi_q((float)dt, qdot)));
  s_new.pos = s_prev.pos + s_prev.vel * dt;
  s_new.vel = s_prev.vel + ((R_prev * acc) - gravity) * dt;
  s_new.imu_data = imu;
  {
    const Vec3 a = quat_to_mat(kf_dq) * acc;  // (the attitude increment BEFORE this sample's rotation)
    kf_dp = (kf_dp + kf_dv * dt) + a * ((0.5 * dt) * dt);
    kf_dv = kf_dv + a * dt;
  }
  kf_dq = quat_normalized(quat_mul(kf_dq, quat_exp(gyro * dt)));
  kf_dt += dt;
  states.push_back(s_new);
  if (states.size() >= STATES_QUEUE_SIZE) states.pop_front();
  q_w_i = s_new.q_w_i;
  pos = s_new.pos;
  vel = s_new.vel;
}

bool VIMOTION::viFindStateIdx(double time, int& idx_in_q) {
  int idx = 9999;
  for (int i = (int)states.size() - 1; i >= 0; i--) {
    if ((states[i].imu_data.timestamp - time) > 0) {
      idx = i;
    } else {
      idx = i;
      break;
    }
  }
  if (idx > 0 && idx != 9999) {
    idx_in_q = idx;
    return true;
  }
  return false;
}

bool VIMOTION::viGetIMURollPitchAtTime(double time, double& roll, double& pitch) {
  int idx;
  if (!viFindStateIdx(time, idx)) return false;
  Vec3 rpy = Q2rpy(quat_normalized(states[idx].q_w_i));  // SE3(q,pos).so3().unit_quaternion()
  roll = rpy.x;
  pitch = rpy.y;
  return true;
}

bool VIMOTION::viGetCorrFrameState(double time, SE3& T_c_w) {
  int idx;
  if (!viFindStateIdx(time, idx)) return false;
  SE3 T_w_i = se3_from_quat(states[idx].q_w_i, states[idx].pos);
  T_c_w = se3_inverse(se3_mul(T_w_i, T_i_c));
  return true;
}

void VIMOTION::viVisionRPCompensation(double time, SE3& T_c_w) {
  SE3 T_w_i_before = se3_mul(se3_inverse(T_c_w), T_c_i);
  Vec3 rpy_before = Q2rpy(T_w_i_before.q), rpy_vimotion{0, 0, 0};
  if (viGetIMURollPitchAtTime(time, rpy_vimotion.x, rpy_vimotion.y)) {
    rpy_vimotion.z = rpy_before.z;
    Vec3 ryp_after = rpy_before * (1 - para_2) + rpy_vimotion * para_2;
    SE3 T_w_i_after{rpy2Q(ryp_after), T_w_i_before.t};  // SE3(SO3(Quaterniond)) normalises
    T_w_i_after.q = quat_normalized(T_w_i_after.q);
    T_c_w = se3_inverse(se3_mul(T_w_i_after, T_i_c));
  }
}

void VIMOTION::viCorrectionFromVision(double t_curr, const SE3& Tcw_curr, double t_last, const SE3& Tcw_last, double) {
  Vec3 acc_bias_est{0, 0, 0}, gyro_bias_est{0, 0, 0};
  int idx_curr, idx_last, idx_mid;
  if (viFindStateIdx(t_last, idx_last) && viFindStateIdx(t_curr, idx_curr)) {
    if (idx_last == idx_curr) return;
    double dt = t_curr - t_last;
    idx_mid = idx_last + (int)std::floor((idx_curr - idx_last) / 2);
    SE3 T_w_iA = se3_mul(se3_inverse(Tcw_last), T_c_i);
    SE3 T_w_iB = se3_mul(se3_inverse(Tcw_curr), T_c_i);
    SE3 T_w_ia = se3_from_quat(states[idx_last].q_w_i, states[idx_last].pos);
    SE3 T_w_ib = se3_from_quat(states[idx_curr].q_w_i, states[idx_curr].pos);
    SE3 T_w_im = se3_from_quat(states[idx_mid].q_w_i, states[idx_mid].pos);
    SE3 T_iB_iA = se3_mul(se3_inverse(T_w_iB), T_w_iA);
    SE3 T_ib_ia = se3_mul(se3_inverse(T_w_ib), T_w_ia);
    Quat Q_B_A = T_iB_iA.q, Q_b_a = T_ib_ia.q;
    // Eigen Quaternion::inverse() = conjugate / squaredNorm
    double n2 = quat_sqnorm(Q_b_a);
    Quat Q_b_a_inv{Q_b_a.w / n2, -Q_b_a.x / n2, -Q_b_a.y / n2, -Q_b_a.z / n2};
    Quat Q_B_b = quat_mul(Q_B_A, Q_b_a_inv);
    gyro_bias_est = {Q_B_b.x / dt, Q_B_b.y / dt, Q_B_b.z / dt};
    int cnt = idx_curr - idx_last + 1;
    Vec3 vel_imu{0, 0, 0};
    for (int i = idx_last; i <= idx_curr; i++) vel_imu = vel_imu + states[i].vel;
    vel_imu = vel_imu * (1.0 / cnt);
    Vec3 vel_vision_world = (T_w_iB.t - T_w_iA.t) * (1.0 / dt);  // Eigen: vector / dt
    vel_vision_world = {(T_w_iB.t.x - T_w_iA.t.x) / dt, (T_w_iB.t.y - T_w_iA.t.y) / dt, (T_w_iB.t.z - T_w_iA.t.z) / dt};
    Vec3 diff_vel_world = vel_vision_world - vel_imu;
    Quat qm = T_w_im.q;
    double nm2 = quat_sqnorm(qm);
    Quat qm_inv{qm.w / nm2, -qm.x / nm2, -qm.y / nm2, -qm.z / nm2};
    Vec3 diff_vel_local = quat_to_mat(qm_inv) * diff_vel_world;
    acc_bias_est = {-diff_vel_local.x / dt, -diff_vel_local.y / dt, -diff_vel_local.z / dt};
    SE3 T_diff = se3_mul(T_w_iB, se3_inverse(T_w_ib));
    for (size_t i = idx_curr; i < states.size(); i++) {
      SE3 newT = se3_mul(T_diff, se3_from_quat(states[i].q_w_i, states[i].pos));
      states[i].q_w_i = newT.q;
      states[i].pos = newT.t;
      states[i].vel = states[i].vel + diff_vel_world;
    }
    if (std::isnan(acc_bias_est.x)) acc_bias_est = {0, 0, 0};
    if (std::isnan(gyro_bias_est.x)) gyro_bias_est = {0, 0, 0};
    double ba_est_norm = norm(acc_bias_est);
    if (ba_est_norm > ba_sat) acc_bias_est = acc_bias_est * (ba_sat / ba_est_norm);
    double bw_est_norm = norm(gyro_bias_est);
    if (ba_est_norm > bw_sat) gyro_bias_est = gyro_bias_est * (bw_sat / bw_est_norm);  // quirk A17
    if (dt < 0.1) {
      acc_bias = (1 - para_3) * acc_bias + (para_3)*acc_bias_est;
      gyro_bias = (1 - para_3) * gyro_bias + (para_4)*gyro_bias_est;
    }
  }
}

// ------------------------------------------------------------------------------------------ F2FTracking
F2FTracking::F2FTracking(const Config& cfg_in, uint64_t seed) : cfg(cfg_in) {
  d_camera.cam_type = cfg.cam_type;
  d_camera.img_w = cfg.image_width;
  d_camera.img_h = cfg.image_height;
  memcpy(d_camera.K0, cfg.cam0_intrinsics, sizeof(double) * 4);
  memcpy(d_camera.D0, cfg.cam0_distortion, sizeof(double) * 4);
  memcpy(d_camera.K1, cfg.cam1_intrinsics, sizeof(double) * 4);
  memcpy(d_camera.D1, cfg.cam1_distortion, sizeof(double) * 4);
  for (int i = 0; i < 9; i++) {
    d_camera.R0.m[i / 3][i % 3] = cfg.R0[i];
    d_camera.R1.m[i / 3][i % 3] = cfg.R1[i];
  }
  memcpy(d_camera.P0_, cfg.P0, sizeof(double) * 12);
  memcpy(d_camera.P1_, cfg.P1, sizeof(double) * 12);
  d_camera.cam0_fx = cfg.P0[0];
  d_camera.cam0_fy = cfg.P0[5];
  d_camera.cam0_cx = cfg.P0[2];
  d_camera.cam0_cy = cfg.P0[6];
  d_camera.T_cam0_cam1 = se3_from_mat44(cfg.T_cam0_cam1);
  d_camera.T_cam1_cam0 = se3_inverse(d_camera.T_cam0_cam1);
  d_camera.cam_scale_factor = cfg.depth_factor;
  feature_dem = new FeatureDEM(cfg.image_width, cfg.image_height, cfg.feature_para);
  // f2f_tracking.cpp:18-19: only vifusion_para1..4 are forwarded (quirk A19)
  vimotion = new VIMOTION(se3_from_mat44(cfg.T_imu_cam0), 9.81, cfg.vifusion_para[0], cfg.vifusion_para[1],
                          cfg.vifusion_para[2], cfg.vifusion_para[3]);
  curr_frame = &frames[0];
  last_frame = &frames[1];
  iir_ratio = (float)cfg.dr_para[0];
  range = (float)cfg.dr_para[1];
  enable_dummy = !(cfg.dr_para[2] < 0.5);
  frameCount = 0;
  has_localmap_feedback = false;
  vo_tracking_state = UnInit;
  has_imu = false;
  skip_n_imgs = cfg.skip_first_n_imgs;
  need_equal_hist = cfg.need_equal_hist != 0;
  T_c_w_last_keyframe = se3_identity();
  continus_tracking_fail_cnt = 0;
  trackingfail_cnt = 0;
  lm_id_counter = 100;
  ransac_seed = seed;
  dbg_of_inlier = dbg_F_inlier = dbg_pnp_inlier = 0;
}
F2FTracking::~F2FTracking() {
  delete feature_dem;
  delete vimotion;
}

void F2FTracking::imu_feed(double time, Vec3 acc, Vec3 gyro, Quat& q, Vec3& p, Vec3& v) {
  IMUSTATE s{acc, gyro, time};
  if (!vimotion->imu_initialized) {
    has_imu = true;
    vimotion->viIMUinitialization(s, q, p, v);
  } else {
    vimotion->viIMUPropagation(s, q, p, v);
  }
}

LandMarkInFrame F2FTracking::make_landmark(Vec2 pt2d, Vec2 pt2d_undist, const SE3& T_c_w, bool is_inlier) {
  LandMarkInFrame lm;
  lm.lm_id = lm_id_counter++;
  lm.lm_1st_obs_2d = lm.lm_2d_undistort = pt2d_undist;
  lm.lm_2d_plane = pt2d;
  lm.lm_1st_obs_frame_pose = T_c_w;
  lm.is_tracking_inlier = is_inlier;
  lm.is_belong_to_kf = false;
  lm.lm_3d_w = {0, 0, 0};
  lm.lm_3d_c = {0, 0, 0};
  lm.has_3d = false;
  return lm;
}

static inline Vec3 world2cameraT_c_w(Vec3 p, const SE3& T) { return se3_act(T, p); }
static inline Vec3 camera2worldT_c_w(Vec3 p, const SE3& T) { return se3_act(se3_inverse(T), p); }

// CameraFrame::recover3DPts_c_FromStereo (camera_frame.cpp:93-180) on the arrays getAll2dPlaneUndistort3d_cvPf hands it: the matcher's
// seeds, calcOpticalFlowPyrLK img0 -> img1, undistortPoints(K1, D1, R1, P1), trignaulationPtFromStereo and the rand()-drawn dummy depth
// of every failure, in landmark order.  meas = pt3ds, mask = maskHas3DInf.
void F2FTracking::recover3DPts_c_FromStereo(const uint8_t* img0, const uint8_t* img1, int n_, const float* p0, const float* p0u, const float* p3,
                                            const uint8_t* has_3d, const SE3& T_c_w, float rng, Vec3* meas, uint8_t* meas_mask) {
  const size_t n = (size_t)n_;
  std::vector<float> p1(p0, p0 + 2 * n), proj(2 * n), p1u(2 * n);
  std::vector<uint8_t> status(n);
  if (n) {
    project_points(p3, (int)n, se3_mul(d_camera.T_cam1_cam0, T_c_w), d_camera.K1, d_camera.D1, proj.data());
    for (size_t i = 0; i < n; i++)
      if (has_3d[i]) {
        p1[2 * i] = proj[2 * i];
        p1[2 * i + 1] = proj[2 * i + 1];
      }
    calc_optical_flow_pyr_lk(img0, img1, d_camera.img_w, d_camera.img_h, p0, p1.data(), status.data(), (int)n, 31, 5, 30, 0.001, 1, 1e-4f);
    undistort_points(p1.data(), (int)n, d_camera.K1, d_camera.D1, d_camera.R1, d_camera.P1_, p1u.data());
  }
  for (size_t i = 0; i < n; i++) {
    bool ok = false;
    if (status[i] == 1) {
      Vec3 pc = triangulate_dlt({(double)p0u[2 * i], (double)p0u[2 * i + 1]}, {(double)p1u[2 * i], (double)p1u[2 * i + 1]},
                                d_camera.P0_, d_camera.P1_);
      if (!(pc.z < 0 || pc.z > rng)) {  // quirk A13
        meas[i] = pc;
        ok = true;
      }
    }
    if (!ok) {  // quirk A11: rand()-drawn dummy depth
      float d_rand = (float)(0.3 + (float)rnd.next() / ((float)(2147483647 / (0.4))));
      double depth = d_rand;
      meas[i] = {((double)p0u[2 * i] - d_camera.cam0_cx) * depth / d_camera.cam0_fx,
                 ((double)p0u[2 * i + 1] - d_camera.cam0_cy) * depth / d_camera.cam0_fy, depth};
    }
    meas_mask[i] = ok;
  }
}

void F2FTracking::depthInnovation(CameraFrame& f) {  // camera_frame.cpp:93-180,236-330
  const size_t n = f.landmarks.size();
  std::vector<Vec3> tri(n), meas(n);
  std::vector<uint8_t> tri_mask(n), meas_mask(n);
  // recover3DPts_c_FromTriangulation
  for (size_t i = 0; i < n; i++) {
    LandMarkInFrame& lm = f.landmarks[i];
    Vec3 baseline = lm.lm_1st_obs_frame_pose.t - f.T_c_w.t;  // quirk A12
    tri[i] = {0, 0, 0};
    tri_mask[i] = 0;
    if (norm(baseline) >= 0.2) {
      Vec3 pw = triangulate_two_view(lm.lm_1st_obs_2d, lm.lm_2d_undistort, lm.lm_1st_obs_frame_pose, f.T_c_w,
                                     d_camera.cam0_fx, d_camera.cam0_fy, d_camera.cam0_cx, d_camera.cam0_cy);
      Vec3 pc = world2cameraT_c_w(pw, f.T_c_w);
      if (pc.z >= 0.5 && pc.z <= range) {
        tri[i] = pc;
        tri_mask[i] = 1;
      }
    }
  }
  if (d_camera.cam_type == DEPTH_D435) {
    // recover3DPts_c_FromDepthImg (camera_frame.cpp:182-234): nearest depth pixel (round half away from zero), metres =
    // Z16 / cam_scale_factor, valid in [0.3, range]; otherwise a rand()-drawn dummy depth through the PLANE position
    for (size_t i = 0; i < n; i++) {
      const LandMarkInFrame& lm = f.landmarks[i];
      const float ptx = (float)std::round(lm.lm_2d_plane.x), pty = (float)std::round(lm.lm_2d_plane.y);
      const int ix = (int)std::lrint(ptx), iy = (int)std::lrint(pty);  // Mat::at<ushort>(Point2f) -> Point(cvRound)
      bool ok = false;
      const float z = (float)(f.d_img[(size_t)iy * d_camera.img_w + ix] / d_camera.cam_scale_factor);
      if (z >= 0.3 && z <= range) {
        meas[i] = {((double)ptx - d_camera.cam0_cx) * (double)z / d_camera.cam0_fx,
                   ((double)pty - d_camera.cam0_cy) * (double)z / d_camera.cam0_fy, (double)z};
        ok = true;
      } else {
        float d_rand = (float)(0.3 + (float)rnd.next() / ((float)(2147483647 / (0.4))));
        double depth = d_rand;
        meas[i] = {(lm.lm_2d_plane.x - d_camera.cam0_cx) * depth / d_camera.cam0_fx,
                   (lm.lm_2d_plane.y - d_camera.cam0_cy) * depth / d_camera.cam0_fy, depth};
      }
      meas_mask[i] = ok;
    }
  } else {
    // recover3DPts_c_FromStereo
    std::vector<float> p0(2 * n), p0u(2 * n), p3(3 * n);
    std::vector<uint8_t> has(n);
    for (size_t i = 0; i < n; i++) {  // getAll2dPlaneUndistort3d_cvPf: cv::Point2f / Point3f copies of the landmarks
      const LandMarkInFrame& lm = f.landmarks[i];
      p0[2 * i] = (float)lm.lm_2d_plane.x;
      p0[2 * i + 1] = (float)lm.lm_2d_plane.y;
      p0u[2 * i] = (float)lm.lm_2d_undistort.x;
      p0u[2 * i + 1] = (float)lm.lm_2d_undistort.y;
      p3[3 * i] = (float)lm.lm_3d_w.x;
      p3[3 * i + 1] = (float)lm.lm_3d_w.y;
      p3[3 * i + 2] = (float)lm.lm_3d_w.z;
      has[i] = lm.has_3d ? 1 : 0;
    }
    recover3DPts_c_FromStereo(f.img0.data(), f.img1.data(), (int)n, p0.data(), p0u.data(), p3.data(), has.data(), f.T_c_w, range, meas.data(),
                              meas_mask.data());
  }
  for (size_t i = 0; i < n; i++) {
    LandMarkInFrame& lm = f.landmarks[i];
    Vec3 lm_c_measure;
    if (!meas_mask[i] && !tri_mask[i]) {
      if (d_camera.cam_type != DEPTH_D435 && !lm.has_3d && enable_dummy) {  // camera_frame.cpp:288-301: stereo types only
        lm_c_measure = meas[i];
        lm.lm_3d_c = lm_c_measure;
        lm.lm_3d_w = camera2worldT_c_w(lm_c_measure, f.T_c_w);
        lm.has_3d = true;
      }
      continue;
    }
    lm_c_measure = meas_mask[i] ? meas[i] : tri[i];
    if (lm.has_3d) {
      Vec3 lm_c = world2cameraT_c_w(lm.lm_3d_w, f.T_c_w);
      Vec3 upd = lm_c * (double)iir_ratio + lm_c_measure * (double)(1 - iir_ratio);  // float ratio, float (1-ratio)
      lm.lm_3d_c = upd;
      lm.lm_3d_w = camera2worldT_c_w(upd, f.T_c_w);
    } else {
      lm.lm_3d_c = lm_c_measure;
      lm.lm_3d_w = camera2worldT_c_w(lm_c_measure, f.T_c_w);
      lm.has_3d = true;
    }
  }
}

static void eraseNoDepthPoint(CameraFrame& f) {
  std::vector<LandMarkInFrame> keep;
  for (auto& lm : f.landmarks)
    if (lm.has_3d) keep.push_back(lm);
  f.landmarks.swap(keep);
}

bool F2FTracking::init_frame() {
  std::vector<Pt2f> pts2d;
  feature_dem->detect(curr_frame->img0.data(), pts2d);
  std::vector<float> src(2 * pts2d.size()), und(2 * pts2d.size());
  for (size_t i = 0; i < pts2d.size(); i++) {
    src[2 * i] = pts2d[i].x;
    src[2 * i + 1] = pts2d[i].y;
  }
  und = src;  // DEPTH_D435: the undistorted position is the pixel position (f2f_tracking.cpp:410-421)
  if (!pts2d.empty() && d_camera.cam_type != DEPTH_D435)  // undistortPoints in both stereo modes (f2f_tracking.cpp:422-437)
    undistort_points(src.data(), (int)pts2d.size(), d_camera.K0, d_camera.D0, d_camera.R0, d_camera.P0_, und.data());
  for (size_t i = 0; i < pts2d.size(); i++)
    curr_frame->landmarks.push_back(make_landmark({(double)pts2d[i].x, (double)pts2d[i].y},
                                                  {(double)und[2 * i], (double)und[2 * i + 1]}, curr_frame->T_c_w, true));
  depthInnovation(*curr_frame);
  eraseNoDepthPoint(*curr_frame);
  int valid = 0;
  for (auto& lm : curr_frame->landmarks) valid += (lm.has_3d && lm.is_tracking_inlier);
  if (valid > 30) {
    pose_records.push_back({(int)curr_frame->frame_id, curr_frame->T_c_w});  // f2f_tracking.cpp:443-446
    T_c_w_last_keyframe = curr_frame->T_c_w;
    return true;
  }
  return false;
}

static inline Vec3 world2cameraT_c_w_(Vec3 p, const SE3& T) { return se3_act(T, p); }

bool F2FTracking::lk_tracking(CameraFrame& from, CameraFrame& to, const SE3& guess, bool use_guess) {
  const int n = (int)from.landmarks.size();
  std::vector<float> from_plane(2 * n), tracked(2 * n), from_und(2 * n), tracked_und(2 * n), from_p3d(3 * n);
  for (int i = 0; i < n; i++) {
    const LandMarkInFrame& lm = from.landmarks[i];
    from_plane[2 * i] = (float)lm.lm_2d_plane.x;
    from_plane[2 * i + 1] = (float)lm.lm_2d_plane.y;
    from_und[2 * i] = (float)lm.lm_2d_undistort.x;
    from_und[2 * i + 1] = (float)lm.lm_2d_undistort.y;
    from_p3d[3 * i] = (float)lm.lm_3d_w.x;
    from_p3d[3 * i + 1] = (float)lm.lm_3d_w.y;
    from_p3d[3 * i + 2] = (float)lm.lm_3d_w.z;
  }
  tracked = from_plane;
  std::vector<uint8_t> mask_tracked(n);
  if (n) {
    if (use_guess) {
      if (d_camera.cam_type == DEPTH_D435) {  // lkorb_tracking.cpp:41-52: pinhole projection of the float-narrowed landmark
        for (int i = 0; i < n; i++) {
          Vec3 pc = world2cameraT_c_w_({(double)from_p3d[3 * i], (double)from_p3d[3 * i + 1], (double)from_p3d[3 * i + 2]}, guess);
          tracked[2 * i] = (float)(d_camera.cam0_fx * pc.x / pc.z + d_camera.cam0_cx);
          tracked[2 * i + 1] = (float)(d_camera.cam0_fy * pc.y / pc.z + d_camera.cam0_cy);
        }
      } else {
        project_points(from_p3d.data(), n, guess, d_camera.K0, d_camera.D0, tracked.data());
      }
    }
    calc_optical_flow_pyr_lk(from.img0.data(), to.img0.data(), d_camera.img_w, d_camera.img_h, from_plane.data(),
                             tracked.data(), mask_tracked.data(), n, 31, 10, 30, 0.001, 1, 1e-4f);
  }
  if (d_camera.cam_type == STEREO_RECT || d_camera.cam_type == DEPTH_D435) {  // lkorb_tracking.cpp:76-85
    from_und = from_plane;
    tracked_und = tracked;
  } else if (n) {
    undistort_points(tracked.data(), n, d_camera.K0, d_camera.D0, d_camera.R0, d_camera.P0_, tracked_und.data());
  }
  to.landmarks.clear();
  const int w = d_camera.img_w - 1, h = d_camera.img_h - 1;
  int of_inlier_cnt = 0;
  std::vector<int> surv;  // ascending indices of survivors (what remains in the erased-in-place vectors)
  for (int i = n - 1; i >= 0; i--) {
    if (mask_tracked[i] == 1 && tracked[2 * i] > 0 && tracked[2 * i + 1] > 0 && tracked[2 * i] < w && tracked[2 * i + 1] < h) {
      of_inlier_cnt++;
      LandMarkInFrame lm = from.landmarks[i];
      lm.lm_2d_plane = {(double)tracked[2 * i], (double)tracked[2 * i + 1]};
      lm.lm_2d_undistort = {(double)tracked_und[2 * i], (double)tracked_und[2 * i + 1]};
      to.landmarks.push_back(lm);  // descending order (quirk A1)
      surv.push_back(i);
    }
  }
  std::reverse(surv.begin(), surv.end());
  dbg_of_inlier = of_inlier_cnt;
  dbg_F_inlier = dbg_pnp_inlier = 0;
  if (of_inlier_cnt < 10) return false;
  const int m = (int)surv.size();
  std::vector<float> m1(2 * m), m2(2 * m);
  for (int k = 0; k < m; k++) {
    m1[2 * k] = from_und[2 * surv[k]];
    m1[2 * k + 1] = from_und[2 * surv[k] + 1];
    m2[2 * k] = tracked_und[2 * surv[k]];
    m2[2 * k + 1] = tracked_und[2 * surv[k] + 1];
  }
  std::vector<uint8_t> maskF(m);
  find_fundamental_ransac(m1.data(), m2.data(), m, 5.0, 0.99, 0, maskF.data());
  for (int i = 0; i < m; i++)
    if (maskF[i] == 0) to.landmarks[i].is_tracking_inlier = false;  // mirrored index (quirk A1)
  int F_inlier_cnt = 0;
  for (auto& lm : to.landmarks) F_inlier_cnt += lm.is_tracking_inlier;
  dbg_F_inlier = F_inlier_cnt;
  if (F_inlier_cnt < 10) return false;
  std::vector<float> p2d, p3d;
  for (auto& lm : to.landmarks)
    if (lm.has_3d && lm.is_tracking_inlier) {
      p2d.push_back((float)lm.lm_2d_undistort.x);
      p2d.push_back((float)lm.lm_2d_undistort.y);
      p3d.push_back((float)lm.lm_3d_w.x);
      p3d.push_back((float)lm.lm_3d_w.y);
      p3d.push_back((float)lm.lm_3d_w.z);
    }
  const int np = (int)p2d.size() / 2;
  std::vector<uint8_t> mask_pnp(std::max(np, 1), 0);
  // r_, t_ start as the guess (ITERATIVE) or zero (P3P); untouched if RANSAC finds no model
  SE3 T = use_guess ? se3_from_mat(quat_to_mat(guess.q), guess.t) : se3_identity();
  int pnp_inliers = solve_pnp_ransac(p3d.data(), p2d.data(), np, d_camera.cam0_fx, d_camera.cam0_fy, d_camera.cam0_cx,
                                     d_camera.cam0_cy, use_guess, 100, 3.0, 0.99,
                                     0, T, mask_pnp.data());
  int indexLM = 0;  // CameraFrame::updateLMState
  for (auto& lm : to.landmarks)
    if (lm.has_3d && lm.is_tracking_inlier) {
      if (mask_pnp[indexLM] == 0) lm.is_tracking_inlier = false;
      indexLM++;
    }
  to.T_c_w = T;
  dbg_T_pnp = T;
  dbg_pnp_inlier = pnp_inliers;
  return pnp_inliers >= 10;
}

void F2FTracking::image_feed(double time, const uint8_t* img0_in, const uint8_t* img1_in, bool& new_keyframe,
                             bool& reset_cmd) {
  new_keyframe = false;
  reset_cmd = false;
  frameCount++;
  std::swap(last_frame, curr_frame);
  curr_frame->clear();
  curr_frame->frame_id = frameCount;
  curr_frame->frame_time = time;
  const size_t npx = (size_t)cfg.image_width * cfg.image_height;
  curr_frame->img0.assign(img0_in, img0_in + npx);
  if (d_camera.cam_type == DEPTH_D435) {  // f2f_tracking.cpp:116-119: the second image is the Z16 depth image
    const uint16_t* d = reinterpret_cast<const uint16_t*>(img1_in);
    curr_frame->d_img.assign(d, d + npx);
  } else {
    curr_frame->img1.assign(img1_in, img1_in + npx);
  }
  if (skip_n_imgs > 0) {
    skip_n_imgs--;
    return;
  }
  if (need_equal_hist) {
    equalize_hist(curr_frame->img0.data(), curr_frame->img0.data(), cfg.image_width, cfg.image_height);
    if (d_camera.cam_type != DEPTH_D435)
      equalize_hist(curr_frame->img1.data(), curr_frame->img1.data(), cfg.image_width, cfg.image_height);
  }
  const int state_in = vo_tracking_state;
  switch (vo_tracking_state) {
    case UnInit: {
      Mat3 R_w_c = {{{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}}};
      curr_frame->T_c_w = se3_inverse(se3_from_mat(R_w_c, {0, 0, 0}));
      if (has_imu) {
        if (vimotion->imu_initialized) {
          Quat q_init;
          vimotion->viVisiontrigger(q_init);
          Mat3 R = quat_to_mat(q_init) * quat_to_mat(vimotion->T_i_c.q);
          curr_frame->T_c_w = se3_inverse(se3_from_mat(R, {0, 0, 0}));
        } else {
          break;
        }
      }
      if (init_frame()) {
        new_keyframe = true;
        vo_tracking_state = Tracking;
      }
      break;
    }
    case Tracking: {
      // STEP1: recover from the local-map feedback (f2f_tracking.cpp:189-219)
      if (has_localmap_feedback) {
        int corr_id = (int)correction_inf.frame_id;
        int old_pose_idx = 0;
        for (int i = (int)pose_records.size() - 1; i >= 0; i--)
          if (pose_records[i].frame_id == corr_id) {
            old_pose_idx = i;
            break;
          }
        SE3 old_T_c_w = pose_records.at(old_pose_idx).T_c_w;
        SE3 old_T_c_w_inv = se3_inverse(old_T_c_w);
        SE3 update_T_c_w = correction_inf.T_c_w;
        for (size_t i = old_pose_idx; i < pose_records.size(); i++) {
          SE3 T_diff = se3_mul(pose_records[i].T_c_w, old_T_c_w_inv);
          pose_records[i].T_c_w = se3_mul(T_diff, update_T_c_w);
        }
        SE3 T_diff = se3_mul(last_frame->T_c_w, old_T_c_w_inv);
        last_frame->T_c_w = se3_mul(T_diff, update_T_c_w);
        // correctLMP3DWByLMP3DCandT (camera_frame.cpp:332-342) iterates `for(auto lm:landmarks)` BY VALUE: a no-op
        // forceCorrectLM3DW (camera_frame.cpp:344-360): ids narrowed to int, first landmark with that id
        for (int i = 0; i < correction_inf.lm_count; i++) {
          int id = (int)correction_inf.lm_id.at(i);
          for (auto& lm : last_frame->landmarks)
            if (lm.lm_id == id) {
              lm.lm_3d_w = correction_inf.lm_3d.at(i);
              break;
            }
        }
        // forceMarkOutlier (camera_frame.cpp:362-378)
        for (int i = 0; i < correction_inf.lm_outlier_count; i++) {
          int id = (int)correction_inf.lm_outlier_id.at(i);
          for (auto& lm : last_frame->landmarks)
            if (lm.lm_id == id) lm.is_tracking_inlier = false;
        }
        has_localmap_feedback = false;
      }
      SE3 imu_guess = se3_identity();
      bool has_imu_guess = false;
      if (has_imu) has_imu_guess = vimotion->viGetCorrFrameState(time, imu_guess);
      bool tracking_success = lk_tracking(*last_frame, *curr_frame, imu_guess, has_imu_guess);
      if (!tracking_success) {
        continus_tracking_fail_cnt++;
        std::swap(last_frame, curr_frame);
        if (continus_tracking_fail_cnt >= 2) {
          vo_tracking_state = TrackingFail;
          continus_tracking_fail_cnt = 0;
        }
        break;
      }
      continus_tracking_fail_cnt = 0;
      if (has_imu) vimotion->viVisionRPCompensation(curr_frame->frame_time, curr_frame->T_c_w);
      // OptimizeInFrame::optimize
      {
        std::vector<Vec3> p3;
        std::vector<Vec2> p2;
        std::vector<int64_t> ids;
        for (auto& lm : curr_frame->landmarks)
          if (lm.has_3d && lm.is_tracking_inlier) {
            p3.push_back(lm.lm_3d_w);
            p2.push_back(lm.lm_2d_undistort);
            ids.push_back(lm.lm_id);
          }
        dbg_T_pre = curr_frame->T_c_w;
        bool ok = optimize_in_frame(curr_frame->T_c_w, p3.data(), p2.data(), ids.data(), (int)p3.size(), d_camera.cam0_fx,
                                    d_camera.cam0_fy, d_camera.cam0_cx, d_camera.cam0_cy);
        dbg_T_lm = curr_frame->T_c_w;
        if (!ok) {
          continus_tracking_fail_cnt++;
          std::swap(last_frame, curr_frame);
          if (continus_tracking_fail_cnt >= 2) {
            vo_tracking_state = TrackingFail;
            continus_tracking_fail_cnt = 0;
          }
          break;
        }
      }
      // calReprjInlierOutlier(1.5) + eraseReprjOutlier   camera_frame.cpp:18-28,43-91
      double mean_reprojection_error = 0;
      {
        std::vector<double> distances, valid;
        for (auto& lm : curr_frame->landmarks) {
          Vec3 pc = world2cameraT_c_w(lm.lm_3d_w, curr_frame->T_c_w);
          double u = d_camera.cam0_fx * pc.x / pc.z + d_camera.cam0_cx, v = d_camera.cam0_fy * pc.y / pc.z + d_camera.cam0_cy;
          double ex = lm.lm_2d_undistort.x - u, ey = lm.lm_2d_undistort.y - v;
          double d = std::sqrt(ex * ex + ey * ey);
          distances.push_back(d);
          if (d < 3.0) valid.push_back(d);
        }
        double sum = 0;
        for (double d : valid) sum += d;
        mean_reprojection_error = sum / (double)valid.size();
        std::sort(valid.begin(), valid.end());
        double sh = 3.0;
        if (!valid.empty()) sh = 1.5 * valid[valid.size() / 2];  // reference: .at() would throw on empty
        if (sh >= 3.0) sh = 3.0;
        for (size_t i = 0; i < curr_frame->landmarks.size(); i++)
          curr_frame->landmarks[i].is_tracking_inlier = !(distances[i] > sh);
        std::vector<LandMarkInFrame> keep;
        for (auto& lm : curr_frame->landmarks)
          if (lm.is_tracking_inlier) keep.push_back(lm);
        curr_frame->landmarks.swap(keep);
      }
      curr_frame->reprojection_error = mean_reprojection_error;
      if (has_imu)
        vimotion->viCorrectionFromVision(curr_frame->frame_time, curr_frame->T_c_w, last_frame->frame_time,
                                         last_frame->T_c_w, mean_reprojection_error);
      // redetect
      {
        std::vector<Pt2f> existed, newPts;
        for (auto& lm : curr_frame->landmarks) existed.push_back({(float)lm.lm_2d_plane.x, (float)lm.lm_2d_plane.y});
        int orig_size = (int)curr_frame->landmarks.size();
        feature_dem->redetect(curr_frame->img0.data(), existed, newPts);
        std::vector<float> src(2 * newPts.size()), und(2 * newPts.size());
        for (size_t i = 0; i < newPts.size(); i++) {
          src[2 * i] = newPts[i].x;
          src[2 * i + 1] = newPts[i].y;
        }
        und = src;
        if (d_camera.cam_type == STEREO_UNRECT && !newPts.empty())
          undistort_points(src.data(), (int)newPts.size(), d_camera.K0, d_camera.D0, d_camera.R0, d_camera.P0_, und.data());
        bool add_as_inliers = orig_size < 60;
        for (size_t i = 0; i < newPts.size(); i++)
          curr_frame->landmarks.push_back(make_landmark({(double)newPts[i].x, (double)newPts[i].y},
                                                        {(double)und[2 * i], (double)und[2 * i + 1]}, curr_frame->T_c_w,
                                                        add_as_inliers));
      }
      depthInnovation(*curr_frame);
      eraseNoDepthPoint(*curr_frame);
      pose_records.push_back({(int)curr_frame->frame_id, curr_frame->T_c_w});  // STEP7, f2f_tracking.cpp:329-337
      if (pose_records.size() >= 1000) pose_records.pop_front();
      SE3 T_diff_key_curr = se3_mul(T_c_w_last_keyframe, se3_inverse(curr_frame->T_c_w));
      Vec3 t = T_diff_key_curr.t, r = so3_log(T_diff_key_curr.q);
      double t_norm = std::fabs(t.x) + std::fabs(t.y) + std::fabs(t.z);
      double r_norm = std::fabs(r.x) + std::fabs(r.y) + std::fabs(r.z);
      if (frameCount < 40 && (frameCount % 5) == 0) {
        new_keyframe = true;
        T_c_w_last_keyframe = curr_frame->T_c_w;
      }
      if (t_norm >= 0.05 || r_norm >= 0.2) {
        new_keyframe = true;
        T_c_w_last_keyframe = curr_frame->T_c_w;
      }
      break;
    }
    case TrackingFail: {
      trackingfail_cnt++;
      if ((trackingfail_cnt % 3) == 0) {
        if (vimotion->viGetCorrFrameState(curr_frame->frame_time, curr_frame->T_c_w)) {
          if (init_frame()) {
            new_keyframe = true;
            vo_tracking_state = Tracking;
          } else {
            std::swap(last_frame, curr_frame);
          }
        } else {
          std::swap(last_frame, curr_frame);
        }
        trackingfail_cnt = 0;
      } else {
        std::swap(last_frame, curr_frame);
        if ((trackingfail_cnt % 2) == 0) reset_cmd = true;
      }
      break;
    }
  }
  if (new_keyframe) {  // (addition) hand the gyro preintegration since the previous keyframe to the KeyFrame payload, restart it
    const bool chained = state_in == Tracking && vimotion != nullptr;  // a keyframe of init_frame() starts a new chain
    kf_imu_dq = vimotion ? vimotion->kf_dq : quat_identity();
    kf_imu_dt = vimotion ? vimotion->kf_dt : 0.0;
    kf_imu_valid = chained && kf_imu_dt > 0;
    kf_imu_dp = vimotion ? vimotion->kf_dp : Vec3{0, 0, 0};
    kf_imu_va = kf_va_next;
    if (vimotion) {
      vimotion->kf_dq = quat_identity();
      vimotion->kf_dt = 0;
      vimotion->kf_dp = vimotion->kf_dv = Vec3{0, 0, 0};
      kf_va_next = vimotion->states.empty() ? Vec3{0, 0, 0} : vimotion->states.back().vel;
    }
  }
}

void F2FTracking::correction_feed(const CorrectionInfStruct& corr) {  // f2f_tracking.cpp:40-44
  correction_inf = corr;
  has_localmap_feedback = true;
}

void F2FTracking::getKeyFrameInf(KeyFrameStruct& kf) const {
  kf = KeyFrameStruct();
  kf.frame_id = curr_frame->frame_id;
  kf.T_c_w = curr_frame->T_c_w;
  for (auto& lm : curr_frame->landmarks)
    if (lm.has_3d && lm.is_tracking_inlier) {
      kf.lm_3d.push_back(lm.lm_3d_w);
      kf.lm_2d.push_back(lm.lm_2d_undistort);
      kf.lm_id.push_back(lm.lm_id);
    }
  kf.lm_count = (int)kf.lm_id.size();
  kf.imu_dq = kf_imu_dq;
  kf.imu_dt = kf_imu_dt;
  kf.imu_valid = kf_imu_valid;
  kf.imu_dp = kf_imu_dp;
  kf.imu_va = kf_imu_va;
}

}  // namespace ref

// ------------------------------------------------------------------------------------------ C entry points (ctypes)
extern "C" {
int ref_glibc_rand_check(int n, int* out) {
  ref::GlibcRand g;
  for (int i = 0; i < n; i++) out[i] = g.next();
  return n;
}
int ref_config_sizeof(void) { return (int)sizeof(ref::Config); }
int ref_config_load_yaml(const char* path, ref::Config* c, char* err, int errlen) {
  return ref::config_load_yaml(path, *c, err, errlen) ? 1 : 0;
}
int ref_config_finalize(ref::Config* c) { return ref::config_finalize(*c) ? 1 : 0; }

void* ref_tracker_create(const ref::Config* cfg, uint64_t seed) { return new ref::F2FTracking(*cfg, seed); }
// CameraFrame::recover3DPts_c_FromStereo alone, on a tracker's rig and rand() generator; out3 = pt3ds [n][3], mask = maskHas3DInf
void ref_tracker_stereo_depth(void* h, const uint8_t* img0, const uint8_t* img1, int n, const float* pt2d_plane, const float* pt2d_undistort,
                              const float* pt3d_w, const uint8_t* has_depth, const double* T_c_w7, float range, double* out3, uint8_t* mask) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  ref::SE3 T;
  T.t = {T_c_w7[0], T_c_w7[1], T_c_w7[2]};
  T.q.x = T_c_w7[3], T.q.y = T_c_w7[4], T.q.z = T_c_w7[5], T.q.w = T_c_w7[6];
  std::vector<ref::Vec3> meas((size_t)n);
  f->recover3DPts_c_FromStereo(img0, img1, n, pt2d_plane, pt2d_undistort, pt3d_w, has_depth, T, range, meas.data(), mask);
  for (int i = 0; i < n; i++) out3[3 * i] = meas[i].x, out3[3 * i + 1] = meas[i].y, out3[3 * i + 2] = meas[i].z;
}

void ref_tracker_destroy(void* h) { delete (ref::F2FTracking*)h; }
// acc/gyro already remapped to the FLVIS IMU frame (vo_tracking.cpp:331-357); out10 = q(wxyz) p(3) v(3)
void ref_tracker_imu(void* h, double t, const double* acc, const double* gyro, double* out10) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  ref::Quat q;
  ref::Vec3 p, v;
  f->imu_feed(t, {acc[0], acc[1], acc[2]}, {gyro[0], gyro[1], gyro[2]}, q, p, v);
  if (out10) {
    double o[10] = {q.w, q.x, q.y, q.z, p.x, p.y, p.z, v.x, v.y, v.z};
    memcpy(out10, o, sizeof(o));
  }
}
// returns flags: bit0 new_keyframe, bit1 reset_cmd; state in *state; pose7 (tx ty tz qx qy qz qw) of curr_frame
int ref_tracker_image(void* h, double t, const uint8_t* img0, const uint8_t* img1, int* state, double* pose7,
                      int* n_landmarks, int* dbg3) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  bool kf = false, rst = false;
  f->image_feed(t, img0, img1, kf, rst);
  *state = f->vo_tracking_state;
  const ref::SE3& T = f->curr_frame->T_c_w;
  double o[7] = {T.t.x, T.t.y, T.t.z, T.q.x, T.q.y, T.q.z, T.q.w};
  memcpy(pose7, o, sizeof(o));
  *n_landmarks = (int)f->curr_frame->landmarks.size();
  if (dbg3) {
    dbg3[0] = f->dbg_of_inlier;
    dbg3[1] = f->dbg_F_inlier;
    dbg3[2] = f->dbg_pnp_inlier;
  }
  return (kf ? 1 : 0) | (rst ? 2 : 0);
}
// landmark dump of curr_frame: ids, 2d plane, 2d undist, 3d world, flags(has_3d | inlier<<1)
int ref_tracker_landmarks(void* h, int cap, int64_t* ids, double* p2d, double* p2u, double* p3w, uint8_t* flags) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  int n = (int)f->curr_frame->landmarks.size();
  for (int i = 0; i < n && i < cap; i++) {
    const ref::LandMarkInFrame& lm = f->curr_frame->landmarks[i];
    ids[i] = lm.lm_id;
    p2d[2 * i] = lm.lm_2d_plane.x;
    p2d[2 * i + 1] = lm.lm_2d_plane.y;
    p2u[2 * i] = lm.lm_2d_undistort.x;
    p2u[2 * i + 1] = lm.lm_2d_undistort.y;
    p3w[3 * i] = lm.lm_3d_w.x;
    p3w[3 * i + 1] = lm.lm_3d_w.y;
    p3w[3 * i + 2] = lm.lm_3d_w.z;
    flags[i] = (uint8_t)((lm.has_3d ? 1 : 0) | (lm.is_tracking_inlier ? 2 : 0));
  }
  return n;
}
int ref_tracker_keyframe(void* h, int cap, int64_t* frame_id, double* pose7, int64_t* ids, double* p2u, double* p3w) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  ref::KeyFrameStruct kf;
  f->getKeyFrameInf(kf);
  *frame_id = kf.frame_id;
  double o[7] = {kf.T_c_w.t.x, kf.T_c_w.t.y, kf.T_c_w.t.z, kf.T_c_w.q.x, kf.T_c_w.q.y, kf.T_c_w.q.z, kf.T_c_w.q.w};
  memcpy(pose7, o, sizeof(o));
  for (int i = 0; i < kf.lm_count && i < cap; i++) {
    ids[i] = kf.lm_id[i];
    p2u[2 * i] = kf.lm_2d[i].x;
    p2u[2 * i + 1] = kf.lm_2d[i].y;
    p3w[3 * i] = kf.lm_3d[i].x;
    p3w[3 * i + 1] = kf.lm_3d[i].y;
    p3w[3 * i + 2] = kf.lm_3d[i].z;
  }
  return kf.lm_count;
}
// (addition) the gyro preintegration attached to the last keyframe: dq (w, x, y, z) = body rotation since the previous keyframe
int ref_tracker_keyframe_imu(void* h, double* dq_wxyz, double* dt) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  dq_wxyz[0] = f->kf_imu_dq.w, dq_wxyz[1] = f->kf_imu_dq.x, dq_wxyz[2] = f->kf_imu_dq.y, dq_wxyz[3] = f->kf_imu_dq.z;
  *dt = f->kf_imu_dt;
  return f->kf_imu_valid ? 1 : 0;
}
// ... and the position part: dp (body frame of the previous keyframe) and that keyframe's body velocity (world)
void ref_tracker_keyframe_imu_pos(void* h, double* dp3, double* va3) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  dp3[0] = f->kf_imu_dp.x, dp3[1] = f->kf_imu_dp.y, dp3[2] = f->kf_imu_dp.z;
  va3[0] = f->kf_imu_va.x, va3[1] = f->kf_imu_va.y, va3[2] = f->kf_imu_va.z;
}
// F2FTracking::correction_feed (f2f_tracking.cpp:40-44) with the CorrectionInf fields flattened
void ref_tracker_correction_feed(void* h, int64_t frame_id, const double* pose7, int lm_count, const int64_t* lm_id,
                                 const double* lm_3d, int outlier_count, const int64_t* outlier_id) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  ref::CorrectionInfStruct c;
  c.frame_id = frame_id;
  c.T_c_w.t = {pose7[0], pose7[1], pose7[2]};
  c.T_c_w.q.x = pose7[3], c.T_c_w.q.y = pose7[4], c.T_c_w.q.z = pose7[5], c.T_c_w.q.w = pose7[6];
  c.lm_count = lm_count;
  for (int i = 0; i < lm_count; i++) {
    c.lm_id.push_back(lm_id[i]);
    c.lm_3d.push_back({lm_3d[3 * i], lm_3d[3 * i + 1], lm_3d[3 * i + 2]});
  }
  c.lm_outlier_count = outlier_count;
  for (int i = 0; i < outlier_count; i++) c.lm_outlier_id.push_back(outlier_id[i]);
  f->correction_feed(c);
}
// pose_records dump: rows (frame_id, pose7), oldest first
// stage poses of the last Tracking frame (tests): out21 = pose7 after solvePnPRansac, pose7 after OptimizeInFrame
void ref_tracker_stage_poses(void* h, double* out21) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  const ref::SE3* T[3] = {&f->dbg_T_pnp, &f->dbg_T_lm, &f->dbg_T_pre};
  for (int k = 0; k < 3; k++) {
    const double o[7] = {T[k]->t.x, T[k]->t.y, T[k]->t.z, T[k]->q.x, T[k]->q.y, T[k]->q.z, T[k]->q.w};
    memcpy(out21 + 7 * k, o, sizeof(o));
  }
}
int ref_tracker_pose_records(void* h, int cap, double* rows8) {
  ref::F2FTracking* f = (ref::F2FTracking*)h;
  int n = (int)f->pose_records.size();
  for (int i = 0; i < n && i < cap; i++) {
    const ref::SE3& T = f->pose_records[i].T_c_w;
    double o[8] = {(double)f->pose_records[i].frame_id, T.t.x, T.t.y, T.t.z, T.q.x, T.q.y, T.q.z, T.q.w};
    memcpy(rows8 + 8 * i, o, sizeof(o));
  }
  return n;
}
// ---- standalone VIMOTION handles (tests/test_oracle_vimotion.py): the filter driven exactly as F2FTracking drives it
static ref::SE3 pose7_to_se3(const double* p) {
  ref::SE3 T;
  T.t = {p[0], p[1], p[2]};
  T.q.x = p[3], T.q.y = p[4], T.q.z = p[5], T.q.w = p[6];
  return T;
}
static void se3_to_pose7(const ref::SE3& T, double* p) {
  double o[7] = {T.t.x, T.t.y, T.t.z, T.q.x, T.q.y, T.q.z, T.q.w};
  memcpy(p, o, sizeof(o));
}
void* ref_vi_create(const double* T_i_c_pose7, double g, const double* para6) {
  return new ref::VIMOTION(pose7_to_se3(T_i_c_pose7), g, para6[0], para6[1], para6[2], para6[3], para6[4], para6[5]);
}
void ref_vi_destroy(void* h) { delete (ref::VIMOTION*)h; }
void ref_vi_feed(void* h, double t, const double* acc, const double* gyro, double* out10) {  // f2f_tracking.cpp:46-57
  ref::VIMOTION* v = (ref::VIMOTION*)h;
  ref::IMUSTATE s{{acc[0], acc[1], acc[2]}, {gyro[0], gyro[1], gyro[2]}, t};
  ref::Quat q;
  ref::Vec3 p, vel;
  if (!v->imu_initialized)
    v->viIMUinitialization(s, q, p, vel);
  else
    v->viIMUPropagation(s, q, p, vel);
  double o[10] = {q.w, q.x, q.y, q.z, p.x, p.y, p.z, vel.x, vel.y, vel.z};
  memcpy(out10, o, sizeof(o));
}
void ref_vi_vision_trigger(void* h, double* q_wxyz) {
  ref::Quat q;
  ((ref::VIMOTION*)h)->viVisiontrigger(q);
  q_wxyz[0] = q.w, q_wxyz[1] = q.x, q_wxyz[2] = q.y, q_wxyz[3] = q.z;
}
int ref_vi_get_corr_frame_state(void* h, double time, double* pose7) {
  ref::SE3 T = ref::se3_identity();
  bool ok = ((ref::VIMOTION*)h)->viGetCorrFrameState(time, T);
  se3_to_pose7(T, pose7);
  return ok ? 1 : 0;
}
void ref_vi_rp_compensation(void* h, double time, double* pose7_inout) {
  ref::SE3 T = pose7_to_se3(pose7_inout);
  ((ref::VIMOTION*)h)->viVisionRPCompensation(time, T);
  se3_to_pose7(T, pose7_inout);
}
void ref_vi_correction(void* h, double t_curr, const double* pose7_curr, double t_last, const double* pose7_last) {
  ((ref::VIMOTION*)h)->viCorrectionFromVision(t_curr, pose7_to_se3(pose7_curr), t_last, pose7_to_se3(pose7_last), 0.0);
}
// rows of 11: t, q (w x y z), pos, vel; biases6: acc bias, gyro bias.  Returns the queue length.
int ref_vi_states(void* h, int cap, double* rows11, double* biases6) {
  ref::VIMOTION* v = (ref::VIMOTION*)h;
  int n = (int)v->states.size();
  for (int i = 0; i < n && i < cap; i++) {
    const ref::MOTION_STATE& s = v->states[i];
    double o[11] = {s.imu_data.timestamp, s.q_w_i.w, s.q_w_i.x, s.q_w_i.y, s.q_w_i.z, s.pos.x, s.pos.y, s.pos.z, s.vel.x, s.vel.y, s.vel.z};
    memcpy(rows11 + 11 * i, o, sizeof(o));
  }
  double b[6] = {v->acc_bias.x, v->acc_bias.y, v->acc_bias.z, v->gyro_bias.x, v->gyro_bias.y, v->gyro_bias.z};
  memcpy(biases6, b, sizeof(b));
  return n;
}
}
