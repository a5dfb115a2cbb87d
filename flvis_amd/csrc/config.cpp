// flvis_amd: configuration surface of the drop-in boundary (host side).
//   flvis_config_load      accepts the reference's yaml files unchanged (flat "key: scalar" / "key: [list]" with '#' comments,
//                          src/utils/include/yamlRead.h; keys per launch/EuRoC_MAV/euroc.yaml, launch/d435i/*_stereo.yaml)
//   flvis_config_finalize  derives what TrackingNodeletClass::onInit derives (src/frontend/vo_tracking.cpp:155-306): camera
//                          type, skip/equalise flags and the rectification R0,R1,P0,P1 of cv::stereoRectify(CALIB_ZERO_DISPARITY,
//                          alpha = 0) restated from OpenCV 3.x's cvStereoRectify (external dependency of the reference).
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/flvis_hip.h"

namespace {

struct M3h {
  double m[3][3];
};
struct V3h {
  double v[3];
};
M3h mul(const M3h& a, const M3h& b) {
  M3h r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
V3h mulv(const M3h& a, const V3h& x) {
  V3h r;
  for (int i = 0; i < 3; i++) r.v[i] = a.m[i][0] * x.v[0] + a.m[i][1] * x.v[1] + a.m[i][2] * x.v[2];
  return r;
}
M3h transp(const M3h& a) {
  M3h r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
M3h ident() { return M3h{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
V3h rotvec_from_mat(const M3h& R) {
  double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
  double c = (tr - 1) * 0.5;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  double theta = std::acos(c);
  V3h r{{R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]}};
  double s = std::sqrt(r.v[0] * r.v[0] + r.v[1] * r.v[1] + r.v[2] * r.v[2]);  // 2 sin(theta)
  if (s < 1e-12) return V3h{{0.5 * r.v[0], 0.5 * r.v[1], 0.5 * r.v[2]}};
  double k = theta / s;
  return V3h{{k * r.v[0], k * r.v[1], k * r.v[2]}};
}
M3h mat_from_rotvec(const V3h& r) {
  double th = std::sqrt(r.v[0] * r.v[0] + r.v[1] * r.v[1] + r.v[2] * r.v[2]);
  if (th < 1e-15) return ident();
  double x = r.v[0] / th, y = r.v[1] / th, z = r.v[2] / th, c = std::cos(th), s = std::sin(th), C = 1 - c;
  return M3h{{{c + x * x * C, x * y * C - z * s, x * z * C + y * s},
              {y * x * C + z * s, c + y * y * C, y * z * C - x * s},
              {z * x * C - y * s, z * y * C + x * s, c + z * z * C}}};
}

// cv::undistortPoints(pt, K, D, R, P) for one point (5 iterations)
void undistort_pt(double u, double v, const double* K, const double* D, const M3h& R, const double* P, double& ou, double& ov) {
  double x = (u - K[2]) / K[0], y = (v - K[3]) / K[1], x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    double r2 = x * x + y * y;
    double icdist = 1. / (1 + (D[1] * r2 + D[0]) * r2);
    double dX = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x), dY = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
    x = (x0 - dX) * icdist;
    y = (y0 - dY) * icdist;
  }
  double xx = R.m[0][0] * x + R.m[0][1] * y + R.m[0][2], yy = R.m[1][0] * x + R.m[1][1] * y + R.m[1][2];
  double ww = 1. / (R.m[2][0] * x + R.m[2][1] * y + R.m[2][2]);
  ou = (double)(float)(xx * ww * P[0] + P[2]);
  ov = (double)(float)(yy * ww * P[5] + P[6]);
}

void stereo_rectify(const double* K1, const double* D1, const double* K2, const double* D2, int nx, int ny, const M3h& R,
                    const V3h& T, M3h& R1, M3h& R2, double* P1, double* P2) {
  V3h om = rotvec_from_mat(R);
  for (int i = 0; i < 3; i++) om.v[i] *= -0.5;
  M3h r_r = mat_from_rotvec(om);
  V3h t = mulv(r_r, T);
  int idx = std::fabs(t.v[0]) > std::fabs(t.v[1]) ? 0 : 1;
  double c = t.v[idx], nt = std::sqrt(t.v[0] * t.v[0] + t.v[1] * t.v[1] + t.v[2] * t.v[2]);
  V3h uu{{0, 0, 0}};
  uu.v[idx] = c > 0 ? 1 : -1;
  V3h ww{{t.v[1] * uu.v[2] - t.v[2] * uu.v[1], t.v[2] * uu.v[0] - t.v[0] * uu.v[2], t.v[0] * uu.v[1] - t.v[1] * uu.v[0]}};
  double nw = std::sqrt(ww.v[0] * ww.v[0] + ww.v[1] * ww.v[1] + ww.v[2] * ww.v[2]);
  if (nw > 0.0) {
    double k = std::acos(std::fabs(c) / nt) / nw;
    for (int i = 0; i < 3; i++) ww.v[i] *= k;
  }
  M3h wR = mat_from_rotvec(ww);
  R1 = mul(wR, transp(r_r));
  R2 = mul(wR, r_r);
  t = mulv(R2, T);
  double fc_new = DBL_MAX;
  const double* Ks[2] = {K1, K2};
  const double* Ds[2] = {D1, D2};
  for (int k = 0; k < 2; k++) {
    double dk1 = Ds[k][0];
    double fc = idx == 0 ? Ks[k][1] : Ks[k][0];
    if (dk1 < 0) fc *= 1 + dk1 * (nx * nx + ny * ny) / (4 * fc * fc);
    fc_new = std::min(fc_new, fc);
  }
  double ccx[2], ccy[2];
  const M3h* Rs[2] = {&R1, &R2};
  const double Pn[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < 2; k++) {
    double ax = 0, ay = 0;
    for (int i = 0; i < 4; i++) {
      int j = (i < 2) ? 0 : 1;
      double px = (double)(float)((i % 2) * (nx - 1)), py = (double)(float)(j * (ny - 1));
      double ux, uy;
      undistort_pt(px, py, Ks[k], Ds[k], ident(), Pn, ux, uy);
      V3h X = mulv(*Rs[k], V3h{{ux, uy, 1.0}});
      ax += (double)(float)(fc_new * X.v[0] / X.v[2]);
      ay += (double)(float)(fc_new * X.v[1] / X.v[2]);
    }
    ccx[k] = (nx - 1) / 2 - ax / 4;  // integer (nx-1)/2 as in cvStereoRectify
    ccy[k] = (ny - 1) / 2 - ay / 4;
  }
  ccx[0] = ccx[1] = (ccx[0] + ccx[1]) * 0.5;
  ccy[0] = ccy[1] = (ccy[0] + ccy[1]) * 0.5;
  for (int i = 0; i < 12; i++) P1[i] = P2[i] = 0;
  P1[0] = P1[5] = P2[0] = P2[5] = fc_new;
  P1[2] = ccx[0];
  P1[6] = ccy[0];
  P2[2] = ccx[1];
  P2[6] = ccy[1];
  P1[10] = P2[10] = 1;
  P2[4 * idx + 3] = t.v[idx] * fc_new;
  double inner[2][4];
  double* Ps[2] = {P1, P2};
  for (int k = 0; k < 2; k++) {
    const int N = 9;
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    for (int y = 0; y < N; y++)
      for (int x = 0; x < N; x++) {
        double qx, qy;
        undistort_pt((double)((float)x * nx / (N - 1)), (double)((float)y * ny / (N - 1)), Ks[k], Ds[k], *Rs[k], Ps[k], qx, qy);
        if (x == 0) iX0 = std::max(iX0, (float)qx);
        if (x == N - 1) iX1 = std::min(iX1, (float)qx);
        if (y == 0) iY0 = std::max(iY0, (float)qy);
        if (y == N - 1) iY1 = std::min(iY1, (float)qy);
      }
    inner[k][0] = iX0;
    inner[k][1] = iY0;
    inner[k][2] = iX1 - iX0;
    inner[k][3] = iY1 - iY0;
  }
  double cx1 = ccx[0], cy1 = ccy[0], cx2 = ccx[1], cy2 = ccy[1];
  double s0 = std::max(std::max(std::max(cx1 / (cx1 - inner[0][0]), cy1 / (cy1 - inner[0][1])),
                                (nx - cx1) / (inner[0][0] + inner[0][2] - cx1)),
                       (ny - cy1) / (inner[0][1] + inner[0][3] - cy1));
  s0 = std::max(std::max(std::max(std::max(cx2 / (cx2 - inner[1][0]), cy2 / (cy2 - inner[1][1])),
                                  (nx - cx2) / (inner[1][0] + inner[1][2] - cx2)),
                         (ny - cy2) / (inner[1][1] + inner[1][3] - cy2)),
                s0);
  fc_new *= s0;
  P1[0] = P1[5] = P2[0] = P2[5] = fc_new;
  P2[4 * idx + 3] = s0 * P2[4 * idx + 3];
}

void mat44_inverse_rigid(const double* m, double* o) {
  // [R t; 0 1]^-1 = [R^T  -R^T t]
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) o[4 * i + j] = m[4 * j + i];
  for (int i = 0; i < 3; i++) o[4 * i + 3] = -(o[4 * i] * m[3] + o[4 * i + 1] * m[7] + o[4 * i + 2] * m[11]);
  o[12] = o[13] = o[14] = 0;
  o[15] = 1;
}
void mat44_mul(const double* a, const double* b, double* o) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j];
      o[4 * i + j] = s;
    }
}

}  // namespace

extern "C" int flvis_config_finalize(flvis_cfg* c) {
  if (!c) return FLVIS_ERR_INVALID_ARG;
  switch (c->type_of_vi) {
    case 1:
      c->cam_type = 1;  // STEREO_UNRECT
      c->imu_type = 1;  // EuRoC axis remap
      c->skip_first_n_imgs = 0;
      c->need_equal_hist = 1;
      break;
    case 3:
      c->cam_type = 0;  // STEREO_RECT
      c->imu_type = 0;  // D435i remap
      c->skip_first_n_imgs = 50;
      c->need_equal_hist = 0;
      break;
    case 5:
      c->cam_type = 0;
      c->imu_type = 2;  // pixhawk
      c->skip_first_n_imgs = 50;
      c->need_equal_hist = 0;
      break;
    case 0:
    case 2: {
      // DEPTH_D435 (vo_tracking.cpp:142-170): pinhole K from cam0_intrinsics, no distortion, no second camera
      c->cam_type = 2;
      c->imu_type = c->type_of_vi == 0 ? 0 : 2;
      c->skip_first_n_imgs = 50;
      c->need_equal_hist = 0;
      const double P[12] = {c->cam0_intrinsics[0], 0, c->cam0_intrinsics[2], 0, 0, c->cam0_intrinsics[1], c->cam0_intrinsics[3], 0,
                            0, 0, 1, 0};
      memcpy(c->P0, P, sizeof(P));
      memset(c->P1, 0, sizeof(c->P1));
      for (int i = 0; i < 9; i++) c->R0[i] = c->R1[i] = (i % 4 == 0) ? 1.0 : 0.0;
      return c->depth_factor > 0 ? FLVIS_OK : FLVIS_ERR_CONFIG;
    }
    case 4: {
      // VI_TYPE_KITTI_STEREO (vo_tracking.cpp:146,265-306): STEREO_RECT without IMU.  The rig comes from the two projection
      // matrices (flvis_config_load put their first three rows into P0 / P1): K0 = K1 = K0_rect = P0(0:3,0:3), no distortion,
      // R0 = R1 = I, T_c0_c1 = [I | K^-1 * P1(:,3)] with K^-1 as Eigen's 3x3 inverse computes it (cofactors * (1 / det)),
      // T_i_c0 is the dummy SE3(), init(..., 0, false): no skipped frames, no equalizeHist.
      c->cam_type = 0;
      c->imu_type = 3;  // NONE: imu_callback has no remap for it; flvis_imu_feed refuses samples for such a rig
      c->skip_first_n_imgs = 0;
      c->need_equal_hist = 0;
      const double fx = c->P0[0], fy = c->P0[5], cx = c->P0[2], cy = c->P0[6];
      if (!(fx > 0) || !(fy > 0)) return FLVIS_ERR_CONFIG;
      const double K[4] = {fx, fy, cx, cy}, Z[4] = {0, 0, 0, 0};
      memcpy(c->cam0_intrinsics, K, sizeof(K));
      memcpy(c->cam1_intrinsics, K, sizeof(K));
      memcpy(c->cam0_distortion, Z, sizeof(Z));
      memcpy(c->cam1_distortion, Z, sizeof(Z));
      for (int i = 0; i < 9; i++) c->R0[i] = c->R1[i] = (i % 4 == 0) ? 1.0 : 0.0;
      // Eigen compute_inverse_size3 on K = [fx 0 cx; 0 fy cy; 0 0 1]: cofactor(i,j) * invdet, det = fx * cof00 (+ 0 + 0)
      const double c00 = fy * 1.0 - cy * 0.0, det = fx * c00, invdet = 1.0 / det;
      const double i00 = c00 * invdet, i01 = (cx * 0.0 - 0.0 * 1.0) * invdet, i02 = (0.0 * cy - cx * fy) * invdet;
      const double i10 = (cy * 0.0 - 0.0 * 1.0) * invdet, i11 = (fx * 1.0 - cx * 0.0) * invdet, i12 = (0.0 * cx - fx * cy) * invdet;
      const double i20 = (0.0 * 0.0 - 0.0 * fy) * invdet, i21 = (0.0 * 0.0 - fx * 0.0) * invdet, i22 = (fx * fy - 0.0 * 0.0) * invdet;
      const double p0 = c->P1[3], p1 = c->P1[7], p2 = c->P1[11];  // P1(:,3); the 4x4's last row is zero in the yaml
      const double tx = i00 * p0 + i01 * p1 + i02 * p2, ty = i10 * p0 + i11 * p1 + i12 * p2, tz = i20 * p0 + i21 * p1 + i22 * p2;
      const double T01[16] = {1, 0, 0, tx, 0, 1, 0, ty, 0, 0, 1, tz, 0, 0, 0, 1};
      const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      memcpy(c->T_cam0_cam1, T01, sizeof(T01));
      memcpy(c->T_imu_cam0, eye, sizeof(eye));
      return FLVIS_OK;
    }
    default:
      return FLVIS_ERR_CONFIG;
  }
  double Tinv[16];
  mat44_inverse_rigid(c->T_cam0_cam1, Tinv);  // T_c1_c0
  M3h R{{{Tinv[0], Tinv[1], Tinv[2]}, {Tinv[4], Tinv[5], Tinv[6]}, {Tinv[8], Tinv[9], Tinv[10]}}};
  V3h T{{Tinv[3], Tinv[7], Tinv[11]}};
  M3h R0, R1;
  stereo_rectify(c->cam0_intrinsics, c->cam0_distortion, c->cam1_intrinsics, c->cam1_distortion, c->image_width,
                 c->image_height, R, T, R0, R1, c->P0, c->P1);
  for (int i = 0; i < 9; i++) {
    c->R0[i] = R0.m[i / 3][i % 3];
    c->R1[i] = R1.m[i / 3][i % 3];
  }
  return FLVIS_OK;
}

extern "C" int flvis_config_load(const char* path, flvis_cfg* c, char* err, int errlen) {
  auto fail = [&](const std::string& m) {
    if (err && errlen > 0) snprintf(err, errlen, "%s", m.c_str());
    return (int)FLVIS_ERR_CONFIG;
  };
  if (!path || !c) return FLVIS_ERR_INVALID_ARG;
  std::ifstream f(path);
  if (!f) return fail(std::string("cannot open ") + path);
  std::map<std::string, std::vector<double>> kv;
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
    size_t a = key.find_first_not_of(" \t");
    if (a == std::string::npos) continue;
    key = key.substr(a);
    key.erase(key.find_last_not_of(" \t") + 1);
    std::string val = line.substr(col + 1);
    if (val.find('[') != std::string::npos) {
      acc = val;
      if (val.find(']') != std::string::npos)
        flush_list(key, acc);
      else
        in_list = true;
    } else {
      size_t b = val.find_first_not_of(" \t\r");
      if (b == std::string::npos) {
        acc = "";
        in_list = true;
        continue;
      }
      val = val.substr(b);
      val.erase(val.find_last_not_of(" \t\r") + 1);
      if (val == "True" || val == "true")
        kv[key] = {1};
      else if (val == "False" || val == "false")
        kv[key] = {0};
      else
        kv[key] = {atof(val.c_str())};
    }
  }
  std::string missing;
  auto need = [&](const char* k, size_t n, double* dst) -> bool {
    auto it = kv.find(k);
    if (it == kv.end() || it->second.size() < n) {
      missing = k;
      return false;
    }
    for (size_t i = 0; i < n; i++) dst[i] = it->second[i];
    return true;
  };
  memset(c, 0, sizeof(*c));
  double v;
  if (!need("type_of_vi", 1, &v)) return fail("yaml key missing: type_of_vi");
  c->type_of_vi = (int)v;
  if (!need("image_width", 1, &v)) return fail("yaml key missing: image_width");
  c->image_width = (int)v;
  if (!need("image_height", 1, &v)) return fail("yaml key missing: image_height");
  c->image_height = (int)v;
  const bool depth_mode = c->type_of_vi == 0 || c->type_of_vi == 2;
  if (!need("cam0_intrinsics", 4, c->cam0_intrinsics) || !need("cam0_distortion_coeffs", 4, c->cam0_distortion))
    return fail("yaml key missing or short: " + missing);
  if (depth_mode) {  // vo_tracking.cpp:149-154 reads only cam0, depth_factor and T_imu_cam0
    if (!need("depth_factor", 1, &c->depth_factor) || !need("T_imu_cam0", 16, c->T_imu_cam0))
      return fail("yaml key missing or short: " + missing);
    const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(c->T_cam0_cam1, eye, sizeof(eye));
  } else if (!need("cam1_intrinsics", 4, c->cam1_intrinsics) || !need("cam1_distortion_coeffs", 4, c->cam1_distortion)) {
    return fail("yaml key missing or short: " + missing);
  }
  if (depth_mode) {
  } else if (c->type_of_vi == 4) {  // KITTI: vo_tracking.cpp:267-270 reads the two 4x4 projection matrices
    double a[16], b[16];
    if (!need("cam0_projection_matrix", 16, a) || !need("cam1_projection_matrix", 16, b)) return fail("yaml key missing or short: " + missing);
    memcpy(c->P0, a, sizeof(double) * 12);
    memcpy(c->P1, b, sizeof(double) * 12);
  } else if (c->type_of_vi == 1) {
    double a[16], b[16], m[16], ai[16];
    if (!need("T_mavimu_cam0", 16, a) || !need("T_mavimu_cam1", 16, b) || !need("T_imu_mavimu", 16, m))
      return fail("yaml key missing or short: " + missing);
    mat44_inverse_rigid(a, ai);
    mat44_mul(ai, b, c->T_cam0_cam1);  // T_c0_c1 = T_mavimu_cam0^-1 * T_mavimu_cam1
    mat44_mul(m, a, c->T_imu_cam0);    // T_i_c0  = T_imu_mavimu * T_mavimu_cam0
  } else {
    if (!need("T_imu_cam0", 16, c->T_imu_cam0) || !need("T_cam0_cam1", 16, c->T_cam0_cam1))
      return fail("yaml key missing or short: " + missing);
  }
  const char* vk[6] = {"vifusion_para1", "vifusion_para2", "vifusion_para3", "vifusion_para4", "vifusion_para5", "vifusion_para6"};
  const char* fk[6] = {"feature_para1", "feature_para2", "feature_para3", "feature_para4", "feature_para5", "feature_para6"};
  const char* dk[3] = {"dr_para1", "dr_para2", "dr_para3"};
  for (int i = 0; i < 6; i++)
    if (!need(vk[i], 1, &c->vifusion_para[i]) || !need(fk[i], 1, &c->feature_para[i])) return fail("yaml key missing: " + missing);
  for (int i = 0; i < 3; i++)
    if (!need(dk[i], 1, &c->dr_para[i])) return fail("yaml key missing: " + missing);
  if (!need("window_size", 1, &v)) return fail("yaml key missing: window_size");
  c->window_size = (int)v;
  if (c->window_size < 3 || c->window_size > 100) c->window_size = 10;  // vo_localmap.cpp:443-447
  int rc = flvis_config_finalize(c);
  if (rc != FLVIS_OK) return fail("unsupported type_of_vi for this path");
  return FLVIS_OK;
}
