// TEST INFRASTRUCTURE (CPU oracle, see oracle/README.md): restatement of STEP 1.5 / 1.6 of the loop-closing keyframe construction
// (SURVEY 8f-4): the 3-D position of every ORB keypoint of a keyframe -- stereo LK into the second image + DLT triangulation, or a
// lookup in the depth image -- and the removal of the keypoints without one.
//
//   src/backend/vo_loopclosing.cpp:255-350   the cam_type switch (STEREO_RECT / STEREO_UNRECT (empty) / DEPTH_D435)
//   src/backend/vo_loopclosing.cpp:352-372   lm_2d / lm_3d / lm_descriptor keep the entries whose mask is true, in order
//   src/processing/triangulation.cpp:9-54    triangulationPt + trignaulationPtFromStereo (range 100.0f, triangulation.h:24)
//
// parity unpinned: the arithmetic is cv::calcOpticalFlowPyrLK (restated in ref_image.cpp) and Eigen::JacobiSVD (restated in
// ref_geom.cpp); neither library is available here.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ref_api.h"

extern "C" {

// cam_type 0 STEREO_RECT (img1 = mono8), 1 STEREO_UNRECT (the reference's case is empty: nothing is kept), 2 DEPTH_D435 (img1 = Z16).
// kps [n][6] (x, y, ...), desc [n][32].  out: lm_2d [<=n][2] float, lm_3d [<=n][3], lm_desc [<=n][32]; returns lm_count.
int ref_lc_keyframe_landmarks(const uint8_t* img0, const void* img1, int w, int h, int cam_type, const double* P0, const double* P1,
                              const double* K4, const float* kps, const uint8_t* desc, int n, float* lm_2d, double* lm_3d,
                              uint8_t* lm_desc) {
  std::vector<uint8_t> mask((size_t)n, 0);
  std::vector<double> p3((size_t)n * 3, 0.0);
  if (cam_type == 0) {
    std::vector<float> p0((size_t)n * 2), p1((size_t)n * 2);
    std::vector<uint8_t> status((size_t)n, 0);
    for (int i = 0; i < n; i++) {
      p0[2 * i] = p1[2 * i] = kps[6 * i];
      p0[2 * i + 1] = p1[2 * i + 1] = kps[6 * i + 1];
    }
    if (n > 0)
      ref::calc_optical_flow_pyr_lk(img0, (const uint8_t*)img1, w, h, p0.data(), p1.data(), status.data(), n, 31, 5, 30, 0.001, 1, 1e-4f);
    const float range = 100.0f;
    for (int i = 0; i < n; i++) {
      if (status[i] != 1) continue;
      const ref::Vec3 pc = ref::triangulate_dlt({(double)p0[2 * i], (double)p0[2 * i + 1]}, {(double)p1[2 * i], (double)p1[2 * i + 1]}, P0, P1);
      if (pc.z < 0 || pc.z > range) continue;
      mask[i] = 1;
      p3[3 * i] = pc.x, p3[3 * i + 1] = pc.y, p3[3 * i + 2] = pc.z;
    }
  } else if (cam_type == 2) {
    const uint16_t* dimg = (const uint16_t*)img1;
    for (int i = 0; i < n; i++) {
      const float x = kps[6 * i], y = kps[6 * i + 1];
      // img1.at<ushort>(Point2f): Point2f -> Point2i rounds to nearest-even (cv::saturate_cast<int>(float) = cvRound)
      int ix = (int)std::nearbyint(x), iy = (int)std::nearbyint(y);
      ix = ix < 0 ? 0 : (ix > w - 1 ? w - 1 : ix);
      iy = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy);
      const double d = (double)(dimg[(size_t)iy * w + ix] / 1000);  // `(ushort)/1000` is an INTEGER division in the reference (:331)
      if (d >= 0.3 && d <= 10) {
        mask[i] = 1;
        p3[3 * i] = ((double)x - K4[2]) / K4[0] * d;
        p3[3 * i + 1] = ((double)y - K4[3]) / K4[1] * d;
        p3[3 * i + 2] = d;
      }
    }
  }
  int k = 0;
  for (int i = 0; i < n; i++) {
    if (!mask[i]) continue;
    lm_2d[2 * k] = kps[6 * i];
    lm_2d[2 * k + 1] = kps[6 * i + 1];
    memcpy(lm_3d + 3 * k, &p3[3 * i], 3 * sizeof(double));
    memcpy(lm_desc + (size_t)32 * k, desc + (size_t)32 * i, 32);
    k++;
  }
  return k;
}
}
