// flvis_amd: CameraFrame::recover3DPts_c_FromStereo (src/processing/camera_frame.cpp:93-180) as one call on caller arrays -- the
// kernel-level drop-in of BASELINE configs[1] (HIP front-end pieces under the reference's own frame loop; SURVEY 8b:
// flvis_hip_stereo_depth).  Two small kernels around the batched LK matcher (lk_kernel.hip):
//   k_sd_seeds  the matcher's initial guesses: the pixel itself, or -- for landmarks that carry depth -- the world point projected
//               into camera 1 by cv::projectPoints with T_cam1_cam0 * T_c_w (camera_frame.cpp:108-122);
//   k_sd_post   cv::undistortPoints(K1, D1, R1, P1) of the matches, Triangulation::trignaulationPtFromStereo (DLT, valid unless
//               z < 0 or z > range: triangulation.cpp:40-54), and for every failure a rand()-drawn dummy depth in [0.3, 0.7) through
//               the undistorted pixel (camera_frame.cpp:149-176), drawn in landmark order from the set's glibc generator.
// The same device functions as the tracker's own k_depth_seeds / k_depth_innovate (track_kernels.hip).
#include <cstring>

#include "dev_common.hpp"
#include "dev_geom.hpp"
#include "img_kernels.hpp"
#include "pipeline.hpp"

namespace flvis {

struct SdCam {
  double K1[4], D1[4], R1[9], P0[12], P1[12], T_c1_c0[7];
  double fx, fy, cx, cy;
};

__global__ __launch_bounds__(256) void k_sd_seeds(SdCam cam, const float* __restrict__ pt2d_plane, const float* __restrict__ pt3d_w,
                                                  const uint8_t* __restrict__ has_depth, const int* __restrict__ count, int cap,
                                                  const double* __restrict__ T_c_w7, float* __restrict__ seeds) {
  const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count[s] || i >= cap) return;
  const size_t k = (size_t)s * cap + i;
  float* p1 = seeds + 2 * k;
  p1[0] = pt2d_plane[2 * k];
  p1[1] = pt2d_plane[2 * k + 1];
  if (has_depth[k]) {
    const SE3d T1c = se3_mul(load_pose7(cam.T_c1_c0), load_pose7(T_c_w7 + 7 * s));
    const float p3[3] = {pt3d_w[3 * k], pt3d_w[3 * k + 1], pt3d_w[3 * k + 2]};
    project_point(p3, q_to_mat(T1c.q), T1c.t, cam.K1, cam.D1, p1);
  }
}

FD int sd_rand_next(int* r35) {  // glibc rand(), TYPE_3: r[i] = r[i - 31] + r[i - 3], result >> 1
  const int pos = r35[34];
  const int n = (int)((unsigned)r35[(pos + 34 - 31) % 34] + (unsigned)r35[(pos + 34 - 3) % 34]);
  r35[pos] = n;
  r35[34] = (pos + 1) % 34;
  return (int)(((unsigned)n) >> 1);
}

constexpr int SD_T = 1024;
__global__ __launch_bounds__(SD_T) void k_sd_post(SdCam cam, const float* __restrict__ pt2d_undistort, const float* __restrict__ matched,
                                                  const uint8_t* __restrict__ status, const int* __restrict__ count, int cap, float range,
                                                  int* __restrict__ rand_state35, double* __restrict__ pt3d_c,
                                                  uint8_t* __restrict__ mask_has_3d) {
  const int s = blockIdx.x, t = threadIdx.x;
  const int n = min(count[s], cap);
  __shared__ int s_cnt[SD_T / 64];
  __shared__ float s_rnd[SD_T];
  int* rs = rand_state35 + 35 * s;
  for (int base = 0; base < n; base += SD_T) {  // (batches in landmark order: the generator is consumed in that order)
    const int i = base + t;
    const bool valid = i < n;
    const size_t k = (size_t)s * cap + i;
    bool ok = false;
    V3 pc{0, 0, 0};
    float u0x = 0.f, u0y = 0.f;
    if (valid) {
      u0x = pt2d_undistort[2 * k];
      u0y = pt2d_undistort[2 * k + 1];
      if (status[k] == 1) {
        const float src[2] = {matched[2 * k], matched[2 * k + 1]};
        float u1[2];
        undistort_point(src, cam.K1, cam.D1, cam.R1, cam.P1, u1);
        pc = triangulate_dlt((double)u0x, (double)u0y, (double)u1[0], (double)u1[1], cam.P0, cam.P1);
        ok = !(pc.z < 0 || pc.z > (double)range);
      }
    }
    int nfail;
    const int frank = block_rank<SD_T / 64>(valid && !ok, s_cnt, nfail);
    if (t == 0)
      for (int q = 0; q < nfail; q++) s_rnd[q] = (float)(0.3 + (double)((float)sd_rand_next(rs) / ((float)(2147483647 / (0.4)))));
    __syncthreads();
    if (valid) {
      if (!ok) {  // DepthCamera::pixel2camera(undistorted pixel, fx, fy, cx, cy, d_rand)
        const double depth = (double)s_rnd[frank];
        pc = V3{((double)u0x - cam.cx) * depth / cam.fx, ((double)u0y - cam.cy) * depth / cam.fy, depth};
      }
      pt3d_c[3 * k] = pc.x;
      pt3d_c[3 * k + 1] = pc.y;
      pt3d_c[3 * k + 2] = pc.z;
      mask_has_3d[k] = ok ? 1 : 0;
    }
    __syncthreads();
  }
}

void launch_stereo_depth_seeds(hipStream_t st, const flvis_sd_cam& c, const float* pt2d_plane, const float* pt3d_w, const uint8_t* has_depth,
                               const int* count, int cap, int n_sets, const double* d_T_c_w7, float* seeds) {
  SdCam cam;
  static_assert(sizeof(SdCam) == sizeof(flvis_sd_cam), "SdCam layout");
  memcpy(&cam, &c, sizeof(cam));
  hipLaunchKernelGGL(k_sd_seeds, dim3((cap + 255) / 256, n_sets), dim3(256), 0, st, cam, pt2d_plane, pt3d_w, has_depth, count, cap, d_T_c_w7, seeds);
}
void launch_stereo_depth_post(hipStream_t st, const flvis_sd_cam& c, const float* pt2d_undistort, const float* matched, const uint8_t* status,
                              const int* count, int cap, int n_sets, float range, int* rand_state35, double* pt3d_c, uint8_t* mask) {
  SdCam cam;
  memcpy(&cam, &c, sizeof(cam));
  hipLaunchKernelGGL(k_sd_post, dim3(n_sets), dim3(SD_T), 0, st, cam, pt2d_undistort, matched, status, count, cap, range, rand_state35, pt3d_c, mask);
}

}  // namespace flvis
