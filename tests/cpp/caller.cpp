// A C++ caller of the C ABI (include/flvis_hip.h), the way the reference's nodelet would bind it: no Python, no torch, no HIP
// headers -- host buffers in, host structs out.  Built and run by tests/test_cpp_caller.py (g++ / hipcc, links libflvis_hip.so).
//
//   caller <config.yaml> [frames]
//
// Feeds a static synthetic stereo pair (a noise texture, the right image shifted by a constant disparity = a fronto-parallel
// wall) and IMU samples of a rig at rest through flvis_imu_feed / flvis_image_feed_host for a D435i-stereo configuration and
// checks what the reference's process() loop would consume: the tracking state, the KeyFrame message of the first frame, the
// landmark depths; then the loop-closing nodelet's calls (vocabulary file, keyframes in, events / similarity row / poses out).  Exit codes: 0 ok, 3 no GPU (flvis_hip_create refused: the library has no CPU fallback), 1 failure.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "flvis_hip.h"

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ < 0) {                                                                                   \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ctx ? flvis_last_error(ctx) : "-"); \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: caller <config.yaml> [frames]\n");
    return 1;
  }
  const int extra_frames = argc > 2 ? std::atoi(argv[2]) : 6;
  flvis_ctx* ctx = nullptr;
  flvis_cfg cfg;
  char err[256] = {0};
  if (flvis_config_load(argv[1], &cfg, err, sizeof(err)) != FLVIS_OK) {
    std::fprintf(stderr, "config: %s\n", err);
    return 1;
  }
  int rc = flvis_hip_create(0, FLVIS_STREAM_NEW, &ctx);
  if (rc == FLVIS_ERR_NO_DEVICE) {
    std::printf("no device: %s\n", flvis_version());
    return 3;
  }
  if (rc != FLVIS_OK) return 1;
  const int w = cfg.image_width, h = cfg.image_height, n_frames = cfg.skip_first_n_imgs + extra_frames;
  CHECK(flvis_tracker_create(ctx, &cfg, 1, 0xF1715, n_frames));
  // a band-limited noise texture: the left image; the right one sees it shifted by `disp` pixels (wall at fx * b / disp)
  const int disp = 8;
  std::vector<uint8_t> tex((size_t)(w + 64) * h), img0((size_t)w * h), img1((size_t)w * h);
  uint32_t lcg = 12345u;
  std::vector<float> nz((size_t)(w + 64 + 8) * (h + 8));
  for (auto& v : nz) {
    lcg = lcg * 1664525u + 1013904223u;
    v = (float)(lcg >> 24);
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w + 64; x++) {  // 5 x 5 box blur of the noise: corners that LK can track
      float s = 0;
      for (int dy = 0; dy < 5; dy++)
        for (int dx = 0; dx < 5; dx++) s += nz[(size_t)(y + dy) * (w + 64 + 8) + x + dx];
      float v = (s / 25.f - 128.f) * 3.f + 128.f;
      tex[(size_t)y * (w + 64) + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      img0[(size_t)y * w + x] = tex[(size_t)y * (w + 64) + x + 32];
      img1[(size_t)y * w + x] = tex[(size_t)y * (w + 64) + x + 32 + disp];  // a point moves LEFT by disp in the right camera
    }
  flvis_frame_out out;
  int kf_count = 0, tracked = 0;
  double t_imu = 0.0;
  flvis_keyframe kf;
  std::vector<int64_t> ids(1024);
  std::vector<double> p2(2048), p3(3072);
  std::vector<uint8_t> kimg0((size_t)w * h), kimg1((size_t)w * h);
  for (int f = 0; f < n_frames; f++) {
    const double t = f * 0.05;
    for (; t_imu <= t + 1e-9; t_imu += 0.005) {  // 200 Hz, rig at rest: sensor-frame acceleration of a D435i lying level
      const double acc[3] = {0.0, -9.81, 0.0}, gyro[3] = {0.0, 0.0, 0.0};
      CHECK(flvis_imu_feed(ctx, 0, t_imu, acc, gyro));
    }
    const flvis_image a = {img0.data(), w, h, w, 1, t}, b = {img1.data(), w, h, w, 1, t};
    CHECK(flvis_image_feed_host(ctx, &a, &b, &out, /*with_local_map=*/1, /*hold_buffers=*/0));
    if (f < cfg.skip_first_n_imgs) {
      if (out.state != 0 || out.n_landmarks != 0) {
        std::fprintf(stderr, "frame %d: a skipped start-up frame was processed\n", f);
        return 1;
      }
      continue;
    }
    tracked += out.state == 1;
    if (out.new_keyframe) {
      const int n = flvis_get_keyframe_msg(ctx, 0, 1024, &kf, ids.data(), p2.data(), p3.data(), kimg0.data(), kimg1.data());
      if (n <= 30 || kf.lm_count != n || kf.command != 0 || std::fabs(kf.stamp - t) > 1e-12 || kf.frame_id != out.frame_id) {
        std::fprintf(stderr, "frame %d: bad KeyFrame message (n=%d stamp=%g frame_id=%lld)\n", f, n, kf.stamp, (long long)kf.frame_id);
        return 1;
      }
      if (kimg0 != img0 || kimg1 != img1) {
        std::fprintf(stderr, "frame %d: the KeyFrame images are not the images that were fed\n", f);
        return 1;
      }
      kf_count++;
    }
  }
  // the wall is at fx * baseline / disp in front of the camera: every landmark's depth along the optical axis must say so
  std::vector<double> l2d(2048), l2u(2048), l3(3072);
  std::vector<uint8_t> fl(1024);
  const int nl = flvis_get_landmarks(ctx, 0, 1024, ids.data(), l2d.data(), l2u.data(), l3.data(), fl.data());
  if (nl < 30) {
    std::fprintf(stderr, "only %d landmarks\n", nl);
    return 1;
  }
  const double want_z = cfg.P0[0] * std::fabs(cfg.T_cam0_cam1[3]) / disp;
  int good = 0;
  for (int i = 0; i < nl; i++) {
    // world frame of the init pose: x forward (f2f_tracking.cpp:153-161): the wall is at x = want_z
    if ((fl[i] & 1) && std::fabs(l3[3 * i] - want_z) < 0.05 * want_z) good++;
  }
  int64_t counters[3];
  CHECK(flvis_get_counters(ctx, counters));
  std::printf("%s: frames %lld tracked %d keyframes %d landmarks %d (depth ok %d, wall at %.2f m) ba_runs %lld lanes %d\n", flvis_version(),
              (long long)counters[0], tracked, kf_count, nl, good, want_z, (long long)counters[2], flvis_tracker_lanes(ctx));
  // ---- the loop-closing nodelet's calls (vo_loopclosing.cpp onInit / kfmsgProcess / pgoProcess) from the same caller -----------------
  // a vocabulary FILE as `Vocabulary voc(path)` reads it: DBoW3's plain binary layout, 64 words directly under the root
  const char* voc_path = "/tmp/flvis_cpp_caller_voc.dbow3";
  {
    std::FILE* f = std::fopen(voc_path, "wb");
    if (!f) return 1;
    const uint64_t magic = 88877711233ull;
    const uint8_t compressed = 0;
    const uint32_t n_nodes = 65, n_words = 64;
    const int32_t head[4] = {64, 1, 0, 0};  // k, L, scoring L1_NORM, weighting TF_IDF
    std::fwrite(&magic, 8, 1, f);
    std::fwrite(&compressed, 1, 1, f);
    std::fwrite(&n_nodes, 4, 1, f);
    std::fwrite(head, 4, 4, f);
    for (uint32_t i = 1; i < n_nodes; i++) {
      const uint32_t parent = 0;
      const double weight = 1.0 + 0.01 * i;
      const int32_t mat[3] = {32, 1, 0};  // cols, rows, CV_8U
      uint8_t d[32];
      for (int b = 0; b < 32; b++) {
        lcg = lcg * 1664525u + 1013904223u;
        d[b] = (uint8_t)(lcg >> 24);
      }
      std::fwrite(&i, 4, 1, f);
      std::fwrite(&parent, 4, 1, f);
      std::fwrite(&weight, 8, 1, f);
      std::fwrite(mat, 4, 3, f);
      std::fwrite(d, 1, 32, f);
    }
    std::fwrite(&n_words, 4, 1, f);
    for (uint32_t wid = 0; wid < n_words; wid++) {
      const uint32_t nid = wid + 1;
      std::fwrite(&wid, 4, 1, f);
      std::fwrite(&nid, 4, 1, f);
    }
    std::fclose(f);
  }
  CHECK(flvis_hip_bow_load_vocabulary(ctx, voc_path));
  if (flvis_hip_bow_load_vocabulary(ctx, "/tmp/flvis_no_such_vocabulary.dbow3") != FLVIS_ERR_CONFIG) {
    std::fprintf(stderr, "a missing vocabulary file must be refused\n");
    return 1;
  }
  const flvis_lc_params lcp = {25, 18, 50, 20, 2, 20, 0.5, 0.5, 0.12};  // launch/KITTI/KITTI.yaml:110-127
  flvis_loop_closer* lc = nullptr;
  CHECK(flvis_loop_closer_create(ctx, &cfg, &lcp, 1, 8, nullptr, &lc));
  const int stream0 = 0;
  flvis_lc_event ev;
  std::vector<double> row(8), poses(7 * 8);
  int n_row = 0, n_pose = 0, lm_kept = 0, bow_n = 0;
  for (int k = 0; k < 3; k++) {
    const flvis_image a = {img0.data(), w, h, w, 1, 0.0}, b = {img1.data(), w, h, w, 1, 0.0};
    const double T[7] = {0.1 * k, 0, 0, 0, 0, 0, 1};
    int64_t kf_id = -1;
    CHECK(flvis_loop_closer_add_keyframes_host(lc, 1, &stream0, &a, &b, T, &kf_id));
    CHECK(flvis_loop_closer_process(lc, &ev));
    if (kf_id != k || ev.kf_curr != k || ev.candidate || ev.loop_accepted) {  // fewer than 50 keyframes: nothing to close yet
      std::fprintf(stderr, "loop closer: keyframe %d came back as %lld / event kf_curr %lld\n", k, (long long)kf_id, (long long)ev.kf_curr);
      return 1;
    }
  }
  CHECK(flvis_loop_closer_similarity_row(lc, 0, row.data(), 8, &n_row));
  CHECK(flvis_loop_closer_poses(lc, 0, poses.data(), 8, &n_pose));
  CHECK(flvis_loop_closer_keyframe(lc, 0, 2, 0, nullptr, nullptr, nullptr, &lm_kept, nullptr, nullptr, &bow_n));
  bool lc_ok = n_row == 3 && n_pose == 3 && lm_kept > 30 && bow_n > 5 && std::fabs(poses[14] - 0.2) < 1e-15 && poses[20] == 1.0;
  for (int j = 0; j < 3 && lc_ok; j++) lc_ok = std::fabs(row[j] - 1.0) < 1e-12;  // three times the same image: every score is 1
  std::printf("loop closer: %d keyframes, %d landmarks kept, %d words, scores %.15g %.15g %.15g\n", n_pose, lm_kept, bow_n, row[0], row[1], row[2]);
  flvis_loop_closer_destroy(lc);
  flvis_hip_destroy(ctx);
  // (points whose stereo match fails get the reference's rand() dummy depth, camera_frame.cpp:153-168: not all sit on the wall)
  if (!lc_ok) return 1;
  if (tracked != extra_frames || kf_count < 1 || good < nl / 2) return 1;
  std::printf("caller OK\n");
  return 0;
}
