// ORACLE (test infrastructure, NOT product code): native driver of the CPU restatement for bench.py's cpu_baseline leg.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Runs T independent streams through the restated front-end (F2FTracking::image_feed, src/frontend/f2f_tracking.cpp:59-400)
// + local map (LocalMapNodeletClass::frame_callback, src/backend/vo_localmap.cpp:87-380) on n_threads host threads, one
// stream per thread at a time -- the reference's own threading is one tracking worker + one local-map callback per stream
// (vo_tracking.cpp:320-321, vo_localmap.cpp:384), so "one thread per stream, as many streams as cores" is its fair
// throughput configuration.  The start-up frames [0, first) carry IMU samples only and are not timed.
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "ref_tracking.hpp"

namespace {
struct StreamRun {
  ref::F2FTracking* trk = nullptr;
  ref::LocalMap* lmap = nullptr;
};
}  // namespace

extern "C" {

// img0_frames / img1_frames: n_frames pointers, each to a block [T][h][w] (u8; depth rigs: img1 blocks are [T][h][w] u16).
// imu: [first + n_frames][T][spf][7] (t, acc xyz, gyro xyz in the FLVIS IMU frame), imu_cnt: [first + n_frames][T].
// Outputs (optional): poses [T][n_frames][7], states [T][n_frames], frame_ms [T][n_frames].  Returns the wall-clock seconds of
// the timed part (all threads released together after every stream has consumed its untimed prefix), < 0 on bad arguments.
double ref_run_streams(const ref::Config* cfg, int T, int n_frames, int first, double frame_hz, const uint8_t* const* img0_frames,
                       const uint8_t* const* img1_frames, const double* imu, const int* imu_cnt, int spf, const uint64_t* seeds,
                       int with_local_map, int n_threads, double* poses, int* states, double* frame_ms) {
  if (!cfg || T <= 0 || n_frames <= 0 || first < 0 || !img0_frames || !img1_frames || !imu || !imu_cnt || n_threads <= 0) return -1.0;
  const size_t img_px = (size_t)cfg->image_width * cfg->image_height;
  const size_t img1_bytes = img_px * (cfg->cam_type == ref::DEPTH_D435 ? 2 : 1);
  std::vector<StreamRun> runs(T);
  std::vector<uint8_t> blank(img_px * 2, 0);
  auto feed_imu = [&](StreamRun& r, int f, int s) {
    const int n = imu_cnt[(size_t)f * T + s];
    const double* rows = imu + (((size_t)f * T + s) * spf) * 7;
    ref::Quat q;
    ref::Vec3 p, v;
    for (int k = 0; k < n; k++)
      r.trk->imu_feed(rows[7 * k], {rows[7 * k + 1], rows[7 * k + 2], rows[7 * k + 3]},
                      {rows[7 * k + 4], rows[7 * k + 5], rows[7 * k + 6]}, q, p, v);
  };
  for (int s = 0; s < T; s++) {  // untimed: construction and the skipped start-up frames
    runs[s].trk = new ref::F2FTracking(*cfg, seeds ? seeds[s] : 0xF1715ull + s);
    runs[s].lmap = new ref::LocalMap(cfg->window_size, cfg->P0[0], cfg->P0[5], cfg->P0[2], cfg->P0[6]);
    for (int f = 0; f < first; f++) {
      feed_imu(runs[s], f, s);
      bool kf = false, rst = false;
      runs[s].trk->image_feed(f / frame_hz, blank.data(), blank.data(), kf, rst);
    }
  }
  std::atomic<int> next_stream{0};
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  auto worker = [&]() {
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
    for (;;) {
      const int s = next_stream.fetch_add(1);
      if (s >= T) break;
      StreamRun& r = runs[s];
      for (int j = 0; j < n_frames; j++) {
        const int f = first + j;
        const auto t0 = std::chrono::steady_clock::now();
        feed_imu(r, f, s);
        bool kf = false, rst = false;
        r.trk->image_feed(f / frame_hz, img0_frames[j] + (size_t)s * img_px, img1_frames[j] + (size_t)s * img1_bytes, kf, rst);
        if (kf && with_local_map) {
          ref::KeyFrameStruct k;
          r.trk->getKeyFrameInf(k);
          ref::CorrectionInfStruct c;
          r.lmap->frame_callback(k, c);
        }
        const size_t o = (size_t)s * n_frames + j;
        if (poses) {
          const ref::SE3& Tc = r.trk->curr_frame->T_c_w;
          const double p7[7] = {Tc.t.x, Tc.t.y, Tc.t.z, Tc.q.x, Tc.q.y, Tc.q.z, Tc.q.w};
          memcpy(poses + o * 7, p7, sizeof(p7));
        }
        if (states) states[o] = r.trk->vo_tracking_state;
        if (frame_ms) frame_ms[o] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
    }
  };
  const int nt = n_threads < T ? n_threads : T;
  std::vector<std::thread> ths;
  for (int i = 0; i < nt; i++) ths.emplace_back(worker);
  while (ready.load() < nt) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& t : ths) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto& r : runs) {
    delete r.trk;
    delete r.lmap;
  }
  return secs;
}

int ref_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
