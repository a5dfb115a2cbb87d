// flvis_amd: kernel-argument bundle and launch prototypes of the batched front-end / local-map kernels.
#pragma once
#include "pipeline.hpp"

namespace flvis {

constexpr int NEW_MAX = 1024;  // capacity of FeatureDEM's output per stream and frame

struct Pipe {
  int S;
  CamParams cam;
  StreamState* st;          // [S]
  Landmark* lm;             // [2][S][NMAX]
  MotionState* vi;          // [S][VI_QUEUE]
  const unsigned long long* seeds;  // [S] RANSAC seeds
  double* imu_in;           // [S][IMU_MAX][7]  (t, acc, gyro) in the FLVIS IMU frame
  int* n_imu;               // [S]
  double* imu_out;          // [S][IMU_OUT_CAP][11]  F2FTracking::imu_feed's outputs per sample: (t, q_w_i wxyz, pos_w_i, vel_w_i)
  float* prev_pts;          // [S][NMAX][2]
  float* next_pts;          // [S][NMAX][2]
  uint8_t* lk_status;       // [S][NMAX]
  int* lk_count;            // [S]
  int* lk_slot;             // [S][NMAX] temporal LK: template-cache slot of each previous point (Landmark::tslot), -1: none
  long long* lk_tag;        // [S] identity of the LK launch's template image (frame id of the stream's current / last frame)
  int tc_cap;               // slots per stream of the lane's template cache (0: no cache)
  const uint32_t* tc;       // the cache itself (k_track_prepare compares the slot headers) and its slot size in dwords
  int tc_stride;
  // templates ahead of the stereo matcher (round 6, lk_kernel.hip k_lk_templates_ahead): k_track_collect leaves the survivors' pixels, their
  // count and the frame's id here and gives survivor j the cache slot j; k_depth_seeds looks the slots up for the stereo launch
  int tpl_ahead;            // 0: the stereo launch computes every template itself (rounds 4-5)
  float* tpl_pts;           // [S][NMAX][2]
  int* tpl_count;           // [S]
  long long* tpl_tag;       // [S]
  float* m1;                // [S][NMAX][2]  F-RANSAC inputs (ascending survivors)
  float* m2;
  double* tri;              // [S][NMAX][3]
  uint8_t* tri_mask;        // [S][NMAX]
  float* new_xy;            // [S][NEW_MAX][2]
  int* n_new;               // [S]
  double* exist_xy;         // [S][NMAX][2]
  int* n_exist;             // [S]
  int* act_img;             // [S]
  int* act_track;           // [S]
  int* det_mode;            // [S] 0 none, 1 detect (init), 2 redetect
  int* det_maxc;            // [S]
  int* gftt_act;            // [S] corner detection planned at frame begin (init frames; tracking frames speculatively)
  int* gftt_maxc;           // [S] its maxCorners
  int* img_slot;            // [S] image slot of the current frame
  int* img_slot_in;         // [S] image slot the NEXT frame's left image is written to (k_frame_end; known before k_frame_head runs)
  FrameOut* out;            // [S]
  double* traj;             // [S][traj_cap][9]  (t, pose7, state|kf<<4) or nullptr
  int traj_cap;
  // keyframe queue tracker -> local map (the `/vo_kf` topic of the reference): frame_end appends, the local-map worker
  // consumes at its own pace (monotonic counters, slot = counter % KFQ)
  KeyFrameDev* kfq;         // [S][KFQ]
  unsigned* kfq_tail;       // [S] keyframes produced
  unsigned* kfq_head;       // [S] keyframes consumed
  int* ba_busy;             // [S] a worker workgroup owns the stream's window
  unsigned* ba_plan;        // [BA_PLAN_SLOTS][3 + S] per local-map HIP stream: arrival counter, tag of the launch whose list is valid, count, the streams
                            // with a keyframe waiting (written by the first workgroup of a launch to arrive, k_ba_worker)
  int ba_remap;             // workgroup r of a local-map launch serves the r-th stream of that list (1) or stream r (0)
  // local map
  WindowDev* win;           // [S]
  KeyFrameDev* kfs_ring;    // [S][BA_WMAX]
  CorrectionDev* corr;      // [S]
  double* ba_scratch;       // [S][ba_scratch_stride]
  size_t ba_scratch_stride;
  int imu_factor;          // window BA: add the gyro rotation-preintegration edge between consecutive keyframes (off by default)
  double imu_sigma_a;      // accelerometer noise density [m/s^2/sqrt(Hz)] of the factor's position rows (information 1 / (sigma_a^2 dt^3 / 3)); <= 0: rotation rows only
  double imu_sigma_g;      // its gyro noise density [rad/s/sqrt(Hz)]: information = 1 / (sigma_g^2 dt)
  int ba_lds_bytes;         // dynamic LDS of a k_ba_worker workgroup: what it leaves of a CU's 160 KB is there for the tracker's waves
  int ba_drain;             // keyframes a local-map workgroup takes from its stream's queue per launch (0: until the queue is empty) ...
  int ba_backlog;           // ... and it goes on while the queue still holds this many or more: the bound the back-pressure relies on
  int ba_mfma;              // Schur complement of the window solver on the matrix cores (v_mfma_f64_16x16x4_f64) or as register tiles
  // FLVIS_PNP_TAIL=cv: the final solve of solvePnPRansac(ITERATIVE) as OpenCV runs it (DLT start + CvLevMarq on the inliers,
  // cv_solvers.hpp: find_extrinsic_iterative) in a kernel of its own behind k_ransac_pnp; workspace [S][pnp_tail_stride] doubles
  int pnp_tail_cv;
  double* pnp_tail_ws;
  size_t pnp_tail_stride;
  int kf_check;             // keyframe payloads carry a checksum the local-map worker verifies (FLVIS_KF_CHECK=1: stress test of the hand-over's fences)
  int ba_balance;           // Schur accumulate: lanes per pose pair in proportion to the landmarks the pair shares (1) or 16 each (0)
  long long* counters;      // [8]: frames, keyframes, ba_runs, track_fail frames ...
  // local-map feedback (F2FTracking::correction_feed, dead in the reference's v2)
  int* rec_id;              // [S][POSE_REC]  ID_POSE::frame_id (an int in the reference)
  double* rec_T;            // [S][POSE_REC][7]
  // device table of this frame's input image bases (entry 0: img0, entry 1: img1 or, on DEPTH_D435 rigs, the Z16 depth image
  // [S][h][w]).  Rounds 1-3 uploaded it per frame and read the images through it; since round 4 the image bases are kernel arguments
  // (img_plain, in_img1) and the table is only filled by the copying mode of the inputs (FLVIS_INPUT_ZEROCOPY=0)
  const uint8_t* const* in_tab;
  // ... and the second image's base itself (a kernel argument: it changes from frame to frame with the caller's buffer)
  const uint8_t* in_img1;
  CorrectionDev* corr_in;   // [S] correction waiting for the stream's next Tracking frame (valid flag), or nullptr
  KJoin kj;                 // joins folded into THIS launch (set by the host in front of it, cleared behind it)
};

int ba_lds_budget_max();
void launch_imu_feed(hipStream_t st, const Pipe& p);
void launch_frame_begin(hipStream_t st, const Pipe& p, const double* d_time);
// cv::solvePnPRansac on caller arrays (one workgroup per correspondence set): the loop closing's geometric check
int pnp_ransac_max_points();
constexpr int EPNP_DBG_N = 160;  // doubles per set of launch_epnp_sets's output
void launch_epnp_sets(hipStream_t st, const float* p3d, const float* p2d, const int* count, int cap, int n_sets, const double* K4, double* out);
void launch_pnp_ransac_sets(hipStream_t st, const float* p3d, const float* p2d, const int* count, int cap, int n_sets, const double* K4,
                            int iterative, const double* guess7, const unsigned long long* seeds, int max_iters, double reproj_px,
                            double conf, double* pose7, unsigned char* mask, int* n_inliers);
void launch_store_progress(hipStream_t st, long long* host_word, long long v);  // stream-ordered store into host-mapped memory
void launch_store_flag(hipStream_t st, long long* word, long long v);  // stream-ordered store of a sequence number into a device word
void launch_wait_flag(hipStream_t st, const long long* flag, int n_words, long long seq, long long* err_word);  // stream-ordered wait for an upload's sequence block
void launch_frame_head(hipStream_t st, const Pipe& p, const double* d_time, long long* host_progress, long long frame_no);  // imu_feed + frame_begin in one launch
// ... + track_prepare (only when nothing runs between the two: no local-map feedback to apply)
void launch_frame_head_prepare(hipStream_t st, const Pipe& p, const double* d_time, long long* host_progress, long long frame_no);
void launch_apply_correction(hipStream_t st, const Pipe& p);
void launch_track_prepare(hipStream_t st, const Pipe& p);
void launch_track_collect(hipStream_t st, const Pipe& p);
void launch_ransac_f(hipStream_t st, const Pipe& p, bool with_collect = false);  // (with_collect: k_track_collect's work as its prologue)
void launch_ransac_pnp(hipStream_t st, const Pipe& p);
void launch_pnp_tail_cv(hipStream_t st, const Pipe& p);
void launch_track_post(hipStream_t st, const Pipe& p);
void launch_pose_lm(hipStream_t st, const Pipe& p);
void launch_reproj_filter(hipStream_t st, const Pipe& p);
void launch_vi_correction(hipStream_t st, const Pipe& p);  // viCorrectionFromVision of the streams k_reproj_filter marked
void launch_add_new(hipStream_t st, const Pipe& p);
void launch_depth_seeds(hipStream_t st, const Pipe& p);        // stereo-LK seeds of the current landmarks (critical path)
void launch_add_new_seeds(hipStream_t st, const Pipe& p);      // the two above in one launch
void launch_depth_triangulate(hipStream_t st, const Pipe& p);  // two-view triangulation for k_depth_innovate (beside the stereo LK)
void launch_depth_innovate(hipStream_t st, const Pipe& p);
void launch_frame_end(hipStream_t st, const Pipe& p);
// local map
constexpr int BA_PLAN_SLOTS = 8;  // (= Pipeline::NBA: one list per local-map HIP stream, launches on one stream do not overlap)
void launch_ba_worker(hipStream_t st, const Pipe& p, int plan_slot, unsigned launch_tag);
hipError_t ba_kernels_init();
hipError_t track_kernels_init();
size_t ba_scratch_doubles();

}  // namespace flvis
