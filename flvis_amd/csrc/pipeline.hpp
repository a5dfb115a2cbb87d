// flvis_amd: device-resident state of the batched front-end (F2FTracking mirror) and local map (LocalMap mirror).
//
// One context tracks S independent streams.  All per-stream state lives in HBM; the host only enqueues a fixed
// sequence of kernels per frame step (no host round trip unless the caller asks for outputs), every kernel masks
// itself by the stream's state machine flags.  Reference anchors: src/frontend/f2f_tracking.cpp (state machine),
// src/processing/camera_frame.cpp / landmark.cpp (frame + landmark records), src/processing/vi_motion.cpp (IMU filter),
// src/backend/vo_localmap.cpp + poselmbag.cpp (sliding window).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "img_kernels.hpp"

namespace flvis {

constexpr int NMAX = 1024;        // landmark capacity per stream and frame slot
constexpr int IMU_MAX = 64;       // IMU samples accepted per stream between two frames
constexpr int IMU_OUT_CAP = 512;  // rows of the per-stream ring behind F2FTracking::imu_feed's outputs (flvis_get_imu_states)
constexpr int VI_QUEUE = 400;     // STATES_QUEUE_SIZE (src/processing/include/vi_motion.h:10)
constexpr int KF_MAXLM = 1024;    // landmarks per keyframe payload
constexpr int KFQ = 32;            // per-stream keyframe queue between the tracker and the local map (48 KB per entry)
constexpr int BA_WMAX = 16;       // window sizes supported by the LDS-resident solver
constexpr int BA_LMAX = 4096;     // landmarks in the window
constexpr int BA_EMAX = 8192;     // observations (edges) in the window
constexpr int POSE_REC = 1024;    // ring behind F2FTracking::pose_records (the reference keeps < 1000, f2f_tracking.cpp:334)

enum { ST_UNINIT = 0, ST_TRACKING = 1, ST_TRACKFAIL = 2 };
enum { PH_IDLE = 0, PH_TRACK = 1, PH_INIT = 2 };
enum { CAM_STEREO_RECT = 0, CAM_STEREO_UNRECT = 1, CAM_DEPTH = 2 };  // enum TYPEOFCAMERA (depth_camera.h:6-9)

// LandMarkInFrame (src/processing/include/landmark.h:8-35), AoS so that a frame-to-frame copy is one record move
struct Landmark {
  long long id;
  double p3w[3];
  double p2d[2];
  double p2u[2];
  double p3c[3];
  double first2d[2];
  double first_pose[7];  // tx ty tz qx qy qz qw
  unsigned char has3d, inlier;
  short tslot;  // slot of the LK template cache that holds this landmark's templates at p2d in its frame's left image (k_depth_seeds), -1: none
  unsigned char pad[4];
};
static_assert(sizeof(Landmark) == 21 * 8, "Landmark layout");

struct MotionState {  // MOTION_STATE (vi_motion.h:12-17)
  double pos[3], vel[3], q[4] /*w x y z*/, acc[3], gyro[3], t;
};

struct CamParams {
  int cam_type, w, h;
  double fx, fy, cx, cy;  // rectified (from P0)
  double K0[4], D0[4], K1[4], D1[4];
  double R0[9], R1[9], P0[12], P1[12];
  double T_c1_c0[7];  // pose7
  double T_i_c[7], T_c_i[7];
  float iir_ratio, range;
  double depth_scale;  // DEPTH_D435: cam_scale_factor (Z16 units per metre)
  int enable_dummy, need_equal_hist, skip_first_n;
  double vi_para[4];
  DemParams dem;
  int gftt_num;
  double gftt_ql;
  int gftt_dis;
  int window;
  unsigned long long seed;
};

struct StreamState {
  int state, phase, ok, cur;
  int skip_n, has_imu, cont_fail, tf_cnt;
  long long frameCount, lm_id_counter;
  int n_lm[2];
  long long frame_id[2];
  double frame_time[2];
  double T_c_w[2][7];
  double T_kf[7];
  int use_guess;
  double guess[7];
  int new_kf, reset_cmd;
  int n_surv, of_cnt, f_cnt, pnp_cnt;
  int orig_size, n_new;
  double reproj_err;
  int rnd_r[34], rnd_pos;
  // VIMOTION scalars
  int vi_initialized, vi_first, vi_head, vi_count;  // ring: oldest at head, count entries
  double acc_bias[3], gyro_bias[3];
  // local map
  int lm_state;  // 0 UN_INITIALIZED, 1 SLIDING_WINDOW
  int kf_pending;
  // pose_records (f2f_tracking.h:59, ID_POSE): ring in Pipe::rec_id / rec_T, oldest at rec_head
  int rec_head, rec_count;
  double dbg_T_pnp[7], dbg_T_lm[7], dbg_T_pre[7];  // pose right after PnP-RANSAC / after the pose LM of the last Tracking frame (tests)
  double kf_dq[4], kf_dt;  // gyro rotation preintegration since the last keyframe (w, x, y, z), see KeyFrameDev::imu_dq
  int feeds;  // image_feed calls seen by this stream (= row of the device-side trajectory the frame is recorded in)
  // position / velocity preintegration of the bias-corrected accelerometer samples since the last keyframe (body frame of that keyframe)
  // and the filter's body velocity (world) when that keyframe was made: inputs of the IMU factor's position rows
  double kf_dp[3], kf_dv[3], kf_va[3];
  long long imu_seen;  // IMU samples integrated so far (= rows ever written to the stream's IMU-state output ring)
  int vi_corr_due;     // k_reproj_filter: this Tracking frame's viCorrectionFromVision is still to run (k_vi_correction)
  int pad_vi;
};

struct FrameOut {  // per stream, per image_feed
  int state, new_keyframe, reset_cmd, n_landmarks;
  long long frame_id;
  double T_c_w[7];
  int of_cnt, f_cnt, pnp_cnt, pad;
  double reproj_err;
};

// KeyFrame payload (msg/KeyFrame.msg without the images): what KeyFrameMsg::pub packs (src/utils/keyframe_msg.cpp:30-124)
struct KeyFrameDev {
  long long frame_id;
  int lm_count, valid;
  double T_c_w[7];
  double stamp;  // header.stamp of the message: the frame's image time
  // (addition, not in KeyFrame.msg) gyro rotation preintegration since the previous keyframe of the chain: dq (w, x, y, z) =
  // R_body(previous)^T R_body(this) as integrated from the bias-corrected gyro samples, over imu_dt seconds
  double imu_dq[4], imu_dt;
  int imu_valid, imu_pad;
  // ... with the position part: dp = preintegrated body displacement over the same interval (body frame of the previous keyframe),
  // va = the filter's body velocity (world frame) when the previous keyframe was made
  double imu_dp[3], imu_va[3];
  long long lm_id[KF_MAXLM];
  double lm_2d[KF_MAXLM][2];
  double lm_3d[KF_MAXLM][3];
};

// CorrectionInf (msg/CorrectionInf.msg)
struct CorrectionDev {
  long long frame_id;
  int lm_count, lm_outlier_count, valid, pad;
  double T_c_w[7];
  long long lm_id[BA_LMAX];
  double lm_3d[BA_LMAX][3];
  long long lm_outlier_id[BA_EMAX];
};

// sliding-window graph of one stream (PoseLMBag + the g2o graph the callback maintains)
struct WindowDev {
  // PoseLMBag
  int newest, oldest, wp_init, initialized;
  long long pose_frame_id[BA_WMAX];
  double bag_pose[BA_WMAX][7];
  int n_lm;
  long long lm_id[BA_LMAX];
  int lm_count[BA_LMAX];
  double lm_p3d[BA_LMAX][3];  // bag's p3d_w (running mean during init / first seen)
  // graph
  double pose_est[BA_WMAX][7];
  int pose_fixed[BA_WMAX], pose_present[BA_WMAX];
  double lm_est[BA_LMAX][3];  // vertex estimate, parallel to lm_id (vertex exists iff bag entry exists)
  int n_edge;
  long long edge_next_id;
  long long e_id[BA_EMAX];
  long long e_lm[BA_EMAX];  // landmark id
  int e_pose[BA_EMAX];
  int e_lidx[BA_EMAX];  // index of e_lm in the bag arrays (kept consistent across bag compaction)
  double e_uv[BA_EMAX][2];
  // IMU rotation factor (optional): the preintegration that links ring slot j to its predecessor (j - 1 + W) % W
  double imu_dq[BA_WMAX][4], imu_dt[BA_WMAX];
  int imu_has[BA_WMAX];
  double imu_dp[BA_WMAX][3], imu_va[BA_WMAX][3];
  // keyframe queue (the `kfs` deque of vo_localmap.cpp:55): ring of the last `window` payloads, storage in Pipe::kfs_ring
  int kfs_head, kfs_size;
  int solve;        // set by the bookkeeping kernel when this keyframe triggers an optimisation
  int overflow;     // a capacity (BA_LMAX / BA_EMAX) was exceeded; the window stops accepting work
  long long ba_runs;
};

// host helpers shared by the tracker set-up and the one-call entry points (pipeline.cpp)
void pose7_from_mat44(const double* m44, double* out7, bool inverse);
void glibc_seed(unsigned s, int* r34);  // the state srand(s) leaves: the last 34 words of glibc's TYPE_3 table

}  // namespace flvis
