// ORACLE (test infrastructure, NOT product code): front-end state machine declarations.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#pragma once
#include <deque>
#include <vector>

#include "ref_api.h"

namespace ref {

enum TYPEOFCAMERA { STEREO_RECT = 0, STEREO_UNRECT = 1, DEPTH_D435 = 2 };  // src/processing/include/depth_camera.h:6-9

// resolved configuration (what TrackingNodeletClass::onInit derives from the yaml, src/frontend/vo_tracking.cpp:114-306)
struct Config {
  int type_of_vi;
  int image_width, image_height;
  double cam0_intrinsics[4], cam0_distortion[4], cam1_intrinsics[4], cam1_distortion[4];
  double T_imu_cam0[16];   // row-major 4x4 (EuRoC: T_imu_mavimu * T_mavimu_cam0)
  double T_cam0_cam1[16];  // row-major 4x4
  double vifusion_para[6], feature_para[6], dr_para[3];
  int window_size;
  // derived
  int cam_type, has_imu_type, skip_first_n_imgs, need_equal_hist;
  double R0[9], R1[9], P0[12], P1[12];
  double depth_factor;  // depth modes (type_of_vi 0, 2): Z16 units per metre (vo_tracking.cpp:153)
};
// fills cam_type/skip/equalize flags and R0,R1,P0,P1 (cv::stereoRectify restated).  false on unsupported type_of_vi.
bool config_finalize(Config& c);
// reads one of the reference's yaml files (flat key: scalar | [list], '#' comments; src/utils/include/yamlRead.h)
bool config_load_yaml(const char* path, Config& c, char* err, int errlen);

struct DepthCamera {  // src/processing/include/depth_camera.h
  int cam_type;
  int img_w, img_h;
  double cam0_fx, cam0_fy, cam0_cx, cam0_cy;
  double K0[4], D0[4], K1[4], D1[4];
  Mat3 R0, R1;
  double P0_[12], P1_[12];
  SE3 T_cam0_cam1, T_cam1_cam0;
  double cam_scale_factor;  // depth_camera.cpp:23
};

struct LandMarkInFrame {  // src/processing/include/landmark.h
  int64_t lm_id;
  Vec3 lm_3d_w;
  Vec2 lm_2d_plane, lm_2d_undistort;
  Vec3 lm_3d_c;
  bool has_3d, is_belong_to_kf, is_tracking_inlier;
  Vec2 lm_1st_obs_2d;
  SE3 lm_1st_obs_frame_pose;
};

struct CameraFrame {  // src/processing/include/camera_frame.h
  int64_t frame_id = 0;
  double frame_time = 0;
  std::vector<uint8_t> img0, img1;
  std::vector<uint16_t> d_img;  // DEPTH_D435: CV_16UC1 aligned to cam0 (f2f_tracking.cpp:116-119)
  std::vector<LandMarkInFrame> landmarks;
  SE3 T_c_w = se3_identity();
  double reprojection_error = 0;
  void clear() {
    T_c_w = se3_identity();
    img0.clear();
    img1.clear();
    landmarks.clear();
  }
};

struct GlibcRand {  // glibc random_r TYPE_3, the generator behind rand() (camera_frame.cpp:153,168; never seeded => seed 1)
  int32_t r[34];
  int pos;
  GlibcRand() { seed(1); }
  void seed(unsigned s);
  int next();
};

struct IMUSTATE {
  Vec3 acc_raw, gyro_raw;
  double timestamp;
};
struct MOTION_STATE {
  Vec3 pos, vel;
  Quat q_w_i;
  IMUSTATE imu_data;
};

class VIMOTION {  // src/processing/vi_motion.cpp
 public:
  VIMOTION(const SE3& T_i_c, double g, double p1, double p2, double p3, double p4, double p5 = 0.5, double p6 = 0.1);
  void viIMUinitialization(const IMUSTATE& imu, Quat& q, Vec3& p, Vec3& v);
  void viIMUPropagation(const IMUSTATE& imu, Quat& q, Vec3& p, Vec3& v);
  void viVisiontrigger(Quat& init_orientation);
  void viVisionRPCompensation(double time, SE3& T_c_w);
  bool viGetCorrFrameState(double time, SE3& T_c_w);
  void viCorrectionFromVision(double t_curr, const SE3& Tcw_curr, double t_last, const SE3& Tcw_last, double err);
  bool viGetIMURollPitchAtTime(double time, double& roll, double& pitch);
  bool viFindStateIdx(double time, int& idx);
  SE3 T_i_c, T_c_i;
  double para_1, para_2, para_3, para_4, magnitude_g, ba_sat, bw_sat;
  Vec3 acc_bias, gyro_bias, gravity;
  std::deque<MOTION_STATE> states;
  MOTION_STATE init_state;
  bool imu_initialized, is_first_data;
  // gyro rotation preintegration since the last keyframe (not in the reference: input of the optional IMU factor of the window BA)
  Quat kf_dq = quat_identity();
  double kf_dt = 0;
  // ... and the position / velocity preintegration of the bias-corrected accelerometer samples in the body frame of that keyframe:
  // dp += dv dt + 1/2 (dR f) dt^2, dv += (dR f) dt (before dR advances); input of the factor's position rows
  Vec3 kf_dp{0, 0, 0}, kf_dv{0, 0, 0};
};

enum TRACKINGSTATE { UnInit, Tracking, TrackingFail };

class F2FTracking {  // src/frontend/f2f_tracking.cpp
 public:
  F2FTracking(const Config& cfg, uint64_t ransac_seed);
  ~F2FTracking();
  void imu_feed(double time, Vec3 acc, Vec3 gyro, Quat& q_w_i, Vec3& pos_w_i, Vec3& vel_w_i);
  void image_feed(double time, const uint8_t* img0, const uint8_t* img1, bool& new_keyframe, bool& reset_cmd);
  void getKeyFrameInf(KeyFrameStruct& kf) const;  // CameraFrame::getKeyFrameInf + pose (what KeyFrameMsg::pub sends)
  Quat kf_imu_dq = quat_identity();  // preintegrated body rotation between the previous keyframe and the last published one
  double kf_imu_dt = 0;
  bool kf_imu_valid = false;
  Vec3 kf_imu_dp{0, 0, 0};   // preintegrated body displacement over the same interval (body frame of the previous keyframe)
  Vec3 kf_imu_va{0, 0, 0};   // the filter's body velocity (world) when the previous keyframe was made
  Vec3 kf_va_next{0, 0, 0};  // ... when the last one was made (becomes kf_imu_va of the next keyframe)
  // local-map feedback (f2f_tracking.cpp:40-44): dead in v2 (vo_tracking.cpp:373-385 unpacks the message and drops it);
  // restated for the SURVEY 8f-2 row, applied at the next Tracking frame (f2f_tracking.cpp:189-219)
  void correction_feed(const CorrectionInfStruct& corr);
  struct ID_POSE {  // f2f_tracking.h:19-22 (frame_id is an int there)
    int frame_id;
    SE3 T_c_w;
  };
  std::deque<ID_POSE> pose_records;
  CorrectionInfStruct correction_inf;
  bool has_localmap_feedback;

  Config cfg;
  DepthCamera d_camera;
  FeatureDEM* feature_dem;
  VIMOTION* vimotion;
  CameraFrame *curr_frame, *last_frame;
  CameraFrame frames[2];
  int vo_tracking_state;
  bool has_imu, need_equal_hist, enable_dummy;
  int skip_n_imgs;
  float iir_ratio, range;
  int64_t frameCount;
  SE3 T_c_w_last_keyframe;
  int continus_tracking_fail_cnt, trackingfail_cnt;
  int64_t lm_id_counter;  // landmark.cpp:3 static id_index = 100 (per process => per stream here)
  GlibcRand rnd;
  uint64_t ransac_seed;
  // per-frame diagnostics (tests): counts printed by lkorb_tracking.cpp:191
  int dbg_of_inlier, dbg_F_inlier, dbg_pnp_inlier;
  SE3 dbg_T_pnp = se3_identity(), dbg_T_lm = se3_identity(), dbg_T_pre = se3_identity();  // pose right after solvePnPRansac / after OptimizeInFrame (tests)
  // CameraFrame::recover3DPts_c_FromStereo on the arrays getAll2dPlaneUndistort3d_cvPf hands it (also called on its own by the tests)
  void recover3DPts_c_FromStereo(const uint8_t* img0, const uint8_t* img1, int n, const float* p0, const float* p0u, const float* p3,
                                 const uint8_t* has_3d, const SE3& T_c_w, float rng, Vec3* meas, uint8_t* meas_mask);

 private:
  bool init_frame();
  bool lk_tracking(CameraFrame& from, CameraFrame& to, const SE3& guess, bool use_guess);
  void depthInnovation(CameraFrame& f);
  LandMarkInFrame make_landmark(Vec2 pt2d, Vec2 pt2d_undist, const SE3& T_c_w, bool is_inlier);
};

}  // namespace ref
