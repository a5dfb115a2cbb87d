// ORACLE (test infrastructure, NOT product code): declarations shared by the oracle translation units.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#pragma once
#include <cstdint>
#include <vector>

namespace ref {

struct Pt2f {
  float x, y;
};

void equalize_hist(const uint8_t* src, uint8_t* dst, int w, int h);
void cvt_bgr_to_gray(const uint8_t* src, int channels, uint8_t* dst, int w, int h);
void pyr_down(const uint8_t* src, int w, int h, uint8_t* dst);
int lk_num_levels(int w, int h, int win, int max_level);
void calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_pts,
                              float* next_pts, uint8_t* status, int n, int win, int max_level, int max_iter,
                              double eps, int use_initial_flow, float min_eig_thr);
void min_eigen_map(const uint8_t* img, int w, int h, float* eig);
int good_features_to_track(const uint8_t* img, int w, int h, int max_corners, double quality, double min_distance,
                           float* out_xy);

// src/processing/include/feature_dem.h:40-50
class FeatureDEM {
 public:
  struct Scored {
    Pt2f pt;
    float score;
  };
  FeatureDEM(int image_width, int image_height, const double f_para[6]);
  void detect(const uint8_t* img, std::vector<Pt2f>& newPts) const;
  void redetect(const uint8_t* img, const std::vector<Pt2f>& existedPts, std::vector<Pt2f>& newPts) const;
  float calHarrisR(const uint8_t* img, float ptx, float pty) const;

  int width, height, regionWidth, regionHeight, boundary_dis;
  unsigned max_region_feature_num, min_region_feature_num;
  int gftt_num;
  double gftt_ql;
  int gftt_dis;

 private:
  void fillIntoRegion(const uint8_t* img, const std::vector<Pt2f>& pts, std::vector<Scored> (&region)[16],
                      bool existed) const;
};


}  // namespace ref

#include "ref_math.hpp"
namespace ref {
int poly_real_roots(const double* a, int deg, double* roots);
void project_points(const float* p3d, int n, const SE3& T, const double K[4], const double D[4], float* out);
void undistort_points(const float* src, int n, const double K[4], const double D[4], const Mat3& R, const double P[12],
                      float* dst);
Vec3 triangulate_dlt(Vec2 pt1, Vec2 pt2, const double P1[12], const double P2[12]);
Vec3 triangulate_two_view(Vec2 pt1, Vec2 pt2, const SE3& T1, const SE3& T2, double fx, double fy, double cx, double cy);
int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters);
struct CvRNG;
bool cv_get_subset(CvRNG& rng, int count, int modelPoints, int maxAttempts, bool (*check)(const int* idx, int m, const void* ctx),
                   const void* ctx, int* idx);
int seven_point(const double x1[][2], const double x2[][2], double F[3][9]);
int find_fundamental_ransac(const float* m1, const float* m2, int n, double thr, double conf, uint64_t seed,
                            uint8_t* mask);
int p3p_grunert(const Vec3 P[3], const Vec3 f[3], Mat3 Rs[4], Vec3 ts[4]);
bool solve_spd6(const double H[36], const double b[6], double x[6]);
int solve_pnp_ransac(const float* p3d, const float* p2d, int n, double fx, double fy, double cx, double cy,
                     bool iterative_flag, int iterations, double reproj_err, double conf, uint64_t seed, SE3& T,
                     uint8_t* mask);
bool optimize_in_frame(SE3& T_c_w, const Vec3* lm_3d_w, const Vec2* lm_2d, const int64_t* lm_id, int n, double fx,
                       double fy, double cx, double cy);
}  // namespace ref

#include <deque>
#include <map>
namespace ref {

// ---- wire structs (msg/KeyFrame.msg, msg/CorrectionInf.msg; src/utils/include/keyframe_msg.h, correction_inf_msg.h)
struct KeyFrameStruct {
  int64_t frame_id = 0;
  int lm_count = 0;
  std::vector<int64_t> lm_id;
  std::vector<Vec2> lm_2d;
  std::vector<Vec3> lm_3d;
  SE3 T_c_w = se3_identity();
  // optional (no field of msg/KeyFrame.msg): gyro preintegration since the previous keyframe -- the relative rotation of the
  // IMU body frame R_b(prev)^T R_b(this) as a unit quaternion, the integrated time, and whether it is valid
  Quat imu_dq = quat_identity();
  double imu_dt = 0;
  bool imu_valid = false;
  Vec3 imu_dp{0, 0, 0}, imu_va{0, 0, 0};  // preintegrated displacement (body frame of the previous keyframe), its body velocity (world)
};
struct CorrectionInfStruct {
  int64_t frame_id = 0;
  SE3 T_c_w = se3_identity();
  int lm_count = 0;
  std::vector<int64_t> lm_id;
  std::vector<Vec3> lm_3d;
  int lm_outlier_count = 0;
  std::vector<int64_t> lm_outlier_id;
};

// src/backend/include/poselmbag.h
struct LM_ITEM {
  int64_t id;
  int count;
  Vec3 p3d_w;
};
struct POSE_ITEM {
  int64_t relevent_frame_id;
  int64_t pose_id;
  SE3 pose;
};
class PoseLMBag {
 public:
  std::vector<LM_ITEM> lm_sub_bag;
  std::vector<POSE_ITEM> pose_sub_bag;
  int pose_buffer_size, newest, oldest, wp_init;
  bool pose_sub_bag_initialized;
  explicit PoseLMBag(int pose_buffer_size_in);
  void reset();
  bool hasTheLM(int64_t id_in, int& idx) const;
  bool addLMObservation(int64_t id_in, Vec3 p3d_w_in);
  bool addLMObservationSlidingWindow(int64_t id_in, Vec3 p3d_w_in);
  bool removeLMObservation(int64_t id_in);
  void addPose(int64_t id_in, const SE3& pose_in);
  int64_t getPoseIdByReleventFrameId(int64_t frame_id) const;
};

// the part of g2o::SparseOptimizer the local map uses: pose vertices (ring slots), landmark vertices, projection edges
struct BAGraph {
  struct PoseV {
    SE3 est;
    bool fixed;
    bool present;
  };
  struct Edge {
    int64_t id;
    int64_t lm;
    int pose;
    Vec2 z;
  };
  // optional IMU rotation factor between two pose slots (north_star's "IMU-preintegration factors"; the reference has no such
  // edge, vo_localmap.cpp:191-206,263-280): residual Log(dq^T R_b(a)^T R_b(b)), information w * I
  struct ImuEdge {
    int a, b;
    Quat dq;
    double w;
    // position rows (wp > 0): preintegrated body displacement dp in the body frame of a, the body velocity va (world) of a, dt
    Vec3 dp{0, 0, 0}, va{0, 0, 0};
    double dt = 0, wp = 0;
  };
  std::vector<ImuEdge> imu_edges;
  Quat q_c_b = quat_identity();  // rotation IMU body -> camera (T_c_i)
  Vec3 t_c_b{0, 0, 0};           // ... and its translation: the body origin in the camera frame
  double K[4];
  std::vector<PoseV> poses;
  std::map<int64_t, Vec3> lms;
  std::map<int64_t, Edge> edges;  // by edge id
  void optimize(int iterations);
  void remove_pose(int slot);
  void remove_lm(int64_t id);
};

void imu_edge_linearize(const SE3& Ta, const SE3& Tb, Quat q_c_b, Quat dq, double r[3], double Ja[3][3], double Jb[3][3]);
void imu_edge_linearize_pos(const SE3& Ta, const SE3& Tb, Quat q_c_b, Vec3 t_c_b, Vec3 dp, Vec3 va, double dt, double r[3], double Ja[3][6],
                            double Jb[3][6]);
class LocalMap {
 public:
  enum State { UN_INITIALIZED, SLIDING_WINDOW, OPTIMIZING, FAIL };
  LocalMap(int window, double fx, double fy, double cx, double cy);
  bool frame_callback(const KeyFrameStruct& kf, CorrectionInfStruct& out);
  void reset();
  PoseLMBag bag;
  BAGraph graph;
  std::deque<KeyFrameStruct> kfs;
  int window_size;
  State state;
  int64_t edge_id;
  // IMU factor (off by default): per pose slot the preintegrated rotation from the chronologically previous keyframe
  bool imu_factor = false;
  double imu_sigma_g = 0.002;  // gyro noise density [rad/s/sqrt(Hz)]: information 1 / (sigma^2 * dt)
  std::vector<Quat> slot_dq;
  std::vector<double> slot_dt;
  std::vector<char> slot_has;
  std::vector<Vec3> slot_dp, slot_va;
  double imu_sigma_a = 0;  // accelerometer noise density [m/s^2/sqrt(Hz)]; <= 0: rotation rows only
  void set_imu_factor(bool on, double sigma_g, Quat q_c_b);
  void set_imu_factor_pos(double sigma_a, Vec3 t_c_b);
  void rebuild_imu_edges();
};

}  // namespace ref
