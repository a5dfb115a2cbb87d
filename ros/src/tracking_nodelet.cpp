// flvis/TrackingNodeletClass on the MI355X: the ROS surface of the reference's tracking nodelet
// (src/frontend/vo_tracking.cpp:41-488) around libflvis_hip.so.
//   in : /vo/input_image_0, /vo/input_image_1 (sensor_msgs/Image, ExactTime, queue 2; mono8 or bgr8 / bgra8), /imu (sensor_msgs/Imu)
//        /vo_localmap_feedback (flvis/CorrectionInf: ignored unless ~use_localmap_feedback is set -- the reference unpacks it and
//        drops it, vo_tracking.cpp:373-385)
//   out: /vo_kf (flvis/KeyFrame, as KeyFrameMsg::pub fills it), /vo_camera_pose (geometry_msgs/PoseStamped, T_w_c),
//        /imu_pose (geometry_msgs/PoseStamped), /imu_odom (nav_msgs/Odometry), /imu_path (nav_msgs/Path): F2FTracking::imu_feed's
//        q_w_i / pos_w_i / vel_w_i per IMU sample, as imu_callback publishes them (vo_tracking.cpp:362-369); /imu_pose is the topic
//        the reference's EuRoC launch file records as its estimated trajectory (launch/flvis_euroc_mav.launch:83-103)
//   param: /yamlconfigfile (the reference's yaml files unchanged)
// Compile-gated by ros/CMakeLists.txt (needs catkin + the reference's `flvis` message package); never compiled in this repository's
// build image, which has no ROS.
#include <geometry_msgs/PoseStamped.h>
#include <message_filters/subscriber.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <message_filters/sync_policies/exact_time.h>
#include <message_filters/synchronizer.h>
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/Imu.h>

#include <mutex>
#include <string>
#include <vector>

#include <flvis/CorrectionInf.h>
#include <flvis/KeyFrame.h>

#include "flvis_hip.h"

namespace flvis_hip {

class TrackingNodelet : public nodelet::Nodelet {
 public:
  ~TrackingNodelet() override {
    if (ctx_) flvis_hip_destroy(ctx_);
  }

 private:
  typedef message_filters::sync_policies::ExactTime<sensor_msgs::Image, sensor_msgs::Image> SyncPolicy;

  void onInit() override {
    ros::NodeHandle& nh = getMTNodeHandle();
    std::string yaml;
    nh.getParam("/yamlconfigfile", yaml);
    getPrivateNodeHandle().param("use_localmap_feedback", use_feedback_, false);
    char err[256] = {0};
    if (flvis_config_load(yaml.c_str(), &cfg_, err, sizeof(err)) != FLVIS_OK) {
      NODELET_FATAL("flvis_config_load(%s): %s", yaml.c_str(), err);
      return;
    }
    if (flvis_hip_create(0, FLVIS_STREAM_NEW, &ctx_) != FLVIS_OK || flvis_tracker_create(ctx_, &cfg_, 1, 0xF1715, 0) != FLVIS_OK) {
      NODELET_FATAL("cannot create the HIP tracker: %s", ctx_ ? flvis_last_error(ctx_) : "no MI355X visible (no CPU fallback exists)");
      return;
    }
    kf_pub_ = nh.advertise<flvis::KeyFrame>("/vo_kf", 1);
    pose_pub_ = nh.advertise<geometry_msgs::PoseStamped>("/vo_camera_pose", 10);
    imu_pose_pub_ = nh.advertise<geometry_msgs::PoseStamped>("/imu_pose", 10);
    imu_odom_pub_ = nh.advertise<nav_msgs::Odometry>("/imu_odom", 10);
    imu_path_pub_ = nh.advertise<nav_msgs::Path>("/imu_path", 10);
    imu_path_.header.frame_id = "map";
    img0_sub_.subscribe(nh, "/vo/input_image_0", 3);
    img1_sub_.subscribe(nh, "/vo/input_image_1", 3);
    sync_.reset(new message_filters::Synchronizer<SyncPolicy>(SyncPolicy(2), img0_sub_, img1_sub_));
    sync_->registerCallback(boost::bind(&TrackingNodelet::imageCallback, this, _1, _2));
    imu_sub_ = nh.subscribe<sensor_msgs::Imu>("/imu", 10, &TrackingNodelet::imuCallback, this);
    corr_sub_ = nh.subscribe<flvis::CorrectionInf>("/vo_localmap_feedback", 2, &TrackingNodelet::correctionCallback, this);
    ids_.resize(1024);
    p2_.resize(2 * 1024);
    p3_.resize(3 * 1024);
  }

  // sensor-frame sample; the library applies the axis remap of imu_callback (vo_tracking.cpp:331-357) for the rig's imu type
  void imuCallback(const sensor_msgs::ImuConstPtr& m) {
    if (cfg_.imu_type == 3) return;  // KITTI rig: no IMU
    const double a[3] = {m->linear_acceleration.x, m->linear_acceleration.y, m->linear_acceleration.z};
    const double g[3] = {m->angular_velocity.x, m->angular_velocity.y, m->angular_velocity.z};
    double q[4], p[3], v[3];  // q_w_i (w, x, y, z), pos_w_i, vel_w_i: the outputs of F2FTracking::imu_feed (f2f_tracking.cpp:46-57)
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (flvis_imu_feed_out(ctx_, 0, m->header.stamp.toSec(), a, g, q, p, v) != FLVIS_OK) {
        NODELET_WARN_THROTTLE(1.0, "imu_feed: %s", flvis_last_error(ctx_));
        return;
      }
    }
    // pose_imu_pub->pubPose, odom_imu_pub->pubOdom, imu_path_pub->pubPathT_w_c (vo_tracking.cpp:367-369)
    geometry_msgs::PoseStamped ps;
    ps.header.stamp = m->header.stamp;
    ps.header.frame_id = "map";
    ps.pose.position.x = p[0];
    ps.pose.position.y = p[1];
    ps.pose.position.z = p[2];
    ps.pose.orientation.w = q[0];
    ps.pose.orientation.x = q[1];
    ps.pose.orientation.y = q[2];
    ps.pose.orientation.z = q[3];
    imu_pose_pub_.publish(ps);
    nav_msgs::Odometry od;
    od.header = ps.header;
    od.pose.pose = ps.pose;
    od.twist.twist.linear.x = v[0];
    od.twist.twist.linear.y = v[1];
    od.twist.twist.linear.z = v[2];
    imu_odom_pub_.publish(od);
    imu_path_.header.stamp = m->header.stamp;
    imu_path_.poses.push_back(ps);
    if (imu_path_.poses.size() >= 400) imu_path_.poses.erase(imu_path_.poses.begin());  // RVIZPath(nh, "/imu_path", "map", 1, 400), rviz_path.cpp:43-46
    imu_path_pub_.publish(imu_path_);
  }

  void correctionCallback(const flvis::CorrectionInfConstPtr& c) {
    if (!use_feedback_) return;  // v2 behaviour of the reference: the message is received and dropped
    std::vector<double> lm3(3 * c->lm_3d_data.size());
    for (size_t i = 0; i < c->lm_3d_data.size(); i++) {
      lm3[3 * i] = c->lm_3d_data[i].x;
      lm3[3 * i + 1] = c->lm_3d_data[i].y;
      lm3[3 * i + 2] = c->lm_3d_data[i].z;
    }
    const double T[7] = {c->T_c_w.translation.x, c->T_c_w.translation.y, c->T_c_w.translation.z, c->T_c_w.rotation.x,
                         c->T_c_w.rotation.y,    c->T_c_w.rotation.z,    c->T_c_w.rotation.w};
    std::lock_guard<std::mutex> lk(mu_);
    flvis_correction_feed(ctx_, 0, c->frame_id, T, c->lm_count, c->lm_id_data.data.data(), lm3.data(), c->lm_outlier_count,
                          c->lm_outlier_id_data.data.data());
  }

  static int channelsOf(const std::string& enc) { return enc == "bgr8" || enc == "rgb8" ? 3 : (enc == "bgra8" || enc == "rgba8" ? 4 : 1); }

  void imageCallback(const sensor_msgs::ImageConstPtr& m0, const sensor_msgs::ImageConstPtr& m1) {
    const double t = m0->header.stamp.toSec();
    const flvis_image a = {m0->data.data(), (int)m0->width, (int)m0->height, (int)m0->step, channelsOf(m0->encoding), t};
    const flvis_image b = {m1->data.data(), (int)m1->width, (int)m1->height, (int)m1->step, channelsOf(m1->encoding), t};  // 16UC1 on depth rigs
    flvis_frame_out out;
    flvis::KeyFrame kf;
    bool have_kf = false;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (flvis_image_feed_host(ctx_, &a, &b, &out, /*with_local_map=*/0, /*hold_buffers=*/0) != FLVIS_OK) {
        NODELET_ERROR_THROTTLE(1.0, "image_feed: %s", flvis_last_error(ctx_));
        return;
      }
      if (out.new_keyframe) {
        // KeyFrameMsg::pub (src/utils/keyframe_msg.cpp:30-124): the images the tracker worked on, the (id, undistorted pixel,
        // world point) triples of the landmarks with depth and inlier flag, the pose
        const size_t px = (size_t)cfg_.image_width * cfg_.image_height;
        kf.img0.data.resize(px);
        kf.img1.data.resize(cfg_.cam_type == 2 ? 2 * px : px);
        flvis_keyframe k;
        const int n = flvis_get_keyframe_msg(ctx_, 0, 1024, &k, ids_.data(), p2_.data(), p3_.data(), kf.img0.data.data(), kf.img1.data.data());
        if (n > 0) {
          have_kf = true;
          kf.header.stamp = m0->header.stamp;
          kf.frame_id = k.frame_id;
          kf.command = k.command;
          kf.img0.height = kf.img1.height = cfg_.image_height;
          kf.img0.width = kf.img1.width = cfg_.image_width;
          kf.img0.encoding = "mono8";
          kf.img0.step = cfg_.image_width;
          kf.img1.encoding = cfg_.cam_type == 2 ? "16UC1" : "mono8";
          kf.img1.step = cfg_.cam_type == 2 ? 2 * cfg_.image_width : cfg_.image_width;
          kf.lm_count = n;
          kf.lm_id_data.layout.dim.resize(1);
          kf.lm_id_data.layout.dim[0].label = "lm_id";
          kf.lm_id_data.layout.dim[0].size = kf.lm_id_data.layout.dim[0].stride = n;
          kf.lm_id_data.data.assign(ids_.begin(), ids_.begin() + n);
          kf.lm_2d_data.resize(n);
          kf.lm_3d_data.resize(n);
          for (int i = 0; i < n; i++) {
            kf.lm_2d_data[i].x = p2_[2 * i];
            kf.lm_2d_data[i].y = p2_[2 * i + 1];
            kf.lm_2d_data[i].z = 0;
            kf.lm_3d_data[i].x = p3_[3 * i];
            kf.lm_3d_data[i].y = p3_[3 * i + 1];
            kf.lm_3d_data[i].z = p3_[3 * i + 2];
          }
          kf.T_c_w.translation.x = k.T_c_w[0];
          kf.T_c_w.translation.y = k.T_c_w[1];
          kf.T_c_w.translation.z = k.T_c_w[2];
          kf.T_c_w.rotation.x = k.T_c_w[3];
          kf.T_c_w.rotation.y = k.T_c_w[4];
          kf.T_c_w.rotation.z = k.T_c_w[5];
          kf.T_c_w.rotation.w = k.T_c_w[6];
        }
      }
    }
    if (have_kf) kf_pub_.publish(kf);
    if (out.state == 1) {  // T_w_c = T_c_w^-1, what /vo_camera_pose carries
      const double qx = out.T_c_w[3], qy = out.T_c_w[4], qz = out.T_c_w[5], qw = out.T_c_w[6], tx = out.T_c_w[0], ty = out.T_c_w[1], tz = out.T_c_w[2];
      const double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)},
                              {2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)},
                              {2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)}};
      geometry_msgs::PoseStamped p;
      p.header.stamp = m0->header.stamp;
      p.header.frame_id = "map";
      p.pose.position.x = -(R[0][0] * tx + R[1][0] * ty + R[2][0] * tz);
      p.pose.position.y = -(R[0][1] * tx + R[1][1] * ty + R[2][1] * tz);
      p.pose.position.z = -(R[0][2] * tx + R[1][2] * ty + R[2][2] * tz);
      p.pose.orientation.w = qw;
      p.pose.orientation.x = -qx;
      p.pose.orientation.y = -qy;
      p.pose.orientation.z = -qz;
      pose_pub_.publish(p);
    }
  }

  flvis_ctx* ctx_ = nullptr;
  flvis_cfg cfg_;
  bool use_feedback_ = false;
  std::mutex mu_;  // flvis_imu_feed may arrive concurrently with flvis_image_feed_host (MT nodelet handle)
  ros::Publisher kf_pub_, pose_pub_, imu_pose_pub_, imu_odom_pub_, imu_path_pub_;
  nav_msgs::Path imu_path_;
  ros::Subscriber imu_sub_, corr_sub_;
  message_filters::Subscriber<sensor_msgs::Image> img0_sub_, img1_sub_;
  std::shared_ptr<message_filters::Synchronizer<SyncPolicy>> sync_;
  std::vector<int64_t> ids_;
  std::vector<double> p2_, p3_;
};

}  // namespace flvis_hip

PLUGINLIB_EXPORT_CLASS(flvis_hip::TrackingNodelet, nodelet::Nodelet)
