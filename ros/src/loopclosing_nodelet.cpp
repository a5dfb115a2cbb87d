// flvis/LoopClosingNodeletClass on the MI355X: subscribes the KeyFrame stream of the tracker (/vo_kf), keeps the keyframe
// database on the GPU and runs the reference nodelet's work (src/backend/vo_loopclosing.cpp:118-1119) through flvis_loop_closer:
//   onInit            <- :948-1119  camera + LC_PARAS from /yamlconfigfile, vocabulary from /voc, /vo_kf subscription, worker thread
//   keyframeCallback  <- :178-184   queue
//   worker            <- kfmsgProcess (:186-391) + pgoProcess (:393-518): one keyframe in, ORB / bag of words / 3-D landmarks,
//                        similarity row, candidate, geometric check, pose graph; tf map -> odom (:213-223, :503-511) and the
//                        corrected path on /vision_path_lc_all (:382, :916-932) out
// One worker thread handles a keyframe completely before the next one (the reference's two threads race for the newest keyframe).
// The /loop_closure_img debug image (:688-729) is not produced.  Compile-gated by ros/CMakeLists.txt; never compiled in this
// repository's build image (no ROS).
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>
#include <tf/transform_broadcaster.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <flvis/KeyFrame.h>
#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/Path.h>

#include "flvis_hip.h"

namespace flvis_hip {

class LoopClosingNodelet : public nodelet::Nodelet {
 public:
  ~LoopClosingNodelet() override {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
    if (lc_) flvis_loop_closer_destroy(lc_);
    if (ctx_) flvis_hip_destroy(ctx_);
  }

 private:
  void onInit() override {
    ros::NodeHandle& nh = getNodeHandle();
    std::string yaml, voc;
    nh.getParam("/yamlconfigfile", yaml);
    nh.getParam("/voc", voc);
    int max_kf = 20000;
    getPrivateNodeHandle().param("max_keyframes", max_kf, max_kf);
    char err[256] = {0};
    flvis_lc_params prm;
    if (flvis_config_load(yaml.c_str(), &cfg_, err, sizeof(err)) != FLVIS_OK || flvis_lc_params_load(yaml.c_str(), &prm, err, sizeof(err)) != FLVIS_OK) {
      NODELET_FATAL("%s: %s", yaml.c_str(), err);
      return;
    }
    if (flvis_hip_create(0, FLVIS_STREAM_NEW, &ctx_) != FLVIS_OK) {
      NODELET_FATAL("no MI355X visible (no CPU fallback exists)");
      return;
    }
    if (flvis_hip_bow_load_vocabulary(ctx_, voc.c_str()) != FLVIS_OK ||
        flvis_loop_closer_create(ctx_, &cfg_, &prm, 1, max_kf, /*h_orb_pattern=*/nullptr, &lc_) != FLVIS_OK) {
      NODELET_FATAL("loop closer: %s", flvis_last_error(ctx_));
      return;
    }
    path_pub_ = nh.advertise<nav_msgs::Path>("/vision_path_lc_all", 1);
    kf_sub_ = nh.subscribe<flvis::KeyFrame>("/vo_kf", 1000, &LoopClosingNodelet::keyframeCallback, this);
    worker_ = std::thread(&LoopClosingNodelet::work, this);
  }

  void keyframeCallback(const flvis::KeyFrameConstPtr& kf) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      queue_.push_back(kf);
    }
    cv_.notify_one();
  }

  void work() {
    for (;;) {
      flvis::KeyFrameConstPtr kf;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (stop_) return;
        kf = queue_.front();
        queue_.pop_front();
      }
      if (kf->command != 0) continue;  // KFMSG_CMD_RESET_LM (:200)
      const flvis_image a = {kf->img0.data.data(), (int)kf->img0.width, (int)kf->img0.height, (int)kf->img0.step, 1, 0.0};
      const flvis_image b = {kf->img1.data.data(), (int)kf->img1.width, (int)kf->img1.height, (int)kf->img1.step, 1, 0.0};
      const double T[7] = {kf->T_c_w.translation.x, kf->T_c_w.translation.y, kf->T_c_w.translation.z, kf->T_c_w.rotation.x,
                           kf->T_c_w.rotation.y,    kf->T_c_w.rotation.z,    kf->T_c_w.rotation.w};
      const int stream = 0;
      int64_t id = 0;
      if (flvis_loop_closer_add_keyframes_host(lc_, 1, &stream, &a, &b, T, &id) != FLVIS_OK) {
        NODELET_ERROR_THROTTLE(1.0, "loop closer: %s", flvis_last_error(ctx_));
        continue;
      }
      stamps_.push_back(kf->header.stamp);
      flvis_lc_event ev;
      if (flvis_loop_closer_process(lc_, &ev) != FLVIS_OK) {
        NODELET_ERROR_THROTTLE(1.0, "loop closer: %s", flvis_last_error(ctx_));
        continue;
      }
      if (ev.loop_accepted)
        NODELET_INFO("loop %lld -> %lld: %d matches, %d inliers%s", (long long)ev.kf_prev, (long long)ev.kf_curr, ev.n_matches, ev.n_inliers,
                     ev.optimised ? ", pose graph optimised" : "");
      broadcastMapToOdom();
      publishPath(ev.optimised != 0);
    }
  }

  // T_map_odom = T_odom_map^-1 as tf map -> odom (:213-223)
  void broadcastMapToOdom() {
    double d[7];
    if (flvis_loop_closer_drift(lc_, 0, d) != FLVIS_OK) return;
    const tf::Quaternion q(d[3], d[4], d[5], d[6]);
    const tf::Transform T_odom_map(q, tf::Vector3(d[0], d[1], d[2]));
    br_.sendTransform(tf::StampedTransform(T_odom_map.inverse(), ros::Time::now(), "map", "odom"));
  }

  // the path of camera poses T_w_c in the map frame: one pose appended per keyframe (:382), the whole path rewritten after an
  // optimisation (:916-932)
  void publishPath(bool rewrite) {
    int n = 0;
    poses_.resize(7 * stamps_.size());
    if (flvis_loop_closer_poses(lc_, 0, poses_.data(), (int)stamps_.size(), &n) != FLVIS_OK) return;
    const size_t first = rewrite ? 0 : path_.poses.size();
    if (rewrite) path_.poses.clear();
    path_.header.frame_id = "map";
    for (size_t i = first; i < (size_t)n; i++) {
      const tf::Transform T_c_w(tf::Quaternion(poses_[7 * i + 3], poses_[7 * i + 4], poses_[7 * i + 5], poses_[7 * i + 6]),
                                tf::Vector3(poses_[7 * i], poses_[7 * i + 1], poses_[7 * i + 2]));
      const tf::Transform T_w_c = T_c_w.inverse();
      geometry_msgs::PoseStamped p;
      p.header.frame_id = "map";
      p.header.stamp = stamps_[i];
      p.pose.position.x = T_w_c.getOrigin().x();
      p.pose.position.y = T_w_c.getOrigin().y();
      p.pose.position.z = T_w_c.getOrigin().z();
      p.pose.orientation.x = T_w_c.getRotation().x();
      p.pose.orientation.y = T_w_c.getRotation().y();
      p.pose.orientation.z = T_w_c.getRotation().z();
      p.pose.orientation.w = T_w_c.getRotation().w();
      path_.poses.push_back(p);
    }
    path_.header.stamp = ros::Time::now();
    path_pub_.publish(path_);
  }

  flvis_ctx* ctx_ = nullptr;
  flvis_loop_closer* lc_ = nullptr;
  flvis_cfg cfg_;
  ros::Subscriber kf_sub_;
  ros::Publisher path_pub_;
  tf::TransformBroadcaster br_;
  nav_msgs::Path path_;
  std::vector<ros::Time> stamps_;
  std::vector<double> poses_;
  std::deque<flvis::KeyFrameConstPtr> queue_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::thread worker_;
  bool stop_ = false;
};

}  // namespace flvis_hip

PLUGINLIB_EXPORT_CLASS(flvis_hip::LoopClosingNodelet, nodelet::Nodelet)
