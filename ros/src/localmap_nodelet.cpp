// flvis/LocalMapNodeletClass on the MI355X: subscribes the KeyFrame stream of the tracker, runs the sliding-window bundle
// adjustment of LocalMapNodeletClass::frame_callback (src/backend/vo_localmap.cpp:87-380) through flvis_ba_push_keyframe and
// publishes the flvis/CorrectionInf it produces on /vo_localmap_feedback (src/backend/vo_localmap.cpp:452, correction_inf_msg.cpp:13-64).
// The optimiser needs the rectified intrinsics only: they come from the same yaml file (/yamlconfigfile) as in the reference
// (vo_localmap.cpp:395-441).  Compile-gated by ros/CMakeLists.txt; never compiled in this repository's build image (no ROS).
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>
#include <ros/ros.h>

#include <string>
#include <vector>

#include <flvis/CorrectionInf.h>
#include <flvis/KeyFrame.h>

#include "flvis_hip.h"

namespace flvis_hip {

class LocalMapNodelet : public nodelet::Nodelet {
 public:
  ~LocalMapNodelet() override {
    if (ctx_) flvis_hip_destroy(ctx_);
  }

 private:
  void onInit() override {
    ros::NodeHandle& nh = getNodeHandle();
    std::string yaml;
    nh.getParam("/yamlconfigfile", yaml);
    char err[256] = {0};
    if (flvis_config_load(yaml.c_str(), &cfg_, err, sizeof(err)) != FLVIS_OK) {
      NODELET_FATAL("flvis_config_load(%s): %s", yaml.c_str(), err);
      return;
    }
    if (flvis_hip_create(0, FLVIS_STREAM_NEW, &ctx_) != FLVIS_OK || flvis_tracker_create(ctx_, &cfg_, 1, 0, 0) != FLVIS_OK) {
      NODELET_FATAL("cannot create the HIP local map: %s", ctx_ ? flvis_last_error(ctx_) : "no MI355X visible (no CPU fallback exists)");
      return;
    }
    corr_pub_ = nh.advertise<flvis::CorrectionInf>("/vo_localmap_feedback", 1);
    kf_sub_ = nh.subscribe<flvis::KeyFrame>("/vo_kf", 2, &LocalMapNodelet::keyframeCallback, this);
    out_id_.resize(4096);
    out_3d_.resize(3 * 4096);
    out_oid_.resize(8192);
  }

  void keyframeCallback(const flvis::KeyFrameConstPtr& kf) {
    if (kf->command != 0) return;  // KFMSG_CMD_RESET_LM: never published by the reference's v2 (vo_tracking.cpp:431)
    const int n = kf->lm_count;
    std::vector<double> p2(2 * n), p3(3 * n);
    for (int i = 0; i < n; i++) {
      p2[2 * i] = kf->lm_2d_data[i].x;
      p2[2 * i + 1] = kf->lm_2d_data[i].y;
      p3[3 * i] = kf->lm_3d_data[i].x;
      p3[3 * i + 1] = kf->lm_3d_data[i].y;
      p3[3 * i + 2] = kf->lm_3d_data[i].z;
    }
    const double T[7] = {kf->T_c_w.translation.x, kf->T_c_w.translation.y, kf->T_c_w.translation.z, kf->T_c_w.rotation.x,
                         kf->T_c_w.rotation.y,    kf->T_c_w.rotation.z,    kf->T_c_w.rotation.w};
    int64_t fid = 0;
    double To[7];
    int lm_count = 0, out_count = 0;
    const int rc = flvis_ba_push_keyframe(ctx_, 0, kf->frame_id, T, n, kf->lm_id_data.data.data(), p2.data(), p3.data(), 4096, &fid, To,
                                          &lm_count, out_id_.data(), out_3d_.data(), &out_count, out_oid_.data());
    if (rc < 0) {
      NODELET_ERROR_THROTTLE(1.0, "ba_push_keyframe: %s", flvis_last_error(ctx_));
      return;
    }
    if (rc == 0) return;  // the window is still filling
    flvis::CorrectionInf c;  // CorrectionInfMsg::pub (src/utils/correction_inf_msg.cpp:13-64)
    c.frame_id = fid;
    c.T_c_w.translation.x = To[0];
    c.T_c_w.translation.y = To[1];
    c.T_c_w.translation.z = To[2];
    c.T_c_w.rotation.x = To[3];
    c.T_c_w.rotation.y = To[4];
    c.T_c_w.rotation.z = To[5];
    c.T_c_w.rotation.w = To[6];
    c.lm_count = lm_count;
    c.lm_id_data.data.assign(out_id_.begin(), out_id_.begin() + lm_count);
    c.lm_3d_data.resize(lm_count);
    for (int i = 0; i < lm_count; i++) {
      c.lm_3d_data[i].x = out_3d_[3 * i];
      c.lm_3d_data[i].y = out_3d_[3 * i + 1];
      c.lm_3d_data[i].z = out_3d_[3 * i + 2];
    }
    c.lm_outlier_count = out_count;
    c.lm_outlier_id_data.data.assign(out_oid_.begin(), out_oid_.begin() + out_count);
    corr_pub_.publish(c);
  }

  flvis_ctx* ctx_ = nullptr;
  flvis_cfg cfg_;
  ros::Publisher corr_pub_;
  ros::Subscriber kf_sub_;
  std::vector<int64_t> out_id_, out_oid_;
  std::vector<double> out_3d_;
};

}  // namespace flvis_hip

PLUGINLIB_EXPORT_CLASS(flvis_hip::LocalMapNodelet, nodelet::Nodelet)
