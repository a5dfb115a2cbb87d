#pragma once
#include "ros/ros.h"
namespace nodelet {
class Nodelet {
 public:
  virtual ~Nodelet() {}
  virtual void onInit() = 0;
 protected:
  ros::NodeHandle& getNodeHandle() { return nh_; }
  ros::NodeHandle& getMTNodeHandle() { return nh_; }
  ros::NodeHandle& getPrivateNodeHandle() { return nh_; }
 private:
  ros::NodeHandle nh_;
};
}  // namespace nodelet
#define NODELET_FATAL(...) std::printf(__VA_ARGS__)
#define NODELET_INFO(...) std::printf(__VA_ARGS__)
#define NODELET_WARN_THROTTLE(rate, ...) std::printf(__VA_ARGS__)
#define NODELET_ERROR_THROTTLE(rate, ...) std::printf(__VA_ARGS__)
