#pragma once
#include "ros/ros.h"
namespace message_filters {
template <typename M>
class Subscriber {
 public:
  void subscribe(ros::NodeHandle&, const std::string&, uint32_t) {}
};
}
