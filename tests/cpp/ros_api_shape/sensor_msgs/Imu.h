#pragma once
#include <memory>
#include "geometry_msgs/PoseStamped.h"
namespace sensor_msgs {
struct Imu {
  std_msgs::Header header;
  geometry_msgs::Quaternion orientation;
  geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
typedef std::shared_ptr<Imu const> ImuConstPtr;
}
