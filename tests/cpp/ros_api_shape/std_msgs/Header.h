#pragma once
#include <vector>
#include <string>
#include "ros/ros.h"
namespace std_msgs {
struct Header {
  uint32_t seq = 0;
  ros::Time stamp;
  std::string frame_id;
};
struct MultiArrayDimension {
  std::string label;
  uint32_t size = 0, stride = 0;
};
struct MultiArrayLayout {
  std::vector<MultiArrayDimension> dim;
  uint32_t data_offset = 0;
};
struct Int64MultiArray {
  MultiArrayLayout layout;
  std::vector<int64_t> data;
};
struct UInt8MultiArray {
  MultiArrayLayout layout;
  std::vector<uint8_t> data;
};
}  // namespace std_msgs
