// flvis/KeyFrame as the reference's msg/KeyFrame.msg defines it
#pragma once
#include <memory>
#include <vector>
#include "geometry_msgs/PoseStamped.h"
#include "sensor_msgs/Image.h"
namespace flvis {
struct KeyFrame {
  std_msgs::Header header;
  int64_t frame_id = 0;
  int8_t command = 0;
  sensor_msgs::Image img0, img1;
  int32_t lm_count = 0;
  std_msgs::Int64MultiArray lm_id_data;
  std::vector<geometry_msgs::Vector3> lm_2d_data, lm_3d_data;
  std_msgs::UInt8MultiArray lm_descriptor_data;
  geometry_msgs::Transform T_c_w;
};
typedef std::shared_ptr<KeyFrame const> KeyFrameConstPtr;
}
