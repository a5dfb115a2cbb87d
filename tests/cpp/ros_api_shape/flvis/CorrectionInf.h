// flvis/CorrectionInf as the reference's msg/CorrectionInf.msg defines it
#pragma once
#include <memory>
#include <vector>
#include "geometry_msgs/PoseStamped.h"
namespace flvis {
struct CorrectionInf {
  int64_t frame_id = 0;
  geometry_msgs::Transform T_c_w;
  int32_t lm_count = 0;
  std_msgs::Int64MultiArray lm_id_data;
  std::vector<geometry_msgs::Vector3> lm_3d_data;
  int32_t lm_outlier_count = 0;
  std_msgs::Int64MultiArray lm_outlier_id_data;
};
typedef std::shared_ptr<CorrectionInf const> CorrectionInfConstPtr;
}
