#pragma once
#include "std_msgs/Header.h"
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct Twist { Vector3 linear, angular; };
}  // namespace geometry_msgs
