#pragma once
#include <string>
#include "ros/ros.h"
namespace tf {
struct Vector3 {
  Vector3(double x = 0, double y = 0, double z = 0) : v{x, y, z} {}
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  double v[3];
};
struct Quaternion {
  Quaternion(double x = 0, double y = 0, double z = 0, double w = 1) : q{x, y, z, w} {}
  double x() const { return q[0]; }
  double y() const { return q[1]; }
  double z() const { return q[2]; }
  double w() const { return q[3]; }
  double q[4];
};
struct Transform {
  Transform() {}
  Transform(const Quaternion& q, const Vector3& t) : q_(q), t_(t) {}
  Transform inverse() const { return *this; }
  const Vector3& getOrigin() const { return t_; }
  Quaternion getRotation() const { return q_; }
  Quaternion q_;
  Vector3 t_;
};
struct StampedTransform : Transform {
  StampedTransform(const Transform& T, const ros::Time&, const std::string&, const std::string&) : Transform(T) {}
};
class TransformBroadcaster {
 public:
  void sendTransform(const StampedTransform&) {}
};
}
