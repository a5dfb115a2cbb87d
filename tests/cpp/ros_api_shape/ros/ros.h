// declarations-only shape of the roscpp subset used by ros/src/*.cpp (see README.md beside this directory)
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>

#include "boost/bind.hpp"

namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  static Time now() { return Time(); }
};
class Publisher {
 public:
  template <typename M>
  void publish(const M&) const {}
};
class Subscriber {};
class NodeHandle {
 public:
  template <typename T>
  bool getParam(const std::string&, T&) const { return true; }
  template <typename T>
  void param(const std::string&, T& v, const T& d) const { v = d; }
  template <typename M>
  Publisher advertise(const std::string&, uint32_t) { return Publisher(); }
  template <typename M, typename C>
  Subscriber subscribe(const std::string&, uint32_t, void (C::*)(const std::shared_ptr<M const>&), C*) { return Subscriber(); }
};
}  // namespace ros
