#pragma once
#include <functional>
#include <memory>
#include "message_filters/subscriber.h"
namespace message_filters {
template <typename Policy>
class Synchronizer {
 public:
  typedef typename Policy::Message0 M0;
  typedef typename Policy::Message1 M1;
  Synchronizer(const Policy&, Subscriber<M0>&, Subscriber<M1>&) {}
  void registerCallback(const std::function<void(const std::shared_ptr<M0 const>&, const std::shared_ptr<M1 const>&)>&) {}
};
}
