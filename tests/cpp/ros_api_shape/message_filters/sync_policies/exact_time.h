#pragma once
#include <cstdint>
namespace message_filters { namespace sync_policies {
template <typename M0, typename M1>
struct ExactTime {
  typedef M0 Message0;
  typedef M1 Message1;
  explicit ExactTime(uint32_t) {}
};
} }
