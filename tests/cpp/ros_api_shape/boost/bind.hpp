// shape only: boost::bind(&C::member, this, _1, _2) as message_filters callbacks use it
#pragma once
#include <functional>
namespace boost {
template <typename R, typename C, typename A, typename B, typename P1, typename P2>
std::function<R(A, B)> bind(R (C::*f)(A, B), C* self, P1, P2) {
  return [=](A a, B b) { return (self->*f)(a, b); };
}
}  // namespace boost
namespace { struct flvis_shape_ph1 {} _1; struct flvis_shape_ph2 {} _2; }
