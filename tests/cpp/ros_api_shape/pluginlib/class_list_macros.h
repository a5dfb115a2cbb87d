#pragma once
#define PLUGINLIB_EXPORT_CLASS(cls, base) static_assert(std::is_base_of<base, cls>::value, "plugin class must derive from its base");
#include <type_traits>
