// look-alike of the ROS message header std_msgs/ColorRGBA.h plus the ros::Time the reference's estd.h expects to arrive with it
// (TEST INFRASTRUCTURE)
#pragma once
#include <chrono>
#include <ostream>
namespace std_msgs { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }
namespace ros {
struct Time {
  double t = 0;
  static Time now() { Time x; x.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); return x; }
  double toSec() const { return t; }
  unsigned long long toNSec() const { return (unsigned long long)(t * 1e9); }
};
inline std::ostream& operator<<(std::ostream& os, const Time& t) { return os << t.t; }
}  // namespace ros
