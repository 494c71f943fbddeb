// look-alike of <ros/ros.h> (TEST INFRASTRUCTURE): the handle / publisher / subscriber types the reference's class DECLARATIONS name (no behaviour)
#pragma once
#include <string>
#include <dirent.h>     // (the real <ros/ros.h> drags these in; Map.cpp / MapPoint.cpp use mkdir / opendir / usleep without including them)
#include <sys/stat.h>
#include <unistd.h>
#include <ros/time.h>
#include <boost/shared_ptr.hpp>
namespace ros {
struct Publisher { template <class M> void publish(const M&) const {} unsigned getNumSubscribers() const { return 0; } };
struct Subscriber {};
struct ServiceServer {};
struct ServiceClient {};
struct Timer {};
struct Rate { explicit Rate(double) {} bool sleep() { return true; } };
struct NodeHandle {
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  template <class M> Publisher advertise(const std::string&, unsigned) { return Publisher(); }
  template <class M, class T> Subscriber subscribe(const std::string&, unsigned, void (T::*)(M), T*) { return Subscriber(); }
  template <class V> bool param(const std::string&, V& out, const V& def) const { out = def; return false; }
  template <class V> bool getParam(const std::string&, V&) const { return false; }
};
inline bool ok() { return true; }
inline void spinOnce() {}
}  // namespace ros
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_WARN_STREAM(x) ((void)0)
#define ROS_ERROR_STREAM(x) ((void)0)
