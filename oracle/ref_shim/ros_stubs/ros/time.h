// look-alike of <ros/time.h> (TEST INFRASTRUCTURE): ros::Time comes from the std_msgs look-alike of oracle/ref_shim
#pragma once
#include <std_msgs/ColorRGBA.h>
namespace ros { struct Duration { double d = 0; Duration() {} explicit Duration(double x) : d(x) {} double toSec() const { return d; } void sleep() const {} }; }
