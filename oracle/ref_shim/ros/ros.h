// look-alike of <ros/ros.h> (TEST INFRASTRUCTURE): ros::Time only (see std_msgs/ColorRGBA.h)
#pragma once
#include <std_msgs/ColorRGBA.h>
