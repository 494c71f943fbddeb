#pragma once
#include <ros/ros.h>
namespace image_transport { struct Publisher {}; struct Subscriber {}; struct ImageTransport { explicit ImageTransport(const ros::NodeHandle&) {} }; }
