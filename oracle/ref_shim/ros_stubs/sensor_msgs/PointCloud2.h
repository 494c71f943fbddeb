#pragma once
#include <std_msgs/Header.h>
namespace sensor_msgs { struct PointCloud2 { std_msgs::Header header; }; }
