// look-alike of <std_msgs/Header.h> (TEST INFRASTRUCTURE)
#pragma once
#include <cstdint>
#include <string>
#include <ros/time.h>
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
