// look-alike of the generated ROS message header (TEST INFRASTRUCTURE): fields of cslam_msgs/msg/CvKeyPoint.msg
#pragma once
#include <cstdint>
namespace ccmslam_msgs { struct CvKeyPoint { float fPoint2f_x = 0, fPoint2f_y = 0, angle = 0; int8_t octave = 0; uint8_t response = 0, size = 0; }; }
