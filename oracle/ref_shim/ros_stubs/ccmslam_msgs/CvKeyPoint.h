// look-alike of the generated <ccmslam_msgs/CvKeyPoint.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/CvKeyPoint.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
namespace ccmslam_msgs {
struct CvKeyPoint {
  float fPoint2f_x;
  float fPoint2f_y;
  uint8_t size;
  float angle;
  uint8_t response;
  int8_t octave;
  typedef boost::shared_ptr<CvKeyPoint> Ptr;
  typedef boost::shared_ptr<CvKeyPoint const> ConstPtr;
};
typedef boost::shared_ptr<CvKeyPoint> CvKeyPointPtr;
typedef boost::shared_ptr<CvKeyPoint const> CvKeyPointConstPtr;
}
