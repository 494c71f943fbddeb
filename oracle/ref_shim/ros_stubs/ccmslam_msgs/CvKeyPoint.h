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
  typedef float _fPoint2f_x_type;
  float fPoint2f_x;
  typedef float _fPoint2f_y_type;
  float fPoint2f_y;
  typedef uint8_t _size_type;
  uint8_t size;
  typedef float _angle_type;
  float angle;
  typedef uint8_t _response_type;
  uint8_t response;
  typedef int8_t _octave_type;
  int8_t octave;
  typedef boost::shared_ptr<CvKeyPoint> Ptr;
  typedef boost::shared_ptr<CvKeyPoint const> ConstPtr;
};
typedef boost::shared_ptr<CvKeyPoint> CvKeyPointPtr;
typedef boost::shared_ptr<CvKeyPoint const> CvKeyPointConstPtr;
}
