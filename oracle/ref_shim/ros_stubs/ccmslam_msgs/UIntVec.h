// look-alike of the generated <ccmslam_msgs/UIntVec.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/UIntVec.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
namespace ccmslam_msgs {
struct UIntVec {
  typedef std::vector<uint32_t> _uintvec_type;
  std::vector<uint32_t> uintvec;
  typedef boost::shared_ptr<UIntVec> Ptr;
  typedef boost::shared_ptr<UIntVec const> ConstPtr;
};
typedef boost::shared_ptr<UIntVec> UIntVecPtr;
typedef boost::shared_ptr<UIntVec const> UIntVecConstPtr;
}
