// look-alike of the generated <ccmslam_msgs/MPred.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/MPred.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
namespace ccmslam_msgs {
struct MPred {
  uint32_t mnId;
  uint8_t mClientId;
  uint32_t mUniqueId;
  uint8_t mbAck;
  boost::array<float, 3> mPosPred;
  boost::array<float, 3> mPosPar;
  uint8_t mbNormalAndDepthChanged;
  uint8_t mbServerBA;
  uint16_t mpPredKFId;
  uint8_t mpPredKFClientId;
  uint16_t mpParKFId;
  uint8_t mpParKFClientId;
  uint8_t mbBad;
  uint8_t mbMultiUse;
  typedef boost::shared_ptr<MPred> Ptr;
  typedef boost::shared_ptr<MPred const> ConstPtr;
};
typedef boost::shared_ptr<MPred> MPredPtr;
typedef boost::shared_ptr<MPred const> MPredConstPtr;
}
