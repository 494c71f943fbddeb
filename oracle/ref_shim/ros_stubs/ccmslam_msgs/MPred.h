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
  typedef uint32_t _mnId_type;
  uint32_t mnId;
  typedef uint8_t _mClientId_type;
  uint8_t mClientId;
  typedef uint32_t _mUniqueId_type;
  uint32_t mUniqueId;
  typedef uint8_t _mbAck_type;
  uint8_t mbAck;
  typedef boost::array<float, 3> _mPosPred_type;
  boost::array<float, 3> mPosPred;
  typedef boost::array<float, 3> _mPosPar_type;
  boost::array<float, 3> mPosPar;
  typedef uint8_t _mbNormalAndDepthChanged_type;
  uint8_t mbNormalAndDepthChanged;
  typedef uint8_t _mbServerBA_type;
  uint8_t mbServerBA;
  typedef uint16_t _mpPredKFId_type;
  uint16_t mpPredKFId;
  typedef uint8_t _mpPredKFClientId_type;
  uint8_t mpPredKFClientId;
  typedef uint16_t _mpParKFId_type;
  uint16_t mpParKFId;
  typedef uint8_t _mpParKFClientId_type;
  uint8_t mpParKFClientId;
  typedef uint8_t _mbBad_type;
  uint8_t mbBad;
  typedef uint8_t _mbMultiUse_type;
  uint8_t mbMultiUse;
  typedef boost::shared_ptr<MPred> Ptr;
  typedef boost::shared_ptr<MPred const> ConstPtr;
};
typedef boost::shared_ptr<MPred> MPredPtr;
typedef boost::shared_ptr<MPred const> MPredConstPtr;
}
