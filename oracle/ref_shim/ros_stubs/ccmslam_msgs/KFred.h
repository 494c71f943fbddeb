// look-alike of the generated <ccmslam_msgs/KFred.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/KFred.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
namespace ccmslam_msgs {
struct KFred {
  uint16_t mnId;
  uint8_t mClientId;
  uint32_t mUniqueId;
  uint8_t mbAck;
  boost::array<float, 16> mTcpred;
  boost::array<float, 16> mTcpar;
  uint16_t mpPred_KfId;
  uint8_t mpPred_KfClientId;
  uint16_t mpPar_KfId;
  uint8_t mpPar_KfClientId;
  uint8_t mbServerBA;
  uint8_t mbBad;
  typedef boost::shared_ptr<KFred> Ptr;
  typedef boost::shared_ptr<KFred const> ConstPtr;
};
typedef boost::shared_ptr<KFred> KFredPtr;
typedef boost::shared_ptr<KFred const> KFredConstPtr;
}
