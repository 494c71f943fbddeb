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
  typedef uint16_t _mnId_type;
  uint16_t mnId;
  typedef uint8_t _mClientId_type;
  uint8_t mClientId;
  typedef uint32_t _mUniqueId_type;
  uint32_t mUniqueId;
  typedef uint8_t _mbAck_type;
  uint8_t mbAck;
  typedef boost::array<float, 16> _mTcpred_type;
  boost::array<float, 16> mTcpred;
  typedef boost::array<float, 16> _mTcpar_type;
  boost::array<float, 16> mTcpar;
  typedef uint16_t _mpPred_KfId_type;
  uint16_t mpPred_KfId;
  typedef uint8_t _mpPred_KfClientId_type;
  uint8_t mpPred_KfClientId;
  typedef uint16_t _mpPar_KfId_type;
  uint16_t mpPar_KfId;
  typedef uint8_t _mpPar_KfClientId_type;
  uint8_t mpPar_KfClientId;
  typedef uint8_t _mbServerBA_type;
  uint8_t mbServerBA;
  typedef uint8_t _mbBad_type;
  uint8_t mbBad;
  typedef boost::shared_ptr<KFred> Ptr;
  typedef boost::shared_ptr<KFred const> ConstPtr;
};
typedef boost::shared_ptr<KFred> KFredPtr;
typedef boost::shared_ptr<KFred const> KFredConstPtr;
}
