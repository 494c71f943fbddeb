// look-alike of the generated <ccmslam_msgs/MP.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/MP.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
namespace ccmslam_msgs {
struct MP {
  uint8_t bSentOnce;
  uint32_t mnId;
  uint8_t mClientId;
  uint32_t mUniqueId;
  int16_t mnFirstKFid;
  uint8_t mnFirstKfClientId;
  uint8_t mbAck;
  boost::array<float, 3> mPosPred;
  boost::array<float, 3> mPosPar;
  uint8_t mbPoseChanged;
  uint8_t mbServerBA;
  std::vector<uint16_t> mObservations_KFIDs;
  std::vector<uint8_t> mObservations_KFClientIDs;
  std::vector<uint16_t> mObservations_n;
  boost::array<float, 3> mNormalVector;
  uint8_t mbNormalAndDepthChanged;
  boost::array<uint8_t, 32> mDescriptor;
  uint16_t mpPredKFId;
  uint8_t mpPredKFClientId;
  uint16_t mpParKFId;
  uint8_t mpParKFClientId;
  uint8_t mbBad;
  float mfMinDistance;
  float mfMaxDistance;
  uint8_t mbMultiUse;
  typedef boost::shared_ptr<MP> Ptr;
  typedef boost::shared_ptr<MP const> ConstPtr;
};
typedef boost::shared_ptr<MP> MPPtr;
typedef boost::shared_ptr<MP const> MPConstPtr;
}
