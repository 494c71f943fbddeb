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
  typedef uint8_t _bSentOnce_type;
  uint8_t bSentOnce;
  typedef uint32_t _mnId_type;
  uint32_t mnId;
  typedef uint8_t _mClientId_type;
  uint8_t mClientId;
  typedef uint32_t _mUniqueId_type;
  uint32_t mUniqueId;
  typedef int16_t _mnFirstKFid_type;
  int16_t mnFirstKFid;
  typedef uint8_t _mnFirstKfClientId_type;
  uint8_t mnFirstKfClientId;
  typedef uint8_t _mbAck_type;
  uint8_t mbAck;
  typedef boost::array<float, 3> _mPosPred_type;
  boost::array<float, 3> mPosPred;
  typedef boost::array<float, 3> _mPosPar_type;
  boost::array<float, 3> mPosPar;
  typedef uint8_t _mbPoseChanged_type;
  uint8_t mbPoseChanged;
  typedef uint8_t _mbServerBA_type;
  uint8_t mbServerBA;
  typedef std::vector<uint16_t> _mObservations_KFIDs_type;
  std::vector<uint16_t> mObservations_KFIDs;
  typedef std::vector<uint8_t> _mObservations_KFClientIDs_type;
  std::vector<uint8_t> mObservations_KFClientIDs;
  typedef std::vector<uint16_t> _mObservations_n_type;
  std::vector<uint16_t> mObservations_n;
  typedef boost::array<float, 3> _mNormalVector_type;
  boost::array<float, 3> mNormalVector;
  typedef uint8_t _mbNormalAndDepthChanged_type;
  uint8_t mbNormalAndDepthChanged;
  typedef boost::array<uint8_t, 32> _mDescriptor_type;
  boost::array<uint8_t, 32> mDescriptor;
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
  typedef float _mfMinDistance_type;
  float mfMinDistance;
  typedef float _mfMaxDistance_type;
  float mfMaxDistance;
  typedef uint8_t _mbMultiUse_type;
  uint8_t mbMultiUse;
  typedef boost::shared_ptr<MP> Ptr;
  typedef boost::shared_ptr<MP const> ConstPtr;
};
typedef boost::shared_ptr<MP> MPPtr;
typedef boost::shared_ptr<MP const> MPConstPtr;
}
