// look-alike of the generated <ccmslam_msgs/KF.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/KF.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
#include <ccmslam_msgs/CvKeyPoint.h>
#include <ccmslam_msgs/Descriptor.h>
namespace ccmslam_msgs {
struct KF {
  uint8_t bSentOnce;
  uint16_t mnId;
  uint8_t mClientId;
  uint32_t mUniqueId;
  double dTimestamp;
  uint8_t mbAck;
  int16_t mnGridCols;
  int16_t mnGridRows;
  float mfGridElementWidthInv;
  float mfGridElementHeightInv;
  float fx;
  float fy;
  float cx;
  float cy;
  float invfx;
  float invfy;
  int16_t N;
  std::vector<ccmslam_msgs::CvKeyPoint> mvKeysUn;
  std::vector<ccmslam_msgs::Descriptor> mDescriptors;
  boost::array<float, 16> mTcpred;
  boost::array<float, 16> mTcpar;
  uint8_t mbPoseChanged;
  uint8_t mbServerBA;
  boost::array<float, 16> mT_SC;
  int8_t mnScaleLevels;
  float mfScaleFactor;
  float mfLogScaleFactor;
  boost::array<float, 8> mvScaleFactors;
  boost::array<float, 8> mvLevelSigma2;
  boost::array<float, 8> mvInvLevelSigma2;
  int16_t mnMinX;
  int16_t mnMinY;
  int16_t mnMaxX;
  int16_t mnMaxY;
  boost::array<float, 9> mK;
  std::vector<uint32_t> mvpMapPoints_Ids;
  std::vector<uint8_t> mvpMapPoints_ClientIds;
  std::vector<uint16_t> mvpMapPoints_VectId;
  uint16_t mpPred_KfId;
  uint8_t mpPred_KfClientId;
  uint16_t mpPar_KfId;
  uint8_t mpPar_KfClientId;
  uint8_t mbBad;
  typedef boost::shared_ptr<KF> Ptr;
  typedef boost::shared_ptr<KF const> ConstPtr;
};
typedef boost::shared_ptr<KF> KFPtr;
typedef boost::shared_ptr<KF const> KFConstPtr;
}
