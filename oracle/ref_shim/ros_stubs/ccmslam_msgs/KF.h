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
  typedef uint8_t _bSentOnce_type;
  uint8_t bSentOnce;
  typedef uint16_t _mnId_type;
  uint16_t mnId;
  typedef uint8_t _mClientId_type;
  uint8_t mClientId;
  typedef uint32_t _mUniqueId_type;
  uint32_t mUniqueId;
  typedef double _dTimestamp_type;
  double dTimestamp;
  typedef uint8_t _mbAck_type;
  uint8_t mbAck;
  typedef int16_t _mnGridCols_type;
  int16_t mnGridCols;
  typedef int16_t _mnGridRows_type;
  int16_t mnGridRows;
  typedef float _mfGridElementWidthInv_type;
  float mfGridElementWidthInv;
  typedef float _mfGridElementHeightInv_type;
  float mfGridElementHeightInv;
  typedef float _fx_type;
  float fx;
  typedef float _fy_type;
  float fy;
  typedef float _cx_type;
  float cx;
  typedef float _cy_type;
  float cy;
  typedef float _invfx_type;
  float invfx;
  typedef float _invfy_type;
  float invfy;
  typedef int16_t _N_type;
  int16_t N;
  typedef std::vector<ccmslam_msgs::CvKeyPoint> _mvKeysUn_type;
  std::vector<ccmslam_msgs::CvKeyPoint> mvKeysUn;
  typedef std::vector<ccmslam_msgs::Descriptor> _mDescriptors_type;
  std::vector<ccmslam_msgs::Descriptor> mDescriptors;
  typedef boost::array<float, 16> _mTcpred_type;
  boost::array<float, 16> mTcpred;
  typedef boost::array<float, 16> _mTcpar_type;
  boost::array<float, 16> mTcpar;
  typedef uint8_t _mbPoseChanged_type;
  uint8_t mbPoseChanged;
  typedef uint8_t _mbServerBA_type;
  uint8_t mbServerBA;
  typedef boost::array<float, 16> _mT_SC_type;
  boost::array<float, 16> mT_SC;
  typedef int8_t _mnScaleLevels_type;
  int8_t mnScaleLevels;
  typedef float _mfScaleFactor_type;
  float mfScaleFactor;
  typedef float _mfLogScaleFactor_type;
  float mfLogScaleFactor;
  typedef boost::array<float, 8> _mvScaleFactors_type;
  boost::array<float, 8> mvScaleFactors;
  typedef boost::array<float, 8> _mvLevelSigma2_type;
  boost::array<float, 8> mvLevelSigma2;
  typedef boost::array<float, 8> _mvInvLevelSigma2_type;
  boost::array<float, 8> mvInvLevelSigma2;
  typedef int16_t _mnMinX_type;
  int16_t mnMinX;
  typedef int16_t _mnMinY_type;
  int16_t mnMinY;
  typedef int16_t _mnMaxX_type;
  int16_t mnMaxX;
  typedef int16_t _mnMaxY_type;
  int16_t mnMaxY;
  typedef boost::array<float, 9> _mK_type;
  boost::array<float, 9> mK;
  typedef std::vector<uint32_t> _mvpMapPoints_Ids_type;
  std::vector<uint32_t> mvpMapPoints_Ids;
  typedef std::vector<uint8_t> _mvpMapPoints_ClientIds_type;
  std::vector<uint8_t> mvpMapPoints_ClientIds;
  typedef std::vector<uint16_t> _mvpMapPoints_VectId_type;
  std::vector<uint16_t> mvpMapPoints_VectId;
  typedef uint16_t _mpPred_KfId_type;
  uint16_t mpPred_KfId;
  typedef uint8_t _mpPred_KfClientId_type;
  uint8_t mpPred_KfClientId;
  typedef uint16_t _mpPar_KfId_type;
  uint16_t mpPar_KfId;
  typedef uint8_t _mpPar_KfClientId_type;
  uint8_t mpPar_KfClientId;
  typedef uint8_t _mbBad_type;
  uint8_t mbBad;
  typedef boost::shared_ptr<KF> Ptr;
  typedef boost::shared_ptr<KF const> ConstPtr;
};
typedef boost::shared_ptr<KF> KFPtr;
typedef boost::shared_ptr<KF const> KFConstPtr;
}
