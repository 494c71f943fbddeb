// look-alike of the generated <ccmslam_msgs/Map.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/Map.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
#include <ccmslam_msgs/KF.h>
#include <ccmslam_msgs/KFred.h>
#include <ccmslam_msgs/MP.h>
#include <ccmslam_msgs/MPred.h>
#include <std_msgs/Header.h>
namespace ccmslam_msgs {
struct Map {
  typedef std_msgs::Header _header_type;
  std_msgs::Header header;
  typedef uint32_t _mMsgId_type;
  uint32_t mMsgId;
  typedef std::vector<ccmslam_msgs::KF> _Keyframes_type;
  std::vector<ccmslam_msgs::KF> Keyframes;
  typedef std::vector<ccmslam_msgs::KFred> _KFUpdates_type;
  std::vector<ccmslam_msgs::KFred> KFUpdates;
  typedef std::vector<ccmslam_msgs::MP> _MapPoints_type;
  std::vector<ccmslam_msgs::MP> MapPoints;
  typedef std::vector<ccmslam_msgs::MPred> _MPUpdates_type;
  std::vector<ccmslam_msgs::MPred> MPUpdates;
  typedef std::vector<uint16_t> _vAckKFs_type;
  std::vector<uint16_t> vAckKFs;
  typedef std::vector<uint32_t> _vAckMPs_type;
  std::vector<uint32_t> vAckMPs;
  typedef uint16_t _WeakAckKF_type;
  uint16_t WeakAckKF;
  typedef uint32_t _WeakAckMP_type;
  uint32_t WeakAckMP;
  typedef uint16_t _ClosestKf_Id_type;
  uint16_t ClosestKf_Id;
  typedef uint8_t _ClosestKf_ClientId_type;
  uint8_t ClosestKf_ClientId;
  typedef boost::shared_ptr<Map> Ptr;
  typedef boost::shared_ptr<Map const> ConstPtr;
};
typedef boost::shared_ptr<Map> MapPtr;
typedef boost::shared_ptr<Map const> MapConstPtr;
}
