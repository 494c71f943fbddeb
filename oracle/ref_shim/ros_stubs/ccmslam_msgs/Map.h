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
  std_msgs::Header header;
  uint32_t mMsgId;
  std::vector<ccmslam_msgs::KF> Keyframes;
  std::vector<ccmslam_msgs::KFred> KFUpdates;
  std::vector<ccmslam_msgs::MP> MapPoints;
  std::vector<ccmslam_msgs::MPred> MPUpdates;
  std::vector<uint16_t> vAckKFs;
  std::vector<uint32_t> vAckMPs;
  uint16_t WeakAckKF;
  uint32_t WeakAckMP;
  uint16_t ClosestKf_Id;
  uint8_t ClosestKf_ClientId;
  typedef boost::shared_ptr<Map> Ptr;
  typedef boost::shared_ptr<Map const> ConstPtr;
};
typedef boost::shared_ptr<Map> MapPtr;
typedef boost::shared_ptr<Map const> MapConstPtr;
}
