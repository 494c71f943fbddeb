// look-alike of the generated <ccmslam_msgs/Descriptor.h> (TEST INFRASTRUCTURE, scripts/gen_msg_stubs.py): the fields of cslam_msgs/msg/Descriptor.msg
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <boost/array.hpp>
#include <boost/shared_ptr.hpp>
#include <ros/time.h>
namespace ccmslam_msgs {
struct Descriptor {
  typedef boost::array<uint8_t, 32> _mDescriptor_type;
  boost::array<uint8_t, 32> mDescriptor;
  typedef boost::shared_ptr<Descriptor> Ptr;
  typedef boost::shared_ptr<Descriptor const> ConstPtr;
};
typedef boost::shared_ptr<Descriptor> DescriptorPtr;
typedef boost::shared_ptr<Descriptor const> DescriptorConstPtr;
}
