#pragma once
#include <boost/shared_ptr.hpp>
namespace std_msgs { struct Bool { unsigned char data = 0; typedef boost::shared_ptr<Bool const> ConstPtr; }; typedef boost::shared_ptr<Bool const> BoolConstPtr; }
