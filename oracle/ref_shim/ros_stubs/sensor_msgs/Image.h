#pragma once
#include <vector>
#include <boost/shared_ptr.hpp>
#include <std_msgs/Header.h>
namespace sensor_msgs { struct Image { std_msgs::Header header; unsigned height = 0, width = 0; std::string encoding; std::vector<unsigned char> data; typedef boost::shared_ptr<Image const> ConstPtr; }; typedef boost::shared_ptr<Image const> ImageConstPtr; typedef boost::shared_ptr<Image> ImagePtr; }
