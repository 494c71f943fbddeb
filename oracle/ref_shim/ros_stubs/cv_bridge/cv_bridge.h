// look-alike of <cv_bridge/cv_bridge.h> (TEST INFRASTRUCTURE)
#pragma once
#include <opencv2/opencv.hpp>
#include <sensor_msgs/Image.h>
namespace cv_bridge { struct CvImage { std_msgs::Header header; std::string encoding; cv::Mat image; }; typedef boost::shared_ptr<CvImage> CvImagePtr; typedef boost::shared_ptr<CvImage const> CvImageConstPtr; }
