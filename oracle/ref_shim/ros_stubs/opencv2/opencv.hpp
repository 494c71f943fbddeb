// <opencv2/opencv.hpp> for the check against the reference's REAL class headers (TEST INFRASTRUCTURE): the look-alike of oracle/ref_shim plus the legacy
// C-API type PnPSolver.h names in its declarations
#pragma once
#include_next <opencv2/opencv.hpp>
struct CvMat { int rows = 0, cols = 0; union { double* db; float* fl; unsigned char* ptr; } data; };
// cv::undistortPoints for the reference's Frame.cpp compiled as it is (Frame.cpp:295, 321): declared here, defined by the test harness
// (oracle/real_graph_support.cpp) on top of the oracle's restatement of OpenCV 4.2's cvUndistortPointsInternal (oracle/match_ref.cpp:633-690)
namespace cv { void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& dist, const Mat& R, const Mat& P); }
