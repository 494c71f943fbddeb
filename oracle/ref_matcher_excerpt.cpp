// ref_matcher_excerpt.cpp — TEST INFRASTRUCTURE.  Wraps two member functions of the reference's ORBmatcher that need nothing but
// cv::Mat and std::vector: ORBmatcher::ComputeThreeMaxima and ORBmatcher::DescriptorDistance (cslam/src/ORBmatcher.cpp:1606-1669).
// oracle/Makefile.ref extracts exactly those lines from the reference source into oracle/_ref/gen/ORBmatcher_1606_1669.inc at build
// time (nothing of the reference is stored in this repository); this file supplies the class shell they are members of.
// The rest of ORBmatcher.cpp needs the whole Frame / KeyFrame / MapPoint / Map object graph and is not compiled.
#include <cstdint>
#include <vector>
#include <opencv2/opencv.hpp>
using namespace std;

namespace cslam_ref_excerpt {
class ORBmatcher {
 public:
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
  void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
};
#include "ORBmatcher_1606_1669.inc"
int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return ORBmatcher::DescriptorDistance(a, b); }
void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) { ORBmatcher m; m.ComputeThreeMaxima(histo, L, ind1, ind2, ind3); }
}  // namespace cslam_ref_excerpt
