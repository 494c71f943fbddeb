// ref_orb_driver.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's OWN cslam::ORBextractor
// (/root/reference/cslam/src/ORBextractor.cpp, compiled verbatim by oracle/Makefile.ref against the look-alike OpenCV API in
// oracle/ref_shim/opencv2/) and around the static helpers of the reference's ORBmatcher.cpp.  oracle/_ref/liborb_ref.so is what
// oracle/orb_ref.cpp / oracle/match_ref.cpp are pinned against in tests/test_ref_orb.py.  Nothing in the product links or loads this.
#include <cstdint>
#include <cstring>
#include <vector>

#include <cslam/ORBextractor.h>

struct ref_kp { float x, y, size, angle, response; int32_t octave; };

// the reference's ORBmatcher statics, compiled from ORBmatcher.cpp:1607-1669 (see ref_matcher_excerpt.cpp)
namespace cslam_ref_excerpt {
int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
}

extern "C" {

void* ref_orb_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th) { return new cslam::ORBextractor(nfeatures, scale, nlevels, ini_th, min_th); }
void ref_orb_destroy(void* p) { delete (cslam::ORBextractor*)p; }

// (*extractor)(image, cv::Mat(), keys, descriptors) exactly as Frame::ExtractORB calls it (Frame.cpp:120-123)
int ref_orb_extract(void* p, const uint8_t* img, int w, int h, int stride, ref_kp* kps, uint8_t* desc, int cap) {
  cslam::ORBextractor* ex = (cslam::ORBextractor*)p;
  cv::Mat im(h, w, CV_8UC1, (void*)img, (size_t)stride);
  std::vector<cv::KeyPoint> keys;
  cv::Mat descriptors;
  (*ex)(im, cv::Mat(), keys, descriptors);
  const int n = (int)keys.size();
  for (int i = 0; i < n && i < cap; i++) {
    kps[i] = {keys[i].pt.x, keys[i].pt.y, keys[i].size, keys[i].angle, keys[i].response, keys[i].octave};
    std::memcpy(desc + 32 * (size_t)i, descriptors.ptr(i), 32);
  }
  return n;
}
// level of mvImagePyramid of the last extraction (the ROI without its 19-px border), row-major, tightly packed
int ref_orb_get_level(void* p, int level, uint8_t* out, int* w, int* h) {
  cslam::ORBextractor* ex = (cslam::ORBextractor*)p;
  const cv::Mat& m = ex->mvImagePyramid[level];
  *w = m.cols; *h = m.rows;
  if (out) for (int r = 0; r < m.rows; r++) std::memcpy(out + (size_t)r * m.cols, m.ptr(r), (size_t)m.cols);
  return m.rows * m.cols;
}
// pixel of the bordered buffer around a level (row / col relative to the level's origin, may be negative down to -19)
int ref_orb_border_pixel(void* p, int level, int row, int col) {
  cslam::ORBextractor* ex = (cslam::ORBextractor*)p;
  const cv::Mat& m = ex->mvImagePyramid[level];
  return *(m.data + (ptrdiff_t)row * (ptrdiff_t)m.step.v + col);
}
void ref_orb_tables(void* p, float* sf, float* isf, float* s2, float* is2) {
  cslam::ORBextractor* ex = (cslam::ORBextractor*)p;
  const std::vector<float> a = ex->GetScaleFactors(), b = ex->GetInverseScaleFactors(), c = ex->GetScaleSigmaSquares(), d = ex->GetInverseScaleSigmaSquares();
  for (size_t i = 0; i < a.size(); i++) { sf[i] = a[i]; isf[i] = b[i]; s2[i] = c[i]; is2[i] = d[i]; }
}

int ref_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  cv::Mat ma(1, 32, CV_8U, (void*)a), mb(1, 32, CV_8U, (void*)b);
  return cslam_ref_excerpt::DescriptorDistance(ma, mb);
}
void ref_three_maxima(const int32_t* counts, int L, int32_t* out3) {
  std::vector<std::vector<int>> histo((size_t)L);
  for (int i = 0; i < L; i++) histo[(size_t)i].assign((size_t)counts[i], 0);
  int a = -1, b = -1, c = -1;   // the callers initialise all three to -1 (ORBmatcher.cpp:283-285)
  cslam_ref_excerpt::ComputeThreeMaxima(histo.data(), L, a, b, c);
  out3[0] = a; out3[1] = b; out3[2] = c;
}

}  // extern "C"
