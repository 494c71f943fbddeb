// ORBextractor_hip.cpp — DROP-IN replacement for the translation unit cslam/src/ORBextractor.cpp of the reference.
// Defines the constructor and operator() that the reference's own header declares (cslam/include/cslam/ORBextractor.h:103-138), compiled
// against that header; the work goes through the C ABI of libccm_hip.so (ccm_orb_*, include/ccm_hip.h) to the MI355X.  The protected
// helpers of the class (ComputePyramid, ComputeKeyPointsOctTree, DistributeOctTree, ComputeKeyPointsOld) and ExtractorNode::DivideNode
// are only ever called from inside ORBextractor.cpp; they are not defined here.
// The header gives the class no member for a device handle and defines the destructor inline, so the handles live in a registry keyed by
// the object's address; an extractor lives as long as its Tracking object (two per agent, Tracking.cpp:73-76), the registry is emptied at
// process exit.
#include <cslam/ORBextractor.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "../include/ccm_hip.h"

namespace cslam {
namespace {
struct Registry {
  std::mutex mu;
  std::map<const ORBextractor*, std::pair<ccm_ctx*, ccm_orb*>> m;
  ~Registry() { for (auto& kv : m) { ccm_orb_destroy(kv.second.second); ccm_ctx_destroy(kv.second.first); } }
};
Registry& registry() { static Registry r; return r; }
}  // namespace

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
  ccm_ctx* ctx = nullptr;
  ccm_orb* orb = nullptr;
  const char* dev = std::getenv("CCM_DEVICE");
  if (ccm_ctx_create(dev ? std::atoi(dev) : 0, &ctx) != CCM_OK || ccm_orb_create(ctx, _nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST, &orb) != CCM_OK) {
    cout << COUTFATAL << "ORBextractor: " << ccm_last_error(ctx) << endl;
    throw estd::infrastructure_ex();
  }
  // the float tables the accessors return (ORBextractor.cpp:584-600) come from the library, which computes them with the same f32 arithmetic
  mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
  ccm_orb_get_table(orb, 0, mvScaleFactor.data(), nlevels);
  ccm_orb_get_table(orb, 1, mvInvScaleFactor.data(), nlevels);
  ccm_orb_get_table(orb, 2, mvLevelSigma2.data(), nlevels);
  ccm_orb_get_table(orb, 3, mvInvLevelSigma2.data(), nlevels);
  mnFeaturesPerLevel.resize(nlevels);
  ccm_orb_features_per_level(orb, mnFeaturesPerLevel.data(), nlevels);
  mvImagePyramid.resize(nlevels);
  std::lock_guard<std::mutex> lk(registry().mu);
  registry().m[this] = std::make_pair(ctx, orb);
}

void ORBextractor::operator()(InputArray _image, InputArray _mask, vector<KeyPoint>& _keypoints, OutputArray _descriptors) {
  (void)_mask;   // ignored by the reference as well (ORBextractor.h:100)
  if (_image.empty()) return;
  Mat image = _image.getMat();
  assert(image.type() == CV_8UC1);
  ccm_orb* orb;
  {
    std::lock_guard<std::mutex> lk(registry().mu);
    orb = registry().m[this].second;
  }
  // mvImagePyramid is a public member that Frame / Tracking may read: un-bordered level images
  std::vector<uint8_t*> lv(nlevels);
  for (int l = 0; l < nlevels; l++) {
    int lw = 0, lh = 0;
    ccm_orb_level_size(orb, image.cols, image.rows, l, &lw, &lh);
    mvImagePyramid[l].create(lh, lw, CV_8UC1);
    lv[l] = mvImagePyramid[l].data;
  }
  const int cap = ccm_orb_max_keypoints(orb);
  std::vector<ccm_keypoint> kps(cap);
  Mat desc(cap, 32, CV_8U);
  int n = 0;
  if (ccm_orb_extract(orb, image.data, image.cols, image.rows, (int)image.step, kps.data(), desc.data, cap, &n, lv.data()) != CCM_OK) {
    cout << COUTFATAL << "ccm_orb_extract failed" << endl;
    throw estd::infrastructure_ex();
  }
  if (n == 0) _descriptors.release();
  else {
    _descriptors.create(n, 32, CV_8U);
    Mat out = _descriptors.getMat();
    for (int i = 0; i < n; i++) std::memcpy(out.ptr(i), desc.ptr(i), 32);
  }
  _keypoints.clear();
  _keypoints.reserve(n);
  for (int i = 0; i < n; i++) _keypoints.push_back(KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave));
}

}  // namespace cslam
