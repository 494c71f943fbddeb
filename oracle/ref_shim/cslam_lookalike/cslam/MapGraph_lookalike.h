// MapGraph_lookalike.h — LOOK-ALIKES of the reference's map classes (TEST INFRASTRUCTURE).
// cslam::KeyFrame, MapPoint, Map, Frame and Communicator with exactly the members that the hot-path translation units touch
// (cslam/src/Optimizer.cpp, our shim/Optimizer_hip.cpp), same names and types as cslam/include/cslam/{KeyFrame,MapPoint,Map,Frame}.h
// (line numbers cited per member).  The real headers cannot be parsed here: they include ROS messages, PCL, DBoW2, cereal and the
// dense-mapping back end.  This directory is put BEFORE the reference's include directory on the search path, so that the REAL
// <cslam/Optimizer.h>, <cslam/Converter.h>, <cslam/Datatypes.h>, <cslam/config.h>, <cslam/estd.h> pick these classes up through their own
// `#include <cslam/KeyFrame.h>` lines.  Two uses:
//   oracle/Makefile.ref   compiles the reference's Optimizer.cpp + Converter.cc VERBATIM against them (oracle/_ref/liboptimizer_ref.so)
//   shim/Makefile         compiles and syntax-checks OUR drop-in shim/Optimizer_hip.cpp against the same classes
// and oracle/ref_optimizer_driver.cpp builds the same synthetic map for both, so the two can be compared through the class API.
// Mutexes, communication, serialisation and the covisibility bookkeeping of the real classes are out of scope and absent.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <vector>
#include <unistd.h>   // usleep: reaches Optimizer.cpp through <ros/ros.h> in a real build
#include <boost/shared_ptr.hpp>
#include <boost/enable_shared_from_this.hpp>
#include <opencv2/opencv.hpp>
#include <cslam/config.h>
#include <cslam/estd.h>
#include <cslam/Datatypes.h>
#include <thirdparty/DBoW2/DBoW2/BowVector.h>      // the reference's vendored DBoW2 headers (header-only value types)
#include <thirdparty/DBoW2/DBoW2/FeatureVector.h>

#define FRAME_GRID_ROWS 48   // Frame.h:51-52
#define FRAME_GRID_COLS 75

namespace cslam {
using namespace estd;
class KeyFrame;
class MapPoint;
class Map;
class Frame;
class Communicator {};   // Optimizer.h only names the type (commptr)

class KeyFrame : public boost::enable_shared_from_this<KeyFrame> {
 public:
  typedef boost::shared_ptr<KeyFrame> kfptr;
  typedef boost::shared_ptr<MapPoint> mpptr;
  typedef boost::shared_ptr<Map> mapptr;
  // identifiers and BA bookkeeping (KeyFrame.h:282-312)
  idpair mId = defpair;
  size_t mUniqueId = 0;
  idpair mBALocalForKF = defpair, mBAFixedForKF = defpair, mBAGlobalForKF = defpair;
  cv::Mat mTcwGBA, mTcwBefGBA;
  bool mbUpdatedByServer = false;   // :129
  // keypoints and scale pyramid (:322-345)
  int N = 0;
  std::vector<cv::KeyPoint> mvKeysUn;
  int mnScaleLevels = 8;
  float mfScaleFactor = 1.2f, mfLogScaleFactor = 0.f;
  std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  cv::Mat mK;
  // pose (KeyFrame.cpp:288-384): Tcw, Ow = -Rwc * tcw
  void SetPose(const cv::Mat& Tcw_, bool bLock, bool bIgnorePoseMutex = false) {
    (void)bLock; (void)bIgnorePoseMutex;
    Tcw_.copyTo(Tcw);
    cv::Mat Rcw = Tcw.rowRange(0, 3).colRange(0, 3);
    cv::Mat tcw = Tcw.rowRange(0, 3).col(3);
    cv::Mat Rwc = Rcw.t();
    Ow = -Rwc * tcw;
  }
  cv::Mat GetPose() { return Tcw.clone(); }
  cv::Mat GetCameraCenter() { return Ow.clone(); }
  cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
  cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
  bool isBad() { return mbBad; }
  // observations (:390, KeyFrame.cpp:500-530, 611-615)
  std::vector<mpptr> mvpMapPoints;
  std::vector<mpptr> GetMapPointMatches() { return mvpMapPoints; }
  void EraseMapPointMatch(mpptr pMP, bool bLock = false);
  mpptr GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }                                                   // KeyFrame.cpp:617-621
  std::set<mpptr> GetMapPoints();                                                                                         // :592-609
  void AddMapPoint(mpptr pMP, const size_t& idx, bool bLock = false) { (void)bLock; mvpMapPoints[idx] = pMP; }         // :478-498
  void RemapMapPointMatch(mpptr pMP, const size_t& idx_now, const size_t& idx_new) { mvpMapPoints[idx_now] = nullptr; mvpMapPoints[idx_new] = pMP; }
  // what the matcher reads besides the above (KeyFrame.h:287-345): descriptors, BoW feature vector, image bounds, the grid
  std::vector<cv::KeyPoint> mvKeys;
  cv::Mat mDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  int mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS;
  float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  std::vector<std::vector<std::vector<size_t> > > mGrid;
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r) const;   // body = KeyFrame.cpp:1162-1201 (ref_frame_excerpt.cpp)
  bool IsInImage(const float& x, const float& y) const;                                         // body = KeyFrame.cpp:1203-1206
  // covisibility graph / spanning tree / loop edges, as plain containers filled by the harness
  std::vector<kfptr> mvpOrderedConnectedKeyFrames;
  std::vector<int> mvOrderedWeights;
  std::map<kfptr, int> mConnectedKeyFrameWeights;
  idpair mFuseTargetForKF = defpair;         // KeyFrame.h:294 (LocalMapping::SearchInNeighbors marks its first-level fuse targets, Mapping.cpp:481-482)
  std::vector<kfptr> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
  std::vector<kfptr> GetBestCovisibilityKeyFrames(const int& N) {   // KeyFrame.cpp:443-451: the N strongest neighbours (all of them when there are fewer)
    if ((int)mvpOrderedConnectedKeyFrames.size() < N) return mvpOrderedConnectedKeyFrames;
    return std::vector<kfptr>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.begin() + N);
  }
  // KeyFrame.cpp (GetCovisiblesByWeight): the ordered list down to weight w — with the reference's own corner case: when NO neighbour is weaker than w,
  // upper_bound returns end() and the method returns an EMPTY list, not all neighbours.  (Round 5: found by running the shim on the reference's real KeyFrame.cpp,
  // tests/test_shim_real_gpu.py; the earlier look-alike returned every neighbour with weight >= w.)
  std::vector<kfptr> GetCovisiblesByWeight(const int& w) {
    if (mvpOrderedConnectedKeyFrames.empty()) return std::vector<kfptr>();
    std::vector<int>::iterator it = std::upper_bound(mvOrderedWeights.begin(), mvOrderedWeights.end(), w, [](int a, int b) { return a > b; });
    if (it == mvOrderedWeights.end()) return std::vector<kfptr>();
    return std::vector<kfptr>(mvpOrderedConnectedKeyFrames.begin(), mvpOrderedConnectedKeyFrames.begin() + (it - mvOrderedWeights.begin()));
  }
  int GetWeight(kfptr pKF) { auto it = mConnectedKeyFrameWeights.find(pKF); return it == mConnectedKeyFrameWeights.end() ? 0 : it->second; }
  kfptr mpParent;
  std::set<kfptr> mspChildrens, mspLoopEdges;
  kfptr GetParent() { return mpParent; }
  bool hasChild(kfptr pKF) { return mspChildrens.count(pKF) != 0; }
  std::set<kfptr> GetLoopEdges() { return mspLoopEdges; }
  bool mbBad = false;
 protected:
  cv::Mat Tcw, Ow;
};

class MapPoint : public boost::enable_shared_from_this<MapPoint> {
 public:
  typedef boost::shared_ptr<KeyFrame> kfptr;
  typedef boost::shared_ptr<MapPoint> mpptr;
  typedef boost::shared_ptr<Map> mapptr;
  idpair mId = defpair;                      // MapPoint.h:217-250
  size_t mUniqueId = 0;
  idpair mBALocalForKF = defpair, mBAGlobalForKF = defpair;
  idpair mCorrectedByKF_LC = defpair, mCorrectedByKF_MM = defpair;
  size_t mCorrectedReference_LC = 0, mCorrectedReference_MM = 0;
  cv::Mat mPosGBA;
  bool mbUpdatedByServer = false;
  // MapPoint.cpp:338-363: a point whose position was locked (by the server's corrections) ignores later writes on a CLIENT; bLock locks it
  void SetWorldPos(const cv::Mat& Pos, bool bLock, bool bIgnorePosMutex = false) {
    (void)bIgnorePosMutex;
    if (mbPoseLock && mSysState == eSystemState::CLIENT) return;
    Pos.copyTo(mWorldPos);
    if (bLock) mbPoseLock = true;
  }
  bool IsPosLocked() { return mbPoseLock; }                                                                                                         // MapPoint.h:135
  bool mbPoseLock = false;                                                                                                                          // MapPoint.h:277
  eSystemState mSysState = eSystemState::CLIENT;
  cv::Mat GetWorldPos() { return mWorldPos.clone(); }                                                                                              // :393-397
  cv::Mat GetNormal() { return mNormalVector.clone(); }
  kfptr GetReferenceKeyFrame() { return mpRefKF; }
  std::map<kfptr, size_t> GetObservations() { return mObservations; }                                                                              // :511-515
  int GetIndexInKeyFrame(kfptr pKF, bool = false) { auto it = mObservations.find(pKF); return it == mObservations.end() ? -1 : (int)it->second; }   // :746-753
  void EraseObservation(kfptr pKF, bool bLock = false, bool bSuppressMapAction = false);   // :442-509 (reduced: no map / communication side effects)
  bool isBad() { return mbBad; }
  typedef boost::shared_ptr<Frame> frameptr;
  int Observations() { return nObs; }                                                                                  // MapPoint.cpp:517-521
  void AddObservation(kfptr pKF, size_t idx, bool bLock = false) { (void)bLock; if (mObservations.count(pKF)) return; mObservations[pKF] = idx; nObs++; }   // :399-440
  bool IsInKeyFrame(kfptr pKF) { return mObservations.count(pKF) != 0; }                                               // :755-759
  void Replace(mpptr pMP, bool bLock = false);                                                                         // :560-640, reduced (see below)
  bool mbDoNotReplace = false;
  void ComputeDistinctiveDescriptors();                                                                               // :929-994, the reference's own lines (ref_frame_excerpt.cpp)
  cv::Mat GetDescriptor() { return mDescriptor.clone(); }                                                             // :740-744
  cv::Mat mDescriptor;
  int PredictScale(const float& currentDist, kfptr pKF);      // bodies = MapPoint.cpp:836-869 (oracle/ref_mappoint_excerpt.cpp)
  int PredictScale(const float& currentDist, frameptr pF);
  // tracking scratch written by Frame::isInFrustum, read by SearchByProjection (MapPoint.h:224-228)
  float mTrackProjX = 0, mTrackProjY = 0, mTrackViewCos = 0;
  bool mbTrackInView = false;
  int mnTrackScaleLevel = 0;
  mpptr mpReplaced;
  void UpdateNormalAndDepth();   // body = the reference's own lines MapPoint.cpp:779-823 (oracle/ref_mappoint_excerpt.cpp)
#ifdef CCM_LOOKALIKE_MAPPOINT_SETTER
  // the OPTIONAL three-line patch of INTEGRATION.md (not in the reference): with it shim/Optimizer_hip.cpp replaces the per-point UpdateNormalAndDepth()
  // calls of a global BA's write-back by one batched device call
  void SetNormalAndDepth(const cv::Mat& normal, float minDist, float maxDist) { std::unique_lock<std::mutex> l(mMutexPos); normal.copyTo(mNormalVector); mfMinDistance = minDist; mfMaxDistance = maxDist; }
#endif
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  // state read by the excerpt above (names as in MapPoint.h:268-300)
  cv::Mat mWorldPos, mNormalVector;
  std::map<kfptr, size_t> mObservations;
  kfptr mpRefKF;
  float mfMinDistance = 0, mfMaxDistance = 0;
  bool mbBad = false;
  int nObs = 0;
  std::mutex mMutexFeatures, mMutexPos;
  static std::mutex mGlobalMutex;   // MapPoint.h:253
};

inline std::set<KeyFrame::mpptr> KeyFrame::GetMapPoints() { std::set<mpptr> s; for (auto& p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
inline void KeyFrame::EraseMapPointMatch(mpptr pMP, bool) {   // KeyFrame.cpp:518-530
  int idx = pMP->GetIndexInKeyFrame(shared_from_this());
  if (idx >= 0) mvpMapPoints[idx] = nullptr;
}
inline void MapPoint::Replace(mpptr pMP, bool) {   // MapPoint.cpp:560-640 without the map / communication side effects: this point's observations move to pMP
  if (pMP.get() == this) return;
  std::map<kfptr, size_t> obs = mObservations;
  mObservations.clear(); mbBad = true; mpReplaced = pMP;
  for (auto& kv : obs) {
    kfptr pKF = kv.first;
    if (!pMP->IsInKeyFrame(pKF)) { pKF->mvpMapPoints[kv.second] = pMP; pMP->AddObservation(pKF, kv.second); }
    else pKF->mvpMapPoints[kv.second] = nullptr;
  }
}
inline void MapPoint::EraseObservation(kfptr pKF, bool, bool) {   // MapPoint.cpp:442-509: drop the observation; a new reference keyframe if it was this one;
  bool bBad = false;                                              // the point turns bad when two or fewer observers remain
  if (mObservations.count(pKF)) {
    nObs--;
    mObservations.erase(pKF);
    if (mpRefKF == pKF && !mObservations.empty()) mpRefKF = mObservations.begin()->first;
    if (nObs <= 2) bBad = true;
  }
  if (bBad) mbBad = true;   // SetBadFlag: the harness does not model the map clean-up
}

class Map : public boost::enable_shared_from_this<Map> {
 public:
  typedef boost::shared_ptr<KeyFrame> kfptr;
  typedef boost::shared_ptr<MapPoint> mpptr;
  std::set<size_t> msuAssClients;            // Map.h:93-100
  size_t mMapId = 0;
  std::vector<kfptr> mvpKeyFrameOrigins;     // :163
  std::vector<kfptr> mvpKeyFrames;
  std::vector<mpptr> mvpMapPoints;
  std::vector<kfptr> GetAllKeyFrames() { return mvpKeyFrames; }
  std::vector<mpptr> GetAllMapPoints() { return mvpMapPoints; }
  long unsigned int GetMaxKFidUnique() { long unsigned int m = 0; for (auto& k : mvpKeyFrames) m = std::max<long unsigned int>(m, k->mUniqueId); return m; }
  mpptr GetMpPtr(size_t MpId, size_t ClientId) { for (auto& p : mvpMapPoints) if (p->mId.first == MpId && p->mId.second == ClientId && !p->isBad()) return p; return nullptr; }
  mpptr GetMpPtr(idpair id) { return GetMpPtr(id.first, id.second); }
  bool LockMapUpdate() { std::unique_lock<std::mutex> lock(mMutexMapUpdate); if (!mbLockMapUpdate) { mbLockMapUpdate = true; return true; } return false; }   // :175
  void UnLockMapUpdate() { std::unique_lock<std::mutex> lock(mMutexMapUpdate); mbLockMapUpdate = false; }
 private:
  std::mutex mMutexMapUpdate;
  bool mbLockMapUpdate = false;
};

class Frame : public boost::enable_shared_from_this<Frame> {
 public:
  typedef boost::shared_ptr<MapPoint> mpptr;
  typedef boost::shared_ptr<KeyFrame> kfptr;
  int N = 0;                                  // Frame.h:125-168
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  cv::Mat mDescriptors;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  // grid (Frame.h:100-102, 147-149, 171-174); the three function bodies are the reference's own lines (oracle/ref_frame_excerpt.cpp)
  static float mfGridElementWidthInv, mfGridElementHeightInv;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
  std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
  void AssignFeaturesToGrid();                                                                                    // Frame.cpp:103-118
  bool PosInGrid(const cv::KeyPoint& kp, int& posX, int& posY);                                                  // :254-265
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const;   // :200-253
  int mnScaleLevels = 8;
  float mfScaleFactor = 1.2f, mfLogScaleFactor = 0.f;
  std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2;
  std::vector<mpptr> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mTcw;
  idpair mId = defpair;
  std::vector<float> mvInvLevelSigma2;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  cv::Mat mRcw, mRwc, mtcw, mOw;
  void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); UpdatePoseMatrices(); }   // Frame.cpp:125-129
  void UpdatePoseMatrices();                                                // :131-137, the reference's own lines (oracle/ref_frame_excerpt.cpp)
  bool isInFrustum(mpptr pMP, float viewingCosLimit);                       // :139-198, likewise
};

}  // namespace cslam
