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
  // covisibility graph / spanning tree / loop edges, as plain containers filled by the harness
  std::vector<kfptr> mvpOrderedConnectedKeyFrames;
  std::vector<int> mvOrderedWeights;
  std::map<kfptr, int> mConnectedKeyFrameWeights;
  std::vector<kfptr> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
  std::vector<kfptr> GetCovisiblesByWeight(const int& w) {   // KeyFrame.cpp: the ordered list down to weight w
    std::vector<kfptr> out;
    for (size_t i = 0; i < mvpOrderedConnectedKeyFrames.size(); i++) if (mvOrderedWeights[i] >= w) out.push_back(mvpOrderedConnectedKeyFrames[i]);
    return out;
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
  void SetWorldPos(const cv::Mat& Pos, bool bLock, bool bIgnorePosMutex = false) { (void)bLock; (void)bIgnorePosMutex; Pos.copyTo(mWorldPos); }   // MapPoint.cpp:338-363
  cv::Mat GetWorldPos() { return mWorldPos.clone(); }                                                                                              // :393-397
  cv::Mat GetNormal() { return mNormalVector.clone(); }
  kfptr GetReferenceKeyFrame() { return mpRefKF; }
  std::map<kfptr, size_t> GetObservations() { return mObservations; }                                                                              // :511-515
  int GetIndexInKeyFrame(kfptr pKF, bool = false) { auto it = mObservations.find(pKF); return it == mObservations.end() ? -1 : (int)it->second; }   // :746-753
  void EraseObservation(kfptr pKF, bool bLock = false, bool bSuppressMapAction = false);   // :442-509 (reduced: no map / communication side effects)
  bool isBad() { return mbBad; }
  void UpdateNormalAndDepth();   // body = the reference's own lines MapPoint.cpp:779-823 (oracle/ref_mappoint_excerpt.cpp)
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

inline void KeyFrame::EraseMapPointMatch(mpptr pMP, bool) {   // KeyFrame.cpp:518-530
  int idx = pMP->GetIndexInKeyFrame(shared_from_this());
  if (idx >= 0) mvpMapPoints[idx] = nullptr;
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

class Frame {
 public:
  typedef boost::shared_ptr<MapPoint> mpptr;
  int N = 0;                                  // Frame.h:125-168
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<mpptr> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mTcw;
  idpair mId = defpair;
  std::vector<float> mvInvLevelSigma2;
  float fx = 0, fy = 0, cx = 0, cy = 0;
  void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }   // Frame.cpp:125-129 (UpdatePoseMatrices omitted)
};

}  // namespace cslam
