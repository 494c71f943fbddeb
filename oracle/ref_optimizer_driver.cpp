// ref_optimizer_driver.cpp — TEST INFRASTRUCTURE.  Builds a synthetic Map / KeyFrame / MapPoint / Frame object graph (the look-alike
// classes of oracle/ref_shim/cslam_lookalike) from flat arrays, calls the static methods of cslam::Optimizer THROUGH THE REFERENCE'S OWN
// HEADER (cslam/include/cslam/Optimizer.h) and reads the graph back.  The same file is linked twice:
//   oracle/_ref/liboptimizer_ref.so        with the reference's cslam/src/Optimizer.cpp + Converter.cc + g2o, all compiled verbatim
//                                          (oracle/Makefile.ref) — the reference behaviour
//   shim/liboptimizer_hip_shim.so          with OUR drop-in shim/Optimizer_hip.cpp (-> libccm_hip.so on the MI355X)
// so that tests/test_ref_optimizer.py / tests/test_shim_gpu.py can hand both the identical map and compare what they leave behind.
//   shim/liboptimizer_hip_shim_real.so     (round 5, -DCCM_REAL_CLASSES) with OUR shim/Optimizer_hip.cpp AND the reference's own cslam/src/{KeyFrame,MapPoint,Map,
//                                          Frame}.cpp + Converter.cc compiled as they are against the reference's REAL headers: the object graph below is
//                                          then made of the reference's real classes (their mutexes, their accessors, their std::map<idpair, ...> containers)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <cstddef>
#include <set>
#include <vector>

#ifdef CCM_REAL_CLASSES
// The real KeyFrame / MapPoint / Map keep their state protected and are only constructible through the tracking pipeline, ROS messages or the save / load
// path; a TEST HARNESS may cheat: every header of the standard library and of the third-party look-alikes is included FIRST (so that the keyword games below
// never reach them), then the reference's class headers are read with their members made accessible.  Layout and code of the classes are untouched — the
// member functions that run are the ones compiled from the reference's .cpp files.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <mutex>
#include <numeric>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <boost/shared_ptr.hpp>
#include <opencv2/opencv.hpp>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <ros/ros.h>
#include <cslam/config.h>
#include <cslam/estd.h>
#include <cslam/Datatypes.h>
#include <cslam/Converter.h>
#define private public
#define protected public
#include <cslam/KeyFrame.h>
#include <cslam/MapPoint.h>
#include <cslam/Map.h>
#include <cslam/Frame.h>
#undef private
#undef protected
#endif

#include <cslam/Optimizer.h>

#ifndef CCM_REAL_CLASSES
namespace cslam { std::mutex MapPoint::mGlobalMutex; }
#endif

namespace {
using cslam::Frame;
using cslam::KeyFrame;
using cslam::Map;
using cslam::MapPoint;
typedef boost::shared_ptr<KeyFrame> kfptr;
typedef boost::shared_ptr<MapPoint> mpptr;

#ifdef CCM_REAL_CLASSES
// contiguous raw storage, objects placed with the reference's save / load constructors (KeyFrame.cpp:33-49, MapPoint.cpp:32-46) and destroyed in place
template <class T> struct Store {
  T* p = nullptr; size_t n = 0, made = 0;
  void alloc(size_t k) { n = k; p = static_cast<T*>(::operator new(sizeof(T) * (k ? k : 1))); static_assert(alignof(T) <= alignof(std::max_align_t), "plain operator new is aligned enough"); }
  T& operator[](size_t i) { return p[i]; }
  ~Store() { for (size_t i = made; i-- > 0;) p[i].~T(); if (p) ::operator delete(p); }
};
#endif
struct MapG {
  // contiguous stores: shared_ptr ordering (std::map<kfptr, ...> iteration, std::set<kfptr>) is ADDRESS order = index order here
  // (declared first = destroyed last: the shared_ptr vectors and the map below only refer to them)
#ifdef CCM_REAL_CLASSES
  Store<KeyFrame> kf_store;
  Store<MapPoint> mp_store;
  boost::shared_ptr<cslam::Communicator> comm;   // never dereferenced: the three Communicator methods the map classes call are no-ops of the harness (real_graph_support.cpp)
#else
  std::unique_ptr<KeyFrame[]> kf_store;
  std::unique_ptr<MapPoint[]> mp_store;
#endif
  std::vector<kfptr> kfs;
  std::vector<mpptr> mps;
  boost::shared_ptr<Map> map;
  std::vector<int> obs_mp, obs_kf;
#ifdef CCM_REAL_CLASSES
  ~MapG() {   // break the pointer cycles of the real graph before the stores go (keyframes hold their points and neighbours, points their observers, the map both)
    for (size_t k = 0; k < kf_store.made; k++) { KeyFrame& kf = kf_store[k]; kf.mvpMapPoints.clear(); kf.mConnectedKeyFrameWeights.clear(); kf.mvpOrderedConnectedKeyFrames.clear(); kf.mpParent.reset(); kf.mspChildrens.clear(); kf.mspLoopEdges.clear(); kf.mspComm.clear(); kf.mpMap.reset(); }
    for (size_t p = 0; p < mp_store.made; p++) { MapPoint& mp = mp_store[p]; mp.mObservations.clear(); mp.mpRefKF.reset(); mp.mspComm.clear(); mp.mpMap.reset(); }
    if (map) { map->mmpKeyFrames.clear(); map->mmpMapPoints.clear(); map->mvpKeyFrameOrigins.clear(); map->mspComm.clear(); }
    kfs.clear(); mps.clear(); map.reset();
  }
#endif
};

cv::Mat mat44(const float* T) { cv::Mat m(4, 4, CV_32F); std::memcpy(m.data, T, 64); return m; }
}  // namespace

extern "C" {

// kf_*: per keyframe; keypoints of keyframe k are kp_xy/kp_oct[kp_off[k] .. kp_off[k+1]); observation o: map point obs_mp[o] is seen by
// keyframe obs_kf[o] at keypoint index obs_kp[o] (AddObservation order = o); covisibility weight = shared map points, connected when >= cov_th
void* mapg_create(int n_kf, const int32_t* kf_id, const int32_t* kf_client, const int32_t* kf_uid, const float* kf_Tcw, const uint8_t* kf_bad,
                  const float* K4, const int32_t* kp_off, const float* kp_xy, const int32_t* kp_oct, int n_mp, const int32_t* mp_id,
                  const int32_t* mp_client, const int32_t* mp_uid, const float* mp_pos, const uint8_t* mp_bad, int n_obs, const int32_t* obs_mp,
                  const int32_t* obs_kf, const int32_t* obs_kp, int n_levels, float scale_factor, int map_id, int cov_th) {
  MapG* g = new MapG();
#ifdef CCM_REAL_CLASSES
  // the map as a SERVER map (the state MapFusionGBA runs in; mapg_lock_points switches the points of a client test), one communicator pointer like every live map has
  g->comm = boost::shared_ptr<cslam::Communicator>(static_cast<cslam::Communicator*>(::operator new(64)), [](cslam::Communicator* c) { ::operator delete(c); });
  g->map.reset(new Map(ros::NodeHandle(), ros::NodeHandle(), (size_t)map_id, cslam::eSystemState::SERVER));
  g->map->mspComm.insert(g->comm);
  g->kf_store.alloc((size_t)n_kf);
  g->mp_store.alloc((size_t)n_mp);
  for (int k = 0; k < n_kf; k++) { new (&g->kf_store[k]) KeyFrame(cslam::vocptr(), g->map, KeyFrame::dbptr(), g->comm, cslam::eSystemState::SERVER, (size_t)kf_uid[k]); g->kf_store.made++; }
  for (int p = 0; p < n_mp; p++) { new (&g->mp_store[p]) MapPoint(g->map, g->comm, cslam::eSystemState::SERVER, (size_t)mp_uid[p]); g->mp_store.made++; }
#else
  g->kf_store.reset(new KeyFrame[n_kf]);
  g->mp_store.reset(new MapPoint[n_mp]);
  g->map.reset(new Map());
  g->map->mMapId = (size_t)map_id;
#endif
  std::vector<float> sf(n_levels), s2(n_levels), is2(n_levels);
  sf[0] = 1.0f; s2[0] = 1.0f;
  for (int i = 1; i < n_levels; i++) { sf[i] = sf[i - 1] * scale_factor; s2[i] = sf[i] * sf[i]; }   // ORBextractor.cpp:586-591
  for (int i = 0; i < n_levels; i++) is2[i] = 1.0f / s2[i];
  for (int k = 0; k < n_kf; k++) {
    KeyFrame& kf = g->kf_store[k];
    kf.mId = std::make_pair((size_t)kf_id[k], (size_t)kf_client[k]);
    kf.mUniqueId = (size_t)kf_uid[k];
    kf.mbBad = kf_bad[k] != 0;
    kf.fx = K4[0]; kf.fy = K4[1]; kf.cx = K4[2]; kf.cy = K4[3];
    kf.mK = cv::Mat::eye(3, 3, CV_32F);
    kf.mK.at<float>(0, 0) = K4[0]; kf.mK.at<float>(1, 1) = K4[1]; kf.mK.at<float>(0, 2) = K4[2]; kf.mK.at<float>(1, 2) = K4[3];
    kf.mnScaleLevels = n_levels; kf.mfScaleFactor = scale_factor; kf.mfLogScaleFactor = std::log(scale_factor);
    kf.mvScaleFactors = sf; kf.mvLevelSigma2 = s2; kf.mvInvLevelSigma2 = is2;
    kf.N = kp_off[k + 1] - kp_off[k];
    kf.mvKeysUn.resize(kf.N);
    for (int i = 0; i < kf.N; i++) { const int s = kp_off[k] + i; kf.mvKeysUn[i] = cv::KeyPoint(kp_xy[2 * s], kp_xy[2 * s + 1], 31.f * sf[kp_oct[s]], -1, 0, kp_oct[s]); }
    kf.mvpMapPoints.assign(kf.N, mpptr());
#ifdef CCM_REAL_CLASSES
    kf.mvbMapPointsLock.assign(kf.N, false);                           // (what the tracking-side constructor sizes with mvpMapPoints, KeyFrame.cpp:74-77)
    kf.mHalfBaseline = 0; kf.invfx = 1.0f / kf.fx; kf.invfy = 1.0f / kf.fy; kf.mbOmitSending = true;   // (nothing to publish while the graph is being built)
    g->kfs.push_back(kfptr(&kf, [](KeyFrame*) {}));   // before SetPose: the real method may ask for shared_from_this()
    kf.SetPose(mat44(kf_Tcw + 16 * (size_t)k), false, false);
    kf.mbOmitSending = false;
#else
    kf.SetPose(mat44(kf_Tcw + 16 * (size_t)k), false);
    g->kfs.push_back(kfptr(&kf, [](KeyFrame*) {}));
#endif
    g->map->msuAssClients.insert((size_t)kf_client[k]);
  }
  for (int p = 0; p < n_mp; p++) {
    MapPoint& mp = g->mp_store[p];
    mp.mId = std::make_pair((size_t)mp_id[p], (size_t)mp_client[p]);
    mp.mUniqueId = (size_t)mp_uid[p];
    mp.mbBad = mp_bad[p] != 0;
    cv::Mat pos(3, 1, CV_32F);
    for (int c = 0; c < 3; c++) pos.at<float>(c) = mp_pos[3 * (size_t)p + c];
#ifdef CCM_REAL_CLASSES
    g->mps.push_back(mpptr(&mp, [](MapPoint*) {}));
    mp.mbOmitSending = true; mp.mfMinDistance = 0; mp.mfMaxDistance = 0; mp.mnVisible = 1; mp.mnFound = 1;
    mp.SetWorldPos(pos, false, false);
    mp.mbOmitSending = false;
    mp.mNormalVector = cv::Mat::zeros(3, 1, CV_32F);
#else
    mp.SetWorldPos(pos, false);
    mp.mNormalVector = cv::Mat::zeros(3, 1, CV_32F);
    g->mps.push_back(mpptr(&mp, [](MapPoint*) {}));
#endif
  }
  g->obs_mp.assign(obs_mp, obs_mp + n_obs); g->obs_kf.assign(obs_kf, obs_kf + n_obs);
  std::vector<std::map<int, int>> shared(n_kf);   // covisibility weights
  std::vector<std::vector<int>> seen_by(n_mp);
  for (int o = 0; o < n_obs; o++) {
    MapPoint& mp = g->mp_store[obs_mp[o]];
    KeyFrame& kf = g->kf_store[obs_kf[o]];
    if (!mp.mpRefKF) mp.mpRefKF = g->kfs[obs_kf[o]];                  // the creating keyframe
    mp.mObservations[g->kfs[obs_kf[o]]] = (size_t)obs_kp[o];        // AddObservation
    mp.nObs++;
    kf.mvpMapPoints[obs_kp[o]] = g->mps[obs_mp[o]];                 // AddMapPoint
    seen_by[obs_mp[o]].push_back(obs_kf[o]);
  }
  for (int p = 0; p < n_mp; p++) for (int a : seen_by[p]) for (int b : seen_by[p]) if (a != b) shared[a][b]++;
  for (int k = 0; k < n_kf; k++) {   // KeyFrame::UpdateConnections: neighbours with >= cov_th shared points, best first
    std::vector<std::pair<int, int>> v;
    for (auto& kv : shared[k]) if (kv.second >= cov_th) v.push_back({kv.second, kv.first});
    std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
    KeyFrame& kf = g->kf_store[k];
    for (auto& e : v) { kf.mvpOrderedConnectedKeyFrames.push_back(g->kfs[e.second]); kf.mvOrderedWeights.push_back(e.first); kf.mConnectedKeyFrameWeights[g->kfs[e.second]] = e.first; }
    if (k > 0 && kf_client[k] == kf_client[k - 1]) { kf.mpParent = g->kfs[k - 1]; g->kf_store[k - 1].mspChildrens.insert(g->kfs[k]); }
  }
#ifdef CCM_REAL_CLASSES
  // what Map::AddKeyFrame / AddMapPoint do on a server map (Map.cpp), minus the communicator hand-over: file the object under its id, keep the maxima
  for (int k = 0; k < n_kf; k++) {
    KeyFrame& kf = g->kf_store[k];
    g->map->mmpKeyFrames[kf.mId] = g->kfs[k];
    if (kf.mId.first > g->map->mnMaxKFid) g->map->mnMaxKFid = kf.mId.first;
    if (kf.mUniqueId > g->map->mnMaxKFidUnique) g->map->mnMaxKFidUnique = kf.mUniqueId;
    g->map->mnLastKfIdUnique = kf.mUniqueId;
  }
  for (int p = 0; p < n_mp; p++) {
    MapPoint& mp = g->mp_store[p];
    g->map->mmpMapPoints[mp.mId] = g->mps[p];
    if (mp.mId.first > g->map->mnMaxMPid) g->map->mnMaxMPid = mp.mId.first;
    if (mp.mUniqueId > g->map->mnMaxMPidUnique) g->map->mnMaxMPidUnique = mp.mUniqueId;
  }
#else
  g->map->mvpKeyFrames = g->kfs;
  g->map->mvpMapPoints = g->mps;
#endif
  if (n_kf) g->map->mvpKeyFrameOrigins.push_back(g->kfs[0]);
  for (int p = 0; p < n_mp; p++) g->mp_store[p].UpdateNormalAndDepth();   // as after point creation (Mapping.cpp)
  return g;
}
void mapg_destroy(void* h) { delete (MapG*)h; }
// MapPoint::mbPoseLock / mSysState of every point (MapPoint.h:277): locked[p] != 0 marks a position the server has fixed; on a CLIENT (server_state == 0)
// SetWorldPos then returns without writing (MapPoint.cpp:340-341)
void mapg_lock_points(void* h, const uint8_t* locked, int server_state) {
  MapG* g = (MapG*)h;
  for (size_t p = 0; p < g->mps.size(); p++) { g->mp_store[p].mbPoseLock = locked[p] != 0; g->mp_store[p].mSysState = server_state ? cslam::eSystemState::SERVER : cslam::eSystemState::CLIENT; }
}

// cslam::Optimizer::LocalBundleAdjustmentClient(pKF, pbStopFlag, pMap, ClientId, SysState)   (Optimizer.h:84-86)
int mapg_local_ba(void* h, int kf_index, int client_id, int server_state, uint8_t* stop_flag) {
  MapG* g = (MapG*)h;
  try {
    cslam::Optimizer::LocalBundleAdjustmentClient(g->kfs[kf_index], (bool*)stop_flag, g->map, (size_t)client_id,
                                                  server_state ? cslam::eSystemState::SERVER : cslam::eSystemState::CLIENT);
  } catch (std::exception& e) { return -1; }
  return 0;
}
// cslam::Optimizer::MapFusionGBA(pMap, ClientId, nIterations, pbStopFlag, nLoopKF, bRobust)   (Optimizer.h:93-94)
int mapg_map_fusion_gba(void* h, int client_id, int n_iterations, uint8_t* stop_flag, int loop_first, int loop_second, int robust) {
  MapG* g = (MapG*)h;
  try {
    cslam::Optimizer::MapFusionGBA(g->map, (size_t)client_id, n_iterations, (bool*)stop_flag, std::make_pair((size_t)loop_first, (size_t)loop_second), robust != 0);
  } catch (std::exception& e) { return -1; }
  return 0;
}
// cslam::Optimizer::BundleAdjustmentClient(vpKF, vpMP, ClientId, nIterations, pbStopFlag, nLoopKF, bRobust)   (Optimizer.h:79-81)
int mapg_bundle_adjustment_client(void* h, int client_id, int n_iterations, uint8_t* stop_flag, int robust) {
  MapG* g = (MapG*)h;
  try {
    cslam::Optimizer::BundleAdjustmentClient(g->kfs, g->mps, (size_t)client_id, n_iterations, (bool*)stop_flag, std::make_pair((size_t)0, (size_t)0), robust != 0);
  } catch (std::exception& e) { return -1; }
  return 0;
}

// state after the call.  Any pointer may be NULL.  kf_gba / mp_gba: mTcwGBA / mPosGBA when the call filled them (else zeros), *_flag says so.
void mapg_get_state(void* h, float* kf_Tcw, float* kf_gba, uint8_t* kf_gba_flag, float* mp_pos, float* mp_gba, uint8_t* mp_gba_flag, uint8_t* mp_bad,
                    float* mp_normal, float* mp_dmin, float* mp_dmax, uint8_t* obs_alive) {
  MapG* g = (MapG*)h;
  for (size_t k = 0; k < g->kfs.size(); k++) {
    if (kf_Tcw) { cv::Mat T = g->kfs[k]->GetPose(); std::memcpy(kf_Tcw + 16 * k, T.data, 64); }
    const bool has = !g->kfs[k]->mTcwGBA.empty();
    if (kf_gba_flag) kf_gba_flag[k] = has;
    if (kf_gba) { if (has) { cv::Mat T = g->kfs[k]->mTcwGBA.clone(); std::memcpy(kf_gba + 16 * k, T.data, 64); } else std::memset(kf_gba + 16 * k, 0, 64); }
  }
  for (size_t p = 0; p < g->mps.size(); p++) {
    MapPoint& mp = *g->mps[p];
    if (mp_pos) for (int c = 0; c < 3; c++) mp_pos[3 * p + c] = mp.mWorldPos.at<float>(c);
    const bool has = !mp.mPosGBA.empty();
    if (mp_gba_flag) mp_gba_flag[p] = has;
    if (mp_gba) for (int c = 0; c < 3; c++) mp_gba[3 * p + c] = has ? mp.mPosGBA.at<float>(c) : 0.f;
    if (mp_bad) mp_bad[p] = mp.mbBad;
    if (mp_normal) for (int c = 0; c < 3; c++) mp_normal[3 * p + c] = mp.mNormalVector.at<float>(c);
    if (mp_dmin) mp_dmin[p] = mp.mfMinDistance;
    if (mp_dmax) mp_dmax[p] = mp.mfMaxDistance;
  }
  if (obs_alive) for (size_t o = 0; o < g->obs_mp.size(); o++) obs_alive[o] = g->mps[g->obs_mp[o]]->mObservations.count(g->kfs[g->obs_kf[o]]) ? 1 : 0;
}

// cslam::Optimizer::PoseOptimizationClient(Frame&)   (Optimizer.h:88): a Frame with n keypoints, each with a map point
int mapg_pose_optimization(float* Tcw, int n, const float* kp_xy, const int32_t* kp_oct, const float* mp_pos, const float* K4, int n_levels,
                           float scale_factor, uint8_t* outlier) {
#ifdef CCM_REAL_CLASSES
  // The real Frame has two constructors: a copy and the tracking one (Frame.cpp:56-101), which extracts ORB features from an image.  So the harness gives it an image:
  // a block texture that FAST finds corners in, run through the drop-in ORBextractor of this library (shim/ORBextractor_hip.cpp, on the MI355X), K with zero distortion
  // (UndistortKeyPoints returns at its first line, Frame.cpp:262-266).  What the constructor builds — scale tables, image bounds, the calibration statics, the grid — is
  // the reference's own code; the harness then replaces the OBSERVATION SET (keypoints, map points, outlier flags) by the test's, so that the same call can be made on
  // the look-alike Frame, and the pose goes in through the real Frame::SetPose.
  const int W = 752, H = 480;
  cv::Mat im(H, W, CV_8UC1);
  unsigned lcg = 12345u;
  for (int by = 0; by < H; by += 12)
    for (int bx = 0; bx < W; bx += 12) {
      lcg = lcg * 1664525u + 1013904223u;
      const unsigned char v = (unsigned char)(40 + (lcg >> 24) % 176);
      for (int y = by; y < by + 12 && y < H; y++) for (int x = bx; x < bx + 12 && x < W; x++) im.data[(size_t)y * W + x] = v;
    }
  cv::Mat Kc = cv::Mat::eye(3, 3, CV_32F);
  Kc.at<float>(0, 0) = K4[0]; Kc.at<float>(1, 1) = K4[1]; Kc.at<float>(0, 2) = K4[2]; Kc.at<float>(1, 2) = K4[3];
  cv::Mat dist = cv::Mat::zeros(4, 1, CV_32F);
  boost::shared_ptr<cslam::ORBextractor> ex(new cslam::ORBextractor(1000, scale_factor, n_levels, 20, 7));
  Frame::mbInitialComputations = true;
  Frame F(im, 0.0, ex, cslam::vocptr(), Kc, dist, (size_t)0);
  if (F.N <= 0 || (int)F.mvInvLevelSigma2.size() != n_levels) return -101;   // the extractor found nothing in the texture / the tables are not the extractor's
  const int n_extracted = F.N;
  (void)n_extracted;
  F.N = n;
  F.mvKeys.resize(n); F.mvKeysUn.resize(n); F.mvpMapPoints.assign(n, mpptr()); F.mvbOutlier.assign(n, false);
  boost::shared_ptr<cslam::Communicator> comm(static_cast<cslam::Communicator*>(::operator new(64)), [](cslam::Communicator* c) { ::operator delete(c); });
  boost::shared_ptr<Map> map(new Map(ros::NodeHandle(), ros::NodeHandle(), (size_t)0, cslam::eSystemState::CLIENT));
  map->mspComm.insert(comm);
  Store<MapPoint> store;
  store.alloc((size_t)n);
  for (int i = 0; i < n; i++) {
    F.mvKeys[i] = F.mvKeysUn[i] = cv::KeyPoint(kp_xy[2 * i], kp_xy[2 * i + 1], 31.f, -1, 0, kp_oct[i]);
    new (&store[i]) MapPoint(map, comm, cslam::eSystemState::CLIENT, (size_t)i);
    store.made++;
    MapPoint& mp = store[i];
    cv::Mat pos(3, 1, CV_32F);
    for (int c = 0; c < 3; c++) pos.at<float>(c) = mp_pos[3 * (size_t)i + c];
    F.mvpMapPoints[i] = mpptr(&mp, [](MapPoint*) {});
    mp.mbOmitSending = true;
    mp.SetWorldPos(pos, false, false);
    mp.mbOmitSending = false;
  }
  F.SetPose(mat44(Tcw));
  const int nin = cslam::Optimizer::PoseOptimizationClient(F);
  std::memcpy(Tcw, F.mTcw.data, 64);
  for (int i = 0; i < n; i++) outlier[i] = F.mvbOutlier[i] ? 1 : 0;
  F.mvpMapPoints.clear();
  for (size_t i = 0; i < store.made; i++) { store[i].mspComm.clear(); store[i].mpMap.reset(); }
  map->mspComm.clear();
  return nin;
#else
  Frame F;
  F.N = n;
  F.fx = K4[0]; F.fy = K4[1]; F.cx = K4[2]; F.cy = K4[3];
  F.mvInvLevelSigma2.resize(n_levels);
  float s = 1.0f;
  for (int i = 0; i < n_levels; i++) { F.mvInvLevelSigma2[i] = 1.0f / (s * s); s = s * scale_factor; }
  F.mTcw = mat44(Tcw);
  std::unique_ptr<MapPoint[]> store(new MapPoint[n > 0 ? n : 1]);
  F.mvKeysUn.resize(n); F.mvpMapPoints.resize(n); F.mvbOutlier.assign(n, false);
  for (int i = 0; i < n; i++) {
    F.mvKeysUn[i] = cv::KeyPoint(kp_xy[2 * i], kp_xy[2 * i + 1], 31.f, -1, 0, kp_oct[i]);
    cv::Mat pos(3, 1, CV_32F);
    for (int c = 0; c < 3; c++) pos.at<float>(c) = mp_pos[3 * (size_t)i + c];
    store[i].SetWorldPos(pos, false);
    F.mvpMapPoints[i] = mpptr(&store[i], [](MapPoint*) {});
  }
  const int nin = cslam::Optimizer::PoseOptimizationClient(F);
  std::memcpy(Tcw, F.mTcw.data, 64);
  for (int i = 0; i < n; i++) outlier[i] = F.mvbOutlier[i] ? 1 : 0;
  return nin;
#endif
}

namespace {
g2o::Sim3 sim3_of(const double* p) { return g2o::Sim3(Eigen::Quaterniond(p[3], p[0], p[1], p[2]), Eigen::Vector3d(p[4], p[5], p[6]), p[7]); }
void sim3_out(const g2o::Sim3& S, double* p) {
  p[0] = S.rotation().x(); p[1] = S.rotation().y(); p[2] = S.rotation().z(); p[3] = S.rotation().w();
  p[4] = S.translation()[0]; p[5] = S.translation()[1]; p[6] = S.translation()[2]; p[7] = S.scale();
}
}  // namespace

// cslam::Optimizer::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale)   (Optimizer.h:97-98).  match_mp[i] = index of the map point of
// keyframe 2 matched to keypoint i of keyframe 1 (-1: none); returns nIn, keep[i] = 0 where the call nulled vpMatches1[i]
int mapg_optimize_sim3(void* h, int kf1, int kf2, const int32_t* match_mp, double* s8, float th2, int fix_scale, uint8_t* keep) {
  MapG* g = (MapG*)h;
  std::vector<mpptr> vpMatches1((size_t)g->kfs[kf1]->N);
  for (int i = 0; i < g->kfs[kf1]->N; i++) vpMatches1[i] = match_mp[i] >= 0 ? g->mps[match_mp[i]] : mpptr();
  g2o::Sim3 S = sim3_of(s8);
  int nin = -1;
  try { nin = cslam::Optimizer::OptimizeSim3(g->kfs[kf1], g->kfs[kf2], vpMatches1, S, th2, fix_scale != 0); } catch (std::exception&) { return -1; }
  sim3_out(S, s8);
  for (int i = 0; i < g->kfs[kf1]->N; i++) keep[i] = vpMatches1[i] ? 1 : 0;
  return nin;
}

// cslam::Optimizer::OptimizeEssentialGraphLoopClosure / OptimizeEssentialGraphMapFusion   (Optimizer.h:101-110).  corrected / noncorrected: lists
// of (keyframe index, Sim3 as 8 doubles); loop connections: pairs (keyframe a -> keyframe b); loop edges of the spanning structure are set on the
// keyframes beforehand (loop_edge pairs, symmetric).  map_fusion != 0 selects the MapFusion variant (which takes no Sim3 maps).
int mapg_essential_graph(void* h, int map_fusion, int loop_kf, int cur_kf, int n_corr, const int32_t* corr_kf, const double* corr_s8, int n_non,
                         const int32_t* non_kf, const double* non_s8, int n_conn, const int32_t* conn_a, const int32_t* conn_b, int n_le, const int32_t* le_a,
                         const int32_t* le_b, int fix_scale) {
  MapG* g = (MapG*)h;
  cslam::Optimizer::KeyFrameAndPose Corrected, NonCorrected;
  for (int i = 0; i < n_corr; i++) Corrected[g->kfs[corr_kf[i]]] = sim3_of(corr_s8 + 8 * (size_t)i);
  for (int i = 0; i < n_non; i++) NonCorrected[g->kfs[non_kf[i]]] = sim3_of(non_s8 + 8 * (size_t)i);
  std::map<kfptr, std::set<kfptr> > LoopConnections;
  for (int i = 0; i < n_conn; i++) LoopConnections[g->kfs[conn_a[i]]].insert(g->kfs[conn_b[i]]);
  for (int i = 0; i < n_le; i++) { g->kfs[le_a[i]]->mspLoopEdges.insert(g->kfs[le_b[i]]); g->kfs[le_b[i]]->mspLoopEdges.insert(g->kfs[le_a[i]]); }
  const bool bFixScale = fix_scale != 0;
  try {
    if (map_fusion) cslam::Optimizer::OptimizeEssentialGraphMapFusion(g->map, g->kfs[loop_kf], g->kfs[cur_kf], LoopConnections, bFixScale);
    else cslam::Optimizer::OptimizeEssentialGraphLoopClosure(g->map, g->kfs[loop_kf], g->kfs[cur_kf], NonCorrected, Corrected, LoopConnections, bFixScale);
  } catch (std::exception&) { return -1; }
  return 0;
}

// cslam::Converter (Converter.cc:40-119) on plain arrays — pins ccm_slam_amd/host/ccm_convert.h and the oracle's converters
void mapg_to_se3quat(const float* Tcw, double* qt7) {
  const g2o::SE3Quat T = cslam::Converter::toSE3Quat(mat44(Tcw));
  qt7[0] = T.rotation().x(); qt7[1] = T.rotation().y(); qt7[2] = T.rotation().z(); qt7[3] = T.rotation().w();
  for (int i = 0; i < 3; i++) qt7[4 + i] = T.translation()[i];
}
void mapg_se3quat_to_cvmat(const double* qt7, float* Tcw) {
  const g2o::SE3Quat T(Eigen::Quaterniond(qt7[3], qt7[0], qt7[1], qt7[2]), Eigen::Vector3d(qt7[4], qt7[5], qt7[6]));
  cv::Mat m = cslam::Converter::toCvMat(T);
  std::memcpy(Tcw, m.data, 64);
}
// MapPoint::UpdateNormalAndDepth of the reference (MapPoint.cpp:779-823) on every point of the graph
void mapg_update_normals(void* h) { MapG* g = (MapG*)h; for (auto& p : g->mps) p->UpdateNormalAndDepth(); }

}  // extern "C"
