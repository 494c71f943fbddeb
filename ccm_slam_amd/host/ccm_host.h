// ccm_host.h — host-side C++ mirror of the reference's class API for the hot path, on top of the C ABI
// (include/ccm_hip.h).  This is what replaces the three reference translation units
//   cslam/src/ORBextractor.cpp, cslam/src/ORBmatcher.cpp, cslam/src/Optimizer.cpp
// in a drop-in build (INTEGRATION.md shows the glue that walks the reference's shared_ptr graph).
//
// The reference's data model (Frame / KeyFrame / MapPoint / Map, OpenCV types) is OUT OF SCOPE and not
// available in this image, so the classes below take *views*: plain structs of pointers into the caller's
// arrays, holding exactly the fields the reference methods read.  With CCM_HAVE_OPENCV defined the
// ORBextractor additionally offers the cv::InputArray / cv::OutputArray operator() of the reference.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/ccm_hip.h"

namespace cslam {

// thrown where the reference throws estd::infrastructure_ex (cslam/include/cslam/estd.h:74-81)
struct infrastructure_ex : std::runtime_error { using std::runtime_error::runtime_error; };

// one device context per calling thread (SURVEY §8b threading)
class HipContext {
 public:
  explicit HipContext(int device = 0);
  ~HipContext();
  ccm_ctx* get() const { return ctx_; }
 private:
  ccm_ctx* ctx_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------
// ORBextractor — cslam/include/cslam/ORBextractor.h:103-138
// ---------------------------------------------------------------------------------------------------
struct KeyPoint { float x, y, size, angle, response; int octave; };   // cv::KeyPoint minus class_id

class ORBextractor {
 public:
  ORBextractor(HipContext& ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  ~ORBextractor();
  // operator()(image, mask /*ignored, as in the reference*/, keypoints, descriptors): 8-bit single-channel image
  void operator()(const uint8_t* image, int cols, int rows, int step, std::vector<KeyPoint>& keypoints,
                  std::vector<uint8_t>& descriptors /* N x 32, row-major like cv::Mat CV_8U */);
  int GetLevels() const { return nlevels_; }
  float GetScaleFactor() const { return scaleFactor_; }
  std::vector<float> GetScaleFactors() const { return table(0); }
  std::vector<float> GetInverseScaleFactors() const { return table(1); }
  std::vector<float> GetScaleSigmaSquares() const { return table(2); }
  std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
  // mvImagePyramid (public member of the reference): un-bordered levels of the last frame
  struct Level { int cols, rows; std::vector<uint8_t> data; };
  std::vector<Level> mvImagePyramid;
  bool keepPyramid = false;   // fill mvImagePyramid on every call (costs a D2H of ~1.1 MB)
 private:
  std::vector<float> table(int which) const;
  ccm_orb* orb_ = nullptr;
  int nlevels_; float scaleFactor_;
};

// ---------------------------------------------------------------------------------------------------
// ORBmatcher — cslam/include/cslam/ORBmatcher.h:100-139
// ---------------------------------------------------------------------------------------------------
// What SearchByProjection reads from a Frame (cslam/include/cslam/Frame.h:131-152)
struct FrameView {
  int N = 0;
  const KeyPoint* mvKeysUn = nullptr;       // undistorted keypoints (x, y, octave, angle)
  const uint8_t* mDescriptors = nullptr;    // N x 32
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  const float* mvScaleFactors = nullptr;    // nlevels
  // mvpMapPoints as indices: -1 = no map point; >= 0 = a map point with Observations() > 0 ("claimed")
  int32_t* mvpMapPoints = nullptr;          // in/out, length N
};
// What the loop reads from each candidate map point (MapPoint.h:224-228,289)
struct TrackedMapPoints {
  int n = 0;
  const uint8_t* mbTrackInView = nullptr;   // && !isBad()
  const float* mTrackProjX = nullptr; const float* mTrackProjY = nullptr;
  const int32_t* mnTrackScaleLevel = nullptr;
  const float* mTrackViewCos = nullptr;
  const uint8_t* mDescriptor = nullptr;     // n x 32
};
// Last-frame side of SearchByProjection(Frame&, const Frame&, th): projections are computed by the caller
// (f32, ORBmatcher.cpp:1381-1397) — valid[i] = has map point && !outlier && in front && inside the image
struct LastFrameProjections {
  int n = 0;
  const uint8_t* valid = nullptr; const float* u = nullptr; const float* v = nullptr;
  const int32_t* octave = nullptr;          // LastFrame.mvKeys[i].octave
  const float* angle = nullptr;             // LastFrame.mvKeysUn[i].angle
  const uint8_t* mpDescriptor = nullptr;    // n x 32, pMP->GetDescriptor()
};

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened: ascending node ids + CSR of feature indices
struct FeatureVectorView { int nn = 0; const int32_t* node = nullptr; const int32_t* off = nullptr; const int32_t* idx = nullptr; };
// What the BoW / triangulation / initialisation searches read from a KeyFrame or Frame
struct KeysView {
  int N = 0;
  const KeyPoint* keys = nullptr;           // mvKeysUn
  const uint8_t* desc = nullptr;            // N x 32
  const uint8_t* hasMapPoint = nullptr;     // vpMapPoints[i] && !isBad()
  FeatureVectorView fv;
};

// Device-resident grid of one Frame / KeyFrame (ccm_frame_*): undistorted keypoints, bounds, 75x48 cells, descriptors.
// Replaces Frame::UndistortKeyPoints / ComputeImageBounds / AssignFeaturesToGrid (Frame.cpp:103-118, 284-347) and serves
// batched GetFeaturesInArea + Hamming queries to the matcher.
class FrameGridDev {
 public:
  FrameGridDev(HipContext& ctx, const float K[4], const float* distCoef, int nDist, int width, int height);
  ~FrameGridDev();
  FrameGridDev(const FrameGridDev&) = delete;
  FrameGridDev& operator=(const FrameGridDev&) = delete;
  // mvKeys + mDescriptors in; mvKeysUn out (same order), bounds available afterwards
  void SetKeyPoints(const std::vector<KeyPoint>& mvKeys, const uint8_t* mDescriptors, std::vector<KeyPoint>& mvKeysUn);
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  ccm_frame* get() const { return f_; }
 private:
  HipContext& ctx_;
  ccm_frame* f_ = nullptr;
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
  ORBmatcher(HipContext& ctx, float nnratio = 0.6f, bool checkOri = true) : ctx_(ctx), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  // ORBmatcher.cpp:1653-1669 (host popcount; the batched form is ccm_hamming_*)
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b);
  // ORBmatcher.cpp:71-148.  Returns nmatches; F.mvpMapPoints[idx] = index of the matched map point.
  int SearchByProjection(FrameView& F, const TrackedMapPoints& mps, float th);
  // same, with the candidate lists and distances produced on the device from the frame's resident grid (no host grid walk)
  int SearchByProjection(FrameGridDev& grid, FrameView& F, const TrackedMapPoints& mps, float th);
  int SearchByProjection(FrameGridDev& grid, FrameView& CurrentFrame, const LastFrameProjections& last, float th);
  // ORBmatcher.cpp:1350-1476.
  int SearchByProjection(FrameView& CurrentFrame, const LastFrameProjections& last, float th);
  // SearchByBoW(kfptr pKF, Frame& F, ...) — ORBmatcher.cpp:178-306.  matchesF[F.N]: KF feature whose map point goes to F[i], or -1
  int SearchByBoW(const KeysView& KF, const KeysView& F, std::vector<int32_t>& matchesF);
  // SearchByBoW(kfptr, kfptr, ...) — ORBmatcher.cpp:565-698.  matches12[KF1.N]: feature of KF2 or -1
  int SearchByBoW_KF(const KeysView& KF1, const KeysView& KF2, std::vector<int32_t>& matches12);
  // SearchForTriangulation — ORBmatcher.cpp:700-852.  F12 row-major 3x3 f32; (ex,ey) epipole in image 2 (:708-714)
  int SearchForTriangulation(const KeysView& KF1, const KeysView& KF2, const float F12[9], float ex, float ey, const float* sigma2_2,
                             const float* scaleFactors2, std::vector<int32_t>& matches12);
  // SearchForInitialization — ORBmatcher.cpp:448-563.  F2 needs the grid bounds; prevMatched (x,y pairs) is updated in place
  int SearchForInitialization(const KeysView& F1, const FrameView& F2, std::vector<float>& vbPrevMatched, std::vector<int32_t>& vnMatches12,
                              int windowSize = 10);
  // The projected window search shared by Fuse (:854-993, chi2 gate), Fuse with Sim3 (:995-1122), SearchByProjection(kfptr,
  // Scw, ...) (:308-446, skips and claims vpMatched) and both directions of SearchBySim3 (:1124-1348, TH_HIGH).  The caller
  // supplies the outcome of the f32 projection / depth / viewing-angle tests (valid, u, v, predicted level) and applies the
  // map mutations (Replace / AddObservation / RemapMapPointMatch) in order from bestIdx, as the reference does after its loop.
  struct ProjectedPoints { int n = 0; const uint8_t* valid = nullptr; const float* u = nullptr; const float* v = nullptr;
                           const int32_t* level = nullptr; const uint8_t* desc = nullptr; const uint8_t* noClaim = nullptr;
                           // optional: the window candidates of every point as the caller's KeyFrame::GetFeaturesInArea returned them (CSR over the n points,
                           // unfiltered, in that order).  A drop-in build has the reference's KeyFrame.cpp, whose grid was filled with the Frame's float
                           // bounds but is read with the keyframe's int-truncated ones (KeyFrame.cpp:54-61, 1167-1171): only its own lookup has that order.
                           const int32_t* candOff = nullptr; const int32_t* candIdx = nullptr; };
  int ProjectedSearch(const FrameView& KF, const float* invLevelSigma2, const ProjectedPoints& P, float th, bool chi2Gate, int distThreshold,
                      int32_t* matched /* nullable in/out [KF.N] */, bool claim, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist);
  int ProjectedSearch(FrameGridDev& grid, const FrameView& KF, const float* invLevelSigma2, const ProjectedPoints& P, float th, bool chi2Gate,
                      int distThreshold, int32_t* matched, bool claim, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist);
  // SearchBySim3's agreement step (:1318-1345) on the two directional results
  static int MutualAgreement(const std::vector<int32_t>& vnMatch1, const std::vector<int32_t>& vnMatch2, std::vector<int32_t>& matches12);
  friend class TriangulationBatch;
  friend class FuseBatch;
 private:
  void deviceWindows(FrameGridDev& grid, const std::vector<float>& u, const std::vector<float>& v, const std::vector<float>& r,
                     const std::vector<int32_t>& minl, const std::vector<int32_t>& maxl, const std::vector<uint8_t>& qdesc,
                     std::vector<int32_t>& off, std::vector<int32_t>& idx, std::vector<uint16_t>& dist);
  // distances of every (query, candidate) slot in one device launch
  void distances(const std::vector<uint8_t>& qdesc, int Q, const uint8_t* tdesc, int T, const std::vector<int32_t>& off,
                 const std::vector<int32_t>& idx, std::vector<uint16_t>& dist);
  HipContext& ctx_;
  float mfNNratio; bool mbCheckOrientation;
};

// The per-keyframe fan-out of LocalMapping::CreateNewMapPoints (cslam/src/Mapping.cpp:277-470): SearchForTriangulation(pKF1, pKF2_j, F12_j, ...) for up to 20
// covisible neighbours j of the new keyframe.  The Hamming work of ALL of them — every (feature of KF1 without a map point, feature of neighbour j in the same
// vocabulary node) pair — goes to the device as ONE ccm_hamming_csr_multi launch with one read-back when the batch is built; resolve(j, ...) then replays the
// reference's sequential rules of the call against neighbour j (epipole distance, epipolar line, first-best claim, rotation histogram) from the stored distances.
// Between two calls of the reference's loop the keyframes GAIN map points (the triangulated matches of the previous neighbour): resolve() takes the map-point flags
// as they are at ITS call and skips what the reference would skip then (:745-749, :763), so the sequence of resolve() calls returns exactly what the sequence
// of SearchForTriangulation calls returns.  Everything the batch needs is copied at build time (the views may go away).
class TriangulationBatch {
 public:
  TriangulationBatch(ORBmatcher& m, const KeysView& KF1, const std::vector<KeysView>& KF2);
  int neighbours() const { return (int)nb_.size(); }
  int64_t candidates() const { return n_cand_; }
  // has1_now / has2_now: nullptr = the flags of the build
  int resolve(int j, const uint8_t* has1_now, const uint8_t* has2_now, const float F12[9], float ex, float ey, const float* sigma2_2, const float* scaleFactors2,
              std::vector<int32_t>& matches12) const;
 private:
  struct Nb { std::vector<KeyPoint> keys; std::vector<uint8_t> has; std::vector<int32_t> q_of, off, idx; std::vector<uint16_t> dist; };
  std::vector<KeyPoint> keys1_; std::vector<uint8_t> has1_;
  std::vector<Nb> nb_;
  int64_t n_cand_ = 0;
  bool check_ori_;
};

// The fan-out of LocalMapping::SearchInNeighbors (cslam/src/Mapping.cpp:497-503): matcher.Fuse(pKFi, vpMapPointMatches) for every target keyframe — up to 20 covisible
// neighbours and 5 second neighbours of each — and the projected window searches of the same shape elsewhere (Fuse with Sim3 per loop keyframe, MapMatcher / LoopFinder).
// Fuse never claims features inside its loop (ORBmatcher.cpp:854-993: the best candidate of a point depends on that point alone), and what the calls change between
// each other — a point Replace()d by an earlier call is bad, a point has meanwhile been added to the keyframe — only makes a later call SKIP points (:884-888).  So the
// Hamming work of ALL targets goes to the device as ONE ccm_hamming_csr_multi launch when the batch is built (each target = one search against its own descriptor set),
// and resolve(s, skip_now) replays call s from the stored distances: bit for bit what ProjectedSearch(KF_s, ..., chi2Gate, distThreshold, matched = nullptr) returns
// for the points that are not skipped at that moment.  Everything the batch needs is copied at build time.
class FuseBatch {
 public:
  struct Target { FrameView KF; const float* invLevelSigma2 = nullptr; ORBmatcher::ProjectedPoints P; float th = 3.0f; };
  FuseBatch(ORBmatcher& m, const std::vector<Target>& targets, bool chi2Gate = true, int distThreshold = ORBmatcher::TH_LOW);
  int targets() const { return (int)tg_.size(); }
  int64_t candidates() const { return n_cand_; }
  // skip_now: nullable [P.n of target s]; != 0: the reference's loop would `continue` for this point now (isBad() / IsInKeyFrame(pKF) turned true since the build)
  int resolve(int s, const uint8_t* skip_now, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) const;
 private:
  struct Tg { int n_pts = 0; std::vector<int32_t> q_of, off, idx; std::vector<uint16_t> dist; };
  std::vector<Tg> tg_;
  int64_t n_cand_ = 0;
  int dist_threshold_;
};

// ---------------------------------------------------------------------------------------------------
// ORBVocabulary::transform (DBoW2 TemplatedVocabulary<FORB>, thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1260) as
// KeyFrame::ComputeBoW / Frame::ComputeBoW call it (levelsup = 4), and MapPoint::ComputeDistinctiveDescriptors
// (MapPoint.cpp:929-994) batched over many map points — SURVEY §8f rows 1 and 3.
// ---------------------------------------------------------------------------------------------------
struct BowVector { std::vector<int32_t> word; std::vector<double> value; };                       // ascending word ids, L1-normalised
struct FeatureVector { std::vector<int32_t> node, off, idx; };                                     // ascending nodes, CSR, features in index order
class ORBVocabulary {
 public:
  // flat tree, see ccm_vocab_create (include/ccm_hip.h)
  ORBVocabulary(HipContext& ctx, int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc,
                const int32_t* word_id, const double* weight);
  ~ORBVocabulary();
  void transform(const uint8_t* descriptors, int N, BowVector& v, FeatureVector& fv, int levelsup = 4) const;
 private:
  ccm_vocab* voc_ = nullptr;
};
// best descriptor per map point: desc rows off[p]..off[p+1] are the observations of point p; returns local indices
std::vector<int32_t> ComputeDistinctiveDescriptors(HipContext& ctx, const uint8_t* desc, const std::vector<int32_t>& off);

// ---------------------------------------------------------------------------------------------------
// Optimizer — cslam/include/cslam/Optimizer.h:84-112 (numerics; graph walking is the integrator's glue)
// ---------------------------------------------------------------------------------------------------
struct BAProblem {   // owning, f64 like g2o; filled from KeyFrames / MapPoints via Converter (Converter.cc:40-119)
  std::vector<double> cam_qt;   // n_cam*7
  std::vector<uint8_t> cam_fixed;
  std::vector<double> cam_K;    // n_cam*4
  std::vector<double> pt_xyz;   // n_pt*3
  std::vector<int32_t> e_cam, e_pt;
  std::vector<double> e_obs, e_info;
  int n_cam() const { return (int)cam_fixed.size(); }
  int n_pt() const { return (int)pt_xyz.size() / 3; }
  int n_edge() const { return (int)e_cam.size(); }
};

class Optimizer {
 public:
  // PoseOptimizationClient (Optimizer.cpp:215-347): returns nInitialCorrespondences - nBad; outlier[] = mvbOutlier
  static int PoseOptimizationClient(HipContext& ctx, double cam_qt[7], int n, const double* Xw, const double* obs,
                                    const double* invSigma2, const double K[4], std::vector<uint8_t>& outlier);
  // LocalBundleAdjustmentClient numerics (Optimizer.cpp:532-602): optimize(5) Huber sqrt(5.991) -> outliers to level 1,
  // kernel off -> optimize(10).  pbStopFlag as in the reference.  to_erase[e] = 1 for observations the reference erases.
  static void LocalBundleAdjustmentClient(HipContext& ctx, BAProblem& p, bool* pbStopFlag, std::vector<uint8_t>& to_erase);
  // BundleAdjustmentClient / MapFusionGBA numerics (Optimizer.cpp:163-167, 786-797): optimize(nIterations), Huber sqrt(5.99)
  static void GlobalBundleAdjustment(HipContext& ctx, BAProblem& p, int nIterations, bool* pbStopFlag, bool bRobust,
                                     ccm_ba_stats* stats = nullptr);
  // OptimizeSim3 (Optimizer.cpp:861-1056).  Pair i of the valid correspondences (:911-947): P1c / P2c = the points in
  // their own keyframe's camera frame, obs = undistorted keypoints, invSigma2 per octave.  g2oS12 = [qx qy qz qw tx ty tz s]
  // in/out; keep[i] = 0 where the reference nulls vpMatches1[idx]; returns nIn (0 => g2oS12 untouched).
  static int OptimizeSim3(HipContext& ctx, double g2oS12[8], int n, const double* P1c, const double* P2c, const double* obs1,
                          const double* obs2, const double* invSigma2_1, const double* invSigma2_2, const double K1[4],
                          const double K2[4], float th2, bool bFixScale, std::vector<uint8_t>& keep);
  // OptimizeEssentialGraphLoopClosure / MapFusion numerics (Optimizer.cpp:1058-1331, 1333-1566): the caller builds the vertex
  // list (Siw per keyframe as [qx qy qz qw tx ty tz s], fixed = pLoopKF) and the edge list (i, j, Sji) exactly as the two
  // functions do (:1122-1260: loop connections, spanning tree, loop edges, covisibility >= minFeat) and applies the write-back
  // (:1268-1330) from the optimised vertices.  optimize(20) with setUserLambdaInit(1e-16).
  static void OptimizeEssentialGraph(HipContext& ctx, std::vector<double>& vScw, const std::vector<uint8_t>& fixed, bool bFixScale,
                                     const std::vector<int32_t>& e_i, const std::vector<int32_t>& e_j, const std::vector<double>& Sji,
                                     ccm_pg_stats* stats = nullptr);
};

}  // namespace cslam
