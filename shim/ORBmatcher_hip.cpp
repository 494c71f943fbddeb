// ORBmatcher_hip.cpp — DROP-IN replacement for the translation unit cslam/src/ORBmatcher.cpp of the reference.
// Defines the methods that the reference's own header declares (cslam/include/cslam/ORBmatcher.h:100-139), compiled against that header and the
// reference's Frame / KeyFrame / MapPoint classes.  Every method has the same three steps:
//   1. flatten what the method reads from the map objects (keypoints, descriptors, feature vectors, which features hold a map point, and — for
//      the projected searches — the outcome of the f32 projection / depth / viewing-angle tests, computed here with the SAME cv::Mat expressions
//      the reference uses, so OpenCV rounds them identically),
//   2. one call into libccm_host.so (ccm_host_c.h), which gathers the window / bucket candidates, computes ALL Hamming distances of the call in one
//      ccm_hamming_csr launch on the MI355X and replays the reference's sequential acceptance rules (ratio test, claims, rotation histogram),
//   3. apply the result to the map objects in the order the reference's loop would have (Replace / AddObservation / AddMapPoint /
//      RemapMapPointMatch re-checked per point, because those mutations feed the `isBad()` / `IsInKeyFrame()` tests of later points).
// The keyframe window lists come from the caller's own KeyFrame::GetFeaturesInArea (KeyFrame.cpp stays the reference's in a drop-in build): a
// keyframe's grid was filled with the Frame's float image bounds but is read with int-truncated ones (KeyFrame.cpp:54-61, 1167-1171), so only that
// lookup has the reference's candidate order.
// There is no CPU path: if the device call fails the method throws estd::infrastructure_ex, as the reference does for fatal conditions.
#include <cslam/ORBmatcher.h>

#include <climits>
#include <memory>
#include <set>
#include <cstdlib>
#include <cstring>

#include "../ccm_slam_amd/host/ccm_host_c.h"

namespace cslam {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace {
typedef boost::shared_ptr<MapPoint> mpptr;
typedef boost::shared_ptr<KeyFrame> kfptr;

int device() { static const int d = std::getenv("CCM_DEVICE") ? std::atoi(std::getenv("CCM_DEVICE")) : 0; return d; }
int checked(int rc, const char* what) {
  if (rc <= -1000) { cout << COUTFATAL << "ORBmatcher::" << what << ": the MI355X path failed (libccm_host / libccm_hip)" << endl; throw estd::infrastructure_ex(); }
  return rc;
}

// keypoints + descriptors of a Frame / KeyFrame as flat arrays
struct FlatKeys {
  std::vector<float> x, y, angle; std::vector<int32_t> oct; const uint8_t* desc = nullptr; cv::Mat keep; int N = 0;
  FlatKeys(const std::vector<cv::KeyPoint>& k, const cv::Mat& d) : x(k.size()), y(k.size()), angle(k.size()), oct(k.size()), N((int)k.size()) {
    for (int i = 0; i < N; i++) { x[i] = k[i].pt.x; y[i] = k[i].pt.y; angle[i] = k[i].angle; oct[i] = k[i].octave; }
    keep = d.isContinuous() ? d : d.clone();
    desc = keep.data;
  }
};
// a DBoW2::FeatureVector (node -> feature indices) as CSR
struct FlatFeatVec {
  std::vector<int32_t> node, off, idx;
  explicit FlatFeatVec(const DBoW2::FeatureVector& fv) {
    off.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
      node.push_back((int32_t)it->first);
      for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
      off.push_back((int32_t)idx.size());
    }
  }
  int n() const { return (int)node.size(); }
};
void put_desc(std::vector<uint8_t>& dst, size_t i, const cv::Mat& d) { std::memcpy(dst.data() + 32 * i, d.data, 32); }

// The tests that precede the candidate loop of the keyframe searches (depth, image, distance range, optional viewing angle, predicted level) and the
// window lookup, for map points given in world coordinates of a camera [Rcw | tcw] with centre Ow.
struct Projection {
  std::vector<uint8_t> valid, desc; std::vector<float> u, v; std::vector<int32_t> level, cand_off, cand_idx;
  explicit Projection(size_t n) : valid(n, 0), desc(32 * n, 0), u(n, 0.f), v(n, 0.f), level(n, 0), cand_off(n + 1, 0) {}
};
// (p3Dc = the point in the keyframe's camera; dist3D = the distance its method feeds to the range test and PredictScale)
void project_into_keyframe(Projection& P, size_t i, const kfptr& pKF, const mpptr& pMP, const cv::Mat& p3Dc, float dist3D, float th) {
  const float z = p3Dc.at<float>(2);
  const float invz = 1.0 / z;          // == 1.0f / z: a double quotient of two floats rounds to float like the float division does (53 >= 2*24 + 2)
  const float x = p3Dc.at<float>(0) * invz, y = p3Dc.at<float>(1) * invz;
  const float u = pKF->fx * x + pKF->cx, v = pKF->fy * y + pKF->cy;
  if (!pKF->IsInImage(u, v)) return;
  if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) return;
  P.valid[i] = 1; P.u[i] = u; P.v[i] = v; P.level[i] = pMP->PredictScale(dist3D, pKF);
  put_desc(P.desc, i, pMP->GetDescriptor());
  const std::vector<size_t> win = pKF->GetFeaturesInArea(u, v, th * pKF->mvScaleFactors[P.level[i]]);
  for (size_t k = 0; k < win.size(); k++) P.cand_idx.push_back((int32_t)win[k]);
}
void close_row(Projection& P, size_t i) { P.cand_off[i + 1] = (int32_t)P.cand_idx.size(); }

int window_search(const kfptr& pKF, const FlatKeys& K, Projection& P, bool chi2Gate, int distThreshold, int32_t* matched, bool claim, const uint8_t* noClaim,
                  std::vector<int32_t>& bestIdx, const char* what) {
  const int n = (int)P.valid.size();
  bestIdx.assign(n, -1);
  std::vector<int32_t> bestDist(n, INT_MAX);
  if (n == 0 || K.N == 0) return 0;
  if (P.cand_idx.empty()) P.cand_idx.push_back(0);
  return checked(ccmh_projected_window_search_cand(device(), K.x.data(), K.y.data(), K.oct.data(), K.desc, K.N, pKF->mvInvLevelSigma2.data(), n, P.valid.data(),
                                                   P.u.data(), P.v.data(), P.level.data(), P.desc.data(), P.cand_off.data(), P.cand_idx.data(), chi2Gate ? 1 : 0,
                                                   distThreshold, matched, claim ? 1 : 0, noClaim, bestIdx.data(), bestDist.data()), what);
}
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// ORBmatcher.cpp:1653-1669.  The per-pair form is only used by callers outside the matcher (MapPoint.cpp:645,968, Tracking.cpp:491); the searches below
// never call it: their distances come from the device in one batch.
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
  uint64_t wa[4], wb[4];
  std::memcpy(wa, a.ptr<uchar>(), 32); std::memcpy(wb, b.ptr<uchar>(), 32);
  int dist = 0;
  for (int i = 0; i < 4; i++) dist += __builtin_popcountll(wa[i] ^ wb[i]);
  return dist;
}

// M1 — ORBmatcher.cpp:71-148.  Local map points come from the observations of the local keyframes (Tracking::UpdateLocalPoints), so an assigned point
// has Observations() > 0 and blocks its feature for the points after it, which is the rule libccm_host replays.
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<mpptr>& vpMapPoints, const float th) {
  const int n = (int)vpMapPoints.size();
  if (n == 0 || F.N == 0) return 0;
  FlatKeys K(F.mvKeysUn, F.mDescriptors);
  std::vector<uint8_t> inView(n, 0), desc(32 * (size_t)n, 0);
  std::vector<float> px(n, 0.f), py(n, 0.f), vcos(n, 0.f);
  std::vector<int32_t> lvl(n, 0);
  for (int i = 0; i < n; i++) {
    const mpptr& p = vpMapPoints[i];
    if (!p->mbTrackInView || p->isBad()) continue;
    inView[i] = 1; px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; lvl[i] = p->mnTrackScaleLevel; vcos[i] = p->mTrackViewCos;
    put_desc(desc, i, p->GetDescriptor());
  }
  const int32_t FOREIGN = INT_MAX;
  std::vector<int32_t> table(F.N, -1);
  for (int j = 0; j < F.N; j++) if (F.mvpMapPoints[j] && F.mvpMapPoints[j]->Observations() > 0) table[j] = FOREIGN;
  const int nmatches = checked(ccmh_search_by_projection_mp(device(), K.x.data(), K.y.data(), K.oct.data(), K.desc, K.N, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX,
                                                            Frame::mnMaxY, F.mvScaleFactors.data(), n, inView.data(), px.data(), py.data(), lvl.data(), vcos.data(),
                                                            desc.data(), th, mfNNratio, table.data()), "SearchByProjection(Frame, MapPoints)");
  for (int j = 0; j < F.N; j++) if (table[j] >= 0 && table[j] != FOREIGN) F.mvpMapPoints[j] = vpMapPoints[table[j]];
  return nmatches;
}

// M2 — ORBmatcher.cpp:1350-1476
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th) {
  const int nLast = LastFrame.N;
  if (nLast == 0 || CurrentFrame.N == 0) return 0;
  FlatKeys K(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors);
  const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
  const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
  std::vector<uint8_t> valid(nLast, 0), desc(32 * (size_t)nLast, 0);
  std::vector<float> u(nLast, 0.f), v(nLast, 0.f), angle(nLast, 0.f);
  std::vector<int32_t> oct(nLast, 0);
  for (int i = 0; i < nLast; i++) {
    const mpptr& pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const cv::Mat x3Dc = Rcw * pMP->GetWorldPos() + tcw;             // one gemm, as in the reference
    const float invzc = 1.0 / x3Dc.at<float>(2);
    if (invzc < 0) continue;
    const float pu = CurrentFrame.fx * x3Dc.at<float>(0) * invzc + CurrentFrame.cx;
    const float pv = CurrentFrame.fy * x3Dc.at<float>(1) * invzc + CurrentFrame.cy;
    if (pu < Frame::mnMinX || pu > Frame::mnMaxX || pv < Frame::mnMinY || pv > Frame::mnMaxY) continue;
    valid[i] = 1; u[i] = pu; v[i] = pv;
    oct[i] = LastFrame.mvKeys[i].octave; angle[i] = LastFrame.mvKeysUn[i].angle;
    put_desc(desc, i, pMP->GetDescriptor());
  }
  const int32_t FOREIGN = INT_MAX;
  std::vector<int32_t> table(CurrentFrame.N, -1);
  for (int j = 0; j < CurrentFrame.N; j++) if (CurrentFrame.mvpMapPoints[j] && CurrentFrame.mvpMapPoints[j]->Observations() > 0) table[j] = FOREIGN;
  const int nmatches = checked(ccmh_search_by_projection_last(device(), K.x.data(), K.y.data(), K.oct.data(), K.angle.data(), K.desc, K.N, Frame::mnMinX, Frame::mnMinY,
                                                              Frame::mnMaxX, Frame::mnMaxY, CurrentFrame.mvScaleFactors.data(), nLast, valid.data(), u.data(), v.data(),
                                                              oct.data(), angle.data(), desc.data(), th, mbCheckOrientation ? 1 : 0, table.data()),
                               "SearchByProjection(Frame, LastFrame)");
  for (int j = 0; j < CurrentFrame.N; j++) if (table[j] >= 0 && table[j] != FOREIGN) CurrentFrame.mvpMapPoints[j] = LastFrame.mvpMapPoints[table[j]];
  return nmatches;
}

// ORBmatcher.cpp:1478-1604 — the relocalisation search has no caller in CCM-SLAM: Tracking never relocalises against a keyframe.
int ORBmatcher::SearchByProjection(Frame&, kfptr, const std::set<mpptr>&, const float, const int) {
  cout << COUTFATAL << "ORBmatcher::SearchByProjection(Frame, KeyFrame, sAlreadyFound): not part of the CCM-SLAM hot path, not provided by the MI355X build" << endl;
  throw estd::infrastructure_ex();
}

// M9 — ORBmatcher.cpp:308-446
int ORBmatcher::SearchByProjection(kfptr pKF, cv::Mat Scw, const std::vector<mpptr>& vpPoints, std::vector<mpptr>& vpMatched, int th) {
  const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
  const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
  const cv::Mat Rcw = sRcw / scw;
  const cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
  const cv::Mat Ow = -Rcw.t() * tcw;
  std::set<mpptr> already(vpMatched.begin(), vpMatched.end());
  already.erase(mpptr());
  const size_t n = vpPoints.size();
  FlatKeys K(pKF->mvKeysUn, pKF->mDescriptors);
  Projection P(n);
  std::vector<uint8_t> noClaim(n, 0);
  for (size_t i = 0; i < n; i++) {
    const mpptr& pMP = vpPoints[i];
    if (!pMP->isBad() && !already.count(pMP)) {
      const cv::Mat p3Dw = pMP->GetWorldPos();
      const cv::Mat p3Dc = Rcw * p3Dw + tcw;
      if (!(p3Dc.at<float>(2) < 0.0f)) {
        const cv::Mat PO = p3Dw - Ow;
        const float dist = cv::norm(PO);
        if (!(PO.dot(pMP->GetNormal()) < 0.5 * dist)) project_into_keyframe(P, i, pKF, pMP, p3Dc, dist, (float)th);
      }
      noClaim[i] = pMP->GetIndexInKeyFrame(pKF) != -1;      // observed already: re-mapped inside the keyframe instead of matched (:412-432)
    }
    close_row(P, i);
  }
  std::vector<int32_t> matched(K.N, -1), best;
  for (int j = 0; j < K.N; j++) if (vpMatched[j]) matched[j] = INT_MAX;
  window_search(pKF, K, P, false, TH_LOW, matched.data(), true, noClaim.data(), best, "SearchByProjection(KeyFrame, Scw)");
  int nmatches = 0;
  for (size_t i = 0; i < n; i++) {
    if (best[i] < 0) continue;
    const mpptr& pMP = vpPoints[i];
    const int existing = pMP->GetIndexInKeyFrame(pKF);
    if (existing != -1) {
      // a point the keyframe already observes found a feature: it moves there.  (The reference guards this with a distance test, :419-426, whose two
      // operands are the same pair of descriptor rows — dist(dMP, row bestIdx) against bestDist — so the guard never holds and the point always moves.)
      pKF->RemapMapPointMatch(pMP, existing, best[i]);
    } else {
      vpMatched[best[i]] = pMP;
      nmatches++;
    }
  }
  return nmatches;
}

namespace {
// flat view of one side of the BoW searches
struct BowSide {
  FlatKeys K; FlatFeatVec fv; std::vector<uint8_t> has;
  BowSide(const std::vector<cv::KeyPoint>& k, const cv::Mat& d, const DBoW2::FeatureVector& f) : K(k, d), fv(f), has(k.size(), 0) {}
};
int bow_search(int mode, BowSide& A, BowSide& B, const float* F12, float ex, float ey, const float* sigma2, const float* sf, float ratio, bool ori,
               std::vector<int32_t>& out, const char* what) {
  static const int32_t none = 0;
  static const float ones[8] = {1, 1, 1, 1, 1, 1, 1, 1}, eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  auto p = [](const std::vector<int32_t>& v) { return v.empty() ? &none : v.data(); };
  return checked(ccmh_search_bow(device(), mode, p(A.fv.node), A.fv.off.data(), p(A.fv.idx), A.fv.n(), p(B.fv.node), B.fv.off.data(), p(B.fv.idx), B.fv.n(), A.has.data(),
                                 B.has.data(), A.K.desc, A.K.x.data(), A.K.y.data(), A.K.angle.data(), A.K.N, B.K.desc, B.K.x.data(), B.K.y.data(), B.K.oct.data(),
                                 B.K.angle.data(), B.K.N, F12 ? F12 : eye, ex, ey, sigma2 ? sigma2 : ones, sf ? sf : ones, ratio, ori ? 1 : 0, out.data()), what);
}
}  // namespace

// M3 — ORBmatcher.cpp:178-306
int ORBmatcher::SearchByBoW(kfptr pKF, Frame& F, std::vector<mpptr>& vpMapPointMatches) {
  const std::vector<mpptr> vpMapPointsKF = pKF->GetMapPointMatches();
  vpMapPointMatches = std::vector<mpptr>(F.N, mpptr());
  if (F.N == 0 || vpMapPointsKF.empty()) return 0;
  BowSide A(pKF->mvKeysUn, pKF->mDescriptors, pKF->mFeatVec), B(F.mvKeys, F.mDescriptors, F.mFeatVec);   // the frame side reads mvKeys[].angle (:262)
  for (size_t i = 0; i < vpMapPointsKF.size(); i++) A.has[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
  std::vector<int32_t> kfOfF(F.N, -1);
  const int nmatches = bow_search(0, A, B, nullptr, 0.f, 0.f, nullptr, nullptr, mfNNratio, mbCheckOrientation, kfOfF, "SearchByBoW(KeyFrame, Frame)");
  for (int j = 0; j < F.N; j++) if (kfOfF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[kfOfF[j]];
  return nmatches;
}

// M4 — ORBmatcher.cpp:565-698
int ORBmatcher::SearchByBoW(kfptr pKF1, kfptr pKF2, std::vector<mpptr>& vpMatches12) {
  const std::vector<mpptr> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
  vpMatches12 = std::vector<mpptr>(vpMapPoints1.size(), mpptr());
  if (vpMapPoints1.empty() || vpMapPoints2.empty()) return 0;
  BowSide A(pKF1->mvKeysUn, pKF1->mDescriptors, pKF1->mFeatVec), B(pKF2->mvKeysUn, pKF2->mDescriptors, pKF2->mFeatVec);
  for (size_t i = 0; i < vpMapPoints1.size(); i++) A.has[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
  for (size_t i = 0; i < vpMapPoints2.size(); i++) B.has[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
  std::vector<int32_t> m12(vpMapPoints1.size(), -1);
  const int nmatches = bow_search(1, A, B, nullptr, 0.f, 0.f, nullptr, nullptr, mfNNratio, mbCheckOrientation, m12, "SearchByBoW(KeyFrame, KeyFrame)");
  for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
  return nmatches;
}

namespace {
// (round 5) LocalMapping::CreateNewMapPoints (Mapping.cpp:286, :335) calls SearchForTriangulation for up to 20 covisibility neighbours of ONE new keyframe, one after
// the other.  Mapping.cpp is not ours to change, but its first call tells what the next ones will be: on a call for a keyframe it has not seen, the shim asks the
// keyframe for the same neighbour list (GetBestCovisibilityKeyFrames(20)), sends the Hamming work of ALL of them to the MI355X as one launch
// (ccmh_tri_batch_create -> ccm_hamming_csr_multi) and keeps the tables, per calling thread, until a call for another keyframe arrives; every call — the first
// included — is then answered from the tables with the map-point flags AS THEY ARE AT THAT CALL (the loop creates map points between the calls), i.e. with exactly
// the reference's result for that call.  What the tables depend on — keypoints, descriptors, feature vectors — does not change after a keyframe is built.  A
// neighbour the prediction missed takes the single-call path.  CCM_SHIM_TRI_BATCH=0 switches the prediction off.
struct TriFanOut {
  const KeyFrame* kf1 = nullptr; idpair id1; int N1 = 0;
  std::vector<const KeyFrame*> nb; std::vector<idpair> nb_id;
  std::vector<uint8_t> has1_build;   // which features of keyframe 1 held a map point when the tables were built (those are not among the queries)
  void* h = nullptr;
  void drop() { if (h) ccmh_tri_batch_destroy(h); h = nullptr; kf1 = nullptr; nb.clear(); nb_id.clear(); has1_build.clear(); }
  ~TriFanOut() { drop(); }
};
thread_local TriFanOut g_tri;
bool tri_batch_enabled() { static const bool on = !(std::getenv("CCM_SHIM_TRI_BATCH") && std::atoi(std::getenv("CCM_SHIM_TRI_BATCH")) == 0); return on; }
}  // namespace

// M5 — ORBmatcher.cpp:700-852
int ORBmatcher::SearchForTriangulation(kfptr pKF1, kfptr pKF2, cv::Mat F12, std::vector<pair<size_t, size_t> >& vMatchedPairs) {
  vMatchedPairs.clear();
  if (pKF1->N == 0 || pKF2->N == 0) return 0;
  if (tri_batch_enabled()) {
    TriFanOut& c = g_tri;
    if (c.kf1 != pKF1.get() || c.id1 != pKF1->mId || c.N1 != pKF1->N) {
      // a keyframe this thread has not been asked about: predict the fan-out and run its Hamming work in one launch
      c.drop();
      std::vector<kfptr> vpN = pKF1->GetBestCovisibilityKeyFrames(20);                   // Mapping.cpp:285-286
      bool listed = false;
      for (size_t j = 0; j < vpN.size(); j++) listed |= vpN[j] == pKF2;
      if (!listed) vpN.push_back(pKF2);
      std::vector<kfptr> use;
      for (size_t j = 0; j < vpN.size(); j++) if (vpN[j] && vpN[j]->N > 0) use.push_back(vpN[j]);
      if (use.size() >= 2) {
        BowSide A(pKF1->mvKeysUn, pKF1->mDescriptors, pKF1->mFeatVec);
        for (int i = 0; i < pKF1->N; i++) A.has[i] = pKF1->GetMapPoint(i) ? 1 : 0;
        std::vector<std::unique_ptr<BowSide>> B;
        for (size_t j = 0; j < use.size(); j++) {
          B.emplace_back(new BowSide(use[j]->mvKeysUn, use[j]->mDescriptors, use[j]->mFeatVec));
          for (int i = 0; i < use[j]->N; i++) B.back()->has[i] = use[j]->GetMapPoint(i) ? 1 : 0;
        }
        static const int32_t none = 0;
        auto p = [](const std::vector<int32_t>& v) { return v.empty() ? &none : v.data(); };
        const size_t n = use.size();
        std::vector<const int32_t*> n2(n), o2(n), i2(n), oct2(n); std::vector<const uint8_t*> has2(n), d2(n); std::vector<const float*> x2(n), y2(n), a2(n);
        std::vector<int32_t> nn2(n), N2(n);
        for (size_t j = 0; j < n; j++) {
          n2[j] = p(B[j]->fv.node); o2[j] = B[j]->fv.off.data(); i2[j] = p(B[j]->fv.idx); nn2[j] = B[j]->fv.n(); has2[j] = B[j]->has.data(); d2[j] = B[j]->K.desc;
          x2[j] = B[j]->K.x.data(); y2[j] = B[j]->K.y.data(); oct2[j] = B[j]->K.oct.data(); a2[j] = B[j]->K.angle.data(); N2[j] = B[j]->K.N;
        }
        c.h = ccmh_tri_batch_create(device(), mfNNratio, mbCheckOrientation ? 1 : 0, p(A.fv.node), A.fv.off.data(), p(A.fv.idx), A.fv.n(), A.has.data(), A.K.desc,
                                    A.K.x.data(), A.K.y.data(), A.K.angle.data(), A.K.N, (int)n, n2.data(), o2.data(), i2.data(), nn2.data(), has2.data(), d2.data(),
                                    x2.data(), y2.data(), oct2.data(), a2.data(), N2.data());
        if (!c.h) checked(-1000, "SearchForTriangulation (fan-out)");
        c.has1_build = A.has;
        for (size_t j = 0; j < n; j++) { c.nb.push_back(use[j].get()); c.nb_id.push_back(use[j]->mId); }
      }
      // the keyframe is remembered whether or not a batch was built (fewer than two usable neighbours: c.h stays null and every call takes the single-call path
      // below) — otherwise each of its calls would ask for the covisibility list again
      c.kf1 = pKF1.get(); c.id1 = pKF1->mId; c.N1 = pKF1->N;
    }
    if (c.h) {
      for (size_t j = 0; j < c.nb.size(); j++) {
        if (c.nb[j] != pKF2.get() || c.nb_id[j] != pKF2->mId) continue;
        const cv::Mat C2 = pKF2->GetRotation() * pKF1->GetCameraCenter() + pKF2->GetTranslation();
        const float invz = 1.0f / C2.at<float>(2);
        const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
        const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
        std::vector<uint8_t> has1(pKF1->N), has2(pKF2->N);
        bool lost = false;                                                               // a feature that has LOST its map point since (culling by another thread)
        for (int i = 0; i < pKF1->N; i++) { has1[i] = pKF1->GetMapPoint(i) ? 1 : 0; lost |= c.has1_build[i] && !has1[i]; }   // is not in the tables: single call
        if (lost) break;
        for (int i = 0; i < pKF2->N; i++) has2[i] = pKF2->GetMapPoint(i) ? 1 : 0;
        cv::Mat F = F12.isContinuous() ? F12 : F12.clone();
        std::vector<int32_t> m12(pKF1->N, -1);
        const int nmatches = checked(ccmh_tri_batch_resolve(c.h, (int)j, has1.data(), has2.data(), F.ptr<float>(), ex, ey, pKF2->mvLevelSigma2.data(),
                                                            pKF2->mvScaleFactors.data(), m12.data()), "SearchForTriangulation (fan-out)");
        vMatchedPairs.reserve(nmatches);
        for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return nmatches;
      }
    }
  }
  // the epipole of camera 1 in image 2, in the reference's f32 arithmetic
  const cv::Mat C2 = pKF2->GetRotation() * pKF1->GetCameraCenter() + pKF2->GetTranslation();
  const float invz = 1.0f / C2.at<float>(2);
  const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
  const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
  BowSide A(pKF1->mvKeysUn, pKF1->mDescriptors, pKF1->mFeatVec), B(pKF2->mvKeysUn, pKF2->mDescriptors, pKF2->mFeatVec);
  for (int i = 0; i < pKF1->N; i++) A.has[i] = pKF1->GetMapPoint(i) ? 1 : 0;      // only features WITHOUT a map point take part (:745, :763)
  for (int i = 0; i < pKF2->N; i++) B.has[i] = pKF2->GetMapPoint(i) ? 1 : 0;
  cv::Mat F = F12.isContinuous() ? F12 : F12.clone();
  std::vector<int32_t> m12(pKF1->N, -1);
  const int nmatches = bow_search(2, A, B, F.ptr<float>(), ex, ey, pKF2->mvLevelSigma2.data(), pKF2->mvScaleFactors.data(), mfNNratio, mbCheckOrientation, m12,
                                  "SearchForTriangulation");
  vMatchedPairs.reserve(nmatches);
  for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
  return nmatches;
}

// M6 — ORBmatcher.cpp:448-563
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) {
  const int N1 = (int)F1.mvKeysUn.size(), N2 = (int)F2.mvKeysUn.size();
  vnMatches12 = std::vector<int>(N1, -1);
  if (N1 == 0 || N2 == 0) return 0;
  FlatKeys A(F1.mvKeysUn, F1.mDescriptors), B(F2.mvKeysUn, F2.mDescriptors);
  std::vector<float> prev(2 * (size_t)N1);
  for (int i = 0; i < N1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
  std::vector<int32_t> m12(N1, -1);
  const int nmatches = checked(ccmh_search_for_initialization(device(), A.x.data(), A.y.data(), A.oct.data(), A.angle.data(), A.desc, N1, B.x.data(), B.y.data(),
                                                              B.oct.data(), B.angle.data(), B.desc, N2, Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY,
                                                              prev.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data()), "SearchForInitialization");
  for (int i = 0; i < N1; i++) { vnMatches12[i] = m12[i]; vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]); }
  return nmatches;
}

// M10 — ORBmatcher.cpp:1124-1348
int ORBmatcher::SearchBySim3(kfptr pKF1, kfptr pKF2, std::vector<mpptr>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th) {
  const cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
  const cv::Mat sR12 = s12 * R12;
  const cv::Mat sR21 = (1.0 / s12) * R12.t();
  const cv::Mat t21 = -sR21 * t12;
  const std::vector<mpptr> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
  const int N1 = (int)vp1.size(), N2 = (int)vp2.size();
  std::vector<bool> done1(N1, false), done2(N2, false);
  for (int i = 0; i < N1; i++) {
    const mpptr& pMP = vpMatches12[i];
    if (!pMP) continue;
    done1[i] = true;
    const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
    if (idx2 >= 0 && idx2 < N2) done2[idx2] = true;
  }
  // one direction: the points of `from` (camera frame [Rw | tw]) carried by [sR | t] into the camera of `into`
  auto direction = [&](const std::vector<mpptr>& pts, const std::vector<bool>& done, const cv::Mat& Rw, const cv::Mat& tw, const cv::Mat& sR, const cv::Mat& t,
                       const kfptr& into, std::vector<int32_t>& best, const char* what) {
    FlatKeys K(into->mvKeysUn, into->mDescriptors);
    Projection P(pts.size());
    for (size_t i = 0; i < pts.size(); i++) {
      const mpptr& pMP = pts[i];
      if (pMP && !done[i] && !pMP->isBad()) {
        const cv::Mat pa = Rw * pMP->GetWorldPos() + tw;
        const cv::Mat pb = sR * pa + t;
        if (!(pb.at<float>(2) < 0.0)) project_into_keyframe(P, i, into, pMP, pb, (float)cv::norm(pb), th);
      }
      close_row(P, i);
    }
    window_search(into, K, P, false, TH_HIGH, nullptr, false, nullptr, best, what);
  };
  std::vector<int32_t> vnMatch1, vnMatch2;
  direction(vp1, done1, R1w, t1w, sR21, t21, pKF2, vnMatch1, "SearchBySim3 (1 -> 2)");
  direction(vp2, done2, R2w, t2w, sR12, t12, pKF1, vnMatch2, "SearchBySim3 (2 -> 1)");
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == i1) { vpMatches12[i1] = vp2[idx2]; nFound++; }
  }
  return nFound;
}

namespace {
// the tests that precede Fuse's candidate loop (ORBmatcher.cpp:868-913) for every point of the list, against one keyframe
void fuse_project(const kfptr& pKF, const vector<mpptr>& vpMapPoints, float th, Projection& P) {
  const cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
  for (size_t i = 0; i < vpMapPoints.size(); i++) {
    const mpptr& pMP = vpMapPoints[i];
    if (pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)) {
      const cv::Mat p3Dw = pMP->GetWorldPos();
      const cv::Mat p3Dc = Rcw * p3Dw + tcw;
      if (!(p3Dc.at<float>(2) < 0.0f)) {
        const cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (!(PO.dot(pMP->GetNormal()) < 0.5 * dist3D)) project_into_keyframe(P, i, pKF, pMP, p3Dc, dist3D, th);
      }
    }
    close_row(P, i);
  }
}

// (round 6) LocalMapping::SearchInNeighbors (Mapping.cpp:469-503) calls Fuse(pKFi, vpMapPointMatches) for every fuse target of the new keyframe — its (up to 20) best
// covisibility neighbours and up to five second neighbours of each — one after the other on the SAME vector.  Mapping.cpp is not ours to change, but its first call tells
// what the next ones will be: the first target carries the current keyframe's id in mFuseTargetForKF (Mapping.cpp:481), the current keyframe is among its covisible
// keyframes, and the target list follows from the covisibility lists by the reference's own rules.  On a call with a vector this thread has not seen, the shim replays
// that walk, projects the points into ALL predicted targets and sends their Hamming work to the MI355X as one launch (ccmh_fuse_batch_create_cand ->
// ccm_hamming_csr_multi); every call — the first included — is then answered from the tables with the points' flags AS THEY ARE AT THAT CALL: what the calls change for
// each other is only that a point has turned bad (Replace) or has joined the keyframe meanwhile, and such a point is skipped exactly as the reference's loop would
// (ORBmatcher.cpp:884-888).  Poses, positions, descriptors and distance ranges do not change inside the loop.  A call the prediction did not foresee drops the tables and
// takes the single-call path.  CCM_SHIM_FUSE_BATCH=0 switches the prediction off.
struct FuseFanOut {
  const void* pts = nullptr; size_t n = 0; float th = 0.f; const void* first = nullptr; const void* last = nullptr;
  std::vector<const KeyFrame*> tg; std::vector<idpair> tg_id; size_t cursor = 0;
  void* h = nullptr;
  void drop() { if (h) ccmh_fuse_batch_destroy(h); h = nullptr; pts = nullptr; tg.clear(); tg_id.clear(); cursor = 0; }
  ~FuseFanOut() { drop(); }
};
thread_local FuseFanOut g_fuse;
bool fuse_batch_enabled() { static const bool on = !(std::getenv("CCM_SHIM_FUSE_BATCH") && std::atoi(std::getenv("CCM_SHIM_FUSE_BATCH")) == 0); return on; }

void fuse_predict(FuseFanOut& c, const kfptr& pKF, const vector<mpptr>& vpMapPoints, float th) {
  const idpair curId = pKF->mFuseTargetForKF;
  kfptr cur;
  {
    const std::vector<kfptr> cov = pKF->GetVectorCovisibleKeyFrames();
    for (size_t j = 0; j < cov.size(); j++) if (cov[j] && cov[j]->mId == curId) { cur = cov[j]; break; }
  }
  if (!cur || (size_t)cur->N != vpMapPoints.size()) return;                 // not the fan-out of SearchInNeighbors (vpMapPointMatches has one entry per feature)
  // Mapping.cpp:472-492 replayed; the marks the reference has already set are reproduced with a local set (a second neighbour is tested against the marks of the
  // first-level neighbours BEFORE it in the walk only)
  std::vector<kfptr> targets;
  std::set<const KeyFrame*> marked;
  const std::vector<kfptr> vpNeighKFs = cur->GetBestCovisibilityKeyFrames(20);
  for (size_t a = 0; a < vpNeighKFs.size(); a++) {
    const kfptr& pKFi = vpNeighKFs[a];
    if (pKFi->isBad() || marked.count(pKFi.get())) continue;
    if (!(pKFi->mFuseTargetForKF == curId)) return;                         // the reference marked every first-level target before its first call: not this walk
    targets.push_back(pKFi); marked.insert(pKFi.get());
    const std::vector<kfptr> vpSecond = pKFi->GetBestCovisibilityKeyFrames(5);
    for (size_t b = 0; b < vpSecond.size(); b++) {
      const kfptr& pKFi2 = vpSecond[b];
      if (pKFi2->isBad() || marked.count(pKFi2.get()) || pKFi2->mId == curId) continue;
      targets.push_back(pKFi2);
    }
  }
  if (targets.size() < 2 || targets[0] != pKF) return;
  const int S = (int)targets.size();
  const size_t n = vpMapPoints.size();
  std::vector<int32_t> kf_off(S + 1, 0), pt_off(S + 1, 0), oct, level, cand_off, cand_base(S, 0), cand_idx;
  std::vector<float> kx, ky, u, v;
  std::vector<uint8_t> kdesc, valid, pdesc;
  std::vector<const float*> isig(S);
  for (int s = 0; s < S; s++) {
    const kfptr& t = targets[s];
    FlatKeys K(t->mvKeysUn, t->mDescriptors);
    kx.insert(kx.end(), K.x.begin(), K.x.end()); ky.insert(ky.end(), K.y.begin(), K.y.end()); oct.insert(oct.end(), K.oct.begin(), K.oct.end());
    kdesc.insert(kdesc.end(), K.desc, K.desc + (size_t)K.N * 32);
    kf_off[s + 1] = kf_off[s] + K.N;
    Projection P(n);
    fuse_project(t, vpMapPoints, th, P);
    valid.insert(valid.end(), P.valid.begin(), P.valid.end()); u.insert(u.end(), P.u.begin(), P.u.end()); v.insert(v.end(), P.v.begin(), P.v.end());
    level.insert(level.end(), P.level.begin(), P.level.end()); pdesc.insert(pdesc.end(), P.desc.begin(), P.desc.end());
    pt_off[s + 1] = pt_off[s] + (int32_t)n;
    cand_off.insert(cand_off.end(), P.cand_off.begin(), P.cand_off.end());
    cand_base[s] = (int32_t)cand_idx.size();
    cand_idx.insert(cand_idx.end(), P.cand_idx.begin(), P.cand_idx.end());
    isig[s] = t->mvInvLevelSigma2.data();
  }
  if (cand_idx.empty()) cand_idx.push_back(0);
  c.h = ccmh_fuse_batch_create_cand(device(), S, kf_off.data(), kx.data(), ky.data(), oct.data(), kdesc.data(), isig.data(), pt_off.data(), valid.data(), u.data(), v.data(),
                                    level.data(), pdesc.data(), cand_off.data(), cand_base.data(), cand_idx.data(), 1, ORBmatcher::TH_LOW);
  if (!c.h) checked(-1000, "Fuse (fan-out)");
  c.pts = vpMapPoints.data(); c.n = n; c.th = th; c.first = vpMapPoints.front().get(); c.last = vpMapPoints.back().get(); c.cursor = 0;
  for (int s = 0; s < S; s++) { c.tg.push_back(targets[s].get()); c.tg_id.push_back(targets[s]->mId); }
}
}  // namespace

// M7 — ORBmatcher.cpp:854-993
int ORBmatcher::Fuse(kfptr pKF, const vector<mpptr>& vpMapPoints, const float th) {
  const size_t n = vpMapPoints.size();
  std::vector<int32_t> best;
  bool answered = false;
  if (fuse_batch_enabled() && n > 0 && pKF->N > 0) {
    FuseFanOut& c = g_fuse;
    auto entry = [&]() -> int {
      if (!c.h || c.pts != vpMapPoints.data() || c.n != n || c.th != th || c.first != vpMapPoints.front().get() || c.last != vpMapPoints.back().get()) return -1;
      for (size_t e = c.cursor; e < c.tg.size(); e++) if (c.tg[e] == pKF.get() && c.tg_id[e] == pKF->mId) return (int)e;
      return -1;
    };
    int e = entry();
    if (e < 0) { c.drop(); fuse_predict(c, pKF, vpMapPoints, th); e = entry(); }
    if (e >= 0) {
      std::vector<uint8_t> skip(n, 0);
      for (size_t i = 0; i < n; i++) { const mpptr& pMP = vpMapPoints[i]; skip[i] = !(pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)); }
      best.assign(n, -1);
      std::vector<int32_t> bestDist(n, INT_MAX);
      checked(ccmh_fuse_batch_resolve(c.h, e, skip.data(), (int)n, best.data(), bestDist.data()), "Fuse (fan-out)");
      c.cursor = (size_t)e + 1;
      if (c.cursor == c.tg.size()) c.drop();
      answered = true;
    }
  }
  if (!answered) {
    FlatKeys K(pKF->mvKeysUn, pKF->mDescriptors);
    Projection P(n);
    fuse_project(pKF, vpMapPoints, th, P);
    window_search(pKF, K, P, true, TH_LOW, nullptr, false, nullptr, best, "Fuse");
  }
  // the map is mutated point by point: a point replaced (now bad) or adopted by the keyframe (Replace moved the observation) earlier in this loop is
  // skipped when its turn comes, exactly as the reference's per-point tests would
  int nFused = 0;
  for (size_t i = 0; i < n; i++) {
    if (best[i] < 0) continue;
    const mpptr& pMP = vpMapPoints[i];
    if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
    mpptr pMPinKF = pKF->GetMapPoint(best[i]);
    if (pMPinKF) {
      if (!pMPinKF->isBad() && !pMPinKF->mbDoNotReplace) {
        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
        else pMPinKF->Replace(pMP);
      }
    } else {
      pMP->AddObservation(pKF, best[i]);
      pKF->AddMapPoint(pMP, best[i]);
    }
    nFused++;
  }
  return nFused;
}

// M8 — ORBmatcher.cpp:995-1122
int ORBmatcher::Fuse(kfptr pKF, cv::Mat Scw, const std::vector<mpptr>& vpPoints, float th, vector<mpptr>& vpReplacePoint) {
  const cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
  const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
  const cv::Mat Rcw = sRcw / scw;
  const cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
  const cv::Mat Ow = -Rcw.t() * tcw;
  const std::set<mpptr> already = pKF->GetMapPoints();
  const size_t n = vpPoints.size();
  FlatKeys K(pKF->mvKeysUn, pKF->mDescriptors);
  Projection P(n);
  for (size_t i = 0; i < n; i++) {
    const mpptr& pMP = vpPoints[i];
    if (!pMP->isBad() && !already.count(pMP)) {
      const cv::Mat p3Dw = pMP->GetWorldPos();
      const cv::Mat p3Dc = Rcw * p3Dw + tcw;
      if (!(p3Dc.at<float>(2) < 0.0f)) {
        const cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (!(PO.dot(pMP->GetNormal()) < 0.5 * dist3D)) project_into_keyframe(P, i, pKF, pMP, p3Dc, dist3D, th);
      }
    }
    close_row(P, i);
  }
  std::vector<int32_t> best;
  window_search(pKF, K, P, false, TH_LOW, nullptr, false, nullptr, best, "Fuse(Scw)");
  int nFused = 0;
  for (size_t i = 0; i < n; i++) {
    if (best[i] < 0) continue;
    const mpptr& pMP = vpPoints[i];
    mpptr pMPinKF = pKF->GetMapPoint(best[i]);
    if (pMPinKF) {
      if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
    } else {
      pMP->AddObservation(pKF, best[i]);
      pKF->AddMapPoint(pMP, best[i]);
    }
    nFused++;
  }
  return nFused;
}

}  // namespace cslam
