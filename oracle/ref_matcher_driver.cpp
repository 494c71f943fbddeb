// ref_matcher_driver.cpp — TEST INFRASTRUCTURE.  C entry points around the reference's OWN cslam::ORBmatcher (cslam/src/ORBmatcher.cpp, whole file,
// compiled verbatim by oracle/Makefile.ref) and the grid functions of its Frame / KeyFrame (Frame.cpp:103-118, 200-265, KeyFrame.cpp:1162-1206,
// extracted at build time).  Each entry point takes the SAME flat arrays as the corresponding ora_* function of oracle/match_ref.cpp, builds the
// look-alike Frame / KeyFrame / MapPoint objects (oracle/ref_shim/cslam_lookalike) the reference method reads, calls the method through the
// reference's own header (ORBmatcher.h:100-139) and flattens what it wrote.  oracle/_ref/libmatcher_ref.so is what the matcher oracle is pinned
// against (tests/test_ref_matcher.py).  Nothing in the product links or loads this.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include <cslam/ORBmatcher.h>

namespace cslam { std::mutex MapPoint::mGlobalMutex; }

namespace {
using cslam::Frame;
using cslam::KeyFrame;
using cslam::MapPoint;
typedef boost::shared_ptr<KeyFrame> kfptr;
typedef boost::shared_ptr<MapPoint> mpptr;

cv::Mat desc_mat(const uint8_t* d, int n) { cv::Mat m(n, 32, CV_8U); if (n) std::memcpy(m.data, d, (size_t)n * 32); return m; }
cv::Mat desc_row(const uint8_t* d) { cv::Mat m(1, 32, CV_8U); std::memcpy(m.data, d, 32); return m; }

// what the Frame constructor derives from the image bounds (Frame.cpp:64-77) + AssignFeaturesToGrid (the reference's own code)
void setup_frame(Frame& F, const float* kx, const float* ky, const int32_t* oct, const float* angle, const uint8_t* desc, int N, float minX, float minY,
                 float maxX, float maxY, const float* scale_factors, int n_levels) {
  F.N = N;
  F.mvKeysUn.resize(N);
  for (int i = 0; i < N; i++) F.mvKeysUn[i] = cv::KeyPoint(kx[i], ky[i], 31.f, angle ? angle[i] : -1.f, 0, oct[i]);
  F.mvKeys = F.mvKeysUn;
  F.mDescriptors = desc_mat(desc, N);
  F.mvpMapPoints.assign(N, mpptr());
  F.mvbOutlier.assign(N, false);
  Frame::mnMinX = minX; Frame::mnMinY = minY; Frame::mnMaxX = maxX; Frame::mnMaxY = maxY;
  Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(Frame::mnMaxX - Frame::mnMinX);
  Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(Frame::mnMaxY - Frame::mnMinY);
  F.mnScaleLevels = n_levels;
  F.mvScaleFactors.assign(scale_factors, scale_factors + n_levels);
  F.AssignFeaturesToGrid();
}
void fill_fv(DBoW2::FeatureVector& fv, const int32_t* node, const int32_t* off, const int32_t* idx, int nn) {
  for (int k = 0; k < nn; k++) for (int s = off[k]; s < off[k + 1]; s++) fv.addFeature((DBoW2::NodeId)node[k], (unsigned)idx[s]);
}
struct Pool {   // map points with contiguous addresses (shared_ptr order = index order) and no-op deleters
  std::unique_ptr<MapPoint[]> store; int n;
  explicit Pool(int n_) : store(new MapPoint[n_ > 0 ? n_ : 1]), n(n_) {}
  mpptr at(int i) { return mpptr(&store[i], [](MapPoint*) {}); }
  int index(const mpptr& p) const { return p ? (int)(p.get() - store.get()) : -1; }
};
}  // namespace

extern "C" {

// Frame::GetFeaturesInArea (Frame.cpp:200-253) for a batch of queries: CSR of candidate indices in the reference's enumeration order
int64_t ref_grid_candidates(const float* kx, const float* ky, const int32_t* oct, int N, float minX, float minY, float maxX, float maxY, const float* qx,
                            const float* qy, const float* qr, const int32_t* qminl, const int32_t* qmaxl, int Q, int32_t* off, int32_t* idx, int64_t cap) {
  Frame F;
  const float sf[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  std::vector<uint8_t> d((size_t)N * 32, 0);
  setup_frame(F, kx, ky, oct, nullptr, d.data(), N, minX, minY, maxX, maxY, sf, 8);
  int64_t n = 0;
  for (int q = 0; q < Q; q++) {
    off[q] = (int32_t)n;
    const std::vector<size_t> v = F.GetFeaturesInArea(qx[q], qy[q], qr[q], qminl[q], qmaxl[q]);
    for (size_t k = 0; k < v.size(); k++) { if (idx && n < cap) idx[n] = (int32_t)v[k]; n++; }
  }
  off[Q] = (int32_t)n;
  return n;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<mpptr>&, th)   (ORBmatcher.cpp:71-148); flat layout = ora_search_by_projection_mp
int ref_search_by_projection_mp(const float* kx, const float* ky, const int32_t* oct, const uint8_t* fdesc, int N, float minX, float minY, float maxX,
                                float maxY, const float* scale_factors, int n_mp, const uint8_t* mp_in_view, const float* mp_proj_x,
                                const float* mp_proj_y, const int32_t* mp_level, const float* mp_view_cos, const uint8_t* mp_desc, float th,
                                float nnratio, int32_t* frame_mp) {
  Frame F;
  setup_frame(F, kx, ky, oct, nullptr, fdesc, N, minX, minY, maxX, maxY, scale_factors, 8);
  Pool claimed(1), pts(n_mp);
  claimed.store[0].nObs = 1;                                             // "F.mvpMapPoints[idx] && Observations() > 0"
  for (int i = 0; i < N; i++) if (frame_mp[i] >= 0) F.mvpMapPoints[i] = claimed.at(0);
  std::vector<mpptr> vpMapPoints(n_mp);
  for (int i = 0; i < n_mp; i++) {
    MapPoint& p = pts.store[i];
    p.mbTrackInView = mp_in_view[i] != 0; p.mTrackProjX = mp_proj_x[i]; p.mTrackProjY = mp_proj_y[i]; p.mnTrackScaleLevel = mp_level[i];
    p.mTrackViewCos = mp_view_cos[i]; p.mDescriptor = desc_row(mp_desc + 32 * (size_t)i);
    p.nObs = 1;   // a local map point has observers: once assigned it blocks later candidates, as in the reference
    vpMapPoints[i] = pts.at(i);
  }
  cslam::ORBmatcher matcher(nnratio, true);
  const int n = matcher.SearchByProjection(F, vpMapPoints, th);
  for (int i = 0; i < N; i++) { const mpptr& p = F.mvpMapPoints[i]; if (p && p.get() >= pts.store.get() && p.get() < pts.store.get() + n_mp) frame_mp[i] = pts.index(p); }
  return n;
}

// ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th)   (ORBmatcher.cpp:1350-1476).  The reference projects the last frame's map
// points itself (cv::Mat f32 arithmetic, :1381-1397); l_valid / l_u / l_v return that projection (same expressions, same look-alike cv:: arithmetic)
// so that the flat oracle, which takes the projection as input, can be given identical numbers.
int ref_search_by_projection_last(const float* kx, const float* ky, const int32_t* oct, const float* kangle, const uint8_t* fdesc, int N, float minX,
                                  float minY, float maxX, float maxY, const float* scale_factors, const float* Tcw16, const float* K4, int n_last,
                                  const float* Tlw16, const uint8_t* l_has_mp, const uint8_t* l_outlier, const float* l_Xw, const int32_t* l_octave,
                                  const float* l_angle, const uint8_t* l_mp_desc, float th, int check_orientation, int32_t* cur_mp,
                                  uint8_t* l_valid, float* l_u, float* l_v) {
  Frame C, L;
  setup_frame(C, kx, ky, oct, kangle, fdesc, N, minX, minY, maxX, maxY, scale_factors, 8);
  C.fx = K4[0]; C.fy = K4[1]; C.cx = K4[2]; C.cy = K4[3];
  C.mTcw = cv::Mat(4, 4, CV_32F); std::memcpy(C.mTcw.data, Tcw16, 64);
  L.N = n_last;
  L.mTcw = cv::Mat(4, 4, CV_32F); std::memcpy(L.mTcw.data, Tlw16, 64);
  L.mvKeys.resize(n_last); L.mvKeysUn.resize(n_last); L.mvpMapPoints.assign(n_last, mpptr()); L.mvbOutlier.assign(n_last, false);
  Pool pts(n_last), claimed(1);
  claimed.store[0].nObs = 1;
  for (int i = 0; i < N; i++) if (cur_mp[i] >= 0) C.mvpMapPoints[i] = claimed.at(0);
  const cv::Mat Rcw = C.mTcw.rowRange(0, 3).colRange(0, 3);
  const cv::Mat tcw = C.mTcw.rowRange(0, 3).col(3);
  for (int i = 0; i < n_last; i++) {
    L.mvKeys[i] = cv::KeyPoint(0, 0, 31.f, l_angle[i], 0, l_octave[i]);
    L.mvKeysUn[i] = L.mvKeys[i];
    L.mvbOutlier[i] = l_outlier[i] != 0;
    l_valid[i] = 0; l_u[i] = 0; l_v[i] = 0;
    if (!l_has_mp[i]) continue;
    MapPoint& p = pts.store[i];
    cv::Mat pos(3, 1, CV_32F);
    for (int c = 0; c < 3; c++) pos.at<float>(c) = l_Xw[3 * (size_t)i + c];
    p.SetWorldPos(pos, false);
    p.mDescriptor = desc_row(l_mp_desc + 32 * (size_t)i);
    p.nObs = 1;
    L.mvpMapPoints[i] = pts.at(i);
    if (l_outlier[i]) continue;
    // the projection as written at :1381-1397
    cv::Mat x3Dw = p.GetWorldPos();
    cv::Mat x3Dc = Rcw * x3Dw + tcw;
    const float xc = x3Dc.at<float>(0), yc = x3Dc.at<float>(1);
    const float invzc = 1.0 / x3Dc.at<float>(2);
    if (invzc < 0) continue;
    float u = C.fx * xc * invzc + C.cx, v = C.fy * yc * invzc + C.cy;
    if (u < Frame::mnMinX || u > Frame::mnMaxX) continue;
    if (v < Frame::mnMinY || v > Frame::mnMaxY) continue;
    l_valid[i] = 1; l_u[i] = u; l_v[i] = v;
  }
  cslam::ORBmatcher matcher(0.9f, check_orientation != 0);
  const int n = matcher.SearchByProjection(C, L, th);
  for (int i = 0; i < N; i++) {
    const mpptr& p = C.mvpMapPoints[i];
    if (!p) continue;
    if (p.get() >= pts.store.get() && p.get() < pts.store.get() + n_last) cur_mp[i] = pts.index(p);
  }
  // features whose assignment the rotation check removed are null again: report them as free unless they were claimed on entry
  for (int i = 0; i < N; i++) if (!C.mvpMapPoints[i]) cur_mp[i] = -1;
  return n;
}

// ORBmatcher::SearchByBoW(kfptr, Frame&, vector<mpptr>&)   (ORBmatcher.cpp:178-306); flat layout = ora_search_by_bow_kf_frame
int ref_search_by_bow_kf_frame(const int32_t* kf_node, const int32_t* kf_off, const int32_t* kf_idx, int kf_nn, const int32_t* f_node,
                               const int32_t* f_off, const int32_t* f_idx, int f_nn, const uint8_t* kf_has_mp, const uint8_t* kf_desc,
                               const float* kf_angle, int kf_n, const uint8_t* f_desc, const float* f_angle, int f_n, float nnratio, int check_ori,
                               int32_t* matches_f) {
  std::unique_ptr<KeyFrame> kfs(new KeyFrame());
  kfptr pKF(kfs.get(), [](KeyFrame*) {});
  pKF->N = kf_n;
  pKF->mvKeysUn.resize(kf_n);
  for (int i = 0; i < kf_n; i++) pKF->mvKeysUn[i] = cv::KeyPoint(0, 0, 31.f, kf_angle[i], 0, 0);
  pKF->mDescriptors = desc_mat(kf_desc, kf_n);
  Pool pts(kf_n);
  pKF->mvpMapPoints.assign(kf_n, mpptr());
  for (int i = 0; i < kf_n; i++) if (kf_has_mp[i]) pKF->mvpMapPoints[i] = pts.at(i);
  fill_fv(pKF->mFeatVec, kf_node, kf_off, kf_idx, kf_nn);
  Frame F;
  F.N = f_n;
  F.mvKeysUn.resize(f_n);
  for (int i = 0; i < f_n; i++) F.mvKeysUn[i] = cv::KeyPoint(0, 0, 31.f, f_angle[i], 0, 0);
  F.mvKeys = F.mvKeysUn;
  F.mDescriptors = desc_mat(f_desc, f_n);
  fill_fv(F.mFeatVec, f_node, f_off, f_idx, f_nn);
  std::vector<mpptr> vpMapPointMatches;
  cslam::ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchByBoW(pKF, F, vpMapPointMatches);
  for (int i = 0; i < f_n; i++) matches_f[i] = (i < (int)vpMapPointMatches.size()) ? pts.index(vpMapPointMatches[i]) : -1;
  return n;
}

// ORBmatcher::SearchByBoW(kfptr, kfptr, vector<mpptr>&)   (ORBmatcher.cpp:565-698); flat layout = ora_search_by_bow_kf_kf
int ref_search_by_bow_kf_kf(const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const int32_t* n2, const int32_t* o2, const int32_t* i2, int nn2,
                            const uint8_t* has_mp1, const uint8_t* has_mp2, const uint8_t* desc1, const float* angle1, int N1, const uint8_t* desc2,
                            const float* angle2, int N2, float nnratio, int check_ori, int32_t* matches12) {
  std::unique_ptr<KeyFrame[]> kfs(new KeyFrame[2]);
  kfptr k1(&kfs[0], [](KeyFrame*) {}), k2(&kfs[1], [](KeyFrame*) {});
  Pool p1(N1), p2(N2);
  auto setup = [](kfptr& k, Pool& p, const uint8_t* has, const uint8_t* desc, const float* angle, int N, const int32_t* node, const int32_t* off, const int32_t* idx, int nn) {
    k->N = N;
    k->mvKeysUn.resize(N);
    for (int i = 0; i < N; i++) k->mvKeysUn[i] = cv::KeyPoint(0, 0, 31.f, angle[i], 0, 0);
    k->mDescriptors = desc_mat(desc, N);
    k->mvpMapPoints.assign(N, mpptr());
    for (int i = 0; i < N; i++) if (has[i]) k->mvpMapPoints[i] = p.at(i);
    fill_fv(k->mFeatVec, node, off, idx, nn);
  };
  setup(k1, p1, has_mp1, desc1, angle1, N1, n1, o1, i1, nn1);
  setup(k2, p2, has_mp2, desc2, angle2, N2, n2, o2, i2, nn2);
  std::vector<mpptr> vpMatches12;
  cslam::ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchByBoW(k1, k2, vpMatches12);
  for (int i = 0; i < N1; i++) matches12[i] = (i < (int)vpMatches12.size()) ? p2.index(vpMatches12[i]) : -1;
  return n;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs)   (ORBmatcher.cpp:700-852); flat layout = ora_search_for_triangulation, except that the
// reference computes the epipole from the keyframe poses itself (:706-712): T1 / T2 = the two poses (4x4 f32), K4 of keyframe 2; ex / ey are returned for the oracle
int ref_search_for_triangulation(const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const int32_t* n2, const int32_t* o2, const int32_t* i2, int nn2,
                                 const uint8_t* has_mp1, const uint8_t* has_mp2, const uint8_t* desc1, const float* x1, const float* y1, const float* angle1, int N1,
                                 const uint8_t* desc2, const float* x2, const float* y2, const int32_t* oct2, const float* angle2, int N2, const float* F12,
                                 const float* T1, const float* T2, const float* K4, const float* sigma2_2, const float* sf2, int check_ori, int32_t* matches12,
                                 float* ex_out, float* ey_out) {
  std::unique_ptr<KeyFrame[]> kfs(new KeyFrame[2]);
  kfptr k1(&kfs[0], [](KeyFrame*) {}), k2(&kfs[1], [](KeyFrame*) {});
  Pool p1(N1), p2(N2);
  auto setup = [](kfptr& k, Pool& p, const uint8_t* has, const uint8_t* desc, const float* x, const float* y, const int32_t* oct, const float* angle, int N,
                  const int32_t* node, const int32_t* off, const int32_t* idx, int nn, const float* T) {
    k->N = N;
    k->mvKeysUn.resize(N);
    for (int i = 0; i < N; i++) k->mvKeysUn[i] = cv::KeyPoint(x[i], y[i], 31.f, angle[i], 0, oct ? oct[i] : 0);
    k->mDescriptors = desc_mat(desc, N);
    k->mvpMapPoints.assign(N, mpptr());
    for (int i = 0; i < N; i++) if (has[i]) k->mvpMapPoints[i] = p.at(i);
    fill_fv(k->mFeatVec, node, off, idx, nn);
    cv::Mat Tm(4, 4, CV_32F); std::memcpy(Tm.data, T, 64);
    k->SetPose(Tm, false);
  };
  setup(k1, p1, has_mp1, desc1, x1, y1, nullptr, angle1, N1, n1, o1, i1, nn1, T1);
  setup(k2, p2, has_mp2, desc2, x2, y2, oct2, angle2, N2, n2, o2, i2, nn2, T2);
  k2->fx = K4[0]; k2->fy = K4[1]; k2->cx = K4[2]; k2->cy = K4[3];
  k2->mvLevelSigma2.assign(sigma2_2, sigma2_2 + 8); k2->mvScaleFactors.assign(sf2, sf2 + 8);
  {   // the epipole as written at :706-712
    cv::Mat Cw = k1->GetCameraCenter();
    cv::Mat C2 = k2->GetRotation() * Cw + k2->GetTranslation();
    const float invz = 1.0f / C2.at<float>(2);
    *ex_out = k2->fx * C2.at<float>(0) * invz + k2->cx;
    *ey_out = k2->fy * C2.at<float>(1) * invz + k2->cy;
  }
  cv::Mat F(3, 3, CV_32F); std::memcpy(F.data, F12, 36);
  std::vector<std::pair<size_t, size_t> > vMatchedPairs;
  cslam::ORBmatcher matcher(0.6f, check_ori != 0);
  const int n = matcher.SearchForTriangulation(k1, k2, F, vMatchedPairs);
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  for (auto& pr : vMatchedPairs) matches12[pr.first] = (int32_t)pr.second;
  return n;
}

// The fan-out of LocalMapping::CreateNewMapPoints (Mapping.cpp:277-470): ONE new keyframe against n_nb covisible neighbours, SearchForTriangulation per neighbour in
// order; between the calls every matched pair becomes a map point of BOTH keyframes (Mapping.cpp:437-452: AddMapPoint on both), so later calls see fewer
// free features.  The keyframes live for the whole call and keyframe 1 lists the neighbours as its covisibility neighbours (GetBestCovisibilityKeyFrames).
// Arrays of pointers: one entry per neighbour.  matches_out [n_nb][N1].
int ref_triangulation_fan_out(const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const uint8_t* has_mp1, const uint8_t* desc1, const float* x1,
                              const float* y1, const float* angle1, int N1, const float* T1, int n_nb, const int32_t* const* n2, const int32_t* const* o2,
                              const int32_t* const* i2, const int32_t* nn2, const uint8_t* const* has_mp2, const uint8_t* const* desc2, const float* const* x2,
                              const float* const* y2, const int32_t* const* oct2, const float* const* angle2, const int32_t* N2, const float* F12s /* n_nb x 9 */,
                              const float* T2s /* n_nb x 16 */, const float* K4, const float* sigma2_2, const float* sf2, int check_ori, int32_t* matches_out,
                              int32_t* n_out) {
  std::unique_ptr<KeyFrame[]> kfs(new KeyFrame[n_nb + 1]);
  std::vector<kfptr> k(n_nb + 1);
  for (int j = 0; j <= n_nb; j++) k[j] = kfptr(&kfs[j], [](KeyFrame*) {});
  std::vector<std::unique_ptr<Pool>> pools;
  auto setup = [&](kfptr& kf, const uint8_t* has, const uint8_t* desc, const float* x, const float* y, const int32_t* oct, const float* angle, int N,
                   const int32_t* node, const int32_t* off, const int32_t* idx, int nn, const float* T, size_t id) {
    pools.emplace_back(new Pool(2 * N));   // (the second half: points created during the call)
    Pool& p = *pools.back();
    kf->N = N;
    kf->mId = std::make_pair(id, (size_t)0); kf->mUniqueId = id;
    kf->mvKeysUn.resize(N);
    for (int i = 0; i < N; i++) kf->mvKeysUn[i] = cv::KeyPoint(x[i], y[i], 31.f, angle[i], 0, oct ? oct[i] : 0);
    kf->mDescriptors = desc_mat(desc, N);
    kf->mvpMapPoints.assign(N, mpptr());
    for (int i = 0; i < N; i++) if (has[i]) kf->mvpMapPoints[i] = p.at(i);
    fill_fv(kf->mFeatVec, node, off, idx, nn);
    cv::Mat Tm(4, 4, CV_32F); std::memcpy(Tm.data, T, 64);
    kf->SetPose(Tm, false);
    kf->fx = K4[0]; kf->fy = K4[1]; kf->cx = K4[2]; kf->cy = K4[3];
    kf->mvLevelSigma2.assign(sigma2_2, sigma2_2 + 8); kf->mvScaleFactors.assign(sf2, sf2 + 8);
  };
  setup(k[0], has_mp1, desc1, x1, y1, nullptr, angle1, N1, n1, o1, i1, nn1, T1, 100);
  for (int j = 0; j < n_nb; j++) {
    setup(k[j + 1], has_mp2[j], desc2[j], x2[j], y2[j], oct2[j], angle2[j], N2[j], n2[j], o2[j], i2[j], nn2[j], T2s + 16 * (size_t)j, 101 + (size_t)j);
    k[0]->mvpOrderedConnectedKeyFrames.push_back(k[j + 1]); k[0]->mvOrderedWeights.push_back(1000 - j); k[0]->mConnectedKeyFrameWeights[k[j + 1]] = 1000 - j;
  }
  cslam::ORBmatcher matcher(0.6f, check_ori != 0);
  for (int j = 0; j < n_nb; j++) {
    cv::Mat F(3, 3, CV_32F); std::memcpy(F.data, F12s + 9 * (size_t)j, 36);
    std::vector<std::pair<size_t, size_t> > vMatchedPairs;
    n_out[j] = matcher.SearchForTriangulation(k[0], k[j + 1], F, vMatchedPairs);
    int32_t* m = matches_out + (size_t)j * N1;
    for (int i = 0; i < N1; i++) m[i] = -1;
    for (auto& pr : vMatchedPairs) {
      m[pr.first] = (int32_t)pr.second;
      // the triangulated pair becomes a map point of both keyframes (Mapping.cpp:437-452)
      mpptr pMP = pools[0]->at(N1 + (int)pr.first);
      k[0]->mvpMapPoints[pr.first] = pMP;
      k[j + 1]->mvpMapPoints[pr.second] = pMP;
    }
  }
  return 0;
}

namespace {
// a keyframe as the projected searches read it: pose, intrinsics, keypoints, descriptors, pyramid tables, image bounds and the grid (filled with
// the reference's own Frame::AssignFeaturesToGrid and copied, as the KeyFrame constructor copies F.mGrid, KeyFrame.cpp:60-66)
void setup_keyframe(kfptr& k, const float* kx, const float* ky, const int32_t* oct, const uint8_t* desc, int N, float minX, float minY, float maxX, float maxY,
                    const float* sf, const float* inv_sigma2, const float* K4, const float* T16) {
  Frame F;
  setup_frame(F, kx, ky, oct, nullptr, desc, N, minX, minY, maxX, maxY, sf, 8);
  k->N = N; k->mvKeysUn = F.mvKeysUn; k->mvKeys = F.mvKeys; k->mDescriptors = F.mDescriptors;
  k->mvpMapPoints.assign(N, mpptr());
  k->mnGridCols = FRAME_GRID_COLS; k->mnGridRows = FRAME_GRID_ROWS;
  k->mfGridElementWidthInv = Frame::mfGridElementWidthInv; k->mfGridElementHeightInv = Frame::mfGridElementHeightInv;
  k->mnMinX = (int)minX; k->mnMinY = (int)minY; k->mnMaxX = (int)maxX; k->mnMaxY = (int)maxY;
  k->mGrid.assign(FRAME_GRID_COLS, std::vector<std::vector<size_t> >(FRAME_GRID_ROWS));
  for (int i = 0; i < FRAME_GRID_COLS; i++) for (int j = 0; j < FRAME_GRID_ROWS; j++) k->mGrid[i][j] = F.mGrid[i][j];
  k->mnScaleLevels = 8; k->mfScaleFactor = sf[1] / sf[0]; k->mfLogScaleFactor = std::log(k->mfScaleFactor);
  k->mvScaleFactors.assign(sf, sf + 8); k->mvInvLevelSigma2.assign(inv_sigma2, inv_sigma2 + 8);
  k->mvLevelSigma2.resize(8); for (int i = 0; i < 8; i++) k->mvLevelSigma2[i] = 1.0f / inv_sigma2[i];
  k->fx = K4[0]; k->fy = K4[1]; k->cx = K4[2]; k->cy = K4[3];
  cv::Mat Tm(4, 4, CV_32F); std::memcpy(Tm.data, T16, 64);
  k->SetPose(Tm, false);
}
void setup_points(Pool& pts, int n, const float* Xw, const float* normal, const float* dmin, const float* dmax, const uint8_t* desc, int nobs) {
  for (int i = 0; i < n; i++) {
    MapPoint& p = pts.store[i];
    cv::Mat pos(3, 1, CV_32F), nr(3, 1, CV_32F);
    for (int c = 0; c < 3; c++) { pos.at<float>(c) = Xw[3 * (size_t)i + c]; nr.at<float>(c) = normal[3 * (size_t)i + c]; }
    p.SetWorldPos(pos, false); p.mNormalVector = nr;
    p.mfMinDistance = dmin[i]; p.mfMaxDistance = dmax[i];
    p.mDescriptor = desc_row(desc + 32 * (size_t)i);
    p.nObs = nobs;
  }
}
// the tests that precede the candidate loop of Fuse / SearchByProjection(kfptr, Scw ...) as written at ORBmatcher.cpp:868-913 / 326-353 (depth, image bounds,
// distance range, viewing angle, PredictScale — the reference's own MapPoint::PredictScale): what the flat oracle takes as input
void project_points(kfptr& k, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow, Pool& pts, int n, uint8_t* valid, float* u_out, float* v_out, int32_t* level) {
  for (int i = 0; i < n; i++) {
    valid[i] = 0; u_out[i] = 0; v_out[i] = 0; level[i] = 0;
    MapPoint& p = pts.store[i];
    cv::Mat p3Dw = p.GetWorldPos();
    cv::Mat p3Dc = Rcw * p3Dw + tcw;
    if (p3Dc.at<float>(2) < 0.0f) continue;
    const float invz = 1 / p3Dc.at<float>(2);
    const float x = p3Dc.at<float>(0) * invz, y = p3Dc.at<float>(1) * invz;
    const float u = k->fx * x + k->cx, v = k->fy * y + k->cy;
    if (!k->IsInImage(u, v)) continue;
    const float maxDistance = p.GetMaxDistanceInvariance(), minDistance = p.GetMinDistanceInvariance();
    cv::Mat PO = p3Dw - Ow;
    const float dist3D = cv::norm(PO);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    cv::Mat Pn = p.GetNormal();
    if (PO.dot(Pn) < 0.5 * dist3D) continue;
    valid[i] = 1; u_out[i] = u; v_out[i] = v; level[i] = p.PredictScale(dist3D, k);
  }
}
}  // namespace

// ORBmatcher::Fuse(pKF, vpMapPoints, th)   (ORBmatcher.cpp:854-993).  kf_has_mp[j]: feature j already holds a (well observed) map point.  best_idx[i] = the
// feature the call fused point i with (recovered from what it did to the map: AddObservation / AddMapPoint or Replace), -1 = not fused.
int ref_fuse(const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX, float minY, float maxX, float maxY, const float* sf,
             const float* inv_sigma2, const float* K4, const float* T16, const uint8_t* kf_has_mp, int n_pts, const float* Xw, const float* normal, const float* dmin,
             const float* dmax, const uint8_t* pdesc, float th, int32_t* best_idx, uint8_t* valid, float* u, float* v, int32_t* level) {
  std::unique_ptr<KeyFrame> store(new KeyFrame());
  kfptr k(store.get(), [](KeyFrame*) {});
  setup_keyframe(k, kx, ky, oct, kdesc, N, minX, minY, maxX, maxY, sf, inv_sigma2, K4, T16);
  Pool inkf(N), pts(n_pts);
  for (int j = 0; j < N; j++) if (kf_has_mp[j]) { inkf.store[j].nObs = 5; inkf.store[j].mObservations[k] = (size_t)j; k->mvpMapPoints[j] = inkf.at(j); }
  setup_points(pts, n_pts, Xw, normal, dmin, dmax, pdesc, 2);
  cv::Mat Rcw = k->GetRotation(), tcw = k->GetTranslation(), Ow = k->GetCameraCenter();
  project_points(k, Rcw, tcw, Ow, pts, n_pts, valid, u, v, level);
  std::vector<mpptr> vp(n_pts);
  for (int i = 0; i < n_pts; i++) vp[i] = pts.at(i);
  cslam::ORBmatcher matcher(0.6f, true);
  const int n = matcher.Fuse(k, vp, th);
  for (int i = 0; i < n_pts; i++) {
    best_idx[i] = -1;
    MapPoint& p = pts.store[i];
    mpptr target = p.mpReplaced ? p.mpReplaced : pts.at(i);   // replaced by the point already in the keyframe, or added itself
    for (int j = 0; j < N; j++) if (k->mvpMapPoints[j] == target && (p.mpReplaced || p.IsInKeyFrame(k))) { best_idx[i] = j; break; }
  }
  return n;
}

// The fan-out of LocalMapping::SearchInNeighbors (Mapping.cpp:469-503), replayed as the reference writes it: the current keyframe's best covisibility neighbours become
// fuse targets and are marked (mFuseTargetForKF), each contributes up to five second neighbours (a LATER first-level neighbour listed there is not marked yet and is
// pushed twice), then matcher.Fuse(pKFi, vpMapPointMatches) runs for every target in order on the SAME vector of the current keyframe's points.  All S keyframes share
// one feature set (frame geometry) and differ in pose and in which features hold points.  nb1 [n1]: the current keyframe's covisibility list (indices into the S
// keyframes); nb2 [S][5]: every keyframe's own best covisibility list (-1 = none, -2 = the current keyframe).  Per call c: call_target[c], n_out[c] and best_idx[c][i] =
// the feature point i was fused with IN THAT CALL (recovered from what the call did to the map), -1 otherwise.
int ref_fuse_fan_out(const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX, float minY, float maxX, float maxY, const float* sf,
                     const float* inv_sigma2, const float* K4, int S, const float* T16s, const uint8_t* kf_has_mp, int n1, const int32_t* nb1, const int32_t* nb2,
                     int n_pts, const float* Xw, const float* normal, const float* dmin, const float* dmax, const uint8_t* pdesc, float th, int max_calls,
                     int32_t* call_target, int32_t* n_out, int32_t* best_idx) {
  std::unique_ptr<KeyFrame[]> kfs(new KeyFrame[S + 1]);
  std::vector<kfptr> k(S + 1);
  for (int j = 0; j <= S; j++) k[j] = kfptr(&kfs[j], [](KeyFrame*) {});
  std::vector<std::unique_ptr<Pool>> inkf;
  for (int s = 0; s < S; s++) {
    setup_keyframe(k[s], kx, ky, oct, kdesc, N, minX, minY, maxX, maxY, sf, inv_sigma2, K4, T16s + 16 * (size_t)s);
    k[s]->mId = std::make_pair((size_t)(10 + s), (size_t)0); k[s]->mUniqueId = 10 + s;
    inkf.emplace_back(new Pool(N));
    for (int j = 0; j < N; j++) if (kf_has_mp[(size_t)s * N + j]) { inkf[s]->store[j].nObs = 3 + (j % 5); inkf[s]->store[j].mObservations[k[s]] = (size_t)j; k[s]->mvpMapPoints[j] = inkf[s]->at(j); }
  }
  kfptr cur = k[S];
  cur->mId = std::make_pair((size_t)500, (size_t)0); cur->mUniqueId = 500; cur->N = n_pts;
  Pool pts(n_pts);
  setup_points(pts, n_pts, Xw, normal, dmin, dmax, pdesc, 4);
  cur->mvpMapPoints.assign(n_pts, mpptr());
  for (int i = 0; i < n_pts; i++) { cur->mvpMapPoints[i] = pts.at(i); pts.store[i].mObservations[cur] = (size_t)i; }
  for (int a = 0; a < n1; a++) { cur->mvpOrderedConnectedKeyFrames.push_back(k[nb1[a]]); cur->mvOrderedWeights.push_back(1000 - a); cur->mConnectedKeyFrameWeights[k[nb1[a]]] = 1000 - a; }
  for (int s = 0; s < S; s++)
    for (int b = 0; b < 5; b++) {
      const int o = nb2[5 * s + b];
      if (o == -1) continue;
      kfptr other = o == -2 ? cur : k[o];
      k[s]->mvpOrderedConnectedKeyFrames.push_back(other); k[s]->mvOrderedWeights.push_back(900 - b); k[s]->mConnectedKeyFrameWeights[other] = 900 - b;
    }
  // ---- Mapping.cpp:472-492 ----
  const std::vector<kfptr> vpNeighKFs = cur->GetBestCovisibilityKeyFrames(20);
  std::vector<kfptr> vpTargetKFs;
  for (std::vector<kfptr>::const_iterator vit = vpNeighKFs.begin(); vit != vpNeighKFs.end(); vit++) {
    kfptr pKFi = *vit;
    if (pKFi->isBad() || pKFi->mFuseTargetForKF == cur->mId) continue;
    vpTargetKFs.push_back(pKFi);
    pKFi->mFuseTargetForKF = cur->mId;
    const std::vector<kfptr> vpSecondNeighKFs = pKFi->GetBestCovisibilityKeyFrames(5);
    for (std::vector<kfptr>::const_iterator vit2 = vpSecondNeighKFs.begin(); vit2 != vpSecondNeighKFs.end(); vit2++) {
      kfptr pKFi2 = *vit2;
      if (pKFi2->isBad() || pKFi2->mFuseTargetForKF == cur->mId || pKFi2->mId == cur->mId) continue;
      vpTargetKFs.push_back(pKFi2);
    }
  }
  // ---- Mapping.cpp:495-503 ----
  cslam::ORBmatcher matcher(0.6f, true);
  std::vector<mpptr> vpMapPointMatches = cur->GetMapPointMatches();
  int c = 0;
  for (std::vector<kfptr>::iterator vit = vpTargetKFs.begin(); vit != vpTargetKFs.end() && c < max_calls; vit++, c++) {
    kfptr pKFi = *vit;
    std::vector<char> in_before(n_pts), rep_before(n_pts);
    for (int i = 0; i < n_pts; i++) { in_before[i] = pts.store[i].IsInKeyFrame(pKFi); rep_before[i] = pts.store[i].mpReplaced != nullptr; }
    n_out[c] = matcher.Fuse(pKFi, vpMapPointMatches, th);
    call_target[c] = (int32_t)(pKFi.get() - kfs.get());
    int32_t* b = best_idx + (size_t)c * n_pts;
    for (int i = 0; i < n_pts; i++) {
      b[i] = -1;
      MapPoint& p = pts.store[i];
      if (!rep_before[i] && p.mpReplaced) b[i] = p.mpReplaced->GetIndexInKeyFrame(pKFi);
      else if (!in_before[i] && !p.isBad() && p.IsInKeyFrame(pKFi)) b[i] = p.GetIndexInKeyFrame(pKFi);
    }
  }
  return c;
}

namespace {
void decompose_scw(const float* S16, cv::Mat& Rcw, cv::Mat& tcw, cv::Mat& Ow) {   // ORBmatcher.cpp:316-321 / 1004-1008
  cv::Mat Scw(4, 4, CV_32F); std::memcpy(Scw.data, S16, 64);
  cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
  const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
  Rcw = sRcw / scw;
  tcw = Scw.rowRange(0, 3).col(3) / scw;
  Ow = -Rcw.t() * tcw;
}
}  // namespace

// ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)   (ORBmatcher.cpp:995-1122): Sim3 pose, no chi2 gate.  best_idx[i] as above; replace_kf_feature[i] = the
// feature whose map point was returned in vpReplacePoint[i] (-1: none)
int ref_fuse_sim3(const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX, float minY, float maxX, float maxY, const float* sf,
                  const float* inv_sigma2, const float* K4, const float* S16, const uint8_t* kf_has_mp, int n_pts, const float* Xw, const float* normal,
                  const float* dmin, const float* dmax, const uint8_t* pdesc, float th, int32_t* best_idx, uint8_t* valid, float* u, float* v, int32_t* level) {
  std::unique_ptr<KeyFrame> store(new KeyFrame());
  kfptr k(store.get(), [](KeyFrame*) {});
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  setup_keyframe(k, kx, ky, oct, kdesc, N, minX, minY, maxX, maxY, sf, inv_sigma2, K4, I16);
  Pool inkf(N), pts(n_pts);
  for (int j = 0; j < N; j++) if (kf_has_mp[j]) { inkf.store[j].nObs = 5; inkf.store[j].mObservations[k] = (size_t)j; k->mvpMapPoints[j] = inkf.at(j); }
  setup_points(pts, n_pts, Xw, normal, dmin, dmax, pdesc, 2);
  cv::Mat Rcw, tcw, Ow;
  decompose_scw(S16, Rcw, tcw, Ow);
  project_points(k, Rcw, tcw, Ow, pts, n_pts, valid, u, v, level);
  std::vector<mpptr> vp(n_pts), vpReplacePoint(n_pts);
  for (int i = 0; i < n_pts; i++) vp[i] = pts.at(i);
  cv::Mat Scw(4, 4, CV_32F); std::memcpy(Scw.data, S16, 64);
  cslam::ORBmatcher matcher(0.6f, true);
  const int n = matcher.Fuse(k, Scw, vp, th, vpReplacePoint);
  for (int i = 0; i < n_pts; i++) {
    best_idx[i] = -1;
    if (vpReplacePoint[i]) best_idx[i] = vpReplacePoint[i]->GetIndexInKeyFrame(k);   // the occupant: a point of the keyframe, or an earlier point of this call
    else if (pts.store[i].IsInKeyFrame(k)) best_idx[i] = pts.store[i].GetIndexInKeyFrame(k);
  }
  return n;
}

// ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)   (ORBmatcher.cpp:308-446).  matched[N] in/out: >= 0 = vpMatched[idx] is set (on entry by
// an unrelated point: 1 000 000; on exit by point i: i).  existing_idx[i] >= 0: point i is already observed by the keyframe at that feature (the CCM-specific
// remap branch :412-432, which does not count as a match).
int ref_search_by_projection_sim3(const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX, float minY, float maxX, float maxY,
                                  const float* sf, const float* inv_sigma2, const float* K4, const float* S16, int n_pts, const float* Xw, const float* normal,
                                  const float* dmin, const float* dmax, const uint8_t* pdesc, const int32_t* existing_idx, int th, int32_t* matched, uint8_t* valid,
                                  float* u, float* v, int32_t* level) {
  std::unique_ptr<KeyFrame> store(new KeyFrame());
  kfptr k(store.get(), [](KeyFrame*) {});
  float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  setup_keyframe(k, kx, ky, oct, kdesc, N, minX, minY, maxX, maxY, sf, inv_sigma2, K4, I16);
  Pool pts(n_pts), other(1);
  setup_points(pts, n_pts, Xw, normal, dmin, dmax, pdesc, 2);
  for (int i = 0; i < n_pts; i++) if (existing_idx[i] >= 0) { pts.store[i].mObservations[k] = (size_t)existing_idx[i]; k->mvpMapPoints[existing_idx[i]] = pts.at(i); }
  cv::Mat Rcw, tcw, Ow;
  decompose_scw(S16, Rcw, tcw, Ow);
  project_points(k, Rcw, tcw, Ow, pts, n_pts, valid, u, v, level);
  std::vector<mpptr> vp(n_pts), vpMatched(N);
  for (int i = 0; i < n_pts; i++) vp[i] = pts.at(i);
  for (int j = 0; j < N; j++) if (matched[j] >= 0) vpMatched[j] = other.at(0);
  cv::Mat Scw(4, 4, CV_32F); std::memcpy(Scw.data, S16, 64);
  cslam::ORBmatcher matcher(0.75f, true);
  const int n = matcher.SearchByProjection(k, Scw, vp, vpMatched, th);
  for (int j = 0; j < N; j++) { const mpptr& p = vpMatched[j]; if (p && p.get() >= pts.store.get() && p.get() < pts.store.get() + n_pts) matched[j] = pts.index(p); }
  return n;
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)   (ORBmatcher.cpp:1124-1348).  Every feature j with has_mp{1,2}[j] holds a map point at
// Xw{1,2}[j]; pre12[i] >= 0: vpMatches12[i] is set on entry to KF2's point at that feature.  Outputs: matches12[i] = KF2 feature whose point the call wrote
// into vpMatches12[i] (-1 none; entries set on entry keep their value), and both directions' projection (valid, u, v, level) by the expressions of :1171-1205 /
// :1251-1285 for the flat oracle.
int ref_search_by_sim3(const float* kx1, const float* ky1, const int32_t* oct1, const uint8_t* desc1, int N1, const float* T1, const uint8_t* has_mp1, const float* Xw1,
                       const float* dmin1, const float* dmax1, const uint8_t* mdesc1, const float* kx2, const float* ky2, const int32_t* oct2, const uint8_t* desc2,
                       int N2, const float* T2, const uint8_t* has_mp2, const float* Xw2, const float* dmin2, const float* dmax2, const uint8_t* mdesc2, float minX,
                       float minY, float maxX, float maxY, const float* sf, const float* inv_sigma2, const float* K4, float s12, const float* R12_9,
                       const float* t12_3, float th, const int32_t* pre12, int32_t* matches12, uint8_t* valid1, float* u1, float* v1, int32_t* level1, uint8_t* valid2,
                       float* u2, float* v2, int32_t* level2) {
  std::unique_ptr<KeyFrame> s1(new KeyFrame()), s2(new KeyFrame());
  kfptr k1(s1.get(), [](KeyFrame*) {}), k2(s2.get(), [](KeyFrame*) {});
  setup_keyframe(k1, kx1, ky1, oct1, desc1, N1, minX, minY, maxX, maxY, sf, inv_sigma2, K4, T1);
  setup_keyframe(k2, kx2, ky2, oct2, desc2, N2, minX, minY, maxX, maxY, sf, inv_sigma2, K4, T2);
  Pool p1(N1), p2(N2);
  std::vector<float> nrm((size_t)3 * std::max(N1, N2), 0.0f);
  setup_points(p1, N1, Xw1, nrm.data(), dmin1, dmax1, mdesc1, 2);
  setup_points(p2, N2, Xw2, nrm.data(), dmin2, dmax2, mdesc2, 2);
  for (int j = 0; j < N1; j++) if (has_mp1[j]) { k1->mvpMapPoints[j] = p1.at(j); p1.store[j].mObservations[k1] = (size_t)j; }
  for (int j = 0; j < N2; j++) if (has_mp2[j]) { k2->mvpMapPoints[j] = p2.at(j); p2.store[j].mObservations[k2] = (size_t)j; }
  cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F);
  std::memcpy(R12.data, R12_9, 36); std::memcpy(t12.data, t12_3, 12);
  std::vector<mpptr> vpMatches12(N1);
  for (int i = 0; i < N1; i++) if (pre12[i] >= 0) vpMatches12[i] = p2.at(pre12[i]);
  {  // the projections as written in the method
    cv::Mat R1w = k1->GetRotation(), t1w = k1->GetTranslation(), R2w = k2->GetRotation(), t2w = k2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    auto dir = [&](Pool& P, int N, const uint8_t* has, const cv::Mat& Rw, const cv::Mat& tw, const cv::Mat& sR, const cv::Mat& t, kfptr& target, uint8_t* valid, float* u,
                   float* v, int32_t* level) {
      for (int i = 0; i < N; i++) {
        valid[i] = 0; u[i] = 0; v[i] = 0; level[i] = 0;
        if (!has[i]) continue;
        MapPoint& mp = P.store[i];
        cv::Mat p3Dw = mp.GetWorldPos();
        cv::Mat pa = Rw * p3Dw + tw;
        cv::Mat pb = sR * pa + t;
        if (pb.at<float>(2) < 0.0) continue;
        const float invz = 1.0 / pb.at<float>(2);
        const float x = pb.at<float>(0) * invz, y = pb.at<float>(1) * invz;
        const float uu = target->fx * x + target->cx, vv = target->fy * y + target->cy;
        if (!target->IsInImage(uu, vv)) continue;
        const float dist3D = cv::norm(pb);
        if (dist3D < mp.GetMinDistanceInvariance() || dist3D > mp.GetMaxDistanceInvariance()) continue;
        valid[i] = 1; u[i] = uu; v[i] = vv; level[i] = mp.PredictScale(dist3D, target);
      }
    };
    dir(p1, N1, has_mp1, R1w, t1w, sR21, t21, k2, valid1, u1, v1, level1);
    dir(p2, N2, has_mp2, R2w, t2w, sR12, t12, k1, valid2, u2, v2, level2);
  }
  cslam::ORBmatcher matcher(0.75f, true);
  const int n = matcher.SearchBySim3(k1, k2, vpMatches12, s12, R12, t12, th);
  for (int i = 0; i < N1; i++) matches12[i] = vpMatches12[i] ? p2.index(vpMatches12[i]) : -1;
  return n;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)   (ORBmatcher.cpp:448-563); flat layout = ora_search_for_initialization
int ref_search_for_initialization(const float* x1, const float* y1, const int32_t* oct1, const float* angle1, const uint8_t* desc1, int N1, const float* x2,
                                  const float* y2, const int32_t* oct2, const float* angle2, const uint8_t* desc2, int N2, float minX, float minY,
                                  float maxX, float maxY, float* prev_xy, int window, float nnratio, int check_ori, int32_t* matches12) {
  Frame F1, F2;
  const float sf[8] = {1, 1, 1, 1, 1, 1, 1, 1};
  setup_frame(F1, x1, y1, oct1, angle1, desc1, N1, minX, minY, maxX, maxY, sf, 8);
  setup_frame(F2, x2, y2, oct2, angle2, desc2, N2, minX, minY, maxX, maxY, sf, 8);
  std::vector<cv::Point2f> vbPrevMatched(N1);
  for (int i = 0; i < N1; i++) vbPrevMatched[i] = cv::Point2f(prev_xy[2 * i], prev_xy[2 * i + 1]);
  std::vector<int> vnMatches12;
  cslam::ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, window);
  for (int i = 0; i < N1; i++) { matches12[i] = vnMatches12[i]; prev_xy[2 * i] = vbPrevMatched[i].x; prev_xy[2 * i + 1] = vbPrevMatched[i].y; }
  return n;
}

}  // extern "C"

// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cpp:929-994) for P points: point p is observed by keyframes 0 .. n_p-1 (n_p = off[p+1]-off[p]) at feature 0 of
// each, whose descriptor is desc[off[p] + k]; keyframes are allocated contiguously so that the observation map (ordered by address) iterates in index order.
// best_local_idx[p] = which observation's descriptor the point took (first bytewise-equal one), -1 if none.
extern "C" void ref_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int P, int32_t* best_local_idx) {
  int max_n = 0;
  for (int p = 0; p < P; p++) max_n = std::max(max_n, off[p + 1] - off[p]);
  std::unique_ptr<KeyFrame[]> kfs(new KeyFrame[max_n > 0 ? max_n : 1]);
  for (int p = 0; p < P; p++) {
    const int n = off[p + 1] - off[p];
    MapPoint mp;
    for (int k = 0; k < n; k++) {
      kfs[k].mDescriptors = desc_row(desc + 32 * (size_t)(off[p] + k)).clone();
      kfptr kf(&kfs[k], [](KeyFrame*) {});
      mp.mObservations[kf] = 0;
    }
    mp.ComputeDistinctiveDescriptors();
    best_local_idx[p] = -1;
    if (mp.mDescriptor.empty()) continue;
    for (int k = 0; k < n; k++) if (std::memcmp(mp.mDescriptor.data, desc + 32 * (size_t)(off[p] + k), 32) == 0) { best_local_idx[p] = k; break; }
  }
}

// Frame::isInFrustum (Frame.cpp:139-198, with SetPose / UpdatePoseMatrices :125-137) for n points: what Tracking::SearchLocalPoints feeds M1 with.
extern "C" void ref_is_in_frustum(const float* T16, const float* K4, float minX, float minY, float maxX, float maxY, float scale_factor, int n_levels, int n,
                                  const float* Xw, const float* normal, const float* dmin, const float* dmax, float cos_limit, uint8_t* in_view, float* u, float* v,
                                  int32_t* level, float* view_cos, float* Ow_out) {
  boost::shared_ptr<Frame> F(new Frame());
  F->fx = K4[0]; F->fy = K4[1]; F->cx = K4[2]; F->cy = K4[3];
  Frame::mnMinX = minX; Frame::mnMinY = minY; Frame::mnMaxX = maxX; Frame::mnMaxY = maxY;
  F->mnScaleLevels = n_levels; F->mfScaleFactor = scale_factor; F->mfLogScaleFactor = std::log(scale_factor);
  cv::Mat Tm(4, 4, CV_32F); std::memcpy(Tm.data, T16, 64);
  F->SetPose(Tm);
  for (int c = 0; c < 3; c++) Ow_out[c] = F->mOw.at<float>(c);   // the f32 camera centre UpdatePoseMatrices produced
  Pool pts(n);
  std::vector<uint8_t> d((size_t)32 * (n > 0 ? n : 1), 0);
  setup_points(pts, n, Xw, normal, dmin, dmax, d.data(), 2);
  for (int i = 0; i < n; i++) {
    MapPoint& p = pts.store[i];
    const bool in = F->isInFrustum(pts.at(i), cos_limit);
    in_view[i] = (in && p.mbTrackInView) ? 1 : 0;
    u[i] = in ? p.mTrackProjX : 0; v[i] = in ? p.mTrackProjY : 0; level[i] = in ? p.mnTrackScaleLevel : 0; view_cos[i] = in ? p.mTrackViewCos : 0;
  }
}
