// ccm_host.cpp — implementation of the host-side mirror (see ccm_host.h).  Plain C++17, no HIP headers:
// everything device-side goes through the C ABI.
#include "ccm_host.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

namespace cslam {

static void check(int rc, ccm_ctx* ctx, const char* what) {
  if (rc != CCM_OK) throw infrastructure_ex(std::string(what) + ": " + ccm_last_error(ctx));
}

HipContext::HipContext(int device) { check(ccm_ctx_create(device, &ctx_), nullptr, "ccm_ctx_create"); }
HipContext::~HipContext() { ccm_ctx_destroy(ctx_); }

// ---- ORBextractor ----------------------------------------------------------------------------------
ORBextractor::ORBextractor(HipContext& ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    : nlevels_(nlevels), scaleFactor_(scaleFactor) {
  check(ccm_orb_create(ctx.get(), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, &orb_), ctx.get(), "ccm_orb_create");
}
ORBextractor::~ORBextractor() { ccm_orb_destroy(orb_); }

std::vector<float> ORBextractor::table(int which) const {
  std::vector<float> t(nlevels_);
  ccm_orb_get_table(orb_, which, t.data(), nlevels_);
  return t;
}

void ORBextractor::operator()(const uint8_t* image, int cols, int rows, int step, std::vector<KeyPoint>& keypoints,
                              std::vector<uint8_t>& descriptors) {
  if (!image || cols <= 0 || rows <= 0) { keypoints.clear(); descriptors.clear(); return; }   // if(_image.empty()) return; (:1219)
  const int cap = ccm_orb_max_keypoints(orb_);
  static_assert(sizeof(KeyPoint) == sizeof(ccm_keypoint), "KeyPoint must match ccm_keypoint");
  keypoints.resize(cap);
  descriptors.resize((size_t)cap * 32);
  std::vector<uint8_t*> pyr;
  if (keepPyramid) {
    mvImagePyramid.resize(nlevels_);
    pyr.resize(nlevels_);
    for (int l = 0; l < nlevels_; l++) {
      Level& L = mvImagePyramid[l];
      ccm_orb_level_size(orb_, cols, rows, l, &L.cols, &L.rows);
      L.data.resize((size_t)L.cols * L.rows);
      pyr[l] = L.data.data();
    }
  }
  int n = 0;
  const int rc = ccm_orb_extract(orb_, image, cols, rows, step, reinterpret_cast<ccm_keypoint*>(keypoints.data()), descriptors.data(),
                                 cap, &n, keepPyramid ? pyr.data() : nullptr);
  if (rc != CCM_OK) throw infrastructure_ex(std::string("ccm_orb_extract: ") + ccm_last_error(nullptr));
  keypoints.resize(n);
  descriptors.resize((size_t)n * 32);
}

// ---- Frame grid (Frame.cpp:87-88, 103-118, 200-265), FRAME_GRID_COLS 75 x ROWS 48 -------------------
namespace {
constexpr int kGridCols = 75, kGridRows = 48;

struct FrameGrid {
  const FrameView& F;
  float wInv, hInv;
  std::vector<int> start;   // CSR over cells (column-major: cell = ix * rows + iy), insertion order kept
  std::vector<int> items;
  explicit FrameGrid(const FrameView& f) : F(f) {
    wInv = static_cast<float>(kGridCols) / static_cast<float>(F.mnMaxX - F.mnMinX);
    hInv = static_cast<float>(kGridRows) / static_cast<float>(F.mnMaxY - F.mnMinY);
    std::vector<int> cellOf(F.N, -1);
    start.assign(kGridCols * kGridRows + 1, 0);
    for (int i = 0; i < F.N; i++) {
      const int px = (int)std::round((F.mvKeysUn[i].x - F.mnMinX) * wInv);   // PosInGrid uses round()
      const int py = (int)std::round((F.mvKeysUn[i].y - F.mnMinY) * hInv);
      if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
      cellOf[i] = px * kGridRows + py;
      start[cellOf[i] + 1]++;
    }
    for (size_t c = 1; c < start.size(); c++) start[c] += start[c - 1];
    items.resize(start.back());
    std::vector<int> pos(start.begin(), start.end() - 1);
    for (int i = 0; i < F.N; i++) if (cellOf[i] >= 0) items[pos[cellOf[i]]++] = i;
  }
  // appends the candidate indices in the reference's order (ix-major, iy, insertion)
  void featuresInArea(float x, float y, float r, int minLevel, int maxLevel, std::vector<int32_t>& out) const {
    const int nMinCellX = std::max(0, (int)std::floor((x - F.mnMinX - r) * wInv));
    if (nMinCellX >= kGridCols) return;
    const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((x - F.mnMinX + r) * wInv));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - F.mnMinY - r) * hInv));
    if (nMinCellY >= kGridRows) return;
    const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((y - F.mnMinY + r) * hInv));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const int c = ix * kGridRows + iy;
        for (int s = start[c]; s < start[c + 1]; s++) {
          const int k = items[s];
          const KeyPoint& kp = F.mvKeysUn[k];
          if (bCheckLevels) {
            if (kp.octave < minLevel) continue;
            if (maxLevel >= 0 && kp.octave > maxLevel) continue;
          }
          if (std::fabs(kp.x - x) < r && std::fabs(kp.y - y) < r) out.push_back(k);
        }
      }
  }
};

void threeMaxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {   // ORBmatcher.cpp:1607-1648
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) ind3 = -1;
}
}  // namespace

int ORBmatcher::DescriptorDistance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4); std::memcpy(&y, b + 4 * i, 4);
    dist += __builtin_popcount(x ^ y);
  }
  return dist;
}

// Device does every Hamming distance of every (map point, window candidate) pair in one launch; the host then
// replays the reference's loop over the map points IN ORDER on the precomputed distances, so that features
// claimed by an earlier map point are skipped by later ones exactly as ORBmatcher.cpp:113-115 does.
int ORBmatcher::SearchByProjection(FrameView& F, const TrackedMapPoints& mps, float th) {
  FrameGrid grid(F);
  const bool bFactor = th != 1.0;
  std::vector<int32_t> q_of;        // compact query -> map point index
  std::vector<int32_t> off(1, 0), idx;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < mps.n; i++) {
    if (!mps.mbTrackInView[i]) continue;
    const int lvl = mps.mnTrackScaleLevel[i];
    float r = (mps.mTrackViewCos[i] > 0.998) ? 2.5f : 4.0f;   // RadiusByViewingCos (:150-156)
    if (bFactor) r *= th;
    const size_t before = idx.size();
    grid.featuresInArea(mps.mTrackProjX[i], mps.mTrackProjY[i], r * F.mvScaleFactors[lvl], lvl - 1, lvl, idx);
    if (idx.size() == before) continue;   // vIndices.empty()
    q_of.push_back(i);
    off.push_back((int32_t)idx.size());
    qdesc.insert(qdesc.end(), mps.mDescriptor + (size_t)i * 32, mps.mDescriptor + (size_t)i * 32 + 32);
  }
  const int Q = (int)q_of.size();
  if (Q == 0) return 0;
  std::vector<uint16_t> dist(idx.size());
  check(ccm_hamming_csr(ctx_.get(), qdesc.data(), Q, F.mDescriptors, F.N, off.data(), idx.data(), dist.data(), nullptr, nullptr, nullptr),
        ctx_.get(), "ccm_hamming_csr");
  int nmatches = 0;
  for (int q = 0; q < Q; q++) {
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int k = idx[s];
      if (F.mvpMapPoints[k] >= 0) continue;   // F.mvpMapPoints[idx] && Observations() > 0
      const int d = dist[s];
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = F.mvKeysUn[k].octave; bestIdx = k; }
      else if (d < bestDist2) { bestLevel2 = F.mvKeysUn[k].octave; bestDist2 = d; }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      F.mvpMapPoints[bestIdx] = q_of[q];
      nmatches++;
    }
  }
  return nmatches;
}

FrameGridDev::FrameGridDev(HipContext& ctx, const float K[4], const float* distCoef, int nDist, int width, int height) : ctx_(ctx) {
  check(ccm_frame_create(ctx.get(), K, distCoef, nDist, width, height, &f_), ctx.get(), "ccm_frame_create");
  float b[4];
  ccm_frame_bounds(f_, b);
  mnMinX = b[0]; mnMinY = b[1]; mnMaxX = b[2]; mnMaxY = b[3];
}
FrameGridDev::~FrameGridDev() { ccm_frame_destroy(f_); }

void FrameGridDev::SetKeyPoints(const std::vector<KeyPoint>& mvKeys, const uint8_t* mDescriptors, std::vector<KeyPoint>& mvKeysUn) {
  static_assert(sizeof(KeyPoint) == sizeof(ccm_keypoint), "KeyPoint must mirror ccm_keypoint");
  const int n = (int)mvKeys.size();
  check(ccm_frame_set_keypoints(f_, reinterpret_cast<const ccm_keypoint*>(mvKeys.data()), mDescriptors, n), ctx_.get(), "ccm_frame_set_keypoints");
  std::vector<float> xy(2 * (size_t)std::max(n, 1));
  check(ccm_frame_get(f_, xy.data(), nullptr, nullptr), ctx_.get(), "ccm_frame_get");
  mvKeysUn = mvKeys;                                                  // kp = mvKeys[i]; kp.pt = undistorted (Frame.cpp:304-311)
  for (int i = 0; i < n; i++) { mvKeysUn[i].x = xy[2 * i]; mvKeysUn[i].y = xy[2 * i + 1]; }
}

// every candidate list (GetFeaturesInArea order) and Hamming distance of a batch of window queries in one device call
void ORBmatcher::deviceWindows(FrameGridDev& grid, const std::vector<float>& u, const std::vector<float>& v, const std::vector<float>& r,
                               const std::vector<int32_t>& minl, const std::vector<int32_t>& maxl, const std::vector<uint8_t>& qdesc,
                               std::vector<int32_t>& off, std::vector<int32_t>& idx, std::vector<uint16_t>& dist) {
  const int Q = (int)u.size();
  off.assign(Q + 1, 0);
  int64_t n = 0;
  idx.resize((size_t)Q * 16 + 1024); dist.resize(idx.size());   // typical lists hold a handful of features; grow once if short
  int rc = ccm_frame_window_search(grid.get(), Q, u.data(), v.data(), r.data(), minl.data(), maxl.data(), qdesc.data(), off.data(), idx.data(),
                                   dist.data(), (int64_t)idx.size(), &n);
  if (rc == CCM_E_ARG && n > (int64_t)idx.size()) {
    idx.resize((size_t)n); dist.resize((size_t)n);
    rc = ccm_frame_window_search(grid.get(), Q, u.data(), v.data(), r.data(), minl.data(), maxl.data(), qdesc.data(), off.data(), idx.data(),
                                 dist.data(), n, &n);
  }
  check(rc, ctx_.get(), "ccm_frame_window_search");
  idx.resize((size_t)n); dist.resize((size_t)n);
}

// SearchByProjection(Frame&, vector<mpptr>&, th) with the device grid: ONE ccm_frame_window_search call produces every
// candidate list (GetFeaturesInArea order) and distance; the ordered claim replay below is unchanged.
int ORBmatcher::SearchByProjection(FrameGridDev& grid, FrameView& F, const TrackedMapPoints& mps, float th) {
  const bool bFactor = th != 1.0;
  std::vector<int32_t> q_of, minl, maxl;
  std::vector<float> u, v, r;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < mps.n; i++) {
    if (!mps.mbTrackInView[i]) continue;
    const int lvl = mps.mnTrackScaleLevel[i];
    float rad = (mps.mTrackViewCos[i] > 0.998) ? 2.5f : 4.0f;   // RadiusByViewingCos (:150-156)
    if (bFactor) rad *= th;
    q_of.push_back(i);
    u.push_back(mps.mTrackProjX[i]); v.push_back(mps.mTrackProjY[i]); r.push_back(rad * F.mvScaleFactors[lvl]);
    minl.push_back(lvl - 1); maxl.push_back(lvl);
    qdesc.insert(qdesc.end(), mps.mDescriptor + (size_t)i * 32, mps.mDescriptor + (size_t)i * 32 + 32);
  }
  const int Q = (int)q_of.size();
  if (Q == 0) return 0;
  std::vector<int32_t> off, idx;
  std::vector<uint16_t> dist;
  deviceWindows(grid, u, v, r, minl, maxl, qdesc, off, idx, dist);
  int nmatches = 0;
  for (int q = 0; q < Q; q++) {
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int k = idx[s];
      if (F.mvpMapPoints[k] >= 0) continue;   // F.mvpMapPoints[idx] && Observations() > 0
      const int d = dist[s];
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = F.mvKeysUn[k].octave; bestIdx = k; }
      else if (d < bestDist2) { bestLevel2 = F.mvKeysUn[k].octave; bestDist2 = d; }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      F.mvpMapPoints[bestIdx] = q_of[q];
      nmatches++;
    }
  }
  return nmatches;
}

int ORBmatcher::SearchByProjection(FrameView& C, const LastFrameProjections& last, float th) {
  FrameGrid grid(C);
  std::vector<int32_t> q_of, off(1, 0), idx;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < last.n; i++) {
    if (!last.valid[i]) continue;
    const int oct = last.octave[i];
    const size_t before = idx.size();
    grid.featuresInArea(last.u[i], last.v[i], th * C.mvScaleFactors[oct], oct - 1, oct + 1, idx);
    if (idx.size() == before) continue;
    q_of.push_back(i);
    off.push_back((int32_t)idx.size());
    qdesc.insert(qdesc.end(), last.mpDescriptor + (size_t)i * 32, last.mpDescriptor + (size_t)i * 32 + 32);
  }
  const int Q = (int)q_of.size();
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;   // upstream quirk: 30-degree bins (:1358)
  if (Q > 0) {
    std::vector<uint16_t> dist(idx.size());
    check(ccm_hamming_csr(ctx_.get(), qdesc.data(), Q, C.mDescriptors, C.N, off.data(), idx.data(), dist.data(), nullptr, nullptr, nullptr),
          ctx_.get(), "ccm_hamming_csr");
    for (int q = 0; q < Q; q++) {
      int bestDist = 256, bestIdx2 = -1;
      for (int s = off[q]; s < off[q + 1]; s++) {
        const int k = idx[s];
        if (C.mvpMapPoints[k] >= 0) continue;
        if (dist[s] < bestDist) { bestDist = dist[s]; bestIdx2 = k; }
      }
      if (bestDist <= TH_HIGH) {
        C.mvpMapPoints[bestIdx2] = q_of[q];
        nmatches++;
        if (mbCheckOrientation) {
          float rot = last.angle[q_of[q]] - C.mvKeysUn[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdx2);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int k : rotHist[i]) { C.mvpMapPoints[k] = -1; nmatches--; }
  }
  return nmatches;
}

// SearchByProjection(Frame&, const Frame& LastFrame, th) with the device grid (ORBmatcher.cpp:1350-1476)
int ORBmatcher::SearchByProjection(FrameGridDev& grid, FrameView& C, const LastFrameProjections& last, float th) {
  std::vector<int32_t> q_of, minl, maxl;
  std::vector<float> u, v, r;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < last.n; i++) {
    if (!last.valid[i]) continue;
    const int oct = last.octave[i];
    q_of.push_back(i);
    u.push_back(last.u[i]); v.push_back(last.v[i]); r.push_back(th * C.mvScaleFactors[oct]);
    minl.push_back(oct - 1); maxl.push_back(oct + 1);
    qdesc.insert(qdesc.end(), last.mpDescriptor + (size_t)i * 32, last.mpDescriptor + (size_t)i * 32 + 32);
  }
  const int Q = (int)q_of.size();
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;   // upstream quirk: 30-degree bins (:1358)
  if (Q > 0) {
    std::vector<int32_t> off, idx;
    std::vector<uint16_t> dist;
    deviceWindows(grid, u, v, r, minl, maxl, qdesc, off, idx, dist);
    for (int q = 0; q < Q; q++) {
      int bestDist = 256, bestIdx2 = -1;
      for (int s = off[q]; s < off[q + 1]; s++) {
        const int k = idx[s];
        if (C.mvpMapPoints[k] >= 0) continue;
        if (dist[s] < bestDist) { bestDist = dist[s]; bestIdx2 = k; }
      }
      if (bestDist <= TH_HIGH) {
        C.mvpMapPoints[bestIdx2] = q_of[q];
        nmatches++;
        if (mbCheckOrientation) {
          float rot = last.angle[q_of[q]] - C.mvKeysUn[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdx2);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int k : rotHist[i]) { C.mvpMapPoints[k] = -1; nmatches--; }
  }
  return nmatches;
}

// ---- BoW-bucketed / triangulation / initialisation searches ---------------------------------------------
namespace {
// common node ids of two ordered FeatureVectors, ascending (the merge-join with lower_bound jumps, ORBmatcher.cpp:199-283)
std::vector<std::pair<int, int>> commonNodes(const FeatureVectorView& a, const FeatureVectorView& b) {
  std::vector<std::pair<int, int>> r;
  int i = 0, j = 0;
  while (i < a.nn && j < b.nn) {
    if (a.node[i] == b.node[j]) { r.emplace_back(i, j); i++; j++; }
    else if (a.node[i] < b.node[j]) i = (int)(std::lower_bound(a.node, a.node + a.nn, b.node[j]) - a.node);
    else j = (int)(std::lower_bound(b.node, b.node + b.nn, a.node[i]) - b.node);
  }
  return r;
}
int histBin(float rot) {
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * (1.0f / ORBmatcher::HISTO_LENGTH));
  if (bin == ORBmatcher::HISTO_LENGTH) bin = 0;
  return bin;
}
// queries = features of `a` in bucket order that pass `keep`; candidate list of each = the whole bucket of `b`
template <typename Keep>
void bucketQueries(const KeysView& a, const KeysView& b, Keep keep, std::vector<int32_t>& q_of, std::vector<int32_t>& off,
                   std::vector<int32_t>& idx, std::vector<uint8_t>& qdesc) {
  off.assign(1, 0);
  for (auto pr : commonNodes(a.fv, b.fv))
    for (int s1 = a.fv.off[pr.first]; s1 < a.fv.off[pr.first + 1]; s1++) {
      const int i1 = a.fv.idx[s1];
      if (!keep(i1)) continue;
      q_of.push_back(i1);
      idx.insert(idx.end(), b.fv.idx + b.fv.off[pr.second], b.fv.idx + b.fv.off[pr.second + 1]);
      off.push_back((int32_t)idx.size());
      qdesc.insert(qdesc.end(), a.desc + (size_t)i1 * 32, a.desc + (size_t)i1 * 32 + 32);
    }
}
}  // namespace

void ORBmatcher::distances(const std::vector<uint8_t>& qdesc, int Q, const uint8_t* tdesc, int T, const std::vector<int32_t>& off,
                           const std::vector<int32_t>& idx, std::vector<uint16_t>& dist) {
  dist.assign(std::max<size_t>(idx.size(), 1), 0);
  if (Q == 0 || idx.empty()) return;
  check(ccm_hamming_csr(ctx_.get(), qdesc.data(), Q, tdesc, T, off.data(), idx.data(), dist.data(), nullptr, nullptr, nullptr), ctx_.get(),
        "ccm_hamming_csr");
}

int ORBmatcher::SearchByBoW(const KeysView& KF, const KeysView& F, std::vector<int32_t>& matchesF) {
  matchesF.assign(F.N, -1);
  std::vector<int32_t> q_of, off, idx; std::vector<uint8_t> qdesc; std::vector<uint16_t> dist;
  bucketQueries(KF, F, [&](int i) { return KF.hasMapPoint[i] != 0; }, q_of, off, idx, qdesc);
  distances(qdesc, (int)q_of.size(), F.desc, F.N, off, idx, dist);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  for (size_t q = 0; q < q_of.size(); q++) {
    int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int realIdxF = idx[s];
      if (matchesF[realIdxF] >= 0) continue;
      const int d = dist[s];
      if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdxF = realIdxF; }
      else if (d < bestDist2) bestDist2 = d;
    }
    if (bestDist1 <= TH_LOW && static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
      matchesF[bestIdxF] = q_of[q];
      if (mbCheckOrientation) rotHist[histBin(KF.keys[q_of[q]].angle - F.keys[bestIdxF].angle)].push_back(bestIdxF);
      nmatches++;
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int k : rotHist[i]) { matchesF[k] = -1; nmatches--; }
    }
  }
  return nmatches;
}

int ORBmatcher::SearchByBoW_KF(const KeysView& K1, const KeysView& K2, std::vector<int32_t>& matches12) {
  matches12.assign(K1.N, -1);
  std::vector<int32_t> q_of, off, idx; std::vector<uint8_t> qdesc; std::vector<uint16_t> dist;
  bucketQueries(K1, K2, [&](int i) { return K1.hasMapPoint[i] != 0; }, q_of, off, idx, qdesc);
  distances(qdesc, (int)q_of.size(), K2.desc, K2.N, off, idx, dist);
  std::vector<char> vbMatched2(K2.N, 0);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  for (size_t q = 0; q < q_of.size(); q++) {
    int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int idx2 = idx[s];
      if (vbMatched2[idx2] || !K2.hasMapPoint[idx2]) continue;
      const int d = dist[s];
      if (d < bestDist1) { bestDist2 = bestDist1; bestDist1 = d; bestIdx2 = idx2; }
      else if (d < bestDist2) bestDist2 = d;
    }
    if (bestDist1 < TH_LOW && static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {   // strict '<' here (:641)
      matches12[q_of[q]] = bestIdx2;
      vbMatched2[bestIdx2] = 1;
      if (mbCheckOrientation) rotHist[histBin(K1.keys[q_of[q]].angle - K2.keys[bestIdx2].angle)].push_back(q_of[q]);
      nmatches++;
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int k : rotHist[i]) { matches12[k] = -1; nmatches--; }
    }
  }
  return nmatches;
}

namespace {
// the sequential part of SearchForTriangulation (ORBmatcher.cpp:745-845) from the distances of every (query, candidate) slot: queries in bucket order, the
// candidate list of each = the whole vocabulary node of keyframe 2.  has1 / has2: the map-point flags at the time of the call (a query whose feature has a map
// point by now is skipped like the reference's `if(pMP1) continue`, a candidate with one like `if(vbMatched2[idx2] || pMP2) continue`).
int resolveTriangulation(const KeyPoint* keys1, int N1, const uint8_t* has1, const KeyPoint* keys2, const uint8_t* has2, const std::vector<int32_t>& q_of,
                         const std::vector<int32_t>& off, const std::vector<int32_t>& idx, const uint16_t* dist, const float F12[9], float ex, float ey,
                         const float* sigma2_2, const float* sf2, bool checkOrientation, std::vector<int32_t>& matches12) {
  const int TH_LOW = ORBmatcher::TH_LOW, HISTO_LENGTH = ORBmatcher::HISTO_LENGTH;
  matches12.assign(N1, -1);
  std::vector<int> rotHist[ORBmatcher::HISTO_LENGTH];
  int nmatches = 0;
  for (size_t q = 0; q < q_of.size(); q++) {
    if (has1 && has1[q_of[q]]) continue;
    const KeyPoint& kp1 = keys1[q_of[q]];
    int bestDist = TH_LOW, bestIdx2 = -1;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int idx2 = idx[s];
      if (has2[idx2]) continue;          // vbMatched2 is never set by the reference (:761,:789)
      const int d = dist[s];
      if (d > TH_LOW || d > bestDist) continue;
      const KeyPoint& kp2 = keys2[idx2];
      const float distex = ex - kp2.x, distey = ey - kp2.y;
      if (distex * distex + distey * distey < 100 * sf2[kp2.octave]) continue;
      // CheckDistEpipolarLine (:159-176)
      const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
      const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
      const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
      const float num = a * kp2.x + b * kp2.y + c;
      const float den = a * a + b * b;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (dsqr < 3.84 * sigma2_2[kp2.octave]) { bestIdx2 = idx2; bestDist = d; }
    }
    if (bestIdx2 >= 0) {
      matches12[q_of[q]] = bestIdx2;
      nmatches++;
      if (checkOrientation) rotHist[histBin(kp1.angle - keys2[bestIdx2].angle)].push_back(q_of[q]);
    }
  }
  if (checkOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int k : rotHist[i]) { matches12[k] = -1; nmatches--; }
    }
  }
  return nmatches;
}
}  // namespace

int ORBmatcher::SearchForTriangulation(const KeysView& K1, const KeysView& K2, const float F12[9], float ex, float ey, const float* sigma2_2,
                                       const float* sf2, std::vector<int32_t>& matches12) {
  std::vector<int32_t> q_of, off, idx; std::vector<uint8_t> qdesc; std::vector<uint16_t> dist;
  bucketQueries(K1, K2, [&](int i) { return K1.hasMapPoint[i] == 0; }, q_of, off, idx, qdesc);   // only features WITHOUT a map point (:745-749)
  distances(qdesc, (int)q_of.size(), K2.desc, K2.N, off, idx, dist);
  return resolveTriangulation(K1.keys, K1.N, nullptr, K2.keys, K2.hasMapPoint, q_of, off, idx, dist.data(), F12, ex, ey, sigma2_2, sf2, mbCheckOrientation, matches12);
}

TriangulationBatch::TriangulationBatch(ORBmatcher& m, const KeysView& K1, const std::vector<KeysView>& K2) : check_ori_(m.mbCheckOrientation) {
  keys1_.assign(K1.keys, K1.keys + K1.N);
  has1_.assign(K1.hasMapPoint, K1.hasMapPoint + K1.N);
  nb_.resize(K2.size());
  // the queries of all neighbours back to back; candidate indices stay local to the neighbour's own descriptor set
  std::vector<uint8_t> qdesc, tdesc;
  std::vector<int32_t> q_off(1, 0), t_off(1, 0), off_all(1, 0), idx_all;
  for (size_t j = 0; j < K2.size(); j++) {
    Nb& nb = nb_[j];
    nb.keys.assign(K2[j].keys, K2[j].keys + K2[j].N);
    nb.has.assign(K2[j].hasMapPoint, K2[j].hasMapPoint + K2[j].N);
    std::vector<uint8_t> qd;
    bucketQueries(K1, K2[j], [&](int i) { return K1.hasMapPoint[i] == 0; }, nb.q_of, nb.off, nb.idx, qd);
    qdesc.insert(qdesc.end(), qd.begin(), qd.end());
    tdesc.insert(tdesc.end(), K2[j].desc, K2[j].desc + (size_t)K2[j].N * 32);
    const int32_t base = off_all.back();
    for (size_t q = 1; q < nb.off.size(); q++) off_all.push_back(base + nb.off[q]);
    idx_all.insert(idx_all.end(), nb.idx.begin(), nb.idx.end());
    q_off.push_back((int32_t)(qdesc.size() / 32));
    t_off.push_back((int32_t)(tdesc.size() / 32));
  }
  n_cand_ = (int64_t)idx_all.size();
  std::vector<uint16_t> dist_all(std::max<size_t>(idx_all.size(), 1), 0);
  if (!idx_all.empty())
    check(ccm_hamming_csr_multi(m.ctx_.get(), (int)K2.size(), qdesc.data(), q_off.data(), tdesc.data(), t_off.data(), off_all.data(), idx_all.data(), dist_all.data(),
                                nullptr, nullptr, nullptr), m.ctx_.get(), "ccm_hamming_csr_multi");
  size_t at = 0;
  for (Nb& nb : nb_) { nb.dist.assign(dist_all.begin() + at, dist_all.begin() + at + nb.idx.size()); at += nb.idx.size(); }
}

int TriangulationBatch::resolve(int j, const uint8_t* has1_now, const uint8_t* has2_now, const float F12[9], float ex, float ey, const float* sigma2_2, const float* sf2,
                                std::vector<int32_t>& matches12) const {
  const Nb& nb = nb_.at((size_t)j);
  // (the queries were chosen with the flags of the build; a feature that has gained a map point since is skipped, one that had one then cannot lose it here:
  // LocalMapping only adds points to the new keyframe between these calls)
  return resolveTriangulation(keys1_.data(), (int)keys1_.size(), has1_now ? has1_now : nullptr, nb.keys.data(), has2_now ? has2_now : nb.has.data(), nb.q_of, nb.off, nb.idx,
                              nb.dist.data(), F12, ex, ey, sigma2_2, sf2, check_ori_, matches12);
}

int ORBmatcher::SearchForInitialization(const KeysView& F1, const FrameView& F2, std::vector<float>& prev, std::vector<int32_t>& vnMatches12, int windowSize) {
  vnMatches12.assign(F1.N, -1);
  FrameGrid grid(F2);
  std::vector<int32_t> q_of, off(1, 0), idx; std::vector<uint8_t> qdesc; std::vector<uint16_t> dist;
  for (int i1 = 0; i1 < F1.N; i1++) {
    const int level1 = F1.keys[i1].octave;
    if (level1 > 0) continue;
    const size_t before = idx.size();
    grid.featuresInArea(prev[2 * i1], prev[2 * i1 + 1], (float)windowSize, level1, level1, idx);
    if (idx.size() == before) continue;
    q_of.push_back(i1); off.push_back((int32_t)idx.size());
    qdesc.insert(qdesc.end(), F1.desc + (size_t)i1 * 32, F1.desc + (size_t)i1 * 32 + 32);
  }
  distances(qdesc, (int)q_of.size(), F2.mDescriptors, F2.N, off, idx, dist);
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance(F2.N, INT32_MAX), vnMatches21(F2.N, -1);
  int nmatches = 0;
  for (size_t q = 0; q < q_of.size(); q++) {
    const int i1 = q_of[q];
    int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int i2 = idx[s];
      const int d = dist[s];
      if (vMatchedDistance[i2] <= d) continue;
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
      else if (d < bestDist2) bestDist2 = d;
    }
    if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * mfNNratio) {
      if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }   // steal-back (:506-510)
      vnMatches12[i1] = bestIdx2;
      vnMatches21[bestIdx2] = i1;
      vMatchedDistance[bestIdx2] = bestDist;
      nmatches++;
      if (mbCheckOrientation) rotHist[histBin(F1.keys[i1].angle - F2.mvKeysUn[bestIdx2].angle)].push_back(i1);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i]) if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < F1.N; i1++)
    if (vnMatches12[i1] >= 0) { prev[2 * i1] = F2.mvKeysUn[vnMatches12[i1]].x; prev[2 * i1 + 1] = F2.mvKeysUn[vnMatches12[i1]].y; }
  return nmatches;
}

int ORBmatcher::ProjectedSearch(const FrameView& KF, const float* invLevelSigma2, const ProjectedPoints& P, float th, bool chi2Gate,
                                int distThreshold, int32_t* matched, bool claim, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
  bestIdx.assign(P.n, -1); bestDist.assign(P.n, INT32_MAX);
  std::unique_ptr<FrameGrid> grid;
  if (!P.candOff) grid.reset(new FrameGrid(KF));
  std::vector<int32_t> q_of, off(1, 0), idx; std::vector<uint8_t> qdesc; std::vector<uint16_t> dist;
  for (int i = 0; i < P.n; i++) {
    if (!P.valid[i]) continue;
    const int lvl = P.level[i];
    const size_t before = idx.size();
    std::vector<int32_t> win;
    if (P.candOff) win.assign(P.candIdx + P.candOff[i], P.candIdx + P.candOff[i + 1]);
    else grid->featuresInArea(P.u[i], P.v[i], th * KF.mvScaleFactors[lvl], -1, -1, win);   // KeyFrame::GetFeaturesInArea: no level filter
    for (int k : win) {
      const int kpLevel = KF.mvKeysUn[k].octave;
      if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
      if (chi2Gate) {
        const float ex = P.u[i] - KF.mvKeysUn[k].x, ey = P.v[i] - KF.mvKeysUn[k].y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
      }
      idx.push_back(k);
    }
    if (idx.size() == before) continue;
    q_of.push_back(i); off.push_back((int32_t)idx.size());
    qdesc.insert(qdesc.end(), P.desc + (size_t)i * 32, P.desc + (size_t)i * 32 + 32);
  }
  distances(qdesc, (int)q_of.size(), KF.mDescriptors, KF.N, off, idx, dist);
  int nacc = 0;
  for (size_t q = 0; q < q_of.size(); q++) {
    const int i = q_of[q];
    int bd = INT32_MAX, bi = -1;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int k = idx[s];
      if (matched && matched[k] >= 0) continue;     // claims made earlier in this loop are honoured
      if (dist[s] < bd) { bd = dist[s]; bi = k; }
    }
    if (bd <= distThreshold) {
      bestIdx[i] = bi; bestDist[i] = bd;
      if (matched && claim && !(P.noClaim && P.noClaim[i])) matched[bi] = i;
      nacc++;
    }
  }
  return nacc;
}

// ProjectedSearch with the keyframe's grid, candidate lists and distances produced on the device: the level window
// [lvl-1, lvl] is applied by the window kernel, the chi2 gate (Fuse, :930-938) during the ordered resolution
int ORBmatcher::ProjectedSearch(FrameGridDev& grid, const FrameView& KF, const float* invLevelSigma2, const ProjectedPoints& P, float th, bool chi2Gate,
                                int distThreshold, int32_t* matched, bool claim, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
  bestIdx.assign(P.n, -1); bestDist.assign(P.n, INT32_MAX);
  std::vector<int32_t> q_of, minl, maxl;
  std::vector<float> u, v, r;
  std::vector<uint8_t> qdesc;
  for (int i = 0; i < P.n; i++) {
    if (!P.valid[i]) continue;
    const int lvl = P.level[i];
    q_of.push_back(i);
    u.push_back(P.u[i]); v.push_back(P.v[i]); r.push_back(th * KF.mvScaleFactors[lvl]);
    minl.push_back(lvl - 1); maxl.push_back(lvl);
    qdesc.insert(qdesc.end(), P.desc + (size_t)i * 32, P.desc + (size_t)i * 32 + 32);
  }
  if (q_of.empty()) return 0;
  std::vector<int32_t> off, idx;
  std::vector<uint16_t> dist;
  deviceWindows(grid, u, v, r, minl, maxl, qdesc, off, idx, dist);
  int nacc = 0;
  for (size_t q = 0; q < q_of.size(); q++) {
    const int i = q_of[q];
    int bd = INT32_MAX, bi = -1;
    for (int s = off[q]; s < off[q + 1]; s++) {
      const int k = idx[s];
      if (chi2Gate) {
        const float ex = P.u[i] - KF.mvKeysUn[k].x, ey = P.v[i] - KF.mvKeysUn[k].y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * invLevelSigma2[KF.mvKeysUn[k].octave] > 5.99) continue;
      }
      if (matched && matched[k] >= 0) continue;     // claims made earlier in this loop are honoured
      if (dist[s] < bd) { bd = dist[s]; bi = k; }
    }
    if (bd <= distThreshold) {
      bestIdx[i] = bi; bestDist[i] = bd;
      if (matched && claim && !(P.noClaim && P.noClaim[i])) matched[bi] = i;
      nacc++;
    }
  }
  return nacc;
}

// ---- FuseBatch: the projected window searches of many target keyframes, Hamming work in one launch ----
FuseBatch::FuseBatch(ORBmatcher& m, const std::vector<Target>& targets, bool chi2Gate, int distThreshold) : tg_(targets.size()), dist_threshold_(distThreshold) {
  const int S = (int)targets.size();
  std::vector<int32_t> q_off(S + 1, 0), t_off(S + 1, 0), cand_off(1, 0), cand_idx;
  std::vector<uint8_t> qdesc, tdesc;
  for (int s = 0; s < S; s++) {
    const Target& T = targets[s];
    const FrameView& KF = T.KF; const ORBmatcher::ProjectedPoints& P = T.P;
    Tg& g = tg_[s];
    g.n_pts = P.n; g.off.assign(1, 0);
    std::unique_ptr<FrameGrid> grid;
    if (!P.candOff) grid.reset(new FrameGrid(KF));
    std::vector<int32_t> win;
    for (int i = 0; i < P.n; i++) {          // candidate lists exactly as ProjectedSearch builds them (window, level window [lvl - 1, lvl], chi2 gate)
      if (!P.valid[i]) continue;
      const int lvl = P.level[i];
      const size_t before = g.idx.size();
      win.clear();
      if (P.candOff) win.assign(P.candIdx + P.candOff[i], P.candIdx + P.candOff[i + 1]);
      else grid->featuresInArea(P.u[i], P.v[i], T.th * KF.mvScaleFactors[lvl], -1, -1, win);
      for (int k : win) {
        const int kpLevel = KF.mvKeysUn[k].octave;
        if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
        if (chi2Gate) {
          const float ex = P.u[i] - KF.mvKeysUn[k].x, ey = P.v[i] - KF.mvKeysUn[k].y;
          const float e2 = ex * ex + ey * ey;
          if (e2 * T.invLevelSigma2[kpLevel] > 5.99) continue;
        }
        g.idx.push_back(k);
      }
      if (g.idx.size() == before) continue;
      g.q_of.push_back(i); g.off.push_back((int32_t)g.idx.size());
      qdesc.insert(qdesc.end(), P.desc + (size_t)i * 32, P.desc + (size_t)i * 32 + 32);
    }
    q_off[s + 1] = q_off[s] + (int32_t)g.q_of.size();
    t_off[s + 1] = t_off[s] + KF.N;
    tdesc.insert(tdesc.end(), KF.mDescriptors, KF.mDescriptors + (size_t)KF.N * 32);
    const int32_t base = cand_off.back();
    for (size_t q = 1; q < g.off.size(); q++) cand_off.push_back(base + g.off[q]);
    cand_idx.insert(cand_idx.end(), g.idx.begin(), g.idx.end());
  }
  n_cand_ = (int64_t)cand_idx.size();
  if (!n_cand_) return;
  std::vector<uint16_t> dist(cand_idx.size());
  check(ccm_hamming_csr_multi(m.ctx_.get(), S, qdesc.data(), q_off.data(), tdesc.data(), t_off.data(), cand_off.data(), cand_idx.data(), dist.data(),
                              nullptr, nullptr, nullptr), m.ctx_.get(), "ccm_hamming_csr_multi");
  size_t at = 0;
  for (int s = 0; s < S; s++) { Tg& g = tg_[s]; g.dist.assign(dist.begin() + at, dist.begin() + at + g.idx.size()); at += g.idx.size(); }
}

int FuseBatch::resolve(int s, const uint8_t* skip_now, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) const {
  const Tg& g = tg_.at(s);
  bestIdx.assign(g.n_pts, -1); bestDist.assign(g.n_pts, INT32_MAX);
  int nacc = 0;
  for (size_t q = 0; q < g.q_of.size(); q++) {
    const int i = g.q_of[q];
    if (skip_now && skip_now[i]) continue;
    int bd = INT32_MAX, bi = -1;
    for (int c = g.off[q]; c < g.off[q + 1]; c++) if (g.dist[c] < bd) { bd = g.dist[c]; bi = g.idx[c]; }   // first minimum wins (:941-945)
    if (bd <= dist_threshold_) { bestIdx[i] = bi; bestDist[i] = bd; nacc++; }
  }
  return nacc;
}

int ORBmatcher::MutualAgreement(const std::vector<int32_t>& vnMatch1, const std::vector<int32_t>& vnMatch2, std::vector<int32_t>& matches12) {
  int nFound = 0;
  matches12.assign(vnMatch1.size(), -1);
  for (size_t i1 = 0; i1 < vnMatch1.size(); i1++) {
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == (int)i1) { matches12[i1] = idx2; nFound++; }
  }
  return nFound;
}

// ---- vocabulary transform / distinctive descriptors ------------------------------------------------------------
ORBVocabulary::ORBVocabulary(HipContext& ctx, int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc,
                             const int32_t* word_id, const double* weight) {
  check(ccm_vocab_create(ctx.get(), n_nodes, L, child_off, child_id, node_desc, word_id, weight, &voc_), ctx.get(), "ccm_vocab_create");
}
ORBVocabulary::~ORBVocabulary() { ccm_vocab_destroy(voc_); }

void ORBVocabulary::transform(const uint8_t* descriptors, int N, BowVector& v, FeatureVector& fv, int levelsup) const {
  v.word.clear(); v.value.clear(); fv.node.clear(); fv.off.assign(1, 0); fv.idx.clear();
  if (N <= 0) return;
  std::vector<int32_t> word(N), node(N); std::vector<double> w(N);
  if (ccm_bow_transform(voc_, descriptors, N, levelsup, word.data(), w.data(), node.data()) != CCM_OK)
    throw infrastructure_ex(std::string("ccm_bow_transform: ") + ccm_last_error(nullptr));
  // TF_IDF weighting + L1 norm (ORBvoc): BowVector::addWeight in feature order, FeatureVector::addFeature, then normalize
  std::vector<std::pair<int32_t, int>> bw, fn;   // (word, feature), (node, feature) for kept features
  for (int i = 0; i < N; i++) if (w[i] > 0) { bw.emplace_back(word[i], i); fn.emplace_back(node[i], i); }   // "not stopped" (TemplatedVocabulary.h:1158)
  std::stable_sort(bw.begin(), bw.end(), [](const std::pair<int32_t, int>& a, const std::pair<int32_t, int>& b) { return a.first < b.first; });
  for (size_t s = 0; s < bw.size();) {
    const int32_t id = bw[s].first;
    double acc = 0.0;
    bool first = true;
    for (; s < bw.size() && bw[s].first == id; s++) { if (first) { acc = w[bw[s].second]; first = false; } else acc += w[bw[s].second]; }
    v.word.push_back(id); v.value.push_back(acc);
  }
  double norm = 0.0;
  for (double x : v.value) norm += std::fabs(x);
  if (norm > 0.0) for (double& x : v.value) x /= norm;
  std::stable_sort(fn.begin(), fn.end(), [](const std::pair<int32_t, int>& a, const std::pair<int32_t, int>& b) { return a.first < b.first; });
  for (size_t s = 0; s < fn.size();) {
    const int32_t id = fn[s].first;
    fv.node.push_back(id);
    for (; s < fn.size() && fn[s].first == id; s++) fv.idx.push_back(fn[s].second);
    fv.off.push_back((int32_t)fv.idx.size());
  }
}

std::vector<int32_t> ComputeDistinctiveDescriptors(HipContext& ctx, const uint8_t* desc, const std::vector<int32_t>& off) {
  std::vector<int32_t> best(off.empty() ? 0 : off.size() - 1);
  if (!best.empty()) check(ccm_distinctive_descriptors(ctx.get(), desc, off.data(), (int)best.size(), best.data()), ctx.get(), "ccm_distinctive_descriptors");
  return best;
}

// ---- Optimizer ---------------------------------------------------------------------------------------
int Optimizer::PoseOptimizationClient(HipContext& ctx, double cam_qt[7], int n, const double* Xw, const double* obs,
                                      const double* invSigma2, const double K[4], std::vector<uint8_t>& outlier) {
  outlier.assign(std::max(n, 1), 0);
  int ninl = 0;
  check(ccm_pose_optimize(ctx.get(), cam_qt, n, Xw, obs, invSigma2, K, outlier.data(), &ninl), ctx.get(), "ccm_pose_optimize");
  outlier.resize(n);
  return ninl;
}

int Optimizer::OptimizeSim3(HipContext& ctx, double g2oS12[8], int n, const double* P1c, const double* P2c, const double* obs1,
                            const double* obs2, const double* invSigma2_1, const double* invSigma2_2, const double K1[4],
                            const double K2[4], float th2, bool bFixScale, std::vector<uint8_t>& keep) {
  keep.assign(std::max(n, 1), 0);
  int nin = 0;
  check(ccm_sim3_optimize(ctx.get(), g2oS12, n, P1c, P2c, obs1, obs2, invSigma2_1, invSigma2_2, K1, K2, (double)th2,
                          bFixScale ? 1 : 0, keep.data(), &nin), ctx.get(), "ccm_sim3_optimize");
  keep.resize(n);
  return nin;
}

void Optimizer::OptimizeEssentialGraph(HipContext& ctx, std::vector<double>& vScw, const std::vector<uint8_t>& fixed, bool bFixScale,
                                       const std::vector<int32_t>& e_i, const std::vector<int32_t>& e_j, const std::vector<double>& Sji,
                                       ccm_pg_stats* stats) {
  const int n_vert = (int)fixed.size(), n_edge = (int)e_i.size();
  if ((int)vScw.size() != 8 * n_vert || (int)e_j.size() != n_edge || (int)Sji.size() != 8 * n_edge) throw infrastructure_ex("OptimizeEssentialGraph: inconsistent sizes");
  check(ccm_pose_graph_optimize(ctx.get(), n_vert, vScw.data(), fixed.data(), bFixScale ? 1 : 0, n_edge, e_i.data(), e_j.data(), Sji.data(), 20, 1e-16,
                                nullptr, stats), ctx.get(), "ccm_pose_graph_optimize");
}

static ccm_ba_problem make_problem(BAProblem& p, const uint8_t* level, double huber) {
  ccm_ba_problem c{};
  c.n_cam = p.n_cam(); c.n_pt = p.n_pt(); c.n_edge = p.n_edge();
  c.cam_qt = p.cam_qt.data(); c.cam_fixed = p.cam_fixed.data(); c.cam_K = p.cam_K.data(); c.pt_xyz = p.pt_xyz.data();
  c.e_cam = p.e_cam.data(); c.e_pt = p.e_pt.data(); c.e_obs = p.e_obs.data(); c.e_info = p.e_info.data();
  c.e_level = level; c.huber_delta = huber;
  return c;
}

void Optimizer::LocalBundleAdjustmentClient(HipContext& ctx, BAProblem& p, bool* pbStopFlag, std::vector<uint8_t>& to_erase) {
  const int ne = p.n_edge();
  to_erase.assign(ne, 0);
  if (pbStopFlag && *pbStopFlag) return;   // :532-534
  const double thHuberMono = (double)(float)std::sqrt(5.991);   // const float thHuberMono = sqrt(5.991) (:468)
  std::vector<uint8_t> level(ne, 0), dpos(ne, 1);
  std::vector<double> chi2(ne, 0.0);
  ccm_ba_options opt{}; opt.max_iters = 5;
  ccm_ba_problem c = make_problem(p, level.data(), thHuberMono);
  const volatile unsigned char* stop = reinterpret_cast<const volatile unsigned char*>(pbStopFlag);
  check(ccm_ba_optimize(ctx.get(), &c, &opt, stop, chi2.data(), dpos.data(), nullptr), ctx.get(), "ccm_ba_optimize");
  bool bDoMore = !(pbStopFlag && *pbStopFlag);
  if (bDoMore) {
    for (int e = 0; e < ne; e++) if (chi2[e] > 5.991 || !dpos[e]) level[e] = 1;   // setLevel(1); kernel dropped for all (:548-560)
    opt.max_iters = 10;
    c = make_problem(p, level.data(), 0.0);
    check(ccm_ba_optimize(ctx.get(), &c, &opt, stop, chi2.data(), dpos.data(), nullptr), ctx.get(), "ccm_ba_optimize");
  }
  for (int e = 0; e < ne; e++) to_erase[e] = (chi2[e] > 5.991 || !dpos[e]) ? 1 : 0;   // :574-586
}

void Optimizer::GlobalBundleAdjustment(HipContext& ctx, BAProblem& p, int nIterations, bool* pbStopFlag, bool bRobust, ccm_ba_stats* stats) {
  const double thHuber2D = (double)(float)std::sqrt(5.99);   // :759
  ccm_ba_options opt{}; opt.max_iters = nIterations;
  ccm_ba_problem c = make_problem(p, nullptr, bRobust ? thHuber2D : 0.0);
  check(ccm_ba_optimize(ctx.get(), &c, &opt, reinterpret_cast<const volatile unsigned char*>(pbStopFlag), nullptr, nullptr, stats),
        ctx.get(), "ccm_ba_optimize");
}

}  // namespace cslam

// ---- C entry points (ccm_host_c.h): the Python test-suite and the drop-in translation units under shim/ ---------------------------------------
#include "ccm_host_c.h"
namespace {
// one context per (calling thread, device), created on first use and kept for the thread's lifetime: the reference's threads (tracking, mapping, loop
// finder ...) each call the matcher / optimizer from their own loop, so the per-call cost is a map lookup, not a stream + buffer set-up
cslam::HipContext& thread_context(int device) {
  thread_local std::map<int, std::unique_ptr<cslam::HipContext>> ctxs;
  auto& c = ctxs[device];
  if (!c) c.reset(new cslam::HipContext(device));
  return *c;
}
}  // namespace
extern "C" {
int ccmh_search_by_projection_mp(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* fdesc, int N,
                                 float minX, float minY, float maxX, float maxY, const float* scale_factors, int n_mp,
                                 const uint8_t* in_view, const float* px, const float* py, const int32_t* lvl, const float* vcos,
                                 const uint8_t* mp_desc, float th, float nnratio, int32_t* frame_mp) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    std::vector<cslam::KeyPoint> kps(N);
    for (int i = 0; i < N; i++) kps[i] = cslam::KeyPoint{kx[i], ky[i], 31.f, 0.f, 0.f, oct[i]};
    cslam::FrameView F; F.N = N; F.mvKeysUn = kps.data(); F.mDescriptors = fdesc; F.mnMinX = minX; F.mnMinY = minY; F.mnMaxX = maxX; F.mnMaxY = maxY;
    F.mvScaleFactors = scale_factors; F.mvpMapPoints = frame_mp;
    cslam::TrackedMapPoints M; M.n = n_mp; M.mbTrackInView = in_view; M.mTrackProjX = px; M.mTrackProjY = py; M.mnTrackScaleLevel = lvl;
    M.mTrackViewCos = vcos; M.mDescriptor = mp_desc;
    cslam::ORBmatcher m(ctx, nnratio, true);
    return m.SearchByProjection(F, M, th);
  } catch (const std::exception&) { return -1000; }
}

int ccmh_search_by_projection_last(int device, const float* kx, const float* ky, const int32_t* oct, const float* kangle,
                                   const uint8_t* fdesc, int N, float minX, float minY, float maxX, float maxY,
                                   const float* scale_factors, int n_last, const uint8_t* valid, const float* u, const float* v,
                                   const int32_t* l_oct, const float* l_angle, const uint8_t* l_desc, float th, int check_ori,
                                   int32_t* cur_mp) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    std::vector<cslam::KeyPoint> kps(N);
    for (int i = 0; i < N; i++) kps[i] = cslam::KeyPoint{kx[i], ky[i], 31.f, kangle[i], 0.f, oct[i]};
    cslam::FrameView F; F.N = N; F.mvKeysUn = kps.data(); F.mDescriptors = fdesc; F.mnMinX = minX; F.mnMinY = minY; F.mnMaxX = maxX; F.mnMaxY = maxY;
    F.mvScaleFactors = scale_factors; F.mvpMapPoints = cur_mp;
    cslam::LastFrameProjections L; L.n = n_last; L.valid = valid; L.u = u; L.v = v; L.octave = l_oct; L.angle = l_angle; L.mpDescriptor = l_desc;
    cslam::ORBmatcher m(ctx, 0.9f, check_ori != 0);
    return m.SearchByProjection(F, L, th);
  } catch (const std::exception&) { return -1000; }
}

int ccmh_local_ba(int device, int n_cam, int n_pt, int n_edge, double* cam_qt, const uint8_t* cam_fixed, const double* cam_K,
                  double* pt_xyz, const int32_t* e_cam, const int32_t* e_pt, const double* e_obs, const double* e_info, uint8_t* to_erase) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    cslam::BAProblem p;
    p.cam_qt.assign(cam_qt, cam_qt + 7 * (size_t)n_cam); p.cam_fixed.assign(cam_fixed, cam_fixed + n_cam);
    p.cam_K.assign(cam_K, cam_K + 4 * (size_t)n_cam); p.pt_xyz.assign(pt_xyz, pt_xyz + 3 * (size_t)n_pt);
    p.e_cam.assign(e_cam, e_cam + n_edge); p.e_pt.assign(e_pt, e_pt + n_edge);
    p.e_obs.assign(e_obs, e_obs + 2 * (size_t)n_edge); p.e_info.assign(e_info, e_info + n_edge);
    std::vector<uint8_t> er;
    cslam::Optimizer::LocalBundleAdjustmentClient(ctx, p, nullptr, er);
    std::memcpy(cam_qt, p.cam_qt.data(), sizeof(double) * p.cam_qt.size());
    std::memcpy(pt_xyz, p.pt_xyz.data(), sizeof(double) * p.pt_xyz.size());
    std::memcpy(to_erase, er.data(), er.size());
    return 0;
  } catch (const std::exception&) { return -1000; }
}

// SearchByProjection(Frame, map points) through the device grid: raw (distorted) keypoints in, match table + undistorted xy out
int ccmh_search_by_projection_mp_dev(int device, const float* K, const float* dist, int n_dist, int w, int h, const void* kps_raw, const uint8_t* fdesc,
                                     int N, const float* scale_factors, int n_mp, const uint8_t* in_view, const float* px, const float* py,
                                     const int32_t* lvl, const float* vcos, const uint8_t* mp_desc, float th, float nnratio, int32_t* frame_mp,
                                     float* xy_un_out) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    cslam::FrameGridDev grid(ctx, K, dist, n_dist, w, h);
    std::vector<cslam::KeyPoint> keys((const cslam::KeyPoint*)kps_raw, (const cslam::KeyPoint*)kps_raw + N), keysUn;
    grid.SetKeyPoints(keys, fdesc, keysUn);
    for (int i = 0; i < N; i++) { xy_un_out[2 * i] = keysUn[i].x; xy_un_out[2 * i + 1] = keysUn[i].y; }
    cslam::FrameView F;
    F.N = N; F.mvKeysUn = keysUn.data(); F.mDescriptors = fdesc;
    F.mnMinX = grid.mnMinX; F.mnMinY = grid.mnMinY; F.mnMaxX = grid.mnMaxX; F.mnMaxY = grid.mnMaxY;
    F.mvScaleFactors = scale_factors; F.mvpMapPoints = frame_mp;
    cslam::TrackedMapPoints mps;
    mps.n = n_mp; mps.mbTrackInView = in_view; mps.mTrackProjX = px; mps.mTrackProjY = py; mps.mnTrackScaleLevel = lvl; mps.mTrackViewCos = vcos;
    mps.mDescriptor = mp_desc;
    cslam::ORBmatcher m(ctx, nnratio, true);
    return m.SearchByProjection(grid, F, mps, th);
  } catch (const std::exception&) { return -1000; }
}

// SearchByProjection(Frame, LastFrame) through the device grid; keypoints given already undistorted (no distortion coefficients)
int ccmh_search_by_projection_last_dev(int device, const void* kps_un, const uint8_t* cdesc, int N, int w, int h, const float* scale_factors, int n_last,
                                       const uint8_t* valid, const float* u, const float* v, const int32_t* oct, const float* angle, const uint8_t* mp_desc,
                                       float th, int check_ori, int32_t* frame_mp) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    const float K[4] = {1.f, 1.f, 0.f, 0.f};
    cslam::FrameGridDev grid(ctx, K, nullptr, 0, w, h);
    std::vector<cslam::KeyPoint> keys((const cslam::KeyPoint*)kps_un, (const cslam::KeyPoint*)kps_un + N), keysUn;
    grid.SetKeyPoints(keys, cdesc, keysUn);
    cslam::FrameView C;
    C.N = N; C.mvKeysUn = keysUn.data(); C.mDescriptors = cdesc;
    C.mnMinX = grid.mnMinX; C.mnMinY = grid.mnMinY; C.mnMaxX = grid.mnMaxX; C.mnMaxY = grid.mnMaxY;
    C.mvScaleFactors = scale_factors; C.mvpMapPoints = frame_mp;
    cslam::LastFrameProjections L;
    L.n = n_last; L.valid = valid; L.u = u; L.v = v; L.octave = oct; L.angle = angle; L.mpDescriptor = mp_desc;
    cslam::ORBmatcher m(ctx, 0.9f, check_ori != 0);
    return m.SearchByProjection(grid, C, L, th);
  } catch (const std::exception&) { return -1000; }
}

int ccmh_optimize_sim3(int device, double* sim3, int n, const double* P1c, const double* P2c, const double* obs1, const double* obs2,
                       const double* info1, const double* info2, const double* K1, const double* K2, float th2, int fix_scale, uint8_t* keep) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    std::vector<uint8_t> k;
    const int nin = cslam::Optimizer::OptimizeSim3(ctx, sim3, n, P1c, P2c, obs1, obs2, info1, info2, K1, K2, th2, fix_scale != 0, k);
    if (n) std::memcpy(keep, k.data(), k.size());
    return nin;
  } catch (const std::exception&) { return -1000; }
}

int ccmh_orb_extract(int device, int nfeatures, const uint8_t* img, int w, int h, void* kps_out, uint8_t* desc_out, int cap) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    cslam::ORBextractor ex(ctx, nfeatures, 1.2f, 8, 20, 7);
    std::vector<cslam::KeyPoint> k; std::vector<uint8_t> d;
    ex(img, w, h, w, k, d);
    const int n = std::min<int>((int)k.size(), cap);
    std::memcpy(kps_out, k.data(), sizeof(cslam::KeyPoint) * n);
    std::memcpy(desc_out, d.data(), (size_t)n * 32);
    return n;
  } catch (const std::exception&) { return -1000; }
}
}

// ---- C wrappers for the BoW / triangulation / initialisation searches --------------------------------------
namespace {
std::vector<cslam::KeyPoint> mk_keys(const float* x, const float* y, const int32_t* oct, const float* ang, int N) {
  std::vector<cslam::KeyPoint> k(N);
  for (int i = 0; i < N; i++) k[i] = cslam::KeyPoint{x[i], y[i], 31.f, ang ? ang[i] : 0.f, 0.f, oct ? oct[i] : 0};
  return k;
}
}
extern "C" {
// mode 0: SearchByBoW(KF,F) -> out[N2] ; 1: SearchByBoW(KF,KF) -> out[N1] ; 2: SearchForTriangulation -> out[N1]
int ccmh_search_bow(int device, int mode, const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const int32_t* n2,
                    const int32_t* o2, const int32_t* i2, int nn2, const uint8_t* has1, const uint8_t* has2, const uint8_t* d1,
                    const float* x1, const float* y1, const float* a1, int N1, const uint8_t* d2, const float* x2, const float* y2,
                    const int32_t* oct2, const float* a2, int N2, const float* F12, float ex, float ey, const float* sigma2_2,
                    const float* sf2, float nnratio, int check_ori, int32_t* out) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    auto k1 = mk_keys(x1, y1, nullptr, a1, N1), k2 = mk_keys(x2, y2, oct2, a2, N2);
    cslam::KeysView A; A.N = N1; A.keys = k1.data(); A.desc = d1; A.hasMapPoint = has1; A.fv = cslam::FeatureVectorView{nn1, n1, o1, i1};
    cslam::KeysView B; B.N = N2; B.keys = k2.data(); B.desc = d2; B.hasMapPoint = has2; B.fv = cslam::FeatureVectorView{nn2, n2, o2, i2};
    cslam::ORBmatcher m(ctx, nnratio, check_ori != 0);
    std::vector<int32_t> r; int n = 0;
    if (mode == 0) n = m.SearchByBoW(A, B, r);
    else if (mode == 1) n = m.SearchByBoW_KF(A, B, r);
    else n = m.SearchForTriangulation(A, B, F12, ex, ey, sigma2_2, sf2, r);
    std::memcpy(out, r.data(), r.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}

// TriangulationBatch through C (shim/ORBmatcher_hip.cpp, tests): neighbour arrays as arrays of pointers, one entry per neighbour
void* ccmh_tri_batch_create(int device, float nnratio, int check_ori, const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const uint8_t* has1,
                            const uint8_t* d1, const float* x1, const float* y1, const float* a1, int N1, int n_nb, const int32_t* const* n2,
                            const int32_t* const* o2, const int32_t* const* i2, const int32_t* nn2, const uint8_t* const* has2, const uint8_t* const* d2,
                            const float* const* x2, const float* const* y2, const int32_t* const* oct2, const float* const* a2, const int32_t* N2) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    auto k1 = mk_keys(x1, y1, nullptr, a1, N1);
    cslam::KeysView A; A.N = N1; A.keys = k1.data(); A.desc = d1; A.hasMapPoint = has1; A.fv = cslam::FeatureVectorView{nn1, n1, o1, i1};
    std::vector<std::vector<cslam::KeyPoint>> k2((size_t)n_nb);
    std::vector<cslam::KeysView> B((size_t)n_nb);
    for (int j = 0; j < n_nb; j++) {
      k2[j] = mk_keys(x2[j], y2[j], oct2[j], a2[j], N2[j]);
      B[j].N = N2[j]; B[j].keys = k2[j].data(); B[j].desc = d2[j]; B[j].hasMapPoint = has2[j]; B[j].fv = cslam::FeatureVectorView{nn2[j], n2[j], o2[j], i2[j]};
    }
    cslam::ORBmatcher m(ctx, nnratio, check_ori != 0);
    return new cslam::TriangulationBatch(m, A, B);
  } catch (const std::exception&) { return nullptr; }
}
int ccmh_tri_batch_resolve(void* h, int j, const uint8_t* has1_now, const uint8_t* has2_now, const float* F12, float ex, float ey, const float* sigma2_2,
                           const float* sf2, int32_t* matches12) {
  try {
    std::vector<int32_t> r;
    const int n = static_cast<cslam::TriangulationBatch*>(h)->resolve(j, has1_now, has2_now, F12, ex, ey, sigma2_2, sf2, r);
    std::memcpy(matches12, r.data(), r.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}
long long ccmh_tri_batch_candidates(void* h) { return h ? (long long)static_cast<cslam::TriangulationBatch*>(h)->candidates() : 0; }
void ccmh_tri_batch_destroy(void* h) { delete static_cast<cslam::TriangulationBatch*>(h); }

int ccmh_search_for_initialization(int device, const float* x1, const float* y1, const int32_t* oct1, const float* a1, const uint8_t* d1, int N1,
                                   const float* x2, const float* y2, const int32_t* oct2, const float* a2, const uint8_t* d2, int N2,
                                   float minX, float minY, float maxX, float maxY, float* prev_xy, int window, float nnratio, int check_ori,
                                   int32_t* matches12) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    auto k1 = mk_keys(x1, y1, oct1, a1, N1), k2 = mk_keys(x2, y2, oct2, a2, N2);
    cslam::KeysView A; A.N = N1; A.keys = k1.data(); A.desc = d1;
    std::vector<int32_t> dummy(N2, -1);
    cslam::FrameView F; F.N = N2; F.mvKeysUn = k2.data(); F.mDescriptors = d2; F.mnMinX = minX; F.mnMinY = minY; F.mnMaxX = maxX; F.mnMaxY = maxY;
    F.mvpMapPoints = dummy.data();
    std::vector<float> prev(prev_xy, prev_xy + 2 * (size_t)N1);
    std::vector<int32_t> r;
    cslam::ORBmatcher m(ctx, nnratio, check_ori != 0);
    const int n = m.SearchForInitialization(A, F, prev, r, window);
    std::memcpy(prev_xy, prev.data(), prev.size() * sizeof(float));
    std::memcpy(matches12, r.data(), r.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}
}

extern "C" int ccmh_projected_window_search(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX,
                                            float minY, float maxX, float maxY, const float* scale_factors, const float* inv_sigma2, int n_pts,
                                            const uint8_t* valid, const float* u, const float* v, const int32_t* level, const uint8_t* pdesc, float th,
                                            int chi2_gate, int dist_threshold, int32_t* matched, int claim, const uint8_t* no_claim,
                                            int32_t* best_idx, int32_t* best_dist) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    auto kps = mk_keys(kx, ky, oct, nullptr, N);
    std::vector<int32_t> dummy(N, -1);
    cslam::FrameView F; F.N = N; F.mvKeysUn = kps.data(); F.mDescriptors = kdesc; F.mnMinX = minX; F.mnMinY = minY; F.mnMaxX = maxX; F.mnMaxY = maxY;
    F.mvScaleFactors = scale_factors; F.mvpMapPoints = dummy.data();
    cslam::ORBmatcher::ProjectedPoints P; P.n = n_pts; P.valid = valid; P.u = u; P.v = v; P.level = level; P.desc = pdesc; P.noClaim = no_claim;
    cslam::ORBmatcher m(ctx);
    std::vector<int32_t> bi, bd;
    const int n = m.ProjectedSearch(F, inv_sigma2, P, th, chi2_gate != 0, dist_threshold, matched, claim != 0, bi, bd);
    std::memcpy(best_idx, bi.data(), bi.size() * sizeof(int32_t));
    std::memcpy(best_dist, bd.data(), bd.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}

// the same search on candidate lists the caller obtained from its own KeyFrame::GetFeaturesInArea (CSR over the n points; th and the bounds are unused)
extern "C" int ccmh_projected_window_search_cand(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N,
                                                 const float* inv_sigma2, int n_pts, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                                 const uint8_t* pdesc, const int32_t* cand_off, const int32_t* cand_idx, int chi2_gate, int dist_threshold,
                                                 int32_t* matched, int claim, const uint8_t* no_claim, int32_t* best_idx, int32_t* best_dist) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    auto kps = mk_keys(kx, ky, oct, nullptr, N);
    cslam::FrameView F; F.N = N; F.mvKeysUn = kps.data(); F.mDescriptors = kdesc;
    cslam::ORBmatcher::ProjectedPoints P; P.n = n_pts; P.valid = valid; P.u = u; P.v = v; P.level = level; P.desc = pdesc; P.noClaim = no_claim;
    P.candOff = cand_off; P.candIdx = cand_idx;
    cslam::ORBmatcher m(ctx);
    std::vector<int32_t> bi, bd;
    const int n = m.ProjectedSearch(F, inv_sigma2, P, 0.f, chi2_gate != 0, dist_threshold, matched, claim != 0, bi, bd);
    std::memcpy(best_idx, bi.data(), bi.size() * sizeof(int32_t));
    std::memcpy(best_dist, bd.data(), bd.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}

// the same search through the device grid (bounds must be the plain image rectangle: 0, 0, maxX, maxY)
extern "C" int ccmh_projected_window_search_dev(int device, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float maxX,
                                                float maxY, const float* scale_factors, const float* inv_sigma2, int n_pts, const uint8_t* valid,
                                                const float* u, const float* v, const int32_t* level, const uint8_t* pdesc, float th, int chi2_gate,
                                                int dist_threshold, int32_t* matched, int claim, const uint8_t* no_claim, int32_t* best_idx,
                                                int32_t* best_dist) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    auto kps = mk_keys(kx, ky, oct, nullptr, N);
    const float K[4] = {1.f, 1.f, 0.f, 0.f};
    cslam::FrameGridDev grid(ctx, K, nullptr, 0, (int)maxX, (int)maxY);
    std::vector<cslam::KeyPoint> keysUn;
    grid.SetKeyPoints(kps, kdesc, keysUn);
    std::vector<int32_t> dummy(N, -1);
    cslam::FrameView F; F.N = N; F.mvKeysUn = keysUn.data(); F.mDescriptors = kdesc; F.mnMinX = grid.mnMinX; F.mnMinY = grid.mnMinY; F.mnMaxX = grid.mnMaxX;
    F.mnMaxY = grid.mnMaxY; F.mvScaleFactors = scale_factors; F.mvpMapPoints = dummy.data();
    cslam::ORBmatcher::ProjectedPoints P; P.n = n_pts; P.valid = valid; P.u = u; P.v = v; P.level = level; P.desc = pdesc; P.noClaim = no_claim;
    cslam::ORBmatcher m(ctx);
    std::vector<int32_t> bi, bd;
    const int n = m.ProjectedSearch(grid, F, inv_sigma2, P, th, chi2_gate != 0, dist_threshold, matched, claim != 0, bi, bd);
    std::memcpy(best_idx, bi.data(), bi.size() * sizeof(int32_t));
    std::memcpy(best_dist, bd.data(), bd.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}

// FuseBatch through a flat interface (tests): S targets given as concatenated arrays with offsets; every target is searched with the chi2 gate and TH_LOW like Fuse
extern "C" void* ccmh_fuse_batch_create(int device, int S, const int32_t* kf_off /* [S+1] features */, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc,
                                        const float* bounds4 /* [S][4] minX minY maxX maxY */, const float* scale_factors, const float* inv_sigma2,
                                        const int32_t* pt_off /* [S+1] points */, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                        const uint8_t* pdesc, float th) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    cslam::ORBmatcher m(ctx);
    std::vector<std::vector<cslam::KeyPoint>> keys(S);
    std::vector<std::vector<int32_t>> dummy(S);
    std::vector<cslam::FuseBatch::Target> tg(S);
    for (int s = 0; s < S; s++) {
      const int k0 = kf_off[s], N = kf_off[s + 1] - k0, p0 = pt_off[s];
      keys[s] = mk_keys(kx + k0, ky + k0, oct + k0, nullptr, N);
      dummy[s].assign(N, -1);
      cslam::FrameView& F = tg[s].KF;
      F.N = N; F.mvKeysUn = keys[s].data(); F.mDescriptors = kdesc + (size_t)k0 * 32; F.mnMinX = bounds4[4 * s]; F.mnMinY = bounds4[4 * s + 1]; F.mnMaxX = bounds4[4 * s + 2];
      F.mnMaxY = bounds4[4 * s + 3]; F.mvScaleFactors = scale_factors; F.mvpMapPoints = dummy[s].data();
      cslam::ORBmatcher::ProjectedPoints& P = tg[s].P;
      P.n = pt_off[s + 1] - p0; P.valid = valid + p0; P.u = u + p0; P.v = v + p0; P.level = level + p0; P.desc = pdesc + (size_t)p0 * 32;
      tg[s].invLevelSigma2 = inv_sigma2; tg[s].th = th;
    }
    return new cslam::FuseBatch(m, tg);
  } catch (const std::exception&) { return nullptr; }
}
// the same with every target's window candidates supplied by the caller (its own KeyFrame::GetFeaturesInArea, as ccmh_projected_window_search_cand): target s owns the
// points pt_off[s] .. pt_off[s+1]; cand_off has one entry per point plus one per target (target s: entries pt_off[s] + s .. pt_off[s+1] + s, starting at 0), its
// candidate indices start at cand_base[s] in cand_idx
extern "C" void* ccmh_fuse_batch_create_cand(int device, int S, const int32_t* kf_off, const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc,
                                             const float* const* inv_sigma2 /* [S] */, const int32_t* pt_off, const uint8_t* valid, const float* u, const float* v,
                                             const int32_t* level, const uint8_t* pdesc, const int32_t* cand_off, const int32_t* cand_base, const int32_t* cand_idx,
                                             int chi2_gate, int dist_threshold) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    cslam::ORBmatcher m(ctx);
    std::vector<std::vector<cslam::KeyPoint>> keys(S);
    std::vector<cslam::FuseBatch::Target> tg(S);
    for (int s = 0; s < S; s++) {
      const int k0 = kf_off[s], N = kf_off[s + 1] - k0, p0 = pt_off[s];
      keys[s] = mk_keys(kx + k0, ky + k0, oct + k0, nullptr, N);
      cslam::FrameView& F = tg[s].KF;
      F.N = N; F.mvKeysUn = keys[s].data(); F.mDescriptors = kdesc + (size_t)k0 * 32;
      cslam::ORBmatcher::ProjectedPoints& P = tg[s].P;
      P.n = pt_off[s + 1] - p0; P.valid = valid + p0; P.u = u + p0; P.v = v + p0; P.level = level + p0; P.desc = pdesc + (size_t)p0 * 32;
      P.candOff = cand_off + p0 + s; P.candIdx = cand_idx + cand_base[s];
      tg[s].invLevelSigma2 = inv_sigma2[s]; tg[s].th = 0.f;
    }
    return new cslam::FuseBatch(m, tg, chi2_gate != 0, dist_threshold);
  } catch (const std::exception&) { return nullptr; }
}
extern "C" long long ccmh_fuse_batch_candidates(void* h) { return h ? (long long)((cslam::FuseBatch*)h)->candidates() : -1; }
extern "C" int ccmh_fuse_batch_resolve(void* h, int s, const uint8_t* skip_now, int n_pts, int32_t* best_idx, int32_t* best_dist) {
  try {
    std::vector<int32_t> bi, bd;
    const int n = ((cslam::FuseBatch*)h)->resolve(s, skip_now, bi, bd);
    if ((int)bi.size() != n_pts) return -1001;
    std::memcpy(best_idx, bi.data(), bi.size() * sizeof(int32_t));
    std::memcpy(best_dist, bd.data(), bd.size() * sizeof(int32_t));
    return n;
  } catch (const std::exception&) { return -1000; }
}
extern "C" void ccmh_fuse_batch_destroy(void* h) { delete (cslam::FuseBatch*)h; }

extern "C" int ccmh_bow_transform(int device, int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc,
                                  const int32_t* word_id, const double* weight, const uint8_t* desc, int N, int levelsup, int32_t* bow_ids,
                                  double* bow_vals, int32_t* fv_nodes, int32_t* fv_off, int32_t* fv_idx, int32_t* sizes /* [n_bow, n_fv_nodes, n_fv_idx] */) {
  try {
    cslam::HipContext& ctx = thread_context(device);
    cslam::ORBVocabulary voc(ctx, n_nodes, L, child_off, child_id, node_desc, word_id, weight);
    cslam::BowVector v; cslam::FeatureVector fv;
    voc.transform(desc, N, v, fv, levelsup);
    std::memcpy(bow_ids, v.word.data(), v.word.size() * 4); std::memcpy(bow_vals, v.value.data(), v.value.size() * 8);
    std::memcpy(fv_nodes, fv.node.data(), fv.node.size() * 4); std::memcpy(fv_off, fv.off.data(), fv.off.size() * 4);
    std::memcpy(fv_idx, fv.idx.data(), fv.idx.size() * 4);
    sizes[0] = (int32_t)v.word.size(); sizes[1] = (int32_t)fv.node.size(); sizes[2] = (int32_t)fv.idx.size();
    return 0;
  } catch (const std::exception&) { return -1000; }
}

// C entry points of the f32 <-> f64 boundary (ccm_convert.h) for the tests and for non-C++ callers
#include "ccm_convert.h"
extern "C" {
void ccmh_to_se3quat(const float* Tcw16, double* qt7) { ccmh::toSE3Quat(Tcw16, qt7); }
void ccmh_se3quat_to_cvmat(const double* qt7, float* Tcw16) { ccmh::toCvMat(qt7, Tcw16); }
void ccmh_sim3_to_cvse3(const double* s8, float* Tcw16) { ccmh::sim3ToCvSE3(s8, Tcw16); }
}
