// match_ref.cpp — CPU ORACLE (test infrastructure, NOT product code) for the ORBmatcher path.
//
// Restates, on flat arrays, the reference code cited at each function.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into oracle/.
// PARITY PIN: the reference ships no tests or golden vectors for this path and
// cannot be compiled here (needs OpenCV/Boost/ROS).  This file is pinned by self-evident KATs
// (popcount vs bit counting) and by the hand-checked fixtures in tests/golden/; beyond that
// parity is "unpinned" against a running reference binary.
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" {

// ORBmatcher::DescriptorDistance — cslam/src/ORBmatcher.cpp:1653-1669 (SWAR popcount on 8 int32)
int ora_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  uint32_t pa[8], pb[8];
  std::memcpy(pa, a, 32);
  std::memcpy(pb, b, 32);
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    unsigned int v = pa[i] ^ pb[i];
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// Best / second-best scan as written in ORBmatcher.cpp:102-134 (strict '<', first minimum wins),
// over all T targets in ascending index — the dense "brute force" of BASELINE.json north_star.
void ora_hamming_dense_best2(const uint8_t* q, int Q, const uint8_t* t, int T, int32_t* best_idx,
                             int32_t* best_dist, int32_t* second_dist) {
  for (int i = 0; i < Q; i++) {
    int bestDist = 256, bestDist2 = 256, bestIdx = -1;
    for (int j = 0; j < T; j++) {
      const int dist = ora_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2;
  }
}

// same scan over an ordered candidate list per query (the list GetFeaturesInArea returns)
void ora_hamming_csr(const uint8_t* q, int Q, const uint8_t* t, const int32_t* cand_off, const int32_t* cand_idx,
                     uint16_t* cand_dist, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
  for (int i = 0; i < Q; i++) {
    int bestDist = 256, bestDist2 = 256, bestIdx = -1;
    for (int s = cand_off[i]; s < cand_off[i + 1]; s++) {
      const int j = cand_idx[s];
      const int dist = ora_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (cand_dist) cand_dist[s] = (uint16_t)dist;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx = j; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (best_idx) { best_idx[i] = bestIdx; best_dist[i] = bestDist; second_dist[i] = bestDist2; }
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Frame grid: Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea
// cslam/src/Frame.cpp:103-118, 255-265, 200-253 ; FRAME_GRID_COLS 75, ROWS 48 (Frame.h:51-52)
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int GRID_COLS = 75, GRID_ROWS = 48;

struct Grid {
  float minX, minY, maxX, maxY, wInv, hInv;
  std::vector<int> cell[GRID_COLS][GRID_ROWS];
  const float* kx; const float* ky; const int32_t* oct; int N;
};

void build_grid(Grid& g, const float* kx, const float* ky, const int32_t* oct, int N, float minX, float minY,
                float maxX, float maxY) {
  g.minX = minX; g.minY = minY; g.maxX = maxX; g.maxY = maxY;
  g.wInv = static_cast<float>(GRID_COLS) / static_cast<float>(maxX - minX);   // Frame.cpp:87
  g.hInv = static_cast<float>(GRID_ROWS) / static_cast<float>(maxY - minY);   // Frame.cpp:88
  g.kx = kx; g.ky = ky; g.oct = oct; g.N = N;
  for (int i = 0; i < N; i++) {
    // PosInGrid: round(), Frame.cpp:257-258
    const int posX = (int)std::round((kx[i] - minX) * g.wInv);
    const int posY = (int)std::round((ky[i] - minY) * g.hInv);
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
    g.cell[posX][posY].push_back(i);
  }
}

// Frame::GetFeaturesInArea, Frame.cpp:200-253 (floor/ceil cell range, |dx|<r && |dy|<r,
// candidate order = ix-major, then iy, then insertion order)
void features_in_area(const Grid& g, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) {
  out.clear();
  const int nMinCellX = std::max(0, (int)std::floor((x - g.minX - r) * g.wInv));
  if (nMinCellX >= GRID_COLS) return;
  const int nMaxCellX = std::min((int)GRID_COLS - 1, (int)std::ceil((x - g.minX + r) * g.wInv));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - g.minY - r) * g.hInv));
  if (nMinCellY >= GRID_ROWS) return;
  const int nMaxCellY = std::min((int)GRID_ROWS - 1, (int)std::ceil((y - g.minY + r) * g.hInv));
  if (nMaxCellY < 0) return;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<int>& vCell = g.cell[ix][iy];
      for (size_t j = 0; j < vCell.size(); j++) {
        const int k = vCell[j];
        if (bCheckLevels) {
          if (g.oct[k] < minLevel) continue;
          if (maxLevel >= 0 && g.oct[k] > maxLevel) continue;
        }
        const float distx = g.kx[k] - x;
        const float disty = g.ky[k] - y;
        if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(k);
      }
    }
}

// ORBmatcher::ComputeThreeMaxima, ORBmatcher.cpp:1607-1648
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
}  // namespace

extern "C" {

// Candidate CSR for a batch of window queries — the lists the reference builds one at a time
// inside the Search* loops.  Returns the total number of candidates; cand_idx may be NULL to size.
int64_t ora_grid_candidates(const float* kx, const float* ky, const int32_t* oct, int N, float minX, float minY,
                            float maxX, float maxY, const float* qx, const float* qy, const float* qr,
                            const int32_t* qminl, const int32_t* qmaxl, int Q, int32_t* cand_off, int32_t* cand_idx,
                            int64_t cap) {
  Grid* g = new Grid();
  build_grid(*g, kx, ky, oct, N, minX, minY, maxX, maxY);
  std::vector<int> v;
  int64_t n = 0;
  for (int i = 0; i < Q; i++) {
    cand_off[i] = (int32_t)n;
    features_in_area(*g, qx[i], qy[i], qr[i], qminl[i], qmaxl[i], v);
    for (int k : v) { if (cand_idx && n < cap) cand_idx[n] = k; n++; }
  }
  cand_off[Q] = (int32_t)n;
  delete g;
  return n;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<mpptr>&, th) — ORBmatcher.cpp:71-148.
// Flat restatement.  Per map point: in_view (mbTrackInView && !isBad()), projection (mTrackProjX/Y),
// predicted level, view cosine, descriptor.  frame_mp[idx] >= 0 stands for
// "F.mvpMapPoints[idx] && Observations()>0" (a claimed feature); the function assigns
// frame_mp[bestIdx] = iMP exactly where the reference assigns the pointer, so later map points see
// the claims of earlier ones.  Returns nmatches.
int ora_search_by_projection_mp(const float* kx, const float* ky, const int32_t* oct, const uint8_t* fdesc, int N,
                                float minX, float minY, float maxX, float maxY, const float* scale_factors,
                                int n_mp, const uint8_t* mp_in_view, const float* mp_proj_x, const float* mp_proj_y,
                                const int32_t* mp_level, const float* mp_view_cos, const uint8_t* mp_desc, float th,
                                float nnratio, int32_t* frame_mp /* in/out [N], -1 = free */) {
  const int TH_HIGH = 100;
  Grid* g = new Grid();
  build_grid(*g, kx, ky, oct, N, minX, minY, maxX, maxY);
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < n_mp; iMP++) {
    if (!mp_in_view[iMP]) continue;
    const int nPredictedLevel = mp_level[iMP];
    float r = (mp_view_cos[iMP] > 0.998) ? 2.5f : 4.0f;   // RadiusByViewingCos :150-156
    if (bFactor) r *= th;
    features_in_area(*g, mp_proj_x[iMP], mp_proj_y[iMP], r * scale_factors[nPredictedLevel], nPredictedLevel - 1,
                     nPredictedLevel, vIndices);
    if (vIndices.empty()) continue;
    const uint8_t* MPdescriptor = mp_desc + (size_t)iMP * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vIndices) {
      if (frame_mp[idx] >= 0) continue;
      const int dist = ora_descriptor_distance(MPdescriptor, fdesc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = oct[idx]; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = oct[idx]; bestDist2 = dist; }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      frame_mp[bestIdx] = iMP;
      nmatches++;
    }
  }
  delete g;
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th) — ORBmatcher.cpp:1350-1476.
// Flat restatement: the f32 projection (:1381-1397) is done by the caller-visible helper below so
// that GPU and oracle share inputs; here we take per-last-frame-feature: valid (has MP && !outlier
// && invzc>=0 && inside image bounds), u, v, last octave, last angle (mvKeysUn[i].angle), MP descriptor.
int ora_search_by_projection_last(const float* kx, const float* ky, const int32_t* oct, const float* kangle,
                                  const uint8_t* fdesc, int N, float minX, float minY, float maxX, float maxY,
                                  const float* scale_factors, int n_last, const uint8_t* l_valid, const float* l_u,
                                  const float* l_v, const int32_t* l_octave, const float* l_angle,
                                  const uint8_t* l_mp_desc, float th, int check_orientation,
                                  int32_t* cur_mp /* in/out [N] */) {
  const int TH_HIGH = 100, HISTO_LENGTH = 30;
  Grid* g = new Grid();
  build_grid(*g, kx, ky, oct, N, minX, minY, maxX, maxY);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;   // upstream quirk kept: bins are 30 degrees wide (:1358)
  std::vector<int> vIndices2;
  for (int i = 0; i < n_last; i++) {
    if (!l_valid[i]) continue;
    const int nLastOctave = l_octave[i];
    const float radius = th * scale_factors[nLastOctave];
    features_in_area(*g, l_u[i], l_v[i], radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = l_mp_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (cur_mp[i2] >= 0) continue;
      const int dist = ora_descriptor_distance(dMP, fdesc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      cur_mp[bestIdx2] = i;
      nmatches++;
      if (check_orientation) {
        float rot = l_angle[i] - kangle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0; j < rotHist[i].size(); j++) { cur_mp[rotHist[i][j]] = -1; nmatches--; }
  }
  delete g;
  return nmatches;
}

}  // extern "C"

// ================================================================================================
// BoW-bucketed searches.  A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) is passed
// flat: ascending node ids fv_node[nn], fv_off[nn+1], fv_idx[] (feature indices in insertion order).
// ================================================================================================
namespace {
struct FV { const int32_t* node; const int32_t* off; const int32_t* idx; int nn; };
// the merge-join of two ordered maps with lower_bound jumps (ORBmatcher.cpp:199-283) visits exactly the
// common node ids in ascending order; returns pairs (position in a, position in b)
std::vector<std::pair<int, int>> common_nodes(const FV& a, const FV& b) {
  std::vector<std::pair<int, int>> r;
  int i = 0, j = 0;
  while (i < a.nn && j < b.nn) {
    if (a.node[i] == b.node[j]) { r.push_back({i, j}); i++; j++; }
    else if (a.node[i] < b.node[j]) i = (int)(std::lower_bound(a.node, a.node + a.nn, b.node[j]) - a.node);
    else j = (int)(std::lower_bound(b.node, b.node + b.nn, a.node[i]) - b.node);
  }
  return r;
}
int hist_bin(float rot) {   // rotation histogram bin, factor = 1/HISTO_LENGTH (upstream quirk)
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * (1.0f / 30));
  if (bin == 30) bin = 0;
  return bin;
}
}  // namespace

extern "C" {

// ORBmatcher::SearchByBoW(kfptr pKF, Frame &F, vector<mpptr>&) — ORBmatcher.cpp:178-306.
// kf_has_mp[i] = vpMapPointsKF[i] && !isBad().  matches_f[F.N] out: KF feature index whose map point was
// assigned to the frame feature, or -1.  Returns nmatches.
int ora_search_by_bow_kf_frame(const int32_t* kf_node, const int32_t* kf_off, const int32_t* kf_idx, int kf_nn,
                               const int32_t* f_node, const int32_t* f_off, const int32_t* f_idx, int f_nn,
                               const uint8_t* kf_has_mp, const uint8_t* kf_desc, const float* kf_angle,
                               const uint8_t* f_desc, const float* f_angle, int f_n, float nnratio, int check_ori,
                               int32_t* matches_f) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  for (int i = 0; i < f_n; i++) matches_f[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  const FV a{kf_node, kf_off, kf_idx, kf_nn}, b{f_node, f_off, f_idx, f_nn};
  for (auto pr : common_nodes(a, b)) {
    for (int s1 = kf_off[pr.first]; s1 < kf_off[pr.first + 1]; s1++) {
      const int realIdxKF = kf_idx[s1];
      if (!kf_has_mp[realIdxKF]) continue;
      int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
      for (int s2 = f_off[pr.second]; s2 < f_off[pr.second + 1]; s2++) {
        const int realIdxF = f_idx[s2];
        if (matches_f[realIdxF] >= 0) continue;
        const int dist = ora_descriptor_distance(kf_desc + (size_t)realIdxKF * 32, f_desc + (size_t)realIdxF * 32);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist1 <= TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
        matches_f[bestIdxF] = realIdxKF;
        if (check_ori) rotHist[hist_bin(kf_angle[realIdxKF] - f_angle[bestIdxF])].push_back(bestIdxF);
        nmatches++;
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int k : rotHist[i]) { matches_f[k] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByBoW(kfptr pKF1, kfptr pKF2, vector<mpptr>&) — ORBmatcher.cpp:565-698 (strict < TH_LOW).
// matches12[n1] out: feature index in KF2 or -1.
int ora_search_by_bow_kf_kf(const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const int32_t* n2, const int32_t* o2,
                            const int32_t* i2, int nn2, const uint8_t* has_mp1, const uint8_t* has_mp2, const uint8_t* desc1,
                            const float* angle1, int N1, const uint8_t* desc2, const float* angle2, int N2, float nnratio,
                            int check_ori, int32_t* matches12) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  std::vector<char> vbMatched2(N2, 0);
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  const FV a{n1, o1, i1, nn1}, b{n2, o2, i2, nn2};
  for (auto pr : common_nodes(a, b)) {
    for (int s1 = o1[pr.first]; s1 < o1[pr.first + 1]; s1++) {
      const int idx1 = i1[s1];
      if (!has_mp1[idx1]) continue;
      int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
      for (int s2 = o2[pr.second]; s2 < o2[pr.second + 1]; s2++) {
        const int idx2 = i2[s2];
        if (vbMatched2[idx2] || !has_mp2[idx2]) continue;
        const int dist = ora_descriptor_distance(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
      if (bestDist1 < TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
        matches12[idx1] = bestIdx2;
        vbMatched2[bestIdx2] = 1;
        if (check_ori) rotHist[hist_bin(angle1[idx1] - angle2[bestIdx2])].push_back(idx1);
        nmatches++;
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int k : rotHist[i]) { matches12[k] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchForTriangulation — ORBmatcher.cpp:700-852 (+ CheckDistEpipolarLine :159-176).
// F12 row-major 3x3 f32, epipole (ex,ey) in image 2 computed by the caller (:708-714), sigma2 = pKF2->mvLevelSigma2,
// sf2 = pKF2->mvScaleFactors.  Note: the reference never sets vbMatched2 (kept).
int ora_search_for_triangulation(const int32_t* n1, const int32_t* o1, const int32_t* i1, int nn1, const int32_t* n2, const int32_t* o2,
                                 const int32_t* i2, int nn2, const uint8_t* has_mp1, const uint8_t* has_mp2, const uint8_t* desc1,
                                 const float* x1, const float* y1, const float* angle1, int N1, const uint8_t* desc2, const float* x2,
                                 const float* y2, const int32_t* oct2, const float* angle2, int N2, const float* F12, float ex, float ey,
                                 const float* sigma2_2, const float* sf2, int check_ori, int32_t* matches12) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  int nmatches = 0;
  const FV a{n1, o1, i1, nn1}, b{n2, o2, i2, nn2};
  for (auto pr : common_nodes(a, b)) {
    for (int s1 = o1[pr.first]; s1 < o1[pr.first + 1]; s1++) {
      const int idx1 = i1[s1];
      if (has_mp1[idx1]) continue;
      int bestDist = TH_LOW, bestIdx2 = -1;
      for (int s2 = o2[pr.second]; s2 < o2[pr.second + 1]; s2++) {
        const int idx2 = i2[s2];
        if (has_mp2[idx2]) continue;
        const int dist = ora_descriptor_distance(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
        if (dist > TH_LOW || dist > bestDist) continue;
        const float distex = ex - x2[idx2], distey = ey - y2[idx2];
        if (distex * distex + distey * distey < 100 * sf2[oct2[idx2]]) continue;
        // CheckDistEpipolarLine
        const float la = x1[idx1] * F12[0] + y1[idx1] * F12[3] + F12[6];
        const float lb = x1[idx1] * F12[1] + y1[idx1] * F12[4] + F12[7];
        const float lc = x1[idx1] * F12[2] + y1[idx1] * F12[5] + F12[8];
        const float num = la * x2[idx2] + lb * y2[idx2] + lc;
        const float den = la * la + lb * lb;
        if (den == 0) continue;
        const float dsqr = num * num / den;
        if (dsqr < 3.84 * sigma2_2[oct2[idx2]]) { bestIdx2 = idx2; bestDist = dist; }
      }
      if (bestIdx2 >= 0) {
        matches12[idx1] = bestIdx2;
        nmatches++;
        if (check_ori) rotHist[hist_bin(angle1[idx1] - angle2[bestIdx2])].push_back(idx1);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int k : rotHist[i]) { matches12[k] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchForInitialization — ORBmatcher.cpp:448-563.  prev_xy in/out (vbPrevMatched), matches12 out.
int ora_search_for_initialization(const float* x1, const float* y1, const int32_t* oct1, const float* angle1, const uint8_t* desc1, int N1,
                                  const float* x2, const float* y2, const int32_t* oct2, const float* angle2, const uint8_t* desc2, int N2,
                                  float minX, float minY, float maxX, float maxY, float* prev_xy, int window, float nnratio,
                                  int check_ori, int32_t* matches12) {
  const int TH_LOW = 50, HISTO_LENGTH = 30;
  Grid* g = new Grid();
  build_grid(*g, x2, y2, oct2, N2, minX, minY, maxX, maxY);
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vMatchedDistance(N2, INT32_MAX), vnMatches21(N2, -1);
  int nmatches = 0;
  std::vector<int> vIndices2;
  for (int i1 = 0; i1 < N1; i1++) {
    const int level1 = oct1[i1];
    if (level1 > 0) continue;
    features_in_area(*g, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window, level1, level1, vIndices2);
    if (vIndices2.empty()) continue;
    int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      const int dist = ora_descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnratio) {
      if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
      matches12[i1] = bestIdx2;
      vnMatches21[bestIdx2] = i1;
      vMatchedDistance[bestIdx2] = bestDist;
      nmatches++;
      if (check_ori) rotHist[hist_bin(angle1[i1] - angle2[bestIdx2])].push_back(i1);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < N1; i1++) if (matches12[i1] >= 0) { prev_xy[2 * i1] = x2[matches12[i1]]; prev_xy[2 * i1 + 1] = y2[matches12[i1]]; }
  delete g;
  return nmatches;
}

// MapPoint::UpdateNormalAndDepth (MapPoint.cpp:779-823), batch form with flat arrays.  cv::Mat CV_32F semantics restated ([EXT] OpenCV):
// Mat - Mat elementwise in f32; cv::norm(CV_32F) = sqrt of the squares accumulated in double; Mat / s = convertTo(alpha = (float)(1.0 / s));
// Mat + MatExpr(scaled) = f32 multiply then f32 add (no FMA in an x86-64 baseline build).
void ora_update_normal_and_depth(int n_pt, const float* pos, const int32_t* obs_off, const int32_t* obs_kf, const float* kf_center,
                                 const int32_t* ref_kf, const int32_t* ref_level, const float* scale_factors, int n_levels, float* normal,
                                 float* min_dist, float* max_dist) {
  for (int i = 0; i < n_pt; i++) {
    if (obs_off[i + 1] == obs_off[i]) continue;   // observations.empty() -> return (:796-797)
    const float* P = pos + 3 * (size_t)i;
    float nrm[3] = {0.f, 0.f, 0.f};               // cv::Mat::zeros(3,1,CV_32F)
    int n = 0;
    for (int o = obs_off[i]; o < obs_off[i + 1]; o++) {
      const float* Owi = kf_center + 3 * (size_t)obs_kf[o];
      const float ni[3] = {P[0] - Owi[0], P[1] - Owi[1], P[2] - Owi[2]};                               // normali = mWorldPos - Owi
      const double len = std::sqrt((double)ni[0] * ni[0] + (double)ni[1] * ni[1] + (double)ni[2] * ni[2]);   // cv::norm(normali)
      const float a = (float)(1.0 / len);
      for (int c = 0; c < 3; c++) { volatile float t = ni[c] * a; nrm[c] = nrm[c] + t; }             // normal = normal + normali / norm
      n++;
    }
    const float* Or = kf_center + 3 * (size_t)ref_kf[i];
    const float PC[3] = {P[0] - Or[0], P[1] - Or[1], P[2] - Or[2]};                                     // PC = Pos - pRefKF->GetCameraCenter()
    const float dist = (float)std::sqrt((double)PC[0] * PC[0] + (double)PC[1] * PC[1] + (double)PC[2] * PC[2]);
    const float levelScaleFactor = scale_factors[ref_level[i]];
    max_dist[i] = dist * levelScaleFactor;                                                              // mfMaxDistance
    min_dist[i] = max_dist[i] / scale_factors[n_levels - 1];                                            // mfMinDistance
    const float an = (float)(1.0 / (double)n);
    for (int c = 0; c < 3; c++) normal[3 * (size_t)i + c] = nrm[c] * an;                               // mNormalVector = normal / n
  }
}

// ComputeThreeMaxima on a histogram given by its bin sizes (pinning test against oracle/_ref)
void ora_three_maxima(const int32_t* counts, int L, int32_t* out3) {
  std::vector<std::vector<int>> histo((size_t)L);
  for (int i = 0; i < L; i++) histo[(size_t)i].assign((size_t)counts[i], 0);
  int a = -1, b = -1, c = -1;
  three_maxima(histo.data(), L, a, b, c);
  out3[0] = a; out3[1] = b; out3[2] = c;
}

}  // extern "C"

// ================================================================================================
// Projected window searches on a KeyFrame: the common inner loop of
//   Fuse(kfptr, vector<mpptr>, th)                      ORBmatcher.cpp:854-993   (chi2 gate :942-951, TH_LOW)
//   Fuse(kfptr, Scw, points, th, replace)               ORBmatcher.cpp:995-1122  (no gate, TH_LOW)
//   SearchByProjection(kfptr, Scw, points, matched, th) ORBmatcher.cpp:308-446   (skips vpMatched, claims, TH_LOW)
//   SearchBySim3 (both directions)                       ORBmatcher.cpp:1124-1348 (no gate, TH_HIGH, mutual check)
// The f32 projection / depth / viewing-angle tests that precede the loop (cv::Mat arithmetic on the KeyFrame
// pose) stay with the caller: valid[i], u[i], v[i], level[i] are their outcome.
// KeyFrame::GetFeaturesInArea(x,y,r) (KeyFrame.cpp:1162-1201) has no level filter; the loops filter
// kpLevel in [nPredictedLevel-1, nPredictedLevel] themselves.
//   chi2_gate:   skip candidates with e2 * invSigma2[kpLevel] > 5.99  (Fuse only)
//   matched:     nullable in/out [N]; >= 0 entries are skipped; when `claim` is set, an accepted point i with
//                no_claim[i] == 0 writes matched[bestIdx] = i (SearchByProjection's vpMatched)
//   best_idx[i] = accepted feature or -1, best_dist[i] = its distance (INT32_MAX if no candidate)
// Returns the number of accepted points.
extern "C" int ora_projected_window_search(const float* kx, const float* ky, const int32_t* oct, const uint8_t* kdesc, int N, float minX,
                                           float minY, float maxX, float maxY, const float* scale_factors, const float* inv_sigma2,
                                           int n_pts, const uint8_t* valid, const float* u, const float* v, const int32_t* level,
                                           const uint8_t* pdesc, float th, int chi2_gate, int dist_threshold, int32_t* matched, int claim,
                                           const uint8_t* no_claim, int32_t* best_idx, int32_t* best_dist) {
  Grid* g = new Grid();
  build_grid(*g, kx, ky, oct, N, minX, minY, maxX, maxY);
  int n_acc = 0;
  std::vector<int> vIndices;
  for (int i = 0; i < n_pts; i++) {
    best_idx[i] = -1; best_dist[i] = INT32_MAX;
    if (!valid[i]) continue;
    const int nPredictedLevel = level[i];
    const float radius = th * scale_factors[nPredictedLevel];
    features_in_area(*g, u[i], v[i], radius, -1, -1, vIndices);
    if (vIndices.empty()) continue;
    int bestDist = INT32_MAX, bestIdx = -1;
    for (int idx : vIndices) {
      if (matched && matched[idx] >= 0) continue;
      const int kpLevel = oct[idx];
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (chi2_gate) {
        const float ex = u[i] - kx[idx], ey = v[i] - ky[idx];
        const float e2 = ex * ex + ey * ey;
        if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
      }
      const int dist = ora_descriptor_distance(pdesc + (size_t)i * 32, kdesc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= dist_threshold) {
      best_idx[i] = bestIdx; best_dist[i] = bestDist;
      if (matched && claim && !(no_claim && no_claim[i])) matched[bestIdx] = i;
      n_acc++;
    }
  }
  delete g;
  return n_acc;
}

// MapPoint::ComputeDistinctiveDescriptors — cslam/src/MapPoint.cpp:929-994, batched flat restatement
extern "C" void ora_distinctive_descriptors(const uint8_t* desc, const int32_t* off, int P, int32_t* best_local_idx) {
  for (int p = 0; p < P; p++) {
    const int N = off[p + 1] - off[p];
    if (N <= 0) { best_local_idx[p] = -1; continue; }
    const uint8_t* D = desc + (size_t)off[p] * 32;
    std::vector<std::vector<float>> Distances(N, std::vector<float>(N, 0.f));
    for (int i = 0; i < N; i++) {
      Distances[i][i] = 0;
      for (int j = i + 1; j < N; j++) {
        const int distij = ora_descriptor_distance(D + (size_t)i * 32, D + (size_t)j * 32);
        Distances[i][j] = (float)distij; Distances[j][i] = (float)distij;
      }
    }
    int BestMedian = INT32_MAX, BestIdx = 0;
    for (int i = 0; i < N; i++) {
      std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
      std::sort(vDists.begin(), vDists.end());
      const int median = vDists[(size_t)(0.5 * (N - 1))];
      if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    best_local_idx[p] = BestIdx;
  }
}

// ================================================================================================
// DBoW2: TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
// cslam/thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1190 (TF_IDF weighting, L1 norm as ORBvoc uses) and the
// per-feature descent :1218-1260; FORB::distance = the SWAR popcount (FORB.cpp:81-101); BowVector::addWeight /
// normalize (BowVector.cpp:34-84); FeatureVector::addFeature (FeatureVector.cpp:31-45).
// Flat tree: see ccm_vocab_create in include/ccm_hip.h.  Outputs: per feature (word, weight, node); bow_ids/bow_vals =
// the normalised BowVector in ascending word order (returns its size); the FeatureVector follows from `node` grouped
// in ascending node order with features in index order (tests build it with numpy).
// ================================================================================================
#include <map>
extern "C" int ora_bow_transform(int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc,
                                 const int32_t* word_id, const double* weight, const uint8_t* desc, int N, int levelsup,
                                 int32_t* word_out, double* weight_out, int32_t* node_out, int32_t* bow_ids, double* bow_vals) {
  (void)n_nodes;
  std::map<int32_t, double> v;
  for (int f = 0; f < N; f++) {
    const int nid_level = L - levelsup;
    int nid = 0, final_id = 0, current_level = 0;
    while (child_off[final_id + 1] > child_off[final_id]) {
      ++current_level;
      const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
      int best = child_id[c0];
      double best_d = ora_descriptor_distance(desc + (size_t)f * 32, node_desc + (size_t)best * 32);
      for (int s = c0 + 1; s < c1; s++) {
        const int id = child_id[s];
        const double d = ora_descriptor_distance(desc + (size_t)f * 32, node_desc + (size_t)id * 32);
        if (d < best_d) { best_d = d; best = id; }
      }
      final_id = best;
      if (current_level == nid_level) nid = final_id;
    }
    word_out[f] = word_id[final_id]; weight_out[f] = weight[final_id]; node_out[f] = nid;
    if (weight[final_id] > 0) v[word_id[final_id]] += weight[final_id];   // addWeight; "stopped" words (w == 0) are skipped
  }
  double norm = 0.0;
  for (auto& kv : v) norm += std::fabs(kv.second);
  int n = 0;
  for (auto& kv : v) { bow_ids[n] = kv.first; bow_vals[n] = norm > 0.0 ? kv.second / norm : kv.second; n++; }
  return n;
}


// ------------------------------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints / ComputeImageBounds (cslam/src/Frame.cpp:284-347).  cv::undistortPoints is [EXT]: OpenCV
// 4.2.0 imgproc/src/undistort.dispatch.cpp cvUndistortPointsInternal with cameraMatrix = mK (CV_32F widened to f64),
// distCoeffs = mDistCoef (k1 k2 p1 p2 [k3]; k4..k6, s1..s4, tau = 0), R = I, P = mK and the wrapper's default criteria
// TermCriteria(MAX_ITER, 5, 0.01): five fixed-point iterations in f64, no epsilon exit, result stored as f32.
namespace {
struct UndCam { double fx, fy, cx, cy, k[14]; };
void und_point(const UndCam& c, float xin, float yin, float* xo, float* yo) {
  double x = xin, y = yin;
  const double u = x, v = y;
  const double ifx = 1. / c.fx, ify = 1. / c.fy;
  x = (x - c.cx) * ifx;
  y = (y - c.cy) * ify;
  const double* k = c.k;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
    if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
    const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
    const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double RR[3][3] = {{c.fx, 0, c.cx}, {0, c.fy, c.cy}, {0, 0, 1}};
  const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
  const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
  const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}
UndCam make_cam(const float* K, const float* dist, int n_dist) {
  UndCam c{};
  c.fx = K[0]; c.fy = K[1]; c.cx = K[2]; c.cy = K[3];
  for (int i = 0; i < 14; i++) c.k[i] = 0;
  for (int i = 0; i < n_dist && i < 5; i++) c.k[i] = dist[i];
  return c;
}
}  // namespace

extern "C" void ora_undistort_points(const float* K, const float* dist, int n_dist, const float* xy_in, int n, float* xy_out) {
  if (n_dist == 0 || dist[0] == 0.0f) {   // mDistCoef.at<float>(0)==0.0 => mvKeysUn = mvKeys (Frame.cpp:286-290)
    for (int i = 0; i < 2 * n; i++) xy_out[i] = xy_in[i];
    return;
  }
  const UndCam c = make_cam(K, dist, n_dist);
  for (int i = 0; i < n; i++) und_point(c, xy_in[2 * i], xy_in[2 * i + 1], &xy_out[2 * i], &xy_out[2 * i + 1]);
}

// ComputeImageBounds (Frame.cpp:314-347): bounds[4] = mnMinX mnMinY mnMaxX mnMaxY
extern "C" void ora_image_bounds(const float* K, const float* dist, int n_dist, int w, int h, float* bounds) {
  if (n_dist == 0 || dist[0] == 0.0f) { bounds[0] = 0.f; bounds[1] = 0.f; bounds[2] = (float)w; bounds[3] = (float)h; return; }
  const UndCam c = make_cam(K, dist, n_dist);
  const float cx[4] = {0.f, (float)w, 0.f, (float)w}, cy[4] = {0.f, 0.f, (float)h, (float)h};
  float ux[4], uy[4];
  for (int i = 0; i < 4; i++) und_point(c, cx[i], cy[i], &ux[i], &uy[i]);
  bounds[0] = std::min(ux[0], ux[2]); bounds[2] = std::max(ux[1], ux[3]);
  bounds[1] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
}

// AssignFeaturesToGrid (Frame.cpp:103-118) flattened: cell c = x * GRID_ROWS + y, members in push_back order
extern "C" void ora_build_grid(const float* kx, const float* ky, int N, float minX, float minY, float maxX, float maxY,
                               int32_t* cell_off /* COLS*ROWS+1 */, int32_t* cell_idx /* N */) {
  Grid* g = new Grid();
  std::vector<int32_t> oct(N > 0 ? N : 1, 0);
  build_grid(*g, kx, ky, oct.data(), N, minX, minY, maxX, maxY);
  int n = 0;
  for (int x = 0; x < GRID_COLS; x++)
    for (int y = 0; y < GRID_ROWS; y++) {
      cell_off[x * GRID_ROWS + y] = n;
      for (int k : g->cell[x][y]) cell_idx[n++] = k;
    }
  cell_off[GRID_COLS * GRID_ROWS] = n;
  delete g;
}


// ------------------------------------------------------------------------------------------------------------------
// Frame::isInFrustum (cslam/src/Frame.cpp:139-198) over a batch of map points, flat restatement.
// frame = [Rcw 9 | tcw 3 | Ow 3 | fx fy cx cy | minX maxX minY maxY | logScaleFactor] (f32) + nScaleLevels.
// cv::Mat arithmetic is [EXT] (OpenCV 4.2.0): mRcw*P+mtcw = gemm small-matrix path (f32 sum of three products, then
// (float)(t*alpha + c*beta) in f64; baseline non-FMA build); cv::norm / Mat::dot accumulate in f64.
extern "C" void ora_is_in_frustum(const float* frame, int nScaleLevels, int n, const float* P, const float* Pn, const float* dmin,
                                  const float* dmax, float viewingCosLimit, uint8_t* in_view, float* pu, float* pv, int32_t* lvl,
                                  float* pcos) {
  const float* R = frame; const float* t = frame + 9; const float* Ow = frame + 12;
  const float fx = frame[15], fy = frame[16], cx = frame[17], cy = frame[18];
  const float minX = frame[19], maxX = frame[20], minY = frame[21], maxY = frame[22], logSF = frame[23];
  for (int i = 0; i < n; i++) {
    in_view[i] = 0; pu[i] = 0; pv[i] = 0; lvl[i] = 0; pcos[i] = 0;      // pMP->mbTrackInView = false (:141)
    const float* X = P + 3 * (size_t)i;
    float Pc[3];
    for (int r = 0; r < 3; r++) {
      const float acc = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2];
      Pc[r] = (float)((double)acc * 1.0 + (double)t[r] * 1.0);
    }
    const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
    if (PcZ < 0.0f) continue;
    const float invz = 1.0f / PcZ;
    const float u = fx * PcX * invz + cx;
    const float v = fy * PcY * invz + cy;
    if (u < minX || u > maxX) continue;
    if (v < minY || v > maxY) continue;
    const float maxDistance = 1.2f * dmax[i];      // MapPoint::GetMaxDistanceInvariance (MapPoint.cpp:831-835)
    const float minDistance = 0.8f * dmin[i];
    const float PO[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
    double s2 = 0;
    for (int k = 0; k < 3; k++) s2 += (double)PO[k] * PO[k];
    const float dist = (float)std::sqrt(s2);
    if (dist < minDistance || dist > maxDistance) continue;
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)PO[k] * Pn[3 * (size_t)i + k];
    const float viewCos = (float)(dot / dist);
    if (viewCos < viewingCosLimit) continue;
    const float ratio = dmax[i] / dist;             // MapPoint::PredictScale (MapPoint.cpp:854-869)
    int nScale = (int)std::ceil(std::log(ratio) / logSF);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nScaleLevels) nScale = nScaleLevels - 1;
    in_view[i] = 1; pu[i] = u; pv[i] = v; lvl[i] = nScale; pcos[i] = viewCos;
  }
}
