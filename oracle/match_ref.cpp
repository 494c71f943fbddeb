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
