// orb_ref.cpp — CPU ORACLE (test infrastructure, NOT product code) for cslam::ORBextractor.
//
// Restates cslam/src/ORBextractor.cpp (ctor :579-639, ComputePyramid :1280-1304,
// ComputeKeyPointsOctTree :933-1024, DistributeOctTree/DivideNode :650-931, IC_Angle :68-95,
// computeOrbDescriptor :100-316, operator() :1216-1278) together with the OpenCV primitives it calls.
// OpenCV is NOT under /root/reference and not installed here; its primitives are restated from the
// published algorithms of the pinned version OpenCV 4.2.0 (the version ROS Noetic ships, readme.md:56):
//   cv::resize INTER_LINEAR 8UC1      imgproc/resize.cpp   (11-bit fixed-point coefficients, HResizeLinear/VResizeLinear)
//   cv::copyMakeBorder REFLECT_101    not needed: no result ever reads the 19-px border
//   cv::FAST(img,kps,t,true) 9/16     features2d/fast.cpp, fast_score.cpp (FAST_t<16>, cornerScore<16>)
//   cv::GaussianBlur 7x7 sigma 2 8U   imgproc/smooth.dispatch.cpp + fixedpoint (ufixedpoint16 bit-exact path):
//                                     8.8 fixed-point kernel {18,34,48,56,48,34,18}/256, exact separable sums,
//                                     rounding (v + 2^15) >> 16, BORDER_REFLECT_101
//   cv::fastAtan2                     core/mathfuncs_core.simd.hpp scalar polynomial
//   cvRound                           round half to even
//   cos/sin(float)                    glibc cosf/sinf — the REAL libm of this machine is called here
//
// PARITY PIN: the reference has no tests/golden vectors for this path and cannot be compiled here
// ("parity unpinned" against a running reference binary).  This oracle is pinned by hand-derived KATs in
// tests/test_oracle_orb.py (FAST on synthetic corners, blur kernel sum/impulse response, resize of
// constant/ramp images, descriptor of a flat patch == 0, octree invariants).
//
// Defined tie-break: DistributeOctTree sorts pair<int,ExtractorNode*> (:852), i.e. ties
// in node size are broken by heap address in the reference (not reproducible even run to run).  The oracle
// breaks them by node creation order (earlier node = lower address).
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <list>
#include <algorithm>
#include "orb_pattern.h"

#include "cv_prims.h"   // the restated OpenCV primitives (shared with the look-alike cv:: API of oracle/ref_shim)

namespace {
using namespace cvprims;

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

// ---- ExtractorNode / DistributeOctTree (ORBextractor.cpp:650-931) -----------------------------------
struct Pt2i { int x, y; };
struct Node {
  std::vector<KP> vKeys;
  Pt2i UL, UR, BL, BR;
  std::list<Node>::iterator lit;
  bool bNoMore = false;
  int seq = 0;   // creation order: the oracle's stand-in for the heap address used as sort tie-break
  void Divide(Node& n1, Node& n2, Node& n3, Node& n4) const {
    const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
    n1.UL = UL; n1.UR = {UL.x + halfX, UL.y}; n1.BL = {UL.x, UL.y + halfY}; n1.BR = {UL.x + halfX, UL.y + halfY};
    n2.UL = n1.UR; n2.UR = UR; n2.BL = n1.BR; n2.BR = {UR.x, UL.y + halfY};
    n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = BL; n3.BR = {n1.BR.x, BL.y};
    n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = BR;
    for (size_t i = 0; i < vKeys.size(); i++) {
      const KP& kp = vKeys[i];
      if (kp.x < n1.UR.x) { if (kp.y < n1.BR.y) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp); }
      else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
      else n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
  }
};

std::vector<KP> distribute_octree(const std::vector<KP>& vToDistributeKeys, int minX, int maxX, int minY, int maxY, int N) {
  int seq = 0;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> lNodes;
  std::vector<Node*> vpIniNodes(nIni);
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.UL = {(int)(hX * static_cast<float>(i)), 0};
    ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = {ni.UL.x, maxY - minY};
    ni.BR = {ni.UR.x, maxY - minY};
    ni.seq = seq++;
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
    const KP& kp = vToDistributeKeys[i];
    vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
  }
  std::list<Node>::iterator lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
    else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
    else lit++;
  }
  bool bFinish = false;
  std::vector<std::pair<int, Node*>> vSizeAndPointerToNode;
  auto addChild = [&](Node& n, bool count, int& nToExpand) {
    if (n.vKeys.size() > 0) {
      n.seq = seq++;
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (count) nToExpand++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  auto cmp = [](const std::pair<int, Node*>& a, const std::pair<int, Node*>& b) {
    if (a.first != b.first) return a.first < b.first;
    return a.second->seq < b.second->seq;   // reference: pointer value
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) { lit++; continue; }
      Node n1, n2, n3, n4;
      lit->Divide(n1, n2, n3, n4);
      addChild(n1, true, nToExpand); addChild(n2, true, nToExpand); addChild(n3, true, nToExpand); addChild(n4, true, nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
    else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<std::pair<int, Node*>> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), cmp);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          vPrev[j].second->Divide(n1, n2, n3, n4);
          int dummy = 0;
          addChild(n1, false, dummy); addChild(n2, false, dummy); addChild(n3, false, dummy); addChild(n4, false, dummy);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  std::vector<KP> vResultKeys;
  for (std::list<Node>::iterator l = lNodes.begin(); l != lNodes.end(); l++) {
    std::vector<KP>& vNodeKeys = l->vKeys;
    KP* pKP = &vNodeKeys[0];
    float maxResponse = pKP->response;
    for (size_t k = 1; k < vNodeKeys.size(); k++)
      if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
    vResultKeys.push_back(*pKP);
  }
  return vResultKeys;
}

// ---- the extractor -----------------------------------------------------------------------------------
struct Orb {
  int nfeatures, nlevels, iniThFAST, minThFAST; float scaleFactor;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  // last frame
  std::vector<std::vector<uint8_t>> pyr, blur; std::vector<int> lw, lh;
  std::vector<std::vector<KP>> cand;

  Orb(int nf, float sf, int nl, int ini, int mn) : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf) {
    mvScaleFactor.resize(nl); mvLevelSigma2.resize(nl);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor; mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; }
    mvInvScaleFactor.resize(nl); mvInvLevelSigma2.resize(nl);
    for (int i = 0; i < nl; i++) { mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    mnFeaturesPerLevel.resize(nl);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cvRoundF(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = (int)std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  void level_size(int w, int h, int level, int& ow, int& oh) const {
    const float scale = mvInvScaleFactor[level];
    ow = cvRoundF((float)w * scale); oh = cvRoundF((float)h * scale);   // :1285
  }

  float ic_angle(const uint8_t* img, int step, float px, float py) const {   // :68-95
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)cvRoundF(py) * step + cvRoundF(px);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
      int v_sum = 0;
      const int d = umax[v];
      for (int u = -d; u <= d; ++u) {
        const int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
  }

  void descriptor(const KP& kpt, const uint8_t* img, int step, uint8_t* desc) const {   // :100-316
    const float factorPI = (float)(M_PI / 180.f);
    const float angle = (float)kpt.angle * factorPI;
    const float a = (float)cosf(angle), b = (float)sinf(angle);
    const uint8_t* center = img + (size_t)cvRoundF(kpt.y) * step + cvRoundF(kpt.x);
    const int8_t* pat = kOrbPattern31;
    for (int i = 0; i < 32; i++) {
      int val = 0;
      for (int k = 0; k < 8; k++) {
        const int idx = (i * 8 + k) * 4;
        const float x0 = pat[idx], y0 = pat[idx + 1], x1 = pat[idx + 2], y1 = pat[idx + 3];
        const int t0 = center[cvRoundF(x0 * b + y0 * a) * step + cvRoundF(x0 * a - y0 * b)];
        const int t1 = center[cvRoundF(x1 * b + y1 * a) * step + cvRoundF(x1 * a - y1 * b)];
        val |= (t0 < t1) << k;
      }
      desc[i] = (uint8_t)val;
    }
  }

  int extract(const uint8_t* img, int w, int h, int stride, KP* kps_out, uint8_t* desc_out, int cap) {
    // ComputePyramid (:1280-1304).  The 19-px REFLECT_101 border is never read by any later stage
    // (keypoints are >= 19 px from every level edge; the blur re-derives the same reflection), so the
    // levels are kept un-bordered.
    pyr.assign(nlevels, {}); blur.assign(nlevels, {}); lw.assign(nlevels, 0); lh.assign(nlevels, 0); cand.assign(nlevels, {});
    for (int level = 0; level < nlevels; level++) {
      level_size(w, h, level, lw[level], lh[level]);
      pyr[level].resize((size_t)lw[level] * lh[level]);
      if (level == 0) for (int y = 0; y < h; y++) std::memcpy(&pyr[0][(size_t)y * w], img + (size_t)y * stride, w);
      else resize_linear_u8(pyr[level - 1].data(), lw[level - 1], lh[level - 1], lw[level - 1], pyr[level].data(), lw[level], lh[level], lw[level]);
    }
    // ComputeKeyPointsOctTree (:933-1024)
    std::vector<std::vector<KP>> allKeypoints(nlevels);
    const float W = 30;
    for (int level = 0; level < nlevels; ++level) {
      const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
      const int maxBorderX = lw[level] - EDGE_THRESHOLD + 3, maxBorderY = lh[level] - EDGE_THRESHOLD + 3;
      std::vector<KP>& vToDistributeKeys = cand[level];
      const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
      const int nCols = (int)(width / W), nRows = (int)(height / W);
      if (nCols <= 0 || nRows <= 0) continue;
      const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
      const uint8_t* L = pyr[level].data();
      const int step = lw[level];
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          std::vector<KP> vKeysCell;
          const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
          fast9_16(L + (size_t)y0 * step + x0, cw, ch, step, iniThFAST, vKeysCell);
          if (vKeysCell.empty()) fast9_16(L + (size_t)y0 * step + x0, cw, ch, step, minThFAST, vKeysCell);
          for (KP& k : vKeysCell) { k.x += j * wCell; k.y += i * hCell; vToDistributeKeys.push_back(k); }
        }
      }
      std::vector<KP>& keypoints = allKeypoints[level];
      if (!vToDistributeKeys.empty())
        keypoints = distribute_octree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY, mnFeaturesPerLevel[level]);
      const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
      for (KP& k : keypoints) { k.x += minBorderX; k.y += minBorderY; k.octave = level; k.size = (float)scaledPatchSize; }
    }
    for (int level = 0; level < nlevels; ++level)
      for (KP& k : allKeypoints[level]) k.angle = ic_angle(pyr[level].data(), lw[level], k.x, k.y);
    // operator() epilogue (:1232-1277)
    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KP>& keypoints = allKeypoints[level];
      if (keypoints.empty()) continue;
      blur[level].resize(pyr[level].size());
      gaussian_blur7(pyr[level].data(), lw[level], lh[level], lw[level], blur[level].data(), lw[level]);
      for (size_t i = 0; i < keypoints.size(); i++) {
        if (offset + (int)i >= cap) break;
        descriptor(keypoints[i], blur[level].data(), lw[level], desc_out + (size_t)(offset + i) * 32);
      }
      if (level != 0) { const float scale = mvScaleFactor[level]; for (KP& k : keypoints) { k.x *= scale; k.y *= scale; } }
      for (size_t i = 0; i < keypoints.size() && offset + (int)i < cap; i++) kps_out[offset + i] = keypoints[i];
      offset += (int)keypoints.size();
    }
    return std::min(offset, cap);
  }
};

}  // namespace

extern "C" {

void* ora_orb_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th) { return new Orb(nfeatures, scale, nlevels, ini_th, min_th); }
void ora_orb_destroy(void* p) { delete (Orb*)p; }
int ora_orb_extract(void* p, const uint8_t* img, int w, int h, int stride, void* kps, uint8_t* desc, int cap) {
  return ((Orb*)p)->extract(img, w, h, stride, (KP*)kps, desc, cap);
}
void ora_orb_level_size(void* p, int w, int h, int level, int* lw, int* lh) { ((Orb*)p)->level_size(w, h, level, *lw, *lh); }
int ora_orb_get_level(void* p, int level, uint8_t* out) { Orb* o = (Orb*)p; std::memcpy(out, o->pyr[level].data(), o->pyr[level].size()); return (int)o->pyr[level].size(); }
int ora_orb_get_blur(void* p, int level, uint8_t* out) { Orb* o = (Orb*)p; std::memcpy(out, o->blur[level].data(), o->blur[level].size()); return (int)o->blur[level].size(); }
int ora_orb_get_candidates(void* p, int level, void* out, int cap) {
  Orb* o = (Orb*)p; const int n = (int)o->cand[level].size();
  if (out) std::memcpy(out, o->cand[level].data(), sizeof(KP) * std::min(n, cap));
  return n;
}
void ora_orb_tables(void* p, float* sf, float* isf, float* s2, float* is2, int32_t* nfeat, int32_t* umax16) {
  Orb* o = (Orb*)p;
  for (int i = 0; i < o->nlevels; i++) { sf[i] = o->mvScaleFactor[i]; isf[i] = o->mvInvScaleFactor[i]; s2[i] = o->mvLevelSigma2[i]; is2[i] = o->mvInvLevelSigma2[i]; nfeat[i] = o->mnFeaturesPerLevel[i]; }
  for (int i = 0; i < 16; i++) umax16[i] = o->umax[i];
}
// primitives, exposed for unit tests
void ora_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) { resize_linear_u8(src, sw, sh, sw, dst, dw, dh, dw); }
void ora_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst) { gaussian_blur7(src, w, h, w, dst, w); }
void ora_gaussian_kernel7(int32_t* out) { int k[7]; gaussian_kernel7_fixed(k); for (int i = 0; i < 7; i++) out[i] = k[i]; }
int ora_fast9_16(const uint8_t* img, int w, int h, int threshold, void* out, int cap) {
  std::vector<KP> v; fast9_16(img, w, h, w, threshold, v);
  std::memcpy(out, v.data(), sizeof(KP) * std::min<int>((int)v.size(), cap));
  return (int)v.size();
}
float ora_fast_atan2(float y, float x) { return fast_atan2(y, x); }
int ora_distribute_octree(const void* kps, int n, int minX, int maxX, int minY, int maxY, int N, void* out, int cap) {
  std::vector<KP> v((const KP*)kps, (const KP*)kps + n);
  std::vector<KP> r = distribute_octree(v, minX, maxX, minY, maxY, N);
  std::memcpy(out, r.data(), sizeof(KP) * std::min<int>((int)r.size(), cap));
  return (int)r.size();
}

}  // extern "C"
