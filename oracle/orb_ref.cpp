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

namespace {

struct KP { float x, y, size, angle, response; int32_t octave; };

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }

// ---- cv::resize, INTER_LINEAR, 8UC1 (fixed point, INTER_RESIZE_COEF_BITS = 11) -------------------
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    // saturate_cast<short>(float) = cvRound then clamp
    ialpha[dx * 2] = (short)std::min(std::max(cvRoundF((1.f - fx) * 2048.f), -32768), 32767);
    ialpha[dx * 2 + 1] = (short)std::min(std::max(cvRoundF(fx * 2048.f), -32768), 32767);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = (short)std::min(std::max(cvRoundF((1.f - fy) * 2048.f), -32768), 32767);
    ibeta[dy * 2 + 1] = (short)std::min(std::max(cvRoundF(fy * 2048.f), -32768), 32767);
  }
  std::vector<int> row0(dw), row1(dw);
  for (int dy = 0; dy < dh; dy++) {
    const int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);       // clip(sy + k, 0, ssize.height)
    const int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
    const uint8_t* S0 = src + (size_t)sy0 * sstride;
    const uint8_t* S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; dx++) {
      const int sx = xofs[dx];
      const int sx1 = std::min(sx + 1, sw - 1);   // a1 == 0 whenever sx+1 would be out of range
      row0[dx] = S0[sx] * ialpha[dx * 2] + S0[sx1] * ialpha[dx * 2 + 1];
      row1[dx] = S1[sx] * ialpha[dx * 2] + S1[sx1] * ialpha[dx * 2 + 1];
    }
    const short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; dx++)
      D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---- cv::GaussianBlur(7x7, sigma 2) on CV_8U, OpenCV >= 4.1.1 fixed-point path --------------------
// kernel: getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED with 8 fractional bits
void gaussian_kernel7_fixed(int out[7]) {
  const int n = 7;
  const double sigma = 2.0;
  const double scale2X = -0.125 / (sigma * sigma);   // sd_minus_0_125 / (sigmaX*sigmaX), x stepped by 2
  double v[4], sum = 0;
  for (int i = 0, x = 1 - n; i < 3; i++, x += 2) { v[i] = std::exp((double)(x * x) * scale2X); sum += v[i]; }
  sum *= 2; sum += 1.0;
  const double mul1 = 1.0 / sum;
  double k[7];
  for (int i = 0; i < 3; i++) { k[i] = v[i] * mul1; k[6 - i] = k[i]; }
  k[3] = mul1;
  double err = 0; int64_t isum = 0;
  for (int i = 0; i < 3; i++) {
    const double adj = k[i] * 256.0 + err;
    const int64_t v0 = cvRoundD(adj);
    err = adj - (double)v0;
    out[i] = out[6 - i] = (int)v0;
    isum += v0;
  }
  out[3] = (int)(256 - 2 * isum);
}

inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
  return p;
}

void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  int kx[7];
  gaussian_kernel7_fixed(kx);
  std::vector<uint16_t> tmp((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t s = 0;
      for (int k = 0; k < 7; k++) s += (uint32_t)kx[k] * src[(size_t)y * sstride + reflect101(x + k - 3, w)];
      tmp[(size_t)y * w + x] = (uint16_t)s;   // 8.8 fixed point, sum of weights = 256 -> fits 16 bits
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t s = 0;
      for (int k = 0; k < 7; k++) s += (uint32_t)kx[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
      dst[(size_t)y * dstride + x] = (uint8_t)((s + (1u << 15)) >> 16);
    }
}

// ---- cv::FAST(roi, keypoints, threshold, true), TYPE_9_16 --------------------------------------------
const int kOff16[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                           {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]); a = std::min(a, (int)d[k + 5]); a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]); a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]); b = std::max(b, (int)d[k + 4]); b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]); b = std::max(b, (int)d[k + 7]); b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

// img: pointer to the ROI's (0,0) pixel; appends KeyPoint(j, i-1, 7, -1, score) in raster order
void fast9_16(const uint8_t* img, int cols, int rows, int step, int threshold, std::vector<KP>& out) {
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kOff16[k][0] + kOff16[k][1] * step;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  std::vector<uint8_t> bufm((size_t)cols * 3, 0);
  uint8_t* buf[3] = {bufm.data(), bufm.data() + cols, bufm.data() + 2 * cols};
  std::vector<int> cpm((size_t)(cols + 1) * 3, 0);
  int* cpbuf[3] = {cpm.data() + 1, cpm.data() + 1 + (cols + 1), cpm.data() + 1 + 2 * (cols + 1)};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * step + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        const int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          const int vt = v - threshold; int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x < vt) { if (++count > K) { cornerpos[ncorners++] = j; curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold); break; } }
            else count = 0;
          }
        }
        if (d & 2) {
          const int vt = v + threshold; int count = 0;
          for (int k = 0; k < N; k++) {
            const int x = ptr[pixel[k]];
            if (x > vt) { if (++count > K) { cornerpos[ncorners++] = j; curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold); break; } }
            else count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      const int j = cornerpos[k];
      const int score = prev[j];
      if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
          score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
        out.push_back(KP{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0});
    }
  }
}

// ---- cv::fastAtan2 (OpenCV 3.x/4.x scalar) ---------------------------------------------------------
float fast_atan2(float y, float x) {
  static const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

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
