// frame_math.h — per-frame geometry of cslam::Frame restated for host + device (cslam/src/Frame.cpp:139-330).
// cv::undistortPoints is [EXT] (OpenCV 4.2.0, imgproc/src/undistort.dispatch.cpp, cvUndistortPointsInternal): the call
// at Frame.cpp:298 / :323 passes K (CV_32F, widened to f64), D = (k1 k2 p1 p2 [k3]), R = empty, P = K and the default
// TermCriteria(MAX_ITER, 5, 0.01) => exactly five fixed-point iterations in f64, no epsilon test, result cast to f32.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "orb_math.h"

#define FR_HD __host__ __device__ __forceinline__

constexpr int kGridCols = 75, kGridRows = 48;   // FRAME_GRID_COLS / ROWS (cslam/include/cslam/Frame.h:51-52)

struct FrameCam {
  double fx, fy, cx, cy;      // mK (f32 values widened)
  double k1, k2, p1, p2, k3;  // mDistCoef
};

// one point of cv::undistortPoints(src, dst, K, D, noArray(), K)
FR_HD void frame_undistort_point(const FrameCam& c, float xin, float yin, float& xout, float& yout) {
  const double u = xin, v = yin;
  const double ifx = 1. / c.fx, ify = 1. / c.fy;
  double x = (u - c.cx) * ifx, y = (v - c.cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    // k[5..7] = 0 (rational model unused): numerator is exactly 1
    const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
    if (icdist < 0) { x = (u - c.cx) * ifx; y = (v - c.cy) * ify; break; }
    const double deltaX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
    const double deltaY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y + 0 * r2 + 0 * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  // RR = P * R = K: xx = fx x + 0 y + cx, ww = 1 / (0 x + 0 y + 1)
  const double xx = c.fx * x + 0 * y + c.cx, yy = 0 * x + c.fy * y + c.cy, ww = 1. / (0 * x + 0 * y + 1);
  xout = (float)(xx * ww);
  yout = (float)(yy * ww);
}

struct FrameBounds { float minX, minY, maxX, maxY, wInv, hInv; };

// Frame::PosInGrid (Frame.cpp:254-265): f32 arithmetic, round half away from zero
FR_HD int frame_cell_of(const FrameBounds& b, float x, float y) {
  const int posX = (int)roundf((x - b.minX) * b.wInv);
  const int posY = (int)roundf((y - b.minY) * b.hInv);
  if (posX < 0 || posX >= kGridCols || posY < 0 || posY >= kGridRows) return -1;
  return posX * kGridRows + posY;   // mGrid[x][y]: x-major
}

// cell range of Frame::GetFeaturesInArea (Frame.cpp:205-219); returns false when the window misses the grid
FR_HD bool frame_cell_range(const FrameBounds& b, float x, float y, float r, int& x0, int& x1, int& y0, int& y1) {
  x0 = max(0, (int)floorf((x - b.minX - r) * b.wInv));
  if (x0 >= kGridCols) return false;
  x1 = min(kGridCols - 1, (int)ceilf((x - b.minX + r) * b.wInv));
  if (x1 < 0) return false;
  y0 = max(0, (int)floorf((y - b.minY - r) * b.hInv));
  if (y0 >= kGridRows) return false;
  y1 = min(kGridRows - 1, (int)ceilf((y - b.minY + r) * b.hInv));
  if (y1 < 0) return false;
  return true;
}

// Frame::isInFrustum (Frame.cpp:139-198) for one map point, [EXT] cv::Mat arithmetic restated:
//  * mRcw*P+mtcw is one cv::gemm(A 3x3, B 3x1, alpha 1, C, beta 1): OpenCV's small-matrix path (core/src/matmul.simd.hpp,
//    case len == 3) forms t = a0*b0 + a1*b1 + a2*b2 in f32, left to right, then d = (float)(t*alpha + c*beta) in f64;
//    the baseline (non-FMA) build is restated — an AVX2-dispatched OpenCV may contract the f32 sum (parity unpinned);
//  * P - mOw in f32; cv::norm (NORM_L2, CV_32F) accumulates v*v in f64 and returns sqrt in f64, stored to a float;
//  * Mat::dot accumulates in f64; viewCos = (float)(dot / (double)dist);
//  * PredictScale: ceil(logf(mfMaxDistance / dist) / mfLogScaleFactor) clamped to [0, nLevels) (MapPoint.cpp:854-869).
struct FrustumFrame {
  float Rcw[9], tcw[3], Ow[3];
  float fx, fy, cx, cy;
  float minX, maxX, minY, maxY;
  float logScaleFactor;
  int nScaleLevels;
};

FR_HD bool frame_in_frustum(const FrustumFrame& f, const float P[3], const float Pn[3], float mfMinDistance, float mfMaxDistance,
                            float viewingCosLimit, float& u_out, float& v_out, int& level_out, float& cos_out) {
  float Pc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float t = f.Rcw[3 * i] * P[0] + f.Rcw[3 * i + 1] * P[1] + f.Rcw[3 * i + 2] * P[2];
    Pc[i] = (float)((double)t * 1.0 + (double)f.tcw[i] * 1.0);
  }
  if (Pc[2] < 0.0f) return false;
  const float invz = 1.0f / Pc[2];
  const float u = f.fx * Pc[0] * invz + f.cx;
  const float v = f.fy * Pc[1] * invz + f.cy;
  if (u < f.minX || u > f.maxX) return false;
  if (v < f.minY || v > f.maxY) return false;
  const float maxDistance = 1.2f * mfMaxDistance, minDistance = 0.8f * mfMinDistance;   // MapPoint.cpp:825-835
  const float PO[3] = {P[0] - f.Ow[0], P[1] - f.Ow[1], P[2] - f.Ow[2]};
  double s2 = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) s2 += (double)PO[i] * PO[i];
  const float dist = (float)sqrt(s2);
  if (dist < minDistance || dist > maxDistance) return false;
  double dot = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) dot += (double)PO[i] * Pn[i];
  const float viewCos = (float)(dot / dist);
  if (viewCos < viewingCosLimit) return false;
  const float ratio = mfMaxDistance / dist;
  int nScale = (int)ceilf(orbm::logf_glibc(ratio) / f.logScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= f.nScaleLevels) nScale = f.nScaleLevels - 1;
  u_out = u; v_out = v; level_out = nScale; cos_out = viewCos;
  return true;
}
