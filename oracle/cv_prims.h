// cv_prims.h — the OpenCV 4.2.0 primitives that cslam::ORBextractor calls, RESTATED (TEST INFRASTRUCTURE).
// OpenCV is not under /root/reference and is not installed in this image, so these [EXT] primitives exist in exactly one place:
// here.  Both users include this header:
//   oracle/orb_ref.cpp                  the oracle restatement of ORBextractor.cpp
//   oracle/ref_shim/opencv2/*.hpp       the look-alike cv:: API against which the REFERENCE'S OWN ORBextractor.cpp is compiled
//                                       verbatim into oracle/_ref/liborb_ref.so (oracle/Makefile.ref)
// so "oracle == _ref" proves the reference's own code (pyramid sizing, cell loop, threshold fallback, octree, orientation,
// descriptor taps, ordering, scaling) was restated faithfully, while these primitives stay pinned only by the hand-derived KATs of
// tests/test_oracle_orb.py (parity vs a real OpenCV build: unpinned).
//   cv::resize INTER_LINEAR 8UC1      imgproc/resize.cpp   (11-bit fixed-point coefficients, HResizeLinear/VResizeLinear)
//   cv::FAST(img,kps,t,true) 9/16     features2d/fast.cpp, fast_score.cpp (FAST_t<16>, cornerScore<16>)
//   cv::GaussianBlur 7x7 sigma 2 8U   imgproc/smooth.dispatch.cpp + fixedpoint (ufixedpoint16 bit-exact path)
//   cv::fastAtan2                     core/mathfuncs_core.simd.hpp scalar polynomial
//   cvRound                           round half to even
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace cvprims {

struct KP { float x, y, size, angle, response; int32_t octave; };

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }

// ---- cv::resize, INTER_LINEAR, 8UC1 (fixed point, INTER_RESIZE_COEF_BITS = 11) -------------------
inline void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
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
inline void gaussian_kernel7_fixed(int out[7]) {
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

inline void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
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

inline int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
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
inline void fast9_16(const uint8_t* img, int cols, int rows, int step, int threshold, std::vector<KP>& out) {
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
inline float fast_atan2(float y, float x) {
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

}  // namespace cvprims
