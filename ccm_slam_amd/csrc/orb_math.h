// orb_math.h — the scalar float math of the ORB path, written once for host and device so the
// GPU kernels produce the exact bits the reference's CPU build produces.
//
// What the reference calls (cslam/src/ORBextractor.cpp):
//   :94   fastAtan2((float)m_01,(float)m_10)          -> OpenCV core/mathfuncs_core (3.x / 4.x scalar path)
//   :105  (float)cos(angle), (float)sin(angle)         -> glibc cosf / sinf (float overloads of <cmath>)
//   :111  cvRound(x*b + y*a)                           -> round-half-to-even of an f32 expression, no FMA
//                                                         (cslam is built -O3 without -march, CMakeLists.txt:33)
// glibc (pinned: 2.35, the version in this image; the algorithm is unchanged since 2.28) implements
// sinf/cosf as Szabolcs Nagy's double-precision polynomial scheme (sysdeps/ieee754/flt-32/
// s_sincosf.h, s_sinf.c, s_cosf.c; public in ARM optimized-routines, MIT).  It is NOT correctly
// rounded (1.3 % of inputs differ by 1 ulp from the correctly rounded value), so bit parity needs
// the same scheme: restated below, constants from the published table.  tests/test_orb_math.py
// checks it bit-for-bit against this container's libm on 2^24-spaced and random inputs.
// Only |x| < 120 is supported (angles are in [0, 2*pi]); callers never exceed that.
#pragma once
#include <stdint.h>
#include <math.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define ORB_HD __host__ __device__ __forceinline__
#else
#define ORB_HD static inline
#endif

namespace orbm {

ORB_HD uint32_t as_u32(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
ORB_HD uint32_t abstop12(float x) { return (as_u32(x) >> 20) & 0x7ff; }

struct SinCosTab { double c0, c1, c2, c3, c4, s1, s2, s3; };

// polynomial of s_sincosf.h:sinf_poly; neg selects the second table entry (computes -cos)
ORB_HD float sinf_poly(double x, double x2, bool neg, int n) {
  const double c0 = neg ? -0x1p0 : 0x1p0;
  const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
  const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
  const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
  const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    const double x3 = x * x2;
    const double t1 = s2 + x2 * s3;
    const double x7 = x3 * x2;
    const double s = x + x3 * s1;
    return (float)(s + x7 * t1);
  } else {
    const double x4 = x2 * x2;
    const double t2 = c3 + x2 * c4;
    const double t1 = c0 + x2 * c1;
    const double x6 = x4 * x2;
    const double c = t1 + x4 * c2;
    return (float)(c + x6 * t2);
  }
}

// reduce_fast: quadrant in bits 24..31 of x * (2/pi * 2^24)
ORB_HD double reduce_fast(double x, int* np) {
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
  const double r = x * hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * hpi;
}

ORB_HD float sinf_glibc(float y) {
  double x = y;
  const float pio4 = 0x1.921FB6p-1f;
  if (abstop12(y) < abstop12(pio4)) {
    const double s = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return sinf_poly(x, s, false, 0);
  }
  int n;
  x = reduce_fast(x, &n);
  const double sign = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;   // sign[] = {1,-1,-1,1}
  return sinf_poly(x * sign, x * x, (n & 2) != 0, n);
}

ORB_HD float cosf_glibc(float y) {
  double x = y;
  const float pio4 = 0x1.921FB6p-1f;
  if (abstop12(y) < abstop12(pio4)) {
    const double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return sinf_poly(x, x2, false, 1);
  }
  int n;
  x = reduce_fast(x, &n);
  const double sign = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;
  return sinf_poly(x * sign, x * x, (n & 2) != 0, n ^ 1);
}

// cv::fastAtan2 scalar path of OpenCV 3.x / 4.x (core/src/mathfuncs_core.simd.hpp): degrees in [0,360)
ORB_HD float fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;   // (float)DBL_EPSILON
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    // f32 division, correctly rounded: done in f64 (53 >= 2*24+2 bits makes the double rounding exact)
    c = (float)((double)ay / (double)(ax + eps));
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = (float)((double)ax / (double)(ay + eps));
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// cvRound(float): round half to even (lrintf / cvtss2si in the default rounding mode)
ORB_HD int cv_round(float v) {
#ifdef __HIP_DEVICE_COMPILE__
  return __float2int_rn(v);
#else
  return (int)lrintf(v);
#endif
}


// glibc logf (sysdeps/ieee754/flt-32/e_logf.c, Szabolcs Nagy's scheme; table and polynomial read from this image's
// libm.so.6 and identical to the published __logf_data): log(x) = log1p(z/c - 1) + log(c) + k ln2 in f64, NOT correctly
// rounded, so MapPoint::PredictScale's ceil(log(ratio)/mfLogScaleFactor) (MapPoint.cpp:845,862) needs the same scheme.
// Positive normal finite x only (ratio = mfMaxDistance / dist > 0); the special-case prologue of the libm is omitted.
ORB_HD float logf_glibc(float x) {
  const double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
      {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
      {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
      {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
      {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  const uint32_t ix = as_u32(x);
  if (ix == 0x3f800000u) return 0.0f;
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> (23 - 4)) % 16);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  union { uint32_t u; float f; } zf; zf.u = iz;
  const double z = (double)zf.f;
  const double invc = T[i][0], logc = T[i][1];
  const double r = z * invc - 1;
  const double y0 = logc + (double)k * Ln2;
  const double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

}  // namespace orbm
