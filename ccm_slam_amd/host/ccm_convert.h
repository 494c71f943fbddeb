// ccm_convert.h — the f32 <-> f64 boundary of every optimiser call, host side of the product.
// The map stores poses and points as CV_32F cv::Mat; g2o — and the device solver — work in f64 on (unit quaternion, translation).
// cslam::Converter (cslam/src/Converter.cc:40-119) crosses that boundary with Eigen; these functions do the same arithmetic on plain
// arrays so that a drop-in shim needs neither Eigen nor g2o:
//   toSE3Quat(Tcw f32 4x4)   Converter::toSE3Quat (:40-50): R, t widened to f64, Eigen::Quaterniond(R) [the branchy trace method of
//                            Eigen/src/Geometry/Quaternion.h], then g2o::SE3Quat(R, t)'s normalizeRotation (se3quat.h:58-60, 280-285:
//                            w >= 0, unit norm)
//   toCvMat(qt)              Converter::toCvMat(SE3Quat) (:52-56, 74-82): to_homogeneous_matrix = Quaterniond::toRotationMatrix
//                            (se3quat.h:271-277), every entry rounded to f32
//   toVector3d / toCvMat(p)  (:113-119, 94-101): f32 -> f64 widening / f64 -> f32 rounding of a map point
// Pinned against the reference's own Converter.cc (compiled verbatim into oracle/_ref/liboptimizer_ref.so): tests/test_ref_optimizer.py.
#pragma once
#include <cmath>

namespace ccmh {

// cv::Mat 4x4 CV_32F, row-major -> [qx qy qz qw tx ty tz]
inline void toSE3Quat(const float Tcw[16], double qt[7]) {
  double m[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = (double)Tcw[4 * i + j];
  double q[4];   // x y z w
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t;
    q[1] = (m[0][2] - m[2][0]) * t;
    q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
  if (q[3] < 0) for (int c = 0; c < 4; c++) q[c] *= -1;                                        // normalizeRotation
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n > 0) for (int c = 0; c < 4; c++) q[c] /= n;
  qt[0] = q[0]; qt[1] = q[1]; qt[2] = q[2]; qt[3] = q[3];
  qt[4] = (double)Tcw[3]; qt[5] = (double)Tcw[7]; qt[6] = (double)Tcw[11];
}

// [qx qy qz qw] -> 3x3 row-major f64 (Eigen::Quaterniond::toRotationMatrix)
inline void quatToRotation(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// [qx qy qz qw tx ty tz] -> cv::Mat 4x4 CV_32F, row-major
inline void toCvMat(const double qt[7], float Tcw[16]) {
  double R[9];
  quatToRotation(qt, R);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tcw[4 * i + j] = (float)R[3 * i + j]; Tcw[4 * i + 3] = (float)qt[4 + i]; }
  Tcw[12] = 0.f; Tcw[13] = 0.f; Tcw[14] = 0.f; Tcw[15] = 1.f;
}

inline void toVector3d(const float p[3], double out[3]) { out[0] = (double)p[0]; out[1] = (double)p[1]; out[2] = (double)p[2]; }
inline void toCvMat3(const double p[3], float out[3]) { out[0] = (float)p[0]; out[1] = (float)p[1]; out[2] = (float)p[2]; }

// Sim3 [qx qy qz qw tx ty tz s] -> SE3 pose [R | t / s] as cv::Mat 4x4 CV_32F: the essential-graph write-back (Optimizer.cpp:1272-1281)
inline void sim3ToCvSE3(const double s8[8], float Tcw[16]) {
  double R[9];
  quatToRotation(s8, R);
  const double inv = 1. / s8[7];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tcw[4 * i + j] = (float)R[3 * i + j]; Tcw[4 * i + 3] = (float)(s8[4 + i] * inv); }
  Tcw[12] = 0.f; Tcw[13] = 0.f; Tcw[14] = 0.f; Tcw[15] = 1.f;
}

}  // namespace ccmh
