// ba_math.h — f64 device/host math shared by the BA and pose-optimisation kernels.
// Follows the arithmetic of vendored g2o (paths under cslam/thirdparty/g2o/g2o/):
//   SE3Quat::map / exp / operator* / normalizeRotation   types/se3quat.h:104-110,217-285
//   EdgeSE3ProjectXYZ::linearizeOplus / cam_project       types/types_six_dof_expmap.cpp:103-147
//   RobustKernelHuber::robustify                          core/robust_kernel_impl.cpp:78-90
// Eigen's Quaterniond(R), q*v and toRotationMatrix are restated from their published algorithms.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define BA_HD __host__ __device__ __forceinline__

struct BaPose { double qx, qy, qz, qw, tx, ty, tz; };

BA_HD BaPose ba_load_pose(const double* p) { return BaPose{p[0], p[1], p[2], p[3], p[4], p[5], p[6]}; }
BA_HD void ba_store_pose(double* p, const BaPose& T) { p[0] = T.qx; p[1] = T.qy; p[2] = T.qz; p[3] = T.qw; p[4] = T.tx; p[5] = T.ty; p[6] = T.tz; }

BA_HD void ba_normalize_rotation(BaPose& T) {
  if (T.qw < 0) { T.qx = -T.qx; T.qy = -T.qy; T.qz = -T.qz; T.qw = -T.qw; }
  // one reciprocal instead of four divisions (f64 division is ~10 dependent instructions on gfx950); differs from
  // Eigen's coeffs()/norm() by at most 1 ulp per component, far inside the stated pose tolerance
  const double inv = 1.0 / sqrt(T.qx * T.qx + T.qy * T.qy + T.qz * T.qz + T.qw * T.qw);
  T.qx *= inv; T.qy *= inv; T.qz *= inv; T.qw *= inv;
}

// v + w*uv + u x uv with uv = 2 (u x v)
BA_HD void ba_qrot(double qx, double qy, double qz, double qw, const double v[3], double out[3]) {
  double uv0 = qy * v[2] - qz * v[1], uv1 = qz * v[0] - qx * v[2], uv2 = qx * v[1] - qy * v[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  out[0] = v[0] + qw * uv0 + (qy * uv2 - qz * uv1);
  out[1] = v[1] + qw * uv1 + (qz * uv0 - qx * uv2);
  out[2] = v[2] + qw * uv2 + (qx * uv1 - qy * uv0);
}

BA_HD void ba_map(const BaPose& T, const double X[3], double out[3]) {
  ba_qrot(T.qx, T.qy, T.qz, T.qw, X, out);
  out[0] += T.tx; out[1] += T.ty; out[2] += T.tz;
}

BA_HD void ba_q_to_R(const BaPose& T, double R[9]) {
  const double tx = 2 * T.qx, ty = 2 * T.qy, tz = 2 * T.qz;
  const double twx = tx * T.qw, twy = ty * T.qw, twz = tz * T.qw;
  const double txx = tx * T.qx, txy = ty * T.qx, txz = tz * T.qx;
  const double tyy = ty * T.qy, tyz = tz * T.qy, tzz = tz * T.qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

BA_HD void ba_R_to_q(const double m[9], BaPose& T) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    T.qw = 0.5 * t;
    t = 0.5 / t;
    T.qx = (m[7] - m[5]) * t; T.qy = (m[2] - m[6]) * t; T.qz = (m[3] - m[1]) * t;
  } else {
    // Eigen picks the largest diagonal entry i and sets j = (i+1)%3, k = (j+1)%3; written out per case so that m[] is never
    // indexed with a run-time value (that would move the matrix to scratch memory on the device)
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > ((i == 0) ? m[0] : m[4])) i = 2;
    if (i == 0) {          // j = 1, k = 2
      t = sqrt(m[0] - m[4] - m[8] + 1.0);
      T.qx = 0.5 * t; t = 0.5 / t;
      T.qw = (m[7] - m[5]) * t; T.qy = (m[3] + m[1]) * t; T.qz = (m[6] + m[2]) * t;
    } else if (i == 1) {   // j = 2, k = 0
      t = sqrt(m[4] - m[8] - m[0] + 1.0);
      T.qy = 0.5 * t; t = 0.5 / t;
      T.qw = (m[2] - m[6]) * t; T.qz = (m[7] + m[5]) * t; T.qx = (m[1] + m[3]) * t;
    } else {               // j = 0, k = 1
      t = sqrt(m[8] - m[0] - m[4] + 1.0);
      T.qz = 0.5 * t; t = 0.5 / t;
      T.qw = (m[3] - m[1]) * t; T.qx = (m[2] + m[6]) * t; T.qy = (m[5] + m[7]) * t;
    }
  }
}

// T <- exp([omega, upsilon]) * T     (VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-76)
BA_HD BaPose ba_oplus(const double u[6], const BaPose& T) {
  const double o0 = u[0], o1 = u[1], o2 = u[2];
  const double theta = sqrt(o0 * o0 + o1 * o1 + o2 * o2);
  const double Om[9] = {0, -o2, o1, o2, 0, -o0, -o1, o0, 0};
  double Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += Om[i * 3 + k] * Om[k * 3 + j];
      Om2[i * 3 + j] = s;
    }
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    double st, ct;
    sincos(theta, &st, &ct);
    const double it = 1.0 / theta, it2 = it * it;
    const double a = st * it, b = (1 - ct) * it2, c = (theta - st) * it2 * it;
    for (int i = 0; i < 9; i++) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  BaPose E;
  ba_R_to_q(R, E);
  E.tx = V[0] * u[3] + V[1] * u[4] + V[2] * u[5];
  E.ty = V[3] * u[3] + V[4] * u[4] + V[5] * u[5];
  E.tz = V[6] * u[3] + V[7] * u[4] + V[8] * u[5];
  ba_normalize_rotation(E);
  // E * T
  BaPose r;
  const double tv[3] = {T.tx, T.ty, T.tz};
  double rt[3];
  ba_qrot(E.qx, E.qy, E.qz, E.qw, tv, rt);
  r.tx = E.tx + rt[0]; r.ty = E.ty + rt[1]; r.tz = E.tz + rt[2];
  r.qw = E.qw * T.qw - E.qx * T.qx - E.qy * T.qy - E.qz * T.qz;
  r.qx = E.qw * T.qx + E.qx * T.qw + E.qy * T.qz - E.qz * T.qy;
  r.qy = E.qw * T.qy + E.qy * T.qw + E.qz * T.qx - E.qx * T.qz;
  r.qz = E.qw * T.qz + E.qz * T.qw + E.qx * T.qy - E.qy * T.qx;
  ba_normalize_rotation(r);
  return r;
}

// The same update for the single-workgroup LM kernels, where it sits on the serial path of every trial (pose optimisation: ~45 trials per call, every lane executing it
// redundantly): ba_oplus is ~380 mostly dependent f64 instructions (sincos, four square roots, four divisions, a rotation matrix turned into a quaternion) ~ 1 us.
// For 1e-5 <= theta < 0.5 — every LM step that matters — the rotation part of exp is written directly as the unit quaternion (omega sin(theta/2)/theta, cos(theta/2)) and
// the coefficients sin(theta/2)/theta, cos(theta/2), (1 - cos theta)/theta^2, (theta - sin theta)/theta^3 are even power series in theta^2 (truncation < 1e-17 relative
// at theta = 0.5; three independent Horner chains), so no sincos, no matrix -> quaternion conversion and one normalisation are left.  Same rotation as ba_oplus to 1 ulp; the translation
// differs by up to 1e-11 relative, because the closed forms (1 - cos t)/t^2 and (t - sin t)/t^3 cancel for small angles and the series do not (tests/test_ba_math_host.py builds
// both for the host and compares them over 2 000 000 updates); outside the range (g2o's small-angle branch R = I + Om + Om^2 below 1e-5 included) it IS ba_oplus.
BA_HD BaPose ba_oplus_fast(const double u[6], const BaPose& T) {
  const double o0 = u[0], o1 = u[1], o2 = u[2];
  const double x = o0 * o0 + o1 * o1 + o2 * o2;          // theta^2
  if (!(x >= 1e-10 && x < 0.25)) return ba_oplus(u, T);
  const double y = 0.25 * x;                               // (theta / 2)^2
  // sin(h)/h = sum (-1)^k y^k / (2k+1)!,  cos(h) = sum (-1)^k y^k / (2k)!   (h = theta / 2, y <= 0.0625)
  // (Horner steps as explicit fused multiply-adds: four dependent chains of 6 - 7 steps on the serial path)
  const double sh = fma(y, fma(y, fma(y, fma(y, fma(y, fma(y, 1.0 / 6227020800.0, -1.0 / 39916800), 1.0 / 362880), -1.0 / 5040), 1.0 / 120), -1.0 / 6), 1.0);
  const double cw = fma(y, fma(y, fma(y, fma(y, fma(y, fma(y, fma(y, -1.0 / 87178291200.0, 1.0 / 479001600.0), -1.0 / 3628800), 1.0 / 40320), -1.0 / 720), 1.0 / 24), -1.0 / 2), 1.0);
  // b = (1 - cos theta)/theta^2 = sum (-1)^k x^k / (2k+2)!,  c = (theta - sin theta)/theta^3 = sum (-1)^k x^k / (2k+3)!   (x < 0.25)
  const double b = fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, -1.0 / 20922789888000.0, 1.0 / 87178291200.0), -1.0 / 479001600.0), 1.0 / 3628800), -1.0 / 40320), 1.0 / 720), -1.0 / 24), 1.0 / 2);
  const double c = fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, fma(x, -1.0 / 355687428096000.0, 1.0 / 1307674368000.0), -1.0 / 6227020800.0), 1.0 / 39916800), -1.0 / 362880), 1.0 / 5040), -1.0 / 120), 1.0 / 6);
  const double hs = 0.5 * sh;                              // sin(theta/2) / theta
  BaPose E;
  E.qx = hs * o0; E.qy = hs * o1; E.qz = hs * o2; E.qw = cw;   // unit up to rounding; cw > 0.96
  // V upsilon = upsilon + b (omega x upsilon) + c (omega x (omega x upsilon))
  const double w0 = o1 * u[5] - o2 * u[4], w1 = o2 * u[3] - o0 * u[5], w2 = o0 * u[4] - o1 * u[3];
  const double z0 = o1 * w2 - o2 * w1, z1 = o2 * w0 - o0 * w2, z2 = o0 * w1 - o1 * w0;
  E.tx = u[3] + b * w0 + c * z0; E.ty = u[4] + b * w1 + c * z1; E.tz = u[5] + b * w2 + c * z2;
  BaPose r;
  const double tv[3] = {T.tx, T.ty, T.tz};
  double rt[3];
  ba_qrot(E.qx, E.qy, E.qz, E.qw, tv, rt);
  r.tx = E.tx + rt[0]; r.ty = E.ty + rt[1]; r.tz = E.tz + rt[2];
  r.qw = E.qw * T.qw - E.qx * T.qx - E.qy * T.qy - E.qz * T.qz;
  r.qx = E.qw * T.qx + E.qx * T.qw + E.qy * T.qz - E.qz * T.qy;
  r.qy = E.qw * T.qy + E.qy * T.qw + E.qz * T.qx - E.qx * T.qz;
  r.qz = E.qw * T.qz + E.qz * T.qw + E.qx * T.qy - E.qy * T.qx;
  ba_normalize_rotation(r);
  return r;
}

// Huber: rho0 (robustified chi2) and rho1 (weight)
BA_HD void ba_huber(double e2, double delta, double& rho0, double& rho1) {
  // the reference's g2o fork keeps delta^2 in a FLOAT member (robust_kernel_impl.h:84, set in RobustKernelHuber::setDelta, .cpp:65-69):
  // the f32-rounded square is the inlier threshold and the constant of the outlier branch
  const double dsqr = (double)(float)(delta * delta);
  if (delta <= 0 || e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
  else { const double s = sqrt(e2); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// residual of one observation: e = obs - K * proj(T*X); returns camera-frame depth
BA_HD double ba_residual(const BaPose& T, const double K[4], const double X[3], double ox, double oy, double& e0, double& e1) {
  double Xc[3];
  ba_map(T, X, Xc);
  const double invz = 1.0 / Xc[2];
  e0 = ox - (Xc[0] * invz * K[0] + K[2]);
  e1 = oy - (Xc[1] * invz * K[1] + K[3]);
  return Xc[2];
}

// Jacobians: Ji (2x3, d e/d point) and Jj (2x6, d e/d pose [rot, trans]) from the camera-frame point (x, y, 1 / z), the focal lengths and the rotation
// matrix.  The row Schur kernel re-derives the Hpl block of a PARTNER observation from these four numbers + the partner camera's 11 instead of reading
// its 18 stored values (ba.hip: ba_schur_row3), with the very expressions ba_jacobians evaluates.
BA_HD void ba_jac_from_xc(double x, double y, double iz, double fx, double fy, const double R[9], double Ji[6], double Jj[12]) {
  const double iz2 = iz * iz;
  const double tmp[6] = {fx, 0, -x * iz * fx, 0, fy, -y * iz * fy};
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int q = 0; q < 3; q++) s += tmp[r * 3 + q] * R[q * 3 + c];
      Ji[r * 3 + c] = -iz * s;
    }
  Jj[0] = x * y * iz2 * fx; Jj[1] = -(1 + (x * x * iz2)) * fx; Jj[2] = y * iz * fx; Jj[3] = -iz * fx; Jj[4] = 0; Jj[5] = x * iz2 * fx;
  Jj[6] = (1 + y * y * iz2) * fy; Jj[7] = -x * y * iz2 * fy; Jj[8] = -x * iz * fy; Jj[9] = 0; Jj[10] = -iz * fy; Jj[11] = y * iz2 * fy;
}
BA_HD void ba_jacobians(const BaPose& T, const double K[4], const double X[3], double Ji[6], double Jj[12]) {
  double Xc[3];
  ba_map(T, X, Xc);
  const double iz = 1.0 / Xc[2];   // one reciprocal for the ~12 divisions by z / z^2 of the reference formulas
  double R[9];
  ba_q_to_R(T, R);
  ba_jac_from_xc(Xc[0], Xc[1], iz, K[0], K[1], R, Ji, Jj);
}

// pose-only Jacobian (EdgeSE3ProjectXYZOnlyPose::linearizeOplus, types_six_dof_expmap.cpp:266-288)
BA_HD void ba_jacobian_pose_only(const double Xc[3], const double K[4], double J[12]) {
  const double x = Xc[0], y = Xc[1], invz = 1.0 / Xc[2], invz_2 = invz * invz, fx = K[0], fy = K[1];
  J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
  J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
}

// symmetric 3x3 stored as {a00,a01,a02,a11,a12,a22}; inverse by cofactors (same layout)
BA_HD void ba_sym3_inv(const double a[6], double r[6]) {
  const double c00 = a[3] * a[5] - a[4] * a[4];
  const double c01 = a[4] * a[2] - a[1] * a[5];
  const double c02 = a[1] * a[4] - a[3] * a[2];
  const double det = c00 * a[0] + c01 * a[1] + c02 * a[2];
  const double id = 1.0 / det;
  r[0] = c00 * id; r[1] = c01 * id; r[2] = c02 * id;
  r[3] = (a[0] * a[5] - a[2] * a[2]) * id;
  r[4] = (a[2] * a[1] - a[0] * a[4]) * id;
  r[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}

// 6x6 SPD inverse via Cholesky (row-major full storage); returns false if not positive definite
BA_HD bool ba_spd6_inv(const double A[36], double Inv[36]) {
  double L[36];
  for (int i = 0; i < 36; i++) L[i] = 0;
  for (int j = 0; j < 6; j++) {
    double d = A[j * 6 + j];
    for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k];
    if (!(d > 0.0)) return false;
    d = sqrt(d);
    L[j * 6 + j] = d;
    for (int i = j + 1; i < 6; i++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = s / d;
    }
  }
  // invert L (lower) -> Li
  double Li[36];
  for (int i = 0; i < 36; i++) Li[i] = 0;
  for (int c = 0; c < 6; c++) {
    Li[c * 6 + c] = 1.0 / L[c * 6 + c];
    for (int r = c + 1; r < 6; r++) {
      double s = 0;
      for (int k = c; k < r; k++) s -= L[r * 6 + k] * Li[k * 6 + c];
      Li[r * 6 + c] = s / L[r * 6 + r];
    }
  }
  // Inv = Li^T Li
  for (int r = 0; r < 6; r++)
    for (int c = 0; c <= r; c++) {
      double s = 0;
      for (int k = r; k < 6; k++) s += Li[k * 6 + r] * Li[k * 6 + c];
      Inv[r * 6 + c] = s; Inv[c * 6 + r] = s;
    }
  return true;
}
