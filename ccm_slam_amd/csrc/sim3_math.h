// sim3_math.h — g2o::Sim3 (thirdparty/g2o/g2o/types/sim3.h) in f64, host + device.
// Same operation order as the flat restatement in oracle/ba_ref.cpp so the two agree to rounding; the
// quaternion is deliberately NOT re-normalised by the product or the exponential map (upstream does not).
#pragma once
#include "ba_math.h"

struct Sim3d { double qx, qy, qz, qw, tx, ty, tz, s; };

BA_HD Sim3d sim3_load(const double* p) { return Sim3d{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]}; }
BA_HD void sim3_store(double* p, const Sim3d& S) { p[0] = S.qx; p[1] = S.qy; p[2] = S.qz; p[3] = S.qw; p[4] = S.tx; p[5] = S.ty; p[6] = S.tz; p[7] = S.s; }

// Sim3(const Vector7d&) (sim3.h:72-140); u = [omega(3), upsilon(3), sigma]
BA_HD Sim3d sim3_exp(const double u[7]) {
  const double sigma = u[6];
  const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  const double Om[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
  double Om2[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double a = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) a += Om[i * 3 + k] * Om[k * 3 + j];
      Om2[i * 3 + j] = a;
    }
  Sim3d S;
  S.s = exp(sigma);
  const double eps = 0.00001;
  double A, B, C;
  double R[9];
  if (theta < eps) {
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i];
  } else {
    const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
  }
  if (fabs(sigma) < eps) {
    C = 1;
    if (theta < eps) { A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta2 = theta * theta;
      A = (1 - cos(theta)) / theta2;
      B = (theta - sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (theta < eps) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / sigma2;
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double a = S.s * sin(theta), b = S.s * cos(theta);
      const double theta2 = theta * theta, sigma2 = sigma * sigma;
      const double c = theta2 + sigma2;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  BaPose q;
  ba_R_to_q(R, q);
  S.qx = q.qx; S.qy = q.qy; S.qz = q.qz; S.qw = q.qw;
  double t[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double W = A * Om[i * 3 + k] + B * Om2[i * 3 + k] + C * ((i == k) ? 1.0 : 0.0);
      acc += W * u[3 + k];
    }
    t[i] = acc;
  }
  S.tx = t[0]; S.ty = t[1]; S.tz = t[2];
  return S;
}

// Sim3::operator* (sim3.h:272-278)
BA_HD Sim3d sim3_mul(const Sim3d& a, const Sim3d& b) {
  Sim3d r;
  r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
  r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
  r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
  r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
  const double bt[3] = {b.tx, b.ty, b.tz};
  double rt[3];
  ba_qrot(a.qx, a.qy, a.qz, a.qw, bt, rt);
  r.tx = a.s * rt[0] + a.tx; r.ty = a.s * rt[1] + a.ty; r.tz = a.s * rt[2] + a.tz;
  r.s = a.s * b.s;
  return r;
}

// Sim3::inverse (sim3.h:240-243)
BA_HD Sim3d sim3_inv(const Sim3d& a) {
  Sim3d r;
  r.qx = -a.qx; r.qy = -a.qy; r.qz = -a.qz; r.qw = a.qw;
  const double k = -1. / a.s;
  const double v[3] = {k * a.tx, k * a.ty, k * a.tz};
  double t[3];
  ba_qrot(r.qx, r.qy, r.qz, r.qw, v, t);
  r.tx = t[0]; r.ty = t[1]; r.tz = t[2];
  r.s = 1. / a.s;
  return r;
}

// Sim3::map (sim3.h:142-144)
BA_HD void sim3_map(const Sim3d& S, const double X[3], double out[3]) {
  double r[3];
  ba_qrot(S.qx, S.qy, S.qz, S.qw, X, r);
  out[0] = S.s * r[0] + S.tx; out[1] = S.s * r[1] + S.ty; out[2] = S.s * r[2] + S.tz;
}

// Eigen::PartialPivLU<Matrix3d>(W).solve(t) as used by Sim3::log (sim3.h:216): unblocked LU with row pivoting on the
// largest |entry| of the column, unit-lower forward and upper back substitution.  Register-only (no dynamic indexing).
BA_HD void sim3_lu3_solve(const double W[9], const double t[3], double x[3]) {
  double a00 = W[0], a01 = W[1], a02 = W[2], a10 = W[3], a11 = W[4], a12 = W[5], a20 = W[6], a21 = W[7], a22 = W[8];
  double y0 = t[0], y1 = t[1], y2 = t[2];
#define SIM3_SWAP(p, q) { const double tmp_ = p; p = q; q = tmp_; }
  // column 0: rows 0..2 (first maximum wins, like Eigen's maxCoeff)
  {
    int best = 0; double bv = fabs(a00);
    if (fabs(a10) > bv) { bv = fabs(a10); best = 1; }
    if (fabs(a20) > bv) { best = 2; }
    if (best == 1) { SIM3_SWAP(a00, a10) SIM3_SWAP(a01, a11) SIM3_SWAP(a02, a12) SIM3_SWAP(y0, y1) }
    else if (best == 2) { SIM3_SWAP(a00, a20) SIM3_SWAP(a01, a21) SIM3_SWAP(a02, a22) SIM3_SWAP(y0, y2) }
  }
  a10 /= a00; a20 /= a00;
  a11 -= a10 * a01; a12 -= a10 * a02;
  a21 -= a20 * a01; a22 -= a20 * a02;
  // column 1: rows 1..2
  if (fabs(a21) > fabs(a11)) { SIM3_SWAP(a10, a20) SIM3_SWAP(a11, a21) SIM3_SWAP(a12, a22) SIM3_SWAP(y1, y2) }
#undef SIM3_SWAP
  a21 /= a11;
  a22 -= a21 * a12;
  y1 -= a10 * y0;
  y2 -= a20 * y0; y2 -= a21 * y1;
  x[2] = y2 / a22;
  x[1] = (y1 - a12 * x[2]) / a11;
  x[0] = (y0 - a01 * x[1] - a02 * x[2]) / a00;
}

// Sim3::log (sim3.h:146-237) -> [omega(3), upsilon(3), sigma]
BA_HD void sim3_log(const Sim3d& S, double res[7]) {
  const double sigma = log(S.s);
  BaPose q{S.qx, S.qy, S.qz, S.qw, 0, 0, 0};
  double R[9];
  ba_q_to_R(q, R);
  const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double omega[3];
  const double eps = 0.00001;
  double A, B, C;
  const bool small = d > 1 - eps;
  double theta = 0;
  if (small) {
#pragma unroll
    for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
  } else {
    theta = acos(d);
    const double k = theta / (2 * sqrt(1 - d * d));
#pragma unroll
    for (int i = 0; i < 3; i++) omega[i] = k * dR[i];
  }
  if (fabs(sigma) < eps) {
    C = 1;
    if (small) { A = 1. / 2.; B = 1. / 6.; }
    else {
      const double theta2 = theta * theta;
      A = (1 - cos(theta)) / (theta2);
      B = (theta - sin(theta)) / (theta2 * theta);
    }
  } else {
    C = (S.s - 1) / sigma;
    if (small) {
      const double sigma2 = sigma * sigma;
      A = ((sigma - 1) * S.s + 1) / (sigma2);
      B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma);
    } else {
      const double theta2 = theta * theta;
      const double a = S.s * sin(theta), b = S.s * cos(theta);
      const double c = theta2 + sigma * sigma;
      A = (a * sigma + (1 - b) * theta) / (theta * c);
      B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
    }
  }
  const double Om[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
  double W[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) acc += Om[i * 3 + k] * Om[k * 3 + j];
      W[i * 3 + j] = A * Om[i * 3 + j] + B * acc + C * ((i == j) ? 1.0 : 0.0);
    }
  const double t[3] = {S.tx, S.ty, S.tz};
  double ups[3];
  sim3_lu3_solve(W, t, ups);
#pragma unroll
  for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[i + 3] = ups[i]; }
  res[6] = sigma;
}
