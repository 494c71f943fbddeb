// sim3opt.hip — Optimizer::OptimizeSim3 (cslam/src/Optimizer.cpp:861-1056) in ONE kernel launch.
//
// Reference structure: one VertexSim3Expmap (7 DoF, scale optionally frozen), per map-point pair two edges whose
// point vertices are fixed — EdgeSim3ProjectXYZ (x1 = S12 X2) and EdgeInverseSim3ProjectXYZ (x2 = S12^-1 X1)
// (g2o/types/types_seven_dof_expmap.h:133-172) — info invSigma2*I, Huber delta (float)sqrt(th2), BlockSolverX +
// LinearSolverDense (7x7), Levenberg: optimize(5) -> drop pairs with chi2 > th2 on either edge ->
// optimize(10 if any dropped else 5) -> count inliers; fewer than 10 survivors of the first pass => return 0 and
// leave S12 untouched (:1015-1016).  Neither edge overrides linearizeOplus, so upstream differentiates numerically
// (base_binary_edge.hpp:129-196: central differences of the error through the vertex oplus, delta 1e-9).  That
// scheme defines the reference's Jacobian to ~1e-7, so it is reproduced, not replaced by analytic derivatives.
//
// MI355X design: like the pose optimiser this is a tiny, latency-bound problem (<= a few hundred pairs, 7 unknowns),
// so the whole schedule (up to 15 LM iterations x 10 trials) runs in one 4-wave workgroup with the problem staged in
// LDS.  The 14 perturbed estimates S(+-delta e_d) and their inverses are the same for every edge: 14 lanes compute
// them once per iteration into LDS, then each thread differentiates its own pairs (28 maps + projections per pair).
// H (28 unique) and b (7) are reduced with one 64-value halving butterfly + one barrier.
#include "common.h"
#include "sim3_math.h"
#include "block_red.h"
#include <cfloat>
#include <cstring>

namespace {

using namespace blockred;

struct Sim3OptArgs {
  int n, fix_scale;
  double th2;
  const double* P1c; const double* P2c; const double* obs1; const double* obs2; const double* info1; const double* info2;
  double K1[4], K2[4];
  double* sim3;       // in/out [8]
  int* n_in;          // out
  double* err;        // scratch [4n]
  uint8_t* alive;     // scratch [n]
  uint8_t* inlier;    // out [n]
};

constexpr int kPertDoubles = 14 * 16;   // per perturbation: S (8) | S^-1 (8)

__device__ __forceinline__ bool chol7_solve(const double* H, double lambda, const double* b, double* x) {
  double L[49], inv[7];
#pragma unroll
  for (int i = 0; i < 49; i++) L[i] = H[i];
#pragma unroll
  for (int i = 0; i < 7; i++) L[i * 8] += lambda;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    double d = L[j * 7 + j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j * 7 + k] * L[j * 7 + k];
    if (!(d > 0.0) || !isfinite(d)) return false;
    d = sqrt(d);
    L[j * 7 + j] = d;
    const double id = 1.0 / d;
    inv[j] = id;
#pragma unroll
    for (int i = j + 1; i < 7; i++) {
      double s = L[i * 7 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * 7 + k] * L[j * 7 + k];
      L[i * 7 + j] = s * id;
    }
  }
#pragma unroll
  for (int i = 0; i < 7; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * 7 + k] * x[k];
    x[i] = s * inv[i];
  }
#pragma unroll
  for (int i = 6; i >= 0; i--) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < 7; k++) s -= L[k * 7 + i] * x[k];
    x[i] = s * inv[i];
  }
  return true;
}

__device__ __forceinline__ void proj_err(const Sim3d& X, const double* P, const double* obs, const double* K, double& e0, double& e1) {
  const double p3[3] = {P[0], P[1], P[2]};
  double p[3];
  sim3_map(X, p3, p);
  e0 = obs[0] - ((p[0] / p[2]) * K[0] + K[2]);
  e1 = obs[1] - ((p[1] / p[2]) * K[1] + K[3]);
}

__global__ __launch_bounds__(kThreads) void sim3opt_kernel(Sim3OptArgs a, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  BlockRed red{sm, 0, tid & 63, tid >> 6};
  double* pert = sm + kRedDoubles;
  const double delta = (double)(float)sqrt((float)a.th2);
  const double K1[4] = {a.K1[0], a.K1[1], a.K1[2], a.K1[3]};
  const double K2[4] = {a.K2[0], a.K2[1], a.K2[2], a.K2[3]};
  uint8_t* g_inlier = a.inlier;
  const int n = a.n;
  if (use_lds) {
    double* base = pert + kPertDoubles;
    double* l1 = base; double* l2 = base + 3 * (size_t)n; double* o1 = base + 6 * (size_t)n; double* o2 = base + 8 * (size_t)n;
    double* i1 = base + 10 * (size_t)n; double* i2 = base + 11 * (size_t)n; double* le = base + 12 * (size_t)n;
    uint8_t* lb = reinterpret_cast<uint8_t*>(base + 16 * (size_t)n);
    for (int i = tid; i < 3 * n; i += kThreads) { l1[i] = a.P1c[i]; l2[i] = a.P2c[i]; }
    for (int i = tid; i < 2 * n; i += kThreads) { o1[i] = a.obs1[i]; o2[i] = a.obs2[i]; }
    for (int i = tid; i < n; i += kThreads) { i1[i] = a.info1[i]; i2[i] = a.info2[i]; }
    a.P1c = l1; a.P2c = l2; a.obs1 = o1; a.obs2 = o2; a.info1 = i1; a.info2 = i2; a.err = le;
    a.alive = lb; a.inlier = lb + n;
  }
  for (int i = tid; i < n; i += kThreads) {
    a.alive[i] = 1; a.inlier[i] = 1;
    a.err[4 * i] = 0; a.err[4 * i + 1] = 0; a.err[4 * i + 2] = 0; a.err[4 * i + 3] = 0;
  }
  __syncthreads();
  Sim3d S = sim3_load(a.sim3);

  auto oplus = [&](const Sim3d& X, const double* upd) {
    double u[7];
#pragma unroll
    for (int k = 0; k < 7; k++) u[k] = upd[k];
    if (a.fix_scale) u[6] = 0;                       // VertexSim3Expmap::oplusImpl (types_seven_dof_expmap.h:58-67)
    return sim3_mul(sim3_exp(u), X);
  };
  auto chi2_active = [&](const Sim3d& X) -> double {
    const Sim3d Xi = sim3_inv(X);
    double c = 0.0;
    for (int i = tid; i < n; i += kThreads) {
      if (!a.alive[i]) continue;
      double e0, e1, rho0, w;
      proj_err(X, a.P2c + 3 * i, a.obs1 + 2 * i, K1, e0, e1);
      a.err[4 * i] = e0; a.err[4 * i + 1] = e1;
      ba_huber((e0 * e0 + e1 * e1) * a.info1[i], delta, rho0, w);
      c += rho0;
      proj_err(Xi, a.P1c + 3 * i, a.obs2 + 2 * i, K2, e0, e1);
      a.err[4 * i + 2] = e0; a.err[4 * i + 3] = e1;
      ba_huber((e0 * e0 + e1 * e1) * a.info2[i], delta, rho0, w);
      c += rho0;
    }
    return red.sum1(c);
  };

  auto optimize = [&](int iters) {
    double cnt = 0.0;
    for (int i = tid; i < n; i += kThreads) cnt += a.alive[i] ? 1.0 : 0.0;
    if ((int)red.sum1(cnt) == 0) return;
    int nBadLM = 0;
    double lambda = 0, ni = 2;
    bool err_current = false;
    double carriedChi = 0;
    for (int iter = 0; iter < iters; iter++) {
      double currentChi = err_current ? carriedChi : chi2_active(S);
      const double iniChi = currentChi;
      // the 14 perturbed estimates (and inverses) of the numeric differentiation, once per iteration
      if (tid < 14) {
        const int d = tid >> 1;
        double add[7] = {0, 0, 0, 0, 0, 0, 0};
        const double dl = (tid & 1) ? -1e-9 : 1e-9;
#pragma unroll
        for (int k = 0; k < 7; k++) if (k == d) add[k] = dl;
        const Sim3d Sp = oplus(S, add);
        const Sim3d Si = sim3_inv(Sp);
        sim3_store(pert + tid * 16, Sp);
        sim3_store(pert + tid * 16 + 8, Si);
      }
      __syncthreads();
      double acc[64];
#pragma unroll
      for (int k = 0; k < 64; k++) acc[k] = 0;
      const double scalar = 1.0 / (2 * 1e-9);
      for (int i = tid; i < n; i += kThreads) {
        if (!a.alive[i]) continue;
#pragma unroll 1
        for (int side = 0; side < 2; side++) {   // not unrolled: the 14 perturbed Sim3 of both sides at once do not fit the register file
          const double* P = side ? a.P1c + 3 * i : a.P2c + 3 * i;
          const double* ob = side ? a.obs2 + 2 * i : a.obs1 + 2 * i;
          const double* KK = side ? K2 : K1;
          double J[14];
#pragma unroll
          for (int d = 0; d < 7; d++) {
            const Sim3d Xp = sim3_load(pert + (2 * d) * 16 + side * 8);
            const Sim3d Xm = sim3_load(pert + (2 * d + 1) * 16 + side * 8);
            double p0, p1, m0, m1;
            proj_err(Xp, P, ob, KK, p0, p1);
            proj_err(Xm, P, ob, KK, m0, m1);
            J[d] = scalar * (p0 - m0); J[7 + d] = scalar * (p1 - m1);
          }
          const double om = side ? a.info2[i] : a.info1[i];
          const double e0 = a.err[4 * i + 2 * side], e1 = a.err[4 * i + 2 * side + 1];
          double rho0, w;
          ba_huber((e0 * e0 + e1 * e1) * om, delta, rho0, w);
          const double o0 = -om * e0 * w, o1 = -om * e1 * w, wom = w * om;
          int k = 0;
#pragma unroll
          for (int r = 0; r < 7; r++)
#pragma unroll
            for (int c = r; c < 7; c++) acc[k++] += (J[r] * wom) * J[c] + (J[7 + r] * wom) * J[7 + c];
#pragma unroll
          for (int r = 0; r < 7; r++) acc[28 + r] += J[r] * o0 + J[7 + r] * o1;
        }
      }
      red.sum64<35>(acc);
      double H[49], B[7];
      {
        int k = 0;
#pragma unroll
        for (int r = 0; r < 7; r++)
#pragma unroll
          for (int c = r; c < 7; c++) { H[r * 7 + c] = acc[k]; H[c * 7 + r] = acc[k]; k++; }
#pragma unroll
        for (int r = 0; r < 7; r++) B[r] = acc[28 + r];
      }
      if (iter == 0) {
        double m = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) m = fmax(fabs(H[j * 8]), m);
        lambda = 1e-5 * m; ni = 2; nBadLM = 0;
      }
      int qmax = 0;
      double rho = 0;
      do {
        const Sim3d backup = S;
        double xs[7] = {0, 0, 0, 0, 0, 0, 0};
        const bool ok2 = chol7_solve(H, lambda, B, xs);
        if (!ok2) { for (int k = 0; k < 7; k++) xs[k] = 0; }
        S = oplus(S, xs);
        double tempChi = chi2_active(S);
        if (!ok2) tempChi = DBL_MAX;
        double scale = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) scale += xs[j] * (lambda * xs[j] + B[j]);
        scale += 1e-3;
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && isfinite(tempChi)) {
          const double t2 = 2 * rho - 1;
          double alpha = 1. - t2 * t2 * t2;
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          err_current = true; carriedChi = tempChi;
        } else {
          lambda *= ni; ni *= 2;
          S = backup;
          err_current = false;
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
      if (nBadLM >= 3) break;
    }
  };

  auto over = [&](int i) {
    const double c12 = (a.err[4 * i] * a.err[4 * i] + a.err[4 * i + 1] * a.err[4 * i + 1]) * a.info1[i];
    const double c21 = (a.err[4 * i + 2] * a.err[4 * i + 2] + a.err[4 * i + 3] * a.err[4 * i + 3]) * a.info2[i];
    return c12 > a.th2 || c21 > a.th2;
  };

  optimize(5);
  double bad = 0.0;
  for (int i = tid; i < n; i += kThreads)
    if (over(i)) { a.alive[i] = 0; a.inlier[i] = 0; bad += 1.0; }
  const int nBad = (int)red.sum1(bad);
  int nIn = 0;
  const bool go_on = (n - nBad) >= 10;
  if (go_on) {
    optimize(nBad > 0 ? 10 : 5);
    double good = 0.0;
    for (int i = tid; i < n; i += kThreads) {
      if (!a.alive[i]) continue;
      if (over(i)) a.inlier[i] = 0; else good += 1.0;
    }
    nIn = (int)red.sum1(good);
  }
  if (use_lds) for (int i = tid; i < n; i += kThreads) g_inlier[i] = a.inlier[i];
  if (tid == 0) {
    if (go_on) sim3_store(a.sim3, S);
    *a.n_in = nIn;
  }
}

}  // namespace

extern "C" int ccm_sim3_optimize(ccm_ctx* ctx, double sim3[8], int n, const double* P1c, const double* P2c, const double* obs1,
                                 const double* obs2, const double* info1, const double* info2, const double K1[4],
                                 const double K2[4], double th2, int fix_scale, uint8_t* inlier, int* n_inlier) {
  if (!ctx || !sim3 || n < 0 || (n && (!P1c || !P2c || !obs1 || !obs2 || !info1 || !info2 || !inlier)) || !K1 || !K2 || !n_inlier)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_sim3_optimize: bad args");
  *n_inlier = 0;
  if (n == 0) return CCM_OK;   // nCorrespondences - nBad < 10 (:1015-1016): nothing to do, S12 untouched
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // device block: [sim3 8 | n_in (8 B) | P1c 3n | P2c 3n | obs1 2n | obs2 2n | info1 n | info2 n | err 4n] doubles, then
  // inlier / alive bytes.  One H2D from the pinned staging buffer, two small D2H.
  const size_t n_in_d = 9 + 12 * (size_t)n;
  const size_t nd = n_in_d + 4 * (size_t)n;
  void* scratch = nullptr;
  int rc = ccm_scratch(ctx, nd * sizeof(double) + 2 * (size_t)n + 64, &scratch);
  if (rc) return rc;
  void* pin = nullptr;
  rc = ccm_pin_scratch(ctx, n_in_d * sizeof(double) + (size_t)n + 64, &pin);
  if (rc) return rc;
  double* d = (double*)scratch;
  double* h = (double*)pin;
  Sim3OptArgs a;
  a.n = n; a.fix_scale = fix_scale ? 1 : 0; a.th2 = th2;
  a.sim3 = d; a.n_in = (int*)(d + 8);
  a.P1c = d + 9; a.P2c = d + 9 + 3 * (size_t)n; a.obs1 = d + 9 + 6 * (size_t)n; a.obs2 = d + 9 + 8 * (size_t)n;
  a.info1 = d + 9 + 10 * (size_t)n; a.info2 = d + 9 + 11 * (size_t)n; a.err = d + n_in_d;
  uint8_t* bytes_base = (uint8_t*)(d + nd);
  a.inlier = bytes_base; a.alive = bytes_base + n;
  for (int k = 0; k < 4; k++) { a.K1[k] = K1[k]; a.K2[k] = K2[k]; }
  memcpy(h, sim3, 8 * sizeof(double));
  h[8] = 0;
  memcpy(h + 9, P1c, 3 * (size_t)n * sizeof(double));
  memcpy(h + 9 + 3 * (size_t)n, P2c, 3 * (size_t)n * sizeof(double));
  memcpy(h + 9 + 6 * (size_t)n, obs1, 2 * (size_t)n * sizeof(double));
  memcpy(h + 9 + 8 * (size_t)n, obs2, 2 * (size_t)n * sizeof(double));
  memcpy(h + 9 + 10 * (size_t)n, info1, (size_t)n * sizeof(double));
  memcpy(h + 9 + 11 * (size_t)n, info2, (size_t)n * sizeof(double));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d, h, n_in_d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ccm_prof_scope ps(ctx, CCM_K_SIM3OPT);
    const size_t lds_head = (kRedDoubles + kPertDoubles) * sizeof(double);
    const size_t lds_full = lds_head + 16 * (size_t)n * sizeof(double) + 2 * (size_t)n + 16;
    const int use_lds = lds_full <= 150 * 1024;
    const size_t lds_bytes = use_lds ? lds_full : lds_head;
    if (lds_bytes > 64 * 1024) {
      CCM_LDS_ATTR(ctx, CCM_LDS_SIM3OPT, sim3opt_kernel, 150 * 1024);
    }
    hipLaunchKernelGGL(sim3opt_kernel, dim3(1), dim3(kThreads), lds_bytes, ctx->stream, a, use_lds);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  uint8_t* h_out = (uint8_t*)(h + 9);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(h, d, 9 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_out, a.inlier, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(sim3, h, 8 * sizeof(double));
  memcpy(inlier, h_out, (size_t)n);
  int nin = 0;
  memcpy(&nin, h + 8, sizeof(int));
  *n_inlier = nin;
  return CCM_OK;
}
