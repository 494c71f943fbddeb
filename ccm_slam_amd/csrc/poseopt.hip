// poseopt.hip — motion-only pose optimisation, the whole 4-round Levenberg–Marquardt schedule of
// Optimizer::PoseOptimizationClient (cslam/src/Optimizer.cpp:215-347) in ONE kernel launch.
//
// Reference structure: one VertexSE3Expmap, n unary EdgeSE3ProjectXYZOnlyPose edges
// (g2o/types/types_six_dof_expmap.h:143-171, .cpp:266-288), Huber delta (float)sqrt(5.991),
// LinearSolverDense (6x6), 4 x { reset estimate to Frame.mTcw; initializeOptimization(0); optimize(10);
// classify edges by chi2 > 5.991f }, Huber removed after the third round, early exit when fewer than 10 edges.
//
// MI355X design: the problem is tiny (<= ~2000 edges, 6 unknowns) and is called 2-3 times per tracked
// frame, so launch latency is everything: one workgroup runs all <= 4*10*10 LM trials on the device with
// block-wide f64 reductions in LDS; the only host traffic is the input upload and a 7-double + n-byte
// result download.  Reductions use a fixed tree => run-to-run deterministic.
#include "common.h"
#include "ba_math.h"
#include <cfloat>

namespace {

constexpr int kTPB = 256;
constexpr int kWave = 64;

struct PoseOptArgs {
  int n;
  const double* Xw; const double* obs; const double* info;
  double K[4];
  double* cam;          // in/out [7]
  double* err;          // scratch [2n]
  uint8_t* level;       // scratch [n]
  uint8_t* robust;      // scratch [n]
  uint8_t* outlier;     // out [n]
  int* n_bad;           // out
};

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// block-wide sum of NV values per thread; result broadcast to every thread through LDS
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* [NV * 4] */) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = wsum(v[k]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; k++) lds[k * 4 + w] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = lds[k * 4] + lds[k * 4 + 1] + lds[k * 4 + 2] + lds[k * 4 + 3];
}

__device__ __forceinline__ bool chol6_solve(const double* H, double lambda, const double* b, double* x) {
  double L[36];
  for (int i = 0; i < 36; i++) L[i] = H[i];
  for (int i = 0; i < 6; i++) L[i * 7] += lambda;
  for (int j = 0; j < 6; j++) {
    double d = L[j * 6 + j];
    for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k];
    if (!(d > 0.0) || !isfinite(d)) return false;
    d = sqrt(d);
    L[j * 6 + j] = d;
    for (int i = j + 1; i < 6; i++) {
      double s = L[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = s / d;
    }
  }
  for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * x[k]; x[i] = s / L[i * 6 + i]; }
  for (int i = 5; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
  return true;
}

__global__ __launch_bounds__(kTPB) void poseopt_kernel(PoseOptArgs a) {
  __shared__ double red[28 * 4];
  __shared__ double sT[7], sT0[7], sBackup[7], sH[36], sB[6], sX[6];
  __shared__ double sScal[4];   // lambda, ni, rho, tempChi
  __shared__ int sCtl[4];       // ok2, loop flag
  const int tid = threadIdx.x;
  const double delta = (double)(float)sqrt(5.991);
  if (tid < 7) { sT0[tid] = a.cam[tid]; }
  __syncthreads();
  if (tid == 0) { BaPose T = ba_load_pose(sT0); ba_normalize_rotation(T); ba_store_pose(sT0, T); }
  for (int i = tid; i < a.n; i += kTPB) { a.level[i] = 0; a.robust[i] = 1; a.outlier[i] = 0; a.err[2 * i] = 0; a.err[2 * i + 1] = 0; }
  __syncthreads();
  int nBad = 0;

  // robust chi2 of the active edges at pose sT (computeActiveErrors + activeRobustChi2)
  auto chi2_active = [&]() -> double {
    const BaPose T = ba_load_pose(sT);
    double c[1] = {0.0};
    for (int i = tid; i < a.n; i += kTPB) {
      if (a.level[i] != 0) continue;
      const double X[3] = {a.Xw[3 * i], a.Xw[3 * i + 1], a.Xw[3 * i + 2]};
      double e0, e1;
      ba_residual(T, a.K, X, a.obs[2 * i], a.obs[2 * i + 1], e0, e1);
      a.err[2 * i] = e0; a.err[2 * i + 1] = e1;
      const double c2 = (e0 * e0 + e1 * e1) * a.info[i];
      double rho0, w;
      ba_huber(c2, a.robust[i] ? delta : 0.0, rho0, w);
      c[0] += rho0;
    }
    block_sum<1>(c, red);
    return c[0];
  };

  for (int it = 0; it < 4; it++) {
    if (tid < 7) sT[tid] = sT0[tid];            // vSE3->setEstimate(Converter::toSE3Quat(Frame.mTcw)) (:299)
    __syncthreads();
    int nact_l[1] = {0};
    {
      double cnt[1] = {0.0};
      for (int i = tid; i < a.n; i += kTPB) cnt[0] += (a.level[i] == 0) ? 1.0 : 0.0;
      block_sum<1>(cnt, red);
      nact_l[0] = (int)cnt[0];
    }
    if (nact_l[0] > 0) {
      int nBadLM = 0;
      for (int iter = 0; iter < 10; iter++) {
        double currentChi = chi2_active();
        const double iniChi = currentChi;
        // buildSystem: H (21 unique) and b (6)
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; k++) acc[k] = 0;
        {
          const BaPose T = ba_load_pose(sT);
          for (int i = tid; i < a.n; i += kTPB) {
            if (a.level[i] != 0) continue;
            const double X[3] = {a.Xw[3 * i], a.Xw[3 * i + 1], a.Xw[3 * i + 2]};
            double Xc[3];
            ba_map(T, X, Xc);
            double J[12];
            ba_jacobian_pose_only(Xc, a.K, J);
            const double om = a.info[i], e0 = a.err[2 * i], e1 = a.err[2 * i + 1];
            double rho0, w;
            ba_huber((e0 * e0 + e1 * e1) * om, a.robust[i] ? delta : 0.0, rho0, w);
            const double o0 = -om * e0 * w, o1 = -om * e1 * w, wom = w * om;
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
              for (int c = r; c < 6; c++) acc[k++] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * wom;
#pragma unroll
            for (int r = 0; r < 6; r++) acc[21 + r] += J[r] * o0 + J[6 + r] * o1;
          }
        }
        block_sum<27>(acc, red);
        if (tid == 0) {
          int k = 0;
          for (int r = 0; r < 6; r++) for (int c = r; c < 6; c++) { sH[r * 6 + c] = acc[k]; sH[c * 6 + r] = acc[k]; k++; }
          for (int r = 0; r < 6; r++) sB[r] = acc[21 + r];
          if (iter == 0) {
            double m = 0;
            for (int j = 0; j < 6; j++) m = fmax(fabs(sH[j * 7]), m);
            sScal[0] = 1e-5 * m; sScal[1] = 2;
          }
        }
        __syncthreads();
        int qmax = 0;
        double rho = 0;
        do {
          if (tid == 0) {
            for (int k = 0; k < 7; k++) sBackup[k] = sT[k];
            double xs[6] = {0, 0, 0, 0, 0, 0};
            const bool ok2 = chol6_solve(sH, sScal[0], sB, xs);
            if (!ok2) for (int k = 0; k < 6; k++) xs[k] = 0;
            for (int k = 0; k < 6; k++) sX[k] = xs[k];
            const BaPose Tn = ba_oplus(xs, ba_load_pose(sT));
            ba_store_pose(sT, Tn);
            sCtl[0] = ok2 ? 1 : 0;
          }
          __syncthreads();
          double tempChi = chi2_active();
          if (!sCtl[0]) tempChi = DBL_MAX;
          const double lambda = sScal[0];
          double scale = 0;
          for (int j = 0; j < 6; j++) scale += sX[j] * (lambda * sX[j] + sB[j]);
          scale += 1e-3;
          rho = (currentChi - tempChi) / scale;
          __syncthreads();
          if (rho > 0 && isfinite(tempChi)) {
            if (tid == 0) {
              double alpha = 1. - pow((2 * rho - 1), 3);
              alpha = fmin(alpha, 2. / 3.);
              sScal[0] = lambda * fmax(1. / 3., alpha);
              sScal[1] = 2;
            }
            currentChi = tempChi;
          } else {
            if (tid == 0) { sScal[0] = lambda * sScal[1]; sScal[1] *= 2; }
            if (tid < 7) sT[tid] = sBackup[tid];
          }
          __syncthreads();
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
      }
    }
    // classification (:306-334)
    {
      const BaPose T = ba_load_pose(sT);
      double bad[1] = {0.0};
      for (int i = tid; i < a.n; i += kTPB) {
        if (a.outlier[i]) {
          const double X[3] = {a.Xw[3 * i], a.Xw[3 * i + 1], a.Xw[3 * i + 2]};
          double e0, e1;
          ba_residual(T, a.K, X, a.obs[2 * i], a.obs[2 * i + 1], e0, e1);
          a.err[2 * i] = e0; a.err[2 * i + 1] = e1;
        }
        const double e0 = a.err[2 * i], e1 = a.err[2 * i + 1];
        const float chi2 = (float)((e0 * e0 + e1 * e1) * a.info[i]);
        if (chi2 > 5.991f) { a.outlier[i] = 1; a.level[i] = 1; bad[0] += 1.0; }
        else { a.outlier[i] = 0; a.level[i] = 0; }
        if (it == 2) a.robust[i] = 0;
      }
      block_sum<1>(bad, red);
      nBad = (int)bad[0];
    }
    __syncthreads();
    if (a.n < 10) break;
  }
  if (tid < 7) a.cam[tid] = sT[tid];
  if (tid == 0) *a.n_bad = nBad;
}

}  // namespace

extern "C" int ccm_pose_optimize(ccm_ctx* ctx, double cam_qt[7], int n, const double* Xw, const double* obs, const double* info,
                                 const double K[4], uint8_t* outlier, int* n_inlier) {
  if (!ctx || !cam_qt || n < 0 || (n && (!Xw || !obs || !info || !outlier)) || !K || !n_inlier)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_pose_optimize: bad args");
  if (n < 3) { *n_inlier = 0; return CCM_OK; }   // nInitialCorrespondences < 3 (:290-291)
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // one scratch block: [cam 7 | Xw 3n | obs 2n | info n | err 2n] doubles, then level/robust/outlier bytes, n_bad int
  const size_t nd = 7 + 8 * (size_t)n;
  const size_t bytes = nd * sizeof(double) + 3 * (size_t)n + 64;
  void* scratch = nullptr;
  int rc = ccm_scratch(ctx, bytes, &scratch);
  if (rc) return rc;
  double* d = (double*)scratch;
  PoseOptArgs a;
  a.n = n;
  a.cam = d; a.Xw = d + 7; a.obs = d + 7 + 3 * (size_t)n; a.info = d + 7 + 5 * (size_t)n; a.err = d + 7 + 6 * (size_t)n;
  uint8_t* bytes_base = (uint8_t*)(d + nd);
  a.level = bytes_base; a.robust = bytes_base + n; a.outlier = bytes_base + 2 * (size_t)n;
  a.n_bad = (int*)(bytes_base + ((3 * (size_t)n + 15) & ~(size_t)15));
  for (int k = 0; k < 4; k++) a.K[k] = K[k];
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(a.cam, cam_qt, 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.Xw, Xw, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.obs, obs, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync((void*)a.info, info, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ccm_prof_scope ps(ctx, CCM_K_POSEOPT);
    hipLaunchKernelGGL(poseopt_kernel, dim3(1), dim3(kTPB), 0, ctx->stream, a);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  int n_bad = 0;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(cam_qt, a.cam, 7 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(outlier, a.outlier, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(&n_bad, a.n_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  *n_inlier = n - n_bad;
  return CCM_OK;
}
