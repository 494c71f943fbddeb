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
#include "block_red.h"
#include <chrono>
#include <cfloat>
#include <cstring>

namespace {

using namespace blockred;

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
  unsigned long long* h_ticket; unsigned long long ticket;   // zero-copy calls: written (system scope) after the results are in the pinned block; the host polls it
  long long* dbg;       // CCM_POSEOPT_DBG: phase clocks of thread 0 (10 ns ticks): [0] staging [1] edge passes [2] 27-value reductions [3] solve + update [4] classification [5] passes [6] whole kernel
};
#define POSE_TICK(slot) { if (timing) { const long long tn_ = wall_clock64(); a.dbg[slot] += tn_ - tk; tk = tn_; } }

__device__ __forceinline__ bool chol6_solve(const double* H, double lambda, const double* b, double* x) {
  double L[36], inv[6];
#pragma unroll
  for (int i = 0; i < 36; i++) L[i] = H[i];
#pragma unroll
  for (int i = 0; i < 6; i++) L[i * 7] += lambda;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double d = L[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; k++) d = fma(-L[j * 6 + k], L[j * 6 + k], d);   // (fused: this solve is the serial path of every trial; its rounding has no counterpart in g2o's LDLT to match bit for bit)
    if (!(d > 0.0) || !isfinite(d)) return false;
    const double id = rsqrt(d);  // one reciprocal square root per pivot instead of a square root, a division and 27 more divisions
    L[j * 6 + j] = d * id;
    inv[j] = id;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double s = L[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s = fma(-L[i * 6 + k], L[j * 6 + k], s);
      L[i * 6 + j] = s * id;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s = fma(-L[i * 6 + k], x[k], s);
    x[i] = s * inv[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s = fma(-L[k * 6 + i], x[k], s);
    x[i] = s * inv[i];
  }
  return true;
}

// ONE workgroup of 4 waves.  Every thread carries the pose, H, b and the LM scalars in registers (the serial 6x6
// solve / exp map is executed redundantly by all lanes, SIMT makes that free); edges are strided over the 256
// threads, a thread always owns the same edges, and the only synchronisation is the barrier inside BlockRed
// (2-3 per LM trial).  All threads see bit-identical sums, so every branch below is workgroup-uniform.
// Waves per workgroup, measured on a tracked frame (~1000 edges, host-API time per call): 2 waves 0.176 ms, 4 waves
// 0.153 ms, 8 waves 0.180 ms — more waves shorten the f64 edge pass (~250 instructions per edge) but lengthen the two
// block reductions and barriers of every LM trial.
// (round 4, phase clocks CCM_POSEOPT_DBG=1 on 300 edges: 26 passes; per pass edges 1.68 us — 44 threads own two edges and a lone wave per SIMD issues a dependent f64 instruction
// every ~8 cycles —, the 27-value reduction 1.35 us, solve + update 1.9 us per trial.)  TR: the 27 sums go through LDS (BlockRedT::sum28_lds: 1.15 us, and the kernel around it
// schedules better: 0.141 -> 0.126 ms per call at 300 edges, 0.187 -> 0.144 at 150, 0.278 -> 0.257 at 1000) whenever its 62 KB transpose block fits beside the staged problem.
// Measured and dropped: 8 waves above 256 edges with the serial solve + update on waves 0..3 only and the result broadcast through LDS — the edge passes shrink (43.6 -> 29.6 us)
// but the reductions double (58 us: a cross-lane pre-step and eight waves at every barrier) and the call is slower at every size tried (150 / 300 / 600 / 1000 edges).
constexpr int kPoseWaves = 4, kPoseThreads = 64 * kPoseWaves;
template <bool TR> struct PoseCfg {
  static constexpr int kRed = 2 * kPoseWaves * 64 + (TR ? BlockRedT<kPoseWaves>::kTrDoubles : 0);   // doubles at the head of the LDS: [sum1 / sum27 buffers | transpose block]
};
template <bool TR>
__global__ __launch_bounds__(kPoseThreads) void poseopt_kernel(PoseOptArgs a, int use_lds) {
  constexpr int kPoseRed = PoseCfg<TR>::kRed;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* const tr = sm + 2 * kPoseWaves * 64;
  const int tid = threadIdx.x;
  const bool timing = a.dbg != nullptr && tid == 0;
  long long tk = timing ? wall_clock64() : 0;
  const long long tk0 = tk;
  BlockRedT<kPoseWaves> red{sm, 0, tid & 63, tid >> 6};
  const double delta = (double)(float)sqrt(5.991);
  const double K4[4] = {a.K[0], a.K[1], a.K[2], a.K[3]};
  BaPose T0 = ba_load_pose(a.cam);
  ba_normalize_rotation(T0);
  uint8_t* g_outlier = a.outlier;
  if (use_lds) {
    // a single workgroup has nothing to hide global latency behind: stage the whole problem (64 B + 3 flags per
    // edge) in LDS once; every later pass over the edges runs at LDS latency
    double* base = sm + kPoseRed;
    double* lx = base; double* lo = base + 3 * (size_t)a.n; double* li = base + 5 * (size_t)a.n; double* le = base + 6 * (size_t)a.n;
    uint8_t* lb = reinterpret_cast<uint8_t*>(base + 8 * (size_t)a.n);
    // [Xw | obs | info] are contiguous in the device block and in LDS alike: one flat copy of 6 n doubles, 8 loads per thread in flight before the first LDS store (round 4:
    // as three plain loops the staging was ~8 dependent memory round trips, one per iteration)
    {
      const double* src = a.Xw;   // = din of ccm_pose_optimize
      const int n6 = 6 * a.n;
      for (int b0 = 0; b0 < n6; b0 += 8 * kPoseThreads) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = b0 + u * kPoseThreads + tid; v[u] = i < n6 ? src[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = b0 + u * kPoseThreads + tid; if (i < n6) base[i] = v[u]; }
      }
    }
    a.Xw = lx; a.obs = lo; a.info = li; a.err = le;
    a.level = lb; a.robust = lb + a.n; a.outlier = lb + 2 * (size_t)a.n;
    __syncthreads();
  }
  for (int i = tid; i < a.n; i += kPoseThreads) { a.level[i] = 0; a.robust[i] = 1; a.outlier[i] = 0; a.err[2 * i] = 0; a.err[2 * i + 1] = 0; }
  POSE_TICK(0)
  BaPose T = T0;
  int nBad = 0;

  for (int it = 0; it < 4; it++) {
    T = T0;                                     // vSE3->setEstimate(Converter::toSE3Quat(Frame.mTcw)) (:299)
    double cnt = 0.0;
    for (int i = tid; i < a.n; i += kPoseThreads) cnt += (a.level[i] == 0) ? 1.0 : 0.0;
    const int nact = (int)red.sum1(cnt);
    if (nact > 0) {
      int nBadLM = 0;
      double lambda = 0, ni = 2;
      // One pass over the edges per LM trial: the pass that evaluates the trial state (errors, robust chi2 = tempChi)
      // also accumulates that state's Jacobian products, so that an accepted trial hands the next iteration its
      // linearisation (g2o: computeActiveErrors + linearizeSystem at the top of the next iteration, same values) and the
      // 27 sums travel through one block reduction together with the chi2.  Rejected trials waste the Jacobian work.
      bool lin_current = false;                 // acc[] holds the reduced H / b / chi2 of the current estimate T
      double acc[32];
      auto linearise = [&](const BaPose& P) {   // errors, chi2 (acc[27]) and Jacobian products (acc[0..26]) at P, reduced
#pragma unroll
        for (int k = 0; k < 32; k++) acc[k] = 0;
        // Two edges of a thread per loop trip (i and i + 256), their residual / Jacobian chains written side by side in one basic block: a lone wave per SIMD issues a DEPENDENT f64
        // instruction every ~8 cycles, so the second chain fills the issue slots the first leaves empty (round 5, CCM_DBG=poseopt: a pass over 150 edges 0.95 us, over 300 — 44 threads own two — 1.5 -> 1.37 us, over 1000 2.28 us; the call at 1000 edges 0.257 -> 0.237 ms).  The
        // sums are taken in the same order as before (edge i, then edge i + 256): every result keeps its bits.
        struct EdgeLin { double e0, e1, J[12]; };
        auto chain = [&](int j, EdgeLin& E) {
          const double X[3] = {a.Xw[3 * j], a.Xw[3 * j + 1], a.Xw[3 * j + 2]};
          ba_residual(P, K4, X, a.obs[2 * j], a.obs[2 * j + 1], E.e0, E.e1);
          double Xc[3];
          ba_map(P, X, Xc);
          ba_jacobian_pose_only(Xc, K4, E.J);
        };
        auto accumulate = [&](int i, const EdgeLin& E) {
          const double e0 = E.e0, e1 = E.e1;
          const double* J = E.J;
          a.err[2 * i] = e0; a.err[2 * i + 1] = e1;
          const double om = a.info[i];
          double rho0, w;
          ba_huber((e0 * e0 + e1 * e1) * om, a.robust[i] ? delta : 0.0, rho0, w);
          acc[27] += rho0;
          const double o0 = -om * e0 * w, o1 = -om * e1 * w, wom = w * om;
          // J^T (w Omega) J with the weight folded into one factor and the structural zeros of the pose Jacobian
          // (J[4] = d e0 / d ty = 0, J[9] = d e1 / d tx = 0) left out: 55 multiplies per edge instead of 105
          double Jw[12];
#pragma unroll
          for (int q = 0; q < 12; q++) Jw[q] = J[q] * wom;
          int k = 0;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) {
              const bool a0 = r != 4 && c != 4, a1 = r != 3 && c != 3;   // compile-time after unrolling
              // fused multiply-adds into the running sums (round 4: the file is compiled without contraction so that residuals and Jacobians round as g2o's do; these
              // sums of up to 2 400 terms have no such counterpart to match bit for bit, and a fused add halves their instruction count: 55 instead of ~95 per edge)
              double s = acc[k];
              if (a0) s = fma(Jw[r], J[c], s);
              if (a1) s = fma(Jw[6 + r], J[6 + c], s);
              acc[k++] = s;
            }
#pragma unroll
          for (int r = 0; r < 6; r++) {
            double s = acc[21 + r];
            if (r != 4) s = fma(J[r], o0, s);
            if (r != 3) s = fma(J[6 + r], o1, s);
            acc[21 + r] = s;
          }
        };
        for (int i = tid; i < a.n; i += 2 * kPoseThreads) {
          const int i1 = i + kPoseThreads;
          const bool act0 = a.level[i] == 0, act1 = i1 < a.n && a.level[i1] == 0;
          if (!(act0 || act1)) continue;
          EdgeLin E0, E1;
          chain(act0 ? i : i1, E0);             // a slot whose edge is inactive evaluates the other one again: no value of it is used
          chain(act1 ? i1 : i, E1);
          if (act0) accumulate(i, E0);
          if (act1) accumulate(i1, E1);
        }
        POSE_TICK(1)
        if (TR) red.sum28_lds(acc, tr); else red.sum27(acc);
        if (timing) a.dbg[5] += 1;
        POSE_TICK(2)
      };
      for (int iter = 0; iter < 10; iter++) {
        if (!lin_current) linearise(T);
        double currentChi = acc[27];
        const double iniChi = currentChi;
        double H[36], B[6];
        {
          int k = 0;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = r; c < 6; c++) { H[r * 6 + c] = acc[k]; H[c * 6 + r] = acc[k]; k++; }
#pragma unroll
          for (int r = 0; r < 6; r++) B[r] = acc[21 + r];
        }
        if (iter == 0) {
          double m = 0;
#pragma unroll
          for (int j = 0; j < 6; j++) m = fmax(fabs(H[j * 7]), m);
          lambda = 1e-5 * m; ni = 2; nBadLM = 0;
        }
        int qmax = 0;
        double rho = 0;
        do {
          const BaPose backup = T;
          double xs[6] = {0, 0, 0, 0, 0, 0};
          const bool ok2 = chol6_solve(H, lambda, B, xs);
          if (!ok2) {
#pragma unroll
            for (int k = 0; k < 6; k++) xs[k] = 0;
          }
          T = ba_oplus_fast(xs, T);              // (round 4: the update is ~1 us of every trial's serial path in its closed form, ba_math.h)
          POSE_TICK(3)
          linearise(T);
          double tempChi = acc[27];
          if (!ok2) tempChi = DBL_MAX;
          double scale = 0;
#pragma unroll
          for (int j = 0; j < 6; j++) scale += xs[j] * (lambda * xs[j] + B[j]);
          scale += 1e-3;
          rho = (currentChi - tempChi) / scale;
          if (rho > 0 && isfinite(tempChi)) {
            const double t2 = 2 * rho - 1;
            double alpha = 1. - t2 * t2 * t2;   // pow(x,3): same value to 1 ulp, ~300 fewer f64 instructions
            alpha = fmin(alpha, 2. / 3.);
            lambda *= fmax(1. / 3., alpha);
            ni = 2;
            currentChi = tempChi;
            lin_current = true;                 // acc[] now belongs to the accepted estimate
          } else {
            lambda *= ni; ni *= 2;
            T = backup;
            lin_current = false;                // H, B (registers) still describe the backup estimate for the retry
          }
          qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
      }
      // (as in g2o, err[] keeps the errors of the LAST evaluated state, also when that trial was rejected: the inlier
      // classification below reads e->chi2() without recomputing, Optimizer.cpp:306-334)
    }
    // classification (:306-334)
    double bad = 0.0;
    for (int i = tid; i < a.n; i += kPoseThreads) {
      if (a.outlier[i]) {
        const double X[3] = {a.Xw[3 * i], a.Xw[3 * i + 1], a.Xw[3 * i + 2]};
        double e0, e1;
        ba_residual(T, K4, X, a.obs[2 * i], a.obs[2 * i + 1], e0, e1);
        a.err[2 * i] = e0; a.err[2 * i + 1] = e1;
      }
      const double e0 = a.err[2 * i], e1 = a.err[2 * i + 1];
      const float chi2 = (float)((e0 * e0 + e1 * e1) * a.info[i]);
      if (chi2 > 5.991f) { a.outlier[i] = 1; a.level[i] = 1; bad += 1.0; }
      else { a.outlier[i] = 0; a.level[i] = 0; }
      if (it == 2) a.robust[i] = 0;
    }
    nBad = (int)red.sum1(bad);
    POSE_TICK(4)
    if (a.n < 10) break;
  }
  if (use_lds) for (int i = tid; i < a.n; i += kPoseThreads) g_outlier[i] = a.outlier[i];
  if (tid == 0) { ba_store_pose(a.cam, T); *a.n_bad = nBad; }
  if (a.h_ticket) {   // every thread's result stores are out (system scope) before thread 0 raises the ticket
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.h_ticket, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (timing) a.dbg[6] += wall_clock64() - tk0;
}

}  // namespace

extern "C" int ccm_pose_optimize(ccm_ctx* ctx, double cam_qt[7], int n, const double* Xw, const double* obs, const double* info,
                                 const double K[4], uint8_t* outlier, int* n_inlier) {
  if (!ctx || !cam_qt || n < 0 || (n && (!Xw || !obs || !info || !outlier)) || !K || !n_inlier)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_pose_optimize: bad args");
  if (n < 3) { *n_inlier = 0; return CCM_OK; }   // nInitialCorrespondences < 3 (:290-291)
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // one device block: [cam 7 | n_bad (8 B) | outlier bytes (n, padded to 8) | Xw 3n | obs 2n | info n | err 2n] doubles, then the
  // level / robust bytes.  Inputs travel in one pinned staging buffer => ONE H2D; the results (pose, count, outlier flags)
  // are contiguous at the head of the block => ONE D2H.
  const size_t n_ob = ((size_t)n + 7) / 8;                // doubles holding the outlier bytes
  const size_t n_in = 8 + n_ob + 6 * (size_t)n;           // doubles uploaded
  const size_t nd = n_in + 2 * (size_t)n;
  const size_t bytes = nd * sizeof(double) + 2 * (size_t)n + 64;
  void* scratch = nullptr;
  int rc = ccm_scratch(ctx, bytes, &scratch);
  if (rc) return rc;
  void* pin = nullptr;
  rc = ccm_pin_scratch(ctx, n_in * sizeof(double) + 64, &pin);
  if (rc) return rc;
  double* d = (double*)scratch;
  double* h = (double*)pin;
  PoseOptArgs a;
  a.n = n;
  a.cam = d; a.n_bad = (int*)(d + 7);
  a.outlier = (uint8_t*)(d + 8);
  double* din = d + 8 + n_ob;
  a.Xw = din; a.obs = din + 3 * (size_t)n; a.info = din + 5 * (size_t)n; a.err = din + 6 * (size_t)n;
  uint8_t* bytes_base = (uint8_t*)(d + nd);
  a.level = bytes_base; a.robust = bytes_base + n;
  for (int k = 0; k < 4; k++) a.K[k] = K[k];
  static const bool dbg_env = ccm_dbg("poseopt");
  static long long* d_dbg = nullptr;
  a.dbg = nullptr;
  if (dbg_env) {
    if (!d_dbg) CCM_HIP_CHECK(ctx, hipMalloc(&d_dbg, 8 * sizeof(long long)));
    CCM_HIP_CHECK(ctx, hipMemsetAsync(d_dbg, 0, 8 * sizeof(long long), ctx->stream));
    a.dbg = d_dbg;
  }
  memcpy(h, cam_qt, 7 * sizeof(double));
  h[7] = 0;
  double* hin = h + 8 + n_ob;
  memcpy(hin, Xw, 3 * (size_t)n * sizeof(double));
  memcpy(hin + 3 * (size_t)n, obs, 2 * (size_t)n * sizeof(double));
  memcpy(hin + 5 * (size_t)n, info, (size_t)n * sizeof(double));
  // LDS staging: 64 B of f64 data + 3 flag bytes per edge; the 160 KiB LDS of one CU holds ~2400 edges
  const size_t staged = 8 * (size_t)n * sizeof(double) + 3 * (size_t)n + 16;
  // the 27 sums through LDS whenever the transpose block fits beside the staged problem (up to ~1290 edges); CCM_POSEOPT_SHAPE=0: the cross-lane reduction always
  const bool shape0_env = false;
  // (round 6: with the butterfly's cross-lane moves by DPP (block_red.h, lane_xor.h) the transpose wins up to a few hundred edges only — 0.162 against 0.173 ms per call at 150
  // edges, 0.115 / 0.140 at 300, but 0.234 / 0.202 at 1000: its two barriers and 28 LDS columns grow with the edge passes around them; crossover taken at 600)
  const bool tr_red = !shape0_env && n < 600 && PoseCfg<true>::kRed * sizeof(double) + staged <= 150 * 1024;
  const size_t red_bytes = (tr_red ? PoseCfg<true>::kRed : PoseCfg<false>::kRed) * sizeof(double);
  const size_t lds_full = red_bytes + staged;
  const int use_lds = lds_full <= 150 * 1024;
  // (round 4) a problem that is staged in LDS touches its inputs once and its outputs once: the kernel reads them from / writes them to the PINNED HOST block itself (same
  // layout), which takes the two copy commands and their latency out of the call (0.147 -> ~0.13 ms for 300 edges).  CCM_POSEOPT_COPY=1: through the device block as before.
  const bool copy_env = false;
  const bool zero_copy = use_lds && !copy_env;
  // ... and the host polls a ticket the kernel writes behind its results instead of waiting for the stream (~7 us per call; CCM_POSEOPT_POLL=0: stream wait)
  const bool poll_env = true;
  static thread_local unsigned long long ticket_counter = 0;   // (a ticket only has to differ from the zero written before the launch)
  volatile unsigned long long* h_ticket = reinterpret_cast<volatile unsigned long long*>(h + n_in);
  a.h_ticket = nullptr; a.ticket = 0;
  if (zero_copy) {
    a.cam = h; a.n_bad = (int*)(h + 7); a.outlier = (uint8_t*)(h + 8);
    a.Xw = hin; a.obs = hin + 3 * (size_t)n; a.info = hin + 5 * (size_t)n;
    if (poll_env && !dbg_env) { a.h_ticket = const_cast<unsigned long long*>(h_ticket); a.ticket = ++ticket_counter; *h_ticket = 0; }
  } else CCM_HIP_CHECK(ctx, hipMemcpyAsync(d, h, n_in * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ccm_prof_scope ps(ctx, CCM_K_POSEOPT);
    const size_t lds_bytes = use_lds ? lds_full : red_bytes;
    if (tr_red) {
      CCM_LDS_ATTR(ctx, CCM_LDS_POSEOPT1, poseopt_kernel<true>, 150 * 1024);
      hipLaunchKernelGGL(poseopt_kernel<true>, dim3(1), dim3(kPoseThreads), lds_bytes, ctx->stream, a, use_lds);
    } else {
      if (lds_bytes > 64 * 1024) { CCM_LDS_ATTR(ctx, CCM_LDS_POSEOPT, poseopt_kernel<false>, 150 * 1024); }
      hipLaunchKernelGGL(poseopt_kernel<false>, dim3(1), dim3(kPoseThreads), lds_bytes, ctx->stream, a, use_lds);
    }
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  uint8_t* h_out = (uint8_t*)(h + 8);
  if (!zero_copy) CCM_HIP_CHECK(ctx, hipMemcpyAsync(h, d, (8 + n_ob) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  bool polled = false;
  if (a.h_ticket) {
    const auto t_start = std::chrono::steady_clock::now();
    for (long spin = 0;; spin++) {
      if (*h_ticket == a.ticket) { polled = true; break; }
      if ((spin & 0xffff) == 0xffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 2.0) break;   // a failed launch must not hang the caller
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  if (!polled) CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  else CCM_HIP_CHECK(ctx, hipGetLastError());   // (the ticket says the kernel ran; a sticky asynchronous error is still attributed to THIS call, not to whoever synchronises next)
  if (dbg_env) {
    long long hd[8];
    CCM_HIP_CHECK(ctx, hipMemcpy(hd, d_dbg, sizeof(hd), hipMemcpyDeviceToHost));
    fprintf(stderr, "[ccm_pose_optimize] n %d: kernel %.1f us = staging %.1f + %lld passes (edges %.1f, reductions %.1f) + solve/update %.1f + classification %.1f\n", n, hd[6] * 0.01, hd[0] * 0.01,
            hd[5], hd[1] * 0.01, hd[2] * 0.01, hd[3] * 0.01, hd[4] * 0.01);
  }
  memcpy(cam_qt, h, 7 * sizeof(double));
  memcpy(outlier, h_out, (size_t)n);
  int n_bad = 0;
  memcpy(&n_bad, h + 7, sizeof(int));
  *n_inlier = n - n_bad;
  return CCM_OK;
}
