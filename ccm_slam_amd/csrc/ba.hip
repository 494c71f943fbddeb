// ba.hip — Levenberg–Marquardt Schur-complement bundle adjustment on MI355X (gfx950), f64.
//
// Replaces what Optimizer::{BundleAdjustmentClient, LocalBundleAdjustmentClient, MapFusionGBA}
// (cslam/src/Optimizer.cpp:40-212, 349-644, 646-859) hand to g2o: BlockSolver_6_3 +
// OptimizationAlgorithmLevenberg over EdgeSE3ProjectXYZ edges with a Huber kernel
// (g2o/core/block_solver.hpp:354-604, optimization_algorithm_levenberg.cpp:61-189,
//  types/types_six_dof_expmap.cpp:103-147, core/robust_kernel_impl.cpp:78-90).
//
// MI355X design (not a translation of g2o's pointer graph):
//  * problem flattened to SoA buffers resident in HBM; edges sorted by landmark.  Landmark-side work (Hll, b_l, W,
//    back-substitution, chi2) runs one thread per OBSERVATION on chunks of consecutive landmarks, with the per-landmark
//    sums parked in LDS and added in observation order; camera-side work (Hpp, b_p) one wave per camera over its edge
//    list with a halving-butterfly wave reduction — no atomics anywhere, so every sum has a fixed order and the solver
//    is bit-reproducible run to run;
//  * Schur blocks are *gathered*: the pair structure ((edge_a, edge_c) instances per block) is built once per problem
//    on the device; a workgroup per camera row keeps Y = W D^-1 of the camera's observations in LDS and feeds work
//    units of <= 64 pair instances, two instances per v_mfma_f64_16x16x4, to its 16 waves (off-diagonal AND diagonal
//    blocks, b_schur as a seventh column);
//  * reduced camera system: ONE persistent kernel per LM trial (16 < cameras <= 2048) runs the whole preconditioned CG —
//    S rows in registers, the 16-camera cluster inverse in LDS, grid-wide steps by an atomics-free slot exchange, an
//    adaptively switched coarse level of rigid-body modes per 32 cameras; <= 16 cameras: one workgroup; > 2048: two
//    kernels per CG iteration;
//  * sharding: landmarks (with all their edges) are partitioned across ranks, the
//    camera state is replicated, and ONE RCCL all-reduce per LM trial sums
//    [S blocks | b_schur] over xGMI; every rank then solves the identical reduced system.
//  * LM control (lambda schedule, rho test, stop rules) runs on the host exactly as
//    optimization_algorithm_levenberg.cpp:61-164 does; one 48-byte pinned D2H read per trial.
#include "common.h"
#include "test_internal.h"
#include "ba_types.h"
#include "lane_xor.h"
#include "ba_math.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

int ccm_allreduce_f64(ccm_ctx* ctx, double* d_buf, size_t n);   // comm.hip
int ccm_allreduce_max_f64(ccm_ctx* ctx, double* d_buf, size_t n);

namespace {


inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) { return lanex::wave_sum(v); }   // (xor butterfly 32 ... 1; round 6: by DPP / ds_swizzle where they reach, lane_xor.h)

// deterministic block sum (fixed tree), result valid in thread 0; blockDim.x == kTPB
__device__ __forceinline__ double block_sum(double v, double* lds /* >= kTPB/kWave */) {
  v = wave_sum(v);
  const int w = threadIdx.x / kWave;
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) lds[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kTPB / kWave; i++) s += lds[i];
  return s;
}

// Sums of NS arrays of per-workgroup partials, identical result in every thread of every workgroup (fixed order: thread-strided partial
// sums with four loads in flight, wave tree, waves in order).  The whole workgroup shares the work: with every WAVE summing all partials on its
// own (one load in flight per lane) the 10 000-keyframe map spent ~45 us of each ba_pcg_update and ~20 us of each ba_pcg_spmv on 6 000 /
// 2 500 serial L2 round trips before touching its rows.  red: LDS scratch [NS][kTPB / kWave]; contains a block barrier.
// a value every lane already agrees on, moved to scalar registers (frees 2 VGPRs per double)
__device__ __forceinline__ double to_sgpr(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int NS, int TPB = kTPB>
__device__ __forceinline__ void block_sum_partials(const double* const (&p)[NS], const int (&n)[NS], double (&out)[NS], double* red) {
  const int t = threadIdx.x, lane = t & (kWave - 1), wv = t / kWave;
#pragma unroll
  for (int q = 0; q < NS; q++) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int i = t;
    for (; i + 3 * TPB < n[q]; i += 4 * TPB) {
      const double v0 = p[q][i], v1 = p[q][i + TPB], v2 = p[q][i + 2 * TPB], v3 = p[q][i + 3 * TPB];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; i < n[q]; i += TPB) a0 += p[q][i];
    const double w = wave_sum((a0 + a1) + (a2 + a3));
    if (lane == 0) red[q * (TPB / kWave) + wv] = w;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NS; q++) {
    double tot = red[q * (TPB / kWave)];
#pragma unroll
    for (int w = 1; w < TPB / kWave; w++) tot += red[q * (TPB / kWave) + w];
    out[q] = tot;
  }
}


// ---------------------------------------------------------------------------------------------
// linearisation: landmark side (one thread per own landmark)          [CCM_K_BA_LINEARIZE]
// buildSystem (block_solver.hpp:502-560) for Hll, b_l and the Hpl blocks
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTPB) void ba_linearize_pts(BaDev d, int cur) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.Lloc) return;
  const double X[3] = {d.pt[cur][3 * l], d.pt[cur][3 * l + 1], d.pt[cur][3 * l + 2]};
  double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  const int e0 = d.pt_off[l], e1 = d.pt_off[l + 1];
  for (int e = e0; e < e1; e++) {
    const int c = d.ed_cam[e];
    const BaPose T = ba_load_pose(d.cam[cur] + 7 * (size_t)c);
    const double* Kc = d.K + 4 * (size_t)c;
    const double K4[4] = {Kc[0], Kc[1], Kc[2], Kc[3]};
    double r0, r1;
    ba_residual(T, K4, X, d.obs[2 * (size_t)e], d.obs[2 * (size_t)e + 1], r0, r1);
    double Ji[6], Jj[12];
    ba_jacobians(T, K4, X, Ji, Jj);
    const double om = d.info[e];
    if (om == 0.0) {   // an edge that left through ccm_ba_set_edge_levels: exact zeros, whatever its residual is (g2o never evaluates a level-1 edge; inf * 0 would poison the sums)
      if (d.ed_cslot[e] >= 0) { double* W = d.W + 18 * (size_t)e; for (int k = 0; k < 18; k++) W[k] = 0.0; }
      continue;
    }
    double rho0, w;
    ba_huber((r0 * r0 + r1 * r1) * om, d.huber, rho0, w);
    const double o0 = -om * r0 * w, o1 = -om * r1 * w, wom = w * om;
    b[0] += Ji[0] * o0 + Ji[3] * o1; b[1] += Ji[1] * o0 + Ji[4] * o1; b[2] += Ji[2] * o0 + Ji[5] * o1;
    H[0] += (Ji[0] * Ji[0] + Ji[3] * Ji[3]) * wom; H[1] += (Ji[0] * Ji[1] + Ji[3] * Ji[4]) * wom;
    H[2] += (Ji[0] * Ji[2] + Ji[3] * Ji[5]) * wom; H[3] += (Ji[1] * Ji[1] + Ji[4] * Ji[4]) * wom;
    H[4] += (Ji[1] * Ji[2] + Ji[4] * Ji[5]) * wom; H[5] += (Ji[2] * Ji[2] + Ji[5] * Ji[5]) * wom;
    if (d.ed_cslot[e] >= 0) {
      double* W = d.W + 18 * (size_t)e;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W[i * 3 + j] = (Jj[i] * Ji[j] + Jj[6 + i] * Ji[3 + j]) * wom;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) d.Hll[6 * (size_t)l + i] = H[i];
  d.bl[3 * (size_t)l] = b[0]; d.bl[3 * (size_t)l + 1] = b[1]; d.bl[3 * (size_t)l + 2] = b[2];
}

// Edge-parallel variant (default).  One thread per landmark leaves 150 000 threads = 2.3 waves per SIMD walking
// dependent index -> pose -> observation chains (97 us, 2.7 TB/s); here a workgroup takes a chunk of consecutive
// landmarks with <= 256 observations, one thread per OBSERVATION does the projection, Jacobians, robust weight and the
// 6x3 W block (written as 144 contiguous bytes per thread, contiguous across the workgroup), parks the 9 landmark-side
// contributions in LDS, and one thread per landmark adds them in observation order: the same additions in the same
// order as the loop above, hence bit-identical Hll / b_l.
__global__ __launch_bounds__(kTPB) void ba_linearize_pts_e(BaDev d, int cur, double lambda_fused /* >= 0: also form D^-1 and D^-1 b_l for this lambda (ba_dinv folded in: one launch less per LM iteration) */) {
  __shared__ double hb[kTPB][9];
  const int l0 = d.chunk_off[blockIdx.x], l1 = d.chunk_off[blockIdx.x + 1];
  const int e0 = d.pt_off[l0], ne = d.pt_off[l1] - e0, nl = l1 - l0;
  const int t = threadIdx.x;
  if (t < ne) {
    const int e = e0 + t, l = d.ed_pt[e], c = d.ed_cam[e];
    const double X[3] = {d.pt[cur][3 * (size_t)l], d.pt[cur][3 * (size_t)l + 1], d.pt[cur][3 * (size_t)l + 2]};
    const BaPose T = ba_load_pose(d.cam[cur] + 7 * (size_t)c);
    const double* Kc = d.K + 4 * (size_t)c;
    const double K4[4] = {Kc[0], Kc[1], Kc[2], Kc[3]};
    double r0, r1;
    ba_residual(T, K4, X, d.obs[2 * (size_t)e], d.obs[2 * (size_t)e + 1], r0, r1);
    double Ji[6], Jj[12], Xc[3], Rm[9];
    ba_map(T, X, Xc);
    const double iz = 1.0 / Xc[2];
    ba_q_to_R(T, Rm);
    ba_jac_from_xc(Xc[0], Xc[1], iz, K4[0], K4[1], Rm, Ji, Jj);   // = ba_jacobians(T, K4, X, Ji, Jj)
    const double om = d.info[e];
    const bool off = om == 0.0;   // an edge that left through ccm_ba_set_edge_levels contributes exact zeros whatever its residual is (g2o never evaluates a level-1 edge; inf * 0 would poison the sums)
    double rho0, w;
    ba_huber(off ? 0.0 : (r0 * r0 + r1 * r1) * om, d.huber, rho0, w);
    const double o0 = off ? 0.0 : -om * r0 * w, o1 = off ? 0.0 : -om * r1 * w, wom = off ? 0.0 : w * om;
    if (off) {
#pragma unroll
      for (int k = 0; k < 6; k++) Ji[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 12; k++) Jj[k] = 0.0;
    }
    if (d.E4L && d.ed_cslot[e] >= 0) {   // the compact record in landmark-major order (ba_backsub_chi2_e): coalesced, 32 bytes instead of the 144 of the block
      typedef double v2d __attribute__((ext_vector_type(2)));
      v2d* E = reinterpret_cast<v2d*>(d.E4L + 4 * (size_t)e);
      v2d ea, eb; ea[0] = off ? 0.0 : Xc[0]; ea[1] = off ? 0.0 : Xc[1]; eb[0] = off ? 0.0 : iz; eb[1] = wom;
      E[0] = ea; E[1] = eb;
    }
    hb[t][0] = Ji[0] * o0 + Ji[3] * o1; hb[t][1] = Ji[1] * o0 + Ji[4] * o1; hb[t][2] = Ji[2] * o0 + Ji[5] * o1;
    hb[t][3] = (Ji[0] * Ji[0] + Ji[3] * Ji[3]) * wom; hb[t][4] = (Ji[0] * Ji[1] + Ji[3] * Ji[4]) * wom;
    hb[t][5] = (Ji[0] * Ji[2] + Ji[3] * Ji[5]) * wom; hb[t][6] = (Ji[1] * Ji[1] + Ji[4] * Ji[4]) * wom;
    hb[t][7] = (Ji[1] * Ji[2] + Ji[4] * Ji[5]) * wom; hb[t][8] = (Ji[2] * Ji[2] + Ji[5] * Ji[5]) * wom;
    if (!d.w_free && d.ed_cslot[e] >= 0) {
      double* W = d.W + 18 * (size_t)e;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W[i * 3 + j] = (Jj[i] * Ji[j] + Jj[6 + i] * Ji[3 + j]) * wom;
    }
  }
  __syncthreads();
  if (t < nl) {
    const int l = l0 + t;
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int k = d.pt_off[l] - e0; k < d.pt_off[l + 1] - e0; k++) {
      b[0] += hb[k][0]; b[1] += hb[k][1]; b[2] += hb[k][2];
#pragma unroll
      for (int i = 0; i < 6; i++) H[i] += hb[k][3 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) d.Hll[6 * (size_t)l + i] = H[i];
    d.bl[3 * (size_t)l] = b[0]; d.bl[3 * (size_t)l + 1] = b[1]; d.bl[3 * (size_t)l + 2] = b[2];
    if (lambda_fused >= 0.0) {   // exactly ba_dinv's arithmetic on the values just stored
      double a[6], r[6];
#pragma unroll
      for (int i = 0; i < 6; i++) a[i] = H[i];
      a[0] += lambda_fused; a[3] += lambda_fused; a[5] += lambda_fused;
      ba_sym3_inv(a, r);
#pragma unroll
      for (int i = 0; i < 6; i++) d.Dinv[6 * (size_t)l + i] = r[i];
      d.dl[3 * (size_t)l] = r[0] * b[0] + r[1] * b[1] + r[2] * b[2];
      d.dl[3 * (size_t)l + 1] = r[1] * b[0] + r[3] * b[1] + r[4] * b[2];
      d.dl[3 * (size_t)l + 2] = r[2] * b[0] + r[4] * b[1] + r[5] * b[2];
    }
  }
}

// 27 wave-wide sums (upper triangle of a 6x6 + a 6-vector) with a halving butterfly: 16+8+4+2+1+1 = 32 cross-lane moves
// instead of 27 x 6 = 162; afterwards lane l holds the total of value l >> 1 (values 27..31 are padding).  The selects are
// on bit patterns so that the array stays in registers (see block_red.h).
__device__ __forceinline__ double wave_sum27(double* acc /* [32], entries 27..31 zero */, int lane) {
#pragma unroll
  for (int c = 16, off = 32; c >= 1; c >>= 1, off >>= 1) {
    if ((lane & off) != 0) {   // (round 6: the upper lanes swap their halves, see row2_halve)
#pragma unroll
      for (int k = 0; k < c; k++) { asm volatile("" : "+v"(acc[k]), "+v"(acc[k + c])); const double tmp = acc[k]; acc[k] = acc[k + c]; acc[k + c] = tmp; }
    }
#pragma unroll
    for (int k = 0; k < c; k++) acc[k] = acc[k] + lanex::from_partner_c(acc[k + c], off);
  }
  return acc[0] + lanex::from_partner<1>(acc[0]);
}
// value index k (0..20) of the packed upper triangle -> (a, b), a <= b
__device__ __forceinline__ void tri_ab(int k, int& a, int& b) {
  a = 0;
  int first = 0;                       // index of (a, a)
  while (k >= first + 6 - a) { first += 6 - a; a++; }
  b = a + (k - first);
}

// linearisation: camera side (one wave per free camera)                [CCM_K_BA_CAM]
__global__ __launch_bounds__(kTPB) void ba_linearize_cams(BaDev d, int cur) {
  const int lane = threadIdx.x & (kWave - 1);
  const int i = blockIdx.x * (kTPB / kWave) + threadIdx.x / kWave;
  if (i >= d.Cp) return;
  const int c = d.slot_cam[i];
  const BaPose T = ba_load_pose(d.cam[cur] + 7 * (size_t)c);
  const double K4[4] = {d.K[4 * c], d.K[4 * c + 1], d.K[4 * c + 2], d.K[4 * c + 3]};
  if (d.camRK && lane < 12) {   // rotation matrix + focal lengths of the camera by pose slot: what ba_schur_row3 combines an observation's compact record with
    double Rm[9];
    ba_q_to_R(T, Rm);
    double v = lane == 9 ? K4[0] : lane == 10 ? K4[1] : 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) v = lane == k ? Rm[k] : v;
    d.camRK[12 * (size_t)i + lane] = v;
  }
  double acc[32];   // 27 sums + padding for the halving butterfly
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = 0;
  for (int s = d.cam_off[i] + lane; s < d.cam_off[i + 1]; s += kWave) {
    // (round 4) landmark, observation and information come from the camera-major copies (cam_pt, cam_oi: contiguous per camera) instead of three gathers by
    // edge index; only the landmark position is still gathered
    const int l = d.cam_pt[s];
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d oi0 = reinterpret_cast<const v2d*>(d.cam_oi)[2 * (size_t)s], oi1 = reinterpret_cast<const v2d*>(d.cam_oi)[2 * (size_t)s + 1];
    const double X[3] = {d.pt[cur][3 * l], d.pt[cur][3 * l + 1], d.pt[cur][3 * l + 2]};
    double r0, r1;
    ba_residual(T, K4, X, oi0[0], oi0[1], r0, r1);
    double Xc[3];
    ba_map(T, X, Xc);
    double Jj[12];
    const double iz = 1.0 / Xc[2];
    {
      const double x = Xc[0], y = Xc[1], iz2 = iz * iz, fx = K4[0], fy = K4[1];
      Jj[0] = x * y * iz2 * fx; Jj[1] = -(1 + (x * x * iz2)) * fx; Jj[2] = y * iz * fx; Jj[3] = -iz * fx; Jj[4] = 0; Jj[5] = x * iz2 * fx;
      Jj[6] = (1 + y * y * iz2) * fy; Jj[7] = -x * y * iz2 * fy; Jj[8] = -x * iz * fy; Jj[9] = 0; Jj[10] = -iz * fy; Jj[11] = y * iz2 * fy;
    }
    const double om = oi1[0];
    const bool off = om == 0.0;   // (see ba_linearize_pts_e)
    double rho0, w;
    ba_huber(off ? 0.0 : (r0 * r0 + r1 * r1) * om, d.huber, rho0, w);
    const double o0 = off ? 0.0 : -om * r0 * w, o1 = off ? 0.0 : -om * r1 * w, wom = off ? 0.0 : w * om;
    if (off) {
      Xc[0] = Xc[1] = 0.0;
#pragma unroll
      for (int k = 0; k < 12; k++) Jj[k] = 0.0;
    }
    if (d.E4) {   // what ba_schur_row3 re-derives this observation's Hpl block from (the values the landmark-side kernel computes for the same observation)
      v2d* E = reinterpret_cast<v2d*>(d.E4 + 4 * (size_t)s);
      v2d ea, eb; ea[0] = Xc[0]; ea[1] = Xc[1]; eb[0] = off ? 0.0 : iz; eb[1] = wom;
      E[0] = ea; E[1] = eb;
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) acc[k++] += (Jj[a] * Jj[b] + Jj[6 + a] * Jj[6 + b]) * wom;
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += Jj[a] * o0 + Jj[6 + a] * o1;
  }
  const double tot = wave_sum27(acc, lane);
  if ((lane & 1) == 0) {
    const int k = lane >> 1;
    if (k < 21) { int a, b; tri_ab(k, a, b); double* H = d.Hpp + 36 * (size_t)i; H[a * 6 + b] = tot; H[b * 6 + a] = tot; }
    else if (k < 27) d.bp[6 * (size_t)i + k - 21] = tot;
  }
}

// max |diag| of Hpp (after it has been summed over ranks) and of the own Hll  -> scal[4]
// stage 1: grid-stride partial maxima (one per workgroup) ; stage 2 (final != 0): max of the partials
__global__ __launch_bounds__(kTPB) void ba_maxdiag(BaDev d, const double* hpp_full, double* partial, int n_partial, int final) {
  __shared__ double lds[kTPB];
  double m = 0;
  if (!final) {
    const int stride = gridDim.x * kTPB;
    for (int i = blockIdx.x * kTPB + threadIdx.x; i < d.Cp * 6; i += stride) m = fmax(m, fabs(hpp_full[36 * (size_t)(i / 6) + (i % 6) * 7]));
    for (int i = blockIdx.x * kTPB + threadIdx.x; i < d.Lloc * 3; i += stride) {
      const int l = i / 3, k = i % 3;
      m = fmax(m, fabs(d.Hll[6 * (size_t)l + (k == 0 ? 0 : (k == 1 ? 3 : 5))]));
    }
  } else {
    for (int i = threadIdx.x; i < n_partial; i += kTPB) m = fmax(m, partial[i]);
  }
  lds[threadIdx.x] = m;
  __syncthreads();
  for (int s = kTPB / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) lds[threadIdx.x] = fmax(lds[threadIdx.x], lds[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) { if (final) d.scal[4] = lds[0]; else partial[blockIdx.x] = lds[0]; }
}

// ---------------------------------------------------------------------------------------------
// per LM trial
// ---------------------------------------------------------------------------------------------
// D = Hll + lambda I ; Dinv ; dl = Dinv b_l           (block_solver.hpp:385-396)   [CCM_K_BA_DINV]
__global__ __launch_bounds__(kTPB) void ba_dinv(BaDev d, double lambda) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= d.Lloc) return;
  double a[6], r[6];
#pragma unroll
  for (int i = 0; i < 6; i++) a[i] = d.Hll[6 * (size_t)l + i];
  a[0] += lambda; a[3] += lambda; a[5] += lambda;
  ba_sym3_inv(a, r);
#pragma unroll
  for (int i = 0; i < 6; i++) d.Dinv[6 * (size_t)l + i] = r[i];
  const double b0 = d.bl[3 * (size_t)l], b1 = d.bl[3 * (size_t)l + 1], b2 = d.bl[3 * (size_t)l + 2];
  d.dl[3 * (size_t)l] = r[0] * b0 + r[1] * b1 + r[2] * b2;
  d.dl[3 * (size_t)l + 1] = r[1] * b0 + r[3] * b1 + r[4] * b2;
  d.dl[3 * (size_t)l + 2] = r[2] * b0 + r[4] * b1 + r[5] * b2;
}

// diagonal Schur blocks and b_schur (one wave per free camera)          [CCM_K_BA_SCHUR_DIAG]
//   S_ii = Hpp_i - sum_e W_e Dinv W_e^T ;  bs_i = bp_i - sum_e W_e dl      (block_solver.hpp:398-439)
__global__ __launch_bounds__(kTPB) void ba_schur_diag(BaDev d) {
  const int lane = threadIdx.x & (kWave - 1);
  const int i = blockIdx.x * (kTPB / kWave) + threadIdx.x / kWave;
  if (i >= d.Cp) return;
  double acc[32];   // 27 sums + padding for the halving butterfly
#pragma unroll
  for (int k = 0; k < 32; k++) acc[k] = 0;
  for (int s = d.cam_off[i] + lane; s < d.cam_off[i + 1]; s += kWave) {
    const int e = d.cam_edge[s];
    const int l = d.ed_pt[e];
    double W[18], Di[6], dl[3];
    const double* Wp = d.W + 18 * (size_t)e;
#pragma unroll
    for (int k = 0; k < 18; k++) W[k] = Wp[k];
#pragma unroll
    for (int k = 0; k < 6; k++) Di[k] = d.Dinv[6 * (size_t)l + k];
    dl[0] = d.dl[3 * (size_t)l]; dl[1] = d.dl[3 * (size_t)l + 1]; dl[2] = d.dl[3 * (size_t)l + 2];
    double Y[18];   // W * Dinv
#pragma unroll
    for (int r = 0; r < 6; r++) {
      Y[r * 3 + 0] = W[r * 3] * Di[0] + W[r * 3 + 1] * Di[1] + W[r * 3 + 2] * Di[2];
      Y[r * 3 + 1] = W[r * 3] * Di[1] + W[r * 3 + 1] * Di[3] + W[r * 3 + 2] * Di[4];
      Y[r * 3 + 2] = W[r * 3] * Di[2] + W[r * 3 + 1] * Di[4] + W[r * 3 + 2] * Di[5];
    }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = a; b < 6; b++) { acc[k] += Y[a * 3] * W[b * 3] + Y[a * 3 + 1] * W[b * 3 + 1] + Y[a * 3 + 2] * W[b * 3 + 2]; k++; }
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += W[a * 3] * dl[0] + W[a * 3 + 1] * dl[1] + W[a * 3 + 2] * dl[2];
  }
  const double tot = wave_sum27(acc, lane);
  if ((lane & 1) == 0) {
    const int k = lane >> 1;
    if (k < 21) {
      int a, b; tri_ab(k, a, b);
      const double v = d.Hpp[36 * (size_t)i + a * 6 + b] - tot;
      double* S = d.S + 36 * (size_t)i;
      S[a * 6 + b] = v; S[b * 6 + a] = v;
    } else if (k < 27) d.bs[6 * (size_t)i + k - 21] = d.bp[6 * (size_t)i + k - 21] - tot;
  }
}

// off-diagonal Schur blocks: one workgroup per block, lane = element (r,c), the block's pair instances are
// strided over the 4 waves and summed through LDS in a fixed order            [CCM_K_BA_SCHUR_OFF]
//   S_ij = - sum_inst W_a Dinv W_c^T
// WPB = waves cooperating on one block: 4 when there are few blocks (local BA: latency bound, more parallel
// instance streams), 1 for large maps (bandwidth bound; 4 blocks per workgroup)
template <int WPB>
__global__ __launch_bounds__(kTPB) void ba_schur_off(BaDev d) {
  __shared__ double part[kTPB / kWave][36];
  const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  if (WPB == 1) {
    const int b1 = blockIdx.x * (kTPB / kWave) + wv;   // wave-uniform: instance lists arrive through scalar loads
    if (b1 >= d.nOff || lane >= 36) return;
    const int r1 = lane / 6, c1 = lane % 6;
    double acc1 = 0;
    for (int s = d.inst_off[b1]; s < d.inst_off[b1 + 1]; s++) {
      const int ea = d.inst_a[s], ec = d.inst_c[s];
      const double* Wa = d.W + 18 * (size_t)ea + 3 * r1;
      const double* Wc = d.W + 18 * (size_t)ec + 3 * c1;
      const double* Di = d.Dinv + 6 * (size_t)d.ed_pt[ea];
      const double a0 = Wa[0], a1 = Wa[1], a2 = Wa[2];
      const double y0 = a0 * Di[0] + a1 * Di[1] + a2 * Di[2];
      const double y1 = a0 * Di[1] + a1 * Di[3] + a2 * Di[4];
      const double y2 = a0 * Di[2] + a1 * Di[4] + a2 * Di[5];
      acc1 += y0 * Wc[0] + y1 * Wc[1] + y2 * Wc[2];
    }
    d.S[36 * (size_t)(d.Cp + b1) + lane] = -acc1;
    return;
  }
  const int b = blockIdx.x;
  const int r = lane / 6, c = lane % 6;
  double acc = 0;
  if (lane < 36) {
    const int s0 = d.inst_off[b], s1 = d.inst_off[b + 1];
    for (int s = s0 + wv; s < s1; s += kTPB / kWave) {
      const int ea = d.inst_a[s], ec = d.inst_c[s];
      const int l = d.ed_pt[ea];
      const double* Wa = d.W + 18 * (size_t)ea + 3 * r;
      const double* Wc = d.W + 18 * (size_t)ec + 3 * c;
      const double* Di = d.Dinv + 6 * (size_t)l;
      const double a0 = Wa[0], a1 = Wa[1], a2 = Wa[2];
      const double y0 = a0 * Di[0] + a1 * Di[1] + a2 * Di[2];
      const double y1 = a0 * Di[1] + a1 * Di[3] + a2 * Di[4];
      const double y2 = a0 * Di[2] + a1 * Di[4] + a2 * Di[5];
      acc += y0 * Wc[0] + y1 * Wc[1] + y2 * Wc[2];
    }
    part[wv][lane] = acc;
  }
  __syncthreads();
  if (wv == 0 && lane < 36) d.S[36 * (size_t)(d.Cp + b) + lane] = -(((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
}

// ---- row Schur kernel: off-diagonal AND diagonal blocks of one camera row per workgroup ------------------------------------------------------------
// One workgroup (16 waves) per free camera i.  Y_e = W_e D^-1 (6x3) of ALL observations of camera i is formed once into LDS, then every pair instance
// (observation a of camera i, observation c of camera j > i, same landmark) contributes Y_a W_c^T to S_ij.                  [CCM_K_BA_SCHUR_OFF]
// Round 2 fed pairs of instances to v_mfma_f64_16x16x4 (14 % of the tile used, 64 cycles of the matrix pipe per pair, two 8-byte gathers per lane and
// instruction): 223 us on the 4-agent map.  This form gives an instance to a LANE: the lane fetches the whole W_c row (144 B, nine 16-byte loads), reads
// the Y_a row out of LDS and multiplies on the vector ALUs (108 f64 FMAs).  The sum over the instances of a block is made cheap by giving every work
// unit (<= kRow2Chunk consecutive instances of one block) only kRow2Group = 16 lanes, each accumulating its instances serially, and by reducing the 36
// sums of the FOUR units of a wave pass together with one halving butterfly (xor 8, 4, 2, 1: every step exchanges half of the values a lane still
// holds: 18 + 9 + 5 + 3 exchanges instead of 4 x 36), after which each lane holds <= 3 finished elements.  Summation order: lane-serial, then the fixed
// tree — deterministic, identical on every rank.
// The diagonal block needs no pass of its own: the thread that stages Y_e still holds W_e, so it forms Y_e W_e^T (symmetric: 21 entries) and Y_e b_l
// (6) on the spot; the same butterfly leaves one partial per 16 observations in LDS.
// Phase clocks (CCM_BA_ROW_DBG, 4-agent map, us per row, 7.8 rows per CU): the kernel is a chain of dependent memory round trips, not of flops or
// bytes — staging 4.9 (index -> W, D^-1), block passes 6.6 (table entry -> index vectors -> W_c rows), final sums 2.9; a first version with separate
// own-observation passes spent 10.7 us per row on their four-deep chain (table -> cam_edge -> ed_pt -> b_l).
// Measured and dropped (round 3): an LDS-free "flat" form — a 16-lane group per S block recomputing Y_a = W_a D^-1 from global memory (21 sixteen-byte
// gathers per instance and lane), any number of waves per CU — 393 us against 197 us: a wave-wide 16-byte gather whose lanes address 64 different rows
// costs one L1 line look-up per LANE (~64 cycles per instruction), so the gathers alone were ~140 us; the same holds for the nine W_c gathers per
// instance here (~60 us of this kernel).  What would lift it: nine consecutive lanes fetching one 144-byte row (2.25 line look-ups) and handing it to the
// computing lane through LDS — for which the 70-100 KB of Y leave no room.
// (round 6) One step of the halving butterfly without LDS and without selects.  Until now a step cost, per exchanged value, four v_cndmask (which half a lane keeps, which it
// sends), two ds_bpermute and the addition — 476 instructions per pass for its 68 exchanges, as many as three iterations of the multiplication loop, with a chain of LDS
// latencies in it.  Now the upper lanes SWAP their two halves first (v_swap_b32 under their own exec mask: two per value), after which every lane keeps [0, H) and sends
// [H, N); the partner's value arrives through DPP (lane_xor.h: row_ror:8 / row_half_mirror + quad_perm / quad_perm: vector-ALU moves, nothing goes through LDS).  The sums are the same
// sums (keep + received, pair by pair): same bits.
template <int N, int MASK>
__device__ __forceinline__ void row2_halve(const double (&in)[N], double (&out)[(N + 1) / 2], bool hi) {
  constexpr int H = (N + 1) / 2;   // the lower lanes keep elements [0, H), the upper ones [H, N) (N - H <= H of them, padded with zeros)
  double v[2 * H];
#pragma unroll
  for (int k = 0; k < 2 * H; k++) v[k] = k < N ? in[k] : 0.0;
  if (hi) {   // (a real branch: the empty asm keeps the compiler from turning the swaps back into selects)
#pragma unroll
    for (int k = 0; k < H; k++) { asm volatile("" : "+v"(v[k]), "+v"(v[H + k])); const double t = v[k]; v[k] = v[H + k]; v[H + k] = t; }
  }
#pragma unroll
  for (int k = 0; k < H; k++) out[k] = v[k] + lanex::from_partner<MASK>(v[H + k]);
}
// element range [*e0, *e0 + *cnt) that a lane holds after the four steps (lane bits q3..q0 of its position inside the group)
__device__ __forceinline__ void row2_range(int N, int q, int* e0, int* cnt) {
  int off = 0, cap = N, n = N;   // cap: the (compile-time) array length of the step, n: how many of its entries are real sums (the rest is zero padding)
#pragma unroll
  for (int bit = 8; bit >= 1; bit >>= 1) {
    const int H = (cap + 1) / 2;
    if (q & bit) { off += H; n = max(0, n - H); } else n = min(n, H);
    cap = H;
  }
  *e0 = off; *cnt = n;
}

// (round 5: ba_schur_row2 — this kernel on the STORED 144-byte Hpl blocks, nine divergent 16-byte loads per lane and instance — is gone from the library: 211 us per launch
// against 131 for ba_schur_row3 below on the 4-agent map; DESIGN 4.1 keeps the measurements.  Its reduction helpers above serve row3.)

// Factors of an observation's Hpl block W = Jj^T (wom Ji) from its compact record (x, y, 1 / z | wom) and its camera's record (rotation matrix rows
// R_0 R_1 R_2, fx, fy), W itself never formed.  With a = x / z, b = y / z:
//   wom Ji = -(wom / z) [fx (R_0 - a R_2) ; fy (R_1 - b R_2)]                                                  -> wj0[3], wj1[3]
//   Jj     = [fx (a b, -(1 + a^2), b, -1/z, 0, a / z) ; fy (1 + b^2, -a b, -a, 0, -1/z, b / z)]                 -> pj[5] (column 4 is zero), qj[5] (column 3 is zero)
// (types_six_dof_expmap.cpp:196-226 regrouped: products formed from these differ from those of the stored block in the last bits only)
typedef double ba_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ba_compact_factors(ba_v2d ea, ba_v2d eb, const ba_v2d (&r2)[6], double (&wj0)[3], double (&wj1)[3], double (&pj)[5], double (&qj)[5]) {
  const double iz = eb[0], fx = r2[4][1], fy = r2[5][0];
  const double a = ea[0] * iz, b = ea[1] * iz;
  const double gx = -(iz * fx) * eb[1], gy = -(iz * fy) * eb[1];
  const double R0[3] = {r2[0][0], r2[0][1], r2[1][0]}, R1[3] = {r2[1][1], r2[2][0], r2[2][1]}, R2[3] = {r2[3][0], r2[3][1], r2[4][0]};
#pragma unroll
  for (int c = 0; c < 3; c++) { wj0[c] = gx * __builtin_fma(-a, R2[c], R0[c]); wj1[c] = gy * __builtin_fma(-b, R2[c], R1[c]); }
  const double fxa = fx * a, fyb = fy * b;
  pj[0] = fxa * b; pj[1] = -__builtin_fma(fxa, a, fx); pj[2] = fx * b; pj[3] = -(fx * iz); pj[4] = fxa * iz;
  qj[0] = __builtin_fma(fyb, b, fy); qj[1] = -(fyb * a); qj[2] = -(fy * a); qj[3] = -(fy * iz); qj[4] = fyb * iz;
}

// The same row kernel on COMPACT observation records: an observation's Hpl block W = Jj^T (w Omega) Ji is a function of the landmark in the camera frame
// (x, y, 1 / z), the weight and the camera's rotation and focal lengths, so a pair instance fetches 32 bytes (E4, camera-major) + the column camera's
// 96 bytes (camRK, shared by the unit's 16 lanes) instead of 144 divergent bytes of the stored block.  ba_schur_row2 was bound by the address
// processing of those nine divergent 16-byte loads per lane (one line look-up per lane and load: 3.8 us per 1024-lane pass on a CU); here it is two.  The
// camera's own observations are read in camera-major order (contiguous) and re-derive exactly the stored block for Y and the diagonal block.
// (round 6) Per-wave clocks (s_memrealtime in scalar registers; a 4-agent-map row, 7.8 rows per CU): a wave's pass costs ~2 us (table entry -> index vectors -> records ->
// butterfly -> LDS) + ~1.4 us per iteration WHATEVER the other waves do (1.0 of it the 150 f64 instructions and their 17 LDS reads: a lone wave issues one dependent f64
// instruction every ~14 cycles), and the units being dealt longest first, waves 0..5 run ~4 iterations each while waves 8..15 run < 1: the pass phase is the serial time of
// wave 0, with the vector ALUs a third busy.  The same-bits rule (a lane's serial sums, the fixed butterfly, the unit order) rules out re-cutting the units, so the roles
// were swapped instead: the LAST waves stage the observations, the first ones — which stage nothing on this map — request their first index vectors, records and column
// camera before the barrier, and the serial sum of the diagonal partials went from the tail of the kernel to the last wave's idle time inside the pass phase:
// 138-141 -> 127-130 us on one box.  Built, measured on the same box and dropped in this round:
//   * the next iteration's records by LDS-DMA (global_load_lds_dwordx4 into a per-wave buffer, column camera staged in LDS, index two iterations ahead): the loop of wave 0
//     9.45 -> 7.22 us per row, the kernel 138 -> 146 us — an LDS-DMA instruction costs 60-185 cycles of issue (MI355X_MICROARCH.md) and a pass needs 6 + 2 per iteration;
//     tried again on the form below for the second and later iterations only (their first records requested with the first iteration's, before the barrier): 134-136 us
//     against 130-134;
//   * a persistent, software-pipelined form (a workgroup per CU walking its rows; the next row's scalars in LDS, its records / landmark gather / unit ranges / Hpp requested
//     under the tail of the current row, raw s_barrier so that the prefetches stay in flight): 139 us against 142 — the round trips it hides were not the row's critical
//     path (that is wave 0's serial iterations), and every value carried around the row loop competes with the 36 accumulators: its first versions spilled 36-90 registers
//     and ran 182-208 us;
//   * a unit that is its whole block storing straight into S (85 % of the blocks), the second block range of the final sums requested before the barrier: +-1 us.
//   * (after the butterfly went to DPP, below) the upper lanes of a unit accumulating the block's rows rotated by three, so that the butterfly's first step needs no exchange
//     of halves (54 moves fewer per pass, two more address selects per iteration): 106-108 us against 106-107; the halves swapped by v_swap_b32 instead of three moves:
//     110-112 us;
//   * no second barrier and no final-sum phase at all — the unit that delivers a block's LAST partial sums (an LDS counter per block) adds them up in the units' order and
//     stores the block, the waves end one by one: 128.6 us against 127-130, i.e. the final sums were not on the row's critical path either.
__global__ __launch_bounds__(kRow2TPB) void ba_schur_row3(BaDev d) {
  extern __shared__ __attribute__((aligned(16))) double Ys[];
  typedef double v2d __attribute__((ext_vector_type(2)));
  constexpr int G = kRow2Group, UPW = kWave / G, NW = kRow2TPB / kWave;
  const int per_xcd = gridDim.x >> 3;      // workgroup b runs on XCD b % 8: XCD x walks the contiguous row range [x * per, (x + 1) * per) (rows in flight share W_c rows)
  const int i = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (i >= d.Cp) return;
  const int base = d.cam_off[i], ne = d.cam_off[i + 1] - base;   // ne <= kRowMaxEdges < kRow2TPB: one observation per thread
  // phase clocks: compiled in only with -DCCM_BA_ROW_DBG_BUILD (round 4: as a run-time option their 64-bit time stamp stayed live through the whole kernel and
  // was one of the values the register allocator spilled to scratch memory)
#ifdef CCM_BA_ROW_DBG_BUILD
  long long tk0 = 0;
  if (d.row_dbg && threadIdx.x == 0) tk0 = wall_clock64();
#define ROW3_TICK(slot) { if (d.row_dbg && threadIdx.x == 0) { const long long tn_ = wall_clock64(); atomicAdd((unsigned long long*)(d.row_dbg + slot), (unsigned long long)(tn_ - tk0)); tk0 = tn_; } }
#else
#define ROW3_TICK(slot) {}
#endif
  const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int grp = lane / G, q = lane % G;
  const int zrow = d.max_cam_edges;        // a zero row of Y behind the real ones: what the idle lanes of a unit multiply
  const int n_dgrp = (ne + G - 1) / G;     // 16-observation groups of the diagonal partials
  double* dpart = Ys + 18 * (size_t)(zrow + 1);                           // [ceil(max_cam_edges / 16)][27]
  double* part = dpart + 27 * (size_t)((d.max_cam_edges + G - 1) / G);    // [row_units_max][36]
  // the table entry and the index vectors of this wave's first block pass are requested before anything else: their round trips overlap the staging.
  // (Measured and dropped: requesting the observation records first and the table entry behind them, with the loads made unconditional so that no
  // branch end waits for them, and [D^-1 | b_l] as one 80-byte record per landmark: 5.8-6.0 us of staging per row instead of 4.9.)
  const int u_first = d.row_unit_off[i], n_units = d.row_unit_off[i + 1] - u_first;
  int n_s0 = 0, n_s1 = 0, n_slot = 0, n_j = 0;
  if (wv * UPW + grp < n_units) { const int4 te = d.unit_tab[u_first + wv * UPW + grp]; n_s0 = te.y; n_s1 = te.z; n_slot = te.w; n_j = d.blk_j[te.x]; }
  // ---- Y_e = W_e D^-1 of the camera's observations, one thread per observation; the same thread adds the observation's part of the diagonal block
  //      (Y_e W_e^T, symmetric: entries r <= c) and of b_schur (Y_e b_l) ----
  // (round 6) first-iteration operands of the wave's first pass: index vectors, the partners' records, the column camera — requested BEFORE the barrier by the waves that
  // stage nothing (the two uses exclude each other: what is requested here is not alive across the staging code, which has no register to spare)
  int f_ce1, f_ar0, f_ar1;
  v2d f_ea, f_eb, f_r2[6];
  auto preload_none = [&]() {   // (assigned at the END of the paths that request nothing: constants set ahead of the staging code would sit in registers all through it)
    f_ce1 = 0; f_ar0 = zrow; f_ar1 = zrow;
    f_ea[0] = f_ea[1] = f_eb[0] = f_eb[1] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) { f_r2[k][0] = 0.0; f_r2[k][1] = 0.0; }
  };
  auto preload = [&](int s0, int s1, int jc) {
    const int ce0 = (s0 + q < s1) ? d.inst_cp[s0 + q] : 0;
    f_ar0 = (s0 + q < s1) ? d.inst_al[s0 + q] : zrow;
    f_ce1 = (s0 + G + q < s1) ? d.inst_cp[s0 + G + q] : 0;
    f_ar1 = (s0 + G + q < s1) ? d.inst_al[s0 + G + q] : zrow;
    const v2d* rk = reinterpret_cast<const v2d*>(d.camRK + 12 * (size_t)jc);
#pragma unroll
    for (int k = 0; k < 6; k++) f_r2[k] = rk[k];
    const v2d* Ep = reinterpret_cast<const v2d*>(d.E4 + 4 * (size_t)ce0);
    f_ea = Ep[0]; f_eb = Ep[1];
  };
  const bool stager = (NW - 1 - wv) * (kWave / G) < n_dgrp;   // (wave-uniform) this wave holds observations
  const bool early = !stager && wv * UPW < n_units;
  if (!stager) { if (early) preload(n_s0, n_s1, n_j); else preload_none(); }
  else {
    // (round 6) the observations are staged by the LAST waves (observation = 1023 - thread): the first waves hold the row's longest units (the table lists them longest
    // first) and have their first records on the way while the others stage; the butterfly below pairs the same observations (15 - q ^ 8 = 15 - (q ^ 8)): same sums
    const int t = kRow2TPB - 1 - (int)threadIdx.x;
    double dacc[27];
#pragma unroll
    for (int k = 0; k < 27; k++) dacc[k] = 0.0;
    if (t < ne) {
      const int pt = d.cam_pt[base + t];
      const v2d* Ep = reinterpret_cast<const v2d*>(d.E4 + 4 * (size_t)(base + t));   // camera-major: the workgroup reads one contiguous range
      const v2d* Dp = reinterpret_cast<const v2d*>(d.Dinv + 6 * (size_t)pt);
      const v2d ea = Ep[0], eb = Ep[1];
      v2d d2[3];
#pragma unroll
      for (int k = 0; k < 3; k++) d2[k] = Dp[k];
      const double bl0 = d.bl[3 * (size_t)pt], bl1 = d.bl[3 * (size_t)pt + 1], bl2 = d.bl[3 * (size_t)pt + 2];
      double wf[18], yf[18];
      {   // W_e as the linearisation formed it (same expressions, no contraction: bit-identical to the stored block)
        const double* rk = d.camRK + 12 * (size_t)i;
        double Rm[9], Ji[6], Jj[12];
#pragma unroll
        for (int k = 0; k < 9; k++) Rm[k] = rk[k];
        ba_jac_from_xc(ea[0], ea[1], eb[0], rk[9], rk[10], Rm, Ji, Jj);
        const double wom = eb[1];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int b = 0; b < 3; b++) wf[a * 3 + b] = (Jj[a] * Ji[b] + Jj[6 + a] * Ji[3 + b]) * wom;
      }
      const double D0 = d2[0][0], D1 = d2[0][1], D2 = d2[1][0], D3 = d2[1][1], D4 = d2[2][0], D5 = d2[2][1];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        const double a0 = wf[3 * r], a1 = wf[3 * r + 1], a2 = wf[3 * r + 2];
        yf[3 * r + 0] = a0 * D0 + a1 * D1 + a2 * D2;
        yf[3 * r + 1] = a0 * D1 + a1 * D3 + a2 * D4;
        yf[3 * r + 2] = a0 * D2 + a1 * D4 + a2 * D5;
      }
      v2d* Yp = reinterpret_cast<v2d*>(Ys + 18 * (size_t)t);
#pragma unroll
      for (int k = 0; k < 9; k++) { v2d v; v[0] = yf[2 * k]; v[1] = yf[2 * k + 1]; Yp[k] = v; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row is read back three values at a time below (same thread): Y, W_e and the 27 sums do not fit 128 registers together
      const double* Yr = Ys + 18 * (size_t)t;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        constexpr int kTri[6] = {0, 5, 9, 12, 14, 15};   // compact index of (r, c), r <= c: kTri[r] + c
        const double y0 = Yr[3 * r], y1 = Yr[3 * r + 1], y2 = Yr[3 * r + 2];
#pragma unroll
        for (int c = r; c < 6; c++)
          dacc[kTri[r] + c] = __builtin_fma(y2, wf[3 * c + 2], __builtin_fma(y1, wf[3 * c + 1], y0 * wf[3 * c]));
        dacc[21 + r] = __builtin_fma(y2, bl2, __builtin_fma(y1, bl1, y0 * bl0));
      }
    }
    {   // 27 sums per 16-lane group, <= 2 elements per lane afterwards
      double t1[14], t2[7], t3[4], t4[2];
      row2_halve<27, 8>(dacc, t1, (q & 8) != 0);
      row2_halve<14, 4>(t1, t2, (q & 4) != 0);
      row2_halve<7, 2>(t2, t3, (q & 2) != 0);
      row2_halve<4, 1>(t3, t4, (q & 1) != 0);
      int e0, cnt;
      row2_range(27, q, &e0, &cnt);
      const int g = t / G;
      if (g < n_dgrp) {
#pragma unroll
        for (int k = 0; k < 2; k++) if (k < cnt) dpart[27 * (size_t)g + e0 + k] = t4[k];
      }
    }
    preload_none();
  }
  if (threadIdx.x < 18) Ys[18 * (size_t)zrow + threadIdx.x] = 0.0;
  __syncthreads();
  ROW3_TICK(0)
  // ---- off-diagonal blocks: units of <= kRow2Chunk pair instances, longest first, four per wave pass ----
  for (int p = wv; p * UPW < n_units; p += NW) {
    const int uu = p * UPW + grp;
    // (round 4) the table entry of a LATER pass is loaded when that pass begins: prefetched one pass ahead, its four values stayed live across the inner loop,
    // were spilled to scratch memory there (16 bytes per lane and pass, 12 more per lane around the loop) and reloaded before the butterfly — rocprofv3 counted
    // 59.5 MB written per launch for 18.9 MB of S blocks (WRITE_SIZE is exact for this store pattern: scripts/write_size_probe.hip), the rest was scratch.
    // A row of the 4-agent map has ~57 units, i.e. ONE pass per wave: the prefetch bought nothing there.
    if (p != wv) {
      n_s0 = n_s1 = n_slot = n_j = 0;
      if (p * UPW + grp < n_units) { const int4 te = d.unit_tab[u_first + p * UPW + grp]; n_s0 = te.y; n_s1 = te.z; n_slot = te.w; n_j = d.blk_j[te.x]; }
    }
    if (p != wv || !early) preload(n_s0, n_s1, n_j);
    const int s0 = n_s0, s1 = n_s1, slot = n_slot, jc = n_j;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0.0;
    // the table lists the units longest first, so group 0 of the pass sets the trip count of the wave
    const int nit = __builtin_amdgcn_readfirstlane((s1 - s0 + G - 1) / G);
    // Y_a W_c^T with W_c = Jj^T (wom Ji) never formed: V = Y_a (wom Ji)^T is 6 x 2, then V Jj (two structural zeros): 36 + 60 multiply-adds
    auto multiply = [&](v2d ea, v2d eb, const v2d (&r2)[6], int ar) {
      const double* Yp = Ys + 18 * (size_t)ar;
      double wj0[3], wj1[3], pj[5], qj[5];
      ba_compact_factors(ea, eb, r2, wj0, wj1, pj, qj);
#pragma unroll
      for (int r = 0; r < 6; r++) {   // the Y row comes out of LDS three values at a time: 16 waves leave 128 registers per lane
        const double y0 = Yp[3 * r], y1 = Yp[3 * r + 1], y2 = Yp[3 * r + 2];
        const double v0 = __builtin_fma(y2, wj0[2], __builtin_fma(y1, wj0[1], y0 * wj0[0]));
        const double v1 = __builtin_fma(y2, wj1[2], __builtin_fma(y1, wj1[1], y0 * wj1[0]));
        acc[6 * r + 0] = __builtin_fma(v1, qj[0], __builtin_fma(v0, pj[0], acc[6 * r + 0]));
        acc[6 * r + 1] = __builtin_fma(v1, qj[1], __builtin_fma(v0, pj[1], acc[6 * r + 1]));
        acc[6 * r + 2] = __builtin_fma(v1, qj[2], __builtin_fma(v0, pj[2], acc[6 * r + 2]));
        acc[6 * r + 3] = __builtin_fma(v0, pj[3], acc[6 * r + 3]);
        acc[6 * r + 4] = __builtin_fma(v1, qj[3], acc[6 * r + 4]);
        acc[6 * r + 5] = __builtin_fma(v1, qj[4], __builtin_fma(v0, pj[4], acc[6 * r + 5]));
      }
    };
    // the first iteration stands in front of the loop: its operands are there (requested before the barrier, or at the head of the pass) and are dead before the loop begins
    multiply(f_ea, f_eb, f_r2, f_ar0);
    int ce_n = f_ce1, ar_n = f_ar1;                            // index vectors one iteration ahead of the records they address
    preload_none();   // (what flows around the pass loop is constants: the operands themselves must not stay alive through the loop below)
    int jq = jc;
    for (int it = 1; it < nit; it++) {
      const int ce = ce_n, ar = ar_n;
      // the block's column camera: the same 96 bytes for the 16 lanes of the unit.  Requested every iteration, first (the index is laundered: left alone,
      // the compiler hoists these loop-invariant loads out of the loop, has no registers for their 24 values and reloads them from scratch memory — extra
      // dependent L2 round trips in every iteration)
      asm volatile("" : "+v"(jq));
      const v2d* rk = reinterpret_cast<const v2d*>(d.camRK + 12 * (size_t)jq);
      v2d r2[6];
#pragma unroll
      for (int k = 0; k < 6; k++) r2[k] = rk[k];
      const v2d* Ep = reinterpret_cast<const v2d*>(d.E4 + 4 * (size_t)ce);
      const v2d ea = Ep[0], eb = Ep[1];
      const int sn = s0 + (it + 1) * G + q;
      ce_n = (sn < s1) ? d.inst_cp[sn] : 0;
      ar_n = (sn < s1) ? d.inst_al[sn] : zrow;
      multiply(ea, eb, r2, ar);
    }
    double t1[18], t2[9], t3[5], t4[3];
    row2_halve<36, 8>(acc, t1, (q & 8) != 0);
    row2_halve<18, 4>(t1, t2, (q & 4) != 0);
    row2_halve<9, 2>(t2, t3, (q & 2) != 0);
    row2_halve<5, 1>(t3, t4, (q & 1) != 0);
    int e0, cnt;
    row2_range(36, q, &e0, &cnt);
    if (uu < n_units) {
      double* pu = part + 36 * (size_t)slot;
#pragma unroll
      for (int k = 0; k < 3; k++) if (k < cnt) pu[e0 + k] = t4[k];
    }
    n_s0 = n_s1 = n_slot = n_j = 0;   // (constants flow around the pass loop: a later pass loads its own table entry, and the first pass's must not stay alive through the loop)
  }
  // (round 6) the diagonal block and b_schur are summed HERE, by the last wave — the one with the row's shortest units, done long before the first waves —, not behind
  // the barrier, where the serial sum over the 16-observation groups (~30 dependent LDS reads) was the tail every other wave waited for.  The partial sums are complete since
  // the barrier above.  Eight reads are requested at a time, the additions keep their order.
  if (wv == NW - 1 && lane < 27) {
    const int e = lane;
    int r = 0, c = 0;
    double hv;
    if (e < 21) {
      r = e < 6 ? 0 : e < 11 ? 1 : e < 15 ? 2 : e < 18 ? 3 : e < 20 ? 4 : 5;
      c = e - (r == 0 ? 0 : r == 1 ? 5 : r == 2 ? 9 : r == 3 ? 12 : r == 4 ? 14 : 15);
      hv = d.Hpp[36 * (size_t)i + 6 * r + c];
    } else hv = d.bp[6 * (size_t)i + e - 21];
    double sum = 0;
    for (int g0 = 0; g0 < n_dgrp; g0 += 8) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (g0 + k < n_dgrp) ? dpart[27 * (size_t)(g0 + k) + e] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; k++) if (g0 + k < n_dgrp) sum += v[k];
    }
    if (e < 21) {
      const double v = hv - sum;
      d.S[36 * (size_t)i + 6 * r + c] = v; d.S[36 * (size_t)i + 6 * c + r] = v;   // the upper triangle is mirrored: S_ii is exactly symmetric
    } else d.bs[6 * (size_t)i + e - 21] = hv - sum;
  }
  ROW3_TICK(1)   // (wave 0's share of the block passes)
  // the unit range of the thread's first block of the final sums is requested BEFORE the barrier: its round trip runs while the slower waves finish
  const int grp36 = threadIdx.x / 36, el = threadIdx.x % 36;
  constexpr int kGroups = kRow2TPB / 36;
  const int fb0 = d.rowblk_off[i] + grp36, fb_end = d.rowblk_off[i + 1];
  int f_ub = 0, f_n = 0;
  if (grp36 < kGroups && fb0 < fb_end) { f_ub = d.blk_unit0[fb0]; f_n = d.inst_off[fb0 + 1] - d.inst_off[fb0]; }
  __syncthreads();
  ROW3_TICK(2)   // (waiting for the slowest wave)
  // ---- final sums: per block over its units (slot = creation order: a block's units are consecutive); diagonal block + b_schur over the 16-observation groups ----
  {
    if (grp36 < kGroups)
      for (int b = fb0; b < fb_end; b += kGroups) {
        double sum = 0;
        const int ub = (b == fb0) ? f_ub : d.blk_unit0[b];
        const int ni = (b == fb0) ? f_n : d.inst_off[b + 1] - d.inst_off[b];
        const int ue = ub + max(1, (ni + d.unit_chunk - 1) / d.unit_chunk);
        for (int u = ub; u < ue; u++) sum += part[36 * (size_t)(u - u_first) + el];
        d.S[36 * (size_t)(d.Cp + b) + el] = -sum;
      }
  }
  ROW3_TICK(3)
#ifdef CCM_BA_ROW_DBG_BUILD
  if (d.row_dbg && threadIdx.x == 0) atomicAdd((unsigned long long*)(d.row_dbg + 5), 1ull);
#endif
#undef ROW3_TICK
}

// (round 5: ba_schur_row4 — two rows per CU, 8-wave workgroups, own observations as 72 bytes — was built and measured in round 4 (146 us against 134 for row3: a CU holds
// 16 waves either way, and two half-size workgroups pay two passes per row) and is gone from the library; DESIGN 4.1 keeps the measurement.)

// ---- PCG on (S + lambda I_diag) x = bs ---------------------------------------------------------
// Preconditioner: block-Jacobi over CLUSTERS of kClu consecutive camera slots (dense 96x96 blocks).  Keyframes of
// one agent are consecutive and covisibility is mostly local in time, so a cluster captures the strong
// couplings; measured on the 300-KF / 2-agent test problem it halves the iteration count of 6x6 block-Jacobi
// (336 -> 178 at lambda = 0.3).  Per LM trial one workgroup per cluster assembles its dense block from the
// block-CSR rows, factors it (Cholesky in LDS), forms the explicit inverse W = L^-T L^-1 and stores it; per
// PCG iteration the same workgroup applies z_c = W r_c (dense 96x96 mat-vec).

// weight of camera slot k towards the SECOND node of its interval (the first one gets 1 - t): the coarse space interpolates the rigid-body twists of
// nodes placed every `agg` cameras linearly in the camera index (hat functions)
__device__ __forceinline__ double coarse_hat_t(int k, int agg) { return ((double)(k % agg) + 0.5) * (1.0 / (double)agg); }

// cluster part of the coarse restriction P^T r: 6 values for the first node of the cluster's interval, 6 for the second.  The 6 products of every
// (camera, component), weighted for either node, go through LDS and are added in a fixed order.  prod: LDS scratch of 12 * 96 doubles; needs all kTPB threads (barriers).
// (round 4: the hat weights are applied by the 96 threads that form the products; the 12 serial sums used to evaluate `k % agg` and a multiply per term: ~5 us of a 12.8 us launch)
__device__ __forceinline__ void mk_restrict(const BaDev& d, int c, int s0, int m, const double* rc, double* prod) {
  const int t = threadIdx.x;
  if (t < m) {
    const double* P = d.mk_P + 36 * (size_t)(s0 + t / 6) + 6 * (t % 6);
    const double rv = rc[t];
    const double w1 = coarse_hat_t(s0 + t / 6, d.agg), w0 = 1.0 - w1;
#pragma unroll
    for (int cc = 0; cc < 6; cc++) { const double pr = P[cc] * rv; prod[t * 6 + cc] = w0 * pr; prod[6 * kCluN + t * 6 + cc] = w1 * pr; }
  }
  __syncthreads();
  if (t < 12) {
    const int cc = t % 6, second = t / 6;
    const double* pw = prod + (second ? 6 * kCluN : 0) + cc;
    double sv = 0;
    for (int q = 0; q < m; q++) sv += pw[q * 6];
    // node-major, four contributor slots per node (first-node parts of clusters 2n / 2n + 1, second-node parts of clusters 2n - 2 / 2n - 1): the gather of
    // ba_pcg_coarse_apply is four coalesced streams; slots no cluster writes stay zero from the allocation
    d.mk_cpart[(size_t)(2 * second + (c & 1)) * (6 * (size_t)(d.mk_na + 1)) + 6 * (size_t)((c >> 1) + second) + cc] = sv;
  }
}

// (the start kernel of this path, ba_pcg_init_tiles, is defined after the cluster factorisation it shares with the persistent kernel)

// iteration k: q_k = A z_k + beta_k q_{k-1}, p_k = z_k + beta_k p_{k-1}, partial p.q   [CCM_K_BA_PCG_SPMV]
// (round 5: until then q = A (z + beta p) with the direction formed on the fly for every neighbour block — two gathered vectors, nine load requests per lane and block of which
// three were the block; A p_k = A z_k + beta A p_{k-1} is the same vector and needs ONE gathered vector and two row-local recurrences.)
// TWO waves per block row: the row product is a chain of dependent loads (index -> block -> vector), so splitting a row's blocks over two
// waves halves that chain; each wave keeps 8 blocks in flight (lane = g*8 + r: group g takes every 16th block starting at its own offset,
// r is the block row).  Workgroup = 16 waves = 8 rows: measured on the 10 000-keyframe map (rocprofv3), a launch costs ~10 us of dependent
// round trips plus ~10 ns per WORKGROUP dispatched, whatever the workgroups do (a pass that only adds six numbers per row: 20 us for 2 500
// workgroups), so 5 000 two-row workgroups spent most of the kernel being dispatched.  (Also measured and dropped: a two-pass symmetric
// form that reads every block once — upper pass 61 us, lower pass 20 us against 56 us for this kernel: the traffic was never the limit.)
// (round 4) 66 VGPRs left ONE 16-wave workgroup per CU (7 waves per SIMD); the kernel is bound by the bytes it keeps in flight (per wave 8 blocks of 288 B behind
// an index load), so it is held to 64 registers: two workgroups per CU.
// (round 5) F32: the blocks come from S32, the f32 copy of S made once per trial (ba_s_to_f32): 24 instead of 48 bytes per lane and block.  The solve stays an f64 solve:
// every kMkReplaceEvery-th iteration (and once more when the recurrence reports convergence) the residual is REPLACED by the true one, r = b - (S + lambda I) x formed with
// the f64 blocks — XMODE: the same kernel with x as the source vector, no direction update, no dot products —, so what the rounding of S to f32 puts into the recurrence
// never accumulates into the answer (offline, scripts/offline/precond_study.py f32: the same iteration counts and the same 8e-9 final error as the all-f64 solve with a
// replacement every 4 or 8 iterations; without any the error stalls at 4e-7 ... 1.6e-6).  Only maps above 2048 free cameras (S32 != nullptr).
template <bool F32, bool XMODE>
__global__ __launch_bounds__(kSpmvTPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void ba_pcg_spmv(BaDev d, int k) {
  __shared__ double half_sum[kRowsPerWG][2][8];
  __shared__ double lds[kRowsPerWG];
  __shared__ double redp[4 * (kSpmvTPB / kWave)];
  // the "done" flag is read ONCE per workgroup: workgroup 0 sets it further down in this very launch, and waves of another workgroup that read it at
  // different times would leave the block-wide sums below with missing members
  __shared__ int s_done;
  if (threadIdx.x == 0) s_done = d.pcg_flag[0];
  __syncthreads();
  if (s_done) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int rl = wv >> 1, h = wv & 1;            // local row, half
  // XCD-aware row assignment: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only), so give
  // XCD x the CONTIGUOUS row chunk x: covisible cameras are close in index, hence a block S_ij and its mirror use
  // (row i and, transposed, row j) are read by the same XCD and the second read can hit that XCD's 4 MiB L2.
  double rz_k = 0, beta = 0;
  if (!XMODE) {
    const int nco = d.mk_on ? d.n_wg_upd : 0, npr = k ? d.n_wg_upd : 0;
    const double* const ps[4] = {d.prz[k & 1], d.mk_cry[k & 1], d.prz[(k + 1) & 1], d.mk_cry[(k + 1) & 1]};
    const int ns[4] = {d.n_wg_upd, nco, npr, k ? nco : 0};
    double sm[4];
    block_sum_partials<4, kSpmvTPB>(ps, ns, sm, redp);
    rz_k = sm[0];
    if (d.mk_on) rz_k += sm[1];
    if (k == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) d.pcg_scal[0] = rz_k; }
    else {
      double rz_prev = sm[2];
      if (d.mk_on) rz_prev += sm[3];
      beta = rz_k / rz_prev;
    }
  }
  if (!XMODE) {
  const double rz0 = (k == 0) ? rz_k : d.pcg_scal[0];
  // convergence test (identical in every workgroup): sqrt(rz_k / rz_0) <= rel_tol, or exact zero residual
  if (rz_k <= d.pcg_scal[1] * rz0 || !(rz_k > 0.0)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.pcg_flag[0] = 1; d.pcg_flag[1] = k; if (rz_k != rz_k) d.pcg_flag[2] = 1; }
    return;
  }
  }
  const double lambda = to_sgpr(d.pcg_scal[2]);
  beta = to_sgpr(beta);                          // (uniform values in scalar registers: the kernel is held to 64 vector registers)
  const double* pold = d.p[k & 1];
  double* pnew = d.p[(k + 1) & 1];
  const double* zsrc = XMODE ? d.x : d.z;        // XMODE: q = (S + lambda I) x
  const int g = lane >> 3, r = lane & 7;
  // (round 4) the grid is at most two workgroups per CU; a workgroup takes several groups of 8 rows one after the other (XCD x keeps the contiguous chunk x of them): the
  // sums above are formed 512 times per launch instead of once per 8 rows, and no workgroup waits for a slot
  const int per_xcd = d.n_wg_spmv >> 3, g_per_xcd = gridDim.x >> 3;   // both padded to multiples of 8
  for (int li = blockIdx.x >> 3; li < per_xcd; li += g_per_xcd) {
  const int wg = (blockIdx.x & 7) * per_xcd + li;
  const int i = wg * kRowsPerWG + rl;
  double acc = 0;
  if (i < d.Cp && r < 6) {
    const int e0 = d.row_off[i], e1 = d.row_off[i + 1];
    // two blocks of the group per trip, both index pairs loaded before either block: the chain index -> block / vector is walked once for the two (round 5; a row of ~40
    // blocks gives a group two or three of them).  The sums are taken in the old order.
    auto block_row = [&](uint32_t bt, double v[6]) {
      if (F32) {
        const float* B = d.S32 + 36 * (size_t)(bt & ~kTransposeBit);
        if (bt & kTransposeBit) {
#pragma unroll
          for (int c = 0; c < 6; c++) v[c] = (double)B[c * 6 + r];
        } else {
          typedef float v2f __attribute__((ext_vector_type(2)));
          const v2f* B2 = reinterpret_cast<const v2f*>(B + r * 6);   // rows start at multiples of 24 bytes: 8-byte aligned
          const v2f b0 = B2[0], b1 = B2[1], b2 = B2[2];
          v[0] = (double)b0[0]; v[1] = (double)b0[1]; v[2] = (double)b1[0]; v[3] = (double)b1[1]; v[4] = (double)b2[0]; v[5] = (double)b2[1];
        }
      } else {
        const double* B = d.S + 36 * (size_t)(bt & ~kTransposeBit);
        if (bt & kTransposeBit) {
#pragma unroll
          for (int c = 0; c < 6; c++) v[c] = B[c * 6 + r];
        } else {
#pragma unroll
          for (int c = 0; c < 6; c++) v[c] = B[r * 6 + c];
        }
      }
    };
    for (int s = e0 + h * 8 + g; s < e1; s += 32) {
      const bool two = s + 16 < e1;
      const int sb = two ? s + 16 : s;
      const int ja = d.row_col[s], jb = d.row_col[sb];
      const uint32_t bta = d.row_blk[s], btb = d.row_blk[sb];
      double va[6], vb[6], za[6], zb[6];
      block_row(bta, va);
      block_row(btb, vb);
#pragma unroll
      for (int c = 0; c < 6; c++) { za[c] = zsrc[6 * (size_t)ja + c]; zb[c] = zsrc[6 * (size_t)jb + c]; }
#pragma unroll
      for (int c = 0; c < 6; c++) acc += va[c] * za[c];
      if (two) {
#pragma unroll
        for (int c = 0; c < 6; c++) acc += vb[c] * zb[c];
      }
    }
  }
  // sum over the 8 groups (lanes with equal r): fixed xor tree
  acc = lanex::add_partner<8>(acc);
  acc = lanex::add_partner<16>(acc);
  acc = lanex::add_partner<32>(acc);
  if (lane < 8) half_sum[rl][h][lane] = acc;
  __syncthreads();
  double pq = 0, qv = 0;
  if (h == 0 && i < d.Cp) {
    if (lane < 6) {
      const size_t gi = 6 * (size_t)i + lane;
      const double src_i = XMODE ? d.x[gi] : d.z[gi];
      qv = (half_sum[rl][0][lane] + half_sum[rl][1][lane]) + lambda * src_i;
      if (XMODE) d.qx[gi] = qv;
      else {
        // q_k = (S + lambda I) z_k + beta q_{k-1}, p_k = z_k + beta p_{k-1}: row-local (first iteration of a solve: beta = 0 and neither old vector is read)
        double pi = src_i;
        if (k) { qv += beta * d.q[gi]; pi += beta * pold[gi]; }
        d.q[gi] = qv;
        pnew[gi] = pi;
        pq = pi * qv;
      }
    }
    pq = wave_sum(lane < 6 ? pq : 0.0);
    if (lane == 0) lds[rl] = pq;
  } else if (h == 0 && lane == 0) lds[rl] = 0.0;
  __syncthreads();
  if (!XMODE && threadIdx.x == 0) {
    double tot = lds[0];
#pragma unroll
    for (int q = 1; q < kRowsPerWG; q++) tot += lds[q];
    d.ppq[wg] = tot;
  }
  }
}

// (round 5: ba_pcg_spmv_sym — every stored block read once, S_ij^T p_i scattered to the slot of block (i, j) among row j's lower entries — was built and measured in round 4
// (41.6 + 14.3 us per CG iteration against 46.5 + 9.9 on the 10 000-keyframe map: 141.8 against 142.4 ms per call, a wash: the product is bound by the requests in flight, not
// by its bytes) and is gone from the library; DESIGN 4.1 keeps the measurement.)

// alpha = rz_k / p.q ; x += alpha p ; r -= alpha q ; z = W r (cluster-wise dense) ; partial rz_{k+1}   [CCM_K_BA_PCG_UPDATE]
// one workgroup per cluster
// REPLACE (round 5, f32 product): x was already advanced by ba_pcg_xupdate and q holds (S + lambda I) x formed with the f64 blocks: r = b - q instead of r -= alpha q
template <bool REPLACE>
__global__ __launch_bounds__(kTPB) void ba_pcg_update(BaDev d, int k) {
  __shared__ double rc[kCluN];
  __shared__ double zpart[8][kCluN];
  __shared__ double red[kTPB / kWave];
  const int t = threadIdx.x, c = blockIdx.x;
  const int s0 = c * kClu, s1 = min(d.Cp, s0 + kClu);
  const int m = 6 * (s1 - s0);
  // independent loads first (the kernel is otherwise a chain of dependent round trips)
  const int done = d.pcg_flag[0];
  const double* p = d.p[(k + 1) & 1];
  const size_t g = 6 * (size_t)s0 + t;
  double xv = 0, rv = 0, qv = 0, pv = 0;
  if (t < m) { xv = d.x[g]; rv = d.r[g]; qv = REPLACE ? d.qx[g] : d.q[g]; pv = p[g]; }
  __shared__ double redp[3 * (kTPB / kWave)];
  double rz_k, pq;
  {
    const double* const ps[3] = {d.prz[k & 1], d.mk_cry[k & 1], d.ppq};
    const int ns[3] = {d.n_wg_upd, d.mk_on ? d.n_wg_upd : 0, d.n_wg_spmv};
    double sm[3];
    block_sum_partials<3>(ps, ns, sm, redp);
    rz_k = sm[0];
    if (d.mk_on) rz_k += sm[1];
    pq = sm[2];
  }
  if (done) return;
  if (!(pq > 0.0)) {   // not positive definite (or NaN): solver failure -> LM rejects the step
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.pcg_flag[0] = 1; d.pcg_flag[1] = k; d.pcg_flag[2] = 1; }
    return;
  }
  const double alpha = rz_k / pq;
  if (t < m) {
    if (REPLACE) rv = d.bs[g] - qv;
    else { d.x[g] = xv + alpha * pv; rv -= alpha * qv; }
    d.r[g] = rv;
    rc[t] = rv;
  } else if (t < kCluN) rc[t] = 0.0;
  __syncthreads();
  // z = W r_c: thread (rp, seg) = (t % 48, t / 48) takes rows 2 rp, 2 rp + 1 and the columns [24 seg, 24 seg + 24); W is symmetric, so W[col][row]
  // puts consecutive threads on consecutive addresses, and the 24 16-byte loads of a thread are all in flight at once: the whole 74 KB block
  // of the cluster arrives in ONE memory round trip (two threads per row with 8 loads in flight took six; rocprofv3: 13.5 us per launch on the
  // 10 000-keyframe map).  Rows / columns beyond a short last cluster: W is the identity there and r is zero.
  // (round 4: W is stored as f32 — 37 KB per cluster and iteration instead of 74; thread (rp, seg) = (t % 24, t / 24) takes rows 4 rp .. 4 rp + 3 and the
  // columns [12 seg, 12 seg + 12): twelve 16-byte loads in flight)
  const float* W = d.Wc + (size_t)c * kCluN * kCluN;
  if (t < 192) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int rp = t % 24, seg = t / 24;
    const v4f* Wp = reinterpret_cast<const v4f*>(W + (size_t)(12 * seg) * kCluN + 4 * rp);
    v4f w[12];
#pragma unroll
    for (int q = 0; q < 12; q++) w[q] = Wp[(size_t)q * (kCluN / 4)];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int q = 0; q < 12; q++) { const double rq = rc[12 * seg + q]; s0 += (double)w[q][0] * rq; s1 += (double)w[q][1] * rq; s2 += (double)w[q][2] * rq; s3 += (double)w[q][3] * rq; }
    zpart[seg][4 * rp] = s0; zpart[seg][4 * rp + 1] = s1; zpart[seg][4 * rp + 2] = s2; zpart[seg][4 * rp + 3] = s3;
  }
  __syncthreads();
  double rz = 0;
  if (t < m) {
    const double z = (((zpart[0][t] + zpart[1][t]) + (zpart[2][t] + zpart[3][t])) + ((zpart[4][t] + zpart[5][t]) + (zpart[6][t] + zpart[7][t])));
    d.z[g] = z;
    rz = rc[t] * z;
  }
  rz = wave_sum(rz);
  if ((t & (kWave - 1)) == 0) red[t / kWave] = rz;
  __syncthreads();
  if (t == 0) {
    d.prz[(k + 1) & 1][c] = ((red[0] + red[1]) + red[2]) + red[3];
    if (c == 0) d.pcg_flag[1] = k + 1;
  }
  if (d.mk_on) {
    __shared__ double prod[12 * kCluN];
    mk_restrict(d, c, s0, m, rc, prod);
  }
}

// (round 5: ba_pcg_update_coarse — update + coarse correction in one 512-thread workgroup per interval, the coarse residual following r's recurrence with P^T q left by the product
// kernel — was built and measured in round 4 (151.0 against 151.7 ms per call on the 10 000-keyframe map: what the saved launch gives, the product's extra epilogue takes) and is
// gone from the library; DESIGN 4.1 keeps the measurement.)

// x += alpha p alone (round 5: the first half of an iteration that replaces its residual), alpha from the same partial sums as ba_pcg_update; one workgroup per cluster
__global__ __launch_bounds__(kTPB) void ba_pcg_xupdate(BaDev d, int k) {
  __shared__ double redp[3 * (kTPB / kWave)];
  const int t = threadIdx.x, c = blockIdx.x;
  const int s0 = c * kClu, s1 = min(d.Cp, s0 + kClu);
  const int m = 6 * (s1 - s0);
  const int done = d.pcg_flag[0];
  const size_t g = 6 * (size_t)s0 + t;
  double xv = 0, pv = 0;
  if (t < m) { xv = d.x[g]; pv = d.p[(k + 1) & 1][g]; }
  double rz_k, pq;
  {
    const double* const ps[3] = {d.prz[k & 1], d.mk_cry[k & 1], d.ppq};
    const int ns[3] = {d.n_wg_upd, d.mk_on ? d.n_wg_upd : 0, d.n_wg_spmv};
    double sm[3];
    block_sum_partials<3>(ps, ns, sm, redp);
    rz_k = sm[0];
    if (d.mk_on) rz_k += sm[1];
    pq = sm[2];
  }
  if (done || !(pq > 0.0)) return;   // (a non-positive p.q is reported by the ba_pcg_update that follows)
  if (t < m) d.x[g] = xv + (rz_k / pq) * pv;
}
__global__ void ba_s_to_f32(const double* __restrict__ S, float* __restrict__ S32, size_t n4) {   // n4 = number of 4-element groups
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  typedef double v2d __attribute__((ext_vector_type(2)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v2d a = reinterpret_cast<const v2d*>(S)[2 * i], b = reinterpret_cast<const v2d*>(S)[2 * i + 1];
  v4f o; o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)b[0]; o[3] = (float)b[1];
  reinterpret_cast<v4f*>(S32)[i] = o;
}
constexpr int kMkReplaceEvery = 8;   // iterations between two residual replacements of the f32 product

// Coarse part of the preconditioner in the multi-kernel PCG, one workgroup per INTERVAL (two clusters) after ba_pcg_init_tiles / ba_pcg_update:
// rc = P^T r per coarse node (the first-node parts of the node's interval's two clusters + the second-node parts of the previous interval's),
// y = Ac^-1[rows of the two nodes of the interval] rc (every interval recomputes these 12 values: 12 x Nc multiply-adds, cheaper than another
// grid-wide step), z += P (w0 y_a + w1 y_a+1) for the interval's cameras, and the coarse part of r.z = rc . y, every node counted once (by its own
// interval; the last node by the last interval).
// (round 4: a workgroup per cluster read the 60 KB of parts and the 90 KB of inverse rows twice per interval, the parts cluster-major — 16 us per launch on the
// 10 000-keyframe map, bound by the 94 MB that 625 workgroups pulled from L2; parts node-major in four contributor slots, one workgroup per interval)
__global__ __launch_bounds__(kTPB) void ba_pcg_coarse_apply(BaDev d, int par) {
  extern __shared__ __attribute__((aligned(16))) double rcs[];   // [6 * (na + 1)]
  __shared__ double ys[12];
  if (d.pcg_flag[0]) return;
  const int t = threadIdx.x, agg = blockIdx.x;
  const int lane = t & (kWave - 1), wv = t / kWave;
  const int nca = 6 * (d.mk_na + 1), n_clu = d.n_wg_upd;
  // the thread's prolongation row and its z entry are requested first (7 loads; also requesting the first columns of the 12 inverse rows up here was measured:
  // 27 instead of 12 us per launch — 48 loads per thread queued ahead of the gather that the whole workgroup waits for)
  const int c_own = 2 * agg + (t >> 7), tl = t & 127;
  const int s0_own = c_own * kClu;
  const int m_own = c_own < n_clu ? 6 * (min(d.Cp, s0_own + kClu) - s0_own) : 0;
  double Prow[6] = {0, 0, 0, 0, 0, 0}, z_old = 0;
  if (tl < m_own) {
    const double* P = d.mk_P + 36 * (size_t)(s0_own + tl / 6) + 6 * (tl % 6);
#pragma unroll
    for (int cc = 0; cc < 6; cc++) Prow[cc] = P[cc];
    z_old = d.z[6 * (size_t)s0_own + tl];
  }
  // the loads of a batch of 8 entries per thread are all issued before the first LDS store
  for (int base = 0; base < nca; base += 8 * kTPB) {
    double va[8], vb[8], vc[8], vd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int e = base + k * kTPB + t;
      va[k] = vb[k] = vc[k] = vd[k] = 0.0;
      if (e < nca) { va[k] = d.mk_cpart[e]; vb[k] = d.mk_cpart[(size_t)nca + e]; vc[k] = d.mk_cpart[2 * (size_t)nca + e]; vd[k] = d.mk_cpart[3 * (size_t)nca + e]; }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int e = base + k * kTPB + t;
      if (e < nca) rcs[e] = (((0.0 + va[k]) + vb[k]) + vc[k]) + vd[k];   // clusters 2n, 2n + 1 (first-node parts), 2n - 2, 2n - 1 (second-node parts); absent: + 0.0
    }
  }
  __syncthreads();
  {
    // y = Ac^-1[12 rows of the interval's two nodes] rc: every thread takes a strided slice of all rows, one column of the twelve per trip, then wave
    // trees and the four wave sums in order.
    __shared__ double yred[12][kTPB / kWave];
    double a12[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* ar = d.mk_Ainv32 + (size_t)(6 * agg) * d.mk_Nc;   // f32 copy (round 4): the 12 rows are re-read by every interval in every CG iteration
    // one column of the twelve rows per trip (12 loads in flight per thread), accumulated in the plain loop's order.  Measured on one box, loads queued per thread and trip
    // (round 5, 10 000-keyframe map): 96 -> 20.6 us per launch, 48 (rounds 3 - 4) -> 12.6, 36 -> 12.2, 24 -> 11.9, 12 -> 11.3: the queue of a deeper batch delays the
    // loads the other waves of the CU are waiting for
    for (int jj = t; jj < nca; jj += kTPB) {
      float w[12];
      const double rv = rcs[jj];
#pragma unroll
      for (int rr = 0; rr < 12; rr++) w[rr] = ar[(size_t)rr * d.mk_Nc + jj];
#pragma unroll
      for (int rr = 0; rr < 12; rr++) a12[rr] += (double)w[rr] * rv;
    }
#pragma unroll
    for (int rr = 0; rr < 12; rr++) { const double w = wave_sum(a12[rr]); if (lane == 0) yred[rr][wv] = w; }
    __syncthreads();
    if (t < 12) ys[t] = ((yred[t][0] + yred[t][1]) + yred[t][2]) + yred[t][3];
  }
  __syncthreads();
  if (tl < m_own) {   // threads 0 .. 95: the interval's first cluster, 128 .. 223: its second one
    const double w1 = coarse_hat_t(s0_own + tl / 6, d.agg), w0 = 1.0 - w1;
    double zc = 0;
#pragma unroll
    for (int cc = 0; cc < 6; cc++) zc += Prow[cc] * (w0 * ys[cc] + w1 * ys[6 + cc]);
    d.z[6 * (size_t)s0_own + tl] = z_old + zc;
  }
  if (t == 0) {
    double sv = 0;
    for (int rr = 0; rr < 6; rr++) sv += rcs[6 * agg + rr] * ys[rr];
    if (agg == d.mk_na - 1) for (int rr = 0; rr < 6; rr++) sv += rcs[6 * (agg + 1) + rr] * ys[6 + rr];
    d.mk_cry[par][2 * agg] = sv;
    if (2 * agg + 1 < n_clu) d.mk_cry[par][2 * agg + 1] = 0.0;
  }
}

// ---- coarse level of the two-level preconditioner ------------------------------------------------------------------
// Cluster block-Jacobi cannot see the smooth error modes of a long trajectory (rigid drifts of whole map sections), so the
// CG iteration count grows with the map.  Coarse space: one rigid-body twist (6 unknowns) per NODE, nodes placed every `agg` (BaDev::agg: 16 or 32)
// consecutive cameras, interpolated LINEARLY in the camera index between the two nodes of a camera's interval (hat functions:
// camera k of interval a takes (1 - t) xi_a + t xi_a+1, t = (k % agg + 1/2) / agg) and prolongated to camera k by the adjoint
// P_k = Ad(T_cw,k) (a world-frame twist xi moves camera k by the left perturbation Ad(T_cw) xi).  Ac = P^T (S + lambda I) P is dense and
// small (6 (Cp / agg + 1) unknowns, 756 for the 4-agent map with intervals of 16); its explicit inverse is formed by the tile kernels of dense_chol.hip and the
// persistent PCG adds P Ac^-1 P^T r to the cluster-Jacobi term.  Round 2 used piecewise-CONSTANT twists per aggregate (same size of Ac); offline
// on a 1000-keyframe 4-agent system (same matrices, CG to 1e-8): cluster-Jacobi alone 357 iterations at lambda 0.3, + constant aggregates 136,
// + hats 82 (lambda 3: 346 / 116 / 74; lambda 30: 269 / 83 / 59) — a continuous interpolant represents the smooth drift modes that a
// step function can only follow with its jumps.
// hysteresis of the adaptive switch (iterations of the previous solve): intervals of 32 cameras: a coarse build (0.38 ms) is worth ~35 CG iterations; the
// finer spaces converge in fewer iterations with the level ON, so their thresholds sit lower (with 80 / 35 a 16-camera space switches itself off after every
// good solve: 2391 instead of 779 CG iterations per call on the 3-agent map; swept on the 3- and 4-agent maps: 40 / 15)
constexpr int kCoarseOnIters = 80, kCoarseOffIters = 35, kCoarseOnItersFine = 40, kCoarseOffItersFine = 15;

__global__ void ba_coarse_P(BaDev d, int cur, double* Pm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.Cp) return;
  const BaPose T = ba_load_pose(d.cam[cur] + 7 * (size_t)d.slot_cam[i]);
  double R[9];
  ba_q_to_R(T, R);
  const double t[3] = {T.tx, T.ty, T.tz};
  // [t]x R
  double tR[9];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    tR[0 * 3 + c] = -t[2] * R[3 + c] + t[1] * R[6 + c];
    tR[1 * 3 + c] = t[2] * R[0 + c] - t[0] * R[6 + c];
    tR[2 * 3 + c] = -t[1] * R[0 + c] + t[0] * R[3 + c];
  }
  double* P = Pm + 36 * (size_t)i;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      P[r * 6 + c] = R[r * 3 + c]; P[r * 6 + 3 + c] = 0.0;
      P[(3 + r) * 6 + c] = tR[r * 3 + c]; P[(3 + r) * 6 + 3 + c] = R[r * 3 + c];
    }
}

// one workgroup (16 waves) per pair of camera intervals (a <= b) that share S blocks: the four weighted sums
//   F^pq_ab = sum over the S blocks (i in a, j in b) of w^p_i w^q_j P_i^T (S_ij [+ lambda I]) P_j,   w^0 = 1 - t, w^1 = t,
// (an off-diagonal S block inside one interval also contributes its transpose, w^q_i w^p_j (...)^T) go to stage[pair][2 p + q]; ba_coarse_sum then
// adds the up to four F that meet in every pair of coarse NODES (n, n') = (a + p, b + q).  The work is pure latency (entry -> block index -> cameras ->
// three 6x6 matrices), so every wave takes kCoarseBatch consecutive entries at a time, stages their matrices in its own LDS slice with coalesced loads
// (all chains of a batch in flight together) and multiplies out of LDS.  The 16 partial sums are added in a fixed order => deterministic.
constexpr int kCoarseTPB = 1024;
constexpr int kCoarseBatch = 4;
__global__ __launch_bounds__(kCoarseTPB) void ba_coarse_assemble(BaDev d, const double* Pm, const int* cb_off, const int* cb_ent, int n_cb,
                                                                 const int* blk_i, const int* blk_j, double lambda, double* stage_out /* [n_cb][4][36] */) {
  constexpr int kNW = kCoarseTPB / kWave;
  __shared__ double stage[kNW][kCoarseBatch][3][36];   // [S block | P_i | P_j]
  __shared__ double part[kNW][4][36];
  const int w = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const bool el = lane < 36;
  const int r = el ? lane / 6 : 0, c = el ? lane % 6 : 0;
  const int end = cb_off[w + 1];
  double acc[4] = {0, 0, 0, 0};
  for (int s0 = cb_off[w] + wv * kCoarseBatch; s0 < end; s0 += kNW * kCoarseBatch) {
    int code[kCoarseBatch], ci[kCoarseBatch], cj[kCoarseBatch];
#pragma unroll
    for (int u = 0; u < kCoarseBatch; u++) code[u] = (s0 + u < end) ? cb_ent[s0 + u] : -1;
#pragma unroll
    for (int u = 0; u < kCoarseBatch; u++) { const int blk = (code[u] >= 0) ? code[u] >> 1 : 0; ci[u] = blk_i[blk]; cj[u] = blk_j[blk]; }
    double vb[kCoarseBatch], vi[kCoarseBatch], vj[kCoarseBatch];
#pragma unroll
    for (int u = 0; u < kCoarseBatch; u++) {
      const int blk = (code[u] >= 0) ? code[u] >> 1 : 0;
      vb[u] = el ? d.S[36 * (size_t)blk + lane] : 0.0;
      vi[u] = el ? Pm[36 * (size_t)ci[u] + lane] : 0.0;
      vj[u] = el ? Pm[36 * (size_t)cj[u] + lane] : 0.0;
    }
    if (el) {
#pragma unroll
      for (int u = 0; u < kCoarseBatch; u++) {
        stage[wv][u][0][lane] = vb[u] + ((ci[u] == cj[u] && r == c) ? lambda : 0.0);
        stage[wv][u][1][lane] = vi[u];
        stage[wv][u][2][lane] = vj[u];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is private to this wave and LDS executes a wave's accesses in order
#pragma unroll
    for (int u = 0; u < kCoarseBatch; u++) {
      if (code[u] < 0) continue;
      const double* B = stage[wv][u][0];
      const double* Pi = stage[wv][u][1];
      const double* Pj = stage[wv][u][2];
      double m = 0;
#pragma unroll
      for (int q = 0; q < 6; q++) {
        double tq = 0;
#pragma unroll
        for (int p = 0; p < 6; p++) tq += Pi[p * 6 + r] * B[p * 6 + q];
        m += tq * Pj[q * 6 + c];
      }
      const double mt = __shfl(m, c * 6 + r, kWave);
      const double ti = coarse_hat_t(ci[u], d.agg), tj = coarse_hat_t(cj[u], d.agg);
      const double wi[2] = {1.0 - ti, ti}, wj[2] = {1.0 - tj, tj};
      const bool both = (code[u] & 1) != 0;   // off-diagonal S block inside one interval: its mirror image belongs to the same pair
#pragma unroll
      for (int pq = 0; pq < 4; pq++) {
        const int p = pq >> 1, q = pq & 1;
        double v = (wi[p] * wj[q]) * m;
        if (both) v += (wi[q] * wj[p]) * mt;
        acc[pq] += v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (el) {
#pragma unroll
    for (int pq = 0; pq < 4; pq++) part[wv][pq][lane] = acc[pq];
  }
  __syncthreads();
  if (threadIdx.x < 4 * 36) {
    const int pq = threadIdx.x / 36, e = threadIdx.x % 36;
    double tot = 0;
#pragma unroll
    for (int k = 0; k < kNW; k++) tot += part[k][pq][e];
    stage_out[((size_t)w * 4 + pq) * 36 + e] = tot;
  }
}

// Ac[n][n'] (6x6, all node pairs; identity on the padding up to Nc) = sum over p, q of F^pq of the interval pair (n - p, n' - q); a pair stored as
// (b, a) with b < a is read transposed with p and q exchanged.  cb_key: the sorted keys a * na + b of the interval pairs that exist.
__global__ void ba_coarse_sum(const double* stage, const unsigned* cb_key, int n_cb, int na, int nn, double* Ac, int Nc) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Nc * Nc) return;
  const int row = (int)(idx / Nc), col = (int)(idx % Nc);
  if (row >= 6 * nn || col >= 6 * nn) { Ac[idx] = (row == col) ? 1.0 : 0.0; return; }
  int n = row / 6, r = row % 6, n2 = col / 6, c = col % 6;
  if (row > col) { int tmp = n; n = n2; n2 = tmp; tmp = r; r = c; c = tmp; }   // the mirror element runs the SAME additions in the same order: Ac is bit-symmetric
  double sum = 0;
#pragma unroll
  for (int pq = 0; pq < 4; pq++) {
    const int p = pq >> 1, q = pq & 1;
    int a = n - p, b = n2 - q;
    if (a < 0 || b < 0 || a >= na || b >= na) continue;
    int sel = 2 * p + q, e = r * 6 + c;
    if (a > b) { const int tmp = a; a = b; b = tmp; sel = 2 * q + p; e = c * 6 + r; }
    const unsigned key = (unsigned)a * (unsigned)na + (unsigned)b;
    int lo = 0, hi = n_cb;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cb_key[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < n_cb && cb_key[lo] == key) sum += stage[((size_t)lo * 4 + sel) * 36 + e];
  }
  Ac[idx] = sum;
}

// ---- persistent PCG: the WHOLE solve of one LM trial in one launch of co-resident workgroups -------------------
// The multi-kernel iteration above costs two launches (~22 us on gba_c4) for ~15 MB of traffic: launch gaps and the
// dependent index -> block -> vector chains dominate, not HBM.  Here one 16-wave workgroup OWNS one preconditioner
// cluster (16 cameras) for the entire solve:
//   * it assembles and factors its 96x96 block once and keeps the explicit inverse W in LDS (72 KiB) — the
//     preconditioner apply never touches global memory again;
//   * x, r, q of its rows live in LDS; only z and p (needed by covisible neighbours) go through global memory, with
//     device-coherent (sc1) loads / stores, so NO cache flush or invalidate is needed at the grid barriers and the
//     S blocks of the cluster's rows stay resident in the XCD's L2 from one iteration to the next;
//   * per iteration the p = z + beta p_old entries of the cluster's ~100-200 distinct neighbour columns are fetched
//     ONCE into LDS (one coherent round trip, all 1024 threads), the row products then read S from L2 and p from
//     LDS; the CSR structure of the rows is staged in LDS as well;
//   * one wave per block row, 16 blocks in flight per wave;
//   * workgroup b runs on XCD b % 8 and takes cluster (b%8)*per_xcd + b/8, so the S blocks shared by neighbouring
//     clusters are served by one XCD's L2;
//   * two grid barriers per iteration (after q = S p for p.q, after z = W r for r.z) built on one monotonic
//     device-scope counter; the dot products are reduced from per-workgroup partials in a fixed order, so every
//     workgroup derives bit-identical alpha / beta / convergence decisions and the loop stays grid-uniform.
// A barrier gives up after a bounded spin (abort flag -> solver failure -> LM rejects the step) so a scheduling
// accident can never hang the device.  Cooperative launch guarantees co-residency; the host uses this path when
// the cluster count fits (<= 4096 free cameras on 256 CUs), the multi-kernel path otherwise.
// One spin = two dependent device-coherent loads (~0.8 us measured inside a solve) + the sleep: ~8 ms per exchange at worst.  Round 5: was 3 000 000 (seconds — two
// contexts of one process that launched at the same time each held part of the chip for that long before giving up).  With the per-device lease
// (ccm_coresident_scope) two such kernels never meet; what a launch can still wait for is another context's ORDINARY kernels to leave the CUs its last workgroups need,
// i.e. single kernel durations (<= a few ms).  Under a CONTINUOUS foreign load — scripts/gpu_soak_concurrency.py: three global BAs, two local BAs, an ORB batch stream and
// a pose loop at once — the dispatcher does not promise that a whole CU ever falls free for the last workgroups (one launch in 14 681 waited in vain, 60 000 spins =
// ~60 ms, at the first setting of this round): the bound is what such a launch costs before its trial is repeated on the multi-kernel solver, hence short.
constexpr long kPersMaxSpins = 8000;
constexpr int kPersCooldownTrials = 8;   // LM trials a single-rank handle spends on the multi-kernel solver after a persistent launch gave up, before it tries again

struct PersArgs {
  double lambda, rel_tol;
  int max_it, n_clu;
  unsigned* bar;        // [1] abort flag
  int test_abort;              // CCM_BA_TEST_ABORT: the last workgroup leaves at once (exercises the abort -> multi-kernel fallback)
  const int* coff; const int* cij; const uint32_t* cblk;   // per cluster: entries inside its own 16x16 block (local row << 4 | column, S block)
  unsigned long long* slots;   // [2][2][grid]: p.q exchange, r.z exchange (word-major 16-byte slots)
  unsigned long long epoch_base;   // unique per launch: stale slots of earlier solves never validate
  const int* uoff;      // [n_clu+1] offsets into ucol
  const int* ucol;      // distinct columns of each cluster's rows, ascending
  const int* loc;       // [n_row_entries] position of the entry's column in its cluster's ucol list
  long long* dbg;       // optional [16] phase clocks of workgroup 0 (wall_clock64 ticks, 10 ns), accumulated over iterations
  // coarse level (nullptr = cluster-Jacobi only): explicit inverse [Nc x Nc], prolongation blocks [Cp][36], aggregates
  const double* Ainv; const double* Pm; int na, Nc;   // na camera intervals, na + 1 coarse nodes
  double* cparts;              // [12][grid]: the units' parts of the coarse restriction P^T q (6 for the first node of the unit's interval, 6 for the second), component-major; exchanged like p, q and z
  // the cluster inverse across trials (round 4): [grid][96 * 48] the unit's own 48 rows of W, transposed, as the kernel keeps them in LDS.  w_load != 0: this
  // launch takes them from here instead of assembling and factoring the cluster block (a stale W — another lambda, an earlier linearisation — is still a
  // symmetric positive definite block-Jacobi preconditioner: PCG stays exact, only the iteration count moves; the host decides, lm_trial)
  double* wsave; int w_load;
};

// a value every lane already agrees on, moved to scalar registers (frees 2 VGPRs per double in the PCG loop)
__device__ __forceinline__ double pers_uniform(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double coh_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coh_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid-wide "barrier + deterministic sum" in one step, without atomics (256 arrivals on one counter serialise at the
// memory side: measured 6 us per barrier): every workgroup publishes its partial in its own 16-byte slot as
// {bits(value), bits(value) ^ key(epoch)} and then polls ALL slots (one per thread) until every slot validates against
// the current epoch's key; the xor check also rejects torn 16-byte reads.  The values are then summed in a fixed order,
// so every workgroup obtains bit-identical totals.  All cross-workgroup data (z, p, q, slots) moves with device-coherent
// accesses, so the only ordering needed is: drain this workgroup's stores, meet, publish.
__device__ __forceinline__ bool pers_exchange(int t /* threadIdx.x */, unsigned long long* slots /* [2][nwg] */, int nwg, int me, double v_thread, bool force_nan,
                                              unsigned long long epoch, unsigned* abort_flag, double* red /* [kPersWaves + 4] */,
                                              double* total, int* poll_fail /* LDS, sticky: an aborted exchange ends the solve */) {
  // v_thread: this thread's share of the unit's partial.  The unit sum, the publish and the grid-wide sum share two block
  // barriers: wave sums -> LDS, (drain stores, barrier), thread 0 adds the 16 wave sums in a fixed order and publishes,
  // wave 0 polls (lane = source units t, t+64, ...; nwg <= 256), (barrier), everybody reads the grid total.
  // The slot words are stored WORD-MAJOR (word w of unit i at slots[w * nwg + i]) so that a wave's load of one word is
  // 512 contiguous bytes.  Variants measured and dropped: polling with 4 or 16 waves (slows the units still computing),
  // wide slots that also carried the coarse components (8-12 us per exchange; they now travel like p, q and z), and
  // pushing the value into per-unit inboxes (64K scattered write-through stores per exchange).
  const unsigned long long key = 0x9E3779B97F4A7C15ull * epoch;
  {
    const double ws = wave_sum(v_thread);
    if ((t & (kWave - 1)) == 0) red[t / kWave] = ws;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    double mine = 0;
#pragma unroll
    for (int w = 0; w < kPersWaves; w++) mine += red[w];
    if (force_nan) mine = __longlong_as_double(0x7ff8000000000000ll);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
    __hip_atomic_store(slots + me, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(slots + nwg + me, bits ^ key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t < kWave) {
    constexpr int kPer = 4;    // slots per lane
    unsigned done = 0;
    double val[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) { val[j] = 0; if (t + j * kWave >= nwg) done |= 1u << j; }
    bool ok_w = true;
    // (measured and dropped, round 3: a second poll request ~0.25 us behind the first so that an incomplete first answer does not cost a whole further round
    // trip: the per-unit time inside an exchange went from 3.4 to 4.1-4.4 us — twice the polling traffic on the slot lines slows every answer down)
    for (long spins = 0;; spins++) {
#pragma unroll
      for (int j = 0; j < kPer; j++) {
        if (done & (1u << j)) continue;   // a slot that validated is not read again
        const unsigned long long* in = slots + t + j * kWave;
        const unsigned long long b0 = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b1 = __hip_atomic_load(in + nwg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((b0 ^ b1) == key) { done |= 1u << j; val[j] = __longlong_as_double((long long)b0); }
      }
      if (__all(done == (1u << kPer) - 1u)) break;
      if (spins > kPersMaxSpins || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok_w = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    const double v = wave_sum(((val[0] + val[1]) + val[2]) + val[3]);
    if (t == 0) {
      red[kPersWaves] = v;
      if (!ok_w) { *poll_fail = 1; __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
  __syncthreads();
  const bool alive = *poll_fail == 0;
  *total = pers_uniform(red[kPersWaves]);
  // no trailing barrier: red[0..15] is rewritten only after every thread has passed the barrier above, red[16] only
  // after the first barrier of the next exchange
  return alive;
}

#define PERS_TICK(slot) { if (timing) { const long long tn_ = wall_clock64(); tacc[slot] += tn_ - tacc[12]; tacc[12] = tn_; } }

__device__ __forceinline__ double pers_bcast(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  double r = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  asm volatile("" : "+v"(r));   // keep the value in VGPRs: as an SGPR-pair operand it trips the register-class verifier in this function
  return r;
}

// Assemble the damped 96x96 block of one cluster from the block-CSR rows, factor it (Cholesky blocked by camera) and
// leave W = (block)^-1 in A.  Kept out of line so that its register-hungry 6x6 temporaries do not compete with the
// register-resident S rows of the PCG loop.
__device__ __noinline__ void pers_assemble_cluster(double* A, double* Li, const int* cij, const uint32_t* cblk, int c_lo, int c_hi, const double* S, int m,
                                                   double lambda, bool has) {
  constexpr int N = kCluN;
  const int t = threadIdx.x;
  // ---- init 1: assemble the damped dense block of the cluster ----
  for (int i = t; i < N * N; i += kPersTPB) { A[i] = 0; Li[i] = 0; }
  __syncthreads();
  if (has) {   // the cluster's own entries come from a host-built list: up to 10 independent (index, block) chains per thread in flight
    const int grp = t / 36, e = t % 36, r = e / 6, cc = e % 6;
    constexpr int kGroups = kPersTPB / 36;
    if (grp < kGroups)
#pragma unroll 4
      for (int s = c_lo + grp; s < c_hi; s += kGroups) {
        const int ij = cij[s], il = ij >> 4, jl = ij & 15;
        const uint32_t bt = cblk[s];
        const double* B = S + 36 * (size_t)(bt & ~kTransposeBit);
        const double v = (bt & kTransposeBit) ? B[cc * 6 + r] : B[e];
        A[(6 * il + r) * N + 6 * jl + cc] = v + ((il == jl && r == cc) ? lambda : 0.0);
      }
  }
  if (m + t < N) A[(m + t) * N + m + t] = 1.0;   // unit diagonal on the padding rows of a short cluster: the factorisation needs no special case
  __syncthreads();
}

// Factor the dense SPD block in A (lower triangle read; m live rows, unit diagonal beyond) and leave W = A^-1 in A; Li (all zero on entry) is scratch.
__device__ __noinline__ void pers_factor_dense(double* A, double* Li, int* ibuf, int m, long long* tacc, bool timing) {
  constexpr int N = kCluN;
  const int t = threadIdx.x;
  PERS_TICK(7)
  // ---- init 2: Cholesky A = L L^T on 16x16 tiles: 6 steps of (diagonal tile | panel | trailing update) instead of 16
  // camera-sized ones.  Diagonal tile: one wave, lane = row with its 16 entries in registers, pivot-row entries by
  // v_readlane, reciprocal square roots.  Panel: one thread per row below, multiplies by the stored reciprocals.
  // Trailing update: rank-16 on the f64 matrix cores (4 k-steps per 16x16 tile).  104 us (6-column steps, scalar
  // update) -> 43 us (6-column steps, MFMA update) -> 42 us measured inside the two-cluster solver (diagonal tiles 20, panels 15,
  // trailing updates 7; then 14 us for L^-1 and 5 us for W).  Tried and dropped (round 2): right-looking diagonal tile with scalar-register
  // broadcasts + the tile's inverse in the same wave + the panel as one MFMA product per tile: 51 us — the 16 dependent pivots
  // (rsqrt, broadcast, update: ~300 cycles each) set the time of a tile whichever way the off-path work is arranged.
  // Round 4, measured and dropped as well: the same fused updates applied right-looking inside the registers (column c's multiples leave the later columns as 15 - c
  // independent multiply-adds, so that the next pivot waits for one of them instead of a dot product of c terms; bit-identical): both factorisations of the two-cluster
  // solve 59.7 -> 79.9 us — the broadcasts of the freshly scaled column (v_readlane into scalar registers, then the wait states before a vector instruction may read them)
  // land on the pivot path, while the left-looking dot product broadcasts values that were final long before.
  double* invd = Li;   // 16 reciprocal pivots of the current step (Li is all zero otherwise and is restored below)
  {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int lane = t & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(t / kWave);
    const int i16 = lane & 15, kq = lane >> 4;
    for (int T = 0; T < 6 && 16 * T < m; T++) {
      const int b0 = 16 * T;
      // Diagonal tile AND panel in one instruction stream (round 3): lanes 0..15 of waves 0 and 1 each factor the 16 rows of the diagonal tile (the same
      // arithmetic, so both hold the same pivot rows), lanes 16..63 carry rows BELOW the tile (wave 0: the first 48, wave 1: the next 32) and form their panel
      // entries L_ic = (A_ic - sum_{k<c} L_ik L_ck) / L_cc with the multipliers arriving by v_readlane from lanes 0..15 of their own wave — the column step a
      // row below the diagonal takes anyway.  The separate panel pass (one thread per row, 16 dependent steps on LDS operands, a barrier) cost 15 of the 42 us.
      const int r0 = b0 + 16;
      if (wave < 2) {
        const int rl = lane & 15;
        const int prow = r0 + 48 * wave + (lane - 16);   // panel row of lanes 16..63
        const bool is_panel = lane >= 16;
        const int rowi = is_panel ? min(prow, N - 1) : b0 + rl;
        double row[16];
#pragma unroll
        for (int k = 0; k < 16; k++) row[k] = A[rowi * N + b0 + k];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 16; c++) {
          double sv = row[c];
#pragma unroll
          for (int k = 0; k < c; k++) sv = fma(-row[k], pers_bcast(row[k], c), sv);   // (fused: the dot product is a serial chain on the pivot path; nothing downstream matches these bits)
          double dd = pers_bcast(sv, c);
          if (!(dd > 0.0)) { bad = true; dd = 1.0; }      // (padding rows of a short cluster carry a unit diagonal)
          const double inv = rsqrt(dd);
          row[c] = (!is_panel && rl == c) ? dd * inv : sv * inv;     // lanes above the diagonal hold unused values
          if (wave == 0 && lane == c) invd[c] = inv;
        }
        if (wave == 0 && bad && lane == 0) ibuf[1] = 1;
        if (!is_panel) {
          if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) if (k <= lane) A[(b0 + lane) * N + b0 + k] = row[k];
          }
        } else if (prow < m) {
#pragma unroll
          for (int k = 0; k < 16; k++) A[prow * N + b0 + k] = row[k];
        }
      }
      __syncthreads();
      {   // trailing update A_IJ -= X_I X_J^T over the lower tiles I >= J > T
        const int nt = 5 - T, ntiles = nt * (nt + 1) / 2;
        for (int tile = wave; tile < ntiles; tile += kPersWaves) {
          int Ii = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
          while (Ii * (Ii + 1) / 2 > tile) Ii--;
          while ((Ii + 1) * (Ii + 2) / 2 <= tile) Ii++;
          const int I = T + 1 + Ii, Jt = T + 1 + tile - Ii * (Ii + 1) / 2;
          v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; kk++)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(16 * I + i16) * N + b0 + 4 * kk + kq], A[(16 * Jt + i16) * N + b0 + 4 * kk + kq], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * I + kq + 4 * r, col = 16 * Jt + i16;
            if (col <= row) A[row * N + col] -= acc[r];
          }
        }
      }
      __syncthreads();
    }
  }
  if (t < 16) Li[t] = 0.0;   // the reciprocal-pivot scratch lives in Li's first row
  __syncthreads();
  PERS_TICK(8)
  // ---- init 3: Li = L^-1 on 16x16 tiles (6 x 6 of them; Li is all zero on entry) ----
  // 3a: the six diagonal tiles by forward substitution, one wave per tile, lane = column (pivot rows are LDS broadcasts);
  // 3b: tiles at block distance dl = 1..5, one wave per tile: Li_IJ = -Li_II * sum_{K=J}^{I-1} L_IK Li_KJ on the f64 matrix
  // cores (the 16x16 sum is parked in the mirror tile (J, I) of Li, which is zero before and after).  8 us instead of
  // 35 us for the 6x6-blocked scalar version with its 15 barrier-separated block distances.
  {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int lane = t & (kWave - 1), wave = t / kWave;
    if (wave < 6 && lane < 16) {
      const int b0 = 16 * wave, c = lane;
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        double sv = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; k++) sv = fma(-A[(b0 + r) * N + b0 + k], x[k], sv);
        const double dg = A[(b0 + r) * N + b0 + r];
        x[r] = (b0 + r < m) ? sv / dg : 0.0;          // rows beyond the cluster's cameras are padding
      }
#pragma unroll
      for (int r = 0; r < 16; r++) if (r >= c) Li[(b0 + r) * N + b0 + c] = x[r];
    }
    __syncthreads();
    const int i16 = lane & 15, kq = lane >> 4;
    for (int dl = 1; dl < 6; dl++) {
      if (wave < 6 - dl) {
        const int J = wave, I = wave + dl;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int K = J; K < I; K++)
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            const int k = 4 * kk + kq;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(16 * I + i16) * N + 16 * K + k], Li[(16 * K + k) * N + 16 * J + i16], acc, 0, 0, 0);
          }
        double* scr = Li + (16 * J) * N + 16 * I;     // mirror tile (J, I): rows = k, columns = j of the parked sum
#pragma unroll
        for (int r = 0; r < 4; r++) scr[(kq + 4 * r) * N + i16] = acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave reads it back
        v4d out = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int k = 4 * kk + kq;
          out = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[(16 * I + i16) * N + 16 * I + k], scr[k * N + i16], out, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; r++) { Li[(16 * I + kq + 4 * r) * N + 16 * J + i16] = -out[r]; scr[(kq + 4 * r) * N + i16] = 0.0; }
      }
      __syncthreads();
    }
  }
  PERS_TICK(9)
  // ---- init 4: W = Li^T Li into the A region (the factor is dead): 36 tiles of 16x16 on the f64 matrix cores, one wave
  // per tile; operand rows of Li are read straight from LDS (16 consecutive doubles per k: conflict-free); Li is lower
  // triangular and zero beyond m, so tile (I, J) starts at k = 16 max(I, J).  4 us instead of 78 us as a scalar loop.
  {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int lane = t & (kWave - 1), wave = t / kWave;
    const int i16 = lane & 15, kq = lane >> 4;
    for (int tile = wave; tile < 36; tile += kPersWaves) {
      const int I = tile / 6, J = tile % 6;
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      for (int k0 = 16 * max(I, J); k0 < N; k0 += 4) {
        const double* Lk = Li + (k0 + kq) * N;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lk[16 * I + i16], Lk[16 * J + i16], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) A[(16 * I + kq + 4 * r) * N + 16 * J + i16] = acc[r];
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void pers_factor_cluster(double* A, double* Li, int* ibuf, const int* cij, const uint32_t* cblk, int c_lo, int c_hi,
                                                    const double* S, int s0, int s1, double lambda, bool has, long long* tacc, bool timing) {
  pers_assemble_cluster(A, Li, cij, cblk, c_lo, c_hi, S, 6 * (s1 - s0), lambda, has);
  pers_factor_dense(A, Li, ibuf, 6 * (s1 - s0), tacc, timing);
}

// ---- reduced systems of 17..32 free cameras (a local-BA window): EXACT solve in ONE workgroup -------------------------------------------------------
// The two 16-camera clusters are eliminated block-wise with the tile factorisation above (f64 matrix cores):
//   W11 = A11^-1,  T = W11 A12,  S22 = A22 - A12^T T,  W22 = S22^-1,  x2 = W22 (b2 - A12^T W11 b1),  x1 = W11 b1 - T x2.
// Two 96x96 LDS regions are all a factorisation leaves room for, so T waits in a global scratch (74 KB, L2) while S22 is factored.  This replaces
// ~19 iterations of the persistent PCG over 4 workgroups (11 us each: two grid exchanges per iteration) by ~6 block products; g2o itself solves
// these systems directly (LinearSolverDense / Eigen LDLT, Optimizer.cpp:371-375), so the exact solve is also the closer restatement.
static inline size_t dense2_lds_bytes() { return (size_t)(2 * kCluN * kCluN + 5 * kCluN + 8 * kCluN) * sizeof(double) + 16 + 14 * sizeof(long long) + 64 * sizeof(int); }

__global__ __launch_bounds__(kPersTPB) void ba_solve_dense2(BaDev d, double lambda, const int* coff, const int* cij, const uint32_t* cblk, double* gTt /* [96][96] T transposed */,
                                                               long long* dbg /* nullable: [9] phase clocks (10 ns ticks) + launches */, int cur, int add_lambda_term) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  typedef double v4d __attribute__((ext_vector_type(4)));
  constexpr int N = kCluN;
  double* R1 = sm;
  double* R2 = sm + N * N;
  double* b1 = R2 + N * N; double* b2 = b1 + N; double* t1 = b2 + N; double* cv = t1 + N; double* x2 = cv + N;
  double* zpart = x2 + N;                                   // [8][N]
  int* ibuf = reinterpret_cast<int*>(zpart + 8 * N);        // [4]
  long long* tacc = reinterpret_cast<long long*>(ibuf + 4);
  int* roff = reinterpret_cast<int*>(tacc + 14);            // [kDense2MaxCp + 1] row offsets of the block CSR
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(t / kWave);
  const int i16 = lane & 15, kq = lane >> 4;
  const int Cp = d.Cp, m2 = 6 * (Cp - kClu);
  long long tprev = (dbg && t == 0) ? wall_clock64() : 0;
  const bool timing = dbg != nullptr && t == 0;
  if (timing) { for (int q = 0; q < 12; q++) tacc[q] = 0; tacc[12] = wall_clock64(); }
#define D2_TICK(slot) { if (dbg && t == 0) { const long long tn_ = wall_clock64(); dbg[slot] += tn_ - tprev; tprev = tn_; } }
  if (t < 4) ibuf[t] = 0;
  if (t <= Cp) roff[t] = d.row_off[t];
  if (t < N) { b1[t] = d.bs[t]; b2[t] = (t < m2) ? d.bs[N + t] : 0.0; }
  // v_out[row] = sum_k M[k][row] * v_in[k] over the 96 x 96 region M (row-major, read down its columns: conflict-free); 8 column parts
  auto matvec_t = [&](const double* M, const double* vin) {
    const int row = t % N, prt = t / N;
    if (prt < 8) {
      double sv = 0;
#pragma unroll
      for (int k = 0; k < 12; k++) sv += M[(12 * prt + k) * N + row] * vin[12 * prt + k];
      zpart[prt * N + row] = sv;
    }
    __syncthreads();
    double r = 0;
    if (t < N) {
      r = zpart[t];
#pragma unroll
      for (int q = 1; q < 8; q++) r += zpart[q * N + t];
    }
    return r;                                               // valid for t < N
  };
  pers_assemble_cluster(R1, R2, cij, cblk, coff[0], coff[1], d.S, N, lambda, true);
  D2_TICK(0)
  pers_factor_dense(R1, R2, ibuf, N, tacc, timing);          // R1 = W11 (symmetric)
  D2_TICK(1)
  {
    const double v = matvec_t(R1, b1);
    if (t < N) t1[t] = v;
  }
  // A12 (rows: cluster 0, columns: cluster 1) into R2 from the block-CSR rows of cluster 0
  for (int e = t; e < N * N; e += kPersTPB) R2[e] = 0.0;
  __syncthreads();
  {
    const int e0 = roff[0], ne = roff[kClu] - e0;
    for (int idx = t; idx < 6 * ne; idx += kPersTPB) {
      const int sidx = e0 + idx / 6, r = idx % 6;
      const int j = d.row_col[sidx];
      if (j < kClu) continue;
      int i = 0;
#pragma unroll
      for (int step = 8; step > 0; step >>= 1) if (i + step < kClu && roff[i + step] <= sidx) i += step;
      const uint32_t bt = d.row_blk[sidx];
      const double* B = d.S + 36 * (size_t)(bt & ~kTransposeBit);
#pragma unroll
      for (int c = 0; c < 6; c++) R2[(6 * i + r) * N + 6 * (j - kClu) + c] = (bt & kTransposeBit) ? B[c * 6 + r] : B[r * 6 + c];
    }
  }
  __syncthreads();
  D2_TICK(2)
  // T = W11 A12: 36 tiles of 16 x 16, waves 0..3 own three, the others two
  v4d acc[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int tile = wave + kPersWaves * q;
    v4d a4 = {0.0, 0.0, 0.0, 0.0};
    if (tile < 36) {
      const int I = tile / 6, J = tile % 6;
      for (int k0 = 0; k0 < N; k0 += 4)
        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(R1[(k0 + kq) * N + 16 * I + i16], R2[(k0 + kq) * N + 16 * J + i16], a4, 0, 0, 0);   // W11 symmetric: read down the column
    }
    acc[q] = a4;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int tile = wave + kPersWaves * q;
    if (tile < 36) {
      const int I = tile / 6, J = tile % 6;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 16 * I + kq + 4 * r, col = 16 * J + i16;
        R1[row * N + col] = acc[q][r];
        gTt[col * N + row] = acc[q][r];
      }
    }
  }
  __syncthreads();
  D2_TICK(3)
  // P = A12^T T (registers) and c = b2 - A12^T t1
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int tile = wave + kPersWaves * q;
    v4d a4 = {0.0, 0.0, 0.0, 0.0};
    if (tile < 36) {
      const int I = tile / 6, J = tile % 6;
      for (int k0 = 0; k0 < N; k0 += 4)
        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(R2[(k0 + kq) * N + 16 * I + i16], R1[(k0 + kq) * N + 16 * J + i16], a4, 0, 0, 0);
    }
    acc[q] = a4;
  }
  {
    const double v = matvec_t(R2, t1);
    if (t < N) cv[t] = b2[t] - v;
  }
  __syncthreads();
  D2_TICK(4)
  // S22 = A22 - P, factored in place
  pers_assemble_cluster(R1, R2, cij, cblk, coff[1], coff[2], d.S, m2, lambda, true);
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int tile = wave + kPersWaves * q;
    if (tile < 36) {
      const int I = tile / 6, J = tile % 6;
#pragma unroll
      for (int r = 0; r < 4; r++) R1[(16 * I + kq + 4 * r) * N + 16 * J + i16] -= acc[q][r];
    }
  }
  __syncthreads();
  D2_TICK(5)
  pers_factor_dense(R1, R2, ibuf, m2, tacc, timing);         // R1 = W22
  D2_TICK(6)
  {
    const double v = matvec_t(R1, cv);
    if (t < N) x2[t] = v;
  }
  __syncthreads();
  {   // x1 = t1 - T x2, T^T from the global scratch (coalesced down its columns)
    const int row = t % N, prt = t / N;
    if (prt < 8) {
      double sv = 0;
#pragma unroll
      for (int k = 0; k < 12; k++) sv += gTt[(12 * prt + k) * N + row] * x2[12 * prt + k];
      zpart[prt * N + row] = sv;
    }
    __syncthreads();
    if (t < N) {
      double r = zpart[t];
#pragma unroll
      for (int q = 1; q < 8; q++) r += zpart[q * N + t];
      d.x[t] = t1[t] - r;
      if (t < m2) d.x[N + t] = x2[t];
    }
  }
  // (round 4) the camera update of the trial rides in this launch: ba_update_cams' arithmetic for the window's <= 32 free cameras (one launch less per LM trial
  // of a local BA; the window has one workgroup's worth of cameras, so its partial of the gain denominator is the whole of it)
  __syncthreads();
  {
    double sc = 0;
    if (t < Cp) {
      const int c = d.slot_cam[t];
      double u[6];
#pragma unroll
      for (int q = 0; q < 6; q++) u[q] = d.x[6 * (size_t)t + q];
      const BaPose T = ba_load_pose(d.cam[cur] + 7 * (size_t)c);
      const BaPose Tn = ba_oplus(u, T);
      ba_store_pose(d.cam[cur ^ 1] + 7 * (size_t)c, Tn);
#pragma unroll
      for (int q = 0; q < 6; q++) sc += u[q] * ((add_lambda_term ? lambda * u[q] : 0.0) + d.bp[6 * (size_t)t + q]);
    }
    if (t < kWave) {   // Cp <= 32: all terms sit in wave 0; ba_update_cams' block_sum adds the four wave sums of its 256 threads, the other three being zero
      const double s = wave_sum(sc);
      if (t == 0) d.part_cam[0] = ((s + 0.0) + 0.0) + 0.0;
    }
  }
  D2_TICK(7)
  if (dbg && t == 0) { dbg[8] += 1; dbg[9] += tacc[8]; dbg[10] += tacc[9]; }
#undef D2_TICK
  if (t == 0) { d.pcg_flag[0] = 1; d.pcg_flag[1] = 1; d.pcg_flag[2] = ibuf[1]; d.pcg_flag[3] = 0; }
}

// ---- reduced systems of 17..50 free cameras (the reference's configured local-BA window, conf/config.yaml:78): EXACT Cholesky solve in ONE workgroup with the matrix in
// the CU's REGISTER FILE (round 6) --------------------------------------------------------------------------------------------------------------------------------
// A window's reduced camera system is dense (every keyframe of the window sees the others' points) and small: 300 x 300 for 50 free cameras.  Its lower triangle is
// 361 KB in f64 — more than the 160 KB of LDS that limited ba_solve_dense2 to two 96 x 96 blocks, but less than the CU's 512 KB of vector registers.  So the trailing
// matrix lives in registers as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4 (4 doubles = 8 registers per lane and tile): 8 waves x 24 tiles = 192 tiles
// = the 190 lower tiles of a 19 x 19 tile matrix (n <= 304).  Right-looking tile Cholesky, one step per tile column T:
//   (1) the owners of column T's tiles park them in an LDS panel buffer (row = matrix row below the diagonal, 16 columns; double-buffered, so a step costs two barriers),
//   (2) diagonal tile and panel in ONE instruction stream (as pers_factor_dense): lanes 0..15 of every wave factor the diagonal tile redundantly, lanes 16..63 carry 48
//       rows below it with the multipliers arriving by v_readlane — and the right-hand side rides along as one more panel row, which makes y = L^-1 b the forward substitution's result for free,
//   (3) every wave brings ITS tiles (I, J > T) up to date: four f64 MFMAs per tile, operands from the panel buffer, accumulators never leave the registers; the owners of column T's tiles
//       take the finished L tiles back into the (now dead) accumulators for the backward pass.
// Backward substitution L^T x = y row by row from the bottom: 16 lanes solve the diagonal tile (saved in LDS), the owners of row I's tiles subtract L_IJ^T x_I from y_J.
// Replaces the persistent PCG for 33..50 free cameras (a 1.9 x step in the cost of an LM trial between 32 and 33 cameras) and ba_solve_dense2 below that;
// g2o solves these windows directly as well (LinearSolverEigen on a dense pattern, Optimizer.cpp:371-375 / linear_solver_eigen.h:106-136).
// broadcast of one lane's double, register class left to the compiler (a scalar-register pair can be the multiplier of v_fma_f64 directly)
__device__ __forceinline__ double cr_bcast(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// 1 / sqrt(x) for a pivot x > 0 of a damped SPD block (no denormal scaling, no special cases): v_rsq_f64's estimate (relative error d ~ 2^-23 or better) and ONE step of
// cubic convergence, y (1 + e/2 + 3 e^2 / 8) with e = 1 - x y^2 (error ~ d^3): four dependent operations on the pivot path instead of the six of two Newton steps
__device__ __forceinline__ double cr_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

constexpr int kCRWaves = 8, kCRTPB = kCRWaves * kWave, kCRSlots = 24, kCRMaxNT = 19, kCRStride = 18 /* doubles per panel row: 144 B, 16-byte aligned, spreads the lanes' rows over the banks */;
constexpr int kCRRows = 16 * kCRMaxNT + 16;          // panel rows incl. the right-hand side's
static inline size_t cholreg_lds_bytes() {
  return (size_t)(2 * kCRRows * kCRStride + kCRMaxNT * 256 + 4 * kCRRows) * sizeof(double) + 4 * sizeof(int) + 16 * sizeof(long long);
}

// Where every lane of ba_solve_cholreg finds its four entries of every tile: the element's offset into S (in doubles; bit 30: a diagonal entry, lambda is added),
// -1: a structural zero, -2: unit diagonal of a padding row.  Depends on the block structure only, so it is built once per handle (first trial) — the solve then starts
// with one coalesced 16-byte load per lane and tile instead of three dependent round trips (row offsets -> CSR entries -> LDS table -> S).
constexpr int kCRDiagBit = 1 << 30;
__global__ __launch_bounds__(kCRTPB) void ba_cholreg_table(BaDev d, int* table /* [8][24][64][4] */) {
  __shared__ int idx[(kCholRegMaxCp + 1) * (kCholRegMaxCp + 1)];
  __shared__ int roff[kCholRegMaxCp + 2];
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave, i16 = lane & 15, kq = lane >> 4;
  const int Cp = d.Cp, n = 6 * Cp, NT = (n + 15) / 16, ntiles = NT * (NT + 1) / 2;
  if (t <= Cp) roff[t] = d.row_off[t];
  for (int e = t; e < Cp * Cp; e += kCRTPB) idx[e] = -1;
  __syncthreads();
  const int ne = roff[Cp];
  for (int sidx = t; sidx < ne; sidx += kCRTPB) {      // camera pair -> S block, from the block-CSR rows (both triangles are listed; lower entries carry the transpose bit)
    int i = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) if (i + step < Cp && roff[i + step] <= sidx) i += step;
    idx[i * Cp + d.row_col[sidx]] = (int)d.row_blk[sidx];
  }
  __syncthreads();
  for (int s = 0; s < kCRSlots; s++) {
    const int rho = kCRWaves * s + wave;
    int o[4] = {-1, -1, -1, -1};
    if (rho < ntiles) {
      int m = (int)((sqrtf(8.0f * (float)rho + 1.0f) - 1.0f) * 0.5f);
      while (m * (m + 1) / 2 > rho) m--;
      while ((m + 1) * (m + 2) / 2 <= rho) m++;
      const int J = NT - 1 - m, I = J + (rho - m * (m + 1) / 2);
      const int col = 16 * J + i16, cj = col / 6, c6 = col % 6;
      for (int r = 0; r < 4; r++) {
        const int row = 16 * I + kq + 4 * r;
        if (row < n && col < n) {
          const int ci = row / 6, r6 = row % 6;
          const int bt = idx[ci * Cp + cj];
          if (bt != -1) o[r] = (36 * (int)((uint32_t)bt & ~kTransposeBit) + (((uint32_t)bt & kTransposeBit) ? c6 * 6 + r6 : r6 * 6 + c6)) | (row == col ? kCRDiagBit : 0);
        } else if (row == col) o[r] = -2;
      }
    }
    reinterpret_cast<int4*>(table)[(wave * kCRSlots + s) * kWave + lane] = make_int4(o[0], o[1], o[2], o[3]);
  }
}

// Phase (3) of ba_solve_cholreg for the slots S, S + 1, ... of one wave: (-A_IJ) += L_IT L_JT^T in place, with the operands of slot S + 1 requested BEFORE slot S
// multiplies.  The slots a step still updates are a prefix [0, nact) of the wave's slots, so the blocks are NESTED (slot S + 1 lives inside slot S's block): what a
// block loads for the next one is simply in scope there — as sibling blocks behind their own `if`, every operand register became a phi of "loaded" and "not loaded",
// and the register allocator answered with copies and scratch spills.
struct CrStep { const double* P; double* Pn; int T, NT, R_n, wave, i16, kq, tvec; };
typedef double cr_v4d __attribute__((ext_vector_type(4)));
typedef double cr_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cr_load_ops(const CrStep& c, int s, double (&xa)[4], double (&xb)[4]) {
  const int ij = __builtin_amdgcn_readlane(c.tvec, s);
  const int I = min(max((ij >> 8) - c.T, 0), c.NT - 1 - c.T), J = min(max((ij & 255) - c.T, 0), c.NT - 1 - c.T);   // (clamped: the last block of a chain loads for a slot that is not updated)
  // k-step kk of the four MFMAs takes k = 4 kq + kk from lane group kq — on both operands, so the sixteen products of an entry are the same ones, summed in another
  // order — which makes a lane's four operand values CONTIGUOUS: two 16-byte LDS reads per operand, conflict-free at the row stride of 18 doubles
  const cr_v2d* pa = reinterpret_cast<const cr_v2d*>(c.P + (16 * I + c.i16) * kCRStride + 4 * c.kq);
  const cr_v2d* pb = reinterpret_cast<const cr_v2d*>(c.P + (16 * J + c.i16) * kCRStride + 4 * c.kq);
  const cr_v2d a01 = pa[0], a23 = pa[1], b01 = pb[0], b23 = pb[1];
  xa[0] = a01[0]; xa[1] = a01[1]; xa[2] = a23[0]; xa[3] = a23[1]; xb[0] = b01[0]; xb[1] = b01[1]; xb[2] = b23[0]; xb[3] = b23[1];
}
template <int S>
__device__ __forceinline__ void cr_update_chain(cr_v4d (&acc)[kCRSlots], const CrStep& c, int nact, const double (&xa)[4], const double (&xb)[4]) {
  if constexpr (S < kCRSlots) {
    if (S < nact) {
      double na[4], nb[4];
      if constexpr (S + 1 < kCRSlots) cr_load_ops(c, S + 1, na, nb);
#ifndef CR_EXPERIMENT_NO_MFMA   /* timing experiment only (results are wrong): what the phase costs without its matrix instructions */
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %5, %0\n\tv_mfma_f64_16x16x4_f64 %0, %2, %6, %0\n\tv_mfma_f64_16x16x4_f64 %0, %3, %7, %0\n\tv_mfma_f64_16x16x4_f64 %0, %4, %8, %0"
                   : "+v"(acc[S]) : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]));
#else
      acc[S][0] += xa[0] * xb[0] + xa[1] * xb[1] + xa[2] * xb[2] + xa[3] * xb[3];
#endif
      const int rho = kCRWaves * S + c.wave;
      if (rho >= c.R_n) {                                    // column T + 1: up to date now -> the other panel buffer (the accumulators hold minus the matrix)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[S]));   // the compiler does not see the MFMAs inside the asm above: 18 wait states before a vector instruction reads the result of a 16x16 f64 MFMA
        const int rb = 16 * (rho - c.R_n);
#pragma unroll
        for (int r = 0; r < 4; r++) c.Pn[(rb + c.kq + 4 * r) * kCRStride + c.i16] = -acc[S][r];
      }
      if constexpr (S + 1 < kCRSlots) cr_update_chain<S + 1>(acc, c, nact, na, nb);
    }
  }
}

__global__ __launch_bounds__(kCRTPB) void ba_solve_cholreg(BaDev d, double lambda, double* gL /* [n_pad][n_pad] scratch: the factor's rows below the diagonal tiles, written once and read once */, const int* table,
                                                             long long* dbg /* nullable: [8] phase clocks (10 ns ticks) + launches */, int cur, int add_lambda_term) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  typedef double v4d __attribute__((ext_vector_type(4)));
  typedef double v2d __attribute__((ext_vector_type(2)));
  double* Pbuf = sm;                                   // [2][kCRRows][kCRStride]
  double* Dg = Pbuf + 2 * kCRRows * kCRStride;         // [NT][16][16] factored diagonal tiles (row-major, lower)
  double* invd = Dg + kCRMaxNT * 256;                  // [n_pad] 1 / L_cc
  double* yv = invd + kCRRows;                         // [n_pad] y = L^-1 b, then the backward pass's running right-hand side
  double* bv = yv + kCRRows;                           // [n_pad] right-hand side, updated tile column by tile column
  double* xv = bv + kCRRows;                           // [n_pad] solution
  int* ibuf = reinterpret_cast<int*>(xv + kCRRows);    // [4]
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(t / kWave);
  const int i16 = lane & 15, kq = lane >> 4;
  const int Cp = d.Cp, n = 6 * Cp, NT = (n + 15) / 16, n_pad = 16 * NT, ntiles = NT * (NT + 1) / 2;
  long long tprev = (dbg && t == 0) ? wall_clock64() : 0;
#define CR_TICK(slot) { if (dbg && t == 0) { const long long tn_ = wall_clock64(); dbg[slot] += tn_ - tprev; tprev = tn_; } }
  if (t < 4) ibuf[t] = 0;
  // Tiles are RANKED column by column from the last column to the first (inside a column from the diagonal down) and dealt to the waves round-robin: the tile of rank
  // rho belongs to wave rho % 8, slot rho / 8.  The tiles that step T still has to update (column > T) are then exactly the ranks below R(T) = (NT-T-1)(NT-T)/2 — a
  // PREFIX of every wave's slots, the same length (+-1) in every wave at every step — and column T's own tiles are the NT - T ranks that follow.
  int tvec = 0xffff;                                     // lane s: (I << 8 | J) of this wave's slot s; read with v_readlane where a slot needs it
  if (lane < kCRSlots) {
    const int rho = kCRWaves * lane + wave;
    int m = (int)((sqrtf(8.0f * (float)rho + 1.0f) - 1.0f) * 0.5f);      // columns after this tile's column
    while (m * (m + 1) / 2 > rho) m--;
    while ((m + 1) * (m + 2) / 2 <= rho) m++;
    const int J = NT - 1 - m, I = J + (rho - m * (m + 1) / 2);
    if (rho < ntiles) tvec = (I << 8) | J;
  }
#define CR_TIJ(s) __builtin_amdgcn_readlane(tvec, (s))
  // ---- MINUS the damped system into the accumulators (the update is then acc += L_I L_J^T with the operands as they lie in LDS); padding rows beyond n: unit diagonal.
  // Every lane fetches its own four entries of every tile straight from the S blocks through the handle's offset table (ba_cholreg_table): one coalesced 16-byte load per
  // tile, then ~96 independent loads per lane, all in flight together, no staging pass and no barrier (a staged form — all threads fill a dense area with coalesced reads, the
  // waves then pick their tiles — took 33 us in three barrier-separated passes: the latency of three rounds of loads, not their bytes) ----
  v4d acc[kCRSlots];
  {
    const int4* tab = reinterpret_cast<const int4*>(table) + (size_t)wave * kCRSlots * kWave + lane;
#pragma unroll
    for (int s = 0; s < kCRSlots; s++) {
      const int4 o4 = tab[s * kWave];
      const int o[4] = {o4.x, o4.y, o4.z, o4.w};
      v4d a4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double v = 0.0;
        if (o[r] >= 0) { v = d.S[o[r] & (kCRDiagBit - 1)]; if (o[r] & kCRDiagBit) v += lambda; }
        else if (o[r] == -2) v = 1.0;
        a4[r] = -v;
      }
      acc[s] = a4;
    }
  }
  for (int e = t; e < n_pad + 16; e += kCRTPB) { bv[e] = (e < n) ? d.bs[e] : 0.0; yv[e] = 0.0; xv[e] = 0.0; }
  CR_TICK(0)
  // ---- right-looking tile Cholesky ----
  // column 0 -> panel buffer 0 (row 0..15: diagonal tile, then the rows below, then the right-hand side's chunk); from then on step T parks column T + 1 as it updates it
  {
    const int R_0 = (NT - 1) * NT / 2;
#pragma unroll
    for (int s = 0; s < kCRSlots; s++) {
      const int rho = kCRWaves * s + wave;
      if (rho >= R_0 && rho < R_0 + NT) {
        const int rb = 16 * (rho - R_0);
#pragma unroll
        for (int r = 0; r < 4; r++) Pbuf[(rb + kq + 4 * r) * kCRStride + i16] = -acc[s][r];
      }
    }
    if (t < 16) Pbuf[n_pad * kCRStride + t] = bv[t];
  }
  for (int T = 0; T < NT; T++) {
    double* P = Pbuf + (T & 1) * (kCRRows * kCRStride);
    double* Pn = Pbuf + ((T + 1) & 1) * (kCRRows * kCRStride);
    const int rows_below = n_pad - 16 * (T + 1);
    const int R_T = (NT - T - 1) * (NT - T) / 2;             // ranks [0, R_T): columns > T (this step's trailing update); [R_T, R_T + NT - T): column T, diagonal tile first
    const int R_n = (NT - T - 2) * (NT - T - 1) / 2;         // ranks [R_n, R_T): column T + 1
    __syncthreads();
    CR_TICK(4)
    // (2) diagonal tile + panel (+ right-hand side) in one instruction stream
    if (wave == 0 || 48 * wave <= rows_below) {
      const bool is_panel = lane >= 16;
      const int pr = 48 * wave + (lane - 16);              // panel row of lanes 16..63 (pr == rows_below: the right-hand side)
      const int rowi = is_panel ? 16 + min(pr, rows_below) : lane;
      double row[16];
      {
        const v2d* p2 = reinterpret_cast<const v2d*>(P + rowi * kCRStride);
#pragma unroll
        for (int k = 0; k < 8; k++) { const v2d v = p2[k]; row[2 * k] = v[0]; row[2 * k + 1] = v[1]; }
      }
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        double s0 = row[c], s1 = 0.0;                      // the dot product as two interleaved chains (it sits on the pivot path)
#pragma unroll
        for (int k = 0; k < c; k++) { if (k & 1) s1 = fma(-row[k], cr_bcast(row[k], c), s1); else s0 = fma(-row[k], cr_bcast(row[k], c), s0); }
        const double sv = s0 + s1;
        double dd = cr_bcast(sv, c);
        if (!(dd > 0.0)) { bad = true; dd = 1.0; }
        const double inv = cr_rsqrt(dd);                   // (uniform: every lane computes it from the broadcast pivot)
        row[c] = sv * inv;                                 // diagonal lane c: sv == dd, so this is L_cc = dd * inv; lanes above the diagonal hold unused values
        if (wave == 0 && lane == 0) invd[16 * T + c] = inv;
      }
      if (wave == 0 && bad && lane == 0) ibuf[1] = 1;
      if (!is_panel) {
        if (wave == 0) {
#pragma unroll
          for (int k = 0; k < 16; k++) Dg[T * 256 + lane * 16 + k] = (k <= lane) ? row[k] : 0.0;
        }
      } else if (pr < rows_below) {
        v2d* p2 = reinterpret_cast<v2d*>(P + rowi * kCRStride);
        v2d* g2 = reinterpret_cast<v2d*>(gL + (size_t)(16 * (T + 1) + pr) * n_pad + 16 * T);   // for the backward pass (the registers of column T's tiles are NOT rewritten here:
#pragma unroll                                                                                  // a conditional write of an accumulator inside the step loop costs copies and spills)
        for (int k = 0; k < 8; k++) { v2d v; v[0] = row[2 * k]; v[1] = row[2 * k + 1]; p2[k] = v; g2[k] = v; }
      } else if (pr == rows_below) {
#pragma unroll
        for (int k = 0; k < 16; k++) yv[16 * T + k] = row[k];
      }
    }
    __syncthreads();
    CR_TICK(5)
    // (3) right-hand side; trailing update of this wave's tiles (a prefix of its slots), column T + 1 parked for the next step as soon as it is up to date
    if (t < rows_below) {
      const double* pr_ = P + (16 + t) * kCRStride;
      double sv = bv[16 * (T + 1) + t];
#pragma unroll
      for (int k = 0; k < 16; k++) sv = fma(-pr_[k], yv[16 * T + k], sv);
      bv[16 * (T + 1) + t] = sv;
      if (t < 16) Pn[(16 + rows_below - 16) * kCRStride + t] = sv;       // the next step's right-hand side row (its rows_below is 16 less)
    }
    const int nact = min(max((R_T - wave + kCRWaves - 1) / kCRWaves, 0), kCRSlots);     // this wave's slots [0, nact) hold tiles of the columns > T
    {
      CrStep cs; cs.P = P; cs.Pn = Pn; cs.T = T; cs.NT = NT; cs.R_n = R_n; cs.wave = wave; cs.i16 = i16; cs.kq = kq; cs.tvec = tvec;
      double fa[4], fb[4];
      cr_load_ops(cs, 0, fa, fb);
      cr_update_chain<0>(acc, cs, nact, fa, fb);
    }
    CR_TICK(6)
  }
  __syncthreads();
  // the factor's tiles below the diagonal back into the (dead) accumulators for the backward pass: one batch of loads per wave
#pragma unroll
  for (int s = 0; s < kCRSlots; s++) {
    const int ij = CR_TIJ(s), I = min(ij >> 8, NT - 1), J = min(ij & 255, NT - 1);
    const double* g = gL + (size_t)(16 * I + kq) * n_pad + 16 * J + i16;
    v4d a4;
#pragma unroll
    for (int r = 0; r < 4; r++) a4[r] = g[(size_t)4 * r * n_pad];
    acc[s] = a4;
  }
  CR_TICK(1)
  // ---- backward substitution L^T x = y, tile row by tile row from the bottom ----
  for (int I = NT - 1; I >= 0; I--) {
    if (t < 16) {
      double v = yv[16 * I + t];
      double Lc[16];
#pragma unroll
      for (int c = 0; c < 16; c++) Lc[c] = Dg[I * 256 + c * 16 + t];      // column t of the diagonal tile
      const double myinv = invd[16 * I + t];
      double xres = 0.0;
#pragma unroll
      for (int c = 15; c >= 0; c--) {
        const double xc = cr_bcast(v * myinv, c);                          // lane c: x_c = y'_c / L_cc
        if (t < c) v = fma(-Lc[c], xc, v);
        if (t == c) xres = xc;
      }
      xv[16 * I + t] = xres;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kCRSlots; s++) {
      const int ij = CR_TIJ(s);
      if ((ij >> 8) == I && (ij & 255) < I) {
        const int J = ij & 255;
        double pv = 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) pv = fma(acc[s][r], xv[16 * I + kq + 4 * r], pv);
        pv = lanex::add_partner<16>(pv);
        pv = lanex::add_partner<32>(pv);
        if (kq == 0) yv[16 * J + i16] -= pv;
      }
    }
    __syncthreads();
  }
  if (t < n) d.x[t] = xv[t];
  CR_TICK(2)
  // the camera update of the trial rides in this launch (as in ba_solve_dense2): ba_update_cams' arithmetic for the window's <= 50 free cameras, all in wave 0
  {
    double sc = 0;
    if (t < Cp) {
      const int c = d.slot_cam[t];
      double u[6];
#pragma unroll
      for (int q = 0; q < 6; q++) u[q] = xv[6 * t + q];
      const BaPose Tc = ba_load_pose(d.cam[cur] + 7 * (size_t)c);
      const BaPose Tn = ba_oplus(u, Tc);
      ba_store_pose(d.cam[cur ^ 1] + 7 * (size_t)c, Tn);
#pragma unroll
      for (int q = 0; q < 6; q++) sc += u[q] * ((add_lambda_term ? lambda * u[q] : 0.0) + d.bp[6 * (size_t)t + q]);
    }
    if (t < kWave) {   // Cp <= 50: all terms sit in wave 0; ba_update_cams' block_sum adds the four wave sums of its 256 threads, the other three being zero
      const double s = wave_sum(sc);
      if (t == 0) d.part_cam[0] = ((s + 0.0) + 0.0) + 0.0;
    }
  }
  CR_TICK(3)
  if (dbg && t == 0) dbg[7] += 1;
#undef CR_TICK
#undef CR_TIJ
  if (t == 0) { d.pcg_flag[0] = 1; d.pcg_flag[1] = 1; d.pcg_flag[2] = ibuf[1]; d.pcg_flag[3] = 0; }
}

__global__ __launch_bounds__(kPersTPB) void ba_pcg_persist(BaDev d, PersArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int N = kCluN;
  double* A = sm;                          // [96][96]: dense cluster block -> Cholesky factor -> W = inverse
  double* Li = sm + N * N;                 // [96][96]: inverse factor during init, then staged structure + neighbour p
  double* rc = Li + N * N;                 // r of the own rows
  double* xs = rc + N;                     // x
  double* ps = xs + N;                     // p
  double* qs = ps + N;                     // q
  double* zs = qs + N;                     // z
  double* zpart = zs + N;                  // [8][96]
  double* red = zpart + 8 * N;             // [16]
  int* ibuf = reinterpret_cast<int*>(red + kPersWaves + 4);   // [0]=ok flag of the barrier, [1]=bad pivot
  if (a.test_abort && blockIdx.x == gridDim.x - 1) return;
  const int t = threadIdx.x, lane = t & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(t / kWave);
  const int nwg = gridDim.x;               // padded to a multiple of 8
  const int per_xcd = nwg >> 3;
  // TWO workgroups per cluster: unit u owns rows [8u, 8u+8) (x, p, q, z of those rows and their S blocks in registers);
  // both units of a cluster factor the same 96x96 block redundantly and keep the full r of the cluster, so the only
  // extra exchange is the partner's q (48 values) after the first barrier.
  const int u = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int c = u >> 1, hu = u & 1;
  const bool has = c < a.n_clu;
  const int s0 = has ? c * kClu : 0, s1 = has ? min(d.Cp, s0 + kClu) : 0;
  const int nrows = s1 - s0;
  const int m = 6 * nrows;
  const int o0 = has ? min(s1, s0 + (kClu / 2) * hu) : 0, o1 = has ? min(s1, o0 + kClu / 2) : 0;   // own rows
  const int nown = o1 - o0;
  const int mo = 6 * nown;                 // own unknowns
  const int ob = 6 * (o0 - s0);            // their offset inside the cluster vectors
  unsigned long long* slots_pq = a.slots;
  unsigned long long* slots_rz = a.slots + 2 * (size_t)nwg;
  unsigned long long epoch = a.epoch_base;
  const double lambda = a.lambda;
  if (t < 4) ibuf[t] = (t == 0) ? 1 : 0;   // [2] = exchange failure flag
  // optional phase clocks of workgroup 0 / thread 0, accumulated in LDS so that they cost no registers
  long long* tacc = reinterpret_cast<long long*>(ibuf + 4);   // [13]: 12 phases + last stamp
  const bool timing = a.dbg != nullptr && blockIdx.x == 0 && t == 0;
  if (timing) { for (int q = 0; q < 12; q++) tacc[q] = 0; tacc[12] = wall_clock64(); }
  // keep only the unit's own 48 rows of W, transposed (WT[col][row]: conflict-free for the mat-vec); the other half of the
  // A region then holds the coarse level: Ac^-1 rows of the two nodes of the unit's interval (as f32: it is only a preconditioner, and 12 rows in f64
  // would not fit) | coarse residual | own P_k | y of the two nodes
  if (a.w_load) {   // the inverse an earlier launch of this handle left behind: 36 KB per unit out of L2 instead of ~80 us of assembly + tile factorisation
    const double* src = a.wsave + (size_t)u * (N * (N / 2));
    double wreg[5];
    int nw = 0;
    for (int e = t; e < N * (N / 2); e += kPersTPB, nw++) wreg[nw] = src[e];
    nw = 0;
    for (int e = t; e < N * (N / 2); e += kPersTPB, nw++) A[e] = wreg[nw];
    __syncthreads();
  } else {
    pers_factor_cluster(A, Li, ibuf, a.cij, a.cblk, has ? a.coff[c] : 0, has ? a.coff[c + 1] : 0, d.S, s0, s1, lambda, has, tacc, timing);
    double wreg[5];
    int nw = 0;
    for (int e = t; e < N * (N / 2); e += kPersTPB, nw++) wreg[nw] = A[(e / (N / 2)) * N + ob + e % (N / 2)];
    __syncthreads();
    nw = 0;
    double* dst = a.wsave ? a.wsave + (size_t)u * (N * (N / 2)) : nullptr;
    for (int e = t; e < N * (N / 2); e += kPersTPB, nw++) { A[e] = wreg[nw]; if (dst) dst[e] = wreg[nw]; }
  }
  const bool coarse = a.Ainv != nullptr;
  const int Nc = a.Nc, nca = 6 * (a.na + 1);
  float* ainv_l = reinterpret_cast<float*>(A + N * (N / 2));   // [12][Nc] f32: up to the whole half (Nc <= 768: intervals of 16 cameras on the 4-agent map)
  // coarse residual | own P_k | y of the two nodes: behind the staged structure in the Li region (dead once W is formed; the structure lists that move in
  // below end at pl + 6 kPersColCap + 256)
  double* rco = Li + (32 + 2 * kPersIdxCap + kPersColCap) / 2 + 6 * kPersColCap + 256;   // [Nc]
  double* pown = rco + Nc;                   // [8][36]
  double* ypart = pown + 8 * 36;             // [12]
  static_assert((32 + 2 * kPersIdxCap + kPersColCap) % 2 == 0 && (32 + 2 * kPersIdxCap + kPersColCap) / 2 + 6 * kPersColCap + 256 + 768 + 8 * 36 + 12 <= kCluN * kCluN,
                "coarse vectors do not fit behind the staged structure");
  const int aggu = d.agg >> 3;               // units per interval (an interval is a multiple of the 8 cameras of a unit)
  const int agg = u / aggu;                  // interval of the unit's cameras: nodes agg and agg + 1
  if (coarse) {
    {   // the 12 rows are one contiguous range of Ac^-1; all of a thread's loads in flight at once (as a rolled loop: nine dependent round trips per launch)
      const double* src = a.Ainv + (size_t)(6 * agg) * Nc;
      double tmpa[9];
#pragma unroll
      for (int q = 0; q < 9; q++) { const int e = t + q * kPersTPB; tmpa[q] = (has && e < 12 * Nc) ? src[e] : 0.0; }
#pragma unroll
      for (int q = 0; q < 9; q++) { const int e = t + q * kPersTPB; if (e < 12 * Nc) ainv_l[e] = (float)tmpa[q]; }
    }
    for (int e = t; e < Nc; e += kPersTPB) rco[e] = 0.0;
    for (int e = t; e < 8 * 36; e += kPersTPB) pown[e] = (e / 36 < nown) ? a.Pm[36 * (size_t)o0 + e] : 0.0;
  }
  __syncthreads();
  // Loop-invariant per-thread indices (everything derived from threadIdx) must NOT stay live across the PCG loop: the
  // 60 VGPRs of S blocks leave ~60 for the rest, and LLVM hoists every such index out of the loop and then spills it
  // (29-50 scratch reloads per iteration measured).  tq / lq are re-laundered copies of t / lane, opaque to the
  // optimiser, refreshed at the top of every iteration: the indices are recomputed (a few integer ops) instead.
  int tq = t, lq = lane;
  // unit part of the coarse restriction P^T v for the own rows: wave 0, lane = (component c, camera k); the 8 camera
  // terms of a component sit in 8 neighbouring lanes; every camera's term goes with weight 1 - t to the first node of the unit's interval and with t
  // to the second.  Published component-major for the other units.
  auto coarse_restrict = [&](const double* vec /* LDS, own 48 entries */) {
    if (wv == 0) {
      const int cc = lq >> 3, kk = lq & 7;
      double sv = 0;
      if (cc < 6 && kk < nown) {
#pragma unroll
        for (int rr = 0; rr < 6; rr++) sv += pown[kk * 36 + rr * 6 + cc] * vec[6 * kk + rr];
      }
      const double w1 = coarse_hat_t(8 * u + kk, d.agg);
      double s0v = (1.0 - w1) * sv, s1v = w1 * sv;
      s0v = lanex::add_partner<1>(s0v); s1v = lanex::add_partner<1>(s1v);
      s0v = lanex::add_partner<2>(s0v); s1v = lanex::add_partner<2>(s1v);
      s0v = lanex::add_partner<4>(s0v); s1v = lanex::add_partner<4>(s1v);
      if (cc < 6 && kk == 0) { coh_store(a.cparts + (size_t)cc * nwg + u, s0v); coh_store(a.cparts + (size_t)(6 + cc) * nwg + u, s1v); }
    }
  };
  // component tq % 6 of node tq / 6: the first-node parts of the 4 units of interval n plus the second-node parts of the 4 units of interval n - 1
  // (valid after the exchange that follows coarse_restrict)
  auto coarse_gather = [&]() {
    double sgm = 0;
    if (coarse && tq < nca) {
      const int n = tq / 6, cc = tq % 6;
      const double* c0 = a.cparts + (size_t)cc * nwg + aggu * n;
      const double* c1 = a.cparts + (size_t)(6 + cc) * nwg + aggu * (n - 1);
      for (int mm = 0; mm < aggu; mm++) {
        if (n < a.na && aggu * n + mm < nwg) sgm += coh_load(c0 + mm);
        if (n >= 1 && aggu * (n - 1) + mm < nwg) sgm += coh_load(c1 + mm);
      }
    }
    return sgm;
  };
  // PCG start: x = 0, r = bs, z = W r, p_{-1} = 0
  if (t < N) { xs[t] = 0; ps[t] = 0; qs[t] = 0; zs[t] = 0; rc[t] = (t < m) ? d.bs[6 * (size_t)s0 + t] : 0.0; }   // xs/ps/zs: own rows; rc/qs: cluster
  __syncthreads();
  PERS_TICK(10)
  // the Li region is dead now: off[32] | loc[kPersIdxCap] | blk[kPersIdxCap] | ucol[kPersColCap] ints, then p of the
  // neighbour columns [6 * kPersColCap] doubles and the two half-row sums [16][2][8]
  int* l_off = reinterpret_cast<int*>(Li);
  int* l_loc = l_off + 32;
  uint32_t* l_blk = reinterpret_cast<uint32_t*>(l_loc + kPersIdxCap);
  int* l_ucol = reinterpret_cast<int*>(l_blk + kPersIdxCap);
  double* pl = reinterpret_cast<double*>(l_ucol + kPersColCap);
  double* half_sum = pl + 6 * kPersColCap;
  const int e_base = has ? d.row_off[o0] : 0;
  const int n_ent = has ? d.row_off[o1] - e_base : 0;
  const int u_base = has ? a.uoff[u] : 0;
  const int nu = has ? a.uoff[u + 1] - u_base : 0;
  if (has) {
    if (t <= nown) l_off[t] = d.row_off[o0 + t] - e_base;
    for (int s = t; s < n_ent; s += kPersTPB) { l_loc[s] = a.loc[e_base + s]; l_blk[s] = d.row_blk[e_base + s]; }
    for (int s = t; s < nu; s += kPersTPB) l_ucol[s] = a.ucol[u_base + s];
  }
  if (t < 6) pl[t] = 0.0;   // padding entries of the register-resident rows point at column slot 0
  __syncthreads();
  // ---- the S blocks of the own rows go into REGISTERS once: two waves per row, 10 entries in flight per wave (60 of
  // the 64 lanes: with 8 groups of 8 lanes a quarter of the lanes idled and the same 100-block capacity cost 12 more
  // VGPRs, which the 128-register budget of a 16-wave workgroup does not have), lane (g, r) = (lane / 6, lane % 6)
  // keeps row r of its entries (transposition resolved here).  Rows longer than 20*kPersRegEnt blocks
  // read the tail from global memory every iteration.
  constexpr int kPersRegEnt = 5;
  const int row_l = wv >> 1, half = wv & 1, g = lane / 6, r = lane % 6;
  const bool row_ok = has && row_l < nown && g < 10;
  double sreg[kPersRegEnt][6];
  int jreg[kPersRegEnt];
  {
    const int e0 = row_ok ? l_off[row_l] : 0;
    const int e_end = row_ok ? l_off[row_l + 1] : 0;
#pragma unroll
    for (int k = 0; k < kPersRegEnt; k++) {
      const int s = e0 + half * 10 + g + 20 * k;
      const bool v = row_ok && s < e_end;
      jreg[k] = v ? 6 * l_loc[s] : 0;
      const uint32_t bt = v ? l_blk[s] : 0u;
      const double* B = d.S + 36 * (size_t)(bt & ~kTransposeBit);
#pragma unroll
      for (int q = 0; q < 6; q++) sreg[k][q] = v ? ((bt & kTransposeBit) ? B[q * 6 + r] : B[r * 6 + q]) : 0.0;
    }
  }
  auto apply_W = [&]() {          // z(own rows) = W[own rows, :] rc (+ coarse correction); publishes them; returns this thread's share of r.z
    {
      const int row = tq % (N / 2), prt = tq / (N / 2);       // 48 rows x 8 column parts (12 columns each) = waves 0..5
      if (prt < 8) {
        double sv = 0;
        if (row < mo) {
          const int c0 = prt * 12;
#pragma unroll
          for (int col = 0; col < 12; col++) sv += A[(c0 + col) * (N / 2) + row] * rc[c0 + col];
        }
        zpart[prt * (N / 2) + row] = sv;
      } else if (coarse && wv < 12) {                       // waves 6..11: y of component wv - 6 of both nodes = Ac^-1[row] . coarse residual
        const float* ar0 = ainv_l + (wv - 6) * Nc;
        const float* ar1 = ainv_l + (wv - 6 + 6) * Nc;
        double acc0 = 0, acc1 = 0;
        for (int j = lq; j < nca; j += kWave) { const double rv = rco[j]; acc0 += (double)ar0[j] * rv; acc1 += (double)ar1[j] * rv; }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1);
        if (lq == 0) { ypart[wv - 6] = acc0; ypart[wv - 6 + 6] = acc1; }
      }
    }
    __syncthreads();
    double rz = 0;
    if (tq < mo) {
      double z = zpart[tq];
#pragma unroll
      for (int q = 1; q < 8; q++) z += zpart[q * (N / 2) + tq];
      if (coarse) {
        const int kk = tq / 6, rr = tq % 6;
        const double w1 = coarse_hat_t(8 * u + kk, d.agg), w0 = 1.0 - w1;
#pragma unroll
        for (int cc = 0; cc < 6; cc++) z += pown[kk * 36 + rr * 6 + cc] * (w0 * ypart[cc] + w1 * ypart[6 + cc]);
      }
      zs[tq] = z;
      coh_store(d.z + 6 * (size_t)o0 + tq, z);
      rz = rc[ob + tq] * z;
    }
    return rz;
  };
  int fail = 0, k = 0;
  double rz = 0;
  bool alive = true;
  if (coarse) {   // coarse residual of r0 = b: one extra exchange before the first preconditioner application
    coarse_restrict(rc + ob);
    double dummy = 0;
    alive = pers_exchange(tq, slots_rz, nwg, u, 0.0, false, ++epoch, a.bar + 1, red, &dummy, ibuf + 2);
    const double cg0 = coarse_gather();
    if (t < nca) rco[t] += cg0;
    __syncthreads();
  }
  {
    const double rz_t = apply_W();
    if (t < mo) coh_store(d.p[0] + 6 * (size_t)o0 + t, 0.0);
    const bool alive2 = pers_exchange(tq, slots_rz, nwg, u, rz_t, has && ibuf[1], ++epoch, a.bar + 1, red, &rz, ibuf + 2);   // bad pivot -> NaN -> grid-wide failure
    alive = alive && alive2;
  }
  const double rz0 = rz;
  const double thresh = a.rel_tol * a.rel_tol;
  double rz_prev = rz;
  if (!alive) fail = 1;
  PERS_TICK(11)
  for (; alive && k < a.max_it; k++) {
    tq = t; asm volatile("" : "+v"(tq)); lq = tq & (kWave - 1);
    if (rz <= thresh * rz0 || !(rz > 0.0)) { if (rz != rz) fail = 1; break; }
    const double beta = pers_uniform((k == 0) ? 0.0 : rz / rz_prev);
    const double* pold = d.p[k & 1];
    double* pnew = d.p[(k + 1) & 1];
    // ---- p = z + beta p_old of every neighbour column, once, into LDS ----
    for (int idx = tq; idx < 6 * nu; idx += kPersTPB) {
      const int j = l_ucol[idx / 6], q = idx % 6;
      pl[idx] = coh_load(d.z + 6 * (size_t)j + q) + beta * coh_load(pold + 6 * (size_t)j + q);
    }
    __syncthreads();
    PERS_TICK(0)
    // ---- q = (S + lambda I) p for the own rows: registers x LDS ----
    {
      double acc = 0;
#pragma unroll
      for (int kk = 0; kk < kPersRegEnt; kk++) {
        // tie the entry's LDS address to the running sum: left alone, the compiler hoists all 30 p loads (60 VGPRs) above
        // the first multiply and pushes half of the S registers into scratch; the other three waves of the SIMD cover
        // the LDS latency of one entry at a time
        int jo = jreg[kk];
        asm volatile("" : "+v"(jo), "+v"(acc));
        const double* pj = pl + jo;
#pragma unroll
        for (int q = 0; q < 6; q++) acc += sreg[kk][q] * pj[q];
      }
      {   // tail of very long rows (bounds re-read from LDS: two broadcast loads instead of two live registers)
        const int gq = lq / 6, rq = lq - 6 * gq;
        const bool rowq = has && row_l < nown && gq < 10;
        const int e_endq = rowq ? l_off[row_l + 1] : 0;
        for (int s = (rowq ? l_off[row_l] : 0) + half * 10 + gq + 20 * kPersRegEnt; s < e_endq; s += 20) {
          const uint32_t bt = l_blk[s];
          const double* B = d.S + 36 * (size_t)(bt & ~kTransposeBit);
          const double* pj = pl + 6 * l_loc[s];
#pragma unroll
          for (int q = 0; q < 6; q++) acc += ((bt & kTransposeBit) ? B[q * 6 + rq] : B[rq * 6 + q]) * pj[q];
        }
      }
      acc += __shfl_down(acc, 30, kWave);                    // groups g and g + 5
      const double pr = acc + __shfl_down(acc, 6, kWave);     // (0,1) at g = 0, (2,3) at g = 2
      acc = (pr + __shfl_down(pr, 12, kWave)) + __shfl_down(acc, 24, kWave);
      if (lq < 6) half_sum[(row_l * 2 + half) * 8 + lq] = acc;
    }
    __syncthreads();
    double pq_t = 0;
    if (tq < mo) {
      const int rw = tq / 6, cc = tq % 6;
      const double pi = zs[tq] + beta * ps[tq];
      const double qv = (half_sum[(rw * 2) * 8 + cc] + half_sum[(rw * 2 + 1) * 8 + cc]) + lambda * pi;
      ps[tq] = pi;
      qs[ob + tq] = qv;
      coh_store(pnew + 6 * (size_t)o0 + tq, pi);
      coh_store(d.q + 6 * (size_t)o0 + tq, qv);     // the partner unit needs it for its copy of r
      pq_t = pi * qv;
    }
    if (coarse) { __syncthreads(); coarse_restrict(qs + ob); }
    PERS_TICK(1)
    double pq = 0;
    long long tw0 = 0;
    if (a.dbg && t == 0) tw0 = wall_clock64();
    alive = pers_exchange(tq, slots_pq, nwg, u, pq_t, false, ++epoch, a.bar + 1, red, &pq, ibuf + 2);
    if (a.dbg && t == 0) a.dbg[32 + 2 * u] += wall_clock64() - tw0;   // per unit: time inside exchange 1 (its own wait for the slowest unit + the exchange latency)
    PERS_TICK(2)
    if (!alive) { fail = 1; break; }
    const double q_partner = (tq < m && !(tq >= ob && tq < ob + mo)) ? coh_load(d.q + 6 * (size_t)s0 + tq) : 0.0;
    const double cg = coarse_gather();       // P^T q of every aggregate
    if (!(pq > 0.0)) { fail = 1; break; }   // not positive definite (or NaN): solver failure -> LM rejects the step
    PERS_TICK(3)
    const double alpha = pers_uniform(rz / pq);
    if (tq < mo) xs[tq] += alpha * ps[tq];
    if (tq < m) rc[tq] -= alpha * ((tq >= ob && tq < ob + mo) ? qs[tq] : q_partner);
    if (coarse && tq < nca) rco[tq] -= alpha * cg;  // P^T r follows the recurrence of r: no second gather per iteration
    __syncthreads();
    const double rz_t2 = apply_W();
    PERS_TICK(4)
    rz_prev = rz;
    if (a.dbg && t == 0) tw0 = wall_clock64();
    alive = pers_exchange(tq, slots_rz, nwg, u, rz_t2, false, ++epoch, a.bar + 1, red, &rz, ibuf + 2);
    if (a.dbg && t == 0) a.dbg[33 + 2 * u] += wall_clock64() - tw0;
    PERS_TICK(5)
    if (!alive) { fail = 1; break; }
    PERS_TICK(6)
  }
#undef PERS_TICK
  if (timing) { for (int q = 0; q < 12; q++) a.dbg[q] += tacc[q]; a.dbg[12] += k; a.dbg[13] += 1; }
  if (t < mo) d.x[6 * (size_t)o0 + t] = xs[t];
  if (blockIdx.x == 0 && t == 0) {
    d.pcg_flag[0] = 1; d.pcg_flag[1] = k; d.pcg_flag[2] = fail;
    d.pcg_flag[3] = ibuf[2] | (int)__hip_atomic_load(a.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an exchange timed out: not a numeric failure
  }
}

// Multi-kernel PCG start with the tile factorisation of the persistent kernel (one 16-wave workgroup per cluster): assembles
// and inverts the damped cluster block, stores W for ba_pcg_update, sets x = 0, r = b, z = W r, p = 0 and the r.z partial.
// Its column-by-column predecessor took 1.43 ms per trial on 625 clusters (96 steps of three barriers); this one ~0.1 ms.
__global__ __launch_bounds__(kPersTPB) void ba_pcg_init_tiles(BaDev d, double lambda, double rel_tol, const int* coff, const int* cij, const uint32_t* cblk) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int N = kCluN;
  double* A = sm;
  double* Li = sm + N * N;
  double* rc = Li + N * N;                   // [96]
  double* red = rc + N;                      // [16]
  int* ibuf = reinterpret_cast<int*>(red + kPersWaves);
  const int t = threadIdx.x, c = blockIdx.x;
  const int s0 = c * kClu, s1 = min(d.Cp, s0 + kClu);
  const int m = 6 * (s1 - s0);
  if (t < 4) ibuf[t] = 0;
  __syncthreads();
  pers_factor_cluster(A, Li, ibuf, cij, cblk, coff[c], coff[c + 1], d.S, s0, s1, lambda, true, nullptr, false);
  float* W = d.Wc + (size_t)c * N * N;
  for (int e = t; e < N * N; e += kPersTPB) W[e] = (float)A[e];
  if (t < N) rc[t] = (t < m) ? d.bs[6 * (size_t)s0 + t] : 0.0;
  __syncthreads();
  double rz = 0;
  if (t < m) {
    double z = 0;
    for (int col = 0; col < m; col++) z += (double)(float)A[col * N + t] * rc[col];   // W is symmetric: column access is conflict-free; rounded as ba_pcg_update will read it (ONE preconditioner for the whole solve)
    const size_t g = 6 * (size_t)s0 + t;
    d.x[g] = 0; d.r[g] = rc[t]; d.z[g] = z; d.p[0][g] = 0;
    rz = rc[t] * z;
  }
  rz = wave_sum(rz);
  if ((t & (kWave - 1)) == 0) red[t / kWave] = rz;
  __syncthreads();
  if (t == 0) {
    double tot = 0;
    for (int w = 0; w < kPersWaves; w++) tot += red[w];
    d.prz[0][c] = tot;
    d.prz[1][c] = 0;
    if (ibuf[1]) d.pcg_flag[2] = 1;
    if (c == 0) { d.pcg_scal[1] = rel_tol * rel_tol; d.pcg_scal[2] = lambda; }
  }
  if (d.mk_on) mk_restrict(d, c, s0, m, rc, Li);
}

static inline size_t pers_lds_bytes() {
  return (size_t)(2 * kCluN * kCluN + 5 * kCluN + 8 * kCluN + kPersWaves + 4) * sizeof(double) + 16 + 14 * sizeof(long long);
}

// ---- very small reduced systems (<= 16 free cameras: initial maps, tiny local windows): the whole PCG in ONE workgroup ----
// Vectors and the 6x6 block-Jacobi preconditioner live in LDS and an iteration is a few block barriers (~2 us).  Typical
// local-BA sizes (tens of cameras) go through the persistent kernel instead: its 16-camera cluster preconditioner needs
// a third of the iterations (lba_c2: 59 -> 19 per solve).

template <int TPB>
__device__ __forceinline__ double block_dot_small(double v, double* red /* [16] */) {
  v = wave_sum(v);
  if (TPB == kWave) return v;   // single wave: the butterfly already left the full sum in every lane
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int i = 0; i < TPB / kWave; i++) s += red[i];
  return s;   // identical in every thread
}

template <int kSmallTPB>
__global__ __launch_bounds__(kSmallTPB) void ba_pcg_small(BaDev d, double lambda, double rel_tol, int max_it, int stage_S, int n_entries) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = 6 * d.Cp;
  double* xs = sm; double* rs = xs + n; double* zs = rs + n; double* ps = zs + n; double* qs = ps + n;
  double* Mi = qs + n;               // [Cp][36]
  double* red = Mi + 36 * d.Cp;      // [16] + flag word
  // optional LDS copy of the whole reduced matrix and its block-CSR structure (local-BA sizes fit)
  double* Sl = red + 18;             // [(Cp+nOff)*36]
  int* rowoff_l = reinterpret_cast<int*>(Sl + (stage_S ? 36 * (size_t)(d.Cp + d.nOff) : 0));
  int* rowcol_l = rowoff_l + (d.Cp + 1);
  uint32_t* rowblk_l = reinterpret_cast<uint32_t*>(rowcol_l + n_entries);
  const double* Sm = d.S; const int* roff = d.row_off; const int* rcol = d.row_col; const uint32_t* rblk = d.row_blk;
  if (stage_S) {
    for (int i = threadIdx.x; i < 36 * (d.Cp + d.nOff); i += kSmallTPB) Sl[i] = d.S[i];
    for (int i = threadIdx.x; i <= d.Cp; i += kSmallTPB) rowoff_l[i] = d.row_off[i];
    for (int i = threadIdx.x; i < n_entries; i += kSmallTPB) { rowcol_l[i] = d.row_col[i]; rowblk_l[i] = d.row_blk[i]; }
    Sm = Sl; roff = rowoff_l; rcol = rowcol_l; rblk = rowblk_l;
  }
  // all LDS in the dynamic region (a static __shared__ in front of it would misalign the f64 arrays, guide G17)
  int& fail_s = *reinterpret_cast<int*>(red + 16);
  const int t = threadIdx.x, lane = t & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(t / kWave);
  if (t == 0) fail_s = 0;
  __syncthreads();
  if (t < d.Cp) {
    double A[36], Inv[36];
    const double* S = d.S + 36 * (size_t)t;
#pragma unroll
    for (int k = 0; k < 36; k++) A[k] = S[k];
#pragma unroll
    for (int k = 0; k < 6; k++) A[k * 7] += lambda;
    if (!ba_spd6_inv(A, Inv)) { fail_s = 1; for (int k = 0; k < 36; k++) Inv[k] = (k % 7 == 0) ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < 36; k++) Mi[36 * t + k] = Inv[k];
  }
  for (int i = t; i < n; i += kSmallTPB) { xs[i] = 0; rs[i] = d.bs[i]; }
  __syncthreads();
  for (int i = t; i < n; i += kSmallTPB) {
    const int row = i / 6, a = i % 6;
    double s = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) s += Mi[36 * row + a * 6 + c] * rs[6 * row + c];
    zs[i] = s; ps[i] = s;
  }
  __syncthreads();
  double part = 0;
  for (int i = t; i < n; i += kSmallTPB) part += rs[i] * zs[i];
  double rz = block_dot_small<kSmallTPB>(part, red);
  const double rz0 = rz;
  int k = 0, fail = fail_s;
  for (; k < max_it; k++) {
    if (rz <= rel_tol * rel_tol * rz0 || !(rz > 0.0)) { if (rz != rz) fail = 1; break; }
    // q = (S + lambda I) p : one wave per block row, 8 blocks in flight (same lane mapping as ba_pcg_spmv)
    for (int i = wv; i < d.Cp; i += kSmallTPB / kWave) {
      const int g = lane >> 3, r = lane & 7;
      double acc = 0;
      if (r < 6) {
        for (int s = roff[i] + g; s < roff[i + 1]; s += 8) {
          const int j = rcol[s];
          const uint32_t bt = rblk[s];
          const double* B = Sm + 36 * (size_t)(bt & ~kTransposeBit);
          if (bt & kTransposeBit) {
#pragma unroll
            for (int c = 0; c < 6; c++) acc += B[c * 6 + r] * ps[6 * j + c];
          } else {
#pragma unroll
            for (int c = 0; c < 6; c++) acc += B[r * 6 + c] * ps[6 * j + c];
          }
        }
      }
      acc = lanex::add_partner<8>(acc); acc = lanex::add_partner<16>(acc); acc = lanex::add_partner<32>(acc);
      if (lane < 6) qs[6 * i + lane] = acc + lambda * ps[6 * i + lane];
    }
    __syncthreads();
    part = 0;
    for (int i = t; i < n; i += kSmallTPB) part += ps[i] * qs[i];
    const double pq = block_dot_small<kSmallTPB>(part, red);
    if (!(pq > 0.0)) { fail = 1; break; }
    const double alpha = rz / pq;
    for (int i = t; i < n; i += kSmallTPB) { xs[i] += alpha * ps[i]; rs[i] -= alpha * qs[i]; }
    __syncthreads();
    part = 0;
    for (int i = t; i < n; i += kSmallTPB) {
      const int row = i / 6, a = i % 6;
      double s = 0;
#pragma unroll
      for (int c = 0; c < 6; c++) s += Mi[36 * row + a * 6 + c] * rs[6 * row + c];
      zs[i] = s;
      part += rs[i] * s;
    }
    const double rz_new = block_dot_small<kSmallTPB>(part, red);
    const double beta = rz_new / rz;
    for (int i = t; i < n; i += kSmallTPB) ps[i] = zs[i] + beta * ps[i];
    rz = rz_new;
    __syncthreads();
  }
  for (int i = t; i < n; i += kSmallTPB) d.x[i] = xs[i];
  if (t == 0) { d.pcg_flag[0] = 1; d.pcg_flag[1] = k; d.pcg_flag[2] = fail; }
}

// ---- apply the step to a trial state, chi2 of the trial ---------------------------------------
// cameras: T_trial = exp(dx) * T ; partial of sum x (lambda x + b_p)                   [CCM_K_BA_UPDATE]
__global__ __launch_bounds__(kTPB) void ba_update_cams(BaDev d, int cur, double lambda, int add_lambda_term, unsigned* pers_flags /* nullable */) {
  __shared__ double lds[kTPB / kWave];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (pers_flags && i < 4) pers_flags[i] = 0u;   // abort flag of the persistent PCG kernel, cleared for the next trial (saves a 5 us memset launch)
  double sc = 0;
  if (i < d.Cp) {
    const int c = d.slot_cam[i];
    double u[6];
#pragma unroll
    for (int a = 0; a < 6; a++) u[a] = d.x[6 * (size_t)i + a];
    const BaPose T = ba_load_pose(d.cam[cur] + 7 * (size_t)c);
    const BaPose Tn = ba_oplus(u, T);
    ba_store_pose(d.cam[cur ^ 1] + 7 * (size_t)c, Tn);
#pragma unroll
    for (int a = 0; a < 6; a++) sc += u[a] * ((add_lambda_term ? lambda * u[a] : 0.0) + d.bp[6 * (size_t)i + a]);
  }
  const double s = block_sum(sc, lds);
  if (threadIdx.x == 0) d.part_cam[blockIdx.x] = s;
}

// landmarks: dX = Dinv (b_l - sum_e W_e^T dx_cam) ; X_trial = X + dX ; robust chi2 of the own
// edges at the trial state (block_solver.hpp:461-481, sparse_optimizer.cpp:61-114,422-435) [CCM_K_BA_BACKSUB]
__global__ __launch_bounds__(kTPB) void ba_backsub_chi2(BaDev d, int cur, double lambda, int chi2_only) {
  __shared__ double lds[kTPB / kWave];
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  double chi = 0, sc = 0;
  if (l < d.Lloc) {
    const int e0 = d.pt_off[l], e1 = d.pt_off[l + 1];
    double X[3] = {d.pt[cur][3 * (size_t)l], d.pt[cur][3 * (size_t)l + 1], d.pt[cur][3 * (size_t)l + 2]};
    const int trial = chi2_only ? cur : (cur ^ 1);
    if (!chi2_only) {
      double cl[3] = {d.bl[3 * (size_t)l], d.bl[3 * (size_t)l + 1], d.bl[3 * (size_t)l + 2]};
      const double bl0 = cl[0], bl1 = cl[1], bl2 = cl[2];
      for (int e = e0; e < e1; e++) {
        const int cs = d.ed_cslot[e];
        if (cs < 0) continue;
        const double* W = d.W + 18 * (size_t)e;
        const double* xp = d.x + 6 * (size_t)cs;
#pragma unroll
        for (int r = 0; r < 6; r++) { cl[0] -= W[r * 3] * xp[r]; cl[1] -= W[r * 3 + 1] * xp[r]; cl[2] -= W[r * 3 + 2] * xp[r]; }
      }
      const double* Di = d.Dinv + 6 * (size_t)l;
      const double dx0 = Di[0] * cl[0] + Di[1] * cl[1] + Di[2] * cl[2];
      const double dx1 = Di[1] * cl[0] + Di[3] * cl[1] + Di[4] * cl[2];
      const double dx2 = Di[2] * cl[0] + Di[4] * cl[1] + Di[5] * cl[2];
      sc = dx0 * (lambda * dx0 + bl0) + dx1 * (lambda * dx1 + bl1) + dx2 * (lambda * dx2 + bl2);
      X[0] += dx0; X[1] += dx1; X[2] += dx2;
      d.pt[trial][3 * (size_t)l] = X[0]; d.pt[trial][3 * (size_t)l + 1] = X[1]; d.pt[trial][3 * (size_t)l + 2] = X[2];
    }
    for (int e = e0; e < e1; e++) {
      const int c = d.ed_cam[e];
      const BaPose T = ba_load_pose(d.cam[trial] + 7 * (size_t)c);
      const double K4[4] = {d.K[4 * c], d.K[4 * c + 1], d.K[4 * c + 2], d.K[4 * c + 3]};
      double r0, r1;
      const double zc = ba_residual(T, K4, X, d.obs[2 * (size_t)e], d.obs[2 * (size_t)e + 1], r0, r1);
      const double c2 = d.info[e] != 0.0 ? (r0 * r0 + r1 * r1) * d.info[e] : 0.0;   // a deactivated edge adds an exact zero whatever its residual is
      double rho0, w;
      ba_huber(c2, d.huber, rho0, w);
      chi += rho0;
      if (d.info[e] != 0.0) d.edge_chi2[e] = c2;   // an edge moved to level 1 (ccm_ba_set_edge_levels) keeps the chi2 of the pass before, like g2o's _error
      d.edge_depth[e] = zc > 0.0;
    }
  }
  const double s0 = block_sum(chi, lds);
  const double s1 = block_sum(sc, lds);
  if (threadIdx.x == 0) { d.part_pt[2 * blockIdx.x] = s0; d.part_pt[2 * blockIdx.x + 1] = s1; }
}

// Edge-parallel variant (default, see ba_linearize_pts_e): one thread per observation forms W_e^T dp and later the
// trial's residual, one thread per landmark does the 3x3 back-substitution in between.  (The W^T dp products of an
// observation are summed before they are subtracted from b_l, so the landmark step differs from the loop above in the
// last bits; the order is fixed, results are reproducible.)
__global__ __launch_bounds__(kTPB) void ba_backsub_chi2_e(BaDev d, int cur, double lambda, int chi2_only) {
  __shared__ double v[kTPB][3], Xs[kTPB][3];
  __shared__ double lds[kTPB / kWave];
  const int l0 = d.chunk_off[blockIdx.x], l1 = d.chunk_off[blockIdx.x + 1];
  const int e0 = d.pt_off[l0], ne = d.pt_off[l1] - e0, nl = l1 - l0;
  const int t = threadIdx.x;
  const int trial = chi2_only ? cur : (cur ^ 1);
  if (!chi2_only) {
    if (t < ne) {
      const int e = e0 + t, cs = d.ed_cslot[e];
      double a0 = 0, a1 = 0, a2 = 0;
      if (cs >= 0 && d.w_free) {
        // W^T dx = (wom Ji)^T (Jj dx) from the compact record and the camera's record: 32 coalesced bytes + 96 bytes of a 192 KB table instead of 144
        const ba_v2d* Ep = reinterpret_cast<const ba_v2d*>(d.E4L + 4 * (size_t)e);
        const ba_v2d* rk = reinterpret_cast<const ba_v2d*>(d.camRK + 12 * (size_t)cs);
        const double* xp = d.x + 6 * (size_t)cs;
        const ba_v2d ea = Ep[0], eb = Ep[1];
        ba_v2d r2[6];
#pragma unroll
        for (int k = 0; k < 6; k++) r2[k] = rk[k];
        double wj0[3], wj1[3], pj[5], qj[5];
        ba_compact_factors(ea, eb, r2, wj0, wj1, pj, qj);
        const double x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3], x4 = xp[4], x5 = xp[5];
        const double u0 = __builtin_fma(pj[4], x5, __builtin_fma(pj[3], x3, __builtin_fma(pj[2], x2, __builtin_fma(pj[1], x1, pj[0] * x0))));
        const double u1 = __builtin_fma(qj[4], x5, __builtin_fma(qj[3], x4, __builtin_fma(qj[2], x2, __builtin_fma(qj[1], x1, qj[0] * x0))));
        a0 = __builtin_fma(wj1[0], u1, wj0[0] * u0); a1 = __builtin_fma(wj1[1], u1, wj0[1] * u0); a2 = __builtin_fma(wj1[2], u1, wj0[2] * u0);
      } else if (cs >= 0) {
        const double* W = d.W + 18 * (size_t)e;
        const double* xp = d.x + 6 * (size_t)cs;
#pragma unroll
        for (int r = 0; r < 6; r++) { a0 += W[r * 3] * xp[r]; a1 += W[r * 3 + 1] * xp[r]; a2 += W[r * 3 + 2] * xp[r]; }
      }
      v[t][0] = a0; v[t][1] = a1; v[t][2] = a2;
    }
    __syncthreads();
  }
  double chi = 0, sc = 0;
  if (t < nl) {
    const int l = l0 + t;
    double X[3] = {d.pt[cur][3 * (size_t)l], d.pt[cur][3 * (size_t)l + 1], d.pt[cur][3 * (size_t)l + 2]};
    if (!chi2_only) {
      const double bl0 = d.bl[3 * (size_t)l], bl1 = d.bl[3 * (size_t)l + 1], bl2 = d.bl[3 * (size_t)l + 2];
      double cl[3] = {bl0, bl1, bl2};
      for (int k = d.pt_off[l] - e0; k < d.pt_off[l + 1] - e0; k++) { cl[0] -= v[k][0]; cl[1] -= v[k][1]; cl[2] -= v[k][2]; }
      const double* Di = d.Dinv + 6 * (size_t)l;
      const double dx0 = Di[0] * cl[0] + Di[1] * cl[1] + Di[2] * cl[2];
      const double dx1 = Di[1] * cl[0] + Di[3] * cl[1] + Di[4] * cl[2];
      const double dx2 = Di[2] * cl[0] + Di[4] * cl[1] + Di[5] * cl[2];
      sc = dx0 * (lambda * dx0 + bl0) + dx1 * (lambda * dx1 + bl1) + dx2 * (lambda * dx2 + bl2);
      X[0] += dx0; X[1] += dx1; X[2] += dx2;
      d.pt[trial][3 * (size_t)l] = X[0]; d.pt[trial][3 * (size_t)l + 1] = X[1]; d.pt[trial][3 * (size_t)l + 2] = X[2];
    }
    Xs[t][0] = X[0]; Xs[t][1] = X[1]; Xs[t][2] = X[2];
  }
  __syncthreads();
  if (t < ne) {
    const int e = e0 + t, c = d.ed_cam[e], ll = d.ed_pt[e] - l0;
    const double X[3] = {Xs[ll][0], Xs[ll][1], Xs[ll][2]};
    const BaPose T = ba_load_pose(d.cam[trial] + 7 * (size_t)c);
    const double K4[4] = {d.K[4 * c], d.K[4 * c + 1], d.K[4 * c + 2], d.K[4 * c + 3]};
    double r0, r1;
    const double zc = ba_residual(T, K4, X, d.obs[2 * (size_t)e], d.obs[2 * (size_t)e + 1], r0, r1);
    const double c2 = d.info[e] != 0.0 ? (r0 * r0 + r1 * r1) * d.info[e] : 0.0;   // (see ba_backsub_chi2)
    double rho0, w;
    ba_huber(c2, d.huber, rho0, w);
    chi = rho0;
    if (d.info[e] != 0.0) d.edge_chi2[e] = c2;     // (see ba_backsub_chi2)
    d.edge_depth[e] = zc > 0.0;
  }
  const double s0 = block_sum(chi, lds);
  const double s1 = block_sum(sc, lds);
  if (threadIdx.x == 0) { d.part_pt[2 * blockIdx.x] = s0; d.part_pt[2 * blockIdx.x + 1] = s1; }
}

// final reduction of the trial scalars (single workgroup, fixed order)
// stop_local / abort_local: this rank's view of the caller's stop flag and of a failed persistent-PCG launch; together with the
// kernel-side give-up flag they ride in the same all-reduce as chi2, so that every rank of a sharded run takes the same
// decision (a rank leaving the LM loop alone would leave its peers waiting in the next collective)
// h_out != nullptr (one rank): the eight read-back words also go straight to the handle's pinned host block, followed by `ticket` (system-scope release): the host polls the
// ticket instead of queueing a 64-byte copy and waiting for the stream (read_scalars_polled)
__global__ __launch_bounds__(kTPB) void ba_reduce_scalars(BaDev d, int stop_local, int abort_local, int pers_trial, double* h_out, unsigned long long ticket) {
  __shared__ double lds[kTPB / kWave];
  double chi = 0, sc = 0;
  for (int i = threadIdx.x; i < d.n_part; i += kTPB) { chi += d.part_pt[2 * i]; sc += d.part_pt[2 * i + 1]; }
  for (int i = threadIdx.x; i < d.n_wg_cam; i += kTPB) sc += d.part_cam[i];
  const double a = block_sum(chi, lds);
  const double b = block_sum(sc, lds);
  if (threadIdx.x == 0) {
    d.scal[0] = a; d.scal[1] = b;
    d.scal[2] = stop_local ? 1.0 : 0.0;
    // 1: the persistent kernel gave up on this rank (bounded spin); 4096: its LAUNCH was refused on this rank.  Summed over the ranks of a sharded handle, so every rank
    // can tell "somebody gave up" (> 0: repeat the trial on the multi-kernel solver, then cool down) from "somebody cannot launch it at all" (>= 4096: for good)
    const double s3 = abort_local ? 4096.0 : (pers_trial && d.pcg_flag[3]) ? 1.0 : 0.0;
    d.scal[3] = s3;
    if (h_out) {
      h_out[0] = a; h_out[1] = b; h_out[2] = stop_local ? 1.0 : 0.0; h_out[3] = s3;
      h_out[4] = d.scal[4]; h_out[5] = d.scal[5]; h_out[6] = d.scal[6]; h_out[7] = d.scal[7];   // [6..7]: the four PCG flags
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(h_out) + 8, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void ba_scatter_points(double* full, const double* own, int lb /* first own landmark slot */, int Lloc) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Lloc) return;
  const int g = lb + l;
  full[3 * (size_t)g] = own[3 * (size_t)l]; full[3 * (size_t)g + 1] = own[3 * (size_t)l + 1]; full[3 * (size_t)g + 2] = own[3 * (size_t)l + 2];
}

}  // namespace

// =================================================================================================
// host side
// =================================================================================================

namespace {

template <typename T>
int dev_upload(ccm_ba* ba, const std::vector<T>& v, T** out) {
  ccm_ctx* ctx = ba->ctx;
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  size_t actual = 0;
  if (int rc = ccm_pool_get(ctx, bytes, &p, &actual)) return rc;
  ba->allocs.push_back({p, actual});
  if (!v.empty()) CCM_HIP_CHECK(ctx, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  *out = (T*)p;
  return CCM_OK;
}
template <typename T>
int dev_alloc(ccm_ba* ba, size_t n, T** out, bool zero = true) {
  ccm_ctx* ctx = ba->ctx;
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  size_t actual = 0;
  if (int rc = ccm_pool_get(ctx, bytes, &p, &actual)) return rc;
  ba->allocs.push_back({p, actual});
  if (zero) CCM_HIP_CHECK(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
  *out = (T*)p;
  return CCM_OK;
}

// reduced systems with at most this many off-diagonal blocks use the one-workgroup-per-block Schur kernel; above, the row kernel (which also forms the
// diagonal blocks and b_schur).  CCM_BA_ROW_MIN_BLOCKS overrides (experiments).
static inline int row_min_blocks() { return 256; }

// S and b_schur of the current linearisation and D^-1 (one rank's part): the row kernel on large maps (it also forms the diagonal blocks and b_schur), the
// per-block kernels otherwise.  Also what the test hooks below call, so that they see the kernels the LM loop runs.
static int launch_schur(ccm_ba* ba) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  if (!d.Cp) return CCM_OK;
  if (!(d.nOff > row_min_blocks() && d.row_units_max)) {   // the row kernel also forms the diagonal blocks and b_schur
    ccm_prof_scope ps(ctx, CCM_K_BA_SCHUR_DIAG);
    hipLaunchKernelGGL(ba_schur_diag, dim3(d.n_wg_wave4), dim3(kTPB), 0, ctx->stream, d);
  }
  if (d.nOff) {
    ccm_prof_scope ps(ctx, CCM_K_BA_SCHUR_OFF);
    if (d.nOff <= row_min_blocks()) hipLaunchKernelGGL(ba_schur_off<4>, dim3(d.nOff), dim3(kTPB), 0, ctx->stream, d);
    else if (d.row_units_max) {
      const size_t lds_row = ((size_t)(d.max_cam_edges + 1) * 18 + 27 * (size_t)ccm_div_up(d.max_cam_edges, kRow2Group) + (size_t)d.row_units_max * 36) * sizeof(double);
      CCM_LDS_ATTR(ctx, CCM_LDS_BA_ROW3, ba_schur_row3, 158 * 1024);
      hipLaunchKernelGGL(ba_schur_row3, dim3(8 * ccm_div_up(d.Cp, 8)), dim3(kRow2TPB), lds_row, ctx->stream, d);
    } else hipLaunchKernelGGL(ba_schur_off<1>, dim3(ccm_div_up(d.nOff, kTPB / kWave)), dim3(kTPB), 0, ctx->stream, d);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

#define RC(x) do { int _rc = (x); if (_rc != CCM_OK) return _rc; } while (0)

// Collectives of a BA handle.  A handle created for ONE rank never enters a collective, even when its context carries a
// multi-rank communicator (bench.py runs a single-rank local BA on rank 0 of a sharded job: an all-reduce issued by that
// rank alone would wait for its peers forever).
static inline int ba_allreduce_sum(ccm_ba* ba, double* buf, size_t n) {
  if (ba->nranks <= 1 && ba->ctx->comm_nranks > 1) return CCM_OK;
  if (ba->nranks <= 1 && !ba->ctx->comm && !ba->ctx->loop_group) return CCM_OK;   // no communicator: nothing is issued (and nothing is timed)
  ccm_prof_scope ps(ba->ctx, CCM_K_BA_ALLREDUCE);
  return ccm_allreduce_f64(ba->ctx, buf, n);
}
static inline int ba_allreduce_max(ccm_ba* ba, double* buf, size_t n) {
  if (ba->nranks <= 1 && ba->ctx->comm_nranks > 1) return CCM_OK;
  if (ba->nranks <= 1 && !ba->ctx->comm && !ba->ctx->loop_group) return CCM_OK;   // no communicator: nothing is issued (and nothing is timed)
  ccm_prof_scope ps(ba->ctx, CCM_K_BA_ALLREDUCE);
  return ccm_allreduce_max_f64(ba->ctx, buf, n);
}

}  // namespace

// Host-only partition of landmark slots into nranks contiguous ranges balanced by weight
// (pair instances + edges).  Exported for the CPU tests of the sharding logic.
extern "C" int ccm_ba_partition(const int64_t* weight, int n, int nranks, int32_t* begin_out /* nranks+1 */) {
  if (!begin_out || nranks < 1 || n < 0 || (n && !weight)) return CCM_E_ARG;
  int64_t total = 0;
  for (int i = 0; i < n; i++) total += weight[i];
  begin_out[0] = 0;
  int64_t acc = 0;
  int r = 1;
  for (int i = 0; i < n && r < nranks; i++) {
    acc += weight[i];
    // close shard r-1 once it has reached r/nranks of the total weight
    while (r < nranks && acc * nranks >= total * r) { begin_out[r] = i + 1; r++; }
  }
  for (; r < nranks; r++) begin_out[r] = n;
  begin_out[nranks] = n;
  return CCM_OK;
}

// ccm_ba_create lives in ba_build.hip (structure build on the device).  Occupancy question it asks about the persistent PCG kernel: can `grid`
// workgroups be co-resident on this device?
int ccm_ba_pers_grid_fits(ccm_ctx* ctx, int grid) {
  const size_t lds = pers_lds_bytes();
  int per_cu = 0, n_cu = 0;
  const bool ok = hipFuncSetAttribute((const void*)ba_pcg_persist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)ba_pcg_persist, kPersTPB, lds) == hipSuccess &&
                  hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && per_cu * n_cu >= grid;
  (void)hipGetLastError();
  return ok ? 1 : 0;
}

extern "C" void ccm_ba_destroy(ccm_ba* ba) {
  if (!ba) return;
  if (ba->ctx) { hipSetDevice(ba->ctx->device); hipStreamSynchronize(ba->ctx->stream); }
  for (auto& pr : ba->allocs) ccm_pool_put(ba->ctx, pr.first, pr.second);
  if (ba->h_rb) { if (ba->ctx) ba->ctx->rb_free.push_back(ba->h_rb); else hipHostFree(ba->h_rb); }   // (the stream is drained: no kernel still writes its ticket there)
  delete ba;
}

int ccm_ba_state_from_raw(ccm_ba* ba);                                       // ba_build.hip
int ccm_ba_points_to_raw_order(ccm_ba* ba, const double* d_pt_slots);         // ba_build.hip

extern "C" int ccm_ba_reset_state(ccm_ba* ba, const double* cam_qt, const double* pt_xyz) {
  if (!ba || !cam_qt || (ba->n_pt && !pt_xyz)) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_raw_cam, cam_qt, 7 * (size_t)ba->n_cam * sizeof(double), hipMemcpyDefault, ctx->stream));
  if (ba->n_pt) CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_raw_pt, pt_xyz, 3 * (size_t)ba->n_pt * sizeof(double), hipMemcpyDefault, ctx->stream));
  RC(ccm_ba_state_from_raw(ba));   // SE3Quat(q, t) normalises the rotation (se3quat.h:61-63); landmarks gathered into slot order
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

// SparseOptimizer::push() / pop() over all vertices (sparse_optimizer.cpp:600-613), device to device: the saved estimate stays
// in HBM, so a benchmark can re-run the same optimisation without touching the host.
extern "C" int ccm_ba_push_state(ccm_ba* ba) {
  if (!ba) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t nc = 7 * (size_t)ba->n_cam, np = 3 * (size_t)std::max(ba->Lloc, 1);
  if (!ba->d_saved_cam) RC(dev_alloc<double>(ba, nc, &ba->d_saved_cam, false));
  if (!ba->d_saved_pt) RC(dev_alloc<double>(ba, np, &ba->d_saved_pt, false));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_saved_cam, ba->d.cam[ba->cur], nc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (ba->Lloc) CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_saved_pt, ba->d.pt[ba->cur], 3 * (size_t)ba->Lloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return CCM_OK;
}
extern "C" int ccm_ba_pop_state(ccm_ba* ba) {
  if (!ba) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  if (!ba->d_saved_cam || !ba->d_saved_pt) return ccm_set_error(ctx, CCM_E_STATE, "ccm_ba_pop_state: nothing pushed");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ba->cur = 0;
  for (int k = 0; k < 2; k++) {
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d.cam[k], ba->d_saved_cam, 7 * (size_t)ba->n_cam * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (ba->Lloc) CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d.pt[k], ba->d_saved_pt, 3 * (size_t)ba->Lloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  return CCM_OK;
}

extern "C" int ccm_ba_counts(const ccm_ba* ba, int64_t* n_active_edges, int64_t* n_active_pts, int64_t* n_free_cams,
                             int64_t* n_blocks, int64_t* n_pairs) {
  if (!ba) return CCM_E_ARG;
  if (n_active_edges) *n_active_edges = ba->Eloc;
  if (n_active_pts) *n_active_pts = ba->Lloc;
  if (n_free_cams) *n_free_cams = ba->Cp;
  if (n_blocks) *n_blocks = ba->Cp + ba->nOff;
  if (n_pairs) *n_pairs = ba->n_inst;
  return CCM_OK;
}

namespace {

// the trial scalars and the PCG flags in ONE copy into pinned memory (a pageable destination is staged by the runtime:
// the two copies + syncs of a trial used to leave the GPU idle for ~90 us)
int read_scalars(ccm_ba* ba, double out[6], int flags[4] = nullptr) {
  ccm_ctx* ctx = ba->ctx;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->h_rb, ba->d.scal, 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out, ba->h_rb, 6 * sizeof(double));
  if (flags) memcpy(flags, ba->h_rb + 6, 4 * sizeof(int));
  return CCM_OK;
}

// The same read-back without a copy command and without waiting for the stream: ba_reduce_scalars wrote the words and then `ticket` into the pinned block (round 4: the
// 64-byte copy + stream wait were ~25 us of every LM trial, 18 trials per 4-agent call, ~25 per local BA).  CCM_BA_POLL=0: the copy.  A ticket that does not arrive within
// two seconds falls back to the stream wait (a failed launch must not hang the caller).
bool poll_enabled() { return true; }
int read_scalars_polled(ccm_ba* ba, double out[6], int flags[4], unsigned long long ticket) {
  volatile unsigned long long* tk = reinterpret_cast<volatile unsigned long long*>(ba->h_rb) + 8;
  bool got = false;
  double t_start = 0;
  for (long spin = 0; !got; spin++) {
    if (*tk == ticket) { got = true; break; }
    if ((spin & 0xffff) == 0xffff) {
      if (spin == 0xffff) t_start = now_ms();
      else if (now_ms() - t_start > 2000.0) break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  if (!got) return read_scalars(ba, out, flags);
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  memcpy(out, ba->h_rb, 6 * sizeof(double));
  if (flags) memcpy(flags, ba->h_rb + 6, 4 * sizeof(int));
  // the ticket says the trial's kernels ran; a sticky asynchronous error (a fault in a LATER launch of this stream cannot exist yet, an earlier one would have
  // kept the ticket from arriving) is still picked up here so that it is attributed to this call, not to whoever synchronises next
  CCM_HIP_CHECK(ba->ctx, hipGetLastError());
  return CCM_OK;
}

// chi2 of the current state (computeActiveErrors + activeRobustChi2)
int eval_chi2(ccm_ba* ba, double* chi) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  {
    ccm_prof_scope ps(ctx, CCM_K_BA_CHI2);
    if (d.chunk_off) hipLaunchKernelGGL(ba_backsub_chi2_e, dim3(d.n_chunk), dim3(kTPB), 0, ctx->stream, d, ba->cur, 0.0, 1);
    else hipLaunchKernelGGL(ba_backsub_chi2, dim3(d.n_wg_pt), dim3(kTPB), 0, ctx->stream, d, ba->cur, 0.0, 1);
  }
  hipMemsetAsync(d.part_cam, 0, sizeof(double) * d.n_wg_cam, ctx->stream);
  const bool poll = ba->nranks == 1 && poll_enabled();
  const unsigned long long ticket = ++ba->rb_ticket;
  hipLaunchKernelGGL(ba_reduce_scalars, dim3(1), dim3(kTPB), 0, ctx->stream, d, ba->stop_local(), 0, 0, poll ? ba->h_rb : (double*)nullptr, ticket);
  RC(ba_allreduce_sum(ba, d.scal, 4));
  double s[6];
  if (poll) { RC(read_scalars_polled(ba, s, nullptr, ticket)); } else { RC(read_scalars(ba, s)); }
  *chi = s[0];
  ba->stop_any = s[2] > 0.0;
  return CCM_OK;
}

// lambda_next >= 0: the damping of the FIRST trial on this linearisation is already known (every LM iteration but the first): the landmark-side kernel also forms
// D^-1 and D^-1 b_l for it and lm_trial skips its ba_dinv launch for that lambda
int build_system(ccm_ba* ba, double lambda_next = -1.0) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  ba->dinv_done_lambda = -1.0;
  if (d.Lloc) {
    ccm_prof_scope ps(ctx, CCM_K_BA_LINEARIZE);
    if (d.chunk_off) {
      hipLaunchKernelGGL(ba_linearize_pts_e, dim3(d.n_chunk), dim3(kTPB), 0, ctx->stream, d, ba->cur, lambda_next);
      if (lambda_next >= 0.0) ba->dinv_done_lambda = lambda_next;
    } else hipLaunchKernelGGL(ba_linearize_pts, dim3(d.n_wg_pt), dim3(kTPB), 0, ctx->stream, d, ba->cur);
  }
  if (d.Cp) {
    ccm_prof_scope ps(ctx, CCM_K_BA_CAM);
    hipLaunchKernelGGL(ba_linearize_cams, dim3(d.n_wg_wave4), dim3(kTPB), 0, ctx->stream, d, ba->cur);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

// lambda_init = tau * max diag  (computeLambdaInit)
int max_diag(ccm_ba* ba, double* out) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  const double* hpp = d.Hpp;
  if (ba->nranks > 1 && d.Cp) {
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_hpp_full, d.Hpp, 36 * (size_t)d.Cp * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    RC(ba_allreduce_sum(ba, ba->d_hpp_full, 36 * (size_t)d.Cp));
    hpp = ba->d_hpp_full;
  }
  const int nb = std::max(1, std::min(d.n_wg_pt, 512));   // partials live in part_pt (>= 2*n_wg_pt doubles)
  hipLaunchKernelGGL(ba_maxdiag, dim3(nb), dim3(kTPB), 0, ctx->stream, d, hpp, d.part_pt, nb, 0);
  hipLaunchKernelGGL(ba_maxdiag, dim3(1), dim3(kTPB), 0, ctx->stream, d, hpp, d.part_pt, nb, 1);
  RC(ba_allreduce_max(ba, d.scal + 4, 1));
  double s[6];
  RC(read_scalars(ba, s));
  *out = s[4];
  return CCM_OK;
}

__global__ void ba_f64_to_f32(const double* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// coarse operator of the current linearisation at this lambda and its explicit inverse (needs S of the trial)
int coarse_build(ccm_ba* ba, double lambda) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  ccm_prof_scope ps(ctx, CCM_K_BA_COARSE);
  const int Nc = ba->coarse_Nc, na = ba->coarse_na, nn = na + 1;
  hipLaunchKernelGGL(ba_coarse_P, dim3(ccm_div_up(d.Cp, kTPB)), dim3(kTPB), 0, ctx->stream, d, ba->cur, ba->d_cP);
  hipLaunchKernelGGL(ba_coarse_assemble, dim3(ba->coarse_ncb), dim3(kCoarseTPB), 0, ctx->stream, d, (const double*)ba->d_cP,
                     (const int*)ba->d_cb_off, (const int*)ba->d_cb_ent, ba->coarse_ncb, (const int*)ba->d_blk_i, (const int*)ba->d_blk_j, lambda, ba->d_cstage);
  hipLaunchKernelGGL(ba_coarse_sum, dim3(ccm_div_up((int64_t)Nc * Nc, kTPB)), dim3(kTPB), 0, ctx->stream, (const double*)ba->d_cstage, (const unsigned*)ba->d_cb_key,
                     ba->coarse_ncb, na, nn, ba->d_cA, Nc);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  RC(ccm_dense_chol_inverse_dev(ctx, ba->d_cA, Nc, ba->d_cLinv, ba->d_cX, ba->d_cAinv, ba->d_cinfo));
  if (ba->d_cAinv32) {   // multi-kernel PCG: the rows the clusters re-read every CG iteration, as f32
    const size_t n = (size_t)Nc * Nc;
    hipLaunchKernelGGL(ba_f64_to_f32, dim3((unsigned)ccm_div_up((int64_t)n, kTPB)), dim3(kTPB), 0, ctx->stream, (const double*)ba->d_cAinv, ba->d_cAinv32, n);
    CCM_HIP_CHECK(ctx, hipGetLastError());
  }
  return CCM_OK;
}

int coarse_prepare(ccm_ba* ba, double lambda) {
  // lambda window inside which a stale coarse operator is kept.  Measured on gba_c4 (10 calls each): window 4: 6 builds / 926 CG iterations /
  // 21.9 ms per call, 16: 4 builds / 976 / 20.4 ms, 64: 4 builds / 982 / 20.8 ms.  The stale-iterations guard below still forces a rebuild when
  // a reused operator costs a third more iterations than a fresh one did.
  const double win = 16.0;
  // (round 4) UPWARDS the window is wider: lambda only grows along a chain of rejected trials (x2, x4, x8, ...), where every trial has its own lambda and a
  // rebuilt operator would serve that one solve — a 768-unknown build costs ~50 CG iterations, an operator built at a 10 - 60 times smaller lambda 5 - 15;
  // the stale-iterations guard still forces a rebuild when a reused operator does badly
  // (measured, one box, complete calls: gba_c4 15.6 -> 15.2 ms with 2 builds instead of 3 and 567 instead of 538 CG iterations; gba_c3 unchanged; on the
  // multi-kernel path of gba_c5, where a CG iteration costs ~100 us against a 2.5 ms build, the wider window LOSES 10 ms (1546 instead of 1438 iterations): 16 there)
  const double win_up_env = 0.0;
  const double win_up = win_up_env > 0 ? win_up_env : (ba->pers_grid ? 64.0 : win);
  const bool need = !ba->coarse_reuse || !ba->coarse_valid || ba->coarse_stale_bad || lambda > win_up * ba->coarse_lambda_built || lambda < ba->coarse_lambda_built / win;
  ba->coarse_fresh = need;
  if (!need) return CCM_OK;
  RC(coarse_build(ba, lambda));
  ba->coarse_valid = true; ba->coarse_stale_bad = false; ba->coarse_lambda_built = lambda;
  return CCM_OK;
}

// one LM trial: solve with lambda, write trial state, return tempChi, scale, ok
int lm_trial(ccm_ba* ba, double lambda, const ccm_ba_options& opt, double* temp_chi, double* scale, bool* ok, int* pcg_iters) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  const int cur = ba->cur;
  if (!ba->pers_grid && ba->pers_grid_built && ba->pers_cooldown > 0 && --ba->pers_cooldown == 0) ba->pers_grid = ba->pers_grid_built;   // back to the persistent solver after an abort
  if (d.Lloc && ba->dinv_done_lambda != lambda) {   // (the linearisation already formed D^-1 for this lambda: build_system)
    ccm_prof_scope ps(ctx, CCM_K_BA_DINV);
    hipLaunchKernelGGL(ba_dinv, dim3(d.n_wg_pt), dim3(kTPB), 0, ctx->stream, d, lambda);
  }
  ba->dinv_done_lambda = -1.0;   // the next trial on this linearisation has another lambda
  *ok = true;
  *pcg_iters = 0;
  bool small_path = false, pers_trial = false, pers_launch_failed = false, cams_updated = false;
  int small_flags[4] = {0, 0, 0, 0};
  if (d.Cp) {
    RC(launch_schur(ba));
    RC(ba_allreduce_sum(ba, ba->d_red, ba->red_count));
    // ---- PCG ----
    const double tol_default = 1e-8;   // (ccm_ba_options.pcg_rel_tol overrides; the parity tests run at 1e-8)
    const double tol = opt.pcg_rel_tol > 0 ? opt.pcg_rel_tol : tol_default;
    const int max_it = opt.pcg_max_iters > 0 ? opt.pcg_max_iters : 1000;
    int flags[4] = {0, 0, 0, 0};
    const bool dense2_on = true;
    // (round 6) 17..50 free cameras: exact Cholesky with the matrix in one CU's register file (ba_solve_cholreg).  CCM_BA_CHOLREG=0 keeps the earlier solvers
    // (two-cluster exact solve up to 32 cameras, persistent PCG above) for A / B measurements and for the tests that compare the paths (read when the handle is created).
    if (ba->d_cholreg_L) {   // (decided at create time: ba_build.hip, cholreg_win)
      CCM_LDS_ATTR(ctx, CCM_LDS_BA_CHOLREG, ba_solve_cholreg, cholreg_lds_bytes());
      if (!ba->cholreg_table_built) {   // structure only: once per handle (ccm_ba_set_edge_levels keeps the block structure)
        hipLaunchKernelGGL(ba_cholreg_table, dim3(1), dim3(kCRTPB), 0, ctx->stream, d, ba->d_cholreg_tab);
        ba->cholreg_table_built = true;
      }
      {
        ccm_prof_scope ps(ctx, CCM_K_BA_PCG_PERSIST);
        hipLaunchKernelGGL(ba_solve_cholreg, dim3(1), dim3(kCRTPB), cholreg_lds_bytes(), ctx->stream, d, lambda, ba->d_cholreg_L, (const int*)ba->d_cholreg_tab,
                           ccm_dbg("cholreg") ? (long long*)ba->d_cholreg_dbg : (long long*)nullptr, cur, ba->rank == 0 ? 1 : 0);
      }
      small_path = true;
      cams_updated = true;   // (the solve's launch also applied the step to the cameras)
    } else if (d.Cp > kSmallMaxCp && d.Cp <= kDense2MaxCp && ba->d_dense_T && ba->d_pers_coff && dense2_on) {
      // exact block solve in one workgroup (see ba_solve_dense2): one launch, flags read back with the trial scalars
      CCM_LDS_ATTR(ctx, CCM_LDS_BA_DENSE2, ba_solve_dense2, dense2_lds_bytes());
      {
        ccm_prof_scope ps(ctx, CCM_K_BA_PCG_PERSIST);
        hipLaunchKernelGGL(ba_solve_dense2, dim3(1), dim3(kPersTPB), dense2_lds_bytes(), ctx->stream, d, lambda, (const int*)ba->d_pers_coff, (const int*)ba->d_pers_cij,
                           (const uint32_t*)ba->d_pers_cblk, ba->d_dense_T,
                           ccm_dbg("dense2") ? (long long*)(ba->d_dense_T + kCluN * kCluN) : (long long*)nullptr, cur, ba->rank == 0 ? 1 : 0);
      }
      small_path = true;
      cams_updated = true;   // (the solve's launch also applied the step to the cameras)
    } else if (d.Cp <= kSmallMaxCp) {
      // one launch, no host round trip: the flags are read back after the trial kernels are queued
      size_t lds = (size_t)(5 * 6 * d.Cp + 36 * d.Cp + 18) * sizeof(double);
      const size_t lds_S = 36 * (size_t)(d.Cp + d.nOff) * sizeof(double) + ((size_t)d.Cp + 1 + 2 * (size_t)ba->n_row_entries) * sizeof(int) + 16;
      const int stage_S = (lds + lds_S <= 150 * 1024) ? 1 : 0;
      if (stage_S) lds += lds_S;
      CCM_LDS_ATTR(ctx, CCM_LDS_BA_SMALL, ba_pcg_small<1024>, 150 * 1024);
      {
        ccm_prof_scope ps(ctx, CCM_K_BA_PCG_SPMV);
        // measured on lba_c2 (30 free cameras, ~64 PCG iterations per solve): 16 waves 460 us, 1 wave 1420 us — the
        // row products are LDS-latency chains, so more waves (rows in flight) win even with barriers
        hipLaunchKernelGGL(ba_pcg_small<1024>, dim3(1), dim3(1024), lds, ctx->stream, d, lambda, tol, max_it, stage_S, (int)ba->n_row_entries);
      }
      small_path = true;
    } else {
    bool persist_ok = false;
    if (ba->pers_grid) {
      // whole solve in one launch; flags are read back together with the trial scalars
      PersArgs pa;
      pa.lambda = lambda; pa.rel_tol = tol; pa.max_it = max_it; pa.n_clu = ccm_div_up(d.Cp, kClu);
      pa.bar = ba->d_pers_bar; pa.slots = (unsigned long long*)ba->d_pers_part;
      pa.epoch_base = (++ba->pers_launch) << 20;
      pa.uoff = ba->d_pers_uoff; pa.ucol = ba->d_pers_ucol; pa.loc = ba->d_pers_loc;
      pa.coff = ba->d_pers_coff; pa.cij = ba->d_pers_cij; pa.cblk = ba->d_pers_cblk;
      pa.test_abort = getenv("CCM_BA_TEST_ABORT") ? 1 : 0;
      pa.dbg = ccm_dbg("pers") ? (long long*)(ba->d_pers_bar + 4) : nullptr;
      pa.Ainv = nullptr; pa.Pm = nullptr; pa.na = 0; pa.Nc = 0; pa.cparts = nullptr;
      // Cluster inverse across trials.  Offline (1000-keyframe 4-agent reduced systems, numpy PCG to 1e-8): W built at a lambda 10 / 100 / 1000 times away
      // costs 0-2 / 2-3 / 5 more CG iterations of 24-46; W of the INITIAL linearisation used three LM iterations later costs 13-19 more (the robust weights move),
      // later linearisations hardly differ (the map itself disagreed, see the measurements below).  Policy: reuse inside an LM iteration (rejected trials: only lambda moved) within a lambda window, and across
      // iterations as long as the solve that first used a carried-over inverse did not need noticeably more iterations than the last fresh one (the counts
      // are deterministic, so the decision is — on every rank of a sharded run alike).  CCM_BA_W_REUSE=0 never, 1 (default) inside an iteration only, 2 adaptive across iterations.
      // MEASURED on gba_c4 (CCM_BA_TRIAL_DBG, one box): a launch that loads W is 55-65 us shorter than one that factors (589 against 646 us at 39 CG iterations);
      // inside an LM iteration a W built at lambda / 2 ... lambda / 8 costs no iteration, at lambda / 64 three (28 against 25), at lambda / 32 on a lambda-dominated
      // system (1.4e4) ten (21 against 11); ACROSS linearisations it is erratic: 32 against 31, 41 / 41, 43 / 43, but 87 against 48 (second linearisation), 58 / 43,
      // 66 / 41 — and every bad solve also trips the coarse level's stale guard (6 builds instead of 4): 19.2 ms per call against 17.0.  Hence the default:
      // mode 1 (same linearisation only) with a window of 8: 3 of the 7 rejected trials of gba_c4 load, ~0.2 ms of 17.
      const int w_mode = 1;
      const double w_win = 8.5;
      pa.wsave = ba->d_pers_wsave;
      {
        const bool same_lin = ba->w_lin_id == ba->lin_id;
        const bool in_win = ba->w_lambda_built > 0 && lambda <= w_win * ba->w_lambda_built && lambda >= ba->w_lambda_built / w_win;
        const bool reuse = ba->d_pers_wsave && w_mode > 0 && ba->w_valid && in_win && !pa.test_abort && (same_lin || (w_mode >= 2 && !ba->w_stale_bad));
        pa.w_load = reuse ? 1 : 0;
        ba->w_loaded = reuse;
        // (advisor, round 4) what this launch will leave in wsave becomes reusable only AFTER its flags have been read and show a clean solve (below): until
        // then no inverse is on offer, so a repeat of this trial on the multi-kernel path, a failed launch or a bad pivot can never hand a half-written or
        // patched W to a later trial
        if (!reuse) { ba->w_valid = false; ba->w_pending = ba->d_pers_wsave != nullptr; ba->w_lambda_built = lambda; ba->w_lin_id = ba->lin_id; ba->w_stale_bad = false; }
      }
      // (round 4) a strongly damped system does not need the coarse level: above the call's first lambda (g2o's 1e-5 max diag(H): the scale at which the damping
      // takes over the smooth modes as well) cluster-Jacobi alone converges in 10-24 iterations on the 4-agent map, while a stale coarse operator carried up
      // there from a 16 times smaller lambda needed 46 and a fresh build costs ~50 iterations' worth.  CCM_BA_COARSE_LMAX scales the limit (0 = no limit).
      const double coarse_lmax = 1.0;
      const bool damped = ba->coarse_force == 0 && coarse_lmax > 0 && ba->lambda_first > 0 && lambda >= coarse_lmax * ba->lambda_first;
      const bool use_coarse = ba->coarse_na && !damped && (ba->coarse_force > 0 || (ba->coarse_force == 0 && ba->coarse_active));
      ba->coarse_used = use_coarse;
      ba->coarse_skipped_damped = damped;
      if (use_coarse) {
        if (ccm_dbg("coarse")) {
          hipStreamSynchronize(ctx->stream); const double tc0 = now_ms();
          RC(coarse_prepare(ba, lambda));
          const double tc1 = now_ms(); hipStreamSynchronize(ctx->stream);
          fprintf(stderr, "[ccm_ba] coarse_build: enqueue %.3f ms, complete %.3f ms\n", tc1 - tc0, now_ms() - tc0);
        } else
        RC(coarse_prepare(ba, lambda));
        pa.Ainv = ba->d_cAinv; pa.Pm = ba->d_cP; pa.na = ba->coarse_na; pa.Nc = ba->coarse_Nc; pa.cparts = ba->d_cparts;
      }
      static const bool trial_dbg = ccm_dbg("trial");   // development: wall clock of every persistent solve (adds two stream syncs per trial)
      double tdbg0 = 0;
      if (trial_dbg) { hipStreamSynchronize(ctx->stream); tdbg0 = now_ms(); }
      {
        // per-device lease: a persistent launch of ANOTHER context of this process (a second Map's global BA, a local BA beside it) is ordered before this
        // one on the GPU; nothing is recorded or waited for while this is the only context on the device
        ccm_coresident_scope lease(ctx);
        ccm_prof_scope ps(ctx, CCM_K_BA_PCG_PERSIST);
        // A cooperative launch guarantees co-residency but, measured with rocprofv3 on MI355X / ROCm 7.2, leaves the
        // GPU idle for ~0.55 ms before the kernel starts (18 launches: 10.5 ms of 53); an ordinary launch starts after
        // ~13 us.  The grid was sized at create time to fit the device at the kernel's occupancy, so on a stream whose
        // earlier work has drained all workgroups become resident; if they ever do not (device shared with another
        // long-running kernel), the bounded spins of pers_exchange abort the solve, pcg_flag[3] reports it and the trial
        // is repeated on the multi-kernel path below.  (The cooperative launch itself was removed in round 5.)
        hipLaunchKernelGGL(ba_pcg_persist, dim3(ba->pers_grid), dim3(kPersTPB), pers_lds_bytes(), ctx->stream, d, pa);
        const hipError_t le = hipGetLastError();
        if (le == hipSuccess) persist_ok = true;
        else { (void)hipGetLastError(); ba->pers_grid = 0; ba->pers_grid_built = 0; ba->w_valid = false; pers_launch_failed = true; }   // the launch itself was refused: multi-kernel path from now on
      }
      if (persist_ok) small_path = pers_trial = true;
      if (trial_dbg) {
        hipStreamSynchronize(ctx->stream);
        int fl[4] = {0, 0, 0, 0};
        hipMemcpy(fl, d.pcg_flag, sizeof(fl), hipMemcpyDeviceToHost);
        fprintf(stderr, "[ccm_ba] trial: lin %d lambda %.4g persist %.1f us, %d CG iterations, coarse %s, W %s\n", ba->lin_id, lambda, (now_ms() - tdbg0) * 1e3, fl[1],
                !use_coarse ? "off" : ba->coarse_fresh ? "built" : "reused", pa.w_load ? "loaded" : "factored");
      }
    }
    if (!persist_ok) {
      d.mk_on = 0;
      const double coarse_lmax_mk = 1.0;
      const bool damped_mk = ba->coarse_force == 0 && coarse_lmax_mk > 0 && ba->lambda_first > 0 && lambda >= coarse_lmax_mk * ba->lambda_first;
      ba->coarse_skipped_damped = damped_mk;
      if (d.mk_cpart && ba->coarse_na && !damped_mk && (ba->coarse_force > 0 || (ba->coarse_force == 0 && ba->coarse_active))) {
        RC(coarse_prepare(ba, lambda));
        d.mk_on = 1;
      }
      ba->coarse_used = d.mk_on != 0;
      CCM_HIP_CHECK(ctx, hipMemsetAsync(d.pcg_flag, 0, 4 * sizeof(int), ctx->stream));
      if (d.S32) {   // the f32 copy of this trial's S for the product kernel (114 -> 57 MB on the 10 000-keyframe map, once per trial)
        const size_t n4 = 9 * (size_t)(d.Cp + d.nOff);
        hipLaunchKernelGGL(ba_s_to_f32, dim3((unsigned)ccm_div_up((int64_t)n4, kTPB)), dim3(kTPB), 0, ctx->stream, (const double*)d.S, d.S32, n4);
      }
      {
        const size_t lds_tiles = (size_t)(2 * kCluN * kCluN + kCluN + kPersWaves) * sizeof(double) + 16;
        CCM_LDS_ATTR(ctx, CCM_LDS_BA_TILES, ba_pcg_init_tiles, lds_tiles);
        if (!ba->d_pers_coff) return ccm_set_error(ctx, CCM_E_STATE, "ccm_ba: cluster entry lists missing");
        hipLaunchKernelGGL(ba_pcg_init_tiles, dim3(d.n_wg_upd), dim3(kPersTPB), lds_tiles, ctx->stream, d, lambda, tol, (const int*)ba->d_pers_coff,
                           (const int*)ba->d_pers_cij, (const uint32_t*)ba->d_pers_cblk);
        if (d.mk_on) hipLaunchKernelGGL(ba_pcg_coarse_apply, dim3(d.mk_na), dim3(kTPB), 6 * (size_t)(d.mk_na + 1) * sizeof(double), ctx->stream, d, 0);
      }
      // The host learns that the solve has converged from the flags it reads back between chunks of queued iterations; every iteration queued beyond the converged
      // one is three launches that return at once (~3 us each + their gaps).  Round 5: the first chunk is sized by the PREVIOUS solve of the handle (consecutive
      // trials of a call need similar counts: 60 - 75 on the 10 000-keyframe map), the following ones are short — ~3 instead of ~12 idle iterations per solve
      // (1680 -> ~1510 launches of each kernel for the 1440 iterations of that call).  The counts are deterministic, so every rank of a sharded run queues alike.
      const int sym_grid = 512;   // two 16-wave workgroups per CU
      const dim3 g_spmv(std::min(d.n_wg_spmv, sym_grid));
      const size_t lds_ca = 6 * (size_t)(d.mk_na + 1) * sizeof(double);
      const bool f32 = d.S32 != nullptr;
      // the true residual r = b - (S + lambda I) x in place of the recurrence's (f64 blocks), followed by z = M^-1 r and its dot products: what iteration kk's update leaves
      auto replace_residual = [&](int kk) {
        { ccm_prof_scope ps(ctx, CCM_K_BA_PCG_SPMV); hipLaunchKernelGGL((ba_pcg_spmv<false, true>), g_spmv, dim3(kSpmvTPB), 0, ctx->stream, d, kk); }
        ccm_prof_scope ps(ctx, CCM_K_BA_PCG_UPDATE);
        hipLaunchKernelGGL(ba_pcg_update<true>, dim3(d.n_wg_upd), dim3(kTPB), 0, ctx->stream, d, kk);
        if (d.mk_on) hipLaunchKernelGGL(ba_pcg_coarse_apply, dim3(d.mk_na), dim3(kTPB), lds_ca, ctx->stream, d, (kk + 1) & 1);
      };
      int k = 0, verified_at = -1;
      // (second half of round 5) the first chunk reaches two iterations BEYOND the previous solve's count — an iteration queued in vain is three launches that return at
      // once (~8 us), a chunk that stops short is a read-back with the device idle (~40 us) plus most of the next chunk in vain —, and the question that follows a
      // verification (below) is ONE iteration, not a chunk: 13.5 -> ~7 iterations queued in vain per solve on the 10 000-keyframe map.
      bool first_chunk = true, after_verify = false;
      while (k < max_it) {
        const int chunk = first_chunk ? (ba->mk_prev_iters > 0 ? std::max(8, ba->mk_prev_iters + 2) : 24) : after_verify ? 1 : 6;
        first_chunk = false; after_verify = false;
        const int kend = std::min(max_it, k + chunk);
        for (; k < kend; k++) {
          {
            ccm_prof_scope ps(ctx, CCM_K_BA_PCG_SPMV);
            if (f32) hipLaunchKernelGGL((ba_pcg_spmv<true, false>), g_spmv, dim3(kSpmvTPB), 0, ctx->stream, d, k);
            else hipLaunchKernelGGL((ba_pcg_spmv<false, false>), g_spmv, dim3(kSpmvTPB), 0, ctx->stream, d, k);
          }
          if (f32 && (k + 1) % kMkReplaceEvery == 0) {
            { ccm_prof_scope ps(ctx, CCM_K_BA_PCG_UPDATE); hipLaunchKernelGGL(ba_pcg_xupdate, dim3(d.n_wg_upd), dim3(kTPB), 0, ctx->stream, d, k); }
            replace_residual(k);
          } else {
            ccm_prof_scope ps(ctx, CCM_K_BA_PCG_UPDATE);
            hipLaunchKernelGGL(ba_pcg_update<false>, dim3(d.n_wg_upd), dim3(kTPB), 0, ctx->stream, d, k);
            if (d.mk_on) hipLaunchKernelGGL(ba_pcg_coarse_apply, dim3(d.mk_na), dim3(kTPB), lds_ca, ctx->stream, d, (k + 1) & 1);
          }
        }
        CCM_HIP_CHECK(ctx, hipMemcpyAsync(flags, d.pcg_flag, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
        CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (flags[0]) {
          // f32 product: the recurrence says "converged at iteration N" (the launches queued behind that test returned at once, so x = x_N).  Unless r_N is a replaced
          // residual already, form the true one and put the question again: the test at the top of iteration N then speaks about b - (S + lambda I) x_N itself.
          const int N = flags[1];
          if (f32 && !flags[2] && N > 0 && N % kMkReplaceEvery != 0 && verified_at != N) {
            verified_at = N;
            CCM_HIP_CHECK(ctx, hipMemsetAsync(d.pcg_flag, 0, 2 * sizeof(int), ctx->stream));
            replace_residual(N - 1);
            k = N;
            after_verify = true;
            flags[0] = 0;
            continue;
          }
          break;
        }
      }
      *pcg_iters = flags[0] ? flags[1] : k;
      ba->mk_prev_iters = *pcg_iters;
    }
    }
    if (flags[2]) *ok = false;   // not SPD / NaN: linear solver failure (levenberg.cpp:126-127)
  }
  if (d.Cp && !cams_updated) {
    ccm_prof_scope ps(ctx, CCM_K_BA_UPDATE);
    hipLaunchKernelGGL(ba_update_cams, dim3(d.n_wg_cam), dim3(kTPB), 0, ctx->stream, d, cur, lambda, ba->rank == 0 ? 1 : 0, ba->d_pers_bar /* nullptr when the handle has no persistent solver; cleared also while it is cooling down after an abort */);
  } else if (!d.Cp) hipMemsetAsync(d.part_cam, 0, sizeof(double) * d.n_wg_cam, ctx->stream);
  {
    ccm_prof_scope ps(ctx, CCM_K_BA_BACKSUB);
    if (d.chunk_off) hipLaunchKernelGGL(ba_backsub_chi2_e, dim3(d.n_chunk), dim3(kTPB), 0, ctx->stream, d, cur, lambda, 0);
    else hipLaunchKernelGGL(ba_backsub_chi2, dim3(d.n_wg_pt), dim3(kTPB), 0, ctx->stream, d, cur, lambda, 0);
  }
  // a sharded run repeats the trial on EVERY rank when the persistent kernel gave up (or could not be launched) on ANY rank:
  // the repeat issues the same collectives again, and all ranks must keep bit-identical camera states
  const bool poll = ba->nranks == 1 && poll_enabled();
  const unsigned long long ticket = ++ba->rb_ticket;
  {
    ccm_prof_scope ps(ctx, CCM_K_BA_REDUCE);
    hipLaunchKernelGGL(ba_reduce_scalars, dim3(1), dim3(kTPB), 0, ctx->stream, d, ba->stop_local(), (pers_launch_failed && ba->nranks > 1) ? 1 : 0,
                       pers_trial ? 1 : 0, poll ? ba->h_rb : (double*)nullptr, ticket);
  }
  RC(ba_allreduce_sum(ba, d.scal, 4));
  double s[6];
  if (poll) { RC(read_scalars_polled(ba, s, small_flags, ticket)); } else { RC(read_scalars(ba, s, small_flags)); }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  ba->stop_any = s[2] > 0.0;
  // s[3] is summed over the ranks: the persistent kernel gave up (or could not be launched) SOMEWHERE.  Every rank repeats the trial on the multi-kernel
  // path, whatever its own solver did — the repeat issues the Schur and scalar all-reduces again, so a rank that skipped it would fall out of step with
  // its peers, and all ranks must keep bit-identical camera states.  The repeat runs with pers_grid = 0 and contributes 0: it cannot recurse.
  if (s[3] > 0.0) {
    // (round 5) the give-up is no longer for good: the handle takes the multi-kernel path for kPersCooldownTrials trials (this repeat included) and then tries
    // the persistent kernel again — whatever held the CUs (another process on the device, a long foreign kernel) has usually gone by then.  s[3] is the
    // reduced value, so every rank of a sharded run counts the same trials.  The carried-over cluster inverse is dropped: the aborted launch may have
    // written part of it.
    if (pers_trial) { ccm_coresident_note_abort(ctx); ba->pers_aborts++; }
    ba->pers_grid = 0; ba->pers_cooldown = kPersCooldownTrials;
    // (round 6) a sharded handle cools down like a single-rank one: s[3] is the all-reduced value, so every rank counts the same trials and returns to the persistent
    // kernel at the same trial.  Only a refused LAUNCH on some rank (>= 4096 in the sum) is for good — that rank cannot come back, and all ranks must keep the same solver path.
    if (ba->nranks > 1 && s[3] >= 4096.0) ba->pers_grid_built = 0;
    ba->w_valid = false; ba->w_pending = false;
    return lm_trial(ba, lambda, opt, temp_chi, scale, ok, pcg_iters);
  }
  if (small_path) { *pcg_iters = small_flags[1]; if (small_flags[2]) *ok = false; }
  if (pers_trial) {   // iteration guard of the carried-over cluster inverse
    if (!ba->w_loaded) ba->w_fresh_iters = *pcg_iters;
    else if (ba->w_lin_id != ba->lin_id && *pcg_iters > ba->w_fresh_iters + ba->w_fresh_iters / 8 + 3) ba->w_stale_bad = true;
    if (ba->w_pending) { ba->w_valid = !small_flags[2]; ba->w_pending = false; }   // the launch that factored has finished cleanly: its W may be loaded from now on
    if (small_flags[2]) ba->w_valid = false;   // a failed solve (bad pivot, NaN) never leaves an inverse behind to be reused
  }
  if (ba->coarse_na && (small_path || d.mk_cpart)) {
    if (ba->coarse_used) {
      if (ba->coarse_fresh) ba->coarse_fresh_iters = *pcg_iters;
      else if (*pcg_iters > ba->coarse_fresh_iters + ba->coarse_fresh_iters / 3 + 8) ba->coarse_stale_bad = true;
    }
    const int on_env = 0;
    const int off_env = 0;
    const int on_it = on_env ? on_env : (d.agg < kAggWide ? kCoarseOnItersFine : kCoarseOnIters);
    const int off_it = off_env ? off_env : (d.agg < kAggWide ? kCoarseOffItersFine : kCoarseOffIters);
    if (ba->coarse_skipped_damped) { /* the level was left out because of the damping, not by the switch: the switch keeps its state */ }
    else if (!ba->coarse_used && *pcg_iters >= on_it) ba->coarse_active = true;
    else if (ba->coarse_used && *pcg_iters <= off_it) ba->coarse_active = false;
  }
  *temp_chi = s[0];
  *scale = s[1];
  return CCM_OK;
}

}  // namespace

// Test hook for the sharded path: this rank's PARTIAL reduced camera system [S blocks | b_schur] of the
// current state at the given lambda, exactly the buffer lm_trial hands to the RCCL all-reduce, without the all-reduce.
// Summing the downloads of all ranks' handles must reproduce the single-rank system (tests/test_ba_gpu.py).
// Test hook: coarse operator Ac = P^T (S + lambda I) P of the current state and its inverse ([6 na]^2 each, row-major), plus
// the prolongation blocks P_k ([Cp][36]).  *na = 0 when the coarse level is not in use for this problem.
int ccm_internal::ba_debug_coarse(ccm_ba* ba, double lambda, int* na, double* Ac, double* Ainv, double* Pm, size_t cap) {
  if (!ba || !na) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  *na = ba->coarse_na;   // camera intervals; the coarse system has na + 1 nodes of 6 unknowns
  if (!ba->coarse_na || !Ac) return CCM_OK;
  const size_t nc = 6 * ((size_t)ba->coarse_na + 1), Nc = (size_t)ba->coarse_Nc;
  if (cap < nc * nc) return ccm_set_error(ctx, CCM_E_ARG, "ccm_ba_debug_coarse: buffer too small");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  RC(build_system(ba));
  if (d.Lloc) hipLaunchKernelGGL(ba_dinv, dim3(d.n_wg_pt), dim3(kTPB), 0, ctx->stream, d, lambda);
  RC(launch_schur(ba));
  // the inverse destroys d_cA: assemble twice
  RC(coarse_build(ba, lambda));
  std::vector<double> buf(Nc * Nc);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(buf.data(), ba->d_cAinv, buf.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (Ainv) for (size_t r = 0; r < nc; r++) for (size_t c = 0; c < nc; c++) Ainv[r * nc + c] = buf[r * Nc + c];
  hipLaunchKernelGGL(ba_coarse_sum, dim3(ccm_div_up((int64_t)Nc * Nc, kTPB)), dim3(kTPB), 0, ctx->stream, (const double*)ba->d_cstage, (const unsigned*)ba->d_cb_key,
                     ba->coarse_ncb, ba->coarse_na, ba->coarse_na + 1, ba->d_cA, (int)Nc);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(buf.data(), ba->d_cA, buf.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t r = 0; r < nc; r++) for (size_t c = 0; c < nc; c++) Ac[r * nc + c] = buf[r * Nc + c];
  if (Pm) { CCM_HIP_CHECK(ctx, hipMemcpy(Pm, ba->d_cP, 36 * (size_t)d.Cp * sizeof(double), hipMemcpyDeviceToHost)); }
  return CCM_OK;
}

int ccm_internal::ba_debug_partial_reduced(ccm_ba* ba, double lambda, double* out, size_t cap, size_t* count) {
  if (!ba || !count) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  *count = ba->red_count;
  if (!out) return CCM_OK;
  if (cap < ba->red_count) return ccm_set_error(ctx, CCM_E_ARG, "ccm_ba_debug_partial_reduced: buffer too small");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  RC(build_system(ba));
  if (d.Lloc) hipLaunchKernelGGL(ba_dinv, dim3(d.n_wg_pt), dim3(kTPB), 0, ctx->stream, d, lambda);
  RC(launch_schur(ba));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(out, ba->d_red, ba->red_count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

static void pers_dbg_dump(ccm_ba* ba) {
  if (ba->d.row_dbg) {
    long long h[8];
    hipMemcpy(h, ba->d.row_dbg, sizeof(h), hipMemcpyDeviceToHost);
    const double nw = (double)std::max<long long>(h[5], 1);
    fprintf(stderr, "[ccm_ba] row Schur kernel, thread 0 of every workgroup (%lld workgroups): us per row: staging + diagonal %.2f block passes %.2f wait %.2f final sums %.2f\n",
            h[5], h[0] * 0.01 / nw, h[1] * 0.01 / nw, h[2] * 0.01 / nw, h[3] * 0.01 / nw);
  }
  if (ba->d_cholreg_dbg && ccm_dbg("cholreg")) {
    long long h[8];
    hipMemcpy(h, ba->d_cholreg_dbg, sizeof(h), hipMemcpyDeviceToHost);
    const double nl = (double)std::max<long long>(h[7], 1);
    fprintf(stderr, "[ccm_ba] register-resident Cholesky solve (%d free cameras): %lld launches; us/launch: assemble %.1f factor %.1f (barrier wait %.1f, diagonal + panel %.1f, trailing update + park %.1f) backward %.1f camera update %.1f\n", ba->d.Cp, h[7],
            h[0] * 0.01 / nl, (h[1] + h[4] + h[5] + h[6]) * 0.01 / nl, h[4] * 0.01 / nl, h[5] * 0.01 / nl, h[6] * 0.01 / nl, h[2] * 0.01 / nl, h[3] * 0.01 / nl);
  }
  if (ba->d_dense_T && ccm_dbg("dense2")) {
    long long h[11];
    hipMemcpy(h, ba->d_dense_T + kCluN * kCluN, sizeof(h), hipMemcpyDeviceToHost);
    const double nl = (double)std::max<long long>(h[8], 1);
    fprintf(stderr, "[ccm_ba] exact two-cluster solve: %lld launches; us/launch: assemble1 %.1f factor1 %.1f t1+A12 %.1f T %.1f P+c %.1f assemble2 %.1f factor2 %.1f x %.1f | both factorisations: cholesky %.1f inverse-factor %.1f\n", h[8],
            h[0] * 0.01 / nl, h[1] * 0.01 / nl, h[2] * 0.01 / nl, h[3] * 0.01 / nl, h[4] * 0.01 / nl, h[5] * 0.01 / nl, h[6] * 0.01 / nl, h[7] * 0.01 / nl, h[9] * 0.01 / nl, h[10] * 0.01 / nl);
  }
  if (!ba->pers_grid || !ccm_dbg("pers")) return;
  long long h[16];
  hipMemcpy(h, ba->d_pers_bar + 4, sizeof(h), hipMemcpyDeviceToHost);
  {
    std::vector<long long> w(2 * (size_t)ba->pers_grid);
    hipMemcpy(w.data(), (long long*)(ba->d_pers_bar + 4) + 32, w.size() * sizeof(long long), hipMemcpyDeviceToHost);
    const double itn = (double)std::max<long long>(h[12], 1);
    for (int x = 0; x < 2; x++) {
      std::vector<double> v;
      for (int g = 0; g < ba->pers_grid; g++) if (w[2 * g + x] > 0) v.push_back(w[2 * g + x] * 0.01 / itn);
      std::sort(v.begin(), v.end());
      if (!v.empty()) fprintf(stderr, "[ccm_ba] persistent PCG, time inside exchange %d per unit and iteration (us): min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f  (%zu units)\n",
                              x + 1, v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
    }
  }
  const double it = (double)std::max<long long>(h[12], 1), nl = (double)std::max<long long>(h[13], 1);
  fprintf(stderr, "[ccm_ba] persistent PCG, workgroup 0: %lld iterations in %lld launches; us/iteration: stage_p %.2f spmv+dot %.2f barrier1 %.2f "
          "sum_pq %.2f update+W %.2f barrier2 %.2f sum_rz %.2f | us/launch: assemble %.1f cholesky %.1f inverse %.1f W %.1f start %.1f\n",
          h[12], h[13], h[0] * 0.01 / it, h[1] * 0.01 / it, h[2] * 0.01 / it, h[3] * 0.01 / it, h[4] * 0.01 / it, h[5] * 0.01 / it, h[6] * 0.01 / it,
          h[7] * 0.01 / nl, h[8] * 0.01 / nl, h[9] * 0.01 / nl, h[10] * 0.01 / nl, h[11] * 0.01 / nl);
}

extern "C" int ccm_ba_run(ccm_ba* ba, const ccm_ba_options* opt_in, const volatile unsigned char* stop_flag, ccm_ba_stats* stats) {
  if (!ba) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  if (ba->nranks > 1 && ((!ctx->comm && !ctx->loop_group) || ctx->comm_nranks != ba->nranks || ctx->comm_rank != ba->rank))
    return ccm_set_error(ctx, CCM_E_STATE, "ccm_ba_run: sharded problem needs a matching communicator (ccm_comm_init)");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ccm_ba_options opt{};
  if (opt_in) opt = *opt_in;
  ba->coarse_active = false;   // every run starts from the same preconditioner state
  ba->mk_prev_iters = 0;
  ba->coarse_valid = false; ba->coarse_stale_bad = false; ba->coarse_fresh_iters = 0;
  ba->w_valid = false; ba->w_stale_bad = false; ba->w_fresh_iters = 0; ba->w_lambda_built = 0; ba->lin_id = 0; ba->w_lin_id = -1;
  ba->stop_flag = stop_flag; ba->stop_any = false;
  ba->hist_chi2.clear(); ba->hist_lambda.clear(); ba->hist_trials.clear();
  if (ba->nranks > 1 && !ba->pers_agreed) {
    // the ranks of a sharded run must take the same PCG path from the first trial (bit-identical replicated solves): the persistent kernel is used only
    // when EVERY rank can run it (occupancy, CU count and CCM_BA_NO_PERSIST may differ between ranks)
    const double mine = ba->pers_grid ? 0.0 : 1.0;
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d.scal + 5, &mine, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    RC(ba_allreduce_max(ba, ba->d.scal + 5, 1));
    double any = 0;
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(&any, ba->d.scal + 5, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (any > 0.0) { ba->pers_grid = 0; ba->pers_grid_built = 0; }
    ba->pers_agreed = true;
  }
  const double t_start = now_ms();
  ccm_ba_stats st{};
  st.ms_setup = ba->ms_setup;
  st.n_schur_blocks = ba->Cp + ba->nOff;
  st.n_pair_instances = ba->n_inst;
  double lambda = 0, ni = 2;
  int nBad = 0;
  double currentChi = 0;
  int reason = 0;
  bool have_chi = false;
  int rc = CCM_OK;
  const bool empty = (ba->n_act_edges == 0);
  for (int it = 0; it < opt.max_iters && !empty; it++) {
    // optimize()'s `!terminate()` (sparse_optimizer.cpp:382).  A sharded run acts on the flag as reduced with the last trial's
    // scalars; before the first trial that is the reduction inside eval_chi2, which changes no state.
    if (ba->nranks == 1 && ba->stop_requested()) { reason = 1; break; }
    // computeActiveErrors + activeRobustChi2 — equal to the chi2 of the last accepted state, which the
    // previous iteration already evaluated (same state => same value); only iteration 0 needs a pass.
    if (!have_chi) { if ((rc = eval_chi2(ba, &currentChi))) return rc; have_chi = true; st.chi2_initial = currentChi; }
    if (ba->nranks > 1 && ba->stop_requested()) { reason = 1; break; }
    const double iniChi = currentChi;
    if ((rc = build_system(ba, it > 0 ? lambda : -1.0))) return rc;
    ba->lin_id++;
    if (it == 0) {
      if (opt.lambda_init > 0) lambda = opt.lambda_init;
      else { double md = 0; if ((rc = max_diag(ba, &md))) return rc; lambda = 1e-5 * md; }
      ba->lambda_first = lambda;
      ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      double tempChi = 0, scale = 0; bool ok2 = true; int pit = 0;
      if ((rc = lm_trial(ba, lambda, opt, &tempChi, &scale, &ok2, &pit))) return rc;
      st.pcg_iters += pit;
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      scale += 1e-3;
      rho /= scale;
      if (opt.verbose && ba->rank == 0)
        fprintf(stderr, "[ccm_ba] it %d trial %d lambda %.6g chi %.9g -> %.9g rho %.4g pcg %d%s\n", it, qmax, lambda, currentChi, tempChi, rho, pit, ok2 ? "" : " (solver failed)");
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        ba->cur ^= 1;                        // discardTop(): the trial state becomes the estimate
      } else {
        lambda *= ni; ni *= 2;               // pop(): the estimate stays
      }
      qmax++; st.lm_trials++;
      if (ba->trial_cb) ba->trial_cb(ba->trial_cb_user, it, qmax, tempChi, (rho > 0 && std::isfinite(tempChi)) ? 1 : 0);
    } while (rho < 0 && qmax < 10 && !ba->stop_requested());
    st.iters_done++;
    ba->hist_chi2.push_back(currentChi); ba->hist_lambda.push_back(lambda); ba->hist_trials.push_back(qmax);
    if (qmax == 10 || rho == 0) { reason = 2; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { reason = 3; break; }
  }
  st.stop_reason = reason;
  st.chi2_final = currentChi;
  st.lambda_final = lambda;
  st.ms_iters = now_ms() - t_start;
  st.ms_total = st.ms_iters + st.ms_setup;
  if (stats) *stats = st;
  pers_dbg_dump(ba);
  return CCM_OK;
}

// the camera-major copy of the informations follows ba_deactivate_edges
__global__ void ba_refresh_cam_info(BaDev d) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.cam_off[d.Cp]) return;
  d.cam_oi[4 * (size_t)s + 2] = d.info[d.cam_edge[s]];
}
// edges whose level is not 0 leave the optimisation: their information becomes 0, so every sum they enter gets an exact zero; an edge whose level is 0
// (again) gets the information it was created with
__global__ void ba_deactivate_edges(double* info, const double* info_orig, const int* loc_edge_orig, const uint8_t* level, int Eloc) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < Eloc) info[k] = level[loc_edge_orig[k]] != 0 ? 0.0 : info_orig[k];
}

// The second stage of Optimizer::LocalBundleAdjustmentClient (Optimizer.cpp:545-566: outlier edges -> setLevel(1), robust kernel off,
// initializeOptimization(0), optimize(10)) on the SAME handle: edges with e_level != 0 are taken out by zeroing their information (exact zeros in every
// sum: the same normal equations as a rebuilt structure, whose rows / blocks they would merely not have), the Huber delta is replaced, the estimate stays.
// Edges that were inactive at ccm_ba_create are not part of the handle and cannot come back; every other edge can: the call is a pure function of
// (e_level, huber_delta) and the information stored at create time (d_info_orig), so a handle that is re-run (push / pop state) passes its first-stage
// levels and delta again and gets the first-stage problem back.  ccm_ba_download keeps returning, for a deactivated edge, the chi2 of the last pass it
// took part in (g2o leaves the _error of a level-1 edge alone).
extern "C" int ccm_ba_set_edge_levels(ccm_ba* ba, const uint8_t* e_level, double huber_delta) {
  if (!ba || !e_level) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (ba->n_edge) {
    void* d_lvl = nullptr;
    RC(ccm_scratch(ctx, (size_t)ba->n_edge, &d_lvl));
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_lvl, e_level, (size_t)ba->n_edge, hipMemcpyDefault, ctx->stream));
    if (ba->Eloc && !ba->d_info_orig) {   // first call: keep what ccm_ba_create stored
      RC(dev_alloc<double>(ba, (size_t)ba->Eloc, &ba->d_info_orig, false));
      CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->d_info_orig, ba->d.info, (size_t)ba->Eloc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (ba->Eloc) hipLaunchKernelGGL(ba_deactivate_edges, dim3(ccm_div_up(ba->Eloc, kTPB)), dim3(kTPB), 0, ctx->stream, const_cast<double*>(ba->d.info), (const double*)ba->d_info_orig,
                                     (const int*)ba->d_loc_edge_orig, (const uint8_t*)d_lvl, ba->Eloc);
    if (ba->Eloc && ba->d.Cp && ba->d.cam_oi) hipLaunchKernelGGL(ba_refresh_cam_info, dim3(ccm_div_up(ba->Eloc, kTPB)), dim3(kTPB), 0, ctx->stream, ba->d);
    CCM_HIP_CHECK(ctx, hipGetLastError());
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // e_level may be a pageable host array
  }
  ba->d.huber = huber_delta;
  return CCM_OK;
}

extern "C" int ccm_ba_history(const ccm_ba* ba, int cap, double* chi2_per_iter, double* lambda_per_iter, int32_t* trials_per_iter, int* n_iters) {
  if (!ba || !n_iters) return CCM_E_ARG;
  *n_iters = (int)ba->hist_chi2.size();
  for (int i = 0; i < *n_iters && i < cap; i++) {
    if (chi2_per_iter) chi2_per_iter[i] = ba->hist_chi2[i];
    if (lambda_per_iter) lambda_per_iter[i] = ba->hist_lambda[i];
    if (trials_per_iter) trials_per_iter[i] = ba->hist_trials[i];
  }
  return CCM_OK;
}

extern "C" int ccm_ba_set_trial_callback(ccm_ba* ba, ccm_ba_trial_cb cb, void* user) {
  if (!ba) return CCM_E_ARG;
  ba->trial_cb = cb; ba->trial_cb_user = user;
  return CCM_OK;
}

extern "C" int ccm_ba_download(ccm_ba* ba, double* cam_qt, double* pt_xyz, double* chi2_per_edge) {
  if (!ba) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const bool want_pt = pt_xyz && ba->Lp, want_chi2 = chi2_per_edge && ba->Eloc, want_map = want_chi2 && ba->loc_edge_orig.empty();
  // everything is copied into the context's pinned block behind the device work (one synchronisation), then placed into the caller's arrays
  const size_t b_cam = cam_qt ? 7 * (size_t)ba->n_cam * sizeof(double) : 0, b_pt = want_pt ? 3 * (size_t)ba->n_pt * sizeof(double) : 0,
               b_c2 = want_chi2 ? (size_t)ba->Eloc * sizeof(double) : 0;
  void* pin = nullptr;
  RC(ccm_pin_scratch(ctx, b_cam + b_pt + b_c2 + 1024, &pin));
  double* h_cam = (double*)pin;
  double* h_pt = (double*)((char*)pin + ((b_cam + 255) & ~(size_t)255));
  double* h_c2 = (double*)((char*)h_pt + ((b_pt + 255) & ~(size_t)255));
  if (cam_qt) CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_cam, d.cam[ba->cur], b_cam, hipMemcpyDeviceToHost, ctx->stream));
  if (want_pt) {
    // optimised landmarks into the caller's numbering on the device (landmarks without an active edge keep the values uploaded at create / reset)
    const double* src = d.pt[ba->cur];
    if (ba->nranks > 1) {
      CCM_HIP_CHECK(ctx, hipMemsetAsync(ba->d_pt_full, 0, 3 * (size_t)ba->Lp * sizeof(double), ctx->stream));
      if (ba->Lloc) hipLaunchKernelGGL(ba_scatter_points, dim3(ccm_div_up(ba->Lloc, kTPB)), dim3(kTPB), 0, ctx->stream, ba->d_pt_full, d.pt[ba->cur], ba->lp_begin, ba->Lloc);
      RC(ba_allreduce_sum(ba, ba->d_pt_full, 3 * (size_t)ba->Lp));
      src = ba->d_pt_full;
    }
    RC(ccm_ba_points_to_raw_order(ba, src));
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_pt, ba->d_raw_pt, b_pt, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (want_chi2) {
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(h_c2, d.edge_chi2, b_c2, hipMemcpyDeviceToHost, ctx->stream));
    if (want_map) {   // local edge -> the caller's edge index (built on the device, fetched on first use)
      ba->loc_edge_orig.resize(ba->Eloc);
      CCM_HIP_CHECK(ctx, hipMemcpyAsync(ba->loc_edge_orig.data(), ba->d_loc_edge_orig, (size_t)ba->Eloc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (cam_qt)   // only the poses that were vertices of the problem
    for (int i = 0; i < ba->Cp; i++) std::memcpy(cam_qt + 7 * (size_t)ba->slot_cam[i], h_cam + 7 * (size_t)ba->slot_cam[i], 7 * sizeof(double));
  if (want_pt) std::memcpy(pt_xyz, h_pt, b_pt);
  if (want_chi2)
    // e->chi2(): value of the last evaluated LM trial (g2o keeps _error of the last computeActiveErrors, even when that trial was rejected).  Only the
    // own (shard-local) active edges are written; inactive (level != 0) edges keep whatever the caller passed in, as g2o leaves their _error untouched.
    for (int k = 0; k < ba->Eloc; k++) chi2_per_edge[ba->loc_edge_orig[k]] = h_c2[k];
  return CCM_OK;
}

// e->isDepthPositive() (types_six_dof_expmap.h:97-101) for EVERY edge (active or not) at the given state.
// Host arithmetic on the downloaded state: O(n_edge), off the hot path (called once after optimize()).
static void depth_positive_all(const ccm_ba_problem* P, const double* cam_qt, const double* pt_xyz, uint8_t* depth_pos) {
  for (int e = 0; e < P->n_edge; e++) {
    BaPose T = ba_load_pose(cam_qt + 7 * (size_t)P->e_cam[e]);
    double Xc[3];
    ba_map(T, pt_xyz + 3 * (size_t)P->e_pt[e], Xc);
    depth_pos[e] = Xc[2] > 0.0;
  }
}

extern "C" int ccm_ba_depth_positive(const ccm_ba_problem* P, const double* cam_qt, const double* pt_xyz, uint8_t* depth_pos) {
  if (!P || !cam_qt || !pt_xyz || !depth_pos) return CCM_E_ARG;
  depth_positive_all(P, cam_qt, pt_xyz, depth_pos);
  return CCM_OK;
}

extern "C" int ccm_ba_optimize(ccm_ctx* ctx, ccm_ba_problem* prob, const ccm_ba_options* opt,
                               const volatile unsigned char* stop_flag, double* chi2_per_edge, uint8_t* depth_pos,
                               ccm_ba_stats* stats) {
  ccm_ba* ba = nullptr;
  // one rank, always: a one-shot call made by a single agent on a context that carries a multi-rank communicator must not
  // turn into a collective (sharding is opt-in through ccm_ba_create(rank, nranks), called by ALL ranks)
  int rc = ccm_ba_create(ctx, prob, 0, 1, &ba);
  if (rc) return rc;
  rc = ccm_ba_run(ba, opt, stop_flag, stats);
  if (rc == CCM_OK) rc = ccm_ba_download(ba, prob->cam_qt, prob->pt_xyz, chi2_per_edge);
  if (rc == CCM_OK && depth_pos) depth_positive_all(prob, prob->cam_qt, prob->pt_xyz, depth_pos);
  ccm_ba_destroy(ba);
  return rc;
}
