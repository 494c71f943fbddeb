// posegraph.hip — Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion numerics (cslam/src/Optimizer.cpp:1058-1331,
// 1333-1566): 7-DoF pose graph over keyframes.  Reference structure: one VertexSim3Expmap per keyframe (estimate Siw,
// one vertex fixed, _fix_scale), one EdgeSim3 per spanning-tree / loop / covisibility link (vertex(0) = i, vertex(1) = j,
// measurement Sji, information I7; error = log(Sji * Siw * Sjw^-1), types_seven_dof_expmap.h:98-121), BlockSolver_7_3 +
// LinearSolverEigen, Levenberg with setUserLambdaInit(1e-16), optimize(20).  EdgeSim3 does not override linearizeOplus:
// both 7x7 Jacobians are g2o's central differences through the vertex oplus (base_binary_edge.hpp:129-196, delta 1e-9),
// which is where g2o spends its time on this problem (28 exp/log evaluations per edge per iteration).
//
// Device design: the numeric differentiation is embarrassingly parallel — one lane per (edge, vertex side, dimension,
// sign) — so `pg_linearize` runs 28 lanes per edge.  H is assembled by gather (one wave per 7x7 block walks the edges that
// contribute to it: no atomics, fixed order).  The damped system is solved EXACTLY by default (dense MFMA-f64 Cholesky,
// dense_chol.hip): only an exact step reproduces which LM trials g2o accepts.  CCM_PG_SOLVER=pcg (and graphs above 24 000
// unknowns) use PCG with a spanning-forest preconditioner instead — ~110 iterations whatever the chain length, several times
// faster on large graphs, but an inexact-Newton path that ends in a different (equally converged) point.  The LM control
// loop stays on the host like optimization_algorithm_levenberg.cpp:61-164.  The graph walk that picks the edges and the SE3 / map-point write-back
// (Optimizer.cpp:1268-1330) stay with the caller.
#include "common.h"
#include "sim3_math.h"
#include "lane_xor.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

namespace {

constexpr int kWave = 64;
constexpr int kTPB = 256;
constexpr uint32_t kT = 0x80000000u;   // transpose bit of a row entry

struct PgDev {
  int V, F, E, nBlk;          // vertices, free vertices, active edges, blocks (F diagonal first)
  int fix_scale;
  double* S[2];               // [V*8] current / trial
  const int* slot;            // [V] free slot or -1
  const int* free_v;          // [F] vertex of slot
  const int *ei, *ej;         // [E]
  const double* meas;         // [E*8]
  double* err;                // [E*7] errors of the last evaluated state (computeActiveErrors)
  double* J;                  // [E][2][49]  J[k*7+d]
  double* H;                  // [nBlk*49]
  double* b;                  // [F*7]
  // structure
  const int* blk_off; const int* blk_edge;        // per block: contributing edges; entry = edge*4 + (side_row*2 + side_col)
  const int* vtx_off; const int* vtx_edge;        // per free slot: incident (edge*2 + side)
  const int* row_off; const int* row_col; const uint32_t* row_blk;
  // PCG
  double *x, *r, *z, *q, *p[2], *Minv;
  double *ppq, *prz[2];
  double* scal;               // [0]=rz0 [1]=thresh^2 [2]=lambda ; [4]=chi2 [5]=scale
  int* flag;                  // [0]=done [1]=iters [2]=fail
  double* part;               // partial sums of chi2 / scale
  // spanning-forest preconditioner (see pg_tree_setup); use_tree = 0 -> block-Jacobi
  int use_tree;
  const int* t_parent; const int* t_code; const int* t_order; const int* t_pos; const int* t_out;
  const int* ends_off; const int* ends_idx;
  double* Binv; double* Dinv;  // [F*49] each
  int n_wg_row, n_wg_upd, n_wg_edge;
};

__device__ __forceinline__ double wave_sum(double v) { return lanex::wave_sum(v); }   // (xor butterfly 32 ... 1, lane_xor.h)
__device__ __forceinline__ double sum_partials(const double* __restrict__ p, int n) {
  const int lane = threadIdx.x & (kWave - 1);
  double s = 0;
  for (int i = lane; i < n; i += kWave) s += p[i];
  return wave_sum(s);
}
__device__ __forceinline__ double block_sum(double v, double* lds) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int w = 0; w < kTPB / kWave; w++) s += lds[w];
  return s;
}

__device__ __forceinline__ Sim3d pg_oplus(const Sim3d& X, const double* upd, int fix_scale) {
  double u[7];
#pragma unroll
  for (int k = 0; k < 7; k++) u[k] = upd[k];
  if (fix_scale) u[6] = 0;                       // VertexSim3Expmap::oplusImpl (types_seven_dof_expmap.h:58-67)
  return sim3_mul(sim3_exp(u), X);
}
__device__ __forceinline__ void pg_edge_error(const Sim3d& C, const Sim3d& Si, const Sim3d& Sj, double err[7]) {
  const Sim3d Eo = sim3_mul(sim3_mul(C, Si), sim3_inv(Sj));   // EdgeSim3::computeError (:104-112)
  sim3_log(Eo, err);
}

// errors + chi2 of state `which` (computeActiveErrors + activeChi2): thread per edge       [also used for the trial]
__global__ __launch_bounds__(kTPB) void pg_chi2(PgDev d, int which) {
  __shared__ double lds[kTPB / kWave];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double c2 = 0;
  if (e < d.E) {
    const Sim3d C = sim3_load(d.meas + 8 * (size_t)e);
    const Sim3d Si = sim3_load(d.S[which] + 8 * (size_t)d.ei[e]), Sj = sim3_load(d.S[which] + 8 * (size_t)d.ej[e]);
    double er[7];
    pg_edge_error(C, Si, Sj, er);
#pragma unroll
    for (int k = 0; k < 7; k++) { d.err[7 * (size_t)e + k] = er[k]; c2 += er[k] * er[k]; }
  }
  const double s = block_sum(c2, lds);
  if (threadIdx.x == 0) d.part[blockIdx.x] = s;
}
__global__ __launch_bounds__(kTPB) void pg_reduce(PgDev d, int n, int dst) {
  __shared__ double lds[kTPB / kWave];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += kTPB) s += d.part[i];
  const double tot = block_sum(s, lds);
  if (threadIdx.x == 0) d.scal[dst] = tot;
}

// numeric Jacobians: 32 lanes per edge; lane l < 28 -> side = l / 14, dim = (l % 14) / 2, sign = l & 1
__global__ __launch_bounds__(kTPB) void pg_linearize(PgDev d, int cur) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const int l = threadIdx.x & 31;
  if (e >= d.E) return;
  const int vi = d.ei[e], vj = d.ej[e];
  const Sim3d C = sim3_load(d.meas + 8 * (size_t)e);
  Sim3d Si = sim3_load(d.S[cur] + 8 * (size_t)vi), Sj = sim3_load(d.S[cur] + 8 * (size_t)vj);
  const int side = l / 14, dim = (l % 14) >> 1;
  const bool active = l < 28 && (side == 0 ? d.slot[vi] >= 0 : d.slot[vj] >= 0);
  double er[7] = {0, 0, 0, 0, 0, 0, 0};
  if (active) {
    double add[7] = {0, 0, 0, 0, 0, 0, 0};
    const double dl = (l & 1) ? -1e-9 : 1e-9;
#pragma unroll
    for (int k = 0; k < 7; k++) if (k == dim) add[k] = dl;
    if (side == 0) Si = pg_oplus(Si, add, d.fix_scale); else Sj = pg_oplus(Sj, add, d.fix_scale);
    pg_edge_error(C, Si, Sj, er);
  }
  // column = scalar * (e+ - e-): the minus lane sits next to the plus lane
  const double scalar = 1.0 / (2 * 1e-9);
  double* Jout = d.J + 98 * (size_t)e + 49 * side;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const double other = __shfl_xor(er[k], 1, kWave);
    if (l < 28 && !(l & 1)) Jout[k * 7 + dim] = active ? scalar * (er[k] - other) : 0.0;
  }
}

// H blocks by gather: one wave per block, lane = element (p, q); b by gather: one wave per free vertex
__global__ __launch_bounds__(kTPB) void pg_assemble(PgDev d) {
  const int w = blockIdx.x * (kTPB / kWave) + threadIdx.x / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  if (w < d.nBlk) {
    if (lane < 49) {
      const int p = lane / 7, q = lane % 7;
      double acc = 0;
      for (int s = d.blk_off[w]; s < d.blk_off[w + 1]; s++) {
        const int code = d.blk_edge[s];
        const int e = code >> 2, sr = (code >> 1) & 1, sc = code & 1;
        const double* A = d.J + 98 * (size_t)e + 49 * sr;
        const double* B = d.J + 98 * (size_t)e + 49 * sc;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += A[k * 7 + p] * B[k * 7 + q];
      }
      d.H[49 * (size_t)w + lane] = acc;
    }
  } else if (w < d.nBlk + d.F) {
    const int a = w - d.nBlk;
    if (lane < 7) {
      double acc = 0;
      for (int s = d.vtx_off[a]; s < d.vtx_off[a + 1]; s++) {
        const int code = d.vtx_edge[s];
        const int e = code >> 1, side = code & 1;
        const double* Jm = d.J + 98 * (size_t)e + 49 * side;
        const double* er = d.err + 7 * (size_t)e;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += Jm[k * 7 + lane] * (-er[k]);
      }
      d.b[7 * (size_t)a + lane] = acc;
    }
  }
}

// ---- block-Jacobi PCG on (H + lambda I) x = b, 7x7 blocks ----
__global__ __launch_bounds__(kTPB) void pg_pcg_init(PgDev d, double lambda, double rel_tol) {
  __shared__ double lds[kTPB / kWave];
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  double rz = 0;
  bool bad = false;
  if (a < d.F) {
    // Minv = (H_aa + lambda I)^-1 by Cholesky + triangular inverse, in registers
    double L[49], X[28];
#pragma unroll
    for (int i = 0; i < 49; i++) L[i] = d.H[49 * (size_t)a + i];
#pragma unroll
    for (int i = 0; i < 7; i++) L[i * 8] += lambda;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      double dj = L[j * 7 + j];
#pragma unroll
      for (int k = 0; k < j; k++) dj -= L[j * 7 + k] * L[j * 7 + k];
      if (!(dj > 0.0)) { bad = true; dj = 1.0; }
      dj = sqrt(dj);
      L[j * 7 + j] = dj;
      const double inv = 1.0 / dj;
#pragma unroll
      for (int i = j + 1; i < 7; i++) {
        double s = L[i * 7 + j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= L[i * 7 + k] * L[j * 7 + k];
        L[i * 7 + j] = s * inv;
      }
    }
#pragma unroll
    for (int col = 0; col < 7; col++)
#pragma unroll
      for (int i = col; i < 7; i++) {
        double s = (i == col) ? 1.0 : 0.0;
#pragma unroll
        for (int k = col; k < i; k++) s -= L[i * 7 + k] * X[k * (k + 1) / 2 + col];
        X[i * (i + 1) / 2 + col] = s / L[i * 7 + i];
      }
    double r[7], z[7];
#pragma unroll
    for (int i = 0; i < 7; i++) r[i] = d.b[7 * (size_t)a + i];
#pragma unroll
    for (int p = 0; p < 7; p++)
#pragma unroll
      for (int q = 0; q < 7; q++) {
        double s = 0;
#pragma unroll
        for (int k = (p > q ? p : q); k < 7; k++) s += X[k * (k + 1) / 2 + p] * X[k * (k + 1) / 2 + q];
        d.Minv[49 * (size_t)a + p * 7 + q] = s;
        L[p * 7 + q] = s;
      }
#pragma unroll
    for (int p = 0; p < 7; p++) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 7; q++) s += L[p * 7 + q] * r[q];
      z[p] = s;
      rz += r[p] * s;
      const size_t g = 7 * (size_t)a + p;
      d.x[g] = 0; d.r[g] = r[p]; d.z[g] = s; d.p[0][g] = 0;
    }
  }
  const double tot = block_sum(rz, lds);
  if (threadIdx.x == 0) d.prz[0][blockIdx.x] = tot;
  if (bad) d.flag[2] = 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.scal[1] = rel_tol * rel_tol; d.scal[2] = lambda; }
}

// iteration k: p = z + beta p_old on the fly, q = (H + lambda I) p, partial p.q — one wave per block row
__global__ __launch_bounds__(kTPB) void pg_pcg_spmv(PgDev d, int k) {
  __shared__ double lds[kTPB / kWave];
  if (d.flag[0]) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = threadIdx.x / kWave;
  const int i = blockIdx.x * (kTPB / kWave) + wv;
  const double rz_k = sum_partials(d.prz[k & 1], d.n_wg_upd);
  double beta = 0;
  if (k == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) d.scal[0] = rz_k; }
  else beta = rz_k / sum_partials(d.prz[(k + 1) & 1], d.n_wg_upd);
  const double rz0 = (k == 0) ? rz_k : d.scal[0];
  if (rz_k <= d.scal[1] * rz0 || !(rz_k > 0.0)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.flag[0] = 1; d.flag[1] = k; if (rz_k != rz_k) d.flag[2] = 1; }
    return;
  }
  const double lambda = d.scal[2];
  const double* pold = d.p[k & 1];
  double* pnew = d.p[(k + 1) & 1];
  const int g = lane >> 3, r = lane & 7;
  double acc = 0;
  if (i < d.F && r < 7) {
    for (int s = d.row_off[i] + g; s < d.row_off[i + 1]; s += 8) {
      const int j = d.row_col[s];
      const uint32_t bt = d.row_blk[s];
      const double* B = d.H + 49 * (size_t)(bt & ~kT);
#pragma unroll
      for (int c = 0; c < 7; c++) {
        const double v = (bt & kT) ? B[c * 7 + r] : B[r * 7 + c];
        acc += v * (d.z[7 * (size_t)j + c] + beta * pold[7 * (size_t)j + c]);
      }
    }
  }
  acc = lanex::add_partner<8>(acc);
  acc = lanex::add_partner<16>(acc);
  acc = lanex::add_partner<32>(acc);
  double pq = 0;
  if (i < d.F && lane < 7) {
    const double pi = d.z[7 * (size_t)i + lane] + beta * pold[7 * (size_t)i + lane];
    const double qv = acc + lambda * pi;
    d.q[7 * (size_t)i + lane] = qv;
    pnew[7 * (size_t)i + lane] = pi;
    pq = pi * qv;
  }
  pq = wave_sum(pq);
  if (lane == 0) lds[wv] = pq;
  __syncthreads();
  if (threadIdx.x == 0) d.ppq[blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
}

// alpha = rz / p.q ; x += alpha p ; r -= alpha q ; z = Minv r ; partial r.z — 8 lanes per vertex (7 active)
__global__ __launch_bounds__(kTPB) void pg_pcg_update(PgDev d, int k) {
  __shared__ double lds[kTPB / kWave];
  __shared__ double rs[kTPB];
  const int t = threadIdx.x;
  const int a = blockIdx.x * (kTPB / 8) + (t >> 3), c = t & 7;
  const bool on = a < d.F && c < 7;
  const int done = d.flag[0];
  const double rz_k = sum_partials(d.prz[k & 1], d.n_wg_upd);
  const double pq = sum_partials(d.ppq, d.n_wg_row);
  if (done) return;
  if (!(pq > 0.0)) {
    if (blockIdx.x == 0 && t == 0) { d.flag[0] = 1; d.flag[1] = k; d.flag[2] = 1; }
    return;
  }
  const double alpha = rz_k / pq;
  const size_t g = 7 * (size_t)a + c;
  double rv = 0;
  if (on) {
    const double* p = d.p[(k + 1) & 1];
    d.x[g] += alpha * p[g];
    rv = d.r[g] - alpha * d.q[g];
    d.r[g] = rv;
  }
  rs[t] = rv;
  __syncthreads();
  double rz = 0;
  if (on) {
    const double* M = d.Minv + 49 * (size_t)a + 7 * c;
    double z = 0;
#pragma unroll
    for (int q = 0; q < 7; q++) z += M[q] * rs[(t & ~7) + q];
    d.z[g] = z;
    rz = rv * z;
  }
  const double tot = block_sum(rz, lds);
  if (t == 0) {
    d.prz[(k + 1) & 1][blockIdx.x] = tot;
    if (blockIdx.x == 0) d.flag[1] = k + 1;
  }
}

// trial state = oplus(current, x) for free vertices, copy for fixed ones; partial of sum x (lambda x + b)
__global__ __launch_bounds__(kTPB) void pg_apply(PgDev d, int cur, double lambda) {
  __shared__ double lds[kTPB / kWave];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  double sc = 0;
  if (v < d.V) {
    const int a = d.slot[v];
    Sim3d S = sim3_load(d.S[cur] + 8 * (size_t)v);
    if (a >= 0) {
      double u[7];
#pragma unroll
      for (int k = 0; k < 7; k++) { u[k] = d.x[7 * (size_t)a + k]; sc += u[k] * (lambda * u[k] + d.b[7 * (size_t)a + k]); }
      S = pg_oplus(S, u, d.fix_scale);
    }
    sim3_store(d.S[cur ^ 1] + 8 * (size_t)v, S);
  }
  const double s = block_sum(sc, lds);
  if (threadIdx.x == 0) d.part[blockIdx.x] = s;
}


// ---- spanning-forest preconditioner ------------------------------------------------------------------------------
// Block-Jacobi PCG needs O(chain length) iterations on a pose graph (measured 923 per solve at 400 keyframes, ~4000 at
// 2000) because the low-frequency modes of the keyframe chain are invisible to a local preconditioner.  The tree part of
// H, T = sum over the edges of a spanning forest of [Jk Jp]^T [Jk Jp], can be inverted exactly and in parallel:
// with B_root = I and, down the tree, W_k = -Jp B_p^-1, B_k = W_k^-1 Jk, every tree residual reads
// Jk dk + Jp dp = W_k (B_k dk - B_p dp), i.e. in the transported variables y = B d the tree is a plain difference operator
// G and T = B^T G^T D G B with D_k = W_k^T W_k.  Hence T^-1 = B^-1 P D^-1 P^T B^-T where P sums along root paths (P^T sums
// over subtrees): two prefix scans in DFS order and three small mat-vecs per node.  The off-tree (covisibility) edges have
// bounded stretch, so the preconditioned system needs ~110 iterations whatever the number of keyframes (measured 105 /
// 117 at 120 / 400 keyframes in an offline study of the same matrices).  The forest is chosen on the host (Kruskal by
// |i - j|, rooted at the fixed vertices), the recursion runs on the device once per LM iteration (J changes, lambda does
// not enter: it is 1e-16 in the reference).
constexpr int kPgTreeMaxF = 2600;   // DFS-ordered 7-vectors of all nodes must fit the LDS of the single apply workgroup

// 7x7 inverse by Gauss-Jordan with partial pivoting, one wave, matrices in LDS ([49] row-major); A is destroyed
__device__ __forceinline__ void inv7_wave(double* A, double* Ainv, int lane) {
  const int r = lane / 7, c = lane % 7;
  if (lane < 49) Ainv[lane] = (r == c) ? 1.0 : 0.0;
  __syncthreads();
  for (int col = 0; col < 7; col++) {
    int piv = col;
    double bv = fabs(A[col * 7 + col]);
    for (int rr = col + 1; rr < 7; rr++) { const double v = fabs(A[rr * 7 + col]); if (v > bv) { bv = v; piv = rr; } }   // uniform: every lane computes it
    __syncthreads();
    if (piv != col && lane < 7) {
      const double a0 = A[col * 7 + lane], a1 = A[piv * 7 + lane]; A[col * 7 + lane] = a1; A[piv * 7 + lane] = a0;
      const double b0 = Ainv[col * 7 + lane], b1 = Ainv[piv * 7 + lane]; Ainv[col * 7 + lane] = b1; Ainv[piv * 7 + lane] = b0;
    }
    __syncthreads();
    const double pinv = 1.0 / A[col * 7 + col];
    __syncthreads();
    if (lane < 7) { A[col * 7 + lane] *= pinv; Ainv[col * 7 + lane] *= pinv; }
    __syncthreads();
    double fa = 0, fi = 0, f = 0;
    if (lane < 49 && r != col) { f = A[r * 7 + col]; fa = A[col * 7 + c]; fi = Ainv[col * 7 + c]; }
    __syncthreads();
    if (lane < 49 && r != col) { A[r * 7 + c] -= f * fa; Ainv[r * 7 + c] -= f * fi; }
    __syncthreads();
  }
}

// one wave walks the forest in DFS preorder (parents first): Binv_k = Jk^-1 W_k, Dinv_k = W_k^-1 W_k^-T
__global__ __launch_bounds__(kWave) void pg_tree_setup(PgDev d, double lambda) {
  __shared__ double Jk[49], Jp[49], W[49], Wi[49], Ji[49], tmp[49];
  const int lane = threadIdx.x;
  const int r = lane / 7, c = lane % 7;
  for (int pos = 0; pos < d.F; pos++) {
    const int a = d.t_order[pos];
    const int code = d.t_code[a], par = d.t_parent[a];
    if (code < 0) {
      // root of a component without a fixed vertex: no tree edge, D = its own damped diagonal block, B = I
      if (lane < 49) { W[lane] = d.H[49 * (size_t)a + lane] + ((r == c) ? lambda + 1e-12 : 0.0); }
      __syncthreads();
      inv7_wave(W, Wi, lane);
      if (lane < 49) { d.Dinv[49 * (size_t)a + lane] = Wi[lane]; d.Binv[49 * (size_t)a + lane] = (r == c) ? 1.0 : 0.0; }
      __syncthreads();
      continue;
    }
    const int e = code >> 1, side = code & 1;
    if (lane < 49) {
      double jk = d.J[98 * (size_t)e + 49 * side + lane], jp = d.J[98 * (size_t)e + 49 * (side ^ 1) + lane];
      if (d.fix_scale) {   // the scale row / column of a fixed-scale graph is identically zero: decouple it with a unit entry
        if (r == 6 || c == 6) { jk = (r == 6 && c == 6) ? 1.0 : 0.0; jp = (r == 6 && c == 6) ? -1.0 : 0.0; }
      }
      Jk[lane] = jk; Jp[lane] = jp;
    }
    __syncthreads();
    if (lane < 49) {
      if (par < 0) W[lane] = Jk[lane];       // parent is a fixed vertex: residual = Jk dk, B = I
      else {
        double acc = 0;
        const double* Bp = d.Binv + 49 * (size_t)par;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += Jp[r * 7 + k] * Bp[k * 7 + c];
        W[lane] = -acc;
      }
      tmp[lane] = (par < 0) ? Jk[lane] : W[lane];
    }
    __syncthreads();
    if (lane < 49) W[lane] = tmp[lane];
    __syncthreads();
    if (lane < 49) tmp[lane] = W[lane];       // keep W: inv7 destroys its input
    __syncthreads();
    inv7_wave(tmp, Wi, lane);                 // Wi = W^-1
    if (lane < 49) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) acc += Wi[r * 7 + k] * Wi[c * 7 + k];   // W^-1 W^-T
      d.Dinv[49 * (size_t)a + lane] = acc;
    }
    if (par < 0) {
      if (lane < 49) d.Binv[49 * (size_t)a + lane] = (r == c) ? 1.0 : 0.0;
    } else {
      if (lane < 49) tmp[lane] = Jk[lane];
      __syncthreads();
      inv7_wave(tmp, Ji, lane);               // Ji = Jk^-1
      if (lane < 49) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += Ji[r * 7 + k] * W[k * 7 + c];   // B^-1 = Jk^-1 W
        d.Binv[49 * (size_t)a + lane] = acc;
      }
    }
    __threadfence();
    __syncthreads();
  }
}

// inclusive scan over positions of the F x 7 array in LDS (component-wise), 1024 threads
__device__ __forceinline__ void pg_scan7(double* buf, int F, double* tot /* [128*8] */) {
  const int t = threadIdx.x;
  const int c = t & 7, chunk = t >> 3;            // 128 chunks x 8 lanes (7 components)
  const int per = (F + 127) / 128;
  const int p0 = chunk * per, p1 = min(F, p0 + per);
  double s = 0;
  if (c < 7) for (int p = p0; p < p1; p++) { s += buf[p * 7 + c]; buf[p * 7 + c] = s; }
  tot[chunk * 8 + c] = s;
  __syncthreads();
  if (t < 7) { double run = 0; for (int k = 0; k < 128; k++) { const double v = tot[k * 8 + t]; tot[k * 8 + t] = run; run += v; } }   // exclusive chunk offsets
  __syncthreads();
  if (c < 7) { const double o = tot[chunk * 8 + c]; for (int p = p0; p < p1; p++) buf[p * 7 + c] += o; }
  __syncthreads();
}

// z = B^-1 P D^-1 P^T B^-T r and r.z, one 1024-thread workgroup; slot = which prz buffer receives r.z
__global__ __launch_bounds__(1024) void pg_precond(PgDev d, int slot_out, int k_next) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* buf = sm;                         // [F*7]
  double* tot = sm + 7 * (size_t)d.F;       // [128*8]
  double* red = tot + 1024;                 // [16]
  if (d.flag[0]) return;
  const int t = threadIdx.x;
  const int F = d.F;
  // a. t = B^-T r, stored at the node's DFS position
  for (int i = t; i < 8 * F; i += 1024) {
    const int a = i >> 3, c = i & 7;
    if (c < 7) {
      const double* Bi = d.Binv + 49 * (size_t)a;
      double acc = 0;
#pragma unroll
      for (int q = 0; q < 7; q++) acc += Bi[q * 7 + c] * d.r[7 * (size_t)a + q];
      buf[d.t_pos[a] * 7 + c] = acc;
    }
  }
  __syncthreads();
  pg_scan7(buf, F, tot);
  // b. u = subtree sums, w = D^-1 u  (u through global scratch q, w through global scratch p-unused slot? -> x is live, use q and z)
  for (int i = t; i < 8 * F; i += 1024) {
    const int a = i >> 3, c = i & 7;
    if (c < 7) {
      const int p0 = d.t_pos[a], p1 = d.t_out[a];
      d.q[7 * (size_t)a + c] = buf[(p1 - 1) * 7 + c] - (p0 > 0 ? buf[(p0 - 1) * 7 + c] : 0.0);
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = t; i < 8 * F; i += 1024) {
    const int a = i >> 3, c = i & 7;
    if (c < 7) {
      const double* Di = d.Dinv + 49 * (size_t)a + 7 * c;
      double acc = 0;
#pragma unroll
      for (int q = 0; q < 7; q++) acc += Di[q] * d.q[7 * (size_t)a + q];
      d.z[7 * (size_t)a + c] = acc;            // w, parked in z
    }
  }
  __threadfence_block();
  __syncthreads();
  // c. ancestor sums: difference array over DFS positions, then a prefix scan
  for (int i = t; i < 8 * F; i += 1024) {
    const int pos = i >> 3, c = i & 7;
    if (c < 7) {
      double v = d.z[7 * (size_t)d.t_order[pos] + c];
      for (int s = d.ends_off[pos]; s < d.ends_off[pos + 1]; s++) v -= d.z[7 * (size_t)d.ends_idx[s] + c];
      buf[pos * 7 + c] = v;
    }
  }
  __syncthreads();
  pg_scan7(buf, F, tot);
  // d. z = B^-1 y ; r.z
  double rz = 0;
  for (int i = t; i < 8 * F; i += 1024) {
    const int a = i >> 3, c = i & 7;
    if (c < 7) {
      const double* Bi = d.Binv + 49 * (size_t)a + 7 * c;
      const double* y = buf + d.t_pos[a] * 7;
      double acc = 0;
#pragma unroll
      for (int q = 0; q < 7; q++) acc += Bi[q] * y[q];
      d.q[7 * (size_t)a + c] = acc;            // final z, staged in q (z still holds w for other threads)
      rz += d.r[7 * (size_t)a + c] * acc;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = t; i < 7 * F; i += 1024) d.z[i] = d.q[i];
  rz = wave_sum(rz);
  if ((t & (kWave - 1)) == 0) red[t / kWave] = rz;
  __syncthreads();
  if (t == 0) {
    double s = 0;
    for (int w = 0; w < 16; w++) s += red[w];
    d.prz[slot_out][0] = s;
    // The tree inverse amplifies the rounding noise of r in the low-frequency directions, so on a right-hand side that is
    // itself noise (LM trials at convergence) r.z can stall above rel_tol^2 * r0.z0 although the iterate has reached the
    // accuracy f64 allows.  Stop when r.z has not improved by 10 % for 40 iterations.
    if (k_next == 0) { d.scal[6] = s; d.scal[7] = 0; }
    else if (s < 0.9 * d.scal[6]) { d.scal[6] = s; d.scal[7] = (double)k_next; }
    else if ((double)k_next - d.scal[7] > 40.0) { d.flag[0] = 1; d.flag[1] = k_next; d.flag[3] = 1; }
  }
}

// tree-preconditioned variant of the PCG update: alpha, x, r only (z and r.z come from pg_precond)
__global__ __launch_bounds__(kTPB) void pg_pcg_update_xr(PgDev d, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int done = d.flag[0];
  const double rz_k = sum_partials(d.prz[k & 1], d.n_wg_upd);
  const double pq = sum_partials(d.ppq, d.n_wg_row);
  if (done) return;
  if (!(pq > 0.0)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.flag[0] = 1; d.flag[1] = k; d.flag[2] = 1; }
    return;
  }
  const double alpha = rz_k / pq;
  if (i < 7 * d.F) {
    const double* p = d.p[(k + 1) & 1];
    d.x[i] += alpha * p[i];
    d.r[i] -= alpha * d.q[i];
  }
  if (i == 0) d.flag[1] = k + 1;
}

// tree variant of the start: x = 0, r = b, p_{-1} = 0 (z and r.z come from pg_precond)
__global__ __launch_bounds__(kTPB) void pg_pcg_start(PgDev d, double lambda, double rel_tol) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 7 * d.F) { d.x[i] = 0; d.r[i] = d.b[i]; d.p[0][i] = 0; }
  if (i == 0) { d.scal[1] = rel_tol * rel_tol; d.scal[2] = lambda; }
}


// ---- exact dense solve (default) -----------------------------------------------------------------------------------
// g2o solves the pose graph with a sparse direct factorisation; reproducing its LM path (which steps are accepted)
// needs the weakly constrained chain modes of the step resolved exactly, which is precisely what an iterative solver
// finds last.  The system is small (7 x keyframes <= ~14 000 unknowns), so the default scatters the block matrix into a
// dense array and factors it with the blocked MFMA-f64 Cholesky of dense_chol.hip.
// dense A (row-major N x N, N = 7F rounded up to 64, full symmetric, identity on the padding) = H + lambda I; rhs = b
__global__ __launch_bounds__(kTPB) void pg_dense_fill(PgDev d, const int* blk_a, const int* blk_b, double* A, double* rhs, double lambda, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = 7 * d.F;
  if (i < d.nBlk * 49) {
    const int k = i / 49, el = i % 49, p = el / 7, q = el % 7;
    const int a = blk_a[k], b = blk_b[k];
    double v = d.H[i];
    if (a == b && p == q) v += lambda;
    A[(7 * (size_t)a + p) * N + (7 * (size_t)b + q)] = v;
    if (a != b) A[(7 * (size_t)b + q) * N + (7 * (size_t)a + p)] = v;
  }
  if (i < N) { rhs[i] = (i < n) ? d.b[i] : 0.0; if (i >= n) A[(size_t)i * N + i] = 1.0; }
}

// tile-sparse form (ccm_tsc): vertex a sits at dense offset dpos[a] (nested-dissection order, pieces padded to whole tiles); only tiles of the
// lower triangle exist; padding unknowns (dsrc < 0) get a unit diagonal
__global__ __launch_bounds__(kTPB) void pg_tile_fill(PgDev d, const int* blk_a, const int* blk_b, const int* dpos, const int* dsrc, const int* tid, int T, double* tiles,
                                                     double* rhs, double lambda, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nBlk * 49) {
    const int k = i / 49, el = i % 49, p = el / 7, q = el % 7;
    const int a = blk_a[k], b = blk_b[k];
    double v = d.H[i];
    if (a == b && p == q) v += lambda;
    const int ra = dpos[a] + p, rb = dpos[b] + q;
    const int ta = ra >> 6, tb = rb >> 6;
    if (ta >= tb) tiles[(size_t)tid[(size_t)ta * T + tb] * 4096 + (size_t)(ra & 63) * 64 + (rb & 63)] = v;
    if (a != b && tb >= ta) tiles[(size_t)tid[(size_t)tb * T + ta] * 4096 + (size_t)(rb & 63) * 64 + (ra & 63)] = v;
  }
  if (i < N) {
    const int src = dsrc[i];
    rhs[i] = src >= 0 ? d.b[src] : 0.0;
    if (src < 0) tiles[(size_t)tid[(size_t)(i >> 6) * T + (i >> 6)] * 4096 + (size_t)(i & 63) * 65] = 1.0;
  }
}
__global__ __launch_bounds__(kTPB) void pg_tile_extract(const double* rhs, const int* dsrc, int N, double* x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N && dsrc[i] >= 0) x[dsrc[i]] = rhs[i];
}

// Nested-dissection order of the free vertices for the tile-sparse factorisation: recursive bisection by the middle level of a BFS from a
// pseudo-peripheral vertex (essential graphs are chains with short covisibility links and a few loop edges: a BFS level is a handful of
// keyframes); pieces = leaves (<= leaf vertices, BFS order: banded inside) and separators, emitted children first, separator after them.
// Every piece starts on a tile boundary (64 unknowns) so that sibling subtrees share no tile and eliminate in parallel.
static void pg_nd_order(int F, const std::vector<std::vector<int>>& adj, int leaf, std::vector<std::vector<int>>* pieces) {
  std::vector<int> mark(F, -1), dist(F, 0), queue;
  int stamp = 0;
  // BFS inside the vertex set tagged `tag` in `mark`; returns the visit order, dist[] filled
  auto bfs = [&](int start, int tag, std::vector<int>& order) {
    order.clear(); order.push_back(start); dist[start] = 0; mark[start] = tag + 1;   // tag + 1 = visited in this pass
    for (size_t h = 0; h < order.size(); h++) {
      const int u = order[h];
      for (int v : adj[u]) if (mark[v] == tag) { mark[v] = tag + 1; dist[v] = dist[u] + 1; order.push_back(v); }
    }
  };
  std::vector<std::vector<int>> stack;
  {
    std::vector<int> all(F);
    for (int a = 0; a < F; a++) all[a] = a;
    stack.push_back(all);
  }
  // explicit recursion: entries are either vertex sets to dissect or (flagged by a leading -1) finished pieces to emit in order
  std::vector<int> order, order2;
  while (!stack.empty()) {
    std::vector<int> S = std::move(stack.back()); stack.pop_back();
    if (!S.empty() && S[0] == -1) { pieces->emplace_back(S.begin() + 1, S.end()); continue; }
    if (S.empty()) continue;
    const int tag = (stamp += 2);
    for (int v : S) mark[v] = tag;
    bfs(S[0], tag, order);
    if (order.size() < S.size()) {   // disconnected: no separator between components; small ones share pieces (a piece costs at least one tile)
      std::vector<std::vector<int>> comps(1, order);
      for (int v : S) if (mark[v] == tag) { bfs(v, tag, order); comps.push_back(order); }
      std::vector<int> bin(1, -1);
      for (auto& cc : comps) {
        if ((int)cc.size() > leaf / 2) { stack.push_back(cc); continue; }
        if ((int)bin.size() - 1 + (int)cc.size() > leaf) { stack.push_back(bin); bin.assign(1, -1); }
        bin.insert(bin.end(), cc.begin(), cc.end());
      }
      if (bin.size() > 1) stack.push_back(bin);
      continue;
    }
    // pseudo-peripheral start: restart from the last vertex reached
    for (int v : S) mark[v] = tag;
    bfs(order.back(), tag, order2);
    if ((int)S.size() <= leaf) { order2.insert(order2.begin(), -1); stack.push_back(order2); continue; }
    const int depth = dist[order2.back()];
    if (depth < 2) { order2.insert(order2.begin(), -1); stack.push_back(order2); continue; }   // a clique-like blob: no useful separator
    // middle level by vertex count
    std::vector<int> cnt(depth + 1, 0);
    for (int v : order2) cnt[dist[v]]++;
    int m = 1, acc = cnt[0];
    while (m < depth - 1 && acc + cnt[m] < (int)S.size() / 2) { acc += cnt[m]; m++; }
    std::vector<int> A, B, sep(1, -1);
    for (int v : order2) { if (dist[v] < m) A.push_back(v); else if (dist[v] == m) sep.push_back(v); else B.push_back(v); }
    stack.push_back(sep);   // popped last: after both halves
    stack.push_back(B);
    stack.push_back(A);
  }
}

typedef std::vector<std::pair<void*, size_t>> PgAllocs;   // pooled blocks (ccm_pool_get / ccm_pool_put)
template <typename T>
int up(ccm_ctx* ctx, PgAllocs& allocs, const std::vector<T>& v, T** out) {
  void* p = nullptr;
  size_t actual = 0;
  if (int rc = ccm_pool_get(ctx, std::max<size_t>(v.size(), 1) * sizeof(T), &p, &actual)) return rc;
  allocs.push_back({p, actual});
  if (!v.empty()) CCM_HIP_CHECK(ctx, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  *out = (T*)p;
  return CCM_OK;
}
template <typename T>
int al(ccm_ctx* ctx, PgAllocs& allocs, size_t n, T** out) {
  void* p = nullptr;
  size_t actual = 0;
  if (int rc = ccm_pool_get(ctx, std::max<size_t>(n, 1) * sizeof(T), &p, &actual)) return rc;
  allocs.push_back({p, actual});
  CCM_HIP_CHECK(ctx, hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(T), ctx->stream));
  *out = (T*)p;
  return CCM_OK;
}

}  // namespace

#define PG_RC(x) do { int rc_ = (x); if (rc_) { hipStreamSynchronize(ctx->stream); for (auto& p_ : allocs) ccm_pool_put(ctx, p_.first, p_.second); return rc_; } } while (0)

extern "C" int ccm_pose_graph_optimize(ccm_ctx* ctx, int n_vert, double* sim3, const uint8_t* fixed, int fix_scale, int n_edge,
                                       const int32_t* e_i, const int32_t* e_j, const double* meas, int max_iters, double lambda_init,
                                       const volatile unsigned char* stop_flag, ccm_pg_stats* stats) {
  if (!ctx || n_vert < 0 || n_edge < 0 || (n_vert && (!sim3 || !fixed)) || (n_edge && (!e_i || !e_j || !meas)) || max_iters < 0)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_pose_graph_optimize: bad args");
  ccm_pg_stats st{};
  if (stats) *stats = st;
  for (int e = 0; e < n_edge; e++)
    if (e_i[e] < 0 || e_i[e] >= n_vert || e_j[e] < 0 || e_j[e] >= n_vert || e_i[e] == e_j[e])
      return ccm_set_error(ctx, CCM_E_ARG, "ccm_pose_graph_optimize: edge vertex index out of range");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // ---- structure (host): slots, active edges, blocks, gather lists, block-CSR rows ----
  std::vector<int> slot(n_vert, -1), free_v;
  for (int v = 0; v < n_vert; v++) if (!fixed[v]) { slot[v] = (int)free_v.size(); free_v.push_back(v); }
  const int F = (int)free_v.size();
  std::vector<int> ei, ej; std::vector<double> ms;
  for (int e = 0; e < n_edge; e++) {
    if (fixed[e_i[e]] && fixed[e_j[e]]) continue;       // not in the active set (all vertices fixed)
    ei.push_back(e_i[e]); ej.push_back(e_j[e]);
    ms.insert(ms.end(), meas + 8 * (size_t)e, meas + 8 * (size_t)e + 8);
  }
  const int E = (int)ei.size();
  if (F == 0 || E == 0) return CCM_OK;
  std::map<std::pair<int, int>, int> blk;
  std::vector<std::pair<int, int>> keys;
  for (int a = 0; a < F; a++) { blk[{a, a}] = a; keys.push_back({a, a}); }
  for (int e = 0; e < E; e++) {
    const int a = slot[ei[e]], b = slot[ej[e]];
    if (a < 0 || b < 0) continue;
    const std::pair<int, int> k{std::min(a, b), std::max(a, b)};
    if (!blk.count(k)) { blk[k] = (int)keys.size(); keys.push_back(k); }
  }
  const int nBlk = (int)keys.size();
  std::vector<std::vector<int>> bl(nBlk), vl(F);
  for (int e = 0; e < E; e++) {
    const int a = slot[ei[e]], b = slot[ej[e]];
    if (a >= 0) { bl[a].push_back(e * 4 + 0); vl[a].push_back(e * 2 + 0); }                  // Ji^T Ji
    if (b >= 0) { bl[b].push_back(e * 4 + 3); vl[b].push_back(e * 2 + 1); }                  // Jj^T Jj
    if (a >= 0 && b >= 0) bl[blk[{std::min(a, b), std::max(a, b)}]].push_back(e * 4 + (a < b ? 1 : 2));   // rows = lower slot's side
  }
  std::vector<int> blk_off(nBlk + 1, 0), blk_edge, vtx_off(F + 1, 0), vtx_edge;
  for (int k = 0; k < nBlk; k++) { blk_edge.insert(blk_edge.end(), bl[k].begin(), bl[k].end()); blk_off[k + 1] = (int)blk_edge.size(); }
  for (int a = 0; a < F; a++) { vtx_edge.insert(vtx_edge.end(), vl[a].begin(), vl[a].end()); vtx_off[a + 1] = (int)vtx_edge.size(); }
  std::vector<std::vector<std::pair<int, uint32_t>>> rows(F);
  for (int k = 0; k < nBlk; k++) {
    const int a = keys[k].first, b = keys[k].second;
    rows[a].push_back({b, (uint32_t)k});
    if (a != b) rows[b].push_back({a, (uint32_t)k | kT});
  }
  std::vector<int> row_off(F + 1, 0), row_col; std::vector<uint32_t> row_blk;
  for (int a = 0; a < F; a++) {
    std::sort(rows[a].begin(), rows[a].end(), [](const std::pair<int, uint32_t>& x, const std::pair<int, uint32_t>& y) { return x.first < y.first; });
    for (auto& pr : rows[a]) { row_col.push_back(pr.first); row_blk.push_back(pr.second); }
    row_off[a + 1] = (int)row_col.size();
  }
  // ---- spanning forest for the preconditioner: Kruskal by |i - j| (local edges first), all fixed vertices = one root ----
  const bool use_tree = F <= kPgTreeMaxF;
  std::vector<int> t_parent(F, -1), t_code(F, -1), t_order, t_pos(F, 0), t_out(F, 0), ends_off(F + 2, 0), ends_idx;
  if (use_tree) {
    std::vector<int> eo(E);
    for (int e = 0; e < E; e++) eo[e] = e;
    std::stable_sort(eo.begin(), eo.end(), [&](int x, int y) { return std::abs(ei[x] - ej[x]) < std::abs(ei[y] - ej[y]); });
    std::vector<int> uf(F + 1);                       // node F = the fixed super-node
    for (int i = 0; i <= F; i++) uf[i] = i;
    auto find = [&](int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
    std::vector<std::vector<std::pair<int, int>>> adj(F + 1);   // (neighbour, edge)
    for (int e : eo) {
      const int a = slot[ei[e]] < 0 ? F : slot[ei[e]], b = slot[ej[e]] < 0 ? F : slot[ej[e]];
      const int ra = find(a), rb = find(b);
      if (ra == rb) continue;
      uf[ra] = rb;
      adj[a].push_back({b, e}); adj[b].push_back({a, e});
    }
    // DFS preorder from the fixed super-node, then from every still unvisited node (components without a fixed vertex)
    std::vector<char> seen(F + 1, 0);
    std::vector<std::pair<int, int>> stack;          // (node, next child index)
    auto dfs = [&](int root) {
      stack.clear(); stack.push_back({root, 0}); seen[root] = 1;
      if (root < F) { t_pos[root] = (int)t_order.size(); t_order.push_back(root); }
      while (!stack.empty()) {
        auto& top = stack.back();
        const int u = top.first;
        if (top.second < (int)adj[u].size()) {
          const auto [v, e] = adj[u][top.second++];
          if (seen[v]) continue;
          seen[v] = 1;
          t_parent[v] = (u == F) ? -1 : u;
          t_code[v] = e * 2 + (slot[ei[e]] == v ? 0 : 1);
          t_pos[v] = (int)t_order.size(); t_order.push_back(v);
          stack.push_back({v, 0});
        } else {
          if (u < F) t_out[u] = (int)t_order.size();
          stack.pop_back();
        }
      }
    };
    dfs(F);
    for (int a = 0; a < F; a++) if (!seen[a]) dfs(a);
    std::vector<std::vector<int>> ends(F + 1);
    for (int a = 0; a < F; a++) ends[t_out[a]].push_back(a);
    for (int pos = 0; pos <= F; pos++) { ends_idx.insert(ends_idx.end(), ends[pos].begin(), ends[pos].end()); ends_off[pos + 1] = (int)ends_idx.size(); }
  }
  // ---- device state ----
  PgAllocs allocs;
  PgDev d{};
  d.V = n_vert; d.F = F; d.E = E; d.nBlk = nBlk; d.fix_scale = fix_scale ? 1 : 0;
  std::vector<double> s0(sim3, sim3 + 8 * (size_t)n_vert);
  int *p_i = nullptr; uint32_t* p_u = nullptr; double* p_d = nullptr;
  PG_RC(up(ctx, allocs, s0, &d.S[0])); PG_RC(up(ctx, allocs, s0, &d.S[1]));
  PG_RC(up(ctx, allocs, slot, &p_i)); d.slot = p_i;
  PG_RC(up(ctx, allocs, free_v, &p_i)); d.free_v = p_i;
  PG_RC(up(ctx, allocs, ei, &p_i)); d.ei = p_i;
  PG_RC(up(ctx, allocs, ej, &p_i)); d.ej = p_i;
  PG_RC(up(ctx, allocs, ms, &p_d)); d.meas = p_d;
  PG_RC(up(ctx, allocs, blk_off, &p_i)); d.blk_off = p_i;
  PG_RC(up(ctx, allocs, blk_edge, &p_i)); d.blk_edge = p_i;
  PG_RC(up(ctx, allocs, vtx_off, &p_i)); d.vtx_off = p_i;
  PG_RC(up(ctx, allocs, vtx_edge, &p_i)); d.vtx_edge = p_i;
  PG_RC(up(ctx, allocs, row_off, &p_i)); d.row_off = p_i;
  PG_RC(up(ctx, allocs, row_col, &p_i)); d.row_col = p_i;
  PG_RC(up(ctx, allocs, row_blk, &p_u)); d.row_blk = p_u;
  // solver: exact dense Cholesky (default, reproduces the reference's LM path) or tree-preconditioned PCG (CCM_PG_SOLVER=pcg, or
  // graphs whose dense matrix would not fit the budget below: faster, but an inexact-Newton path)
  const char* solver_env = getenv("CCM_PG_SOLVER");
  const size_t n_dense = 7 * (size_t)F;
  const int N_dense = (int)((n_dense + 63) / 64) * 64;
  // exact solver, default form: tile-sparse, level-scheduled Cholesky over a nested-dissection order (ccm_tsc: only the non-zero tiles are
  // stored, so the size limit is the tile-pattern table, not a dense array); CCM_PG_SOLVER=dense keeps the round-1 form (dense array of
  // <= 24 000 unknowns, natural order, one tile column after the other) for comparison
  const bool want_exact = !(solver_env && !strcmp(solver_env, "pcg"));
  // (<= 20 000 keyframes: the symbolic step keeps dense T x T tile-pattern tables on host and device, T ~ 4000 tiles at that size, and packs tile
  // coordinates as (i << 16) | j in a signed int, i.e. T < 32768; larger graphs take the tree-preconditioned PCG path)
  const bool use_tiles = want_exact && !(solver_env && !strcmp(solver_env, "dense")) && n_dense <= 140000;
  const bool use_dense = want_exact && (use_tiles || n_dense <= 24000);
  double *d_A = nullptr, *d_rhs = nullptr, *d_linv = nullptr; int *d_info = nullptr, *d_blk_a = nullptr, *d_blk_b = nullptr;
  ccm_tile_plan plan;
  ccm_tsc tsc;
  int *d_dpos = nullptr, *d_dsrc = nullptr;
  int N_tiles = 0;
  if (use_tiles) {
    std::vector<int> ka(nBlk), kb(nBlk);
    for (int k = 0; k < nBlk; k++) { ka[k] = keys[k].first; kb[k] = keys[k].second; }
    PG_RC(up(ctx, allocs, ka, &d_blk_a)); PG_RC(up(ctx, allocs, kb, &d_blk_b));
    std::vector<std::vector<int>> adj(F);
    for (int k = 0; k < nBlk; k++) if (ka[k] != kb[k]) { adj[ka[k]].push_back(kb[k]); adj[kb[k]].push_back(ka[k]); }
    std::vector<std::vector<int>> pieces;
    const int nd_leaf = 31;   // 31 vertices = 217 unknowns = 4 tiles (2000 keyframes: leaf 31: 17.2 ms / 11 levels, 63: 18.9 / 13, 127: 23.4 / 19)
    pg_nd_order(F, adj, std::max(nd_leaf, 9), &pieces);
    std::vector<int> dpos(F, 0), dsrc;
    for (auto& pc : pieces) {
      for (int v : pc) { dpos[v] = (int)dsrc.size(); for (int c = 0; c < 7; c++) dsrc.push_back(7 * v + c); }
      while (dsrc.size() % 64) dsrc.push_back(-1);
    }
    N_tiles = (int)dsrc.size();
    const int T = N_tiles / 64;
    std::vector<char> nz((size_t)T * T, 0);
    for (int t = 0; t < T; t++) nz[(size_t)t * T + t] = 1;
    for (int k = 0; k < nBlk; k++) {
      const int a0 = dpos[ka[k]] / 64, a1 = (dpos[ka[k]] + 6) / 64, b0 = dpos[kb[k]] / 64, b1 = (dpos[kb[k]] + 6) / 64;
      for (int ta = a0; ta <= a1; ta++) for (int tb = b0; tb <= b1; tb++) nz[(size_t)std::max(ta, tb) * T + std::min(ta, tb)] = 1;
    }
    PG_RC(ccm_tsc_create(ctx, T, nz, &tsc));
    PG_RC(up(ctx, allocs, dpos, &d_dpos)); PG_RC(up(ctx, allocs, dsrc, &d_dsrc)); PG_RC(al(ctx, allocs, (size_t)N_tiles, &d_rhs));
    if (ccm_dbg("pg")) fprintf(stderr, "[ccm_pg] tile-sparse cholesky: %d unknowns in %d pieces -> %d tiles columns, %d non-zero tiles, %d levels\n", (int)n_dense, (int)pieces.size(), T, tsc.n_tiles, tsc.n_levels);
  } else if (use_dense) {
    std::vector<int> ka(nBlk), kb(nBlk);
    for (int k = 0; k < nBlk; k++) { ka[k] = keys[k].first; kb[k] = keys[k].second; }
    PG_RC(up(ctx, allocs, ka, &d_blk_a)); PG_RC(up(ctx, allocs, kb, &d_blk_b));
    PG_RC(al(ctx, allocs, (size_t)N_dense * N_dense, &d_A)); PG_RC(al(ctx, allocs, (size_t)N_dense, &d_rhs)); PG_RC(al(ctx, allocs, 4, &d_info));
    PG_RC(al(ctx, allocs, (size_t)N_dense * 64, &d_linv));
    // tile-level sparsity: the essential graph is a near-banded chain plus loop edges, so most 64x64 tiles of the factor
    // stay zero; the factorisation and the substitutions visit only the tiles a symbolic elimination marks
    const int T = N_dense / 64;
    std::vector<char> nz((size_t)T * T, 0);
    for (int t = 0; t < T; t++) nz[(size_t)t * T + t] = 1;
    for (int k = 0; k < nBlk; k++) {
      const int a0 = 7 * ka[k] / 64, a1 = (7 * ka[k] + 6) / 64, b0 = 7 * kb[k] / 64, b1 = (7 * kb[k] + 6) / 64;
      for (int ta = a0; ta <= a1; ta++) for (int tb = b0; tb <= b1; tb++) { nz[(size_t)std::max(ta, tb) * T + std::min(ta, tb)] = 1; }
    }
    std::vector<int> col_rows, upd_pairs, row_cols;
    ccm_tile_plan_symbolic(T, nz, &plan, &col_rows, &upd_pairs, &row_cols);
    int *p_cr = nullptr, *p_up = nullptr, *p_rc = nullptr;
    PG_RC(up(ctx, allocs, col_rows, &p_cr)); PG_RC(up(ctx, allocs, upd_pairs, &p_up)); PG_RC(up(ctx, allocs, row_cols, &p_rc));
    plan.d_col_rows = p_cr; plan.d_upd_pairs = p_up; plan.d_row_cols = p_rc;
  }
  d.use_tree = use_tree ? 1 : 0;
  if (use_tree) {
    PG_RC(up(ctx, allocs, t_parent, &p_i)); d.t_parent = p_i;
    PG_RC(up(ctx, allocs, t_code, &p_i)); d.t_code = p_i;
    PG_RC(up(ctx, allocs, t_order, &p_i)); d.t_order = p_i;
    PG_RC(up(ctx, allocs, t_pos, &p_i)); d.t_pos = p_i;
    PG_RC(up(ctx, allocs, t_out, &p_i)); d.t_out = p_i;
    PG_RC(up(ctx, allocs, ends_off, &p_i)); d.ends_off = p_i;
    PG_RC(up(ctx, allocs, ends_idx, &p_i)); d.ends_idx = p_i;
    PG_RC(al(ctx, allocs, 49 * (size_t)F, &d.Binv)); PG_RC(al(ctx, allocs, 49 * (size_t)F, &d.Dinv));
  }
  d.n_wg_row = ccm_div_up(F, kTPB / kWave); d.n_wg_upd = use_tree ? 1 : ccm_div_up(F, kTPB / 8); d.n_wg_edge = ccm_div_up(E, kTPB);
  const int n_wg_init = ccm_div_up(F, kTPB), n_wg_v = ccm_div_up(n_vert, kTPB);
  PG_RC(al(ctx, allocs, 7 * (size_t)E, &d.err)); PG_RC(al(ctx, allocs, 98 * (size_t)E, &d.J));
  PG_RC(al(ctx, allocs, 49 * (size_t)nBlk, &d.H)); PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.b));
  PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.x)); PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.r)); PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.z));
  PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.q)); PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.p[0])); PG_RC(al(ctx, allocs, 7 * (size_t)F, &d.p[1]));
  PG_RC(al(ctx, allocs, 49 * (size_t)F, &d.Minv));
  PG_RC(al(ctx, allocs, (size_t)d.n_wg_row, &d.ppq));
  // prz is written by pg_pcg_init (n_wg_init workgroups) and pg_pcg_update (n_wg_upd): size for the larger, sum n_wg_upd
  PG_RC(al(ctx, allocs, (size_t)std::max(d.n_wg_upd, n_wg_init), &d.prz[0])); PG_RC(al(ctx, allocs, (size_t)std::max(d.n_wg_upd, n_wg_init), &d.prz[1]));
  PG_RC(al(ctx, allocs, 8, &d.scal)); PG_RC(al(ctx, allocs, 4, &d.flag));
  PG_RC(al(ctx, allocs, (size_t)std::max({d.n_wg_edge, n_wg_v, 1}), &d.part));
  auto cleanup = [&]() { hipStreamSynchronize(ctx->stream); for (auto& p : allocs) ccm_pool_put(ctx, p.first, p.second); if (use_tiles) ccm_tsc_destroy(ctx, &tsc); };
  auto read_scal = [&](int idx, double* out) -> int {
    CCM_HIP_CHECK(ctx, hipMemcpyAsync(out, d.scal + idx, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return CCM_OK;
  };
  auto chi2_of = [&](int which, double* out) -> int {
    hipLaunchKernelGGL(pg_chi2, dim3(d.n_wg_edge), dim3(kTPB), 0, ctx->stream, d, which);
    hipLaunchKernelGGL(pg_reduce, dim3(1), dim3(kTPB), 0, ctx->stream, d, d.n_wg_edge, 4);
    return read_scal(4, out);
  };
  // ---- Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-164) ----
  int cur = 0, rc = CCM_OK;
  double lambda = lambda_init, ni = 2;
  int nBad = 0;
  double currentChi = 0;
  for (int iter = 0; iter < max_iters && rc == CCM_OK; iter++) {
    if (stop_flag && *stop_flag) break;
    if ((rc = chi2_of(cur, &currentChi))) break;
    if (iter == 0) st.chi2_initial = currentChi;
    const double iniChi = currentChi;
    hipLaunchKernelGGL(pg_linearize, dim3(ccm_div_up((int64_t)E * 32, kTPB)), dim3(kTPB), 0, ctx->stream, d, cur);
    hipLaunchKernelGGL(pg_assemble, dim3(ccm_div_up(nBlk + F, kTPB / kWave)), dim3(kTPB), 0, ctx->stream, d);
    if (use_tree && !use_dense) hipLaunchKernelGGL(pg_tree_setup, dim3(1), dim3(kWave), 0, ctx->stream, d, lambda);
    if (iter == 0 && !(lambda_init > 0)) {   // computeLambdaInit without a user value: tau * max diagonal
      std::vector<double> Hd(49 * (size_t)F);
      if (hipMemcpyAsync(Hd.data(), d.H, Hd.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = ccm_set_error(ctx, CCM_E_HIP, "pose graph: H readback"); break; }
      double m = 0;
      for (int a = 0; a < F; a++) for (int k = 0; k < 7; k++) m = std::max(m, std::fabs(Hd[49 * (size_t)a + k * 8]));
      lambda = 1e-5 * m;
    }
    double rho = 0;
    int qmax = 0;
    do {
      // solve (H + lambda I) x = b
      hipMemsetAsync(d.flag, 0, 4 * sizeof(int), ctx->stream);
      hipMemsetAsync(d.prz[0], 0, sizeof(double) * (size_t)std::max(d.n_wg_upd, n_wg_init), ctx->stream);   // stale partials of the previous solve
      hipMemsetAsync(d.prz[1], 0, sizeof(double) * (size_t)std::max(d.n_wg_upd, n_wg_init), ctx->stream);
      const size_t lds_pc = (7 * (size_t)F + 1024 + 16) * sizeof(double);
      if (use_dense) {
        int info = 0;
        if (use_tiles) {
          if ((rc = ccm_tsc_clear(ctx, &tsc))) break;
          hipLaunchKernelGGL(pg_tile_fill, dim3(ccm_div_up(std::max(nBlk * 49, N_tiles), kTPB)), dim3(kTPB), 0, ctx->stream, d, d_blk_a, d_blk_b, (const int*)d_dpos,
                             (const int*)d_dsrc, (const int*)tsc.d_tid, tsc.T, tsc.d_tiles, d_rhs, lambda, N_tiles);
          if ((rc = ccm_tsc_solve(ctx, &tsc, d_rhs))) break;
          hipLaunchKernelGGL(pg_tile_extract, dim3(ccm_div_up(N_tiles, kTPB)), dim3(kTPB), 0, ctx->stream, (const double*)d_rhs, (const int*)d_dsrc, N_tiles, d.x);
          if (hipMemcpyAsync(&info, tsc.d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            rc = ccm_set_error(ctx, CCM_E_HIP, "pose graph: tile solve readback"); break;
          }
        } else {
        hipMemsetAsync(d_A, 0, (size_t)N_dense * N_dense * sizeof(double), ctx->stream);
        hipLaunchKernelGGL(pg_dense_fill, dim3(ccm_div_up(std::max(nBlk * 49, N_dense), kTPB)), dim3(kTPB), 0, ctx->stream, d, d_blk_a, d_blk_b, d_A, d_rhs,
                           lambda, N_dense);
        if ((rc = ccm_dense_chol_solve_dev(ctx, d_A, N_dense, d_rhs, d_linv, d_info, &plan))) break;
        if (hipMemcpyAsync(d.x, d_rhs, n_dense * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
          rc = ccm_set_error(ctx, CCM_E_HIP, "pose graph: dense solve readback"); break;
        }
        }
        const bool ok_d = info == 0;     // > 0: leading minor not positive definite -> solver failure, LM rejects the step
        if (!ok_d) hipMemsetAsync(d.x, 0, 7 * (size_t)F * sizeof(double), ctx->stream);
        hipLaunchKernelGGL(pg_apply, dim3(n_wg_v), dim3(kTPB), 0, ctx->stream, d, cur, lambda);
        hipLaunchKernelGGL(pg_reduce, dim3(1), dim3(kTPB), 0, ctx->stream, d, n_wg_v, 5);
        double tempChi = 0, scale = 0;
        if ((rc = chi2_of(cur ^ 1, &tempChi))) break;
        if ((rc = read_scal(5, &scale))) break;
        st.lm_trials++;
        if (!ok_d) tempChi = DBL_MAX;
        scale += 1e-3;
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          cur ^= 1;
        } else {
          lambda *= ni; ni *= 2;
        }
        qmax++;
        continue;
      }
      if (use_tree) {
        if (!(ctx->lds_attr_done & (1u << CCM_LDS_PG_PRECOND))) {
          if (hipFuncSetAttribute((const void*)pg_precond, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess) { rc = ccm_set_error(ctx, CCM_E_HIP, "pose graph: LDS attribute"); break; }
          ctx->lds_attr_done |= 1u << CCM_LDS_PG_PRECOND;
        }
        hipLaunchKernelGGL(pg_pcg_start, dim3(ccm_div_up(7 * F, kTPB)), dim3(kTPB), 0, ctx->stream, d, lambda, 1e-10);
        hipLaunchKernelGGL(pg_precond, dim3(1), dim3(1024), lds_pc, ctx->stream, d, 0, 0);
      } else {
        hipLaunchKernelGGL(pg_pcg_init, dim3(n_wg_init), dim3(kTPB), 0, ctx->stream, d, lambda, 1e-10);
      }
      // pg_pcg_init leaves its partial sums in the first n_wg_init slots; the consumers sum n_wg_upd (>= n_wg_init) slots,
      // the remainder is zero from the allocation / stays zero
      int flags[4] = {0, 0, 0, 0};
      const int max_it = std::min(50000, 10 * 7 * F + 100);
      int k = 0;
      while (k < max_it) {
        const int kend = std::min(max_it, k + 32);
        for (; k < kend; k++) {
          hipLaunchKernelGGL(pg_pcg_spmv, dim3(d.n_wg_row), dim3(kTPB), 0, ctx->stream, d, k);
          if (use_tree) {
            hipLaunchKernelGGL(pg_pcg_update_xr, dim3(ccm_div_up(7 * F, kTPB)), dim3(kTPB), 0, ctx->stream, d, k);
            hipLaunchKernelGGL(pg_precond, dim3(1), dim3(1024), lds_pc, ctx->stream, d, (k + 1) & 1, k + 1);
          } else {
            hipLaunchKernelGGL(pg_pcg_update, dim3(d.n_wg_upd), dim3(kTPB), 0, ctx->stream, d, k);
          }
        }
        if (hipMemcpyAsync(flags, d.flag, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = ccm_set_error(ctx, CCM_E_HIP, "pose graph: flag readback"); break; }
        if (flags[0]) break;
      }
      if (rc) break;
      const bool ok2 = !flags[2];
      st.pcg_iters += flags[0] ? flags[1] : k;
      if (!ok2) hipMemsetAsync(d.x, 0, 7 * (size_t)F * sizeof(double), ctx->stream);
      hipLaunchKernelGGL(pg_apply, dim3(n_wg_v), dim3(kTPB), 0, ctx->stream, d, cur, lambda);
      hipLaunchKernelGGL(pg_reduce, dim3(1), dim3(kTPB), 0, ctx->stream, d, n_wg_v, 5);
      double tempChi = 0, scale = 0;
      if ((rc = chi2_of(cur ^ 1, &tempChi))) break;
      if ((rc = read_scal(5, &scale))) break;
      st.lm_trials++;
      if (!ok2) tempChi = DBL_MAX;
      scale += 1e-3;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        cur ^= 1;
      } else {
        lambda *= ni; ni *= 2;
      }
      qmax++;
    } while (rho < 0 && qmax < 10 && !(stop_flag && *stop_flag));
    if (rc) break;
    st.iters_done++;
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  if (rc == CCM_OK) {
    st.chi2_final = currentChi; st.lambda_final = lambda;
    if (hipMemcpyAsync(sim3, d.S[cur], 8 * (size_t)n_vert * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
      rc = ccm_set_error(ctx, CCM_E_HIP, "pose graph: result readback");
  }
  cleanup();
  if (stats) *stats = st;
  return rc;
}
