// dense_chol.hip — dense f64 Cholesky solve on the device, used for the small SPD system of the pose-graph optimiser
// (posegraph.hip: 7 x keyframes unknowns).  g2o factors that system with a sparse direct solver; reproducing its
// Levenberg-Marquardt path needs the step exact, and the matrix is small enough (<= ~2 GB) to treat densely.
//
// Blocked right-looking factorisation with 64x64 tiles, row-major lower triangle in place:
//   chol_diag_wave (1 wave)        L_jj = chol(A_jj), Li_jj = L_jj^-1 kept for the panel and the solves
//   chol_panel  (1 wg / tile row)  L_ij = A_ij Li_jj^T                     } one 64x64x64 product per workgroup on the
//   chol_update (1 wg / tile pair) A_ik -= L_ij L_kj^T  (i >= k > j)       } f64 matrix cores (v_mfma_f64_16x16x4_f64)
// then tile-wise forward / backward substitution.  MFMA operand layout (guide, "f64 MFMA does NOT use these maps"):
// A: lane l holds A[l & 15][l >> 4], B: lane l holds B[l >> 4][l & 15], D: reg r of lane l is D[(l >> 4) + 4 r][l & 15].
#include "common.h"
#include "test_internal.h"
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int NB = 64;            // tile size
constexpr int LD = NB + 1;        // padded LDS row stride (doubles): 16 lanes reading one column hit 16 different banks
constexpr int kTPB = 256;
typedef double v4d __attribute__((ext_vector_type(4)));

// D(64x64) = As(64x64) * Bs(64x64)^T, both in LDS with stride LD; wave w owns rows 16w..16w+15, acc[s] = 16x16 subtile s
__device__ __forceinline__ void tile_abt(const double* As, const double* Bs, v4d acc[4], int wave, int lane) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll 4
  for (int kk = 0; kk < NB / 4; kk++) {
    const int k = 4 * kk + kq;
    const double a = As[(16 * wave + i) * LD + k];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const double b = Bs[(16 * s + i) * LD + k];
      acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[s], 0, 0, 0);
    }
  }
}

// all 16 loads of a thread are issued before the first LDS store: as a rolled loop every tile paid the global latency 16
// times, which was most of the 13-16 us of the one-product kernels
__device__ __forceinline__ void load_tile(double* dst, const double* src, int ld_src) {
  double v[NB * NB / kTPB];
#pragma unroll
  for (int q = 0; q < NB * NB / kTPB; q++) { const int e = threadIdx.x + q * kTPB; v[q] = src[(size_t)(e / NB) * ld_src + e % NB]; }
#pragma unroll
  for (int q = 0; q < NB * NB / kTPB; q++) { const int e = threadIdx.x + q * kTPB; dst[(e / NB) * LD + e % NB] = v[q]; }
}

// L_jj and its inverse (info: first failing global column + 1, like LAPACK) by ONE wave: the diagonal tile is the serial
// link of the factorisation (the coarse level of the BA preconditioner is re-inverted for every LM trial, the pose-graph
// system has up to 219 of them per factorisation).  A 256-thread version with three block barriers per column took ~130 us per tile.
// Measured alternatives with one wave: tile in registers and every multiplier travelling by v_readlane, 88 us; tile in
// LDS, left-looking, two LDS reads per term, 96 us (nothing hides the LDS latency of a lone wave).  This one keeps lane
// i's ROW in registers and reads only the pivot row from LDS (uniform address = broadcast, two doubles per read), so a
// term costs one FMA plus half a read with no dependence between the reads: left-looking column form, fully unrolled,
// four accumulators.  L^-1 is a forward substitution, lane = column, x in registers, L rows broadcast the same way; it runs
// on a second wave one column behind the factorisation (58 us -> 38 with FMA and the prefetched pivot row -> 30 with
// all tile loads in flight -> 28 pipelined).
__device__ __forceinline__ double bcast64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// Two waves in a software pipeline: wave 0 factors column c while wave 1 does row c-1 of the forward substitution for
// L^-1 (it needs row c-1 of L and its reciprocal pivot, both final after step c-1); one block barrier per column.
// Ajj: the diagonal tile (row stride N doubles), Lg: where its inverse factor goes (64 x 64, contiguous), j: tile index for `info`
__device__ __forceinline__ void chol_diag_body(double* Ajj, size_t N, int j, double* Lg, int* info) {
  constexpr int LR = NB + 2;   // even row stride: rows stay 16-byte aligned for the paired broadcast reads
  __shared__ __attribute__((aligned(16))) double L[NB * LR];
  __shared__ double rdiag[NB];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {   // each wave brings half of the rows, all its loads in flight at once (a rolled loop pays the global latency per row)
    double tmp[NB / 2];
#pragma unroll
    for (int r = 0; r < NB / 2; r++) tmp[r] = Ajj[(size_t)(wv * (NB / 2) + r) * N + lane];
#pragma unroll
    for (int r = 0; r < NB / 2; r++) L[(wv * (NB / 2) + r) * LR + lane] = tmp[r];
  }
  __syncthreads();
  int bad_col = 0;
  double row[NB], cur[NB], nxt[NB];   // wave 0: lane i's row of the tile, current / next pivot row (ping-pong indexing of one 2-D array instead of the copy: 52 us, the array leaves the registers)
  double x[NB];                        // wave 1: lane t's column of L^-1
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < NB; k++) { row[k] = L[lane * LR + k]; cur[k] = 0.0; nxt[k] = 0.0; }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c <= NB; c++) {
    if (wv == 0) {
      if (c < NB) {
        // cur[k] = L_ck (k < c), fetched during the previous column.  lane i: A_ic - sum_{k<c} L_ik L_ck
        double acc[4] = {row[c], 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < c; k++) acc[k & 3] = __builtin_fma(-row[k], cur[k], acc[k & 3]);
        // prefetch the old part of the NEXT pivot row (entries k < c were stored in earlier columns) so that the LDS
        // latency overlaps the sqrt chain below; its newest entry L_{c+1,c} comes by v_readlane from lane c+1's register
        if (c + 1 < NB) {
#pragma unroll
          for (int k = 0; k < c; k++) nxt[k] = L[(c + 1) * LR + k];
        }
        const double sv = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        double dd = bcast64(sv, c);
        if (!(dd > 0.0)) { if (!bad_col) bad_col = c + 1; dd = 1.0; }
        const double inv = rsqrt(dd), l = dd * inv;
        row[c] = (lane == c) ? l : sv * inv;           // lanes < c hold the (unused) upper part
        L[lane * LR + c] = row[c];
        if (lane == c) rdiag[c] = inv;
        if (c + 1 < NB) nxt[c] = bcast64(row[c], c + 1);
#pragma unroll
        for (int k = 0; k <= c; k++) cur[k] = nxt[k];
      }
    } else if (c >= 1) {
      // X = L^-1, lane = column t: x_r = (delta_rt - sum_{k<r} L_rk x_k) / L_rr  (x_k = 0 for k < t, so the bounds are uniform)
      const int r = c - 1;
      const double* Lrow = L + r * LR;
      double acc[4] = {(r == lane) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < r; k++) acc[k & 3] = __builtin_fma(-Lrow[k], x[k], acc[k & 3]);
      x[r] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * rdiag[r];
      Lg[r * NB + lane] = x[r];
    }
    __syncthreads();
  }
  if (wv == 0) {
    if (bad_col && lane == 0 && *info == 0) *info = j * NB + bad_col;
#pragma unroll
    for (int r = 0; r < NB; r++) Ajj[(size_t)r * N + lane] = (lane <= r) ? L[r * LR + lane] : 0.0;
  }
}
// The same tile, BLOCKED by 16 columns (round 3).  The unblocked form above pays, for every one of its 64 columns, a chain of up to 16 dependent
// multiply-adds on operands fetched from LDS, a pivot broadcast, a store and a block barrier (~460 ns per column).  Here wave 0 takes 16 columns at a time:
//   (a) the block's 16 entries of every row are brought up to date with all EARLIER columns in one pass of independent chains (16 accumulators per lane,
//       the multipliers L_ck of the block's 16 pivot rows read from LDS at uniform addresses, two per read) — off the pivot chain;
//   (b) the 16 pivots of the block then run on registers alone: lane i keeps its row, a pivot row's entries travel by v_readlane, rows below the block
//       form their panel entries in the same instruction stream (the 16 x 16 tile factorisation of the BA cluster solver, 64 rows tall): ~200 ns per pivot;
//   (c) the block's columns go to LDS for the later blocks and for wave 1, which forms L^-1 one BLOCK behind (four barriers per tile instead of 64).
__device__ __forceinline__ void chol_diag_body_blk(double* Ajj, size_t N, int j, double* Lg, int* info) {
  constexpr int LR = NB + 2, KB = 16;
  __shared__ __attribute__((aligned(16))) double L[NB * LR];
  __shared__ double rdiag[NB];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    double tmp[NB / 2];
#pragma unroll
    for (int r = 0; r < NB / 2; r++) tmp[r] = Ajj[(size_t)(wv * (NB / 2) + r) * N + lane];
#pragma unroll
    for (int r = 0; r < NB / 2; r++) L[(wv * (NB / 2) + r) * LR + lane] = tmp[r];
  }
  __syncthreads();
  // The two waves run DIFFERENT loops with the same number of block barriers (wave 0 arrives at barrier b after block b is in LDS, wave 1 before it reads
  // it): in one shared loop the register allocator keeps wave 1's 64 values of x alive through wave 0's code as well, and both spill.
  // INVARIANT: each branch executes EXACTLY kBlocks barriers, one per trip of its `b` loop and none elsewhere (s_barrier counts arrivals per wave, not
  // call sites; the HIP model does not promise that, so this relies on the gfx950 barrier and on both trip counts being the one constant below).
  // The column-by-column form chol_diag_body (one shared barrier site per column) stays in the build behind CCM_CHOL_DIAG=columns and is compared with
  // this one by tests/test_ba_gpu.py::test_formulations_of_the_large_map_path_agree.
  constexpr int kBlocks = NB / KB;
  static_assert(NB % KB == 0 && kBlocks >= 1, "both waves must run the same whole number of 16-column blocks");
  if (wv == 0) {
    int bad_col = 0;
#pragma unroll
    for (int b = 0; b < kBlocks; b++) {
      const int c0 = KB * b;
      double row[KB];   // lane i's entries of the block's 16 columns (its earlier columns stay in LDS)
#pragma unroll
      for (int c = 0; c < KB; c += 2) { const double2 v = *reinterpret_cast<const double2*>(&L[lane * LR + c0 + c]); row[c] = v.x; row[c + 1] = v.y; }
      // (a) row[c] -= sum_{k < c0} L_ik L_{c0 + c, k}: both operands from LDS (the lane's own earlier entries; the pivot rows at uniform addresses)
#pragma unroll 2
      for (int k = 0; k < c0; k += 2) {
        const double2 a2 = *reinterpret_cast<const double2*>(&L[lane * LR + k]);
#pragma unroll
        for (int c = 0; c < KB; c++) {
          const double2 l2 = *reinterpret_cast<const double2*>(&L[(c0 + c) * LR + k]);
          row[c] = __builtin_fma(-a2.y, l2.y, __builtin_fma(-a2.x, l2.x, row[c]));
        }
      }
      // (b) the block's 16 pivots on registers
#pragma unroll
      for (int c = 0; c < KB; c++) {
        double s0 = row[c], s1 = 0.0;
#pragma unroll
        for (int k = 0; k < c; k++) {
          const double m = bcast64(row[k], c0 + c);
          if (k & 1) s1 = __builtin_fma(-row[k], m, s1); else s0 = __builtin_fma(-row[k], m, s0);
        }
        const double sv = s0 + s1;
        double dd = bcast64(sv, c0 + c);
        if (!(dd > 0.0)) { if (!bad_col) bad_col = c0 + c + 1; dd = 1.0; }
        const double inv = rsqrt(dd);
        row[c] = (lane == c0 + c) ? dd * inv : sv * inv;   // lanes above the diagonal hold unused values
        if (lane == c0 + c) rdiag[c0 + c] = inv;
      }
      // (c) the block's columns for the later blocks and for wave 1
#pragma unroll
      for (int c = 0; c < KB; c += 2) { double2 v; v.x = row[c]; v.y = row[c + 1]; *reinterpret_cast<double2*>(&L[lane * LR + c0 + c]) = v; }
      __syncthreads();
    }
    if (bad_col && lane == 0 && *info == 0) *info = j * NB + bad_col;
#pragma unroll
    for (int r = 0; r < NB; r++) Ajj[(size_t)r * N + lane] = (lane <= r) ? L[r * LR + lane] : 0.0;
  } else {
    double x[NB];     // lane t's column of L^-1
#pragma unroll
    for (int b = 0; b < kBlocks; b++) {   // (kBlocks barriers, see the invariant above)
      __syncthreads();
      // X = L^-1, lane = column t, rows of block b: x_r = (delta_rt - sum_{k<r} L_rk x_k) / L_rr  (x_k = 0 for k < t, so the bounds are uniform)
#pragma unroll
      for (int rr = 0; rr < KB; rr++) {
        const int r = KB * b + rr;
        const double* Lrow = L + r * LR;
        double acc[4] = {(r == lane) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < r; k++) acc[k & 3] = __builtin_fma(-Lrow[k], x[k], acc[k & 3]);
        x[r] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * rdiag[r];
        Lg[r * NB + lane] = x[r];
        asm volatile("" ::: "memory");   // keeps the next row's LDS reads behind this row
      }
    }
  }
}
__global__ __launch_bounds__(128) void chol_diag_wave(double* A, int N, int j, double* Linv_all, int* info) {
  chol_diag_body(A + ((size_t)j * NB) * N + (size_t)j * NB, (size_t)N, j, Linv_all + (size_t)j * NB * NB, info);
}
__global__ __launch_bounds__(128) void chol_diag_wave_blk(double* A, int N, int j, double* Linv_all, int* info) {
  chol_diag_body_blk(A + ((size_t)j * NB) * N + (size_t)j * NB, (size_t)N, j, Linv_all + (size_t)j * NB * NB, info);
}

// ---- tile-sparse, level-scheduled form (ccm_tsc_*): non-zero tiles of L stored compactly (64 x 64, row-major, contiguous), left-looking
// (every tile GATHERS its updates, no two workgroups write one tile), so all tile columns of one elimination-tree level run in one launch.
// One level = gather -> diagonal factor -> panel; the pose graph's nested-dissection order gives ~20-40 levels for 2000 keyframes where the
// right-looking column-by-column form made 3 launches for each of 219 columns and their diagonal tiles (27 us each) formed one chain.
__global__ __launch_bounds__(kTPB) void tsc_gather(double* tiles, const int* tid, int T, const int* gt, const int* gk_off, const int* gk) {
  __shared__ double As[NB * LD], Bs[NB * LD];
  const int g = blockIdx.x;
  const int i = gt[g] >> 16, j = gt[g] & 0xffff;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int q = gk_off[g]; q < gk_off[g + 1]; q++) {
    const int k = gk[q];
    load_tile(As, tiles + (size_t)tid[(size_t)i * T + k] * (NB * NB), NB);
    if (i != j) load_tile(Bs, tiles + (size_t)tid[(size_t)j * T + k] * (NB * NB), NB);
    __syncthreads();
    tile_abt(As, i != j ? Bs : As, acc, wave, lane);
    __syncthreads();
  }
  double* Aij = tiles + (size_t)tid[(size_t)i * T + j] * (NB * NB);
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) Aij[(size_t)(16 * wave + (lane >> 4) + 4 * r) * NB + 16 * s + (lane & 15)] -= acc[s][r];
}
static inline bool diag_blocked() { static const bool v = !(getenv("CCM_CHOL_DIAG") && !strcmp(getenv("CCM_CHOL_DIAG"), "columns")); return v; }

template <bool kBlocked>
__global__ __launch_bounds__(128) void tsc_diag(double* tiles, const int* tid, int T, const int* cols, double* Linv_all, int* info) {
  const int j = cols[blockIdx.x];
  if (kBlocked) chol_diag_body_blk(tiles + (size_t)tid[(size_t)j * T + j] * (NB * NB), (size_t)NB, j, Linv_all + (size_t)j * NB * NB, info);
  else chol_diag_body(tiles + (size_t)tid[(size_t)j * T + j] * (NB * NB), (size_t)NB, j, Linv_all + (size_t)j * NB * NB, info);
}
__global__ __launch_bounds__(kTPB) void tsc_panel(double* tiles, const int* tid, int T, const int* pt, const double* Linv_all) {
  __shared__ double As[NB * LD], Bs[NB * LD];
  const int i = pt[blockIdx.x] >> 16, j = pt[blockIdx.x] & 0xffff;
  double* Aij = tiles + (size_t)tid[(size_t)i * T + j] * (NB * NB);
  load_tile(As, Aij, NB);
  load_tile(Bs, Linv_all + (size_t)j * NB * NB, NB);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  tile_abt(As, Bs, acc, wave, lane);
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) Aij[(size_t)(16 * wave + (lane >> 4) + 4 * r) * NB + 16 * s + (lane & 15)] = acc[s][r];
}
// substitutions, one workgroup per tile column of the level, thread (t, q) = (row or column of the tile, quarter of the inner index):
//   forward   y_j = Li_j (b_j - sum_{k in row_cols(j)} L_jk y_k)        backward   x_j = Li_j^T (y_j - sum_{i in col_rows(j)} L_ij^T x_i)
// quarter sums are added in quarter order => deterministic
__global__ __launch_bounds__(kTPB) void tsc_subst(const double* tiles, const int* tid, int T, const int* cols, const int* l_off, const int* l_idx,
                                                  const double* Linv_all, double* b, int backward) {
  __shared__ double part[4][NB];
  __shared__ double v[NB];
  const int j = cols[blockIdx.x];
  const int t = threadIdx.x & 63, q = threadIdx.x >> 6;
  double s = 0;
  for (int e = l_off[j]; e < l_off[j + 1]; e++) {
    const int o = l_idx[e];
    const double* bo = b + (size_t)o * NB + 16 * q;
    if (!backward) {
      const double* Lrow = tiles + (size_t)tid[(size_t)j * T + o] * (NB * NB) + (size_t)t * NB + 16 * q;   // L_jk, row t
#pragma unroll
      for (int c = 0; c < 16; c++) s += Lrow[c] * bo[c];
    } else {
      const double* Lcol = tiles + (size_t)tid[(size_t)o * T + j] * (NB * NB) + (size_t)(16 * q) * NB + t;   // L_ij, column t
#pragma unroll
      for (int r = 0; r < 16; r++) s += Lcol[(size_t)r * NB] * bo[r];
    }
  }
  part[q][t] = s;
  __syncthreads();
  if (q == 0) v[t] = b[(size_t)j * NB + t] - (((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]);
  __syncthreads();
  const double* Li = Linv_all + (size_t)j * NB * NB;
  double z = 0;
  if (!backward) {
#pragma unroll
    for (int c = 0; c < 16; c++) { const int k = 16 * q + c; if (k <= t) z += Li[t * NB + k] * v[k]; }
  } else {
#pragma unroll
    for (int c = 0; c < 16; c++) { const int k = 16 * q + c; if (k >= t) z += Li[k * NB + t] * v[k]; }
  }
  part[q][t] = z;
  __syncthreads();
  if (q == 0) b[(size_t)j * NB + t] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
}

// tile row i = j + 1 + blockIdx.x:  L_ij = A_ij * Li_jj^T   (in place)
__global__ __launch_bounds__(kTPB) void chol_panel(double* A, int N, int j, const double* Linv_all, const int* rows /* nullptr: all rows below j */) {
  __shared__ double As[NB * LD], Bs[NB * LD];
  const int i = rows ? rows[blockIdx.x] : j + 1 + blockIdx.x;
  double* Aij = A + ((size_t)i * NB) * N + (size_t)j * NB;
  load_tile(As, Aij, N);
  load_tile(Bs, Linv_all + (size_t)j * NB * NB, NB);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  tile_abt(As, Bs, acc, wave, lane);
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) Aij[(size_t)(16 * wave + (lane >> 4) + 4 * r) * N + 16 * s + (lane & 15)] = acc[s][r];
}

// trailing update: pair index p -> (i, k) with i >= k > j;  A_ik -= L_ij L_kj^T
__global__ __launch_bounds__(kTPB) void chol_update(double* A, int N, int j, const int* pairs /* nullptr: every pair below j */) {
  __shared__ double As[NB * LD], Bs[NB * LD];
  const int p = blockIdx.x;
  int i, k;
  if (pairs) { i = pairs[p] >> 16; k = pairs[p] & 0xffff; }
  else {
    int ii = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while (ii * (ii + 1) / 2 > p) ii--;
    while ((ii + 1) * (ii + 2) / 2 <= p) ii++;
    const int kk = p - ii * (ii + 1) / 2;
    i = j + 1 + ii; k = j + 1 + kk;
  }
  load_tile(As, A + ((size_t)i * NB) * N + (size_t)j * NB, N);
  load_tile(Bs, A + ((size_t)k * NB) * N + (size_t)j * NB, N);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  tile_abt(As, Bs, acc, wave, lane);
  double* Aik = A + ((size_t)i * NB) * N + (size_t)k * NB;
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) Aik[(size_t)(16 * wave + (lane >> 4) + 4 * r) * N + 16 * s + (lane & 15)] -= acc[s][r];
}

// forward: y_j = Li_jj b_j ; backward: x_j = Li_jj^T b_j  (one workgroup, 64 outputs)
__global__ __launch_bounds__(NB) void chol_solve_diag(double* b, int j, const double* Linv_all, int transpose) {
  __shared__ double v[NB];
  const int t = threadIdx.x;
  const double* Li = Linv_all + (size_t)j * NB * NB;
  v[t] = b[(size_t)j * NB + t];
  __syncthreads();
  double s = 0;
  if (!transpose) { for (int k = 0; k <= t; k++) s += Li[t * NB + k] * v[k]; }
  else { for (int k = t; k < NB; k++) s += Li[k * NB + t] * v[k]; }
  b[(size_t)j * NB + t] = s;
}
// forward: b_i -= L_ij y_j (i > j) ; backward: b_k -= L_jk^T x_j (k < j);  one workgroup per tile
__global__ __launch_bounds__(NB) void chol_solve_update(const double* A, int N, double* b, int j, int transpose, const int* list /* nullptr: all */) {
  __shared__ double v[NB];
  const int t = threadIdx.x;
  v[t] = b[(size_t)j * NB + t];
  __syncthreads();
  double s = 0;
  if (!transpose) {
    const int i = list ? list[blockIdx.x] : j + 1 + blockIdx.x;
    const double* Lij = A + ((size_t)i * NB + t) * N + (size_t)j * NB;
    for (int k = 0; k < NB; k++) s += Lij[k] * v[k];
    b[(size_t)i * NB + t] -= s;
  } else {
    const int k = list ? list[blockIdx.x] : blockIdx.x;
    const double* Ljk = A + ((size_t)j * NB) * N + (size_t)k * NB + t;
    for (int r = 0; r < NB; r++) s += Ljk[(size_t)r * N] * v[r];
    b[(size_t)k * NB + t] -= s;
  }
}

// load a tile transposed: dst[c][r] = src[r][c]
__device__ __forceinline__ void load_tile_t(double* dst, const double* src, int ld_src) {
  double v[NB * NB / kTPB];
#pragma unroll
  for (int q = 0; q < NB * NB / kTPB; q++) { const int e = threadIdx.x + q * kTPB; v[q] = src[(size_t)(e / NB) * ld_src + e % NB]; }
#pragma unroll
  for (int q = 0; q < NB * NB / kTPB; q++) { const int e = threadIdx.x + q * kTPB; dst[(e % NB) * LD + e / NB] = v[q]; }
}

// D(64x16) = As(64x64) * Bs(16x64)^T: wave w owns rows 16w..16w+15
__device__ __forceinline__ void tile_abt16(const double* As, const double* Bs, v4d& acc, int wave, int lane) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll 4
  for (int kk = 0; kk < NB / 4; kk++) {
    const int k = 4 * kk + kq;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[(16 * wave + i) * LD + k], Bs[i * LD + k], acc, 0, 0, 0);
  }
}
// strip of 16 columns of a tile, transposed: dst[c][r] = src[r][c0 + c]
__device__ __forceinline__ void load_strip_t(double* dst, const double* src, int ld_src) {
  double v[NB * 16 / kTPB];
#pragma unroll
  for (int q = 0; q < NB * 16 / kTPB; q++) { const int e = threadIdx.x + q * kTPB; v[q] = src[(size_t)(e / 16) * ld_src + e % 16]; }
#pragma unroll
  for (int q = 0; q < NB * 16 / kTPB; q++) { const int e = threadIdx.x + q * kTPB; dst[(e % 16) * LD + e / 16] = v[q]; }
}

// X = L^-1 (lower triangular, tiles in X's lower triangle), X_jj = Li_jj, X_ij = -Li_ii * sum_{k=j}^{i-1} L_ik X_kj, one launch per tile ROW (round 5).  Until then one
// workgroup per column strip walked down its column (chol_tri_inverse): T - j - 1 links of i - j dependent 64 x 64 products each — 435 products in a row for the first
// column of a 30-tile operator (the 10 000-keyframe map: 717 us), 66 for 12 tiles (116 us; now 11 launches of 5.2 us + the diagonal copy = 63 us, and ~230 us for 30 tiles).
// Launch k holds every product that has row k of X as its right factor: workgroup (i, j, strip), i > k >= j, adds L_ik X_kj to the sum of X_ij (kept in X_ij's own
// storage; the first term, k = j, writes), and the workgroups of row i = k + 1 — whose sum is complete with this term — finish it, X_ij = -Li_ii * sum.  The terms enter
// every sum in the column walk's order through the same accumulator chain, so X has the same bits; the chain is T - 1 launches of one or two products.
__global__ __launch_bounds__(kTPB) void chol_tri_diag(int N, const double* Linv_all, double* X) {   // X_jj = Li_jj
  const int j = blockIdx.x;
  for (int e = threadIdx.x; e < NB * NB; e += kTPB) X[((size_t)j * NB + e / NB) * N + (size_t)j * NB + e % NB] = Linv_all[(size_t)j * NB * NB + e];
}
__global__ __launch_bounds__(kTPB) void chol_tri_step(const double* A, int N, const double* Linv_all, double* X, int k) {
  __shared__ double As[NB * LD], Bs[16 * LD];
  const int nj = k + 1;
  const int i = k + 1 + (int)blockIdx.x / nj, j = (int)blockIdx.x % nj, c0 = 16 * blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i16 = lane & 15, kq = lane >> 4;
  double* Xij = X + ((size_t)i * NB) * N + (size_t)j * NB + c0;
  v4d acc = {0, 0, 0, 0};
  if (j != k) {
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = Xij[(size_t)(16 * wave + kq + 4 * r) * N + i16];
  }
  load_tile(As, A + ((size_t)i * NB) * N + (size_t)k * NB, N);                        // L_ik
  load_strip_t(Bs, X + ((size_t)k * NB) * N + (size_t)j * NB + c0, N);                // strip of X_kj, transposed
  __syncthreads();
  tile_abt16(As, Bs, acc, wave, lane);
  __syncthreads();
  if (i == k + 1) {
    // Bs <- acc^T (so that Li_ii * acc = As * Bs^T), As <- Li_ii
#pragma unroll
    for (int r = 0; r < 4; r++) Bs[i16 * LD + 16 * wave + kq + 4 * r] = acc[r];
    load_tile(As, Linv_all + (size_t)i * NB * NB, NB);
    __syncthreads();
    v4d out = {0, 0, 0, 0};
    tile_abt16(As, Bs, out, wave, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) Xij[(size_t)(16 * wave + kq + 4 * r) * N + i16] = -out[r];
  } else {
#pragma unroll
    for (int r = 0; r < 4; r++) Xij[(size_t)(16 * wave + kq + 4 * r) * N + i16] = acc[r];
  }
}

// Ainv = X^T X: strip s (16 columns) of tile (p, q), p >= q: sum_{k >= p} X_kp^T X_kq ; mirrored into (q, p)
__global__ __launch_bounds__(kTPB) void chol_xtx(const double* X, int N, double* Ainv) {
  __shared__ double As[NB * LD], Bs[16 * LD];
  const int T = N / NB;
  const int pi = blockIdx.x, c0 = 16 * blockIdx.y;
  int p = (int)((sqrtf(8.0f * (float)pi + 1.0f) - 1.0f) * 0.5f);
  while (p * (p + 1) / 2 > pi) p--;
  while ((p + 1) * (p + 2) / 2 <= pi) p++;
  const int q = pi - p * (p + 1) / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i16 = lane & 15, kq = lane >> 4;
  v4d acc = {0, 0, 0, 0};
  for (int k = p; k < T; k++) {
    load_tile_t(As, X + ((size_t)k * NB) * N + (size_t)p * NB, N);            // X_kp^T
    load_strip_t(Bs, X + ((size_t)k * NB) * N + (size_t)q * NB + c0, N);      // strip of X_kq, transposed => As * Bs^T = X_kp^T X_kq[:, strip]
    __syncthreads();
    tile_abt16(As, Bs, acc, wave, lane);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = 16 * wave + kq + 4 * r, col = c0 + i16;
    Ainv[((size_t)p * NB + row) * N + (size_t)q * NB + col] = acc[r];
    if (p != q) Ainv[((size_t)q * NB + col) * N + (size_t)p * NB + row] = acc[r];
  }
}

}  // namespace

// Solves A x = b for a symmetric positive definite A.  d_A: N x N row-major with N = n rounded up to 64 (ccm_dense_padded)
// and the padding rows / columns already set to the identity; only the lower triangle is read; destroyed.  d_b: [N], in: rhs,
// out: x.  d_linv: scratch [N/64][64*64].  d_info: device int, set to (first non-positive pivot + 1) or left 0.
int ccm_dense_chol_solve_dev(ccm_ctx* ctx, double* d_A, int N, double* d_b, double* d_linv, int* d_info, const ccm_tile_plan* plan) {
  if (N % NB) return ccm_set_error(ctx, CCM_E_ARG, "dense cholesky: N must be a multiple of 64");
  const int T = N / NB;
  if (plan && (plan->T != T || T >= 32768)) return ccm_set_error(ctx, CCM_E_ARG, "dense cholesky: tile plan does not match");
  CCM_HIP_CHECK(ctx, hipMemsetAsync(d_info, 0, sizeof(int), ctx->stream));
  for (int j = 0; j < T; j++) {
    if (diag_blocked()) hipLaunchKernelGGL(chol_diag_wave_blk, dim3(1), dim3(128), 0, ctx->stream, d_A, N, j, d_linv, d_info);
    else hipLaunchKernelGGL(chol_diag_wave, dim3(1), dim3(128), 0, ctx->stream, d_A, N, j, d_linv, d_info);
    if (plan) {
      const int nr = plan->h_col_off[j + 1] - plan->h_col_off[j], np = plan->h_upd_off[j + 1] - plan->h_upd_off[j];
      if (nr) hipLaunchKernelGGL(chol_panel, dim3(nr), dim3(kTPB), 0, ctx->stream, d_A, N, j, (const double*)d_linv, plan->d_col_rows + plan->h_col_off[j]);
      if (np) hipLaunchKernelGGL(chol_update, dim3(np), dim3(kTPB), 0, ctx->stream, d_A, N, j, plan->d_upd_pairs + plan->h_upd_off[j]);
      continue;
    }
    const int rem = T - j - 1;
    if (rem > 0) {
      hipLaunchKernelGGL(chol_panel, dim3(rem), dim3(kTPB), 0, ctx->stream, d_A, N, j, (const double*)d_linv, (const int*)nullptr);
      hipLaunchKernelGGL(chol_update, dim3(rem * (rem + 1) / 2), dim3(kTPB), 0, ctx->stream, d_A, N, j, (const int*)nullptr);
    }
  }
  for (int j = 0; j < T; j++) {
    hipLaunchKernelGGL(chol_solve_diag, dim3(1), dim3(NB), 0, ctx->stream, d_b, j, (const double*)d_linv, 0);
    if (plan) {
      const int nr = plan->h_col_off[j + 1] - plan->h_col_off[j];
      if (nr) hipLaunchKernelGGL(chol_solve_update, dim3(nr), dim3(NB), 0, ctx->stream, (const double*)d_A, N, d_b, j, 0, plan->d_col_rows + plan->h_col_off[j]);
    } else if (T - j - 1 > 0) hipLaunchKernelGGL(chol_solve_update, dim3(T - j - 1), dim3(NB), 0, ctx->stream, (const double*)d_A, N, d_b, j, 0, (const int*)nullptr);
  }
  for (int j = T - 1; j >= 0; j--) {
    hipLaunchKernelGGL(chol_solve_diag, dim3(1), dim3(NB), 0, ctx->stream, d_b, j, (const double*)d_linv, 1);
    if (plan) {
      const int nc = plan->h_row_off[j + 1] - plan->h_row_off[j];
      if (nc) hipLaunchKernelGGL(chol_solve_update, dim3(nc), dim3(NB), 0, ctx->stream, (const double*)d_A, N, d_b, j, 1, plan->d_row_cols + plan->h_row_off[j]);
    } else if (j > 0) hipLaunchKernelGGL(chol_solve_update, dim3(j), dim3(NB), 0, ctx->stream, (const double*)d_A, N, d_b, j, 1, (const int*)nullptr);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

// Symbolic tile factorisation: eliminating tile column j connects every pair of rows below it.
void ccm_tile_plan_symbolic(int T, const std::vector<char>& nz_in, ccm_tile_plan* plan, std::vector<int>* col_rows, std::vector<int>* upd_pairs,
                            std::vector<int>* row_cols) {
  std::vector<char> nz = nz_in;
  plan->T = T;
  plan->h_col_off.assign(1, 0); plan->h_upd_off.assign(1, 0); plan->h_row_off.assign(1, 0);
  col_rows->clear(); upd_pairs->clear(); row_cols->clear();
  std::vector<int> rows;
  for (int j = 0; j < T; j++) {
    rows.clear();
    for (int i = j + 1; i < T; i++) if (nz[(size_t)i * T + j]) rows.push_back(i);
    for (int i : rows) col_rows->push_back(i);
    for (size_t a = 0; a < rows.size(); a++)
      for (size_t b = 0; b <= a; b++) { nz[(size_t)rows[a] * T + rows[b]] = 1; upd_pairs->push_back((rows[a] << 16) | rows[b]); }
    plan->h_col_off.push_back((int)col_rows->size());
    plan->h_upd_off.push_back((int)upd_pairs->size());
  }
  for (int j = 0; j < T; j++) {
    for (int k = 0; k < j; k++) if (nz[(size_t)j * T + k]) row_cols->push_back(k);
    plan->h_row_off.push_back((int)row_cols->size());
  }
}

// ---- tile-sparse Cholesky with level scheduling: plan, storage, solve ----
template <typename Tv>
static int tsc_up(ccm_ctx* ctx, ccm_tsc* p, const std::vector<Tv>& v, Tv** out) {
  void* d = nullptr; size_t actual = 0;
  if (int rc = ccm_pool_get(ctx, std::max<size_t>(v.size(), 1) * sizeof(Tv), &d, &actual)) return rc;
  p->blocks.push_back({d, actual});
  if (!v.empty()) CCM_HIP_CHECK(ctx, hipMemcpyAsync(d, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice, ctx->stream));
  *out = (Tv*)d;
  return CCM_OK;
}
int ccm_tsc_create(ccm_ctx* ctx, int T, const std::vector<char>& nz, ccm_tsc* p) {
  if (T <= 0 || T >= 32768) return ccm_set_error(ctx, CCM_E_ARG, "tile cholesky: bad tile count (tile coordinates are packed as (i << 16) | j in a signed int)");
  ccm_tile_plan sym;
  std::vector<int> col_rows, upd_pairs, row_cols;
  ccm_tile_plan_symbolic(T, nz, &sym, &col_rows, &upd_pairs, &row_cols);   // fill pattern: col_rows(j) = rows below, row_cols(j) = columns left
  p->T = T; p->owner = ctx;
  // elimination levels: a column can be eliminated when every column it has a tile in is done
  std::vector<int> level(T, 0);
  int nl = 0;
  for (int j = 0; j < T; j++) {
    int l = 0;
    for (int e = sym.h_row_off[j]; e < sym.h_row_off[j + 1]; e++) l = std::max(l, level[row_cols[e]] + 1);
    level[j] = l; nl = std::max(nl, l + 1);
  }
  p->n_levels = nl;
  p->tid.assign((size_t)T * T, -1);
  int nt = 0;
  for (int j = 0; j < T; j++) {
    p->tid[(size_t)j * T + j] = nt++;
    for (int e = sym.h_col_off[j]; e < sym.h_col_off[j + 1]; e++) p->tid[(size_t)col_rows[e] * T + j] = nt++;
  }
  p->n_tiles = nt;
  std::vector<int> cols, gt, gk_off(1, 0), gk, pt;
  p->lvl_col_off.assign(1, 0); p->lvl_gt_off.assign(1, 0); p->lvl_pt_off.assign(1, 0);
  for (int l = 0; l < nl; l++) {
    for (int j = 0; j < T; j++) {
      if (level[j] != l) continue;
      cols.push_back(j);
      const int* rj = row_cols.data() + sym.h_row_off[j]; const int nj = sym.h_row_off[j + 1] - sym.h_row_off[j];
      if (nj) { gt.push_back((j << 16) | j); for (int a = 0; a < nj; a++) gk.push_back(rj[a]); gk_off.push_back((int)gk.size()); }
      for (int e = sym.h_col_off[j]; e < sym.h_col_off[j + 1]; e++) {
        const int i = col_rows[e];
        pt.push_back((i << 16) | j);
        const int* ri = row_cols.data() + sym.h_row_off[i]; const int ni = sym.h_row_off[i + 1] - sym.h_row_off[i];
        const size_t before = gk.size();
        for (int a = 0, b = 0; a < ni && b < nj;) {   // columns k < j where both L_ik and L_jk are non-zero
          if (ri[a] == rj[b]) { gk.push_back(ri[a]); a++; b++; } else if (ri[a] < rj[b]) a++; else b++;
        }
        if (gk.size() > before) { gt.push_back((i << 16) | j); gk_off.push_back((int)gk.size()); }
      }
    }
    p->lvl_col_off.push_back((int)cols.size()); p->lvl_gt_off.push_back((int)gt.size()); p->lvl_pt_off.push_back((int)pt.size());
  }
  if (int rc = tsc_up(ctx, p, p->tid, &p->d_tid)) return rc;
  if (int rc = tsc_up(ctx, p, cols, &p->d_cols)) return rc;
  if (int rc = tsc_up(ctx, p, gt, &p->d_gt)) return rc;
  if (int rc = tsc_up(ctx, p, gk_off, &p->d_gk_off)) return rc;
  if (int rc = tsc_up(ctx, p, gk, &p->d_gk)) return rc;
  if (int rc = tsc_up(ctx, p, pt, &p->d_pt)) return rc;
  if (int rc = tsc_up(ctx, p, sym.h_row_off, &p->d_rc_off)) return rc;
  if (int rc = tsc_up(ctx, p, row_cols, &p->d_rc)) return rc;
  if (int rc = tsc_up(ctx, p, sym.h_col_off, &p->d_cr_off)) return rc;
  if (int rc = tsc_up(ctx, p, col_rows, &p->d_cr)) return rc;
  std::vector<double> none;
  void* d = nullptr; size_t actual = 0;
  if (int rc = ccm_pool_get(ctx, (size_t)nt * NB * NB * sizeof(double), &d, &actual)) return rc;
  p->blocks.push_back({d, actual}); p->d_tiles = (double*)d;
  if (int rc = ccm_pool_get(ctx, (size_t)T * NB * NB * sizeof(double), &d, &actual)) return rc;
  p->blocks.push_back({d, actual}); p->d_linv = (double*)d;
  if (int rc = ccm_pool_get(ctx, 256, &d, &actual)) return rc;
  p->blocks.push_back({d, actual}); p->d_info = (int*)d;
  return CCM_OK;
}
void ccm_tsc_destroy(ccm_ctx* ctx, ccm_tsc* p) {
  for (auto& b : p->blocks) ccm_pool_put(ctx, b.first, b.second);
  p->blocks.clear();
}
ccm_tsc::~ccm_tsc() { if (owner) ccm_tsc_destroy(owner, this); }
int ccm_tsc_clear(ccm_ctx* ctx, ccm_tsc* p) {
  CCM_HIP_CHECK(ctx, hipMemsetAsync(p->d_tiles, 0, (size_t)p->n_tiles * NB * NB * sizeof(double), ctx->stream));
  return CCM_OK;
}
// factorises the tiles in place and solves for d_b ([64 T], in: rhs, out: x); *d_info = first non-positive pivot + 1 or 0
int ccm_tsc_solve(ccm_ctx* ctx, ccm_tsc* p, double* d_b) {
  const int T = p->T;
  CCM_HIP_CHECK(ctx, hipMemsetAsync(p->d_info, 0, sizeof(int), ctx->stream));
  for (int l = 0; l < p->n_levels; l++) {
    const int c0 = p->lvl_col_off[l], nc = p->lvl_col_off[l + 1] - c0;
    const int g0 = p->lvl_gt_off[l], ng = p->lvl_gt_off[l + 1] - g0;
    const int p0 = p->lvl_pt_off[l], np = p->lvl_pt_off[l + 1] - p0;
    if (ng) hipLaunchKernelGGL(tsc_gather, dim3(ng), dim3(kTPB), 0, ctx->stream, p->d_tiles, (const int*)p->d_tid, T, (const int*)p->d_gt + g0, (const int*)p->d_gk_off + g0, (const int*)p->d_gk);
    if (diag_blocked()) hipLaunchKernelGGL(tsc_diag<true>, dim3(nc), dim3(128), 0, ctx->stream, p->d_tiles, (const int*)p->d_tid, T, (const int*)p->d_cols + c0, p->d_linv, p->d_info);
    else hipLaunchKernelGGL(tsc_diag<false>, dim3(nc), dim3(128), 0, ctx->stream, p->d_tiles, (const int*)p->d_tid, T, (const int*)p->d_cols + c0, p->d_linv, p->d_info);
    if (np) hipLaunchKernelGGL(tsc_panel, dim3(np), dim3(kTPB), 0, ctx->stream, p->d_tiles, (const int*)p->d_tid, T, (const int*)p->d_pt + p0, (const double*)p->d_linv);
  }
  for (int l = 0; l < p->n_levels; l++) {
    const int c0 = p->lvl_col_off[l], nc = p->lvl_col_off[l + 1] - c0;
    hipLaunchKernelGGL(tsc_subst, dim3(nc), dim3(kTPB), 0, ctx->stream, (const double*)p->d_tiles, (const int*)p->d_tid, T, (const int*)p->d_cols + c0,
                       (const int*)p->d_rc_off, (const int*)p->d_rc, (const double*)p->d_linv, d_b, 0);
  }
  for (int l = p->n_levels - 1; l >= 0; l--) {
    const int c0 = p->lvl_col_off[l], nc = p->lvl_col_off[l + 1] - c0;
    hipLaunchKernelGGL(tsc_subst, dim3(nc), dim3(kTPB), 0, ctx->stream, (const double*)p->d_tiles, (const int*)p->d_tid, T, (const int*)p->d_cols + c0,
                       (const int*)p->d_cr_off, (const int*)p->d_cr, (const double*)p->d_linv, d_b, 1);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

// Test hook: host matrix (n x n row-major, symmetric positive definite) and rhs in, solution out.
int ccm_internal::debug_dense_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info) {
  if (!ctx || !A || !b || !x || !info || n <= 0) return ccm_set_error(ctx, CCM_E_ARG, "ccm_debug_dense_solve: bad args");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int N = ((n + NB - 1) / NB) * NB;
  std::vector<double> Ap((size_t)N * N, 0.0), bp(N, 0.0);
  for (int i = 0; i < N; i++) {
    if (i < n) { for (int c = 0; c < n; c++) Ap[(size_t)i * N + c] = A[(size_t)i * n + c]; bp[i] = b[i]; }
    else Ap[(size_t)i * N + i] = 1.0;
  }
  double *dA = nullptr, *db = nullptr, *dl = nullptr; int* di = nullptr;
  CCM_HIP_CHECK(ctx, hipMalloc(&dA, Ap.size() * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&db, N * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&dl, (size_t)N * NB * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&di, sizeof(int)));
  hipMemcpyAsync(dA, Ap.data(), Ap.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  hipMemcpyAsync(db, bp.data(), N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  int rc = ccm_dense_chol_solve_dev(ctx, dA, N, db, dl, di);
  if (rc == CCM_OK) {
    hipMemcpyAsync(bp.data(), db, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(info, di, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, "ccm_debug_dense_solve: sync");
    for (int i = 0; i < n; i++) x[i] = bp[i];
  }
  hipFree(dA); hipFree(db); hipFree(dl); hipFree(di);
  return rc;
}

// Test hook for the tile-sparse, level-scheduled form: the tile pattern is taken from the non-zeros of the host matrix (its fill is added
// symbolically), the matrix is scattered into the compact tile storage, factored and solved.  *levels / *tiles report the plan.
int ccm_internal::debug_tile_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info, int* levels, int* tiles) {
  if (!ctx || !A || !b || !x || !info || n <= 0) return ccm_set_error(ctx, CCM_E_ARG, "ccm_debug_tile_solve: bad args");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int N = ((n + NB - 1) / NB) * NB, T = N / NB;
  std::vector<char> nz((size_t)T * T, 0);
  for (int i = 0; i < n; i++) for (int c = 0; c <= i; c++) if (A[(size_t)i * n + c] != 0.0) nz[(size_t)(i / NB) * T + c / NB] = 1;
  for (int t = 0; t < T; t++) nz[(size_t)t * T + t] = 1;
  ccm_tsc p;
  if (int rc = ccm_tsc_create(ctx, T, nz, &p)) return rc;
  std::vector<double> ht((size_t)p.n_tiles * NB * NB, 0.0), hb(N, 0.0);
  for (int i = 0; i < N; i++) {
    if (i >= n) { ht[(size_t)p.tid[(size_t)(i / NB) * T + i / NB] * NB * NB + (size_t)(i % NB) * (NB + 1)] = 1.0; continue; }
    hb[i] = b[i];
    for (int c = 0; c <= i; c++) {
      const double v = A[(size_t)i * n + c];
      if (v != 0.0) ht[(size_t)p.tid[(size_t)(i / NB) * T + c / NB] * NB * NB + (size_t)(i % NB) * NB + c % NB] = v;
    }
  }
  double* db = nullptr;
  CCM_HIP_CHECK(ctx, hipMalloc(&db, N * sizeof(double)));
  hipMemcpyAsync(p.d_tiles, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  hipMemcpyAsync(db, hb.data(), N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  int rc = ccm_tsc_solve(ctx, &p, db);
  if (rc == CCM_OK) {
    hipMemcpyAsync(hb.data(), db, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(info, p.d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, "ccm_debug_tile_solve: sync");
    for (int i = 0; i < n; i++) x[i] = hb[i];
  }
  if (levels) *levels = p.n_levels;
  if (tiles) *tiles = p.n_tiles;
  hipStreamSynchronize(ctx->stream);
  hipFree(db);
  return rc;
}

// Explicit inverse of a symmetric positive definite matrix (same layout contract as ccm_dense_chol_solve_dev; d_A is
// destroyed, d_X is N x N scratch whose lower tiles receive L^-1, d_Ainv receives the full symmetric inverse).
int ccm_dense_chol_inverse_dev(ccm_ctx* ctx, double* d_A, int N, double* d_linv, double* d_X, double* d_Ainv, int* d_info) {
  if (N % NB) return ccm_set_error(ctx, CCM_E_ARG, "dense cholesky: N must be a multiple of 64");
  const int T = N / NB;
  CCM_HIP_CHECK(ctx, hipMemsetAsync(d_info, 0, sizeof(int), ctx->stream));
  for (int j = 0; j < T; j++) {
    if (diag_blocked()) hipLaunchKernelGGL(chol_diag_wave_blk, dim3(1), dim3(128), 0, ctx->stream, d_A, N, j, d_linv, d_info);
    else hipLaunchKernelGGL(chol_diag_wave, dim3(1), dim3(128), 0, ctx->stream, d_A, N, j, d_linv, d_info);
    const int rem = T - j - 1;
    if (rem > 0) {
      hipLaunchKernelGGL(chol_panel, dim3(rem), dim3(kTPB), 0, ctx->stream, d_A, N, j, (const double*)d_linv, (const int*)nullptr);
      hipLaunchKernelGGL(chol_update, dim3(rem * (rem + 1) / 2), dim3(kTPB), 0, ctx->stream, d_A, N, j, (const int*)nullptr);
    }
  }
  hipLaunchKernelGGL(chol_tri_diag, dim3(T), dim3(kTPB), 0, ctx->stream, N, (const double*)d_linv, d_X);
  for (int k = 0; k + 1 < T; k++)
    hipLaunchKernelGGL(chol_tri_step, dim3((T - k - 1) * (k + 1), NB / 16), dim3(kTPB), 0, ctx->stream, (const double*)d_A, N, (const double*)d_linv, d_X, k);
  hipLaunchKernelGGL(chol_xtx, dim3(T * (T + 1) / 2, NB / 16), dim3(kTPB), 0, ctx->stream, (const double*)d_X, N, d_Ainv);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

// Test hook: inverse of a host SPD matrix (n x n row-major) through the tile kernels above.
int ccm_internal::debug_dense_inverse(ccm_ctx* ctx, const double* A, int n, double* Ainv, int* info) {
  if (!ctx || !A || !Ainv || !info || n <= 0) return ccm_set_error(ctx, CCM_E_ARG, "ccm_debug_dense_inverse: bad args");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int N = ((n + NB - 1) / NB) * NB;
  std::vector<double> Ap((size_t)N * N, 0.0);
  for (int i = 0; i < N; i++) {
    if (i < n) { for (int c = 0; c < n; c++) Ap[(size_t)i * N + c] = A[(size_t)i * n + c]; }
    else Ap[(size_t)i * N + i] = 1.0;
  }
  double *dA = nullptr, *dl = nullptr, *dX = nullptr, *dI = nullptr; int* di = nullptr;
  CCM_HIP_CHECK(ctx, hipMalloc(&dA, Ap.size() * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&dX, Ap.size() * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&dI, Ap.size() * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&dl, (size_t)N * NB * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMalloc(&di, sizeof(int)));
  hipMemcpyAsync(dA, Ap.data(), Ap.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  hipMemsetAsync(dX, 0, Ap.size() * sizeof(double), ctx->stream);
  int rc = ccm_dense_chol_inverse_dev(ctx, dA, N, dl, dX, dI, di);
  if (rc == CCM_OK) {
    hipMemcpyAsync(Ap.data(), dI, Ap.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(info, di, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, "ccm_debug_dense_inverse: sync");
    for (int i = 0; i < n; i++) for (int c = 0; c < n; c++) Ainv[(size_t)i * n + c] = Ap[(size_t)i * N + c];
  }
  hipFree(dA); hipFree(dX); hipFree(dI); hipFree(dl); hipFree(di);
  return rc;
}
