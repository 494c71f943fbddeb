// hamming.hip — 256-bit Hamming distance kernels (gfx950, wave64).
//
// Reference: ORBmatcher::DescriptorDistance (cslam/src/ORBmatcher.cpp:1653-1669) evaluated
// inside the best / second-best scans of the Search* methods (e.g. :102-134, :220-245,
// :1408-1433).  Integer work, bit-exact.
//
// Design notes (MI355X):
//  * dense Q x T: one lane owns one query (8 dwords in VGPRs).  The target row index is
//    wave-uniform, so target words arrive through the scalar cache (s_load_dwordx8) and the
//    inner step is 8 x (v_xor_b32 with an SGPR operand + v_bcnt_u32_b32 accumulate) — no LDS
//    traffic at all.  T is split across blockIdx.y so that small problems still put >= 1
//    wave on every SIMD; partial (best,second) pairs are merged in split order, which keeps
//    the reference's "first minimum wins" tie rule (lowest target index).
//  * CSR (windowed) search: one wave per query, lanes over that query's candidate list; the
//    query words are wave-uniform (scalar loads), each lane gathers one 32-byte target row
//    (2 x global_load_dwordx4).  Per-slot distances are written for the host-side ordered
//    resolution pass; best/second come from two wave min-reductions over (dist,slot) keys.
#include "common.h"

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ int ham256(const uint32_t (&a)[8], const uint32_t* __restrict__ b) {
  int d = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]);
  return d;
}

// partial results layout: [split][Q]
__global__ __launch_bounds__(256) void hamming_dense_partial(
    const uint32_t* __restrict__ q, int Q, const uint32_t* __restrict__ t, int T, int t_per_split,
    int32_t* __restrict__ p_best_idx, int32_t* __restrict__ p_best, int32_t* __restrict__ p_second) {
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  const int split = blockIdx.y;
  const int t0 = split * t_per_split;
  const int t1 = min(T, t0 + t_per_split);
  uint32_t qa[8];
  if (qi < Q) {
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
    uint4 lo = qp[0], hi = qp[1];
    qa[0] = lo.x; qa[1] = lo.y; qa[2] = lo.z; qa[3] = lo.w;
    qa[4] = hi.x; qa[5] = hi.y; qa[6] = hi.z; qa[7] = hi.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) qa[k] = 0;
  }
  int best = 256, second = 256, best_idx = -1;
  for (int ti = t0; ti < t1; ++ti) {       // ti is wave-uniform -> scalar loads of the target row
    const uint32_t* tp = t + (size_t)ti * 8;
    const int d = ham256(qa, tp);
    if (d < best) { second = best; best = d; best_idx = ti; }
    else if (d < second) { second = d; }
  }
  if (qi < Q) {
    const size_t o = (size_t)split * Q + qi;
    p_best_idx[o] = best_idx; p_best[o] = best; p_second[o] = second;
  }
}

// (round 5: the one-launch form of this search — hamming_dense_fused, the last workgroup of a query block to take a ticket merging the splits — was built and measured in
// round 4 (29.1 us against 15.9 for partial + merge at 2000 x 2000: the device-scope fence before the ticket costs more than the launch it saves) and is gone from the library.)

__global__ __launch_bounds__(256) void hamming_dense_merge(
    int Q, int n_split, const int32_t* __restrict__ p_best_idx, const int32_t* __restrict__ p_best,
    const int32_t* __restrict__ p_second, int32_t* __restrict__ best_idx, int32_t* __restrict__ best,
    int32_t* __restrict__ second) {
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= Q) return;
  int b = 256, s = 256, bi = -1;
  // eight splits' partial results are requested together (one memory round trip per batch instead of one per split: 13.7 -> ~4 us at 2000 x 2000,
  // where the 32 splits were 32 dependent round trips), then combined in ascending target order: strict '<' keeps the first minimum
  for (int sp0 = 0; sp0 < n_split; sp0 += 8) {
    int pb[8], ps[8], pi[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const bool v = sp0 + u < n_split;
      const size_t o = (size_t)(v ? sp0 + u : 0) * Q + qi;
      pb[u] = v ? p_best[o] : 256; ps[u] = v ? p_second[o] : 256; pi[u] = v ? p_best_idx[o] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (pb[u] < b) { s = min(b, ps[u]); b = pb[u]; bi = pi[u]; }
      else           { s = min(s, pb[u]); }
    }
  }
  best_idx[qi] = bi; best[qi] = b; second[qi] = s;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, kWave));
  return v;
}

// one wave per query; blockDim = 256 -> 4 queries per workgroup
__global__ __launch_bounds__(256) void hamming_csr_kernel(
    const uint32_t* __restrict__ q, int Q, const uint32_t* __restrict__ t,
    const int32_t* __restrict__ cand_off, const int32_t* __restrict__ cand_idx,
    uint16_t* __restrict__ cand_dist, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
    int32_t* __restrict__ second_dist, long long n_cand) {
  const int lane = threadIdx.x & (kWave - 1);
  const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave));
  if (qi >= Q) return;
  const int c0 = cand_off[qi], c1 = cand_off[qi + 1];
  if ((long long)c1 > n_cand) return;   // list lies (partly) beyond the buffers the caller sized: never touch it
  uint32_t qa[8];
  {
    const uint32_t* qp = q + (size_t)qi * 8;   // wave-uniform address
#pragma unroll
    for (int k = 0; k < 8; ++k) qa[k] = qp[k];
  }
  // running (best, second) as keys (dist << 20 | slot-in-list); slot < 2^20 is checked on the host
  uint32_t kbest = 0xFFFFFFFFu, ksecond = 0xFFFFFFFFu;
  for (int base = c0; base < c1; base += kWave) {
    const int s = base + lane;
    uint32_t key = 0xFFFFFFFFu;
    if (s < c1) {
      const int ti = cand_idx[s];
      const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)ti * 8);
      const uint4 lo = tp[0], hi = tp[1];
      int d = __builtin_popcount(qa[0] ^ lo.x) + __builtin_popcount(qa[1] ^ lo.y) +
              __builtin_popcount(qa[2] ^ lo.z) + __builtin_popcount(qa[3] ^ lo.w) +
              __builtin_popcount(qa[4] ^ hi.x) + __builtin_popcount(qa[5] ^ hi.y) +
              __builtin_popcount(qa[6] ^ hi.z) + __builtin_popcount(qa[7] ^ hi.w);
      if (cand_dist) cand_dist[s] = (uint16_t)d;
      key = ((uint32_t)d << 20) | (uint32_t)(s - c0);
    }
    if (best_idx) {
      const uint32_t m1 = wave_min_u32(key);
      const uint32_t key2 = (key == m1) ? 0xFFFFFFFFu : key;   // keys are unique (slot bits)
      const uint32_t m2 = wave_min_u32(key2);
      // merge chunk (m1,m2) after running (kbest,ksecond): smaller key = smaller dist, then earlier slot
      if (m1 < kbest) { ksecond = min(kbest, m2); kbest = m1; }
      else            { ksecond = min(ksecond, m1); }
    }
  }
  if (best_idx && lane == 0) {
    if (kbest == 0xFFFFFFFFu) { best_idx[qi] = -1; best_dist[qi] = 256; second_dist[qi] = 256; }
    else {
      best_idx[qi] = cand_idx[c0 + (int)(kbest & 0xFFFFFu)];
      best_dist[qi] = (int)(kbest >> 20);
      second_dist[qi] = (ksecond == 0xFFFFFFFFu) ? 256 : (int)(ksecond >> 20);
    }
  }
}


// (round 5) The same search with G lanes per query (G = 8, 16, 32; 64 / G queries per wave) and an optional per-query TARGET BASE.  The reference's window and BoW-bucket
// searches have short lists — ~20 candidates for a SearchForTriangulation query, ~30 for a projected window —, so a wave per query leaves more than half of its
// lanes idle and every query pays a whole wave's launch slot; and Mapping.cpp:335 / :503 runs up to 20 such searches per new keyframe, each against ANOTHER
// keyframe's descriptors.  ccm_hamming_csr_multi sends all of them out as one launch: the queries of all searches back to back, q_tbase[q] = first row of the
// target set query q searches in (its candidate indices are local to that set).  Results per slot are the same integers whatever G is: the distance of a slot does
// not depend on its neighbours, and best / second are the minimum and the second minimum of the keys (distance << 20 | slot), which no grouping changes.
template <int G>
__global__ __launch_bounds__(256) void hamming_csr_group_kernel(
    const uint32_t* __restrict__ q, int Q, const uint32_t* __restrict__ t, const int32_t* __restrict__ q_tbase /* nullable */,
    const int32_t* __restrict__ cand_off, const int32_t* __restrict__ cand_idx,
    uint16_t* __restrict__ cand_dist, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
    int32_t* __restrict__ second_dist, long long n_cand) {
  const int sub = threadIdx.x & (G - 1);
  const int qi = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) / G);
  const bool live = qi < Q;
  int c0 = 0, c1 = 0;
  if (live) { c0 = cand_off[qi]; c1 = cand_off[qi + 1]; if ((long long)c1 > n_cand) c1 = c0; }   // a list beyond the buffers the caller sized is never touched
  uint4 qlo = make_uint4(0, 0, 0, 0), qhi = qlo;
  size_t tb = 0;
  if (live) {
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi * 8);
    qlo = qp[0]; qhi = qp[1];
    if (q_tbase) tb = (size_t)q_tbase[qi] * 8;
  }
  uint32_t kbest = 0xFFFFFFFFu, ksecond = 0xFFFFFFFFu;
  // every group of the wave runs as many rounds as the longest list among them (the shuffles below need all lanes)
  int rounds = (c1 - c0 + G - 1) / G;
#pragma unroll
  for (int off = G; off < kWave; off <<= 1) rounds = max(rounds, __shfl_xor(rounds, off, kWave));
  for (int r = 0; r < rounds; r++) {
    const int s = c0 + r * G + sub;
    uint32_t key = 0xFFFFFFFFu;
    if (s < c1) {
      const int ti = cand_idx[s];
      const uint4* tp = reinterpret_cast<const uint4*>(t + tb + (size_t)ti * 8);
      const uint4 lo = tp[0], hi = tp[1];
      const int d = __builtin_popcount(qlo.x ^ lo.x) + __builtin_popcount(qlo.y ^ lo.y) + __builtin_popcount(qlo.z ^ lo.z) + __builtin_popcount(qlo.w ^ lo.w) +
                    __builtin_popcount(qhi.x ^ hi.x) + __builtin_popcount(qhi.y ^ hi.y) + __builtin_popcount(qhi.z ^ hi.z) + __builtin_popcount(qhi.w ^ hi.w);
      if (cand_dist) cand_dist[s] = (uint16_t)d;
      key = ((uint32_t)d << 20) | (uint32_t)(s - c0);
    }
    if (best_idx) {
      uint32_t m1 = key;
#pragma unroll
      for (int off = G / 2; off >= 1; off >>= 1) m1 = min(m1, (uint32_t)__shfl_xor((int)m1, off, kWave));
      uint32_t m2 = (key == m1) ? 0xFFFFFFFFu : key;   // keys are unique (slot bits)
#pragma unroll
      for (int off = G / 2; off >= 1; off >>= 1) m2 = min(m2, (uint32_t)__shfl_xor((int)m2, off, kWave));
      if (m1 < kbest) { ksecond = min(kbest, m2); kbest = m1; }
      else            { ksecond = min(ksecond, m1); }
    }
  }
  if (best_idx && live && sub == 0) {
    if (kbest == 0xFFFFFFFFu) { best_idx[qi] = -1; best_dist[qi] = 256; second_dist[qi] = 256; }
    else {
      best_idx[qi] = cand_idx[c0 + (int)(kbest & 0xFFFFFu)];
      best_dist[qi] = (int)(kbest >> 20);
      second_dist[qi] = (ksecond == 0xFFFFFFFFu) ? 256 : (int)(ksecond >> 20);
    }
  }
}

// group width for a mean list length: the smallest power of two >= mean / 2 lanes (two rounds for an average list), 8 .. 64
static int csr_group_width(long long n_cand, int Q) {
  const double mean = Q > 0 ? (double)n_cand / (double)Q : 0.0;
  int g = 8;
  while (g < 64 && 2 * g < mean) g <<= 1;
  return g;
}

// ---- MapPoint::ComputeDistinctiveDescriptors (cslam/src/MapPoint.cpp:929-994), batched ----------------------
// one wave per map point: lane i owns observation i (rows i, i+64, ... when a point has more than 64), computes
// its N distances into a lane-private LDS row, selects the median (element of rank (int)(0.5*(N-1)) of the sorted
// row, self-distance 0 included) by rank counting, then the wave picks the FIRST observation with the least median.
constexpr int kDistinctMaxN = 256;
__global__ __launch_bounds__(64) void distinctive_kernel(const uint32_t* __restrict__ desc, const int32_t* __restrict__ off, int P,
                                                         int32_t* __restrict__ best_out) {
  extern __shared__ uint16_t rows[];   // [64][kDistinctMaxN]
  const int lane = threadIdx.x;
  const int p = blockIdx.x;
  if (p >= P) return;
  const int o0 = off[p], N = off[p + 1] - o0;
  if (N <= 0) { if (lane == 0) best_out[p] = -1; return; }
  const int kth = (int)(0.5 * (N - 1));
  uint32_t bestKey = 0xFFFFFFFFu;   // (median << 16) | index : min = least median, then first index
  uint16_t* my = rows + lane * kDistinctMaxN;
  for (int base = 0; base < N; base += 64) {
    const int i = base + lane;
    uint32_t key = 0xFFFFFFFFu;
    if (i < N) {
      uint32_t a[8];
      const uint4* ap = reinterpret_cast<const uint4*>(desc + (size_t)(o0 + i) * 8);
      const uint4 lo = ap[0], hi = ap[1];
      a[0] = lo.x; a[1] = lo.y; a[2] = lo.z; a[3] = lo.w; a[4] = hi.x; a[5] = hi.y; a[6] = hi.z; a[7] = hi.w;
      for (int j = 0; j < N; j++) {   // j is wave-uniform: the j-th descriptor arrives through scalar loads
        const uint32_t* bp = desc + (size_t)(o0 + j) * 8;
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) d += __builtin_popcount(a[k] ^ bp[k]);
        my[j] = (uint16_t)d;
      }
      int median = 0;
      for (int j = 0; j < N; j++) {
        const int dj = my[j];
        int rank = 0;
        for (int l = 0; l < N; l++) { const int dl = my[l]; rank += (dl < dj) || (dl == dj && l < j); }
        if (rank == kth) { median = dj; break; }
      }
      key = ((uint32_t)median << 16) | (uint32_t)i;
    }
    key = wave_min_u32(key);
    bestKey = min(bestKey, key);
  }
  if (lane == 0) best_out[p] = (int)(bestKey & 0xFFFFu);
}


// ---- DBoW2 vocabulary tree descent (TemplatedVocabulary::transform, DBoW2/TemplatedVocabulary.h:1218-1260) -----
// one wave per feature: at every level the lanes take one child each (k <= 64), the wave picks the child with the
// least FORB::distance, FIRST minimum winning (strict '<' in child order); records the node at level nid_level.
__global__ __launch_bounds__(256) void bow_descend_kernel(const uint32_t* __restrict__ desc, int N, const int32_t* __restrict__ child_off,
                                                          const int32_t* __restrict__ child_id, const uint32_t* __restrict__ node_desc,
                                                          int nid_level, int32_t* __restrict__ leaf_out, int32_t* __restrict__ nid_out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int f = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
  if (f >= N) return;
  uint32_t a[8];
  {
    const uint32_t* ap = desc + (size_t)f * 8;   // wave-uniform
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = ap[k];
  }
  int node = 0, level = 0, nid = 0;
  while (true) {
    const int c0 = child_off[node], c1 = child_off[node + 1];
    if (c1 <= c0) break;   // leaf
    ++level;
    uint32_t best = 0xFFFFFFFFu;
    for (int base = c0; base < c1; base += kWave) {
      uint32_t key = 0xFFFFFFFFu;
      if (base + lane < c1) {
        const int id = child_id[base + lane];
        const uint4* bp = reinterpret_cast<const uint4*>(node_desc + (size_t)id * 8);
        const uint4 lo = bp[0], hi = bp[1];
        const int d = __builtin_popcount(a[0] ^ lo.x) + __builtin_popcount(a[1] ^ lo.y) + __builtin_popcount(a[2] ^ lo.z) + __builtin_popcount(a[3] ^ lo.w) +
                      __builtin_popcount(a[4] ^ hi.x) + __builtin_popcount(a[5] ^ hi.y) + __builtin_popcount(a[6] ^ hi.z) + __builtin_popcount(a[7] ^ hi.w);
        key = ((uint32_t)d << 20) | (uint32_t)(base + lane - c0);
      }
      best = min(best, wave_min_u32(key));
    }
    node = child_id[c0 + (int)(best & 0xFFFFFu)];
    if (level == nid_level) nid = node;
  }
  if (lane == 0) { leaf_out[f] = node; nid_out[f] = nid; }
}

}  // namespace

extern "C" int ccm_hamming_dense_best2_dev(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, int T,
                                           int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist) {
  if (!ctx || Q < 0 || T < 0) return ccm_set_error(ctx, CCM_E_ARG, "hamming_dense: bad args");
  if (Q == 0) return CCM_OK;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int qblocks = ccm_div_up(Q, 256);
  // choose the T split so that the grid has >= ~2048 waves when the problem allows it
  int n_split = 1;
  if (T > 0) {
    const int waves_q = ccm_div_up(Q, 64);
    n_split = std::max(1, std::min(ccm_div_up(T, 64), ccm_div_up(2048, waves_q)));
  }
  const int t_per_split = T > 0 ? ccm_div_up(T, n_split) : 0;
  if (T > 0) n_split = ccm_div_up(T, t_per_split);
  void* scratch = nullptr;
  const size_t part = (size_t)n_split * Q * sizeof(int32_t);
  int rc = ccm_scratch(ctx, 3 * part, &scratch);
  if (rc) return rc;
  int32_t* p_idx = (int32_t*)scratch;
  int32_t* p_best = p_idx + (size_t)n_split * Q;
  int32_t* p_second = p_best + (size_t)n_split * Q;
  {
    ccm_prof_scope ps(ctx, CCM_K_HAMMING_DENSE);
    hipLaunchKernelGGL(hamming_dense_partial, dim3(qblocks, n_split), dim3(256), 0, ctx->stream,
                       (const uint32_t*)d_q, Q, (const uint32_t*)d_t, T, t_per_split, p_idx, p_best, p_second);
    hipLaunchKernelGGL(hamming_dense_merge, dim3(qblocks), dim3(256), 0, ctx->stream, Q, n_split, p_idx, p_best,
                       p_second, d_best_idx, d_best_dist, d_second_dist);
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

extern "C" int ccm_hamming_dense_best2(ccm_ctx* ctx, const uint8_t* q, int Q, const uint8_t* t, int T,
                                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
  if (!ctx || Q < 0 || T < 0 || (Q && (!q || !best_idx || !best_dist || !second_dist)) || (T && !t))
    return ccm_set_error(ctx, CCM_E_ARG, "hamming_dense: bad args");
  if (Q == 0) return CCM_OK;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bq = ccm_align256((size_t)Q * 32), bt = ccm_align256((size_t)std::max(T, 1) * 32);
  void* io = nullptr;
  { int rc0 = ccm_io_scratch(ctx, bq + bt + (size_t)Q * 3 * sizeof(int32_t), &io); if (rc0) return rc0; }
  uint8_t* d_q = (uint8_t*)io; uint8_t* d_t = d_q + bq; int32_t* d_out = (int32_t*)(d_t + bt);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_q, q, (size_t)Q * 32, hipMemcpyHostToDevice, ctx->stream));
  if (T) CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_t, t, (size_t)T * 32, hipMemcpyHostToDevice, ctx->stream));
  int rc = ccm_hamming_dense_best2_dev(ctx, d_q, Q, d_t, T, d_out, d_out + Q, d_out + 2 * (size_t)Q);
  if (rc == CCM_OK) {
    hipMemcpyAsync(best_idx, d_out, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(best_dist, d_out + Q, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(second_dist, d_out + 2 * (size_t)Q, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, std::string("hamming_dense: ") + hipGetErrorString(e));
  }
  return rc;
}

// one launch of the windowed search: a wave per query for long lists (the round-1 kernel), G lanes per query below
static int csr_launch(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, const int32_t* d_q_tbase, const int32_t* d_cand_off, const int32_t* d_cand_idx,
                      int64_t n_cand, uint16_t* d_cand_dist, int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist) {
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int G = csr_group_width((long long)n_cand, Q);
  {
    ccm_prof_scope ps(ctx, CCM_K_HAMMING_CSR);
#define CCM_CSR_G(g) hipLaunchKernelGGL(hamming_csr_group_kernel<g>, dim3(ccm_div_up((int64_t)Q * g, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_q, Q, \
                                        (const uint32_t*)d_t, d_q_tbase, d_cand_off, d_cand_idx, d_cand_dist, d_best_idx, d_best_dist, d_second_dist, (long long)n_cand)
    if (G >= 64 && !d_q_tbase)
      hipLaunchKernelGGL(hamming_csr_kernel, dim3(ccm_div_up(Q, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)d_q, Q,
                         (const uint32_t*)d_t, d_cand_off, d_cand_idx, d_cand_dist, d_best_idx, d_best_dist, d_second_dist, (long long)n_cand);
    else if (G >= 64) CCM_CSR_G(64);
    else if (G == 32) CCM_CSR_G(32);
    else if (G == 16) CCM_CSR_G(16);
    else CCM_CSR_G(8);
#undef CCM_CSR_G
  }
  CCM_HIP_CHECK(ctx, hipGetLastError());
  return CCM_OK;
}

extern "C" int ccm_hamming_csr_dev(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, int T,
                                   const int32_t* d_cand_off, const int32_t* d_cand_idx, int64_t n_cand,
                                   uint16_t* d_cand_dist, int32_t* d_best_idx, int32_t* d_best_dist,
                                   int32_t* d_second_dist) {
  (void)T;
  if (!ctx || Q < 0) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr: bad args");
  if (d_best_idx && (!d_best_dist || !d_second_dist)) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr: best2 outputs must be all set or all NULL");
  if (Q == 0) return CCM_OK;
  return csr_launch(ctx, d_q, Q, d_t, nullptr, d_cand_off, d_cand_idx, n_cand, d_cand_dist, d_best_idx, d_best_dist, d_second_dist);
}

extern "C" int ccm_hamming_csr_multi_dev(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, const int32_t* d_q_tbase,
                                         const int32_t* d_cand_off, const int32_t* d_cand_idx, int64_t n_cand,
                                         uint16_t* d_cand_dist, int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist) {
  if (!ctx || Q < 0) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: bad args");
  if (d_best_idx && (!d_best_dist || !d_second_dist)) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: best2 outputs must be all set or all NULL");
  if (Q == 0) return CCM_OK;
  return csr_launch(ctx, d_q, Q, d_t, d_q_tbase, d_cand_off, d_cand_idx, n_cand, d_cand_dist, d_best_idx, d_best_dist, d_second_dist);
}

// S searches, one launch, one read-back.  Search s owns the query rows q_off[s] .. q_off[s+1] and the target rows t_off[s] .. t_off[s+1]; the candidate
// indices of its queries are LOCAL to its target set.
extern "C" int ccm_hamming_csr_multi(ccm_ctx* ctx, int S, const uint8_t* q, const int32_t* q_off, const uint8_t* t, const int32_t* t_off,
                                     const int32_t* cand_off, const int32_t* cand_idx, uint16_t* cand_dist,
                                     int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
  if (!ctx || S < 0 || (S && (!q_off || !t_off))) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: bad args");
  if (S == 0) return CCM_OK;
  const int Q = q_off[S], T = t_off[S];
  if (q_off[0] != 0 || t_off[0] != 0 || Q < 0 || T < 0 || (Q && (!q || !cand_off))) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: bad offsets");
  if (Q == 0) return CCM_OK;
  const int64_t n_cand = cand_off[Q];
  if (cand_off[0] != 0 || n_cand < 0 || (n_cand && (!cand_idx || !t))) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: bad candidate list (cand_off must start at 0)");
  void* pin = nullptr;
  { int rc0 = ccm_pin_scratch(ctx, (size_t)Q * 4 + 256, &pin); if (rc0) return rc0; }
  int32_t* h_base = (int32_t*)pin;
  for (int s = 0; s < S; s++) {
    if (q_off[s + 1] < q_off[s] || t_off[s + 1] < t_off[s]) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: offsets must ascend");
    const int Ts = t_off[s + 1] - t_off[s];
    for (int i = q_off[s]; i < q_off[s + 1]; i++) {
      h_base[i] = t_off[s];
      const int64_t len = (int64_t)cand_off[i + 1] - cand_off[i];
      if (len < 0 || len >= (1 << 20)) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: candidate list length out of range");
      for (int c = cand_off[i]; c < cand_off[i + 1]; c++)
        if (cand_idx[c] < 0 || cand_idx[c] >= Ts) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr_multi: candidate index out of range");
    }
  }
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bq = ccm_align256((size_t)Q * 32), bt = ccm_align256((size_t)std::max(T, 1) * 32), bo = ccm_align256((size_t)(Q + 1) * 4), bb = ccm_align256((size_t)Q * 4);
  const size_t bi = ccm_align256((size_t)std::max<int64_t>(n_cand, 1) * 4), bd = ccm_align256((size_t)std::max<int64_t>(n_cand, 1) * 2);
  void* io = nullptr;
  { int rc0 = ccm_io_scratch(ctx, bq + bt + bo + bb + bi + bd + (size_t)Q * 12, &io); if (rc0) return rc0; }
  uint8_t* d_q = (uint8_t*)io; uint8_t* d_t = d_q + bq; int32_t* d_off = (int32_t*)(d_t + bt); int32_t* d_base = (int32_t*)((uint8_t*)d_off + bo);
  int32_t* d_idx = (int32_t*)((uint8_t*)d_base + bb); uint16_t* d_dist = (uint16_t*)((uint8_t*)d_idx + bi); int32_t* d_out = (int32_t*)((uint8_t*)d_dist + bd);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_q, q, (size_t)Q * 32, hipMemcpyHostToDevice, ctx->stream));
  if (T) CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_t, t, (size_t)T * 32, hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_off, cand_off, (size_t)(Q + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_base, h_base, (size_t)Q * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_cand) CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_idx, cand_idx, (size_t)n_cand * 4, hipMemcpyHostToDevice, ctx->stream));
  int rc = ccm_hamming_csr_multi_dev(ctx, d_q, Q, d_t, d_base, d_off, d_idx, n_cand, d_dist, best_idx ? d_out : nullptr,
                                     best_idx ? d_out + Q : nullptr, best_idx ? d_out + 2 * (size_t)Q : nullptr);
  if (rc == CCM_OK) {
    if (cand_dist && n_cand) hipMemcpyAsync(cand_dist, d_dist, (size_t)n_cand * 2, hipMemcpyDeviceToHost, ctx->stream);
    if (best_idx) {
      hipMemcpyAsync(best_idx, d_out, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
      hipMemcpyAsync(best_dist, d_out + Q, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
      hipMemcpyAsync(second_dist, d_out + 2 * (size_t)Q, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, std::string("hamming_csr_multi: ") + hipGetErrorString(e));
  }
  return rc;
}

extern "C" int ccm_hamming_csr(ccm_ctx* ctx, const uint8_t* q, int Q, const uint8_t* t, int T,
                               const int32_t* cand_off, const int32_t* cand_idx, uint16_t* cand_dist,
                               int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
  if (!ctx || Q < 0 || T < 0 || (Q && (!q || !cand_off)))
    return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr: bad args");
  if (Q == 0) return CCM_OK;
  const int64_t n_cand = cand_off[Q];
  if (n_cand < 0 || (n_cand && (!cand_idx || !t))) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr: bad candidate list");
  for (int i = 0; i < Q; ++i) {
    const int64_t len = (int64_t)cand_off[i + 1] - cand_off[i];
    if (len < 0 || len >= (1 << 20)) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr: candidate list length out of range");
  }
  for (int64_t s = 0; s < n_cand; ++s)
    if (cand_idx[s] < 0 || cand_idx[s] >= T) return ccm_set_error(ctx, CCM_E_ARG, "hamming_csr: candidate index out of range");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bq = ccm_align256((size_t)Q * 32), bt = ccm_align256((size_t)std::max(T, 1) * 32), bo = ccm_align256((size_t)(Q + 1) * 4);
  const size_t bi = ccm_align256((size_t)std::max<int64_t>(n_cand, 1) * 4), bd = ccm_align256((size_t)std::max<int64_t>(n_cand, 1) * 2);
  void* io = nullptr;
  { int rc0 = ccm_io_scratch(ctx, bq + bt + bo + bi + bd + (size_t)Q * 12, &io); if (rc0) return rc0; }
  uint8_t* d_q = (uint8_t*)io; uint8_t* d_t = d_q + bq; int32_t* d_off = (int32_t*)(d_t + bt); int32_t* d_idx = (int32_t*)((uint8_t*)d_off + bo);
  uint16_t* d_dist = (uint16_t*)((uint8_t*)d_idx + bi); int32_t* d_out = (int32_t*)((uint8_t*)d_dist + bd);
  hipMemcpyAsync(d_q, q, (size_t)Q * 32, hipMemcpyHostToDevice, ctx->stream);
  if (T) hipMemcpyAsync(d_t, t, (size_t)T * 32, hipMemcpyHostToDevice, ctx->stream);
  hipMemcpyAsync(d_off, cand_off, (size_t)(Q + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
  if (n_cand) hipMemcpyAsync(d_idx, cand_idx, (size_t)n_cand * 4, hipMemcpyHostToDevice, ctx->stream);
  int rc = ccm_hamming_csr_dev(ctx, d_q, Q, d_t, T, d_off, d_idx, n_cand, d_dist, best_idx ? d_out : nullptr,
                               best_idx ? d_out + Q : nullptr, best_idx ? d_out + 2 * (size_t)Q : nullptr);
  if (rc == CCM_OK) {
    if (cand_dist && n_cand) hipMemcpyAsync(cand_dist, d_dist, (size_t)n_cand * 2, hipMemcpyDeviceToHost, ctx->stream);
    if (best_idx) {
      hipMemcpyAsync(best_idx, d_out, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
      hipMemcpyAsync(best_dist, d_out + Q, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
      hipMemcpyAsync(second_dist, d_out + 2 * (size_t)Q, (size_t)Q * 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = ccm_set_error(ctx, CCM_E_HIP, std::string("hamming_csr: ") + hipGetErrorString(e));
  }
  return rc;
}

extern "C" int ccm_distinctive_descriptors(ccm_ctx* ctx, const uint8_t* desc, const int32_t* off, int P, int32_t* best_local_idx) {
  if (!ctx || P < 0 || (P && (!off || !best_local_idx))) return ccm_set_error(ctx, CCM_E_ARG, "distinctive: bad args");
  if (P == 0) return CCM_OK;
  const int64_t total = off[P];
  if (total < 0 || (total && !desc)) return ccm_set_error(ctx, CCM_E_ARG, "distinctive: bad offsets");
  for (int p = 0; p < P; p++) {
    const int n = off[p + 1] - off[p];
    if (n < 0 || n > kDistinctMaxN) return ccm_set_error(ctx, CCM_E_ARG, "distinctive: a map point has more than 256 observations (or a negative count)");
  }
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bd = ccm_align256((size_t)std::max<int64_t>(total, 1) * 32), bo = ccm_align256((size_t)(P + 1) * 4);
  void* io = nullptr;
  { int rc0 = ccm_io_scratch(ctx, bd + bo + (size_t)P * 4, &io); if (rc0) return rc0; }
  uint8_t* d_desc = (uint8_t*)io; int32_t* d_off = (int32_t*)(d_desc + bd); int32_t* d_best = (int32_t*)((uint8_t*)d_off + bo);
  if (total) CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_desc, desc, (size_t)total * 32, hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_off, off, (size_t)(P + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(distinctive_kernel, dim3(P), dim3(64), 64 * kDistinctMaxN * sizeof(uint16_t), ctx->stream, (const uint32_t*)d_desc, d_off, P, d_best);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(best_local_idx, d_best, (size_t)P * 4, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

// ---- vocabulary handle ------------------------------------------------------------------------------------
struct ccm_vocab {
  ccm_ctx* ctx; int n_nodes, L;
  int32_t *d_child_off, *d_child_id; uint32_t* d_node_desc;
  std::vector<int32_t> word_id; std::vector<double> weight;
};

extern "C" int ccm_vocab_create(ccm_ctx* ctx, int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc,
                                const int32_t* word_id, const double* weight, ccm_vocab** out) {
  if (!ctx || !out || n_nodes <= 0 || !child_off || !node_desc || !word_id || !weight || L <= 0)
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_vocab_create: bad args");
  const int n_child = child_off[n_nodes];
  if (n_child < 0 || (n_child && !child_id)) return ccm_set_error(ctx, CCM_E_ARG, "ccm_vocab_create: bad children");
  for (int i = 0; i < n_nodes; i++) {
    const int k = child_off[i + 1] - child_off[i];
    if (k < 0 || k >= (1 << 20)) return ccm_set_error(ctx, CCM_E_ARG, "ccm_vocab_create: bad child count");
  }
  for (int i = 0; i < n_child; i++) if (child_id[i] <= 0 || child_id[i] >= n_nodes) return ccm_set_error(ctx, CCM_E_ARG, "ccm_vocab_create: child id out of range");
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ccm_vocab* v = new ccm_vocab();
  v->ctx = ctx; v->n_nodes = n_nodes; v->L = L;
  v->word_id.assign(word_id, word_id + n_nodes); v->weight.assign(weight, weight + n_nodes);
  const bool ok = hipMalloc(&v->d_child_off, (size_t)(n_nodes + 1) * 4) == hipSuccess && hipMalloc(&v->d_child_id, (size_t)std::max(n_child, 1) * 4) == hipSuccess &&
                  hipMalloc(&v->d_node_desc, (size_t)n_nodes * 32) == hipSuccess &&
                  hipMemcpyAsync(v->d_child_off, child_off, (size_t)(n_nodes + 1) * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                  (!n_child || hipMemcpyAsync(v->d_child_id, child_id, (size_t)n_child * 4, hipMemcpyHostToDevice, ctx->stream) == hipSuccess) &&
                  hipMemcpyAsync(v->d_node_desc, node_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                  hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) {   // nothing of a half-built vocabulary survives
    (void)hipGetLastError();
    ccm_vocab_destroy(v);
    return ccm_set_error(ctx, CCM_E_HIP, "ccm_vocab_create: device allocation / upload failed");
  }
  *out = v;
  return CCM_OK;
}

extern "C" void ccm_vocab_destroy(ccm_vocab* v) {
  if (!v) return;
  hipSetDevice(v->ctx->device); hipStreamSynchronize(v->ctx->stream);
  hipFree(v->d_child_off); hipFree(v->d_child_id); hipFree(v->d_node_desc);
  delete v;
}

extern "C" int ccm_bow_transform(ccm_vocab* v, const uint8_t* desc, int N, int levelsup, int32_t* word_out, double* weight_out, int32_t* node_out) {
  if (!v || N < 0 || (N && (!desc || !word_out || !weight_out || !node_out))) return ccm_set_error(v ? v->ctx : nullptr, CCM_E_ARG, "ccm_bow_transform: bad args");
  if (N == 0) return CCM_OK;
  ccm_ctx* ctx = v->ctx;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bd = ccm_align256((size_t)N * 32);
  void* io = nullptr;
  { int rc0 = ccm_io_scratch(ctx, bd + (size_t)N * 8, &io); if (rc0) return rc0; }
  uint8_t* d_desc = (uint8_t*)io; int32_t* d_leaf = (int32_t*)(d_desc + bd); int32_t* d_nid = d_leaf + N;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(d_desc, desc, (size_t)N * 32, hipMemcpyHostToDevice, ctx->stream));
  const int nid_level = v->L - levelsup;   // if <= 0 the node stays 0 (root), TemplatedVocabulary.h:1227
  hipLaunchKernelGGL(bow_descend_kernel, dim3(ccm_div_up(N, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)d_desc, N, v->d_child_off, v->d_child_id,
                     v->d_node_desc, nid_level, d_leaf, d_nid);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  std::vector<int32_t> leaf(N);
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(leaf.data(), d_leaf, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(node_out, d_nid, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < N; i++) { word_out[i] = v->word_id[leaf[i]]; weight_out[i] = v->weight[leaf[i]]; }
  return CCM_OK;
}
