// ba_build.hip — ccm_ba_create: the structure of a bundle-adjustment problem, built ON THE DEVICE.
//
// What g2o does in initializeOptimization(0) + buildStructure (sparse_optimizer.cpp:199-267, block_solver.hpp:143-295) — active set, vertex
// ordering, Hpp / Hll / Hpl / Hschur block pattern — plus the index structures of our own kernels (per-camera edge lists, block-CSR rows, the
// persistent solver's column lists, the row Schur kernel's unit table, the coarse level's block lists).  Round 2 did the edge sort, the row lists
// and the local arrays on the host (15 of the 16 ms of a warm ccm_ba_create on the 4-agent map); here the host only copies the caller's flat
// arrays to the device and reads back a handful of sizes:
//   mark (active edges, touched vertices) -> slot scans -> stable radix sort of the edges by (landmark slot, pose slot) -> landmark edge ranges
//   -> gather of the local edge arrays -> per-camera edge lists (stable sort by pose slot) -> pair structure (ba_structure.hip) -> block-CSR
//   rows by counting ranks -> per-unit column bitmaps (persistent PCG) -> cluster entry lists -> row-kernel unit table (ranked in LDS) ->
//   coarse block lists (stable sort + run-length encode).
// Every list has the order the host version produced (the kernels' summation orders depend on it), so results are bit-identical to round 2;
// tests/test_ba_structure_gpu.py recomputes every array with numpy from the flat problem and compares.
// Host work that remains: the greedy chunking of consecutive landmarks (a sequential scan over Lp + 1 prefix sums read back once) and, for a
// sharded handle, the landmark partition (ccm_ba_partition over the per-landmark weights).
#include "common.h"
#include "ba_types.h"
#include "ba_math.h"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <rocprim/rocprim.hpp>

int ccm_ba_build_pairs(ccm_ctx* ctx, const int* d_pt_off, const int* d_cslot, int Lp, int Cp, int lb, int le, int eb,
                       uint32_t** d_U_out, int* nOff_out, int** d_inst_off, int** d_inst_a, int** d_inst_c, int64_t* n_inst,
                       std::vector<std::pair<void*, size_t>>& keep);   // ba_structure.hip
int ccm_ba_pers_grid_fits(ccm_ctx* ctx, int grid);                      // ba.hip: occupancy of the persistent PCG kernel
int ccm_ba_state_from_raw(ccm_ba* ba);                                  // below

namespace {

constexpr int kB = 256;

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// device-side counters of one build, read back in one copy
struct BuildSizes {
  int Cp, Lp, Eact, bad, max_cam_edges, n_fix_edges, worst_units, n_units, pers_bad, n_ucol, n_cij, ncb, pad[4];
};

#define BB_RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
#define BB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return ccm_set_error(ctx, CCM_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct Tmp {   // temporaries of the build: back to the pool when the build ends (after a stream sync)
  ccm_ctx* ctx;
  std::vector<std::pair<void*, size_t>> blocks;
  template <typename T> int get(size_t n, T** out) {
    void* p = nullptr; size_t actual = 0;
    if (int rc = ccm_pool_get(ctx, std::max<size_t>(n, 1) * sizeof(T), &p, &actual)) return rc;
    blocks.push_back({p, actual});
    *out = (T*)p;
    return CCM_OK;
  }
  ~Tmp() { hipStreamSynchronize(ctx->stream); for (auto& b : blocks) ccm_pool_put(ctx, b.first, b.second); }
};

template <typename T>
int keep_get(ccm_ba* ba, size_t n, T** out, bool zero = false) {
  ccm_ctx* ctx = ba->ctx;
  void* p = nullptr; size_t actual = 0;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  if (int rc = ccm_pool_get(ctx, bytes, &p, &actual)) return rc;
  ba->allocs.push_back({p, actual});
  if (zero) ba->zero_list.push_back({p, bytes});   // cleared together by flush_zero_list: ~45 separate fills were 0.2 ms of a local window's 0.75 ms structure build
  *out = (T*)p;
  return CCM_OK;
}

// one launch clears every buffer a handle asked to have zeroed (keep_get(..., true)): (pointer, size) pairs as kernel arguments; a workgroup takes one 64 KB
// chunk of one buffer (chunk c of the concatenation: linear search over the <= 96 prefix counts)
constexpr int kZeroMax = 96;
constexpr unsigned kZeroChunk16 = 4096;   // 16-byte words per workgroup
struct ZeroTab { void* p[kZeroMax]; unsigned n16[kZeroMax]; unsigned first_chunk[kZeroMax + 1]; int n; };
__global__ __launch_bounds__(256) void bb_zero_many(ZeroTab tab) {
  int k = 0;
  while (k + 1 < tab.n && blockIdx.x >= tab.first_chunk[k + 1]) k++;
  const unsigned c = blockIdx.x - tab.first_chunk[k];
  uint4* q = reinterpret_cast<uint4*>(tab.p[k]);
  const unsigned end = min(tab.n16[k], (c + 1) * kZeroChunk16);
  for (unsigned i = c * kZeroChunk16 + threadIdx.x; i < end; i += 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
}
int flush_zero_list(ccm_ba* ba) {
  ccm_ctx* ctx = ba->ctx;
  size_t k = 0;
  while (k < ba->zero_list.size()) {
    ZeroTab tab;
    tab.n = 0;
    unsigned chunks = 0;
    for (; k < ba->zero_list.size() && tab.n < kZeroMax; k++) {
      // pool blocks are 256-byte aligned and at least as large as the request rounded up to 16 bytes (smallest bucket: 4 KiB)
      const size_t n16 = (ba->zero_list[k].second + 15) / 16;
      if (n16 > 0xffffffffull / 2) { BB_HIP(hipMemsetAsync(ba->zero_list[k].first, 0, ba->zero_list[k].second, ctx->stream)); continue; }
      tab.p[tab.n] = ba->zero_list[k].first; tab.n16[tab.n] = (unsigned)n16; tab.first_chunk[tab.n] = chunks;
      chunks += (unsigned)((n16 + kZeroChunk16 - 1) / kZeroChunk16);
      tab.n++;
    }
    tab.first_chunk[tab.n] = chunks;
    if (chunks) hipLaunchKernelGGL(bb_zero_many, dim3(chunks), dim3(256), 0, ctx->stream, tab);
    BB_HIP(hipGetLastError());
  }
  ba->zero_list.clear();
  return CCM_OK;
}

int scan_excl(ccm_ctx* ctx, Tmp& tmp, const int* in, int* out, size_t n) {
  size_t bytes = 0;
  BB_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), ctx->stream));
  char* scratch = nullptr;
  BB_RC(tmp.get(bytes, &scratch));
  BB_HIP(rocprim::exclusive_scan(scratch, bytes, in, out, 0, n, rocprim::plus<int>(), ctx->stream));
  return CCM_OK;
}

template <typename K>
int sort_pairs(ccm_ctx* ctx, Tmp& tmp, K* keys, int* vals, size_t n, unsigned bits, K** keys_sorted, int** vals_sorted) {
  *keys_sorted = keys; *vals_sorted = vals;
  if (n == 0) return CCM_OK;
  K* k2 = nullptr; int* v2 = nullptr;
  BB_RC(tmp.get(n, &k2)); BB_RC(tmp.get(n, &v2));
  rocprim::double_buffer<K> kb(keys, k2);
  rocprim::double_buffer<int> vb(vals, v2);
  size_t bytes = 0;
  BB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kb, vb, (unsigned)n, 0u, bits, ctx->stream));
  char* scratch = nullptr;
  BB_RC(tmp.get(bytes, &scratch));
  BB_HIP(rocprim::radix_sort_pairs(scratch, bytes, kb, vb, (unsigned)n, 0u, bits, ctx->stream));
  *keys_sorted = kb.current(); *vals_sorted = vb.current();
  return CCM_OK;
}

inline unsigned bits_for(uint64_t n_values) {   // smallest b with 2^b >= n_values (at least 1)
  unsigned b = 1;
  while (b < 63 && ((uint64_t)1 << b) < n_values) b++;
  return b;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------------------
// active edges (level 0) mark their two vertices; an index out of range is reported (ccm_ba_create fails)
__global__ void bb_mark(int n_edge, const int* ecam, const int* ept, const uint8_t* lvl, int n_cam, int n_pt, int* cam_has, int* pt_has, BuildSizes* sz) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edge) return;
  const int c = ecam[e], p = ept[e];
  if (c < 0 || c >= n_cam || p < 0 || p >= n_pt) { sz->bad = 1; return; }
  if (lvl && lvl[e] != 0) return;
  cam_has[c] = 1; pt_has[p] = 1;
}

// flag[i] = has[i] (and not fixed); flag[n] = 0 closes the scan
__global__ void bb_flags(int n, const int* has, const uint8_t* fixed, int* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  flag[i] = (i < n && has[i] && !(fixed && fixed[i])) ? 1 : 0;
}

// slot[i] = rank among the flagged (or -1), inv[rank] = i, *total = number flagged
__global__ void bb_slots(int n, const int* flag, const int* pre, int* slot, int* inv, int* total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) { *total = pre[n]; return; }
  const int f = flag[i];
  slot[i] = f ? pre[i] : -1;
  if (f) inv[pre[i]] = i;
}

// sort key of every edge: (landmark slot, pose slot + 1) — fixed cameras (slot -1) first inside a landmark; inactive edges sort last
__global__ void bb_keys(int n_edge, const int* ecam, const int* ept, const uint8_t* lvl, int n_cam, int n_pt, const int* cam_slot, const int* pt_slot, unsigned cb,
                        unsigned pb, unsigned long long* keys, int* vals) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edge) return;
  const bool in_range = ecam[e] >= 0 && ecam[e] < n_cam && ept[e] >= 0 && ept[e] < n_pt;   // an index out of range was reported by bb_mark: the build fails at the first read-back
  const bool act = in_range && !(lvl && lvl[e] != 0);
  unsigned long long k = (((unsigned long long)1 << (pb + cb)) - 1ull);
  if (act) k = ((unsigned long long)(unsigned)pt_slot[ept[e]] << cb) | (unsigned long long)(unsigned)(cam_slot[ecam[e]] + 1);
  keys[e] = k;
  vals[e] = e;
}

// landmark edge ranges from the sorted keys (every landmark slot has at least one active edge), pose slot per sorted position, active count
__global__ void bb_ptoff(int n_edge, const unsigned long long* keys, unsigned cb, unsigned pb, int* pt_off, int* cslot_g, BuildSizes* sz) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edge) return;
  const unsigned long long sent = ((unsigned long long)1 << pb) - 1ull;
  const unsigned long long cur = keys[k] >> cb;
  if (cur == sent) return;
  cslot_g[k] = (int)(keys[k] & (((unsigned long long)1 << cb) - 1ull)) - 1;
  const unsigned long long nxt = (k + 1 < n_edge) ? (keys[k + 1] >> cb) : sent;
  if (nxt != cur) pt_off[cur + 1] = k + 1;
  if (nxt == sent) sz->Eact = k + 1;
}

// weight of a landmark for the shard balance: pair instances + edges (as the host version)
__global__ void bb_weights(int Lp, const int* pt_off, const int* cslot_g, long long* w) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Lp) return;
  int kf = 0;
  for (int k = pt_off[l]; k < pt_off[l + 1]; k++) kf += cslot_g[k] >= 0;
  w[l] = (long long)kf * (kf + 1) / 2 + (pt_off[l + 1] - pt_off[l]);
}

__global__ void bb_ptoff_local(int n, const int* g_pt_off, int eb, int* pt_off) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < n) pt_off[l] = g_pt_off[l] - eb;
}

// local edge arrays of the own landmarks, in sorted order
__global__ void bb_gather(int Eloc, int eb, int lb, const unsigned long long* keys, const int* vals, unsigned cb, const int* ecam, const double* obs_raw,
                          const double* info_raw, int* ed_cam, int* ed_cslot, int* ed_pt, double* obs, double* info, int* loc_orig, unsigned* ckey, int* cval) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Eloc) return;
  const unsigned long long key = keys[eb + k];
  const int e = vals[eb + k];
  const int cs = (int)(key & (((unsigned long long)1 << cb) - 1ull)) - 1;
  ed_cam[k] = ecam[e]; ed_cslot[k] = cs; ed_pt[k] = (int)(key >> cb) - lb;
  obs[2 * (size_t)k] = obs_raw[2 * (size_t)e]; obs[2 * (size_t)k + 1] = obs_raw[2 * (size_t)e + 1];
  info[k] = info_raw[e];
  loc_orig[k] = e;
  ckey[k] = (unsigned)(cs + 1); cval[k] = k;
}

// off[s] = first position whose key is >= s  (s = 0 .. nq-1)
__global__ void bb_lower_bound(const unsigned* a, int n, int nq, int* off) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nq) return;
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < (unsigned)s) lo = mid + 1; else hi = mid; }
  off[s] = lo;
}

// per-camera edge lists (edges of fixed cameras — key 0 — are skipped), landmark of every list slot, longest list
__global__ void bb_cam_lists(int Eloc, const int* off_all /* [Cp+2] */, const int* cval, const int* ed_pt, int* cam_edge, int* cam_pt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int nfix = off_all[1];
  if (q >= Eloc - nfix) return;
  const int k = cval[nfix + q];
  cam_edge[q] = k; cam_pt[q] = ed_pt[k];
}
// camera-major copy of (observation, information) (ba.hip refreshes it after ccm_ba_set_edge_levels has changed the informations)
__global__ void bb_cam_oi_bounded(int Eloc, const int* __restrict__ off_all, const int* __restrict__ cam_edge, const double* __restrict__ obs, const double* __restrict__ info,
                                  double* __restrict__ cam_oi) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= Eloc - off_all[1]) return;     // off_all[1] = edges of fixed cameras: they have no list slot
  const int e = cam_edge[s];
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d a, b; a[0] = obs[2 * (size_t)e]; a[1] = obs[2 * (size_t)e + 1]; b[0] = info[e]; b[1] = 0.0;
  reinterpret_cast<v2d*>(cam_oi)[2 * (size_t)s] = a; reinterpret_cast<v2d*>(cam_oi)[2 * (size_t)s + 1] = b;
}
__global__ void bb_cam_off(int Cp, const int* off_all, int* cam_off, BuildSizes* sz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > Cp) return;
  cam_off[i] = off_all[i + 1] - off_all[1];
  if (i < Cp) atomicMax(&sz->max_cam_edges, off_all[i + 2] - off_all[i + 1]);
  if (i == 0) sz->n_fix_edges = off_all[1];
}

// block list -> (row, column) per block, sort keys of the column lists, row ranges
__global__ void bb_block_ij(int nOff, const uint32_t* U, int Cp, int* bi, int* bj, unsigned* jkey, int* jval) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nOff) return;
  const uint32_t u = U[b];
  const int i = (int)(u / (uint32_t)Cp), j = (int)(u % (uint32_t)Cp);
  bi[Cp + b] = i; bj[Cp + b] = j;
  jkey[b] = (unsigned)j; jval[b] = b;
}
__global__ void bb_diag_ij(int Cp, int* bi, int* bj) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Cp) { bi[i] = i; bj[i] = i; }
}
__global__ void bb_rowblk_off(int Cp, const uint32_t* U, int nOff, int* rowblk_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > Cp) return;
  const unsigned long long key = (unsigned long long)i * (unsigned long long)Cp;
  int lo = 0, hi = nOff;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if ((unsigned long long)U[mid] < key) lo = mid + 1; else hi = mid; }
  rowblk_off[i] = lo;
}
// block-CSR rows of the full symmetric pattern, columns ascending: lower part (transposed blocks), diagonal, upper part.
// row_off[i] = i + rowblk_off[i] + low_off[i]
__global__ void bb_row_off(int Cp, const int* rowblk_off, const int* low_off, int* row_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= Cp) row_off[i] = i + rowblk_off[i] + low_off[i];
}
__global__ void bb_row_fill(int Cp, int nOff, const int* bi, const int* bj, const unsigned* jkey_sorted, const int* jval_sorted, const int* rowblk_off,
                            const int* low_off, const int* row_off, int* row_col, uint32_t* row_blk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nOff) {
    {   // lower entry: position t of the column-grouped list belongs to row j = key, block b = value (ascending b = ascending i inside the group)
      const int j = (int)jkey_sorted[t], b = jval_sorted[t];
      const int pos = row_off[j] + (t - low_off[j]);
      row_col[pos] = bi[Cp + b]; row_blk[pos] = (uint32_t)(Cp + b) | kTransposeBit;
    }
    {   // upper entry of block t
      const int i = bi[Cp + t];
      const int pos = row_off[i] + (low_off[i + 1] - low_off[i]) + 1 + (t - rowblk_off[i]);
      row_col[pos] = bj[Cp + t]; row_blk[pos] = (uint32_t)(Cp + t);
    }
  }
  if (t < Cp) {
    const int pos = row_off[t] + (low_off[t + 1] - low_off[t]);
    row_col[pos] = t; row_blk[pos] = (uint32_t)t;
  }
}

// persistent PCG: per unit (8 rows, two units per 16-camera cluster) the ascending list of distinct columns its rows touch and every entry's
// position in it.  A bitmap over the columns in LDS; the position of a column = number of set bits below it.
__device__ __forceinline__ void pers_unit_rows(int u, int Cp, int& r0, int& r1) {
  r0 = min(Cp, (u >> 1) * kClu + (u & 1) * (kClu / 2));
  r1 = min(min(Cp, r0 + kClu / 2), ((u >> 1) + 1) * kClu);
}
template <int FILL>
__global__ __launch_bounds__(kB) void bb_pers_units(int Cp, int n_words, const int* row_off, const int* row_col, int* ucnt, const int* uoff, int* ucol, int* loc,
                                                  BuildSizes* sz) {
  extern __shared__ unsigned bm[];      // [n_words] bitmap | [n_words] exclusive popcount prefix
  unsigned* wpre = bm + n_words;
  __shared__ int s_tot;
  const int u = blockIdx.x, t = threadIdx.x;
  int r0, r1;
  pers_unit_rows(u, Cp, r0, r1);
  for (int w = t; w < n_words; w += kB) bm[w] = 0u;
  __syncthreads();
  const int e0 = row_off[r0], e1 = row_off[r1];
  for (int s = e0 + t; s < e1; s += kB) { const int c = row_col[s]; atomicOr(&bm[c >> 5], 1u << (c & 31)); }
  __syncthreads();
  if (t == 0) {   // n_words <= 2048: a serial prefix is a few microseconds and runs once per problem
    int acc = 0;
    for (int w = 0; w < n_words; w++) { wpre[w] = (unsigned)acc; acc += __popc(bm[w]); }
    s_tot = acc;
  }
  __syncthreads();
  if (!FILL) {
    if (t == 0) { ucnt[u] = s_tot; if (s_tot > kPersColCap || e1 - e0 > kPersIdxCap) sz->pers_bad = 1; }
    return;
  }
  const int base = uoff[u];
  for (int w = t; w < n_words; w += kB) {
    unsigned m = bm[w];
    int pos = base + (int)wpre[w];
    while (m) { const int b = __ffs(m) - 1; ucol[pos++] = 32 * w + b; m &= m - 1; }
  }
  for (int s = e0 + t; s < e1; s += kB) { const int c = row_col[s]; loc[s] = (int)wpre[c >> 5] + __popc(bm[c >> 5] & ((1u << (c & 31)) - 1u)); }
}

// entries of the block-CSR rows that fall inside each cluster's own 16x16 block, in (row, CSR position) order: local row << 4 | local column, S block
template <int FILL>
__global__ __launch_bounds__(kB) void bb_cluster_entries(int Cp, const int* row_off, const int* row_col, const uint32_t* row_blk, int* ccnt, const int* coff, int* cij,
                                                       uint32_t* cblk) {
  __shared__ int s_scan[kB];
  __shared__ int s_run;
  const int c = blockIdx.x, t = threadIdx.x;
  const int r0 = c * kClu, r1 = min(Cp, r0 + kClu);
  const int e0 = row_off[r0], e1 = row_off[r1];
  if (t == 0) s_run = 0;
  __syncthreads();
  for (int sb = e0; sb < e1; sb += kB) {
    const int s = sb + t;
    int col = -1;
    if (s < e1) col = row_col[s];
    const int in = (s < e1 && col >= r0 && col < r1) ? 1 : 0;
    s_scan[t] = in;
    __syncthreads();
    for (int off = 1; off < kB; off <<= 1) {   // inclusive Hillis-Steele scan
      const int v = (t >= off) ? s_scan[t - off] : 0;
      __syncthreads();
      s_scan[t] += v;
      __syncthreads();
    }
    const int run = s_run;
    if (FILL && in) {
      int lo = r0, hi = r1;   // row of entry s: the last row whose offset is <= s
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_off[mid] <= s) lo = mid; else hi = mid; }
      const int pos = coff[c] + run + s_scan[t] - 1;
      cij[pos] = ((lo - r0) << 4) | (col - r0);
      cblk[pos] = row_blk[s];
    }
    __syncthreads();
    if (t == kB - 1) s_run = run + s_scan[kB - 1];
    __syncthreads();
  }
  if (!FILL && t == 0) ccnt[c] = s_run;
}

// row Schur kernel: work units = (block, chunk of <= `chunk` consecutive pair instances), at least one per block; a block's units are consecutive
__global__ void bb_unit_counts(int nOff, int chunk, const int* inst_off, int* nub) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= nOff) nub[t] = (t < nOff) ? max(1, (inst_off[t + 1] - inst_off[t] + chunk - 1) / chunk) : 0;
}
__global__ void bb_unit_offsets(int nOff, int Cp, const int* blk_pref, const int* rowblk_off, int* row_unit_off, int* blk_unit0, BuildSizes* sz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= Cp) row_unit_off[t] = blk_pref[rowblk_off[t]];
  if (t < Cp) atomicMax(&sz->worst_units, blk_pref[rowblk_off[t + 1]] - blk_pref[rowblk_off[t]]);
  if (t == 0) sz->n_units = blk_pref[nOff];
  if (t <= nOff) blk_unit0[t] = blk_pref[t];
}
// one workgroup per camera row: the row's units in creation order (block after block), then ranked LONGEST FIRST (stable) — the order the waves of the row
// kernel are dealt their units in, four per pass, so that the units of a pass have similar lengths; the slot keeps the creation index (where the partial
// sum goes and where the final per-block sums find it)
__global__ __launch_bounds__(kB) void bb_unit_fill(int Cp, int chunk, const int* rowblk_off, const int* inst_off, const int* row_unit_off, const int* blk_unit0, int4* tab) {
  extern __shared__ int su[];   // [n][3]: block, first, end
  const int i = blockIdx.x, t = threadIdx.x;
  const int base = row_unit_off[i], n = row_unit_off[i + 1] - base;
  int* blk = su; int* s0a = su + n; int* s1a = su + 2 * n;
  for (int b = rowblk_off[i] + t; b < rowblk_off[i + 1]; b += kB) {
    int c = blk_unit0[b] - base, s0 = inst_off[b];
    const int end = inst_off[b + 1];
    do { const int s1 = min(end, s0 + chunk); blk[c] = b; s0a[c] = s0; s1a[c] = s1; c++; s0 = s1; } while (s0 < end);
  }
  __syncthreads();
  for (int c = t; c < n; c += kB) {
    const int len = s1a[c] - s0a[c];
    int rank = 0;
    for (int o = 0; o < n; o++) { const int lo = s1a[o] - s0a[o]; rank += (lo > len || (lo == len && o < c)) ? 1 : 0; }
    tab[base + rank] = make_int4(blk[c], s0a[c], s1a[c], c);
  }
}

// rank of every observation inside its camera's list, then per pair instance the rank of its row-side observation
__global__ void bb_edge_rank(const int* cam_off, const int* cam_edge, int* rank) {
  const int i = blockIdx.x;
  for (int s = cam_off[i] + threadIdx.x; s < cam_off[i + 1]; s += blockDim.x) rank[cam_edge[s]] = s - cam_off[i];
}
__global__ void bb_inst_rank(const int* inst_a, const int* rank, int n, int* inst_al) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) inst_al[s] = rank[inst_a[s]];
}
// position of every observation in camera-major order (the order of cam_edge; the caller presets -1: observations of fixed cameras stay there), then per
// pair instance the position of its column-side observation
__global__ void bb_edge_cpos(const int* cam_off, const int* cam_edge, int* cpos) {
  const int i = blockIdx.x;
  for (int s = cam_off[i] + threadIdx.x; s < cam_off[i + 1]; s += blockDim.x) cpos[cam_edge[s]] = s;
}
__global__ void bb_inst_cpos(const int* inst_c, const int* cpos, int n, int* inst_cp) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) inst_cp[s] = cpos[inst_c[s]];
}

// coarse level: every S block (diagonal first, then the off-diagonal ones in block order) with the key of its aggregate pair
__global__ void bb_coarse_keys(int Cp, int nOff, const int* bi, const int* bj, int na, int agg, unsigned* keys, int* vals) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= Cp + nOff) return;
  const int a = bi[k] / agg, a2 = bj[k] / agg;   // i < j => a <= a2
  keys[k] = (unsigned)a * (unsigned)na + (unsigned)a2;
  vals[k] = (k < Cp) ? 2 * k : 2 * k + (a == a2 ? 1 : 0);
}
__global__ void bb_coarse_ab(const unsigned* uniq, const int* n_runs, int na, int* cb_ab, BuildSizes* sz) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) sz->ncb = *n_runs;
  if (r >= *n_runs) return;
  cb_ab[2 * r] = (int)(uniq[r] / (unsigned)na); cb_ab[2 * r + 1] = (int)(uniq[r] % (unsigned)na);
}
__global__ void bb_counts_tail(unsigned* counts, const int* n_runs) { counts[*n_runs] = 0u; }

// initial estimate: SE3Quat(q, t) normalises the rotation (se3quat.h:61-63); own landmarks in slot order
__global__ void bb_state_init(int n_cam, int Lloc, int lb, const double* raw_cam, const double* raw_pt, const int* slot_pt, double* cam0, double* cam1, double* pt0,
                              double* pt1) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_cam) {
    BaPose T = ba_load_pose(raw_cam + 7 * (size_t)t);
    ba_normalize_rotation(T);
    ba_store_pose(cam0 + 7 * (size_t)t, T); ba_store_pose(cam1 + 7 * (size_t)t, T);
  }
  if (t < Lloc) {
    const int p = slot_pt[lb + t];
#pragma unroll
    for (int c = 0; c < 3; c++) { const double v = raw_pt[3 * (size_t)p + c]; pt0[3 * (size_t)t + c] = v; pt1[3 * (size_t)t + c] = v; }
  }
}

// optimised landmarks back into the caller's numbering (landmarks without an active edge keep the values passed in)
__global__ void bb_points_out(int Lp, const int* slot_pt, const double* pt_slots, double* out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= Lp) return;
  const int p = slot_pt[l];
#pragma unroll
  for (int c = 0; c < 3; c++) out[3 * (size_t)p + c] = pt_slots[3 * (size_t)l + c];
}

inline int grid_for(int64_t n) { return std::max(1, ccm_div_up(n, kB)); }

static inline int row_min_blocks() { return 256; }

}  // namespace

// landmarks in the caller's numbering, on the device: raw input overwritten by the optimised own / gathered landmarks (ccm_ba_download)
int ccm_ba_points_to_raw_order(ccm_ba* ba, const double* d_pt_slots /* [Lp][3] */) {
  ccm_ctx* ctx = ba->ctx;
  if (ba->Lp) hipLaunchKernelGGL(bb_points_out, dim3(grid_for(ba->Lp)), dim3(kB), 0, ctx->stream, ba->Lp, (const int*)ba->d_slot_pt, d_pt_slots, ba->d_raw_pt);
  BB_HIP(hipGetLastError());
  return CCM_OK;
}

// cam[0] = cam[1] = normalised raw cameras, pt[0] = pt[1] = raw landmarks of the own slots
int ccm_ba_state_from_raw(ccm_ba* ba) {
  ccm_ctx* ctx = ba->ctx;
  BaDev& d = ba->d;
  const int n = std::max(ba->n_cam, ba->Lloc);
  hipLaunchKernelGGL(bb_state_init, dim3(grid_for(n)), dim3(kB), 0, ctx->stream, ba->n_cam, ba->Lloc, ba->lp_begin, (const double*)ba->d_raw_cam, (const double*)ba->d_raw_pt,
                     (const int*)ba->d_slot_pt, d.cam[0], d.cam[1], d.pt[0], d.pt[1]);
  BB_HIP(hipGetLastError());
  ba->cur = 0;
  return CCM_OK;
}

extern "C" int ccm_ba_create(ccm_ctx* ctx, const ccm_ba_problem* P, int rank, int nranks, ccm_ba** out) {
  if (!ctx || !P || !out || nranks < 1 || rank < 0 || rank >= nranks) return ccm_set_error(ctx, CCM_E_ARG, "ccm_ba_create: bad args");
  if (P->n_cam <= 0 || P->n_pt < 0 || P->n_edge < 0 || !P->cam_qt || !P->cam_fixed || !P->cam_K ||
      (P->n_pt && !P->pt_xyz) || (P->n_edge && (!P->e_cam || !P->e_pt || !P->e_obs || !P->e_info)))
    return ccm_set_error(ctx, CCM_E_ARG, "ccm_ba_create: incomplete problem");
  const double t0 = now_ms();
  const bool setup_dbg = ccm_dbg("setup");
  double t_last = t0;
  auto lap = [&](const char* what) {
    if (setup_dbg) { hipStreamSynchronize(ctx->stream); const double t = now_ms(); fprintf(stderr, "[ccm_ba] setup %-26s %7.2f ms\n", what, t - t_last); t_last = t; }
  };
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ccm_ba* ba = new ccm_ba();
  ba->ctx = ctx; ba->rank = rank; ba->nranks = nranks;
  const int n_cam = ba->n_cam = P->n_cam, n_pt = ba->n_pt = P->n_pt, n_edge = ba->n_edge = P->n_edge;
  hipStream_t st = ctx->stream;
  int rc_build = CCM_OK;
  {
  std::vector<int> chunk(1, 0);   // landmark chunks: uploaded asynchronously from this pageable vector, so it must outlive ~Tmp (which drains the stream) on every path
  Tmp tmp{ctx, {}};
  auto body = [&]() -> int {
    BaDev& d = ba->d;
    // ---- the caller's flat arrays -> HBM (hipMemcpyDefault: the arrays of ccm_ba_problem may live in host OR device memory; staging plain host arrays
    // through a pinned block first was measured and is slower than the runtime's own pageable path: 3.1 vs 2.5 ms on the 4-agent map).  Index arrays first:
    // the structure kernels queue behind them while the host stages the big ones ----
    int *r_ecam = nullptr, *r_ept = nullptr; uint8_t *r_lvl = nullptr, *r_fixed = nullptr;
    double *r_obs = nullptr, *r_info = nullptr, *dK = nullptr;
    BB_RC(tmp.get((size_t)n_edge, &r_ecam)); BB_RC(tmp.get((size_t)n_edge, &r_ept));
    if (P->e_level) BB_RC(tmp.get((size_t)n_edge, &r_lvl));
    BB_RC(tmp.get((size_t)n_cam, &r_fixed));
    BB_RC(tmp.get(2 * (size_t)n_edge, &r_obs)); BB_RC(tmp.get((size_t)n_edge, &r_info));
    BB_RC(keep_get(ba, 4 * (size_t)n_cam, &dK));
    BB_RC(keep_get(ba, 7 * (size_t)n_cam, &ba->d_raw_cam)); BB_RC(keep_get(ba, 3 * (size_t)std::max(n_pt, 1), &ba->d_raw_pt));
    if (n_edge) {
      BB_HIP(hipMemcpyAsync(r_ecam, P->e_cam, (size_t)n_edge * sizeof(int), hipMemcpyDefault, st));
      BB_HIP(hipMemcpyAsync(r_ept, P->e_pt, (size_t)n_edge * sizeof(int), hipMemcpyDefault, st));
      if (r_lvl) BB_HIP(hipMemcpyAsync(r_lvl, P->e_level, (size_t)n_edge, hipMemcpyDefault, st));
    }
    BB_HIP(hipMemcpyAsync(r_fixed, P->cam_fixed, (size_t)n_cam, hipMemcpyDefault, st));
    lap("H2D index arrays");
    // ---- active set (initializeOptimization(0), sparse_optimizer.cpp:199-267) and vertex slots ----
    BuildSizes* sz = nullptr;
    BB_RC(tmp.get(1, &sz));
    BB_HIP(hipMemsetAsync(sz, 0, sizeof(BuildSizes), st));
    int *cam_has = nullptr, *pt_has = nullptr, *flag_c = nullptr, *flag_p = nullptr, *pre_c = nullptr, *pre_p = nullptr, *cam_slot = nullptr, *pt_slot = nullptr;
    BB_RC(tmp.get((size_t)n_cam, &cam_has)); BB_RC(tmp.get((size_t)std::max(n_pt, 1), &pt_has));
    BB_RC(tmp.get((size_t)n_cam + 1, &flag_c)); BB_RC(tmp.get((size_t)n_pt + 1, &flag_p));
    BB_RC(tmp.get((size_t)n_cam + 1, &pre_c)); BB_RC(tmp.get((size_t)n_pt + 1, &pre_p));
    BB_RC(tmp.get((size_t)n_cam, &cam_slot)); BB_RC(tmp.get((size_t)std::max(n_pt, 1), &pt_slot));
    int *d_slot_cam = nullptr;
    BB_RC(keep_get(ba, (size_t)n_cam, &d_slot_cam)); BB_RC(keep_get(ba, (size_t)std::max(n_pt, 1), &ba->d_slot_pt));
    BB_HIP(hipMemsetAsync(cam_has, 0, (size_t)n_cam * sizeof(int), st));
    BB_HIP(hipMemsetAsync(pt_has, 0, (size_t)std::max(n_pt, 1) * sizeof(int), st));
    if (n_edge) hipLaunchKernelGGL(bb_mark, dim3(grid_for(n_edge)), dim3(kB), 0, st, n_edge, (const int*)r_ecam, (const int*)r_ept, (const uint8_t*)r_lvl, n_cam, n_pt, cam_has, pt_has, sz);
    hipLaunchKernelGGL(bb_flags, dim3(grid_for(n_cam + 1)), dim3(kB), 0, st, n_cam, (const int*)cam_has, (const uint8_t*)r_fixed, flag_c);
    hipLaunchKernelGGL(bb_flags, dim3(grid_for(n_pt + 1)), dim3(kB), 0, st, n_pt, (const int*)pt_has, (const uint8_t*)nullptr, flag_p);
    BB_RC(scan_excl(ctx, tmp, flag_c, pre_c, (size_t)n_cam + 1));
    BB_RC(scan_excl(ctx, tmp, flag_p, pre_p, (size_t)n_pt + 1));
    hipLaunchKernelGGL(bb_slots, dim3(grid_for(n_cam + 1)), dim3(kB), 0, st, n_cam, (const int*)flag_c, (const int*)pre_c, cam_slot, d_slot_cam, &sz->Cp);
    hipLaunchKernelGGL(bb_slots, dim3(grid_for(n_pt + 1)), dim3(kB), 0, st, n_pt, (const int*)flag_p, (const int*)pre_p, pt_slot, ba->d_slot_pt, &sz->Lp);
    // ---- edges sorted by (landmark slot, pose slot), fixed cameras (slot -1) first, stable; inactive edges last ----
    const unsigned cb = bits_for((uint64_t)n_cam + 1), pb = bits_for((uint64_t)n_pt + 1);
    unsigned long long* keys = nullptr; int* vals = nullptr;
    BB_RC(tmp.get((size_t)n_edge, &keys)); BB_RC(tmp.get((size_t)n_edge, &vals));
    unsigned long long* keys_s = keys; int* vals_s = vals;
    int *g_pt_off = nullptr, *cslot_g = nullptr;
    BB_RC(tmp.get((size_t)n_pt + 2, &g_pt_off)); BB_RC(tmp.get((size_t)n_edge, &cslot_g));
    BB_HIP(hipMemsetAsync(g_pt_off, 0, sizeof(int), st));
    if (n_edge) {
      hipLaunchKernelGGL(bb_keys, dim3(grid_for(n_edge)), dim3(kB), 0, st, n_edge, (const int*)r_ecam, (const int*)r_ept, (const uint8_t*)r_lvl, n_cam, n_pt, (const int*)cam_slot,
                         (const int*)pt_slot, cb, pb, keys, vals);
      BB_RC(sort_pairs(ctx, tmp, keys, vals, (size_t)n_edge, pb + cb, &keys_s, &vals_s));
      hipLaunchKernelGGL(bb_ptoff, dim3(grid_for(n_edge)), dim3(kB), 0, st, n_edge, (const unsigned long long*)keys_s, cb, pb, g_pt_off, cslot_g, sz);
    }
    BB_HIP(hipGetLastError());
    // ---- the big arrays (the host stages them while the kernels above run) ----
    if (n_edge) {
      BB_HIP(hipMemcpyAsync(r_obs, P->e_obs, 2 * (size_t)n_edge * sizeof(double), hipMemcpyDefault, st));
      BB_HIP(hipMemcpyAsync(r_info, P->e_info, (size_t)n_edge * sizeof(double), hipMemcpyDefault, st));
    }
    BB_HIP(hipMemcpyAsync(dK, P->cam_K, 4 * (size_t)n_cam * sizeof(double), hipMemcpyDefault, st));
    BB_HIP(hipMemcpyAsync(ba->d_raw_cam, P->cam_qt, 7 * (size_t)n_cam * sizeof(double), hipMemcpyDefault, st));
    if (n_pt) BB_HIP(hipMemcpyAsync(ba->d_raw_pt, P->pt_xyz, 3 * (size_t)n_pt * sizeof(double), hipMemcpyDefault, st));
    d.K = dK;
    // ---- sizes (first read-back) ----
    BuildSizes hs;
    BB_HIP(hipMemcpyAsync(&hs, sz, sizeof(hs), hipMemcpyDeviceToHost, st));
    BB_HIP(hipStreamSynchronize(st));
    if (hs.bad) return ccm_set_error(ctx, CCM_E_ARG, "ccm_ba_create: edge index out of range");
    const int Cp = ba->Cp = hs.Cp, Lp = ba->Lp = hs.Lp;
    ba->n_act_edges = hs.Eact;
    lap("sort + H2D observations");
    ba->slot_cam.resize(Cp);
    std::vector<int> h_pt_off((size_t)Lp + 1, 0);
    if (Cp) BB_HIP(hipMemcpyAsync(ba->slot_cam.data(), d_slot_cam, (size_t)Cp * sizeof(int), hipMemcpyDeviceToHost, st));
    BB_HIP(hipMemcpyAsync(h_pt_off.data(), g_pt_off, ((size_t)Lp + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    // ---- shard the landmarks: weight = pair instances + edges ----
    std::vector<int32_t> shard(nranks + 1);
    if (nranks == 1) { shard[0] = 0; shard[1] = Lp; BB_HIP(hipStreamSynchronize(st)); }
    else {
      long long* d_w = nullptr;
      BB_RC(tmp.get((size_t)Lp, &d_w));
      std::vector<int64_t> weight(Lp);
      if (Lp) {
        hipLaunchKernelGGL(bb_weights, dim3(grid_for(Lp)), dim3(kB), 0, st, Lp, (const int*)g_pt_off, (const int*)cslot_g, d_w);
        BB_HIP(hipMemcpyAsync(weight.data(), d_w, (size_t)Lp * sizeof(int64_t), hipMemcpyDeviceToHost, st));
      }
      BB_HIP(hipStreamSynchronize(st));
      ccm_ba_partition(weight.data(), Lp, nranks, shard.data());
    }
    const int lb = ba->lp_begin = shard[rank], le = ba->lp_end = shard[rank + 1];
    const int Lloc = ba->Lloc = le - lb;
    const int eb = h_pt_off[lb], ee = h_pt_off[le];
    const int Eloc = ba->Eloc = ee - eb;
    d.n_cam = n_cam; d.Cp = Cp; d.Lloc = Lloc; d.Eloc = Eloc; d.huber = P->huber_delta;
    d.slot_cam = d_slot_cam;
    // ---- local edge arrays + per-camera edge lists ----
    int *p_pt_off = nullptr, *p_ed_cam = nullptr, *p_ed_cslot = nullptr, *p_ed_pt = nullptr, *p_cam_off = nullptr, *p_cam_edge = nullptr, *p_cam_pt = nullptr;
    double *p_obs = nullptr, *p_info = nullptr;
    BB_RC(keep_get(ba, (size_t)Lloc + 1, &p_pt_off)); BB_RC(keep_get(ba, (size_t)Eloc, &p_ed_cam)); BB_RC(keep_get(ba, (size_t)Eloc, &p_ed_cslot));
    BB_RC(keep_get(ba, (size_t)Eloc, &p_ed_pt)); BB_RC(keep_get(ba, 2 * (size_t)Eloc, &p_obs)); BB_RC(keep_get(ba, (size_t)Eloc, &p_info));
    BB_RC(keep_get(ba, (size_t)Eloc, &ba->d_loc_edge_orig));
    BB_RC(keep_get(ba, (size_t)Cp + 1, &p_cam_off)); BB_RC(keep_get(ba, (size_t)Eloc, &p_cam_edge)); BB_RC(keep_get(ba, (size_t)Eloc, &p_cam_pt));
    unsigned* ckey = nullptr; int* cval = nullptr; int* off_all = nullptr;
    BB_RC(tmp.get((size_t)Eloc, &ckey)); BB_RC(tmp.get((size_t)Eloc, &cval)); BB_RC(tmp.get((size_t)Cp + 2, &off_all));
    hipLaunchKernelGGL(bb_ptoff_local, dim3(grid_for(Lloc + 1)), dim3(kB), 0, st, Lloc + 1, (const int*)(g_pt_off + lb), eb, p_pt_off);
    unsigned* ckey_s = ckey; int* cval_s = cval;
    if (Eloc) {
      hipLaunchKernelGGL(bb_gather, dim3(grid_for(Eloc)), dim3(kB), 0, st, Eloc, eb, lb, (const unsigned long long*)keys_s, (const int*)vals_s, cb, (const int*)r_ecam,
                         (const double*)r_obs, (const double*)r_info, p_ed_cam, p_ed_cslot, p_ed_pt, p_obs, p_info, ba->d_loc_edge_orig, ckey, cval);
      BB_RC(sort_pairs(ctx, tmp, ckey, cval, (size_t)Eloc, bits_for((uint64_t)Cp + 1), &ckey_s, &cval_s));
    }
    hipLaunchKernelGGL(bb_lower_bound, dim3(grid_for(Cp + 2)), dim3(kB), 0, st, (const unsigned*)ckey_s, Eloc, Cp + 2, off_all);
    if (Eloc) hipLaunchKernelGGL(bb_cam_lists, dim3(grid_for(Eloc)), dim3(kB), 0, st, Eloc, (const int*)off_all, (const int*)cval_s, (const int*)p_ed_pt, p_cam_edge, p_cam_pt);
    double* p_cam_oi = nullptr;
    BB_RC(keep_get(ba, 4 * (size_t)Eloc, &p_cam_oi));
    // (the number of list slots, Eloc minus the edges of fixed cameras, lives on the device: the kernel bounds itself through off_all)
    if (Eloc) hipLaunchKernelGGL(bb_cam_oi_bounded, dim3(grid_for(Eloc)), dim3(kB), 0, st, Eloc, (const int*)off_all, (const int*)p_cam_edge, (const double*)p_obs, (const double*)p_info, p_cam_oi);
    d.cam_oi = p_cam_oi;
    hipLaunchKernelGGL(bb_cam_off, dim3(grid_for(Cp + 1)), dim3(kB), 0, st, Cp, (const int*)off_all, p_cam_off, sz);
    BB_HIP(hipGetLastError());
    d.pt_off = p_pt_off; d.ed_cam = p_ed_cam; d.ed_cslot = p_ed_cslot; d.ed_pt = p_ed_pt; d.obs = p_obs; d.info = p_info;
    d.cam_off = p_cam_off; d.cam_edge = p_cam_edge; d.cam_pt = p_cam_pt;
    // ---- the greedy landmark chunks (host: a sequential scan over the prefix sums read back above) — here, behind the launches of the camera lists, so that the
    // ~0.2 ms of the scan on a 150 000-landmark map pass while the device works (round 5: it ran at the end of the build, with the device idle) ----
    bool chunk_fits = true;
    for (int l = 0; l < Lloc && chunk_fits; l++) {   // chunks of consecutive landmarks with <= kTPB landmarks and <= kTPB observations
      const int o1 = h_pt_off[lb + l + 1] - eb;
      if (o1 - (h_pt_off[lb + l] - eb) > kTPB) chunk_fits = false;
      else if (o1 - (h_pt_off[lb + chunk.back()] - eb) > kTPB || l + 1 - chunk.back() > kTPB) chunk.push_back(l);
    }
    d.chunk_off = nullptr; d.n_chunk = 0;
    if (chunk_fits && Lloc) {
      chunk.push_back(Lloc);
      int* p_ch = nullptr;
      BB_RC(keep_get(ba, chunk.size(), &p_ch));
      BB_HIP(hipMemcpyAsync(p_ch, chunk.data(), chunk.size() * sizeof(int), hipMemcpyHostToDevice, st));   // (`chunk` outlives the stream's drain)
      d.chunk_off = p_ch; d.n_chunk = (int)chunk.size() - 1;
    }
    lap("local arrays + camera lists");
    // ---- global off-diagonal block structure + own pair instances (ba_structure.hip) ----
    uint32_t* d_U = nullptr; int nOff = 0;
    int *d_inst_off = nullptr, *d_inst_a = nullptr, *d_inst_c = nullptr;
    {
      int64_t n_inst = 0;
      BB_RC(ccm_ba_build_pairs(ctx, g_pt_off, cslot_g, Lp, Cp, lb, le, eb, &d_U, &nOff, &d_inst_off, &d_inst_a, &d_inst_c, &n_inst, ba->allocs));
      ba->n_inst = n_inst;
    }
    ba->nOff = d.nOff = nOff;
    d.inst_off = d_inst_off; d.inst_a = d_inst_a; d.inst_c = d_inst_c;
    lap("pair structure");
    // ---- block-CSR rows (full symmetric pattern) ----
    const int64_t n_ent = (int64_t)Cp + 2 * (int64_t)nOff;
    ba->n_row_entries = n_ent;
    int *p_rowblk = nullptr, *p_row_off = nullptr, *p_row_col = nullptr; uint32_t* p_row_blk = nullptr;
    BB_RC(keep_get(ba, (size_t)Cp + nOff, &ba->d_blk_i)); BB_RC(keep_get(ba, (size_t)Cp + nOff, &ba->d_blk_j));
    BB_RC(keep_get(ba, (size_t)Cp + 1, &p_rowblk)); BB_RC(keep_get(ba, (size_t)Cp + 1, &p_row_off));
    BB_RC(keep_get(ba, (size_t)n_ent, &p_row_col)); BB_RC(keep_get(ba, (size_t)n_ent, &p_row_blk));
    unsigned* jkey = nullptr; int* jval = nullptr; int* low_off = nullptr;
    BB_RC(tmp.get((size_t)nOff, &jkey)); BB_RC(tmp.get((size_t)nOff, &jval)); BB_RC(tmp.get((size_t)Cp + 1, &low_off));
    unsigned* jkey_s = jkey; int* jval_s = jval;
    if (Cp) hipLaunchKernelGGL(bb_diag_ij, dim3(grid_for(Cp)), dim3(kB), 0, st, Cp, ba->d_blk_i, ba->d_blk_j);
    if (nOff) {
      hipLaunchKernelGGL(bb_block_ij, dim3(grid_for(nOff)), dim3(kB), 0, st, nOff, (const uint32_t*)d_U, Cp, ba->d_blk_i, ba->d_blk_j, jkey, jval);
      BB_RC(sort_pairs(ctx, tmp, jkey, jval, (size_t)nOff, bits_for((uint64_t)Cp), &jkey_s, &jval_s));
    }
    hipLaunchKernelGGL(bb_rowblk_off, dim3(grid_for(Cp + 1)), dim3(kB), 0, st, Cp, (const uint32_t*)d_U, nOff, p_rowblk);
    hipLaunchKernelGGL(bb_lower_bound, dim3(grid_for(Cp + 1)), dim3(kB), 0, st, (const unsigned*)jkey_s, nOff, Cp + 1, low_off);
    hipLaunchKernelGGL(bb_row_off, dim3(grid_for(Cp + 1)), dim3(kB), 0, st, Cp, (const int*)p_rowblk, (const int*)low_off, p_row_off);
    hipLaunchKernelGGL(bb_row_fill, dim3(grid_for(std::max(Cp, nOff))), dim3(kB), 0, st, Cp, nOff, (const int*)ba->d_blk_i, (const int*)ba->d_blk_j, (const unsigned*)jkey_s,
                       (const int*)jval_s, (const int*)p_rowblk, (const int*)low_off, (const int*)p_row_off, p_row_col, p_row_blk);
    BB_HIP(hipGetLastError());
    d.rowblk_off = p_rowblk; d.row_off = p_row_off; d.row_col = p_row_col; d.row_blk = p_row_blk;
    // ---- persistent PCG: per unit the ascending list of distinct columns + every entry's position in it; cluster entry lists ----
    // (round 6) windows of 17..50 free cameras are solved exactly by ba_solve_cholreg (ba.hip), which needs the block-CSR rows only: none of the persistent solver's
    // lists, no cluster entry lists, no coarse level.  CCM_BA_CHOLREG=0 (read here and in lm_trial) keeps the earlier solvers and their structures.
    const char* cholreg_env = getenv("CCM_BA_CHOLREG");
    const bool cholreg_win = Cp > kSmallMaxCp && Cp <= kCholRegMaxCp && !(cholreg_env && cholreg_env[0] == '0');
    const bool pers_wanted = Cp > kSmallMaxCp && !cholreg_win;
    const int n_cl = ccm_div_up(std::max(Cp, 1), kClu);
    const int pers_grid_want = ((2 * n_cl + 7) / 8) * 8;
    const bool pers_try = pers_wanted && !getenv("CCM_BA_NO_PERSIST") && pers_grid_want <= 4 * kWave && ccm_ba_pers_grid_fits(ctx, pers_grid_want);
    int *ucnt = nullptr, *ccnt = nullptr;
    const int n_pu = 2 * n_cl, n_words = ccm_div_up(std::max(Cp, 1), 32);
    if (pers_try) {
      BB_RC(tmp.get((size_t)n_pu + 1, &ucnt));
      BB_RC(keep_get(ba, (size_t)n_pu + 1, &ba->d_pers_uoff)); BB_RC(keep_get(ba, (size_t)n_ent, &ba->d_pers_ucol)); BB_RC(keep_get(ba, (size_t)n_ent, &ba->d_pers_loc));
      BB_HIP(hipMemsetAsync(ucnt + n_pu, 0, sizeof(int), st));
      hipLaunchKernelGGL(bb_pers_units<0>, dim3(n_pu), dim3(kB), 2 * (size_t)n_words * sizeof(unsigned), st, Cp, n_words, (const int*)p_row_off, (const int*)p_row_col, ucnt,
                         (const int*)nullptr, (int*)nullptr, (int*)nullptr, sz);
      BB_RC(scan_excl(ctx, tmp, ucnt, ba->d_pers_uoff, (size_t)n_pu + 1));
      hipLaunchKernelGGL(bb_pers_units<1>, dim3(n_pu), dim3(kB), 2 * (size_t)n_words * sizeof(unsigned), st, Cp, n_words, (const int*)p_row_off, (const int*)p_row_col, (int*)nullptr,
                         (const int*)ba->d_pers_uoff, ba->d_pers_ucol, ba->d_pers_loc, sz);
    }
    if (pers_wanted) {
      BB_RC(tmp.get((size_t)n_cl + 1, &ccnt));
      BB_RC(keep_get(ba, (size_t)n_cl + 1, &ba->d_pers_coff)); BB_RC(keep_get(ba, (size_t)n_cl * kClu * kClu, &ba->d_pers_cij)); BB_RC(keep_get(ba, (size_t)n_cl * kClu * kClu, &ba->d_pers_cblk));
      BB_HIP(hipMemsetAsync(ccnt + n_cl, 0, sizeof(int), st));
      hipLaunchKernelGGL(bb_cluster_entries<0>, dim3(n_cl), dim3(kB), 0, st, Cp, (const int*)p_row_off, (const int*)p_row_col, (const uint32_t*)p_row_blk, ccnt, (const int*)nullptr,
                         (int*)nullptr, (uint32_t*)nullptr);
      BB_RC(scan_excl(ctx, tmp, ccnt, ba->d_pers_coff, (size_t)n_cl + 1));
      hipLaunchKernelGGL(bb_cluster_entries<1>, dim3(n_cl), dim3(kB), 0, st, Cp, (const int*)p_row_off, (const int*)p_row_col, (const uint32_t*)p_row_blk, (int*)nullptr,
                         (const int*)ba->d_pers_coff, ba->d_pers_cij, ba->d_pers_cblk);
    }
    BB_HIP(hipGetLastError());
    // ---- row Schur kernel: unit offsets (the table itself needs the sizes read back below) ----
    int *nub = nullptr, *blk_pref = nullptr, *p_row_u = nullptr, *p_blk_u = nullptr;
    BB_RC(tmp.get((size_t)nOff + 1, &nub)); BB_RC(tmp.get((size_t)nOff + 1, &blk_pref));
    BB_RC(keep_get(ba, (size_t)Cp + 1, &p_row_u)); BB_RC(keep_get(ba, (size_t)nOff + 1, &p_blk_u));
    const int chunk_env = 0;   // experiments: 16 / 32 / 64 / 128
    const int unit_chunk = d.unit_chunk = chunk_env >= kRow2Group ? chunk_env : kRow2Chunk;
    hipLaunchKernelGGL(bb_unit_counts, dim3(grid_for(nOff + 1)), dim3(kB), 0, st, nOff, unit_chunk, (const int*)d_inst_off, nub);
    BB_RC(scan_excl(ctx, tmp, nub, blk_pref, (size_t)nOff + 1));
    hipLaunchKernelGGL(bb_unit_offsets, dim3(grid_for(std::max(nOff, Cp) + 1)), dim3(kB), 0, st, nOff, Cp, (const int*)blk_pref, (const int*)p_rowblk, p_row_u, p_blk_u, sz);
    // ---- rank of every observation inside its camera's list / of every pair instance's row-side observation ----
    int *p_rank = nullptr, *p_al = nullptr;
    BB_RC(tmp.get((size_t)Eloc, &p_rank)); BB_RC(keep_get(ba, (size_t)std::max<int64_t>(ba->n_inst, 1), &p_al));
    if (Cp) hipLaunchKernelGGL(bb_edge_rank, dim3(Cp), dim3(kB), 0, st, (const int*)p_cam_off, (const int*)p_cam_edge, p_rank);
    if (ba->n_inst) hipLaunchKernelGGL(bb_inst_rank, dim3(grid_for(ba->n_inst)), dim3(kB), 0, st, (const int*)d_inst_a, (const int*)p_rank, (int)ba->n_inst, p_al);
    d.inst_al = p_al;
    // ---- compact observation records of the row kernel (ba_schur_row3): camera-major positions ----
    d.E4 = nullptr; d.camRK = nullptr; d.inst_cp = nullptr; d.blk_j = ba->d_blk_j + Cp;
    if (Cp && Eloc) {
      int *p_cpos = nullptr, *p_icp = nullptr; double *p_e4 = nullptr, *p_rk = nullptr;
      BB_RC(tmp.get((size_t)Eloc, &p_cpos)); BB_RC(keep_get(ba, (size_t)std::max<int64_t>(ba->n_inst, 1), &p_icp));
      BB_RC(keep_get(ba, 4 * (size_t)Eloc, &p_e4)); BB_RC(keep_get(ba, 12 * (size_t)Cp, &p_rk));   // (written by the linearisation before any kernel reads them)
      BB_HIP(hipMemsetAsync(p_cpos, 0xFF, (size_t)Eloc * sizeof(int), st));
      hipLaunchKernelGGL(bb_edge_cpos, dim3(Cp), dim3(kB), 0, st, (const int*)p_cam_off, (const int*)p_cam_edge, p_cpos);
      if (ba->n_inst) hipLaunchKernelGGL(bb_inst_cpos, dim3(grid_for(ba->n_inst)), dim3(kB), 0, st, (const int*)d_inst_c, (const int*)p_cpos, (int)ba->n_inst, p_icp);
      d.E4 = p_e4; d.camRK = p_rk; d.inst_cp = p_icp;
    }
    // ---- coarse level: block lists of Ac = P^T S P by aggregate pair ----
    // intervals of 16 cameras wherever a unit of the persistent solver can keep its 12 rows of Ac^-1 (f32) in the free half of the LDS region of W
    // (12 Nc floats <= 96 * 48 doubles: Nc <= 768, i.e. up to 2032 free cameras) — measured on a 1000-keyframe map with the coarse level always on:
    // 1351 CG iterations per call with intervals of 32, 901 with 16 — and 32 otherwise and on the multi-kernel path (whose kernels pair two clusters)
    auto coarse_size = [&](int agg, int* na_out, int* Nc_out) { *na_out = ccm_div_up(std::max(Cp, 1), agg); *Nc_out = ((6 * (*na_out + 1) + 63) / 64) * 64; };
    auto pers_fits = [&](int agg, int na_, int Nc_) { return 6 * (size_t)Nc_ <= (size_t)kCluN * kCluN / 2 && (agg / 8) * na_ <= pers_grid_want + agg / 8 - 1; };
    int agg = kAggWide, na = 0, Nc = 0;
    static const int agg_env = getenv("CCM_BA_COARSE_AGG") ? atoi(getenv("CCM_BA_COARSE_AGG")) : 0;        // experiments: 16 / 24 / 32
    const int nc_cap = kCoarseNcCap;
    if (pers_try)
      for (int cand : {kAggFine, kAggMid}) {
        if (agg_env ? cand != agg_env : false) continue;
        coarse_size(cand, &na, &Nc);
        if ((agg_env || Nc <= nc_cap) && pers_fits(cand, na, Nc)) { agg = cand; break; }
      }
    coarse_size(agg, &na, &Nc);
    d.agg = agg;
    // (windows of up to 32 free cameras are solved exactly by ba_solve_dense2 and never see a coarse level: its block lists — a sort, a run-length encode and
    // half a dozen short kernels of a launch-bound 0.7 ms build — are skipped there unless that solver is switched off)
    const bool dense2_off = false;
    const bool coarse_pers = pers_try && pers_fits(agg, na, Nc) && (Cp > kDense2MaxCp || dense2_off);
    const bool coarse_mk = pers_wanted && agg == 2 * kClu && Nc <= 6144;   // multi-kernel PCG (when the persistent kernel is not usable); three Nc^2 f64 buffers: <= 0.9 GB
    unsigned* uq = nullptr; unsigned* cnts = nullptr; int* n_runs = nullptr;
    if (coarse_pers || coarse_mk) {
      const size_t nk = (size_t)Cp + nOff;
      unsigned* ck = nullptr; int* cv = nullptr;
      BB_RC(tmp.get(nk, &ck)); BB_RC(tmp.get(nk, &cv)); BB_RC(tmp.get(nk, &uq)); BB_RC(tmp.get(nk + 1, &cnts)); BB_RC(tmp.get(1, &n_runs));
      hipLaunchKernelGGL(bb_coarse_keys, dim3(grid_for((int64_t)nk)), dim3(kB), 0, st, Cp, nOff, (const int*)ba->d_blk_i, (const int*)ba->d_blk_j, na, agg, ck, cv);
      unsigned* ck_s = ck; int* cv_s = cv;
      BB_RC(sort_pairs(ctx, tmp, ck, cv, nk, bits_for((uint64_t)na * na), &ck_s, &cv_s));
      BB_RC(keep_get(ba, nk, &ba->d_cb_ent));
      BB_HIP(hipMemcpyAsync(ba->d_cb_ent, cv_s, nk * sizeof(int), hipMemcpyDeviceToDevice, st));
      size_t bytes = 0;
      BB_HIP(rocprim::run_length_encode(nullptr, bytes, ck_s, (unsigned)nk, uq, cnts, n_runs, st));
      char* scratch = nullptr;
      BB_RC(tmp.get(bytes, &scratch));
      BB_HIP(rocprim::run_length_encode(scratch, bytes, ck_s, (unsigned)nk, uq, cnts, n_runs, st));
      const size_t max_runs = std::min<size_t>(nk, (size_t)na * (na + 1) / 2);
      BB_RC(keep_get(ba, max_runs + 1, &ba->d_cb_off)); BB_RC(keep_get(ba, 2 * max_runs, &ba->d_cb_ab));
      BB_RC(keep_get(ba, max_runs, &ba->d_cb_key)); BB_RC(keep_get(ba, max_runs * 4 * 36, &ba->d_cstage));
      BB_HIP(hipMemcpyAsync(ba->d_cb_key, uq, max_runs * sizeof(unsigned), hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(bb_counts_tail, dim3(1), dim3(1), 0, st, cnts, (const int*)n_runs);
      BB_RC(scan_excl(ctx, tmp, (const int*)cnts, ba->d_cb_off, max_runs + 1));
      hipLaunchKernelGGL(bb_coarse_ab, dim3(grid_for((int64_t)max_runs)), dim3(kB), 0, st, (const unsigned*)uq, (const int*)n_runs, na, ba->d_cb_ab, sz);
    }
    BB_HIP(hipGetLastError());
    // ---- sizes (second read-back) ----
    BB_HIP(hipMemcpyAsync(&hs, sz, sizeof(hs), hipMemcpyDeviceToHost, st));
    BB_HIP(hipStreamSynchronize(st));
    lap("rows, lists, unit offsets");
    d.max_cam_edges = hs.max_cam_edges;
    // ---- row Schur kernel: the unit table, when every row's partial sums fit the LDS beside its Y ----
    d.unit_tab = nullptr; d.row_unit_off = nullptr; d.blk_unit0 = nullptr; d.row_units_max = 0; d.row_dbg = nullptr;
    if (ccm_dbg("row")) { long long* p_dbg = nullptr; BB_RC(keep_get(ba, 8, &p_dbg, true)); d.row_dbg = p_dbg; }
    if (nOff > row_min_blocks() && d.max_cam_edges <= kRowMaxEdges && (uint64_t)std::max(Eloc, 1) * 144u < (1ull << 31)) {
      // LDS of a row: Y of its observations + a zero row, the diagonal partials (27 per 16 observations), 36 doubles per unit
      const size_t lds_free = 158 * 1024 - ((size_t)(d.max_cam_edges + 1) * 18 + 27 * (size_t)ccm_div_up(d.max_cam_edges, kRow2Group)) * sizeof(double);
      const int units_cap = (int)(lds_free / (36 * sizeof(double)));
      if (hs.worst_units <= units_cap) {
        int4* p_tab = nullptr;
        BB_RC(keep_get(ba, (size_t)hs.n_units, &p_tab));
        hipLaunchKernelGGL(bb_unit_fill, dim3(Cp), dim3(kB), 3 * (size_t)std::max(hs.worst_units, 1) * sizeof(int), st, Cp, unit_chunk, (const int*)p_rowblk,
                           (const int*)d_inst_off, (const int*)p_row_u, (const int*)p_blk_u, p_tab);
        BB_HIP(hipGetLastError());
        d.unit_tab = p_tab; d.row_unit_off = p_row_u; d.blk_unit0 = p_blk_u; d.row_units_max = std::max(hs.worst_units, 1);
      }
    }
    // ---- solver buffers ----
#define AL(field, n, T) { T* _p = nullptr; BB_RC(keep_get<T>(ba, (n), &_p, true)); d.field = _p; }
    AL(cam[0], 7 * (size_t)n_cam, double) AL(cam[1], 7 * (size_t)n_cam, double)
    AL(pt[0], 3 * (size_t)Lloc, double) AL(pt[1], 3 * (size_t)Lloc, double)
    // the Hpl blocks (144 bytes per observation) exist only where a kernel reads them: with the row kernel on compact records and the edge-parallel
    // landmark kernels every consumer re-derives them (32 bytes per observation in landmark-major order for the back-substitution)
    d.E4L = nullptr;
    d.w_free = (d.E4 && d.chunk_off && d.unit_tab && nOff > row_min_blocks()) ? 1 : 0;
    if (d.w_free) { double* p_l = nullptr; BB_RC(keep_get(ba, 4 * (size_t)Eloc, &p_l)); d.E4L = p_l; }
    AL(W, d.w_free ? 1 : 18 * (size_t)Eloc, double) AL(Hll, 6 * (size_t)Lloc, double) AL(bl, 3 * (size_t)Lloc, double)
    AL(Dinv, 6 * (size_t)Lloc, double) AL(dl, 3 * (size_t)Lloc, double) AL(Hpp, 36 * (size_t)Cp, double) AL(bp, 6 * (size_t)Cp, double)
    AL(x, 6 * (size_t)Cp, double) AL(r, 6 * (size_t)Cp, double) AL(z, 6 * (size_t)Cp, double) AL(q, 6 * (size_t)Cp, double)
    AL(p[0], 6 * (size_t)Cp, double) AL(p[1], 6 * (size_t)Cp, double)
    AL(Wc, (size_t)n_cl * kCluN * kCluN, float)
    d.n_wg_spmv = ((ccm_div_up(std::max(Cp, 1), kRowsPerWG) + 7) / 8) * 8;   // padded to 8 (one chunk per XCD)
    d.n_wg_wave4 = ccm_div_up(std::max(Cp, 1), kTPB / kWave);
    d.n_wg_upd = n_cl;   // one workgroup per preconditioner cluster
    d.n_wg_pt = ccm_div_up(std::max(Lloc, 1), kTPB); d.n_wg_cam = ccm_div_up(std::max(Cp, 1), kTPB);
    d.n_part = d.chunk_off ? d.n_chunk : d.n_wg_pt;
    AL(ppq, d.n_wg_spmv, double) AL(prz[0], d.n_wg_upd, double) AL(prz[1], d.n_wg_upd, double)
    AL(pcg_scal, 4, double)
    AL(edge_chi2, Eloc, double) AL(edge_depth, Eloc, uint8_t)
    AL(part_pt, 2 * (size_t)std::max(d.n_wg_pt, d.n_chunk), double) AL(part_cam, d.n_wg_cam, double) AL(scal, 8, double)
#undef AL
    d.pcg_flag = reinterpret_cast<int*>(d.scal + 6);   // [scalars | PCG flags]: one 64-byte read-back per LM trial
    if (!ctx->rb_free.empty()) { ba->h_rb = static_cast<double*>(ctx->rb_free.back()); ctx->rb_free.pop_back(); }
    else if (hipHostMalloc(&ba->h_rb, 128, hipHostMallocCoherent /* polled by the host while the kernel is in flight (read_scalars_polled): fine-grained, explicitly */) != hipSuccess) return ccm_set_error(ctx, CCM_E_HIP, "ccm_ba_create: pinned read-back buffer");
    memset(ba->h_rb, 0, 128);   // (a block that served another handle holds that handle's last ticket)
    ba->red_count = 36 * (size_t)(Cp + nOff) + 6 * (size_t)Cp;
    BB_RC(keep_get(ba, ba->red_count, &ba->d_red, true));
    d.S = ba->d_red; d.bs = ba->d_red + 36 * (size_t)(Cp + nOff);
    BB_RC(keep_get(ba, 3 * (size_t)std::max(Lp, 1), &ba->d_pt_full, true));
    BB_RC(keep_get(ba, 36 * (size_t)std::max(Cp, 1), &ba->d_hpp_full, true));
    auto coarse_buffers = [&](int n_units) -> int {
      BB_RC(keep_get(ba, 36 * (size_t)Cp, &ba->d_cP, true));
      BB_RC(keep_get(ba, (size_t)Nc * Nc, &ba->d_cA, true)); BB_RC(keep_get(ba, (size_t)Nc * Nc, &ba->d_cX, true));
      BB_RC(keep_get(ba, (size_t)Nc * Nc, &ba->d_cAinv, true)); BB_RC(keep_get(ba, (size_t)Nc * 64, &ba->d_cLinv, true));
      BB_RC(keep_get(ba, 4, &ba->d_cinfo, true));
      BB_RC(keep_get(ba, 12 * (size_t)std::max(n_units, 1), &ba->d_cparts, true));
      ba->coarse_na = na; ba->coarse_Nc = Nc; ba->coarse_ncb = hs.ncb;
      if (const char* cf = getenv("CCM_BA_COARSE")) ba->coarse_force = !strcmp(cf, "always") ? 1 : !strcmp(cf, "never") ? -1 : 0;
      return CCM_OK;
    };
    // persistent single-launch PCG: usable when all workgroups (two per cluster) can be co-resident on the device and every unit's lists fit its LDS
    ba->pers_grid = 0;
    d.mk_cpart = nullptr; d.mk_cry[0] = d.mk_cry[1] = nullptr; d.mk_P = nullptr; d.mk_Ainv = nullptr; d.mk_Ainv32 = nullptr; d.mk_on = 0; d.mk_Nc = 0; d.mk_na = 0;
    if (pers_try && !hs.pers_bad) {
      BB_RC(keep_get(ba, 4 + 2 * 16 + 4 * 512, &ba->d_pers_bar, true));   // abort flag + debug clocks (workgroup 0's phases; then per workgroup the time spent in the two exchanges)
      BB_RC(keep_get(ba, 4 * (size_t)pers_grid_want, &ba->d_pers_part, true));   // [2][2][grid] slot words
      ba->pers_grid = pers_grid_want; ba->pers_grid_built = pers_grid_want;
      BB_RC(keep_get(ba, (size_t)pers_grid_want * (kCluN * (kCluN / 2)), &ba->d_pers_wsave, false));   // the units' halves of the cluster inverse, carried from trial to trial (written before read)
      if (coarse_pers) BB_RC(coarse_buffers(pers_grid_want));
    } else if (pers_try) { ba->d_pers_uoff = nullptr; ba->d_pers_ucol = nullptr; ba->d_pers_loc = nullptr; }
    if (Cp > kSmallMaxCp && Cp <= kDense2MaxCp && ba->d_pers_coff)
      BB_RC(keep_get(ba, (size_t)kCluN * kCluN + 16, &ba->d_dense_T, true));   // + phase clocks (CCM_BA_DENSE2_DBG)
    if (cholreg_win) {   // register-resident Cholesky solve: phase clocks [16]; the factor's scratch [n_pad][n_pad] (written before read)
      const size_t np = 16 * (size_t)ccm_div_up(6 * Cp, 16);
      BB_RC(keep_get(ba, 16, &ba->d_cholreg_dbg, true));
      BB_RC(keep_get(ba, np * np, &ba->d_cholreg_L, false));
      BB_RC(keep_get(ba, (size_t)8 * 24 * 64 * 4, &ba->d_cholreg_tab, false));
    }
    // maps too large for the persistent kernel: the same coarse level inside the multi-kernel PCG (kAgg = 2 clusters)
    if (!ba->pers_grid && coarse_mk) {
      BB_RC(coarse_buffers(0));
      double *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
      BB_RC(keep_get(ba, 24 * (size_t)(na + 1), &p1, true)); BB_RC(keep_get(ba, (size_t)n_cl, &p2, true)); BB_RC(keep_get(ba, (size_t)n_cl, &p3, true));
      d.mk_cpart = p1; d.mk_cry[0] = p2; d.mk_cry[1] = p3;
      d.mk_P = ba->d_cP; d.mk_Ainv = ba->d_cAinv; d.mk_Nc = Nc; d.mk_na = na;
      BB_RC(keep_get(ba, (size_t)Nc * Nc, &ba->d_cAinv32, false)); d.mk_Ainv32 = ba->d_cAinv32;
    }
    d.S32 = nullptr; d.qx = nullptr;
    if (!ba->pers_grid && pers_wanted && pers_grid_want > 4 * kWave) {   // the maps that are too large for the persistent kernel (not its fallback on smaller ones)
      float* p32 = nullptr;
      BB_RC(keep_get(ba, 36 * (size_t)(Cp + nOff), &p32, false));
      d.S32 = p32;
      double* pqx = nullptr;
      BB_RC(keep_get(ba, 6 * (size_t)Cp, &pqx, false));
      d.qx = pqx;
    }
    BB_RC(flush_zero_list(ba));   // (no kernel above reads a buffer it asked to have zeroed)
    BB_RC(ccm_ba_state_from_raw(ba));
    BB_HIP(hipStreamSynchronize(st));
    lap("unit table, buffers, state");
    return CCM_OK;
  };
  rc_build = body();
  }   // ~Tmp: stream drained, temporaries back in the pool
  if (rc_build != CCM_OK) { ccm_ba_destroy(ba); return rc_build; }
  ba->ms_setup = now_ms() - t0;
  *out = ba;
  return CCM_OK;
}
