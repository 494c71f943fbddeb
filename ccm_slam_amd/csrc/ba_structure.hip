// ba_structure.hip — the pair structure of the Schur complement, built on the device.
// buildStructure (block_solver.hpp:143-295) needs, for every pair of free cameras that share a landmark, one Hschur block
// and the list of (edge_a, edge_c) observation pairs that contribute to it.  On the 4-agent merged map that is 3.1 M
// pairs in 64 k blocks; building it on the host (generate, two counting-sort passes, run-length split) was 55 % of
// ccm_ba_create.  Here: one thread per landmark counts and emits its pairs (key = ia*Cp + ic, value = both edge positions),
// a stable LSD radix sort groups them by block while keeping the landmark order inside a block (the summation order
// of the gather kernel, hence bit-identical results), run-length encoding yields the block list.  A sharded rank sorts
// the keys of ALL landmarks for the (global, rank-independent) block list and the pairs of its OWN landmarks for the
// instance lists, and finds its per-block ranges by binary search.  rocPRIM supplies scan / sort / run-length encode.
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>

namespace {

constexpr int kTPB = 256;

// edges of a landmark are sorted by pose slot, fixed cameras (slot -1) first; a pair needs two different free cameras
template <int EMIT>
__global__ void pairs_kernel(const int* pt_off, const int* cslot, int l0, int n_l, int Cp, int* cnt, const int* poff, uint32_t* keys,
                             unsigned long long* vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_l) return;
  const int l = l0 + i;
  const int k0 = pt_off[l], k1 = pt_off[l + 1];
  int n = 0;
  int w = EMIT ? poff[i] : 0;
  for (int a = k0; a < k1; a++) {
    const int ia = cslot[a];
    if (ia < 0) continue;
    for (int c = a + 1; c < k1; c++) {
      const int ic = cslot[c];
      if (ic == ia) continue;
      if (EMIT) { keys[w] = (uint32_t)ia * (uint32_t)Cp + (uint32_t)ic; if (vals) vals[w] = ((unsigned long long)(uint32_t)a << 32) | (uint32_t)c; w++; }
      n++;
    }
  }
  if (!EMIT) cnt[i] = n;
}

__global__ void split_kernel(const unsigned long long* vals, int n, int eb, int* inst_a, int* inst_c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  inst_a[i] = (int)(vals[i] >> 32) - eb;
  inst_c[i] = (int)(uint32_t)vals[i] - eb;
}

// inst_off[b] = first own pair with key >= U[b]; inst_off[nOff] = n_own
__global__ void ranges_kernel(const uint32_t* U, int nOff, const uint32_t* own_keys, int n_own, int* inst_off) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nOff) return;
  if (b == nOff) { inst_off[b] = n_own; return; }
  const uint32_t key = U[b];
  int lo = 0, hi = n_own;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (own_keys[mid] < key) lo = mid + 1; else hi = mid; }
  inst_off[b] = lo;
}

struct Tmp {
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

#define ST_RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
#define ST_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return ccm_set_error(ctx, CCM_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// counts + emits the pairs of landmarks [l0, l1) ; returns device keys (and values) plus the number of pairs
int emit_pairs(ccm_ctx* ctx, Tmp& tmp, const int* d_pt_off, const int* d_cslot, int l0, int l1, int Cp, bool with_vals, uint32_t** keys,
               unsigned long long** vals, int* n_out) {
  const int n_l = l1 - l0;
  *keys = nullptr; if (vals) *vals = nullptr; *n_out = 0;
  if (n_l <= 0) return CCM_OK;
  int *cnt = nullptr, *poff = nullptr;
  ST_RC(tmp.get((size_t)n_l + 1, &cnt)); ST_RC(tmp.get((size_t)n_l + 1, &poff));
  ST_HIP(hipMemsetAsync(cnt + n_l, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(pairs_kernel<0>, dim3(ccm_div_up(n_l, kTPB)), dim3(kTPB), 0, ctx->stream, d_pt_off, d_cslot, l0, n_l, Cp, cnt, (const int*)nullptr,
                     (uint32_t*)nullptr, (unsigned long long*)nullptr);
  size_t bytes = 0;
  ST_HIP(rocprim::exclusive_scan(nullptr, bytes, cnt, poff, 0, (size_t)n_l + 1, rocprim::plus<int>(), ctx->stream));
  char* scratch = nullptr;
  ST_RC(tmp.get(bytes, &scratch));
  ST_HIP(rocprim::exclusive_scan(scratch, bytes, cnt, poff, 0, (size_t)n_l + 1, rocprim::plus<int>(), ctx->stream));
  int total = 0;
  ST_HIP(hipMemcpyAsync(&total, poff + n_l, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ST_HIP(hipStreamSynchronize(ctx->stream));
  *n_out = total;
  if (total == 0) return CCM_OK;
  ST_RC(tmp.get((size_t)total, keys));
  if (with_vals) ST_RC(tmp.get((size_t)total, vals));
  hipLaunchKernelGGL(pairs_kernel<1>, dim3(ccm_div_up(n_l, kTPB)), dim3(kTPB), 0, ctx->stream, d_pt_off, d_cslot, l0, n_l, Cp, (int*)nullptr, (const int*)poff,
                     *keys, with_vals ? *vals : (unsigned long long*)nullptr);
  ST_HIP(hipGetLastError());
  return CCM_OK;
}

}  // namespace

// d_pt_off [Lp+1] / d_cslot [E] (device): landmark-sorted edge ranges and the pose slot (-1 = fixed) of every edge position.
// Own landmarks [lb, le), eb = first own edge position.  Outputs (device, blocks appended to `keep`, owned by the caller): the global block
// list d_U [nOff] (key = ia * Cp + ic, ascending), inst_off [nOff+1] / inst_a / inst_c [n_inst].  Two host syncs (pair count, block count).
int ccm_ba_build_pairs(ccm_ctx* ctx, const int* d_pt_off, const int* d_cslot, int Lp, int Cp, int lb, int le, int eb,
                       uint32_t** d_U_out, int* nOff_out, int** d_inst_off, int** d_inst_a, int** d_inst_c, int64_t* n_inst,
                       std::vector<std::pair<void*, size_t>>& keep) {
  if (Cp > 65535) return ccm_set_error(ctx, CCM_E_ARG, "bundle adjustment: more than 65535 free cameras (32-bit block keys)");
  *n_inst = 0; *nOff_out = 0; *d_U_out = nullptr;
  auto keep_get = [&](size_t n, int** out) -> int {
    void* p = nullptr; size_t actual = 0;
    if (int rc = ccm_pool_get(ctx, std::max<size_t>(n, 1) * sizeof(int), &p, &actual)) return rc;
    keep.push_back({p, actual}); *out = (int*)p; return CCM_OK;
  };
  Tmp tmp{ctx, {}};
  unsigned bits = 1;
  while (bits < 32 && ((uint64_t)1 << bits) < (uint64_t)Cp * (uint64_t)Cp) bits++;
  const bool whole = (lb == 0 && le == Lp);
  // ---- own pairs, sorted by block (stable: landmark order inside a block) ----
  uint32_t* ok = nullptr; unsigned long long* ov = nullptr; int n_own = 0;
  ST_RC(emit_pairs(ctx, tmp, d_pt_off, d_cslot, lb, le, Cp, true, &ok, &ov, &n_own));
  uint32_t* ok_sorted = ok; unsigned long long* ov_sorted = ov;
  if (n_own) {
    uint32_t* ok2 = nullptr; unsigned long long* ov2 = nullptr;
    ST_RC(tmp.get((size_t)n_own, &ok2)); ST_RC(tmp.get((size_t)n_own, &ov2));
    rocprim::double_buffer<uint32_t> kb(ok, ok2);
    rocprim::double_buffer<unsigned long long> vb(ov, ov2);
    size_t bytes = 0;
    ST_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kb, vb, (unsigned)n_own, 0u, bits, ctx->stream));
    char* scratch = nullptr;
    ST_RC(tmp.get(bytes, &scratch));
    ST_HIP(rocprim::radix_sort_pairs(scratch, bytes, kb, vb, (unsigned)n_own, 0u, bits, ctx->stream));
    ok_sorted = kb.current(); ov_sorted = vb.current();
  }
  // ---- global block list: keys of all landmarks (the own ones when this rank owns everything) ----
  uint32_t* gk_sorted = ok_sorted; int n_all = n_own;
  if (!whole) {
    uint32_t* gk = nullptr;
    ST_RC(emit_pairs(ctx, tmp, d_pt_off, d_cslot, 0, Lp, Cp, false, &gk, nullptr, &n_all));
    gk_sorted = gk;
    if (n_all) {
      uint32_t* gk2 = nullptr;
      ST_RC(tmp.get((size_t)n_all, &gk2));
      rocprim::double_buffer<uint32_t> kb(gk, gk2);
      size_t bytes = 0;
      ST_HIP(rocprim::radix_sort_keys(nullptr, bytes, kb, (unsigned)n_all, 0u, bits, ctx->stream));
      char* scratch = nullptr;
      ST_RC(tmp.get(bytes, &scratch));
      ST_HIP(rocprim::radix_sort_keys(scratch, bytes, kb, (unsigned)n_all, 0u, bits, ctx->stream));
      gk_sorted = kb.current();
    }
  }
  int nOff = 0;
  uint32_t* U = nullptr; unsigned* counts = nullptr; int* runs = nullptr;
  if (n_all) {
    ST_RC(tmp.get((size_t)n_all, &U)); ST_RC(tmp.get((size_t)n_all, &counts)); ST_RC(tmp.get(1, &runs));
    size_t bytes = 0;
    ST_HIP(rocprim::run_length_encode(nullptr, bytes, gk_sorted, (unsigned)n_all, U, counts, runs, ctx->stream));
    char* scratch = nullptr;
    ST_RC(tmp.get(bytes, &scratch));
    ST_HIP(rocprim::run_length_encode(scratch, bytes, gk_sorted, (unsigned)n_all, U, counts, runs, ctx->stream));
    ST_HIP(hipMemcpyAsync(&nOff, runs, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ST_HIP(hipStreamSynchronize(ctx->stream));
  }
  int* U_keep = nullptr;
  ST_RC(keep_get((size_t)nOff, &U_keep));
  if (nOff) ST_HIP(hipMemcpyAsync(U_keep, U, (size_t)nOff * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  ST_RC(keep_get((size_t)nOff + 1, d_inst_off)); ST_RC(keep_get((size_t)n_own, d_inst_a)); ST_RC(keep_get((size_t)n_own, d_inst_c));
  hipLaunchKernelGGL(ranges_kernel, dim3(ccm_div_up(nOff + 1, kTPB)), dim3(kTPB), 0, ctx->stream, (const uint32_t*)U, nOff, (const uint32_t*)ok_sorted, n_own,
                     *d_inst_off);
  if (n_own) hipLaunchKernelGGL(split_kernel, dim3(ccm_div_up(n_own, kTPB)), dim3(kTPB), 0, ctx->stream, (const unsigned long long*)ov_sorted, n_own, eb, *d_inst_a, *d_inst_c);
  ST_HIP(hipGetLastError());
  *d_U_out = (uint32_t*)U_keep;
  *nOff_out = nOff;
  *n_inst = n_own;
  return CCM_OK;   // ~Tmp waits for the stream before the temporaries go back to the pool
}
