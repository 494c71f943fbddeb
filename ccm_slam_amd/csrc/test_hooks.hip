// test_hooks.hip — TEST-ONLY entry points, built into libccm_testhooks.so (NOT into the product library libccm_hip.so, which it links against).
// They reach into handles through the internal headers; nothing in the product's timed paths knows about them.
#include "common.h"
#include "ba_types.h"
#include "test_internal.h"
#include "lane_xor.h"
#include "../../include/ccm_testhooks.h"
#include <cstring>
#include <string>

// Copies one of the structure arrays that ccm_ba_create built on the device (ba_build.hip) to the host.  *bytes receives the size of the array;
// with out == nullptr only the size is returned.  tests/test_ba_structure_gpu.py recomputes every array with numpy and compares.
extern "C" int ccm_ba_debug_array(ccm_ba* ba, const char* name, void* out, size_t cap_bytes, size_t* bytes) {
  if (!ba || !name || !bytes) return CCM_E_ARG;
  ccm_ctx* ctx = ba->ctx;
  const BaDev& d = ba->d;
  const std::string n(name);
  const void* src = nullptr;
  size_t sz = 0;
  const size_t Cp = (size_t)d.Cp, L = (size_t)d.Lloc, E = (size_t)d.Eloc, nOff = (size_t)d.nOff, ent = (size_t)ba->n_row_entries, I = (size_t)ba->n_inst;
  const size_t n_cl = (Cp + kClu - 1) / kClu;
  auto host_int = [&](const int* p, size_t idx, int* v) -> int {
    CCM_HIP_CHECK(ctx, hipMemcpy(v, p + idx, sizeof(int), hipMemcpyDeviceToHost));
    return CCM_OK;
  };
  int tail = 0;
#define ARR(nm, ptr, count, T) if (n == nm) { src = (ptr); sz = (size_t)(count) * sizeof(T); }
  ARR("slot_cam", d.slot_cam, Cp, int) ARR("slot_pt", ba->d_slot_pt, ba->Lp, int) ARR("loc_edge_orig", ba->d_loc_edge_orig, E, int)
  ARR("pt_off", d.pt_off, L + 1, int) ARR("ed_cam", d.ed_cam, E, int) ARR("ed_cslot", d.ed_cslot, E, int) ARR("ed_pt", d.ed_pt, E, int)
  ARR("obs", d.obs, 2 * E, double) ARR("info", d.info, E, double)
  ARR("cam_off", d.cam_off, Cp + 1, int) ARR("rowblk_off", d.rowblk_off, Cp + 1, int) ARR("row_off", d.row_off, Cp + 1, int)
  ARR("row_col", d.row_col, ent, int) ARR("row_blk", d.row_blk, ent, uint32_t)
  ARR("inst_off", d.inst_off, nOff + 1, int) ARR("inst_a", d.inst_a, I, int) ARR("inst_c", d.inst_c, I, int) ARR("inst_al", d.inst_al, I, int)
  ARR("blk_i", ba->d_blk_i, Cp + nOff, int) ARR("blk_j", ba->d_blk_j, Cp + nOff, int)
  ARR("chunk_off", d.chunk_off, d.chunk_off ? d.n_chunk + 1 : 0, int)
  ARR("pers_uoff", ba->d_pers_uoff, ba->d_pers_uoff ? 2 * n_cl + 1 : 0, int) ARR("pers_loc", ba->d_pers_loc, ba->d_pers_loc ? ent : 0, int)
  ARR("pers_coff", ba->d_pers_coff, ba->d_pers_coff ? n_cl + 1 : 0, int)
  ARR("row_unit_off", d.row_unit_off, d.row_unit_off ? Cp + 1 : 0, int) ARR("blk_unit0", d.blk_unit0, d.blk_unit0 ? nOff + 1 : 0, int)
#undef ARR
  if (n == "cam_edge" || n == "cam_pt") {
    if (int rc = host_int(d.cam_off, Cp, &tail)) return rc;
    src = n == "cam_edge" ? d.cam_edge : d.cam_pt; sz = (size_t)tail * sizeof(int);
  } else if (n == "cam_oi") {
    if (int rc = host_int(d.cam_off, Cp, &tail)) return rc;
    src = d.cam_oi; sz = 4 * (size_t)tail * sizeof(double);
  } else if (n == "pers_ucol") {
    if (ba->d_pers_uoff) { if (int rc = host_int(ba->d_pers_uoff, 2 * n_cl, &tail)) return rc; }
    src = ba->d_pers_ucol; sz = (size_t)tail * sizeof(int);
  } else if (n == "pers_cij" || n == "pers_cblk") {
    if (ba->d_pers_coff) { if (int rc = host_int(ba->d_pers_coff, n_cl, &tail)) return rc; }
    src = n == "pers_cij" ? (const void*)ba->d_pers_cij : (const void*)ba->d_pers_cblk; sz = (size_t)tail * sizeof(int);
  } else if (n == "unit_tab") {
    if (d.row_unit_off) { if (int rc = host_int(d.row_unit_off, Cp, &tail)) return rc; }
    src = d.unit_tab; sz = (size_t)tail * sizeof(int4);
  } else if (n == "cb_off" || n == "cb_ab") {
    src = n == "cb_off" ? ba->d_cb_off : ba->d_cb_ab;
    sz = ba->coarse_na ? (n == "cb_off" ? (size_t)ba->coarse_ncb + 1 : 2 * (size_t)ba->coarse_ncb) * sizeof(int) : 0;
  } else if (n == "cb_ent") {
    src = ba->d_cb_ent; sz = ba->coarse_na ? (Cp + nOff) * sizeof(int) : 0;
  } else if (n == "sizes") {   // Cp, Lp, Lloc, Eloc, nOff, max_cam_edges, row_units_max, pers_grid, coarse_na, coarse_ncb, n_chunk, lp_begin
    const int v[12] = {d.Cp, ba->Lp, d.Lloc, d.Eloc, d.nOff, d.max_cam_edges, d.row_units_max, ba->pers_grid, ba->coarse_na, ba->coarse_ncb, d.n_chunk, ba->lp_begin};
    *bytes = sizeof(v);
    if (out) { if (cap_bytes < sizeof(v)) return CCM_E_ARG; std::memcpy(out, v, sizeof(v)); }
    return CCM_OK;
  } else if (!src && sz == 0 && n != "chunk_off" && n != "cam_oi" && n.rfind("pers_", 0) != 0 && n != "row_unit_off" && n != "blk_unit0") {
    bool known = false;
    for (const char* k : {"slot_cam", "slot_pt", "loc_edge_orig", "pt_off", "ed_cam", "ed_cslot", "ed_pt", "obs", "info", "cam_off", "rowblk_off", "row_off", "row_col", "row_blk",
                          "inst_off", "inst_a", "inst_c", "inst_al", "blk_i", "blk_j"}) known = known || n == k;
    if (!known) return ccm_set_error(ctx, CCM_E_ARG, "ccm_ba_debug_array: unknown array " + n);
  }
  *bytes = sz;
  if (!out || !sz) return CCM_OK;
  if (cap_bytes < sz || !src) return CCM_E_ARG;
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpy(out, src, sz, hipMemcpyDeviceToHost));
  return CCM_OK;
}

// ---- C wrappers of the C++-linkage test entry points that live next to the file-local kernels they exercise (test_internal.h) ----
extern "C" int ccm_ba_debug_partial_reduced(ccm_ba* ba, double lambda, double* out, size_t cap, size_t* count) { return ccm_internal::ba_debug_partial_reduced(ba, lambda, out, cap, count); }
extern "C" int ccm_ba_debug_coarse(ccm_ba* ba, double lambda, int* na, double* Ac, double* Ainv, double* Pm, size_t cap) { return ccm_internal::ba_debug_coarse(ba, lambda, na, Ac, Ainv, Pm, cap); }
extern "C" int ccm_comm_loopback_create(int nranks, void** group) { return ccm_internal::comm_loopback_create(nranks, group); }
extern "C" void ccm_comm_loopback_destroy(void* group) { ccm_internal::comm_loopback_destroy(group); }
extern "C" int ccm_comm_init_loopback(ccm_ctx* ctx, void* group, int rank) { return ccm_internal::comm_init_loopback(ctx, group, rank); }
extern "C" int ccm_debug_dense_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info) { return ccm_internal::debug_dense_solve(ctx, A, b, n, x, info); }
extern "C" int ccm_debug_tile_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info, int* levels, int* tiles) { return ccm_internal::debug_tile_solve(ctx, A, b, n, x, info, levels, tiles); }
extern "C" int ccm_debug_dense_inverse(ccm_ctx* ctx, const double* A, int n, double* Ainv, int* info) { return ccm_internal::debug_dense_inverse(ctx, A, n, Ainv, info); }
extern "C" int ccm_orb_debug_timing(const ccm_orb* o, double out_ms[6]) { return ccm_internal::orb_debug_timing(o, out_ms); }
extern "C" int ccm_orb_debug_level(ccm_orb* o, int level, uint8_t* score_out, uint8_t* blur_out) { return ccm_internal::orb_debug_level(o, level, score_out, blur_out); }
extern "C" int ccm_orb_debug_candidates(ccm_orb* o, int level, ccm_keypoint* out, int cap, int* n_out) { return ccm_internal::orb_debug_candidates(o, level, out, cap, n_out); }
extern "C" int ccm_orb_debug_octree_dev(ccm_ctx* ctx, const int32_t* x, const int32_t* y, const int32_t* response, int n, int W, int H, int N, int32_t* sel_out, int cap, int* n_out, int* overflow) {
  return ccm_internal::orb_debug_octree_dev(ctx, x, y, response, n, W, H, N, sel_out, cap, n_out, overflow);
}

// lane_xor.h against __shfl_xor: out[6][3][64] doubles — for MASK = 1, 2, 4, 8, 16, 32: from_partner<MASK>(v), add_partner<MASK>(v) and __shfl_xor(v, MASK) of the 64 lanes'
// values in[64]; sum_out[0] = lanex::wave_sum, sum_out[1] = the __shfl_xor butterfly 32 ... 1, sum_out[2] = lanex::wave_incl_scan_i32 of the pattern (37 lane mod 101) - 20.  tests/test_lane_xor_gpu.py.
namespace {
template <int MASK>
__device__ void lane_xor_case(const double v, double* out, int slot) {
  const int lane = threadIdx.x;
  out[(3 * slot + 0) * 64 + lane] = lanex::from_partner<MASK>(v);
  out[(3 * slot + 1) * 64 + lane] = lanex::add_partner<MASK>(v);
  out[(3 * slot + 2) * 64 + lane] = __shfl_xor(v, MASK, 64);
}
__global__ __launch_bounds__(64) void lane_xor_probe(const double* in, double* out, double* sum_out) {
  const double v = in[threadIdx.x];
  lane_xor_case<1>(v, out, 0); lane_xor_case<2>(v, out, 1); lane_xor_case<4>(v, out, 2);
  lane_xor_case<8>(v, out, 3); lane_xor_case<16>(v, out, 4); lane_xor_case<32>(v, out, 5);
  double a = lanex::wave_sum(v), b = v;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) b += __shfl_xor(b, off, 64);
  sum_out[threadIdx.x] = a; sum_out[64 + threadIdx.x] = b;
  sum_out[128 + threadIdx.x] = (double)lanex::wave_incl_scan_i32((int)(threadIdx.x * 37 % 101) - 20);   // (integer scan of a fixed pattern)
}
}  // namespace
extern "C" int ccm_debug_lane_xor(ccm_ctx* ctx, const double* in64, double* out_6x3x64, double* sums_2x64 /* [3][64]: wave sums by lane_xor.h, by __shfl_xor, inclusive integer scan */) {
  if (!ctx || !in64 || !out_6x3x64 || !sums_2x64) return CCM_E_ARG;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double* d = nullptr;
  CCM_HIP_CHECK(ctx, hipMalloc(&d, (64 + 6 * 3 * 64 + 192) * sizeof(double)));
  CCM_HIP_CHECK(ctx, hipMemcpy(d, in64, 64 * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(lane_xor_probe, dim3(1), dim3(64), 0, ctx->stream, d, d + 64, d + 64 + 6 * 3 * 64);
  CCM_HIP_CHECK(ctx, hipGetLastError());
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  CCM_HIP_CHECK(ctx, hipMemcpy(out_6x3x64, d + 64, 6 * 3 * 64 * sizeof(double), hipMemcpyDeviceToHost));
  CCM_HIP_CHECK(ctx, hipMemcpy(sums_2x64, d + 64 + 6 * 3 * 64, 192 * sizeof(double), hipMemcpyDeviceToHost));
  CCM_HIP_CHECK(ctx, hipFree(d));
  return CCM_OK;
}
