// test_hooks.hip — TEST-ONLY entry points, built into libccm_testhooks.so (NOT into the product library libccm_hip.so, which it links against).
// They reach into handles through the internal headers; nothing in the product's timed paths knows about them.
#include "common.h"
#include "ba_types.h"
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
  } else if (!src && sz == 0 && n != "chunk_off" && n.rfind("pers_", 0) != 0 && n != "row_unit_off" && n != "blk_unit0") {
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
