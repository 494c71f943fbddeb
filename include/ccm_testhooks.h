/* ccm_testhooks.h — TEST-ONLY entry points of libccm_testhooks.so (ccm_slam_amd/csrc/test_hooks.hip).  Not part of the product's C ABI:
 * libccm_hip.so exports none of these names and include/ccm_hip.h declares none; the library links against the product and reaches the
 * file-local kernels through the C++-linkage functions of ccm_slam_amd/csrc/test_internal.h.  Used by tests/ and scripts/ only. */
#ifndef CCM_TESTHOOKS_H
#define CCM_TESTHOOKS_H
#include "ccm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* one of the structure arrays that ccm_ba_create built on the device (ba_build.hip), by name; out == NULL only queries *bytes */
int  ccm_ba_debug_array(ccm_ba* ba, const char* name, void* out, size_t cap_bytes, size_t* bytes);

/* test hook: DistributeOctTree as the DEVICE runs it (orb_octree_kernel, one workgroup) on one level's candidates: integer positions inside the
 * level's W x H border box, responses 1..255, positions unique; sel_out: indices of the kept candidates in output order.  *overflow = 1 when
 * the set does not fit the kernel's LDS plan (the extractor then selects on the host). */
int  ccm_orb_debug_octree_dev(ccm_ctx* ctx, const int32_t* x, const int32_t* y, const int32_t* response, int n, int W, int H, int N,
                              int32_t* sel_out, int cap, int* n_out, int* overflow);

/* intermediate products for parity tests (host copies; any pointer may be NULL):
 * FAST score map and blurred image of one level of the LAST extracted frame.              */
int  ccm_orb_debug_level(ccm_orb* orb, int level, uint8_t* score_out, uint8_t* blur_out);
/* host wall-clock phases of the last ccm_orb_extract call, ms: [queue phase 1, wait for candidates, octree, queue phase 2,
 * wait + D2H, total] */
int  ccm_orb_debug_timing(const ccm_orb* orb, double out_ms[6]);
/* pre-octree FAST candidates of the last frame: returns count for the level, fills up to cap */
int  ccm_orb_debug_candidates(ccm_orb* orb, int level, ccm_keypoint* out, int cap, int* n_out);

/* test hook: this rank's partial reduced camera system [36*(n_free_cams+n_blocks) S | 6*n_free_cams b]
 * at the current state, i.e. the buffer the per-trial RCCL all-reduce sums; out == NULL only queries *count. */
int  ccm_ba_debug_partial_reduced(ccm_ba* ba, double lambda, double* out, size_t cap, size_t* count);

/* test hook: coarse level of the two-level PCG preconditioner at the current state: *na aggregates (0 = not in use), Ac and
 * its inverse [6 na x 6 na], prolongation blocks P_k = Ad(T_cw,k) [n_free_cams x 36]; cap = doubles available in Ac / Ainv. */
int  ccm_ba_debug_coarse(ccm_ba* ba, double lambda, int* na, double* Ac, double* Ainv, double* Pm, size_t cap);

/* TEST-ONLY in-process communicator: the ranks are threads of one process sharing one GPU, each with its own ccm_ctx; an all-reduce is
 * a rendezvous plus one reduction kernel that leaves the same bits in every rank's buffer (RCCL's contract).  Lets a single-GPU box
 * execute the complete multi-rank control flow of the sharded global BA (tests/test_sharded_loopback_gpu.py). */
int  ccm_comm_loopback_create(int nranks, void** group);
void ccm_comm_loopback_destroy(void* group);
int  ccm_comm_init_loopback(ccm_ctx* ctx, void* group, int rank);

/* test hook for the dense f64 Cholesky (MFMA tiles) behind ccm_pose_graph_optimize: solves A x = b for a host matrix
 * (n x n row-major, symmetric positive definite); *info = 0 or (first non-positive pivot + 1). */
int  ccm_debug_dense_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info);
/* test hook for the tile-sparse, level-scheduled form of the same factorisation (what ccm_pose_graph_optimize uses by default): only the
 * non-zero 64 x 64 tiles of the factor are stored, tile columns of one elimination level run in one launch; the tile pattern is taken from
 * the non-zeros of A.  levels / tiles (nullable) receive the plan's level and tile counts. */
int  ccm_debug_tile_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info, int* levels, int* tiles);
/* same machinery, explicit inverse (used for the coarse level of the BA preconditioner) */
int  ccm_debug_dense_inverse(ccm_ctx* ctx, const double* A, int n, double* Ainv, int* info);

/* test hook for ccm_slam_amd/csrc/lane_xor.h (round 6: the cross-lane moves of every xor butterfly — DPP, v_permlane16_swap / v_permlane32_swap): for MASK = 1, 2, 4, 8, 16, 32
 * the 64 lanes' from_partner<MASK>(v), add_partner<MASK>(v) and __shfl_xor(v, MASK) of in64, then [3][64]: lanex::wave_sum, the __shfl_xor butterfly of the same values, and
 * lanex::wave_incl_scan_i32 of the integer pattern (37 lane mod 101) - 20. */
int  ccm_debug_lane_xor(ccm_ctx* ctx, const double* in64, double* out_6x3x64, double* sums_3x64);

#ifdef __cplusplus
}
#endif
#endif /* CCM_TESTHOOKS_H */
