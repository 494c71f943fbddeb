// test_internal.h — entry points that exist for tests only.  They are NOT part of the C ABI (include/ccm_hip.h): C++ linkage inside namespace
// ccm_internal, defined next to the file-local kernels they exercise, and reached from tests through the `extern "C"` wrappers of
// libccm_testhooks.so (test_hooks.hip, declared in include/ccm_testhooks.h).  Nothing on a product path calls them.
#pragma once
#include "../../include/ccm_hip.h"
#include <cstddef>
#include <cstdint>
namespace ccm_internal {
int  ba_debug_partial_reduced(ccm_ba* ba, double lambda, double* out, size_t cap, size_t* count);
int  ba_debug_coarse(ccm_ba* ba, double lambda, int* na, double* Ac, double* Ainv, double* Pm, size_t cap);
int  comm_loopback_create(int nranks, void** group);
void comm_loopback_destroy(void* group);
int  comm_init_loopback(ccm_ctx* ctx, void* group, int rank);
int  debug_dense_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info);
int  debug_tile_solve(ccm_ctx* ctx, const double* A, const double* b, int n, double* x, int* info, int* levels, int* tiles);
int  debug_dense_inverse(ccm_ctx* ctx, const double* A, int n, double* Ainv, int* info);
int  orb_debug_timing(const ccm_orb* o, double out_ms[6]);
int  orb_debug_level(ccm_orb* o, int level, uint8_t* score_out, uint8_t* blur_out);
int  orb_debug_candidates(ccm_orb* o, int level, ccm_keypoint* out, int cap, int* n_out);
int  orb_debug_octree_dev(ccm_ctx* ctx, const int32_t* x, const int32_t* y, const int32_t* response, int n, int W, int H, int N,
                          int32_t* sel_out, int cap, int* n_out, int* overflow);
}  // namespace ccm_internal
