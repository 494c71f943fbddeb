// common.h — internal helpers shared by the HIP translation units of libccm_hip.so.
// gfx950 only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
// (advisor, round 4) several kernels rely on gfx950 behaviour the HIP model does not promise — 64-lane waves everywhere, the raw `s_waitcnt lgkmcnt(0); s_barrier`
// of orb_pyramid_kernel, wave-divergent loops that meet at the same number of barriers in dense_chol.hip, sc1 loads for cross-XCD coherence in ba_pcg_persist:
// another target must be a build error, not silent corruption.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libccm_hip.so is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
#include <stdint.h>
#include <map>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/ccm_hip.h"

struct ccm_prof_slot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  int64_t launches = 0;
  double total_ms = 0.0;
};

struct ccm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // event pool for per-kernel timing
  int prof_class = -2;  // -2 none, -1 all
  ccm_prof_slot prof[CCM_K_COUNT];
  std::vector<hipEvent_t> ev_pool;
  // kernels whose dynamic-LDS limit has been raised for this context's device (bit per kernel, CCM_LDS_ATTR): per
  // context, not process-wide, so that a second context on another GPU of the same process sets its own
  uint32_t lds_attr_done = 0;
  // RCCL communicator (opaque; ncclComm_t) for the sharded global BA
  void* comm = nullptr;
  int comm_rank = 0, comm_nranks = 1;
  void* loop_group = nullptr;   // test-only in-process communicator (ccm_comm_init_loopback): ranks = threads sharing one GPU
  // reusable staging buffers
  void* d_scratch = nullptr; size_t d_scratch_bytes = 0;
  void* d_io = nullptr; size_t d_io_bytes = 0;   // staging for the host-pointer entry points
  void* h_pin = nullptr; size_t h_pin_bytes = 0; // pinned host staging (one H2D / D2H per small call)
  // size-bucketed cache of device blocks released by BA handles: a local BA builds a fresh problem for every keyframe, and
  // ~45 hipMalloc + hipFree per problem cost more than the host-side structure build itself
  std::multimap<size_t, void*> pool_free; size_t pool_bytes = 0;
  // 128-byte pinned (coherent) read-back blocks of BA handles, kept across handles: a hipHostMalloc + hipHostFree pair per local BA is two driver calls of tens of microseconds
  std::vector<void*> rb_free;
};

int ccm_set_error(ccm_ctx* ctx, int code, const std::string& msg);
// development prints / device phase clocks: CCM_DBG = comma-separated list of topics (pers, trial, coarse, dense2, cholreg, setup, row, orb, pg, poseopt) or "all"; read once
bool ccm_dbg(const char* topic);

// Bracket around the launch of a kernel that needs ALL its workgroups co-resident on the device (ba_pcg_persist: up to one workgroup per CU, grid-wide
// exchanges inside).  The reference runs one global-BA thread per Map (cslam/src/Map.cpp:1401-1402, LoopFinder.cpp:686-688) beside LocalMapping and Tracking
// (ClientHandler.cpp:184): two such launches from two contexts of one process would each get part of the chip and spin until their bounded waits give up.
// The bracket is a per-device, process-wide LEASE kept on the GPU, not on the host: under a short host mutex the launching stream first waits for the event
// the previous lease holder recorded behind ITS kernel, then the kernel is launched and this stream's event becomes the one to wait for — the kernels run one
// after the other whatever stream they are on, no host thread ever blocks (so ranks that are threads of one process, joined by a collective, cannot deadlock),
// and a process with ONE context on the device pays nothing (no event is recorded or waited for).  Kernels that merely want many CUs are not bracketed: they
// finish on their own, and the persistent kernel's bounded waits cover the time its last workgroups need to become resident behind them.
struct ccm_coresident_scope {
  ccm_ctx* ctx; bool chained = false;
  explicit ccm_coresident_scope(ccm_ctx* c);
  ~ccm_coresident_scope();
  ccm_coresident_scope(const ccm_coresident_scope&) = delete;
  ccm_coresident_scope& operator=(const ccm_coresident_scope&) = delete;
};
void ccm_coresident_note_abort(ccm_ctx* ctx);   // a bracketed kernel gave up waiting for its peers (counted per device: ccm_coresidency_stats)

#define CCM_HIP_CHECK(ctx, expr)                                                        \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return ccm_set_error((ctx), CCM_E_HIP,                                            \
                           std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +   \
                               __FILE__ + ":" + std::to_string(__LINE__) + ")");        \
  } while (0)

enum { CCM_LDS_BA_ROW = 0, CCM_LDS_BA_ROW2 = 8, CCM_LDS_BA_SMALL, CCM_LDS_BA_TILES, CCM_LDS_PG_PRECOND, CCM_LDS_POSEOPT, CCM_LDS_SIM3OPT, CCM_LDS_BA_DENSE2, CCM_LDS_ORB_OCT, CCM_LDS_ORB_OCT2, CCM_LDS_BA_ROW3, CCM_LDS_BA_ROW4, CCM_LDS_POSEOPT1, CCM_LDS_BA_CHOLREG };
#define CCM_LDS_ATTR(ctx, bit, func, bytes)                                                                              \
  do {                                                                                                                   \
    if (!((ctx)->lds_attr_done & (1u << (bit)))) {                                                                       \
      CCM_HIP_CHECK((ctx), hipFuncSetAttribute((const void*)(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      (ctx)->lds_attr_done |= 1u << (bit);                                                                               \
    }                                                                                                                    \
  } while (0)

// RAII-ish bracket used around a kernel launch when profiling of its class is enabled.
struct ccm_prof_scope {
  ccm_ctx* ctx; int cls; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; bool on = false;
  ccm_prof_scope(ccm_ctx* c, int k, hipStream_t launch_stream = nullptr);   // nullptr: the context's stream
  ~ccm_prof_scope();
};

// device scratch that grows on demand (never shrinks); contents undefined
int ccm_scratch(ccm_ctx* ctx, size_t bytes, void** out);
int ccm_io_scratch(ccm_ctx* ctx, size_t bytes, void** out);
int ccm_pin_scratch(ccm_ctx* ctx, size_t bytes, void** out);
// pooled device blocks: *actual receives the bucket size to hand back to ccm_pool_put
int ccm_pool_get(ccm_ctx* ctx, size_t bytes, void** out, size_t* actual);
void ccm_pool_put(ccm_ctx* ctx, void* p, size_t actual);
// dense_chol.hip: SPD solve on the device (N multiple of 64, padding = identity), see the definition for the contract
// Tile-level sparsity of a dense SPD system (64 x 64 tiles): which tiles of L can be non-zero, after symbolic fill.  The
// factorisation and the substitutions then touch only those (pose graphs are near-banded: a few tiles per tile row).
// Lists live on the device, the offsets also on the host (grid sizes).  nullptr plan = every lower tile.
struct ccm_tile_plan {
  int T = 0;
  std::vector<int> h_col_off;    // [T+1] rows i > j with L_ij != 0 (panel of step j; forward substitution)
  std::vector<int> h_upd_off;    // [T+1] pairs (i >= k) of those rows (trailing update of step j), packed i << 16 | k
  std::vector<int> h_row_off;    // [T+1] columns k < j with L_jk != 0 (backward substitution)
  const int* d_col_rows = nullptr;
  const int* d_upd_pairs = nullptr;
  const int* d_row_cols = nullptr;
};
// nz: T x T row-major flags of the lower triangle of A (tile (i, j), i >= j, non-zero); returns the host-side lists
void ccm_tile_plan_symbolic(int T, const std::vector<char>& nz, ccm_tile_plan* plan, std::vector<int>* col_rows, std::vector<int>* upd_pairs,
                            std::vector<int>* row_cols);
int ccm_dense_chol_solve_dev(ccm_ctx* ctx, double* d_A, int N, double* d_b, double* d_linv, int* d_info, const ccm_tile_plan* plan = nullptr);
int ccm_dense_chol_inverse_dev(ccm_ctx* ctx, double* d_A, int N, double* d_linv, double* d_X, double* d_Ainv, int* d_info);
// Tile-sparse Cholesky with level scheduling (dense_chol.hip): only the non-zero 64 x 64 tiles of L are stored (tid: tile (i, j) -> slot, i >= j),
// left-looking, all tile columns of one elimination level in one launch.  The caller fills d_tiles (lower triangle; diagonal tiles: lower part)
// after ccm_tsc_clear and calls ccm_tsc_solve; everything lives in pooled blocks released by ccm_tsc_destroy.
struct ccm_tsc {
  int T = 0, n_tiles = 0, n_levels = 0;
  std::vector<int> tid;                                   // host copy [T * T], -1 = zero tile
  std::vector<int> lvl_col_off, lvl_gt_off, lvl_pt_off;   // [n_levels + 1]
  int *d_tid = nullptr, *d_cols = nullptr, *d_gt = nullptr, *d_gk_off = nullptr, *d_gk = nullptr, *d_pt = nullptr;
  int *d_rc_off = nullptr, *d_rc = nullptr, *d_cr_off = nullptr, *d_cr = nullptr;
  double *d_tiles = nullptr, *d_linv = nullptr;
  int* d_info = nullptr;
  std::vector<std::pair<void*, size_t>> blocks;
  ccm_ctx* owner = nullptr;                               // set by ccm_tsc_create: the destructor hands the blocks back to its pool
  ccm_tsc() = default;
  ccm_tsc(const ccm_tsc&) = delete;
  ccm_tsc& operator=(const ccm_tsc&) = delete;
  ~ccm_tsc();
};
int ccm_tsc_create(ccm_ctx* ctx, int T, const std::vector<char>& nz, ccm_tsc* out);
void ccm_tsc_destroy(ccm_ctx* ctx, ccm_tsc* p);
int ccm_tsc_clear(ccm_ctx* ctx, ccm_tsc* p);
int ccm_tsc_solve(ccm_ctx* ctx, ccm_tsc* p, double* d_b);
static inline size_t ccm_align256(size_t n) { return (n + 255) & ~(size_t)255; }

static inline int ccm_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
