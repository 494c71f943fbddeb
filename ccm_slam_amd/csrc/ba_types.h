// ba_types.h — device-side problem description (BaDev), the BA handle (ccm_ba) and the size constants shared by ba.hip (kernels, LM loop) and
// ba_build.hip (structure build on the device).
#pragma once
#include "common.h"

constexpr int kWave = 64;
constexpr int kTPB = 256;
constexpr uint32_t kTransposeBit = 0x80000000u;
constexpr int kRowMaxEdges = 1000;   // 144 B of LDS per observation of the camera
constexpr int kRow2TPB = 1024;       // lane-per-instance row kernel: 16 waves, i.e. 128 VGPRs per lane (36-42 accumulators + the W_c row; the Y row is read three values at a time)
constexpr int kRow2Group = 16;       // lanes that share one work unit (block chunk); 4 units per wave pass
constexpr int kRow2Chunk = 64;       // pair instances per work unit: <= 4 per lane (measured on the 4-agent map with ba_schur_row3: 16: 164 us, 32: 143, 48: 139, 64: 137, 96: 139;
                                     // with ba_schur_row2, whose passes were bound by their divergent loads: 32: 211 us, 64: 219, 128: 271)
constexpr int kClu = 16;             // cameras per preconditioner cluster
constexpr int kCluN = 6 * kClu;      // 96 unknowns
constexpr int kSpmvTPB = 1024;
constexpr int kRowsPerWG = kSpmvTPB / (2 * kWave);
constexpr int kCoarseNcCap = 768;              // (see ba_build.hip: the interval is the smallest of 16 / 24 / 32 cameras whose coarse system stays within this size)
constexpr int kAggFine = 16, kAggMid = 24, kAggWide = 32;   // cameras per interval of the coarse space (BaDev::agg): 16 wherever the persistent solver's LDS holds the 12 rows of
                                             // Ac^-1 a unit needs (maps up to 2032 free cameras), 32 otherwise and on the multi-kernel path (= 2 clusters there)
constexpr int kPersTPB = 1024;
constexpr int kPersWaves = kPersTPB / kWave;
constexpr int kPersIdxCap = 3072;    // CSR entries of one persistent unit's rows (staged in LDS)
constexpr int kPersColCap = 640;     // distinct neighbour columns of one persistent unit
constexpr int kSmallMaxCp = 16;      // one cluster: the single-workgroup kernel; above, the persistent kernel with its 16-camera cluster preconditioner
constexpr int kDense2MaxCp = 2 * kClu;
constexpr int kCholRegMaxCp = 50;    // (round 6) exact register-resident Cholesky in one workgroup: 19 x 19 tiles of 16 x 16 (n = 6 Cp <= 304), ba_solve_cholreg

struct BaDev {
  // sizes
  int n_cam, Cp, Lloc, Eloc, nOff;
  double huber;
  // camera state (all cameras replicated), [2][n_cam*7]
  double* cam[2];
  const double* K;           // [n_cam*4]
  const int* slot_cam;       // [Cp] pose slot -> camera index
  // landmark state (own landmarks), [2][Lloc*3]
  double* pt[2];
  // edges of own landmarks, sorted by landmark
  const int* pt_off;         // [Lloc+1]
  const int* ed_cam;         // [Eloc] camera index
  const int* ed_cslot;       // [Eloc] pose slot or -1 (fixed camera)
  const int* ed_pt;          // [Eloc] local landmark
  const double* obs;         // [Eloc*2]
  const double* info;        // [Eloc]
  // per camera slot: list of local edges
  const int* cam_off;        // [Cp+1]
  const int* cam_edge;       // [..]
  const int* cam_pt;         // [..] landmark of every camera-list slot (ed_pt[cam_edge[s]])
  double* cam_oi;            // [..][4] (round 4) observation and information of every camera-list slot (obs x, obs y, info, 0): what ba_linearize_cams reads contiguously instead of three gathers by edge index; refreshed by ccm_ba_set_edge_levels
  // linear system pieces
  double* W;                 // [Eloc*18]  Hpl block of each edge (pose rows x landmark cols)
  double* Hll;               // [Lloc*6]   symmetric 3x3
  double* bl;                // [Lloc*3]
  double* Dinv;              // [Lloc*6]
  double* dl;                // [Lloc*3]   D^-1 b_l
  double* Hpp;               // [Cp*36]    partial (own edges)
  double* bp;                // [Cp*6]     partial
  // reduced system (contiguous: all-reduced in one call)
  double* S;                 // [(Cp+nOff)*36] diagonal blocks first
  double* bs;                // [Cp*6]
  double* qx;                // [6 Cp] (S + lambda I) x of a residual replacement (q itself follows a recurrence across iterations and must survive it); allocated with S32
  float* S32;                // [(Cp + nOff) * 36] f32 copy of S for the product kernel of the multi-kernel PCG (maps above 2048 free cameras; nullptr otherwise), remade every trial
  const int* inst_off;       // [nOff+1]
  const int* inst_a;         // edge with the block-row camera
  const int* inst_c;         // edge with the block-col camera
  const int* inst_al;        // rank of inst_a's edge inside its camera's edge list (row-centric Schur kernel)
  const int* rowblk_off;     // [Cp+1] off-diagonal blocks (i, j > i) of block row i = [rowblk_off[i], rowblk_off[i+1])
  int max_cam_edges;         // longest per-camera edge list on this rank
  // row-centric Schur kernel: work units = (block, chunk of <= row_chunk consecutive pair instances), dealt to the waves
  const int4* unit_tab;      // [n_units]: block, first instance, end instance, slot of the partial sum inside the row (creation index); per row longest unit first
  const int* row_unit_off;   // [Cp+1] units of block row i
  const int* blk_unit0;      // [nOff+1] first unit of every block (a block's units are consecutive)
  int row_units_max;         // most units in one row (LDS partial sums); 0 = row kernel not usable
  int unit_chunk;            // pair instances per work unit (kRow2Chunk)
  // compact form of the Hpl blocks for the row kernel (ba_schur_row3)
  double* E4;                // [cam_off[Cp]][4] per observation of a free camera, CAMERA-MAJOR (the order of cam_edge): x, y, 1 / z of the landmark in the camera frame, w * information
  double* E4L;               // [Eloc][4] the same four numbers in LANDMARK-major order (the order of the edge arrays), written by the landmark-side linearisation;
                             // nullptr unless w_free
  int w_free;                // the Hpl blocks are never stored: every kernel of the handle's path derives them from E4 / E4L + camRK (large maps: row kernel +
                             // edge-parallel landmark kernels); 0: W is written and the fallback Schur kernels read it
  double* camRK;             // [Cp][12] per pose slot: rotation matrix (row-major), fx, fy, 0
  const int* inst_cp;        // [n_inst] ed_cpos of the pair instance's block-col observation
  const int* blk_j;          // [nOff] column pose slot of every off-diagonal block
  long long* row_dbg;        // nullable (CCM_DBG=row in a -DCCM_BA_ROW_DBG_BUILD build): [8] phase clocks of the row kernel summed over its workgroups (10 ns ticks) + launches
  // block CSR for SpMV (full rows, diag included)
  const int* row_off;        // [Cp+1]
  const int* row_col;        // [..]
  const uint32_t* row_blk;   // [..] block id | transpose bit
  // symmetric product of the multi-kernel PCG (round 4; nullptr: every row reads its lower blocks transposed, ba_pcg_spmv): the row of an upper block (i, j) also forms
  // S_ij^T p_i and stores it at the block's place among row j's lower entries; ba_pcg_update adds a row's lower parts to q
  // PCG
  double *x, *r, *z, *q, *p[2];
  float* Wc;                 // [n_clusters][96*96] explicit inverses of the damped cluster blocks, multi-kernel PCG; f32: only a preconditioner (offline: the same CG iteration counts as f64), half the 74 KB a cluster re-reads in every CG iteration
  double *ppq, *prz[2];      // partials
  double* pcg_scal;          // [0]=rz0 [1]=thresh^2  [2]=lambda
  int* pcg_flag;             // [0]=done [1]=iters [2]=fail
  int n_wg_spmv, n_wg_upd, n_wg_wave4;   // wave4: one wave per camera, 4 per workgroup
  // trial outputs
  double* edge_chi2;         // [Eloc]
  uint8_t* edge_depth;       // [Eloc]
  double* part_pt;           // [n_wg_pt*2]  (robust chi2, scale) partials
  double* part_cam;          // [n_wg_cam]   scale partials (pose part)
  double* scal;              // [0] chi2 [1] scale [2] stop requested on any rank [3] persistent PCG gave up on any rank (0..3 are summed over
                             // ranks in ONE all-reduce per trial) [4] maxdiag [5] - [6..7] = pcg_flag (4 ints)
  int n_wg_pt, n_wg_cam;
  // edge-parallel landmark kernels: chunks of consecutive landmarks with <= kTPB landmarks and <= kTPB observations
  const int* chunk_off;      // [n_chunk+1] landmark ranges; nullptr: a landmark has more than kTPB observations -> thread-per-landmark kernels
  int n_chunk;
  int n_part;                // entries of part_pt written by the last chi2 kernel (n_chunk or n_wg_pt)
  // coarse level of the multi-kernel PCG (maps above 2048 free cameras)
  double* mk_cpart;          // [4][6 (na + 1)] restriction parts P^T r, node-major: slot 0 / 1 = first-node part of cluster 2n / 2n + 1, slot 2 / 3 = second-node part of cluster 2n - 2 / 2n - 1
  double* mk_cry[2];         // [n_clusters] coarse part of r.z per cluster (first cluster of an aggregate), by iteration parity
  const double* mk_P;        // [Cp][36] prolongation blocks
  const double* mk_Ainv;     // [mk_Nc][mk_Nc] coarse inverse
  const float* mk_Ainv32;    // the same rounded to f32: what ba_pcg_coarse_apply reads (12 rows per cluster and CG iteration)
  int mk_on, mk_Nc, mk_na;   // mk_on: this trial's solve uses the coarse level
  int agg;                   // cameras per interval of the coarse space (kAggFine / kAggWide); a multiple of the 8 cameras of a persistent unit
};

struct ccm_ba {
  ccm_ctx* ctx = nullptr;
  int rank = 0, nranks = 1;
  int n_cam = 0, n_pt = 0, n_edge = 0;
  int Cp = 0, Lp = 0, Lloc = 0, Eloc = 0, nOff = 0;
  int64_t n_inst = 0, n_act_edges = 0, n_row_entries = 0;
  int lp_begin = 0, lp_end = 0;
  std::vector<int> slot_cam;            // [Cp] pose slot -> camera index (host copy for ccm_ba_download)
  std::vector<int> loc_edge_orig;       // local edge -> the caller's edge index (host copy, fetched on first use)
  double *d_raw_cam = nullptr, *d_raw_pt = nullptr;   // the caller's cameras [n_cam][7] / landmarks [n_pt][3] as uploaded (create / reset); d_raw_pt also stages the download
  int *d_slot_pt = nullptr, *d_loc_edge_orig = nullptr;   // [Lp] landmark slot -> landmark index; [Eloc]
  std::vector<std::pair<void*, size_t>> allocs;   // pooled blocks (ccm_pool_get)
  std::vector<std::pair<void*, size_t>> zero_list; // blocks to clear before first use (ccm_ba_create: one launch for all of them)
  BaDev d{};
  int cur = 0;
  double* d_red = nullptr; size_t red_count = 0;   // [S | bs]
  unsigned* d_pers_bar = nullptr; double* d_pers_part = nullptr; int pers_grid = 0;   // persistent PCG (0 = not usable)
  double* d_dense_T = nullptr;   // [96][96] scratch of the exact two-cluster solve (17..32 free cameras)
  double* d_cholreg_dbg = nullptr;   // [16] phase clocks of the register-resident Cholesky solve (17..50 free cameras; CCM_DBG=cholreg)
  int* d_cholreg_tab = nullptr;      // [8][24][64][4] offsets into S of every lane's entries of every tile (ba_cholreg_table, built at the first trial)
  bool cholreg_table_built = false;
  double* d_cholreg_L = nullptr;     // [n_pad][n_pad] scratch of its factor (rows below the diagonal tiles)
  int *d_pers_uoff = nullptr, *d_pers_ucol = nullptr, *d_pers_loc = nullptr, *d_pers_coff = nullptr, *d_pers_cij = nullptr;
  uint32_t* d_pers_cblk = nullptr;
  unsigned long long pers_launch = 0;
  // (round 5) pers_grid_built: the grid the handle was created for (0: it has no persistent solver); pers_grid goes to 0 for pers_cooldown trials after a launch
  // gave up waiting for its peers and then returns to it; pers_aborts counts those launches
  int pers_grid_built = 0, pers_cooldown = 0, pers_aborts = 0;
  bool w_pending = false;     // the running / last persistent launch factored the cluster blocks: its wsave becomes valid once its flags have been read clean
  bool pers_agreed = false;   // sharded handle: the ranks have agreed on whether the persistent kernel is used (first ccm_ba_run)
  // coarse level (two-level preconditioner of the persistent PCG); na = 0 -> disabled
  int coarse_na = 0, coarse_Nc = 0, coarse_ncb = 0;
  // The coarse level costs a dense inverse per trial (~0.5 ms) and ~25% per CG iteration; it pays only when the
  // cluster-Jacobi solve is long (small lambda).  Switch with hysteresis on the iteration count of the previous solve
  // (deterministic: the counts are): on after a solve of >= kCoarseOnIters iterations, off after one of <= kCoarseOffIters.
  bool coarse_active = false, coarse_used = false;
  // The coarse operator Ac = P^T (S + lambda I) P is only a preconditioner: a STALE one (built at an earlier trial's lambda or an
  // earlier linearisation point) still gives a fixed SPD M^-1 for the whole solve, so PCG converges to the same tolerance, just a few
  // iterations later.  It is rebuilt when lambda has left [1/4, 4] x the lambda it was built at, or when a solve with the stale
  // operator needed clearly more iterations than the solve right after the last build (counts are deterministic => so is the policy).
  // cluster inverse of the persistent solver carried from trial to trial (ba.hip, lm_trial): what it was built at, and the iteration guard
  double* d_pers_wsave = nullptr;   // [pers_grid][96 * 48]
  float* d_cAinv32 = nullptr;       // [Nc][Nc] multi-kernel path
  bool w_valid = false, w_loaded = false, w_stale_bad = false; double w_lambda_built = 0; int w_fresh_iters = 0, lin_id = 0, w_lin_id = -1;
  bool coarse_valid = false, coarse_fresh = false, coarse_stale_bad = false, coarse_reuse = true;
  double coarse_lambda_built = 0; int coarse_fresh_iters = 0;
  int mk_prev_iters = 0;   // CG iterations of the handle's previous multi-kernel solve (sizes the first chunk of queued iterations of the next one)
  double dinv_done_lambda = -1.0;   // lambda for which the landmark-side linearisation already formed D^-1 (ba_dinv folded in), -1 = none
  double lambda_first = 0; bool coarse_skipped_damped = false;   // the call's first lambda; the coarse level is left out at and above it (lm_trial)
  double* h_rb = nullptr;    // pinned: [6 scalars | 4 flags] of a trial, then the ticket ba_reduce_scalars writes after them (read_scalars_polled)
  unsigned long long rb_ticket = 0;
  int coarse_force = 0;      // CCM_BA_COARSE=always / never (tests), 0 = adaptive
  double *d_cP = nullptr, *d_cA = nullptr, *d_cX = nullptr, *d_cAinv = nullptr, *d_cLinv = nullptr;
  double* d_cparts = nullptr;
  unsigned* d_cb_key = nullptr;   // [ncb] sorted keys a * na + b of the interval pairs that share S blocks
  double* d_cstage = nullptr;     // [ncb][4][36] weighted sums of every interval pair (ba_coarse_assemble -> ba_coarse_sum)
  int *d_cb_off = nullptr, *d_cb_ent = nullptr, *d_cb_ab = nullptr, *d_blk_i = nullptr, *d_blk_j = nullptr, *d_cinfo = nullptr;
  double* d_pt_full = nullptr;
  double* d_hpp_full = nullptr;
  double *d_saved_cam = nullptr, *d_saved_pt = nullptr;   // ccm_ba_push_state
  double* d_info_orig = nullptr;   // the edges' information as ccm_ba_create stored it: ccm_ba_set_edge_levels derives the current one from it, so levels can go back to 0
  double ms_setup = 0;
  // stop flag of the running ccm_ba_run (the reference's bool* pbStopFlag).  One rank: read where g2o calls terminate().
  // Sharded: the local value rides in the per-trial all-reduce and only the reduced value (stop_any) is acted on.
  const volatile unsigned char* stop_flag = nullptr;
  bool stop_any = false;
  bool stop_requested() const { return nranks > 1 ? stop_any : (stop_flag && *stop_flag); }
  int stop_local() const { return (stop_flag && *stop_flag) ? 1 : 0; }
  // per-iteration record of the last run (chi2 after the iteration, lambda, trials) and the optional per-trial callback
  std::vector<double> hist_chi2, hist_lambda; std::vector<int32_t> hist_trials;
  ccm_ba_trial_cb trial_cb = nullptr; void* trial_cb_user = nullptr;
};
