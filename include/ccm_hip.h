/*
 * ccm_hip.h — C ABI of libccm_hip.so, the MI355X (gfx950) back end for CCM-SLAM's
 * ORB-extract / Hamming-match / bundle-adjustment hot path.
 *
 * The reference has no FFI for this path: cslam::ORBextractor, cslam::ORBmatcher and
 * cslam::Optimizer are concrete C++ classes (cslam/include/cslam/ORBextractor.h:103-138,
 * ORBmatcher.h:100-139, Optimizer.h:84-112).  The drop-in replaces their three translation
 * units with the drop-in translation units under shim/ ({Optimizer,ORBextractor,ORBmatcher}_hip.cpp,
 * compiled against the reference's own headers), which flatten the shared_ptr graph into the POD
 * buffers below and call this ABI (matchers: through the host mirror ccm_slam_amd/host/).  Every
 * entry point cites the reference code it stands for.
 *
 * Conventions
 *   - plain C, POD only, caller-owned memory, no exceptions across the boundary;
 *   - return value: 0 = ok, <0 = error (CCM_E_*); ccm_last_error() gives a message;
 *   - handle based and re-entrant: one ccm_ctx per calling thread (own HIP stream);
 *     a ctx must not be used from two threads at once (reference threading: SURVEY §8b).
 *     Any number of contexts of ONE process may work on one device at the same time — the
 *     reference runs Tracking, LocalMapping and one global-BA thread per Map concurrently
 *     (ClientHandler.cpp:184, Map.cpp:1401-1402, LoopFinder.cpp:686-688): launches that need the
 *     whole device to themselves (the persistent reduced solve of ccm_ba_run) are put in order on
 *     the GPU by a per-device lease, see ccm_coresidency_stats;
 *   - "host" entry points take host pointers and do their own H2D/D2H; "_dev" entry points
 *     take device pointers obtained from ccm_dev_alloc (used by bench.py so that inputs are
 *     resident in HBM when the timed region starts).
 */
#ifndef CCM_HIP_H
#define CCM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCM_OK            0
#define CCM_E_ARG        -1   /* bad argument                                   */
#define CCM_E_HIP        -2   /* HIP runtime error (message in ccm_last_error)   */
#define CCM_E_NOGPU      -3   /* no gfx950 device visible: the product never falls back to CPU */
#define CCM_E_NUMERIC    -4   /* reduced system not positive definite / NaN      */
#define CCM_E_COMM       -5   /* RCCL error                                      */
#define CCM_E_STATE      -6   /* call sequence error                             */

typedef struct ccm_ctx ccm_ctx;

/* ---- context ------------------------------------------------------------------------- */
int         ccm_ctx_create(int device_id, ccm_ctx** out);
void        ccm_ctx_destroy(ccm_ctx* ctx);
const char* ccm_last_error(const ccm_ctx* ctx);   /* ctx may be NULL: last global error */
int         ccm_ctx_sync(ccm_ctx* ctx);           /* hipStreamSynchronize on the ctx stream */
int         ccm_device_count(void);
/* library/build identification, e.g. "ccm_hip 0.1 gfx950" */
const char* ccm_version(void);
/* The per-device lease of this process (all pointers nullable): launches = kernels launched under it (each needs all its workgroups
 * co-resident: the persistent PCG of a 33 .. 2048-camera bundle adjustment), chained = those that were ordered behind another
 * context's launch by an event wait on their own stream (no host thread blocks), aborted = those that still gave up waiting for their
 * peers — the trial is then repeated on the multi-kernel solver and the handle returns to the persistent one a few trials later —,
 * contexts = live contexts on the device.  Two PROCESSES sharing one device are not covered: give each its own GPU (north_star: one
 * agent per GPU) or run one of them with CCM_BA_NO_PERSIST=1. */
int         ccm_coresidency_stats(int device_id, int64_t* launches, int64_t* chained, int64_t* aborted, int* contexts);

/* device memory owned by the ctx' device (thin wrappers so callers need no HIP headers) */
int ccm_dev_alloc(ccm_ctx* ctx, size_t bytes, void** dptr);
int ccm_dev_free(ccm_ctx* ctx, void* dptr);
int ccm_memcpy_h2d(ccm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int ccm_memcpy_d2h(ccm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- per-kernel timing with HIP events on the ctx stream ------------------------------
 * bench.py's roofline leg: when enabled, every launch of the selected kernel class is
 * bracketed by hipEventRecord on the launching stream; ccm_prof_read drains the events and
 * returns launches and summed milliseconds since the last ccm_prof_reset.               */
/* Classes, not kernels.  ORB: CCM_K_PYR_RESIZE = orb_pyramid_kernel (all levels of a frame in one launch), CCM_K_FAST_NMS = orb_cells_kernel (per 30-px cell:
 * FAST-9/16 score, 3x3 NMS, the minThFAST retry) AND orb_octree_kernel (DistributeOctTree on the device; in the single-frame call the 7x7 blur tiles ride in
 * its launch), CCM_K_BLUR = orb_blur_frames_kernel of the grouped batch path and orb_blur_kernel of the host-octree path, CCM_K_BRIEF = orientation + steered descriptors (one kernel; CCM_K_ORIENT is unused), CCM_K_FAST_SCORE =
 * the score map of the test hook only. */
enum {
  CCM_K_HAMMING_DENSE = 0, CCM_K_HAMMING_CSR, CCM_K_PYR_RESIZE, CCM_K_FAST_SCORE, CCM_K_FAST_NMS,
  CCM_K_ORIENT, CCM_K_BLUR, CCM_K_BRIEF, CCM_K_BA_LINEARIZE, CCM_K_BA_CAM, CCM_K_BA_DINV,
  CCM_K_BA_SCHUR_DIAG, CCM_K_BA_SCHUR_OFF, CCM_K_BA_PCG_SPMV, CCM_K_BA_PCG_UPDATE,
  CCM_K_BA_BACKSUB, CCM_K_BA_UPDATE, CCM_K_BA_CHI2, CCM_K_POSEOPT, CCM_K_SIM3OPT, CCM_K_BA_PCG_PERSIST,
  CCM_K_BA_COARSE /* coarse operator + dense inverse of the two-level preconditioner */, CCM_K_BA_REDUCE /* trial scalars */,
  CCM_K_BA_ALLREDUCE /* the collectives of a sharded handle (RCCL all-reduce of [S | b_schur], of the trial scalars, of lambda_0's max): queue wait + wire time on the stream */, CCM_K_COUNT
};
int ccm_prof_enable(ccm_ctx* ctx, int kernel_class /* -1: all, -2: none */);
int ccm_prof_reset(ccm_ctx* ctx);
int ccm_prof_read(ccm_ctx* ctx, int kernel_class, int64_t* launches, double* total_ms);

/* ---- 256-bit Hamming ------------------------------------------------------------------
 * Replaces ORBmatcher::DescriptorDistance (cslam/src/ORBmatcher.cpp:1653-1669) and the
 * best / second-best scans inside the Search* methods (ORBmatcher.cpp:102-134, 220-245,
 * 1408-1433 ...).  Descriptors are rows of 32 bytes (cv::Mat N x 32 CV_8U, Frame.h:138).
 * Tie rule everywhere: strict '<' in candidate order, i.e. the FIRST minimum wins
 * (ORBmatcher.cpp:121,129).                                                              */

/* dense brute force: for every query row the best and second-best target over ALL T rows,
 * scanned in ascending target index.  best_idx = -1 / dists = 256 when T == 0.           */
int ccm_hamming_dense_best2(ccm_ctx* ctx, const uint8_t* q, int Q, const uint8_t* t, int T,
                            int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
int ccm_hamming_dense_best2_dev(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, int T,
                                int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist);

/* windowed / bucketed search (reference semantics): query i is compared with the ordered
 * candidate list cand_idx[cand_off[i] .. cand_off[i+1]) (the order GetFeaturesInArea
 * produced, Frame.cpp:228-250).  cand_dist receives one distance per candidate slot — the
 * host-side ordered resolution pass (claimed-feature skipping, ORBmatcher.cpp:113-115)
 * consumes it.  best2 outputs are computed ignoring claims and may be NULL.             */
int ccm_hamming_csr(ccm_ctx* ctx, const uint8_t* q, int Q, const uint8_t* t, int T,
                    const int32_t* cand_off, const int32_t* cand_idx,
                    uint16_t* cand_dist,
                    int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
int ccm_hamming_csr_dev(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, int T,
                        const int32_t* d_cand_off, const int32_t* d_cand_idx, int64_t n_cand,
                        uint16_t* d_cand_dist,
                        int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist);

/* S windowed searches in ONE launch and one read-back (round 5) — the per-keyframe fan-out of LocalMapping: up to 20 SearchForTriangulation calls
 * (cslam/src/Mapping.cpp:335) and as many Fuse calls (:503, :528) per new keyframe, each against ANOTHER keyframe's descriptors.  Search s owns the query
 * rows q_off[s] .. q_off[s+1] of q and the target rows t_off[s] .. t_off[s+1] of t (q_off[0] = t_off[0] = 0); cand_off / cand_idx / cand_dist run over ALL
 * queries back to back, and the candidate indices of a query are LOCAL to its search's target set.  Outputs as ccm_hamming_csr (best_idx local as well).
 * The _dev form takes, instead of the offsets, q_tbase[q] = first target row of the set query q searches in (nullable: one shared set). */
int ccm_hamming_csr_multi(ccm_ctx* ctx, int S, const uint8_t* q, const int32_t* q_off /* S+1 */, const uint8_t* t, const int32_t* t_off /* S+1 */,
                          const int32_t* cand_off, const int32_t* cand_idx, uint16_t* cand_dist,
                          int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);
int ccm_hamming_csr_multi_dev(ccm_ctx* ctx, const uint8_t* d_q, int Q, const uint8_t* d_t, const int32_t* d_q_tbase,
                              const int32_t* d_cand_off, const int32_t* d_cand_idx, int64_t n_cand,
                              uint16_t* d_cand_dist, int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist);

/* MapPoint::ComputeDistinctiveDescriptors (cslam/src/MapPoint.cpp:929-994), batched over P map points:
 * point p owns the descriptor rows off[p] .. off[p+1] (one per non-bad observing keyframe, in observation-map order);
 * best_local_idx[p] = index (within its own list) of the descriptor with the least median Hamming distance to the
 * others — median = element (int)(0.5*(N-1)) of the sorted row including the zero self distance, first minimum wins;
 * -1 for an empty list.  At most 256 observations per point.                                                        */
int ccm_distinctive_descriptors(ccm_ctx* ctx, const uint8_t* desc, const int32_t* off, int P, int32_t* best_local_idx);

/* DBoW2 vocabulary transform: TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
 * (cslam/thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1260) for a batch of descriptors.  The tree is passed flat:
 * node 0 is the root, children of node i are child_id[child_off[i] .. child_off[i+1]) in the vocabulary's child order
 * (a leaf has none), node_desc is n_nodes x 32 bytes, word_id / weight are per node (meaningful on leaves), L = depth.
 * Outputs per feature: word id, its weight (idf), and the node at level L - levelsup (0 when that level is <= 0).
 * BowVector accumulation / L1 normalisation and the FeatureVector map are host work on these arrays
 * (ccm_slam_amd/host: cslam::ORBVocabulary::transform; TemplatedVocabulary.h:1127-1190, BowVector.cpp:34-84).                     */
typedef struct ccm_vocab ccm_vocab;
int  ccm_vocab_create(ccm_ctx* ctx, int n_nodes, int L, const int32_t* child_off, const int32_t* child_id, const uint8_t* node_desc,
                      const int32_t* word_id, const double* weight, ccm_vocab** out);
void ccm_vocab_destroy(ccm_vocab* v);
int  ccm_bow_transform(ccm_vocab* v, const uint8_t* desc, int N, int levelsup, int32_t* word_out, double* weight_out, int32_t* node_out);

/* ---- ORB extraction -------------------------------------------------------------------
 * Replaces ORBextractor::ORBextractor / operator() (cslam/src/ORBextractor.cpp:579-639,
 * 1216-1278).  ccm_keypoint == cv::KeyPoint without class_id.                            */
typedef struct { float x, y, size, angle, response; int32_t octave; } ccm_keypoint;
typedef struct ccm_orb ccm_orb;

int  ccm_orb_create(ccm_ctx* ctx, int nfeatures, float scale_factor, int nlevels,
                    int ini_th_fast, int min_th_fast, ccm_orb** out);
void ccm_orb_destroy(ccm_orb* orb);
/* accessor mirrors of ORBextractor::GetScaleFactors() etc. (ORBextractor.h:114-136): fills
 * up to nlevels floats; which: 0 scale, 1 inv scale, 2 sigma2, 3 inv sigma2.             */
int  ccm_orb_get_table(const ccm_orb* orb, int which, float* out, int cap);
int  ccm_orb_features_per_level(const ccm_orb* orb, int32_t* out, int cap);
/* one frame, host buffers in and out.  kps/desc capacity (cap rows) should be >= ccm_orb_max_keypoints().
 * pyramid_out (nullable): nlevels caller buffers receiving the un-bordered level images
 * (mvImagePyramid, ORBextractor.h:138), each at least level_w*level_h bytes, row stride = level_w. */
int  ccm_orb_extract(ccm_orb* orb, const uint8_t* img, int w, int h, int stride,
                     ccm_keypoint* kps, uint8_t* desc, int cap, int* n_out,
                     uint8_t* const* pyramid_out);
int  ccm_orb_level_size(const ccm_orb* orb, int w, int h, int level, int* lw, int* lh);
/* upper bound of keypoints one frame can return: nfeatures + 67 per level.  DistributeOctTree stops at >= N nodes per level and may overshoot by up to 3
 * (ORBextractor.cpp:837,898), but its FIRST pass splits every root before any count is checked (:759-843): a level returns up to 4 nIni nodes whatever N is
 * (7 features on a 752 x 480 frame: 64 keypoints); nIni = round(W / H) <= 16 is assumed, wider levels are truncated at this capacity.
 * A pyramid level smaller than one 30-px cell contributes no keypoint (its image is still produced); an image whose level is more than twice as high as wide,
 * or empty, is rejected with CCM_E_ARG (the reference divides by zero / throws there). */
int  ccm_orb_max_keypoints(const ccm_orb* orb);
/* host-only (no GPU): DistributeOctTree (ORBextractor.cpp:707-931) on candidates given relative to
 * (minX,minY); sel_out receives the indices of the kept candidates in the reference's output order. */
int  ccm_orb_distribute_octree(const float* x, const float* y, const float* response, int n, int minX, int maxX,
                               int minY, int maxY, int N, int32_t* sel_out, int cap, int* n_out);
/* batch of frames already resident in HBM (d_imgs: n_frames images, tightly packed w*h each);
 * outputs stay on the device: d_kps [n_frames][cap], d_desc [n_frames][cap][32],
 * d_counts [n_frames].  Every stage runs on the device (DistributeOctTree included, as in
 * ccm_orb_extract); frames go out in groups of four per launch, two groups in flight; only a frame
 * whose candidates overflow the octree kernel's LDS plan is redone through the host octree.  */
int  ccm_orb_extract_batch_dev(ccm_orb* orb, const uint8_t* d_imgs, int n_frames, int w, int h,
                               ccm_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts);

/* Per-frame glue between extraction and matching.  A ccm_frame holds, on the device, what the
 * window searches read from a Frame / KeyFrame: undistorted keypoints (Frame::UndistortKeyPoints, Frame.cpp:284-312 —
 * cv::undistortPoints with P = K, five fixed-point iterations), the image bounds (ComputeImageBounds, :314-347), the
 * 75x48 grid (AssignFeaturesToGrid / PosInGrid, :103-118, :254-265) and the descriptors.
 * K = fx fy cx cy (mK, f32); dist = mDistCoef k1 k2 p1 p2 [k3] (n_dist 0, 4 or 5; dist[0] == 0 means "no distortion",
 * :286).  ccm_frame_window_search evaluates a batch of Frame::GetFeaturesInArea(u, v, r, minLevel, maxLevel) calls
 * (:200-253; pass -1, -1 for KeyFrame::GetFeaturesInArea, KeyFrame.cpp:1162-1201) and the Hamming distance of every
 * candidate to the query's descriptor: cand_off[Q+1], cand_idx / cand_dist in the reference's enumeration order
 * (ix-major, iy, insertion order), ready for the host's ordered resolution pass.  Call it with cap = 0 to size. */
typedef struct ccm_frame ccm_frame;
int  ccm_frame_create(ccm_ctx* ctx, const float K[4], const float* dist, int n_dist, int img_w, int img_h, ccm_frame** out);
void ccm_frame_destroy(ccm_frame* f);
int  ccm_frame_bounds(const ccm_frame* f, float bounds[4] /* mnMinX mnMinY mnMaxX mnMaxY */);
int  ccm_frame_set_keypoints(ccm_frame* f, const ccm_keypoint* kps /* mvKeys */, const uint8_t* desc /* n x 32 */, int n);
int  ccm_frame_get(ccm_frame* f, float* xy_un /* n x 2, nullable */, int32_t* cell_off /* 75*48+1, nullable */,
                   int32_t* cell_idx /* <= n, nullable */);
int  ccm_frame_window_search(ccm_frame* f, int Q, const float* u, const float* v, const float* r,
                             const int32_t* min_level, const int32_t* max_level, const uint8_t* qdesc /* Q x 32 */,
                             int32_t* cand_off /* Q+1 */, int32_t* cand_idx, uint16_t* cand_dist, int64_t cap,
                             int64_t* n_cand);

/* Frame::isInFrustum (Frame.cpp:139-198) for a batch of map points — Tracking::SearchLocalPoints' visibility loop.
 * The struct carries what the frame contributes (mRcw row-major, mtcw, mOw, intrinsics, image bounds, mfLogScaleFactor,
 * mnScaleLevels); per point: world position, mean viewing direction (GetNormal), mfMinDistance / mfMaxDistance.
 * Outputs = mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel (PredictScale, MapPoint.cpp:854-869), mTrackViewCos. */
typedef struct {
  float Rcw[9], tcw[3], Ow[3];
  float fx, fy, cx, cy;
  float minX, maxX, minY, maxY;
  float logScaleFactor;
  int32_t nScaleLevels;
} ccm_frustum_frame;
int  ccm_frame_frustum(ccm_ctx* ctx, const ccm_frustum_frame* fr, int n, const float* P /* n x 3 */,
                       const float* normal /* n x 3 */, const float* min_dist, const float* max_dist,
                       float viewing_cos_limit, uint8_t* in_view, float* proj_x, float* proj_y, int32_t* level,
                       float* view_cos);

/* MapPoint::UpdateNormalAndDepth (MapPoint.cpp:779-823) for a batch of map points — called after every BA write-back
 * (Optimizer.cpp:636, 845) and for every new / fused point.  Per point i: world position pos[i] (mWorldPos), the list
 * obs_kf[obs_off[i] .. obs_off[i+1]) of its NON-BAD observing keyframes in the order the shim iterates mObservations (the
 * reference's std::map<kfptr, size_t> is ordered by heap address, so any order is a valid reference order; the f32 sum follows the
 * list), the reference keyframe ref_kf[i] (mpRefKF) and the octave ref_level[i] of its keypoint there; kf_center = GetCameraCenter()
 * of every keyframe, scale_factors = mvScaleFactors (shared by all keyframes of a map).  Outputs mNormalVector, mfMinDistance,
 * mfMaxDistance; a point whose list is empty keeps the values passed in (the reference returns early, :796-797). */
int  ccm_update_normal_and_depth(ccm_ctx* ctx, int n_pt, const float* pos /* n_pt x 3 */, const int32_t* obs_off /* n_pt+1 */,
                                 const int32_t* obs_kf, int n_kf, const float* kf_center /* n_kf x 3 */, const int32_t* ref_kf,
                                 const int32_t* ref_level, const float* scale_factors, int n_levels,
                                 float* normal /* n_pt x 3, in/out */, float* min_dist /* in/out */, float* max_dist /* in/out */);

/* ---- bundle adjustment ----------------------------------------------------------------
 * Replaces the g2o machinery driven by Optimizer::BundleAdjustmentClient /
 * LocalBundleAdjustmentClient / MapFusionGBA (cslam/src/Optimizer.cpp:40-212, 349-644,
 * 646-859): BlockSolver_6_3 + OptimizationAlgorithmLevenberg + EdgeSE3ProjectXYZ + Huber
 * (thirdparty/g2o/g2o/core/block_solver.hpp, optimization_algorithm_levenberg.cpp,
 * types/types_six_dof_expmap.{h,cpp}, core/robust_kernel_impl.cpp).
 * All state is f64.  cam_qt rows: qx qy qz qw tx ty tz (world -> camera, as g2o::SE3Quat).
 * ccm_ba_create / ccm_ba_reset_state only COPY these arrays to their device-side staging buffers (the structure is built on the device), so every
 * pointer may address host memory (pageable or pinned) or memory of the context's device; the one-shot ccm_ba_optimize, ccm_ba_download and
 * ccm_ba_depth_positive write / read host memory. */
typedef struct {
  int32_t n_cam, n_pt, n_edge;
  double*        cam_qt;     /* [n_cam*7]  in/out                                        */
  const uint8_t* cam_fixed;  /* [n_cam]    1 = vertex->setFixed(true)                    */
  const double*  cam_K;      /* [n_cam*4]  fx fy cx cy (e->fx.. come from the KF)        */
  double*        pt_xyz;     /* [n_pt*3]   in/out                                        */
  const int32_t* e_cam;      /* [n_edge]                                                 */
  const int32_t* e_pt;       /* [n_edge]                                                 */
  const double*  e_obs;      /* [n_edge*2] keypoint (undistorted) pixel                  */
  const double*  e_info;     /* [n_edge]   invSigma2 (information = I2 * invSigma2)      */
  const uint8_t* e_level;    /* [n_edge]   nullable; g2o edge level, only level 0 is optimised */
  double         huber_delta;/* <= 0: no robust kernel                                   */
} ccm_ba_problem;

typedef struct {
  int32_t max_iters;         /* optimizer.optimize(n)                                    */
  int32_t pcg_max_iters;     /* <=0: default 1000                                        */
  double  pcg_rel_tol;       /* <=0: default 1e-8 on sqrt(r.z / r0.z0); see DESIGN.md 4.1  */
  double  lambda_init;       /* <=0: tau * max diag(H), tau = 1e-5 (levenberg.cpp:166-180) */
  int32_t verbose;
} ccm_ba_options;

typedef struct {
  int32_t iters_done;        /* value optimize() would return                            */
  int32_t lm_trials;         /* total inner trials                                       */
  int32_t pcg_iters;         /* total PCG iterations                                     */
  int32_t stop_reason;       /* 0 iters exhausted, 1 stop flag, 2 trials exhausted/rho==0, 3 chi2 stagnation, 4 solver failure */
  double  chi2_initial, chi2_final, lambda_final;
  double  ms_setup, ms_total, ms_iters;   /* host wall clock */
  int32_t n_schur_blocks;    /* upper-triangular blocks incl. diagonal                   */
  int64_t n_pair_instances;
} ccm_ba_stats;

typedef struct ccm_ba ccm_ba;
typedef void (*ccm_ba_trial_cb)(void* user, int iteration, int trial_in_iteration, double chi2_trial, int accepted);

/* one-shot: build structure, upload, optimise, write cam_qt/pt_xyz back.  Always a single-rank solve, also on a context
 * that carries a multi-rank communicator (sharding is opt-in through the staged API below, called by all ranks).
 * stop_flag (nullable) is the reference's bool* pbStopFlag, polled between LM trials.
 * chi2_per_edge (nullable, in/out [n_edge]) = e->chi2() as the caller of optimize() sees it: for
 * active (level-0) edges the value of the last evaluated LM trial, inactive edges are left
 * untouched; depth_pos (nullable, [n_edge]) = e->isDepthPositive() at the final estimate.    */
int ccm_ba_optimize(ccm_ctx* ctx, ccm_ba_problem* prob, const ccm_ba_options* opt,
                    const volatile unsigned char* stop_flag,
                    double* chi2_per_edge, uint8_t* depth_pos, ccm_ba_stats* stats);

/* staged API (bench / multi-GPU): create uploads the problem and builds the Schur structure;
 * rank/nranks shard the landmarks (each rank owns a contiguous landmark range balanced by
 * pair count; camera state is replicated).  With nranks > 1 a communicator must be attached
 * before ccm_ba_run, and ccm_ba_run / ccm_ba_download are COLLECTIVE calls: every rank makes them with
 * the same options, and either every rank passes a stop flag or none does (the flag is reduced over
 * the ranks with each trial's scalars, so a flag raised on one rank stops all of them at the same trial). */
int  ccm_ba_create(ccm_ctx* ctx, const ccm_ba_problem* prob, int rank, int nranks, ccm_ba** out);
void ccm_ba_destroy(ccm_ba* ba);
int  ccm_ba_reset_state(ccm_ba* ba, const double* cam_qt, const double* pt_xyz); /* re-upload initial state */
/* SparseOptimizer::push() / pop() for all vertices (sparse_optimizer.cpp:600-613): save the current estimate on the device /
 * make the saved estimate current again (stream-ordered device copies; one level, a second push overwrites the first) */
int  ccm_ba_push_state(ccm_ba* ba);
int  ccm_ba_pop_state(ccm_ba* ba);
int  ccm_ba_run(ccm_ba* ba, const ccm_ba_options* opt, const volatile unsigned char* stop_flag,
                ccm_ba_stats* stats);
int  ccm_ba_download(ccm_ba* ba, double* cam_qt, double* pt_xyz, double* chi2_per_edge);
/* Second stage of Optimizer::LocalBundleAdjustmentClient on the SAME handle (Optimizer.cpp:545-566: outlier edges -> setLevel(1), robust kernel off,
 * optimize(10)): edges with e_level[e] != 0 ([n_edge], the caller's numbering) leave the optimisation, the Huber delta is replaced (<= 0: none), the
 * estimate stays; the next ccm_ba_run is the reference's initializeOptimization(0) + optimize(n) on the reduced edge set.  The call is a pure function of
 * (e_level, huber_delta) and of the informations given to ccm_ba_create: an edge whose level goes back to 0 takes part again with its original information
 * (a handle re-run after ccm_ba_pop_state passes its first-stage levels and delta again).  Only edges inactive at ccm_ba_create are not part of the handle
 * and cannot come back.  A deactivated edge contributes exact zeros whatever its residual is and keeps reporting the chi2 of the last pass it took part in. */
int  ccm_ba_set_edge_levels(ccm_ba* ba, const uint8_t* e_level, double huber_delta);
/* per-iteration record of the last ccm_ba_run: robust chi2 after the iteration, lambda after it, LM trials it took
 * (what g2o prints with setVerbose(true), sparse_optimizer.cpp:400-410); fills min(*n_iters, cap) entries */
int  ccm_ba_history(const ccm_ba* ba, int cap, double* chi2_per_iter, double* lambda_per_iter, int32_t* trials_per_iter,
                    int* n_iters);
/* called on the optimising thread after every LM trial, before the stop flag is polled (the place where
 * OptimizationAlgorithmLevenberg::solve tests terminate(), optimization_algorithm_levenberg.cpp:150); a caller can use it
 * to watch progress or to decide when to raise its stop flag.  cb == NULL removes it. */
int  ccm_ba_set_trial_callback(ccm_ba* ba, ccm_ba_trial_cb cb, void* user);
/* e->isDepthPositive() for every edge of the problem at the given state (host arithmetic, O(n_edge)) */
int  ccm_ba_depth_positive(const ccm_ba_problem* prob, const double* cam_qt, const double* pt_xyz, uint8_t* depth_pos);
/* host-only: split n landmark weights into nranks contiguous ranges (begin_out has nranks+1 entries) */
int  ccm_ba_partition(const int64_t* weight, int n, int nranks, int32_t* begin_out);
/* algorithmic byte count of one LM trial for the roofline */
int  ccm_ba_counts(const ccm_ba* ba, int64_t* n_active_edges, int64_t* n_active_pts,
                   int64_t* n_free_cams, int64_t* n_blocks, int64_t* n_pairs);


/* RCCL communicator for the sharded GBA.  id_bytes is an ncclUniqueId (128 bytes) produced by
 * ccm_comm_unique_id on rank 0 and broadcast by the launcher (bench.py uses torch.distributed). */
int ccm_comm_unique_id(uint8_t id_bytes[128]);
int ccm_comm_init(ccm_ctx* ctx, int nranks, int rank, const uint8_t id_bytes[128]);
int ccm_comm_destroy(ccm_ctx* ctx);

/* motion-only pose optimisation: Optimizer::PoseOptimizationClient (Optimizer.cpp:215-347):
 * 4 rounds x 10 LM iterations on one SE3 vertex with unary EdgeSE3ProjectXYZOnlyPose edges,
 * Huber sqrt(5.991) (dropped for the last round), chi2 threshold 5.991, dense 6x6 solve.
 * cam_qt in: Frame.mTcw as SE3Quat, out: optimised pose.  outlier[n] = Frame.mvbOutlier.
 * Returns via n_inlier the reference's return value (nInitialCorrespondences - nBad).     */
int ccm_pose_optimize(ccm_ctx* ctx, double cam_qt[7], int n, const double* Xw /*n*3*/,
                      const double* obs /*n*2*/, const double* info /*n*/, const double K[4],
                      uint8_t* outlier, int* n_inlier);

/* Sim3 between two keyframes from matched map points: Optimizer::OptimizeSim3 (Optimizer.cpp:861-1056).
 * sim3 = g2oS12 as [qx qy qz qw tx ty tz s] (in: Sim3Solver estimate, out: optimised; untouched when the call
 * returns *n_inlier = 0 because fewer than 10 pairs survive the first pass, :1015-1016).  Per valid pair i
 * (the shim filters null / bad map points and i2 < 0, :911-947): P1c = R1w*X1+t1w and P2c = R2w*X2+t2w (the fixed
 * VertexSBAPointXYZ estimates), obs1/obs2 = undistorted keypoints in KF1/KF2, info = mvInvLevelSigma2[octave].
 * Edges EdgeSim3ProjectXYZ / EdgeInverseSim3ProjectXYZ (types_seven_dof_expmap.h:133-172) with g2o's numeric
 * Jacobians (base_binary_edge.hpp:129-196), Huber (float)sqrt(th2), optimize(5) + optimize(10|5).
 * inlier[i] = 0 where the reference nulls vpMatches1[idx]. */
int  ccm_sim3_optimize(ccm_ctx* ctx, double sim3[8], int n, const double* P1c, const double* P2c,
                       const double* obs1, const double* obs2, const double* info1, const double* info2,
                       const double K1[4], const double K2[4], double th2, int fix_scale,
                       uint8_t* inlier, int* n_inlier);

/* Pose-graph numerics of Optimizer::OptimizeEssentialGraphLoopClosure / MapFusion (Optimizer.cpp:1058-1331, 1333-1566):
 * vertices = keyframes with estimate Siw as [qx qy qz qw tx ty tz s] (fixed[v] = the loop keyframe, fix_scale =
 * bFixScale), edges = EdgeSim3 with vertex(0) = e_i, vertex(1) = e_j, measurement Sji (same 8-double layout) and
 * information I7 (types_seven_dof_expmap.h:98-121); g2o's numeric Jacobians (base_binary_edge.hpp:129-196),
 * BlockSolver_7_3 + Levenberg with setUserLambdaInit(lambda_init = 1e-16 in the reference; <= 0 selects g2o's
 * tau * max-diagonal rule), optimize(max_iters = 20).  The caller keeps the graph walk that chooses the edges
 * (spanning tree, loop edges, covisibility >= minFeat, :1122-1260) and the SE3 / map-point write-back (:1268-1330).
 * The linear solve is a dense f64 Cholesky on the device (exact, like Eigen's sparse LDLT in the reference); the environment
 * variable CCM_PG_SOLVER=pcg selects a tree-preconditioned PCG instead (faster on large graphs, inexact-Newton path). */
typedef struct {
  int32_t iters_done, lm_trials, pcg_iters, reserved;
  double chi2_initial, chi2_final, lambda_final;
} ccm_pg_stats;
int  ccm_pose_graph_optimize(ccm_ctx* ctx, int n_vert, double* sim3 /* n_vert x 8, in/out */, const uint8_t* fixed,
                             int fix_scale, int n_edge, const int32_t* e_i, const int32_t* e_j,
                             const double* meas /* n_edge x 8 */, int max_iters, double lambda_init,
                             const volatile unsigned char* stop_flag /* nullable */, ccm_pg_stats* stats /* nullable */);

#ifdef __cplusplus
}
#endif
#endif /* CCM_HIP_H */
