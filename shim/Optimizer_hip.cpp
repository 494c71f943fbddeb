// Optimizer_hip.cpp — DROP-IN replacement for the translation unit cslam/src/Optimizer.cpp of the reference.
//
// It defines the static methods that the reference's own header declares (cslam/include/cslam/Optimizer.h:79-112) — same names, same
// signatures, compiled against that header — and does what Optimizer.cpp does around g2o: the graph walks that choose vertices and
// edges, the f32 -> f64 conversions on the way in (Converter.cc:40-119, here ccm_convert.h), the write-back with its side effects
// (SetPose, SetWorldPos, UpdateNormalAndDepth, EraseObservation, mTcwGBA ...).  The optimisation itself — everything that was g2o —
// goes through the C ABI of libccm_hip.so (include/ccm_hip.h) to the MI355X: no g2o optimiser, solver, vertex or edge object is created.
// The only g2o type that appears is the VALUE type g2o::Sim3 (header-only arithmetic), because it is part of the reference's interface
// (OptimizeSim3's g2oS12, KeyFrameAndPose) and the essential-graph functions compose their edge measurements from it exactly as the
// reference does.  cslam::Converter (Converter.cc) is a separate translation unit of the reference and stays in the build.
//
// Build: shim/Makefile.  In this repository the map classes are the look-alikes of oracle/ref_shim/cslam_lookalike (the real KeyFrame.h /
// MapPoint.h / Map.h need ROS, PCL, DBoW2 and cereal, none of which is installed); the member names and types used here are those of
// the real headers, so the same file compiles in a catkin workspace with `Optimizer.cpp` replaced by it in cslam/CMakeLists.txt:122-145.
// tests/test_shim_gpu.py runs this file and the reference's Optimizer.cpp on identical synthetic maps and compares what they leave
// behind in the map.
//
// LICENCE AND PROVENANCE.  This file is a derived work of cslam/src/Optimizer.cpp of CCM-SLAM (Copyright (C) Patrik Schmuck, ETH Zurich; GNU GPL v3 or later; itself based
// on ORB-SLAM2 by Raul Mur-Artal) and is distributed under the same licence.  Everything numerical is new (it is a call into libccm_hip.so), but a drop-in must choose the
// SAME vertices and edges in the SAME order and mutate the map in the SAME order as the reference, so these passages follow Optimizer.cpp statement by statement —
// reformatted, with the reference's variable names kept so that a maintainer can diff them — and runs of a few lines of them are textually the reference's:
//   * BundleAdjustmentClient / MapFusionGBA: the vertex / edge collection loops and the write-back loops with the mTcwGBA / mPosGBA branch   (Optimizer.cpp:55-212, 668-859)
//   * PoseOptimizationClient: the edge set-up loop over Frame.mvpMapPoints / mvKeysUn and the outlier bookkeeping of the four rounds         (Optimizer.cpp:235-347)
//   * LocalBundleAdjustmentClient: the local-window walk (local keyframes, local points, fixed cameras), the erase loop with its
//     SetNotErase wait, the pose / point write-back                                                                                         (Optimizer.cpp:351-404, 568-644)
//   * OptimizeSim3: correspondence collection and the inlier bookkeeping                                                                     (Optimizer.cpp:880-1056)
//   * OptimizeEssentialGraphLoopClosure / MapFusion: the four edge walks (loop connections, spanning tree, loop edges, covisibility >= 100)
//     and the corrected-pose / point write-back                                                                                             (Optimizer.cpp:1122-1331, 1376-1566)
// What the shim prints (its fatal conditions, device errors) is its own wording; what it throws is the reference's exception type, because callers catch that.
#include <cslam/Optimizer.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <unistd.h>   // usleep (the wait loops of the reference around SetNotErase, Optimizer.cpp:620-626)

#include "../include/ccm_hip.h"
#include "../ccm_slam_amd/host/ccm_convert.h"

namespace cslam {
namespace {

// Fatal conditions of the graph walks: the reference prints a message and throws estd::infrastructure_ex (Optimizer.cpp:87-90, 480-483, 595-598, 663-666); a drop-in
// keeps the EXCEPTION TYPE (callers catch it) and says in its own words what was wrong and where.
[[noreturn]] void shim_fatal(const char* method, const char* what) {
  std::cout << "[ccm_hip shim] " << method << ": " << what << " (infrastructure_ex)" << std::endl;
  throw estd::infrastructure_ex();
}

// one device context per calling thread: tracking, local mapping and the server's optimisation threads call concurrently
ccm_ctx* thread_ctx() {
  struct Holder {
    ccm_ctx* c = nullptr;
    ~Holder() { if (c) ccm_ctx_destroy(c); }
  };
  static thread_local Holder h;
  if (!h.c) {
    const char* dev = std::getenv("CCM_DEVICE");
    if (ccm_ctx_create(dev ? std::atoi(dev) : 0, &h.c) != CCM_OK) {
      cout << COUTFATAL << "no MI355X context: " << ccm_last_error(nullptr) << endl;
      throw infrastructure_ex();
    }
  }
  return h.c;
}
void check(int rc, const char* what) {
  if (rc != CCM_OK) {
    cout << COUTFATAL << what << ": " << ccm_last_error(thread_ctx()) << endl;
    throw infrastructure_ex();
  }
}

// wall-clock phases of the last bundle-adjustment call made by this thread (ms): [0] graph walk (vertices + edges gathered), [1] flatten (ids -> indices,
// f32 -> f64), [2] ccm_ba_create (structure build: g2o's initializeOptimization + buildStructure), [3] ccm_ba_run (optimize(n)), [4] download (+ depth
// test), [5] keyframe write-back, [6] map-point write-back (SetWorldPos + UpdateNormalAndDepth), [7] whole call.  Read with ccm_shim_last_phases().
// (round 5) [8] GetAll* + camera vertices, [9] dropping the call's references into the flat problem, [10] SCOPE EXIT: everything the call's locals free.  On the
// 4-agent map this was the largest unlabelled part of the call (29 ms after the last lap).  It is not the 150 000 shared_ptr copies Map::GetAllMapPoints() hands
// out (those are dropped inside the write-back loop now, by the thread that has just worked on the point) but glibc: the threaded write-back frees the points' old
// cv::Mat buffers — allocated by whichever thread built the map — back into THAT thread's arena, ~2 fastbin chunks per point, and the first large free() / malloc()
// that arena sees afterwards (here: the 2.4 MB pointer vector at scope exit) runs malloc_consolidate over all of them (measured with MALLOC_ARENA_MAX=1: the phase
// vanishes; on one thread there is no cross-arena garbage either).  With the optional MapPoint setter the write-back reuses the buffers and the
// phase is empty.  [11] what is still unaccounted (total - sum of the others).  Read with ccm_shim_phases().
constexpr int kPhases = 12;
thread_local double g_phase[kPhases] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct PhaseClock {
  double t0, t;
  PhaseClock() : t0(now_ms()), t(t0) { for (double& v : g_phase) v = 0; }
  void lap(int i) { const double n = now_ms(); g_phase[i] += n - t; t = n; }
  ~PhaseClock() {
    g_phase[7] = now_ms() - t0;
    double sum = 0;
    for (int i = 0; i < kPhases - 1; i++) if (i != 7) sum += g_phase[i];
    g_phase[kPhases - 1] = g_phase[7] - sum;
  }
};

// host threads for the per-map-point work of a global bundle adjustment (graph walk, write-back): every map point is independent and the
// reference's accessors take the point's own mutexes, so contiguous chunks of vpMP are handled by a few threads and merged in order.
// Measured on the 4-agent map (150 000 points, 256-core host): walk 87 ms on one thread, 36 on 8, 34 on 32 (shared_ptr reference counts of
// the 2000 keyframes bounce between the cores); write-back 230 / 46 / 17 ms.  CCM_SHIM_THREADS overrides (1 = the calling thread only).
int shim_threads(size_t n_items, unsigned cap = 8) {
  static const int env = std::getenv("CCM_SHIM_THREADS") ? std::atoi(std::getenv("CCM_SHIM_THREADS")) : 0;
  int t = env > 0 ? env : (int)std::min<unsigned>(std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 1u, cap);
  if (n_items < 20000) t = 1;
  return std::max(1, t);
}
template <typename F>
void parallel_chunks(size_t n, int n_thr, F fn /* (chunk index, begin, end) */) {
  if (n_thr <= 1) { fn(0, (size_t)0, n); return; }
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> err((size_t)n_thr);
  for (int t = 0; t < n_thr; t++)
    th.emplace_back([&, t]() {
      try { fn(t, n * (size_t)t / (size_t)n_thr, n * (size_t)(t + 1) / (size_t)n_thr); } catch (...) { err[(size_t)t] = std::current_exception(); }
    });
  for (auto& x : th) x.join();
  for (auto& e : err) if (e) std::rethrow_exception(e);
}

// the call's own copies of the caller's pointer vectors, released on the host threads that have just written the objects back (their reference counts sit in
// those cores' caches; one thread doing 150 000 lock-prefixed decrements on remote lines took ~29 ms at the end of MapFusionGBA on the 4-agent map)
template <typename P>
void release_refs(std::vector<P>& v) {
  parallel_chunks(v.size(), shim_threads(v.size(), 32), [&](int, size_t b, size_t e) { for (size_t i = b; i < e; i++) v[i].reset(); });
  std::vector<P>().swap(v);
}

// vertex id -> index: a direct table when the ids are compact (mUniqueId, GetID of a few clients), a sorted list otherwise
struct IdIndex {
  std::vector<int32_t> table;
  std::vector<std::pair<size_t, int32_t>> sorted;
  void build(const std::vector<size_t>& ids) {
    table.clear(); sorted.clear();
    size_t mx = 0;
    for (size_t v : ids) mx = std::max(mx, v);
    if (mx <= 8 * ids.size() + 4096) {
      table.assign(mx + 1, -1);
      for (size_t i = 0; i < ids.size(); i++) table[ids[i]] = (int32_t)i;
    } else {
      sorted.reserve(ids.size());
      for (size_t i = 0; i < ids.size(); i++) sorted.push_back({ids[i], (int32_t)i});
      std::sort(sorted.begin(), sorted.end());
    }
  }
  int32_t find(size_t id) const {
    if (!table.empty()) return id < table.size() ? table[id] : -1;
    auto it = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(id, (int32_t)INT32_MIN));
    return (it != sorted.end() && it->first == id) ? it->second : -1;
  }
};

// a flat bundle-adjustment problem under construction; cameras and points are numbered in g2o VERTEX-ID order, the order in which g2o
// itself sorts its active vertices (sparse_optimizer.cpp:482-487).  Edges keep their insertion order (the callers index per-edge results by it).
struct FlatBA {
  std::vector<size_t> cam_id, pt_id;                 // insertion order until flatten() sorts them by id
  std::vector<Optimizer::kfptr> cam_kf;
  std::vector<char> cam_is_fixed;
  std::vector<MapPoint*> pt_mp;                      // raw: the caller's containers keep the points alive, and 150 000 shared_ptr copies cost ~15 ms of cache-missing atomics each way
  std::vector<size_t> e_cam_id, e_pt_id;             // per edge: vertex ids
  // per point, only filled when MapPoint offers SetNormalAndDepth (batched UpdateNormalAndDepth of the write-back): id of the reference keyframe, octave of the
  // point's keypoint there, and whether EVERY non-bad observation became an edge (else the point takes the reference's own method)
  std::vector<size_t> pt_ref_cam_id;
  std::vector<int32_t> pt_ref_level;
  std::vector<char> pt_regular;
  bool aux_ok = true;                                // false once flatten() had to reorder the points (the per-point arrays above are in insertion order)
  std::vector<double> pt_xyz_walk;                   // optional: Converter::toVector3d(pMP->GetWorldPos()) taken by the graph walk while it holds the point (3 per point, insertion order)
  // flattened
  std::vector<double> cam_qt, cam_K, pt_xyz, e_obs, e_info;
  std::vector<uint8_t> cam_fix, e_level;
  std::vector<int32_t> e_cam, e_pt;
  IdIndex cam_index, pt_index;

  size_t nEdges() const { return e_cam_id.size(); }
  void reset() {   // keeps every vector's capacity: the global BA of a 4-agent map builds ~60 MB of flat arrays, and allocating (first-touch page faults) and
                   // releasing (munmap) them cost ~50 ms per call — the server thread keeps one FlatBA for its lifetime instead
    cam_id.clear(); pt_id.clear(); cam_kf.clear(); cam_is_fixed.clear(); pt_mp.clear(); e_cam_id.clear(); e_pt_id.clear();
    pt_ref_cam_id.clear(); pt_ref_level.clear(); pt_regular.clear(); aux_ok = true; pt_xyz_walk.clear();
    cam_qt.clear(); cam_K.clear(); pt_xyz.clear(); e_obs.clear(); e_info.clear(); cam_fix.clear(); e_level.clear(); e_cam.clear(); e_pt.clear();
  }
  void addCam(size_t id, Optimizer::kfptr kf, bool is_fixed) { cam_id.push_back(id); cam_kf.push_back(kf); cam_is_fixed.push_back(is_fixed ? 1 : 0); }
  void addPoint(size_t id, const Optimizer::mpptr& mp) { pt_id.push_back(id); pt_mp.push_back(mp.get()); }
  void addPointPos(const cv::Mat& P) { const float p[3] = {P.at<float>(0), P.at<float>(1), P.at<float>(2)}; double d[3]; ccmh::toVector3d(p, d); pt_xyz_walk.insert(pt_xyz_walk.end(), d, d + 3); }
  void addPointAux(size_t ref_cam_id, int level, bool regular) { pt_ref_cam_id.push_back(ref_cam_id); pt_ref_level.push_back(level); pt_regular.push_back(regular ? 1 : 0); }
  void addEdge(size_t p_id, const Optimizer::kfptr& kf, size_t c_id, const cv::KeyPoint& kpUn) {
    const float& invSigma2 = kf->mvInvLevelSigma2[kpUn.octave];
    e_cam_id.push_back(c_id); e_pt_id.push_back(p_id);
    e_obs.push_back((double)kpUn.pt.x); e_obs.push_back((double)kpUn.pt.y); e_info.push_back((double)invSigma2);
  }
  void removePoint(size_t id) {   // the callers remove the point they added last (a point without edges)
    for (size_t i = pt_id.size(); i-- > 0;) if (pt_id[i] == id) { pt_id.erase(pt_id.begin() + i); pt_mp.erase(pt_mp.begin() + i); return; }
  }
  void append(FlatBA& o) {        // merge of a thread's part (points and edges of a later chunk of vpMP)
    pt_id.insert(pt_id.end(), o.pt_id.begin(), o.pt_id.end()); pt_mp.insert(pt_mp.end(), o.pt_mp.begin(), o.pt_mp.end());
    e_cam_id.insert(e_cam_id.end(), o.e_cam_id.begin(), o.e_cam_id.end()); e_pt_id.insert(e_pt_id.end(), o.e_pt_id.begin(), o.e_pt_id.end());
    e_obs.insert(e_obs.end(), o.e_obs.begin(), o.e_obs.end()); e_info.insert(e_info.end(), o.e_info.begin(), o.e_info.end());
    pt_ref_cam_id.insert(pt_ref_cam_id.end(), o.pt_ref_cam_id.begin(), o.pt_ref_cam_id.end());
    pt_ref_level.insert(pt_ref_level.end(), o.pt_ref_level.begin(), o.pt_ref_level.end()); pt_regular.insert(pt_regular.end(), o.pt_regular.begin(), o.pt_regular.end());
    pt_xyz_walk.insert(pt_xyz_walk.end(), o.pt_xyz_walk.begin(), o.pt_xyz_walk.end());
  }
  template <typename P>
  static void sort_by_id(std::vector<size_t>& ids, std::vector<P>& ptrs, std::vector<char>* flags) {
    if (std::is_sorted(ids.begin(), ids.end())) return;
    std::vector<size_t> perm(ids.size());
    for (size_t i = 0; i < perm.size(); i++) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return ids[a] < ids[b]; });
    std::vector<size_t> ids2(ids.size()); std::vector<P> p2(ids.size()); std::vector<char> f2(flags ? ids.size() : 0);
    for (size_t i = 0; i < perm.size(); i++) { ids2[i] = ids[perm[i]]; p2[i] = ptrs[perm[i]]; if (flags) f2[i] = (*flags)[perm[i]]; }
    ids.swap(ids2); ptrs.swap(p2); if (flags) flags->swap(f2);
  }
  // g2o refuses an edge whose camera vertex does not exist (optimizer.vertex(id) == 0 -> addEdge fails): such observations are dropped
  void flatten(bool drop_edges_without_camera = false) {
    sort_by_id(cam_id, cam_kf, &cam_is_fixed);
    if (!std::is_sorted(pt_id.begin(), pt_id.end())) aux_ok = false;
    sort_by_id<MapPoint*>(pt_id, pt_mp, nullptr);
    cam_index.build(cam_id); pt_index.build(pt_id);
    const size_t nc = cam_id.size(), np = pt_id.size();
    cam_qt.resize(7 * nc); cam_K.resize(4 * nc); cam_fix.resize(nc); pt_xyz.resize(3 * np);
    for (size_t i = 0; i < nc; i++) {
      const cv::Mat Tcw = cam_kf[i]->GetPose();                                      // Converter::toSE3Quat(pKF->GetPose())
      float T[16];
      for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T[4 * r + c] = Tcw.at<float>(r, c);
      ccmh::toSE3Quat(T, &cam_qt[7 * i]);
      cam_K[4 * i] = cam_kf[i]->fx; cam_K[4 * i + 1] = cam_kf[i]->fy; cam_K[4 * i + 2] = cam_kf[i]->cx; cam_K[4 * i + 3] = cam_kf[i]->cy;
      cam_fix[i] = cam_is_fixed[i] ? 1 : 0;
    }
    if (aux_ok && pt_xyz_walk.size() == 3 * np) pt_xyz.swap(pt_xyz_walk);             // the walk already read every position (same order: no point was moved)
    else
    parallel_chunks(np, shim_threads(np), [&](int, size_t b, size_t e) {
      for (size_t i = b; i < e; i++) {
        const cv::Mat P = pt_mp[i]->GetWorldPos();                                   // Converter::toVector3d(pMP->GetWorldPos())
        const float p[3] = {P.at<float>(0), P.at<float>(1), P.at<float>(2)};
        ccmh::toVector3d(p, &pt_xyz[3 * i]);
      }
    });
    const size_t ne = e_cam_id.size();
    e_cam.resize(ne); e_pt.resize(ne);
    // ids -> indices: two table look-ups per observation (955 000 on the 4-agent map: ~7 ms on one thread), independent from edge to edge
    std::vector<char> missing((size_t)shim_threads(ne) + 1, 0);
    parallel_chunks(ne, shim_threads(ne), [&](int t, size_t b, size_t e) {
      char miss = 0;
      for (size_t k = b; k < e; k++) { const int32_t ci = cam_index.find(e_cam_id[k]); e_cam[k] = ci; e_pt[k] = pt_index.find(e_pt_id[k]); miss |= ci < 0; }
      missing[(size_t)t] = miss;
    });
    bool any_missing = false;
    for (char m : missing) any_missing |= m != 0;
    size_t w = ne;
    if (any_missing && drop_edges_without_camera) {   // (rare: an observation of a keyframe that is not a vertex) ordered compaction
      w = 0;
      for (size_t k = 0; k < ne; k++) {
        if (e_cam[k] < 0) continue;
        if (w != k) { e_cam[w] = e_cam[k]; e_pt[w] = e_pt[k]; e_cam_id[w] = e_cam_id[k]; e_pt_id[w] = e_pt_id[k]; e_obs[2 * w] = e_obs[2 * k]; e_obs[2 * w + 1] = e_obs[2 * k + 1]; e_info[w] = e_info[k]; }
        w++;
      }
    }
    e_cam.resize(w); e_pt.resize(w); e_cam_id.resize(w); e_pt_id.resize(w); e_obs.resize(2 * w); e_info.resize(w);
    e_level.assign(w, 0);
  }
  ccm_ba_problem problem(double huber) {
    ccm_ba_problem P;
    P.n_cam = (int32_t)cam_id.size(); P.n_pt = (int32_t)pt_id.size(); P.n_edge = (int32_t)e_cam.size();
    P.cam_qt = cam_qt.data(); P.cam_fixed = cam_fix.data(); P.cam_K = cam_K.data(); P.pt_xyz = pt_xyz.data();
    P.e_cam = e_cam.data(); P.e_pt = e_pt.data(); P.e_obs = e_obs.data(); P.e_info = e_info.data(); P.e_level = e_level.data();
    P.huber_delta = huber;
    return P;
  }
  cv::Mat camPose(size_t id) {                                                       // Converter::toCvMat(vSE3->estimate())
    float T[16];
    ccmh::toCvMat(&cam_qt[7 * (size_t)cam_index.find(id)], T);
    cv::Mat m(4, 4, CV_32F);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m.at<float>(r, c) = T[4 * r + c];
    return m;
  }
  cv::Mat pointPos(size_t id) {                                                      // Converter::toCvMat(vPoint->estimate())
    cv::Mat m(3, 1, CV_32F);
    const size_t i = (size_t)pt_index.find(id);
    for (int c = 0; c < 3; c++) m.at<float>(c) = (float)pt_xyz[3 * i + c];
    return m;
  }
};

// optimizer.initializeOptimization(0); optimizer.optimize(n) on the device.  keep != nullptr: the handle outlives the call (*keep == nullptr: it is
// created here; otherwise the problem of the earlier call is reused: the edges whose e_level became non-zero leave, the estimate stays — the second
// stage of the local BA, Optimizer.cpp:545-566); the caller releases it with ccm_ba_destroy.
void run_ba(FlatBA& f, double huber, int iterations, bool* pbStopFlag, std::vector<double>* chi2, std::vector<uint8_t>* depth_pos, ccm_ba** keep = nullptr) {
  ccm_ba_problem P = f.problem(huber);
  ccm_ba_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.max_iters = iterations;
  if (chi2) chi2->resize(f.nEdges(), 0.0);
  if (depth_pos) depth_pos->resize(f.nEdges(), 1);
  // = ccm_ba_optimize (one rank: a one-shot call never turns into a collective), split so that the phases can be read
  ccm_ctx* ctx = thread_ctx();
  ccm_ba* ba = keep ? *keep : nullptr;
  double t = now_ms();
  if (!ba) check(ccm_ba_create(ctx, &P, 0, 1, &ba), "ccm_ba_create");
  else check(ccm_ba_set_edge_levels(ba, f.e_level.data(), huber), "ccm_ba_set_edge_levels");
  double n = now_ms(); g_phase[2] += n - t; t = n;
  int rc = ccm_ba_run(ba, &opt, reinterpret_cast<const volatile unsigned char*>(pbStopFlag), nullptr);
  n = now_ms(); g_phase[3] += n - t; t = n;
  if (rc == CCM_OK) rc = ccm_ba_download(ba, P.cam_qt, P.pt_xyz, chi2 ? chi2->data() : nullptr);
  if (rc == CCM_OK && depth_pos) rc = ccm_ba_depth_positive(&P, P.cam_qt, P.pt_xyz, depth_pos->data());
  if (keep) *keep = ba; else ccm_ba_destroy(ba);
  g_phase[4] += now_ms() - t;
  check(rc, "ccm_ba_run");
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// batched MapPoint::UpdateNormalAndDepth for the write-back of a global BA — only when MapPoint offers SetNormalAndDepth (the optional patch of
// INTEGRATION.md; the reference keeps mNormalVector / mfMinDistance / mfMaxDistance protected without a setter, MapPoint.h:286-305).  Detected at compile
// time, so this translation unit builds against the unpatched header too and then calls the reference's method per point.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, typename = void> struct has_normal_depth_setter : std::false_type {};
template <typename T>
struct has_normal_depth_setter<T, decltype(std::declval<T&>().SetNormalAndDepth(std::declval<const cv::Mat&>(), 0.0f, 0.0f), void())> : std::true_type {};
constexpr bool kBatchedNormals = has_normal_depth_setter<MapPoint>::value;

template <typename MP> void store_normal_depth(MP* p, const float* n3, float mn, float mx, std::true_type) {
  cv::Mat n(3, 1, CV_32F);
  n.at<float>(0) = n3[0]; n.at<float>(1) = n3[1]; n.at<float>(2) = n3[2];
  p->SetNormalAndDepth(n, mn, mx);
}
template <typename MP> void store_normal_depth(MP*, const float*, float, float, std::false_type) {}

// SetWorldPos + UpdateNormalAndDepth (Optimizer.cpp:843-845) of every point of the flattened problem: positions by SetWorldPos as before; normal / depth
// range of the REGULAR points (every non-bad observation is an edge of the problem, reference keyframe among its cameras) by ONE call of
// ccm_update_normal_and_depth (MapPoint.cpp:779-823 on the device, bit-exact: tests/test_frame_gpu.py) over the problem's own edge lists — a point's edges
// are in the order the walk iterated its mObservations, which is the order the reference sums in; the other points take the reference's method.
// edge_skip (local BA): observations erased after the optimisation (they are no longer in mObservations); ref_from_walk: the reference keyframe and octave
// were recorded by the graph walk (global BA: no second GetObservations()), else they are asked of the point now (local BA: EraseObservation may have moved
// mpRefKF, MapPoint.cpp:474-476); pos_lock: SetWorldPos's bLock as the reference call site passes it.
void batched_point_writeback(FlatBA& f, const std::vector<char>* edge_skip, bool ref_from_walk, bool pos_lock, bool ids_are_unique_ids) {
  const size_t np = f.pt_id.size(), nc = f.cam_id.size(), ne = f.e_pt.size();
  std::vector<float> pos(3 * np), center(3 * nc), normal(3 * np, 0.0f), dmin(np, 0.0f), dmax(np, 0.0f);
  std::vector<int32_t> off(np + 1, 0), kf(ne), ref(np, 0), lvl(np, 0);
  for (size_t i = 0; i < np; i++) for (int c = 0; c < 3; c++) pos[3 * i + c] = (float)f.pt_xyz[3 * i + c];   // what pointPos() hands to SetWorldPos
  for (size_t i = 0; i < nc; i++) {
    const cv::Mat Ow = f.cam_kf[i]->GetCameraCenter();                                                      // after the keyframe write-back
    for (int c = 0; c < 3; c++) center[3 * i + c] = Ow.at<float>(c);
  }
  for (size_t k = 0; k < ne; k++) if (!edge_skip || !(*edge_skip)[k]) off[(size_t)f.e_pt[k] + 1]++;
  for (size_t i = 0; i < np; i++) off[i + 1] += off[i];
  {
    std::vector<int32_t> fill(off.begin(), off.end() - 1);
    for (size_t k = 0; k < ne; k++) if (!edge_skip || !(*edge_skip)[k]) kf[(size_t)fill[(size_t)f.e_pt[k]]++] = f.e_cam[k];   // stable: the walk's order inside a point
  }
  std::vector<char> regular(np, 0);
  for (size_t i = 0; i < np; i++) {
    int32_t r = -1;
    if (ref_from_walk) { r = f.pt_regular[i] ? f.cam_index.find(f.pt_ref_cam_id[i]) : -1; lvl[i] = f.pt_ref_level[i]; }
    else if (!f.pt_mp[i]->isBad()) {
      const Optimizer::kfptr pRef = f.pt_mp[i]->GetReferenceKeyFrame();
      const int idx = pRef ? f.pt_mp[i]->GetIndexInKeyFrame(pRef) : -1;
      if (pRef && idx >= 0) { r = f.cam_index.find(ids_are_unique_ids ? pRef->mUniqueId : (size_t)Optimizer::GetID(pRef->mId, true)); lvl[i] = pRef->mvKeysUn[idx].octave; }
    }
    regular[i] = r >= 0 && off[i + 1] > off[i];
    ref[i] = r >= 0 ? r : 0;
  }
  const std::vector<float>& sf = f.cam_kf[0]->mvScaleFactors;                                                // one table per map (ORBextractor parameters)
  check(ccm_update_normal_and_depth(thread_ctx(), (int)np, pos.data(), off.data(), kf.data(), (int)nc, center.data(), ref.data(), lvl.data(), sf.data(),
                                    (int)f.cam_kf[0]->mnScaleLevels, normal.data(), dmin.data(), dmax.data()), "ccm_update_normal_and_depth");
  parallel_chunks(np, shim_threads(np, 32), [&](int, size_t b, size_t e) {
    for (size_t i = b; i < e; i++) {
      MapPoint* pMP = f.pt_mp[i];
      if (pMP->isBad()) continue;
      cv::Mat p(3, 1, CV_32F);
      for (int c = 0; c < 3; c++) p.at<float>(c) = pos[3 * i + c];
      // A point whose position is locked may ignore the write (MapPoint::SetWorldPos returns early on a CLIENT, MapPoint.cpp:340-341): the batch values were
      // computed from the NEW position, so such a point takes the reference's own per-point method, which reads whatever position the point really has.
      const bool locked = pMP->IsPosLocked();
      pMP->SetWorldPos(p, pos_lock);
      if (regular[i] && !locked) store_normal_depth(pMP, &normal[3 * i], dmin[i], dmax[i], has_normal_depth_setter<MapPoint>());
      else pMP->UpdateNormalAndDepth();
    }
  });
}


// ---------------------------------------------------------------------------------------------------------------------------------
// client side
// ---------------------------------------------------------------------------------------------------------------------------------
void Optimizer::GlobalBundleAdjustemntClient(mapptr pMap, size_t ClientId, int nIterations, bool* pbStopFlag, const idpair nLoopKF, const bool bRobust) {
  vector<kfptr> vpKFs = pMap->GetAllKeyFrames();
  vector<mpptr> vpMP = pMap->GetAllMapPoints();
  BundleAdjustmentClient(vpKFs, vpMP, ClientId, nIterations, pbStopFlag, nLoopKF, bRobust);
}

// Optimizer.cpp:40-212
void Optimizer::BundleAdjustmentClient(const vector<kfptr>& vpKFs, const vector<mpptr>& vpMP, size_t ClientId, int nIterations, bool* pbStopFlag,
                                       const idpair nLoopKF, const bool bRobust) {
  const idpair zeropair = make_pair(0, ClientId);
  vector<bool> vbNotIncludedMP;
  vbNotIncludedMP.resize(vpMP.size());
  FlatBA f;
  for (size_t i = 0; i < vpKFs.size(); i++) {
    kfptr pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    if (pKF->mId.first >= IDRANGE) {
      shim_fatal("Optimizer::BundleAdjustmentClient / MapFusionGBA", "keyframe id is not below IDRANGE");
    }
    f.addCam(Optimizer::GetID(pKF->mId, true), pKF, pKF->mId == zeropair);
  }
  const float thHuber2D = sqrt(5.99);
  for (size_t i = 0; i < vpMP.size(); i++) {
    mpptr pMP = vpMP[i];
    if (pMP->isBad()) continue;
    if (pMP->mId.first >= IDRANGE) {
      shim_fatal("Optimizer::BundleAdjustmentClient / MapFusionGBA", "map point id is not below IDRANGE");
    }
    const int id = Optimizer::GetID(pMP->mId, false);
    f.addPoint(id, pMP);
    const map<kfptr, size_t> observations = pMP->GetObservations();
    int nEdges = 0;
    for (map<kfptr, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
      kfptr pKF = mit->first;
      if (pKF->isBad()) continue;
      if (pKF->mId.first >= IDRANGE) {
        shim_fatal("Optimizer::BundleAdjustmentClient / MapFusionGBA", "keyframe id is not below IDRANGE");
      }
      nEdges++;
      f.addEdge(id, pKF, Optimizer::GetID(pKF->mId, true), pKF->mvKeysUn[mit->second]);
    }
    if (nEdges == 0) { f.removePoint(id); vbNotIncludedMP[i] = true; }
    else vbNotIncludedMP[i] = false;
  }
  f.flatten(true);
  run_ba(f, bRobust ? (double)thHuber2D : 0.0, nIterations, pbStopFlag, nullptr, nullptr);
  for (size_t i = 0; i < vpKFs.size(); i++) {
    kfptr pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    cv::Mat pose = f.camPose(Optimizer::GetID(pKF->mId, true));
    if (nLoopKF == zeropair) pKF->SetPose(pose, false);
    else {
      pKF->mTcwGBA.create(4, 4, CV_32F);
      pose.copyTo(pKF->mTcwGBA);
      pKF->mBAGlobalForKF = nLoopKF;
    }
  }
  for (size_t i = 0; i < vpMP.size(); i++) {
    if (vbNotIncludedMP[i]) continue;
    mpptr pMP = vpMP[i];
    if (pMP->isBad()) continue;
    cv::Mat pos = f.pointPos(Optimizer::GetID(pMP->mId, false));
    if (nLoopKF == zeropair) {
      pMP->SetWorldPos(pos, false);
      pMP->UpdateNormalAndDepth();
    } else {
      pMP->mPosGBA.create(3, 1, CV_32F);
      pos.copyTo(pMP->mPosGBA);
      pMP->mBAGlobalForKF = nLoopKF;
    }
  }
}

// Optimizer.cpp:215-347
int Optimizer::PoseOptimizationClient(Frame& Frame) {
  int nInitialCorrespondences = 0;
  float T0[16];
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T0[4 * r + c] = Frame.mTcw.at<float>(r, c);
  double cam_qt[7];
  ccmh::toSE3Quat(T0, cam_qt);                                                        // Converter::toSE3Quat(Frame.mTcw)
  const int N = Frame.N;
  std::vector<double> Xw, obs, info;
  vector<size_t> vnIndexEdgeMono;
  vnIndexEdgeMono.reserve(N);
  {
    unique_lock<mutex> lock(MapPoint::mGlobalMutex);
    for (int i = 0; i < N; i++) {
      mpptr pMP = Frame.mvpMapPoints[i];
      if (pMP) {
        nInitialCorrespondences++;
        Frame.mvbOutlier[i] = false;
        const cv::KeyPoint& kpUn = Frame.mvKeysUn[i];
        obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y);
        const float invSigma2 = Frame.mvInvLevelSigma2[kpUn.octave];
        info.push_back(invSigma2);
        cv::Mat P = pMP->GetWorldPos();
        Xw.push_back(P.at<float>(0)); Xw.push_back(P.at<float>(1)); Xw.push_back(P.at<float>(2));
        vnIndexEdgeMono.push_back(i);
      }
    }
  }
  if (nInitialCorrespondences < 3) return 0;
  const double K[4] = {Frame.fx, Frame.fy, Frame.cx, Frame.cy};
  std::vector<uint8_t> outlier(vnIndexEdgeMono.size(), 0);
  int nInliers = 0;
  // the four rounds of 10 iterations with their inlier / outlier reclassification (:299-338) run inside one kernel launch
  check(ccm_pose_optimize(thread_ctx(), cam_qt, (int)vnIndexEdgeMono.size(), Xw.data(), obs.data(), info.data(), K, outlier.data(), &nInliers), "ccm_pose_optimize");
  for (size_t i = 0; i < vnIndexEdgeMono.size(); i++) Frame.mvbOutlier[vnIndexEdgeMono[i]] = outlier[i] != 0;
  float T[16];
  ccmh::toCvMat(cam_qt, T);                                                           // Converter::toCvMat(SE3quat_recov)
  cv::Mat pose(4, 4, CV_32F);
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) pose.at<float>(r, c) = T[4 * r + c];
  Frame.SetPose(pose);
  return nInliers;
}

// Optimizer.cpp:349-644
void Optimizer::LocalBundleAdjustmentClient(kfptr pKF, bool* pbStopFlag, mapptr pMap, size_t ClientId, eSystemState SysState) {
  PhaseClock pc;
  // Local KeyFrames: breadth-first from the current keyframe (:351-366)
  list<kfptr> lLocalKeyFrames;
  lLocalKeyFrames.push_back(pKF);
  pKF->mBALocalForKF = pKF->mId;
  const vector<kfptr> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (int i = 0, iend = vNeighKFs.size(); i < iend; i++) {
    kfptr pKFi = vNeighKFs[i];
    pKFi->mBALocalForKF = pKF->mId;
    if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
  }
  // Local MapPoints seen in Local KeyFrames (:368-385)
  list<mpptr> lLocalMapPoints;
  for (list<kfptr>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
    vector<mpptr> vpMPs = (*lit)->GetMapPointMatches();
    for (vector<mpptr>::iterator vit = vpMPs.begin(), vend = vpMPs.end(); vit != vend; vit++) {
      mpptr pMP = *vit;
      if (pMP)
        if (!pMP->isBad())
          if (pMP->mBALocalForKF != pKF->mId) {
            lLocalMapPoints.push_back(pMP);
            pMP->mBALocalForKF = pKF->mId;
          }
    }
  }
  // Fixed Keyframes: see local points but are not local (:387-404)
  list<kfptr> lFixedCameras;
  for (list<mpptr>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
    map<kfptr, size_t> observations = (*lit)->GetObservations();
    for (map<kfptr, size_t>::iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
      kfptr pKFi = mit->first;
      if (pKFi->mBALocalForKF != pKF->mId && pKFi->mBAFixedForKF != pKF->mId) {
        pKFi->mBAFixedForKF = pKF->mId;
        if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
      }
    }
  }
  FlatBA f;
  for (list<kfptr>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
    kfptr pKFi = *lit;
    if (pKFi->mId.first >= IDRANGE) {
      shim_fatal("Optimizer::LocalBundleAdjustmentClient", "keyframe id is not below IDRANGE");
    }
    f.addCam(Optimizer::GetID(pKFi->mId, true), pKFi, pKFi->mId.first == 0 && pKFi->mId.second == ClientId);
  }
  for (list<kfptr>::iterator lit = lFixedCameras.begin(), lend = lFixedCameras.end(); lit != lend; lit++) {
    kfptr pKFi = *lit;
    if (pKFi->mId.first >= IDRANGE) {
      shim_fatal("Optimizer::LocalBundleAdjustmentClient", "keyframe id is not below IDRANGE");
    }
    f.addCam(Optimizer::GetID(pKFi->mId, true), pKFi, true);
  }
  vector<kfptr> vpEdgeKFMono;
  vector<mpptr> vpMapPointEdgeMono;
  const float thHuberMono = sqrt(5.991);
  for (list<mpptr>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
    mpptr pMP = *lit;
    if (pMP->mId.first >= IDRANGE) {
      shim_fatal("Optimizer::LocalBundleAdjustmentClient", "map point id is not below IDRANGE");
    }
    const int id = Optimizer::GetID(pMP->mId, false);
    f.addPoint(id, pMP);
    const map<kfptr, size_t> observations = pMP->GetObservations();
    for (map<kfptr, size_t>::const_iterator mit = observations.begin(), mend = observations.end(); mit != mend; mit++) {
      kfptr pKFi = mit->first;
      if (pKFi->mId.first >= IDRANGE) {
        shim_fatal("Optimizer::LocalBundleAdjustmentClient", "keyframe id is not below IDRANGE");
      }
      if (!pKFi->isBad()) {
        f.addEdge(id, pKFi, Optimizer::GetID(pKFi->mId, true), pKFi->mvKeysUn[mit->second]);
        vpEdgeKFMono.push_back(pKFi);
        vpMapPointEdgeMono.push_back(pMP);
      }
    }
  }
  if (pbStopFlag)
    if (*pbStopFlag) return;
  pc.lap(0);
  f.flatten();
  pc.lap(1);
  // optimizer.initializeOptimization(); optimizer.optimize(5);  (:536-537)
  std::vector<double> chi2;
  std::vector<uint8_t> dpos;
  struct Handle { ccm_ba* h = nullptr; ~Handle() { if (h) ccm_ba_destroy(h); } } session;   // both optimisations run on one device-side problem
  run_ba(f, (double)thHuberMono, 5, pbStopFlag, &chi2, &dpos, &session.h);
  bool bDoMore = true;
  if (pbStopFlag)
    if (*pbStopFlag) bDoMore = false;
  if (bDoMore) {
    // outliers to level 1, robust kernel off, optimize(10) (:545-566).  e->chi2() of a level-1 edge keeps the value of the first pass.
    for (size_t i = 0, iend = f.nEdges(); i < iend; i++) {
      mpptr pMP = vpMapPointEdgeMono[i];
      if (pMP->isBad()) continue;
      if (chi2[i] > 5.991 || !dpos[i]) f.e_level[i] = 1;
    }
    run_ba(f, 0.0, 10, pbStopFlag, &chi2, &dpos, &session.h);
  }
  pc.t = now_ms();
  vector<pair<kfptr, mpptr> > vToErase;
  vToErase.reserve(f.nEdges());
  for (size_t i = 0, iend = f.nEdges(); i < iend; i++) {
    mpptr pMP = vpMapPointEdgeMono[i];
    if (pMP->isBad()) continue;
    if (chi2[i] > 5.991 || !dpos[i]) vToErase.push_back(make_pair(vpEdgeKFMono[i], pMP));
  }
  if (SysState != eSystemState::SERVER)
    while (!pMap->LockMapUpdate()) { usleep(params::timings::miLockSleep); }
  if (!vToErase.empty()) {
    for (size_t i = 0; i < vToErase.size(); i++) {
      kfptr pKFi = vToErase[i].first;
      mpptr pMPi = vToErase[i].second;
      pKFi->EraseMapPointMatch(pMPi);
      pMPi->EraseObservation(pKFi);
    }
  }
  for (list<kfptr>::iterator lit = lLocalKeyFrames.begin(), lend = lLocalKeyFrames.end(); lit != lend; lit++) {
    kfptr pKFl = *lit;
    pKFl->SetPose(f.camPose(Optimizer::GetID(pKFl->mId, true)), false);
    pKFl->mbUpdatedByServer = false;
  }
  pc.lap(5);
  static const bool batched_off = std::getenv("CCM_SHIM_NO_BATCHED_NORMALS") != nullptr;
  if (kBatchedNormals && !batched_off && !f.cam_kf.empty() && !f.pt_id.empty()) {
    // (with the optional MapPoint::SetNormalAndDepth) positions as below, normals and distance ranges of all local points by one device call over the
    // problem's edges minus the erased observations
    for (list<mpptr>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++)
      if ((*lit)->isBad() && pMap->GetMpPtr((*lit)->mId)) {
        shim_fatal("Optimizer::LocalBundleAdjustmentClient", "a local map point is flagged bad but the map still holds it");
      }
    std::vector<char> erased(f.nEdges(), 0);
    for (size_t i = 0, iend = f.nEdges(); i < iend; i++) erased[i] = !vpMapPointEdgeMono[i]->isBad() && (chi2[i] > 5.991 || !dpos[i]);
    // (an observation erased above whose point turned bad with it belongs to a point the helper skips)
    for (size_t i = 0, iend = f.nEdges(); i < iend; i++) if (vpMapPointEdgeMono[i]->isBad()) erased[i] = 1;
    batched_point_writeback(f, &erased, false, false, false);
  } else
  for (list<mpptr>::iterator lit = lLocalMapPoints.begin(), lend = lLocalMapPoints.end(); lit != lend; lit++) {
    mpptr pMP = *lit;
    if (pMP->isBad()) {
      mpptr pMPcheck = pMap->GetMpPtr(pMP->mId);
      if (pMPcheck) {
        shim_fatal("Optimizer::LocalBundleAdjustmentClient", "a local map point is flagged bad but the map still holds it");
      }
    } else {
      pMP->SetWorldPos(f.pointPos(Optimizer::GetID(pMP->mId, false)), false);
      pMP->UpdateNormalAndDepth();
    }
  }
  pc.lap(6);
  if (SysState != eSystemState::SERVER) pMap->UnLockMapUpdate();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// server side
// ---------------------------------------------------------------------------------------------------------------------------------
// Optimizer.cpp:646-859
void Optimizer::MapFusionGBA(mapptr pMap, size_t ClientId, int nIterations, bool* pbStopFlag, idpair nLoopKF, const bool bRobust) {
  (void)ClientId;
  PhaseClock pc;
  {   // (everything the call owns lives in this scope, so that what its destructors cost is inside the last phase, not after it)
  vector<kfptr> vpKFs = pMap->GetAllKeyFrames();
  vector<mpptr> vpMP = pMap->GetAllMapPoints();
  const idpair zeropair = make_pair(0, pMap->mMapId);
  if (pMap->mvpKeyFrameOrigins.empty()) {
    shim_fatal("Optimizer::MapFusionGBA", "the map has no origin keyframe (mvpKeyFrameOrigins is empty)");
  }
  idpair FixedId = (*(pMap->mvpKeyFrameOrigins.begin()))->mId;
  std::vector<char> vbNotIncludedMP(vpMP.size(), 0);   // one byte per point: chunks of vpMP are walked by different threads
  static thread_local FlatBA f_keep;
  static thread_local std::vector<FlatBA> part_keep;
  FlatBA& f = f_keep;
  f.reset();
  // the flat arrays keep their capacity from call to call (per thread); the references into the caller's object graph do not outlive the call: a keyframe
  // erased from the map afterwards is destroyed when the reference would destroy it
  struct DropRefs {
    FlatBA& a; std::vector<FlatBA>& parts;
    ~DropRefs() { a.cam_kf.clear(); a.pt_mp.clear(); for (auto& g : parts) { g.cam_kf.clear(); g.pt_mp.clear(); } }
  } drop_refs{f_keep, part_keep};
  size_t maxKFid = 0;
  for (size_t i = 0; i < vpKFs.size(); i++) {
    kfptr pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    f.addCam(pKF->mUniqueId, pKF, pKF->mId == FixedId);
    if (pKF->mUniqueId > maxKFid) maxKFid = pKF->mUniqueId;
  }
  const float thHuber2D = sqrt(5.99);
  pc.lap(8);
  {
    // the map points in contiguous chunks of vpMP, one chunk per host thread (every point is independent; GetObservations() copies under the point's
    // own mutex exactly as in the reference); the chunks are appended in order, so vertices and edges keep the order of the sequential walk
    const int n_thr = shim_threads(vpMP.size());
    std::vector<FlatBA>& part = part_keep;
    if (part.size() < (size_t)n_thr) part.resize((size_t)n_thr);
    for (auto& g : part) g.reset();
    parallel_chunks(vpMP.size(), n_thr, [&](int t, size_t i0, size_t i1) {
      FlatBA& g = t == 0 ? f : part[(size_t)t];
      for (size_t i = i0; i < i1; i++) {
        const mpptr& pMP = vpMP[i];
        if (pMP->isBad()) continue;
        const map<kfptr, size_t> observations = pMP->GetObservations();
        if (observations.size() < 2) { vbNotIncludedMP[i] = true; continue; }
        int nEdges = 0;
        const size_t id = pMP->mUniqueId;
        for (map<kfptr, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
          const kfptr& pKF = mit->first;
          if (!pKF || pKF->isBad() || pKF->mUniqueId > maxKFid) continue;
          nEdges++;
        }
        if (nEdges < 2) { vbNotIncludedMP[i] = true; continue; }
        g.addPoint(id, pMP);
        g.addPointPos(pMP->GetWorldPos());   // (the vertex estimate, Optimizer.cpp:724: read here, while this thread has the point's lines, instead of in a second pass over all points)
        if (kBatchedNormals) {   // what the batched UpdateNormalAndDepth of the write-back needs beside the edges
          int nLive = 0;         // observations the reference's method would use (non-bad keyframes): all of them must be edges
          for (map<kfptr, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) if (mit->first && !mit->first->isBad()) nLive++;
          const kfptr pRef = pMP->GetReferenceKeyFrame();
          map<kfptr, size_t>::const_iterator rit = pRef ? observations.find(pRef) : observations.end();
          const bool regular = nLive == nEdges && pRef && !pRef->isBad() && pRef->mUniqueId <= maxKFid && rit != observations.end();
          g.addPointAux(regular ? pRef->mUniqueId : 0, regular ? pRef->mvKeysUn[rit->second].octave : 0, regular);
        }
        for (map<kfptr, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); mit++) {
          const kfptr& pKF = mit->first;
          if (!pKF || pKF->isBad() || pKF->mUniqueId > maxKFid) continue;
          g.addEdge(id, pKF, pKF->mUniqueId, pKF->mvKeysUn[mit->second]);
        }
      }
    });
    for (int t = 1; t < n_thr; t++) f.append(part[(size_t)t]);
  }
  pc.lap(0);
  f.flatten(true);
  pc.lap(1);
  run_ba(f, bRobust ? (double)thHuber2D : 0.0, nIterations, pbStopFlag, nullptr, nullptr);
  pc.t = now_ms();
  for (size_t i = 0; i < vpKFs.size(); i++) {
    kfptr pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    cv::Mat pose = f.camPose(pKF->mUniqueId);
    if (nLoopKF == zeropair) pKF->SetPose(pose, true);
    else {
      pKF->mTcwGBA.create(4, 4, CV_32F);
      pose.copyTo(pKF->mTcwGBA);
      pKF->mBAGlobalForKF = nLoopKF;
    }
  }
  pc.lap(5);
  // every keyframe has its new pose: the per-point write-back (SetWorldPos + UpdateNormalAndDepth, 150 000 mutex-taking calls after a merge of four
  // agents) is independent from point to point
  static const bool batched_off = std::getenv("CCM_SHIM_NO_BATCHED_NORMALS") != nullptr;
  bool refs_dropped = false;
  if (kBatchedNormals && !batched_off && nLoopKF == zeropair && f.aux_ok && f.pt_regular.size() == f.pt_id.size() && !f.cam_kf.empty()) batched_point_writeback(f, nullptr, true, true, true);
  else {
  refs_dropped = true;
  const int n_wb = shim_threads(vpMP.size(), 32);
  // (measured on the reference's REAL classes, 150 000 points: 63 - 78 ms for this loop on 32 threads against 18 on the look-alike — the real SetWorldPos takes the
  // process-wide MapPoint::mGlobalMutex (MapPoint.cpp:343) and the real UpdateNormalAndDepth two mutexes of every observing keyframe (isBad, GetCameraCenter).  Tried
  // and dropped: all positions on one thread with the normals following behind it (67 ms: the call itself is ~450 ns), positions on four threads then normals on all (66 - 82 ms).)
  parallel_chunks(vpMP.size(), n_wb, [&](int, size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
      mpptr& pMP = vpMP[i];
      if (!vbNotIncludedMP[i] && !pMP->isBad()) {
        cv::Mat pos = f.pointPos(pMP->mUniqueId);
        if (nLoopKF == zeropair) {
          pMP->SetWorldPos(pos, true);
          pMP->UpdateNormalAndDepth();
        } else {
          pMP->mPosGBA.create(3, 1, CV_32F);
          pos.copyTo(pMP->mPosGBA);
          pMP->mBAGlobalForKF = nLoopKF;
        }
      }
      // this call's copy of the pointer (Map::GetAllMapPoints() hands out 150 000 of them by value) is dropped by the thread that has just worked on the point:
      // releasing them in one go afterwards cost 13 - 30 ms on the 4-agent map (one lock-prefixed decrement per point on a line some other core holds)
      pMP.reset();
    }
  });
  }
  pc.lap(6);
  f.cam_kf.clear(); f.pt_mp.clear();   // no keyframe / map point is kept alive between calls; the flat arrays stay allocated
  pc.lap(9);
  // the copies Map::GetAllMapPoints() / GetAllKeyFrames() handed out by value: released on several threads and INSIDE the phase clock (round 5: their
  // destructors used to run after the last lap, 29 ms that no phase showed)
  if (!refs_dropped) release_refs(vpMP);
  release_refs(vpKFs);
  }
  pc.lap(10);
}

// Optimizer.cpp:861-1056
int Optimizer::OptimizeSim3(kfptr pKF1, kfptr pKF2, std::vector<mpptr>& vpMatches1, g2o::Sim3& g2oS12, const float th2, bool bFixScale) {
  const cv::Mat& K1 = pKF1->mK;
  const cv::Mat& K2 = pKF2->mK;
  const cv::Mat R1w = pKF1->GetRotation();
  const cv::Mat t1w = pKF1->GetTranslation();
  const cv::Mat R2w = pKF2->GetRotation();
  const cv::Mat t2w = pKF2->GetTranslation();
  const double k1[4] = {K1.at<float>(0, 0), K1.at<float>(1, 1), K1.at<float>(0, 2), K1.at<float>(1, 2)};
  const double k2[4] = {K2.at<float>(0, 0), K2.at<float>(1, 1), K2.at<float>(0, 2), K2.at<float>(1, 2)};
  const int N = vpMatches1.size();
  const vector<mpptr> vpMapPoints1 = pKF1->GetMapPointMatches();
  std::vector<double> P1c, P2c, obs1, obs2, info1, info2;
  vector<size_t> vnIndexEdge;
  for (int i = 0; i < N; i++) {
    if (!vpMatches1[i]) continue;
    mpptr pMP1 = vpMapPoints1[i];
    mpptr pMP2 = vpMatches1[i];
    const int i2 = pMP2->GetIndexInKeyFrame(pKF2);
    if (pMP1 && pMP2) {
      if (!pMP1->isBad() && !pMP2->isBad() && i2 >= 0) {
        cv::Mat P3D1c = R1w * pMP1->GetWorldPos() + t1w;   // the cv::Mat arithmetic stays with the caller's data types (f32)
        cv::Mat P3D2c = R2w * pMP2->GetWorldPos() + t2w;
        for (int c = 0; c < 3; c++) { P1c.push_back(P3D1c.at<float>(c)); P2c.push_back(P3D2c.at<float>(c)); }
      } else continue;
    } else continue;
    const cv::KeyPoint& kpUn1 = pKF1->mvKeysUn[i];
    obs1.push_back(kpUn1.pt.x); obs1.push_back(kpUn1.pt.y);
    info1.push_back(pKF1->mvInvLevelSigma2[kpUn1.octave]);
    const cv::KeyPoint& kpUn2 = pKF2->mvKeysUn[i2];
    obs2.push_back(kpUn2.pt.x); obs2.push_back(kpUn2.pt.y);
    info2.push_back(pKF2->mvInvLevelSigma2[kpUn2.octave]);
    vnIndexEdge.push_back(i);
  }
  double s8[8] = {g2oS12.rotation().x(), g2oS12.rotation().y(), g2oS12.rotation().z(), g2oS12.rotation().w(),
                  g2oS12.translation()[0], g2oS12.translation()[1], g2oS12.translation()[2], g2oS12.scale()};
  std::vector<uint8_t> keep(vnIndexEdge.size(), 1);
  int nIn = 0;
  check(ccm_sim3_optimize(thread_ctx(), s8, (int)vnIndexEdge.size(), P1c.data(), P2c.data(), obs1.data(), obs2.data(), info1.data(), info2.data(), k1, k2,
                          (double)th2, bFixScale ? 1 : 0, keep.data(), &nIn),
        "ccm_sim3_optimize");
  for (size_t i = 0; i < vnIndexEdge.size(); i++) if (!keep[i]) vpMatches1[vnIndexEdge[i]] = static_cast<mpptr>(NULL);
  if (nIn == 0) return 0;   // fewer than 10 survivors after the first pass: g2oS12 stays as it was (:1015-1016)
  g2oS12 = g2o::Sim3(Eigen::Quaterniond(s8[3], s8[0], s8[1], s8[2]), Eigen::Vector3d(s8[4], s8[5], s8[6]), s8[7]);
  return nIn;
}

namespace {
typedef std::vector<g2o::Sim3, Eigen::aligned_allocator<g2o::Sim3> > Sim3Vec;
void sim3_to8(const g2o::Sim3& S, double* p) {
  p[0] = S.rotation().x(); p[1] = S.rotation().y(); p[2] = S.rotation().z(); p[3] = S.rotation().w();
  p[4] = S.translation()[0]; p[5] = S.translation()[1]; p[6] = S.translation()[2]; p[7] = S.scale();
}
// edge list of an essential graph + the device call; vertices are indexed by mUniqueId like the reference's vScw / vpVertices
struct PoseGraph {
  std::vector<int32_t> e_i, e_j;
  std::vector<double> meas;
  void addEdge(size_t i, size_t j, const g2o::Sim3& Sji) { e_i.push_back((int32_t)i); e_j.push_back((int32_t)j); meas.resize(meas.size() + 8); sim3_to8(Sji, &meas[meas.size() - 8]); }
  // optimizer.initializeOptimization(); optimizer.optimize(20) with setUserLambdaInit(1e-16): vertices = the keyframes that were added (`present`)
  void optimize(const Sim3Vec& vScw, const std::vector<char>& present, size_t fixed_uid, bool bFixScale, Sim3Vec& out) {
    const size_t n = vScw.size();
    std::vector<int32_t> slot(n, -1);
    std::vector<double> sim3;
    std::vector<uint8_t> fixed;
    int nv = 0;
    for (size_t u = 0; u < n; u++) if (present[u]) { slot[u] = nv++; sim3.resize(sim3.size() + 8); sim3_to8(vScw[u], &sim3[sim3.size() - 8]); fixed.push_back(u == fixed_uid ? 1 : 0); }
    std::vector<int32_t> ei(e_i.size()), ej(e_j.size());
    for (size_t k = 0; k < e_i.size(); k++) { ei[k] = slot[e_i[k]]; ej[k] = slot[e_j[k]]; }
    check(ccm_pose_graph_optimize(thread_ctx(), nv, sim3.data(), fixed.data(), bFixScale ? 1 : 0, (int)ei.size(), ei.data(), ej.data(), meas.data(), 20, 1e-16, nullptr,
                                  nullptr),
          "ccm_pose_graph_optimize");
    out = vScw;
    for (size_t u = 0; u < n; u++) if (present[u]) { const double* p = &sim3[8 * (size_t)slot[u]]; out[u] = g2o::Sim3(Eigen::Quaterniond(p[3], p[0], p[1], p[2]), Eigen::Vector3d(p[4], p[5], p[6]), p[7]); }
  }
};
cv::Mat sim3_pose(const g2o::Sim3& CorrectedSiw) {   // [R t/s; 0 1] (:1272-1281)
  double s8[8];
  sim3_to8(CorrectedSiw, s8);
  float T[16];
  ccmh::sim3ToCvSE3(s8, T);
  cv::Mat m(4, 4, CV_32F);
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) m.at<float>(r, c) = T[4 * r + c];
  return m.clone();
}
}  // namespace

// Optimizer.cpp:1058-1331
void Optimizer::OptimizeEssentialGraphLoopClosure(mapptr pMap, kfptr pLoopKF, kfptr pCurKF, const KeyFrameAndPose& NonCorrectedSim3,
                                                  const KeyFrameAndPose& CorrectedSim3, const map<kfptr, set<kfptr> >& LoopConnections, const bool& bFixScale) {
  const vector<kfptr> vpKFs = pMap->GetAllKeyFrames();
  const vector<mpptr> vpMPs = pMap->GetAllMapPoints();
  const unsigned int nMaxKFid = pMap->GetMaxKFidUnique();
  Sim3Vec vScw(nMaxKFid + 1), vCorrectedSwc(nMaxKFid + 1);
  std::vector<char> present(nMaxKFid + 1, 0);
  const int minFeat = params::opt::miEssGraphMinFeats;
  for (size_t i = 0, iend = vpKFs.size(); i < iend; i++) {
    kfptr pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    const size_t nIDi = pKF->mUniqueId;
    KeyFrameAndPose::const_iterator it = CorrectedSim3.find(pKF);
    if (it != CorrectedSim3.end()) vScw[nIDi] = it->second;
    else {
      Eigen::Matrix<double, 3, 3> Rcw = Converter::toMatrix3d(pKF->GetRotation());
      Eigen::Matrix<double, 3, 1> tcw = Converter::toVector3d(pKF->GetTranslation());
      vScw[nIDi] = g2o::Sim3(Rcw, tcw, 1.0);
    }
    present[nIDi] = 1;
  }
  PoseGraph pg;
  set<pair<long unsigned int, long unsigned int> > sInsertedEdges;
  for (map<kfptr, set<kfptr> >::const_iterator mit = LoopConnections.begin(), mend = LoopConnections.end(); mit != mend; mit++) {
    kfptr pKF = mit->first;
    if (pKF->isBad()) continue;
    const size_t nIDi = pKF->mUniqueId;
    const set<kfptr>& spConnections = mit->second;
    const g2o::Sim3 Swi = vScw[nIDi].inverse();
    for (set<kfptr>::const_iterator sit = spConnections.begin(), send = spConnections.end(); sit != send; sit++) {
      if ((*sit)->isBad()) continue;
      const size_t nIDj = (*sit)->mUniqueId;
      if ((nIDi != pCurKF->mUniqueId || nIDj != pLoopKF->mUniqueId) && pKF->GetWeight(*sit) < minFeat) continue;
      pg.addEdge(nIDi, nIDj, vScw[nIDj] * Swi);
      sInsertedEdges.insert(make_pair(min(nIDi, nIDj), max(nIDi, nIDj)));
    }
  }
  for (size_t i = 0, iend = vpKFs.size(); i < iend; i++) {
    kfptr pKF = vpKFs[i];
    const size_t nIDi = pKF->mUniqueId;
    g2o::Sim3 Swi;
    KeyFrameAndPose::const_iterator iti = NonCorrectedSim3.find(pKF);
    if (iti != NonCorrectedSim3.end()) Swi = (iti->second).inverse();
    else Swi = vScw[nIDi].inverse();
    kfptr pParentKF = pKF->GetParent();
    if (pParentKF) {   // spanning tree edge
      const size_t nIDj = pParentKF->mUniqueId;
      KeyFrameAndPose::const_iterator itj = NonCorrectedSim3.find(pParentKF);
      const g2o::Sim3 Sjw = itj != NonCorrectedSim3.end() ? itj->second : vScw[nIDj];
      pg.addEdge(nIDi, nIDj, Sjw * Swi);
    }
    const set<kfptr> sLoopEdges = pKF->GetLoopEdges();
    for (set<kfptr>::const_iterator sit = sLoopEdges.begin(), send = sLoopEdges.end(); sit != send; sit++) {
      kfptr pLKF = *sit;
      const size_t nIDj = pLKF->mUniqueId;
      if (nIDj < nIDi) {
        KeyFrameAndPose::const_iterator itl = NonCorrectedSim3.find(pLKF);
        const g2o::Sim3 Slw = itl != NonCorrectedSim3.end() ? itl->second : vScw[nIDj];
        pg.addEdge(nIDi, nIDj, Slw * Swi);
      }
    }
    const vector<kfptr> vpConnectedKFs = pKF->GetCovisiblesByWeight(minFeat);
    for (vector<kfptr>::const_iterator vit = vpConnectedKFs.begin(); vit != vpConnectedKFs.end(); vit++) {
      kfptr pKFn = *vit;
      if ((*vit)->isBad()) continue;
      if (pKFn && pKFn != pParentKF && !pKF->hasChild(pKFn) && !sLoopEdges.count(pKFn)) {
        const size_t nIDj = pKFn->mUniqueId;
        if (!pKFn->isBad() && nIDj < nIDi) {
          if (sInsertedEdges.count(make_pair(min(nIDi, nIDj), max(nIDi, nIDj)))) continue;
          KeyFrameAndPose::const_iterator itn = NonCorrectedSim3.find(pKFn);
          const g2o::Sim3 Snw = itn != NonCorrectedSim3.end() ? itn->second : vScw[nIDj];
          pg.addEdge(nIDi, nIDj, Snw * Swi);
        }
      }
    }
  }
  Sim3Vec vCorrectedSiw;
  pg.optimize(vScw, present, pLoopKF->mUniqueId, bFixScale, vCorrectedSiw);
  for (size_t i = 0; i < vpKFs.size(); i++) {
    kfptr pKFi = vpKFs[i];
    const size_t nIDi = pKFi->mUniqueId;
    const g2o::Sim3 CorrectedSiw = vCorrectedSiw[nIDi];
    vCorrectedSwc[nIDi] = CorrectedSiw.inverse();
    pKFi->SetPose(sim3_pose(CorrectedSiw), true);
  }
  for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
    mpptr pMP = vpMPs[i];
    if (pMP->isBad()) continue;
    size_t nIDr;
    if (pMP->mCorrectedByKF_LC == pCurKF->mId) nIDr = pMP->mCorrectedReference_LC;
    else nIDr = pMP->GetReferenceKeyFrame()->mUniqueId;
    const g2o::Sim3 Srw = vScw[nIDr];
    const g2o::Sim3 correctedSwr = vCorrectedSwc[nIDr];
    Eigen::Matrix<double, 3, 1> eigP3Dw = Converter::toVector3d(pMP->GetWorldPos());
    Eigen::Matrix<double, 3, 1> eigCorrectedP3Dw = correctedSwr.map(Srw.map(eigP3Dw));
    pMP->SetWorldPos(Converter::toCvMat(eigCorrectedP3Dw), true);
    pMP->UpdateNormalAndDepth();
  }
}

// Optimizer.cpp:1333-1566
void Optimizer::OptimizeEssentialGraphMapFusion(mapptr pMap, kfptr pLoopKF, kfptr pCurKF, const map<kfptr, set<kfptr> >& LoopConnections, const bool& bFixScale) {
  const vector<kfptr> vpKFs = pMap->GetAllKeyFrames();
  const vector<mpptr> vpMPs = pMap->GetAllMapPoints();
  const unsigned int nMaxKFid = pMap->GetMaxKFidUnique();
  Sim3Vec vScw(nMaxKFid + 1), vCorrectedSwc(nMaxKFid + 1);
  std::vector<char> present(nMaxKFid + 1, 0);
  const int minFeat = params::opt::miEssGraphMinFeats;
  for (size_t i = 0, iend = vpKFs.size(); i < iend; i++) {
    kfptr pKF = vpKFs[i];
    if (pKF->isBad()) continue;
    const size_t nIDi = pKF->mUniqueId;
    Eigen::Matrix<double, 3, 3> Rcw = Converter::toMatrix3d(pKF->GetRotation());
    Eigen::Matrix<double, 3, 1> tcw = Converter::toVector3d(pKF->GetTranslation());
    vScw[nIDi] = g2o::Sim3(Rcw, tcw, 1.0);
    present[nIDi] = 1;
  }
  PoseGraph pg;
  set<pair<long unsigned int, long unsigned int> > sInsertedEdges;
  for (map<kfptr, set<kfptr> >::const_iterator mit = LoopConnections.begin(), mend = LoopConnections.end(); mit != mend; mit++) {
    kfptr pKF = mit->first;
    if (pKF->isBad()) continue;
    const size_t nIDi = pKF->mUniqueId;
    const set<kfptr>& spConnections = mit->second;
    const g2o::Sim3 Swi = vScw[nIDi].inverse();
    for (set<kfptr>::const_iterator sit = spConnections.begin(), send = spConnections.end(); sit != send; sit++) {
      if ((*sit)->isBad()) continue;
      const size_t nIDj = (*sit)->mUniqueId;
      if ((nIDi != pCurKF->mUniqueId || nIDj != pLoopKF->mUniqueId) && pKF->GetWeight(*sit) < minFeat) continue;
      pg.addEdge(nIDi, nIDj, vScw[nIDj] * Swi);
      sInsertedEdges.insert(make_pair(min(nIDi, nIDj), max(nIDi, nIDj)));
    }
  }
  for (size_t i = 0, iend = vpKFs.size(); i < iend; i++) {
    kfptr pKF = vpKFs[i];
    const size_t nIDi = pKF->mUniqueId;
    const g2o::Sim3 Swi = vScw[nIDi].inverse();
    kfptr pParentKF = pKF->GetParent();
    if (pParentKF) pg.addEdge(nIDi, pParentKF->mUniqueId, vScw[pParentKF->mUniqueId] * Swi);
    const set<kfptr> sLoopEdges = pKF->GetLoopEdges();
    for (set<kfptr>::const_iterator sit = sLoopEdges.begin(), send = sLoopEdges.end(); sit != send; sit++) {
      const size_t nIDj = (*sit)->mUniqueId;
      if (nIDj < nIDi) pg.addEdge(nIDi, nIDj, vScw[nIDj] * Swi);
    }
    const vector<kfptr> vpConnectedKFs = pKF->GetCovisiblesByWeight(minFeat);
    for (vector<kfptr>::const_iterator vit = vpConnectedKFs.begin(); vit != vpConnectedKFs.end(); vit++) {
      kfptr pKFn = *vit;
      if ((*vit)->isBad()) continue;
      if (pKFn && pKFn != pParentKF && !pKF->hasChild(pKFn) && !sLoopEdges.count(pKFn)) {
        const size_t nIDj = pKFn->mUniqueId;
        if (!pKFn->isBad() && nIDj < nIDi) {
          if (sInsertedEdges.count(make_pair(min(nIDi, nIDj), max(nIDi, nIDj)))) continue;
          pg.addEdge(nIDi, nIDj, vScw[nIDj] * Swi);
        }
      }
    }
  }
  Sim3Vec vCorrectedSiw;
  pg.optimize(vScw, present, pLoopKF->mUniqueId, bFixScale, vCorrectedSiw);
  for (size_t i = 0; i < vpKFs.size(); i++) {
    kfptr pKFi = vpKFs[i];
    const size_t nIDi = pKFi->mUniqueId;
    const g2o::Sim3 CorrectedSiw = vCorrectedSiw[nIDi];
    vCorrectedSwc[nIDi] = CorrectedSiw.inverse();
    pKFi->SetPose(sim3_pose(CorrectedSiw), true);
  }
  for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
    mpptr pMP = vpMPs[i];
    if (pMP->isBad()) continue;
    int nIDr;
    if (pMP->mCorrectedByKF_MM == pCurKF->mId) nIDr = pMP->mCorrectedReference_MM;
    else nIDr = pMP->GetReferenceKeyFrame()->mUniqueId;
    const g2o::Sim3 Srw = vScw[nIDr];
    const g2o::Sim3 correctedSwr = vCorrectedSwc[nIDr];
    Eigen::Matrix<double, 3, 1> eigP3Dw = Converter::toVector3d(pMP->GetWorldPos());
    Eigen::Matrix<double, 3, 1> eigCorrectedP3Dw = correctedSwr.map(Srw.map(eigP3Dw));
    pMP->SetWorldPos(Converter::toCvMat(eigCorrectedP3Dw), true);
    pMP->UpdateNormalAndDepth();
  }
}

}  // namespace cslam

// phases (ms) of the last LocalBundleAdjustmentClient / MapFusionGBA call of the calling thread, see g_phase above
extern "C" void ccm_shim_last_phases(double* out10) { for (int i = 0; i < 10; i++) out10[i] = cslam::g_phase[i]; }
// all phases: fills min(cap, 12) entries, returns 12
extern "C" int ccm_shim_phases(double* out, int cap) { for (int i = 0; i < cap && i < cslam::kPhases; i++) out[i] = cslam::g_phase[i]; return cslam::kPhases; }
