"""Bundle-adjustment / pose-optimisation entry points of the C ABI (Optimizer path).

Python harness over ccm_ba_* / ccm_pose_optimize (include/ccm_hip.h), mirroring the static methods
of cslam::Optimizer on flat problems:
  local_bundle_adjustment  <- Optimizer::LocalBundleAdjustmentClient (cslam/src/Optimizer.cpp:349-644)
  global_bundle_adjustment <- Optimizer::MapFusionGBA                (cslam/src/Optimizer.cpp:646-859)
  pose_optimization        <- Optimizer::PoseOptimizationClient      (cslam/src/Optimizer.cpp:215-347)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import BAOptions, BAProblem, BAStats, Context, check, hooks, lib


TRIAL_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p).value if a is not None else None


class BAHandle:
    """Staged BA (ccm_ba_create / run / download).  Keeps the flattened arrays alive."""

    def __init__(self, ctx: Context, prob: dict, rank: int = 0, nranks: int = 1, resident: "ResidentProblem | None" = None):
        """resident: the flat problem already uploaded to HBM (ResidentProblem): ccm_ba_create then reads device memory only."""
        self.ctx = ctx
        self.prob = prob
        if resident is not None:
            self._keep = resident.keep
            self.cprob = resident.host_cprob
            self._h = C.c_void_p()
            check(lib().ccm_ba_create(ctx.handle, C.byref(resident.dev_cprob), int(rank), int(nranks), C.byref(self._h)), ctx.handle)
            ctx.adopt(self)
            return
        self._keep = {
            "cam_qt": np.ascontiguousarray(prob["cam_qt"], np.float64).copy(),
            "cam_fixed": np.ascontiguousarray(prob["cam_fixed"], np.uint8),
            "cam_K": np.ascontiguousarray(prob["cam_K"], np.float64),
            "pt_xyz": np.ascontiguousarray(prob["pt_xyz"], np.float64).copy(),
            "e_cam": np.ascontiguousarray(prob["e_cam"], np.int32),
            "e_pt": np.ascontiguousarray(prob["e_pt"], np.int32),
            "e_obs": np.ascontiguousarray(prob["e_obs"], np.float64),
            "e_info": np.ascontiguousarray(prob["e_info"], np.float64),
        }
        lvl = prob.get("e_level")
        self._keep["e_level"] = np.ascontiguousarray(lvl, np.uint8) if lvl is not None else None
        k = self._keep
        self.cprob = BAProblem(int(prob["n_cam"]), int(prob["n_pt"]), int(prob["n_edge"]), _vp(k["cam_qt"]),
                               _vp(k["cam_fixed"]), _vp(k["cam_K"]), _vp(k["pt_xyz"]), _vp(k["e_cam"]), _vp(k["e_pt"]),
                               _vp(k["e_obs"]), _vp(k["e_info"]), _vp(k["e_level"]), float(prob["huber_delta"]))
        self._h = C.c_void_p()
        check(lib().ccm_ba_create(ctx.handle, C.byref(self.cprob), int(rank), int(nranks), C.byref(self._h)), ctx.handle)
        ctx.adopt(self)

    def reset(self):
        check(lib().ccm_ba_reset_state(self._h, C.c_void_p(_vp(self._keep["cam_qt"])), C.c_void_p(_vp(self._keep["pt_xyz"]))),
              self.ctx.handle)

    def push_state(self):
        """SparseOptimizer::push(): keep a device-side copy of the current estimate"""
        check(lib().ccm_ba_push_state(self._h), self.ctx.handle)

    def pop_state(self):
        """SparseOptimizer::pop(): the saved estimate becomes current again (device-to-device, stream-ordered)"""
        check(lib().ccm_ba_pop_state(self._h), self.ctx.handle)

    def run(self, max_iters: int, pcg_max_iters: int = 0, pcg_rel_tol: float = 0.0, lambda_init: float = 0.0,
            verbose: int = 0, stop_flag=None) -> BAStats:
        """optimizer.optimize(max_iters).  stop_flag: a 1-element numpy uint8 array playing the reference's bool* pbStopFlag."""
        opt = BAOptions(int(max_iters), int(pcg_max_iters), float(pcg_rel_tol), float(lambda_init), int(verbose))
        st = BAStats()
        fl = stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None
        check(lib().ccm_ba_run(self._h, C.byref(opt), fl, C.byref(st)), self.ctx.handle)
        return st

    def set_edge_levels(self, e_level, huber_delta: float):
        """second stage of the local BA on the same handle (ccm_ba_set_edge_levels): edges with level != 0 leave, new Huber delta"""
        lvl = np.ascontiguousarray(e_level, np.uint8)
        check(lib().ccm_ba_set_edge_levels(self._h, C.c_void_p(_vp(lvl)), C.c_double(huber_delta)), self.ctx.handle)

    def history(self):
        """(chi2 after each LM iteration, lambda after it, trials it took) of the last run"""
        n = C.c_int(0)
        chi, lam, tr = np.zeros(256), np.zeros(256), np.zeros(256, np.int32)
        check(lib().ccm_ba_history(self._h, 256, C.c_void_p(_vp(chi)), C.c_void_p(_vp(lam)), C.c_void_p(_vp(tr)), C.byref(n)), self.ctx.handle)
        k = min(n.value, 256)
        return chi[:k].copy(), lam[:k].copy(), tr[:k].copy()

    def set_trial_callback(self, fn):
        """fn(iteration, trial_in_iteration, chi2_trial, accepted) after every LM trial, on the optimising thread; None removes it"""
        if fn is None:
            self._cb = None
            check(lib().ccm_ba_set_trial_callback(self._h, None, None), self.ctx.handle)
            return
        self._cb = TRIAL_CB(lambda user, it, tr, chi, acc: fn(it, tr, chi, acc))
        check(lib().ccm_ba_set_trial_callback(self._h, self._cb, None), self.ctx.handle)

    def download(self, chi2_in=None):
        cam = self._keep["cam_qt"].copy()
        pts = self._keep["pt_xyz"].copy()
        ne = int(self.prob["n_edge"])
        chi2 = np.zeros(ne, np.float64) if chi2_in is None else np.ascontiguousarray(chi2_in, np.float64).copy()
        dpos = np.zeros(ne, np.uint8)
        check(lib().ccm_ba_download(self._h, C.c_void_p(_vp(cam)), C.c_void_p(_vp(pts)), C.c_void_p(_vp(chi2))),
              self.ctx.handle)
        check(lib().ccm_ba_depth_positive(C.byref(self.cprob), C.c_void_p(_vp(cam)), C.c_void_p(_vp(pts)),
                                          C.c_void_p(_vp(dpos))), self.ctx.handle)
        return cam, pts, chi2, dpos

    def counts(self):
        v = [C.c_int64() for _ in range(5)]
        check(lib().ccm_ba_counts(self._h, *[C.byref(x) for x in v]), self.ctx.handle)
        return dict(zip(("edges", "points", "free_cams", "blocks", "pairs"), (x.value for x in v)))

    def partial_reduced(self, lam: float) -> np.ndarray:
        """Test hook: this rank's partial [S | b_schur] at the current state (ccm_ba_debug_partial_reduced)."""
        n = C.c_size_t()
        check(hooks().ccm_ba_debug_partial_reduced(self._h, C.c_double(lam), None, C.c_size_t(0), C.byref(n)), self.ctx.handle)
        out = np.zeros(n.value, np.float64)
        check(hooks().ccm_ba_debug_partial_reduced(self._h, C.c_double(lam), C.c_void_p(_vp(out)), C.c_size_t(out.size), C.byref(n)),
              self.ctx.handle)
        return out

    def coarse_level(self, lam: float):
        """Test hook (ccm_ba_debug_coarse): (na camera intervals, Ac, Ainv over the na + 1 coarse nodes, P) of the two-level preconditioner, or (0, None, None, None)."""
        na = C.c_int(0)
        check(hooks().ccm_ba_debug_coarse(self._h, C.c_double(lam), C.byref(na), None, None, None, C.c_size_t(0)), self.ctx.handle)
        if na.value == 0:
            return 0, None, None, None
        nc = 6 * (na.value + 1)
        Ac = np.zeros((nc, nc)); Ai = np.zeros((nc, nc)); P = np.zeros((self.counts()["free_cams"], 6, 6))
        check(hooks().ccm_ba_debug_coarse(self._h, C.c_double(lam), C.byref(na), C.c_void_p(_vp(Ac)), C.c_void_p(_vp(Ai)), C.c_void_p(_vp(P)),
                                        C.c_size_t(Ac.size)), self.ctx.handle)
        return na.value, Ac, Ai, P

    def close(self):
        if self._h:
            lib().ccm_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ResidentProblem:
    """The flat arrays of a BA problem uploaded once to HBM; `dev_cprob` is a ccm_ba_problem whose pointers address device memory
    (ccm_ba_create accepts either, include/ccm_hip.h)."""

    def __init__(self, ctx: Context, prob: dict):
        self.ctx = ctx
        k = self.keep = {
            "cam_qt": np.ascontiguousarray(prob["cam_qt"], np.float64).copy(), "cam_fixed": np.ascontiguousarray(prob["cam_fixed"], np.uint8),
            "cam_K": np.ascontiguousarray(prob["cam_K"], np.float64), "pt_xyz": np.ascontiguousarray(prob["pt_xyz"], np.float64).copy(),
            "e_cam": np.ascontiguousarray(prob["e_cam"], np.int32), "e_pt": np.ascontiguousarray(prob["e_pt"], np.int32),
            "e_obs": np.ascontiguousarray(prob["e_obs"], np.float64), "e_info": np.ascontiguousarray(prob["e_info"], np.float64),
            "e_level": np.ascontiguousarray(prob["e_level"], np.uint8) if prob.get("e_level") is not None else None}
        order = ("cam_qt", "cam_fixed", "cam_K", "pt_xyz", "e_cam", "e_pt", "e_obs", "e_info", "e_level")
        self.dptr = {n: (ctx.upload(k[n]) if k[n] is not None else None) for n in order}
        dims = (int(prob["n_cam"]), int(prob["n_pt"]), int(prob["n_edge"]))
        self.host_cprob = BAProblem(*dims, *[_vp(k[n]) for n in order], float(prob["huber_delta"]))
        self.dev_cprob = BAProblem(*dims, *[self.dptr[n] for n in order], float(prob["huber_delta"]))
        self.bytes = sum(k[n].nbytes for n in order if k[n] is not None)

    def close(self):
        for p in self.dptr.values():
            if p:
                self.ctx.free(p)
        self.dptr = {}


def bundle_adjustment(ctx: Context, prob: dict, max_iters: int, chi2_in=None, **kw):
    """optimizer.initializeOptimization(0); optimizer.optimize(max_iters) on the flat problem.
    Returns (cam_qt, pt_xyz, chi2_per_edge, depth_pos, stats)."""
    h = BAHandle(ctx, prob)
    try:
        st = h.run(max_iters, **kw)
        cam, pts, chi2, dpos = h.download(chi2_in)
    finally:
        h.close()
    return cam, pts, chi2, dpos, st


def global_bundle_adjustment(ctx: Context, prob: dict, n_iterations: int = 20, **kw):
    """Optimizer::MapFusionGBA numerics (Optimizer.cpp:786-797): one optimize(nIterations) pass,
    Huber sqrt(5.99) on every edge."""
    return bundle_adjustment(ctx, prob, n_iterations, **kw)


def local_bundle_adjustment(ctx: Context, prob: dict, **kw):
    """Optimizer::LocalBundleAdjustmentClient numerics (Optimizer.cpp:536-602): optimize(5) with Huber
    sqrt(5.991); edges with chi2 > 5.991 or non-positive depth go to level 1 and the kernel is
    dropped; optimize(10); finally every edge with chi2 > 5.991 or non-positive depth is reported
    for erasure (level-1 edges keep the chi2 of the first pass, as g2o leaves their _error alone).
    Returns (cam_qt, pt_xyz, to_erase mask, stats1, stats2)."""
    p1 = dict(prob)
    p1["huber_delta"] = float(np.float32(np.sqrt(np.float32(5.991))))
    level = np.ascontiguousarray(prob.get("e_level") if prob.get("e_level") is not None else np.zeros(prob["n_edge"], np.uint8), np.uint8).copy()
    active = level == 0
    if kw.pop("one_handle", True):
        # both stages on ONE handle (what shim/Optimizer_hip.cpp does): the outliers of the first pass leave through ccm_ba_set_edge_levels
        h = BAHandle(ctx, p1)
        try:
            st1 = h.run(5, **kw)
            _, _, chi2, dpos = h.download()
            level[active & ((chi2 > 5.991) | (dpos == 0))] = 1
            h.set_edge_levels(level, 0.0)
            st2 = h.run(10, **kw)
            cam2, pts2, chi2b, dpos2 = h.download(chi2)
        finally:
            h.close()
        erase = active & ((chi2b > 5.991) | (dpos2 == 0))
        return cam2, pts2, erase, st1, st2
    cam, pts, chi2, dpos, st1 = bundle_adjustment(ctx, p1, 5, **kw)
    out = active & ((chi2 > 5.991) | (dpos == 0))
    level[out] = 1
    p2 = dict(prob)
    p2.update(cam_qt=cam, pt_xyz=pts, e_level=level, huber_delta=0.0)
    cam2, pts2, chi2b, dpos2, st2 = bundle_adjustment(ctx, p2, 10, chi2_in=chi2, **kw)
    erase = active & ((chi2b > 5.991) | (dpos2 == 0))
    return cam2, pts2, erase, st1, st2


def pose_optimization(ctx: Context, cam_qt, Xw, obs, info, K):
    cam = np.ascontiguousarray(cam_qt, np.float64).copy()
    Xw = np.ascontiguousarray(Xw, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    info = np.ascontiguousarray(info, np.float64)
    K = np.ascontiguousarray(K, np.float64)
    n = Xw.shape[0]
    outl = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int(0)
    check(lib().ccm_pose_optimize(ctx.handle, C.c_void_p(_vp(cam)), n, C.c_void_p(_vp(Xw)), C.c_void_p(_vp(obs)),
                                  C.c_void_p(_vp(info)), C.c_void_p(_vp(K)), C.c_void_p(_vp(outl)), C.byref(ninl)),
          ctx.handle)
    return cam, outl[:n], ninl.value


class PoseOptCall:
    """Call object for ccm_pose_optimize: arguments converted once; run() restores the initial pose (7 doubles) and makes exactly one C-ABI call
    (bench.py times this; pose_optimization() above spends ~10 us per call in numpy / ctypes conversions).  Results: .cam, .outlier, .n_inlier."""

    def __init__(self, ctx: Context, cam_qt, Xw, obs, info, K):
        f = lambda a: np.ascontiguousarray(a, np.float64)
        self.ctx = ctx
        self._cam0 = f(cam_qt).copy()
        self.cam = self._cam0.copy()
        self._in = (f(Xw), f(obs), f(info), f(K))
        n = self._in[0].shape[0]
        self.n = n
        self.outlier = np.zeros(max(n, 1), np.uint8)
        self._ninl = C.c_int(0)
        v = lambda a: C.c_void_p(_vp(a))
        self._args = (ctx.handle, v(self.cam), n, v(self._in[0]), v(self._in[1]), v(self._in[2]), v(self._in[3]), v(self.outlier), C.byref(self._ninl))
        self._fn = lib().ccm_pose_optimize

    def run(self):
        self.cam[:] = self._cam0
        rc = self._fn(*self._args)
        if rc:
            check(rc, self.ctx.handle)

    @property
    def n_inlier(self):
        return self._ninl.value


def sim3_optimization(ctx: Context, sim3, P1c, P2c, obs1, obs2, info1, info2, K1, K2, th2: float = 10.0, fix_scale: bool = False):
    """Optimizer::OptimizeSim3 (Optimizer.cpp:861-1056) through ccm_sim3_optimize.  Returns (sim3[8], inlier flags, nIn)."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    s = f(sim3).copy()
    P1c, P2c, obs1, obs2, info1, info2, K1, K2 = map(f, (P1c, P2c, obs1, obs2, info1, info2, K1, K2))
    n = P1c.shape[0] if P1c.ndim == 2 else P1c.size // 3
    inl = np.zeros(max(n, 1), np.uint8)
    nin = C.c_int(0)
    v = lambda a: C.c_void_p(_vp(a))
    check(lib().ccm_sim3_optimize(ctx.handle, v(s), n, v(P1c), v(P2c), v(obs1), v(obs2), v(info1), v(info2), v(K1), v(K2),
                                  C.c_double(th2), int(bool(fix_scale)), v(inl), C.byref(nin)), ctx.handle)
    return s, inl[:n], nin.value


class PGStats(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("lm_trials", C.c_int32), ("pcg_iters", C.c_int32), ("reserved", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double)]


def pose_graph_optimization(ctx: Context, pg: dict, max_iters: int = 20, lambda_init: float = 1e-16):
    """OptimizeEssentialGraph{LoopClosure,MapFusion} numerics (Optimizer.cpp:1058-1566) through ccm_pose_graph_optimize.
    pg: sim3[n,8], fixed[n], fix_scale, e_i, e_j, meas[m,8] (synth.make_pose_graph layout).  Returns (sim3, stats)."""
    sim3 = np.ascontiguousarray(pg["sim3"], np.float64).copy()
    fixed = np.ascontiguousarray(pg["fixed"], np.uint8)
    e_i, e_j = np.ascontiguousarray(pg["e_i"], np.int32), np.ascontiguousarray(pg["e_j"], np.int32)
    meas = np.ascontiguousarray(pg["meas"], np.float64)
    st = PGStats()
    v = lambda a: C.c_void_p(_vp(a))
    check(lib().ccm_pose_graph_optimize(ctx.handle, int(sim3.shape[0]), v(sim3), v(fixed), int(bool(pg["fix_scale"])), int(e_i.size), v(e_i), v(e_j),
                                        v(meas), int(max_iters), C.c_double(lambda_init), None, C.byref(st)), ctx.handle)
    return sim3, st


def debug_dense_solve(ctx: Context, A, b):
    """Test hook: SPD solve through the device's blocked MFMA-f64 Cholesky (ccm_debug_dense_solve)."""
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b); info = C.c_int(0)
    check(hooks().ccm_debug_dense_solve(ctx.handle, C.c_void_p(_vp(A)), C.c_void_p(_vp(b)), int(b.size), C.c_void_p(_vp(x)), C.byref(info)), ctx.handle)
    return x, info.value


def debug_tile_solve(ctx: Context, A, b):
    """Test hook: SPD solve through the tile-sparse, level-scheduled Cholesky (ccm_debug_tile_solve); returns (x, info, levels, tiles)."""
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b); info = C.c_int(0); levels = C.c_int(0); tiles = C.c_int(0)
    check(hooks().ccm_debug_tile_solve(ctx.handle, C.c_void_p(_vp(A)), C.c_void_p(_vp(b)), int(b.size), C.c_void_p(_vp(x)), C.byref(info),
                                     C.byref(levels), C.byref(tiles)), ctx.handle)
    return x, info.value, levels.value, tiles.value


def debug_dense_inverse(ctx: Context, A):
    """Test hook: explicit SPD inverse through the tile kernels of dense_chol.hip (ccm_debug_dense_inverse)."""
    A = np.ascontiguousarray(A, np.float64)
    out = np.zeros_like(A); info = C.c_int(0)
    check(hooks().ccm_debug_dense_inverse(ctx.handle, C.c_void_p(_vp(A)), int(A.shape[0]), C.c_void_p(_vp(out)), C.byref(info)), ctx.handle)
    return out, info.value
