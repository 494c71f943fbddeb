"""ctypes bindings of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Allowed importers: tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (ccm_slam_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class BAStats(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("lm_trials", C.c_int32), ("stop_reason", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("ms_residuals", C.c_double), ("ms_quadratic", C.c_double), ("ms_schur", C.c_double),
                ("ms_linear", C.c_double), ("ms_update", C.c_double), ("ms_structure", C.c_double),
                ("ms_total", C.c_double),
                ("n_free_cams", C.c_int32), ("n_active_pts", C.c_int32), ("n_active_edges", C.c_int32),
                ("n_schur_blocks", C.c_int32),
                ("chi2_hist", C.c_double * 64), ("lambda_hist", C.c_double * 64), ("trials_hist", C.c_int32 * 64)]


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h")) or f == "Makefile"]
    stale = os.path.exists(path) and any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs)
    if force or stale or not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return path


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.ora_ba_chi2.restype = C.c_double
        _LIB.ora_grid_candidates.restype = C.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


# ---- Hamming / matcher -------------------------------------------------------------------------
def descriptor_distance(a: np.ndarray, b: np.ndarray) -> int:
    return int(lib().ora_descriptor_distance(_p(np.ascontiguousarray(a, np.uint8), C.c_uint8),
                                             _p(np.ascontiguousarray(b, np.uint8), C.c_uint8)))


def update_normal_and_depth(pos, obs_off, obs_kf, kf_center, ref_kf, ref_level, scale_factors, normal, min_dist, max_dist):
    """MapPoint::UpdateNormalAndDepth restated (ora_update_normal_and_depth); normal / min_dist / max_dist are copied and returned."""
    f32, i32 = (lambda a: np.ascontiguousarray(a, np.float32)), (lambda a: np.ascontiguousarray(a, np.int32))
    pos, kf_center, sf = f32(pos), f32(kf_center), f32(scale_factors)
    obs_off, obs_kf, ref_kf, ref_level = i32(obs_off), i32(obs_kf), i32(ref_kf), i32(ref_level)
    nrm, dmin, dmax = f32(normal).copy(), f32(min_dist).copy(), f32(max_dist).copy()
    f = lib().ora_update_normal_and_depth
    f.restype = None
    f.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 3
    f(int(ref_kf.size), pos.ctypes.data, obs_off.ctypes.data, obs_kf.ctypes.data, kf_center.ctypes.data, ref_kf.ctypes.data, ref_level.ctypes.data,
      sf.ctypes.data, int(sf.size), nrm.ctypes.data, dmin.ctypes.data, dmax.ctypes.data)
    return nrm, dmin, dmax


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(3, np.int32)
    lib().ora_three_maxima(_p(counts, C.c_int32), int(counts.size), _p(out, C.c_int32))
    return tuple(int(x) for x in out)


def hamming_dense_best2(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    Q, T = q.shape[0], t.shape[0]
    bi, bd, sd = (np.empty(Q, np.int32) for _ in range(3))
    lib().ora_hamming_dense_best2(_p(q, C.c_uint8), Q, _p(t, C.c_uint8), T, _p(bi, C.c_int32), _p(bd, C.c_int32),
                                  _p(sd, C.c_int32))
    return bi, bd, sd


def hamming_csr(q, t, cand_off, cand_idx):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    cand_off = np.ascontiguousarray(cand_off, np.int32)
    cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    Q = q.shape[0]
    dist = np.empty(max(cand_idx.size, 1), np.uint16)
    bi, bd, sd = (np.empty(Q, np.int32) for _ in range(3))
    lib().ora_hamming_csr(_p(q, C.c_uint8), Q, _p(t, C.c_uint8), _p(cand_off, C.c_int32), _p(cand_idx, C.c_int32),
                          _p(dist, C.c_uint16), _p(bi, C.c_int32), _p(bd, C.c_int32), _p(sd, C.c_int32))
    return dist[:cand_idx.size], bi, bd, sd


def grid_candidates(kx, ky, octave, bounds, qx, qy, qr, qminl, qmaxl):
    kx, ky = np.ascontiguousarray(kx, np.float32), np.ascontiguousarray(ky, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
    qminl, qmaxl = np.ascontiguousarray(qminl, np.int32), np.ascontiguousarray(qmaxl, np.int32)
    Q = qx.size
    off = np.empty(Q + 1, np.int32)
    args = [_p(kx, C.c_float), _p(ky, C.c_float), _p(octave, C.c_int32), kx.size] + [C.c_float(b) for b in bounds] + \
           [_p(qx, C.c_float), _p(qy, C.c_float), _p(qr, C.c_float), _p(qminl, C.c_int32), _p(qmaxl, C.c_int32), Q]
    n = lib().ora_grid_candidates(*args, _p(off, C.c_int32), None, C.c_int64(0))
    idx = np.empty(max(n, 1), np.int32)
    lib().ora_grid_candidates(*args, _p(off, C.c_int32), _p(idx, C.c_int32), C.c_int64(n))
    return off, idx[:n]


def build_grid(kx, ky, bounds):
    kx, ky = np.ascontiguousarray(kx, np.float32), np.ascontiguousarray(ky, np.float32)
    off = np.zeros(75 * 48 + 1, np.int32)
    idx = np.zeros(max(kx.size, 1), np.int32)
    lib().ora_build_grid(_p(kx, C.c_float), _p(ky, C.c_float), int(kx.size), *[C.c_float(b) for b in bounds], _p(off, C.c_int32), _p(idx, C.c_int32))
    return off, idx[:off[-1]]


def is_in_frustum(frame24, n_levels, P, normal, dmin, dmax, cos_limit=0.5):
    """Frame::isInFrustum restated (ora_is_in_frustum).  frame24 = Rcw 9 | tcw 3 | Ow 3 | fx fy cx cy | minX maxX minY maxY | logSF."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    frame24, P, normal, dmin, dmax = f32(frame24), f32(P).reshape(-1, 3), f32(normal).reshape(-1, 3), f32(dmin), f32(dmax)
    n = P.shape[0]
    inv = np.zeros(max(n, 1), np.uint8); u = np.zeros(max(n, 1), np.float32); v = np.zeros(max(n, 1), np.float32)
    lvl = np.zeros(max(n, 1), np.int32); cs = np.zeros(max(n, 1), np.float32)
    lib().ora_is_in_frustum(_p(frame24, C.c_float), int(n_levels), n, _p(P, C.c_float), _p(normal, C.c_float), _p(dmin, C.c_float),
                            _p(dmax, C.c_float), C.c_float(cos_limit), _p(inv, C.c_uint8), _p(u, C.c_float), _p(v, C.c_float),
                            _p(lvl, C.c_int32), _p(cs, C.c_float))
    return inv[:n], u[:n], v[:n], lvl[:n], cs[:n]


def undistort_points(K, dist, xy):
    """Frame::UndistortKeyPoints restated (ora_undistort_points)."""
    K = np.ascontiguousarray(K, np.float32); dist = np.ascontiguousarray(dist, np.float32)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty_like(xy)
    lib().ora_undistort_points(_p(K, C.c_float), _p(dist, C.c_float), int(dist.size), _p(xy, C.c_float), xy.shape[0], _p(out, C.c_float))
    return out


def image_bounds(K, dist, w, h):
    K = np.ascontiguousarray(K, np.float32); dist = np.ascontiguousarray(dist, np.float32)
    b = np.zeros(4, np.float32)
    lib().ora_image_bounds(_p(K, C.c_float), _p(dist, C.c_float), int(dist.size), int(w), int(h), _p(b, C.c_float))
    return b


def search_by_projection_mp(kx, ky, octave, fdesc, bounds, scale_factors, mp_in_view, mp_proj_x, mp_proj_y, mp_level,
                            mp_view_cos, mp_desc, th, nnratio, frame_mp):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    kx, ky, sf = f32(kx), f32(ky), f32(scale_factors)
    octave = np.ascontiguousarray(octave, np.int32)
    fdesc = np.ascontiguousarray(fdesc, np.uint8)
    mp_in_view = np.ascontiguousarray(mp_in_view, np.uint8)
    px, py, vc = f32(mp_proj_x), f32(mp_proj_y), f32(mp_view_cos)
    lvl = np.ascontiguousarray(mp_level, np.int32)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    frame_mp = np.ascontiguousarray(frame_mp, np.int32).copy()
    n = lib().ora_search_by_projection_mp(
        _p(kx, C.c_float), _p(ky, C.c_float), _p(octave, C.c_int32), _p(fdesc, C.c_uint8), kx.size,
        *[C.c_float(b) for b in bounds], _p(sf, C.c_float), mp_in_view.size, _p(mp_in_view, C.c_uint8),
        _p(px, C.c_float), _p(py, C.c_float), _p(lvl, C.c_int32), _p(vc, C.c_float), _p(mp_desc, C.c_uint8),
        C.c_float(th), C.c_float(nnratio), _p(frame_mp, C.c_int32))
    return int(n), frame_mp


def search_by_projection_last(kx, ky, octave, kangle, fdesc, bounds, scale_factors, l_valid, l_u, l_v, l_octave,
                              l_angle, l_mp_desc, th, check_orientation, cur_mp):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    kx, ky, ka, sf = f32(kx), f32(ky), f32(kangle), f32(scale_factors)
    octave = np.ascontiguousarray(octave, np.int32)
    fdesc = np.ascontiguousarray(fdesc, np.uint8)
    l_valid = np.ascontiguousarray(l_valid, np.uint8)
    lu, lv, la = f32(l_u), f32(l_v), f32(l_angle)
    lo = np.ascontiguousarray(l_octave, np.int32)
    l_mp_desc = np.ascontiguousarray(l_mp_desc, np.uint8)
    cur_mp = np.ascontiguousarray(cur_mp, np.int32).copy()
    n = lib().ora_search_by_projection_last(
        _p(kx, C.c_float), _p(ky, C.c_float), _p(octave, C.c_int32), _p(ka, C.c_float), _p(fdesc, C.c_uint8), kx.size,
        *[C.c_float(b) for b in bounds], _p(sf, C.c_float), l_valid.size, _p(l_valid, C.c_uint8), _p(lu, C.c_float),
        _p(lv, C.c_float), _p(lo, C.c_int32), _p(la, C.c_float), _p(l_mp_desc, C.c_uint8), C.c_float(th),
        int(check_orientation), _p(cur_mp, C.c_int32))
    return int(n), cur_mp


# ---- bundle adjustment -------------------------------------------------------------------------
TRIAL_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int)


def ba_optimize(prob: dict, max_iters: int, linear_solver: int = 0, lambda_init: float = 0.0, chi2_in=None, stop_flag=None, trial_hook=None):
    """Runs the oracle LM on a copy of the problem state.  Returns (cam_qt, pt_xyz, chi2_per_edge,
    depth_pos, stats)."""
    cam = np.ascontiguousarray(prob["cam_qt"], np.float64).copy()
    pts = np.ascontiguousarray(prob["pt_xyz"], np.float64).copy()
    ne = int(prob["n_edge"])
    chi2 = np.zeros(ne, np.float64) if chi2_in is None else np.ascontiguousarray(chi2_in, np.float64).copy()
    dpos = np.zeros(ne, np.uint8)
    st = BAStats()
    lvl = prob.get("e_level")
    cb = None
    if trial_hook is not None:   # trial_hook(iteration, trial_in_iteration, chi2_trial, accepted) after every LM trial
        cb = TRIAL_CB(lambda user, it, tr, chi, acc: trial_hook(it, tr, chi, acc))
        lib().ora_ba_set_trial_hook(cb, None)
    try:
        _ba_call(prob, ne, cam, pts, lvl, max_iters, linear_solver, lambda_init, stop_flag, chi2, dpos, st)
    finally:
        if cb is not None:
            lib().ora_ba_set_trial_hook(C.cast(None, TRIAL_CB), None)
    return cam, pts, chi2, dpos, st


def _ba_call(prob, ne, cam, pts, lvl, max_iters, linear_solver, lambda_init, stop_flag, chi2, dpos, st):
    lib().ora_ba_optimize(
        int(prob["n_cam"]), int(prob["n_pt"]), ne, _p(cam, C.c_double), _p(np.ascontiguousarray(prob["cam_fixed"], np.uint8), C.c_uint8),
        _p(np.ascontiguousarray(prob["cam_K"], np.float64), C.c_double), _p(pts, C.c_double),
        _p(np.ascontiguousarray(prob["e_cam"], np.int32), C.c_int32), _p(np.ascontiguousarray(prob["e_pt"], np.int32), C.c_int32),
        _p(np.ascontiguousarray(prob["e_obs"], np.float64), C.c_double), _p(np.ascontiguousarray(prob["e_info"], np.float64), C.c_double),
        _p(np.ascontiguousarray(lvl, np.uint8), C.c_uint8) if lvl is not None else None,
        C.c_double(prob["huber_delta"]), int(max_iters), int(linear_solver), C.c_double(lambda_init),
        stop_flag.ctypes.data_as(C.c_void_p) if stop_flag is not None else None,
        _p(chi2, C.c_double), _p(dpos, C.c_uint8), C.byref(st))


def ba_chi2(prob: dict, cam=None, pts=None) -> float:
    cam = np.ascontiguousarray(prob["cam_qt"] if cam is None else cam, np.float64)
    pts = np.ascontiguousarray(prob["pt_xyz"] if pts is None else pts, np.float64)
    lvl = prob.get("e_level")
    return float(lib().ora_ba_chi2(
        int(prob["n_cam"]), int(prob["n_pt"]), int(prob["n_edge"]), _p(cam, C.c_double),
        _p(np.ascontiguousarray(prob["cam_K"], np.float64), C.c_double), _p(pts, C.c_double),
        _p(np.ascontiguousarray(prob["e_cam"], np.int32), C.c_int32), _p(np.ascontiguousarray(prob["e_pt"], np.int32), C.c_int32),
        _p(np.ascontiguousarray(prob["e_obs"], np.float64), C.c_double), _p(np.ascontiguousarray(prob["e_info"], np.float64), C.c_double),
        _p(np.ascontiguousarray(lvl, np.uint8), C.c_uint8) if lvl is not None else None, C.c_double(prob["huber_delta"])))


def ba_partial_system(prob: dict, lam: float, pt_lo: int, pt_hi: int, add_lambda: bool):
    """(Hschur dense (n,n), bschur (n,), partial robust chi2) of the landmark shard [pt_lo, pt_hi)."""
    args = [int(prob["n_cam"]), int(prob["n_pt"]), int(prob["n_edge"]),
            _p(np.ascontiguousarray(prob["cam_qt"], np.float64), C.c_double),
            _p(np.ascontiguousarray(prob["cam_fixed"], np.uint8), C.c_uint8),
            _p(np.ascontiguousarray(prob["cam_K"], np.float64), C.c_double),
            _p(np.ascontiguousarray(prob["pt_xyz"], np.float64), C.c_double),
            _p(np.ascontiguousarray(prob["e_cam"], np.int32), C.c_int32), _p(np.ascontiguousarray(prob["e_pt"], np.int32), C.c_int32),
            _p(np.ascontiguousarray(prob["e_obs"], np.float64), C.c_double), _p(np.ascontiguousarray(prob["e_info"], np.float64), C.c_double),
            _p(np.ascontiguousarray(prob["e_level"], np.uint8), C.c_uint8), C.c_double(prob["huber_delta"]),
            C.c_double(lam), int(pt_lo), int(pt_hi), int(add_lambda)]
    ncp = C.c_int(0)
    lib().ora_ba_partial_system(*args, None, None, None, C.byref(ncp))
    n = ncp.value * 6
    H = np.zeros((n, n), np.float64)
    b = np.zeros(n, np.float64)
    chi = C.c_double(0)
    lib().ora_ba_partial_system(*args, _p(H, C.c_double), _p(b, C.c_double), C.byref(chi), C.byref(ncp))
    return H, b, chi.value


def pose_optimize(cam_qt, Xw, obs, info, K):
    cam = np.ascontiguousarray(cam_qt, np.float64).copy()
    Xw = np.ascontiguousarray(Xw, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    info = np.ascontiguousarray(info, np.float64)
    K = np.ascontiguousarray(K, np.float64)
    n = Xw.shape[0]
    outl = np.zeros(max(n, 1), np.uint8)
    ninl = lib().ora_pose_optimize(_p(cam, C.c_double), n, _p(Xw, C.c_double), _p(obs, C.c_double), _p(info, C.c_double),
                                   _p(K, C.c_double), _p(outl, C.c_uint8))
    return cam, outl[:n], int(ninl)


def sim3_optimize(sim3, P1c, P2c, obs1, obs2, info1, info2, K1, K2, th2=10.0, fix_scale=False):
    """Optimizer::OptimizeSim3 restated (ora_sim3_optimize).  Returns (sim3, inlier flags, nIn)."""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    s = f(sim3).copy()
    P1c, P2c, obs1, obs2, info1, info2, K1, K2 = map(f, (P1c, P2c, obs1, obs2, info1, info2, K1, K2))
    n = P1c.shape[0]
    inl = np.zeros(max(n, 1), np.uint8)
    fn = lib().ora_sim3_optimize
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_double, C.c_int, C.c_void_p]
    nin = fn(s.ctypes.data, n, P1c.ctypes.data, P2c.ctypes.data, obs1.ctypes.data, obs2.ctypes.data, info1.ctypes.data,
             info2.ctypes.data, K1.ctypes.data, K2.ctypes.data, float(th2), int(bool(fix_scale)), inl.ctypes.data)
    return s, inl[:n], int(nin)


class PGStats(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("lm_trials", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double)]


def _vec(fn, inp, nout):
    inp = np.ascontiguousarray(inp, np.float64)
    out = np.zeros(nout)
    f = getattr(lib(), fn)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p]
    f(inp.ctypes.data, out.ctypes.data)
    return out


def sim3_exp_log(u7): return _vec("ora_sim3_exp_log", u7, 7)
def sim3_exp(u7): return _vec("ora_sim3_exp", u7, 8)
def se3_exp(u6): return _vec("ora_se3_exp", u6, 7)


def pose_graph_optimize(pg, max_iters=20, lambda_init=1e-16):
    """OptimizeEssentialGraph* numerics restated (ora_pose_graph_optimize).  Returns (sim3[n,8], stats)."""
    sim3 = np.ascontiguousarray(pg["sim3"], np.float64).copy()
    fixed = np.ascontiguousarray(pg["fixed"], np.uint8)
    e_i, e_j = np.ascontiguousarray(pg["e_i"], np.int32), np.ascontiguousarray(pg["e_j"], np.int32)
    meas = np.ascontiguousarray(pg["meas"], np.float64)
    st = PGStats()
    fn = lib().ora_pose_graph_optimize
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    fn(sim3.shape[0], sim3.ctypes.data, fixed.ctypes.data, int(bool(pg["fix_scale"])), e_i.size, e_i.ctypes.data, e_j.ctypes.data,
       meas.ctypes.data, int(max_iters), float(lambda_init), C.addressof(st))
    return sim3, st


# ---- ORB extractor ---------------------------------------------------------------------------------
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32)])


class OrbOracle:
    """cslam::ORBextractor restated (oracle/orb_ref.cpp)."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7):
        L = lib()
        L.ora_orb_create.restype = C.c_void_p
        self.nlevels, self.nfeatures = nlevels, nfeatures
        self._h = C.c_void_p(L.ora_orb_create(int(nfeatures), C.c_float(scale), int(nlevels), int(ini_th), int(min_th)))

    def close(self):
        if self._h:
            lib().ora_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, img: np.ndarray):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = self.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = lib().ora_orb_extract(self._h, _p(img, C.c_uint8), w, h, w, kps.ctypes.data_as(C.c_void_p), _p(desc, C.c_uint8), cap)
        self._wh = (w, h)
        return kps[:n].copy(), desc[:n].copy()

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        lib().ora_orb_level_size(self._h, w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def level(self, level):
        lw, lh = self.level_size(*self._wh, level)
        out = np.empty((lh, lw), np.uint8)
        lib().ora_orb_get_level(self._h, level, _p(out, C.c_uint8))
        return out

    def blur(self, level):
        lw, lh = self.level_size(*self._wh, level)
        out = np.empty((lh, lw), np.uint8)
        lib().ora_orb_get_blur(self._h, level, _p(out, C.c_uint8))
        return out

    def candidates(self, level):
        n = lib().ora_orb_get_candidates(self._h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().ora_orb_get_candidates(self._h, level, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]

    def tables(self):
        nl = self.nlevels
        sf, isf, s2, is2 = (np.zeros(nl, np.float32) for _ in range(4))
        nf = np.zeros(nl, np.int32)
        um = np.zeros(16, np.int32)
        lib().ora_orb_tables(self._h, _p(sf, C.c_float), _p(isf, C.c_float), _p(s2, C.c_float), _p(is2, C.c_float),
                             _p(nf, C.c_int32), _p(um, C.c_int32))
        return sf, isf, s2, is2, nf, um


def resize_linear_u8(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.empty((dh, dw), np.uint8)
    lib().ora_resize_linear_u8(_p(src, C.c_uint8), src.shape[1], src.shape[0], _p(out, C.c_uint8), dw, dh)
    return out


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.empty_like(src)
    lib().ora_gaussian_blur7(_p(src, C.c_uint8), src.shape[1], src.shape[0], _p(out, C.c_uint8))
    return out


def gaussian_kernel7():
    k = np.zeros(7, np.int32)
    lib().ora_gaussian_kernel7(_p(k, C.c_int32))
    return k


def fast9_16(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(max(cap, 1), KP_DTYPE)
    n = lib().ora_fast9_16(_p(img, C.c_uint8), img.shape[1], img.shape[0], int(threshold), out.ctypes.data_as(C.c_void_p), cap)
    return out[:n]


def fast_atan2(y, x):
    lib().ora_fast_atan2.restype = C.c_float
    return float(lib().ora_fast_atan2(C.c_float(y), C.c_float(x)))


def distribute_octree(kps, minX, maxX, minY, maxY, N):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros(max(kps.size, 1), KP_DTYPE)
    n = lib().ora_distribute_octree(kps.ctypes.data_as(C.c_void_p), kps.size, minX, maxX, minY, maxY, N,
                                    out.ctypes.data_as(C.c_void_p), out.size)
    return out[:n]


# ---- BoW-bucketed / triangulation / initialisation searches ------------------------------------------
def _fv(fv):
    node, off, idx = (np.ascontiguousarray(a, np.int32) for a in fv)
    return node, off, idx


def search_by_bow_kf_frame(fv_kf, fv_f, kf_has_mp, kf_desc, kf_angle, f_desc, f_angle, nnratio, check_ori):
    n1, o1, i1 = _fv(fv_kf); n2, o2, i2 = _fv(fv_f)
    f_n = f_desc.shape[0]
    out = np.zeros(f_n, np.int32)
    n = lib().ora_search_by_bow_kf_frame(_p(n1, C.c_int32), _p(o1, C.c_int32), _p(i1, C.c_int32), n1.size, _p(n2, C.c_int32), _p(o2, C.c_int32),
                                         _p(i2, C.c_int32), n2.size, _p(np.ascontiguousarray(kf_has_mp, np.uint8), C.c_uint8),
                                         _p(np.ascontiguousarray(kf_desc, np.uint8), C.c_uint8), _p(np.ascontiguousarray(kf_angle, np.float32), C.c_float),
                                         _p(np.ascontiguousarray(f_desc, np.uint8), C.c_uint8), _p(np.ascontiguousarray(f_angle, np.float32), C.c_float),
                                         f_n, C.c_float(nnratio), int(check_ori), _p(out, C.c_int32))
    return int(n), out


def search_by_bow_kf_kf(fv1, fv2, has1, has2, d1, a1, d2, a2, nnratio, check_ori):
    n1, o1, i1 = _fv(fv1); n2, o2, i2 = _fv(fv2)
    out = np.zeros(d1.shape[0], np.int32)
    n = lib().ora_search_by_bow_kf_kf(_p(n1, C.c_int32), _p(o1, C.c_int32), _p(i1, C.c_int32), n1.size, _p(n2, C.c_int32), _p(o2, C.c_int32),
                                      _p(i2, C.c_int32), n2.size, _p(np.ascontiguousarray(has1, np.uint8), C.c_uint8), _p(np.ascontiguousarray(has2, np.uint8), C.c_uint8),
                                      _p(np.ascontiguousarray(d1, np.uint8), C.c_uint8), _p(np.ascontiguousarray(a1, np.float32), C.c_float), d1.shape[0],
                                      _p(np.ascontiguousarray(d2, np.uint8), C.c_uint8), _p(np.ascontiguousarray(a2, np.float32), C.c_float), d2.shape[0],
                                      C.c_float(nnratio), int(check_ori), _p(out, C.c_int32))
    return int(n), out


def search_for_triangulation(fv1, fv2, has1, has2, d1, x1, y1, a1, d2, x2, y2, oct2, a2, F12, ex, ey, sigma2_2, sf2, check_ori):
    n1, o1, i1 = _fv(fv1); n2, o2, i2 = _fv(fv2)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    out = np.zeros(d1.shape[0], np.int32)
    n = lib().ora_search_for_triangulation(_p(n1, C.c_int32), _p(o1, C.c_int32), _p(i1, C.c_int32), n1.size, _p(n2, C.c_int32), _p(o2, C.c_int32),
                                           _p(i2, C.c_int32), n2.size, _p(np.ascontiguousarray(has1, np.uint8), C.c_uint8), _p(np.ascontiguousarray(has2, np.uint8), C.c_uint8),
                                           _p(np.ascontiguousarray(d1, np.uint8), C.c_uint8), _p(f32(x1), C.c_float), _p(f32(y1), C.c_float), _p(f32(a1), C.c_float), d1.shape[0],
                                           _p(np.ascontiguousarray(d2, np.uint8), C.c_uint8), _p(f32(x2), C.c_float), _p(f32(y2), C.c_float),
                                           _p(np.ascontiguousarray(oct2, np.int32), C.c_int32), _p(f32(a2), C.c_float), d2.shape[0], _p(f32(F12), C.c_float),
                                           C.c_float(ex), C.c_float(ey), _p(f32(sigma2_2), C.c_float), _p(f32(sf2), C.c_float), int(check_ori), _p(out, C.c_int32))
    return int(n), out


def search_for_initialization(k1, d1, k2, d2, bounds, prev_xy, window, nnratio, check_ori):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    prev = f32(prev_xy).copy()
    out = np.zeros(len(k1), np.int32)
    n = lib().ora_search_for_initialization(_p(f32(k1["x"]), C.c_float), _p(f32(k1["y"]), C.c_float), _p(np.ascontiguousarray(k1["octave"], np.int32), C.c_int32),
                                            _p(f32(k1["angle"]), C.c_float), _p(np.ascontiguousarray(d1, np.uint8), C.c_uint8), len(k1),
                                            _p(f32(k2["x"]), C.c_float), _p(f32(k2["y"]), C.c_float), _p(np.ascontiguousarray(k2["octave"], np.int32), C.c_int32),
                                            _p(f32(k2["angle"]), C.c_float), _p(np.ascontiguousarray(d2, np.uint8), C.c_uint8), len(k2),
                                            *[C.c_float(b) for b in bounds], _p(prev, C.c_float), int(window), C.c_float(nnratio), int(check_ori), _p(out, C.c_int32))
    return int(n), out, prev


def projected_window_search(kx, ky, octave, kdesc, bounds, scale_factors, inv_sigma2, valid, u, v, level, pdesc, th, chi2_gate,
                            dist_threshold, matched=None, claim=False, no_claim=None):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    n_pts = len(u)
    bi = np.zeros(n_pts, np.int32); bd = np.zeros(n_pts, np.int32)
    m = None if matched is None else np.ascontiguousarray(matched, np.int32).copy()
    nc = None if no_claim is None else np.ascontiguousarray(no_claim, np.uint8)
    n = lib().ora_projected_window_search(
        _p(f32(kx), C.c_float), _p(f32(ky), C.c_float), _p(np.ascontiguousarray(octave, np.int32), C.c_int32),
        _p(np.ascontiguousarray(kdesc, np.uint8), C.c_uint8), len(kx), *[C.c_float(b) for b in bounds], _p(f32(scale_factors), C.c_float),
        _p(f32(inv_sigma2), C.c_float), n_pts, _p(np.ascontiguousarray(valid, np.uint8), C.c_uint8), _p(f32(u), C.c_float), _p(f32(v), C.c_float),
        _p(np.ascontiguousarray(level, np.int32), C.c_int32), _p(np.ascontiguousarray(pdesc, np.uint8), C.c_uint8), C.c_float(th), int(chi2_gate),
        int(dist_threshold), _p(m, C.c_int32) if m is not None else None, int(claim), _p(nc, C.c_uint8) if nc is not None else None,
        _p(bi, C.c_int32), _p(bd, C.c_int32))
    return int(n), bi, bd, m


def distinctive_descriptors(desc, off):
    desc = np.ascontiguousarray(desc, np.uint8); off = np.ascontiguousarray(off, np.int32)
    out = np.zeros(off.size - 1, np.int32)
    lib().ora_distinctive_descriptors(_p(desc, C.c_uint8), _p(off, C.c_int32), off.size - 1, _p(out, C.c_int32))
    return out


def bow_transform(vocab: dict, desc, levelsup=4):
    desc = np.ascontiguousarray(desc, np.uint8)
    N = desc.shape[0]
    word = np.zeros(N, np.int32); w = np.zeros(N, np.float64); node = np.zeros(N, np.int32)
    bid = np.zeros(max(N, 1), np.int32); bval = np.zeros(max(N, 1), np.float64)
    n = lib().ora_bow_transform(vocab["n_nodes"], vocab["L"], _p(vocab["child_off"], C.c_int32), _p(vocab["child_id"], C.c_int32),
                                _p(vocab["node_desc"], C.c_uint8), _p(vocab["word_id"], C.c_int32), _p(vocab["weight"], C.c_double),
                                _p(desc, C.c_uint8), N, int(levelsup), _p(word, C.c_int32), _p(w, C.c_double), _p(node, C.c_int32),
                                _p(bid, C.c_int32), _p(bval, C.c_double))
    return word, w, node, bid[:n], bval[:n]


def ba_optimize_fast(prob: dict, max_iters: int):
    """the 'optimistic CPU' line of bench.py: the same restatement built -O3 -march=native (oracle/Makefile: liboracle_fast.so), timing only"""
    path = os.path.join(_HERE, "liboracle_fast.so")
    src = os.path.join(_HERE, "ba_ref.cpp")
    # -march=native code must be built on the machine that runs it: rebuild when the CPU (model + flags) differs from the build's
    import hashlib
    try:
        cpu = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
    except OSError:
        cpu = []
    stamp, want = path + ".cpu", hashlib.sha1("".join(cpu).encode()).hexdigest()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path) or have != want:
        if os.path.exists(path):
            os.remove(path)
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_fast.so"])
        open(stamp, "w").write(want)
    global _LIB
    lib()
    saved = _LIB
    _LIB = C.CDLL(path)
    try:
        _, _, _, _, st = ba_optimize(prob, max_iters)
    finally:
        _LIB = saved
    work = st.ms_total - st.ms_structure
    return {"flags": "-O3 -march=native -funroll-loops (AVX-512, FMA), 1 thread", "ms_per_iter": round(work / max(st.iters_done, 1), 2),
            "ms_per_trial": round(work / max(st.lm_trials, 1), 2), "linear_solve_ms": round(st.ms_linear, 1),
            "note": "g2o's Eigen SimplicialLDLT is single-threaded and G2O_OPENMP is off in the reference build, so more threads would touch at most the "
                    "~20 % outside the linear solve"}


def to_se3quat(Tcw) -> np.ndarray:
    """Converter::toSE3Quat restated: 4x4 CV_32F pose (or an array of them) -> [qx qy qz qw tx ty tz] f64"""
    T = np.ascontiguousarray(Tcw, np.float32).reshape(-1, 16)
    out = np.zeros((T.shape[0], 7))
    f = lib().ora_to_se3quat
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p]
    for i in range(T.shape[0]):
        f(T[i].ctypes.data, out[i].ctypes.data)
    return out


def se3quat_to_cvmat(qt) -> np.ndarray:
    """Converter::toCvMat(SE3Quat) restated: [qx qy qz qw tx ty tz] f64 -> 4x4 CV_32F"""
    q = np.ascontiguousarray(qt, np.float64).reshape(-1, 7)
    out = np.zeros((q.shape[0], 16), np.float32)
    f = lib().ora_se3quat_to_cvmat
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p]
    for i in range(q.shape[0]):
        f(q[i].ctypes.data, out[i].ctypes.data)
    return out.reshape(-1, 4, 4)
