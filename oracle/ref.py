"""ctypes binding of oracle/_ref/ — binaries built from the REFERENCE'S OWN sources (oracle/Makefile.ref) against look-alike
third-party headers (oracle/ref_shim/).  TEST INFRASTRUCTURE: the oracle restatement (oracle/*.cpp) is pinned against these in
tests/test_ref_*.py.  /root/reference exists only in the build container; on the GPU box the prebuilt .so files are used."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build() -> None:
    """(re)build oracle/_ref from /root/reference when it is present; a no-op on the GPU box"""
    subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref", "-s", "-j8"])


def available(name: str = "libg2o_ref.so") -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", name))


def _lib(name):
    if name not in _libs:
        path = name if os.path.isabs(name) else os.path.join(_HERE, "_ref", name)
        if not os.path.exists(path):
            build()
        _libs[name] = C.CDLL(path)
    return _libs[name]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class RefBAStats(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("lm_trials", C.c_int32), ("n_hist", C.c_int32), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("chi2_hist", C.c_double * 64), ("trials_hist", C.c_int32 * 64)]


def g2o_ba_optimize(prob: dict, max_iters: int, dense_solver: bool = False, lambda_init: float = 0.0, chi2_in=None):
    """The reference's g2o on the flat problem, graph built as Optimizer::MapFusionGBA / LocalBundleAdjustmentClient do.
    Returns (cam_qt, pt_xyz, chi2_per_edge, depth_pos, stats) like oracle.ba_optimize."""
    cam = np.ascontiguousarray(prob["cam_qt"], np.float64).copy()
    pts = np.ascontiguousarray(prob["pt_xyz"], np.float64).copy()
    ne = int(prob["n_edge"])
    chi2 = np.zeros(ne) if chi2_in is None else np.ascontiguousarray(chi2_in, np.float64).copy()
    dpos = np.zeros(ne, np.uint8)
    lvl = prob.get("e_level")
    lvl = np.ascontiguousarray(lvl, np.uint8) if lvl is not None else None
    keep = [np.ascontiguousarray(prob[k], t) for k, t in (("cam_fixed", np.uint8), ("cam_K", np.float64), ("e_cam", np.int32), ("e_pt", np.int32),
                                                          ("e_obs", np.float64), ("e_info", np.float64))]
    st = RefBAStats()
    f = _lib("libg2o_ref.so").ref_ba_optimize
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 3 + [C.c_void_p] * 9 + [C.c_double, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(int(prob["n_cam"]), int(prob["n_pt"]), ne, _p(cam), _p(keep[0]), _p(keep[1]), _p(pts), _p(keep[2]), _p(keep[3]), _p(keep[4]), _p(keep[5]), _p(lvl),
      float(prob["huber_delta"]), int(max_iters), int(bool(dense_solver)), float(lambda_init), None, _p(chi2), _p(dpos), C.byref(st))
    return cam, pts, chi2, dpos, st


def g2o_pose_optimize(cam_qt, Xw, obs, info, K):
    cam = np.ascontiguousarray(cam_qt, np.float64).copy()
    Xw, obs, info, K = (np.ascontiguousarray(a, np.float64) for a in (Xw, obs, info, K))
    n = Xw.shape[0]
    outl = np.zeros(max(n, 1), np.uint8)
    f = _lib("libg2o_ref.so").ref_pose_optimize
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    ninl = f(_p(cam), n, _p(Xw), _p(obs), _p(info), _p(K), _p(outl))
    return cam, outl[:n], ninl


def g2o_sim3_optimize(sim3, P1c, P2c, obs1, obs2, info1, info2, K1, K2, th2=10.0, fix_scale=False):
    s = np.ascontiguousarray(sim3, np.float64).copy()
    arrs = [np.ascontiguousarray(a, np.float64) for a in (P1c, P2c, obs1, obs2, info1, info2, K1, K2)]
    n = arrs[0].shape[0]
    inl = np.zeros(max(n, 1), np.uint8)
    f = _lib("libg2o_ref.so").ref_sim3_optimize
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_double, C.c_int, C.c_void_p]
    nin = f(_p(s), n, *[_p(a) for a in arrs], float(th2), int(bool(fix_scale)), _p(inl))
    return s, inl[:n], nin


class RefPGStats(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("lm_trials", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double)]


def g2o_pose_graph_optimize(pg, max_iters=20, lambda_init=1e-16):
    sim3 = np.ascontiguousarray(pg["sim3"], np.float64).copy()
    fixed = np.ascontiguousarray(pg["fixed"], np.uint8)
    e_i, e_j = np.ascontiguousarray(pg["e_i"], np.int32), np.ascontiguousarray(pg["e_j"], np.int32)
    meas = np.ascontiguousarray(pg["meas"], np.float64)
    st = RefPGStats()
    f = _lib("libg2o_ref.so").ref_pose_graph_optimize
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    f(int(sim3.shape[0]), _p(sim3), _p(fixed), int(bool(pg["fix_scale"])), int(e_i.size), _p(e_i), _p(e_j), _p(meas), int(max_iters), float(lambda_init),
      C.byref(st))
    return sim3, st


def _vec(fn, inp, nout):
    inp = np.ascontiguousarray(inp, np.float64)
    out = np.zeros(nout)
    f = getattr(_lib("libg2o_ref.so"), fn)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p]
    f(_p(inp), _p(out))
    return out


def se3_exp(u6): return _vec("ref_se3_exp", u6, 7)
def se3_log(qt7): return _vec("ref_se3_log", qt7, 6)
def sim3_exp(u7): return _vec("ref_sim3_exp", u7, 8)
def sim3_log(s8): return _vec("ref_sim3_log", s8, 7)


def huber(delta, e2):
    out = np.zeros(3)
    f = _lib("libg2o_ref.so").ref_huber
    f.restype = None
    f.argtypes = [C.c_double, C.c_double, C.c_void_p]
    f(float(delta), float(e2), _p(out))
    return out


def edge_se3_project(cam_qt, K, X, obs):
    """EdgeSE3ProjectXYZ at a state: (error[2], J_point[2,3], J_cam[2,6]) from the reference's computeError / linearizeOplus"""
    a = [np.ascontiguousarray(x, np.float64) for x in (cam_qt, K, X, obs)]
    err, Jp, Jc = np.zeros(2), np.zeros((2, 3)), np.zeros((2, 6))
    f = _lib("libg2o_ref.so").ref_edge_se3_project
    f.restype = None
    f.argtypes = [C.c_void_p] * 7
    f(*[_p(x) for x in a], _p(err), _p(Jp), _p(Jc))
    return err, Jp, Jc


# ---- the reference's own ORBextractor (cslam/src/ORBextractor.cpp compiled verbatim) and ORBmatcher statics -----------------------
KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32), ("response", np.float32),
                     ("octave", np.int32)])


class RefOrb:
    """cslam::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) of the reference"""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, lib_path="liborb_ref.so"):
        """lib_path: the library that provides cslam::ORBextractor behind oracle/ref_orb_driver.cpp — the reference's own ORBextractor.cpp
        (default) or shim/liborb_hip_shim.so (OUR drop-in translation unit, on the MI355X)"""
        self._libname = lib_path
        L = _lib(lib_path)
        L.ref_orb_create.restype = C.c_void_p
        L.ref_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        self._h = C.c_void_p(L.ref_orb_create(nfeatures, scale, nlevels, ini_th, min_th))
        self.nfeatures, self.nlevels, self.scale, self.ini_th, self.min_th = nfeatures, nlevels, scale, ini_th, min_th

    def extract_cli(self, img):
        """The same extraction in a separate process whose allocator hands out increasing addresses (oracle/ref_orb_cli.cpp): the
        reference's pointer tie-break in DistributeOctTree becomes node creation order, i.e. deterministic and comparable."""
        import tempfile
        img = np.ascontiguousarray(img, np.uint8)
        exe = os.path.join(_HERE, "_ref", "orb_ref_cli")
        if not os.path.exists(exe):
            build()
        with tempfile.TemporaryDirectory() as d:
            fin, fout = os.path.join(d, "in.raw"), os.path.join(d, "out.bin")
            img.tofile(fin)
            subprocess.check_call([exe, fin, str(img.shape[1]), str(img.shape[0]), str(self.nfeatures), repr(float(self.scale)), str(self.nlevels),
                                   str(self.ini_th), str(self.min_th), fout])
            raw = open(fout, "rb").read()
        n = int(np.frombuffer(raw, np.int32, 1)[0])
        kps = np.frombuffer(raw, KP_DTYPE, n, 4).copy()
        desc = np.frombuffer(raw, np.uint8, n * 32, 4 + n * KP_DTYPE.itemsize).reshape(n, 32).copy()
        return kps, desc

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        cap = 4 * self.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        f = _lib(self._libname).ref_orb_extract
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        n = f(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap)
        assert n <= cap
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l):
        f = _lib(self._libname).ref_orb_get_level
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        w, h = C.c_int(), C.c_int()
        f(self._h, l, None, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        f(self._h, l, _p(out), C.byref(w), C.byref(h))
        return out

    def border_pixel(self, l, row, col):
        f = _lib(self._libname).ref_orb_border_pixel
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        return f(self._h, l, row, col)

    def tables(self):
        out = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        f = _lib(self._libname).ref_orb_tables
        f.restype = None
        f.argtypes = [C.c_void_p] * 5
        f(self._h, *[_p(a) for a in out])
        return out

    def close(self):
        if self._h:
            f = _lib(self._libname).ref_orb_destroy
            f.restype = None
            f.argtypes = [C.c_void_p]
            f(self._h)
            self._h = None


def descriptor_distance(a, b) -> int:
    a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
    f = _lib("liborb_ref.so").ref_descriptor_distance
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p]
    return f(_p(a), _p(b))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(3, np.int32)
    f = _lib("liborb_ref.so").ref_three_maxima
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    f(_p(counts), int(counts.size), _p(out))
    return tuple(int(x) for x in out)
