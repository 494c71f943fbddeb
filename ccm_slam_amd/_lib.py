"""ctypes binding of libccm_hip.so (the C ABI declared in include/ccm_hip.h).

The library is the product: there is no CPU fallback.  Loading works without a GPU (so that the
CPU test-suite can check the exported symbols); creating a context without a gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libccm_hip.so")
_lib = None


class CcmError(RuntimeError):
    pass


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32)]


class BAProblem(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("n_pt", C.c_int32), ("n_edge", C.c_int32),
                ("cam_qt", C.c_void_p), ("cam_fixed", C.c_void_p), ("cam_K", C.c_void_p), ("pt_xyz", C.c_void_p),
                ("e_cam", C.c_void_p), ("e_pt", C.c_void_p), ("e_obs", C.c_void_p), ("e_info", C.c_void_p),
                ("e_level", C.c_void_p), ("huber_delta", C.c_double)]


class BAOptions(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("pcg_max_iters", C.c_int32), ("pcg_rel_tol", C.c_double),
                ("lambda_init", C.c_double), ("verbose", C.c_int32)]


class BAStats(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("lm_trials", C.c_int32), ("pcg_iters", C.c_int32),
                ("stop_reason", C.c_int32), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("ms_setup", C.c_double), ("ms_total", C.c_double),
                ("ms_iters", C.c_double), ("n_schur_blocks", C.c_int32), ("n_pair_instances", C.c_int64)]


# kernel classes for ccm_prof_* (enum in ccm_hip.h)
K = {name: i for i, name in enumerate([
    "HAMMING_DENSE", "HAMMING_CSR", "PYR_RESIZE", "FAST_SCORE", "FAST_NMS", "ORIENT", "BLUR", "BRIEF",
    "BA_LINEARIZE", "BA_CAM", "BA_DINV", "BA_SCHUR_DIAG", "BA_SCHUR_OFF", "BA_PCG_SPMV", "BA_PCG_UPDATE",
    "BA_BACKSUB", "BA_UPDATE", "BA_CHI2", "POSEOPT", "SIM3OPT", "BA_PCG_PERSIST", "BA_COARSE", "BA_REDUCE", "BA_ALLREDUCE"])}


def build(force: bool = False) -> str:
    """Compile libccm_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean", "-s"])
    subprocess.check_call(["make", "-C", src, "-s", "-j4"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CcmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is required; there is no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.ccm_last_error.restype = C.c_char_p
        _lib.ccm_version.restype = C.c_char_p
        _lib.ccm_ctx_destroy.restype = None
        _lib.ccm_ba_destroy.restype = None
        if hasattr(_lib, "ccm_orb_destroy"):
            _lib.ccm_orb_destroy.restype = None
    return _lib


_hooks = None


def hooks():
    """libccm_testhooks.so: the TEST-ONLY entry points (include/ccm_testhooks.h).  A separate library on top of the product; tests/ and scripts/ only."""
    global _hooks
    if _hooks is None:
        lib()   # the product library first: the hooks link against it
        path = os.path.join(_HERE, "libccm_testhooks.so")
        if not os.path.exists(path):
            raise CcmError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _hooks = C.CDLL(path)
        _hooks.ccm_comm_loopback_destroy.restype = None
    return _hooks


def check(rc: int, ctx=None):
    if rc != 0:
        msg = lib().ccm_last_error(ctx)
        raise CcmError(f"libccm_hip error {rc}: {msg.decode() if msg else '?'}")


class Context:
    """One ccm_ctx (device + stream).  Not thread-safe: use one per thread, like the reference uses
    one ORBextractor / optimizer per thread."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().ccm_ctx_create(int(device), C.byref(self._h)))
        self.device = device
        self._children = weakref.WeakSet()   # handles (ORB, BA) that must be destroyed before the ctx

    def adopt(self, child):
        self._children.add(child)

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(lib().ccm_ctx_sync(self._h), self._h)

    def close(self):
        if self._h:
            for ch in list(self._children):
                try:
                    ch.close()
                except Exception:
                    pass
            lib().ccm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # device memory helpers -------------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib().ccm_dev_alloc(self._h, C.c_size_t(nbytes), C.byref(p)), self._h)
        return p.value

    def free(self, dptr: int):
        check(lib().ccm_dev_free(self._h, C.c_void_p(dptr)), self._h)

    def h2d(self, dptr: int, arr):
        check(lib().ccm_memcpy_h2d(self._h, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)), self._h)

    def d2h(self, arr, dptr: int):
        check(lib().ccm_memcpy_d2h(self._h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), C.c_size_t(arr.nbytes)), self._h)

    def upload(self, arr) -> int:
        p = self.alloc(max(arr.nbytes, 16))
        if arr.nbytes:
            self.h2d(p, arr)
        return p

    # profiling -------------------------------------------------------------------------------
    def prof_enable(self, kernel_class: int):
        check(lib().ccm_prof_enable(self._h, int(kernel_class)), self._h)

    def prof_reset(self):
        check(lib().ccm_prof_reset(self._h), self._h)

    def prof_read(self, kernel_class: int):
        n = C.c_int64()
        ms = C.c_double()
        check(lib().ccm_prof_read(self._h, int(kernel_class), C.byref(n), C.byref(ms)), self._h)
        return n.value, ms.value

    # RCCL ------------------------------------------------------------------------------------
    def comm_init(self, nranks: int, rank: int, id_bytes: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        check(lib().ccm_comm_init(self._h, int(nranks), int(rank), buf), self._h)


def coresidency_stats(device: int = 0) -> dict:
    """ccm_coresidency_stats: the per-device lease of this process (launches that need the whole device, how many were ordered behind another
    context's launch, how many gave up waiting for their peers, live contexts)."""
    a, b, c, n = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
    check(lib().ccm_coresidency_stats(int(device), C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
    return dict(launches=a.value, chained=b.value, aborted=c.value, contexts=n.value)


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    check(lib().ccm_comm_unique_id(buf))
    return bytes(buf)
