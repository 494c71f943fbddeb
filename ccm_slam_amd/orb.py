"""ORB extraction entry points of the C ABI — the harness-side mirror of cslam::ORBextractor
(cslam/include/cslam/ORBextractor.h:103-138, cslam/src/ORBextractor.cpp)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, Keypoint, check, hooks, lib

KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32)])


class ORBextractor:
    """Same constructor arguments as the reference: ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)."""

    def __init__(self, ctx: Context, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7):
        self.ctx = ctx
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self._h = C.c_void_p()
        check(lib().ccm_orb_create(ctx.handle, int(nfeatures), C.c_float(scale_factor), int(nlevels), int(ini_th_fast),
                                   int(min_th_fast), C.byref(self._h)), ctx.handle)
        self.cap = int(lib().ccm_orb_max_keypoints(self._h))
        ctx.adopt(self)

    def close(self):
        if self._h:
            lib().ccm_orb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # accessors of ORBextractor.h:114-136
    def _table(self, which):
        out = np.zeros(self.nlevels, np.float32)
        check(lib().ccm_orb_get_table(self._h, which, out.ctypes.data_as(C.c_void_p), self.nlevels), self.ctx.handle)
        return out

    def GetScaleFactors(self):
        return self._table(0)

    def GetInverseScaleFactors(self):
        return self._table(1)

    def GetScaleSigmaSquares(self):
        return self._table(2)

    def GetInverseScaleSigmaSquares(self):
        return self._table(3)

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        check(lib().ccm_orb_features_per_level(self._h, out.ctypes.data_as(C.c_void_p), self.nlevels), self.ctx.handle)
        return out

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        check(lib().ccm_orb_level_size(self._h, w, h, level, C.byref(lw), C.byref(lh)), self.ctx.handle)
        return lw.value, lh.value

    def __call__(self, img: np.ndarray, want_pyramid: bool = False):
        """operator()(image, mask, keypoints, descriptors): returns (keypoints structured array, descriptors Nx32 u8
        [, pyramid list])."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int(0)
        pyr = None
        pyr_ptrs = None
        if want_pyramid:
            pyr = [np.zeros(self.level_size(w, h, l)[::-1], np.uint8) for l in range(self.nlevels)]
            pyr_ptrs = (C.c_void_p * self.nlevels)(*[p.ctypes.data_as(C.c_void_p).value for p in pyr])
        check(lib().ccm_orb_extract(self._h, img.ctypes.data_as(C.c_void_p), w, h, w, kps.ctypes.data_as(C.c_void_p),
                                    desc.ctypes.data_as(C.c_void_p), self.cap, C.byref(n), pyr_ptrs), self.ctx.handle)
        self._wh = (w, h)
        if want_pyramid:
            return kps[:n.value].copy(), desc[:n.value].copy(), pyr
        return kps[:n.value].copy(), desc[:n.value].copy()

    def prepared(self, w: int, h: int):
        """A call object for frames of one size: output arrays and ctypes arguments are made ONCE, run(img) is then exactly one ccm_orb_extract call
        (bench.py times this: the plain __call__ above spends 10 - 15 us per frame in numpy / ctypes allocations and copies that are not the C ABI's)."""
        return _PreparedExtract(self, w, h)

    def debug_level(self, level):
        lw, lh = self.level_size(*self._wh, level)
        score = np.zeros((lh, lw), np.uint8)
        blur = np.zeros((lh, lw), np.uint8)
        check(hooks().ccm_orb_debug_level(self._h, level, score.ctypes.data_as(C.c_void_p), blur.ctypes.data_as(C.c_void_p)), self.ctx.handle)
        return score, blur

    def debug_candidates(self, level):
        n = C.c_int(0)
        check(hooks().ccm_orb_debug_candidates(self._h, level, None, 0, C.byref(n)), self.ctx.handle)
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        check(hooks().ccm_orb_debug_candidates(self._h, level, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), self.ctx.handle)
        return out[:n.value]


class _PreparedExtract:
    def __init__(self, ex: ORBextractor, w: int, h: int):
        self.ex, self.w, self.h = ex, w, h
        self.kps = np.zeros(ex.cap, KP_DTYPE)
        self.desc = np.zeros((ex.cap, 32), np.uint8)
        self.n = C.c_int(0)
        self._fn = lib().ccm_orb_extract
        self._tail = (w, h, w, self.kps.ctypes.data_as(C.c_void_p), self.desc.ctypes.data_as(C.c_void_p), ex.cap, C.byref(self.n), None)

    def run(self, img: np.ndarray) -> int:
        """img: C-contiguous uint8 [h, w].  Results in self.kps[:n] / self.desc[:n]."""
        rc = self._fn(self.ex._h, C.c_void_p(img.ctypes.data), *self._tail)
        if rc:
            check(rc, self.ex.ctx.handle)
        self.ex._wh = (self.w, self.h)
        return self.n.value


class OrbBatchDev:
    """Frames resident in HBM; outputs stay on the device (bench.py)."""

    def __init__(self, ctx: Context, ex: ORBextractor, imgs: np.ndarray):
        self.ctx, self.ex = ctx, ex
        imgs = np.ascontiguousarray(imgs, np.uint8)
        self.n, self.h, self.w = imgs.shape
        self.d_imgs = ctx.upload(imgs)
        self.d_kps = ctx.alloc(self.n * ex.cap * 24)
        self.d_desc = ctx.alloc(self.n * ex.cap * 32)
        self.d_counts = ctx.alloc(self.n * 4)

    def run(self):
        check(lib().ccm_orb_extract_batch_dev(self.ex._h, C.c_void_p(self.d_imgs), self.n, self.w, self.h, C.c_void_p(self.d_kps),
                                              C.c_void_p(self.d_desc), self.ex.cap, C.c_void_p(self.d_counts)), self.ctx.handle)

    def counts(self):
        out = np.zeros(self.n, np.int32)
        self.ctx.d2h(out, self.d_counts)
        return out

    def results(self):
        """[(keypoints, descriptors)] of every frame, downloaded"""
        cnt = self.counts()
        kps = np.zeros(self.n * self.ex.cap, KP_DTYPE)
        desc = np.zeros((self.n * self.ex.cap, 32), np.uint8)
        self.ctx.d2h(kps, self.d_kps)
        self.ctx.d2h(desc, self.d_desc)
        return [(kps[f * self.ex.cap:f * self.ex.cap + cnt[f]].copy(), desc[f * self.ex.cap:f * self.ex.cap + cnt[f]].copy()) for f in range(self.n)]

    def close(self):
        for p in (self.d_imgs, self.d_kps, self.d_desc, self.d_counts):
            self.ctx.free(p)


def distribute_octree(x, y, response, minX, maxX, minY, maxY, N):
    """host-only DistributeOctTree of the product (no GPU needed)"""
    x, y, r = (np.ascontiguousarray(a, np.float32) for a in (x, y, response))
    sel = np.zeros(max(x.size, 1), np.int32)
    n = C.c_int(0)
    check(lib().ccm_orb_distribute_octree(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                          x.size, minX, maxX, minY, maxY, N, sel.ctypes.data_as(C.c_void_p), sel.size, C.byref(n)))
    return sel[:n.value]


def debug_octree_dev(ctx: Context, x, y, response, W, H, N):
    """Test hook: the device octree kernel on one level's candidates; returns (selected indices in output order, overflow flag)."""
    x, y, r = (np.ascontiguousarray(a, np.int32) for a in (x, y, response))
    sel = np.zeros(max(4 * N + 16, 64), np.int32)   # (the first pass turns up to 15 roots into 60 nodes whatever N is)
    n = C.c_int(0); over = C.c_int(0)
    check(hooks().ccm_orb_debug_octree_dev(ctx.handle, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                         int(x.size), int(W), int(H), int(N), sel.ctypes.data_as(C.c_void_p), int(sel.size), C.byref(n), C.byref(over)), ctx.handle)
    return sel[:n.value], over.value


def smoke_check(ctx: Context, oracle):
    from . import synth
    img = synth.gen_image(1234, 0)
    ex = ORBextractor(ctx, 500)
    kps, desc = ex(img)
    o = oracle.OrbOracle(500)
    okps, odesc = o.extract(img)
    assert len(kps) == len(okps) and np.array_equal(desc, odesc), "ORB descriptors differ from the oracle"
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(kps[f], okps[f]), f"ORB keypoint field {f} differs from the oracle"
    ex.close()
