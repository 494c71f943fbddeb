"""Per-frame glue on the device: Frame::UndistortKeyPoints / ComputeImageBounds /
AssignFeaturesToGrid (cslam/src/Frame.cpp:103-118, 284-347) and batched Frame::GetFeaturesInArea window searches
(:200-253) with the Hamming distances of every candidate — ccm_frame_* of the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, check, lib

GRID_COLS, GRID_ROWS = 75, 48


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class FrameGrid:
    def __init__(self, ctx: Context, K, dist, width: int, height: int):
        self.ctx = ctx
        K = np.ascontiguousarray(K, np.float32)
        dist = np.ascontiguousarray(dist, np.float32)
        self._h = C.c_void_p()
        check(lib().ccm_frame_create(ctx.handle, _p(K), _p(dist) if dist.size else None, int(dist.size), int(width), int(height),
                                     C.byref(self._h)), ctx.handle)
        self.N = 0
        ctx.adopt(self)

    @property
    def bounds(self):
        b = np.zeros(4, np.float32)
        check(lib().ccm_frame_bounds(self._h, _p(b)), self.ctx.handle)
        return b

    def set_keypoints(self, kps: np.ndarray, desc: np.ndarray):
        """kps: structured array with the ccm_keypoint layout (x, y, size, angle, response, octave)"""
        kps = np.ascontiguousarray(kps)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        assert kps.dtype.itemsize == 24 and kps.shape[0] == desc.shape[0]
        self.N = int(kps.shape[0])
        self._keep = (kps, desc)
        check(lib().ccm_frame_set_keypoints(self._h, _p(kps), _p(desc), self.N), self.ctx.handle)

    def get(self):
        xy = np.zeros((max(self.N, 1), 2), np.float32)
        off = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
        idx = np.zeros(max(self.N, 1), np.int32)
        check(lib().ccm_frame_get(self._h, _p(xy), _p(off), _p(idx)), self.ctx.handle)
        return xy[:self.N], off, idx[:off[-1]]

    def window_search(self, u, v, r, min_level, max_level, qdesc):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        u, v, r, min_level, max_level = f32(u), f32(v), f32(r), i32(min_level), i32(max_level)
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        Q = int(u.size)
        off = np.zeros(Q + 1, np.int32)
        n = C.c_int64(0)
        args = (self._h, Q, _p(u), _p(v), _p(r), _p(min_level), _p(max_level), _p(qdesc), _p(off))
        cap = 24 * Q + 1024                                    # typical lists are short; one retry with the exact size otherwise
        idx = np.zeros(cap, np.int32)
        dist = np.zeros(cap, np.uint16)
        rc = lib().ccm_frame_window_search(*args, _p(idx), _p(dist), C.c_int64(cap), C.byref(n))
        if rc != 0 and n.value > cap:
            idx = np.zeros(n.value, np.int32)
            dist = np.zeros(n.value, np.uint16)
            rc = lib().ccm_frame_window_search(*args, _p(idx), _p(dist), C.c_int64(idx.size), C.byref(n))
        check(rc, self.ctx.handle)
        return off, idx[:n.value], dist[:n.value]

    def prepared_set_keypoints(self, kps: np.ndarray, desc: np.ndarray):
        """call object: run() = one ccm_frame_set_keypoints call on the arrays given here (arguments converted once)"""
        kps = np.ascontiguousarray(kps)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        assert kps.dtype.itemsize == 24 and kps.shape[0] == desc.shape[0]
        return _Prepared(lib().ccm_frame_set_keypoints, (self._h, _p(kps), _p(desc), int(kps.shape[0])), self.ctx, (kps, desc), owner=self, n=int(kps.shape[0]))

    def prepared_window_search(self, u, v, r, min_level, max_level, qdesc, cap=None):
        """call object: run() = one ccm_frame_window_search call (outputs preallocated for `cap` candidates, default as window_search); .result() -> (off, idx, dist)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        u, v, r, min_level, max_level = f32(u), f32(v), f32(r), i32(min_level), i32(max_level)
        qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        Q = int(u.size)
        cap = int(cap) if cap is not None else 24 * Q + 1024
        off = np.zeros(Q + 1, np.int32); idx = np.zeros(cap, np.int32); dist = np.zeros(cap, np.uint16)
        n = C.c_int64(0)
        args = (self._h, Q, _p(u), _p(v), _p(r), _p(min_level), _p(max_level), _p(qdesc), _p(off), _p(idx), _p(dist), C.c_int64(cap), C.byref(n))
        c = _Prepared(lib().ccm_frame_window_search, args, self.ctx, (u, v, r, min_level, max_level, qdesc, off, idx, dist, n))
        c.result = lambda: (off, idx[:n.value], dist[:n.value])
        return c

    def close(self):
        if self._h:
            lib().ccm_frame_destroy(self._h)
            self._h = C.c_void_p()


class _Prepared:
    """One C-ABI call with its ctypes arguments built once (bench.py times run(); the plain wrappers spend 10 - 30 us per call in numpy / ctypes conversions)."""

    def __init__(self, fn, args, ctx, keep, owner=None, n=None):
        self._fn, self._args, self._ctx, self._keep, self._owner, self._n = fn, args, ctx, keep, owner, n

    def run(self):
        rc = self._fn(*self._args)
        if rc:
            check(rc, self._ctx.handle)
        if self._owner is not None:
            self._owner.N = self._n
            self._owner._keep = self._keep


class FrustumFrame(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("minX", C.c_float), ("maxX", C.c_float), ("minY", C.c_float), ("maxY", C.c_float),
                ("logScaleFactor", C.c_float), ("nScaleLevels", C.c_int32)]


def update_normal_and_depth(ctx: Context, pos, obs_off, obs_kf, kf_center, ref_kf, ref_level, scale_factors, normal, min_dist, max_dist):
    """MapPoint::UpdateNormalAndDepth for a batch of map points (ccm_update_normal_and_depth).  Returns (normal, min_dist, max_dist)."""
    f32, i32 = (lambda a: np.ascontiguousarray(a, np.float32)), (lambda a: np.ascontiguousarray(a, np.int32))
    pos, kf_center, sf = f32(pos).reshape(-1, 3), f32(kf_center).reshape(-1, 3), f32(scale_factors)
    obs_off, obs_kf, ref_kf, ref_level = i32(obs_off), i32(obs_kf), i32(ref_kf), i32(ref_level)
    nrm, dmin, dmax = f32(normal).copy().reshape(-1, 3), f32(min_dist).copy(), f32(max_dist).copy()
    check(lib().ccm_update_normal_and_depth(ctx.handle, int(pos.shape[0]), _p(pos), _p(obs_off), _p(obs_kf), int(kf_center.shape[0]), _p(kf_center),
                                            _p(ref_kf), _p(ref_level), _p(sf), int(sf.size), _p(nrm), _p(dmin), _p(dmax)), ctx.handle)
    return nrm, dmin, dmax


def prepared_is_in_frustum(ctx: Context, frame24, n_levels: int, P, normal, dmin, dmax, cos_limit: float = 0.5):
    """call object: run() = one ccm_frame_frustum call; .result() -> (in_view, u, v, level, cos)"""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    fr = f32(frame24)
    st = FrustumFrame()
    C.memmove(C.byref(st), fr.ctypes.data, 24 * 4)
    st.nScaleLevels = int(n_levels)
    P, normal, dmin, dmax = f32(P).reshape(-1, 3), f32(normal).reshape(-1, 3), f32(dmin), f32(dmax)
    n = P.shape[0]
    inv = np.zeros(max(n, 1), np.uint8); u = np.zeros(max(n, 1), np.float32); v = np.zeros(max(n, 1), np.float32)
    lvl = np.zeros(max(n, 1), np.int32); cs = np.zeros(max(n, 1), np.float32)
    args = (ctx.handle, C.byref(st), n, _p(P), _p(normal), _p(dmin), _p(dmax), C.c_float(cos_limit), _p(inv), _p(u), _p(v), _p(lvl), _p(cs))
    c = _Prepared(lib().ccm_frame_frustum, args, ctx, (st, P, normal, dmin, dmax, inv, u, v, lvl, cs))
    c.result = lambda: (inv[:n], u[:n], v[:n], lvl[:n], cs[:n])
    return c


def is_in_frustum(ctx: Context, frame24, n_levels: int, P, normal, dmin, dmax, cos_limit: float = 0.5):
    """Frame::isInFrustum for a batch of map points (ccm_frame_frustum); frame24 as in oracle.is_in_frustum."""
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    fr = f32(frame24)
    st = FrustumFrame()
    C.memmove(C.byref(st), fr.ctypes.data, 24 * 4)
    st.nScaleLevels = int(n_levels)
    P, normal, dmin, dmax = f32(P).reshape(-1, 3), f32(normal).reshape(-1, 3), f32(dmin), f32(dmax)
    n = P.shape[0]
    inv = np.zeros(max(n, 1), np.uint8); u = np.zeros(max(n, 1), np.float32); v = np.zeros(max(n, 1), np.float32)
    lvl = np.zeros(max(n, 1), np.int32); cs = np.zeros(max(n, 1), np.float32)
    check(lib().ccm_frame_frustum(ctx.handle, C.byref(st), n, _p(P), _p(normal), _p(dmin), _p(dmax), C.c_float(cos_limit), _p(inv), _p(u), _p(v),
                                  _p(lvl), _p(cs)), ctx.handle)
    return inv[:n], u[:n], v[:n], lvl[:n], cs[:n]
