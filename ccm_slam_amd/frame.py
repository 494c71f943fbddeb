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
        check(lib().ccm_frame_window_search(*args, None, None, C.c_int64(0), C.byref(n)), self.ctx.handle)
        idx = np.zeros(max(n.value, 1), np.int32)
        dist = np.zeros(max(n.value, 1), np.uint16)
        check(lib().ccm_frame_window_search(*args, _p(idx), _p(dist), C.c_int64(idx.size), C.byref(n)), self.ctx.handle)
        return off, idx[:n.value], dist[:n.value]

    def close(self):
        if self._h:
            lib().ccm_frame_destroy(self._h)
            self._h = C.c_void_p()
