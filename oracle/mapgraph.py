"""Synthetic Map / KeyFrame / MapPoint graphs for the class-API comparisons (TEST INFRASTRUCTURE).

`MapGraph(lib_path, flat)` hands a flat description of a map to oracle/ref_optimizer_driver.cpp, which builds the look-alike object graph
and calls cslam::Optimizer's static methods through the reference's own header.  The driver is linked twice:
  oracle/_ref/liboptimizer_ref.so   the reference's Optimizer.cpp + Converter.cc + g2o, compiled verbatim         (REF_LIB)
  shim/liboptimizer_hip_shim.so     our drop-in shim/Optimizer_hip.cpp on top of libccm_hip.so (needs the MI355X) (SHIM_LIB)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(_HERE, "_ref", "liboptimizer_ref.so")
SHIM_LIB = os.path.join(os.path.dirname(_HERE), "shim", "liboptimizer_hip_shim.so")
SCALE = np.float32(1.2)
N_LEVELS = 8


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def flat_from_ba_problem(prob: dict, n_agents: int = 1) -> dict:
    """A BA problem of ccm_slam_amd.synth (cameras, points, observations) as a map: keyframe poses as 4x4 CV_32F, map points CV_32F,
    one keypoint per observation (octave from the observation's information 1.2^(-2 octave))."""
    from ccm_slam_amd import synth
    n_cam, n_pt, n_edge = int(prob["n_cam"]), int(prob["n_pt"]), int(prob["n_edge"])
    R = synth.R_from_quat(prob["cam_qt"][:, :4])
    T = np.zeros((n_cam, 4, 4), np.float32)
    T[:, :3, :3] = R.astype(np.float32)
    T[:, :3, 3] = prob["cam_qt"][:, 4:].astype(np.float32)
    T[:, 3, 3] = 1
    per = -(-n_cam // n_agents)
    kf_client = (np.arange(n_cam) // per).astype(np.int32)
    kf_id = (np.arange(n_cam) % per).astype(np.int32)
    e_cam, e_pt = prob["e_cam"].astype(np.int32), prob["e_pt"].astype(np.int32)
    order = np.argsort(e_cam, kind="stable")                      # keypoint index = rank of the observation inside its keyframe
    kp_off = np.zeros(n_cam + 1, np.int32)
    np.add.at(kp_off, e_cam + 1, 1)
    kp_off = np.cumsum(kp_off).astype(np.int32)
    obs_kp = np.zeros(n_edge, np.int32)
    obs_kp[order] = (np.arange(n_edge) - kp_off[e_cam[order]]).astype(np.int32)
    octv = np.rint(-np.log(prob["e_info"]) / (2 * np.log(1.2))).astype(np.int32)
    kp_xy = np.zeros((n_edge, 2), np.float32)
    kp_oct = np.zeros(n_edge, np.int32)
    kp_xy[kp_off[e_cam] + obs_kp] = prob["e_obs"].astype(np.float32)
    kp_oct[kp_off[e_cam] + obs_kp] = octv
    return dict(n_kf=n_cam, kf_id=kf_id, kf_client=kf_client, kf_uid=np.arange(n_cam, dtype=np.int32), kf_Tcw=T, kf_bad=np.zeros(n_cam, np.uint8),
                K4=np.asarray(prob["cam_K"][0], np.float32), kp_off=kp_off, kp_xy=kp_xy, kp_oct=kp_oct, n_mp=n_pt,
                mp_id=np.arange(n_pt, dtype=np.int32), mp_client=np.zeros(n_pt, np.int32), mp_uid=(n_cam + np.arange(n_pt)).astype(np.int32),
                mp_pos=prob["pt_xyz"].astype(np.float32), mp_bad=np.zeros(n_pt, np.uint8), obs_mp=e_pt, obs_kf=e_cam, obs_kp=obs_kp, cov_th=15)


class MapGraph:
    def __init__(self, lib_path: str, flat: dict):
        self.lib = C.CDLL(lib_path)
        self.flat = f = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in flat.items()}
        fn = self.lib.mapg_create
        fn.restype = C.c_void_p
        fn.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_int]
        self.h = C.c_void_p(fn(f["n_kf"], _p(f["kf_id"]), _p(f["kf_client"]), _p(f["kf_uid"]), _p(f["kf_Tcw"]), _p(f["kf_bad"]), _p(f["K4"]), _p(f["kp_off"]),
                               _p(f["kp_xy"]), _p(f["kp_oct"]), f["n_mp"], _p(f["mp_id"]), _p(f["mp_client"]), _p(f["mp_uid"]), _p(f["mp_pos"]), _p(f["mp_bad"]),
                               int(f["obs_mp"].size), _p(f["obs_mp"]), _p(f["obs_kf"]), _p(f["obs_kp"]), N_LEVELS, float(SCALE), 0, int(f.get("cov_th", 15))))

    def close(self):
        if self.h:
            self.lib.mapg_destroy.restype = None
            self.lib.mapg_destroy.argtypes = [C.c_void_p]
            self.lib.mapg_destroy(self.h)
            self.h = None

    def lock_points(self, locked, server: bool = False):
        """MapPoint::mbPoseLock of every point (set by the server's corrections in the real system) and the system state the points live in"""
        locked = np.ascontiguousarray(locked, np.uint8)
        self.lib.mapg_lock_points.restype = None
        self.lib.mapg_lock_points(self.h, _p(locked), int(bool(server)))

    def local_ba(self, kf_index: int, client_id: int = 0, server: bool = False, stop_flag=None) -> int:
        fn = self.lib.mapg_local_ba
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        return fn(self.h, kf_index, client_id, int(server), _p(stop_flag))

    def map_fusion_gba(self, client_id: int = 0, iterations: int = 20, loop_kf=(0, 0), robust: bool = True, stop_flag=None) -> int:
        fn = self.lib.mapg_map_fusion_gba
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        return fn(self.h, client_id, iterations, _p(stop_flag), int(loop_kf[0]), int(loop_kf[1]), int(robust))

    def bundle_adjustment_client(self, client_id: int = 0, iterations: int = 20, robust: bool = True) -> int:
        fn = self.lib.mapg_bundle_adjustment_client
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        return fn(self.h, client_id, iterations, None, int(robust))

    def optimize_sim3(self, kf1: int, kf2: int, match_mp, sim3, th2: float = 10.0, fix_scale: bool = False):
        """cslam::Optimizer::OptimizeSim3.  Returns (sim3[8], keep flags per keypoint of kf1, nIn)."""
        match_mp = np.ascontiguousarray(match_mp, np.int32)
        s8 = np.ascontiguousarray(sim3, np.float64).copy()
        keep = np.zeros(max(match_mp.size, 1), np.uint8)
        fn = self.lib.mapg_optimize_sim3
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        nin = fn(self.h, kf1, kf2, _p(match_mp), _p(s8), float(th2), int(fix_scale), _p(keep))
        return s8, keep[:match_mp.size], nin

    def essential_graph(self, loop_kf: int, cur_kf: int, corrected=(), noncorrected=(), connections=(), loop_edges=(), fix_scale=False, map_fusion=False) -> int:
        """OptimizeEssentialGraphLoopClosure (map_fusion=False) / OptimizeEssentialGraphMapFusion.  corrected / noncorrected: [(kf index, sim3[8])];
        connections / loop_edges: [(kf a, kf b)]."""
        def unpack(lst):
            idx = np.array([k for k, _ in lst], np.int32)
            s8 = np.array([s for _, s in lst], np.float64).reshape(-1, 8)
            return idx, np.ascontiguousarray(s8)
        ck, cs = unpack(corrected); nk, ns = unpack(noncorrected)
        ca = np.array([a for a, _ in connections], np.int32); cb = np.array([b for _, b in connections], np.int32)
        la = np.array([a for a, _ in loop_edges], np.int32); lb = np.array([b for _, b in loop_edges], np.int32)
        fn = self.lib.mapg_essential_graph
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                       C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        return fn(self.h, int(map_fusion), loop_kf, cur_kf, ck.size, _p(ck), _p(cs), nk.size, _p(nk), _p(ns), ca.size, _p(ca), _p(cb), la.size, _p(la), _p(lb),
                  int(fix_scale))

    def state(self) -> dict:
        f = self.flat
        nk, nm, no = f["n_kf"], f["n_mp"], int(f["obs_mp"].size)
        out = dict(kf_Tcw=np.zeros((nk, 4, 4), np.float32), kf_gba=np.zeros((nk, 4, 4), np.float32), kf_gba_flag=np.zeros(nk, np.uint8),
                   mp_pos=np.zeros((nm, 3), np.float32), mp_gba=np.zeros((nm, 3), np.float32), mp_gba_flag=np.zeros(nm, np.uint8), mp_bad=np.zeros(nm, np.uint8),
                   mp_normal=np.zeros((nm, 3), np.float32), mp_dmin=np.zeros(nm, np.float32), mp_dmax=np.zeros(nm, np.float32), obs_alive=np.zeros(no, np.uint8))
        fn = self.lib.mapg_get_state
        fn.restype = None
        fn.argtypes = [C.c_void_p] * 12
        fn(self.h, *[_p(out[k]) for k in ("kf_Tcw", "kf_gba", "kf_gba_flag", "mp_pos", "mp_gba", "mp_gba_flag", "mp_bad", "mp_normal", "mp_dmin", "mp_dmax", "obs_alive")])
        return out


def pose_optimization(lib_path: str, Tcw, kp_xy, kp_oct, mp_pos, K4):
    """cslam::Optimizer::PoseOptimizationClient(Frame&) on a Frame built from the arrays.  Returns (Tcw 4x4 f32, outlier flags, nInliers)."""
    lib = C.CDLL(lib_path)
    T = np.ascontiguousarray(Tcw, np.float32).copy()
    kp_xy, kp_oct, mp_pos, K4 = (np.ascontiguousarray(kp_xy, np.float32), np.ascontiguousarray(kp_oct, np.int32), np.ascontiguousarray(mp_pos, np.float32),
                                 np.ascontiguousarray(K4, np.float32))
    n = kp_oct.size
    outl = np.zeros(max(n, 1), np.uint8)
    fn = lib.mapg_pose_optimization
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_void_p]
    nin = fn(_p(T), n, _p(kp_xy), _p(kp_oct), _p(mp_pos), _p(K4), N_LEVELS, float(SCALE), _p(outl))
    return T.reshape(4, 4), outl[:n], nin


def converter(lib_path: str):
    lib = C.CDLL(lib_path)

    def to_se3quat(Tcw):
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        q = np.zeros(7)
        lib.mapg_to_se3quat.restype = None
        lib.mapg_to_se3quat.argtypes = [C.c_void_p, C.c_void_p]
        lib.mapg_to_se3quat(_p(T), _p(q))
        return q

    def to_cvmat(qt):
        q = np.ascontiguousarray(qt, np.float64).reshape(7)
        T = np.zeros(16, np.float32)
        lib.mapg_se3quat_to_cvmat.restype = None
        lib.mapg_se3quat_to_cvmat.argtypes = [C.c_void_p, C.c_void_p]
        lib.mapg_se3quat_to_cvmat(_p(q), _p(T))
        return T.reshape(4, 4)
    return to_se3quat, to_cvmat
