"""The drop-in translation units EXECUTED on the reference's REAL map classes (round 5).  shim/liboptimizer_hip_shim_real.so links shim/Optimizer_hip.cpp,
ORBmatcher_hip.cpp and ORBextractor_hip.cpp — compiled against the reference's real headers — with the reference's own cslam/src/KeyFrame.cpp, MapPoint.cpp,
Map.cpp, Frame.cpp and Converter.cc compiled as they are (shim/Makefile: REAL_REF_OBJS; look-alikes remain only for what the image does not have: ROS / PCL
headers, OpenCV, Eigen, Boost), and the harness of the look-alike tests built with -DCCM_REAL_CLASSES: the object graph consists of cslam::KeyFrame /
cslam::MapPoint / cslam::Map objects placed with the reference's save / load constructors (KeyFrame.cpp:33-49, MapPoint.cpp:32-46), so every accessor the
shim calls — GetObservations(), GetMapPointMatches(), SetPose(), SetWorldPos(), UpdateNormalAndDepth(), EraseObservation(), EraseMapPointMatch(),
Map::GetAllKeyFrames() / LockMapUpdate() ... — is the reference's own code with its own mutexes (MapPoint.cpp:340-341, :779-823, KeyFrame.cpp:286-327).

Every call is made three times on the same flat map: the reference's Optimizer.cpp + g2o on the look-alike graph (REF), the shim on the look-alike graph
(SHIM) and the shim on the real classes (REAL).  REAL must equal SHIM bit for bit (same flat problem -> same device result -> same write-back, whatever
classes carry it) and REF within the bars of tests/test_shim_gpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

from ccm_slam_amd import synth
from oracle import mapgraph as mg
from tests.test_ref_optimizer import local_window, ulps32

REAL_LIB = mg.SHIM_LIB.replace(".so", "_real.so")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(mg.REF_LIB) and os.path.exists(mg.SHIM_LIB) and os.path.exists(REAL_LIB)), reason="oracle/_ref or shim libraries not built")]
_KEYS = ("kf_Tcw", "kf_gba", "kf_gba_flag", "mp_pos", "mp_gba", "mp_gba_flag", "mp_bad", "mp_normal", "mp_dmin", "mp_dmax", "obs_alive")


def _three(flat, call, prepare=None):
    out = []
    for lib in (mg.REF_LIB, mg.SHIM_LIB, REAL_LIB):
        g = mg.MapGraph(lib, flat)
        if prepare:
            prepare(g)
        assert call(g) == 0, lib
        out.append(g.state())
        g.close()
    return out


def _identical(a, b, what, flat=None):
    for k in _KEYS:
        if k == "obs_alive" and flat is not None:
            # the real MapPoint::SetBadFlag (MapPoint.cpp) also empties the observation map of a point that turned bad; the look-alike keeps it: compare the
            # observations of the points that are still alive, and the bad flags themselves
            live = a["mp_bad"][flat["obs_mp"]] == 0
            assert np.array_equal(a[k][live], b[k][live]), f"{what}: {k} differs between the look-alike graph and the real classes"
            continue
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs between the look-alike graph and the real classes"


def test_the_library_is_made_of_the_real_classes():
    lib = C.CDLL(REAL_LIB)
    assert lib.mapg_uses_real_classes() == 1
    import subprocess
    sym = subprocess.run(["nm", "-D", "-C", "--defined-only", REAL_LIB], capture_output=True, text=True).stdout
    for s in ("cslam::MapPoint::UpdateNormalAndDepth()", "cslam::MapPoint::GetObservations()", "cslam::KeyFrame::SetPose(cv::Mat const&, bool, bool)",
              "cslam::Map::GetAllMapPoints()", "cslam::Optimizer::MapFusionGBA(", "cslam::ORBmatcher::DescriptorDistance("):
        assert s in sym, s


def test_local_bundle_adjustment_on_the_real_classes():
    prob = synth.make_ba_problem(n_agents=1, kfs_per_agent=30, n_points=2500, seed=12, n_fixed=1, mean_track=7)
    flat = mg.flat_from_ba_problem(prob)
    for cur in (17, 3):
        local_kf, local_mp, _, _ = local_window(flat, cur, 15)
        r, s, x = _three(flat, lambda g: g.local_ba(cur, client_id=0))
        _identical(s, x, f"local BA around keyframe {cur}", flat)
        live = (r["mp_bad"][flat["obs_mp"]] == 0) & (x["mp_bad"][flat["obs_mp"]] == 0)      # (the real SetBadFlag empties a bad point's observation map)
        assert 0 < (x["obs_alive"] == 0).sum() and (r["obs_alive"][live] != x["obs_alive"][live]).sum() <= 2 and (r["mp_bad"] != x["mp_bad"]).sum() <= 2
        assert ulps32(r["kf_Tcw"], x["kf_Tcw"]).max() <= 256
        assert np.array_equal(x["kf_Tcw"][~local_kf], flat["kf_Tcw"][~local_kf]) and np.array_equal(x["mp_pos"][~local_mp], flat["mp_pos"][~local_mp])
        keep = (r["mp_bad"] == 0) & (x["mp_bad"] == 0)
        assert ulps32(r["mp_pos"][keep], x["mp_pos"][keep]).max() <= 1024
        assert np.abs(r["mp_normal"][keep] - x["mp_normal"][keep]).max() < 1e-4


def test_local_ba_with_pose_locked_points_on_the_real_classes():
    """MapPoint.cpp:340-341 (the REAL lines now): a locked point of a CLIENT ignores SetWorldPos"""
    lflat = mg.flat_from_ba_problem(synth.make_ba_config("lba_c2"))
    locked = (np.arange(lflat["n_mp"]) % 3 == 0).astype(np.uint8)
    r, s, x = _three(lflat, lambda g: g.local_ba(15, client_id=0), prepare=lambda g: g.lock_points(locked, server=False))
    _identical(s, x, "local BA with locked points", lflat)
    lk = locked.astype(bool)
    assert np.array_equal(x["mp_pos"][lk], lflat["mp_pos"][lk]) and (x["mp_pos"][~lk] != lflat["mp_pos"][~lk]).any(axis=1).sum() > 500
    keep = (r["mp_bad"] == 0) & (x["mp_bad"] == 0)
    assert ulps32(r["mp_pos"][keep], x["mp_pos"][keep]).max() <= 1024


def test_map_fusion_gba_and_bundle_adjustment_client_on_the_real_classes():
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=15, n_points=1500, seed=4, n_fixed=1)
    flat = mg.flat_from_ba_problem(prob, n_agents=2)
    for name, call in (("MapFusionGBA", lambda g: g.map_fusion_gba(0, 4)), ("BundleAdjustmentClient", lambda g: g.bundle_adjustment_client(0, 4)),
                       ("MapFusionGBA without the robust kernel", lambda g: g.map_fusion_gba(0, 4, robust=False))):
        r, s, x = _three(flat, call)
        _identical(s, x, name)
        assert ulps32(r["kf_Tcw"], x["kf_Tcw"]).max() <= 64 and ulps32(r["mp_pos"], x["mp_pos"]).max() <= 256
    r, s, x = _three(flat, lambda g: g.map_fusion_gba(0, 3, loop_kf=(3, 0)))          # results land in mTcwGBA / mPosGBA (Optimizer.cpp:812-818, 841-852)
    _identical(s, x, "MapFusionGBA with a loop keyframe")
    assert np.array_equal(x["kf_Tcw"], flat["kf_Tcw"]) and np.array_equal(x["mp_pos"], flat["mp_pos"]) and x["kf_gba_flag"].sum() > 0
    assert ulps32(r["kf_gba"], x["kf_gba"]).max() <= 64 and ulps32(r["mp_gba"], x["mp_gba"]).max() <= 256


def test_map_fusion_gba_of_a_three_agent_map_on_the_real_classes_with_the_threaded_walk():
    """25 000 map points: above the shim's threshold for the multi-threaded graph walk / write-back, so the reference's per-object mutexes are taken from
    several host threads at once (GetObservations under mMutexFeatures, SetWorldPos under mMutexPos, UpdateNormalAndDepth under both)"""
    prob = synth.make_ba_problem(n_agents=3, kfs_per_agent=60, n_points=25000, seed=31)
    flat = mg.flat_from_ba_problem(prob, n_agents=3)
    out = []
    for lib in (mg.SHIM_LIB, REAL_LIB):
        g = mg.MapGraph(lib, flat)
        assert g.map_fusion_gba(0, 20) == 0
        out.append(g.state())
        g.close()
    _identical(out[0], out[1], "MapFusionGBA, 3 agents")
    assert np.abs(out[1]["kf_Tcw"] - flat["kf_Tcw"]).max() > 1e-4 and (out[1]["mp_dmax"] > out[1]["mp_dmin"]).sum() > 20000


def test_optimize_sim3_and_the_essential_graph_on_the_real_classes():
    from tests.test_shim_gpu import _loop_map, _sim3_of_pose
    prob, flat = _loop_map()
    kf1, kf2 = 20, 22
    seen2 = set(flat["obs_mp"][flat["obs_kf"] == kf2].tolist())
    match = np.full(int(flat["kp_off"][kf1 + 1] - flat["kp_off"][kf1]), -1, np.int32)
    for o in np.where(flat["obs_kf"] == kf1)[0]:
        if int(flat["obs_mp"][o]) in seen2:
            match[flat["obs_kp"][o]] = flat["obs_mp"][o]
    T12 = flat["kf_Tcw"][kf1].astype(np.float64) @ np.linalg.inv(flat["kf_Tcw"][kf2].astype(np.float64))
    T12[:3, 3] += [0.02, -0.01, 0.015]
    s0 = _sim3_of_pose(T12.astype(np.float32), 1.03)
    for fix in (False, True):
        res = []
        for lib in (mg.REF_LIB, mg.SHIM_LIB, REAL_LIB):
            g = mg.MapGraph(lib, flat)
            res.append(g.optimize_sim3(kf1, kf2, match, s0, th2=10.0, fix_scale=fix))
            g.close()
        (sr, kr, nr), (ss, ks, ns), (sx, kx, nx) = res
        assert nx == ns == nr and np.array_equal(kx, ks) and np.array_equal(sx, ss)
        assert np.abs(sr - sx).max() < 1e-5
    T = flat["kf_Tcw"]
    corrected = [(k, _sim3_of_pose(T[k]) + np.r_[np.zeros(4), [0.01 * (k - 54), -0.004 * (k - 54), 0.002], [0.01]]) for k in range(55, 60)]
    noncorrected = [(k, _sim3_of_pose(T[k])) for k in range(55, 60)]
    for map_fusion in (False, True):
        r, s, x = _three(flat, lambda g: g.essential_graph(0, 59, corrected if not map_fusion else (), noncorrected if not map_fusion else (),
                                                           [(59, 0), (59, 1), (58, 0), (57, 2)], [(30, 10)], fix_scale=False, map_fusion=map_fusion))
        # (not bit for bit here: the real Map hands its keyframes out of a std::map<idpair, kfptr> and the real KeyFrame its neighbours out of its own containers, so the
        # pose-graph edges reach the device in another order than from the look-alike's vectors — another summation order in the 7 x 7 blocks, the same optimum)
        assert np.abs(s["kf_Tcw"] - x["kf_Tcw"]).max() < 2e-5 and np.abs(s["mp_pos"] - x["mp_pos"]).max() < 1e-4, (np.abs(s["kf_Tcw"] - x["kf_Tcw"]).max(), np.abs(s["mp_pos"] - x["mp_pos"]).max())
        assert np.array_equal(s["mp_bad"], x["mp_bad"]) and np.array_equal(s["kf_Tcw"][0], x["kf_Tcw"][0])
        assert np.abs(r["kf_Tcw"] - x["kf_Tcw"]).max() < 2e-4 and np.abs(r["mp_pos"] - x["mp_pos"]).max() < 1e-3


def test_pose_optimization_client_on_the_real_frame():
    """cslam::Optimizer::PoseOptimizationClient(Frame&) (Optimizer.cpp:215-347) on the reference's REAL cslam::Frame.  The class has no default constructor: the harness
    builds it with the tracking constructor (Frame.cpp:56-101) from a synthetic image — the ORB extraction inside it is this library's drop-in ORBextractor on the MI355X —,
    then gives it the test's observation set (keypoints, real cslam::MapPoint objects, pose through the real Frame::SetPose); oracle/ref_optimizer_driver.cpp.  The same
    call on the look-alike Frame (SHIM) must give the same pose bit for bit, the reference's own Optimizer.cpp + g2o (REF) within the bar of tests/test_shim_gpu.py."""
    for n, seed, of in ((300, 0, 0.1), (1000, 7, 0.2), (40, 3, 0.3)):
        pp = synth.make_pose_problem(n, seed, of)
        R = synth.R_from_quat(pp["cam_qt"][None, :4])[0]
        T = np.eye(4, dtype=np.float32); T[:3, :3] = R.astype(np.float32); T[:3, 3] = pp["cam_qt"][4:].astype(np.float32)
        octv = np.rint(-np.log(pp["info"]) / (2 * np.log(1.2))).astype(np.int32)
        args = (T, pp["obs"].astype(np.float32), octv, pp["Xw"].astype(np.float32), np.asarray(pp["K"], np.float32))
        Tr, outr, nr = mg.pose_optimization(mg.REF_LIB, *args)
        Ts, outs, ns = mg.pose_optimization(mg.SHIM_LIB, *args)
        Tq, outq, nq = mg.pose_optimization(REAL_LIB, *args)
        assert nq >= 0, f"the real Frame could not be built (code {nq})"
        assert nq == ns == nr
        assert np.array_equal(outq, outs) and np.array_equal(outq, outr)
        assert np.array_equal(Tq, Ts)
        assert ulps32(Tr, Tq).max() <= 2
