"""OUR drop-in translation units (shim/Optimizer_hip.cpp, shim/ORBextractor_hip.cpp -> libccm_hip.so on the MI355X) against the REFERENCE'S OWN
translation units (cslam/src/Optimizer.cpp, ORBextractor.cpp compiled verbatim, oracle/_ref), both driven through the reference's class API
by the same harness on identical synthetic maps / images: what a maintainer gets when the .cpp files are swapped in cslam/CMakeLists.txt.
Covers the graph walks of the shim, the O0 boundary, the write-back side effects and the erase set end to end."""
import os

import numpy as np
import pytest

from ccm_slam_amd import synth
from oracle import mapgraph as mg
from oracle import ref
from tests.test_ref_optimizer import local_window, ulps32

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(mg.REF_LIB) and os.path.exists(mg.SHIM_LIB)), reason="oracle/_ref or shim library not built")]


def _both(flat, call):
    out = []
    for lib in (mg.REF_LIB, mg.SHIM_LIB):
        g = mg.MapGraph(lib, flat)
        assert call(g) == 0
        out.append(g.state())
        g.close()
    return out


def test_local_bundle_adjustment_client_shim_equals_reference():
    prob = synth.make_ba_problem(n_agents=1, kfs_per_agent=30, n_points=2500, seed=12, n_fixed=1, mean_track=7)
    flat = mg.flat_from_ba_problem(prob)
    for cur in (17, 3):
        local_kf, local_mp, edge_in, fixed_kf = local_window(flat, cur, 15)
        r, s = _both(flat, lambda g: g.local_ba(cur, client_id=0))
        assert (r["obs_alive"] != s["obs_alive"]).sum() <= 2                       # the erase set (chi2 within rounding of 5.991 may flip)
        assert 0 < (r["obs_alive"] == 0).sum()
        assert ulps32(r["kf_Tcw"], s["kf_Tcw"]).max() <= 256
        assert np.array_equal(s["kf_Tcw"][~local_kf], flat["kf_Tcw"][~local_kf])   # only the local keyframes are written
        keep = (r["mp_bad"] == 0) & (s["mp_bad"] == 0)
        assert ulps32(r["mp_pos"][keep], s["mp_pos"][keep]).max() <= 1024
        assert np.array_equal(s["mp_pos"][~local_mp], flat["mp_pos"][~local_mp])
        assert np.abs(r["mp_normal"][keep] - s["mp_normal"][keep]).max() < 1e-4 and np.abs(r["mp_dmax"][keep] / s["mp_dmax"][keep] - 1).max() < 1e-4
        assert (r["mp_bad"] != s["mp_bad"]).sum() <= 2


def test_local_ba_respects_a_raised_stop_flag_like_the_reference():
    prob = synth.make_ba_problem(n_agents=1, kfs_per_agent=12, n_points=800, seed=2)
    flat = mg.flat_from_ba_problem(prob)
    flag = np.ones(1, np.uint8)
    r, s = _both(flat, lambda g: g.local_ba(6, client_id=0, stop_flag=flag))
    for st in (r, s):                                                              # early return at Optimizer.cpp:532-534: nothing is written
        assert np.array_equal(st["kf_Tcw"], flat["kf_Tcw"]) and np.array_equal(st["mp_pos"], flat["mp_pos"]) and st["obs_alive"].all()


def test_map_fusion_gba_and_bundle_adjustment_client_shim_equal_reference():
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=15, n_points=1500, seed=4, n_fixed=1)
    flat = mg.flat_from_ba_problem(prob, n_agents=2)
    for call in (lambda g: g.map_fusion_gba(0, 4), lambda g: g.bundle_adjustment_client(0, 4)):
        r, s = _both(flat, call)
        assert ulps32(r["kf_Tcw"], s["kf_Tcw"]).max() <= 64
        assert ulps32(r["mp_pos"], s["mp_pos"]).max() <= 256
        assert np.array_equal(r["mp_normal"] != 0, s["mp_normal"] != 0)
    # loop keyframe given: results land in mTcwGBA / mPosGBA, the map itself is untouched (Optimizer.cpp:812-818, 841-852)
    r, s = _both(flat, lambda g: g.map_fusion_gba(0, 3, loop_kf=(3, 0)))
    assert np.array_equal(r["kf_gba_flag"], s["kf_gba_flag"]) and np.array_equal(r["mp_gba_flag"], s["mp_gba_flag"])
    assert np.array_equal(s["kf_Tcw"], flat["kf_Tcw"]) and np.array_equal(s["mp_pos"], flat["mp_pos"])
    assert ulps32(r["kf_gba"], s["kf_gba"]).max() <= 64 and ulps32(r["mp_gba"], s["mp_gba"]).max() <= 256


def test_pose_optimization_client_shim_equals_reference():
    for n, seed, of in ((300, 0, 0.1), (40, 3, 0.3), (1000, 7, 0.2), (2, 1, 0.0)):
        pp = synth.make_pose_problem(n, seed, of)
        R = synth.R_from_quat(pp["cam_qt"][None, :4])[0]
        T = np.eye(4, dtype=np.float32); T[:3, :3] = R.astype(np.float32); T[:3, 3] = pp["cam_qt"][4:].astype(np.float32)
        octv = np.rint(-np.log(pp["info"]) / (2 * np.log(1.2))).astype(np.int32)
        args = (T, pp["obs"].astype(np.float32), octv, pp["Xw"].astype(np.float32), np.asarray(pp["K"], np.float32))
        Tr, outr, nr = mg.pose_optimization(mg.REF_LIB, *args)
        Ts, outs, ns = mg.pose_optimization(mg.SHIM_LIB, *args)
        assert nr == ns and np.array_equal(outr, outs)
        assert ulps32(Tr, Ts).max() <= 2


def test_orb_extractor_shim_equals_the_reference_translation_unit():
    """cslam::ORBextractor::operator() of shim/ORBextractor_hip.cpp (MI355X) against the reference's ORBextractor.cpp (run with the monotonic
    allocator that makes its pointer tie-break 'creation order'): keypoints, descriptors, mvImagePyramid and the accessor tables, bit for bit."""
    shim = os.path.join(os.path.dirname(mg.SHIM_LIB), "liborb_hip_shim.so")
    if not os.path.exists(shim):
        pytest.skip("shim/liborb_hip_shim.so not built")
    for seed, t, nf in ((1000, 0, 1000), (1002, 4, 2000)):
        img = synth.gen_image(seed, t)
        r = ref.RefOrb(nf)
        s = ref.RefOrb(nf, lib_path=shim)
        rk, rd = r.extract_cli(img)
        sk, sd = s.extract(img)
        assert len(rk) == len(sk)
        for f in rk.dtype.names:
            assert np.array_equal(rk[f], sk[f]), f
        assert np.array_equal(rd, sd)
        r.extract(img)
        for l in range(8):
            assert np.array_equal(r.level(l), s.level(l)), l
        for a, b in zip(r.tables(), s.tables()):
            assert np.array_equal(a, b)


def _loop_map():
    """one agent driving a closed loop; the keyframes near the end of the loop see landmarks of the start (loop closure candidates)"""
    prob = synth.make_ba_problem(n_agents=1, kfs_per_agent=60, n_points=4000, seed=33, n_fixed=1)
    return prob, mg.flat_from_ba_problem(prob)


def _sim3_of_pose(T, s=1.0):
    from oracle import to_se3quat
    q = to_se3quat(T)[0]
    return np.concatenate([q, [s]])


def test_optimize_sim3_shim_equals_reference():
    prob, flat = _loop_map()
    kf1, kf2 = 20, 22
    seen2 = set(flat["obs_mp"][flat["obs_kf"] == kf2].tolist())
    n1 = int(flat["kp_off"][kf1 + 1] - flat["kp_off"][kf1])
    match = np.full(n1, -1, np.int32)
    sel = np.where(flat["obs_kf"] == kf1)[0]
    for o in sel:
        if int(flat["obs_mp"][o]) in seen2:
            match[flat["obs_kp"][o]] = flat["obs_mp"][o]
    assert (match >= 0).sum() >= 40
    T1, T2 = flat["kf_Tcw"][kf1].astype(np.float64), flat["kf_Tcw"][kf2].astype(np.float64)
    T12 = T1 @ np.linalg.inv(T2)
    T12[:3, 3] += [0.02, -0.01, 0.015]                                   # a Sim3Solver-grade initial guess
    s0 = _sim3_of_pose(T12.astype(np.float32), 1.03)
    for fix in (False, True):
        out = []
        for lib in (mg.REF_LIB, mg.SHIM_LIB):
            g = mg.MapGraph(lib, flat)
            out.append(g.optimize_sim3(kf1, kf2, match, s0, th2=10.0, fix_scale=fix))
            g.close()
        (sr, kr, nr), (ss, ks, ns) = out
        assert nr == ns and nr >= 10 and np.array_equal(kr, ks)
        assert np.abs(sr - ss).max() < 1e-5
        if fix:
            assert ss[7] == s0[7]


@pytest.mark.parametrize("map_fusion", [False, True])
def test_essential_graph_optimisation_shim_equals_reference(map_fusion):
    """OptimizeEssentialGraphLoopClosure / MapFusion: the shim builds the same edge set (loop connections, spanning tree, loop edges, covisibility
    above the threshold), the device solves the pose graph, the write-back moves keyframes and map points like the reference's."""
    prob, flat = _loop_map()
    T = flat["kf_Tcw"]
    cur, loop = 59, 0
    corrected = [(k, _sim3_of_pose(T[k]) + np.r_[np.zeros(4), [0.01 * (k - 54), -0.004 * (k - 54), 0.002], [0.0]] * np.r_[np.ones(7), [0]] + np.r_[np.zeros(7), [0.01]])
                 for k in range(55, 60)]
    noncorrected = [(k, _sim3_of_pose(T[k])) for k in range(55, 60)]
    connections = [(59, 0), (59, 1), (58, 0), (57, 2)]
    loop_edges = [(30, 10)]
    out = []
    for lib in (mg.REF_LIB, mg.SHIM_LIB):
        g = mg.MapGraph(lib, flat)
        assert g.essential_graph(loop, cur, corrected if not map_fusion else (), noncorrected if not map_fusion else (), connections, loop_edges,
                                 fix_scale=False, map_fusion=map_fusion) == 0
        out.append(g.state())
        g.close()
    r, s = out
    assert np.abs(r["kf_Tcw"] - flat["kf_Tcw"]).max() > 1e-4 or map_fusion          # the loop-closure variant really moved the keyframes
    assert np.abs(r["kf_Tcw"] - s["kf_Tcw"]).max() < 2e-4, np.abs(r["kf_Tcw"] - s["kf_Tcw"]).max()
    assert np.abs(r["mp_pos"] - s["mp_pos"]).max() < 1e-3, np.abs(r["mp_pos"] - s["mp_pos"]).max()
    assert np.array_equal(r["kf_Tcw"][loop], s["kf_Tcw"][loop])


def test_map_fusion_gba_through_the_shim_at_baseline_scale_is_lossless():
    """BASELINE config 4 (2000 keyframes, 150 000 map points, 0.95 M observations) as a Map / KeyFrame / MapPoint graph, Optimizer::MapFusionGBA(…, 20)
    through shim/Optimizer_hip.cpp: what the call leaves in the map must be EXACTLY the f32 image (Converter::toCvMat) of what the flat C ABI computes
    on the flat problem that the reference's accessors describe (f32 poses through Converter::toSE3Quat, f32 points widened, mvInvLevelSigma2 of the
    observation's octave) — i.e. the threaded graph walk, the flatten and the threaded write-back lose nothing and reorder nothing that matters.
    The reference's own Optimizer.cpp needs ~25 s per LM iteration on this map (bench.py's cpu_baseline), so it is compared at 30 keyframes above."""
    import ctypes as C
    from ccm_slam_amd import optimizer
    from ccm_slam_amd._lib import Context
    prob = synth.make_ba_config("gba_c4")
    flat = mg.flat_from_ba_problem(prob, n_agents=4)
    g = mg.MapGraph(mg.SHIM_LIB, flat)
    assert g.map_fusion_gba(0, 20) == 0
    ph = (C.c_double * 10)()
    g.lib.ccm_shim_last_phases(ph)
    st = g.state()
    g.close()
    assert 0 < ph[2] and 0 < ph[3] and ph[7] >= ph[2] + ph[3]                     # create and run were clocked inside the call
    to_q, to_T = mg.converter(mg.SHIM_LIB)
    n_kf, n_mp = flat["n_kf"], flat["n_mp"]
    cam_qt = np.stack([to_q(flat["kf_Tcw"][k]) for k in range(n_kf)])
    sf = np.ones(mg.N_LEVELS, np.float32)
    for i in range(1, mg.N_LEVELS):
        sf[i] = sf[i - 1] * mg.SCALE
    inv_sigma2 = (np.float32(1.0) / (sf * sf)).astype(np.float32)                  # ORBextractor.cpp:586-597 in f32
    o_kf, o_mp, o_kp = flat["obs_kf"], flat["obs_mp"], flat["obs_kp"]
    slot = flat["kp_off"][o_kf] + o_kp
    keep = np.bincount(o_mp, minlength=n_mp)[o_mp] >= 2                           # Optimizer.cpp:738-741: points with fewer than two observations stay out
    K = np.tile(flat["K4"].astype(np.float64), (n_kf, 1))
    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1                                # mvpKeyFrameOrigins[0]
    prob2 = dict(n_cam=n_kf, n_pt=n_mp, n_edge=int(keep.sum()), cam_qt=cam_qt, cam_fixed=fixed, cam_K=K, pt_xyz=flat["mp_pos"].astype(np.float64),
                 e_cam=o_kf[keep].astype(np.int32), e_pt=o_mp[keep].astype(np.int32), e_obs=flat["kp_xy"][slot[keep]].astype(np.float64),
                 e_info=inv_sigma2[flat["kp_oct"][slot[keep]]].astype(np.float64), e_level=None, huber_delta=float(np.float32(np.sqrt(5.99))))   # const float thHuber2D = sqrt(5.99)
    ctx = Context(0)
    cam, pts, _, _, stats = optimizer.bundle_adjustment(ctx, prob2, 20)
    ctx.close()
    assert stats.iters_done >= 8 and stats.stop_reason == 3                       # ended by chi2 stagnation like the run of tests/golden/gba_c4_full.npz
    T = np.stack([to_T(cam[k]) for k in range(n_kf)])
    assert np.array_equal(T, st["kf_Tcw"])
    seen = np.bincount(o_mp[keep], minlength=n_mp) > 0
    assert np.array_equal(pts[seen].astype(np.float32), st["mp_pos"][seen])
    assert np.array_equal(st["mp_pos"][~seen], flat["mp_pos"][~seen])


def test_batched_normal_and_depth_of_the_patched_shim_equals_the_per_point_method():
    """shim/Optimizer_hip.cpp built against a MapPoint that carries the OPTIONAL setter of INTEGRATION.md (`SetNormalAndDepth`, three lines a maintainer may
    add; the look-alike has it under -DCCM_LOOKALIKE_MAPPOINT_SETTER): the write-back of MapFusionGBA then hands every point's normal and scale-invariance
    distances to ONE ccm_update_normal_and_depth call instead of calling the reference's UpdateNormalAndDepth() per point.  Everything the call leaves in
    the map — poses, positions, normals, distance ranges — must equal, bit for bit, what the unpatched shim leaves (whose per-point calls run the
    reference's own lines MapPoint.cpp:779-823), on a 3-agent map with fixed and free keyframes."""
    import os
    patched = os.path.join(os.path.dirname(mg.SHIM_LIB), "liboptimizer_hip_shim_patched.so")
    if not os.path.exists(patched):
        pytest.skip("patched shim not built")
    prob = synth.make_ba_problem(n_agents=3, kfs_per_agent=60, n_points=25000, seed=31)
    flat = mg.flat_from_ba_problem(prob, n_agents=3)
    out = []
    for lib in (mg.SHIM_LIB, patched):
        g = mg.MapGraph(lib, flat)
        assert g.map_fusion_gba(0, 20) == 0
        out.append(g.state())
        g.close()
    a, b = out
    assert np.abs(a["mp_normal"]).max() > 0.5 and (a["mp_dmax"] > a["mp_dmin"]).sum() > 20000     # the per-point method ran
    for k in ("kf_Tcw", "mp_pos", "mp_normal", "mp_dmin", "mp_dmax"):
        assert np.array_equal(a[k], b[k]), k
    # the local bundle adjustment's write-back the same way (observations erased as outliers leave the lists; EraseObservation may move a point's
    # reference keyframe, so it is asked of the point after the erasures)
    lflat = mg.flat_from_ba_problem(synth.make_ba_config("lba_c2"))
    out = []
    for lib in (mg.SHIM_LIB, patched):
        g = mg.MapGraph(lib, lflat)
        assert g.local_ba(15, client_id=0) == 0
        out.append(g.state())
        g.close()
    a, b = out
    assert (a["obs_alive"] == 0).sum() > 100                                                       # outlier observations were erased
    for k in ("kf_Tcw", "mp_pos", "mp_normal", "mp_dmin", "mp_dmax", "mp_bad", "obs_alive"):
        assert np.array_equal(a[k], b[k]), k


def test_pose_locked_points_keep_their_position_and_get_consistent_normals():
    """MapPoint::SetWorldPos returns without writing when the point's position is locked and the system is a CLIENT (MapPoint.cpp:340-341), so the
    UpdateNormalAndDepth() that follows (Optimizer.cpp:635-636) works from the OLD position.  Every third point of a local window is locked: the
    reference's own Optimizer.cpp, the shim and the shim with the batched write-back (which computes normals from the optimised positions and therefore
    must send locked points through the per-point method) have to leave the same map: locked points bit-identical to the input, their normals /
    distance ranges those of the old position."""
    lflat = mg.flat_from_ba_problem(synth.make_ba_config("lba_c2"))
    locked = (np.arange(lflat["n_mp"]) % 3 == 0).astype(np.uint8)
    libs = [mg.REF_LIB, mg.SHIM_LIB]
    patched = os.path.join(os.path.dirname(mg.SHIM_LIB), "liboptimizer_hip_shim_patched.so")
    if os.path.exists(patched):
        libs.append(patched)
    out = []
    for lib in libs:
        g = mg.MapGraph(lib, lflat)
        g.lock_points(locked, server=False)
        assert g.local_ba(15, client_id=0) == 0
        out.append(g.state())
        g.close()
    r = out[0]
    lk = locked.astype(bool)
    moved = (r["mp_pos"] != lflat["mp_pos"]).any(axis=1)
    assert moved[~lk].sum() > 500 and not moved[lk].any()                       # the reference: locked points did not move, others did
    for s in out[1:]:
        assert np.array_equal(s["mp_pos"][lk], lflat["mp_pos"][lk])
        keep = (r["mp_bad"] == 0) & (s["mp_bad"] == 0)
        assert (r["obs_alive"] != s["obs_alive"]).sum() <= 2
        assert ulps32(r["mp_pos"][keep], s["mp_pos"][keep]).max() <= 1024
        assert np.abs(r["mp_normal"][keep] - s["mp_normal"][keep]).max() < 1e-4
        assert np.abs(r["mp_dmax"][keep] / np.maximum(s["mp_dmax"][keep], 1e-30) - 1).max() < 1e-4
    if len(out) == 3:                                                            # batched and per-point write-back: the same bits
        for k in ("kf_Tcw", "mp_pos", "mp_normal", "mp_dmin", "mp_dmax", "mp_bad", "obs_alive"):
            assert np.array_equal(out[1][k], out[2][k]), k
