"""The prepared call objects of the Python layer (arguments converted once, one C-ABI call per run(): what bench.py times) return exactly what the plain wrappers
return, call after call."""
import numpy as np
import pytest

from ccm_slam_amd import frame, optimizer, orb, synth

pytestmark = pytest.mark.gpu

K4 = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
D4 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)


def test_prepared_calls_equal_the_plain_wrappers(ctx):
    ex = orb.ORBextractor(ctx, 1000)
    pex = ex.prepared(752, 480)
    for t in range(3):
        img = synth.gen_image(1000, t)
        kps, desc = ex(img)
        n = pex.run(img)
        assert n == len(kps) and np.array_equal(pex.kps[:n], kps) and np.array_equal(pex.desc[:n], desc)
    fg = frame.FrameGrid(ctx, K4, D4, 752, 480)
    fg.set_keypoints(kps, desc)
    ref = fg.get()
    psk = fg.prepared_set_keypoints(kps, desc)
    psk.run()
    for a, b in zip(fg.get(), ref):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(5)
    xy = ref[0]
    src = rng.integers(0, len(kps), 700)
    u = (xy[src, 0] + rng.normal(0, 2, 700)).astype(np.float32); v = (xy[src, 1] + rng.normal(0, 2, 700)).astype(np.float32)
    lvl = np.clip(kps["octave"][src] + rng.integers(0, 2, 700), 0, 7).astype(np.int32)
    q = (u, v, (7.0 * 1.2 ** lvl).astype(np.float32), (lvl - 1).astype(np.int32), lvl, desc[src].copy())
    pw = fg.prepared_window_search(*q)
    for _ in range(2):
        pw.run()
        for a, b in zip(pw.result(), fg.window_search(*q)):
            assert np.array_equal(a, b)
    R, t, _ = synth._agent_loop(40, 0)
    R, t = R[5].astype(np.float32), t[5].astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    b = fg.bounds
    frame24 = np.concatenate([R.ravel(), t, Ow, K4, [b[0], b[2], b[1], b[3]], [np.float32(np.log(np.float32(1.2)))]]).astype(np.float32)
    P = (Ow + rng.normal(size=(500, 3)) * 6).astype(np.float32)
    nrm = P - Ow
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    dmax = (np.linalg.norm(P - Ow, axis=1) * rng.uniform(0.6, 4.0, 500)).astype(np.float32)
    dmin = (dmax / np.float32(1.2 ** 7)).astype(np.float32)
    pfr = frame.prepared_is_in_frustum(ctx, frame24, 8, P, nrm, dmin, dmax)
    pfr.run()
    for a, b in zip(pfr.result(), frame.is_in_frustum(ctx, frame24, 8, P, nrm, dmin, dmax)):
        assert np.array_equal(a, b)
    p = synth.make_pose_problem(300, 0, 0.1)
    ref_pose = optimizer.pose_optimization(ctx, p["cam_qt"], p["Xw"], p["obs"], p["info"], p["K"])
    ppo = optimizer.PoseOptCall(ctx, p["cam_qt"], p["Xw"], p["obs"], p["info"], p["K"])
    for _ in range(3):   # run() restores the initial pose: every call is the same optimisation
        ppo.run()
        assert np.array_equal(ppo.cam, ref_pose[0]) and np.array_equal(ppo.outlier[:ppo.n], ref_pose[1]) and ppo.n_inlier == ref_pose[2]
    fg.close(); ex.close()
