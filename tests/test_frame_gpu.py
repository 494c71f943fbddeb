"""Frame glue on the device (ccm_frame_*) vs the oracle: undistorted keypoints, image bounds, the 75x48 grid and the
batched GetFeaturesInArea + Hamming window search.  Everything here is f32 / integer work restated operation by operation,
so the bar is bit-exact: identical floats, identical candidate lists in identical order, identical distances."""
import numpy as np
import pytest

from ccm_slam_amd import synth
from ccm_slam_amd.frame import GRID_COLS, GRID_ROWS, FrameGrid, is_in_frustum

pytestmark = pytest.mark.gpu

EUROC_K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
EUROC_D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)   # cslam/conf/vi_euroc.yaml


def _kps(oracle_lib, seed=1000, t=3, n=1000):
    return oracle_lib.OrbOracle(n).extract(synth.gen_image(seed, t))


@pytest.mark.parametrize("dist", [EUROC_D, np.zeros(4, np.float32), np.array([-0.3, 0.1, 0.001, -0.0005, 0.02], np.float32)])
def test_undistort_bounds_and_grid_match_the_oracle(ctx, oracle_lib, dist):
    kps, desc = _kps(oracle_lib)
    fg = FrameGrid(ctx, EUROC_K, dist, 752, 480)
    b = fg.bounds
    assert np.array_equal(b, oracle_lib.image_bounds(EUROC_K, dist, 752, 480))
    fg.set_keypoints(kps, desc)
    xy, off, idx = fg.get()
    xy_o = oracle_lib.undistort_points(EUROC_K, dist, np.stack([kps["x"], kps["y"]], 1))
    assert np.array_equal(xy, xy_o)                        # bit-identical floats
    off_o, idx_o = oracle_lib.build_grid(xy_o[:, 0], xy_o[:, 1], b)
    assert np.array_equal(off, off_o) and np.array_equal(idx, idx_o)   # same cells, insertion (index) order inside each
    assert off[-1] > 0.9 * len(kps)
    fg.close()


@pytest.mark.parametrize("seed", [0, 1])
def test_window_search_candidates_and_distances(ctx, oracle_lib, seed):
    kps, desc = _kps(oracle_lib, t=5 + seed)
    fg = FrameGrid(ctx, EUROC_K, EUROC_D, 752, 480)
    fg.set_keypoints(kps, desc)
    xy, _, _ = fg.get()
    b = fg.bounds
    rng = np.random.default_rng(seed)
    Q = 3000
    # queries around real features (tracked map points) plus some far outside the image
    base = xy[rng.integers(0, xy.shape[0], Q)] + rng.normal(size=(Q, 2)).astype(np.float32) * 6
    base[:50] = rng.uniform(-400, 1400, (50, 2))
    u, v = base[:, 0].astype(np.float32), base[:, 1].astype(np.float32)
    lvl = rng.integers(0, 8, Q).astype(np.int32)
    r = (rng.choice([2.5, 4.0, 7.0, 15.0], Q) * 1.2 ** lvl).astype(np.float32)
    minl = np.where(rng.random(Q) < 0.2, -1, lvl - 1).astype(np.int32)
    maxl = np.where(minl < 0, -1, lvl).astype(np.int32)
    minl[100:120] = 0; maxl[100:120] = -1                  # bCheckLevels false / open upper end
    qdesc = desc[rng.integers(0, desc.shape[0], Q)].copy()
    qdesc[:, :4] ^= rng.integers(0, 256, (Q, 4), dtype=np.uint8)
    off, idx, dist = fg.window_search(u, v, r, minl, maxl, qdesc)
    off_o, idx_o = oracle_lib.grid_candidates(xy[:, 0], xy[:, 1], kps["octave"], b, u, v, r, minl, maxl)
    assert np.array_equal(off, off_o) and np.array_equal(idx, idx_o)
    dist_o, _, _, _ = oracle_lib.hamming_csr(qdesc, desc, off_o, idx_o)
    assert np.array_equal(dist, dist_o)
    assert idx.size > Q                                     # the test exercises real lists
    fg.close()


def test_empty_frame_and_empty_queries(ctx):
    fg = FrameGrid(ctx, EUROC_K, EUROC_D, 752, 480)
    kp = np.zeros(0, dtype=[("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"), ("octave", "i4")])
    fg.set_keypoints(kp, np.zeros((0, 32), np.uint8))
    off, idx, dist = fg.window_search(np.array([10.0]), np.array([10.0]), np.array([50.0]), [-1], [-1], np.zeros((1, 32), np.uint8))
    assert off.tolist() == [0, 0] and idx.size == 0
    off, idx, dist = fg.window_search([], [], [], [], [], np.zeros((0, 32), np.uint8))
    assert off.tolist() == [0]
    fg.close()


def _frustum_case(seed, n=20000):
    rng = np.random.default_rng(seed)
    R, t, _ = synth._agent_loop(40, 0)
    R, t = R[5].astype(np.float32), t[5].astype(np.float32)
    Ow = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    b = np.array([-135.79564, 895.5073, -92.875015, 565.5531], np.float32)   # minX maxX minY maxY of the EuRoC camera
    frame24 = np.concatenate([R.ravel(), t, Ow, EUROC_K, b, [np.float32(np.log(np.float32(1.2)))]]).astype(np.float32)
    # points all around the camera: in front / behind / outside the image / too far / too close / bad viewing angle
    P = (Ow + rng.normal(size=(n, 3)) * 6).astype(np.float32)
    view = P - Ow
    normal = view / np.linalg.norm(view, axis=1, keepdims=True)
    normal = (normal + rng.normal(size=(n, 3)) * rng.choice([0.05, 0.8], (n, 1))).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True).astype(np.float32)
    d = np.linalg.norm(view, axis=1)
    dmax = (d * rng.uniform(0.6, 4.0, n)).astype(np.float32)
    dmin = (dmax / np.float32(1.2 ** 7)).astype(np.float32)
    return frame24, P, normal.astype(np.float32), dmin, dmax


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frustum_cull_matches_the_oracle_bit_for_bit(ctx, oracle_lib, seed):
    frame24, P, normal, dmin, dmax = _frustum_case(seed)
    got = is_in_frustum(ctx, frame24, 8, P, normal, dmin, dmax, 0.5)
    exp = oracle_lib.is_in_frustum(frame24, 8, P, normal, dmin, dmax, 0.5)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    frac = got[0].mean()
    assert 0.02 < frac < 0.6                                  # every rejection branch and the accept branch are exercised
    assert len(np.unique(got[3][got[0] == 1])) >= 6           # predicted levels span the pyramid


def test_window_search_capacity_shortfall_is_reported_not_overrun(ctx, oracle_lib):
    import ctypes as C
    from ccm_slam_amd._lib import lib
    kps, desc = _kps(oracle_lib, t=9)
    fg = FrameGrid(ctx, EUROC_K, EUROC_D, 752, 480)
    fg.set_keypoints(kps, desc)
    xy, _, _ = fg.get()
    Q = 500
    u = np.ascontiguousarray(xy[:Q, 0]); v = np.ascontiguousarray(xy[:Q, 1])
    r = np.full(Q, 60.0, np.float32); ml = np.full(Q, -1, np.int32)
    qd = np.ascontiguousarray(desc[:Q])
    off = np.zeros(Q + 1, np.int32); idx = np.full(64, -7, np.int32); dist = np.full(64, 9999, np.uint16); n = C.c_int64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().ccm_frame_window_search(fg._h, Q, p(u), p(v), p(r), p(ml), p(ml), p(qd), p(off), p(idx), p(dist), C.c_int64(64), C.byref(n))
    assert rc != 0 and n.value > 64 and off[-1] == n.value          # the true size comes back for the retry
    assert (idx == -7).all() and (dist == 9999).all()               # nothing was written into the short buffers
    off2, idx2, dist2 = fg.window_search(u, v, r, ml, ml, qd)       # the wrapper retries with the exact size
    assert idx2.size == n.value and np.array_equal(off2, off)
    fg.close()


def test_update_normal_and_depth_matches_the_oracle_bit_for_bit(ctx, oracle_lib):
    """MapPoint::UpdateNormalAndDepth (MapPoint.cpp:779-823) batched on the device (second half): f32 normals and distance
    bounds bit-exact for 60 000 map points with 0..30 observers; points without observers keep their values."""
    from tests.test_oracle_match import _normal_depth_case
    from ccm_slam_amd.frame import update_normal_and_depth
    for seed, n_pt, n_kf in ((0, 60000, 2000), (1, 17, 3)):
        pos, off, obs, kfc, ref_kf, ref_level, sf, old = _normal_depth_case(seed, n_pt, n_kf)
        got = update_normal_and_depth(ctx, pos, off, obs, kfc, ref_kf, ref_level, sf, *old)
        exp = oracle_lib.update_normal_and_depth(pos, off, obs, kfc, ref_kf, ref_level, sf, *old)
        for g, e in zip(got, exp):
            assert np.array_equal(g.reshape(-1), e.reshape(-1))
    # bad arguments are refused, not executed
    from ccm_slam_amd._lib import CcmError
    pos, off, obs, kfc, ref_kf, ref_level, sf, old = _normal_depth_case(2, 50, 5)
    bad = obs.copy(); bad[0] = 99
    with pytest.raises(CcmError):
        update_normal_and_depth(ctx, pos, off, bad, kfc, ref_kf, ref_level, sf, *old)
