"""CPU: the matcher oracle against self-evident known answers (the reference ships no golden vectors)."""
import numpy as np

import oracle


def _popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_descriptor_distance_is_popcount():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert oracle.descriptor_distance(a, b) == _popcount_dist(a, b)
    z = np.zeros(32, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, np.full(32, 255, np.uint8)) == 256


def test_dense_best2_brute_force():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (20, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    t[7] = q[3]; t[9] = q[3]   # exact duplicates: first index must win, second-best distance is 0 too
    bi, bd, sd = oracle.hamming_dense_best2(q, t)
    for i in range(20):
        d = np.array([_popcount_dist(q[i], t[j]) for j in range(50)])
        assert bd[i] == d.min() and bi[i] == int(np.argmin(d))
        assert sd[i] == np.sort(d)[1]
    assert bi[3] == 7 and bd[3] == 0 and sd[3] == 0


def test_grid_candidates_window_and_levels():
    # 4 keypoints, Frame grid 75x48 over a 752x480 image (Frame.cpp:87-88, 200-253)
    kx = np.array([100.0, 103.0, 140.0, 100.0], np.float32)
    ky = np.array([100.0, 101.0, 100.0, 100.0], np.float32)
    octv = np.array([0, 1, 0, 3], np.int32)
    bounds = (0.0, 0.0, 752.0, 480.0)
    off, idx = oracle.grid_candidates(kx, ky, octv, bounds, [100.0, 100.0, 100.0], [100.0, 100.0, 100.0], [5.0, 5.0, 50.0],
                                      [-1, 1, 0], [-1, 3, 0])
    assert sorted(idx[off[0]:off[1]]) == [0, 1, 3]        # no level check
    assert sorted(idx[off[1]:off[2]]) == [1, 3]           # levels 1..3
    assert sorted(idx[off[2]:off[3]]) == [0, 2]           # maxLevel = 0 keeps octave 0 only
    # strict inequality: a point exactly r away is excluded (|dx| < r)
    off, idx = oracle.grid_candidates(kx, ky, octv, bounds, [95.0], [100.0], [5.0], [-1], [-1])
    assert 0 not in idx[off[0]:off[1]]


def test_search_by_projection_claims_are_sequential():
    # two map points project onto the same spot; one feature.  The first map point claims it, the second
    # one must then skip it (ORBmatcher.cpp:113-115) even though its descriptor matches better.
    d = np.zeros((1, 32), np.uint8)
    mp_desc = np.zeros((2, 32), np.uint8)
    mp_desc[0, 0] = 0b111          # distance 3
    n, fm = oracle.search_by_projection_mp([100.0], [100.0], [0], d, (0, 0, 752, 480), np.ones(8, np.float32),
                                           [1, 1], [100.0, 100.0], [100.0, 100.0], [0, 0], [1.0, 1.0], mp_desc, 1.0, 0.6,
                                           [-1])
    assert n == 1 and fm[0] == 0
    # not in view -> skipped
    n, fm = oracle.search_by_projection_mp([100.0], [100.0], [0], d, (0, 0, 752, 480), np.ones(8, np.float32),
                                           [0, 1], [100.0, 100.0], [100.0, 100.0], [0, 0], [1.0, 1.0], mp_desc, 1.0, 0.6,
                                           [-1])
    assert n == 1 and fm[0] == 1


def test_search_by_projection_ratio_test_same_level_only():
    # two features at the same level with distances 40 and 50: 40 > 0.6*50 -> rejected;
    # when the second-best sits on another level the ratio test is skipped (ORBmatcher.cpp:137-141)
    f = np.zeros((2, 32), np.uint8)
    f[0, :5] = 0xFF   # 40 bits
    f[1, :6] = 0xFF; f[1, 6] = 0x03  # 50 bits
    mp = np.zeros((1, 32), np.uint8)
    args = ([100.0, 101.0], [100.0, 100.0])
    n, fm = oracle.search_by_projection_mp(*args, [1, 1], f, (0, 0, 752, 480), np.ones(8, np.float32), [1], [100.0], [100.0],
                                           [1], [1.0], mp, 1.0, 0.6, [-1, -1])
    assert n == 0
    n, fm = oracle.search_by_projection_mp(*args, [1, 0], f, (0, 0, 752, 480), np.ones(8, np.float32), [1], [100.0], [100.0],
                                           [1], [1.0], mp, 1.0, 0.6, [-1, -1])
    assert n == 1 and fm[0] == 0


def test_search_by_projection_last_rotation_histogram():
    # 12 consistent matches (rotation 0) and one with a 90 degree rotation: the odd one is removed
    # (HISTO_LENGTH bins of 30 degrees, ORBmatcher.cpp:1358,1437-1445)
    n_f = 13
    kx = np.arange(n_f, dtype=np.float32) * 40 + 50
    ky = np.full(n_f, 200, np.float32)
    desc = np.zeros((n_f, 32), np.uint8)
    for i in range(n_f):
        desc[i, i] = 0xFF
    kangle = np.zeros(n_f, np.float32)
    l_angle = np.zeros(n_f, np.float32); l_angle[5] = 90.0
    n, cur = oracle.search_by_projection_last(kx, ky, np.zeros(n_f, np.int32), kangle, desc, (0, 0, 752, 480),
                                              np.ones(8, np.float32), np.ones(n_f, np.uint8), kx, ky, np.zeros(n_f, np.int32),
                                              l_angle, desc, 7.0, 1, -np.ones(n_f, np.int32))
    assert n == 12 and cur[5] == -1 and (np.delete(cur, 5) == np.delete(np.arange(n_f), 5)).all()


def test_undistort_restatement_inverts_the_distortion_model(oracle_lib):
    """cv::undistortPoints restated (five fixed-point iterations, P = K): applying the forward Brown-Conrady model to the
    result must give back the input pixel to the accuracy five iterations reach; zero distortion is the identity; the
    image bounds are the undistorted corners (Frame.cpp:314-347)."""
    import numpy as np
    K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
    rng = np.random.default_rng(0)
    xy = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    un = oracle_lib.undistort_points(K, D, xy).astype(np.float64)
    x = (un[:, 0] - K[2]) / K[0]; y = (un[:, 1] - K[3]) / K[1]; r2 = x * x + y * y
    cd = 1 + D[0] * r2 + D[1] * r2 * r2
    xd = x * cd + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x); yd = y * cd + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    back = np.stack([xd * K[0] + K[2], yd * K[1] + K[3]], 1)
    err = np.abs(back - xy).max(axis=1)
    centre = np.hypot(xy[:, 0] - K[2], xy[:, 1] - K[3]) < 200
    assert err[centre].max() < 5e-3 and err.max() < 0.5          # five iterations: converged near the centre, close at the corners
    assert np.array_equal(oracle_lib.undistort_points(K, np.zeros(4, np.float32), xy), xy)
    b = oracle_lib.image_bounds(K, D, 752, 480)
    assert b[0] < 0 and b[1] < 0 and b[2] > 752 and b[3] > 480    # barrel distortion: the undistorted image is larger
    assert np.array_equal(oracle_lib.image_bounds(K, np.zeros(4, np.float32), 752, 480), np.array([0, 0, 752, 480], np.float32))
    off, idx = oracle_lib.build_grid(un[:, 0].astype(np.float32), un[:, 1].astype(np.float32), b)
    assert off[-1] == 500 and sorted(idx.tolist()) == list(range(500))
    for c in range(75 * 48):
        seg = idx[off[c]:off[c + 1]]
        assert (np.diff(seg) > 0).all()                           # push_back order


def test_frustum_oracle_on_hand_built_points(oracle_lib):
    """Frame::isInFrustum restated: identity pose, EuRoC intrinsics; one point per branch of Frame.cpp:139-198."""
    import numpy as np
    K = [458.654, 457.296, 367.215, 248.375]
    frame = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1] + [0, 0, 0] + [0, 0, 0] + K + [0, 752, 0, 480] + [np.log(np.float32(1.2))], np.float32)
    P = np.array([[0, 0, 4], [0, 0, -4], [10, 0, 4], [0, 0, 4], [0, 0, 4], [0, 0, 4], [0.5, 0.2, 8]], np.float32)
    nrm = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [1, 0, 0], [0, 0, 1]], np.float32)
    dmax = np.array([8, 8, 8, 3, 80, 8, 8], np.float32)      # 1.2*3 < 4: too far; min = dmax/1.2^7: 80 -> 22 > 4/0.8: too close
    dmin = (dmax / np.float32(1.2 ** 7)).astype(np.float32)
    inv, u, v, lvl, cs = oracle_lib.is_in_frustum(frame, 8, P, nrm, dmin, dmax, 0.5)
    assert inv.tolist() == [1, 0, 0, 0, 0, 0, 1]
    assert u[0] == np.float32(367.215) and v[0] == np.float32(248.375) and cs[0] == 1.0
    assert lvl[0] == int(np.ceil(np.log(np.float32(2.0)) / np.log(np.float32(1.2))))   # ceil(3.80) = 4
    assert lvl[6] in (0, 1)                                                               # ratio just below 1


def _normal_depth_case(seed, n_pt=5000, n_kf=120):
    rng = np.random.default_rng(seed)
    kf_center = (rng.normal(size=(n_kf, 3)) * np.array([8, 5, 1.5])).astype(np.float32)
    pos = (rng.normal(size=(n_pt, 3)) * np.array([10, 6, 2])).astype(np.float32)
    k = np.clip(rng.poisson(6, n_pt), 0, min(30, n_kf))
    k[rng.random(n_pt) < 0.02] = 0                                  # observations.empty(): the point keeps its old values
    off = np.zeros(n_pt + 1, np.int32); off[1:] = np.cumsum(k)
    obs = np.concatenate([rng.permutation(n_kf)[:kk] for kk in k] + [np.zeros(0, np.int64)]).astype(np.int32)
    ref_kf = np.array([obs[off[i]] if k[i] else 0 for i in range(n_pt)], np.int32)      # mpRefKF is one of the observers
    ref_level = rng.integers(0, 8, n_pt).astype(np.int32)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32)   # mvScaleFactor[i] = [i-1] * 1.2f
    old = (rng.normal(size=(n_pt, 3)).astype(np.float32), rng.uniform(0.1, 1, n_pt).astype(np.float32), rng.uniform(5, 9, n_pt).astype(np.float32))
    return pos, off, obs, kf_center, ref_kf, ref_level, sf, old


def test_update_normal_and_depth_oracle_on_hand_numbers(oracle_lib):
    """MapPoint::UpdateNormalAndDepth (MapPoint.cpp:779-823): mean unit viewing direction over the observers, max distance = |P - O_ref| * scale[level]
    of the reference keyframe's keypoint, min distance = max / scale[last]."""
    pos = np.array([[0, 0, 4]], np.float32)
    kfc = np.array([[0, 0, 0], [3, 0, 4], [0, -4, 1]], np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32)
    nrm, dmin, dmax = oracle_lib.update_normal_and_depth(pos, [0, 3], [0, 1, 2], kfc, [2], [3], sf, np.zeros((1, 3)), [0.0], [0.0])
    exp = (np.array([0, 0, 1.0]) + np.array([-1.0, 0, 0]) + np.array([0, 4, 3]) / 5) / 3
    assert np.abs(nrm[0] - exp).max() < 1e-6
    assert abs(dmax[0] - 5 * sf[3]) < 1e-5 and abs(dmin[0] - dmax[0] / sf[7]) < 1e-6
    # f32 semantics: independent numpy evaluation with the same roundings, on a random map
    pos, off, obs, kfc, ref_kf, ref_level, sf, old = _normal_depth_case(3, 400, 40)
    nrm, dmin, dmax = oracle_lib.update_normal_and_depth(pos, off, obs, kfc, ref_kf, ref_level, sf, *old)
    for i in range(400):
        if off[i + 1] == off[i]:
            assert np.array_equal(nrm[i], old[0][i]) and dmin[i] == old[1][i] and dmax[i] == old[2][i]
            continue
        acc = np.zeros(3, np.float32)
        for o in obs[off[i]:off[i + 1]]:
            d = pos[i] - kfc[o]
            a = np.float32(1.0 / np.sqrt((d.astype(np.float64) ** 2).sum()))
            acc = acc + d * a
        assert np.array_equal(nrm[i], acc * np.float32(1.0 / (off[i + 1] - off[i])))
        pc = pos[i] - kfc[ref_kf[i]]
        mx = np.float32(np.sqrt((pc.astype(np.float64) ** 2).sum())) * sf[ref_level[i]]
        assert dmax[i] == mx and dmin[i] == mx / sf[7]
