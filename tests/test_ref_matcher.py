"""Pins the matcher oracle (oracle/match_ref.cpp) to the REFERENCE'S OWN cslam/src/ORBmatcher.cpp: the whole file compiled verbatim
(oracle/Makefile.ref -> oracle/_ref/libmatcher_ref.so) against look-alike Frame / KeyFrame / MapPoint classes whose grid functions are the
reference's own lines (Frame.cpp:103-118, 200-265; KeyFrame.cpp:1162-1206), the vendored DBoW2 FeatureVector, and the look-alike cv:: API.
oracle/ref_matcher_driver.cpp gives every method the same flat inputs as the oracle's restatement and flattens what the method wrote.
Rows of SURVEY 8a pinned here: G (GetFeaturesInArea candidate order), M1, M2 (incl. the reference's own f32 projection), M3, M4, M6.
The GPU parity tests (tests/test_host_mirror_gpu.py) compare the product with the oracle on the same scenarios."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from ccm_slam_amd import synth
from oracle import ref

LIB = os.path.join(os.path.dirname(os.path.abspath(ref.__file__)), "_ref", "libmatcher_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB) and not os.path.isdir("/root/reference/cslam"), reason="oracle/_ref not built")
BOUNDS = (0.0, 0.0, 752.0, 480.0)


@pytest.fixture(scope="module")
def rlib():
    if not os.path.exists(LIB):
        ref.build()
    return C.CDLL(LIB)


@pytest.fixture(scope="module")
def frames():
    o = oracle.OrbOracle(1000)
    out = [o.extract(synth.gen_image(1000, t)) for t in (0, 1)]
    o.close()
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


c = np.ascontiguousarray
fb = [C.c_float(b) for b in BOUNDS]


def test_get_features_in_area_order_is_the_references(rlib, frames):
    """Frame::GetFeaturesInArea + AssignFeaturesToGrid + PosInGrid of the reference vs the oracle's grid: the candidate lists (ix-major, iy, insertion
    order) of 4000 window queries with and without level filters — the order that defines every tie-break of the window searches."""
    kps, _ = frames[0]
    rng = np.random.default_rng(0)
    Q = 4000
    src = rng.integers(0, len(kps), Q)
    qx = (kps["x"][src] + rng.normal(0, 6, Q)).astype(np.float32)
    qy = (kps["y"][src] + rng.normal(0, 6, Q)).astype(np.float32)
    qx[:50] = rng.uniform(-40, 800, 50).astype(np.float32); qy[:50] = rng.uniform(-40, 520, 50).astype(np.float32)   # partly outside the image
    qr = rng.uniform(2, 40, Q).astype(np.float32)
    qmin = rng.integers(-1, 7, Q).astype(np.int32); qmax = (qmin + rng.integers(0, 3, Q)).astype(np.int32)
    qmax[::5] = -1; qmin[::7] = -1
    kx, ky, oc = c(kps["x"]), c(kps["y"]), c(kps["octave"])
    off = np.zeros(Q + 1, np.int32)
    rlib.ref_grid_candidates.restype = C.c_int64
    n = rlib.ref_grid_candidates(_p(kx), _p(ky), _p(oc), len(kps), *fb, _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), Q, _p(off), None, C.c_int64(0))
    idx = np.zeros(max(n, 1), np.int32)
    rlib.ref_grid_candidates(_p(kx), _p(ky), _p(oc), len(kps), *fb, _p(qx), _p(qy), _p(qr), _p(qmin), _p(qmax), Q, _p(off), _p(idx), C.c_int64(n))
    ooff, oidx = oracle.grid_candidates(kx, ky, oc, BOUNDS, qx, qy, qr, qmin, qmax)
    assert n > 10000 and np.array_equal(off, ooff) and np.array_equal(idx[:n], oidx)


def test_search_by_projection_map_points_M1(rlib, frames):
    kps, desc = frames[0]
    N = len(kps)
    rng = np.random.default_rng(0)
    sf = synth.scale_tables()[0]
    n_mp = 3000
    src = rng.integers(0, N, n_mp)
    px = (kps["x"][src] + rng.normal(0, 2.0, n_mp)).astype(np.float32)
    py = (kps["y"][src] + rng.normal(0, 2.0, n_mp)).astype(np.float32)
    lvl = np.clip(kps["octave"][src] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    vcos = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    in_view = (rng.random(n_mp) < 0.9).astype(np.uint8)
    bits = np.unpackbits(desc[src], axis=1)
    mp_desc = np.packbits(bits ^ (rng.random(bits.shape) < 0.08), axis=1)
    frame_mp0 = -np.ones(N, np.int32)
    frame_mp0[rng.integers(0, N, 50)] = 10_000
    for th, ratio in ((3.0, 0.8), (1.0, 0.6)):
        exp_n, exp = oracle.search_by_projection_mp(kps["x"], kps["y"], kps["octave"], desc, BOUNDS, sf, in_view, px, py, lvl, vcos, mp_desc, th, ratio, frame_mp0)
        got = frame_mp0.copy()
        n = rlib.ref_search_by_projection_mp(_p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(desc), N, *fb, _p(sf), n_mp, _p(in_view), _p(px), _p(py),
                                             _p(lvl), _p(vcos), _p(mp_desc), C.c_float(th), C.c_float(ratio), _p(got))
        assert n == exp_n and n > 300
        assert np.array_equal(got, exp)


def test_search_by_projection_last_frame_M2_with_the_references_own_projection(rlib, frames):
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(1)
    sf = synth.scale_tables()[0]
    n_last = len(k1)
    K4 = np.array(synth.EUROC_K, np.float32)
    # last frame at the identity, current frame shifted by ~1 px at 5 m depth; map points back-projected from the last frame's keypoints
    Tl = np.eye(4, dtype=np.float32)
    Tc = np.eye(4, dtype=np.float32); Tc[:3, 3] = [-5.0 / K4[0], -5.0 / K4[1], 0.0]
    z = rng.uniform(3, 8, n_last).astype(np.float32)
    Xw = np.stack([(k1["x"] - K4[2]) / K4[0] * z, (k1["y"] - K4[3]) / K4[1] * z, z], 1).astype(np.float32)
    Xw[:30, 2] = -1.0                                   # behind the camera
    has = (rng.random(n_last) < 0.7).astype(np.uint8)
    outl = (rng.random(n_last) < 0.1).astype(np.uint8)
    cur0 = -np.ones(len(k2), np.int32); cur0[rng.integers(0, len(k2), 30)] = 5000
    for th, ori in ((7.0, 1), (15.0, 0)):
        got = cur0.copy()
        lvalid = np.zeros(n_last, np.uint8); lu = np.zeros(n_last, np.float32); lv = np.zeros(n_last, np.float32)
        n = rlib.ref_search_by_projection_last(_p(c(k2["x"])), _p(c(k2["y"])), _p(c(k2["octave"])), _p(c(k2["angle"])), _p(d2), len(k2), *fb, _p(sf), _p(Tc), _p(K4),
                                               n_last, _p(Tl), _p(has), _p(outl), _p(Xw), _p(c(k1["octave"])), _p(c(k1["angle"])), _p(d1), C.c_float(th), ori, _p(got),
                                               _p(lvalid), _p(lu), _p(lv))
        assert 100 < lvalid.sum() < n_last
        exp_n, exp = oracle.search_by_projection_last(k2["x"], k2["y"], k2["octave"], k2["angle"], d2, BOUNDS, sf, lvalid, lu, lv, k1["octave"], k1["angle"], d1, th, ori,
                                                      cur0)
        # features claimed on entry keep their (foreign) map point in the reference; the oracle reports them with the entry value
        mine = cur0 < 0
        assert n == exp_n and n > 100
        assert np.array_equal(got[mine], exp[mine])


def _feature_vector(desc, seed, n_nodes=60):
    node_of = (desc[:, 0].astype(int) * 7 + (desc[:, 1] >> 5) + seed) % n_nodes
    nodes = np.unique(node_of)
    off, idx = [0], []
    for nd in nodes:
        idx.extend(np.nonzero(node_of == nd)[0].tolist())
        off.append(len(idx))
    return nodes.astype(np.int32), np.array(off, np.int32), np.array(idx, np.int32)


def test_search_by_bow_M3_M4(rlib, frames):
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(2)
    fv1, fv2 = _feature_vector(d1, 0), _feature_vector(d2, 0)
    has1 = (rng.random(len(k1)) < 0.8).astype(np.uint8); has2 = (rng.random(len(k2)) < 0.8).astype(np.uint8)
    a1, a2 = c(k1["angle"]), c(k2["angle"])
    for ratio, ori in ((0.7, 1), (0.9, 0)):
        exp_n, exp = oracle.search_by_bow_kf_frame(fv1, fv2, has1, d1, a1, d2, a2, ratio, ori)
        got = np.zeros(len(k2), np.int32)
        n = rlib.ref_search_by_bow_kf_frame(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), fv1[0].size, _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), fv2[0].size, _p(has1), _p(d1), _p(a1), len(k1),
                                            _p(d2), _p(a2), len(k2), C.c_float(ratio), ori, _p(got))
        assert n == exp_n and n > 100 and np.array_equal(got, exp)
        exp_n, exp = oracle.search_by_bow_kf_kf(fv1, fv2, has1, has2, d1, a1, d2, a2, ratio, ori)
        got = np.zeros(len(k1), np.int32)
        n = rlib.ref_search_by_bow_kf_kf(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), fv1[0].size, _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), fv2[0].size, _p(has1), _p(has2), _p(d1), _p(a1),
                                         len(k1), _p(d2), _p(a2), len(k2), C.c_float(ratio), ori, _p(got))
        assert n == exp_n and n > 100 and np.array_equal(got, exp)


def test_search_for_initialization_M6(rlib, frames):
    (k1, d1), (k2, d2) = frames
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    for window, ratio, ori in ((100, 0.9, 1), (30, 0.7, 0)):
        exp_n, exp, exp_prev = oracle.search_for_initialization(k1, d1, k2, d2, BOUNDS, prev, window, ratio, ori)
        got = np.zeros(len(k1), np.int32); gprev = prev.copy()
        n = rlib.ref_search_for_initialization(_p(c(k1["x"])), _p(c(k1["y"])), _p(c(k1["octave"])), _p(c(k1["angle"])), _p(d1), len(k1), _p(c(k2["x"])), _p(c(k2["y"])),
                                               _p(c(k2["octave"])), _p(c(k2["angle"])), _p(d2), len(k2), *fb, _p(gprev), window, C.c_float(ratio), ori, _p(got))
        assert n == exp_n and n > 50
        assert np.array_equal(got, exp) and np.array_equal(gprev, exp_prev)


def test_search_for_triangulation_M5_with_the_references_own_epipole(rlib, frames):
    """ORBmatcher::SearchForTriangulation: only features WITHOUT a map point, `dist <= TH_LOW && dist <= best` (a later equal candidate replaces an
    earlier one), the epipole test at 100 x scale and the epipolar-line test 3.84 sigma^2 (CheckDistEpipolarLine :159-176); the reference derives the
    epipole from the two keyframe poses with cv::Mat arithmetic, the driver returns it for the flat oracle."""
    (k1, d1), (k2, d2) = frames
    rng = np.random.default_rng(3)
    fv1, fv2 = _feature_vector(d1, 0), _feature_vector(d2, 0)
    has1 = (rng.random(len(k1)) < 0.3).astype(np.uint8); has2 = (rng.random(len(k2)) < 0.3).astype(np.uint8)
    K4 = np.array(synth.EUROC_K, np.float32)
    T1 = np.eye(4, dtype=np.float32)
    T2 = np.eye(4, dtype=np.float32); T2[:3, 3] = [-0.12, 0.01, 0.02]
    # fundamental matrix of a mostly sideways motion: epipolar lines roughly horizontal, so that the 1-px shifted scene passes the line test
    Kk = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float64)
    t = -T2[:3, 3].astype(np.float64)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F12 = (np.linalg.inv(Kk).T @ tx @ np.linalg.inv(Kk)).astype(np.float32)
    sf, s2 = synth.scale_tables()[0], synth.scale_tables()[2]
    for ori in (1, 0):
        got = np.zeros(len(k1), np.int32); ex = C.c_float(0); ey = C.c_float(0)
        n = rlib.ref_search_for_triangulation(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), fv1[0].size, _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), fv2[0].size, _p(has1), _p(has2), _p(d1),
                                              _p(c(k1["x"])), _p(c(k1["y"])), _p(c(k1["angle"])), len(k1), _p(d2), _p(c(k2["x"])), _p(c(k2["y"])), _p(c(k2["octave"])),
                                              _p(c(k2["angle"])), len(k2), _p(F12), _p(T1), _p(T2), _p(K4), _p(s2), _p(sf), ori, _p(got), C.byref(ex), C.byref(ey))
        exp_n, exp = oracle.search_for_triangulation(fv1, fv2, has1, has2, d1, k1["x"], k1["y"], k1["angle"], d2, k2["x"], k2["y"], k2["octave"], k2["angle"], F12,
                                                     ex.value, ey.value, s2, sf, ori)
        assert n == exp_n and n > 30, (n, exp_n)
        assert np.array_equal(got, exp)


def test_triangulation_fan_out_of_a_new_keyframe_M5_x20(rlib, frames):
    """LocalMapping::CreateNewMapPoints (Mapping.cpp:277-470) through the reference's class API: ONE new keyframe, SearchForTriangulation against 20 covisibility
    neighbours one after the other, every matched pair turned into a map point of both keyframes before the next call — against the flat oracle replaying the same
    sequence.  (CPU: the reference's ORBmatcher.cpp; tests/test_shim_matcher_gpu.py runs the same function on shim/ORBmatcher_hip.cpp, whose FIRST call computes the
    Hamming tables of all 20 neighbours in one launch.)"""
    (k1, d1), _ = frames
    o = oracle.OrbOracle(1000)
    nbs = [o.extract(synth.gen_image(1000 + j // 5, 1 + j % 5)) for j in range(20)]
    o.close()
    rng = np.random.default_rng(21)
    fv1 = _feature_vector(d1, 0); fvs = [_feature_vector(d, 0) for _, d in nbs]
    has1 = (rng.random(len(k1)) < 0.3).astype(np.uint8); has2 = [(rng.random(len(k)) < 0.3).astype(np.uint8) for k, _ in nbs]
    K4 = np.array(synth.EUROC_K, np.float32)
    Kk = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float64)
    T1 = np.eye(4, dtype=np.float32)
    T2s = np.tile(np.eye(4, dtype=np.float32), (20, 1, 1)); F12s = np.zeros((20, 9), np.float32)
    for j in range(20):
        T2s[j, :3, 3] = [-0.12 - 0.01 * j, 0.01 * (j % 3), 0.02]
        t = -T2s[j, :3, 3].astype(np.float64)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        F12s[j] = (np.linalg.inv(Kk).T @ tx @ np.linalg.inv(Kk)).astype(np.float32).ravel()
    sf, s2 = synth.scale_tables()[0], synth.scale_tables()[2]
    keep = []

    def ptrs(arrs):
        arrs = [c(a) for a in arrs]
        keep.extend(arrs)
        return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    nn2 = np.array([f[0].size for f in fvs], np.int32); N2 = np.array([len(k) for k, _ in nbs], np.int32)
    got = np.zeros((20, len(k1)), np.int32); n_got = np.zeros(20, np.int32)
    assert rlib.ref_triangulation_fan_out(_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), fv1[0].size, _p(has1), _p(d1), _p(c(k1["x"])), _p(c(k1["y"])), _p(c(k1["angle"])), len(k1), _p(T1),
                                          20, ptrs([f[0] for f in fvs]), ptrs([f[1] for f in fvs]), ptrs([f[2] for f in fvs]), _p(nn2), ptrs(has2), ptrs([d for _, d in nbs]),
                                          ptrs([k["x"] for k, _ in nbs]), ptrs([k["y"] for k, _ in nbs]), ptrs([k["octave"] for k, _ in nbs]), ptrs([k["angle"] for k, _ in nbs]),
                                          _p(N2), _p(F12s), _p(T2s), _p(K4), _p(s2), _p(sf), 0, _p(got), _p(n_got)) == 0
    # the oracle replays the sequence: epipole of camera 1 in image j in the reference's f32 arithmetic (identity rotations: C2 = t2 - t1 = t2)
    has1_now = has1.copy()
    total = 0
    for j, ((k2, d2), fv2) in enumerate(zip(nbs, fvs)):
        C2 = T2s[j, :3, 3]
        invz = np.float32(1.0) / C2[2]
        ex = np.float32(K4[0] * C2[0] * invz + K4[2]); ey = np.float32(K4[1] * C2[1] * invz + K4[3])
        exp_n, exp = oracle.search_for_triangulation(fv1, fv2, has1_now, has2[j], d1, k1["x"], k1["y"], k1["angle"], d2, k2["x"], k2["y"], k2["octave"], k2["angle"], F12s[j],
                                                     float(ex), float(ey), s2, sf, 0)
        assert n_got[j] == exp_n and np.array_equal(got[j], exp), j
        has1_now[exp >= 0] = 1
        total += exp_n
    assert total > 100


def _projected_case(frames, seed, sim3_scale=1.0):
    """A keyframe (frame 0's features) with a non-trivial pose and map points back-projected from its keypoints (jittered, some behind the camera, out of
    range, viewed from the side), set up so that PredictScale lands on / next to the source feature's level."""
    kps, desc = frames[0]
    N = len(kps)
    rng = np.random.default_rng(seed)
    sf, _, _, isig = synth.scale_tables()
    K4 = np.array(synth.EUROC_K, np.float32)
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float64)
    t = np.array([0.4, -0.2, 0.7])
    T = np.eye(4, dtype=np.float32); T[:3, :3] = R; T[:3, 3] = t
    S = np.eye(4, dtype=np.float32); S[:3, :3] = sim3_scale * R; S[:3, 3] = sim3_scale * t
    n_pts = 2500
    src = rng.integers(0, N, n_pts)
    z = rng.uniform(3, 9, n_pts)
    uu = kps["x"][src] + rng.normal(0, 1.5, n_pts); vv = kps["y"][src] + rng.normal(0, 1.5, n_pts)
    Xc = np.stack([(uu - K4[2]) / K4[0] * z, (vv - K4[3]) / K4[1] * z, z], 1)
    Xc[:40, 2] *= -1                                                       # behind the camera
    Xw = (Xc - t) @ R                                                      # R^T (Xc - t)
    Ow = -R.T @ t
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    normal = PO / dist[:, None]
    side = rng.random(n_pts) < 0.05
    normal[side] = np.roll(normal[side], 1, axis=1) * np.array([1, -1, 1])  # viewing angle test fails for most of these
    lvl = np.clip(kps["octave"][src] + rng.integers(0, 2, n_pts), 0, 7)
    dmax = dist * 1.2 ** (lvl - 0.5)
    dmin = dmax / 1.2 ** 7
    far = rng.random(n_pts) < 0.04
    dmax[far] = dist[far] * 0.5; dmin[far] = dmax[far] * 0.1               # outside the scale-invariance range
    bits = np.unpackbits(desc[src], axis=1)
    pdesc = np.packbits(bits ^ (rng.random(bits.shape) < 0.07), axis=1)
    f32 = lambda a: c(np.asarray(a, np.float32))
    return dict(kps=kps, desc=desc, N=N, sf=sf, isig=isig, K4=K4, T=T, S=S, n_pts=n_pts, Xw=f32(Xw), normal=f32(normal), dmin=f32(dmin), dmax=f32(dmax), pdesc=c(pdesc),
                rng=rng)


def _proj_out(n):
    return np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)


def test_fuse_M7_chi2_gate(rlib, frames):
    """ORBmatcher::Fuse(pKF, vpMapPoints, th) (ORBmatcher.cpp:854-993): the reference's own projection + PredictScale feed the oracle's window search; which
    feature each point was fused with is read back from what the reference did to the map (AddObservation/AddMapPoint or Replace)."""
    s = _projected_case(frames, 7)
    kps = s["kps"]
    has = (s["rng"].random(s["N"]) < 0.5).astype(np.uint8)
    for th in (3.0, 2.5):
        best = np.zeros(s["n_pts"], np.int32); valid, u, v, lvl = _proj_out(s["n_pts"])
        n = rlib.ref_fuse(_p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(s["desc"]), s["N"], *fb, _p(s["sf"]), _p(s["isig"]), _p(s["K4"]), _p(s["T"]), _p(has),
                          s["n_pts"], _p(s["Xw"]), _p(s["normal"]), _p(s["dmin"]), _p(s["dmax"]), _p(s["pdesc"]), C.c_float(th), _p(best), _p(valid), _p(u), _p(v), _p(lvl))
        assert 1500 < valid.sum() < s["n_pts"] - 100
        en, ebi, _, _ = oracle.projected_window_search(kps["x"], kps["y"], kps["octave"], s["desc"], BOUNDS, s["sf"], s["isig"], valid, u, v, lvl, s["pdesc"], th, True, 50)
        assert n == en and n > 800
        assert np.array_equal(best, ebi)
        # the gate matters in this scenario: without it the oracle accepts more
        en2, _, _, _ = oracle.projected_window_search(kps["x"], kps["y"], kps["octave"], s["desc"], BOUNDS, s["sf"], s["isig"], valid, u, v, lvl, s["pdesc"], th, False, 50)
        assert en2 > en


def _fuse_fan_out_case(frames, seed=17, S=8):
    """SearchInNeighbors' fan-out: S target keyframes around one pose (same features, poses a few centimetres / tenths of a degree apart, own sets of occupied features),
    a current keyframe whose points project into all of them; first-level neighbours 0..4 (in that order), second neighbours chosen so that the walk meets an already
    marked keyframe, the current keyframe itself, an unmarked LATER first-level neighbour (pushed twice) and keyframes that are second neighbours only."""
    s = _projected_case(frames, seed)
    rng = s["rng"]
    Ts = np.zeros((S, 4, 4), np.float32)
    for k in range(S):
        a = 0.3 + 0.004 * rng.normal()
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = np.array([0.4, -0.2, 0.7]) + 0.01 * rng.normal(size=3)
        Ts[k] = T.astype(np.float32)
    has = (rng.random((S, s["N"])) < 0.5).astype(np.uint8)
    nb1 = np.array([0, 1, 2, 3, 4], np.int32)
    nb2 = -np.ones((S, 5), np.int32)
    nb2[0, :4] = [-2, 1, 5, 6]      # the current keyframe (skipped), a later first-level neighbour (not marked yet: pushed here AND as first-level), two second-only
    nb2[1, :3] = [0, -2, 5]         # an already marked one (skipped), current, 5 again (second neighbours are never marked: pushed again)
    nb2[2, :2] = [7, 3]
    nb2[4, :1] = [6]
    return s, c(Ts), c(has), nb1, c(nb2)


def _run_fuse_fan_out(lib, s, Ts, has, nb1, nb2, max_calls=32):
    kps = s["kps"]; S = Ts.shape[0]; n = s["n_pts"]
    call_target = -np.ones(max_calls, np.int32); n_out = np.zeros(max_calls, np.int32); best = -np.ones((max_calls, n), np.int32)
    lib.ref_fuse_fan_out.restype = C.c_int
    nc = lib.ref_fuse_fan_out(_p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(s["desc"]), s["N"], *fb, _p(s["sf"]), _p(s["isig"]), _p(s["K4"]), S, _p(Ts), _p(has),
                              len(nb1), _p(nb1), _p(nb2), n, _p(s["Xw"]), _p(s["normal"]), _p(s["dmin"]), _p(s["dmax"]), _p(s["pdesc"]), C.c_float(3.0), max_calls,
                              _p(call_target), _p(n_out), _p(best))
    return nc, call_target[:nc], n_out[:nc], best[:nc]


def test_fuse_fan_out_of_search_in_neighbors_M7_xS(rlib, frames):
    """LocalMapping::SearchInNeighbors' first loop (Mapping.cpp:469-503) on the reference's own ORBmatcher.cpp: the target walk and a sequence of Fuse calls that change
    the map for each other.  Pins the ORDER of the targets (incl. the keyframe pushed twice) and that every call fuses a substantial, different set of points; the
    drop-in's prediction of this very sequence is compared with it call by call in tests/test_shim_matcher_gpu.py."""
    s, Ts, has, nb1, nb2 = _fuse_fan_out_case(frames)
    nc, tgt, n_out, best = _run_fuse_fan_out(rlib, s, Ts, has, nb1, nb2)
    assert tgt.tolist() == [0, 1, 5, 6, 1, 5, 2, 7, 3, 3, 4, 6]
    first_visits = [0, 1, 2, 3, 6, 7, 8, 10]
    assert n_out[0] > 800 and (n_out[first_visits] > 100).all()
    fused = best >= 0
    # (what a call did to the map identifies the fused feature of all but a handful of its points: a point added to a feature and displaced again inside the same call)
    assert (fused.sum(1) <= n_out).all() and (fused.sum(1) >= 0.98 * n_out).all()
    # a point fused into a keyframe is in it (or bad) afterwards: the second visit of keyframes 1, 5, 3 and 6 finds nothing left to fuse
    for first, second in ((1, 4), (2, 5), (8, 9), (3, 11)):
        assert not (fused[first] & fused[second]).any() and n_out[second] < n_out[first]
    # the calls change the map for each other: points replaced in an earlier call are bad in the later ones, so a later call on an untouched copy would fuse more
    assert fused[1:].sum() > 1000


def test_fuse_sim3_M8(rlib, frames):
    """ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cpp:995-1122): Sim3 decomposition, no chi2 gate, replacements returned."""
    s = _projected_case(frames, 8, sim3_scale=1.7)
    kps = s["kps"]
    has = (s["rng"].random(s["N"]) < 0.5).astype(np.uint8)
    best = np.zeros(s["n_pts"], np.int32); valid, u, v, lvl = _proj_out(s["n_pts"])
    th = 4.0
    n = rlib.ref_fuse_sim3(_p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(s["desc"]), s["N"], *fb, _p(s["sf"]), _p(s["isig"]), _p(s["K4"]), _p(s["S"]), _p(has),
                           s["n_pts"], _p(s["Xw"]), _p(s["normal"]), _p(s["dmin"]), _p(s["dmax"]), _p(s["pdesc"]), C.c_float(th), _p(best), _p(valid), _p(u), _p(v), _p(lvl))
    assert 1500 < valid.sum() < s["n_pts"] - 100
    en, ebi, _, _ = oracle.projected_window_search(kps["x"], kps["y"], kps["octave"], s["desc"], BOUNDS, s["sf"], s["isig"], valid, u, v, lvl, s["pdesc"], th, False, 50)
    assert n == en and n > 800
    # a feature without a map point takes the FIRST point fused with it (AddMapPoint); later points that pick the same feature find it occupied and are
    # returned as replacements of that first point — in both cases the feature index is the oracle's
    assert np.array_equal(best, ebi)


def test_search_by_projection_sim3_M9_claims(rlib, frames):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cpp:308-446): features matched on entry are skipped, every accepted point
    claims its feature for the points after it; the CCM branch for points the keyframe already observes does not count."""
    s = _projected_case(frames, 9, sim3_scale=0.8)
    kps = s["kps"]; N = s["N"]
    rng = s["rng"]
    matched0 = -np.ones(N, np.int32); matched0[rng.integers(0, N, 120)] = 1_000_000
    existing = -np.ones(s["n_pts"], np.int32)
    already = rng.choice(s["n_pts"], 150, replace=False)
    free = np.flatnonzero(matched0 < 0)
    existing[already] = rng.choice(free, 150, replace=False)          # points the keyframe already observes (at some other feature)
    no_claim = (existing >= 0).astype(np.uint8)
    for th in (10, 4):
        got = matched0.copy(); valid, u, v, lvl = _proj_out(s["n_pts"])
        n = rlib.ref_search_by_projection_sim3(_p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(s["desc"]), N, *fb, _p(s["sf"]), _p(s["isig"]), _p(s["K4"]),
                                               _p(s["S"]), s["n_pts"], _p(s["Xw"]), _p(s["normal"]), _p(s["dmin"]), _p(s["dmax"]), _p(s["pdesc"]), _p(existing), th,
                                               _p(got), _p(valid), _p(u), _p(v), _p(lvl))
        en, ebi, _, em = oracle.projected_window_search(kps["x"], kps["y"], kps["octave"], s["desc"], BOUNDS, s["sf"], s["isig"], valid, u, v, lvl, s["pdesc"], float(th),
                                                        False, 50, matched=matched0, claim=True, no_claim=no_claim)
        # an already-observed point that finds a feature is re-mapped inside the keyframe (:412-432): no claim, and the reference does not count it
        remapped = int(((ebi >= 0) & (no_claim > 0)).sum())
        assert remapped > 30 and n == en - remapped and n > 600
        assert np.array_equal(got, em)
        # claims are order dependent: an order-free search (no claims) accepts more points
        en_free, _, _, _ = oracle.projected_window_search(kps["x"], kps["y"], kps["octave"], s["desc"], BOUNDS, s["sf"], s["isig"], valid, u, v, lvl, s["pdesc"], float(th),
                                                          False, 50, matched=matched0, claim=False)
        assert en_free > en


def test_search_by_sim3_M10(rlib, frames):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cpp:1124-1348): two keyframes whose maps differ by a similarity (s12, R12, t12); both directional searches (no chi2
    gate, TH_HIGH, level window) and the agreement step.  The oracle side = two calls of its projected window search on the reference's own projections +
    the mutual check."""
    kps, desc = frames[0]
    N = len(kps)
    rng = np.random.default_rng(10)
    sf, _, _, isig = synth.scale_tables()
    K4 = np.array(synth.EUROC_K, np.float32)

    def rot(ax, a):
        ax = np.asarray(ax, float) / np.linalg.norm(ax)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx

    R1, t1 = rot([0, 1, 0], 0.2), np.array([0.1, 0.0, 0.3])
    R2, t2 = rot([1, 0, 0], -0.4), np.array([-2.0, 1.0, 0.5])
    s12, R12, t12 = 1.15, rot([0.2, 1, 0.1], 0.03), np.array([0.05, -0.02, 0.1])
    sR21 = (1.0 / s12) * R12.T
    t21 = -sR21 @ t12
    z = rng.uniform(3, 9, N)
    Xc1 = np.stack([(kps["x"] - K4[2]) / K4[0] * z, (kps["y"] - K4[3]) / K4[1] * z, z], 1)          # the scene in camera-1 coordinates
    Xc2 = Xc1 @ sR21.T + t21
    # keyframe 2 sees the same scene: its features are the projections (jittered, shuffled), its descriptors noisy copies
    perm = rng.permutation(N)
    u2 = K4[0] * Xc2[:, 0] / Xc2[:, 2] + K4[2] + rng.normal(0, 1.0, N)
    v2 = K4[1] * Xc2[:, 1] / Xc2[:, 2] + K4[3] + rng.normal(0, 1.0, N)
    outside = (u2 < 1) | (u2 > 750) | (v2 < 1) | (v2 > 478)
    u2[outside] = rng.uniform(1, 750, outside.sum()); v2[outside] = rng.uniform(1, 478, outside.sum())
    f32 = lambda a: c(np.asarray(a, np.float32))
    kx2, ky2, oc2 = f32(u2[perm]), f32(v2[perm]), c(kps["octave"][perm])
    bits = np.unpackbits(desc, axis=1)
    desc2 = c(np.packbits(bits ^ (rng.random(bits.shape) < 0.06), axis=1)[perm])
    # map points: keyframe 1's in world 1, keyframe 2's in world 2 (where keyframe 2's features back-project)
    Xw1 = (Xc1 + rng.normal(0, 0.004, Xc1.shape) - t1) @ R1
    Xc2p = np.stack([(kx2 - K4[2]) / K4[0], (ky2 - K4[3]) / K4[1], np.ones(N)], 1) * Xc2[perm][:, 2:3]
    Xw2 = (Xc2p - t2) @ R2
    lvl1 = np.clip(kps["octave"] + rng.integers(0, 2, N), 0, 7); lvl2 = np.clip(oc2 + rng.integers(0, 2, N), 0, 7)
    d1 = np.linalg.norm(Xc2, axis=1); d2 = np.linalg.norm(Xc2p @ (s12 * R12).T + t12, axis=1)           # distances in the TARGET camera
    dmax1 = d1 * 1.2 ** (lvl1 - 0.5); dmax2 = d2 * 1.2 ** (lvl2 - 0.5)
    dmax1[:30] = d1[:30] * 0.3                                                                           # out of range
    mdesc1 = c(np.packbits(bits ^ (rng.random(bits.shape) < 0.03), axis=1))
    mdesc2 = c(np.packbits(np.unpackbits(desc2, axis=1) ^ (rng.random(bits.shape) < 0.03), axis=1))
    has1 = (rng.random(N) < 0.8).astype(np.uint8); has2 = (rng.random(N) < 0.8).astype(np.uint8)
    inv = np.empty(N, np.int64); inv[perm] = np.arange(N)                                                # feature j of KF1 is feature inv[j] of KF2
    pre12 = -np.ones(N, np.int32)
    pre = rng.choice(np.flatnonzero(has2[inv] > 0), 60, replace=False)
    pre12[pre] = inv[pre]
    T1 = np.eye(4, dtype=np.float32); T1[:3, :3] = R1; T1[:3, 3] = t1
    T2 = np.eye(4, dtype=np.float32); T2[:3, :3] = R2; T2[:3, 3] = t2
    out12 = np.zeros(N, np.int32)
    va1, pu1, pv1, pl1 = _proj_out(N); va2, pu2, pv2, pl2 = _proj_out(N)
    th = 7.5
    n = rlib.ref_search_by_sim3(_p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(desc), N, _p(T1), _p(has1), _p(f32(Xw1)), _p(f32(dmax1 / 1.2 ** 7)),
                                _p(f32(dmax1)), _p(mdesc1), _p(kx2), _p(ky2), _p(oc2), _p(desc2), N, _p(T2), _p(has2), _p(f32(Xw2)), _p(f32(dmax2 / 1.2 ** 7)),
                                _p(f32(dmax2)), _p(mdesc2), *fb, _p(sf), _p(isig), _p(K4), C.c_float(s12), _p(f32(R12)), _p(f32(t12)), C.c_float(th), _p(pre12), _p(out12),
                                _p(va1), _p(pu1), _p(pv1), _p(pl1), _p(va2), _p(pu2), _p(pv2), _p(pl2))
    assert va1.sum() > 500 and va2.sum() > 500
    src1 = va1.copy(); src1[pre12 >= 0] = 0                       # vbAlreadyMatched1
    src2 = va2.copy(); src2[pre12[pre12 >= 0]] = 0                # vbAlreadyMatched2
    _, vn1, _, _ = oracle.projected_window_search(kx2, ky2, oc2, desc2, BOUNDS, sf, isig, src1, pu1, pv1, pl1, mdesc1, th, False, 100)
    _, vn2, _, _ = oracle.projected_window_search(kps["x"], kps["y"], kps["octave"], desc, BOUNDS, sf, isig, src2, pu2, pv2, pl2, mdesc2, th, False, 100)
    exp = pre12.copy()
    i1 = np.flatnonzero(vn1 >= 0)
    agree = i1[vn2[vn1[i1]] == i1]
    exp[agree] = vn1[agree]
    assert n == agree.size and n > 300
    assert np.array_equal(out12, exp)
    # the matches found are the true correspondences of the construction
    assert (exp[agree] == inv[agree]).mean() > 0.98


def _write_orbvoc_txt(path, vocab, k):
    """the ORBvoc.txt text format TemplatedVocabulary::loadFromTextFile reads (TemplatedVocabulary.h:1347-1431): header 'k L scoring weighting' (L1_NORM = 0,
    TF_IDF = 0), then one line per non-root node in node order: parent id, is-leaf, the 32 descriptor bytes, the weight"""
    n = vocab["n_nodes"]
    with open(path, "w") as f:
        f.write(f"{k} {vocab['L']} 0 0\n")
        lines = []
        for i in range(1, n):
            leaf = int(vocab["child_off"][i + 1] == vocab["child_off"][i])
            lines.append(f"{(i - 1) // k} {leaf} " + " ".join(str(int(b)) for b in vocab["node_desc"][i]) + f" {float(vocab['weight'][i])!r}")
        f.write("\n".join(lines))          # no trailing newline: the loader would read an empty line as one more node


def test_bow_transform_against_the_vendored_dbow2(rlib, frames, tmp_path):
    """KeyFrame::ComputeBoW (KeyFrame.cpp:277-286) on the reference's DBoW2 (TemplatedVocabulary::transform, FORB::distance, BowVector::addWeight / normalize,
    FeatureVector::addFeature) vs oracle.bow_transform: BowVector (ids, f64 values bit-exact) and FeatureVector (nodes at levelsup, feature lists)."""
    k, L = 10, 4
    vocab = synth.make_vocabulary(k, L, seed=3)
    path = str(tmp_path / "voc.txt")
    _write_orbvoc_txt(path, vocab, k)
    rlib.ref_vocab_load_text.restype = C.c_void_p
    h = rlib.ref_vocab_load_text(path.encode())
    assert h and rlib.ref_vocab_size(C.c_void_p(h)) == k ** L
    try:
        for (kps, desc), levelsup in ((frames[0], 4), (frames[1], 2), (frames[0], 6)):
            N = len(kps)
            ids = np.zeros(N, np.int32); vals = np.zeros(N, np.float64); nfv = np.zeros(1, np.int32)
            fvn = np.zeros(N, np.int32); fvo = np.zeros(N + 1, np.int32); fvf = np.zeros(N, np.int32)
            n = rlib.ref_bow_transform(C.c_void_p(h), _p(desc), N, levelsup, _p(ids), _p(vals), _p(nfv), _p(fvn), _p(fvo), _p(fvf))
            word, w, node, bid, bval = oracle.bow_transform(vocab, desc, levelsup)
            assert n == len(bid) and n > 300
            assert np.array_equal(ids[:n], bid) and np.array_equal(vals[:n], bval)           # f64 bit-exact (same accumulation + normalisation order)
            # FeatureVector: features with non-stopped words, grouped by node, ascending feature index
            live = np.flatnonzero(w > 0)
            assert 0 < live.size < N                                                         # the synthetic vocabulary has stopped words
            nodes = np.unique(node[live])
            assert nfv[0] == nodes.size and np.array_equal(fvn[:nfv[0]], nodes)
            for j, nd in enumerate(nodes):
                assert np.array_equal(fvf[fvo[j]:fvo[j + 1]], live[node[live] == nd])
            if levelsup >= L:
                assert nodes.size == 1 and nodes[0] == 0                                     # nid_level <= 0: everything under the root
    finally:
        rlib.ref_vocab_free(C.c_void_p(h))


def test_distinctive_descriptors_against_the_references_mappoint(rlib):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cpp:929-994, the reference's own lines) vs oracle.distinctive_descriptors: least-median rule incl. the
    0.5*(N-1) median index and first-minimum tie-break."""
    rng = np.random.default_rng(5)
    P = 600
    n = rng.integers(0, 14, P); n[:5] = (0, 1, 2, 3, 4)
    off = np.zeros(P + 1, np.int32); off[1:] = np.cumsum(n)
    base = rng.integers(0, 2, (P, 256), dtype=np.uint8)
    bits = np.repeat(base, n, axis=0) ^ (rng.random((off[-1], 256)) < 0.15)
    # coarse noise on some points so that medians tie often
    coarse = np.repeat(rng.random(P) < 0.3, n)
    bits[coarse] = np.repeat(base, n, axis=0)[coarse] ^ np.repeat(rng.random((off[-1], 8)) < 0.3, 32, axis=1)[coarse]
    desc = c(np.packbits(bits, axis=1))
    got = np.zeros(P, np.int32)
    rlib.ref_distinctive_descriptors(_p(desc), _p(off), P, _p(got))
    exp = oracle.distinctive_descriptors(desc, off)
    assert (got[n == 0] == -1).all()
    for p in np.flatnonzero(n > 0):                               # equal descriptors (coarse noise) are interchangeable: compare what the point ends up with
        assert np.array_equal(desc[off[p] + got[p]], desc[off[p] + exp[p]]), p
    assert (got[n > 0] == exp[n > 0]).mean() > 0.7


def test_is_in_frustum_against_the_references_frame(rlib):
    """Frame::isInFrustum (Frame.cpp:139-198, the reference's own lines; pose matrices by its UpdatePoseMatrices) vs oracle.is_in_frustum on 20 000 points all
    around the camera: flags, projection, predicted level and viewing cosine bit-exact."""
    for seed in (0, 1):
        rng = np.random.default_rng(seed)
        R, t, _ = synth._agent_loop(40, 0)
        R, t = R[5 + seed].astype(np.float32), t[5 + seed].astype(np.float32)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = R; T[:3, 3] = t
        K4 = np.array(synth.EUROC_K, np.float32)
        b = np.array([-135.79564, 895.5073, -92.875015, 565.5531], np.float32)   # minX maxX minY maxY of the EuRoC camera
        n = 20000
        Ow64 = -(R.T.astype(np.float64) @ t.astype(np.float64))
        P = (Ow64 + rng.normal(size=(n, 3)) * 6).astype(np.float32)
        view = P - Ow64
        normal = view / np.linalg.norm(view, axis=1, keepdims=True)
        normal = normal + rng.normal(size=(n, 3)) * rng.choice([0.05, 0.8], (n, 1))
        normal = c((normal / np.linalg.norm(normal, axis=1, keepdims=True)).astype(np.float32))
        d = np.linalg.norm(view, axis=1)
        dmax = (d * rng.uniform(0.6, 4.0, n)).astype(np.float32)
        dmin = (dmax / np.float32(1.2 ** 7)).astype(np.float32)
        inv = np.zeros(n, np.uint8); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32); lvl = np.zeros(n, np.int32); cs = np.zeros(n, np.float32)
        Ow = np.zeros(3, np.float32)
        rlib.ref_is_in_frustum(_p(T), _p(K4), C.c_float(b[0]), C.c_float(b[2]), C.c_float(b[1]), C.c_float(b[3]), C.c_float(np.float32(1.2)), 8, n, _p(P), _p(normal),
                               _p(dmin), _p(dmax), C.c_float(0.5), _p(inv), _p(u), _p(v), _p(lvl), _p(cs), _p(Ow))
        assert np.abs(Ow - Ow64).max() < 1e-5
        frame24 = np.concatenate([R.ravel(), t, Ow, K4, b, [np.float32(np.log(np.float32(1.2)))]]).astype(np.float32)
        einv, eu, ev, elvl, ecs = oracle.is_in_frustum(frame24, 8, P, normal, dmin, dmax, 0.5)
        assert 500 < inv.sum() < n // 2
        assert np.array_equal(inv, einv)
        m = inv > 0
        assert np.array_equal(u[m], eu[m]) and np.array_equal(v[m], ev[m]) and np.array_equal(lvl[m], elvl[m]) and np.array_equal(cs[m], ecs[m])
