"""GPU parity of the C++ host mirror (ccm_slam_amd/host/, libccm_host.so): ORBmatcher::SearchByProjection
(both overloads) with the device Hamming kernel + ordered host resolution vs the oracle's sequential
restatement — identical match tables (integer work, bit-exact); LocalBundleAdjustmentClient two-stage
protocol vs the oracle; ORBextractor class vs the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from ccm_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib():
    lib = C.CDLL(os.path.join(ROOT, "ccm_slam_amd", "libccm_host.so"))
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _frame(oracle_lib, seed, t):
    kps, desc = oracle_lib.OrbOracle(1000).extract(synth.gen_image(seed, t))
    return kps, desc


def test_search_by_projection_map_points(hostlib, oracle_lib):
    kps, desc = _frame(oracle_lib, 1000, 0)
    N = len(kps)
    rng = np.random.default_rng(0)
    sf = synth.scale_tables()[0]
    # 3000 "local map points": most project near an existing feature and carry a noisy copy of its descriptor
    n_mp = 3000
    src = rng.integers(0, N, n_mp)
    px = (kps["x"][src] + rng.normal(0, 2.0, n_mp)).astype(np.float32)
    py = (kps["y"][src] + rng.normal(0, 2.0, n_mp)).astype(np.float32)
    lvl = np.clip(kps["octave"][src] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    vcos = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    in_view = (rng.random(n_mp) < 0.9).astype(np.uint8)
    mp_desc = desc[src].copy()
    bits = np.unpackbits(mp_desc, axis=1)
    flip = rng.random(bits.shape) < 0.08
    mp_desc = np.packbits(bits ^ flip, axis=1)
    frame_mp0 = -np.ones(N, np.int32)
    frame_mp0[rng.integers(0, N, 50)] = 10_000   # features already holding an observed map point
    bounds = (0.0, 0.0, 752.0, 480.0)
    exp_n, exp = oracle_lib.search_by_projection_mp(kps["x"], kps["y"], kps["octave"], desc, bounds, sf, in_view, px, py, lvl, vcos,
                                                    mp_desc, 3.0, 0.8, frame_mp0)
    got = frame_mp0.copy()
    kx, ky, oc = (np.ascontiguousarray(kps[f]) for f in ("x", "y", "octave"))
    n = hostlib.ccmh_search_by_projection_mp(0, _p(kx), _p(ky), _p(oc), _p(desc), N, *[C.c_float(b) for b in bounds], _p(sf), n_mp,
                                             _p(in_view), _p(px), _p(py), _p(lvl), _p(vcos), _p(mp_desc), C.c_float(3.0), C.c_float(0.8),
                                             _p(got))
    assert n == exp_n and n > 500
    assert np.array_equal(got, exp)


def test_search_by_projection_last_frame(hostlib, oracle_lib):
    k1, d1 = _frame(oracle_lib, 1000, 0)
    k2, d2 = _frame(oracle_lib, 1000, 1)   # scene shifted by one pixel
    rng = np.random.default_rng(1)
    sf = synth.scale_tables()[0]
    n_last = len(k1)
    valid = (rng.random(n_last) < 0.6).astype(np.uint8)
    u = (k1["x"] - np.float32(1.0) + rng.normal(0, 1.0, n_last)).astype(np.float32)
    v = (k1["y"] - np.float32(1.0) + rng.normal(0, 1.0, n_last)).astype(np.float32)
    bounds = (0.0, 0.0, 752.0, 480.0)
    cur0 = -np.ones(len(k2), np.int32)
    args = (k2["x"], k2["y"], k2["octave"], k2["angle"], d2, bounds, sf, valid, u, v, k1["octave"], k1["angle"], d1)
    exp_n, exp = oracle_lib.search_by_projection_last(*args, 7.0, 1, cur0)
    got = cur0.copy()
    kx, ky, oc, ka = (np.ascontiguousarray(k2[f]) for f in ("x", "y", "octave", "angle"))
    lo, la = np.ascontiguousarray(k1["octave"]), np.ascontiguousarray(k1["angle"])
    n = hostlib.ccmh_search_by_projection_last(0, _p(kx), _p(ky), _p(oc), _p(ka), _p(d2), len(k2), *[C.c_float(b) for b in bounds], _p(sf),
                                               n_last, _p(valid), _p(u), _p(v), _p(lo), _p(la), _p(d1), C.c_float(7.0), 1, _p(got))
    assert n == exp_n and n > 200
    assert np.array_equal(got, exp)
    # the same search with the grid, the candidate lists and the distances produced on the device
    got2 = cur0.copy()
    n2 = hostlib.ccmh_search_by_projection_last_dev(0, _p(np.ascontiguousarray(k2)), _p(d2), len(k2), 752, 480, _p(sf), n_last, _p(valid), _p(u), _p(v),
                                                    _p(lo), _p(la), _p(d1), C.c_float(7.0), 1, _p(got2))
    assert n2 == exp_n and np.array_equal(got2, exp)


def test_local_ba_client_two_stage(hostlib, oracle_lib):
    prob = synth.make_ba_config("lba_c2")
    cam = prob["cam_qt"].copy(); pts = prob["pt_xyz"].copy()
    erase = np.zeros(prob["n_edge"], np.uint8)
    rc = hostlib.ccmh_local_ba(0, prob["n_cam"], prob["n_pt"], prob["n_edge"], _p(cam), _p(prob["cam_fixed"]), _p(prob["cam_K"]), _p(pts),
                               _p(prob["e_cam"]), _p(prob["e_pt"]), _p(np.ascontiguousarray(prob["e_obs"])), _p(prob["e_info"]), _p(erase))
    assert rc == 0
    p1 = dict(prob); p1["huber_delta"] = float(np.float32(np.sqrt(np.float32(5.991))))
    ocam, opts, ochi2, odpos, _ = oracle_lib.ba_optimize(p1, 5)
    level = np.zeros(prob["n_edge"], np.uint8); level[(ochi2 > 5.991) | (odpos == 0)] = 1
    p2 = dict(prob); p2.update(cam_qt=ocam, pt_xyz=opts, e_level=level, huber_delta=0.0)
    ocam2, opts2, ochi2b, odpos2, _ = oracle_lib.ba_optimize(p2, 10, chi2_in=ochi2)
    dt, dr = synth.pose_errors(cam, ocam2)
    assert dt.max() <= 1e-5 and dr.max() <= 1e-4
    assert (erase != ((ochi2b > 5.991) | (odpos2 == 0))).sum() <= 2


def test_orb_extractor_class(hostlib, oracle_lib):
    img = synth.gen_image(1002, 4)
    okps, odesc = oracle_lib.OrbOracle(1000).extract(img)
    kps = np.zeros(1100, oracle_lib.KP_DTYPE); desc = np.zeros((1100, 32), np.uint8)
    n = hostlib.ccmh_orb_extract(0, 1000, _p(img), 752, 480, _p(kps), _p(desc), 1100)
    assert n == len(okps) and np.array_equal(desc[:n], odesc) and np.array_equal(kps[:n], okps)


# ---- BoW-bucketed, triangulation and initialisation searches (M3-M6) ------------------------------------
def _feature_vector(desc, seed, n_nodes=60):
    """synthetic DBoW2 FeatureVector: node = f(first descriptor bits) so that similar descriptors mostly share a
    node; returns (node ids ascending, CSR offsets, feature indices in insertion order)"""
    node_of = (desc[:, 0].astype(int) * 7 + (desc[:, 1] >> 5) + seed) % n_nodes
    nodes = np.unique(node_of)
    off = [0]
    idx = []
    for nd in nodes:
        m = np.nonzero(node_of == nd)[0]
        idx.extend(m.tolist())
        off.append(len(idx))
    return nodes.astype(np.int32), np.array(off, np.int32), np.array(idx, np.int32)


def _two_frames(oracle_lib):
    k1, d1 = _frame(oracle_lib, 1000, 0)
    k2, d2 = _frame(oracle_lib, 1000, 1)
    return k1, d1, k2, d2


def _bow_args(k1, d1, k2, d2, fv1, fv2, has1, has2):
    c = np.ascontiguousarray
    return [_p(fv1[0]), _p(fv1[1]), _p(fv1[2]), fv1[0].size, _p(fv2[0]), _p(fv2[1]), _p(fv2[2]), fv2[0].size, _p(has1), _p(has2), _p(d1),
            _p(c(k1["x"])), _p(c(k1["y"])), _p(c(k1["angle"])), len(k1), _p(d2), _p(c(k2["x"])), _p(c(k2["y"])), _p(c(k2["octave"])),
            _p(c(k2["angle"])), len(k2)]


def test_search_by_bow_kf_frame_and_kf_kf(hostlib, oracle_lib):
    k1, d1, k2, d2 = _two_frames(oracle_lib)
    rng = np.random.default_rng(3)
    fv1, fv2 = _feature_vector(d1, 0), _feature_vector(d2, 0)
    has1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    has2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    zero9 = np.zeros(9, np.float32); sf = synth.scale_tables()[0]; s2 = synth.scale_tables()[2]
    # KF -> Frame (<= TH_LOW, ratio 0.7)
    exp_n, exp = oracle_lib.search_by_bow_kf_frame(fv1, fv2, has1, d1, k1["angle"], d2, k2["angle"], 0.7, 1)
    got = np.zeros(len(k2), np.int32)
    n = hostlib.ccmh_search_bow(0, 0, *_bow_args(k1, d1, k2, d2, fv1, fv2, has1, has2), _p(zero9), C.c_float(0), C.c_float(0), _p(s2), _p(sf),
                                C.c_float(0.7), 1, _p(got))
    assert n == exp_n and n > 100 and np.array_equal(got, exp)
    # KF -> KF (strict < TH_LOW, both sides need map points)
    exp_n, exp = oracle_lib.search_by_bow_kf_kf(fv1, fv2, has1, has2, d1, k1["angle"], d2, k2["angle"], 0.75, 1)
    got = np.zeros(len(k1), np.int32)
    n = hostlib.ccmh_search_bow(0, 1, *_bow_args(k1, d1, k2, d2, fv1, fv2, has1, has2), _p(zero9), C.c_float(0), C.c_float(0), _p(s2), _p(sf),
                                C.c_float(0.75), 1, _p(got))
    assert n == exp_n and n > 50 and np.array_equal(got, exp)


def test_search_for_triangulation(hostlib, oracle_lib):
    k1, d1, k2, d2 = _two_frames(oracle_lib)
    rng = np.random.default_rng(4)
    fv1, fv2 = _feature_vector(d1, 0), _feature_vector(d2, 0)
    has1 = (rng.random(len(k1)) < 0.4).astype(np.uint8)
    has2 = (rng.random(len(k2)) < 0.4).astype(np.uint8)
    # pure x-translation between the views: F = [t]_x, epipolar lines are horizontal; epipole far to the right
    F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)
    sf, _, s2, _ = synth.scale_tables()
    exp_n, exp = oracle_lib.search_for_triangulation(fv1, fv2, has1, has2, d1, k1["x"], k1["y"], k1["angle"], d2, k2["x"], k2["y"], k2["octave"],
                                                     k2["angle"], F12, 5000.0, 240.0, s2, sf, 0)
    got = np.zeros(len(k1), np.int32)
    n = hostlib.ccmh_search_bow(0, 2, *_bow_args(k1, d1, k2, d2, fv1, fv2, has1, has2), _p(F12), C.c_float(5000.0), C.c_float(240.0), _p(s2), _p(sf),
                                C.c_float(0.6), 0, _p(got))
    assert n == exp_n and n > 50 and np.array_equal(got, exp)


def test_search_for_triangulation_fan_out_of_twenty_neighbours_in_one_launch(hostlib, oracle_lib):
    """LocalMapping::CreateNewMapPoints (Mapping.cpp:277-470) calls SearchForTriangulation for up to 20 covisible neighbours of a new keyframe and creates
    map points between the calls.  TriangulationBatch sends the Hamming work of ALL neighbours out as one ccm_hamming_csr_multi launch; resolve(j) with the
    map-point flags as they are at call j must return exactly what the oracle's sequential restatement returns for that call — including the features of
    keyframe 1 that gained a map point from an earlier neighbour's matches and the candidates of keyframe 2 that did."""
    import ctypes as C
    k1, d1 = _frame(oracle_lib, 1000, 0)
    nbs = [_frame(oracle_lib, 1000, 1 + (j % 5)) if j < 5 else None for j in range(20)]
    ext = oracle_lib.OrbOracle(1000)
    for j in range(5, 20):                                      # 15 more neighbours: other frames of the stream, other seeds
        nbs[j] = ext.extract(synth.gen_image(1000 + j // 5, j % 5))
    rng = np.random.default_rng(14)
    fv1 = _feature_vector(d1, 0)
    fvs = [_feature_vector(d, 0) for _, d in nbs]
    has1 = (rng.random(len(k1)) < 0.4).astype(np.uint8)
    has2 = [(rng.random(len(k)) < 0.4).astype(np.uint8) for k, _ in nbs]
    sf, _, s2, _ = synth.scale_tables()
    c = np.ascontiguousarray
    keep = []                                                   # arrays the pointer tables refer to

    def ptrs(arrs, ty=C.c_void_p):
        arrs = [c(a) for a in arrs]
        keep.extend(arrs)
        return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    n_nb = len(nbs)
    nn2 = np.array([f[0].size for f in fvs], np.int32); N2 = np.array([len(k) for k, _ in nbs], np.int32)
    hostlib.ccmh_tri_batch_create.restype = C.c_void_p
    hostlib.ccmh_tri_batch_candidates.restype = C.c_longlong
    hostlib.ccmh_tri_batch_candidates.argtypes = [C.c_void_p]
    hostlib.ccmh_tri_batch_destroy.argtypes = [C.c_void_p]
    hostlib.ccmh_tri_batch_destroy.restype = None
    hostlib.ccmh_tri_batch_resolve.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_float, C.c_float] + [C.c_void_p] * 3
    x1, y1, a1 = c(k1["x"]), c(k1["y"]), c(k1["angle"])
    h = hostlib.ccmh_tri_batch_create(0, C.c_float(0.6), 0, _p(fv1[0]), _p(fv1[1]), _p(fv1[2]), fv1[0].size, _p(has1), _p(d1), _p(x1), _p(y1), _p(a1), len(k1), n_nb,
                                      ptrs([f[0] for f in fvs]), ptrs([f[1] for f in fvs]), ptrs([f[2] for f in fvs]), _p(nn2), ptrs(has2), ptrs([d for _, d in nbs]),
                                      ptrs([k["x"] for k, _ in nbs]), ptrs([k["y"] for k, _ in nbs]), ptrs([k["octave"] for k, _ in nbs]), ptrs([k["angle"] for k, _ in nbs]), _p(N2))
    assert h, "ccmh_tri_batch_create failed"
    assert hostlib.ccmh_tri_batch_candidates(h) > 100_000       # ONE launch carried all of them
    has1_now = has1.copy()
    total = 0
    for j, ((k2, d2), fv2) in enumerate(zip(nbs, fvs)):
        F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32) * np.float32(1 + 0.01 * j)
        ex = 5000.0 - 100.0 * j
        has2_now = has2[j].copy()
        if j % 3 == 2:
            has2_now[rng.integers(0, len(k2), 40)] = 1          # the neighbour gained points too (it was another keyframe's neighbour)
        exp_n, exp = oracle_lib.search_for_triangulation(fv1, fv2, has1_now, has2_now, d1, k1["x"], k1["y"], k1["angle"], d2, k2["x"], k2["y"], k2["octave"],
                                                         k2["angle"], F12, ex, 240.0, s2, sf, 0)
        got = np.zeros(len(k1), np.int32)
        n = hostlib.ccmh_tri_batch_resolve(h, j, _p(has1_now), _p(has2_now), _p(F12), C.c_float(ex), C.c_float(240.0), _p(s2), _p(sf), _p(got))
        assert n == exp_n and np.array_equal(got, exp), j
        # ... and what the single call returns (its own launch)
        single = np.zeros(len(k1), np.int32)
        ns = hostlib.ccmh_search_bow(0, 2, *_bow_args(k1, d1, k2, d2, fv1, fv2, has1_now, has2_now), _p(F12), C.c_float(ex), C.c_float(240.0), _p(s2), _p(sf),
                                     C.c_float(0.6), 0, _p(single))
        assert ns == n and np.array_equal(single, got), j
        total += n
        has1_now[got >= 0] = 1                                  # Mapping.cpp:437-452: the matches are triangulated into new map points of keyframe 1
    hostlib.ccmh_tri_batch_destroy(h)
    assert total > 150


def test_search_for_initialization(hostlib, oracle_lib):
    k1, d1 = oracle_lib.OrbOracle(2000).extract(synth.gen_image(1000, 0))
    k2, d2 = oracle_lib.OrbOracle(2000).extract(synth.gen_image(1000, 2))
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32).ravel()
    bounds = (0.0, 0.0, 752.0, 480.0)
    exp_n, exp, exp_prev = oracle_lib.search_for_initialization(k1, d1, k2, d2, bounds, prev, 100, 0.9, 1)
    got = np.zeros(len(k1), np.int32); gprev = prev.copy()
    c = np.ascontiguousarray
    n = hostlib.ccmh_search_for_initialization(0, _p(c(k1["x"])), _p(c(k1["y"])), _p(c(k1["octave"])), _p(c(k1["angle"])), _p(d1), len(k1),
                                               _p(c(k2["x"])), _p(c(k2["y"])), _p(c(k2["octave"])), _p(c(k2["angle"])), _p(d2), len(k2),
                                               *[C.c_float(b) for b in bounds], _p(gprev), 100, C.c_float(0.9), 1, _p(got))
    assert n == exp_n and n > 100
    assert np.array_equal(got, exp) and np.array_equal(gprev, exp_prev)


# ---- projected window searches: Fuse, Fuse(Sim3), SearchByProjection(kfptr,Scw,..), SearchBySim3 (M7-M10) ----
@pytest.mark.parametrize("name,chi2_gate,thr,use_matched,claim", [("fuse", 1, 50, 0, 0), ("fuse_sim3", 0, 50, 0, 0),
                                                                 ("search_by_projection_sim3", 0, 50, 1, 1), ("search_by_sim3", 0, 100, 0, 0)])
def test_projected_window_search(hostlib, oracle_lib, name, chi2_gate, thr, use_matched, claim):
    kps, desc = _frame(oracle_lib, 1001, 0)
    N = len(kps)
    rng = np.random.default_rng(7)
    sf, _, _, is2 = synth.scale_tables()
    n_pts = 2500
    src = rng.integers(0, N, n_pts)
    u = (kps["x"][src] + rng.normal(0, 1.5, n_pts)).astype(np.float32)
    v = (kps["y"][src] + rng.normal(0, 1.5, n_pts)).astype(np.float32)
    level = np.clip(kps["octave"][src] + rng.integers(0, 2, n_pts), 0, 7).astype(np.int32)
    valid = (rng.random(n_pts) < 0.85).astype(np.uint8)
    pdesc = desc[src].copy()
    bits = np.unpackbits(pdesc, axis=1)
    pdesc = np.packbits(bits ^ (rng.random(bits.shape) < 0.06), axis=1)
    matched0 = -np.ones(N, np.int32)
    matched0[rng.integers(0, N, 80)] = 99_999
    no_claim = (rng.random(n_pts) < 0.1).astype(np.uint8)
    bounds = (0.0, 0.0, 752.0, 480.0)
    th = 3.0 if name.startswith("fuse") else 7.5
    exp_n, ebi, ebd, em = oracle_lib.projected_window_search(kps["x"], kps["y"], kps["octave"], desc, bounds, sf, is2, valid, u, v, level, pdesc, th,
                                                             chi2_gate, thr, matched0 if use_matched else None, claim, no_claim)
    c = np.ascontiguousarray
    gbi = np.zeros(n_pts, np.int32); gbd = np.zeros(n_pts, np.int32)
    gm = matched0.copy()
    n = hostlib.ccmh_projected_window_search(0, _p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(desc), N, *[C.c_float(b) for b in bounds],
                                             _p(sf), _p(is2), n_pts, _p(valid), _p(u), _p(v), _p(level), _p(pdesc), C.c_float(th), chi2_gate, thr,
                                             _p(gm) if use_matched else None, claim, _p(no_claim), _p(gbi), _p(gbd))
    assert n == exp_n and n > 300
    assert np.array_equal(gbi, ebi) and np.array_equal(gbd, ebd)
    if use_matched:
        assert np.array_equal(gm, em)
    # same call with the keyframe grid, candidate lists and distances produced on the device
    gbi2 = np.zeros(n_pts, np.int32); gbd2 = np.zeros(n_pts, np.int32)
    gm2 = matched0.copy()
    n2 = hostlib.ccmh_projected_window_search_dev(0, _p(c(kps["x"])), _p(c(kps["y"])), _p(c(kps["octave"])), _p(desc), N, C.c_float(752.0), C.c_float(480.0),
                                                  _p(sf), _p(is2), n_pts, _p(valid), _p(u), _p(v), _p(level), _p(pdesc), C.c_float(th), chi2_gate, thr,
                                                  _p(gm2) if use_matched else None, claim, _p(no_claim), _p(gbi2), _p(gbd2))
    assert n2 == exp_n and np.array_equal(gbi2, ebi) and np.array_equal(gbd2, ebd)
    if use_matched:
        assert np.array_equal(gm2, em)


def test_vocabulary_transform_class(hostlib, oracle_lib):
    """cslam::ORBVocabulary::transform (host mirror): BowVector and FeatureVector identical to the oracle's restatement of
    TemplatedVocabulary::transform (same f64 accumulation order => identical doubles)."""
    vocab = synth.make_vocabulary(10, 4, seed=9)
    _, desc = _frame(oracle_lib, 1000, 0)
    N = len(desc)
    oword, ow, onode, oids, ovals = oracle_lib.bow_transform(vocab, desc, 2)
    ids = np.zeros(N, np.int32); vals = np.zeros(N); fn = np.zeros(N, np.int32); fo = np.zeros(N + 1, np.int32); fi = np.zeros(N, np.int32)
    sizes = np.zeros(3, np.int32)
    rc = hostlib.ccmh_bow_transform(0, vocab["n_nodes"], vocab["L"], _p(vocab["child_off"]), _p(vocab["child_id"]), _p(vocab["node_desc"]),
                                    _p(vocab["word_id"]), _p(vocab["weight"]), _p(desc), N, 2, _p(ids), _p(vals), _p(fn), _p(fo), _p(fi), _p(sizes))
    assert rc == 0
    nb, nn, ni = sizes
    assert np.array_equal(ids[:nb], oids) and np.array_equal(vals[:nb], ovals)
    keep = ow > 0
    exp_nodes = np.unique(onode[keep])
    assert np.array_equal(fn[:nn], exp_nodes) and ni == keep.sum()
    for k, nd in enumerate(exp_nodes):
        assert np.array_equal(fi[fo[k]:fo[k + 1]], np.nonzero(keep & (onode == nd))[0])


def test_optimize_sim3_class_method(hostlib, oracle_lib):
    p = synth.make_sim3_problem(120, 11)
    s = p["sim3"].copy()
    keep = np.zeros(120, np.uint8)
    f = lambda a: _p(np.ascontiguousarray(a, np.float64))
    arrs = [np.ascontiguousarray(p[k], np.float64) for k in ("P1c", "P2c", "obs1", "obs2", "info1", "info2", "K1", "K2")]
    hostlib.ccmh_optimize_sim3.argtypes = [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_float, C.c_int, C.c_void_p]
    nin = hostlib.ccmh_optimize_sim3(0, _p(s), 120, *[_p(a) for a in arrs], 10.0, 0, _p(keep))
    so, keepo, nino = oracle_lib.sim3_optimize(p["sim3"], p["P1c"], p["P2c"], p["obs1"], p["obs2"], p["info1"], p["info2"], p["K1"], p["K2"], 10.0, False)
    assert nin == nino and np.array_equal(keep, keepo)
    assert np.abs(s - so).max() < 1e-6


def test_search_by_projection_through_the_device_grid(hostlib, oracle_lib):
    """Frame glue on the device end to end: raw (distorted) keypoints -> undistort + grid on the GPU -> ONE window-search
    launch sequence -> ordered claim replay.  Expected result: the oracle's undistortion feeding the oracle's M1."""
    kps, desc = _frame(oracle_lib, 1000, 7)
    N = len(kps)
    K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
    xy_o = oracle_lib.undistort_points(K, D, np.stack([kps["x"], kps["y"]], 1))
    bounds = oracle_lib.image_bounds(K, D, 752, 480)
    rng = np.random.default_rng(3)
    sf = synth.scale_tables()[0]
    n_mp = 4000
    src = rng.integers(0, N, n_mp)
    px = (xy_o[src, 0] + rng.normal(0, 2.0, n_mp)).astype(np.float32)
    py = (xy_o[src, 1] + rng.normal(0, 2.0, n_mp)).astype(np.float32)
    lvl = np.clip(kps["octave"][src] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    vcos = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    in_view = (rng.random(n_mp) < 0.9).astype(np.uint8)
    bits = np.unpackbits(desc[src], axis=1)
    mp_desc = np.packbits(bits ^ (rng.random(bits.shape) < 0.08), axis=1)
    frame_mp0 = -np.ones(N, np.int32)
    frame_mp0[rng.integers(0, N, 50)] = 10_000
    exp_n, exp = oracle_lib.search_by_projection_mp(xy_o[:, 0], xy_o[:, 1], kps["octave"], desc, tuple(float(b) for b in bounds), sf, in_view,
                                                    px, py, lvl, vcos, mp_desc, 3.0, 0.8, frame_mp0)
    got = frame_mp0.copy()
    xy_un = np.zeros((N, 2), np.float32)
    kraw = np.ascontiguousarray(kps)
    n = hostlib.ccmh_search_by_projection_mp_dev(0, _p(K), _p(D), 4, 752, 480, _p(kraw), _p(desc), N, _p(sf), n_mp, _p(in_view), _p(px), _p(py),
                                                 _p(lvl), _p(vcos), _p(mp_desc), C.c_float(3.0), C.c_float(0.8), _p(got), _p(xy_un))
    assert np.array_equal(xy_un, xy_o)
    assert n == exp_n and n > 500
    assert np.array_equal(got, exp)


def test_fuse_fan_out_of_twenty_target_keyframes_in_one_launch(hostlib, oracle_lib):
    """LocalMapping::SearchInNeighbors (Mapping.cpp:497-503) calls matcher.Fuse(pKFi, vpMapPointMatches) for every target keyframe (20 neighbours + second neighbours).
    FuseBatch sends the Hamming work of ALL targets out as one ccm_hamming_csr_multi launch; resolve(s) must return exactly what the single projected window search of
    target s returns (host mirror, one launch per call) and what the oracle's sequential restatement returns — also when points have turned bad / have joined the
    keyframe since the batch was built (the reference's loop skips them, ORBmatcher.cpp:884-888)."""
    ext = oracle_lib.OrbOracle(1000)
    S = 20
    frames = [ext.extract(synth.gen_image(1000 + s // 5, s % 5)) for s in range(S)]
    rng = np.random.default_rng(77)
    sf, _, _, is2 = synth.scale_tables()
    c = np.ascontiguousarray
    n_pts = 900                                                 # the current keyframe's map points, projected into every target
    kf_off, pt_off = [0], [0]
    kx, ky, oc, kd, bounds, valid, u, v, level, pdesc = [], [], [], [], [], [], [], [], [], []
    for kps, desc in frames:
        N = len(kps)
        src = rng.integers(0, N, n_pts)
        kx.append(c(kps["x"])); ky.append(c(kps["y"])); oc.append(c(kps["octave"])); kd.append(desc)
        bounds.append([0.0, 0.0, 752.0, 480.0])
        u.append((kps["x"][src] + rng.normal(0, 1.5, n_pts)).astype(np.float32)); v.append((kps["y"][src] + rng.normal(0, 1.5, n_pts)).astype(np.float32))
        level.append(np.clip(kps["octave"][src] + rng.integers(0, 2, n_pts), 0, 7).astype(np.int32))
        valid.append((rng.random(n_pts) < 0.8).astype(np.uint8))
        bits = np.unpackbits(desc[src], axis=1)
        pdesc.append(np.packbits(bits ^ (rng.random(bits.shape) < 0.06), axis=1))
        kf_off.append(kf_off[-1] + N); pt_off.append(pt_off[-1] + n_pts)
    cat = lambda xs, dt: c(np.concatenate(xs).astype(dt))
    KX, KY, OC, KD = cat(kx, np.float32), cat(ky, np.float32), cat(oc, np.int32), c(np.concatenate(kd))
    B = c(np.array(bounds, np.float32)); VA, U, V, LV, PD = cat(valid, np.uint8), cat(u, np.float32), cat(v, np.float32), cat(level, np.int32), c(np.concatenate(pdesc))
    KO, PO = np.array(kf_off, np.int32), np.array(pt_off, np.int32)
    hostlib.ccmh_fuse_batch_create.restype = C.c_void_p
    hostlib.ccmh_fuse_batch_create.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 14 + [C.c_float]
    hostlib.ccmh_fuse_batch_candidates.restype = C.c_longlong
    hostlib.ccmh_fuse_batch_candidates.argtypes = [C.c_void_p]
    hostlib.ccmh_fuse_batch_resolve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    hostlib.ccmh_fuse_batch_destroy.argtypes = [C.c_void_p]; hostlib.ccmh_fuse_batch_destroy.restype = None
    h = hostlib.ccmh_fuse_batch_create(0, S, _p(KO), _p(KX), _p(KY), _p(OC), _p(KD), _p(B), _p(sf), _p(is2), _p(PO), _p(VA), _p(U), _p(V), _p(LV), _p(PD), 3.0)
    assert h, "ccmh_fuse_batch_create failed"
    assert hostlib.ccmh_fuse_batch_candidates(h) > 5 * S * n_pts // 10
    total = 0
    for s in range(S):
        kps, desc = frames[s]
        N = len(kps)
        # points that earlier calls of the loop turned bad (Replace) or added to this keyframe: skipped now
        skip = (rng.random(n_pts) < (0.0 if s == 0 else 0.15)).astype(np.uint8)
        bi = np.zeros(n_pts, np.int32); bd = np.zeros(n_pts, np.int32)
        n = hostlib.ccmh_fuse_batch_resolve(h, s, _p(skip), n_pts, _p(bi), _p(bd))
        valid_now = (valid[s] & (1 - skip)).astype(np.uint8)
        exp_n, ebi, ebd, _ = oracle_lib.projected_window_search(kps["x"], kps["y"], kps["octave"], desc, (0.0, 0.0, 752.0, 480.0), sf, is2, valid_now, u[s], v[s], level[s],
                                                                pdesc[s], 3.0, 1, 50, None, 0, np.zeros(n_pts, np.uint8))
        assert n == exp_n and np.array_equal(bi, ebi) and np.array_equal(bd, ebd), s
        gbi = np.zeros(n_pts, np.int32); gbd = np.zeros(n_pts, np.int32)
        n1 = hostlib.ccmh_projected_window_search(0, _p(kx[s]), _p(ky[s]), _p(oc[s]), _p(desc), N, *[C.c_float(b) for b in (0.0, 0.0, 752.0, 480.0)], _p(sf), _p(is2), n_pts,
                                                  _p(valid_now), _p(u[s]), _p(v[s]), _p(level[s]), _p(pdesc[s]), C.c_float(3.0), 1, 50, None, 0, None, _p(gbi), _p(gbd))
        assert n1 == n and np.array_equal(gbi, bi) and np.array_equal(gbd, bd), s
        total += n
    assert total > 100 * S
    hostlib.ccmh_fuse_batch_destroy(h)
