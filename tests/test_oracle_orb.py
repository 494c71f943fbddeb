"""CPU: known-answer tests pinning the ORB oracle (oracle/orb_ref.cpp).  The reference has no golden
vectors for this path; each check below is derived by hand or by an independent numpy evaluation of
the definition."""
import numpy as np

import oracle
from ccm_slam_amd import orb, synth


def test_gaussian_kernel_is_opencv_fixed_point_kernel():
    k = oracle.gaussian_kernel7()
    # exp(-x^2/8) normalised, 8 fractional bits with error diffusion; centre = 256 - rest
    assert list(k) == [18, 34, 48, 56, 48, 34, 18] and k.sum() == 256


def test_blur_constant_and_impulse():
    img = np.full((40, 50), 93, np.uint8)
    assert (oracle.gaussian_blur7(img) == 93).all()
    img = np.zeros((41, 41), np.uint8)
    img[20, 20] = 255
    out = oracle.gaussian_blur7(img).astype(int)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    exp = (np.outer(k, k) * 255 + (1 << 15)) >> 16
    assert np.array_equal(out[17:24, 17:24], exp)
    assert out.sum() == exp.sum()


def test_blur_reflect101_border():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (20, 24), dtype=np.uint8)
    pad = np.pad(img, 3, mode="reflect")   # numpy 'reflect' == BORDER_REFLECT_101
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    h = sum(k[i] * pad[:, i:i + 24].astype(np.int64) for i in range(7))
    v = sum(k[i] * h[i:i + 20, :] for i in range(7))
    assert np.array_equal(oracle.gaussian_blur7(img), ((v + (1 << 15)) >> 16).astype(np.uint8))


def test_resize_constant_and_identity_like():
    img = np.full((480, 752), 200, np.uint8)
    assert (oracle.resize_linear_u8(img, 627, 400) == 200).all()
    # horizontal ramp stays monotone and within range after 1/1.2 scaling
    ramp = np.tile(np.linspace(0, 255, 752).astype(np.uint8), (480, 1))
    out = oracle.resize_linear_u8(ramp, 627, 400)
    assert (np.diff(out[10].astype(int)) >= 0).all() and out[10, 0] <= 1 and out[10, -1] >= 254


def _fast_score_numpy(img):
    """independent evaluation of the FAST-9/16 definition: largest t such that 9 contiguous ring pixels are all
    > v+t or all < v-t (0 where no t >= 0 exists... returned as -1)"""
    off = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
           (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    im = img.astype(int)
    sc = -np.ones((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = im[y, x]
            ring = np.array([im[y + dy, x + dx] for dx, dy in off])
            best = -1
            for s in range(16):
                arc = ring[[(s + k) % 16 for k in range(9)]]
                best = max(best, (arc - v).min() - 1, (v - arc).min() - 1)
            sc[y, x] = best
    return sc


def test_fast_matches_definition_on_random_patch():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (30, 34), dtype=np.uint8)
    img[8:20, 10:22] = 250   # a bright block on noise: strong corners
    for th in (7, 20, 60):
        kps = oracle.fast9_16(img, th)
        sc = _fast_score_numpy(img)
        m = np.where(sc >= th, sc, 0)
        exp = []
        for y in range(3, 30 - 3):
            for x in range(3, 34 - 3):
                s = m[y, x]
                if s <= 0:
                    continue
                nb = [m[y + dy, x + dx] if 3 <= y + dy < 27 and 3 <= x + dx < 31 else 0
                      for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)]
                if all(s > n for n in nb):
                    exp.append((x, y, s))
        got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps]
        assert got == exp, th


def test_fast_atan2_quadrants_and_accuracy():
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    for ang in np.linspace(1, 359, 97):
        y, x = np.sin(np.deg2rad(ang)), np.cos(np.deg2rad(ang))
        got = oracle.fast_atan2(float(np.float32(y)), float(np.float32(x)))
        assert abs(got - ang) < 0.02   # OpenCV documents ~0.3 deg accuracy for fastAtan2; the polynomial is much better


def test_extractor_tables():
    o = oracle.OrbOracle(1000)
    sf, isf, s2, is2, nf, umax = o.tables()
    assert list(nf) == [217, 181, 151, 126, 105, 87, 73, 60] and nf.sum() == 1000
    assert list(umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert [o.level_size(752, 480, l) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231),
                                                             (302, 193), (252, 161), (210, 134)]
    o2 = oracle.OrbOracle(2000)
    assert list(o2.tables()[4]) == [434, 362, 302, 251, 209, 175, 145, 122]


def test_extract_basic_invariants():
    img = synth.gen_image(1000, 0)
    o = oracle.OrbOracle(1000)
    kps, desc = o.extract(img)
    assert 900 < len(kps) <= 1024 and desc.shape == (len(kps), 32)
    assert (np.diff(kps["octave"]) >= 0).all()                      # level-major order
    for l in range(8):
        k = kps[kps["octave"] == l]
        lw, lh = o.level_size(752, 480, l)
        sf = o.tables()[0][l]
        # keypoints stay >= 19 px from every level edge (EDGE_THRESHOLD), coordinates scaled back by mvScaleFactor
        assert (k["x"] >= np.float32(19) * sf - 1e-3).all() and (k["x"] <= np.float32(lw - 20) * sf + 1e-3).all()
        assert (k["size"] == np.float32(int(31 * sf))).all()
    assert (kps["angle"] >= 0).all() and (kps["angle"] < 360).all()
    # a constant image has no corners at all
    k0, d0 = o.extract(np.full((480, 752), 9, np.uint8))
    assert len(k0) == 0


def test_product_octree_equals_oracle_octree():
    """host-only: the product's DistributeOctTree (ccm_orb_distribute_octree) against the oracle's literal
    std::list restatement, on random candidate sets including heavy ties."""
    rng = np.random.default_rng(11)
    for trial in range(12):
        n = int(rng.integers(1, 4000))
        W, H = 720, 448
        x = rng.integers(3, W - 3, n).astype(np.float32)
        y = rng.integers(3, H - 3, n).astype(np.float32)
        r = rng.integers(7, 60 if trial % 2 else 255, n).astype(np.float32)
        N = int(rng.integers(5, 400))
        kp = np.zeros(n, oracle.KP_DTYPE)
        kp["x"], kp["y"], kp["response"] = x, y, r
        exp = oracle.distribute_octree(kp, 16, 16 + W, 16, 16 + H, N)
        sel = orb.distribute_octree(x, y, r, 16, 16 + W, 16, 16 + H, N)
        assert len(sel) == len(exp)
        assert np.array_equal(x[sel], exp["x"]) and np.array_equal(y[sel], exp["y"]) and np.array_equal(r[sel], exp["response"])
