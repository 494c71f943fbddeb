"""GPU parity: ORB extraction vs the oracle restatement of cslam::ORBextractor — bit-exact keypoints
(all six fields) and descriptors, plus every intermediate product (pyramid levels, FAST score semantics via
the pre-octree candidate lists, blurred levels)."""
import numpy as np
import pytest

from ccm_slam_amd import orb, synth

pytestmark = pytest.mark.gpu


def _compare(ctx, oracle_lib, img, nfeatures, **kw):
    ex = orb.ORBextractor(ctx, nfeatures, **kw)
    kps, desc, pyr = ex(img, want_pyramid=True)
    o = oracle_lib.OrbOracle(nfeatures, **{{"ini_th_fast": "ini_th", "min_th_fast": "min_th", "scale_factor": "scale"}.get(k, k): v for k, v in kw.items()})
    okps, odesc = o.extract(img)
    for l in range(ex.nlevels):
        assert np.array_equal(pyr[l], o.level(l)), f"pyramid level {l}"
        cand = ex.debug_candidates(l)
        ocand = o.candidates(l)
        assert len(cand) == len(ocand), f"candidate count level {l}: {len(cand)} vs {len(ocand)}"
        for f in ("x", "y", "response"):
            assert np.array_equal(cand[f], ocand[f]), f"candidates level {l} field {f}"
        if (okps["octave"] == l).any():
            _, blur = ex.debug_level(l)
            assert np.array_equal(blur, o.blur(l)), f"blur level {l}"
    assert len(kps) == len(okps)
    for f in ("octave", "x", "y", "response", "size", "angle"):
        assert np.array_equal(kps[f], okps[f]), f"keypoint field {f}"
    assert np.array_equal(desc, odesc)
    ex.close()
    return kps, desc


@pytest.mark.parametrize("seed,t", [(1000, 0), (1000, 7), (1003, 0)])
def test_euroc_shape_1000_features(ctx, oracle_lib, seed, t):
    kps, desc = _compare(ctx, oracle_lib, synth.gen_image(seed, t), 1000)
    assert 900 <= len(kps) <= 1000 + 24


def test_init_extractor_2000_features(ctx, oracle_lib):
    kps, _ = _compare(ctx, oracle_lib, synth.gen_image(1001, 3), 2000)
    assert len(kps) > 1500


def test_low_texture_uses_min_threshold(ctx, oracle_lib):
    # smooth image with a few weak blobs: most cells are empty at iniThFAST=20 and retry with minThFAST=7
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:480, 0:752]
    img = (100 + 30 * np.sin(xx / 60.0) * np.cos(yy / 45.0)).astype(np.float64)
    for _ in range(300):
        x, y = rng.integers(30, 720), rng.integers(30, 450)
        img[y - 1:y + 2, x - 1:x + 2] += rng.integers(8, 18)
    _compare(ctx, oracle_lib, np.clip(np.rint(img), 0, 255).astype(np.uint8), 1000)


def test_constant_image_no_keypoints(ctx, oracle_lib):
    ex = orb.ORBextractor(ctx, 1000)
    kps, desc = ex(np.full((480, 752), 77, np.uint8))
    assert len(kps) == 0 and desc.shape == (0, 32)
    ex.close()


def test_other_geometry_and_levels(ctx, oracle_lib):
    img = synth.gen_image(77, 1, w=640, h=360)
    _compare(ctx, oracle_lib, img, 700, nlevels=5, scale_factor=1.3)


def test_accessors_match_reference_tables(ctx, oracle_lib):
    ex = orb.ORBextractor(ctx, 1000)
    o = oracle_lib.OrbOracle(1000)
    sf, isf, s2, is2, nf, _ = o.tables()
    assert np.array_equal(ex.GetScaleFactors(), sf) and np.array_equal(ex.GetInverseScaleFactors(), isf)
    assert np.array_equal(ex.GetScaleSigmaSquares(), s2) and np.array_equal(ex.GetInverseScaleSigmaSquares(), is2)
    assert np.array_equal(ex.features_per_level(), nf)
    assert list(nf) == [217, 181, 151, 126, 105, 87, 73, 60]
    ex.close()


def test_repeatable(ctx):
    img = synth.gen_image(1000, 11)
    ex = orb.ORBextractor(ctx, 1000)
    k1, d1 = ex(img)
    k2, d2 = ex(img)
    assert np.array_equal(d1, d2) and np.array_equal(k1, k2)
    ex.close()


def test_batch_of_frames_in_hbm_equals_frame_by_frame_extraction(ctx, oracle_lib):
    """ccm_orb_extract_batch_dev keeps two frames in flight (frame t+1's device phase 1 overlaps the host keypoint selection of frame t) on two
    buffer sets; every frame's keypoints and descriptors must be what the single-frame entry point (and therefore the oracle) gives."""
    from ccm_slam_amd.orb import OrbBatchDev
    ex = orb.ORBextractor(ctx, 1000)
    imgs = np.stack([synth.gen_image(1000 + (t % 3), t) for t in range(9)])
    bat = OrbBatchDev(ctx, ex, imgs)
    bat.run()
    res = bat.results()
    bat.run()                                   # a second pass over the same buffers
    res2 = bat.results()
    bat.close()
    for t in range(9):
        kps, desc = ex(imgs[t])
        for got in (res[t], res2[t]):
            assert len(got[0]) == len(kps) and np.array_equal(got[1], desc)
            for f in kps.dtype.names:
                assert np.array_equal(got[0][f], kps[f]), (t, f)
    ok, od = oracle_lib.OrbOracle(1000).extract(imgs[4])
    assert np.array_equal(res[4][1], od) and np.array_equal(res[4][0]["x"], ok["x"])
    ex.close()


def test_device_octree_matches_the_sequential_list_walk(ctx, oracle_lib):
    """orb_octree_kernel (round-based restatement of DistributeOctTree, one workgroup per level) against the oracle's sequential walk on random
    candidate sets: one / two / three root nodes, N from 1 to 434, few and many candidates, heavy ties in size and response."""
    rng = np.random.default_rng(5)
    from oracle import KP_DTYPE
    n_checked = 0
    for trial in range(60):
        W = int(rng.integers(120, 760)); H = int(rng.integers(60, 470))
        if W < H: W, H = H, W
        if trial % 7 == 0: W = min(3 * H + int(rng.integers(0, 40)), 4000)
        n = int(rng.integers(1, 6000)); N = int(rng.choice([1, 2, 5, 17, 60, 105, 217, 434]))
        x = rng.integers(0, W, n); y = rng.integers(0, H, n)
        _, ui = np.unique(np.stack([x, y], 1), axis=0, return_index=True); ui.sort()
        x, y = x[ui], y[ui]
        resp = rng.integers(7, 7 + (3 if trial % 3 == 0 else 120), x.size)
        sel, over = orb.debug_octree_dev(ctx, x, y, resp, W, H, N)
        if over: continue
        kps = np.zeros(x.size, KP_DTYPE); kps["x"] = x; kps["y"] = y; kps["response"] = resp
        ref = oracle_lib.distribute_octree(kps, 0, W, 0, H, N)
        assert len(sel) == len(ref), (trial, W, H, x.size, N, len(sel), len(ref))
        assert np.array_equal(x[sel].astype(np.float32), ref["x"]) and np.array_equal(y[sel].astype(np.float32), ref["y"]), (trial, W, H, x.size, N)
        n_checked += 1
    assert n_checked >= 50


def test_levels_that_do_not_fit_the_octree_kernel_fall_back_to_the_host(ctx, oracle_lib, monkeypatch):
    """CCM_ORB_OCT_KCAP shrinks the kernel's candidate capacity: level 0 overflows, the frame is redone through the host octree, same result."""
    monkeypatch.setenv("CCM_ORB_OCT_KCAP", "1024")
    _compare(ctx, oracle_lib, synth.gen_image(1000, 3), 1000)


@pytest.mark.parametrize("w,h,nlevels,scale", [(752, 480, 8, 1.2), (333, 251, 5, 1.7)])
def test_per_level_resize_kernels_still_agree(ctx, oracle_lib, monkeypatch, w, h, nlevels, scale):
    """CCM_ORB_PYR_FUSED=0: one orb_resize_kernel launch per level instead of orb_pyramid_kernel (the default, which every other test of this file runs on)."""
    monkeypatch.setenv("CCM_ORB_PYR_FUSED", "0")
    _compare(ctx, oracle_lib, synth.gen_image(777, 2, w, h), 600, nlevels=nlevels, scale_factor=scale)


_SWEEP = synth.orb_sweep_cases()


@pytest.mark.parametrize("case", range(len(_SWEEP)), ids=[c[0] for c in _SWEEP])
def test_wide_sweep(ctx, oracle_lib, case):
    """ORBextractor.cpp:933-998 / :1280-1304 over 66 images (synth.orb_sweep_cases): widths and heights of every residue mod 4, levels smaller than
    one 30-px cell (64x48, 40x40: no keypoints at all, the pyramid is still produced), single-cell levels with 59-px cells, dense checkerboards and
    noise (up to 33 000 octree candidates on a level: the device octree's LDS plan overflows and the frame is redone through the host octree),
    saturated / step images, nlevels 1 and 12, scale factors 1.1 - 2.0, 7 - 5000 features, iniThFAST == minThFAST.  Bit-exact against the oracle, which
    tests/test_ref_orb.py pins to the reference's own translation unit on the same images."""
    name, img, nf, kw = _SWEEP[case]
    kw = {{"scale": "scale_factor", "ini_th": "ini_th_fast", "min_th": "min_th_fast"}.get(k, k): v for k, v in kw.items()}
    _compare(ctx, oracle_lib, img, nf, **kw)


def test_sweep_images_also_agree_through_the_batch_entry(ctx, oracle_lib):
    """ccm_orb_extract_batch_dev (four frames in flight) on 12 frames of an odd size: identical to frame-by-frame extraction."""
    imgs = np.stack([synth.gen_image(5000 + k, k, 751, 479) for k in range(12)])
    ex = orb.ORBextractor(ctx, 1000)
    b = orb.OrbBatchDev(ctx, ex, imgs)
    b.run()
    res = b.results()
    b.close()
    for k in range(12):
        kps, desc = ex(imgs[k])
        assert len(kps) == len(res[k][0]) and np.array_equal(desc, res[k][1])
        for f in kps.dtype.names:
            assert np.array_equal(kps[f], res[k][0][f])
    ex.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_geometries_and_parameters(ctx, oracle_lib, seed):
    """Seeded random extractor set-ups: image sizes 70 ... 1000 x 70 ... 700 (never more than twice as high as wide on any level that holds a cell: the reference divides by
    nIni = 0 there and the product rejects the frame), 1 - 10 levels, scale factors 1.1 - 1.7, 20 - 3000 features, FAST thresholds 5 - 40 — bit-exact against the oracle."""
    rng = np.random.default_rng(31000 + seed)
    w = int(rng.integers(70, 1001)); h = int(rng.integers(70, min(701, 2 * w - 40)))
    nlevels = int(rng.integers(1, 11)); scale = float(np.float32(rng.uniform(1.1, 1.7)))
    nf = int(rng.choice([20, 100, 500, 1000, 3000])); ini = int(rng.integers(8, 41)); mn = int(rng.integers(5, ini + 1))
    img = synth.gen_image(32000 + seed, seed, w, h)
    if rng.random() < 0.25:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    # a level that holds a cell but is more than twice as high as wide: DistributeOctTree computes nIni = round(W / H) = 0 and divides by it (ORBextractor.cpp:711-716);
    # the product refuses such a frame, the oracle (like the reference) must not be run on it
    ex = orb.ORBextractor(ctx, nf, scale, nlevels, ini, mn)
    tall = False
    for l in range(nlevels):
        lw, lh = ex.level_size(w, h, l)
        W, H = lw - 32, lh - 32
        tall = tall or (W >= 30 and H >= 30 and int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5))) < 1)
    if tall:
        from ccm_slam_amd._lib import CcmError
        with pytest.raises(CcmError, match="twice as high as wide"):
            ex(img)
        ex.close()
        return
    ex.close()
    _compare(ctx, oracle_lib, img, nf, nlevels=nlevels, scale_factor=scale, ini_th_fast=ini, min_th_fast=mn)
