#!/usr/bin/env python3
"""Pin the oracle's restated OpenCV primitives against a REAL OpenCV, on any box that has one.

The build image has no OpenCV, so `oracle/cv_prims.h` (resize INTER_LINEAR 8U, FAST-9/16 + score + 3x3 NMS, 7x7 sigma-2 GaussianBlur 8U, fastAtan2) and
`oracle/match_ref.cpp`'s undistortPoints restate OpenCV 4.2.0 from its source and are pinned by hand-derived known-answer tests only.
This script closes that gap wherever `import cv2` works:

    python scripts/pin_opencv.py            # compares, prints one line per primitive, writes tests/golden/opencv_<version>.npz
    python scripts/pin_opencv.py --no-dump  # compares only

It runs cv2 on the repository's synthetic frames (ccm_slam_amd.synth.gen_image: the frames every ORB test uses) and compares with the oracle through
liboracle.so bit for bit (integers) / exactly (fastAtan2, undistortPoints: f32).  The dumped file holds INPUT SEEDS and cv2's OUTPUTS only — data, no source — and
`tests/test_golden.py::test_oracle_matches_the_opencv_vectors_when_present` consumes every tests/golden/opencv_*.npz it finds, so a vector file produced once on a
maintainer's box keeps the pin alive here.  Exit code: 0 all equal (or cv2 absent: nothing to do), 1 a primitive differs (the report says where).

Reference call sites (cslam/src/ORBextractor.cpp): resize :1293, FAST :978 / :983, GaussianBlur :1259, fastAtan2 :113 (IC_Angle); Frame.cpp:131-160 undistortPoints.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FRAMES = [(1000, 0), (1000, 7), (1003, 0)]                 # (seed, t) of synth.gen_image
K4 = np.array([458.654, 457.296, 367.215, 248.375], np.float32)               # conf/vi_euroc.yaml:9-12
D4 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)


def level_sizes(w, h, nlevels=8, scale=1.2):
    """ORBextractor's level geometry: cvRound(dim * (1 / scale^l)) with the f32 scale table (ORBextractor.cpp:429-445, 1285-1290)"""
    out, sf = [], np.float32(1.0)
    for lvl in range(nlevels):
        inv = np.float32(1.0) / sf
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        sf = np.float32(sf * np.float32(scale))
    return out


def cv2_vectors(cv2):
    """what a real OpenCV computes on the synthetic frames: the arrays that go into the vector file"""
    from ccm_slam_amd import synth
    vec = {"opencv_version": np.array(cv2.__version__), "frames": np.array(FRAMES, np.int32)}
    for fi, (seed, t) in enumerate(FRAMES):
        img = synth.gen_image(seed, t)
        sizes = level_sizes(img.shape[1], img.shape[0])
        cur = img
        for lvl in range(1, 4):                             # three levels of the chain suffice: every level runs the same arithmetic on another geometry
            cur = cv2.resize(cur, sizes[lvl], 0, 0, cv2.INTER_LINEAR)
            vec[f"f{fi}_resize_l{lvl}"] = cur
        vec[f"f{fi}_blur"] = cv2.GaussianBlur(img, (7, 7), 2, 2, cv2.BORDER_REFLECT_101)
        for th in (20, 7):
            roi = np.ascontiguousarray(img[100:260, 200:420])                       # one 160 x 220 region (the extractor calls FAST per cell, :978)
            fast = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
            kps = fast.detect(roi, None)
            vec[f"f{fi}_fast_t{th}"] = np.array([(k.pt[0], k.pt[1], k.response) for k in kps], np.float32).reshape(-1, 3)
    rng = np.random.default_rng(12)
    yx = rng.uniform(-300, 300, (4000, 2)).astype(np.float32)
    yx[:64] = [[0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, 1], [-1, -1], [1, -1]] * 8
    vec["atan2_in"] = yx
    vec["atan2_out"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in yx], np.float32)
    pts = rng.uniform([0, 0], [752, 480], (2000, 2)).astype(np.float32)
    Kmat = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float32)
    und = cv2.undistortPoints(pts.reshape(-1, 1, 2), Kmat, D4, None, Kmat)       # Frame.cpp:150
    vec["undistort_in"] = pts
    vec["undistort_out"] = und.reshape(-1, 2).astype(np.float32)
    return vec


def compare(vec, report=print):
    """the oracle (liboracle.so) against a vector file's cv2 outputs; returns the list of primitives that differ"""
    import oracle
    from ccm_slam_amd import synth
    oracle.build()
    bad = []

    def same(name, got, exp):
        ok = got.shape == exp.shape and np.array_equal(got, exp)
        if not ok:
            bad.append(name)
            n = int((got != exp).sum()) if got.shape == exp.shape else -1
            report(f"  DIFFERS {name}: {n} of {exp.size} entries" + (f", max |d| {np.abs(got.astype(np.float64) - exp.astype(np.float64)).max():g}" if n > 0 else " (shape)"))
        return ok
    for fi, (seed, t) in enumerate(np.asarray(vec["frames"]).tolist()):
        img = synth.gen_image(int(seed), int(t))
        sizes = level_sizes(img.shape[1], img.shape[0])
        cur = img
        for lvl in range(1, 4):
            cur = oracle.resize_linear_u8(cur, sizes[lvl][0], sizes[lvl][1])
            same(f"resize frame {fi} level {lvl}", cur, np.asarray(vec[f"f{fi}_resize_l{lvl}"]))
            cur = np.asarray(vec[f"f{fi}_resize_l{lvl}"])       # continue from cv2's level so that one differing pixel does not cascade
        same(f"GaussianBlur frame {fi}", oracle.gaussian_blur7(img), np.asarray(vec[f"f{fi}_blur"]))
        for th in (20, 7):
            roi = np.ascontiguousarray(img[100:260, 200:420])
            k = oracle.fast9_16(roi, th)
            got = np.stack([k["x"], k["y"], k["response"]], 1).astype(np.float32) if len(k) else np.zeros((0, 3), np.float32)
            same(f"FAST frame {fi} threshold {th}", got, np.asarray(vec[f"f{fi}_fast_t{th}"]))
    yx = np.asarray(vec["atan2_in"])
    same("fastAtan2", np.array([oracle.fast_atan2(float(y), float(x)) for y, x in yx], np.float32), np.asarray(vec["atan2_out"]))
    same("undistortPoints", oracle.undistort_points(K4, D4, np.asarray(vec["undistort_in"])), np.asarray(vec["undistort_out"]))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-dump", action="store_true")
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print("pin_opencv: cv2 is not importable here — nothing to pin (run this on a box with OpenCV; 4.2.0 is the version the oracle restates)")
        return 0
    print(f"pin_opencv: OpenCV {cv2.__version__}")
    vec = cv2_vectors(cv2)
    bad = compare(vec)
    if not args.no_dump:
        path = os.path.join(ROOT, "tests", "golden", f"opencv_{cv2.__version__}.npz")
        np.savez_compressed(path, **vec)
        print(f"pin_opencv: wrote {os.path.relpath(path, ROOT)}")
    if bad:
        print(f"pin_opencv: {len(bad)} primitive(s) differ from OpenCV {cv2.__version__}: " + "; ".join(bad))
        if not cv2.__version__.startswith("4.2"):
            print("  (the oracle restates 4.2.0: GaussianBlur 8U rounds differently before 3.4.1, fastAtan2 has other coefficients in 2.4)")
        return 1
    print("pin_opencv: every primitive equals the oracle's restatement bit for bit")
    return 0


if __name__ == "__main__":
    sys.exit(main())
