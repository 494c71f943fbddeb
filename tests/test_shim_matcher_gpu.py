"""OUR drop-in translation unit shim/ORBmatcher_hip.cpp (-> libccm_host.so -> ccm_hamming_csr on the MI355X) in the place of the reference's
cslam/src/ORBmatcher.cpp: the SAME harness that pins the oracle to the reference (oracle/ref_matcher_driver.cpp: flat inputs -> the reference's Frame /
KeyFrame / MapPoint objects -> the method under test through the class API of cslam/include/cslam/ORBmatcher.h -> what it did to the map) is linked with
our ORBmatcher instead (shim/Makefile: libmatcher_hip_shim.so) and must produce the same tables as the oracle, which tests/test_ref_matcher.py shows
to be the reference's.  Rows M1-M10 of SURVEY 8a, incl. the map mutations of Fuse and the re-map branch of SearchByProjection(KeyFrame, Scw)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
import tests.test_ref_matcher as trm
from ccm_slam_amd import synth

SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shim", "libmatcher_hip_shim.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(SHIM), reason="shim/libmatcher_hip_shim.so not built")]


@pytest.fixture(scope="module")
def slib():
    return C.CDLL(SHIM)


@pytest.fixture(scope="module")
def frames():
    o = oracle.OrbOracle(1000)
    out = [o.extract(synth.gen_image(1000, t)) for t in (0, 1)]
    o.close()
    return out


def test_M1_search_by_projection_map_points(slib, frames):
    trm.test_search_by_projection_map_points_M1(slib, frames)


def test_M2_search_by_projection_last_frame(slib, frames):
    trm.test_search_by_projection_last_frame_M2_with_the_references_own_projection(slib, frames)


def test_M3_M4_search_by_bow(slib, frames):
    trm.test_search_by_bow_M3_M4(slib, frames)


def test_M5_search_for_triangulation(slib, frames):
    trm.test_search_for_triangulation_M5_with_the_references_own_epipole(slib, frames)


def test_M5_fan_out_of_twenty_neighbours_through_the_class_api(slib, frames):
    """the reference's own call sequence (Mapping.cpp:335: one SearchForTriangulation per neighbour, map points created in between) on shim/ORBmatcher_hip.cpp: the first
    call sends the Hamming work of all 20 neighbours to the MI355X in ONE launch (ccm_hamming_csr_multi), the others are answered from its tables — same match tables"""
    trm.test_triangulation_fan_out_of_a_new_keyframe_M5_x20(slib, frames)


def test_M5_fan_out_without_the_prediction_is_the_same(frames, monkeypatch):
    import subprocess, sys
    code = ("import ctypes as C, tests.test_ref_matcher as trm, tests.test_shim_matcher_gpu as t, oracle\n"
            "from ccm_slam_amd import synth\n"
            "o = oracle.OrbOracle(1000); fr = [o.extract(synth.gen_image(1000, k)) for k in (0, 1)]; o.close()\n"
            "trm.test_triangulation_fan_out_of_a_new_keyframe_M5_x20(C.CDLL(t.SHIM), fr)\nprint('ok')\n")
    env = dict(os.environ, CCM_SHIM_TRI_BATCH="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


def test_M6_search_for_initialization(slib, frames):
    trm.test_search_for_initialization_M6(slib, frames)


def test_M7_fuse(slib, frames):
    trm.test_fuse_M7_chi2_gate(slib, frames)


def test_M8_fuse_sim3(slib, frames):
    trm.test_fuse_sim3_M8(slib, frames)


def test_M9_search_by_projection_sim3(slib, frames):
    trm.test_search_by_projection_sim3_M9_claims(slib, frames)


def test_M10_search_by_sim3(slib, frames):
    trm.test_search_by_sim3_M10(slib, frames)


def test_shim_and_reference_libraries_agree_call_by_call(slib, frames):
    """the two libraries side by side on one Fuse scenario: same return value, same fused features, same projections"""
    if not os.path.exists(trm.LIB):
        pytest.skip("oracle/_ref not built")
    rlib = C.CDLL(trm.LIB)
    s = trm._projected_case(frames, 21)
    kps = s["kps"]
    has = (s["rng"].random(s["N"]) < 0.5).astype(np.uint8)
    outs = []
    for lib in (rlib, slib):
        best = np.zeros(s["n_pts"], np.int32); valid, u, v, lvl = trm._proj_out(s["n_pts"])
        p = trm._p; c = trm.c
        n = lib.ref_fuse(p(c(kps["x"])), p(c(kps["y"])), p(c(kps["octave"])), p(s["desc"]), s["N"], *trm.fb, p(s["sf"]), p(s["isig"]), p(s["K4"]), p(s["T"]), p(has),
                         s["n_pts"], p(s["Xw"]), p(s["normal"]), p(s["dmin"]), p(s["dmax"]), p(s["pdesc"]), C.c_float(3.0), p(best), p(valid), p(u), p(v), p(lvl))
        outs.append((n, best, valid, u, v, lvl))
    assert outs[0][0] == outs[1][0] > 500
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


def test_M7_fan_out_of_search_in_neighbors_through_the_class_api(slib, frames):
    """LocalMapping::SearchInNeighbors' sequence of Fuse calls (Mapping.cpp:469-503, replayed by oracle/ref_matcher_driver.cpp::ref_fuse_fan_out) on shim/ORBmatcher_hip.cpp:
    its first call predicts the twelve targets of the walk (one keyframe twice as first-level and second neighbour, second-only keyframes, repeats), projects the points
    into all of them and sends the Hamming work out as ONE launch (ccm_hamming_csr_multi); every call is answered from those tables with the points' flags as they are
    at that call.  Same targets, same return values, same fused feature for every point in every call as the reference's own ORBmatcher.cpp on the same scenario."""
    rlib = C.CDLL(trm.LIB)
    s, Ts, has, nb1, nb2 = trm._fuse_fan_out_case(frames)
    nc_r, tgt_r, n_r, best_r = trm._run_fuse_fan_out(rlib, s, Ts, has, nb1, nb2)
    nc_s, tgt_s, n_s, best_s = trm._run_fuse_fan_out(slib, s, Ts, has, nb1, nb2)
    assert nc_r == nc_s == 12 and np.array_equal(tgt_r, tgt_s)
    assert np.array_equal(n_r, n_s), (n_r, n_s)
    assert np.array_equal(best_r, best_s)
    # a walk that stops early (LocalMapping is interrupted): the tables of the unfinished fan-out must not leak into the next one
    nc2, tgt2, n2, best2 = trm._run_fuse_fan_out(slib, s, Ts, has, nb1, nb2, max_calls=5)
    assert nc2 == 5 and np.array_equal(n2, n_r[:5]) and np.array_equal(best2, best_r[:5])
    nc3, tgt3, n3, best3 = trm._run_fuse_fan_out(slib, s, Ts, has, nb1, nb2)
    assert np.array_equal(n3, n_r) and np.array_equal(best3, best_r)


def test_M7_fan_out_without_the_prediction_is_the_same(frames):
    import subprocess, sys
    code = ("import ctypes as C, numpy as np, tests.test_ref_matcher as trm, tests.test_shim_matcher_gpu as t, oracle\n"
            "from ccm_slam_amd import synth\n"
            "o = oracle.OrbOracle(1000); fr = [o.extract(synth.gen_image(1000, k)) for k in (0, 1)]; o.close()\n"
            "case = trm._fuse_fan_out_case(fr)\n"
            "r = trm._run_fuse_fan_out(C.CDLL(trm.LIB), *case); q = trm._run_fuse_fan_out(C.CDLL(t.SHIM), *case)\n"
            "assert r[0] == q[0] and all(np.array_equal(a, b) for a, b in zip(r[1:], q[1:]))\nprint('ok')\n")
    env = dict(os.environ, CCM_SHIM_FUSE_BATCH="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
