"""The library under the reference's thread model (round 5).  CCM-SLAM runs, in ONE process, a Tracking thread and a LocalMapping thread per
agent (cslam/src/ClientHandler.cpp:184) and one global-BA thread per Map (cslam/src/Map.cpp:1401-1402, started from LoopFinder.cpp:686-688 / the
map-merge path), all of which reach this back end through their own ccm_ctx on the same device.  The reduced solve of a 33 .. 2048-camera bundle
adjustment is ONE persistent kernel that needs all its workgroups co-resident (up to one per CU); two of them launched at the same time from two
contexts would each get part of the chip.  Every such launch goes through a per-device, process-wide lease (common.h: ccm_coresident_scope — an
event chain on the GPU, no host thread blocks), and these tests run the combinations the reference produces:

  * two Maps optimised at once (two threads x two contexts, gba_c3 each: 2 x 150 workgroups on 256 CUs — they CANNOT be co-resident),
  * a global BA beside a frame stream (ORB batches) and a pose-optimisation loop,
  * a local BA at the reference's window size (50 free + 20 fixed keyframes: persistent solver with 7 units) beside tracking.

Bar: every result BIT-IDENTICAL to the same call made alone, and no persistent launch gave up (ccm_coresidency_stats.aborted unchanged).
"""
import threading

import numpy as np
import pytest

from ccm_slam_amd import optimizer, orb, synth
from ccm_slam_amd._lib import Context, coresidency_stats

pytestmark = pytest.mark.gpu


def _run_threads(fns, timeout=600):
    out, err = [None] * len(fns), [None] * len(fns)
    gate = threading.Barrier(len(fns))

    def main(i):
        try:
            gate.wait(timeout=60)
            out[i] = fns[i]()
        except Exception as e:   # noqa: BLE001 — reported below
            err[i] = e
            try:
                gate.abort()
            except Exception:
                pass
    th = [threading.Thread(target=main, args=(i,)) for i in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=timeout)
    assert not any(t.is_alive() for t in th), "a thread is stuck"
    assert all(e is None for e in err), err
    return out


def _gba(prob, iters=20, reps=1, resident=True):
    """one thread's work: own context, create (from HBM) + run(iters), `reps` times; returns everything a caller can see"""
    ctx = Context(0)
    res = optimizer.ResidentProblem(ctx, prob) if resident else None
    outs = []
    for _ in range(reps):
        h = optimizer.BAHandle(ctx, prob, resident=res)
        st = h.run(iters)
        chi, lam, tr = h.history()
        cam, pts, chi2, dpos = h.download()
        outs.append(dict(cam=cam, pts=pts, chi2=chi2, dpos=dpos, chi=chi, lam=lam, tr=tr, st=(st.iters_done, st.lm_trials, st.pcg_iters, st.stop_reason), chi2_final=st.chi2_final))
        h.close()
    if res is not None:
        res.close()
    ctx.close()
    return outs


def _same(a, b, what):
    for k in ("cam", "pts", "chi2", "dpos", "chi", "lam", "tr"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs from the solo run"
    assert a["st"] == b["st"] and a["chi2_final"] == b["chi2_final"], (what, a["st"], b["st"])


def test_two_maps_optimised_at_once_equal_the_solo_runs():
    """Map.cpp:1401-1402: one GBA thread per Map.  gba_c3 twice at the same time: 300 workgroups of ba_pcg_persist for 256 CUs."""
    prob = synth.make_ba_config("gba_c3")
    solo = _gba(prob)[0]
    assert solo["st"][0] >= 10 and solo["st"][1] > solo["st"][0]   # the full call with its rejected trials
    before = coresidency_stats(0)
    both = _run_threads([lambda: _gba(prob, reps=2), lambda: _gba(prob, reps=2)])
    after = coresidency_stats(0)
    for t, outs in enumerate(both):
        for r, o in enumerate(outs):
            _same(o, solo, f"thread {t} repetition {r}")
    assert after["aborted"] == before["aborted"], (before, after)
    assert after["launches"] - before["launches"] == 4 * solo["st"][1]
    assert after["chained"] > before["chained"], "the two contexts' persistent launches were never ordered against each other"


def test_global_ba_beside_a_frame_stream_and_a_pose_loop():
    """LoopFinder.cpp:686-688 starts the GBA thread while the agents keep tracking: ORB batches + pose optimisations on their own contexts."""
    prob = synth.make_ba_config("gba_c3")
    imgs = np.stack([synth.gen_image(7100 + i, i) for i in range(8)])
    pp = synth.make_pose_problem(300, seed=9)

    def orb_leg(reps):
        ctx = Context(0)
        ex = orb.ORBextractor(ctx, 1000)
        b = orb.OrbBatchDev(ctx, ex, imgs)
        outs = []
        for _ in range(reps):
            b.run()
            outs.append(b.results())
        b.close(); ex.close(); ctx.close()
        return outs

    def pose_leg(reps):
        ctx = Context(0)
        outs = [optimizer.pose_optimization(ctx, pp["cam_qt"], pp["Xw"], pp["obs"], pp["info"], pp["K"]) for _ in range(reps)]
        ctx.close()
        return outs

    solo_g, solo_o, solo_p = _gba(prob)[0], orb_leg(1)[0], pose_leg(1)[0]
    before = coresidency_stats(0)
    g, o, p = _run_threads([lambda: _gba(prob, reps=2), lambda: orb_leg(40), lambda: pose_leg(400)])
    after = coresidency_stats(0)
    for r, x in enumerate(g):
        _same(x, solo_g, f"global BA repetition {r}")
    for x in o:
        assert len(x) == len(solo_o)
        for (ka, da), (kb, db) in zip(x, solo_o):
            assert ka.tobytes() == kb.tobytes() and np.array_equal(da, db), "ORB batch differs from the solo run"
    for cam, outl, nin in p:
        assert np.array_equal(cam, solo_p[0]) and np.array_equal(outl, solo_p[1]) and nin == solo_p[2], "pose optimisation differs from the solo run"
    assert after["aborted"] == before["aborted"], (before, after)


@pytest.mark.parametrize("free", [50, 60])
def test_local_ba_at_the_reference_window_size_beside_tracking(free):
    """ClientHandler.cpp:184: LocalMapping (local BA, 50 free + 20 fixed keyframes: conf/config.yaml:78-79) runs beside Tracking.  Since round 6 the 50-camera window is
    solved by ONE workgroup (ba_solve_cholreg) and needs no lease on the chip; a window of 60 free cameras (LocalMapSize + part of LocalMapBuffer) still takes the
    persistent solver and with it the per-device lease."""
    prob = synth.make_ba_config("lba_50", n_fixed=70 - free)
    img = synth.gen_image(7200, 3)
    pp = synth.make_pose_problem(250, seed=11)

    def lba(reps):
        ctx = Context(0)
        outs = []
        for _ in range(reps):
            cam, pts, erase, st1, st2 = optimizer.local_bundle_adjustment(ctx, prob)
            outs.append((cam, pts, erase, (st1.iters_done, st1.lm_trials, st2.iters_done, st2.lm_trials)))
        ctx.close()
        return outs

    def track(reps):
        ctx = Context(0)
        ex = orb.ORBextractor(ctx, 1000)
        outs = []
        for _ in range(reps):
            k, d = ex(img)
            c, o, n = optimizer.pose_optimization(ctx, pp["cam_qt"], pp["Xw"], pp["obs"], pp["info"], pp["K"])
            outs.append((k.tobytes(), d.tobytes(), c.tobytes(), o.tobytes(), n))
        ex.close(); ctx.close()
        return outs

    solo_l, solo_t = lba(1)[0], track(1)[0]
    before = coresidency_stats(0)
    l, t = _run_threads([lambda: lba(6), lambda: track(150)])
    after = coresidency_stats(0)
    for cam, pts, erase, st in l:
        assert np.array_equal(cam, solo_l[0]) and np.array_equal(pts, solo_l[1]) and np.array_equal(erase, solo_l[2]) and st == solo_l[3]
    assert all(x == solo_t for x in t), "tracking leg differs from its solo run"
    assert after["aborted"] == before["aborted"], (before, after)
    if free > 50: assert after["launches"] > before["launches"], "the 60-camera window did not take the persistent solver"
    else: assert after["launches"] == before["launches"], "the 50-camera window asked for the whole chip: it should be solved by one workgroup"


def test_a_handle_returns_to_the_persistent_solver_after_a_give_up(monkeypatch):
    """CCM_BA_TEST_ABORT makes every persistent launch give up (its last workgroup leaves at once): the trial is repeated on the multi-kernel
    solver, the handle cools down for a few trials and then TRIES AGAIN (round 5; before, one give-up was for good) — the result stays within the
    parity bar of the undisturbed run and the give-ups are counted."""
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=60, n_points=5000, seed=21)
    ref = _gba(prob, iters=12, resident=False)[0]
    monkeypatch.setenv("CCM_BA_TEST_ABORT", "1")
    before = coresidency_stats(0)
    got = _gba(prob, iters=12, resident=False)[0]
    after = coresidency_stats(0)
    assert got["st"][0] == ref["st"][0] and got["st"][1] == ref["st"][1] and got["st"][3] == ref["st"][3]
    dt, dr = synth.pose_errors(got["cam"], ref["cam"])
    assert dt.max() < 1e-5 and dr.max() < 1e-4
    n_abort = after["aborted"] - before["aborted"]
    assert n_abort >= 2, "the handle never went back to the persistent kernel after its first give-up"
    assert n_abort <= 2 + ref["st"][1] // 7   # one give-up, then kPersCooldownTrials - 1 further trials on the multi-kernel solver
