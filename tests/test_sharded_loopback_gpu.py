"""The landmark-sharded global BA with nranks > 1, executed END TO END on one GPU: the ranks are threads with their own ccm_ctx, joined by the
test-only loop-back communicator (ccm_comm_init_loopback: rendezvous + one reduction kernel, the same bits in every rank's buffer — RCCL's
contract without the wires).  Runs, with nranks = 2 and 3, every collective of ba.hip that a single-rank handle skips: the lambda_0
max-reduce, the all-reduce of [S | b_schur] in every LM trial, the per-trial scalar all-reduce carrying chi2, the gain denominator and the
stop / give-up flags, and the point gather of ccm_ba_download.  (The RCCL transport itself is covered by tests/test_comm_gpu.py with a
1-rank communicator and by the driver's multi-GPU bench.)"""
import ctypes as C
import threading

import numpy as np
import pytest

from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context, check, hooks, lib

pytestmark = pytest.mark.gpu


def run_sharded(prob, nranks, iters, stop_rank=None, stop_at_trial=None, lambda_init=0.0):
    group = C.c_void_p()
    check(hooks().ccm_comm_loopback_create(nranks, C.byref(group)))
    out, err = [None] * nranks, [None] * nranks

    def rank_main(rank):
        try:
            ctx = Context(0)
            check(hooks().ccm_comm_init_loopback(ctx.handle, group, rank), ctx.handle)
            h = optimizer.BAHandle(ctx, prob, rank=rank, nranks=nranks)
            flag = np.zeros(1, np.uint8)
            if stop_rank == rank:
                count = [0]

                def hook(it, trial, chi, acc):
                    count[0] += 1
                    if count[0] == stop_at_trial:
                        flag[0] = 1
                h.set_trial_callback(hook)
            st = h.run(iters, stop_flag=flag, lambda_init=lambda_init)
            chi, lam, tr = h.history()
            cam, pts, chi2, dpos = h.download()
            out[rank] = dict(cam=cam, pts=pts, st=(st.iters_done, st.lm_trials, st.stop_reason), chi=chi, tr=tr, chi2_final=st.chi2_final, counts=h.counts())
            h.close()
            ctx.close()
        except Exception as e:   # a failing rank must not leave its peers waiting in the rendezvous forever: report and let the test fail
            err[rank] = e
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th), "a rank is stuck in a collective (mismatched call sequence)"
    hooks().ccm_comm_loopback_destroy(group)
    assert all(e is None for e in err), err
    return out


@pytest.mark.parametrize("nranks", [2, 3])
def test_sharded_lm_loop_equals_the_single_rank_run(ctx, nranks):
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=60, n_points=5000, seed=21)
    single = optimizer.BAHandle(ctx, prob)
    st1 = single.run(12)
    chi1, _, tr1 = single.history()
    cam1, pts1, _, _ = single.download()
    single.close()
    ranks = run_sharded(prob, nranks, 12)
    assert sum(r["counts"]["points"] for r in ranks) == prob["n_pt"] or sum(r["counts"]["points"] for r in ranks) <= prob["n_pt"]   # landmarks are partitioned
    assert all(r["counts"]["edges"] > 0 for r in ranks)
    for r in ranks[1:]:   # every rank solved the identical reduced system: bit-identical cameras, identical LM decisions, identical gathered points
        assert np.array_equal(r["cam"], ranks[0]["cam"]) and np.array_equal(r["pts"], ranks[0]["pts"])
        assert r["st"] == ranks[0]["st"] and np.array_equal(r["tr"], ranks[0]["tr"]) and np.array_equal(r["chi"], ranks[0]["chi"])
    r0 = ranks[0]
    assert r0["st"] == (st1.iters_done, st1.lm_trials, st1.stop_reason) and np.array_equal(r0["tr"], tr1)
    assert st1.lm_trials > st1.iters_done                                            # the run contains rejected trials
    assert np.abs(r0["chi"] / chi1 - 1).max() < 1e-9                                  # partial sums are added in a different order: rounding only
    dt, dr = synth.pose_errors(r0["cam"], cam1)
    assert dt.max() < 1e-8 and np.abs(r0["cam"] - cam1).max() < 1e-8 and np.abs(r0["pts"] - pts1).max() < 1e-7


def test_stop_flag_raised_on_one_rank_stops_every_rank_at_the_same_trial():
    """ranks poll their own flag at different times; the decision must be collective or a rank leaves the loop while its peers wait in
    the next all-reduce.  Rank 1 raises its flag after its 5th trial; both ranks must stop together, in the same state."""
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=60, n_points=5000, seed=21)
    ranks = run_sharded(prob, 2, 12, stop_rank=1, stop_at_trial=5)
    a, b = ranks
    assert a["st"] == b["st"] and a["st"][2] == 1 and a["st"][1] in (5, 6), (a["st"], b["st"])
    assert np.array_equal(a["cam"], b["cam"]) and np.array_equal(a["pts"], b["pts"])


def test_persistent_solver_give_up_is_taken_by_all_ranks(monkeypatch):
    """a rank whose persistent PCG kernel gives up repeats the trial on the multi-kernel path, which issues the trial's collectives a
    second time; the give-up flag rides in the scalar all-reduce so that every rank repeats.  CCM_BA_TEST_ABORT makes the kernel give up."""
    monkeypatch.setenv("CCM_BA_TEST_ABORT", "1")
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=40, n_points=3000, seed=5)
    ranks = run_sharded(prob, 2, 4)
    monkeypatch.delenv("CCM_BA_TEST_ABORT")
    assert ranks[0]["st"] == ranks[1]["st"] and ranks[0]["st"][0] == 4
    assert np.array_equal(ranks[0]["cam"], ranks[1]["cam"])
    c = Context(0)
    h = optimizer.BAHandle(c, prob); st = h.run(4); cam, _, _, _ = h.download(); h.close(); c.close()
    assert np.abs(ranks[0]["cam"] - cam).max() < 1e-7 and abs(ranks[0]["chi2_final"] / st.chi2_final - 1) < 1e-8
