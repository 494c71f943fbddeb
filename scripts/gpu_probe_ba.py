"""Development probe (run on the GPU box): BA vs oracle with verbose output and timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from ccm_slam_amd import optimizer, synth, matcher
from ccm_slam_amd._lib import Context, K

ctx = Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=60, n_points=5000, seed=21)
    iters = 8
elif which == "lba":
    prob = synth.make_ba_config("lba_c2"); iters = 15
else:
    prob = synth.make_ba_config(which); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print("problem", which, prob["n_cam"], prob["n_pt"], prob["n_edge"], flush=True)
t = time.time()
h = optimizer.BAHandle(ctx, prob)
print("create s", time.time() - t, h.counts(), flush=True)
ctx.prof_enable(-2 if os.environ.get("CCM_PROBE_NOPROF") else -1)
ctx.prof_reset()
t = time.time()
st = h.run(iters, verbose=1)
print("run s", time.time() - t, "iters", st.iters_done, "trials", st.lm_trials, "pcg", st.pcg_iters, "chi", st.chi2_initial, st.chi2_final,
      "ms_setup", st.ms_setup, "ms_iters", st.ms_iters, "reason", st.stop_reason, flush=True)
for name, k in K.items():
    n, ms = ctx.prof_read(k)
    if n:
        print(f"  {name:16s} launches {n:6d} total {ms:9.3f} ms avg {ms/n*1e3:9.2f} us")
cam, pts, chi2, dpos = h.download()
if prob["n_edge"] < 200000:
    t = time.time()
    ocam, opts, ochi2, odpos, ost = oracle.ba_optimize(prob, iters)
    print("oracle s", time.time() - t, "iters", ost.iters_done, "trials", ost.lm_trials, "chi", ost.chi2_initial, ost.chi2_final)
    print(" oracle chi hist", [ost.chi2_hist[i] for i in range(ost.iters_done)])
    dt, dr = synth.pose_errors(cam, ocam)
    print(" max pose diff t", dt.max(), "r", dr.max(), "pts", np.abs(pts - opts).max(), "chi2 edge diff", np.abs(chi2 - ochi2).max())
h.close()
