"""GPU box: one optimize(20) of a workload with the per-trial log (lambda, chi2, PCG iterations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
prob = synth.make_ba_config(sys.argv[1] if len(sys.argv) > 1 else "gba_c4")
h = optimizer.BAHandle(ctx, prob)
st = h.run(20, verbose=1)
print("iters", st.iters_done, "trials", st.lm_trials, "pcg", st.pcg_iters, "ms", st.ms_iters)
