"""Development probe (GPU box): phase clocks of ccm_ba_create (CCM_DBG=1) on a warm handle, from HBM-resident arrays like bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
prob = synth.make_ba_config(sys.argv[1] if len(sys.argv) > 1 else "gba_c4")
res = optimizer.ResidentProblem(ctx, prob)
for k in range(4):
    print(f"--- create {k}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    h = optimizer.BAHandle(ctx, prob, resident=res)
    print(f"--- create {k}: {(time.perf_counter() - t0) * 1e3:.3f} ms", file=sys.stderr, flush=True)
    h.close()
