"""Development probe (GPU box): ccm_ba_create cost on the global-BA workload, first and repeated calls."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
prob = synth.make_ba_config(sys.argv[1] if len(sys.argv) > 1 else "gba_c4")
for rep in range(3):
    t0 = time.perf_counter(); h = optimizer.BAHandle(ctx, prob); t1 = time.perf_counter(); h.close()
    print(f"rep {rep}: create {1e3*(t1-t0):.2f} ms", flush=True)
