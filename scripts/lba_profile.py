"""Local-BA leg alone (lba_c2, 8 calls as bench.py's local_ba_leg makes them), for `rocprofv3 --kernel-trace --stats`:
per-kernel times of a 30-free-camera window.  Prints the best wall time and the phase split of the last call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import optimizer, synth  # noqa: E402
from ccm_slam_amd._lib import Context  # noqa: E402

ctx = Context(0)
prob = synth.make_ba_config(sys.argv[1] if len(sys.argv) > 1 else "lba_c2")   # lba_50: the reference's configured window
best = 1e9
for r in range(9):
    t0 = time.perf_counter()
    h = optimizer.BAHandle(ctx, prob)
    t1 = time.perf_counter()
    st = h.run(15)
    t2 = time.perf_counter()
    h.download()
    h.close()
    t3 = time.perf_counter()
    if r:
        best = min(best, t3 - t0)
print("local_ba_ms", round(best * 1e3, 3), "last call: create %.3f run %.3f download+close %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3),
      "iters", st.iters_done, "trials", st.lm_trials)
ctx.close()
