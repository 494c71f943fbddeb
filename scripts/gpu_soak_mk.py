"""Soak test (GPU box): the multi-kernel PCG path with the coarse level forced on, repeated runs bit-identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CCM_BA_NO_PERSIST"] = "1"
os.environ["CCM_BA_COARSE"] = "always"
import numpy as np
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
prob = synth.make_ba_config("gba_c4")
h = optimizer.BAHandle(ctx, prob)
ref = None
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    h.reset()
    st = h.run(6)
    cam, pts, _, _ = h.download()
    sig = (st.lm_trials, st.pcg_iters, st.chi2_final)
    if ref is None: ref = (sig, cam.copy(), pts.copy())
    else: assert sig == ref[0] and np.array_equal(cam, ref[1]) and np.array_equal(pts, ref[2]), (r, sig, ref[0])
print("multi-kernel + coarse:", ref[0], "bit-identical over all runs")
h.close()
