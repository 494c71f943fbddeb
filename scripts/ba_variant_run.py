"""GPU box: one complete optimize(20) of a synthetic multi-agent map with whatever CCM_BA_* switches the environment carries; prints ONE JSON line
(iterations, trials per iteration, chi2 after every iteration, CG iterations, the optimised poses' checksum and the poses themselves) — what
tests/test_ba_gpu.py compares across the formulations of the large-map path (compact observation records / stored Hpl blocks, coarse intervals of
16 / 24 / 32 cameras, blocked / column-wise Cholesky tiles: the switches are read once per process, hence a process per variant).
usage: ba_variant_run.py <n_agents> <kfs_per_agent> <n_points> <seed>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context

na, kf, npt, seed = (int(v) for v in sys.argv[1:5])
ctx = Context(0)
prob = synth.make_ba_problem(n_agents=na, kfs_per_agent=kf, n_points=npt, seed=seed)
h = optimizer.BAHandle(ctx, prob)
st = h.run(20)
chi, lam, tr = h.history()
cam, pts, chi2, dpos = h.download()
print(json.dumps({"iters": int(st.iters_done), "trials": [int(v) for v in tr[:st.iters_done]], "chi2": [float(v) for v in chi[:st.iters_done]],
                  "pcg_iters": int(st.pcg_iters), "counts": h.counts(), "cam": cam.ravel().tolist(), "pts_sum": float(np.abs(pts).sum())}))
h.close()
