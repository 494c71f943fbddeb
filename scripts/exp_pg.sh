#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import oracle
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
os.environ["CCM_PG_DBG"] = "1"
for n in (5000, 10000):
    pg = synth.make_pose_graph(n, 0, covis=6)
    optimizer.pose_graph_optimization(ctx, pg, max_iters=1)
    t0 = time.perf_counter(); s, st = optimizer.pose_graph_optimization(ctx, pg); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); so, sto = oracle.pose_graph_optimize(pg); tc = time.perf_counter() - t0
    print(f"n={n} edges={pg['n_edge']}: gpu {tg*1e3:.1f} ms (iters {st.iters_done}, trials {st.lm_trials}) oracle {tc*1e3:.1f} ms (iters {sto.iters_done}, trials {sto.lm_trials}); chi2 {st.chi2_final:.6g} / {sto.chi2_final:.6g}; max diff {np.abs(s-so).max():.2e}", flush=True)
PY
