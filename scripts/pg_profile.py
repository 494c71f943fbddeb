"""Pose-graph optimisation of a 2000-keyframe essential graph alone (for `rocprofv3 --kernel-trace --stats`, scripts/kstats.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import optimizer, synth  # noqa: E402
from ccm_slam_amd._lib import Context  # noqa: E402
ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pg = synth.make_pose_graph(n, 0, covis=6)
optimizer.pose_graph_optimization(ctx, pg, max_iters=1)
t0 = time.perf_counter()
s, st = optimizer.pose_graph_optimization(ctx, pg)
print(f"n={n}: {(time.perf_counter() - t0) * 1e3:.1f} ms, iters {st.iters_done}, trials {st.lm_trials}")
