"""Pose optimisation alone (300 edges, as a tracked frame has them), 60 calls: for scripts/kstats.sh; prints the host-API time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pp = synth.make_pose_problem(n, 0)
best = 1e9
for r in range(60):
    t0 = time.perf_counter()
    out = optimizer.pose_optimization(ctx, pp["cam_qt"].copy(), pp["Xw"], pp["obs"], pp["info"], pp["K"])
    best = min(best, time.perf_counter() - t0)
print("pose_opt n=%d: %.4f ms per call (best of 60)" % (n, best * 1e3))
