"""GPU box: how close to the oracle's own decision boundary must a residual sit before the device's pose optimisation decides differently?
For every planted case of tests/test_poseopt_gpu.py's sweep (one observation moved to s* (1 +- delta), s* = the scale at which the ORACLE's flag of that edge
flips, found by bisection to 1e-13), count the calls whose outlier flags or inlier count differ from the oracle's, per delta.  Output: one JSON line."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from ccm_slam_amd import optimizer, synth  # noqa: E402
from ccm_slam_amd._lib import Context  # noqa: E402

spec = importlib.util.spec_from_file_location("tp", os.path.join(ROOT, "tests", "test_poseopt_gpu.py"))
tp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tp)
oracle.build()
ctx = Context(0)
deltas = [1e-6, 1e-7, 1e-8, 1e-9, 1e-10, 1e-11, 1e-12, 1e-13]
calls = {d: 0 for d in deltas}
flips = {d: 0 for d in deltas}
planted = 0
for k, n, seed, of in tp._sweep_cases():
    if n < 6:
        continue
    p = synth.make_pose_problem(n, seed, of, pose_sigma_t=0.05 * (1 + 2 * (k % 2)), pose_sigma_r_deg=1.0 * (1 + 2 * (k % 2)))
    j = (7 * k) % n
    pl = tp._plant(oracle, p, j)
    if pl is None:
        continue
    planted += 1
    base, off, lo, hi = pl
    for d in deltas:
        for s in (lo * (1 - d), hi * (1 + d)):
            obs = p["obs"].copy(); obs[j] = base + s * off
            cam, outl, ninl = optimizer.pose_optimization(ctx, p["cam_qt"], p["Xw"], obs, p["info"], p["K"])
            ocam, ooutl, oninl = oracle.pose_optimize(p["cam_qt"], p["Xw"], obs, p["info"], p["K"])
            calls[d] += 1
            if ninl != oninl or not np.array_equal(outl, ooutl):
                flips[d] += 1
print(json.dumps({"planted_cases": planted, "calls_per_delta": calls[deltas[0]], "calls_with_a_differing_flag": {f"{d:g}": flips[d] for d in deltas}}))
