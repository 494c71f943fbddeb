import time, numpy as np, sys
sys.path.insert(0, "/root/repo")
import oracle
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context, K, lib
import ctypes as C
ctx = Context(0)
for n in (60, 150, 400):
    p = synth.make_sim3_problem(n, 0)
    args = (p["sim3"], p["P1c"], p["P2c"], p["obs1"], p["obs2"], p["info1"], p["info2"], p["K1"], p["K2"], 10.0, False)
    s, inl, nin = optimizer.sim3_optimization(ctx, *args)
    so, inlo, nino = oracle.sim3_optimize(*args)
    t0 = time.perf_counter()
    for _ in range(50): optimizer.sim3_optimization(ctx, *args)
    tg = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(20): oracle.sim3_optimize(*args)
    tc = (time.perf_counter() - t0) / 20
    print(f"n={n}: gpu {tg*1e3:.3f} ms  oracle {tc*1e3:.3f} ms  maxdiff {np.abs(s-so).max():.2e}  nin {nin}/{nino}")
