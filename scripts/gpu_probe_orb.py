"""Development probe: ORB / matcher / pose-opt timings on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from ccm_slam_amd import orb, synth, matcher, optimizer
from ccm_slam_amd._lib import hooks, Context, K

ctx = Context(0)
ex = orb.ORBextractor(ctx, 1000)
imgs = np.stack([synth.gen_image(1000, t) for t in range(16)])
for _ in range(3): ex(imgs[0])
t = time.perf_counter()
for i in range(16): kps, desc = ex(imgs[i])
dt = (time.perf_counter() - t) / 16
print(f"host-API extract: {dt*1e3:.3f} ms/frame = {1/dt:.1f} fps, n={len(kps)}")
import ctypes as C
from ccm_slam_amd._lib import lib
tm = (C.c_double * 6)(); hooks().ccm_orb_debug_timing(ex._h, tm)
print("  phases ms [queue1, wait cand, octree, queue2, wait+D2H, total]:", [round(x, 4) for x in tm])
ctx.prof_enable(-1); ctx.prof_reset()
b = orb.OrbBatchDev(ctx, ex, imgs)
b.run()
ctx.prof_reset()
t = time.perf_counter(); b.run(); ctx.sync(); dt = (time.perf_counter() - t) / 16
print(f"device-resident batch: {dt*1e3:.3f} ms/frame = {1/dt:.1f} fps, counts={b.counts()[:4]}")
for name in ("PYR_RESIZE", "FAST_SCORE", "FAST_NMS", "BLUR", "BRIEF"):
    n, ms = ctx.prof_read(K[name]); print(f"  {name:12s} launches {n:5d} total {ms:8.3f} ms avg {ms/max(n,1)*1e3:8.2f} us")
ctx.prof_enable(-2)
o = oracle.OrbOracle(1000)
t = time.perf_counter()
for i in range(4): o.extract(imgs[i])
print(f"oracle CPU: {(time.perf_counter()-t)/4*1e3:.2f} ms/frame")
# pose opt
p = synth.make_pose_problem(300, 0, 0.1)
for _ in range(3): optimizer.pose_optimization(ctx, p["cam_qt"], p["Xw"], p["obs"], p["info"], p["K"])
t = time.perf_counter()
for _ in range(20): optimizer.pose_optimization(ctx, p["cam_qt"], p["Xw"], p["obs"], p["info"], p["K"])
print(f"pose-opt n=300: {(time.perf_counter()-t)/20*1e3:.3f} ms")
t = time.perf_counter()
for _ in range(5): oracle.pose_optimize(p["cam_qt"], p["Xw"], p["obs"], p["info"], p["K"])
print(f"pose-opt oracle: {(time.perf_counter()-t)/5*1e3:.3f} ms")
# hamming dense 2000x2000 device resident
d1, d2, _, _ = synth.make_descriptor_sets(2000, 2000, 3)
m = matcher.DenseMatcherDev(ctx, d2, d1)
for _ in range(3): m.run()
ctx.sync(); t = time.perf_counter()
for _ in range(100): m.run()
ctx.sync(); dt = (time.perf_counter() - t) / 100
print(f"hamming dense 2000x2000: {dt*1e6:.1f} us -> {17*4e6/dt/1e12:.3f} Tops/s")
