"""Does a busy GPU run the single-workgroup pose-optimisation kernel faster (clock / power state)?  A second host thread keeps a dense Hamming match (2000 x 2000, ~16 us per
launch, all CUs) running on its own context while the main thread times ccm_pose_optimize; compare with the idle-GPU figure of scripts/poseopt_profile.py."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccm_slam_amd import matcher, optimizer, synth
from ccm_slam_amd._lib import Context

ctx = Context(0)
pp = synth.make_pose_problem(300, 0)


def timed(label):
    best = 1e9
    for r in range(80):
        t0 = time.perf_counter()
        optimizer.pose_optimization(ctx, pp["cam_qt"].copy(), pp["Xw"], pp["obs"], pp["info"], pp["K"])
        best = min(best, time.perf_counter() - t0)
    print("%s: pose_opt n=300 %.4f ms per call (best of 80)" % (label, best * 1e3), flush=True)


timed("idle GPU")
stop = False
ctx2 = Context(0)
rng = np.random.default_rng(1)
q = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
m = matcher.DenseMatcherDev(ctx2, q, q)


def spin():
    while not stop:
        for _ in range(50):
            m.run()
        ctx2.sync()


th = threading.Thread(target=spin)
th.start()
time.sleep(0.3)
timed("busy GPU (dense Hamming looping on another stream)")
stop = True
th.join()
timed("idle GPU again")
