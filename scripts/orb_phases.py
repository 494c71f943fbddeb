"""Host-clock phase split of one ORB extraction (upload+queue | wait for candidates | octree on the host | queue phase 2 | wait + D2H | total), best of 50 frames."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import orb, synth  # noqa: E402
from ccm_slam_amd._lib import hooks, Context, lib  # noqa: E402

ctx = Context(0)
ex = orb.ORBextractor(ctx, 1000)
img = synth.gen_image(1000, 0)
acc = []
for i in range(60):
    ex(img)
    ph = (C.c_double * 6)()
    hooks().ccm_orb_debug_timing(ex._h, ph)
    if i >= 10:
        acc.append(list(ph))
a = np.array(acc)
print("median ms: upload+queue %.3f  wait_cand %.3f  octree %.3f  queue2 %.3f  wait+D2H %.3f  total %.3f" % tuple(np.median(a, 0)))
print("min    ms: upload+queue %.3f  wait_cand %.3f  octree %.3f  queue2 %.3f  wait+D2H %.3f  total %.3f" % tuple(a.min(0)))
