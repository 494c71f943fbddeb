"""ms per frame of ccm_orb_extract_batch_dev on 64 resident frames (best of 5 calls); the octree runs on the device, frames go out in groups of four per launch."""
import os
import sys
import time

import numpy as np

if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401  (bench.py imports torch first: its bundled HIP runtime is the one the process then uses)
    torch.cuda.init()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccm_slam_amd import orb, synth  # noqa: E402
from ccm_slam_amd._lib import Context  # noqa: E402

ctx = Context(0)
imgs = np.stack([synth.gen_image(1000, t) for t in range(64)])
ex = orb.ORBextractor(ctx, 1000)
b = orb.OrbBatchDev(ctx, ex, imgs)
b.run()
for r in range(int(os.environ.get("WARM_CALLS", "0"))):       # the clocks of an idle box take a few hundred ms of load to come up
    b.run()
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    b.run()
    best = min(best, time.perf_counter() - t0)
print("ms/frame %.4f" % (best * 1e3 / 64), "keypoints", int(b.counts().sum()))
