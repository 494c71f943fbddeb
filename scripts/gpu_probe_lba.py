"""Development probe (GPU box): local BA end to end (handle creation + solve + download), repeated like LocalMapping does."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
ctx = Context(0)
prob = synth.make_ba_config("lba_c2")
for rep in range(6):
    t0 = time.perf_counter()
    h = optimizer.BAHandle(ctx, prob)
    t1 = time.perf_counter()
    st = h.run(15)
    t2 = time.perf_counter()
    h.download(); h.close()
    t3 = time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.2f} ms (ms_setup {st.ms_setup:.2f}) run {1e3*(t2-t1):.2f} ms download+close {1e3*(t3-t2):.2f} ms", flush=True)
