"""GPU box, under scripts/kstats.sh: the two Hamming workloads of bench.py's `extra.hamming` (dense 2000 x 2000, windowed 5000 x ~30) for a rocprofv3 kernel table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ccm_slam_amd._lib import Context
ctx = Context(0)
print(bench.hamming_leg(ctx))
ctx.close()
