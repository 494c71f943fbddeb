"""GPU (single device): the RCCL plumbing of the sharded global BA.  A 1-rank communicator is attached and
CCM_FORCE_ALLREDUCE=1 makes libccm_hip issue every ncclAllReduce of the sharded path (sum over [S | b_schur],
max for lambda_0, sum of the trial scalars) on it; the result must equal the run without a communicator.
Multi-rank behaviour is covered on CPU by tests/test_sharding_gloo.py; the driver exercises N = 2,4,8."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context, comm_unique_id
prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=30, n_points=2000, seed=8)
ctx = Context(0)
cam0, pts0, _, _, st0 = optimizer.bundle_adjustment(ctx, prob, 4)
ctx.comm_init(1, 0, comm_unique_id())
h = optimizer.BAHandle(ctx, prob, rank=0, nranks=1)
st1 = h.run(4)
cam1, pts1, _, _ = h.download()
assert st1.iters_done == st0.iters_done and st1.lm_trials == st0.lm_trials
assert np.array_equal(cam0, cam1) and np.array_equal(pts0, pts1), "forced single-rank all-reduce changed the result"
print("COMM_OK", st1.chi2_final)
''' % ROOT


def test_single_rank_communicator_runs_every_collective():
    env = dict(os.environ, CCM_FORCE_ALLREDUCE="1")
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "COMM_OK" in out.stdout, out.stdout + out.stderr
