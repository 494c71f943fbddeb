#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hamming_gpu.py tests/test_host_mirror_gpu.py tests/test_shim_matcher_gpu.py tests/test_frame_gpu.py -q -x 2>&1 | tail -25 ) > gpurun_out/r5d_tests.log 2>&1
( timeout 300 python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from ccm_slam_amd._lib import Context
ctx = Context(0)
print(json.dumps(bench.hamming_leg(ctx), indent=1))
PY
) > gpurun_out/r5d_hamming.log 2>&1
cat gpurun_out/r5d_tests.log; cat gpurun_out/r5d_hamming.log | cut -c1-400
