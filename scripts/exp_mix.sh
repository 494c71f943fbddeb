#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_shim_gpu.py tests/test_frame_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== batch device octree"; timeout 120 python scripts/orb_batch_time.py 2>&1 | tail -1
echo "== batch host octree"; CCM_ORB_HOST_OCTREE=1 timeout 120 python scripts/orb_batch_time.py 2>&1 | tail -1
bash scripts/kstats.sh python scripts/orb_batch_time.py 2>&1 | head -12
