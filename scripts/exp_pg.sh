#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_posegraph_gpu.py tests/test_shim_gpu.py -x -q -m gpu 2>&1 | tail -12
echo "=== tiles (default)"; CCM_PG_DBG=1 timeout 300 python scripts/gpu_probe_posegraph.py 2>&1 | tail -8
echo "=== leaf 31"; CCM_PG_ND_LEAF=31 CCM_PG_DBG=1 timeout 300 python scripts/pg_profile.py 2000 2>&1 | tail -3
echo "=== leaf 127"; CCM_PG_ND_LEAF=127 CCM_PG_DBG=1 timeout 300 python scripts/pg_profile.py 2000 2>&1 | tail -3
bash scripts/kstats.sh python scripts/pg_profile.py 2000 2>&1 | tail -32
