#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_orb_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== single"; timeout 120 python scripts/orb_phases.py 2>&1 | tail -2
echo "== batch"; timeout 120 python scripts/orb_batch_time.py 2>&1 | tail -1
bash scripts/kstats.sh python scripts/orb_batch_time.py 2>&1 | head -3
