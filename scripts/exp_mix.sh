#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "=== gba_c4 probe"; CCM_PROBE_QUIET=1 timeout 120 python scripts/gpu_probe_ba.py gba_c4 3 2>&1 | grep -E "run s|BA_|rror"
echo "=== lba kstats"; bash scripts/kstats.sh python scripts/lba_profile.py 2>&1 | tail -45
echo "=== ba tests"; timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -m gpu 2>&1 | tail -5
