#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -m gpu 2>&1 | tail -4
CCM_PROBE_NOPROF=1 CCM_PROBE_QUIET=1 timeout 300 python scripts/gpu_probe_ba.py gba_c5 20 2>&1 | grep -E "run s"
CCM_PROBE_NOPROF=1 CCM_PROBE_QUIET=1 bash scripts/kstats.sh python scripts/gpu_probe_ba.py gba_c5 20 2>&1 | head -8
