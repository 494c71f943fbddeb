#!/bin/bash
cd $GRAFT_REPO_ROOT
CCM_PROBE_QUIET=1 timeout 120 python scripts/gpu_probe_ba.py gba_c4 12 2>&1 | grep -E "run s|BA_SCHUR|BA_PCG"
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_sharded_loopback_gpu.py -x -q -m gpu 2>&1 | tail -4
