#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "=== $*"; env "$@" CCM_PROBE_QUIET=1 timeout 120 python scripts/gpu_probe_ba.py gba_c4 3 2>&1 | grep -E "run s|BA_SCHUR|rror|row kernel" ; }
run CCM_BA_ROW_DBG=128
run CCM_BA_ROW_DBG=240
