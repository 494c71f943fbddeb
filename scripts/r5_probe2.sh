#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_poseopt_gpu.py tests/test_shim_gpu.py -q -x 2>&1 | tail -5 ) > gpurun_out/r5b_tests.log 2>&1
( CCM_DBG=pers,trial timeout 200 python scripts/lba_profile.py lba_50 ) > gpurun_out/r5b_lba50_dbg.log 2>&1
( timeout 200 python scripts/lba_profile.py lba_50 ) > gpurun_out/r5b_lba50.log 2>&1
( timeout 300 python scripts/shim_gba_probe.py gba_c4 4 ) > gpurun_out/r5b_shim.log 2>&1
tail -3 gpurun_out/r5b_tests.log; tail -40 gpurun_out/r5b_lba50_dbg.log | cut -c1-400; tail -2 gpurun_out/r5b_lba50.log; tail -8 gpurun_out/r5b_shim.log
