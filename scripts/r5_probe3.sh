#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -X faulthandler -m pytest tests/test_shim_real_gpu.py -v -x 2>&1 | grep -E "PASSED|FAILED|Fatal|Error|assert|^E |passed|failed|File \"/root|tests/" | head -60 ) > gpurun_out/r5c_tests.log 2>&1
( timeout 300 python scripts/shim_gba_probe.py gba_c4 3 2>&1 | grep -v "rand seed\|Initialized" ) > gpurun_out/r5c_shim.log 2>&1
cat gpurun_out/r5c_tests.log; tail -12 gpurun_out/r5c_shim.log | cut -c1-700
