#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ba_gpu.py -q -x -k "gba_c5 or limit or large_map or fixture" 2>&1 | tail -6 ) > gpurun_out/r5e_tests.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/r5e_bench.log 2>&1
grep -h '^{"metric"' gpurun_out/r5e_bench.log | tail -1 > gpurun_out/r5e_bench_line.json
cat gpurun_out/r5e_tests.log; tail -c 300 gpurun_out/r5e_bench_line.json
