#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "=== lba dense2 phases"; CCM_BA_DENSE2_DBG=1 CCM_PROBE_QUIET=1 timeout 120 python scripts/gpu_probe_ba.py lba 2>&1 | grep -E "exact two-cluster|run s|BA_PCG_PERSIST|BA_SCHUR"
echo "=== bench"; ( time timeout 600 python bench.py ) > gpurun_out/r02i_bench.log 2>&1
grep -h '^{"metric"' gpurun_out/r02i_bench.log | tail -1 > gpurun_out/r02i_bench_line.json
tail -c 900 gpurun_out/r02i_bench_line.json; tail -4 gpurun_out/r02i_bench.log
