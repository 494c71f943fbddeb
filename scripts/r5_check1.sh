#!/bin/bash
# round 5, first GPU pass: the new tests, then the whole suite, the bench line, the flip probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_concurrency_gpu.py tests/test_poseopt_gpu.py -q -x 2>&1 | tail -30 ) > gpurun_out/r5a_new_tests.log 2>&1
( time timeout 300 python scripts/poseopt_flip_probe.py ) > gpurun_out/r5a_flip.log 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_concurrency_gpu.py 2>&1 | tail -8 ) > gpurun_out/r5a_pytest.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r5a_bench.log 2>&1
grep -h '^{"metric"' gpurun_out/r5a_bench.log | tail -1 > gpurun_out/r5a_bench_line.json
tail -30 gpurun_out/r5a_new_tests.log; tail -3 gpurun_out/r5a_flip.log; tail -5 gpurun_out/r5a_pytest.log; tail -c 300 gpurun_out/r5a_bench_line.json
