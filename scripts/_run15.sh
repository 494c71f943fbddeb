cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_golden.py tests/test_shim_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/run15_tests.log 2>&1
python scripts/orb_batch_time.py > gpurun_out/run15_orb.log 2>&1
for t in 1 8; do echo "== threads $t"; CCM_SHIM_THREADS=$t python scripts/shim_gba_probe.py gba_c4 3 2>&1 | grep "gba_c4"; done > gpurun_out/run15_shim.log 2>&1
