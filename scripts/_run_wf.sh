cd /root/repo
python -m pytest tests/test_ba_gpu.py tests/test_ba_structure_gpu.py tests/test_sharded_loopback_gpu.py tests/test_shim_gpu.py tests/test_golden.py tests/test_host_mirror_gpu.py -x -q 2>&1 | tail -6
python scripts/gpu_soak_ba.py 2>&1 | tail -4
