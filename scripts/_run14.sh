cd $GRAFT_REPO_ROOT
NOPROF=1 bash scripts/full_check.sh r03b > gpurun_out/r03b_full.log 2>&1
bash scripts/profile.sh r03b 20 5 > gpurun_out/r03b_profile.log 2>&1
python scripts/shim_gba_probe.py gba_c4 3 2>&1 | grep -v "^-" > gpurun_out/r03b_shim_probe.log
