cd $GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | head -80 ) > gpurun_out/run1_all.log 2>&1
