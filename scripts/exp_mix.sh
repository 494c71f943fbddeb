#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 2 3 4 5 6 8; do
  echo "== waves $w"
  CCM_POSE_WAVES=$w timeout 60 python scripts/poseopt_profile.py 300 2>&1 | tail -1
  CCM_POSE_WAVES=$w timeout 60 python scripts/poseopt_profile.py 1000 2>&1 | tail -1
done
timeout 300 python -m pytest tests/test_poseopt_gpu.py -x -q -m gpu 2>&1 | tail -2
