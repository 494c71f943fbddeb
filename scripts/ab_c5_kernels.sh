#!/bin/bash
# development probe (GPU box): per-kernel averages of the 10 000-keyframe leg (one profiled run of 2 steps) for every library in ab_libs/, same box
cd $GRAFT_REPO_ROOT
cp ccm_slam_amd/libccm_hip.so /tmp/cur.so
for f in ab_libs/libccm_hip_*.so; do
  cp $f ccm_slam_amd/libccm_hip.so
  echo "== $f"
  bash scripts/kstats.sh python bench.py --gba-only --steps 2 --warmup 1 --workload gba_c5 2>&1 | head -${AB_LINES:-3} | cut -c1-120
done
cp /tmp/cur.so ccm_slam_amd/libccm_hip.so
