#!/bin/bash
# development probe (GPU box): the 10 000-keyframe leg with every library in ab_libs/ on the SAME box (boxes differ by up to 20 % on latency-bound kernels)
cd $GRAFT_REPO_ROOT
cp ccm_slam_amd/libccm_hip.so /tmp/cur.so
for f in ab_libs/libccm_hip_*.so; do
  cp $f ccm_slam_amd/libccm_hip.so
  echo "== $f"
  bash scripts/kstats.sh python bench.py --gba-only --steps 2 --warmup 1 --workload gba_c5 2>&1 | head -${AB_LINES:-3} | cut -c1-120
  python bench.py --gba-only --steps 3 --warmup 1 --workload gba_c5 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('ms_per_step', json.loads(l)['ms_per_step'])"
done
cp /tmp/cur.so ccm_slam_amd/libccm_hip.so
