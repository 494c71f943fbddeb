#!/bin/bash
# development probe (GPU box): a global-BA bench leg (WORKLOAD, default gba_c4) and the local-BA profile with every library in ab_libs/ on the SAME box, interleaved twice
cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-gba_c4}
cp ccm_slam_amd/libccm_hip.so /tmp/cur.so
for rep in 1 2; do
for f in ab_libs/libccm_hip_*.so; do
  cp $f ccm_slam_amd/libccm_hip.so
  echo "== $f"
  python bench.py --gba-only --steps 20 --warmup 3 --workload $W 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', d['ms_per_step'], 'create', d['config'].get('create_ms'), 'run', d['config'].get('run_ms'))"
  python scripts/lba_profile.py lba_50 2>&1 | tail -1
  python scripts/lba_profile.py lba_c2 2>&1 | tail -1
done
done
cp /tmp/cur.so ccm_slam_amd/libccm_hip.so
