#!/bin/bash
# Runs on the GPU box: bench.py --gba-only for a list of workloads x values of ONE environment variable; prints ms per call, CG iterations, coarse builds.
# usage: sweep_env.sh VAR "v1 v2 ..." "gba_c4 gba_c3" [steps]
VAR=$1; VALS=$2; WLS=$3; STEPS=${4:-10}
for wl in $WLS; do for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --gba-only --workload $wl --steps $STEPS --warmup 2 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
k = {x['class']: x for x in d['kernels']}
print('$wl $VAR=$v', 'ms/call', d['ms_per_step'], 'run', c['run_ms'], 'cg', c['pcg_iters_per_step'], 'trials', c['trials_per_iteration'], 'coarse builds', k.get('ba_coarse', {}).get('launches_per_call'), 'persist avg', k.get('ba_pcg_persist', {}).get('avg_us'))"
done; done
