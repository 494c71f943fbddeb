#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "=== $*"; env "$@" timeout 200 python bench.py --gba-only --steps 10 --warmup 2 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('ms_per_step',d['ms_per_step'],'pcg',c['pcg_iters_per_step'],'iters',c['lm_iterations_per_step'],'trials',c['lm_trials_per_step'],'chi2',c['chi2_final']); print([ (k['class'],k['launches_per_call'],k['avg_us']) for k in d['kernels'][:4]])
"; }
run CCM_X=0
run CCM_BA_COARSE_WIN=16
run CCM_BA_COARSE_WIN=64
run CCM_BA_COARSE_ON=60 CCM_BA_COARSE_OFF=25
run CCM_BA_COARSE_ON=60 CCM_BA_COARSE_OFF=25 CCM_BA_COARSE_WIN=16
run CCM_BA_COARSE_ON=120 CCM_BA_COARSE_OFF=35
