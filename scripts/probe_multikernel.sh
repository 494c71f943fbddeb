#!/bin/bash
# development probe (GPU box): multi-kernel PCG after a change of its product kernel — the tests that run on that path, then the 10 000-keyframe bench leg with kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_sharded_loopback_gpu.py -m gpu -x -q -k "config5 or multi_kernel or abort or fallback or limit or fixture or shard or formulations or loop" 2>&1 | tail -5 ) > gpurun_out/mk_tests.log 2>&1
( timeout 300 python bench.py --gba-only --steps 3 --warmup 1 --workload gba_c5 ) > gpurun_out/mk_c5.log 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mk_prof -o c5 -- python $GRAFT_REPO_ROOT/bench.py --gba-only --steps 2 --warmup 1 --workload gba_c5 ) > $GRAFT_REPO_ROOT/gpurun_out/mk_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/mk_tests.log
python - <<'PY'
import json,glob,csv
for l in open('gpurun_out/mk_c5.log'):
    if l.startswith('{'):
        d=json.loads(l); print('c5 ms_per_step', d['ms_per_step'], d['config'].get('run_ms'))
for f in glob.glob('gpurun_out/mk_prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3)
PY
