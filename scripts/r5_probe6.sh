#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_orb_gpu.py tests/test_prepared_calls_gpu.py -q -x 2>&1 | tail -4 ) > gpurun_out/r5f_tests.log 2>&1
( WITH_TORCH=1 WARM_CALLS=3 timeout 200 python scripts/orb_batch_time.py ) > gpurun_out/r5f_orb.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( WITH_TORCH=1 WARM_CALLS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r5f_orb -o orb -- python scripts/orb_batch_time.py ) > gpurun_out/r5f_orb_prof.log 2>&1
python - <<'PY' > gpurun_out/r05b_orb_batch_kernel_stats.txt 2>&1
import csv, glob
f = glob.glob('gpurun_out/prof_r5f_orb/**/orb_kernel_stats.csv', recursive=True)
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r['TotalDurationNs']))[:8]:
    print(r['Name'][:48].ljust(48), 'calls', r['Calls'].rjust(6), 'total_us %10.1f' % (float(r['TotalDurationNs']) / 1e3), 'avg_us %9.2f' % (float(r['AverageNs']) / 1e3))
PY
cat gpurun_out/r5f_tests.log; cat gpurun_out/r5f_orb.log; cat gpurun_out/r05b_orb_batch_kernel_stats.txt
