#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a few bench steps; prints the duration of every ba_pcg_persist launch of the LAST step, in order.
# usage: persist_trace.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ptr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptr -o p -- python bench.py --gba-only --steps 3 --warmup 1 "$@" > /tmp/ptr.log 2>&1
python - <<'PY'
import csv, glob
t = glob.glob("/tmp/ptr/**/*kernel_trace.csv", recursive=True)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(t[0]))))
pers = [(s, e) for s, e, n in rows if "ba_pcg_persist" in n]
last = pers[-18:]
print("persist us (last step):", [round((e - s) / 1e3, 1) for s, e in last], "sum", round(sum(e - s for s, e in last) / 1e3, 1))
# gap before each persist launch
idx = {s: i for i, (s, e, n) in enumerate(rows)}
gaps = []
for s, e in last:
    i = idx[s]
    gaps.append(round((s - rows[i - 1][1]) / 1e3, 1))
print("idle gap before it, us:", gaps)
PY
grep '^{' /tmp/ptr.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'cg', d['config']['pcg_iters_per_step'])"
