#!/bin/bash
# Runs on the GPU box: kernel timeline (start offset, duration, gap to the previous kernel) of the last $1 kernels of a command
N=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kst && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kst -o p -- "$@" > /tmp/kst.log 2>&1
python - $N <<'PY'
import csv, glob, sys
n = int(sys.argv[1])
t = glob.glob("/tmp/kst/**/*kernel_trace.csv", recursive=True)
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:36]) for r in csv.DictReader(open(t[0]))))
ev = ev[-n:]
t0 = ev[0][0]; prev = ev[0][0]
for s, e, name in ev:
    print(f"{(s-t0)/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  gap {(s-prev)/1e3:7.1f}  {name}")
    prev = e
PY
