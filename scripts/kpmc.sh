#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass (counters in $1, space separated) over a command; per-kernel mean counter values
CNT="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kpmc && timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/kpmc -o p -- "$@" > /tmp/kpmc.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kpmc/**/*counter_collection.csv", recursive=True)
if not f: print(open("/tmp/kpmc.log").read()[-1500:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:30]
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    if not any(x in k for x in ("schur", "persist", "linearize", "backsub")): continue
    print(k, " ".join(f"{c}={v[1]/v[0]:.3g}" for c, v in cs.items()))
PY
