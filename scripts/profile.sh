#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + HBM PMC passes for bench.py.  Outputs under gpurun_out/prof_$1
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $CMD > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
python - <<PY
import csv, glob, collections, os
out = "$OUT"
def load(pattern):
    f = glob.glob(os.path.join(out, pattern), recursive=True)
    return f[0] if f else None
st = load("trace/**/*kernel_stats.csv")
if st:
    print("== kernel stats ==")
    for i, l in enumerate(open(st)):
        if i < 25: print(l.rstrip())
for name in ("pmc_fetch", "pmc_write"):
    f = load(name + "/**/*counter_collection.csv")
    if not f: print("no counter csv for", name); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")[:60]
        agg[k][0] += 1; agg[k][1] += float(r.get("Counter_Value", 0))
    print("==", name, "(sum of Counter_Value per kernel; units per rocprofv3: KiB) ==")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f"{k:60s} dispatches {n:6d} total {v:14.1f} per-dispatch {v/n:12.2f}")
PY
