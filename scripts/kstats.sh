#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of an arbitrary command, printed compactly.  usage: kstats.sh <cmd...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kst && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o p -- "$@" > /tmp/kst.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)
if not f: print("no stats; log tail:"); print(open("/tmp/kst.log").read()[-1500:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    print(f'{n[:48]:48s} calls {int(r["Calls"]):6d} total_us {float(r["TotalDurationNs"])/1e3:10.1f} avg_us {float(r["AverageNs"])/1e3:9.2f}')
# idle time before each kernel (gap since the previous kernel ended), summed by the kernel that follows the gap
t = glob.glob("/tmp/kst/**/*kernel_trace.csv", recursive=True)
if t:
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]) for r in csv.DictReader(open(t[0]))))
    import collections
    gap = collections.defaultdict(lambda: [0, 0.0])
    busy = 0.0
    for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
        g = max(0, s1 - e0) / 1e3
        if g < 20000: gap[n1][0] += 1; gap[n1][1] += g
    for s0, e0, n0 in ev: busy += (e0 - s0) / 1e3
    print(f"== span {(ev[-1][1]-ev[0][0])/1e3:.1f} us, kernels busy {busy:.1f} us; idle gaps by following kernel (gaps > 20 ms ignored) ==")
    for n, (c, g) in sorted(gap.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"{n:40s} gaps {c:5d} total_us {g:10.1f} avg_us {g/c:8.2f}")
PY
