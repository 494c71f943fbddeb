#!/bin/bash
# Runs on the GPU box: builds scripts/write_size_probe.hip and prints WRITE_SIZE (KiB, as rocprofv3 reports it) per kernel next to the bytes really written.
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/wsp $GRAFT_REPO_ROOT/scripts/write_size_probe.hip || exit 1
rm -rf /tmp/wsp_out && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/wsp_out -o w -- /tmp/wsp > /tmp/wsp.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/wsp_out/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
true_kib = 65694 * 288 / 1024
for k, v in acc.items():
    print(f"{k:20s} WRITE_SIZE {sum(v)/len(v):10.1f} KiB per launch   written {true_kib:10.1f} KiB   ratio {sum(v)/len(v)/true_kib:5.2f}")
PY
