#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/profile.sh (gpurun_out/prof_<tag>/) into the committed summaries under profiles/:
  <tag>_bench_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary, verbatim
  <tag>_bench_pmc_{fetch,write}_per_kernel.csv   per-kernel sums of the FETCH_SIZE / WRITE_SIZE passes (separate runs)
  pmc_latest.json                       HBM bytes per launch per kernel, gfx950 correction applied (FETCH_SIZE x2)
usage: python scripts/collect_profiles.py r01c
"""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def first(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


shutil.copy(first("trace/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))


def short(name):
    m = re.search(r"([A-Za-z_0-9]+)(?:<[^>]*>)?\(", name)
    return m.group(1) if m else name.strip('"')


per = {}
for which in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(first(f"pmc_{which}/**/*counter_collection.csv"))):
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(os.path.join(dst, f"{tag}_bench_pmc_{which}_per_kernel.csv"), "w") as f:
        f.write("kernel,dispatches,counter_sum_KiB,per_dispatch_KiB\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f'"{k}",{n},{v:.1f},{v / n:.2f}\n')
            per.setdefault(short(k), {})[f"{which}_kib_raw"] = round(v / n, 2)
for k, v in per.items():
    v["hbm_bytes_per_launch"] = int((2.0 * v.get("fetch_kib_raw", 0.0) + v.get("write_kib_raw", 0.0)) * 1024)
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 5 --warmup 1 "
                     f"--no-cpu-baseline`, gba_c4, round 1 ({tag})",
           "correction": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads), WRITE_SIZE as is",
           "kernels": per}, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
print("wrote", sorted(per))
