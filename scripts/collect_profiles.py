#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/profile.sh (gpurun_out/prof_<tag>/) into the committed summaries under profiles/:
  <tag>_gba_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary of the profiled command, verbatim
  <tag>_gba_kernels_by_grid.csv     per (kernel, grid size): launches, average duration (kernel trace), FETCH_SIZE and
                                    WRITE_SIZE per launch (separate PMC passes), HBM bytes per launch
  <tag>_gba_command.txt             the profiled command
  pmc_latest.json                   the same table as JSON — what bench.py's `traffic` fields are read from
Launches are keyed by kernel AND grid size, so kernels that also run on other problem sizes never mix into the figures.
usage: python scripts/collect_profiles.py r02a            (here, after gpurun merged gpurun_out/ back)
       python scripts/collect_profiles.py r02a --on-box   (on the GPU box: writes the compact aggregate only)
"""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
on_box = "--on-box" in sys.argv
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def first(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    return f[0] if f else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"([A-Za-z_0-9]+)(?:<[^>]*>)?\(", name)
    return m.group(1) if m else name.strip('"')


agg_path = os.path.join(src, "by_grid.json")
if on_box or not os.path.exists(agg_path):
    table = collections.defaultdict(lambda: {"launches": 0, "dur_ns": 0.0, "fetch_n": 0, "fetch_kib": 0.0, "write_n": 0, "write_kib": 0.0})
    t = first("trace/**/*kernel_trace.csv")
    for r in csv.DictReader(open(t)):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        e = table[key]
        e["launches"] += 1
        e["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for which in ("fetch", "write"):
        f = first(f"pmc_{which}/**/*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            e = table[key]
            e[which + "_n"] += 1
            e[which + "_kib"] += float(r["Counter_Value"])
    rows = [{"kernel": k, "grid": g, **v} for (k, g), v in table.items()]
    json.dump(rows, open(agg_path, "w"))
    # the per-dispatch CSVs of the PMC passes are tens of MB: drop them once aggregated (the stats CSV stays)
    if on_box:
        for f in glob.glob(os.path.join(src, "pmc_*/**/*counter_collection.csv"), recursive=True) + \
                 glob.glob(os.path.join(src, "pmc_*/**/*kernel_trace.csv"), recursive=True):
            os.remove(f)
        print("aggregated", len(rows), "(kernel, grid) rows ->", agg_path)
        sys.exit(0)

rows = json.load(open(agg_path))
cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "?"
shutil.copy(first("trace/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_gba_kernel_stats.csv"))
open(os.path.join(dst, f"{tag}_gba_command.txt"), "w").write(cmd + "\n")
bl = os.path.join(src, "bench_line_profiled_run.json")
if os.path.exists(bl) and os.path.getsize(bl):
    shutil.copy(bl, os.path.join(dst, f"{tag}_gba_bench_line_profiled_run.json"))
out = []
for r in sorted(rows, key=lambda r: -r["dur_ns"]):
    fetch = r["fetch_kib"] / r["fetch_n"] if r["fetch_n"] else None
    write = r["write_kib"] / r["write_n"] if r["write_n"] else None
    hbm = int((2.0 * (fetch or 0.0) + (write or 0.0)) * 1024) if (fetch is not None or write is not None) else None
    out.append({"kernel": r["kernel"], "grid": r["grid"], "launches": r["launches"], "avg_us": round(r["dur_ns"] / r["launches"] / 1e3, 3),
                "total_ms": round(r["dur_ns"] / 1e6, 3), "fetch_kib_raw_per_launch": None if fetch is None else round(fetch, 2),
                "write_kib_raw_per_launch": None if write is None else round(write, 2), "hbm_bytes_per_launch": hbm})
with open(os.path.join(dst, f"{tag}_gba_kernels_by_grid.csv"), "w") as f:
    f.write("kernel,grid_threads,launches,avg_us,total_ms,FETCH_SIZE_KiB_raw_per_launch,WRITE_SIZE_KiB_raw_per_launch,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
    for e in out:
        f.write(",".join(str(e[k]) for k in ("kernel", "grid", "launches", "avg_us", "total_ms", "fetch_kib_raw_per_launch", "write_kib_raw_per_launch",
                                              "hbm_bytes_per_launch")) + "\n")
workload = re.search(r"--workload (\S+)", cmd)
try:
    import subprocess
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = ""
json.dump({"tag": tag + (f" @ {head}" if head else ""), "source": f"rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (three separate passes) over `{cmd}` ({tag})",
           "workload": workload.group(1) if workload else "gba_c4",
           "correction": "hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, "
                         "both reported by rocprofv3 in KiB; WRITE_SIZE and narrow / gathered reads are uncalibrated per the same section",
           "kernels": out}, open(os.path.join(dst, "pmc_latest.json" if not workload or workload.group(1) == "gba_c4" else f"pmc_{workload.group(1)}.json"), "w"), indent=1)
print("wrote", len(out), "rows for", tag)
