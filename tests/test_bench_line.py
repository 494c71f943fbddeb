"""bench.py's output contract (CPU): the last stdout line is ONE compact JSON line below 4 KB — round 5's 21 KB line was lost by the driver's parser —
and the long record goes to a side file.  The compacting function is fed the committed long records of earlier rounds."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LONG_RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]*_bench_line.json")))


@pytest.mark.parametrize("path", LONG_RECORDS[-6:], ids=lambda p: os.path.basename(p))
def test_compact_line_is_small_and_complete(path):
    import bench
    full = json.load(open(path))
    line = bench.compact_line(full, os.path.join(ROOT, "bench_extra.json"))
    assert "\n" not in line and len(line) < 4000
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "extra_file"):
        assert k in c, k
    assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"]
    assert c["config"]["workload"].startswith("gba_c")
    if full.get("roofline"):
        r = c["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s"
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    if full.get("cpu_baseline"):
        assert {"value", "unit", "cores", "kind", "sample"} <= set(c["cpu_baseline"])


def test_an_oversized_record_still_yields_a_small_line():
    import bench
    full = json.load(open(LONG_RECORDS[-1]))
    full["config"]["workload"] = "x" * 5000
    full["roofline"]["bound"] = "y" * 5000
    line = bench.compact_line(full, "bench_extra.json")
    json.loads(line)
    assert len(line) < 4000 + 5000   # the workload string is the caller's; everything optional is gone
    assert "roofline_trial" not in json.loads(line)


def test_stdout_is_one_json_line_and_stderr_is_empty():
    """the part of the contract that runs without a GPU: --plumbing-only goes through the same descriptor handling as the full run"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-only"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stderr == ""
    lines = out.stdout.splitlines()
    assert len(lines) == 1
    assert json.loads(lines[-1])["plumbing"] == "ok"
