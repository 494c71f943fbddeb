"""GPU box: Optimizer::MapFusionGBA / LocalBundleAdjustmentClient through the drop-in shim TU (shim/Optimizer_hip.cpp) on BASELINE-sized maps,
with the shim's phase clocks (graph walk / flatten / create / run / download / write-back).  usage: shim_gba_probe.py [gba_c4] [reps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ccm_slam_amd import synth
from oracle import mapgraph as mg

PH = ("walk", "flatten", "create", "run", "download", "kf_writeback", "mp_writeback", "total", "vertices", "release", "scope_exit_frees", "unaccounted")


def phases(lib):
    out = (C.c_double * 12)()
    lib.ccm_shim_phases(out, 12)
    return {k: round(v, 3) for k, v in zip(PH, out)}


def gba(name, reps):
    prob = synth.make_ba_config(name)
    flat = mg.flat_from_ba_problem(prob, n_agents={"gba_c3": 3, "gba_c4": 4, "gba_c5": 8}.get(name, 1))
    libs = [("look-alike", mg.SHIM_LIB)]
    real = mg.SHIM_LIB.replace(".so", "_real.so")
    if os.path.exists(real):
        libs.append(("REAL classes", real))
    for tag, path in libs:
        graphs = [mg.MapGraph(path, flat) for _ in range(reps)]     # all graphs first: a graph freed right before a call leaves the heap trimmed
        for r, g in enumerate(graphs):
            t0 = time.perf_counter()
            rc = g.map_fusion_gba(0, 20)
            dt = time.perf_counter() - t0
            print(f"{name} MapFusionGBA [{tag}] rep {r}: rc {rc} wall {1e3 * dt:.2f} ms phases {phases(g.lib)}", flush=True)
        for g in graphs:
            g.close()


def lba(reps):
    prob = synth.make_ba_config("lba_c2")
    flat = mg.flat_from_ba_problem(prob)
    for r in range(reps):
        g = mg.MapGraph(mg.SHIM_LIB, flat)
        t0 = time.perf_counter()
        rc = g.local_ba(15, client_id=0)
        dt = time.perf_counter() - t0
        print(f"lba_c2 LocalBundleAdjustmentClient rep {r}: rc {rc} wall {1e3 * dt:.2f} ms phases {phases(g.lib)}", flush=True)
        g.close()


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "gba_c4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    gba(name, reps)
    lba(reps)
