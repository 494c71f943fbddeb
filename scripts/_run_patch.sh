cd /root/repo
python -m pytest tests/test_shim_gpu.py -q 2>&1 | tail -3
python - <<'PY'
import os, sys, time
sys.path.insert(0, '/root/repo')
from ccm_slam_amd import synth
from oracle import mapgraph as mg
from scripts.shim_gba_probe import phases
prob = synth.make_ba_config("gba_c4")
flat = mg.flat_from_ba_problem(prob, n_agents=4)
for lib in (mg.SHIM_LIB, mg.SHIM_LIB.replace(".so", "_patched.so")):
    gs = [mg.MapGraph(lib, flat) for _ in range(3)]
    for r, g in enumerate(gs):
        t0 = time.perf_counter(); rc = g.map_fusion_gba(0, 20); dt = time.perf_counter() - t0
        print(os.path.basename(lib), r, f"wall {1e3*dt:.1f}", phases(g.lib), flush=True)
    for g in gs: g.close()
PY
