"""Development probe (GPU box): local BA end to end as LocalMapping calls it (two-stage: optimize(5) + levels + optimize(10), create + run + download + tear-down),
for the named window (lba_c2: 30 free cameras, lba_50: the reference's configured window).  usage: gpu_probe_lba.py [lba_c2|lba_50] [reps]
With CCM_DBG=cholreg / dense2 / pers the solvers print their phase clocks at handle destruction."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context
name = sys.argv[1] if len(sys.argv) > 1 else "lba_c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = Context(0)
prob = synth.make_ba_config(name)
best = None
for rep in range(reps):
    t0 = time.perf_counter()
    _cam, _pts, erase, st1, st2 = optimizer.local_bundle_adjustment(ctx, prob)
    dt = (time.perf_counter() - t0) * 1e3
    if rep: best = dt if best is None else min(best, dt)
    print(f"{name} rep {rep}: {dt:.3f} ms, {st1.iters_done}+{st2.iters_done} iterations / {st1.lm_trials}+{st2.lm_trials} trials, erase {int(erase.sum())}, chi2 {st2.chi2_final:.6f}, setup {st1.ms_setup:.3f} ms, runs {st1.ms_total:.3f} + {st2.ms_total:.3f} ms", flush=True)
tr = st1.lm_trials + st2.lm_trials
print(f"{name}: best {best:.3f} ms = {best / tr:.4f} ms per trial ({tr} trials), solver {'default' if os.environ.get('CCM_BA_CHOLREG', '1') != '0' else 'CCM_BA_CHOLREG=0'}")
