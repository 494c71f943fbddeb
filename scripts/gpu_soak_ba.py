"""Soak test (GPU box): repeat the global-BA and local-BA solves many times on one context and check that every run is
bit-identical to the first and that the persistent kernel never fell back to the multi-kernel path."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import Context, K

ctx = Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, iters in (("gba_c4", 12), ("lba_c2", 15)):
    prob = synth.make_ba_config(name)
    h = optimizer.BAHandle(ctx, prob)
    ctx.prof_enable(K["BA_PCG_SPMV"]); ctx.prof_reset()
    ref = None
    t0 = time.time()
    for r in range(reps):
        h.reset()
        st = h.run(iters)
        cam, pts, _, _ = h.download()
        sig = (st.lm_trials, st.pcg_iters, st.chi2_final, float(cam.sum()), float(pts.sum()))
        if ref is None: ref = (sig, cam.copy(), pts.copy())
        else:
            assert sig == ref[0], (r, sig, ref[0])
            assert np.array_equal(cam, ref[1]) and np.array_equal(pts, ref[2]), r
    n_spmv, _ = ctx.prof_read(K["BA_PCG_SPMV"])
    ctx.prof_enable(-2)
    h.close()
    print(f"{name}: {reps} runs bit-identical ({ref[0][0]} trials, {ref[0][1]} CG iterations, chi2 {ref[0][2]:.6f}), "
          f"multi-kernel PCG launches {n_spmv}, {1e3*(time.time()-t0)/reps:.1f} ms per run", flush=True)
    assert n_spmv == 0
    # the structure is built on the device (round 3): a FRESH handle per run must reproduce the same bits (stable sorts, integer atomics only)
    t0 = time.time()
    for r in range(max(reps // 4, 3)):
        h2 = optimizer.BAHandle(ctx, prob)
        st = h2.run(iters)
        cam, pts, _, _ = h2.download()
        h2.close()
        sig = (st.lm_trials, st.pcg_iters, st.chi2_final, float(cam.sum()), float(pts.sum()))
        assert sig == ref[0] and np.array_equal(cam, ref[1]) and np.array_equal(pts, ref[2]), (r, sig, ref[0])
    print(f"{name}: {max(reps // 4, 3)} freshly built handles bit-identical to the first, {1e3*(time.time()-t0)/max(reps // 4, 3):.1f} ms per create + run + download", flush=True)
