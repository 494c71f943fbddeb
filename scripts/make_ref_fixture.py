"""Build container only (needs /root/reference -> oracle/_ref/libg2o_ref.so): the REFERENCE'S OWN g2o (thirdparty/g2o compiled verbatim, graph built as
Optimizer::MapFusionGBA builds it, oracle/ref_g2o_driver.cpp) on a BASELINE-sized global bundle adjustment, run to g2o's own stop rule, stored as
tests/golden/<name>_ref.npz: chi2 after every LM iteration, trials per iteration, the final lambda and chi2, every camera of the final estimate and every
50th landmark.  tests/test_golden.py (oracle, CPU) and tests/test_ba_gpu.py (MI355X) compare against THIS file, i.e. against the reference itself and not
against the oracle's restatement of it.  Takes minutes (the look-alike Eigen's SimplicialLDLT is slow).  usage: make_ref_fixture.py [gba_c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ccm_slam_amd import synth
from oracle import ref

name = sys.argv[1] if len(sys.argv) > 1 else "gba_c3"
prob = synth.make_ba_config(name)
t0 = time.time()
cam, pts, chi2, dpos, st = ref.g2o_ba_optimize(prob, 20)
n = st.n_hist
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"{name}_ref.npz")
np.savez_compressed(out, workload=name, iters_done=st.iters_done, lm_trials=st.lm_trials, chi2_hist=np.array(st.chi2_hist[:n]),
                    trials_hist=np.array(st.trials_hist[:n], np.int32), chi2_final=st.chi2_final, lambda_final=st.lambda_final, cam=cam, pts_every_50th=pts[::50],
                    generator="scripts/make_ref_fixture.py: libg2o_ref.so = /root/reference/cslam/thirdparty/g2o compiled verbatim (oracle/Makefile.ref)")
print(f"{name}: {st.iters_done} iterations / {st.lm_trials} trials, chi2 {st.chi2_final:.9g}, trials per iteration {list(st.trials_hist[:n])}, {time.time() - t0:.0f} s -> {out}")
