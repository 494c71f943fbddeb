"""Generates the committed golden fixtures from the CPU oracle (oracle/) on seeded synthetic inputs.
The reference ships no golden vectors and cannot be compiled or imported here (C++ needing OpenCV/Eigen/
ROS), so these fixtures pin the ORACLE's outputs (and, through the GPU tests, the HIP path) across machines
and commits; they are not outputs of a running reference binary.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from ccm_slam_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    # ORB: one EuRoC-shaped frame, 1000 features
    img = synth.gen_image(1000, 0)
    kps, desc = oracle.OrbOracle(1000).extract(img)
    np.savez_compressed(os.path.join(OUT, "orb_seed1000_t0_n1000.npz"), img_sha=np.frombuffer(__import__("hashlib").sha256(img.tobytes()).digest(), np.uint8),
                        kps=kps, desc=desc)
    # Hamming
    d1, d2, _, _ = synth.make_descriptor_sets(500, 400, 42)
    bi, bd, sd = oracle.hamming_dense_best2(d2, d1)
    np.savez_compressed(os.path.join(OUT, "hamming_500x400_seed42.npz"), best_idx=bi, best_dist=bd, second_dist=sd)
    # local BA (config 2 shape): 5 iterations with Huber sqrt(5.991)
    prob = synth.make_ba_config("lba_c2")
    prob["huber_delta"] = float(np.float32(np.sqrt(np.float32(5.991))))
    cam, pts, chi2, dpos, st = oracle.ba_optimize(prob, 5)
    np.savez_compressed(os.path.join(OUT, "lba_c2_5iters.npz"), cam=cam, pts_head=pts[:200], chi2_hist=np.array([st.chi2_hist[i] for i in range(st.iters_done)]),
                        chi2_initial=st.chi2_initial, iters=st.iters_done, trials=st.lm_trials)
    # pose optimisation
    p = synth.make_pose_problem(300, 0, 0.1)
    pc, outl, ninl = oracle.pose_optimize(p["cam_qt"], p["Xw"], p["obs"], p["info"], p["K"])
    np.savez_compressed(os.path.join(OUT, "poseopt_n300_seed0.npz"), cam=pc, outlier=outl, n_inlier=ninl)
    print("golden fixtures written to", OUT)


PT_STRIDE = 37   # every 37th landmark of the final state is kept (the full point sets would be tens of MB)


def gba_full(name, iters=20):
    """One complete optimize(20) of a BASELINE global-BA configuration on the oracle, run to g2o's stop rule: the per-iteration
    chi2 / lambda / trial-count history, the stop reason and the final estimate.  gba_c4 takes ~70 s, gba_c3 ~35 s, gba_c5 ~10 min
    on one core of this container; run  python tests/golden/make_golden.py gba_c4 gba_c3 gba_c5  to regenerate."""
    prob = synth.make_ba_config(name)
    cam, pts, chi2, dpos, st = oracle.ba_optimize(prob, iters)
    n = st.iters_done
    np.savez_compressed(os.path.join(OUT, f"{name}_full.npz"), cam=cam, pts_sub=pts[::PT_STRIDE].copy(), pt_stride=PT_STRIDE,
                        chi2_hist=np.array([st.chi2_hist[i] for i in range(n)]), lambda_hist=np.array([st.lambda_hist[i] for i in range(n)]),
                        trials_hist=np.array([st.trials_hist[i] for i in range(n)], np.int32), chi2_initial=st.chi2_initial, chi2_final=st.chi2_final,
                        iters=n, trials=st.lm_trials, stop_reason=st.stop_reason, max_iters=iters, n_depth_nonpos=int((dpos == 0).sum()),
                        n_edge=prob["n_edge"])
    print(name, "iterations", n, "trials", st.lm_trials, "stop", st.stop_reason)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for nm in sys.argv[1:]:
            gba_full(nm)
    else:
        main()
