"""Offline studies of the global-BA linear solve on the ORACLE's dense reduced systems (CPU only: oracle + numpy; nothing here is part of the product or of the tests).
DESIGN.md section 4.1 ("Candidates the round's measurements leave for the next one") quotes the figures these print.

  python scripts/offline/precond_study.py warm  [kfs_per_agent]   retry trials warm-started from x(lambda) with an absolute target
  python scripts/offline/precond_study.py ns    [kfs_per_agent]   spectral radius of I - Ac_new Ac_old^-1 (Newton-Schulz refresh of the coarse inverse)
  python scripts/offline/precond_study.py defl  [kfs_per_agent]   previous LM steps / exact slow modes as extra coarse columns; spectrum of M^-1 A
  python scripts/offline/precond_study.py adef  [kfs_per_agent]   additive two-level (the product) vs A-DEF2 / A-DEF1 hybrids
  python scripts/offline/precond_study.py f32   [kfs_per_agent]   S rounded to f32 inside the product, f64 residual replacement every R iterations

The preconditioner is the product's: 16-camera cluster-Jacobi + one rigid-body twist per node, nodes every 16 cameras, hat-function interpolation, prolongation by Ad(T_cw)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle
from ccm_slam_amd import synth

AGG, CL = 16, 96
what = sys.argv[1] if len(sys.argv) > 1 else "adef"
KF = int(sys.argv[2]) if len(sys.argv) > 2 else 120
prob = synth.make_ba_problem(n_agents=4, kfs_per_agent=KF, n_points=75 * KF, seed=11)
NPT = int(prob["n_pt"])


def state(iters):
    cam, pts, _, _, _ = oracle.ba_optimize(prob, iters)
    p = dict(prob); p["cam_qt"] = cam; p["pt_xyz"] = pts
    return p, cam


def system(p, lam, damped=True):
    H, b, _ = oracle.ba_partial_system(p, lam, 0, NPT, damped)
    return H, b


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def Pmat(cam, n):
    nc = n // 6
    free = np.where(np.asarray(prob["cam_fixed"]) == 0)[0]
    na = (nc + AGG - 1) // AGG
    P = np.zeros((n, 6 * (na + 1)))
    for k, ci in enumerate(free):
        R = quat_R(cam[ci, :4]); t = cam[ci, 4:]
        Ad = np.zeros((6, 6)); Ad[:3, :3] = R; Ad[3:, 3:] = R; Ad[3:, :3] = skew(t) @ R
        a = k // AGG; w1 = ((k % AGG) + 0.5) / AGG
        P[6 * k:6 * k + 6, 6 * a:6 * a + 6] += (1 - w1) * Ad
        P[6 * k:6 * k + 6, 6 * (a + 1):6 * (a + 1) + 6] += w1 * Ad
    return P


def levels(A, P):
    n = A.shape[0]
    Ws = [np.linalg.inv(A[s:s + CL, s:s + CL]) for s in range(0, n, CL)]
    W = lambda r: np.concatenate([Ws[i] @ r[s:s + CL] for i, s in enumerate(range(0, n, CL))])
    Aci = np.linalg.inv(P.T @ A @ P)
    Q = lambda r: P @ (Aci @ (P.T @ r))
    return W, Q


def pcg(A, b, M, x0, target, cap=2000):
    x = x0.copy(); r = b - A @ x; z = M(r); p = z.copy(); rz = r @ z; it = 0
    while np.sqrt(abs(rz)) > target and it < cap:
        q = A @ p; al = rz / (p @ q); x += al * p; r -= al * q; z = M(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return x, it


H0, _ = system(state(0)[0], 0.0, False)
LAM0 = 1e-5 * np.max(np.diag(H0))     # g2o's first lambda (tau * max diag)

if what == "warm":
    p, cam = state(3)
    for mult, nus in ((1.0, (2, 4, 8, 16, 32)), (100.0, (2, 4, 8)), (1e4, (2, 4, 8))):
        lam = LAM0 * mult
        H, b = system(p, lam); W, Q = levels(H, Pmat(cam, H.shape[0])); M = lambda r: W(r) + Q(r)
        x, it = pcg(H, b, M, np.zeros_like(b), 1e-8 * np.sqrt(b @ M(b)))
        print(f"lambda {lam:.3g}: cold {it} iterations", flush=True)
        for nu in nus:
            lam *= nu
            H, b = system(p, lam); W, Q = levels(H, Pmat(cam, H.shape[0])); M = lambda r, W=W, Q=Q: W(r) + Q(r)
            tgt = 1e-8 * np.sqrt(b @ M(b))
            xc, itc = pcg(H, b, M, np.zeros_like(b), tgt)
            _, itw = pcg(H, b, M, x, tgt)
            r0 = b - H @ x
            print(f"   -> x{nu}: cold {itc}, warm {itw}  (|r0|_M / |b|_M = {np.sqrt(r0 @ M(r0)) / np.sqrt(b @ M(b)):.3g}, |x' - x| / |x'| = {np.linalg.norm(xc - x) / np.linalg.norm(xc):.3g})", flush=True)
            x = xc
elif what == "ns":
    (p3, c3), (p4, c4) = state(3), state(4)
    Ac = lambda p, cam, lam: (lambda H: Pmat(cam, H.shape[0]).T @ H @ Pmat(cam, H.shape[0]))(system(p, lam)[0])
    rho = lambda A, X: np.max(np.abs(np.linalg.eigvals(np.eye(len(X)) - A @ X)))
    for mult in (1.0, 100.0):
        lam = LAM0 * mult
        X = np.linalg.inv(Ac(p3, c3, lam))
        for nu in (0.33, 3.0, 16.0, 64.0):
            print(f"same linearisation, lambda {lam:.3g} -> x{nu}: rho = {rho(Ac(p3, c3, lam * nu), X):.3g}", flush=True)
        print(f"next linearisation, same lambda: rho = {rho(Ac(p4, c4, lam), X):.3g};  lambda x0.33: rho = {rho(Ac(p4, c4, lam * 0.33), X):.3g}", flush=True)
elif what == "defl":
    steps, lam = [], LAM0
    for it in range(7):
        p, cam = state(it)
        H, b = system(p, lam); P = Pmat(cam, H.shape[0])
        res = []
        for k in (0, 1, 2, 4):
            if k and len(steps) < 1:
                continue
            Pa = P if k == 0 else np.concatenate([P, (lambda e: e / np.linalg.norm(e, axis=0))(np.stack(steps[-k:], 1))], 1)
            W, Q = levels(H, Pa); M = lambda r, W=W, Q=Q: W(r) + Q(r)
            x, n_it = pcg(H, b, M, np.zeros_like(b), 1e-8 * np.sqrt(b @ M(b)))
            res.append(n_it)
            if k == 0:
                x_keep = x
        print(f"LM iteration {it}, lambda {lam:.3g}: hats only {res[0]}; + last 1 / 2 / 4 steps as coarse columns: {res[1:]}", flush=True)
        steps.append(x_keep); lam /= 3.0
    p, cam = state(4); lam = LAM0 / 81
    H, b = system(p, lam); P = Pmat(cam, H.shape[0]); n = H.shape[0]
    W, Q = levels(H, P); M = lambda r: W(r) + Q(r)
    Minv = np.stack([M(np.eye(n)[:, j]) for j in range(n)], 1); Minv = 0.5 * (Minv + Minv.T)
    L = np.linalg.cholesky(Minv)
    w, V = np.linalg.eigh(L.T @ H @ L)
    print("spectrum of M^-1 A: 5 smallest", np.round(w[:5], 4), "largest %.3g" % w[-1])
    _, n0 = pcg(H, b, M, np.zeros(n), 1e-8 * np.sqrt(b @ M(b)))
    for k in (2, 4, 8, 16):
        e = L @ V[:, :k]
        Wk, Qk = levels(H, np.concatenate([P, e / np.linalg.norm(e, axis=0)], 1)); Mk = lambda r: Wk(r) + Qk(r)
        print(f"   {k} exact slowest modes deflated: {pcg(H, b, Mk, np.zeros(n), 1e-8 * np.sqrt(b @ Mk(b)))[1]} iterations (hats only: {n0})", flush=True)
elif what == "f32":
    # S as f32 inside the product, f64 residual replacement every R iterations (round 5 costing of VERDICT r4 item 3): iterations and final error against the f64 solve
    def pcg_f32(A, A32, b, M, target, R, cap=2000):
        x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; it = 0
        while np.sqrt(abs(rz)) > target and it < cap:
            q = A32 @ p; al = rz / (p @ q); x += al * p
            it += 1
            if R and it % R == 0: r = b - A @ x          # the true residual, one f64 product
            else: r -= al * q
            z = M(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn
        return x, it
    p, cam = state(3)
    for mult in (1.0, 10.0, 100.0, 1e4):
        lam = LAM0 * mult
        H, b = system(p, lam)
        D = H - lam * np.eye(len(b))
        H32 = D.astype(np.float32).astype(np.float64) + lam * np.eye(len(b))     # the kernel adds lambda p in f64 beside the f32 blocks
        W, Q = levels(H, Pmat(cam, H.shape[0])); M = lambda r, W=W, Q=Q: W(r) + Q(r)
        tgt = 1e-8 * np.sqrt(b @ M(b))
        xe = np.linalg.solve(H, b)
        x64, it64 = pcg(H, b, M, np.zeros_like(b), tgt)
        row = [f"lambda {lam:.3g}: f64 {it64} it (err {np.linalg.norm(x64 - xe) / np.linalg.norm(xe):.1e})"]
        for R in (0, 4, 8, 10, 12, 14, 16):
            x, it = pcg_f32(H, H32, b, M, tgt, R)
            true_res = b - H @ x
            row.append(f"R={R}: {it} it, err {np.linalg.norm(x - xe) / np.linalg.norm(xe):.1e}, true |r|_M/|b|_M {np.sqrt(true_res @ M(true_res)) / np.sqrt(b @ M(b)):.1e}")
        print(" | ".join(row), flush=True)
else:
    for it, mult in ((0, 1.0), (3, 1 / 27.0), (5, 1 / 243.0), (5, 1.0), (5, 30.0)):
        p, cam = state(it); lam = LAM0 * mult
        H, b = system(p, lam); n = H.shape[0]
        W, Q = levels(H, Pmat(cam, n))
        Madd = lambda r: W(r) + Q(r)
        Mdef2 = lambda r: (lambda z1: z1 + Q(r - H @ z1))(W(r))       # local solve, then the coarse correction of what is left
        Mdef1 = lambda r: (lambda y: y + W(r - H @ y))(Q(r))          # coarse first
        tgt = 1e-8 * np.sqrt(b @ Madd(b))
        xa, ka = pcg(H, b, Madd, np.zeros(n), tgt, 500)
        x2, k2 = pcg(H, b, Mdef2, Q(b), tgt, 500)
        x1, k1 = pcg(H, b, Mdef1, np.zeros(n), tgt, 500)
        print(f"state {it}, lambda {lam:.3g}: additive {ka}, A-DEF2 {k2}, A-DEF1 {k1} iterations; |x_def2 - x_add| / |x| = {np.linalg.norm(x2 - xa) / np.linalg.norm(xa):.1e}", flush=True)

