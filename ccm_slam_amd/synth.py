"""Synthetic EuRoC-shaped inputs for the ORB / matcher / BA hot path.

EuRoC bags are not available offline, so every test and bench run uses the seeded generators in
this module.  Intrinsics are the reference's EuRoC cam0 values (cslam/conf/vi_euroc.yaml:9-12),
image size 752x480, ORB scale pyramid 8 levels x 1.2 (cslam/conf/config.yaml:38-51).

Nothing here touches the GPU or the oracle: the generators only produce numpy arrays laid out the
way include/ccm_hip.h expects them.
"""
from __future__ import annotations

import numpy as np

EUROC_K = (458.654, 457.296, 367.215, 248.375)  # fx fy cx cy
IMG_W, IMG_H = 752, 480
N_LEVELS = 8
SCALE = 1.2


def scale_tables(nlevels: int = N_LEVELS, scale: float = SCALE):
    """mvScaleFactor / mvLevelSigma2 / inverse tables in f32, as ORBextractor's ctor builds them
    (cslam/src/ORBextractor.cpp:584-600)."""
    sf = np.empty(nlevels, np.float32)
    s2 = np.empty(nlevels, np.float32)
    sf[0] = 1.0
    s2[0] = 1.0
    f = np.float32(scale)
    for i in range(1, nlevels):
        sf[i] = np.float32(sf[i - 1] * f)
        s2[i] = np.float32(sf[i] * sf[i])
    return sf, (np.float32(1.0) / sf).astype(np.float32), s2, (np.float32(1.0) / s2).astype(np.float32)


# --------------------------------------------------------------------------------------------
# small SO(3) helpers (f64, vectorised over leading axis)
# --------------------------------------------------------------------------------------------
def quat_from_R(R: np.ndarray) -> np.ndarray:
    """Rotation matrices (n,3,3) -> unit quaternions (n,4) as x y z w with w >= 0."""
    R = np.asarray(R, np.float64)
    n = R.shape[0]
    q = np.empty((n, 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    for k in range(n):
        m = R[k]
        t = tr[k]
        if t > 0:
            s = np.sqrt(t + 1.0)
            w = 0.5 * s
            s = 0.5 / s
            q[k] = ((m[2, 1] - m[1, 2]) * s, (m[0, 2] - m[2, 0]) * s, (m[1, 0] - m[0, 1]) * s, w)
        else:
            i = 0
            if m[1, 1] > m[0, 0]:
                i = 1
            if m[2, 2] > m[i, i]:
                i = 2
            j = (i + 1) % 3
            kk = (j + 1) % 3
            s = np.sqrt(m[i, i] - m[j, j] - m[kk, kk] + 1.0)
            c = np.zeros(3)
            c[i] = 0.5 * s
            s = 0.5 / s
            w = (m[kk, j] - m[j, kk]) * s
            c[j] = (m[j, i] + m[i, j]) * s
            c[kk] = (m[kk, i] + m[i, kk]) * s
            q[k] = (c[0], c[1], c[2], w)
    neg = q[:, 3] < 0
    q[neg] *= -1
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def R_from_quat(q: np.ndarray) -> np.ndarray:
    q = np.asarray(q, np.float64)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def rodrigues(w: np.ndarray) -> np.ndarray:
    """axis-angle (n,3) -> rotation matrices (n,3,3)"""
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w, axis=1)
    K = np.zeros((w.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    th_safe = np.where(th < 1e-12, 1.0, th)
    a = np.where(th < 1e-12, 1.0, np.sin(th_safe) / th_safe)[:, None, None]
    b = np.where(th < 1e-12, 0.5, (1 - np.cos(th_safe)) / (th_safe * th_safe))[:, None, None]
    return np.eye(3)[None] + a * K + b * (K @ K)


# --------------------------------------------------------------------------------------------
# bundle-adjustment scenes
# --------------------------------------------------------------------------------------------
def _agent_loop(n_kf: int, agent: int, loop_len: int | None = None):
    """world->camera poses (R (n,3,3), t (n,3)) of one agent on a closed elliptical loop, looking
    along the tangent; ~0.15 m between consecutive keyframes.  With loop_len > n_kf only the first
    n_kf keyframes of a loop_len-keyframe loop are produced (an open trajectory segment)."""
    n_all = n_kf if loop_len is None else max(loop_len, n_kf)
    per = 0.15 * n_all
    A = per / (2 * np.pi) * 1.22 + 0.35 * agent
    B = A * 0.62
    h = 1.4 + 0.25 * agent
    th = 2 * np.pi * (np.arange(n_kf) / n_all) + 0.37 * agent
    c = np.stack([A * np.cos(th) + 0.8 * agent, B * np.sin(th) - 0.5 * agent, np.full(n_kf, h)], 1)
    f = np.stack([-A * np.sin(th), B * np.cos(th), np.zeros(n_kf)], 1)
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    down = np.tile(np.array([0.0, 0.0, -1.0]), (n_kf, 1))
    right = np.cross(down, f)
    right /= np.linalg.norm(right, axis=1, keepdims=True)
    R = np.stack([right, down, f], 1)  # rows = camera axes in world
    t = -np.einsum("nij,nj->ni", R, c)
    return R, t, c


def make_ba_problem(n_agents: int = 1, kfs_per_agent: int = 70, n_points: int = 4000, seed: int = 0,
                    mean_track: float = 6.0, max_track: int = 30, cross_frac: float = 0.1,
                    outlier_frac: float = 0.02, noise: bool = True, n_fixed: int = 1,
                    fixed_mode: str = "first", pose_sigma_t: float = 0.02, pose_sigma_r_deg: float = 0.5,
                    point_sigma: float = 0.03, huber_delta: float | None = None, loop_len: int | None = None):
    """Build a BA problem in the flat layout of ccm_ba_problem (include/ccm_hip.h).

    Returns a dict of contiguous numpy arrays (cam_qt, cam_fixed, cam_K, pt_xyz, e_cam, e_pt,
    e_obs, e_info, e_level) plus ground truth (gt_cam_qt, gt_pt_xyz) and huber_delta.

    Noise model: pixel noise N(0, (1.2^octave)^2), information 1.2^(-2*octave) taken
    from the reference's f32 mvInvLevelSigma2 table; `outlier_frac` of the observations get an extra
    +-U(10,40) px; the initial state is truth + pose/point noise rounded to f32, because the
    reference stores poses and points as CV_32F and widens them in Converter::toSE3Quat /
    toVector3d (cslam/src/Converter.cc:40-50,113-119).  fixed_mode "first": cameras 0..n_fixed-1 are
    fixed (GBA: the first origin KF, Optimizer.cpp:705); "tail": the LAST n_fixed cameras of every
    agent are fixed (local BA: observers outside the local window, Optimizer.cpp:439-454).
    """
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = EUROC_K
    Rs, ts, cs = [], [], []
    for a in range(n_agents):
        R, t, c = _agent_loop(kfs_per_agent, a, loop_len)
        Rs.append(R), ts.append(t), cs.append(c)
    R_all = np.concatenate(Rs)
    t_all = np.concatenate(ts)
    c_all = np.concatenate(cs)
    n_cam = n_agents * kfs_per_agent

    # ---- tracks ----
    n_gen = int(n_points * 1.25) + 16
    agent = rng.integers(0, n_agents, n_gen)
    start = rng.integers(0, kfs_per_agent, n_gen)
    klen = np.clip(rng.poisson(mean_track, n_gen), 2, min(max_track, kfs_per_agent))
    mid = (start + klen // 2) % kfs_per_agent
    mid_cam = agent * kfs_per_agent + mid
    depth = rng.uniform(2.5, 9.0, n_gen)
    lx = rng.uniform(-0.40, 0.40, n_gen) * depth
    ly = rng.uniform(-0.30, 0.30, n_gen) * depth
    Xc_mid = np.stack([lx, ly, depth], 1)
    # X_w = R^T (X_c - t)
    Xw = np.einsum("nji,nj->ni", R_all[mid_cam], Xc_mid - t_all[mid_cam])

    pt_ids = np.repeat(np.arange(n_gen), klen)
    offs = np.arange(klen.sum()) - np.repeat(np.cumsum(klen) - klen, klen)
    cam_ids = np.repeat(agent, klen) * kfs_per_agent + (np.repeat(start, klen) + offs) % kfs_per_agent
    # cross-agent tracks
    if n_agents > 1 and cross_frac > 0:
        cross = np.nonzero(rng.random(n_gen) < cross_frac)[0]
        other = (agent[cross] + 1 + rng.integers(0, n_agents - 1, cross.size)) % n_agents
        k2 = np.clip(rng.poisson(mean_track * 0.6, cross.size), 1, max_track)
        extra_p, extra_c = [], []
        for p, b, kk in zip(cross, other, k2):
            cb = cs[b]
            near = int(np.argmin(np.sum((cb - c_all[mid_cam[p]]) ** 2, 1)))
            idx = (near - kk // 2 + np.arange(kk)) % kfs_per_agent
            extra_p.append(np.full(kk, p))
            extra_c.append(b * kfs_per_agent + idx)
        if extra_p:
            pt_ids = np.concatenate([pt_ids] + extra_p)
            cam_ids = np.concatenate([cam_ids] + extra_c)
    # visibility
    Xc = np.einsum("nij,nj->ni", R_all[cam_ids], Xw[pt_ids]) + t_all[cam_ids]
    z = Xc[:, 2]
    zs = np.where(z > 1e-6, z, 1.0)
    u = fx * Xc[:, 0] / zs + cx
    v = fy * Xc[:, 1] / zs + cy
    vis = (z > 0.5) & (z < 15.0) & (u > 20) & (u < IMG_W - 20) & (v > 20) & (v < IMG_H - 20)
    pt_ids, cam_ids, u, v, z = pt_ids[vis], cam_ids[vis], u[vis], v[vis], z[vis]
    # unique (pt, cam) pairs, ordered by point then camera id (the reference iterates
    # map<kfptr,size_t> observations per map point, Optimizer.cpp:742-786)
    key = pt_ids.astype(np.int64) * n_cam + cam_ids
    _, first = np.unique(key, return_index=True)
    pt_ids, cam_ids, u, v, z = pt_ids[first], cam_ids[first], u[first], v[first], z[first]
    cnt = np.bincount(pt_ids, minlength=n_gen)
    good = np.nonzero(cnt >= 2)[0][:n_points]
    remap = -np.ones(n_gen, np.int64)
    remap[good] = np.arange(good.size)
    keep = remap[pt_ids] >= 0
    pt_ids, cam_ids, u, v, z = remap[pt_ids[keep]], cam_ids[keep], u[keep], v[keep], z[keep]
    Xw = Xw[good]
    n_pt = good.size
    n_edge = pt_ids.size

    # ---- observations ----
    _, _, _, inv_s2 = scale_tables()
    octave = np.clip(np.ceil(np.log(10.0 / z) / np.log(SCALE)), 0, N_LEVELS - 1).astype(np.int64)
    sigma = SCALE ** octave
    obs = np.stack([u, v], 1)
    if noise:
        obs = obs + rng.normal(size=obs.shape) * sigma[:, None]
        out = rng.random(n_edge) < outlier_frac
        mag = rng.uniform(10, 40, (n_edge, 2)) * rng.choice([-1.0, 1.0], (n_edge, 2))
        obs = obs + out[:, None] * mag
    obs = obs.astype(np.float32).astype(np.float64)  # cv::KeyPoint.pt is f32
    info = inv_s2[octave].astype(np.float64)

    # ---- states ----
    gt_q = quat_from_R(R_all)
    gt_cam = np.concatenate([gt_q, t_all], 1)
    if noise:
        dR = rodrigues(rng.normal(size=(n_cam, 3)) * np.deg2rad(pose_sigma_r_deg))
        R0 = dR @ R_all
        t0 = t_all + rng.normal(size=(n_cam, 3)) * pose_sigma_t
        X0 = Xw + rng.normal(size=Xw.shape) * point_sigma
    else:
        R0, t0, X0 = R_all.copy(), t_all.copy(), Xw.copy()
    fixed = np.zeros(n_cam, np.uint8)
    if fixed_mode == "first":
        fixed[:n_fixed] = 1
    elif fixed_mode == "tail":
        for a in range(n_agents):
            fixed[(a + 1) * kfs_per_agent - n_fixed:(a + 1) * kfs_per_agent] = 1
    else:
        raise ValueError(fixed_mode)
    # fixed cameras keep their true pose (they are well-converged map KFs)
    R0[fixed == 1] = R_all[fixed == 1]
    t0[fixed == 1] = t_all[fixed == 1]
    R0 = R0.astype(np.float32).astype(np.float64)
    t0 = t0.astype(np.float32).astype(np.float64)
    X0 = X0.astype(np.float32).astype(np.float64)
    cam_qt = np.concatenate([quat_from_R(R0), t0], 1)

    if huber_delta is None:
        huber_delta = float(np.float32(np.sqrt(np.float32(5.99))))  # const float thHuber2D = sqrt(5.99), Optimizer.cpp:759
    return dict(
        n_cam=n_cam, n_pt=n_pt, n_edge=n_edge,
        cam_qt=np.ascontiguousarray(cam_qt), cam_fixed=fixed,
        cam_K=np.tile(np.array(EUROC_K, np.float64), (n_cam, 1)),
        pt_xyz=np.ascontiguousarray(X0),
        e_cam=cam_ids.astype(np.int32), e_pt=pt_ids.astype(np.int32),
        e_obs=np.ascontiguousarray(obs), e_info=np.ascontiguousarray(info),
        e_level=np.zeros(n_edge, np.uint8), huber_delta=float(huber_delta),
        gt_cam_qt=gt_cam, gt_pt_xyz=Xw, octave=octave.astype(np.int32),
    )


# BASELINE.json configs -> BA problem sizes
BA_CONFIGS = {
    "lba_c2": dict(n_agents=1, kfs_per_agent=70, n_points=4000, n_fixed=40, fixed_mode="tail", seed=2001),
    # the reference's configured local window (cslam/conf/config.yaml:78-79: Mapping.LocalMapSize 50 + Mapping.LocalMapBuffer 20): 50 free + 20 fixed keyframes
    "lba_50": dict(n_agents=1, kfs_per_agent=70, n_points=5000, n_fixed=20, fixed_mode="tail", seed=2050),
    "gba_c3": dict(n_agents=3, kfs_per_agent=400, n_points=90000, seed=3100),
    "gba_c4": dict(n_agents=4, kfs_per_agent=500, n_points=150000, seed=4100),
    "gba_c5": dict(n_agents=8, kfs_per_agent=1250, n_points=300000, seed=5100),
}


def make_ba_config(name: str, **over):
    kw = dict(BA_CONFIGS[name])
    kw.update(over)
    return make_ba_problem(**kw)


def pose_errors(cam_a: np.ndarray, cam_b: np.ndarray):
    """per-pose translational (m, camera-centre distance) and rotational (deg) difference"""
    Ra, Rb = R_from_quat(cam_a[:, :4]), R_from_quat(cam_b[:, :4])
    ca = -np.einsum("nji,nj->ni", Ra, cam_a[:, 4:])
    cb = -np.einsum("nji,nj->ni", Rb, cam_b[:, 4:])
    dt = np.linalg.norm(ca - cb, axis=1)
    Rrel = np.einsum("nij,nkj->nik", Ra, Rb)
    cosang = np.clip((np.trace(Rrel, axis1=1, axis2=2) - 1) / 2, -1, 1)
    return dt, np.rad2deg(np.arccos(cosang))


# --------------------------------------------------------------------------------------------
# descriptors / images
# --------------------------------------------------------------------------------------------
def make_descriptor_sets(n1: int, n2: int, seed: int, match_frac: float = 0.7, max_flip: int = 60):
    """D1 (n1x32) random; D2 (n2x32): match_frac of the rows are rows of D1 with k~U{0..max_flip}
    flipped bits, the rest random."""
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    nm = min(int(match_frac * n2), n1)
    src = rng.permutation(n1)[:nm]
    dst = rng.permutation(n2)[:nm]
    rows = d1[src].copy()
    bits = np.unpackbits(rows, axis=1)
    for r in range(nm):
        k = int(rng.integers(0, max_flip + 1))
        flip = rng.permutation(256)[:k]
        bits[r, flip] ^= 1
    d2[dst] = np.packbits(bits, axis=1)
    return d1, d2, src, dst


def gen_image(seed: int, t: int = 0, w: int = IMG_W, h: int = IMG_H) -> np.ndarray:
    """Deterministic textured test frame: low-frequency gradient + 64 random rectangles (some
    rotated) + a field of small blobs for FAST corners + integer noise U{-4..4}; the scene is
    translated by (t mod 40, t mod 23) pixels so consecutive frames overlap."""
    rng = np.random.default_rng(seed)
    H2, W2 = h + 64, w + 64
    yy, xx = np.mgrid[0:H2, 0:W2]
    img = 90 + 50 * np.sin(xx / 97.0) * np.cos(yy / 71.0) + 20 * np.sin((xx + yy) / 41.0)
    for _ in range(64):
        x0, y0 = rng.integers(0, W2 - 20), rng.integers(0, H2 - 20)
        ww, hh = rng.integers(12, 140), rng.integers(12, 110)
        val = float(rng.integers(20, 236))
        if rng.random() < 0.5:
            img[y0:y0 + hh, x0:x0 + ww] = val
        else:
            ang = rng.uniform(0, np.pi)
            ca, sa = np.cos(ang), np.sin(ang)
            dx, dy = xx - x0, yy - y0
            m = (np.abs(dx * ca + dy * sa) < ww / 2) & (np.abs(-dx * sa + dy * ca) < hh / 2)
            img[m] = val
    nb = 900
    bx, by = rng.integers(4, W2 - 4, nb), rng.integers(4, H2 - 4, nb)
    bv = rng.integers(0, 256, nb)
    bs = rng.integers(1, 4, nb)
    for x, y, v, s in zip(bx, by, bv, bs):
        img[y - s:y + s + 1, x - s:x + s + 1] = v
    ox, oy = t % 40, t % 23
    img = img[oy:oy + h, ox:ox + w]
    nrng = np.random.default_rng(seed * 7919 + t)
    img = img + nrng.integers(-4, 5, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# Sweep cases on which the REFERENCE ITSELF throws: a level whose border box has W > 0 and H <= 0 (or the reverse) makes DistributeOctTree compute
# nIni = round(W / H) <= 0 and `vpIniNodes.resize(nIni)` raises std::length_error (ORBextractor.cpp:711-716; observed on oracle/_ref/orb_ref_cli).  The
# oracle and the product define the natural continuation — such a level holds no cell, hence no keypoint — and are compared with each other there.
ORB_SWEEP_REFERENCE_THROWS = {"small_64x48", "small_100x70", "small_96x96", "small_91x62", "small_62x91", "small_123x77", "small_200x63", "small_80x80"}


def orb_sweep_cases():
    """(name, image, nfeatures, kwargs) for the wide ORB parity sweep (tests/test_orb_gpu.py: HIP vs oracle; tests/test_ref_orb.py: oracle vs the
    reference's own ORBextractor.cpp).  Everything ORBextractor.cpp:933-998 / :1280-1304 branches on: widths and heights of every residue mod 4 (row
    strides, resize tables), levels smaller than one 30-px cell (they contribute no keypoints: `nCols` or `nRows` is 0 and the cell loops do not
    run), images where EVERY level is, single-cell levels whose cell is up to 59 px wide, dense checkerboards (the most NMS survivors a cell can
    hold, tens of thousands of octree candidates per level), saturated / noise / step images, nlevels 1 and 12, other scale factors, tiny and huge
    feature budgets, iniThFAST == minThFAST.  kwargs use the oracle's names (scale, nlevels, ini_th, min_th)."""
    cases = []
    # 1. sizes around the EuRoC frame with every residue of width and height mod 4 (28 images)
    k = 0
    for dw in range(-3, 4):
        for dh in (-2, -1, 1, 3):
            cases.append((f"size_{IMG_W + dw}x{IMG_H + dh}", gen_image(2000 + k, k, IMG_W + dw, IMG_H + dh), 1000, {}))
            k += 1
    # 2. small images: vanishing levels (64x48: no level holds a cell; 40x40: none either and levels shrink to 11 px; 100x70 / 96x96: some do)
    for w, h, s in ((64, 48, 1), (40, 40, 2), (100, 70, 3), (96, 96, 4), (91, 62, 5), (62, 91, 6), (123, 77, 7), (160, 120, 8), (200, 63, 9), (64, 64, 10),
                    (80, 80, 11), (75, 75, 12)):
        cases.append((f"small_{w}x{h}", gen_image(3000 + s, s, w, h), 300, {}))
    rng = np.random.default_rng(77)
    cases.append(("small_noise_91x91", rng.integers(0, 256, (91, 91), dtype=np.uint8), 200, {}))
    # 3. checkerboards: squares of 2 / 3 / 5 px, a corner at every crossing
    yy, xx = np.mgrid[0:IMG_H, 0:IMG_W]
    for sq in (2, 3, 5):
        cases.append((f"checker_{sq}px", (((xx // sq + yy // sq) & 1) * 215 + 20).astype(np.uint8), 1000, {}))
    cases.append(("checker_3px_2000", (((xx // 3 + yy // 3) & 1) * 255).astype(np.uint8), 2000, {}))
    # 4. pixel statistics
    cases.append(("noise_u8", rng.integers(0, 256, (IMG_H, IMG_W), dtype=np.uint8), 1000, {}))
    cases.append(("noise_binary", (rng.integers(0, 2, (IMG_H, IMG_W)) * 255).astype(np.uint8), 1000, {}))
    cases.append(("all_zero", np.zeros((IMG_H, IMG_W), np.uint8), 1000, {}))
    cases.append(("all_255", np.full((IMG_H, IMG_W), 255, np.uint8), 1000, {}))
    cases.append(("vertical_step", np.where(xx < IMG_W // 2, 40, 200).astype(np.uint8), 1000, {}))
    sparse = np.full((IMG_H, IMG_W), 128, np.uint8)
    for _ in range(40):
        x, y = int(rng.integers(25, IMG_W - 25)), int(rng.integers(25, IMG_H - 25))
        sparse[y - 1:y + 2, x - 1:x + 2] = int(rng.integers(0, 2)) * 255
    cases.append(("forty_blobs", sparse, 1000, {}))
    # 5. extractor parameters
    base = gen_image(4000, 3)
    cases.append(("nlevels_1", base, 1000, {"nlevels": 1}))
    cases.append(("nlevels_12", base, 1000, {"nlevels": 12}))
    cases.append(("nlevels_12_2000", gen_image(4001, 5), 2000, {"nlevels": 12}))
    cases.append(("scale_1.1", base, 1000, {"scale": 1.1}))
    cases.append(("scale_1.5_6", base, 800, {"scale": 1.5, "nlevels": 6}))
    cases.append(("scale_2.0_4", base, 500, {"scale": 2.0, "nlevels": 4}))
    cases.append(("nfeatures_7", base, 7, {}))
    cases.append(("nfeatures_50", base, 50, {}))
    cases.append(("nfeatures_5000", gen_image(4002, 9), 5000, {}))
    cases.append(("th_40_20", base, 1000, {"ini_th": 40, "min_th": 20}))
    cases.append(("th_equal_12", base, 1000, {"ini_th": 12, "min_th": 12}))
    cases.append(("th_1_1", gen_image(4003, 1, 320, 240), 600, {"ini_th": 1, "min_th": 1}))
    cases.append(("vga", gen_image(4004, 2, 640, 480), 1200, {}))
    cases.append(("hd_1280x720", gen_image(4005, 4, 1280, 720), 1500, {}))
    cases.append(("wide_1000x130", gen_image(4006, 6, 1000, 130), 500, {"nlevels": 4}))
    return cases


def make_pose_problem(n: int = 300, seed: int = 0, outlier_frac: float = 0.1, pose_sigma_t: float = 0.05,
                      pose_sigma_r_deg: float = 1.0):
    """One tracked frame for Optimizer::PoseOptimizationClient (Optimizer.cpp:215-347): n map points seen by
    a camera, f32-rounded world positions (MapPoint::GetWorldPos is CV_32F), f32 keypoints with octave-
    dependent noise, a fraction of gross outliers, and a perturbed initial pose (the motion-model prediction)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = EUROC_K
    R, t, _ = _agent_loop(40, 0)
    R, t = R[3], t[3]
    depth = rng.uniform(1.5, 9.0, n)
    u = rng.uniform(30, IMG_W - 30, n)
    v = rng.uniform(30, IMG_H - 30, n)
    Xc = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], 1)
    Xw = (Xc - t) @ R          # R^T (Xc - t)
    Xw = Xw.astype(np.float32).astype(np.float64)
    _, _, _, inv_s2 = scale_tables()
    octave = np.clip(np.ceil(np.log(10.0 / depth) / np.log(SCALE)), 0, N_LEVELS - 1).astype(np.int64)
    Xc2 = Xw @ R.T + t
    obs = np.stack([fx * Xc2[:, 0] / Xc2[:, 2] + cx, fy * Xc2[:, 1] / Xc2[:, 2] + cy], 1)
    obs += rng.normal(size=obs.shape) * (SCALE ** octave)[:, None]
    out = rng.random(n) < outlier_frac
    obs += out[:, None] * rng.uniform(8, 60, (n, 2)) * rng.choice([-1.0, 1.0], (n, 2))
    obs = obs.astype(np.float32).astype(np.float64)
    dR = rodrigues(rng.normal(size=(1, 3)) * np.deg2rad(pose_sigma_r_deg))[0]
    R0 = (dR @ R).astype(np.float32).astype(np.float64)
    t0 = (t + rng.normal(size=3) * pose_sigma_t).astype(np.float32).astype(np.float64)
    cam0 = np.concatenate([quat_from_R(R0[None])[0], t0])
    gt = np.concatenate([quat_from_R(R[None])[0], t])
    return dict(cam_qt=cam0, Xw=Xw, obs=obs, info=inv_s2[octave].astype(np.float64), K=np.array(EUROC_K), gt_cam_qt=gt,
                is_outlier=out)


def make_sim3_problem(n: int = 150, seed: int = 0, outlier_frac: float = 0.1, fix_scale: bool = False,
                      sigma_r_deg: float = 1.0, sigma_t: float = 0.03, sigma_s: float = 0.02):
    """One loop / map-match candidate for Optimizer::OptimizeSim3 (Optimizer.cpp:861-1056): n map-point pairs, each point
    given in its own keyframe's camera frame (P1c = R1w*X1+t1w, P2c likewise, f32-rounded like the cv::Mat products at
    :925-936), undistorted f32 keypoints with octave-dependent noise in both keyframes, a fraction of wrong pairs, and a
    perturbed initial S12 (the Sim3Solver RANSAC estimate).  sim3 layout = [qx qy qz qw tx ty tz s]."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = EUROC_K
    s_true = 1.0 if fix_scale else float(rng.uniform(0.8, 1.25))
    R12 = rodrigues(rng.normal(size=(1, 3)) * np.deg2rad(6.0))[0]
    t12 = rng.uniform(-0.4, 0.4, 3)
    P1, P2 = [], []
    while len(P2) < n:
        depth = rng.uniform(2.0, 9.0)
        u, v = rng.uniform(40, IMG_W - 40), rng.uniform(40, IMG_H - 40)
        X2 = np.array([(u - cx) / fx * depth, (v - cy) / fy * depth, depth])
        X1 = s_true * (R12 @ X2) + t12
        if X1[2] < 0.8:
            continue
        u1, v1 = fx * X1[0] / X1[2] + cx, fy * X1[1] / X1[2] + cy
        if not (20 < u1 < IMG_W - 20 and 20 < v1 < IMG_H - 20):
            continue
        P1.append(X1); P2.append(X2)
    P1, P2 = np.array(P1), np.array(P2)
    _, _, _, inv_s2 = scale_tables()
    oct1 = np.clip(np.ceil(np.log(10.0 / P1[:, 2]) / np.log(SCALE)), 0, N_LEVELS - 1).astype(np.int64)
    oct2 = np.clip(np.ceil(np.log(10.0 / P2[:, 2]) / np.log(SCALE)), 0, N_LEVELS - 1).astype(np.int64)

    def proj(P):
        return np.stack([fx * P[:, 0] / P[:, 2] + cx, fy * P[:, 1] / P[:, 2] + cy], 1)
    obs1 = proj(P1) + rng.normal(size=(n, 2)) * (SCALE ** oct1)[:, None]
    obs2 = proj(P2) + rng.normal(size=(n, 2)) * (SCALE ** oct2)[:, None]
    out = rng.random(n) < outlier_frac
    obs1 += out[:, None] * rng.uniform(8, 40, (n, 2)) * rng.choice([-1.0, 1.0], (n, 2))
    # the two maps disagree slightly about the 3-D points
    P1 = (P1 + rng.normal(size=P1.shape) * 0.01).astype(np.float32).astype(np.float64)
    P2 = (P2 + rng.normal(size=P2.shape) * 0.01).astype(np.float32).astype(np.float64)
    dR = rodrigues(rng.normal(size=(1, 3)) * np.deg2rad(sigma_r_deg))[0]
    s0 = 1.0 if fix_scale else s_true * float(np.exp(rng.normal() * sigma_s))
    sim0 = np.concatenate([quat_from_R((dR @ R12)[None])[0], t12 + rng.normal(size=3) * sigma_t, [s0]])
    gt = np.concatenate([quat_from_R(R12[None])[0], t12, [s_true]])
    return dict(sim3=sim0, P1c=P1, P2c=P2, obs1=obs1.astype(np.float32).astype(np.float64),
                obs2=obs2.astype(np.float32).astype(np.float64), info1=inv_s2[oct1].astype(np.float64),
                info2=inv_s2[oct2].astype(np.float64), K1=np.array(EUROC_K), K2=np.array(EUROC_K), th2=10.0,
                fix_scale=bool(fix_scale), gt_sim3=gt, is_outlier=out)


def make_pose_graph(n_kf: int = 120, seed: int = 0, fix_scale: bool = False, drift_t: float = 0.01, drift_r_deg: float = 0.15,
                    drift_s: float = 0.002, n_loop: int = 3, covis: int = 4):
    """Essential-graph problem for Optimizer::OptimizeEssentialGraph{LoopClosure,MapFusion} (Optimizer.cpp:1058-1566): one
    agent drives a closed loop; the keyframe poses Siw drift (translation, rotation and, unless fix_scale, scale); spanning-
    tree and covisibility edges carry the relative Sim3 of the DRIFTED trajectory (zero error at the start, :1166-1260), the
    loop edges carry the true relative Sim3 between the ends of the loop (:1122-1160).  Vertex 0 (the loop keyframe) is
    fixed.  Layout: sim3[n,8] = qx qy qz qw tx ty tz s (world -> camera), edges (i, j, Sji) with Sji = Sjw * Swi."""
    rng = np.random.default_rng(seed)
    R_true, t_true, _ = _agent_loop(n_kf, 0)

    def mul(a, b):      # Sim3 product (R, t, s)
        return a[0] @ b[0], a[2] * (a[0] @ b[1]) + a[1], a[2] * b[2]

    def inv(a):
        Rt = a[0].T
        return Rt, Rt @ (-a[1] / a[2]), 1.0 / a[2]
    truth = [(R_true[k], t_true[k], 1.0) for k in range(n_kf)]
    est = [truth[0]]
    for k in range(1, n_kf):
        rel = mul(truth[k], inv(truth[k - 1]))                       # S_k,k-1
        dR = rodrigues(rng.normal(size=(1, 3)) * np.deg2rad(drift_r_deg))[0]
        ds = 1.0 if fix_scale else float(np.exp(rng.normal() * drift_s))
        noisy = (dR @ rel[0], rel[1] + rng.normal(size=3) * drift_t, rel[2] * ds)
        est.append(mul(noisy, est[k - 1]))
    e_i, e_j, meas = [], [], []

    def add(i, j, Sji):
        e_i.append(i); e_j.append(j)
        meas.append(np.concatenate([quat_from_R(Sji[0][None])[0], Sji[1], [Sji[2]]]))
    for k in range(1, n_kf):                                          # spanning tree: child i = k, parent j = k-1
        add(k, k - 1, mul(est[k - 1], inv(est[k])))
        for d in range(2, covis + 1):                                 # covisibility edges to earlier keyframes
            if k - d >= 0 and rng.random() < 0.8:
                add(k, k - d, mul(est[k - d], inv(est[k])))
    for q in range(n_loop):                                           # loop connections: last keyframes <-> first ones, true geometry
        i, j = n_kf - 1 - q, q
        add(i, j, mul(truth[j], inv(truth[i])))
    sim3 = np.stack([np.concatenate([quat_from_R(e[0][None])[0], e[1], [e[2]]]) for e in est])
    gt = np.stack([np.concatenate([quat_from_R(e[0][None])[0], e[1], [e[2]]]) for e in truth])
    fixed = np.zeros(n_kf, np.uint8); fixed[0] = 1
    return dict(sim3=sim3, fixed=fixed, fix_scale=bool(fix_scale), e_i=np.array(e_i, np.int32), e_j=np.array(e_j, np.int32),
                meas=np.stack(meas), gt_sim3=gt, n_vert=n_kf, n_edge=len(e_i))


def make_vocabulary(k: int = 10, L: int = 4, seed: int = 0, stop_frac: float = 0.02):
    """Synthetic DBoW2-style vocabulary tree (the real ORBvoc.txt is a missing blob): complete k-ary
    tree of depth L in breadth-first node order, node descriptors = parent's with ~12 % of the bits flipped, leaf words
    numbered in node order, idf-like positive weights with a few stopped (zero-weight) words."""
    rng = np.random.default_rng(seed)
    n_nodes = (k ** (L + 1) - 1) // (k - 1)
    n_inner = (k ** L - 1) // (k - 1)
    child_off = np.zeros(n_nodes + 1, np.int32)
    child_off[1:n_inner + 1] = np.arange(1, n_inner + 1) * k
    child_off[n_inner + 1:] = n_inner * k
    child_id = np.arange(1, n_inner * k + 1, dtype=np.int32)
    desc = np.zeros((n_nodes, 32), np.uint8)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    parent = (np.arange(1, n_nodes) - 1) // k
    level_start = 1
    for lvl in range(1, L + 1):
        cnt = k ** lvl
        ids = np.arange(level_start, level_start + cnt)
        bits = np.unpackbits(desc[parent[ids - 1]], axis=1)
        desc[ids] = np.packbits(bits ^ (rng.random(bits.shape) < 0.12), axis=1)
        level_start += cnt
    word_id = -np.ones(n_nodes, np.int32)
    word_id[n_inner:] = np.arange(n_nodes - n_inner)
    weight = np.zeros(n_nodes)
    weight[n_inner:] = rng.uniform(0.5, 6.0, n_nodes - n_inner)
    weight[n_inner:][rng.random(n_nodes - n_inner) < stop_frac] = 0.0
    return dict(n_nodes=n_nodes, L=L, child_off=child_off, child_id=child_id, node_desc=desc, word_id=word_id, weight=weight)
