"""The structure that ccm_ba_create builds ON THE DEVICE (ccm_slam_amd/csrc/ba_build.hip: active set, vertex slots, edge order, per-camera lists,
pair instances, block-CSR rows, persistent-solver column lists, cluster entry lists, row-kernel unit table, landmark chunks, coarse block lists)
against an independent numpy restatement of what g2o's initializeOptimization(0) + buildStructure (sparse_optimizer.cpp:199-267,
block_solver.hpp:143-295) and our kernels' index conventions prescribe.  Every array must match exactly, for one rank and for the shards of a
2- / 3-rank landmark partition.  The arrays are read through libccm_testhooks.so (test-only entry points, not part of the product library)."""
import ctypes as C
import os

import numpy as np
import pytest

from ccm_slam_amd import optimizer, synth
from ccm_slam_amd._lib import lib

pytestmark = pytest.mark.gpu

KCLU, CHUNK, TPB = 16, 64, 256
TBIT = np.uint32(0x80000000)
_HOOKS = None


def hooks():
    global _HOOKS
    if _HOOKS is None:
        lib()   # the product library first: the hooks link against it
        _HOOKS = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(optimizer.__file__)), "libccm_testhooks.so"))
    return _HOOKS


def dev_array(h, name, dtype):
    n = C.c_size_t(0)
    assert hooks().ccm_ba_debug_array(h._h, name.encode(), None, C.c_size_t(0), C.byref(n)) == 0, name
    out = np.zeros(n.value // np.dtype(dtype).itemsize, dtype)
    if n.value:
        assert hooks().ccm_ba_debug_array(h._h, name.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes), C.byref(n)) == 0, name
    return out


def partition(weight, nranks):
    begin = np.zeros(nranks + 1, np.int32)
    w = np.ascontiguousarray(weight, np.int64)
    assert lib().ccm_ba_partition(w.ctypes.data_as(C.c_void_p), int(w.size), int(nranks), begin.ctypes.data_as(C.c_void_p)) == 0
    return begin


def host_structure(prob, rank=0, nranks=1, agg=32):
    e_cam, e_pt = prob["e_cam"].astype(np.int64), prob["e_pt"].astype(np.int64)
    lvl = prob.get("e_level")
    lvl = np.zeros(e_cam.size, np.uint8) if lvl is None else np.asarray(lvl)
    fixed = np.asarray(prob["cam_fixed"]) != 0
    n_cam, n_pt = int(prob["n_cam"]), int(prob["n_pt"])
    act = np.flatnonzero(lvl == 0)
    cam_has, pt_has = np.zeros(n_cam, bool), np.zeros(n_pt, bool)
    cam_has[e_cam[act]] = True; pt_has[e_pt[act]] = True
    slot_cam = np.flatnonzero(cam_has & ~fixed); slot_pt = np.flatnonzero(pt_has)
    cam_slot = np.full(n_cam, -1, np.int64); cam_slot[slot_cam] = np.arange(slot_cam.size)
    pt_slot = np.full(n_pt, -1, np.int64); pt_slot[slot_pt] = np.arange(slot_pt.size)
    Cp, Lp = slot_cam.size, slot_pt.size
    order = act[np.lexsort((cam_slot[e_cam[act]], pt_slot[e_pt[act]]))]          # stable; fixed cameras (slot -1) first inside a landmark
    g_pt_off = np.concatenate([[0], np.cumsum(np.bincount(pt_slot[e_pt[order]], minlength=Lp))]).astype(np.int64)
    cslot_g = cam_slot[e_cam[order]]
    kf = np.add.reduceat((cslot_g >= 0).astype(np.int64), g_pt_off[:-1]) if Lp else np.zeros(0, np.int64)
    weight = kf * (kf + 1) // 2 + np.diff(g_pt_off)
    shard = partition(weight, nranks) if nranks > 1 else np.array([0, Lp])
    lb, le = int(shard[rank]), int(shard[rank + 1])
    eb, ee = int(g_pt_off[lb]), int(g_pt_off[le])
    S = dict(Cp=Cp, Lp=Lp, Lloc=le - lb, Eloc=ee - eb, lb=lb, slot_cam=slot_cam, slot_pt=slot_pt)
    loc = order[eb:ee]
    S["loc_edge_orig"] = loc
    S["pt_off"] = g_pt_off[lb:le + 1] - eb
    S["ed_cam"] = e_cam[loc]; S["ed_cslot"] = cam_slot[e_cam[loc]]; S["ed_pt"] = pt_slot[e_pt[loc]] - lb
    S["obs"] = prob["e_obs"][loc].reshape(-1); S["info"] = prob["e_info"][loc]
    free = np.flatnonzero(S["ed_cslot"] >= 0)
    cam_edge = free[np.argsort(S["ed_cslot"][free], kind="stable")]
    S["cam_edge"] = cam_edge; S["cam_pt"] = S["ed_pt"][cam_edge]
    S["cam_oi"] = np.column_stack([S["obs"].reshape(-1, 2)[cam_edge], S["info"][cam_edge], np.zeros(cam_edge.size)]).reshape(-1)   # camera-major (obs x, obs y, info, 0)
    S["cam_off"] = np.concatenate([[0], np.cumsum(np.bincount(S["ed_cslot"][free], minlength=Cp))])
    S["max_cam_edges"] = int(np.diff(S["cam_off"]).max()) if Cp else 0

    def pairs(l0, l1):
        ka, kc, kk = [], [], []
        for l in range(l0, l1):
            k0, k1 = int(g_pt_off[l]), int(g_pt_off[l + 1])
            cs = cslot_g[k0:k1]
            for a in range(k1 - k0):
                if cs[a] < 0:
                    continue
                for c in range(a + 1, k1 - k0):
                    if cs[c] == cs[a]:
                        continue
                    kk.append(cs[a] * Cp + cs[c]); ka.append(k0 + a); kc.append(k0 + c)
        return np.array(kk, np.int64), np.array(ka, np.int64), np.array(kc, np.int64)
    kk, ka, kc = pairs(lb, le)
    o = np.argsort(kk, kind="stable")
    kk, ka, kc = kk[o], ka[o], kc[o]
    U = np.unique(pairs(0, Lp)[0]) if nranks > 1 else np.unique(kk)
    nOff = U.size
    S["nOff"] = nOff
    S["inst_off"] = np.concatenate([np.searchsorted(kk, U, "left"), [kk.size]])
    S["inst_a"], S["inst_c"] = ka - eb, kc - eb
    rank_in_cam = np.zeros(max(ee - eb, 1), np.int64)
    rank_in_cam[cam_edge] = np.arange(cam_edge.size) - S["cam_off"][S["ed_cslot"][cam_edge]]
    S["inst_al"] = rank_in_cam[S["inst_a"]] if kk.size else np.zeros(0, np.int64)
    bi, bj = U // max(Cp, 1), U % max(Cp, 1)
    S["blk_i"] = np.concatenate([np.arange(Cp), bi]); S["blk_j"] = np.concatenate([np.arange(Cp), bj])
    S["rowblk_off"] = np.searchsorted(bi, np.arange(Cp + 1), "left")
    rows_col, rows_blk = [[] for _ in range(Cp)], [[] for _ in range(Cp)]
    for b in range(nOff):                                      # lower parts (ascending i = ascending b), then diagonal, then upper parts
        rows_col[bj[b]].append(bi[b]); rows_blk[bj[b]].append(np.uint32(Cp + b) | TBIT)
    for i in range(Cp):
        rows_col[i].append(i); rows_blk[i].append(np.uint32(i))
    for b in range(nOff):
        rows_col[bi[b]].append(bj[b]); rows_blk[bi[b]].append(np.uint32(Cp + b))
    S["row_off"] = np.concatenate([[0], np.cumsum([len(r) for r in rows_col])]).astype(np.int64)
    S["row_col"] = np.array([c for r in rows_col for c in r], np.int64)
    S["row_blk"] = np.array([c for r in rows_blk for c in r], np.uint32)
    n_cl = -(-Cp // KCLU)
    if Cp > KCLU:                                                # persistent-solver lists + cluster entry lists
        uoff, ucol, ploc = [0], [], np.zeros(S["row_col"].size, np.int64)
        for u in range(2 * n_cl):
            r0 = min(Cp, (u >> 1) * KCLU + (u & 1) * (KCLU // 2)); r1 = min(Cp, r0 + KCLU // 2, ((u >> 1) + 1) * KCLU)
            ent = S["row_col"][S["row_off"][r0]:S["row_off"][r1]]
            cols = np.unique(ent)
            ploc[S["row_off"][r0]:S["row_off"][r1]] = np.searchsorted(cols, ent)
            ucol.extend(cols.tolist()); uoff.append(len(ucol))
        S["pers_uoff"], S["pers_ucol"], S["pers_loc"] = np.array(uoff), np.array(ucol, np.int64), ploc
        coff, cij, cblk = [0], [], []
        for c in range(n_cl):
            r0, r1 = c * KCLU, min(Cp, c * KCLU + KCLU)
            for i in range(r0, r1):
                for s in range(S["row_off"][i], S["row_off"][i + 1]):
                    col = S["row_col"][s]
                    if r0 <= col < r1:
                        cij.append(((i - r0) << 4) | (col - r0)); cblk.append(S["row_blk"][s])
            coff.append(len(cij))
        S["pers_coff"], S["pers_cij"], S["pers_cblk"] = np.array(coff), np.array(cij, np.int64), np.array(cblk, np.uint32)
    # row-kernel unit table: per row the units of its blocks (<= CHUNK pair instances each, at least one per block; slot = creation index), listed longest
    # first (stable).  (The diagonal block needs no units: the thread that stages an observation's Y adds its part.)
    tab, row_u, blk_u = [], [0], []
    for i in range(Cp):
        units = []
        for b in range(S["rowblk_off"][i], S["rowblk_off"][i + 1]):
            blk_u.append(len(tab) + len(units))
            s0, end = int(S["inst_off"][b]), int(S["inst_off"][b + 1])
            while True:
                s1 = min(end, s0 + CHUNK); units.append([b, s0, s1, len(units)]); s0 = s1
                if s0 >= end:
                    break
        units.sort(key=lambda u: -(u[2] - u[1]))               # list.sort is stable
        tab.extend(units); row_u.append(len(tab))
    blk_u.append(len(tab))
    S["unit_tab"], S["row_unit_off"], S["blk_unit0"] = np.array(tab, np.int64).reshape(-1, 4), np.array(row_u), np.array(blk_u)
    S["row_units_max"] = int(np.diff(S["row_unit_off"]).max()) if Cp else 0
    chunk, fits = [0], True                                     # greedy chunks: <= 256 landmarks and <= 256 observations
    po = S["pt_off"]
    for l in range(le - lb):
        if po[l + 1] - po[l] > TPB:
            fits = False; break
        if po[l + 1] - po[chunk[-1]] > TPB or l + 1 - chunk[-1] > TPB:
            chunk.append(l)
    S["chunk_off"] = np.array(chunk + [le - lb]) if fits and le > lb else np.zeros(0, np.int64)
    cb = {}
    for i in range(Cp):
        cb.setdefault((i // agg, i // agg), []).append(2 * i)
    for b in range(nOff):
        a, a2 = bi[b] // agg, bj[b] // agg
        cb.setdefault((a, a2), []).append(2 * (Cp + b) + (1 if a == a2 else 0))
    keys = sorted(cb)
    S["cb_ab"] = np.array(keys, np.int64).reshape(-1); S["cb_ent"] = np.array([e for k in keys for e in cb[k]], np.int64)
    S["cb_off"] = np.concatenate([[0], np.cumsum([len(cb[k]) for k in keys])])
    return S


def check(ctx, prob, rank=0, nranks=1):
    h = optimizer.BAHandle(ctx, prob, rank=rank, nranks=nranks)
    sz = dev_array(h, "sizes", np.int32)
    Cp, Lp, Lloc, Eloc, nOff, max_ce, units_max, pers_grid, c_na, c_ncb, n_chunk, lb = (int(v) for v in sz)
    # intervals of the coarse space: 16 cameras where the persistent solver runs and 12 rows of Ac^-1 (6 (Cp / 16 + 1) floats each, padded to 64) fit its LDS
    # (<= 768), 32 otherwise (ba_build.hip)
    nc = lambda a: -(-6 * (-(-Cp // a) + 1) // 64) * 64
    agg = 32 if not pers_grid else 16 if nc(16) <= 768 else 24 if nc(24) <= 768 else 32
    if c_na: assert c_na == -(-Cp // agg)
    S = host_structure(prob, rank, nranks, agg)
    assert (Cp, Lp, Lloc, Eloc, nOff, max_ce, lb) == (S["Cp"], S["Lp"], S["Lloc"], S["Eloc"], S["nOff"], S["max_cam_edges"], S["lb"])
    names = ["slot_cam", "slot_pt", "loc_edge_orig", "pt_off", "ed_cam", "ed_cslot", "ed_pt", "cam_off", "cam_edge", "cam_pt", "rowblk_off", "row_off", "row_col",
             "inst_off", "inst_a", "inst_c", "inst_al", "blk_i", "blk_j", "chunk_off"]
    for nm in names:
        assert np.array_equal(dev_array(h, nm, np.int32), np.asarray(S[nm], np.int64)), nm
    assert np.array_equal(dev_array(h, "row_blk", np.uint32), S["row_blk"])
    assert np.array_equal(dev_array(h, "obs", np.float64), S["obs"]) and np.array_equal(dev_array(h, "info", np.float64), S["info"])
    assert np.array_equal(dev_array(h, "cam_oi", np.float64), S["cam_oi"])
    # windows of 17 ... 50 free cameras take the register-resident Cholesky solve (ba_solve_cholreg), which reads the block-CSR rows only: the handle then carries none
    # of the persistent solver's lists (CCM_BA_CHOLREG=0 at create time keeps the earlier solvers and their structures)
    cholreg_win = KCLU < Cp <= 50 and os.environ.get("CCM_BA_CHOLREG", "1") != "0"
    if cholreg_win:
        assert pers_grid == 0 and c_na == 0
    if Cp > KCLU and not cholreg_win:
        for nm in ("pers_coff", "pers_cij"):
            assert np.array_equal(dev_array(h, nm, np.int32), S[nm]), nm
        assert np.array_equal(dev_array(h, "pers_cblk", np.uint32), S["pers_cblk"])
    if pers_grid:
        for nm in ("pers_uoff", "pers_ucol", "pers_loc"):
            assert np.array_equal(dev_array(h, nm, np.int32), S[nm]), nm
    if units_max:
        assert units_max == S["row_units_max"]
        assert np.array_equal(dev_array(h, "unit_tab", np.int32).reshape(-1, 4), S["unit_tab"])
        assert np.array_equal(dev_array(h, "row_unit_off", np.int32), S["row_unit_off"]) and np.array_equal(dev_array(h, "blk_unit0", np.int32), S["blk_unit0"])
    if c_na:
        assert c_ncb == S["cb_off"].size - 1
        for nm in ("cb_off", "cb_ab", "cb_ent"):
            assert np.array_equal(dev_array(h, nm, np.int32), S[nm]), nm
    h.close()
    return dict(Cp=Cp, nOff=nOff, pers_grid=pers_grid, units_max=units_max, coarse=c_na)


def test_structure_of_a_multi_agent_map_matches_the_restatement(ctx):
    prob = synth.make_ba_problem(n_agents=3, kfs_per_agent=60, n_points=6000, seed=21, n_fixed=1)
    info = check(ctx, prob)
    assert info["pers_grid"] and info["units_max"] and info["coarse"] and info["nOff"] > 256      # persistent solver, row kernel and coarse level all in use


@pytest.mark.parametrize("cholreg", ["1", "0"])
def test_structure_with_fixed_cameras_inactive_edges_and_unobserved_vertices(ctx, cholreg, monkeypatch):
    monkeypatch.setenv("CCM_BA_CHOLREG", cholreg)              # "0": the window's structures for the earlier solvers (cluster entry lists of the exact two-cluster solve)
    prob = synth.make_ba_config("lba_c2")
    rng = np.random.default_rng(5)
    lvl = np.zeros(prob["n_edge"], np.uint8)
    lvl[rng.random(prob["n_edge"]) < 0.07] = 1                  # the second stage of the local BA: outliers moved to level 1
    lvl[prob["e_pt"] == 17] = 1                                 # a landmark that loses all its edges
    lvl[prob["e_cam"] == 3] = 1                                 # a free camera that loses all its edges
    prob = dict(prob, e_level=lvl)
    info = check(ctx, prob)
    assert 16 < info["Cp"] <= 32                                # the exact two-cluster solve's range


def test_structure_of_small_problems(ctx):
    for n_kf, n_pt, seed in ((2, 60, 1), (9, 400, 2), (17, 900, 3)):
        check(ctx, synth.make_ba_problem(n_agents=1, kfs_per_agent=n_kf, n_points=n_pt, seed=seed))


@pytest.mark.parametrize("nranks", [2, 3])
def test_structure_of_every_shard_of_a_landmark_partition(ctx, nranks):
    prob = synth.make_ba_problem(n_agents=2, kfs_per_agent=40, n_points=3000, seed=8, n_fixed=1)
    for rank in range(nranks):
        check(ctx, prob, rank, nranks)
