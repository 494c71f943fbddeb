"""Hamming / window-search entry points of the C ABI (ORBmatcher path).

Mirrors the pieces of cslam::ORBmatcher that are data-parallel (cslam/src/ORBmatcher.cpp):
DescriptorDistance (:1653-1669) and the best / second-best scans of the Search* methods.  The
ordered resolution pass that reproduces the reference's sequential "claimed feature" semantics
(ORBmatcher.cpp:113-115, 1417-1419) lives in the C++ host shim (ccm_slam_amd/host/); the Python
functions here are the harness used by tests and bench.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def hamming_dense_best2(ctx: Context, q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    Q, T = q.shape[0], t.shape[0]
    bi, bd, sd = (np.empty(Q, np.int32) for _ in range(3))
    check(lib().ccm_hamming_dense_best2(ctx.handle, _p(q), Q, _p(t), T, _p(bi), _p(bd), _p(sd)), ctx.handle)
    return bi, bd, sd


def hamming_csr(ctx: Context, q, t, cand_off, cand_idx, want_best2: bool = True):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    cand_off = np.ascontiguousarray(cand_off, np.int32)
    cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    Q, T = q.shape[0], t.shape[0]
    dist = np.empty(max(cand_idx.size, 1), np.uint16)
    if want_best2:
        bi, bd, sd = (np.empty(Q, np.int32) for _ in range(3))
    else:
        bi = bd = sd = None
    check(lib().ccm_hamming_csr(ctx.handle, _p(q), Q, _p(t), T, _p(cand_off), _p(cand_idx), _p(dist), _p(bi), _p(bd),
                                _p(sd)), ctx.handle)
    return dist[:cand_idx.size], bi, bd, sd


def hamming_csr_multi(ctx: Context, qs, ts, offs, idxs, want_best2: bool = True):
    """S windowed searches in one launch (ccm_hamming_csr_multi): qs / ts = lists of descriptor sets, offs / idxs = the CSR of every search (indices local to its own
    target set).  Returns (dist, best_idx, best_dist, second_dist) over all queries back to back plus the per-search query / candidate offsets."""
    qs = [np.ascontiguousarray(q, np.uint8).reshape(-1, 32) for q in qs]
    ts = [np.ascontiguousarray(t, np.uint8).reshape(-1, 32) for t in ts]
    S = len(qs)
    q_off = np.zeros(S + 1, np.int32); t_off = np.zeros(S + 1, np.int32)
    q_off[1:] = np.cumsum([q.shape[0] for q in qs]); t_off[1:] = np.cumsum([t.shape[0] for t in ts])
    q = np.concatenate(qs) if S else np.zeros((0, 32), np.uint8)
    t = np.concatenate(ts) if S else np.zeros((0, 32), np.uint8)
    c_base = np.concatenate([[0], np.cumsum([int(np.asarray(o)[-1]) for o in offs])]).astype(np.int64)
    cand_off = np.concatenate([[0]] + [np.asarray(o, np.int64)[1:] + c_base[s] for s, o in enumerate(offs)]).astype(np.int32)
    cand_idx = np.ascontiguousarray(np.concatenate([np.asarray(i, np.int32) for i in idxs]) if S else np.zeros(0, np.int32), np.int32)
    Q = int(q_off[-1])
    dist = np.empty(max(cand_idx.size, 1), np.uint16)
    bi, bd, sd = ((np.empty(max(Q, 1), np.int32) for _ in range(3)) if want_best2 else (None, None, None))
    check(lib().ccm_hamming_csr_multi(ctx.handle, S, _p(q), _p(q_off), _p(t), _p(t_off), _p(cand_off), _p(cand_idx), _p(dist), _p(bi), _p(bd), _p(sd)), ctx.handle)
    if want_best2:
        bi, bd, sd = bi[:Q], bd[:Q], sd[:Q]
    return dist[:cand_idx.size], bi, bd, sd, q_off, c_base


class CsrMultiDev:
    """A batch of windowed searches resident in HBM (bench.py): run() = one ccm_hamming_csr_multi_dev launch."""

    def __init__(self, ctx: Context, q, t, q_tbase, cand_off, cand_idx):
        self.ctx = ctx
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32); t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        self.Q, self.n_cand = q.shape[0], int(cand_idx.size)
        self.d = [ctx.upload(a) for a in (q, t, np.ascontiguousarray(q_tbase, np.int32), np.ascontiguousarray(cand_off, np.int32), np.ascontiguousarray(cand_idx, np.int32))]
        self.d_dist = ctx.alloc(max(self.n_cand, 1) * 2)
        self.d_out = ctx.alloc(max(self.Q, 1) * 12)

    def run(self):
        Q = self.Q
        check(lib().ccm_hamming_csr_multi_dev(self.ctx.handle, C.c_void_p(self.d[0]), Q, C.c_void_p(self.d[1]), C.c_void_p(self.d[2]), C.c_void_p(self.d[3]), C.c_void_p(self.d[4]),
                                              C.c_int64(self.n_cand), C.c_void_p(self.d_dist), C.c_void_p(self.d_out), C.c_void_p(self.d_out + 4 * Q), C.c_void_p(self.d_out + 8 * Q)), self.ctx.handle)

    def close(self):
        for p in self.d + [self.d_dist, self.d_out]:
            self.ctx.free(p)


class DenseMatcherDev:
    """Device-resident dense best/second search for bench.py (inputs uploaded once)."""

    def __init__(self, ctx: Context, q: np.ndarray, t: np.ndarray):
        self.ctx = ctx
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        self.Q, self.T = q.shape[0], t.shape[0]
        self.d_q, self.d_t = ctx.upload(q), ctx.upload(t)
        self.d_out = ctx.alloc(max(self.Q, 1) * 12)

    def run(self):
        Q = self.Q
        check(lib().ccm_hamming_dense_best2_dev(self.ctx.handle, C.c_void_p(self.d_q), Q, C.c_void_p(self.d_t), self.T,
                                                C.c_void_p(self.d_out), C.c_void_p(self.d_out + 4 * Q),
                                                C.c_void_p(self.d_out + 8 * Q)), self.ctx.handle)

    def result(self):
        out = np.empty(3 * self.Q, np.int32)
        self.ctx.d2h(out, self.d_out)
        return out[:self.Q], out[self.Q:2 * self.Q], out[2 * self.Q:]

    def close(self):
        for p in (self.d_q, self.d_t, self.d_out):
            self.ctx.free(p)


def distinctive_descriptors(ctx: Context, desc, off):
    """MapPoint::ComputeDistinctiveDescriptors batched (MapPoint.cpp:929-994): per point the local index of the
    descriptor with the least median distance to the others."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.int32)
    out = np.zeros(off.size - 1, np.int32)
    check(lib().ccm_distinctive_descriptors(ctx.handle, _p(desc), _p(off), off.size - 1, _p(out)), ctx.handle)
    return out


class Vocabulary:
    """Device copy of a DBoW2 vocabulary tree (ccm_vocab_*); transform() = TemplatedVocabulary::transform on a batch."""

    def __init__(self, ctx: Context, vocab: dict):
        self.ctx = ctx
        self._keep = {k: np.ascontiguousarray(vocab[k]) for k in ("child_off", "child_id", "node_desc", "word_id", "weight")}
        self._h = C.c_void_p()
        k = self._keep
        check(lib().ccm_vocab_create(ctx.handle, int(vocab["n_nodes"]), int(vocab["L"]), _p(k["child_off"]), _p(k["child_id"]), _p(k["node_desc"]),
                                     _p(k["word_id"]), _p(k["weight"]), C.byref(self._h)), ctx.handle)
        ctx.adopt(self)

    def transform(self, desc, levelsup=4):
        """returns (word_id[N], weight[N], node_id[N], BowVector ids, BowVector values (L1-normalised), FeatureVector as
        (node ids ascending, CSR offsets, feature indices))"""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        N = desc.shape[0]
        word = np.zeros(N, np.int32); w = np.zeros(N, np.float64); node = np.zeros(N, np.int32)
        check(lib().ccm_bow_transform(self._h, _p(desc), N, int(levelsup), _p(word), _p(w), _p(node)), self.ctx.handle)
        keep = w > 0                                   # stopped words are skipped (TemplatedVocabulary.h:1158)
        # BowVector::addWeight in feature order, then L1 normalisation in ascending word order (BowVector.cpp:34-84)
        acc = {}
        for wid, wt in zip(word[keep].tolist(), w[keep].tolist()):
            acc[wid] = acc.get(wid, 0.0) + wt
        ids = np.array(sorted(acc), np.int32)
        vals = np.array([acc[i] for i in ids.tolist()], np.float64)
        norm = 0.0
        for x in vals.tolist():
            norm += abs(x)
        if norm > 0.0:
            vals = vals / norm
        fidx = np.nonzero(keep)[0]
        order = np.argsort(node[fidx], kind="stable")
        nodes_sorted = node[fidx][order]
        fv_nodes, starts = np.unique(nodes_sorted, return_index=True)
        fv_off = np.append(starts, nodes_sorted.size).astype(np.int32)
        return word, w, node, ids, vals, (fv_nodes.astype(np.int32), fv_off, fidx[order].astype(np.int32))

    def close(self):
        if self._h:
            lib().ccm_vocab_destroy(self._h)
            self._h = C.c_void_p()
