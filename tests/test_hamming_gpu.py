"""GPU parity: 256-bit Hamming kernels vs the oracle (bit-exact).  Reference semantics:
ORBmatcher::DescriptorDistance and the best/second scans (cslam/src/ORBmatcher.cpp:1653-1669, 102-134)."""
import numpy as np
import pytest

from ccm_slam_amd import matcher, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Q,T,seed", [(1, 1, 0), (7, 3, 1), (64, 64, 2), (257, 1000, 3), (2000, 2000, 4), (1000, 5001, 5)])
def test_dense_best2_matches_oracle(ctx, oracle_lib, Q, T, seed):
    d1, d2, _, _ = synth.make_descriptor_sets(T, Q, seed)
    bi, bd, sd = matcher.hamming_dense_best2(ctx, d2, d1)
    obi, obd, osd = oracle_lib.hamming_dense_best2(d2, d1)
    assert np.array_equal(bd, obd) and np.array_equal(sd, osd) and np.array_equal(bi, obi)


def test_dense_ties_first_minimum_wins(ctx, oracle_lib):
    # all targets identical -> every distance ties; the reference keeps the first index (strict '<')
    t = np.tile(np.arange(32, dtype=np.uint8), (300, 1))
    q = np.zeros((5, 32), np.uint8)
    bi, bd, sd = matcher.hamming_dense_best2(ctx, q, t)
    obi, obd, osd = oracle_lib.hamming_dense_best2(q, t)
    assert np.array_equal(bi, obi) and (bi == 0).all() and np.array_equal(bd, sd) and np.array_equal(sd, osd)


def test_dense_empty_targets(ctx):
    q = np.zeros((3, 32), np.uint8)
    bi, bd, sd = matcher.hamming_dense_best2(ctx, q, np.zeros((0, 32), np.uint8))
    assert (bi == -1).all() and (bd == 256).all() and (sd == 256).all()


def test_dense_extremes(ctx):
    q = np.zeros((2, 32), np.uint8)
    t = np.stack([np.full(32, 255, np.uint8), np.zeros(32, np.uint8)])
    bi, bd, sd = matcher.hamming_dense_best2(ctx, q, t)
    assert list(bi) == [1, 1] and list(bd) == [0, 0] and list(sd) == [256, 256]


@pytest.mark.parametrize("Q,T,mean_c,seed", [(50, 100, 5, 0), (5000, 1000, 30, 1), (300, 2000, 150, 2)])
def test_csr_matches_oracle(ctx, oracle_lib, Q, T, mean_c, seed):
    rng = np.random.default_rng(seed)
    d1, d2, _, _ = synth.make_descriptor_sets(T, Q, seed + 10)
    lens = rng.poisson(mean_c, Q)
    lens[rng.random(Q) < 0.1] = 0  # empty windows
    off = np.zeros(Q + 1, np.int32)
    off[1:] = np.cumsum(lens)
    idx = rng.integers(0, T, off[-1]).astype(np.int32)
    dist, bi, bd, sd = matcher.hamming_csr(ctx, d2, d1, off, idx)
    odist, obi, obd, osd = oracle_lib.hamming_csr(d2, d1, off, idx)
    assert np.array_equal(dist, odist)
    assert np.array_equal(bd, obd) and np.array_equal(sd, osd) and np.array_equal(bi, obi)


@pytest.mark.parametrize("mean_c", [0.7, 3, 11, 20, 45, 90, 400])
def test_csr_every_group_width_matches_oracle(ctx, oracle_lib, mean_c):
    """round 5: the windowed search runs with 8 / 16 / 32 / 64 lanes per query, chosen from the mean list length — every width, with empty, one-element and
    overlong lists mixed in, gives the oracle's integers (distance per slot, first minimum, second minimum)"""
    rng = np.random.default_rng(int(mean_c * 10))
    Q, T = 1500, 900
    d1, d2, _, _ = synth.make_descriptor_sets(T, Q, 77)
    lens = rng.poisson(mean_c, Q)
    lens[rng.random(Q) < 0.08] = 0
    lens[rng.random(Q) < 0.05] = 1
    lens[rng.integers(0, Q, 3)] = 700                       # a few lists far longer than the group is wide
    off = np.zeros(Q + 1, np.int32); off[1:] = np.cumsum(lens)
    idx = rng.integers(0, T, off[-1]).astype(np.int32)
    idx[off[5]:off[6]] = idx[off[5]] if lens[5] else 0      # a list of identical candidates: the FIRST one must win
    dist, bi, bd, sd = matcher.hamming_csr(ctx, d2, d1, off, idx)
    odist, obi, obd, osd = oracle_lib.hamming_csr(d2, d1, off, idx)
    assert np.array_equal(dist, odist) and np.array_equal(bd, obd) and np.array_equal(sd, osd) and np.array_equal(bi, obi)


def test_csr_multi_twenty_searches_in_one_launch_match_the_oracle_search_by_search(ctx, oracle_lib):
    """ccm_hamming_csr_multi: 20 searches, each against ITS OWN target set (candidate indices local to it), one launch, one read-back = the oracle on every search alone;
    includes a search without queries, one without candidates and target sets of different sizes"""
    rng = np.random.default_rng(5)
    qs, ts, offs, idxs = [], [], [], []
    for s in range(20):
        Q = 0 if s == 7 else int(rng.integers(300, 900)); T = int(rng.integers(500, 1200))
        t, q, _, _ = synth.make_descriptor_sets(T, max(Q, 1), 100 + s)
        q = q[:Q]
        lens = rng.poisson(22, Q) if s != 11 else np.zeros(Q, np.int64)
        off = np.zeros(Q + 1, np.int32); off[1:] = np.cumsum(lens)
        qs.append(q); ts.append(t); offs.append(off); idxs.append(rng.integers(0, T, off[-1]).astype(np.int32))
    dist, bi, bd, sd, q_off, c_base = matcher.hamming_csr_multi(ctx, qs, ts, offs, idxs)
    for s in range(20):
        odist, obi, obd, osd = oracle_lib.hamming_csr(qs[s], ts[s], offs[s], idxs[s])
        a, b = int(q_off[s]), int(q_off[s + 1])
        assert np.array_equal(dist[c_base[s]:c_base[s + 1]], odist), s
        assert np.array_equal(bi[a:b], obi) and np.array_equal(bd[a:b], obd) and np.array_equal(sd[a:b], osd), s


def test_full_size_property_self_match(ctx):
    # BASELINE config 5 shape: 2000 x 2000; every row's nearest neighbour in its own set is itself (distance 0)
    rng = np.random.default_rng(9)
    d = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    bi, bd, sd = matcher.hamming_dense_best2(ctx, d, d)
    assert np.array_equal(bi, np.arange(2000)) and (bd == 0).all() and (sd > 0).all()


def test_distinctive_descriptors_matches_oracle(ctx, oracle_lib):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cpp:929-994) batched: median-of-distances medoid, first minimum."""
    rng = np.random.default_rng(12)
    counts = np.concatenate([[0, 1, 2, 3, 64, 65, 200, 256], rng.integers(1, 40, 3000)])
    off = np.zeros(counts.size + 1, np.int32); off[1:] = np.cumsum(counts)
    base = rng.integers(0, 256, (counts.size, 32), dtype=np.uint8)
    desc = np.repeat(base, counts, axis=0)
    bits = np.unpackbits(desc, axis=1)
    desc = np.packbits(bits ^ (rng.random(bits.shape) < 0.1), axis=1)
    desc[off[6]:off[6] + 5] = desc[off[6]]      # exact duplicates -> ties, first index must win
    got = matcher.distinctive_descriptors(ctx, desc, off)
    exp = oracle_lib.distinctive_descriptors(desc, off)
    assert np.array_equal(got, exp) and got[0] == -1 and got[1] == 0


@pytest.mark.parametrize("k,L,levelsup", [(10, 4, 2), (10, 3, 4), (7, 5, 4)])
def test_bow_transform_matches_oracle(ctx, oracle_lib, k, L, levelsup):
    """DBoW2 TemplatedVocabulary::transform: tree descent on the device (bit-exact word / node ids, first-minimum child),
    BowVector accumulation + L1 normalisation on the host (f64, same operation order => identical doubles)."""
    vocab = synth.make_vocabulary(k, L, seed=3)
    rng = np.random.default_rng(5)
    leaves = np.nonzero(vocab["word_id"] >= 0)[0]
    src = vocab["node_desc"][rng.choice(leaves, 1500)]
    bits = np.unpackbits(src, axis=1)
    desc = np.packbits(bits ^ (rng.random(bits.shape) < 0.05), axis=1)
    desc[:20] = rng.integers(0, 256, (20, 32), dtype=np.uint8)
    V = matcher.Vocabulary(ctx, vocab)
    word, w, node, ids, vals, fv = V.transform(desc, levelsup)
    oword, ow, onode, oids, ovals = oracle_lib.bow_transform(vocab, desc, levelsup)
    assert np.array_equal(word, oword) and np.array_equal(node, onode) and np.array_equal(w, ow)
    assert np.array_equal(ids, oids) and np.array_equal(vals, ovals)
    if L - levelsup <= 0:
        assert (node == 0).all()
    # FeatureVector invariants: ascending nodes, every kept feature exactly once, features ascending inside a node
    fn, fo, fi = fv
    assert (np.diff(fn) > 0).all() and fi.size == (w > 0).sum() and all((np.diff(fi[fo[i]:fo[i + 1]]) > 0).all() for i in range(fn.size))
    V.close()
