"""GPU, BASELINE.json full size (C2: 1M pre-aligned reads, ~20k-node 16S-shaped database):
size-independent properties of the path, plus an oracle comparison on a subsample."""
import numpy as np
import pytest

import bench_workload as bw
from oracle import oracle_c
from stag_amd import _lib
from stag_amd.engine import Database

pytestmark = pytest.mark.gpu
N = 1_000_000


@pytest.fixture(scope="module")
def c2():
    wl = bw.config_16s(seed=2)
    text, offs, leaf = bw.make_reads(wl, seed=1000, n=N)
    return wl, text, offs, leaf


def run(db, text, offs, idx=None):
    """classify reads idx (all when None) from the packed text through the host-buffer C ABI."""
    import ctypes
    from stag_amd.engine import BatchResult
    if idx is not None:
        lens = (offs[1:] - offs[:-1])[idx]
        o2 = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(lens, out=o2[1:])
        src = np.repeat(offs[:-1][idx] - o2[:-1], lens) + np.arange(o2[-1])
        text, offs = np.concatenate([text[src], np.zeros(16, np.uint8)]), o2
    n = len(offs) - 1
    res = BatchResult(n, db.max_depth, db.n_keys)
    st = res.as_struct()
    _lib.check(_lib.lib().stag_classify_a2m(db.handle, _lib.ptr(text), _lib.ptr(offs), n, 0.0, ctypes.byref(st),
                                            None, None))
    return res


def test_c2_full_size_properties(c2):
    wl, text, offs, leaf = c2
    t = wl.tables
    with Database(t) as db:
        res = run(db, text, offs)
        assert (res.status == 0).all()
        # (1) every path is a root-to-leaf chain of the tree, as long as the tree is deep
        cf = t.child_first.astype(np.int64)
        prev = np.zeros(N, dtype=np.int64)
        for d in range(db.max_depth):
            node = res.path_node[:, d].astype(np.int64)
            assert ((node >= cf[prev]) & (node < cf[prev + 1])).all()
            prev = node
        assert (res.path_len == db.max_depth).all() and (cf[prev + 1] == cf[prev]).all()
        # (2) probabilities are probabilities; the rank cut picks the arg-max key
        assert ((res.path_prob >= 0) & (res.path_prob <= 1)).all()
        assert ((res.level_prob > 0) & (res.level_prob <= 1)).all()
        assert (res.selected == res.level_prob.argmax(axis=1)).all()
        # (3) n_aligned = number of A/C/G/T/U characters of the text (checksum of checksums)
        is_nt = np.isin(text[:offs[-1]], np.frombuffer(b"ACGTU", dtype=np.uint8))
        csum = np.concatenate([[0], np.cumsum(is_nt)])
        assert np.array_equal(res.n_aligned, (csum[offs[1:]] - csum[offs[:-1]]).astype(np.int32))
        # (4) idempotence: the same batch again on the same handle is bit-identical
        res2 = run(db, text, offs)
        for f in ("path_node", "path_prob", "level_prob", "selected", "n_aligned", "status"):
            assert np.array_equal(getattr(res, f), getattr(res2, f)), f
        # (5) order independence: a shuffled sub-batch with duplicates gives the same rows bit for bit
        rng = np.random.default_rng(4)
        idx = rng.integers(0, N, size=200_000)
        idx[1::2] = idx[0::2]                         # every read twice, adjacent
        idx = rng.permutation(idx)
        sub = run(db, text, offs, idx)
        for f in ("path_node", "path_prob", "level_prob", "selected", "n_aligned"):
            assert np.array_equal(getattr(sub, f), getattr(res, f)[idx]), f
        # (6) sanity of the workload: reads spread over thousands of leaves, and far more of them come
        # back to the leaf they were drawn from than chance (1e-4) would give
        assert len(np.unique(res.path_node[:, db.max_depth - 1])) > 2000
        assert (res.path_node[:, db.max_depth - 1] == leaf).mean() > 0.02
        # (7) oracle on a subsample: lineage identical, probabilities within 1e-9
        m = 1500
        pick = rng.choice(N, size=m, replace=False)
        X = np.stack([oracle_c.encode_a2m(bytes(text[offs[i]:offs[i + 1]]).decode("latin-1"))[0] for i in pick])
        out = oracle_c.DenseDB.from_packed(t).classify(X)
        assert np.array_equal(out["path_node"], res.path_node[pick])
        np.testing.assert_allclose(res.path_prob[pick], out["path_prob"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(res.level_prob[pick], out["level_prob"], rtol=1e-9, atol=0)
        assert np.array_equal(out["selected"], res.selected[pick])


def dense_from_text(text, offs, idx, L):
    return np.stack([oracle_c.encode_a2m(bytes(text[offs[i]:offs[i + 1]]).decode("latin-1"))[0] for i in idx])


def test_c3_shape_full_length_reads():
    """BASELINE configs[2] shape (protein-derived database: ncol 1050, reads cover every column, so
    no row equals the all-gaps row anywhere and the walk stages full rows), 200k reads."""
    wl = bw.make_db(3, 1050, [4, 12, 40, 120, 300, 600, 900], 0.02, name="C3")
    n = 200_000
    text, offs, leaf = bw.make_reads(wl, seed=31, n=n, window=1050)
    t = wl.tables
    with Database(t) as db:
        res = run(db, text, offs)
        assert (res.status == 0).all()
        cf = t.child_first.astype(np.int64)
        prev = np.zeros(n, dtype=np.int64)
        for d in range(db.max_depth):
            node = res.path_node[:, d].astype(np.int64)
            assert ((node >= cf[prev]) & (node < cf[prev + 1])).all()
            prev = node
        is_nt = np.isin(text[:offs[-1]], np.frombuffer(b"ACGTU", dtype=np.uint8))
        csum = np.concatenate([[0], np.cumsum(is_nt)])
        assert np.array_equal(res.n_aligned, (csum[offs[1:]] - csum[offs[:-1]]).astype(np.int32))
        res2 = run(db, text, offs)
        for f in ("path_node", "path_prob", "level_prob", "selected"):
            assert np.array_equal(getattr(res, f), getattr(res2, f)), f
        rng = np.random.default_rng(8)
        idx = rng.permutation(np.repeat(rng.integers(0, n, size=20_000), 2))
        sub = run(db, text, offs, idx)
        for f in ("path_node", "path_prob", "level_prob", "selected"):
            assert np.array_equal(getattr(sub, f), getattr(res, f)[idx]), f
        assert (res.path_node[:, db.max_depth - 1] == leaf).mean() > 0.2     # chance would be 1e-3
        pick = rng.choice(n, size=600, replace=False)
        out = oracle_c.DenseDB.from_packed(t).classify(dense_from_text(text, offs, pick, t.n_features))
        assert np.array_equal(out["path_node"], res.path_node[pick])
        np.testing.assert_allclose(res.path_prob[pick], out["path_prob"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(res.level_prob[pick], out["level_prob"], rtol=1e-9, atol=0)


def test_c4_shape_concatenated_genome_rows():
    """BASELINE configs[3] shape: the concatenated database of 30 marker genes (L = 30 x 5,250 =
    157,500 features, 4,924 words per row, 308 coefficient chunks), genomes lacking some genes
    (zero-filled blocks, classify_genome.py:375).  Encode at that width, then classify."""
    from stag_amd import tables
    n_genes, gcol = 30, 1050
    ncol = n_genes * gcol
    L = 5 * ncol
    wl = bw.make_db(4, ncol, [3, 6, 12, 24, 40, 60, 80], 0.02, name="C4")
    n = 2000
    text, offs, leaf = bw.make_reads(wl, seed=41, n=n, window=ncol, max_ins=0)
    t = wl.tables
    rng = np.random.default_rng(9)
    with Database(t) as db:
        assert db.words_per_row == 4924
        full = db.classify_a2m_text(text, offs, want_bits=True)
        assert (full.status == 0).all()
        pick = rng.choice(n, size=40, replace=False)
        Xp = dense_from_text(text, offs, pick, L)
        assert np.array_equal(tables.unpack_rows(full.bits[pick], L), Xp)
        # genomes lacking genes: zero the 5,250-feature block of every missing gene
        X = tables.unpack_rows(full.bits, L)
        missing = rng.random((n, n_genes)) < 0.1
        X &= ~np.repeat(missing, 5 * gcol, axis=1)
        res = db.classify_onehot(X)
        assert (res.status == 0).all()
        res2 = db.classify_onehot(X[::-1].copy())
        for f in ("path_node", "path_prob", "level_prob", "selected", "n_aligned"):
            assert np.array_equal(getattr(res2, f), getattr(res, f)[::-1]), f
        assert np.array_equal(res.n_aligned, (ncol - X[:, 0::5].sum(axis=1)).astype(np.int32))
        pick = rng.choice(n, size=150, replace=False)
        out = oracle_c.DenseDB.from_packed(t).classify(X[pick])
        assert np.array_equal(out["path_node"], res.path_node[pick])
        np.testing.assert_allclose(res.path_prob[pick], out["path_prob"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(res.level_prob[pick], out["level_prob"], rtol=1e-9, atol=0)
        assert np.array_equal(out["selected"], res.selected[pick])


def test_wide_levels_use_every_grouping_path():
    """A tree whose levels have 3, 150, ~21,000 and ~30,000 nodes: grouping goes through the ballot
    path (<= 32 nodes), the LDS histogram (<= 15,360) and the per-wave path (wider), and the
    deepest scored level mixes single-child nodes, leaves and siblings (BASELINE config C5 has
    50,000 nodes).  Every read is compared with the oracle."""
    wl = bw.make_db(5, 60, [3, 150, 21000, 30000], 0.1, name="wide")
    t = wl.tables
    level = [0, 1]                                       # nodes of depth d are [level[d], level[d + 1])
    while level[-1] < t.n_nodes and t.child_first[level[-1]] > level[-1]:
        level.append(int(t.child_first[level[-1]]))
    widths = np.diff(level).tolist()
    assert widths[:3] == [1, 3] + widths[2:3] and 32 < widths[2] <= 15_360 < widths[3] and t.n_nodes > 50_000
    n = 60_000
    text, offs, leaf = bw.make_reads(wl, seed=51, n=n, window=60)
    with Database(t) as db:
        assert db.max_depth == 4
        res = run(db, text, offs)
        assert (res.status == 0).all()
        X = dense_from_text(text, offs, np.arange(n), t.n_features)
        out = oracle_c.DenseDB.from_packed(t).classify(X)
        assert np.array_equal(out["path_len"], res.path_len)
        assert np.array_equal(out["path_node"], res.path_node)
        np.testing.assert_allclose(res.path_prob, out["path_prob"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(res.level_prob, out["level_prob"], rtol=1e-9, atol=0)
        assert np.array_equal(out["selected"], res.selected)
        res2 = run(db, text, offs, np.arange(n)[::-1].copy())
        assert np.array_equal(res2.path_node, res.path_node[::-1]) and np.array_equal(res2.path_prob, res.path_prob[::-1])


def test_matrix_core_walk_equals_gather_walk_on_c2(c2, monkeypatch):
    """The two walk kernels on 200k C2 reads: same lineage for every read, probabilities within 1e-9 of each
    other (the matrix-core walk rounds coefficients to multiples of 2^-q, q >= 39), and the matrix-core walk is
    bit-identical under shuffling (integer sums do not depend on the order of anything)."""
    wl, text, offs, leaf = c2
    t = wl.tables
    idx = np.arange(0, N, 5)
    monkeypatch.setenv("STAG_WALK", "gather")
    with Database(t) as db:
        ref = run(db, text, offs, idx)
    monkeypatch.setenv("STAG_WALK", "mfma")
    with Database(t) as db:
        got = run(db, text, offs, idx)
        perm = np.random.default_rng(3).permutation(len(idx))
        again = run(db, text, offs, idx[perm])
    assert np.array_equal(got.path_node, ref.path_node) and np.array_equal(got.selected, ref.selected)
    np.testing.assert_allclose(got.path_prob, ref.path_prob, rtol=1e-9, atol=0)
    np.testing.assert_allclose(got.level_prob, ref.level_prob, rtol=1e-9, atol=0)
    assert not np.array_equal(got.path_prob, ref.path_prob)          # (they are different kernels)
    for f in ("path_node", "path_prob", "level_prob", "selected"):
        assert np.array_equal(getattr(again, f), getattr(got, f)[perm]), f
