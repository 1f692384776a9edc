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
