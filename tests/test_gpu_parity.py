"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Bit-exact for encodings, ids, counts and text; 1e-9 relative for
probabilities (north_star); lineage identical unless the oracle's own margin at the first
differing level is below that tolerance."""
import ctypes

import numpy as np
import pytest

import golden_util as gu
from oracle import oracle_c, oracle_np
from stag_amd import _lib, tables
from stag_amd.engine import Database, raise_for_status

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.fixture(autouse=True, params=["mfma", "gather"])
def walk_kernel(request, monkeypatch):
    """Every test of this file runs twice: with STAG_WALK=mfma, which sends every node that can be through the
    matrix-core kernel k_walk_mfma (nodes with more than 16 children or non-finite coefficients still take the
    gather kernel), and with STAG_WALK=gather, which sends every node through k_walk_level.  (Left alone,
    stag_db_create picks per node by the density of the children's rows.)"""
    monkeypatch.setenv("STAG_WALK", request.param)
    return request.param


def encode_raw(strings, L):
    text, offs = tables.pack_text(strings)
    n = len(strings)
    W = _lib.lib().stag_words_per_row(L)
    bits = np.zeros((n, W), dtype=np.uint32)
    n_char, n_al, status = (np.zeros(n, dtype=np.int32) for _ in range(3))
    _lib.check(_lib.lib().stag_encode_a2m(0, L, _lib.ptr(text), _lib.ptr(offs), n, 0.0, _lib.ptr(bits),
                                          _lib.ptr(n_char), _lib.ptr(n_al), _lib.ptr(status)))
    return bits, n_char, n_al, status


def test_encode_golden_bit_exact():
    cases = gu.load("encode.json")
    L = 5 * max(len(c["a2m"]) for c in cases)
    bits, n_char, n_al, status = encode_raw([c["a2m"] for c in cases], L)
    dense = tables.unpack_rows(bits, bits.shape[1] * 32)
    for i, c in enumerate(cases):
        if "error" in c:
            assert status[i] == _lib.SEQ_ZERO_COLUMNS and n_char[i] == 0
            assert not dense[i].any()
            continue
        exp = gu.bits(c["bits"])
        assert n_char[i] * 5 == len(exp)
        assert np.array_equal(dense[i, :len(exp)], exp), c["a2m"]
        assert not dense[i, len(exp):].any()
        assert n_al[i] / n_char[i] * 100 == float(c["pct"])
        assert status[i] in (_lib.SEQ_OK, _lib.SEQ_LENGTH_MISMATCH)


def test_encode_random_vs_oracle_with_filter_and_newlines():
    rng = np.random.default_rng(5)
    ncol = 300
    alphabet = np.array(list("ACGTU-.NRY*acgtn"))
    p = np.array([8, 8, 8, 8, 2, 10, 1, 1, 1, 1, 0.5] + [0.6] * 5)
    p /= p.sum()
    strings, clean = [], []
    for i in range(500):
        s = "".join(rng.choice(alphabet, size=int(rng.integers(1, 700)), p=p))
        clean.append(s)
        if i % 3 == 0:      # line-wrapped record: '\n' / '\r\n' must be skipped
            w = int(rng.integers(1, 80))
            s = ("\r\n" if i % 2 else "\n").join(s[j:j + w] for j in range(0, len(s), w)) + "\n"
        strings.append(s)
    db_L = 5 * ncol
    text, offs = tables.pack_text(strings)
    n = len(strings)
    W = _lib.lib().stag_words_per_row(5 * 700)
    bits = np.zeros((n, W), dtype=np.uint32)
    n_char, n_al, status = (np.zeros(n, dtype=np.int32) for _ in range(3))
    _lib.check(_lib.lib().stag_encode_a2m(0, 5 * 700, _lib.ptr(text), _lib.ptr(offs), n, 40.0, _lib.ptr(bits),
                                          _lib.ptr(n_char), _lib.ptr(n_al), _lib.ptr(status)))
    dense = tables.unpack_rows(bits, W * 32)
    for i, s in enumerate(clean):
        exp, e_char, e_al = oracle_c.encode_a2m(s)
        assert n_char[i] == e_char and n_al[i] == e_al
        assert np.array_equal(dense[i, :5 * e_char], exp)
        assert not dense[i, 5 * e_char:].any()
        if e_char == 0:
            assert status[i] == _lib.SEQ_ZERO_COLUMNS
        elif not oracle_c.passes_filter(e_al, e_char, 40.0):
            assert status[i] == _lib.SEQ_FILTERED
        else:
            assert status[i] == (_lib.SEQ_OK if e_char == 700 else _lib.SEQ_LENGTH_MISMATCH)
    del db_L


@pytest.mark.parametrize("mean_len,n", [(20, 300), (200, 300), (520, 200), (1100, 150), (4000, 40)])
def test_encode_every_lane_grouping(mean_len, n):
    """The encode kernel runs with 16, 32 or 64 lanes per record depending on the average record
    length; every variant, with gap runs, inserts, ragged lengths and an empty record."""
    rng = np.random.default_rng(mean_len)
    strings = []
    for i in range(n):
        m = max(1, int(rng.normal(mean_len, mean_len * 0.1)))
        s = np.full(m, "-", dtype="<U1")
        a, b = sorted(rng.integers(0, m + 1, size=2))
        s[a:b] = rng.choice(list("ACGTUN"), size=b - a)
        s = list(s)
        for _ in range(int(rng.integers(0, 4))):
            s.insert(int(rng.integers(0, len(s) + 1)), str(rng.choice(list("acgt"))))
        strings.append("".join(s))
    strings[n // 2] = ""
    L = 5 * max(len(x) for x in strings)
    bits, n_char, n_al, status = encode_raw(strings, L)
    dense = tables.unpack_rows(bits, bits.shape[1] * 32)
    for i, s in enumerate(strings):
        exp, e_char, e_al = oracle_c.encode_a2m(s)
        assert n_char[i] == e_char and n_al[i] == e_al, (i, len(s))
        assert np.array_equal(dense[i, :5 * e_char], exp), (i, len(s))
        assert not dense[i, 5 * e_char:].any()
        assert status[i] == (_lib.SEQ_ZERO_COLUMNS if e_char == 0 else
                             _lib.SEQ_OK if 5 * e_char == L else _lib.SEQ_LENGTH_MISMATCH)


@pytest.mark.parametrize("case", gu.load("classify_seq.json"), ids=lambda c: c["name"])
def test_classify_seq_golden(case):
    taxonomy, tax_function, classifiers = gu.db_from_json(case)
    X = np.stack([gu.bits(s) for s in case["seqs"]])
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=case["L"]) as db:
        res = db.classify_onehot(X)
        for i, exp in enumerate(case["expected"]):
            st = int(res.status[i])
            if "error" in exp:
                assert st != 0 and gu.ERROR_OF_STATUS[st] in exp["error"], (st, exp["error"])
                with pytest.raises(Exception) as ei:
                    raise_for_status(res.status[i:i + 1])
                assert any(c.__name__ in exp["error"] for c in type(ei.value).__mro__[:-2])
            else:
                assert st == 0
                gu.rows_match(db.result_row(res, i, "seq%d" % i), exp["row"], rtol=RTOL)


def test_sklearn_predict_proba_parity_on_gpu():
    g = gu.load("sklearn_parity.json")
    row = np.array([float(v) for v in g["row"]])
    other = row.copy()
    other[0] -= 100.0       # sibling that never wins: the winner's probability is the sklearn model's
    taxonomy = {"tree_root": ["yes", "other"], "yes": [], "other": []}
    X = np.stack([gu.bits(s) for s in g["X"]])
    with Database.from_dicts(taxonomy, {"0": np.array([0.0, 1.0])}, {"yes": row, "other": other}) as db:
        res = db.classify_onehot(X)
    assert (res.status == 0).all() and (res.path_node[:, 0] == 1).all()
    p_sk = np.array([float(v) for v in g["predict_proba"]])
    np.testing.assert_allclose(res.path_prob[:, 0], p_sk, rtol=RTOL, atol=0)


def random_db(rng, ncol, fanouts, density, scale=1.0):
    L = 5 * ncol
    taxonomy, classifiers = {"tree_root": []}, {}
    frontier, nid = ["tree_root"], 0
    for choices in fanouts:
        nxt = []
        for parent in frontier:
            k = int(rng.choice(choices))
            kids = []
            for _ in range(k):
                nid += 1
                kids.append("n%d" % nid)
            taxonomy[parent] = kids
            for kid in kids:
                taxonomy[kid] = []
                if k == 1:
                    classifiers[kid] = "no_negative_examples"
                else:
                    w = rng.normal(0, scale, size=1 + L) * (rng.random(1 + L) < density)
                    w[0] = rng.normal(0, 1)
                    classifiers[kid] = w
            nxt.extend(kids)
        frontier = nxt
    D = len(fanouts)
    tax_function = {str(l): rng.normal(0, 1, size=1 + D) for l in range(-1, D)}
    return taxonomy, tax_function, classifiers


def compare_with_oracle(db, odb, res, out, X, classifiers):
    """Exact on ids/counts/status; RTOL on probabilities; lineage equal unless the margin is tiny."""
    n = res.n
    assert np.array_equal(res.status, out["status"])
    assert np.array_equal(res.n_aligned, out["n_aligned"])
    n_diverged = 0
    for i in range(n):
        if res.status[i] != 0:
            continue
        g = [db.names[j] for j in res.path_node[i, :res.path_len[i]]]
        e = [odb.names[j] for j in out["path_node"][i, :out["path_len"][i]]]
        if g != e:      # accept only a genuine near-tie at the first differing level
            d = next(k for k in range(min(len(g), len(e))) if g[k] != e[k])
            pg = float(oracle_np.run_logistic_prediction(X[i], classifiers[g[d]]))
            pe = float(oracle_np.run_logistic_prediction(X[i], classifiers[e[d]]))
            assert abs(pg - pe) <= RTOL * max(pg, pe), (i, g, e, pg, pe)
            n_diverged += 1
            continue
        m = len(g)
        np.testing.assert_allclose(res.path_prob[i, :m], out["path_prob"][i, :m], rtol=RTOL, atol=0)
        lead_o = 0
        while lead_o < m and (out["single_mask"][i] >> lead_o) & 1:
            lead_o += 1
        assert res.n_lead_single[i] == lead_o
        np.testing.assert_allclose(res.level_prob[i], out["level_prob"][i], rtol=RTOL, atol=0)
        if res.selected[i] != out["selected"][i]:
            lp = out["level_prob"][i]
            a, b = lp[res.selected[i]], lp[out["selected"][i]]
            assert abs(a - b) <= RTOL * max(a, b)
    assert n_diverged <= max(1, n // 1000)


@pytest.mark.parametrize("ncol,fanouts,density,n,kind", [
    (29, [[4], [1, 2, 3], [0, 1, 2]], 0.3, 50, "onehot"),                 # C1 shape: L=145, tiny batch
    (400, [[3, 5], [1, 2, 6], [1, 3, 9], [2, 3]], 0.05, 3000, "onehot"),   # fan-out > CB, groups > 64 and < 64
    (1500, [[6], [2, 7], [1, 4]], 0.02, 700, "window"),                    # C2 row width, gap-padded reads
    (3200, [[5], [2, 3]], 0.01, 300, "zerofill"),                          # L=16000: 3 LDS chunks per row
    (64, [[11], [1, 17]], 0.5, 2000, "anybits"),                           # dense rows, wide fan-out, -s style bits
    # windows wider than the walk's staging buffer (5 and 7 chunks) with rows dense enough that a slice's
    # coefficients do not fit a wave's buffer: the slice-by-slice path of k_walk_level and its refill branch
    (411, [[5], [2, 3, 18]], 0.3, 500, "onehot"),
    (411, [[3], [2, 20]], 1.0, 300, "onehot"),
    (700, [[4], [2, 3, 17]], 0.3, 400, "onehot"),
    (700, [[2], [19, 3]], 1.0, 200, "anybits"),
])
def test_classify_random_vs_oracle(ncol, fanouts, density, n, kind):
    rng = np.random.default_rng(ncol * 7 + n)
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, fanouts, density)
    X = np.zeros((n, L), dtype=bool)
    sym = rng.integers(0, 5, size=(n, ncol))
    if kind == "anybits":
        X = rng.random((n, L)) < 0.2
    else:
        if kind == "window":        # amplicon-like: '-' outside a 253-column window
            start = rng.integers(0, ncol - 253, size=n)
            cols = np.arange(ncol)[None, :]
            inside = (cols >= start[:, None]) & (cols < start[:, None] + 253)
            sym = np.where(inside, rng.integers(1, 5, size=(n, ncol)), 0)
        X[np.arange(n)[:, None], 5 * np.arange(ncol)[None, :] + sym] = True
        if kind == "zerofill":
            X[:, (5 * np.arange(ncol)[:, None] + np.arange(5)[None, :])[rng.random(ncol) < 0.3].ravel()] = False
    odb = oracle_c.DenseDB(taxonomy, tax_function, classifiers, L)
    out = odb.classify(X)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        assert db.max_depth == odb.max_depth
        res = db.classify_onehot(X)
        compare_with_oracle(db, odb, res, out, X, classifiers)
        # same batch again on the same handle, smaller: workspace reuse must not leak state
        res2 = db.classify_onehot(X[: n // 3])
        assert np.array_equal(res2.path_node, res.path_node[: n // 3])
        assert np.array_equal(res2.path_prob, res.path_prob[: n // 3])


@pytest.mark.parametrize("ncol,density", [(411, 0.3), (411, 1.0), (700, 0.3), (700, 1.0)])
def test_wide_dense_rows_text_and_packed_entries(ncol, density):
    """The same wide, dense shapes through the TEXT entry point (whose rows are stored as window chunks only) and
    through the packed-row entry point: full-length reads, reads with a window at either end, an all-gap read."""
    rng = np.random.default_rng(ncol + int(10 * density))
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[3], [2, 18], [1, 2]], density)
    strings = []
    for i in range(260):
        cols = rng.choice(list("ACGT-N"), size=ncol, p=[0.23, 0.23, 0.23, 0.23, 0.05, 0.03])
        if i % 4 == 1:
            cols[: int(rng.integers(1, ncol))] = "-"
        if i % 4 == 2:
            cols[int(rng.integers(1, ncol)):] = "-"
        if i == 77:
            cols[:] = "-"
        strings.append("".join(cols))
    X = np.stack([oracle_c.encode_a2m(s)[0] for s in strings])
    odb = oracle_c.DenseDB(taxonomy, tax_function, classifiers, L)
    out = odb.classify(X)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        text = db.classify_a2m(strings)                       # rows stay on the GPU, window chunks only
        full = db.classify_a2m(strings, want_bits=True)       # whole rows
        packed = db.classify_onehot(X)
        compare_with_oracle(db, odb, packed, out, X, classifiers)
        for f in ("status", "n_aligned", "path_len", "path_node", "path_prob", "selected", "level_prob"):
            assert np.array_equal(getattr(text, f), getattr(packed, f)), f
            assert np.array_equal(getattr(full, f), getattr(packed, f)), f
        assert np.array_equal(tables.unpack_rows(full.bits, L), X)


@pytest.mark.parametrize("strings", [[""], ["", ""], ["acgt"], ["", "acgt", ""], [""] * 70 + ["A-C"]],
                         ids=["one-empty", "two-empty", "inserts-only", "mixed-empty", "many-empty-then-one"])
def test_batches_of_empty_records(strings):
    """A batch whose text is zero bytes long (one empty record as the whole batch: the configuration that left the
    GPU faulted in the differential fuzz -- the encode buffers stayed unallocated for a text span of 0 bytes and
    the encode kernel wrote its status to address 0), and its neighbours.  The reference raises ZeroDivisionError
    for such a record (align.py:58)."""
    rng = np.random.default_rng(1)
    taxonomy, tax_function, classifiers = random_db(rng, 3, [[3], [2, 2]], 0.5)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=15) as db:
        for want_bits in (True, False):
            res = db.classify_a2m(strings, want_bits=want_bits)
            for i, s in enumerate(strings):
                n_char = oracle_c.encode_a2m(s)[1]
                assert res.n_char[i] == n_char
                assert res.status[i] == (_lib.SEQ_ZERO_COLUMNS if n_char == 0 else _lib.SEQ_OK)
            with pytest.raises(ZeroDivisionError):
                raise_for_status(res.status)


def test_classify_a2m_fused_vs_oracle():
    """text -> encode -> walk on the GPU against oracle encode + oracle classify, with inserts,
    a min_perc_state filter, an all-insert record and a wrong-length record."""
    rng = np.random.default_rng(77)
    ncol = 120
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [2, 3], [1, 2]], 0.1)
    strings = []
    for i in range(400):
        cols = rng.choice(list("ACGT-N"), size=ncol, p=[0.2, 0.2, 0.2, 0.2, 0.15, 0.05])
        if i % 5 == 0:
            cols[: int(rng.integers(20, ncol))] = "-"
        s = list(cols)
        for _ in range(int(rng.integers(0, 4))):
            s.insert(int(rng.integers(0, len(s) + 1)), rng.choice(list("acgt")))
        strings.append("".join(s))
    strings[7] = "acgtacgt"                 # ZeroDivisionError in the reference
    strings[11] = strings[11][:-3]          # wrong number of columns
    min_perc = 55.0
    odb = oracle_c.DenseDB(taxonomy, tax_function, classifiers, L)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        res = db.classify_a2m(strings, min_perc_state=min_perc, want_bits=True)
        keep, X = [], []
        for i, s in enumerate(strings):
            bits, n_char, n_al = oracle_c.encode_a2m(s)
            assert res.n_char[i] == n_char
            if n_char == 0:
                assert res.status[i] == _lib.SEQ_ZERO_COLUMNS
            elif not oracle_c.passes_filter(n_al, n_char, min_perc):
                assert res.status[i] == _lib.SEQ_FILTERED
            elif 5 * n_char != L:
                assert res.status[i] == _lib.SEQ_LENGTH_MISMATCH
            else:
                keep.append(i)
                X.append(bits)
                assert np.array_equal(tables.unpack_rows(res.bits[i:i + 1], L)[0], bits)
        assert len(keep) > 100 and (res.status != 0).sum() > 20
        X = np.stack(X)
        out = odb.classify(X)
        sub = db.classify_onehot(X)
        compare_with_oracle(db, odb, sub, out, X, classifiers)
        keep = np.asarray(keep)
        assert np.array_equal(res.path_node[keep], sub.path_node)
        assert np.array_equal(res.path_prob[keep], sub.path_prob)
        assert np.array_equal(res.selected[keep], sub.selected)
        assert np.array_equal(res.n_aligned[keep], sub.n_aligned)
        with pytest.raises(ZeroDivisionError):
            raise_for_status(res.status)


def test_tsv_rows_match_reference_text():
    g = gu.load("tsv.json")
    taxonomy, tax_function, classifiers = gu.db_from_json(g["db"])
    ids, X = zip(*(oracle_np.parse_onehot_line(l) for l in g["ali_lines"]))
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=g["L"]) as db:
        res = db.classify_onehot(np.stack(X))
        short = db.result_rows(res, ids, long_out=False)
        assert short == g["runs"]["internal"]["return_rows"]
        long_rows = db.result_rows(res, ids, long_out=True)
        exp = [l.split("\t") for l in g["runs"]["long"]["file"].splitlines()[1:]]
        for got, e in zip(long_rows, exp):
            got = list(got)
            got[3] = str(got[3])
            gu.rows_match(got, e, rtol=RTOL)


def test_empty_batch_and_bad_arguments():
    taxonomy = {"tree_root": ["a", "b"], "a": [], "b": []}
    cls = {"a": np.ones(11), "b": -np.ones(11)}
    with Database.from_dicts(taxonomy, {"0": np.array([0.0, 1.0])}, cls) as db:
        res = db.classify_onehot(np.zeros((0, 10), dtype=bool))
        assert res.n == 0
        res = db.classify_a2m([])
        assert res.n == 0
        with pytest.raises(ValueError):
            db.classify_bits(np.zeros((3, 1), dtype=np.uint32))
    bad = tables.pack_db(taxonomy, {"0": np.array([0.0, 1.0])}, cls)
    bad.child_first = np.array([1, 3, 2, 3], dtype=np.int32)   # not monotone
    with pytest.raises(_lib.StagError):
        Database(bad)


# ------------------------------------------------------------------ protein databases (align.py:63-94)
def backtranslate_gpu(pairs, check_length=True):
    """stag_backtranslate_dev on device buffers (tests/hipmem.py supplies the memory)."""
    from hipmem import DevArray
    pt, po = tables.pack_text([p for p, _ in pairs])
    gt, go = tables.pack_text([g for _, g in pairs])
    n = len(pairs)
    d_pt, d_po, d_gt, d_go = DevArray(pt), DevArray(po), DevArray(gt), DevArray(go)
    cap = 3 * int(po[-1]) + 16
    out, ob, oe = DevArray(shape=(cap,), dtype=np.uint8), DevArray(shape=(n,), dtype=np.int64), DevArray(shape=(n,), dtype=np.int64)
    st = DevArray(shape=(n,), dtype=np.int32)
    _lib.check(_lib.lib().stag_backtranslate_dev(d_pt.ptr, int(po[-1]), d_po.ptr, d_gt.ptr, d_go.ptr, n, int(check_length),
                                                 out.ptr, cap, ob.ptr, oe.ptr, st.ptr, None))
    out, ob, oe, st = out.get(), ob.get(), oe.get(), st.get()
    assert np.array_equal(ob, 3 * po[:-1])
    return [None if st[i] else bytes(out[ob[i]:oe[i]]).decode("latin-1") for i in range(n)], st


def host_backtranslate(prot, gene, check_length=True):
    from stag_amd.align import protein2gene_alignment
    return protein2gene_alignment("x", prot, gene, check_length)


def test_backtranslate_dev_golden_and_random(capfd):
    g = gu.load("backtranslate.json")
    for check in (True, False):
        cases = [c for c in g if c["check_length"] == check]
        got, st = backtranslate_gpu([(c["protein"], c["gene"]) for c in cases], check)
        for c, r, s in zip(cases, got, st):
            # characters outside ASCII travel as stand-ins of their class (tables.ascii_stand_in)
            want = c["result"] if c["result"] is None else tables.ascii_stand_in(c["result"]).decode()
            assert r == want, c
            assert s == (_lib.SEQ_BACKTRANSLATE if c["result"] is None else 0)
    rng = np.random.default_rng(123)
    pairs = []
    for i in range(600):
        m = int(rng.integers(0, 400))
        prot = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY-acdefgh.*"), size=m,
                                  p=[0.03] * 20 + [0.2] + [0.02] * 7 + [0.03, 0.03]))
        res = sum(ch != "-" for ch in prot)
        glen = 3 * res + (3 if i % 3 == 0 else 0) + (int(rng.integers(-4, 5)) if i % 7 == 0 else 0)
        pairs.append((prot, "".join(rng.choice(list("ACGT"), size=max(glen, 0)))))
    for check in (True, False):
        got, _ = backtranslate_gpu(pairs, check)
        for (p, gseq), r in zip(pairs, got):
            assert r == host_backtranslate(p, gseq, check)
    capfd.readouterr()


def test_classify_protein_a2m_equals_host_backtranslation(capfd):
    """Protein path fused on the GPU == host back-translation (pinned to the reference) followed by
    the nucleotide path: same kernels downstream, so everything is bit-identical."""
    rng = np.random.default_rng(321)
    naa = 40
    ncol = 3 * naa
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[3], [2, 2, 3]], 0.1)
    prots, genes = [], []
    for i in range(300):
        cols = rng.choice(list("ACDEFGHIKL-"), size=naa, p=[0.08] * 10 + [0.2])
        s = list(cols)
        for _ in range(int(rng.integers(0, 3))):
            s.insert(int(rng.integers(0, len(s) + 1)), rng.choice(list("acd")))
        prot = "".join(s)
        res = sum(ch != "-" for ch in prot)
        glen = 3 * res + (3 if i % 2 else 0)
        if i % 23 == 0:
            glen += 1                                   # length check fails: the reference returns None
        prots.append(prot)
        genes.append("".join(rng.choice(list("ACGTN"), size=glen, p=[0.24, 0.24, 0.24, 0.24, 0.04])))
    pt, po = tables.pack_text(prots)
    gt, go = tables.pack_text(genes)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        res = db.classify_protein_a2m_text(pt, po, gt, go, min_perc_state=30.0, want_bits=True)
        nuc = [host_backtranslate(p, gseq) for p, gseq in zip(prots, genes)]
        bad = np.array([s is None for s in nuc])
        assert bad.sum() >= 10 and np.array_equal(res.status == _lib.SEQ_BACKTRANSLATE, bad)
        ref = db.classify_a2m([s or "" for s in nuc], min_perc_state=30.0, want_bits=True)
        ok = ~bad
        assert (res.status[ok] == 0).sum() > 100
        for f in ("status", "n_char", "n_aligned", "path_len", "selected", "n_lead_single"):
            assert np.array_equal(getattr(res, f)[ok], getattr(ref, f)[ok]), f
        assert np.array_equal(res.bits[ok], ref.bits[ok])
        assert np.array_equal(res.path_node[ok], ref.path_node[ok])
        assert np.array_equal(res.path_prob[ok], ref.path_prob[ok])
        assert np.array_equal(res.level_prob[ok], ref.level_prob[ok])
        with pytest.raises(TypeError):
            raise_for_status(res.status)
    capfd.readouterr()


def test_classify_concat_equals_host_concatenation():
    """classify_genome.py:364-388 on the GPU: concatenated rows == np.concatenate of the genes'
    rows (zeros for a missing gene), and their classification == classify_bits on those rows."""
    rng = np.random.default_rng(55)
    gene_len = [45, 100, 32, 5, 330, 64, 0, 95, 129]          # odd bit offsets, a 5-bit and an empty gene
    L = sum(gene_len)
    assert L % 5 == 0
    taxonomy, tax_function, classifiers = random_db(rng, L // 5, [[3], [2, 2, 2]], 0.15)
    n_genomes = 300
    gene_rows, dense = [], []
    for g in gene_len:
        m = int(rng.integers(0, 2 * n_genomes))
        X = rng.random((m, g)) < 0.3
        dense.append(X)
        gene_rows.append(tables.pack_bool_rows(X, tables.words_per_row(g)) if m else np.zeros((0, 4), dtype=np.uint32))
    src = np.full((n_genomes, len(gene_len)), -1, dtype=np.int32)
    for p, X in enumerate(dense):
        if len(X):
            have = rng.random(n_genomes) < 0.8
            src[have, p] = rng.integers(0, len(X), size=int(have.sum()))
    src[7] = -1                                               # a genome without any gene: all zeros
    want = np.zeros((n_genomes, L), dtype=bool)
    at = 0
    for p, g in enumerate(gene_len):
        rows = src[:, p]
        ok = rows >= 0
        want[ok, at:at + g] = dense[p][rows[ok]]
        at += g
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        res = db.classify_concat(gene_len, gene_rows, src, want_bits=True)
        assert np.array_equal(tables.unpack_rows(res.bits, L), want)
        assert not res.bits.view(np.uint8)[:, (L + 7) // 8:].any() or L % 8      # padding words stay zero
        ref = db.classify_onehot(want)
        for f in ("status", "n_aligned", "path_len", "selected", "n_lead_single", "path_node", "path_prob", "level_prob"):
            assert np.array_equal(getattr(res, f), getattr(ref, f)), f
        with pytest.raises(_lib.StagError):                   # lengths must add up to the database's L
            db.classify_concat(gene_len[:-1], gene_rows[:-1], src[:, :-1])
        bad = src.copy()
        bad[3, 0] = len(dense[0])
        with pytest.raises(_lib.StagError):
            db.classify_concat(gene_len, gene_rows, bad)


def caterpillar_db(rng, ncol, depth):
    """A tree `depth` levels deep with few nodes: every level has 2-3 children, only the first goes on."""
    L = 5 * ncol
    taxonomy, classifiers = {"tree_root": []}, {}
    parent, nid = "tree_root", 0
    for d in range(depth):
        k = 1 if d % 7 == 3 else int(rng.integers(2, 4))
        kids = []
        for _ in range(k):
            nid += 1
            kids.append("n%d" % nid)
        taxonomy[parent] = kids
        for kid in kids:
            taxonomy[kid] = []
            if k == 1:
                classifiers[kid] = "no_negative_examples"
            else:
                w = rng.normal(0, 1, size=1 + L) * (rng.random(1 + L) < 0.1)
                w[0] = rng.normal(0, 1) + (5.0 if kid == kids[0] else -2.0)   # most sequences go on down the spine
                classifiers[kid] = w
        parent = kids[0]
    tax_function = {str(l): rng.normal(0, 1, size=1 + depth) for l in range(-1, depth)}
    return taxonomy, tax_function, classifiers


def test_deep_tree_up_to_the_rank_cut_tile_limit():
    """31 levels with 32 rank-cut keys is what the rank-cut kernel's LDS tile holds (Limits);
    one level more is refused with an error, not computed wrongly."""
    rng = np.random.default_rng(31)
    ncol, n = 40, 700
    L = 5 * ncol
    X = np.zeros((n, L), dtype=bool)
    X[np.arange(n)[:, None], 5 * np.arange(ncol)[None, :] + rng.integers(0, 5, size=(n, ncol))] = True
    taxonomy, tax_function, classifiers = caterpillar_db(rng, ncol, 31)
    odb = oracle_c.DenseDB(taxonomy, tax_function, classifiers, L)
    out = odb.classify(X)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        assert db.max_depth == 31 and db.n_keys == 32
        res = db.classify_onehot(X)
        assert (res.path_len == 31).sum() > 20 and len(set(res.path_len.tolist())) > 3
        compare_with_oracle(db, odb, res, out, X, classifiers)
    taxonomy, tax_function, classifiers = caterpillar_db(rng, ncol, 32)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        with pytest.raises(_lib.StagError, match="rank-cut"):
            db.classify_onehot(X)


def test_encode_at_the_row_width_limit():
    """The encode kernel keeps one row per record in LDS: 65,000 columns (L = 325,000) still fit the
    160 KB, 70,000 do not and are refused (Limits)."""
    rng = np.random.default_rng(65)
    ncol = 65000
    strings = []
    for _ in range(6):
        s = np.full(ncol, "-", dtype="<U1")
        a = int(rng.integers(0, ncol - 5000))
        s[a:a + 5000] = rng.choice(list("ACGTUN"), size=5000)
        strings.append("".join(s))
    strings[2] = strings[2][:100] + "acgt" + strings[2][100:]
    bits, n_char, n_al, status = encode_raw(strings, 5 * ncol)
    dense = tables.unpack_rows(bits, bits.shape[1] * 32)
    for i, s in enumerate(strings):
        exp, e_char, e_al = oracle_c.encode_a2m(s)
        assert n_char[i] == e_char == ncol and n_al[i] == e_al and status[i] == _lib.SEQ_OK
        assert np.array_equal(dense[i, :5 * ncol], exp)
    with pytest.raises(_lib.StagError, match="too large"):
        encode_raw(["ACGT"], 5 * 70000)


def test_classify_concat_more_genomes_than_a_grid_dimension():
    """70,000 genomes in one call (more than the 65,535 a grid's y extent holds)."""
    rng = np.random.default_rng(70)
    gene_len = [10, 25, 5]
    L = sum(gene_len)
    taxonomy, tax_function, classifiers = random_db(rng, L // 5, [[3], [2, 2, 2]], 0.3)
    n_genomes = 70000
    dense = [rng.random((500, g)) < 0.3 for g in gene_len]
    gene_rows = [tables.pack_bool_rows(X, tables.words_per_row(g)) for X, g in zip(dense, gene_len)]
    src = rng.integers(-1, 500, size=(n_genomes, len(gene_len))).astype(np.int32)
    want = np.zeros((n_genomes, L), dtype=bool)
    at = 0
    for p, g in enumerate(gene_len):
        ok = src[:, p] >= 0
        want[ok, at:at + g] = dense[p][src[ok, p]]
        at += g
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        res = db.classify_concat(gene_len, gene_rows, src, want_bits=True)
        assert np.array_equal(tables.unpack_rows(res.bits, L), want)
        ref = db.classify_onehot(want)
        for f in ("status", "n_aligned", "path_len", "selected", "path_node", "path_prob", "level_prob"):
            assert np.array_equal(getattr(res, f), getattr(ref, f)), f


def test_fuzz_sample():
    """The first 120 configurations of tools/fuzz_gpu.py (random trees, widths, reads, filters, arbitrary
    bit rows, protein pairs, output text) -- the long runs are recorded in DESIGN.md."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "fuzz_gpu.py")
    spec = importlib.util.spec_from_file_location("fuzz_gpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for seed in range(120):
        mod.one(seed)
