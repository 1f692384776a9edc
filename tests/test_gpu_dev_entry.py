"""The *_dev entry points of include/stag_hip.h (device buffers in, device results out, the
caller's stream) against the host-buffer entry points on the same inputs: bit-identical, since
both run the same kernels.  And stag_classify_a2m_dev recorded into a HIP graph by stream capture
and replayed: after stag_db_reserve the call only enqueues work on the stream it is given.
Device memory comes from tests/hipmem.py (ctypes on the HIP runtime libstag_hip.so loaded)."""
import ctypes

import numpy as np
import pytest

from hipmem import DevArray, hip
from stag_amd import _lib, tables
from stag_amd.engine import Database
from test_gpu_parity import random_db

pytestmark = pytest.mark.gpu

FIELDS = ("path_node", "path_prob", "path_len", "n_lead_single", "level_prob", "selected", "n_aligned", "status")


def device_results(n, D, K):
    arrs = {
        "path_node": DevArray(shape=(n, D), dtype=np.int32), "path_prob": DevArray(shape=(n, D), dtype=np.float64),
        "path_len": DevArray(shape=(n,), dtype=np.int32), "n_lead_single": DevArray(shape=(n,), dtype=np.int32),
        "level_prob": DevArray(shape=(n, K), dtype=np.float64), "selected": DevArray(shape=(n,), dtype=np.int32),
        "n_aligned": DevArray(shape=(n,), dtype=np.int32), "status": DevArray(shape=(n,), dtype=np.int32),
    }
    res = _lib.Results()
    for f, _ in _lib.Results._fields_:
        setattr(res, f, arrs[f].ptr.value)
    return arrs, res


def same_as_host(arrs, ref, n, rows=None):
    for f in FIELDS:
        got = arrs[f].get()
        want = getattr(ref, f)[:n].reshape(got.shape)
        if rows is not None:
            got, want = got[rows], want[rows]
        assert np.array_equal(got, want), f


def amplicon_strings(rng, ncol, n):
    out = []
    for i in range(n):
        cols = rng.choice(list("ACGT-N"), size=ncol, p=[0.2, 0.2, 0.2, 0.2, 0.15, 0.05])
        if i % 4 == 0:
            cols[: int(rng.integers(10, ncol))] = "-"
        s = list(cols)
        for _ in range(int(rng.integers(0, 3))):
            s.insert(int(rng.integers(0, len(s) + 1)), rng.choice(list("acgt")))
        out.append("".join(s))
    out[3] = "acgt"            # no columns
    out[5] = out[5][:-2]       # wrong number of columns
    return out


def padded(text):
    return np.concatenate([text, np.zeros(-len(text) % 16, dtype=np.uint8)])


def test_encode_dev_equals_host_entry():
    rng = np.random.default_rng(1)
    ncol = 150
    L = 5 * ncol
    strings = amplicon_strings(rng, ncol, 333)
    text, offs = tables.pack_text(strings)
    n = len(strings)
    W = _lib.lib().stag_words_per_row(L)
    bits = np.zeros((n, W), dtype=np.uint32)
    n_char, n_al, status = (np.zeros(n, dtype=np.int32) for _ in range(3))
    _lib.check(_lib.lib().stag_encode_a2m(0, L, _lib.ptr(text), _lib.ptr(offs), n, 35.0, _lib.ptr(bits), _lib.ptr(n_char),
                                          _lib.ptr(n_al), _lib.ptr(status)))
    t16 = padded(text)
    d_text, d_offs = DevArray(t16), DevArray(offs)
    d_bits = DevArray(shape=(n, W), dtype=np.uint32)
    d_char, d_al, d_st = (DevArray(shape=(n,), dtype=np.int32) for _ in range(3))
    _lib.check(_lib.lib().stag_encode_a2m_dev(L, d_text.ptr, len(t16), d_offs.ptr, n, 35.0, d_bits.ptr, d_char.ptr,
                                              d_al.ptr, d_st.ptr, None))
    assert np.array_equal(d_bits.get(), bits)
    assert np.array_equal(d_char.get(), n_char) and np.array_equal(d_al.get(), n_al) and np.array_equal(d_st.get(), status)
    assert len(set(status.tolist())) >= 3
    # a text buffer that is not a multiple of 16 bytes is refused, not read past its end
    if len(text) % 16:
        assert _lib.lib().stag_encode_a2m_dev(L, d_text.ptr, len(text), d_offs.ptr, n, 0.0, d_bits.ptr, d_char.ptr, d_al.ptr,
                                              d_st.ptr, None) != 0


def test_classify_bits_dev_and_a2m_dev_equal_host_entries():
    rng = np.random.default_rng(2)
    ncol = 150
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [1, 2, 5], [0, 2, 3]], 0.08)
    strings = amplicon_strings(rng, ncol, 500)
    text, offs = tables.pack_text(strings)
    n = len(strings)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        ref = db.classify_a2m(strings, min_perc_state=20.0, want_bits=True)
        D, K = db.max_depth, db.n_keys
        # fused: text -> results
        t16 = padded(text)
        d_text, d_offs = DevArray(t16), DevArray(offs)
        arrs, res = device_results(n, D, K)
        db.reserve(n, len(t16))
        _lib.check(_lib.lib().stag_classify_a2m_dev(db.handle, d_text.ptr, len(t16), d_offs.ptr, n, 20.0,
                                                    ctypes.byref(res), None))
        same_as_host(arrs, ref, n)
        # rows -> results, with encode's status passed along (the statuses the walk itself gave --
        # a node without classifier -- are for it to find again)
        from_encode = np.isin(ref.status, (_lib.SEQ_ZERO_COLUMNS, _lib.SEQ_FILTERED, _lib.SEQ_LENGTH_MISMATCH))
        assert from_encode.sum() >= 3 and (~from_encode & (ref.status != 0)).sum() >= 3
        d_bits, d_st = DevArray(ref.bits), DevArray(np.where(from_encode, ref.status, 0).astype(np.int32))
        arrs2, res2 = device_results(n, D, K)
        _lib.check(_lib.lib().stag_classify_bits_dev(db.handle, d_bits.ptr, n, d_st.ptr, ctypes.byref(res2), None))
        same_as_host(arrs2, ref, n, rows=~from_encode)      # n_aligned of a skipped row is recounted from its bits
        assert np.array_equal(arrs2["status"].get(), ref.status)
        assert (arrs2["path_len"].get()[from_encode] == 0).all()
        # outputs the caller does not want may be NULL (status excepted)
        arrs3, res3 = device_results(n, D, K)
        res3.path_prob = None
        res3.level_prob = None
        _lib.check(_lib.lib().stag_classify_a2m_dev(db.handle, d_text.ptr, len(t16), d_offs.ptr, n, 20.0,
                                                    ctypes.byref(res3), None))
        assert np.array_equal(arrs3["path_node"].get(), ref.path_node)
        assert np.array_equal(arrs3["selected"].get(), ref.selected)
        assert not arrs3["path_prob"].get().any()


def test_classify_protein_a2m_dev_equals_host_entry(capfd):
    rng = np.random.default_rng(3)
    naa = 30
    L = 5 * 3 * naa
    taxonomy, tax_function, classifiers = random_db(rng, 3 * naa, [[3], [2, 3, 3]], 0.1)
    prots, genes = [], []
    for i in range(200):
        prot = "".join(rng.choice(list("ACDEFGHIKL-"), size=naa, p=[0.08] * 10 + [0.2]))
        glen = 3 * sum(ch != "-" for ch in prot) + (3 if i % 2 else 0) + (1 if i % 19 == 0 else 0)
        prots.append(prot)
        genes.append("".join(rng.choice(list("ACGT"), size=glen)))
    pt, po = tables.pack_text(prots)
    gt, go = tables.pack_text(genes)
    n = len(prots)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        ref = db.classify_protein_a2m_text(pt, po, gt, go, min_perc_state=10.0)
        p16, g16 = padded(pt), padded(gt)
        d = [DevArray(a) for a in (p16, po, g16, go)]
        arrs, res = device_results(n, db.max_depth, db.n_keys)
        _lib.check(_lib.lib().stag_classify_protein_a2m_dev(db.handle, d[0].ptr, len(p16), d[1].ptr, d[2].ptr, d[3].ptr, n,
                                                            10.0, ctypes.byref(res), None))
        same_as_host(arrs, ref, n)
        assert (ref.status == _lib.SEQ_BACKTRANSLATE).sum() >= 5
    capfd.readouterr()


def test_classify_a2m_dev_under_stream_capture():
    """One call recorded into a HIP graph, then replayed twice on fresh outputs."""
    H = hip()
    vp = ctypes.c_void_p
    H.hipStreamCreate.argtypes = [ctypes.POINTER(vp)]
    H.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
    H.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
    H.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_size_t]
    H.hipGraphLaunch.argtypes = [vp, vp]
    H.hipStreamSynchronize.argtypes = [vp]
    H.hipGraphExecDestroy.argtypes = [vp]
    H.hipGraphDestroy.argtypes = [vp]
    H.hipStreamDestroy.argtypes = [vp]

    rng = np.random.default_rng(4)
    ncol = 200
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[5], [2, 4], [1, 3]], 0.05)
    strings = amplicon_strings(rng, ncol, 2000)
    text, offs = tables.pack_text(strings)
    n = len(strings)
    t16 = padded(text)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        ref = db.classify_a2m(strings)
        d_text, d_offs = DevArray(t16), DevArray(offs)
        arrs, res = device_results(n, db.max_depth, db.n_keys)
        db.reserve(n, len(t16))     # no allocation may happen while the stream is being captured
        st, graph, gexec = vp(), vp(), vp()
        assert H.hipStreamCreate(ctypes.byref(st)) == 0
        try:
            assert H.hipStreamBeginCapture(st, 0) == 0
            rc = _lib.lib().stag_classify_a2m_dev(db.handle, d_text.ptr, len(t16), d_offs.ptr, n, 0.0, ctypes.byref(res), st)
            assert H.hipStreamEndCapture(st, ctypes.byref(graph)) == 0
            _lib.check(rc)
            assert graph.value
            assert H.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
            assert not arrs["path_len"].get().any()          # capture recorded the work, it did not run it
            for _ in range(2):
                for a in arrs.values():
                    assert H.hipMemset(a.ptr, 0, max(a.nbytes, 16)) == 0
                assert H.hipGraphLaunch(gexec, st) == 0
                assert H.hipStreamSynchronize(st) == 0
                same_as_host(arrs, ref, n)
        finally:
            if gexec.value:
                H.hipGraphExecDestroy(gexec)
            if graph.value:
                H.hipGraphDestroy(graph)
            H.hipStreamDestroy(st)


def test_two_handles_on_two_streams_are_independent():
    """Two handles of the same database, each with its own stream and batch, enqueued back to back
    without a synchronisation in between: same results as one after the other."""
    H = hip()
    vp = ctypes.c_void_p
    H.hipStreamCreate.argtypes = [ctypes.POINTER(vp)]
    H.hipStreamSynchronize.argtypes = [vp]
    H.hipStreamDestroy.argtypes = [vp]
    rng = np.random.default_rng(6)
    ncol = 180
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [2, 3], [1, 2, 4]], 0.06)
    batches = [amplicon_strings(rng, ncol, 3000), amplicon_strings(rng, ncol, 1700)]
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db_a, \
            Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db_b:
        refs = [db_a.classify_a2m(b) for b in batches]
        streams, work = [], []
        try:
            for db, strings in zip((db_a, db_b), batches):
                text, offs = tables.pack_text(strings)
                t16 = padded(text)
                st = vp()
                assert H.hipStreamCreate(ctypes.byref(st)) == 0
                streams.append(st)
                arrs, res = device_results(len(strings), db.max_depth, db.n_keys)
                db.reserve(len(strings), len(t16))
                work.append((db, DevArray(t16), len(t16), DevArray(offs), len(strings), arrs, res, st))
            for _ in range(3):      # several rounds in flight on both streams
                for db, d_text, tb, d_offs, n, arrs, res, st in work:
                    _lib.check(_lib.lib().stag_classify_a2m_dev(db.handle, d_text.ptr, tb, d_offs.ptr, n, 0.0,
                                                                ctypes.byref(res), st))
            for st in streams:
                assert H.hipStreamSynchronize(st) == 0
            for (db, d_text, tb, d_offs, n, arrs, res, st), ref in zip(work, refs):
                same_as_host(arrs, ref, n)
        finally:
            for st in streams:
                H.hipStreamDestroy(st)


def test_host_pipeline_sub_batches_pinned_and_pageable(monkeypatch):
    """stag_classify_a2m is a pipeline over sub-batches (text up / kernels / results down on three streams).
    Forced to 97-record sub-batches it must give what one sub-batch gives, bit for bit, with pageable and with
    page-locked (stag_host_alloc) buffers, with and without the packed rows coming back; and a malformed
    offsets array is refused (its sub-batch is validated before its upload is queued)."""
    rng = np.random.default_rng(8)
    ncol = 210
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [1, 2, 5], [0, 2, 3], [2]], 0.07)
    strings = amplicon_strings(rng, ncol, 3001)
    text, offs = tables.pack_text(strings)
    n = len(strings)
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        monkeypatch.setenv("STAG_PIPE_RECORDS", "1000000")
        ref = db.classify_a2m_text(text, offs, min_perc_state=15.0, want_bits=True)
        assert len(set(ref.status.tolist())) >= 3
        monkeypatch.setenv("STAG_PIPE_RECORDS", "97")
        for want_bits in (True, False):
            got = db.classify_a2m_text(text, offs, min_perc_state=15.0, want_bits=want_bits)
            for f in FIELDS + ("n_char",) + (("bits",) if want_bits else ()):
                assert np.array_equal(getattr(got, f), getattr(ref, f)), f
        # page-locked buffers for everything the call touches
        D, K, W = db.max_depth, db.n_keys, db.words_per_row
        p_text, p_offs = _lib.PinnedArray(len(text), np.uint8), _lib.PinnedArray(n + 1, np.int64)
        p_text.a[:] = text
        p_offs.a[:] = offs
        shapes = {"path_node": ((n, D), np.int32), "path_prob": ((n, D), np.float64), "path_len": (n, np.int32),
                  "n_lead_single": (n, np.int32), "level_prob": ((n, K), np.float64), "selected": (n, np.int32),
                  "n_aligned": (n, np.int32), "status": (n, np.int32)}
        pinned = {f: _lib.PinnedArray(*sd) for f, sd in shapes.items()}
        p_bits, p_char = _lib.PinnedArray((n, W), np.uint32), _lib.PinnedArray(n, np.int32)
        res = _lib.Results()
        for f in shapes:
            setattr(res, f, _lib.ptr(pinned[f].a))
        for _ in range(2):
            for arr in list(pinned.values()) + [p_bits, p_char]:
                arr.a[...] = 0
            _lib.check(_lib.lib().stag_classify_a2m(db.handle, _lib.ptr(p_text.a), _lib.ptr(p_offs.a), n, 15.0,
                                                    ctypes.byref(res), _lib.ptr(p_bits.a), _lib.ptr(p_char.a)))
            for f in shapes:
                assert np.array_equal(pinned[f].a, getattr(ref, f).reshape(pinned[f].a.shape)), f
            assert np.array_equal(p_bits.a, ref.bits) and np.array_equal(p_char.a, ref.n_char)
        # offsets that step backwards in the middle (they would send the kernel outside the text)
        bad = offs.copy()
        bad[1500] = bad[1499] - 5
        st = ref.as_struct()
        rc = _lib.lib().stag_classify_a2m(db.handle, _lib.ptr(text), _lib.ptr(bad), n, 0.0, ctypes.byref(st), None, None)
        assert rc == -1 and b"non-decreasing" in _lib.lib().stag_last_error()
        bad = offs.copy()
        bad[0] = -16
        assert _lib.lib().stag_classify_a2m(db.handle, _lib.ptr(text), _lib.ptr(bad), n, 0.0, ctypes.byref(st), None, None) == -1


def test_trimmed_text_entry_equals_the_full_text_entry():
    """stag_classify_a2m_trimmed: records without their flanking '-' runs + the two counts give, field for field and
    bit for bit, what stag_classify_a2m gives for the whole strings (align.py:37-58: a '-' column is the row's
    default) -- packed rows, n_char and the filter included; also across sub-batches of the host pipeline."""
    import io
    import os
    from stag_amd import ingest
    rng = np.random.default_rng(21)
    ncol = 420
    L = 5 * ncol
    from test_gpu_parity import random_db
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [2, 5], [1, 3]], 0.1)
    strings = []
    for i in range(700):
        s = np.full(ncol, "-", dtype="<U1")
        a, b = sorted(rng.integers(0, ncol + 1, size=2))
        if i % 9:
            s[a:b] = rng.choice(list("ACGT-N"), size=b - a, p=[0.22, 0.22, 0.22, 0.22, 0.08, 0.04])
        s = list(s)
        for _ in range(int(rng.integers(0, 3))):
            s.insert(int(rng.integers(0, len(s) + 1)), str(rng.choice(list("acgt"))))
        if i % 50 == 3:
            s = s[:-2]                       # wrong length
        strings.append("".join(s))
    strings[5] = "-" * ncol
    strings[6] = "acgt"
    stream = "".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(strings)).encode()
    (ids, text, offs, lt), = list(ingest.fasta_batches(io.BytesIO(stream), trim=True))
    assert int(offs[-1]) < sum(map(len, strings)) // 2
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        for pipe in (None, "97"):
            if pipe:
                os.environ["STAG_PIPE_RECORDS"] = pipe
            try:
                for want_bits in (True, False):
                    full = db.classify_a2m(strings, min_perc_state=20.0, want_bits=want_bits)
                    trimmed = db.classify_a2m_text(text, offs, min_perc_state=20.0, want_bits=want_bits, lead_trail=lt)
                    for f in ("status", "n_char", "n_aligned", "path_len", "path_node", "path_prob", "selected",
                              "level_prob", "n_lead_single"):
                        assert np.array_equal(getattr(trimmed, f), getattr(full, f)), (f, pipe, want_bits)
                    if want_bits:
                        assert np.array_equal(trimmed.bits, full.bits)
            finally:
                os.environ.pop("STAG_PIPE_RECORDS", None)
        assert set(np.unique(full.status)) >= {0, 5, 6, 7}
