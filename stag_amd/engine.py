"""Device-resident database + batch classification through the C ABI (include/stag_hip.h).

``Database`` is the packed, GPU-resident form of what ``load_db`` returns; ``classify_*``
run the HIP kernels on a batch and return plain NumPy arrays; ``result_rows`` turns them
into the reference's 7 output fields (stag/classify.py:116-122).
"""
import ctypes

import numpy as np

from . import _lib
from .tables import PackedTables, pack_bool_rows, pack_db, pack_text

# the exception the reference raises for each per-sequence status (include/stag_hip.h)
_EXC = {
    _lib.SEQ_MISSING_CLASSIFIER: (KeyError, "classifier missing for a sibling (stag/classify.py:40)"),
    _lib.SEQ_STRING_CLASSIFIER: (TypeError, "'no_negative_examples' classifier evaluated (stag/classify.py:24-25)"),
    _lib.SEQ_RANKCUT_SHAPE: (ValueError, "operands could not be broadcast together in the rank cut (stag/classify.py:25)"),
    _lib.SEQ_EMPTY_PATH: (IndexError, "list index out of range: empty path (stag/classify.py:106)"),
    _lib.SEQ_ZERO_COLUMNS: (ZeroDivisionError, "division by zero: no alignment column (stag/align.py:58)"),
    _lib.SEQ_BACKTRANSLATE: (TypeError, "'NoneType' object is not iterable: gene and protein alignment lengths do not "
                                        "fit (stag/align.py:66-74, then :37)"),
    _lib.SEQ_LENGTH_MISMATCH: (ValueError, "operands could not be broadcast together: alignment length differs "
                                          "from the database's (stag/classify.py:25)"),
}


def raise_for_status(status, ids=None):
    """Raise what the reference would have raised at the first failing sequence (it processes
    sequences in order and does not catch)."""
    bad = np.flatnonzero((status != _lib.SEQ_OK) & (status != _lib.SEQ_FILTERED))
    if bad.size:
        i = int(bad[0])
        exc, msg = _EXC[int(status[i])]
        raise exc("%s [sequence %s]" % (msg, ids[i] if ids is not None else i))


class BatchResult:
    def __init__(self, n, D, K):
        self.n, self.D, self.K = n, D, K
        self.path_node = np.full((n, D), -1, dtype=np.int32)
        self.path_prob = np.zeros((n, D), dtype=np.float64)
        self.path_len = np.zeros(n, dtype=np.int32)
        self.n_lead_single = np.zeros(n, dtype=np.int32)
        self.level_prob = np.zeros((n, max(K, 1)), dtype=np.float64)
        self.selected = np.full(n, -1, dtype=np.int32)
        self.n_aligned = np.zeros(n, dtype=np.int32)
        self.status = np.zeros(n, dtype=np.int32)
        self.bits = None
        self.n_char = None

    def as_struct(self):
        r = _lib.Results()
        for f, _ in _lib.Results._fields_:
            setattr(r, f, _lib.ptr(getattr(self, f)))
        return r


class Database:
    """A single-gene stagDB resident on one GPU."""

    def __init__(self, tables, device=0):
        if not isinstance(tables, PackedTables):
            raise TypeError("tables must be PackedTables (see stag_amd.tables.pack_db)")
        self.tables = tables
        self.device = int(device)
        self._h = ctypes.c_void_p()
        st = tables.as_struct()
        _lib.check(_lib.lib().stag_db_create(ctypes.byref(st), self.device, ctypes.byref(self._h)))
        d, w, k, tb = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _lib.check(_lib.lib().stag_db_info(self._h, ctypes.byref(d), ctypes.byref(w), ctypes.byref(k), ctypes.byref(tb)))
        self.max_depth, self.words_per_row, self.n_keys, self.table_bytes = d.value, w.value, k.value, tb.value
        self.names = tables.names
        self.keys = tables.keys
        self.n_features = tables.n_features
        self._lineage_cache = {}

    @classmethod
    def from_dicts(cls, taxonomy, tax_function, classifiers, n_features=None, device=0):
        return cls(pack_db(taxonomy, tax_function, classifiers, n_features), device)

    @property
    def handle(self):
        if not self._h:
            raise _lib.StagError("database handle already closed")
        return self._h

    def close(self):
        if self._h:
            _lib.lib().stag_db_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------ host-buffer calls
    def encode_a2m(self, strings, min_perc_state=0.0):
        """align.py:37-58 for a list of aligned strings -> (bits[n,W], n_char, n_aligned, status)."""
        text, offs = pack_text(strings)
        n = len(strings)
        bits = np.zeros((n, self.words_per_row), dtype=np.uint32)
        n_char = np.zeros(n, dtype=np.int32)
        n_al = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        _lib.check(_lib.lib().stag_encode_a2m(self.device, self.n_features, _lib.ptr(text), _lib.ptr(offs), n,
                                              float(min_perc_state), _lib.ptr(bits), _lib.ptr(n_char),
                                              _lib.ptr(n_al), _lib.ptr(status)))
        return bits, n_char, n_al, status

    def classify_bits(self, bits, status_in=None):
        """classify_seq (classify.py:94-124) for packed rows uint32[n, words_per_row]."""
        bits = np.ascontiguousarray(bits, dtype=np.uint32)
        if bits.ndim != 2 or bits.shape[1] != self.words_per_row:
            raise ValueError("bits must be [n, %d] uint32" % self.words_per_row)
        n = bits.shape[0]
        res = BatchResult(n, self.max_depth, self.n_keys)
        st = res.as_struct()
        sin = None if status_in is None else np.ascontiguousarray(status_in, dtype=np.int32)
        _lib.check(_lib.lib().stag_classify_bits(self.handle, _lib.ptr(bits), n, _lib.ptr(sin), ctypes.byref(st)))
        res.level_prob = res.level_prob[:, :self.n_keys]
        return res

    def classify_onehot_text(self, text, body_begin, body_end, slow_line=None, raw=None):
        """alignment_reader + classify_seq (classify.py:16-20, 94-124) for lines of a -s file found by
        stag_amd.ingest.onehot_text_batches: the GPU parses the fields.  Lines it flags (fields other than the
        single characters 0 / 1) go through slow_line(raw(i)) -> (id, bool array), the reference's int() per field.
        Returns (result, n_fields)."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        bb = np.ascontiguousarray(body_begin, dtype=np.int64)
        be = np.ascontiguousarray(body_end, dtype=np.int64)
        n = len(bb)
        n_fields = np.zeros(n, dtype=np.int32)
        flag = np.zeros(n, dtype=np.int32)
        _lib.check(_lib.lib().stag_onehot_text_parse(self.handle, _lib.ptr(text), len(text), _lib.ptr(bb), _lib.ptr(be), n,
                                                     _lib.ptr(n_fields), _lib.ptr(flag)))
        slow = np.flatnonzero(flag)
        patch_rows = np.zeros((len(slow), self.words_per_row), dtype=np.uint32)
        for j, i in enumerate(slow):
            if slow_line is None or raw is None:
                raise ValueError("line %d needs the slow path and none was given" % int(i))
            _, x = slow_line(raw(int(i)))
            n_fields[i] = len(x)
            if len(x) == self.n_features:
                patch_rows[j] = pack_bool_rows(np.asarray(x, dtype=bool)[None, :], self.words_per_row)[0]
        status = np.where(n_fields == self.n_features, _lib.SEQ_OK, _lib.SEQ_LENGTH_MISMATCH).astype(np.int32)
        res = BatchResult(n, self.max_depth, self.n_keys)
        st = res.as_struct()
        idx = np.ascontiguousarray(slow, dtype=np.int64)
        _lib.check(_lib.lib().stag_onehot_text_classify(self.handle, n, _lib.ptr(idx), _lib.ptr(patch_rows), len(slow),
                                                        _lib.ptr(status), ctypes.byref(st)))
        res.level_prob = res.level_prob[:, :self.n_keys]
        return res, n_fields

    def classify_onehot(self, X):
        """X: bool [n, L] exactly as the reference holds it."""
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("X must be [n, L]")
        if X.shape[1] != self.n_features:
            status = np.full(X.shape[0], _lib.SEQ_LENGTH_MISMATCH, dtype=np.int32)
            bits = np.zeros((X.shape[0], self.words_per_row), dtype=np.uint32)
            res = self.classify_bits(bits, status)
            ncols = (X.shape[1] + 4) // 5
            res.n_aligned = (ncols - X[:, 0::5].sum(axis=1)).astype(np.int32)
            return res
        return self.classify_bits(pack_bool_rows(X, self.words_per_row))

    def classify_a2m(self, strings, min_perc_state=0.0, want_bits=False):
        """encode + classify on the GPU; rows never leave it unless want_bits."""
        text, offs = pack_text(strings)
        return self.classify_a2m_text(text, offs, min_perc_state, want_bits)

    def classify_a2m_text(self, text, offs, min_perc_state=0.0, want_bits=False, lead_trail=None):
        """Same for records already laid out back to back: record i = text[offs[i]:offs[i+1]]
        (uint8, int64[n+1]); what stag_amd.ingest.fasta_batches produces.  lead_trail (int32 [n, 2], optional):
        the '-' columns dropped from either end of each record (fasta_batches(trim=True))."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
        n = len(offs) - 1
        if n < 0 or (n > 0 and (offs[0] < 0 or offs[-1] > len(text))):
            raise ValueError("offs does not describe records inside text")
        if lead_trail is not None:
            lead_trail = np.ascontiguousarray(lead_trail, dtype=np.int32)
            if lead_trail.size != 2 * n:
                raise ValueError("lead_trail must hold two counts per record")
        if len(text) == 0:
            text = np.zeros(16, dtype=np.uint8)
        res = BatchResult(n, self.max_depth, self.n_keys)
        st = res.as_struct()
        res.n_char = np.zeros(n, dtype=np.int32)
        if want_bits:
            res.bits = np.zeros((n, self.words_per_row), dtype=np.uint32)
        _lib.check(_lib.lib().stag_classify_a2m_trimmed(self.handle, _lib.ptr(text), _lib.ptr(offs), _lib.ptr(lead_trail), n,
                                                        float(min_perc_state), ctypes.byref(st), _lib.ptr(res.bits),
                                                        _lib.ptr(res.n_char)))
        res.level_prob = res.level_prob[:, :self.n_keys]
        return res

    def classify_protein_a2m_text(self, prot_text, prot_offs, gene_text, gene_offs, min_perc_state=0.0, want_bits=False):
        """Protein database: (protein alignment, gene) pairs -> back-translate (align.py:63-94),
        encode and classify on the GPU.  Record i = text[offs[i]:offs[i+1]] in both pairs of arrays."""
        pt = np.ascontiguousarray(prot_text, dtype=np.uint8)
        po = np.ascontiguousarray(prot_offs, dtype=np.int64)
        gt = np.ascontiguousarray(gene_text, dtype=np.uint8)
        go = np.ascontiguousarray(gene_offs, dtype=np.int64)
        n = len(po) - 1
        if n < 0 or len(go) != n + 1 or (n > 0 and (po[-1] > len(pt) or go[-1] > len(gt))):
            raise ValueError("offsets do not describe n records inside their texts")
        if len(pt) == 0:
            pt = np.zeros(16, dtype=np.uint8)
        if len(gt) == 0:
            gt = np.zeros(16, dtype=np.uint8)
        res = BatchResult(n, self.max_depth, self.n_keys)
        st = res.as_struct()
        res.n_char = np.zeros(n, dtype=np.int32)
        if want_bits:
            res.bits = np.zeros((n, self.words_per_row), dtype=np.uint32)
        _lib.check(_lib.lib().stag_classify_protein_a2m(self.handle, _lib.ptr(pt), _lib.ptr(po), _lib.ptr(gt), _lib.ptr(go),
                                                        n, float(min_perc_state), ctypes.byref(st), _lib.ptr(res.bits),
                                                        _lib.ptr(res.n_char)))
        res.level_prob = res.level_prob[:, :self.n_keys]
        return res

    def classify_concat(self, gene_len, gene_rows, src_row, want_bits=False):
        """classify_genome.py:364-388 + :494 on the GPU: gene_len[p] features and packed rows
        gene_rows[p] (uint32 [n_p, W_p]) per marker gene, src_row int32 [n_genomes, n_genes] = the
        row of gene p that belongs to genome j or -1 (zero-filled).  Returns the BatchResult of the
        concatenated rows against this (concatenated) database."""
        n_genes = len(gene_len)
        src = np.ascontiguousarray(src_row, dtype=np.int32).reshape(-1, n_genes)
        n = src.shape[0]
        rows = [np.ascontiguousarray(r, dtype=np.uint32).reshape(len(r), -1) if len(r) else
                np.zeros((0, 4), dtype=np.uint32) for r in gene_rows]
        glen = np.ascontiguousarray(gene_len, dtype=np.int32)
        wpr = np.array([r.shape[1] for r in rows], dtype=np.int32)
        nrows = np.array([r.shape[0] for r in rows], dtype=np.int64)
        ptrs = (ctypes.c_void_p * n_genes)(*[r.ctypes.data if r.size else None for r in rows])
        res = BatchResult(n, self.max_depth, self.n_keys)
        st = res.as_struct()
        if want_bits:
            res.bits = np.zeros((n, self.words_per_row), dtype=np.uint32)
        _lib.check(_lib.lib().stag_classify_concat(self.handle, n_genes, _lib.ptr(glen), ptrs, _lib.ptr(wpr), _lib.ptr(nrows),
                                                   _lib.ptr(src), n, ctypes.byref(st), _lib.ptr(res.bits)))
        res.level_prob = res.level_prob[:, :self.n_keys]
        return res

    # ------------------------------------------------------------------ profiling (bench.py)
    def profile_enable(self, on=True):
        _lib.check(_lib.lib().stag_profile_enable(self.handle, int(bool(on))))

    def profile_read(self):
        ms = np.zeros(len(_lib.PROF_CLASSES), dtype=np.float64)
        cnt = np.zeros(len(_lib.PROF_CLASSES), dtype=np.int64)
        _lib.check(_lib.lib().stag_profile_read(self.handle, _lib.ptr(ms), _lib.ptr(cnt)))
        return {k: (float(m), int(c)) for k, m, c in zip(_lib.PROF_CLASSES, ms, cnt)}

    def reserve(self, n, text_bytes=0):
        _lib.check(_lib.lib().stag_db_reserve(self.handle, int(n), int(text_bytes)))

    # ------------------------------------------------------------------ output fields
    def lineage(self, res, i):
        """(tax list) for sequence i; cached per end position because a position fixes its path."""
        n = int(res.path_len[i])
        if n == 0:
            return []
        key = int(res.path_node[i, n - 1])
        tax = self._lineage_cache.get(key)
        if tax is None:
            tax = [self.names[j] for j in res.path_node[i, :n]]
            self._lineage_cache[key] = tax
        return tax

    def result_row(self, res, i, gene_id, long_out=True):
        """The reference's output fields for sequence i (classify.py:116-122)."""
        tax = self.lineage(res, i)
        sel = int(res.selected[i])
        selected = self.keys[sel] if sel >= 0 else -1        # int -1 when nothing was selected (:61)
        assigned = ";".join(tax[0:int(selected) + 1])
        if not long_out:
            return [gene_id, assigned]
        n = len(tax)
        lead = int(res.n_lead_single[i])
        perc = ["1"] * min(lead, n) + [repr(float(p)) for p in res.path_prob[i, lead:n]]
        per_level = [k + ":" + repr(float(p)) for k, p in zip(self.keys, res.level_prob[i])]
        return [gene_id, assigned, "/".join(tax), selected, "/".join(perc), "/".join(per_level),
                str(int(res.n_aligned[i]))]

    def result_rows(self, res, ids, long_out=True):
        """Rows for every sequence the reference would have yielded (status FILTERED is skipped);
        raises the reference's exception for the first failing sequence."""
        raise_for_status(res.status, ids)
        ok = np.flatnonzero(res.status == _lib.SEQ_OK).tolist()
        D, K = res.path_prob.shape[1], len(self.keys)
        path_len, sel_all = res.path_len.tolist(), res.selected.tolist()
        last = np.maximum(res.path_len.astype(np.int64) - 1, 0)
        end_all = res.path_node[np.arange(res.n), last].tolist() if D else [-1] * res.n
        if not long_out:
            assigned = {}     # (end position, selected) -> the assigned lineage: a position fixes its path
            rows = []
            for i in ok:
                sel = sel_all[i]
                key = (end_all[i] if path_len[i] else -1, sel)
                text = assigned.get(key)
                if text is None:
                    selected = self.keys[sel] if sel >= 0 else -1
                    text = assigned[key] = ";".join(self.lineage(res, i)[0:int(selected) + 1])
                rows.append([ids[i], text])
            return rows
        # the long fields of a whole batch: every float goes through repr() in one pass over flat lists (the same
        # strings as result_row's repr(float(p)); a row then only slices and joins)
        perc_all = list(map(repr, res.path_prob.ravel().tolist()))
        level_all = list(map(repr, res.level_prob[:, :K].ravel().tolist()))
        prefix = [k + ":" for k in self.keys]
        n_aligned = list(map(str, res.n_aligned.tolist()))
        lead_all = res.n_lead_single.tolist()
        rows = []
        joined = {}       # (end position, selected) -> the two lineage strings: a position fixes its path
        for i in ok:
            n, lead, sel = path_len[i], lead_all[i], sel_all[i]
            selected = self.keys[sel] if sel >= 0 else -1
            key = (end_all[i] if n else -1, sel)
            both = joined.get(key)
            if both is None:
                tax = self.lineage(res, i)
                both = joined[key] = (";".join(tax[0:int(selected) + 1]), "/".join(tax))
            perc = ["1"] * min(lead, n) + perc_all[i * D + lead:i * D + n]
            per_level = map(str.__add__, prefix, level_all[i * K:(i + 1) * K])
            rows.append([ids[i], both[0], both[1], selected, "/".join(perc), "/".join(per_level), n_aligned[i]])
        return rows
