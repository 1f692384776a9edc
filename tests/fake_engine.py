"""CPU stand-in for stag_amd.engine.Database, used ONLY by the CPU tests of the host logic
(file formats, output text, rank sharding): same interface, arithmetic done by the C oracle.
It reuses the product's row formatting (Database.result_row) so that text paths are exercised."""
import numpy as np

from oracle import oracle_c
from stag_amd import _lib
from stag_amd.engine import BatchResult, Database
from stag_amd.tables import unpack_rows


class OracleDatabase(Database):
    def __init__(self, tables, device=0):
        self.tables, self.device = tables, device
        self._o = oracle_c.DenseDB.from_packed(tables)
        self.max_depth, self.n_keys = self._o.max_depth, len(tables.keys)
        self.words_per_row = max(((tables.n_features + 31) // 32 + 3) // 4 * 4, 4)
        self.names, self.keys, self.n_features = tables.names, tables.keys, tables.n_features
        self._lineage_cache = {}
        self._h = None

    classify_onehot_text = None     # no device parser here: classify() then reads -s files with ingest.onehot_batches

    def close(self):
        pass

    def classify_bits(self, bits, status_in=None):
        X = unpack_rows(np.ascontiguousarray(bits, dtype=np.uint32), self.n_features)
        n = X.shape[0]
        out = self._o.classify(X)
        res = BatchResult(n, self.max_depth, self.n_keys)
        res.path_node, res.path_prob, res.path_len = out["path_node"], out["path_prob"], out["path_len"]
        res.level_prob, res.selected, res.n_aligned = out["level_prob"], out["selected"], out["n_aligned"]
        res.status = out["status"].copy()
        lead = np.zeros(n, dtype=np.int32)
        for i in range(n):
            while lead[i] < res.path_len[i] and (out["single_mask"][i] >> lead[i]) & 1:
                lead[i] += 1
        res.n_lead_single = lead
        if status_in is not None:
            bad = np.asarray(status_in) != 0
            res.status[bad] = np.asarray(status_in)[bad]
        return res

    def classify_a2m_text(self, text, offs, min_perc_state=0.0, want_bits=False, lead_trail=None):
        strings = [bytes(text[offs[i]:offs[i + 1]]).decode("latin-1") for i in range(len(offs) - 1)]
        if lead_trail is not None:      # the records came without their flanking gap columns
            strings = ["-" * int(a) + s + "-" * int(b) for s, (a, b) in zip(strings, np.asarray(lead_trail).reshape(-1, 2))]
        return self.classify_a2m(strings, min_perc_state, want_bits)

    def classify_concat(self, gene_len, gene_rows, src_row, want_bits=False):
        from stag_amd.tables import pack_bool_rows
        src = np.asarray(src_row).reshape(-1, len(gene_len))
        X = np.zeros((src.shape[0], self.n_features), dtype=bool)
        at = 0
        for p, g in enumerate(gene_len):
            ok = src[:, p] >= 0
            if ok.any():
                X[ok, at:at + g] = unpack_rows(np.asarray(gene_rows[p]), g)[src[ok, p]]
            at += g
        assert at == self.n_features
        return self.classify_bits(pack_bool_rows(X, self.words_per_row))

    def classify_protein_a2m_text(self, ptext, poffs, gtext, goffs, min_perc_state=0.0, want_bits=False):
        n = len(poffs) - 1
        cut = lambda t, o, i: bytes(t[o[i]:o[i + 1]]).decode("latin-1")
        nuc = [oracle_c.protein2gene(cut(ptext, poffs, i), cut(gtext, goffs, i), True) for i in range(n)]
        res = self.classify_a2m([s or "" for s in nuc], min_perc_state, want_bits)
        res.status[[s is None for s in nuc]] = _lib.SEQ_BACKTRANSLATE
        return res

    def classify_a2m(self, strings, min_perc_state=0.0, want_bits=False):
        from stag_amd.tables import pack_bool_rows
        n = len(strings)
        X = np.zeros((n, self.n_features), dtype=bool)
        status = np.zeros(n, dtype=np.int32)
        n_char = np.zeros(n, dtype=np.int32)
        for i, s in enumerate(strings):
            b, nc, na = oracle_c.encode_a2m(s)
            n_char[i] = nc
            if nc == 0:
                status[i] = _lib.SEQ_ZERO_COLUMNS
            elif not oracle_c.passes_filter(na, nc, min_perc_state):
                status[i] = _lib.SEQ_FILTERED
            elif 5 * nc != self.n_features:
                status[i] = _lib.SEQ_LENGTH_MISMATCH
            else:
                X[i] = b
        bits = pack_bool_rows(X, self.words_per_row)
        res = self.classify_bits(bits, status)
        res.n_char = n_char
        res.bits = bits if want_bits else None
        return res
