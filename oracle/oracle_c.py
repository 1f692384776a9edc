"""ctypes wrapper around oracle/liboracle.so (plain-C restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under stag_amd/ imports it.  It does its own tree numbering and
dense-row packing so that it shares no code with the product's table packer.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SEQ_OK, SEQ_MISSING_CLASSIFIER, SEQ_STRING_CLASSIFIER, SEQ_RANKCUT_SHAPE, SEQ_EMPTY_PATH, \
    SEQ_ZERO_COLUMNS = range(6)
TREE_ROOT = "tree_root"
NO_NEG = "no_negative_examples"


def build():
    """Compile liboracle.so with gcc (a few hundred ms)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        c_p = ctypes.c_void_p
        L.oracle_encode_a2m.restype = ctypes.c_int64
        L.oracle_encode_a2m.argtypes = [ctypes.c_char_p, ctypes.c_int64, c_p, ctypes.POINTER(ctypes.c_int64)]
        L.oracle_passes_filter.restype = ctypes.c_int
        L.oracle_passes_filter.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_double]
        L.oracle_protein2gene.restype = ctypes.c_int64
        L.oracle_protein2gene.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_char_p]
        L.oracle_logistic_bits.restype = ctypes.c_double
        L.oracle_logistic_bits.argtypes = [c_p, ctypes.c_int64, c_p]
        L.oracle_classify_bits.restype = None
        L.oracle_classify_bits.argtypes = [ctypes.c_int32, ctypes.c_int32, c_p, c_p, c_p, c_p,
                                           ctypes.c_int32, ctypes.c_int32, c_p, c_p, ctypes.c_int64,
                                           ctypes.c_int32] + [c_p] * 8
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _as_bytes(text):
    """The C restatement reads one byte per character.  The reference classifies characters with str
    methods (align.py:47,82-86), which know letters outside ASCII too: such a character is handed to C
    as an ASCII byte of the same class ('x' lower case, 'X' upper case, '?' neither)."""
    if text.isascii():
        return text.encode("ascii")
    return bytes(ord(c) if ord(c) < 128 else (0x78 if c.islower() else 0x58 if c.isupper() else 0x3F) for c in text)


def encode_a2m(text):
    """-> (bool[5*n_char], n_char, n_aligned)."""
    raw = _as_bytes(text)
    bits = np.zeros(5 * max(len(raw), 1), dtype=np.uint8)
    n_al = ctypes.c_int64(0)
    n_char = lib().oracle_encode_a2m(raw, len(raw), _ptr(bits), ctypes.byref(n_al))
    return bits[:5 * n_char].astype(bool), int(n_char), int(n_al.value)


def passes_filter(n_aligned, n_char, min_perc):
    return bool(lib().oracle_passes_filter(n_aligned, n_char, float(min_perc)))


def protein2gene(prot, gene, check_length=True):
    p, g = _as_bytes(prot), _as_bytes(gene)     # (a gene character outside ASCII comes back as its stand-in)
    out = ctypes.create_string_buffer(3 * len(p) + 4)
    n = lib().oracle_protein2gene(p, len(p), g, len(g), int(check_length), out)
    return None if n < 0 else out.raw[:n].decode("latin-1")


class DenseDB:
    """Tree positions numbered breadth-first (root = 0) + dense float64 rows."""

    def __init__(self, taxonomy, tax_function, classifiers, L):
        names, child_off, child_ids = [TREE_ROOT], [0], []
        i = 0
        limit = 64 * (len(taxonomy) + 2) + 1024
        while i < len(names):
            for kid in taxonomy.get(names[i]) or ():
                child_ids.append(len(names))
                names.append(kid)
                if len(names) > limit:
                    raise ValueError("taxonomy is not a tree")
            child_off.append(len(child_ids))
            i += 1
        self.names = names
        self.L = int(L)
        self.child_off = np.asarray(child_off, dtype=np.int32)
        self.child_ids = np.asarray(child_ids, dtype=np.int32)
        n = len(names)
        self.row_kind = np.zeros(n, dtype=np.uint8)
        self.rows = np.zeros((n, 1 + self.L), dtype=np.float64)
        for j, nm in enumerate(names):
            c = classifiers.get(nm)
            if c is None:
                continue
            if isinstance(c, str):
                self.row_kind[j] = 2
            else:
                self.row_kind[j] = 1
                self.rows[j] = np.asarray(c, dtype=np.float64)
        self.keys = sorted(tax_function)                      # string sort, classify.py:62
        self.n_tf = len(tax_function[self.keys[0]]) - 1 if self.keys else 0
        self.tf = np.zeros((max(len(self.keys), 1), 1 + self.n_tf), dtype=np.float64)
        for k, key in enumerate(self.keys):
            self.tf[k] = np.asarray(tax_function[key], dtype=np.float64)
        depth, lvl, frontier = 0, 0, [0]
        while frontier:
            nxt = []
            for p in frontier:
                nxt.extend(self.child_ids[self.child_off[p]:self.child_off[p + 1]])
            if nxt:
                depth += 1
            frontier = nxt
        self.max_depth = max(depth, 1)

    @classmethod
    def from_packed(cls, t):
        """From flat arrays (names, child_first, row_kind, intercept, row_off, row_feat, row_w, keys,
        tf, n_features) -- used by bench.py where 20k dense rows never exist as a dict."""
        self = cls.__new__(cls)
        n = len(t.names)
        self.names, self.L = list(t.names), int(t.n_features)
        self.child_off = (np.asarray(t.child_first, dtype=np.int64) - 1).astype(np.int32)
        self.child_ids = np.arange(1, n, dtype=np.int32)
        self.row_kind = np.ascontiguousarray(t.row_kind, dtype=np.uint8)
        self.rows = np.zeros((n, 1 + self.L), dtype=np.float64)
        self.rows[:, 0] = t.intercept
        lens = np.diff(t.row_off)
        self.rows[np.repeat(np.arange(n), lens), 1 + np.asarray(t.row_feat)] = t.row_w
        self.keys = list(t.keys)
        self.tf = np.ascontiguousarray(t.tf, dtype=np.float64)
        self.n_tf = self.tf.shape[1] - 1 if len(self.keys) else 0
        depth, frontier_lo, frontier_hi = 0, 0, 1
        while True:
            lo, hi = self.child_off[frontier_lo] + 1, self.child_off[frontier_hi] + 1
            if hi == lo:
                break
            depth, frontier_lo, frontier_hi = depth + 1, lo, hi
            if hi >= n:
                break
        self.max_depth = max(depth, 1)
        return self

    def classify(self, X):
        """X: bool/uint8 [n, L].  Returns dict of arrays (see stag_oracle.c)."""
        X = np.ascontiguousarray(X, dtype=np.uint8)
        n = X.shape[0]
        D, K = self.max_depth, len(self.keys)
        out = {
            "path_node": np.full((n, D), -1, dtype=np.int32),
            "path_prob": np.zeros((n, D), dtype=np.float64),
            "path_len": np.zeros(n, dtype=np.int32),
            "single_mask": np.zeros(n, dtype=np.uint32),
            "level_prob": np.zeros((n, max(K, 1)), dtype=np.float64),
            "selected": np.full(n, -1, dtype=np.int32),
            "n_aligned": np.zeros(n, dtype=np.int32),
            "status": np.zeros(n, dtype=np.int32),
        }
        lib().oracle_classify_bits(len(self.names), self.L, _ptr(self.child_off), _ptr(self.child_ids),
                                   _ptr(self.row_kind), _ptr(self.rows), K, self.n_tf, _ptr(self.tf),
                                   _ptr(X), n, D, _ptr(out["path_node"]), _ptr(out["path_prob"]),
                                   _ptr(out["path_len"]), _ptr(out["single_mask"]), _ptr(out["level_prob"]),
                                   _ptr(out["selected"]), _ptr(out["n_aligned"]), _ptr(out["status"]))
        out["level_prob"] = out["level_prob"][:, :K]
        return out

    def row_fields(self, out, i, gene_id):
        """The reference's 7 output fields for sequence i (classify.py:116-122)."""
        n = int(out["path_len"][i])
        tax = [self.names[j] for j in out["path_node"][i, :n]]
        perc = []
        for d in range(n):          # a placeholder-derived leading 1 is the Python int 1
            p = out["path_prob"][i, d]
            lead_single = all((out["single_mask"][i] >> e) & 1 for e in range(d + 1))
            perc.append("1" if lead_single else str(np.float64(p)))
        sel = int(out["selected"][i])
        selected = self.keys[sel] if sel >= 0 else -1
        per_level = [k + ":" + str(np.float64(p)) for k, p in zip(self.keys, out["level_prob"][i])]
        return [gene_id, ";".join(tax[0:int(selected) + 1]), "/".join(tax), selected,
                "/".join(perc), "/".join(per_level), str(int(out["n_aligned"][i]))]
