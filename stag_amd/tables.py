"""Packs what ``load_db`` returns (stag/databases.py:36-108) into the flat arrays of
``stag_tables_t`` (include/stag_hip.h).

The reference walks ``taxonomy: dict[str, list[str]]`` by name (classify.py:45-52).  Here every
*tree position* gets an integer id in breadth-first order (root ``tree_root`` = 0), so that
the children of a node are consecutive ids in their stored order and a tree level is one id
range: that is what the level-synchronous GPU walk needs.  A name reachable by two paths is
numbered once per path (its classifier row is shared by value), which keeps the result
identical to the name-based walk for any DAG; a cycle makes the reference recurse forever
and is rejected here.
"""
import ctypes

import numpy as np

from . import _lib

TREE_ROOT = "tree_root"            # stag/taxonomy3.py:18, used at classify.py:103
NO_NEG = "no_negative_examples"    # stag/databases.py:104


class PackedTables:
    """Host arrays in the layout of stag_tables_t, plus the names needed to print results."""

    def __init__(self, names, child_first, row_kind, intercept, row_off, row_feat, row_w, keys, tf, n_features):
        self.names = names
        self.child_first = np.ascontiguousarray(child_first, dtype=np.int32)
        self.row_kind = np.ascontiguousarray(row_kind, dtype=np.uint8)
        self.intercept = np.ascontiguousarray(intercept, dtype=np.float64)
        self.row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        self.row_feat = np.ascontiguousarray(row_feat, dtype=np.int32)
        self.row_w = np.ascontiguousarray(row_w, dtype=np.float64)
        self.keys = list(keys)
        self.tf = np.ascontiguousarray(tf, dtype=np.float64).reshape(len(self.keys), -1) if len(keys) else \
            np.zeros((0, 1), dtype=np.float64)
        self.n_features = int(n_features)
        self.n_nodes = len(self.child_first) - 1
        self.n_tf = self.tf.shape[1] - 1 if len(self.keys) else 0

    def as_struct(self):
        t = _lib.Tables()
        t.n_nodes = self.n_nodes
        t.n_features = self.n_features
        t.child_first = _lib.ptr(self.child_first)
        t.row_kind = _lib.ptr(self.row_kind)
        t.intercept = _lib.ptr(self.intercept)
        t.row_off = _lib.ptr(self.row_off)
        t.row_feat = _lib.ptr(self.row_feat)
        t.row_w = _lib.ptr(self.row_w)
        t.n_keys = len(self.keys)
        t.n_tf = self.n_tf
        t.tf = _lib.ptr(self.tf) if len(self.keys) else None
        return t

    @property
    def nnz(self):
        return int(self.row_off[-1])


def number_tree(taxonomy):
    """Breadth-first numbering of tree positions.  Returns (names, child_first[n+1])."""
    names = [TREE_ROOT]
    child_first = []
    limit = 64 * (len(taxonomy) + 2) + 1024
    i = 0
    while i < len(names):
        child_first.append(len(names))
        kids = taxonomy.get(names[i])       # None for names that are not keys: the walk stops there
        if kids:
            names.extend(kids)
            if len(names) > limit:
                raise ValueError("[E::tables] the taxonomy of this database is not a tree (cycle below %r)" % names[i])
        i += 1
    child_first.append(len(names))
    return names, np.asarray(child_first, dtype=np.int32)


def pack_db(taxonomy, tax_function, classifiers, n_features=None):
    """taxonomy / tax_function / classifiers exactly as returned by load_db -> PackedTables."""
    names, child_first = number_tree(taxonomy)
    n = len(names)
    L = n_features
    if L is None:
        for c in classifiers.values():
            if not isinstance(c, str):
                L = len(c) - 1
                break
    if L is None or L < 1:
        raise ValueError("[E::tables] cannot determine the alignment length: no coefficient row and no n_features")
    row_kind = np.zeros(n, dtype=np.uint8)
    intercept = np.zeros(n, dtype=np.float64)
    row_off = np.zeros(n + 1, dtype=np.int64)
    feats, ws = [], []
    packed = {}   # name -> (feat, w) so a shared name is scanned once
    for j, name in enumerate(names):
        c = classifiers.get(name) if j else None
        if c is None:
            row_off[j + 1] = row_off[j]
            continue
        if isinstance(c, str):
            row_kind[j] = 2
            row_off[j + 1] = row_off[j]
            continue
        if name not in packed:
            row = np.asarray(c, dtype=np.float64)
            if row.shape != (1 + L,):
                raise ValueError("[E::tables] classifier %r has %d coefficients, expected %d (operands could not "
                                 "be broadcast together)" % (name, row.size - 1, L))
            f = np.flatnonzero(row[1:]).astype(np.int32)
            packed[name] = (row[0], f, row[1:][f])
        b, f, w = packed[name]
        row_kind[j] = 1
        intercept[j] = b
        feats.append(f)
        ws.append(w)
        row_off[j + 1] = row_off[j] + len(f)
    row_feat = np.concatenate(feats) if feats else np.zeros(0, dtype=np.int32)
    row_w = np.concatenate(ws) if ws else np.zeros(0, dtype=np.float64)
    keys = sorted(tax_function)            # string sort, classify.py:62
    if keys:
        widths = {len(tax_function[k]) for k in keys}
        if len(widths) != 1:
            raise ValueError("[E::tables] tax_function rows have different lengths: %s" % sorted(widths))
        tf = np.stack([np.asarray(tax_function[k], dtype=np.float64) for k in keys])
    else:
        tf = np.zeros((0, 1))
    return PackedTables(names, child_first, row_kind, intercept, row_off, row_feat, row_w, keys, tf, L)


def words_per_row(n_features):
    """uint32 words of one packed row: stag_words_per_row of include/stag_hip.h (multiple of 4, >= 4)."""
    return max(((n_features + 31) // 32 + 3) // 4 * 4, 4)


def ascii_stand_in(s):
    """One byte per CHARACTER of an aligned string, as the kernels read it.  The reference classifies
    characters with str methods (align.py:47 `islower`, :82-86 `isupper` / `islower`), which also accept
    letters outside ASCII; such a character travels as an ASCII stand-in of the same class -- 'x' (a lower-case
    letter: an insert state), 'X' (an upper-case letter) or '?' (anything else) -- so that the byte-wise
    kernels decide as the reference does.  ASCII strings (every real alignment) pass through unchanged."""
    if s.isascii():
        return s.encode("ascii")
    return "".join(c if c.isascii() else "x" if c.islower() else "X" if c.isupper() else "?" for c in s).encode("ascii")


def pack_text(strings):
    """list[str] -> (uint8 text buffer, int64 offsets[n+1]) for the *_a2m entry points."""
    raw = [ascii_stand_in(s) if isinstance(s, str) else bytes(s) for s in strings]
    offs = np.zeros(len(raw) + 1, dtype=np.int64)
    if raw:
        np.cumsum([len(r) for r in raw], out=offs[1:])
    text = np.frombuffer(b"".join(raw) + b"\0" * 16, dtype=np.uint8)
    return text, offs


def pack_bool_rows(X, words_per_row):
    """bool/0-1 matrix [n, L] -> packed rows uint32[n, words_per_row] (bit f -> word f>>5, bit f&31)."""
    X = np.ascontiguousarray(X).astype(bool, copy=False)
    n, L = X.shape
    by = np.packbits(X, axis=1, bitorder="little")
    out = np.zeros((n, words_per_row * 4), dtype=np.uint8)
    out[:, :by.shape[1]] = by
    return out.view(np.uint32)


def unpack_rows(bits, n_features):
    """packed rows -> bool [n, n_features]."""
    by = np.ascontiguousarray(bits).view(np.uint8)
    return np.unpackbits(by, axis=1, bitorder="little")[:, :n_features].astype(bool)
