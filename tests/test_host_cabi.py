"""CPU-only checks of the boundary: libstag_hip.so loads, exports every symbol declared in
include/stag_hip.h, and its host-only entry point (back-translation) matches the reference's
golden vectors.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden_util as gu
from stag_amd import _lib, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "stag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stag_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(L, name), name
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)


def test_words_per_row():
    f = _lib.lib().stag_words_per_row
    assert f(1) == 4 and f(128) == 4 and f(129) == 8 and f(7500) == 236 and f(145) == 8


def test_backtranslate_golden(capfd):
    from stag_amd.align import protein2gene_alignment
    for case in gu.load("backtranslate.json"):
        # the Python-level function takes any string; the C ABI one byte per character (ASCII)
        assert protein2gene_alignment("id", case["protein"], case["gene"], case["check_length"]) == case["result"], case
        if not (case["protein"].isascii() and case["gene"].isascii()):
            continue
        p, g = case["protein"].encode(), case["gene"].encode()
        out = ctypes.create_string_buffer(3 * len(p) + 4)
        n = ctypes.c_int64(0)
        _lib.check(_lib.lib().stag_backtranslate(p, len(p), g, len(g), int(case["check_length"]), out, ctypes.byref(n)))
        got = None if n.value < 0 else out.raw[:n.value].decode()
        assert got == case["result"], case


def test_backtranslate_random_pairs_equal_the_oracle(capfd):
    """Seeded random (protein alignment, gene) pairs, with inserts, odd symbols, stop codons and wrong
    lengths, with and without the length check: the host entry point against the NumPy-free restatement
    of align.py:63-94 that the golden vectors pin."""
    from oracle import oracle_np
    rng = np.random.default_rng(6394)
    alphabet = list("ACDEFGHIKLMNPQRSTVWY") + ["-", "-", "-", "a", "c", "x", ".", "*", "X", "1"]
    for case in range(3000):
        m = int(rng.integers(0, 40))
        prot = "".join(rng.choice(alphabet, size=m))
        res = sum(ch != "-" for ch in prot)
        glen = max(0, 3 * res + int(rng.choice([0, 0, 3, 3, 1, -1, 2, -3, 6])))
        gene = "".join(rng.choice(list("ACGTNacgt-"), size=glen))
        for check in (True, False):
            want = oracle_np.protein2gene_alignment("g", prot, gene, check_length=check)
            p, g = prot.encode(), gene.encode()
            out = ctypes.create_string_buffer(3 * len(p) + 4)
            n = ctypes.c_int64(0)
            _lib.check(_lib.lib().stag_backtranslate(p, len(p), g, len(g), int(check), out, ctypes.byref(n)))
            got = None if n.value < 0 else out.raw[:n.value].decode()
            assert got == want, (prot, gene, check)
    capfd.readouterr()


def test_no_gpu_fails_loudly():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    t = tables.pack_db({"tree_root": ["a", "b"], "a": [], "b": []}, {"0": np.zeros(2)},
                       {"a": np.ones(11), "b": np.ones(11)})
    from stag_amd.engine import Database
    with pytest.raises(_lib.StagError, match="no HIP device"):
        Database(t)


def test_number_tree_and_pack():
    taxonomy = {"tree_root": ["b", "a"], "a": ["a1", "a2", "a3"], "b": ["b1"], "a1": [], "a2": [], "a3": [],
                "b1": ["x"]}     # x is not a key: a leaf
    names, cf = tables.number_tree(taxonomy)
    assert names == ["tree_root", "b", "a", "b1", "a1", "a2", "a3", "x"]
    assert cf.tolist() == [1, 3, 4, 7, 8, 8, 8, 8, 8]
    rng = np.random.default_rng(0)
    L = 40
    cls = {k: rng.normal(size=1 + L) * (rng.random(1 + L) < 0.3) for k in ("a", "b", "a1", "a2")}
    cls["a3"] = tables.NO_NEG
    tf = {"10": np.arange(4.0), "2": np.arange(4.0) + 1, "-1": np.arange(4.0) + 2}
    t = tables.pack_db(taxonomy, tf, cls)
    assert t.keys == ["-1", "10", "2"] and t.n_tf == 3 and t.n_features == L
    assert t.row_kind.tolist() == [0, 1, 1, 0, 1, 1, 2, 0]
    j = names.index("a2")
    f = t.row_feat[t.row_off[j]:t.row_off[j + 1]]
    dense = np.zeros(L)
    dense[f] = t.row_w[t.row_off[j]:t.row_off[j + 1]]
    assert np.array_equal(dense, cls["a2"][1:]) and t.intercept[j] == cls["a2"][0]
    with pytest.raises(ValueError, match="not a tree"):
        tables.number_tree({"tree_root": ["a"], "a": ["tree_root"]})


def test_pack_bool_rows_roundtrip():
    rng = np.random.default_rng(1)
    X = rng.random((7, 145)) < 0.3
    W = _lib.lib().stag_words_per_row(145)
    bits = tables.pack_bool_rows(X, W)
    assert bits.shape == (7, W) and bits.dtype == np.uint32
    assert bool(bits[3, 70 >> 5] >> (70 & 31) & 1) == bool(X[3, 70])
    assert np.array_equal(tables.unpack_rows(bits, 145), X)
