"""stag_classify_a2m_noalloc_dev (include/stag_hip.h: the entry point for callers that cannot
allocate; its workspace is one static device array per GPU) against stag_classify_a2m_dev on the
same inputs, through the C ABI: bit-identical outputs (both run the same kernels), on two calls in a
row, the second with a smaller batch so that the array still holds the first call's contents.
Collected last on purpose (the round's end runs pytest -x)."""
import ctypes

import numpy as np
import pytest

from hipmem import DevArray
from stag_amd import _lib, tables
from stag_amd.engine import Database
from test_gpu_dev_entry import FIELDS, amplicon_strings, device_results, padded
from test_gpu_parity import random_db

pytestmark = pytest.mark.gpu


def run(entry, db, strings, min_perc, null_fields=()):
    text, offs = tables.pack_text(strings)
    t16 = padded(text)
    n = len(strings)
    d_text, d_offs = DevArray(t16), DevArray(offs)
    arrs, res = device_results(n, db.max_depth, db.n_keys)
    for f in null_fields:
        setattr(res, f, None)
    _lib.check(entry(db.handle, d_text.ptr, len(t16), d_offs.ptr, n, min_perc, ctypes.byref(res), None))
    return {f: arrs[f].get() for f in FIELDS if f not in null_fields}


def test_noalloc_entry_equals_classify_a2m_dev_twice_in_a_row():
    L_ = _lib.lib()
    rng = np.random.default_rng(77)
    ncol = 170
    L = 5 * ncol
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [1, 2, 5], [0, 2, 3], [2, 2]], 0.08)
    first = amplicon_strings(rng, ncol, 3000)
    second = amplicon_strings(rng, ncol, 700)          # smaller: the scratch still holds the first call's data
    with Database.from_dicts(taxonomy, tax_function, classifiers, n_features=L) as db:
        for strings, min_perc in ((first, 20.0), (second, 0.0)):
            db.reserve(len(strings), sum(map(len, strings)) + 16)
            want = run(L_.stag_classify_a2m_dev, db, strings, min_perc)
            got = run(L_.stag_classify_a2m_noalloc_dev, db, strings, min_perc)
            for f in FIELDS:
                assert np.array_equal(got[f], want[f]), f
            assert len(set(want["status"].tolist())) >= 3 and (want["path_len"] > 0).any()
        # outputs the caller leaves NULL are written into the scratch instead
        got = run(L_.stag_classify_a2m_noalloc_dev, db, second, 0.0, null_fields=("path_prob", "level_prob", "n_aligned"))
        for f in got:
            assert np.array_equal(got[f], want[f]), f
        # a batch whose workspace exceeds the static array is refused, with the sizes in the message
        n_big = 2_000_000
        d_text, d_offs = DevArray(np.zeros(16, dtype=np.uint8)), DevArray(np.zeros(4, dtype=np.int64))
        rc = L_.stag_classify_a2m_noalloc_dev(db.handle, d_text.ptr, 16, d_offs.ptr, n_big, 0.0, None, None)
        assert rc == -1 and b"static scratch" in L_.stag_last_error()
        # and the handle still works afterwards
        again = run(L_.stag_classify_a2m_noalloc_dev, db, second, 0.0)
        for f in FIELDS:
            assert np.array_equal(again[f], want[f]), f
