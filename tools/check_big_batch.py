"""One call with more than 2^32 bytes of text and of packed rows (5M C2 reads: 7.5 GB of text, 4.7 GB of
rows): the first, middle and last 2,000 reads must come out exactly as when they are classified on their
own.  `python tools/check_big_batch.py [reads]` on a GPU box, from the repository root (needs ~40 GB of
host memory and ~20 GB of HBM)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import bench_workload as bw  # noqa: E402
from stag_amd.engine import Database  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
wl = bw.config_16s(seed=2)
parts, chunk = [], 1_000_000
t0 = time.time()
texts, offs_all, total = [], [np.zeros(1, dtype=np.int64)], 0
for i in range(0, n, chunk):
    m = min(chunk, n - i)
    text, offs, _ = bw.make_reads(wl, seed=500 + i // chunk, n=m)
    texts.append(text[:offs[-1]])
    offs_all.append(offs[1:] + total)
    total += int(offs[-1])
    print("generated %d reads (%.0f s)" % (i + m, time.time() - t0), flush=True)
text = np.concatenate(texts)
offs = np.concatenate(offs_all)
del texts
print("text %.2f GB, rows %.2f GB" % (len(text) / 1e9, n * 944 / 1e9), flush=True)
assert len(text) > 2 ** 32 or n < 3_000_000
with Database(wl.tables, device=0) as db:
    t0 = time.time()
    res = db.classify_a2m_text(text, offs)
    print("classified %d reads in %.1f s (host buffers in, host results out)" % (n, time.time() - t0), flush=True)
    assert (res.status == 0).all()
    for a in (0, n // 2 - 1000, n - 2000):
        b = a + 2000
        sub_text = text[offs[a]:offs[b]]
        sub = db.classify_a2m_text(np.concatenate([sub_text, np.zeros(16, dtype=np.uint8)]), offs[a:b + 1] - offs[a])
        for f in ("status", "n_char", "n_aligned", "path_len", "selected", "n_lead_single", "path_node", "path_prob", "level_prob"):
            assert np.array_equal(getattr(res, f)[a:b], getattr(sub, f)), (a, f)
        print("reads %d..%d identical to their own small batch" % (a, b), flush=True)
    # every path is a chain of the tree
    cf = np.asarray(wl.tables.child_first)
    pn = res.path_node
    parent = np.zeros(len(pn), dtype=np.int64)
    for d in range(pn.shape[1]):
        alive = pn[:, d] >= 0
        c = pn[alive, d]
        assert ((c >= cf[parent[alive]]) & (c < cf[parent[alive] + 1])).all(), d
        parent = np.where(alive, pn[:, d], parent)
    print("all %d paths are chains of the tree" % n)
