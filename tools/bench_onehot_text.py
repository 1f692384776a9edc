"""The -s path (classify.py:16-20 + 94-124) end to end on a GPU box: a file of N one-hot lines of the C2 shape
(7,500 fields, 15 kB per read) through
  host   ingest.onehot_batches (C parser on one core) + Database.classify_bits
  device ingest.onehot_text_batches (lines found on the host) + Database.classify_onehot_text (k_parse_onehot)
and the rows the two produce compared.  `python tools/bench_onehot_text.py [N]` from the repository root."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch      # before libstag_hip.so: one HIP runtime per process, the one torch brings

import bench_workload as bw
from stag_amd import _lib, ingest
from stag_amd.classify import parse_onehot_line
from stag_amd.engine import Database

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
wl = bw.config_16s(seed=2)
L, t = wl.tables.n_features, wl.tables
text, offs, _ = bw.make_reads(wl, seed=5, n=N)
with Database(t) as db:
    W = db.words_per_row
    path = os.path.join(tempfile.mkdtemp(), "reads.tsv")
    t0 = time.perf_counter()
    with open(path, "wb") as fh:
        for a in range(0, N, 10_000):
            b = min(N, a + 10_000)
            sub = np.concatenate([text[offs[a]:offs[b]], np.zeros(-(int(offs[b] - offs[a])) % 16, dtype=np.uint8)])
            res = db.classify_a2m_text(sub, offs[a:b + 1] - offs[a], want_bits=True)
            fh.write(ingest.format_onehot(["read%d" % i for i in range(a, b)], res.bits, L))
    size = os.path.getsize(path)
    print("file: %d lines, %.2f GB, written in %.1f s" % (N, size / 1e9, time.perf_counter() - t0), flush=True)

    def host():
        out = []
        for ids, bits, nf in ingest.onehot_batches(path, L, W, batch_rows=1 << 14, slow_line=parse_onehot_line):
            status = np.where(nf == L, _lib.SEQ_OK, _lib.SEQ_LENGTH_MISMATCH).astype(np.int32)
            out.append(db.classify_bits(bits, status))
        return out

    def device():
        out = []
        for ids, tx, bb, be, raw in ingest.onehot_text_batches(path, batch_rows=1 << 14, slow_line=parse_onehot_line):
            out.append(db.classify_onehot_text(tx, bb, be, parse_onehot_line, raw)[0])
        return out

    results = {}
    for name, fn in (("host", host), ("device", device), ("device", device)):
        t0 = time.perf_counter()
        results[name] = fn()
        dt = time.perf_counter() - t0
        print("%-6s parser: %.2f s, %.1f k reads/s, %.2f GB/s of text" % (name, dt, N / dt / 1e3, size / dt / 1e9), flush=True)
    for f in ("path_node", "path_prob", "selected", "n_aligned", "status"):
        a = np.concatenate([getattr(r, f) for r in results["host"]])
        b = np.concatenate([getattr(r, f) for r in results["device"]])
        assert np.array_equal(a, b), f
    print("results identical", flush=True)

    # ---- the kernel alone, text resident in HBM: one batch of 16,384 lines
    dev = torch.device("cuda", 0)
    batches = ingest.onehot_text_batches(path, batch_rows=1 << 14, slow_line=parse_onehot_line)   # (owns the text buffers)
    ids, tx, bb, be, raw = next(batches)
    pad = np.zeros(len(tx) + (-len(tx) % 16) + 16, dtype=np.uint8)
    pad[:len(tx)] = tx
    d_text, d_bb, d_be = torch.from_numpy(pad).to(dev), torch.from_numpy(bb.copy()).to(dev), torch.from_numpy(be.copy()).to(dev)
    m = len(ids)
    d_bits = torch.zeros((m, W), dtype=torch.int32, device=dev)
    d_nf, d_flag = torch.zeros(m, dtype=torch.int32, device=dev), torch.zeros(m, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(7):
        if rep == 2:
            ev[0].record(st)
        _lib.check(_lib.lib().stag_parse_onehot_dev(L, d_text.data_ptr(), len(pad), d_bb.data_ptr(), d_be.data_ptr(), m,
                                                    d_bits.data_ptr(), d_nf.data_ptr(), d_flag.data_ptr(), st.cuda_stream))
    ev[1].record(st)
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    print("k_parse_onehot: %d lines, %.1f MB of text in %.3f ms = %.2f TB/s (text read + rows written: %.1f MB)"
          % (m, len(tx) / 1e6, ms, len(tx) / ms / 1e9, (len(tx) + m * W * 4) / 1e6), flush=True)
    del d_text, d_bits, batches

# ---- classify() itself (what `stag_amd classify -s` runs) on the same file: database from memory instead of a
# .stagDB, everything else as the command line does it, output file included
import stag_amd.classify as sc
for long_out in (False, True):
    hmm = tempfile.NamedTemporaryFile(delete=False)
    hmm.close()
    sc.table_cache.load = lambda *a, hmm=hmm: (hmm.name, False, t, sc.tool_version)
    out = path + (".long" if long_out else ".short")
    t0 = time.perf_counter()
    sc.classify("synthetic C2", aligned_sequences=path, output=out, long_out=long_out, verbose=1)
    dt = time.perf_counter() - t0
    print("classify(-s, long_out=%s): %.2f s, %.1f k reads/s, output %.1f MB" % (long_out, dt, N / dt / 1e3, os.path.getsize(out) / 1e6),
          flush=True)
    os.remove(out)
os.remove(path)
