"""The C2 batch as 1, 2 and 4 independent parts on as many streams and handles (what overlapping the
small grouping kernels is worth): `python tools/bench_streams.py` on a GPU box, from the repository root."""
import sys, os, ctypes, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench_workload as bw
from stag_amd import _lib
from stag_amd.engine import Database
wl = bw.config_16s(seed=2, scale=1.0)
n = 1_000_000
text, offs, _ = bw.make_reads(wl, seed=1000, n=n)
dev = torch.device("cuda", 0)
lib = _lib.lib()
def setup(parts):
    ctxs = []
    per = n // parts
    for i in range(parts):
        a, b = i * per, (i + 1) * per if i < parts - 1 else n
        t = text[offs[a]:offs[b]]
        t = np.concatenate([t, np.zeros(-len(t) % 16, dtype=np.uint8)])
        o = (offs[a:b + 1] - offs[a]).astype(np.int64)
        db = Database(wl.tables, device=0)
        m = b - a
        db.reserve(m, len(t))
        ctxs.append((db, torch.from_numpy(t).to(dev), torch.from_numpy(o).to(dev), m, len(t), torch.cuda.Stream(dev)))
    return ctxs
for parts in (1, 2, 4):
    ctxs = setup(parts)
    def step():
        for db, dt, do, m, tb, st in ctxs:
            _lib.check(lib.stag_classify_a2m_dev(db.handle, dt.data_ptr(), tb, do.data_ptr(), m, 0.0, None, st.cuda_stream))
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt_ = (time.perf_counter() - t0) / 20
    print("parts", parts, "ms/step %.4f" % (dt_ * 1e3), "Mseq/s %.1f" % (n / dt_ / 1e6), flush=True)
    del ctxs
