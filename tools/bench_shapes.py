"""Throughput of stag_classify_a2m_dev on other shapes than the bench line (C3, full-length reads):
`python tools/bench_shapes.py` on a GPU box, from the repository root.  Numbers: profiles/README.md."""
import sys, os, ctypes, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench_workload as bw
from stag_amd import _lib
from stag_amd.engine import Database
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.lib()
def run(name, wl, n, **kw):
    text, offs, _ = bw.make_reads(wl, seed=7, n=n, **kw)
    text = np.concatenate([text, np.zeros(-len(text) % 16, dtype=np.uint8)])
    db = Database(wl.tables, device=0)
    db.reserve(n, len(text))
    dt = torch.from_numpy(text).to(dev); do = torch.from_numpy(offs).to(dev)
    st = torch.cuda.current_stream(dev)
    def step():
        _lib.check(lib.stag_classify_a2m_dev(db.handle, dt.data_ptr(), len(text), do.data_ptr(), n, 0.0, None, st.cuda_stream))
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    db.profile_enable(True); step(); prof = db.profile_read(); db.profile_enable(False)
    print(name, "n", n, "nodes", wl.tables.n_nodes, "ms/step %.3f" % ms, "Mreads/s %.1f" % (n / ms / 1e3),
          {k: round(v[0], 3) for k, v in prof.items()}, "text GB %.2f" % (len(text) / 1e9), flush=True)
    db.close()
run("C2", bw.config_16s(seed=2), 1_000_000)
run("C3 full-length protein-marker shape", bw.make_db(3, 1050, [4, 12, 40, 120, 300, 600, 900], 0.02, name="C3"), 1_000_000, window=1050)
run("C5 50k nodes", bw.make_db(5, 1500, [6, 30, 200, 1000, 4000, 15000, 30000], 0.02, name="C5"), 1_000_000)
run("full-length 16S", bw.config_16s(seed=2), 300_000, window=1500)
