"""Gather walk (k_walk_level) against matrix-core walk (k_walk_mfma) as the coefficient rows get denser
(VERDICT r1 item 7 / BASELINE config 5): `python tools/density_sweep.py` on a GPU box, from the repository root.
Same tree shape as the bench (scaled to ~2,000 nodes so that the 100 % rows fit in host memory while being
generated), 200k reads, amplicon windows and full-length reads.  Prints a markdown table (profiles/r2_walk_density.md)."""
import ctypes
import os
import subprocess
import sys
import json

sys.path.insert(0, os.getcwd())

CHILD = r'''
import sys, os, ctypes, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench_workload as bw
from stag_amd import _lib
from stag_amd.engine import Database
density, window, n = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sizes = [6, 30, 120, 400, 1200] if sys.argv[4] == "small" else [6, 36, 180, 720, 2500, 6500, 10000]
wl = bw.make_db(2, 1500, sizes, density, name="sweep")
text, offs, _ = bw.make_reads(wl, seed=7, n=n, window=window)
text = np.concatenate([text, np.zeros(-len(text) % 16, dtype=np.uint8)])
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
db = Database(wl.tables, device=0)
db.reserve(n, len(text))
dt = torch.from_numpy(text).to(dev); do = torch.from_numpy(offs).to(dev)
st = torch.cuda.current_stream(dev)
lib = _lib.lib()
def step():
    _lib.check(lib.stag_classify_a2m_dev(db.handle, dt.data_ptr(), len(text), do.data_ptr(), n, 0.0, None, st.cuda_stream))
for _ in range(2): step()
torch.cuda.synchronize()
db.profile_enable(True)
for _ in range(3): step()
prof = db.profile_read()
print(json.dumps({"walk_ms": prof["walk_level"][0] / 3, "launches": prof["walk_level"][1] // 3, "nodes": wl.tables.n_nodes,
                  "nnz_per_row": float(wl.tables.nnz) / max(int((wl.tables.row_kind == 1).sum()), 1)}))
'''

rows = []
CASES = [("small", 253, d, 200000) for d in (0.02, 0.05, 0.10, 0.20, 0.50, 1.00)] + \
        [("small", 1500, d, 200000) for d in (0.02, 0.05, 0.10, 0.20, 0.50, 1.00)] + \
        [("c2", 253, d, 1000000) for d in (0.02, 0.05, 0.10, 0.20)]
for tree, window, density, n_reads in CASES:
    if True:
        res = {}
        for mode in ("gather", "mfma"):
            env = dict(os.environ, STAG_WALK=mode)
            out = subprocess.run([sys.executable, "-c", CHILD, str(density), str(window), str(n_reads), tree], env=env,
                                 capture_output=True, text=True, timeout=900)
            if out.returncode:
                print(out.stderr[-2000:], file=sys.stderr)
                raise SystemExit(1)
            res[mode] = json.loads(out.stdout.strip().splitlines()[-1])
        g, m = res["gather"], res["mfma"]
        rows.append((tree, g["nodes"], n_reads, window, density, g["nnz_per_row"], g["walk_ms"], m["walk_ms"]))
        print("%s tree  window %4d  density %.2f  coefficients/row %6.0f  gather %.3f ms  matrix cores %.3f ms  -> %s" %
              (tree, window, density, g["nnz_per_row"], g["walk_ms"], m["walk_ms"], "mfma" if m["walk_ms"] < g["walk_ms"] else "gather"),
              flush=True)
print()
print("| tree (nodes), reads | reads cover | non-zero fraction of a row | coefficients per row | k_walk_level (gather) ms | k_walk_mfma ms | faster |")
print("|---|---|---|---|---|---|---|")
for tree, nodes, n_reads, window, density, nnz, g, m in rows:
    print("| %s (%d), %dk | %s | %.0f %% | %.0f | %.3f | %.3f | %s |" %
          (tree, nodes, n_reads // 1000, "253-column window" if window == 253 else "all 1,500 columns",
           100 * density, nnz, g, m, "matrix cores" if m < g else "gather"))
