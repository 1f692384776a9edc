"""Seeded synthetic workloads of SURVEY.md 8(d) for bench.py and the full-size GPU tests.

Nothing here is read from the reference or from disk: trees, sparse LASSO-like rows and aligned
reads are generated with numpy.random.default_rng(seed).  The database is produced directly
in the CSR layout of stag_tables_t (20k dense fp64 rows would be 1.2 GB of zeros).
"""
import numpy as np

from stag_amd.tables import PackedTables, TREE_ROOT

SYMS = np.frombuffer(b"-TGCA", dtype=np.uint8)      # symbol code s -> character whose one-hot bit is s


class Workload:
    """tables: PackedTables; consensus: uint8 [n_nodes, ncol] symbol codes 1..4; leaves: node ids."""

    def __init__(self, name, tables, consensus, leaves, ncol, window):
        self.name, self.tables, self.consensus, self.leaves = name, tables, consensus, leaves
        self.ncol, self.window = ncol, window


def make_tree(rng, level_sizes):
    """Breadth-first tree with ~level_sizes[d] nodes at depth d+1; fan-outs geometric, min 1."""
    counts, parents = [], 1
    for target in level_sizes:
        mean = max(target / parents, 1.0)
        k = rng.geometric(1.0 / mean, size=parents) if mean > 1.0 else np.ones(parents, dtype=np.int64)
        if parents == 1:
            k = np.array([target], dtype=np.int64)            # the root's fan-out is not left to one draw
        counts.append(k)
        parents = int(k.sum())
    counts.append(np.zeros(parents, dtype=np.int64))          # the deepest level has no children
    counts = np.concatenate(counts)
    child_first = np.empty(len(counts) + 1, dtype=np.int32)
    child_first[0] = 1
    child_first[1:] = 1 + np.cumsum(counts)
    return child_first


def make_db(seed, ncol, level_sizes, nnz_frac=0.02, name="synthetic"):
    rng = np.random.default_rng(seed)
    L = 5 * ncol
    child_first = make_tree(rng, level_sizes)
    n = len(child_first) - 1
    D = len(level_sizes)
    parent = np.zeros(n, dtype=np.int64)
    for p in range(n):
        parent[child_first[p]:child_first[p + 1]] = p
    # consensus: children inherit from the parent with 3 % substitutions
    cons = np.zeros((n, ncol), dtype=np.uint8)
    cons[0] = rng.integers(1, 5, size=ncol)
    for c in range(1, n):
        row = cons[parent[c]].copy()
        mut = rng.random(ncol) < 0.03
        row[mut] = rng.integers(1, 5, size=int(mut.sum()))
        cons[c] = row
    nnz = max(int(round(nnz_frac * L)), 4)
    row_kind = np.zeros(n, dtype=np.uint8)
    intercept = np.zeros(n)
    feats = [np.zeros(0, dtype=np.int32)] * n
    ws = [np.zeros(0)] * n
    for p in range(n):
        c0, c1 = int(child_first[p]), int(child_first[p + 1])
        k = c1 - c0
        if k < 2:
            continue                                          # single children are never scored
        sib = cons[c0:c1]                                     # [k, ncol]
        cols = np.flatnonzero((sib != sib[0]).any(axis=0))    # columns that tell the siblings apart
        sib_feats = 5 * cols[None, :] + sib[:, cols]          # [k, ncols]: each sibling's own symbol there
        for j in range(k):
            c = c0 + j
            f_pos = sib_feats[j]
            f_neg = np.setdiff1d(np.delete(sib_feats, j, axis=0).ravel(), f_pos)
            f = np.concatenate([f_pos, f_neg])
            w = np.concatenate([np.abs(rng.normal(0, 2, size=len(f_pos))),
                                -np.abs(rng.normal(0, 2, size=len(f_neg)))])
            if len(f) > nnz:
                keep = rng.choice(len(f), size=nnz, replace=False)
                f, w = f[keep], w[keep]
            elif len(f) < nnz:                                # fill with weak, uninformative features
                pool = rng.choice(L, size=min(L, 2 * (nnz - len(f)) + 8), replace=False)
                extra = np.setdiff1d(pool, f)
                if len(extra) < nnz - len(f):                 # (dense rows: take whatever is left)
                    extra = np.setdiff1d(np.arange(L), f)
                extra = extra[: nnz - len(f)]
                f = np.concatenate([f, extra])
                w = np.concatenate([w, rng.normal(0, 0.2, size=len(extra))])
            order = np.argsort(f, kind="stable")
            feats[c] = f[order].astype(np.int32)
            ws[c] = w[order]
            row_kind[c] = 1
            intercept[c] = rng.normal(-1, 1)
    row_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(f) for f in feats], out=row_off[1:])
    keys = sorted(str(l) for l in range(-1, D))
    tf = rng.normal(0, 1, size=(len(keys), 1 + D))
    names = [TREE_ROOT] + ["r%d" % i for i in range(1, n)]
    tables = PackedTables(names, child_first, row_kind, intercept, row_off, np.concatenate(feats),
                          np.concatenate(ws), keys, tf, L)
    leaves = np.flatnonzero(np.diff(child_first) == 0)
    return Workload(name, tables, cons, leaves, ncol, None)


def make_reads(wl, seed, n, window=253, start=None, sub=0.02, n_frac=0.005, max_ins=3):
    """Aligned (A2M) reads: '-' outside a `window`-column window, the leaf consensus inside with
    2 % substitutions, 0.5 % N and 0..max_ins lowercase inserts.  Returns (text uint8, offs int64,
    true leaf ids)."""
    rng = np.random.default_rng(seed)
    ncol = wl.ncol
    window = min(window, ncol)
    if start is None:
        start = (ncol - window) // 2
    chunks, lens, leaves_all = [], [], []
    for b0 in range(0, n, 100_000):
        m = min(100_000, n - b0)
        leaf = rng.choice(wl.leaves, size=m)
        sym = np.zeros((m, ncol), dtype=np.uint8)
        inside = wl.consensus[leaf, start:start + window].copy()
        mut = rng.random(inside.shape) < sub
        inside[mut] = rng.integers(1, 5, size=int(mut.sum()))
        chars = SYMS[inside]
        chars[rng.random(inside.shape) < n_frac] = ord("N")
        base = np.full((m, ncol), ord("-"), dtype=np.uint8)
        base[:, start:start + window] = chars
        del sym
        n_ins = rng.integers(0, max_ins + 1, size=m)
        out_len = ncol + n_ins
        offs = np.zeros(m + 1, dtype=np.int64)
        np.cumsum(out_len, out=offs[1:])
        out = np.empty(int(offs[-1]), dtype=np.uint8)
        pos = np.sort(rng.integers(start, start + window, size=(m, max_ins)), axis=1) if max_ins else \
            np.zeros((m, 0), dtype=np.int64)
        pos = np.where(np.arange(max_ins)[None, :] < n_ins[:, None], pos, ncol + 1)
        cols = np.arange(ncol)[None, :]
        shift = np.zeros((m, ncol), dtype=np.int64)
        for j in range(max_ins):
            shift += cols >= pos[:, j:j + 1]
        out[(offs[:-1, None] + cols + shift).ravel()] = base.ravel()
        for j in range(max_ins):                       # insert j of read r sits just before column pos[r, j]
            has = n_ins > j
            at = offs[:-1][has] + pos[has, j] + j
            out[at] = rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), size=int(has.sum()))
        chunks.append(out)
        lens.append(out_len)
        leaves_all.append(leaf)
    text = np.concatenate(chunks)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.concatenate(lens), out=offs[1:])
    pad = np.zeros(16, dtype=np.uint8)
    return np.concatenate([text, pad]), offs, np.concatenate(leaves_all)


# configurations of BASELINE.json (shapes assumed by SURVEY.md 8)
def config_16s(seed=2, scale=1.0):
    """C2: 16S amplicon stagDB: ncol 1500 (L 7500), 7 ranks, ~20k nodes."""
    sizes = [6, 36, 180, 720, 2500, 6500, 10000]
    sizes = [max(2, int(s * scale)) for s in sizes]
    return make_db(seed, 1500, sizes, 0.02, name="16S_C2")


def config_c3(seed=3):
    """C3: protein marker gene stagDB shape: ncol 1050 (350 aa x 3, L 5250), 7 ranks; reads cover every column."""
    return make_db(seed, 1050, [4, 12, 40, 120, 300, 600, 900], 0.02, name="C3")


def config_c5(seed=5):
    """C5: synthetic ~50k-node taxonomy at the 16S width (ncol 1500, L 7500)."""
    return make_db(seed, 1500, [6, 30, 200, 1000, 4000, 15000, 30000], 0.02, name="C5")


def dense_rows(tables, node_ids=None):
    """Dense [n_nodes, 1+L] fp64 rows for the C oracle (only for node_ids if given, else all)."""
    n, L = tables.n_nodes, tables.n_features
    rows = np.zeros((n, 1 + L))
    rows[:, 0] = tables.intercept
    ids = range(n) if node_ids is None else node_ids
    for c in ids:
        a, b = tables.row_off[c], tables.row_off[c + 1]
        rows[c, 1 + tables.row_feat[a:b]] = tables.row_w[a:b]
    return rows


class LazyClassifiers(dict):
    """classifiers dict for the NumPy oracle that materialises dense rows on first use."""

    def __init__(self, tables):
        super().__init__()
        self.t = tables
        self.index = {nm: i for i, nm in enumerate(tables.names)}

    def __missing__(self, name):
        c = self.index[name]
        if self.t.row_kind[c] != 1:
            raise KeyError(name)
        row = np.zeros(1 + self.t.n_features)
        row[0] = self.t.intercept[c]
        a, b = self.t.row_off[c], self.t.row_off[c + 1]
        row[1 + self.t.row_feat[a:b]] = self.t.row_w[a:b]
        self[name] = row
        return row


def taxonomy_dict(tables):
    cf = tables.child_first
    return {tables.names[i]: [tables.names[j] for j in range(cf[i], cf[i + 1])] for i in range(tables.n_nodes)}


def tax_function_dict(tables):
    return {k: tables.tf[i] for i, k in enumerate(tables.keys)}
