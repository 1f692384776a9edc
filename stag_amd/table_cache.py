"""On-disk cache of the packed tables of a .stagDB.

Loading a database the reference's way (stag/databases.py:36-108) reads one HDF5 dataset per
tree node -- tens of thousands of small gzip datasets, dense rows of 1 + L float64 -- and
classify() then keeps only the non-zero coefficients (stag_amd.tables.pack_db).  The packed form
is what the GPU needs, is 50x smaller, and depends only on the file, so it is kept next to the
user's other caches: $STAG_AMD_CACHE, else ~/.cache/stag_amd; STAG_AMD_CACHE=off disables it.
An entry is keyed by the file's absolute path, size and mtime and holds plain arrays (.npz, no
pickle): the tables, the node names, the HMM text and the two flags load_db checks.
"""
import hashlib
import os
import tempfile

import numpy as np

from .tables import PackedTables

FORMAT = 1


def cache_dir():
    d = os.environ.get("STAG_AMD_CACHE")
    if d is not None and d.lower() in ("", "off", "0", "none"):
        return None
    return d or os.path.join(os.path.expanduser("~"), ".cache", "stag_amd")


def _entry(path):
    d = cache_dir()
    if d is None or not os.path.isfile(path):
        return None
    st = os.stat(path)
    key = "%s|%d|%d|%d" % (os.path.abspath(path), st.st_size, st.st_mtime_ns, FORMAT)
    return os.path.join(d, hashlib.sha1(key.encode()).hexdigest() + ".npz")


def check_inputs(use_proteins, protein_fasta_input, aligned_sequences):
    """The consistency check of stag/databases.py:56-61."""
    if not aligned_sequences:
        if protein_fasta_input and not use_proteins:
            raise ValueError("Protein provided, but the database was constructed on genes.\n")
        elif not protein_fasta_input and use_proteins:
            raise ValueError("Missing protein file (the database was constructed aligning proteins).\n")


def load(path, protein_fasta_input=None, aligned_sequences=None):
    """(hmm temp file path, use_cmalign, PackedTables, db_tool_version) from the cache, or None."""
    entry = _entry(path)
    if entry is None or not os.path.isfile(entry):
        return None
    try:
        with np.load(entry, allow_pickle=False) as z:
            t = PackedTables([str(s) for s in z["names"]], z["child_first"], z["row_kind"], z["intercept"], z["row_off"],
                             z["row_feat"], z["row_w"], [str(s) for s in z["keys"]], z["tf"], int(z["n_features"]))
            use_proteins, use_cmalign = bool(z["use_proteins"]), bool(z["use_cmalign"])
            hmm_text, version = str(z["hmm_text"]), str(z["tool_version"])
    except Exception:
        return None          # unreadable entry: fall back to the database itself
    check_inputs(use_proteins, protein_fasta_input, aligned_sequences)
    hmm_file = tempfile.NamedTemporaryFile(delete=False, mode="w")
    with hmm_file:
        os.chmod(hmm_file.name, 0o644)
        hmm_file.write(hmm_text)
    return hmm_file.name, use_cmalign, t, version


def store(path, tables, hmm_file_path, use_cmalign, use_proteins, db_tool_version):
    """Best effort: a cache that cannot be written is not an error."""
    entry = _entry(path)
    if entry is None:
        return False
    try:
        os.makedirs(os.path.dirname(entry), exist_ok=True)
        with open(hmm_file_path) as fh:
            hmm_text = fh.read()
        tmp = entry + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, names=np.array(tables.names, dtype=str), child_first=tables.child_first, row_kind=tables.row_kind,
                 intercept=tables.intercept, row_off=tables.row_off, row_feat=tables.row_feat, row_w=tables.row_w,
                 keys=np.array(tables.keys, dtype=str), tf=tables.tf, n_features=np.int64(tables.n_features),
                 use_proteins=np.bool_(use_proteins), use_cmalign=np.bool_(use_cmalign), hmm_text=np.array(hmm_text),
                 tool_version=np.array(str(db_tool_version)))
        os.replace(tmp, entry)
        return True
    except Exception:
        return False
