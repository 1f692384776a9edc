"""Test tooling: writes .stagDB files (the HDF5 layout of stag/databases.py:111-147) and genome
databases (the tar layout of stag/train_genome.py:46-60, a14) through tests/csrc/libstagdb_writer.so, so that tests
can build databases of any size where they run.  The product never writes databases (training is out of scope)."""
import ctypes
import os
import tarfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libstagdb_writer.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        L.stagdb_create.restype = ctypes.c_int64
        L.stagdb_create.argtypes = [ctypes.c_char_p]
        L.stagdb_close.argtypes = [ctypes.c_int64]
        L.stagdb_group.argtypes = [ctypes.c_int64, ctypes.c_char_p]
        L.stagdb_strings.argtypes = [ctypes.c_int64, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        L.stagdb_f64.argtypes = [ctypes.c_int64, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
        L.stagdb_bool.argtypes = [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int]
        _lib = L
    return _lib


def _ck(rc, what):
    if rc != 0:
        raise OSError("stagdb_writer: %s failed" % what)


def _strings(f, name, values, gzip):
    arr = (ctypes.c_char_p * max(len(values), 1))(*[v.encode("utf-8") for v in values])
    _ck(lib().stagdb_strings(f, name.encode(), arr, len(values), gzip), name)


def write_stagdb(path, taxonomy, tax_function, classifiers, hmm_text="HMMER3/f\n", use_cmalign=False,
                 align_protein=False, tool_version="0.8.2", gzip_rows=1):
    """taxonomy {node: [children]} (every node a key), tax_function {level: float64[1+D]}, classifiers
    {node: float64[1+L] or the string "no_negative_examples"}: the arguments of the reference's save_to_file.
    (gzip_rows: the reference compresses classifier rows at level 8; level 1 reads back the same and is what keeps
    writing dozens of databases inside a test quick.)"""
    f = lib().stagdb_create(os.fsencode(path))
    if f < 0:
        raise OSError("stagdb_writer: cannot create %s" % path)
    try:
        _strings(f, "tool_version", [tool_version], 0)
        _strings(f, "db_type", ["single_gene"], 0)
        _ck(lib().stagdb_bool(f, b"align_protein", int(align_protein)), "align_protein")
        _ck(lib().stagdb_bool(f, b"use_cmalign", int(use_cmalign)), "use_cmalign")
        _strings(f, "hmm_file", [hmm_text], 6)
        for group in ("taxonomy", "tax_function", "classifiers"):
            _ck(lib().stagdb_group(f, group.encode()), group)
        for node, kids in taxonomy.items():
            _strings(f, "taxonomy/" + node, list(kids), 6)
        for level, row in tax_function.items():
            row = np.ascontiguousarray(row, dtype=np.float64)
            _ck(lib().stagdb_f64(f, ("tax_function/" + str(level)).encode(), row.ctypes.data, len(row), 6), "tax_function")
        for node, row in classifiers.items():
            if isinstance(row, str):
                _strings(f, "classifiers/" + node, [row], 0)
            else:
                row = np.ascontiguousarray(row, dtype=np.float64)
                _ck(lib().stagdb_f64(f, ("classifiers/" + node).encode(), row.ctypes.data, len(row), gzip_rows), "classifiers")
    finally:
        _ck(lib().stagdb_close(f), "close")


def write_genome_db(path, gene_dbs, concatenated_db, thresholds, lengths):
    """The gztar of train_genome.py:46-60: one HDF5 per marker gene (member name = gene name), threshold_file.tsv
    (gene <tab> score: its order is the concatenation order), hmm_lengths_file.tsv (gene <tab> L) and
    concatenated_genes_STAG_database.HDF5.  gene_dbs {gene: path of its .stagDB}; thresholds / lengths {gene: value}
    in concatenation order."""
    tmp = path + ".parts"
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, "threshold_file.tsv"), "w") as fh:
        fh.writelines("%s\t%s\n" % kv for kv in thresholds.items())
    with open(os.path.join(tmp, "hmm_lengths_file.tsv"), "w") as fh:
        fh.writelines("%s\t%s\n" % kv for kv in lengths.items())
    with tarfile.open(path, "w:gz", compresslevel=1) as tar:
        for gene, p in gene_dbs.items():
            tar.add(p, arcname=gene)
        tar.add(os.path.join(tmp, "threshold_file.tsv"), arcname="threshold_file.tsv")
        tar.add(os.path.join(tmp, "hmm_lengths_file.tsv"), arcname="hmm_lengths_file.tsv")
        tar.add(concatenated_db, arcname="concatenated_genes_STAG_database.HDF5")
