"""Loading .stagDB files: mirror of stag/databases.py (zellerlab/stag 0.8.2).

``load_db`` keeps the reference's signature and 6-tuple result (databases.py:36-108) and
``load_genome_DB`` its 6-tuple (databases.py:15-33).  HDF5 access goes through
``libstag_h5.so`` (include/stag_h5.h, a thin reader over libhdf5) so the h5py-2.10 pin of the
reference (it compares vlen strings with ``str``, databases.py:51,101) does not apply; when
that library was not built, ``h5py`` of any version is used if importable (bytes are decoded).
"""
import contextlib
import ctypes
import os
import shutil
import sys
import tarfile
import tempfile

import numpy as np

from . import __version__ as tool_version
from .tables import NO_NEG

_HERE = os.path.dirname(os.path.abspath(__file__))
H5_LIB_PATH = os.path.join(_HERE, "lib", "libstag_h5.so")
c_p = ctypes.c_void_p

H5_EXPORTS = {
    "stag_h5_last_error": (ctypes.c_char_p, []),
    "stag_h5_open": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(c_p)]),
    "stag_h5_close": (None, [c_p]),
    "stag_h5_group_size": (ctypes.c_int, [c_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "stag_h5_member_name": (ctypes.c_int, [c_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]),
    "stag_h5_exists": (ctypes.c_int, [c_p, ctypes.c_char_p]),
    "stag_h5_dataset_info": (ctypes.c_int, [c_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32),
                                            ctypes.POINTER(ctypes.c_int64)]),
    "stag_h5_read_f64": (ctypes.c_int, [c_p, ctypes.c_char_p, c_p, ctypes.c_int64]),
    "stag_h5_read_u8": (ctypes.c_int, [c_p, ctypes.c_char_p, c_p, ctypes.c_int64]),
    "stag_h5_read_strings": (ctypes.c_int, [c_p, ctypes.c_char_p, c_p, ctypes.c_int64, c_p, ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_int64)]),
}
_h5lib = None


def _load_h5lib():
    global _h5lib
    if _h5lib is None and os.path.exists(H5_LIB_PATH):
        L = ctypes.CDLL(H5_LIB_PATH)
        for name, (res, args) in H5_EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _h5lib = L
    return _h5lib


class _NativeFile:
    """The few h5py.File operations load_db needs, over libstag_h5.so."""
    FLOAT, STRING, INTEGER = 0, 1, 2

    def __init__(self, path):
        self.L = _load_h5lib()
        self.h = c_p()
        self._ck(self.L.stag_h5_open(os.fsencode(path), ctypes.byref(self.h)))

    def _ck(self, rc):
        if rc != 0:
            raise OSError("[E::databases] " + (self.L.stag_h5_last_error() or b"?").decode("utf-8", "replace"))

    def close(self):
        if self.h:
            self.L.stag_h5_close(self.h)
            self.h = c_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def keys(self, group):
        n = ctypes.c_int64()
        self._ck(self.L.stag_h5_group_size(self.h, group.encode(), ctypes.byref(n)))
        buf = ctypes.create_string_buffer(70000)
        out = []
        for i in range(n.value):
            self._ck(self.L.stag_h5_member_name(self.h, group.encode(), i, buf, len(buf)))
            out.append(buf.value.decode("utf-8"))
        return out

    def read(self, name):
        """float64 ndarray, bool ndarray or list[str] depending on the dataset's class."""
        cls, n = ctypes.c_int32(), ctypes.c_int64()
        bname = name.encode("utf-8")
        self._ck(self.L.stag_h5_dataset_info(self.h, bname, ctypes.byref(cls), ctypes.byref(n)))
        n = n.value
        if cls.value == self.FLOAT:
            out = np.empty(n, dtype=np.float64)
            self._ck(self.L.stag_h5_read_f64(self.h, bname, out.ctypes.data_as(c_p), n))
            return out
        if cls.value == self.INTEGER:
            out = np.empty(n, dtype=np.uint8)
            self._ck(self.L.stag_h5_read_u8(self.h, bname, out.ctypes.data_as(c_p), n))
            return out.astype(bool)
        lens = np.zeros(max(n, 1), dtype=np.int64)
        need = ctypes.c_int64()
        self._ck(self.L.stag_h5_read_strings(self.h, bname, None, 0, lens.ctypes.data_as(c_p), n, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(max(need.value, 1))
        self._ck(self.L.stag_h5_read_strings(self.h, bname, buf, need.value, lens.ctypes.data_as(c_p), n,
                                             ctypes.byref(need)))
        raw, out, o = buf.raw, [], 0
        for i in range(n):
            out.append(raw[o:o + lens[i]].decode("utf-8"))
            o += int(lens[i])
        return out


class _H5pyFile:
    """Same interface over h5py (any version): bytes are decoded, so h5py >= 3 works."""

    def __init__(self, path):
        import h5py
        self.f = h5py.File(path, "r")

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def keys(self, group):
        return list(self.f[group])

    def read(self, name):
        d = self.f[name]
        if d.dtype.kind == "f":
            return np.array(d, dtype=np.float64)
        if d.dtype.kind in "biu":
            return np.array(d).astype(bool)
        return [v.decode("utf-8") if isinstance(v, bytes) else str(v) for v in d[()]]


def open_stagdb(path):
    if not os.path.isfile(path):
        raise OSError("[E::databases] Error: database file %s not found" % path)
    if _load_h5lib() is not None:
        return _NativeFile(path)
    try:
        return _H5pyFile(path)
    except ImportError:
        raise ImportError("[E::databases] cannot read HDF5: stag_amd/lib/libstag_h5.so is not built "
                          "(make -C stag_amd/csrc h5 needs libhdf5) and h5py is not importable") from None


GENOME_DB_CORE = ("threshold_file.tsv", "hmm_lengths_file.tsv", "concatenated_genes_STAG_database.HDF5")


def _two_column_table(path):
    """{first column: second column} of a two-column TSV, in file order (a line that does not have
    exactly two columns is an error, as it is for the reference's dict(...) over split lines)."""
    table = {}
    with open(path) as fh:
        for line in fh:
            key, value = line.rstrip().split("\t")
            table[key] = value
    return table


def load_genome_DB(database, tool_version=tool_version, verbose=3):
    """stag/databases.py:15-33 for the tar layout of train_genome.py:46-60.  Returns (marker-gene database
    file names, the temp directory they were unpacked into, {gene: threshold}, gene order = the order of
    threshold_file.tsv, {gene: alignment length}, path of the concatenated-genes database).  The caller
    removes the directory."""
    workdir = tempfile.mkdtemp()
    with tarfile.open(database, "r:gz") as tar:
        if hasattr(tarfile, "data_filter"):
            tar.extractall(workdir, filter="data")      # no absolute paths, links out of the tree or devices
        else:
            tar.extractall(workdir)
    unpacked = {name for name in os.listdir(workdir) if os.path.isfile(os.path.join(workdir, name))}
    for name in GENOME_DB_CORE:
        if name not in unpacked:
            raise ValueError(f"[E::align] Error: {name} is missing.")
    thresholds = _two_column_table(os.path.join(workdir, GENOME_DB_CORE[0]))
    lengths = _two_column_table(os.path.join(workdir, GENOME_DB_CORE[1]))
    gene_files = sorted(unpacked.difference(GENOME_DB_CORE))
    return gene_files, workdir, thresholds, list(thresholds), lengths, os.path.join(workdir, GENOME_DB_CORE[2])


def load_db(hdf5_DB_path, protein_fasta_input=None, aligned_sequences=None, dir_output=None):
    """stag/databases.py:36-108: returns (hmm temp file path, use_cmalign, taxonomy, tax_function,
    classifiers, db_tool_version) with the same types; ``dir_output`` dumps the database like
    ``stag unzip_db``."""
    with open_stagdb(hdf5_DB_path) as db_in:
        db_tool_version = db_in.read("tool_version")[0]
        use_proteins = bool(db_in.read("align_protein")[0])
        use_cmalign = bool(db_in.read("use_cmalign")[0])
        if dir_output:
            with open(os.path.join(dir_output, "parameters.tsv"), "w") as params_out:
                params_out.write("Tool version: " + str(db_tool_version) + "\n")
                params_out.write("Use proteins for the alignment: " + str(use_proteins) + "\n")
                params_out.write("Use cmalign instead of hmmalign: " + str(use_cmalign) + "\n")
        if db_in.read("db_type")[0] != "single_gene":
            sys.stderr.write("[E::main] Error: this database is not designed to run with stag classify\n")
            sys.exit(1)
        if not aligned_sequences and not dir_output:
            if protein_fasta_input and not use_proteins:
                raise ValueError("Protein provided, but the database was constructed on genes.\n")
            elif not protein_fasta_input and use_proteins:
                raise ValueError("Missing protein file (the database was constructed aligning proteins).\n")
        hmm_file = tempfile.NamedTemporaryFile(delete=False, mode="w")
        with hmm_file:
            os.chmod(hmm_file.name, 0o644)
            hmm_file.write(db_in.read("hmm_file")[0])
            hmm_file.flush()
            os.fsync(hmm_file.fileno())
        if dir_output:
            shutil.move(hmm_file.name, os.path.join(dir_output, "hmmfile.hmm"))
        taxonomy = {key: list(db_in.read("taxonomy/" + key)) for key in db_in.keys("taxonomy")}
        if dir_output:
            with open(os.path.join(dir_output, "node_hierarchy.tsv"), "w") as tax_out:
                print("Node", "Children", sep="\t", file=tax_out)
                for key, values in taxonomy.items():
                    print(key, *map(str, values), sep="\t", file=tax_out)
        tax_function = {str(key): db_in.read("tax_function/" + key) for key in db_in.keys("tax_function")}
        if dir_output:
            with open(os.path.join(dir_output, "taxonomy_function.tsv"), "w") as tf_out:
                for key, value in tax_function.items():
                    print(key, value, sep="\t", file=tf_out)
        classifiers = {}
        class_out = open(os.path.join(dir_output, "classifiers_weights.tsv"), "w") if dir_output \
            else contextlib.nullcontext()
        with class_out:
            for key in db_in.keys("classifiers"):
                value = db_in.read("classifiers/" + key)
                classifiers[key] = NO_NEG if isinstance(value, list) else value
                if dir_output:
                    print(key, *classifiers[key], sep="\t", file=class_out)
    return hmm_file.name, use_cmalign, taxonomy, tax_function, classifiers, db_tool_version
