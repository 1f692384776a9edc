"""CPU: the .stagDB reader (libstag_h5.so over libhdf5) against a database written by the
REFERENCE's own writer (tests/golden/tiny.stagDB, made by oracle/make_golden.py through
stag.databases.save_to_file), and the genome-database tar layout."""
import ctypes
import hashlib
import os
import re
import tarfile

import numpy as np
import pytest

import golden_util as gu
from stag_amd import databases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(gu.GOLDEN, "tiny.stagDB")

needs_reader = pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built (no libhdf5)")


@needs_reader
def test_h5_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "stag_h5.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(stag_h5_[a-z0-9_]+)\s*\(", text)))
    L = ctypes.CDLL(databases.H5_LIB_PATH)
    assert set(names) == set(databases.H5_EXPORTS)
    for n in names:
        assert hasattr(L, n)


@needs_reader
def test_load_db_matches_reference_writer():
    meta = gu.load("tiny_db.json")
    hmm, use_cmalign, taxonomy, tax_function, classifiers, version = databases.load_db(TINY)
    try:
        assert use_cmalign is False and version == meta["tool_version"]
        assert taxonomy == meta["taxonomy"]
        assert taxonomy["Alpha"] == [] and "Beta-Delta" in taxonomy      # '/' -> '-' at training, leaves = []
        assert sorted(tax_function) == sorted(meta["tax_function"])
        for k, row in meta["tax_function"].items():
            assert np.array_equal(tax_function[k], np.array([float(v) for v in row]))
            assert tax_function[k].dtype == np.float64
        assert sorted(classifiers) == sorted(meta["classifiers"])
        for k, row in meta["classifiers"].items():
            if isinstance(row, str):
                assert classifiers[k] == "no_negative_examples"
            else:
                assert np.array_equal(classifiers[k], np.array([float(v) for v in row]))
        data = open(hmm, "rb").read()
        assert len(data) == meta["hmm_len"] and hashlib.sha1(data).hexdigest() == meta["hmm_sha1"]
    finally:
        os.remove(hmm)


@needs_reader
def test_load_db_errors_and_dump(tmp_path):
    with pytest.raises(ValueError, match="constructed on genes"):
        databases.load_db(TINY, protein_fasta_input="x.faa")
    r = databases.load_db(TINY, protein_fasta_input="x.faa", aligned_sequences="y")   # check skipped with -s
    os.remove(r[0])
    with pytest.raises(OSError):
        databases.load_db(str(tmp_path / "missing.stagDB"))
    out = tmp_path / "dump"
    out.mkdir()
    databases.load_db(TINY, dir_output=str(out))
    assert sorted(os.listdir(out)) == ["classifiers_weights.tsv", "hmmfile.hmm", "node_hierarchy.tsv",
                                       "parameters.tsv", "taxonomy_function.tsv"]
    assert (out / "parameters.tsv").read_text().splitlines()[1] == "Use proteins for the alignment: False"


def test_load_genome_db_layout(tmp_path):
    src = tmp_path / "src"
    src.mkdir()
    (src / "threshold_file.tsv").write_text("COG0012\t210\nCOG0016\t240\n")
    (src / "hmm_lengths_file.tsv").write_text("COG0012\t145\nCOG0016\t150\n")
    for f in ("COG0012", "COG0016", "concatenated_genes_STAG_database.HDF5"):
        (src / f).write_bytes(b"x")
    tar = tmp_path / "g.stagDB"
    with tarfile.open(tar, "w:gz") as t:
        for f in os.listdir(src):
            t.add(src / f, arcname=f)
    files, dirpath, thr, order, lens, concat = databases.load_genome_DB(str(tar))
    assert sorted(files) == ["COG0012", "COG0016"] and order == ["COG0012", "COG0016"]
    assert thr == {"COG0012": "210", "COG0016": "240"} and lens["COG0016"] == "150"
    assert concat == os.path.join(dirpath, "concatenated_genes_STAG_database.HDF5")
    import shutil
    shutil.rmtree(dirpath)
    os.remove(src / "threshold_file.tsv")
    with tarfile.open(tar, "w:gz") as t:
        for f in os.listdir(src):
            t.add(src / f, arcname=f)
    with pytest.raises(ValueError, match="threshold_file.tsv is missing"):
        databases.load_genome_DB(str(tar))


@needs_reader
def test_packed_table_cache_round_trip(tmp_path, monkeypatch):
    """classify() packs a .stagDB once; the second load comes from the .npz and is identical."""
    import shutil
    from stag_amd import table_cache, tables
    monkeypatch.setenv("STAG_AMD_CACHE", str(tmp_path / "cache"))
    db = tmp_path / "copy.stagDB"
    shutil.copy(TINY, db)
    assert table_cache.load(str(db)) is None
    hmm, use_cmalign, taxonomy, tax_function, classifiers, version = databases.load_db(str(db))
    t = tables.pack_db(taxonomy, tax_function, classifiers)
    assert table_cache.store(str(db), t, hmm, use_cmalign, False, version)
    hmm2, cm2, t2, v2 = table_cache.load(str(db))
    assert open(hmm2).read() == open(hmm).read() and cm2 == use_cmalign and v2 == version
    assert t2.names == t.names and t2.keys == t.keys and t2.n_features == t.n_features
    for f in ("child_first", "row_kind", "intercept", "row_off", "row_feat", "row_w", "tf"):
        a, b = getattr(t, f), getattr(t2, f)
        assert a.dtype == b.dtype and np.array_equal(a, b), f
    with pytest.raises(ValueError, match="constructed on genes"):       # databases.py:56-59 still applies
        table_cache.load(str(db), protein_fasta_input="x.faa")
    os.utime(db, ns=(1, 1))                                              # a changed file is a different entry
    assert table_cache.load(str(db)) is None
    monkeypatch.setenv("STAG_AMD_CACHE", "off")
    assert table_cache.cache_dir() is None and not table_cache.store(str(db), t, hmm, use_cmalign, False, version)
    os.remove(hmm)
    os.remove(hmm2)
