"""The BASELINE.json configurations that round 1 left without a test on the real kernels (C1 and C4 also run
on the oracle-backed engine without a GPU: the `engine` fixture of conftest.py).

  C1  the reference's only known-answer check (stag_test_short.py:197-280): the 50 bundled test
      sequences, classified with a database trained on them, must come back with their true lineage.
      tests/golden/c1.stagDB was trained by the reference's own train_all_classifiers_nonmp on those
      sequences (oracle/make_golden.py::gen_c1; hmmalign is not installed, so the alignment is the fixed
      left-justified one, replayed here through stand-in aligner scripts; rank-cut rows are fixed ones).
  C4  `classify_genome -G` end to end on a genome database in the tar layout of train_genome.py:46-60
      whose HDF5 members the reference's save_to_file wrote (genome3.stagDB), against the rows the
      reference's classify_seq / concat_alignments / merge_gene_predictions gave (genome_db.json).
  C5  the 50k-node database at its stated width (L = 7,500), oracle on a subsample.
"""
import json
import os

import numpy as np
import pytest

import bench_workload as bw
import golden_util as gu
from oracle import oracle_c
from stag_amd import classify as sc
from stag_amd import classify_genome as scg
from stag_amd import databases
from stag_amd.engine import Database
from conftest import RTOL
from test_gpu_fullsize import dense_from_text, run

needs_h5 = pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built")


def stand_in_aligner(tmp_path, monkeypatch, replay_cmd):
    """hmmalign / esl-reformat scripts that replay a prepared A2M stream (neither tool is installed)."""
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\n%s\n" % replay_cmd)
    (bindir / "esl-reformat").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
    for t in ("hmmalign", "esl-reformat"):
        os.chmod(bindir / t, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])


@needs_h5
def test_c1_bundled_sequences_get_their_true_lineage(engine, tmp_path, monkeypatch):
    g = gu.load("c1.json")
    a2m = tmp_path / "c1.a2m"
    a2m.write_text("".join(">%s\n%s\n" % (i, a) for i, a in zip(g["ids"], g["aligned"])))
    stand_in_aligner(tmp_path, monkeypatch, "cat %s" % a2m)
    fasta = tmp_path / "sequences.fasta"
    fasta.write_text("".join(">%s\n%s\n" % (i, s) for i, s in zip(g["ids"], g["sequences"])))
    out = tmp_path / "c1.tsv"
    saved = tmp_path / "c1.ali"
    n, rows = sc.classify(os.path.join(gu.GOLDEN, "c1.stagDB"), fasta_input=str(fasta), output=str(out), long_out=True,
                          verbose=1, save_ali_to_file=str(saved))
    assert n == 145 and len(rows) == 50 and g["n_trained"] >= 10
    # the known-answer check of stag_test_short.py:237-280: predicted lineage == true lineage, all 50
    assert [r[1] for r in rows] == g["true_lineage"]
    text = out.read_text().splitlines()
    assert [l.split("\t")[1] for l in text[1:]] == g["true_lineage"]
    # and every field is what the reference's classify_seq gave with the same database
    for got, want in zip(rows, g["reference_rows"]):
        want = list(want)
        want[3] = type(got[3])(want[3])
        gu.rows_match([str(v) if i != 3 else v for i, v in enumerate(got)], want, rtol=RTOL[engine])
    assert len(saved.read_text().splitlines()) == 50


@needs_h5
def test_classify_genome_on_a_genome_stagdb_end_to_end(engine, tmp_path, monkeypatch):
    g = gu.load("genome_db.json")
    order = g["gene_order"]
    mgs = {}
    for mg in order:
        fna = tmp_path / (mg + "_genes.fna")
        fna.write_text("".join(">%s\nACGT\n" % sid for sid, _ in g["a2m"][mg]))
        (tmp_path / (mg + ".fna.a2m")).write_text("".join(">%s\n%s\n" % (sid, s) for sid, s in g["a2m"][mg]))
        mgs[mg] = [str(fna), "no_protein"]
    # classify_genome stores the marker genes as MG_sequences/<gene>.fna first: the replay is found by that name
    stand_in_aligner(tmp_path, monkeypatch, "cat %s/$(basename $2).a2m" % tmp_path)
    spec = tmp_path / "mgs.json"
    spec.write_text(json.dumps(mgs))
    out = tmp_path / "out"
    scg.classify_genome(os.path.join(gu.GOLDEN, "genome3.stagDB"), marker_genes=[str(spec)], output=str(out), verbose=1,
                        long_out=True)
    # per-gene predictions: the files (names and text) the reference's merge_gene_predictions wrote
    assert {f: (out / "genes_predictions" / f).read_text() for f in os.listdir(out / "genes_predictions")} == g["merged_files"]
    # MG_ali/<gene> is what the reference's -S writes, so its concatenation is the reference's text
    path = scg.concat_alignments(None, str(out / "MG_ali"), order, {mg: str(5 * c) for mg, c in g["ncols"].items()})
    assert open(path).read() == g["concatenated"]
    os.remove(path)
    # genome_annotation: the reference's classify_seq rows for the concatenated alignments
    lines = (out / "genome_annotation").read_text().splitlines()
    assert lines[0].split("\t") == sc.OUT_HEADER and len(lines) == 1 + len(g["genome_rows"])
    for line, want in zip(lines[1:], g["genome_rows"]):
        gu.rows_match(line.split("\t"), want, rtol=RTOL[engine])
    assert sorted(os.listdir(out)) == ["MG_ali", "MG_sequences", "genes_predictions", "genome_annotation"]
    # the text round trip (the reference's own route) gives the same file byte for byte
    monkeypatch.setattr(scg, "concat_on_device", lambda *a: None)
    out2 = tmp_path / "out_text"
    scg.classify_genome(os.path.join(gu.GOLDEN, "genome3.stagDB"), marker_genes=[str(spec)], output=str(out2), verbose=1,
                        long_out=True)
    assert (out2 / "genome_annotation").read_text() == (out / "genome_annotation").read_text()


@pytest.mark.gpu
def test_c5_50k_nodes_at_full_width():
    """BASELINE configs[4]: ~50k nodes, L = 7,500 (round 1 ran this tree only with 60 columns)."""
    wl = bw.config_c5()
    t = wl.tables
    assert t.n_nodes > 49_000 and t.n_features == 7500
    n = 300_000
    text, offs, leaf = bw.make_reads(wl, seed=55, n=n)
    with Database(t) as db:
        assert db.max_depth == 7
        res = run(db, text, offs)
        assert (res.status == 0).all() and (res.path_len == 7).all()
        cf = t.child_first.astype(np.int64)
        prev = np.zeros(n, dtype=np.int64)
        for d in range(db.max_depth):
            node = res.path_node[:, d].astype(np.int64)
            assert ((node >= cf[prev]) & (node < cf[prev + 1])).all()
            prev = node
        assert len(np.unique(res.path_node[:, 6])) > 1000
        rng = np.random.default_rng(5)
        idx = rng.permutation(np.repeat(rng.integers(0, n, size=30_000), 2))
        sub = run(db, text, offs, idx)
        for f in ("path_node", "path_prob", "level_prob", "selected", "n_aligned"):
            assert np.array_equal(getattr(sub, f), getattr(res, f)[idx]), f
        pick = rng.choice(n, size=2000, replace=False)
        out = oracle_c.DenseDB.from_packed(t).classify(dense_from_text(text, offs, pick, t.n_features))
        assert np.array_equal(out["path_node"], res.path_node[pick])
        np.testing.assert_allclose(res.path_prob[pick], out["path_prob"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(res.level_prob[pick], out["level_prob"], rtol=1e-9, atol=0)
        assert np.array_equal(out["selected"], res.selected[pick])
