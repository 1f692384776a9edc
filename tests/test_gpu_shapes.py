"""GPU: the three BASELINE.json shapes that round 2 left without a run of their own.

  C3  through the PROTEIN entry point (stag_classify_protein_a2m: back-translation, align.py:63-94, fused in front
      of the encode kernel) at 350 aa = 1,050 nucleotide columns, 10^5 (protein alignment, gene) pairs with stop
      codons, insert residues, gap residues and wrong gene lengths;
  C5  the 50k-node database with every node that can be on the matrix-core walk (STAG_WALK=mfma);
  C4  `classify_genome -G` end to end: 30 marker-gene databases of the C3 width + the concatenated database
      (L = 157,500) in train_genome's tar layout, 1,000 genomes that each hold a gene with p = 0.9.
"""
import json
import os

import numpy as np
import pytest

import bench_workload as bw
import stagdb_writer
from oracle import oracle_c
from stag_amd import _lib, tables
from stag_amd import classify_genome as scg
from stag_amd.engine import Database
from test_gpu_fullsize import dense_from_text, run

gpu = pytest.mark.gpu
AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def protein_pairs(wl, seed, n):
    """(protein alignment, gene) pairs whose back-translation is known by construction: the nucleotide alignment
    `want[i]` (None where align.py:66-74 rejects the pair).  Full-length reads of the C3 database, some codons
    gapped out ('-' in the protein alignment, no nucleotides in the gene), 0-2 insert residues (lower case: three
    nucleotides of the gene that the alignment skips), a stop codon behind every second gene, one gene in a
    hundred a nucleotide too long or too short."""
    rng = np.random.default_rng(seed)
    text, offs, leaf = bw.make_reads(wl, seed=seed, n=n, window=wl.ncol, max_ins=0)
    naa = wl.ncol // 3
    nt = text[:offs[-1]].reshape(n, wl.ncol).copy()
    gap = rng.random((n, naa)) < 0.04                       # codons that are alignment gaps
    nt.reshape(n, naa, 3)[gap] = ord("-")
    prots, genes, want = [], [], []
    res = AA[rng.integers(0, 20, size=(n, naa))]
    res[gap] = ord("-")
    ins_nt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        p = res[i].tobytes()
        g_parts = nt[i].reshape(naa, 3)[~gap[i]].tobytes()
        k = int(rng.integers(0, 3))
        if k:                                               # insert residues: (position in the alignment, residue)
            for at in sorted(rng.integers(0, naa + 1, size=k).tolist(), reverse=True):
                n_before = int((~gap[i, :at]).sum())        # residues before it = codons of the gene before it
                p = p[:at] + bytes([int(rng.integers(97, 123))]) + p[at:]
                g_parts = g_parts[:3 * n_before] + ins_nt[rng.integers(0, 4, size=3)].tobytes() + g_parts[3 * n_before:]
        if i % 2:
            g_parts += b"TAA"
        expect = nt[i].tobytes().decode()
        if i % 100 == 17:
            g_parts = g_parts[:-1] if i % 200 == 17 else g_parts + b"A"
            expect = None
        prots.append(p.decode())
        genes.append(g_parts.decode())
        want.append(expect)
    return prots, genes, want, leaf


@gpu
def test_c3_protein_entry_point_at_350_residues():
    wl = bw.config_c3()
    t = wl.tables
    n = 100_000
    prots, genes, want, leaf = protein_pairs(wl, seed=33, n=n)
    ok = np.array([w is not None for w in want])
    assert 900 < (~ok).sum() < 1100
    pt, po = tables.pack_text(prots)
    gt, go = tables.pack_text(genes)
    with Database(t) as db:
        got = db.classify_protein_a2m_text(pt, po, gt, go)
        assert np.array_equal(got.status == _lib.SEQ_BACKTRANSLATE, ~ok)
        assert (got.status[ok] == 0).all() and (got.n_char[ok] == wl.ncol).all()
        # the nucleotide path on the alignments the pairs were built from: same bits in every field
        nuc_text, nuc_offs = tables.pack_text([w or "" for w in want])
        ref = db.classify_a2m_text(nuc_text, nuc_offs)
        for f in ("n_char", "n_aligned", "path_len", "selected", "path_node", "path_prob", "level_prob"):
            assert np.array_equal(getattr(got, f)[ok], getattr(ref, f)[ok]), f
        # and the host back-translation (pinned to the reference's by tests/golden/backtranslate.json) agrees
        from stag_amd.align import protein2gene_alignment
        import contextlib, io
        rng = np.random.default_rng(1)
        pick = rng.choice(n, size=3000, replace=False)
        with contextlib.redirect_stderr(io.StringIO()):
            for i in pick:
                assert protein2gene_alignment("x", prots[i], genes[i], True) == want[i], i
        # reads mostly come back to the leaf they were drawn from
        assert (got.path_node[ok, db.max_depth - 1] == leaf[ok]).mean() > 0.2
        # oracle on a subsample
        pick = rng.choice(np.flatnonzero(ok), size=400, replace=False)
        X = np.stack([oracle_c.encode_a2m(want[i])[0] for i in pick])
        out = oracle_c.DenseDB.from_packed(t).classify(X)
        assert np.array_equal(out["path_node"], got.path_node[pick])
        np.testing.assert_allclose(got.path_prob[pick], out["path_prob"], rtol=1e-9, atol=0)
        np.testing.assert_allclose(got.level_prob[pick], out["level_prob"], rtol=1e-9, atol=0)
        assert np.array_equal(out["selected"], got.selected[pick])
        assert np.array_equal(out["n_aligned"], got.n_aligned[pick])


@gpu
def test_c5_50k_nodes_on_the_matrix_core_walk(monkeypatch):
    """BASELINE configs[4], the dense half: every node with 2..16 scored children goes through k_walk_mfma (in auto
    mode this 2 % database stays with the gather walk, which is what test_gpu_configs.py runs)."""
    wl = bw.config_c5()
    t = wl.tables
    n = 120_000
    text, offs, leaf = bw.make_reads(wl, seed=56, n=n)
    monkeypatch.setenv("STAG_WALK", "gather")
    with Database(t) as db:
        ref = run(db, text, offs)
    monkeypatch.setenv("STAG_WALK", "mfma")
    with Database(t) as db:
        got = run(db, text, offs)
        assert (got.status == 0).all() and (got.path_len == 7).all()
        perm = np.random.default_rng(3).permutation(n)[:40_000]
        again = run(db, text, offs, perm)
    assert np.array_equal(got.path_node, ref.path_node) and np.array_equal(got.selected, ref.selected)
    np.testing.assert_allclose(got.path_prob, ref.path_prob, rtol=1e-9, atol=0)
    assert not np.array_equal(got.path_prob, ref.path_prob)              # (they are different kernels)
    for f in ("path_node", "path_prob", "level_prob", "selected"):
        assert np.array_equal(getattr(again, f), getattr(got, f)[perm]), f
    rng = np.random.default_rng(6)
    pick = rng.choice(n, size=1500, replace=False)
    out = oracle_c.DenseDB.from_packed(t).classify(dense_from_text(text, offs, pick, t.n_features))
    assert np.array_equal(out["path_node"], got.path_node[pick])
    np.testing.assert_allclose(got.path_prob[pick], out["path_prob"], rtol=1e-9, atol=0)
    np.testing.assert_allclose(got.level_prob[pick], out["level_prob"], rtol=1e-9, atol=0)
    assert np.array_equal(out["selected"], got.selected[pick])


def dicts_of(t):
    """(taxonomy, tax_function, classifiers) as load_db returns them, from packed tables (dense fp64 rows)."""
    taxonomy = bw.taxonomy_dict(t)
    tax_function = bw.tax_function_dict(t)
    classifiers = {}
    rows = bw.dense_rows(t)
    cf = t.child_first
    for p in range(t.n_nodes):
        for c in range(cf[p], cf[p + 1]):
            classifiers[t.names[c]] = rows[c] if t.row_kind[c] == 1 else "no_negative_examples"
    return taxonomy, tax_function, classifiers


needs_writer = pytest.mark.skipif(not stagdb_writer.available(),
                                  reason="tests/csrc/libstagdb_writer.so not built (make -C oracle)")


@gpu
@needs_writer
def test_c4_classify_genome_with_30_marker_genes(tmp_path, monkeypatch):
    genome_run(tmp_path, monkeypatch, n_genes=30, gcol=1050, n_genomes=1000, n_oracle=25, rtol=1e-9)


def genome_run(tmp_path, monkeypatch, n_genes, gcol, n_genomes, n_oracle, rtol):
    from test_gpu_configs import stand_in_aligner
    rng = np.random.default_rng(44)
    genes = ["COG%04d" % (12 + 7 * i) for i in range(n_genes)]
    # ---- the databases: 30 of the C3 width with small trees, and the concatenated one
    gene_wl, gene_db_path = {}, {}
    for i, g in enumerate(genes):
        wl = bw.make_db(400 + i, gcol, [3, 7, 16, 30], 0.02, name=g)
        wl.tables.names[1:] = ["%s_n%d" % (g, j) for j in range(1, wl.tables.n_nodes)]
        gene_wl[g] = wl
        gene_db_path[g] = str(tmp_path / (g + ".stagDB"))
        stagdb_writer.write_stagdb(gene_db_path[g], *dicts_of(wl.tables))
    cwl = bw.make_db(499, n_genes * gcol, [3, 6, 12, 24, 40], 0.004, name="concat")
    concat_path = str(tmp_path / "concat.stagDB")
    stagdb_writer.write_stagdb(concat_path, *dicts_of(cwl.tables))
    genome_db = str(tmp_path / "gtdb_30_like.stagDB")
    stagdb_writer.write_genome_db(genome_db, gene_db_path, concat_path, {g: "60.0" for g in genes},
                                  {g: str(5 * gcol) for g in genes})
    # ---- the genomes: a leaf of the concatenated tree each; gene g of a genome = that leaf's consensus over the
    # gene's columns with substitutions, present with p = 0.9
    ctext, coffs, cleaf = bw.make_reads(cwl, seed=45, n=n_genomes, window=n_genes * gcol, max_ins=0)
    whole = ctext[:coffs[-1]].reshape(n_genomes, n_genes * gcol)
    present = rng.random((n_genomes, n_genes)) < 0.9
    present[:, 0] |= ~present.any(axis=1)
    genome_ids = ["GENOME%05d" % j for j in range(n_genomes)]
    mgs = {}
    for i, g in enumerate(genes):
        rows = np.flatnonzero(present[:, i])
        ids = ["%s_%d##%s" % (genome_ids[j], i + 1, g) for j in rows]
        (tmp_path / (g + "_genes.fna")).write_text("".join(">%s\nACGT\n" % s for s in ids))
        block = whole[rows, i * gcol:(i + 1) * gcol]
        (tmp_path / (g + ".fna.a2m")).write_text("".join(">%s\n%s\n" % (s, block[r].tobytes().decode())
                                                          for r, s in enumerate(ids)))
        mgs[g] = [str(tmp_path / (g + "_genes.fna")), "no_protein"]
    stand_in_aligner(tmp_path, monkeypatch, "cat %s/$(basename $2).a2m" % tmp_path)
    spec = tmp_path / "mgs.json"
    spec.write_text(json.dumps(mgs))
    # ---- fused path (rows concatenated on the GPU) and the reference's text route give the same file
    out = tmp_path / "out"
    scg.classify_genome(genome_db, marker_genes=[str(spec)], output=str(out), verbose=1, long_out=True)
    monkeypatch.setattr(scg, "concat_on_device", lambda *a: None)
    out_text = tmp_path / "out_text"
    scg.classify_genome(genome_db, marker_genes=[str(spec)], output=str(out_text), verbose=1, long_out=True)
    annotation = (out / "genome_annotation").read_text()
    assert annotation == (out_text / "genome_annotation").read_text()
    lines = annotation.splitlines()
    assert len(lines) == 1 + n_genomes
    # genes_predictions: one file per genome, one line per gene it holds
    files = sorted(os.listdir(out / "genes_predictions"))
    assert files == sorted(genome_ids)
    for j in (0, 17, n_genomes - 1):
        assert len((out / "genes_predictions" / genome_ids[j]).read_text().splitlines()) == int(present[j].sum())
    # ---- oracle on a subsample of genome_annotation: the concatenated row = the genes' one-hot columns, zeros
    # where a gene is missing (classify_genome.py:375)
    import golden_util as gu
    from oracle import oracle_np
    taxonomy, tax_function = bw.taxonomy_dict(cwl.tables), bw.tax_function_dict(cwl.tables)
    classifiers = bw.LazyClassifiers(cwl.tables)
    by_id = {l.split("\t")[0]: l.split("\t") for l in lines[1:]}
    assert set(by_id) == set(genome_ids)
    for j in rng.choice(n_genomes, size=n_oracle, replace=False):
        x = oracle_c.encode_a2m(whole[j].tobytes().decode())[0]
        x &= np.repeat(present[j], 5 * gcol)
        want = oracle_np.classify_seq(genome_ids[j], x, taxonomy, tax_function, classifiers)
        gu.rows_match(by_id[genome_ids[j]], [str(v) for v in want], rtol=rtol)
    # a genome mostly comes back to the leaf it was drawn from
    hit = sum(by_id[genome_ids[j]][2].split("/")[-1] == cwl.tables.names[cleaf[j]] for j in range(n_genomes))
    assert hit > 0.25 * n_genomes            # (chance: one in the number of leaves)


@needs_writer
def test_classify_genome_small_genome_database_written_here(engine, tmp_path, monkeypatch):
    """The same scenario at 4 genes x 90 columns x 40 genomes, on the oracle-backed engine without a GPU and on the
    kernels with one: the database writer of the tests, the tar layout, the ids and the two concatenation routes."""
    from conftest import RTOL
    genome_run(tmp_path, monkeypatch, n_genes=4, gcol=90, n_genomes=40, n_oracle=40, rtol=RTOL[engine])
