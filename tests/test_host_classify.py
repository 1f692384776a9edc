"""CPU: host logic of the drop-in (classify(), classify_genome pieces, rank sharding) with the
GPU engine replaced by the oracle-backed stand-in of tests/fake_engine.py.  The arithmetic is
NOT what is under test here (tests/test_gpu_*.py do that on the real kernels); file formats,
output text, return values, error behaviour and the world_size-2 sharding are.

The tests that drive classify() through the aligner pipe, the protein path, -S and classify_genome
take the `engine` fixture: "oracle" is the stand-in (runs here), "gpu" leaves the real
stag_amd.engine.Database in place and is marked gpu, so `pytest -m gpu` runs the very same
scenarios -- stand-in aligner scripts, first-failure ordering, the -S text -- on the HIP kernels."""
import io
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import golden_util as gu
from fake_engine import OracleDatabase
from oracle import oracle_np
from stag_amd import classify as sc
from stag_amd import classify_genome as scg
from stag_amd import databases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = os.path.join(gu.GOLDEN, "tiny.stagDB")


@pytest.fixture
def fake_gpu(monkeypatch):
    monkeypatch.setattr(sc, "Database", OracleDatabase)


from conftest import RTOL      # the `engine` fixture lives there


def golden_tsv_setup(monkeypatch, tmp_path):
    g = gu.load("tsv.json")
    taxonomy, tax_function, classifiers = gu.db_from_json(g["db"])

    def fake_load_db(path, protein_fasta_input=None, aligned_sequences=None, dir_output=None):
        hmm = tempfile.NamedTemporaryFile(delete=False, mode="w")
        hmm.write("NA")
        hmm.close()
        return hmm.name, False, taxonomy, tax_function, classifiers, "0.8.2"

    monkeypatch.setattr(sc, "load_db", fake_load_db)
    ali = tmp_path / "ali.tsv"
    ali.write_text("\n".join(g["ali_lines"]) + "\n")
    return g, str(ali)


def test_classify_output_text_matches_reference(fake_gpu, monkeypatch, tmp_path, capsys):
    g, ali = golden_tsv_setup(monkeypatch, tmp_path)
    out = str(tmp_path / "short.tsv")
    ret = sc.classify("db", aligned_sequences=ali, output=out, verbose=1)
    assert open(out).read() == g["runs"]["short"]["file"]
    assert ret[0] == g["runs"]["short"]["return_len"] and ret[1] == g["runs"]["short"]["return_rows"]
    assert oct(os.stat(out).st_mode & 0o777) == "0o644"
    ret = sc.classify("db", aligned_sequences=ali, output=str(tmp_path / "long.tsv"), verbose=1, long_out=True)
    got = open(tmp_path / "long.tsv").read().splitlines()
    exp = g["runs"]["long"]["file"].splitlines()
    assert got[0] == exp[0] and len(got) == len(exp)
    for a, b in zip(got[1:], exp[1:]):
        gu.rows_match(a.split("\t"), b.split("\t"), rtol=1e-12)
    ret = sc.classify("db", aligned_sequences=ali, verbose=1, internal_call=True, long_out=True,
                      output=str(tmp_path / "none"))
    assert ret[1] == g["runs"]["internal"]["return_rows"] and not os.path.exists(tmp_path / "none")
    capsys.readouterr()
    sc.classify("db", aligned_sequences=ali, verbose=1)
    assert capsys.readouterr().out == g["runs"]["stdout_short"]["file"]


def test_classify_verbose_and_errors(fake_gpu, monkeypatch, tmp_path, capsys):
    g, ali = golden_tsv_setup(monkeypatch, tmp_path)
    sc.classify("db", aligned_sequences=ali, output=str(tmp_path / "o"), verbose=3)
    err = capsys.readouterr().err
    assert err.startswith("Load database: ") and "Classify sequences: " in err
    bad = tmp_path / "bad.tsv"
    bad.write_text(g["ali_lines"][0] + "\n" + "short\t0\t1\t0\n")
    with pytest.raises(ValueError, match="broadcast"):
        sc.classify("db", aligned_sequences=str(bad), output=str(tmp_path / "o2"), verbose=1)
    empty = tmp_path / "empty.tsv"
    empty.write_text("")
    ret = sc.classify("db", aligned_sequences=str(empty), output=str(tmp_path / "o3"), verbose=1)
    assert ret == (None, []) and open(tmp_path / "o3").read() == "sequence\ttaxonomy\n"


def test_classify_from_aligner_pipe_with_fake_tools(engine, monkeypatch, tmp_path):
    """fasta_input path: hmmalign | esl-reformat (stand-in scripts that replay a prepared A2M
    stream) -> native FASTA scanner -> encode + classify -> rows, and the -S file.  Expected
    values are the reference's own output for the same rows through -s (tsv.json)."""
    g, ali = golden_tsv_setup(monkeypatch, tmp_path)
    sym = "-TGCA"
    records = []
    for k, line in enumerate(g["ali_lines"]):
        gid, *f = line.split("\t")
        x = np.array(f, dtype=int).reshape(-1, 5)
        s = "".join(sym[int(np.argmax(col))] for col in x)
        s = s[:5] + "acg"[:k % 4] + s[5:]                  # lowercase inserts are dropped (align.py:47)
        wrapped = "\n".join(s[i:i + 7] for i in range(0, len(s), 7))
        records.append(">%s\n%s\n" % (gid, wrapped))
    a2m = tmp_path / "replay.a2m"
    a2m.write_text("".join(records))
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s\n" % a2m)
    (bindir / "esl-reformat").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
    for t in ("hmmalign", "esl-reformat"):
        os.chmod(bindir / t, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    fasta = tmp_path / "reads.fasta"
    fasta.write_text(">unused\nACGT\n")
    saved = tmp_path / "saved.tsv"
    out = tmp_path / "o.tsv"
    ret = sc.classify("db", fasta_input=str(fasta), output=str(out), verbose=1, save_ali_to_file=str(saved))
    assert open(out).read() == g["runs"]["short"]["file"]
    assert ret[0] == g["runs"]["short"]["return_len"] and ret[1] == g["runs"]["short"]["return_rows"]
    assert saved.read_text() == "\n".join(g["ali_lines"]) + "\n"
    assert saved.read_text().splitlines(True)[0] == g["save_ali_line"]
    # a failing aligner is reported as in align.py:170
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s\nexit 3\n" % a2m)
    with pytest.raises(ValueError, match="hmmalign/cmalign failed"):
        sc.classify("db", fasta_input=str(fasta), output=str(tmp_path / "o2"), verbose=1)


def test_classify_protein_database_with_fake_tools(engine, monkeypatch, tmp_path, capsys):
    """protein_fasta_input path: the protein alignment from the pipe is paired with the gene file
    (text mode, CRLF here), back-translated and classified; rows equal the nucleotide path's."""
    g, ali = golden_tsv_setup(monkeypatch, tmp_path)       # L = 80: 16 columns -> needs a multiple of 3
    taxonomy, tax_function, classifiers = gu.db_from_json(g["db"])
    rng = np.random.default_rng(4)
    L = 5 * 18
    classifiers = {k: (v if isinstance(v, str) else rng.normal(0, 1, size=1 + L)) for k, v in classifiers.items()}

    def fake_load_db(path, protein_fasta_input=None, aligned_sequences=None, dir_output=None):
        hmm = tempfile.NamedTemporaryFile(delete=False, mode="w")
        hmm.close()
        return hmm.name, False, taxonomy, tax_function, classifiers, "0.8.2"

    monkeypatch.setattr(sc, "load_db", fake_load_db)
    prots, genes, nucs = [], [], []
    for i in range(9):
        prot = "".join(rng.choice(list("MKV-"), size=6))
        prot = prot[:2] + "g" * (i % 2) + prot[2:]
        res = sum(c != "-" for c in prot)
        gene = "".join(rng.choice(list("ACGT"), size=3 * res + 3 * (i % 2)))
        prots.append(prot)
        genes.append(gene)
        nucs.append(oracle_np.protein2gene_alignment("r%d" % i, prot, gene))
    (tmp_path / "p.a2m").write_text("".join(">r%d\n%s\n" % (i, p) for i, p in enumerate(prots)))
    gene_file = tmp_path / "genes.fna"
    gene_file.write_bytes("".join(">r%d\r\n%s\r\n" % (i, s) for i, s in enumerate(genes)).encode())
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s\n" % (tmp_path / "p.a2m"))
    (bindir / "esl-reformat").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
    for t in ("hmmalign", "esl-reformat"):
        os.chmod(bindir / t, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    prot_file = tmp_path / "prot.faa"
    prot_file.write_text(">unused\nMKV\n")
    n, rows = sc.classify("db", fasta_input=str(gene_file), protein_fasta_input=str(prot_file), verbose=1,
                          output=str(tmp_path / "o"), long_out=True)
    assert n == L and len(rows) == 9
    for i, row in enumerate(rows):
        x, _ = oracle_np.convert_alignment(nucs[i])
        gu.rows_match(row, oracle_np.classify_seq("r%d" % i, x, taxonomy, tax_function, classifiers), rtol=RTOL[engine])
    # a gene of the wrong length: message of align.py:70-73, then the reference's TypeError
    gene_file.write_text("".join(">r%d\n%s\n" % (i, s + ("A" if i == 4 else "")) for i, s in enumerate(genes)))
    capsys.readouterr()
    saved = tmp_path / "partial.tsv"
    with pytest.raises(TypeError):
        sc.classify("db", fasta_input=str(gene_file), protein_fasta_input=str(prot_file), verbose=1, output=str(tmp_path / "o2"),
                    save_ali_to_file=str(saved))
    assert "length of genes/alignment is not correct" in capsys.readouterr().err
    # by then the reference has written the -S lines of the four sequences before it (classify.py:158-162)
    want = [oracle_np.onehot_line("r%d" % i, oracle_np.convert_alignment(nucs[i])[0]) for i in range(4)]
    assert saved.read_text() == "\n".join(want) + "\n"
    # identifiers that do not pair up: sys.exit(1) (align.py:152-154)
    gene_file.write_text("".join(">q%d\n%s\n" % (i, s) for i, s in enumerate(genes)))
    with pytest.raises(SystemExit):
        sc.classify("db", fasta_input=str(gene_file), protein_fasta_input=str(prot_file), verbose=1, output=str(tmp_path / "o3"))
    # both in one stream: the reference fails at whichever comes first (pair 4 before pair 6)
    gene_file.write_text("".join(">%s%d\n%s\n" % ("q" if i == 6 else "r", i, s + ("A" if i == 4 else ""))
                                 for i, s in enumerate(genes)))
    with pytest.raises(TypeError):
        sc.classify("db", fasta_input=str(gene_file), protein_fasta_input=str(prot_file), verbose=1, output=str(tmp_path / "o4"))


def test_classify_genome_fused_concat_equals_text_path(engine, monkeypatch, tmp_path):
    """classify_genome -G end to end (marker-gene DBs, MG_ali, genes_predictions, genome_annotation):
    the concatenated rows assembled from the packed marker-gene rows give byte for byte the
    genome_annotation of the reference's text round trip (concat_alignments, pinned by concat.json)."""
    rng = np.random.default_rng(12)
    ncols = {"COG0012": 6, "COG0016": 9, "COG0018": 3}
    order = list(ncols)

    def make_db(L):
        tax = {"tree_root": ["d_A", "d_B"], "d_A": ["p_A1", "p_A2"], "d_B": ["p_B1"], "p_A1": [], "p_A2": [], "p_B1": []}
        cls = {k: rng.normal(0, 1, size=1 + L) for k in ("d_A", "d_B", "p_A1", "p_A2")}
        cls["p_B1"] = "no_negative_examples"
        tf = {str(k): rng.normal(0, 1, size=3) for k in (-1, 0, 1)}
        return tax, tf, cls

    dbs = {mg: make_db(5 * c) for mg, c in ncols.items()}
    dbs["concat"] = make_db(5 * sum(ncols.values()))

    def fake_load_db(path, protein_fasta_input=None, aligned_sequences=None, dir_output=None):
        hmm = tempfile.NamedTemporaryFile(delete=False, mode="w")
        hmm.close()
        tax, tf, cls = dbs[os.path.basename(path)]
        return hmm.name, False, tax, tf, dict(cls), "0.8.2"

    def fake_load_genome_DB(database, tool_version, verbose):
        d = tempfile.mkdtemp()
        return ({mg: os.path.join(d, mg) for mg in order}, d, {mg: "10" for mg in order}, order,
                {mg: str(5 * c) for mg, c in ncols.items()}, os.path.join(d, "concat"))

    monkeypatch.setattr(sc, "load_db", fake_load_db)
    monkeypatch.setattr(scg, "load_genome_DB", fake_load_genome_DB)
    monkeypatch.setattr(scg.os.path, "isfile", lambda p: True if os.path.basename(p) in ncols else os.path.exists(p))
    # marker genes: genomeA has all three (two copies of COG0016: the last one wins), genomeB lacks COG0018,
    # genomeC only has COG0018
    have = {"COG0012": ["genomeA_1", "genomeB_4"], "COG0016": ["genomeA_2", "genomeB_9", "genomeA_7"],
            "COG0018": ["genomeC_3", "genomeA_5"]}
    mgs = {}
    for mg, ids in have.items():
        fna = tmp_path / (mg + ".fna")
        fna.write_text("".join(">%s##%s\nACGT\n" % (i, mg) for i in ids))
        (tmp_path / (mg + ".fna.a2m")).write_text("".join(
            ">%s##%s\n%s\n" % (i, mg, "".join(rng.choice(list("ACGT-"), size=ncols[mg]))) for i in ids))
        mgs[mg] = [str(fna), "no_protein"]
    bindir = tmp_path / "bin"
    bindir.mkdir()
    # classify_genome copies the marker genes to MG_sequences/ first: look the replay up by file name
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s/$(basename $2).a2m\n" % tmp_path)
    (bindir / "esl-reformat").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
    for t in ("hmmalign", "esl-reformat"):
        os.chmod(bindir / t, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    spec = tmp_path / "mgs.json"
    spec.write_text(json.dumps(mgs))

    fused = tmp_path / "fused"
    scg.classify_genome("genomedb", marker_genes=[str(spec)], output=str(fused), verbose=1, long_out=True)
    used = []
    real = scg.concat_on_device
    monkeypatch.setattr(scg, "concat_on_device", lambda *a: used.append(real(*a)) or None)
    text = tmp_path / "text"
    scg.classify_genome("genomedb", marker_genes=[str(spec)], output=str(text), verbose=1, long_out=True)
    src = used[0]
    assert src["ids"] == ["genomeA", "genomeB", "genomeC"]
    assert src["src_row"].tolist() == [[0, 2, 1], [1, 1, -1], [-1, -1, 0]]
    a = (fused / "genome_annotation").read_text()
    b = (text / "genome_annotation").read_text()
    assert a == b and len(a.splitlines()) == 4
    for sub in ("MG_ali", "genes_predictions"):
        for f in os.listdir(text / sub):
            assert (fused / sub / f).read_bytes() == (text / sub / f).read_bytes()
    if engine == "gpu":      # and what the kernels gave is what the oracle gives for the same files
        monkeypatch.setattr(sc, "Database", OracleDatabase)
        ref = tmp_path / "oracle"
        scg.classify_genome("genomedb", marker_genes=[str(spec)], output=str(ref), verbose=1, long_out=True)
        want = (ref / "genome_annotation").read_text().splitlines()
        assert a.splitlines()[0] == want[0] and len(want) == 4
        for got_line, want_line in zip(a.splitlines()[1:], want[1:]):
            gu.rows_match(got_line.split("\t"), want_line.split("\t"), rtol=RTOL["gpu"])
        for f in os.listdir(ref / "MG_ali"):
            assert (fused / "MG_ali" / f).read_bytes() == (ref / "MG_ali" / f).read_bytes()


def test_parse_onehot_line_matches_reference_reader():
    for line in ("id1\t0\t1\t0\t0\t1\n", "id2\t0\t10\t0\t-3\t1\n", "id3\t1\n", "x y\t0\t0\n"):
        gid, x = sc.parse_onehot_line(line)
        eid, ex = oracle_np.parse_onehot_line(line)
        assert gid == eid and np.array_equal(x, ex) and x.dtype == bool


@pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built")
def test_classify_on_reference_written_stagdb(fake_gpu, tmp_path):
    """Whole -s path from the HDF5 file the reference's writer produced."""
    rng = np.random.default_rng(9)
    hmm, _, taxonomy, tax_function, classifiers, _ = databases.load_db(TINY)
    os.remove(hmm)
    lines, exp = [], []
    for i in range(40):
        sym = rng.integers(0, 5, size=29)
        x = np.zeros(145, dtype=bool)
        x[5 * np.arange(29) + sym] = True
        lines.append("r%d\t%s" % (i, "\t".join("1" if b else "0" for b in x)))
        exp.append(oracle_np.classify_seq("r%d" % i, x, taxonomy, tax_function, classifiers))
    ali = tmp_path / "a.tsv"
    ali.write_text("\n".join(lines) + "\n")
    out = tmp_path / "o.tsv"
    n, rows = sc.classify(TINY, aligned_sequences=str(ali), output=str(out), long_out=True, verbose=1)
    assert n == 145 and len(rows) == 40
    for a, b in zip(rows, exp):
        gu.rows_match(a, b, rtol=1e-12)
    assert open(out).read().splitlines()[0].split("\t") == sc.OUT_HEADER


@pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built")
def test_classify_uses_the_packed_table_cache(fake_gpu, monkeypatch, tmp_path):
    monkeypatch.setenv("STAG_AMD_CACHE", str(tmp_path / "cache"))
    x = np.zeros(145, dtype=bool)
    x[5 * np.arange(29) + 2] = True
    ali = tmp_path / "a.tsv"
    ali.write_text("r0\t%s\n" % "\t".join("1" if b else "0" for b in x))
    first = sc.classify(TINY, aligned_sequences=str(ali), output=str(tmp_path / "o1"), long_out=True, verbose=1)
    assert len(os.listdir(tmp_path / "cache")) == 1

    def no_load(*a, **k):
        raise AssertionError("load_db called although the packed tables are cached")

    monkeypatch.setattr(sc, "load_db", no_load)
    second = sc.classify(TINY, aligned_sequences=str(ali), output=str(tmp_path / "o2"), long_out=True, verbose=1)
    assert first == second and open(tmp_path / "o1").read() == open(tmp_path / "o2").read()


def test_concat_and_merge_match_reference(tmp_path):
    g = gu.load("concat.json")
    ali = tmp_path / "MG_ali"
    ali.mkdir()
    for mg, lines in g["ali_files"].items():
        (ali / mg).write_text("\n".join(lines) + "\n")
    path = scg.concat_alignments(None, str(ali), g["gene_order"], g["ali_lengths"])
    assert open(path).read() == g["concatenated"]
    os.remove(path)
    # the files merge_gene_predictions of the reference wrote for these predictions (names and contents)
    scg.merge_gene_predictions(None, [], g["predictions"], 1, 1, str(tmp_path), False, False)
    outdir = tmp_path / "genes_predictions"
    assert {f.name: f.read_text() for f in outdir.iterdir()} == g["merged_files"]
    # the two functions do not agree on the genome of "plain.3##MG_4" (classify_genome.py:353 vs :372)
    assert "plain.3\t" in g["concatenated"] and "plain.3" not in g["merged_files"]


WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
import stag_amd.classify as sc
from fake_engine import OracleDatabase
sc.Database = OracleDatabase
sc.READER_BATCH = 7          # many small batches so both ranks own some
n, rows = sc.classify({db!r}, aligned_sequences={ali!r}, output={out!r} + ".rank" + sys.argv[1], long_out=True, verbose=1)
json.dump([n, rows], open({out!r} + ".ret" + sys.argv[1], "w"))
dist.destroy_process_group()
'''


@pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built")
def test_sharded_classify_world2_gloo(fake_gpu, tmp_path):
    """Two ranks (gloo, CPU): batches dealt round-robin, rows gathered to rank 0 in input order;
    the output file must equal the single-process one and only rank 0 writes it."""
    rng = np.random.default_rng(10)
    lines = []
    for i in range(60):
        x = np.zeros(145, dtype=bool)
        x[5 * np.arange(29) + rng.integers(0, 5, size=29)] = True
        lines.append("r%d\t%s" % (i, "\t".join("1" if b else "0" for b in x)))
    ali = tmp_path / "a.tsv"
    ali.write_text("\n".join(lines) + "\n")
    single = tmp_path / "single.tsv"
    n1, rows1 = sc.classify(TINY, aligned_sequences=str(ali), output=str(single), long_out=True, verbose=1)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = WORKER.format(root=ROOT, port=port, db=TINY, ali=str(ali), out=str(tmp_path / "dist"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-2000:]
    assert open(str(tmp_path / "dist") + ".rank0").read() == open(single).read()
    assert not os.path.exists(str(tmp_path / "dist") + ".rank1")
    n0, rows0 = json.load(open(str(tmp_path / "dist") + ".ret0"))
    assert n0 == n1 and [list(map(str, r)) for r in rows0] == [list(map(str, r)) for r in rows1]
    assert json.load(open(str(tmp_path / "dist") + ".ret1")) == [145, []]
    # two malformed rows, one in a batch of each rank (READER_BATCH = 7: r9 is in batch 1, rank 1; r16 in
    # batch 2, rank 0): both ranks stop with the error of the EARLIER one, as the reference's single loop would
    bad = list(lines)
    bad[16] = "r16\t0\t1"
    bad[9] = "r9\t1\t0\t0"
    ali.write_text("\n".join(bad) + "\n")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = WORKER.format(root=ROOT, port=port, db=TINY, ali=str(ali), out=str(tmp_path / "dist2"))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode != 0 and "ValueError" in err.decode() and "[sequence r9]" in err.decode(), err.decode()[-2000:]
    assert not os.path.exists(str(tmp_path / "dist2") + ".rank0")


CLI_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import stag_amd.classify as sc
from fake_engine import OracleDatabase
sc.Database = OracleDatabase
sc.READER_BATCH = 7
from stag_amd.__main__ import main
main(["classify", "-d", {db!r}, "-s", {ali!r}, "-o", {out!r}, "-v", "3", "-l"])
'''


@pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built")
def test_command_line_under_a_launcher_joins_the_process_group(fake_gpu, tmp_path):
    """`torchrun ... -m stag_amd classify`: the command line itself joins the process group from RANK /
    WORLD_SIZE / MASTER_* (nothing else on that path does), so the ranks share the batches and only rank 0
    writes the output file -- instead of every rank classifying everything and writing the same file."""
    rng = np.random.default_rng(11)
    lines = []
    for i in range(50):
        x = np.zeros(145, dtype=bool)
        x[5 * np.arange(29) + rng.integers(0, 5, size=29)] = True
        lines.append("r%d\t%s" % (i, "\t".join("1" if b else "0" for b in x)))
    ali = tmp_path / "a.tsv"
    ali.write_text("\n".join(lines) + "\n")
    single = tmp_path / "single.tsv"
    sc.classify(TINY, aligned_sequences=str(ali), output=str(single), long_out=True, verbose=1)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "dist.tsv"
    marker = tmp_path / "writes"
    code = CLI_WORKER.format(root=ROOT, db=TINY, ali=str(ali), out=str(out))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TMPDIR=str(tmp_path / ("tmp%d" % r)))
        os.makedirs(env["TMPDIR"])
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stderr=subprocess.PIPE))
    errs = []
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-2000:]
        errs.append(err.decode())
    assert out.read_text() == single.read_text()
    # "Classify sequences: ..." is printed by the rank that goes on to write the output (classify.py:164-166)
    assert "Classify sequences: " in errs[0] and "Classify sequences: " not in errs[1]
    # the output goes through a temp file in TMPDIR and a move: only rank 0's TMPDIR ever held one, and
    # neither rank left anything behind (HMM temp files included)
    assert os.listdir(tmp_path / "tmp0") == [] and os.listdir(tmp_path / "tmp1") == []


def test_command_line_classify_and_its_argument_errors(fake_gpu, monkeypatch, tmp_path, capsys):
    """`python -m stag_amd classify ...` (stag/__main__.py:393-422): flags, defaults, the output file,
    and the usage errors (exit status 1 with the reference's message)."""
    from stag_amd.__main__ import main
    g, ali = golden_tsv_setup(monkeypatch, tmp_path)
    db = tmp_path / "db.stagDB"
    db.write_text("placeholder: load_db is replaced")
    out = tmp_path / "cli.tsv"
    main(["classify", "-d", str(db), "-s", ali, "-o", str(out), "-v", "1"])
    assert out.read_text() == g["runs"]["short"]["file"]
    main(["classify", "-d", str(db), "-s", ali, "-o", str(out), "-v", "1", "-l"])
    assert out.read_text().splitlines()[0] == g["runs"]["long"]["file"].splitlines()[0]
    capsys.readouterr()
    for argv, message in ((["classify", "-d", str(db)], "missing <fasta_seqs> (-i) or <aligned_seq> (-s)"),
                          (["classify", "-s", ali], "missing <database> (-d)"),
                          (["classify", "-d", str(db), "-s", ali, "-t", "0"], "number of threads (-t) is less than 1"),
                          (["classify", "-d", str(db), "-s", ali, "-m", "101"], "-m should be between 0 and 100"),
                          (["classify_genome", "-d", str(db), "-o", str(tmp_path / "g")], "you need to provide at least -i, -D, or -G."),
                          (["classify_genome", "-d", str(db), "-G", ali, "-i", ali, "-o", str(tmp_path / "g")],
                           "options -i, -D, and -G are mutually exclusive"),
                          (["classify_genome", "-d", str(db), "-G", ali], "missing output directory (-o)")):
        with pytest.raises(SystemExit) as e:
            main(argv)
        assert e.value.code == 1 and "[E::main] Error: " + message in capsys.readouterr().err


@pytest.mark.skipif(databases._load_h5lib() is None, reason="libstag_h5.so not built")
def test_one_process_driving_several_devices(engine, monkeypatch, tmp_path):
    """STAG_DEVICES: one process, one aligner, one parser, a handle and a worker thread per device (here three
    handles on device 0), batches dealt in input order.  Output, return value and the -S file equal the
    single-device run's; so does the failure the reference would stop at, with the -S lines written before it."""
    from stag_amd import align as sa
    rng = np.random.default_rng(21)
    sym = "-TGCA"
    n = 83
    strings = ["".join(sym[int(v)] for v in rng.integers(0, 5, size=29)) for _ in range(n)]
    a2m = tmp_path / "replay.a2m"
    a2m.write_text("".join(">r%d\n%s\n" % (i, s[:9] + "ac"[: i % 3] + s[9:]) for i, s in enumerate(strings)))
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s\n" % a2m)
    (bindir / "esl-reformat").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
    for t in ("hmmalign", "esl-reformat"):
        os.chmod(bindir / t, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    monkeypatch.setattr(sa, "BATCH_RECORDS", 7)
    monkeypatch.setattr(sc, "READER_BATCH", 7)
    fasta = tmp_path / "reads.fasta"
    fasta.write_text(">unused\nACGT\n")

    def run(tag, devices):
        if devices:
            monkeypatch.setenv("STAG_DEVICES", devices)
        else:
            monkeypatch.delenv("STAG_DEVICES", raising=False)
        out, saved = tmp_path / (tag + ".tsv"), tmp_path / (tag + ".ali")
        ret = sc.classify(TINY, fasta_input=str(fasta), output=str(out), long_out=True, verbose=1, save_ali_to_file=str(saved))
        return ret, out.read_text(), saved.read_text()

    one = run("one", None)
    three = run("three", "0,0,0")
    assert three == one and len(one[0][1]) == n and one[0][0] == 145
    # the pre-aligned path (-s) through the same pool
    monkeypatch.setenv("STAG_DEVICES", "0,0")
    ret = sc.classify(TINY, aligned_sequences=str(tmp_path / "one.ali"), output=str(tmp_path / "s.tsv"), long_out=True, verbose=1)
    assert (tmp_path / "s.tsv").read_text() == one[1] and ret == one[0]
    # a record of the wrong length in the middle: the reference fails there (ValueError: operands could not be
    # broadcast), after writing the -S lines of everything before it
    bad = strings[:40] + [strings[40][:-1]] + strings[41:]
    a2m.write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(bad)))
    for devices in (None, "0,0,0"):
        if devices:
            monkeypatch.setenv("STAG_DEVICES", devices)
        else:
            monkeypatch.delenv("STAG_DEVICES", raising=False)
        saved = tmp_path / "partial.ali"
        with pytest.raises(ValueError, match="broadcast"):
            sc.classify(TINY, fasta_input=str(fasta), output=str(tmp_path / "never.tsv"), verbose=1, save_ali_to_file=str(saved))
        assert len(saved.read_text().splitlines()) == 40 and not os.path.exists(tmp_path / "never.tsv")


def test_protein_and_gene_streams_are_paired_record_by_record(engine, monkeypatch, tmp_path, capsys):
    """The two FASTA batchers cut their streams independently (a gene file with a line outside ASCII or a lone
    carriage return is re-parsed and lands in one batch whole): records must still pair up one to one, as
    the reference's zip() pairs them (align.py:148), and none may be dropped."""
    from stag_amd import align as sa
    g, ali = golden_tsv_setup(monkeypatch, tmp_path)
    taxonomy, tax_function, classifiers = gu.db_from_json(g["db"])
    rng = np.random.default_rng(44)
    L = 5 * 18
    classifiers = {k: (v if isinstance(v, str) else rng.normal(0, 1, size=1 + L)) for k, v in classifiers.items()}

    def fake_load_db(path, protein_fasta_input=None, aligned_sequences=None, dir_output=None):
        hmm = tempfile.NamedTemporaryFile(delete=False, mode="w")
        hmm.close()
        return hmm.name, False, taxonomy, tax_function, classifiers, "0.8.2"

    monkeypatch.setattr(sc, "load_db", fake_load_db)
    monkeypatch.setattr(sa, "BATCH_RECORDS", 5)
    n = 23
    prots, genes, nucs = [], [], []
    for i in range(n):
        prot = "".join(rng.choice(list("MKV-"), size=6))
        res = sum(c != "-" for c in prot)
        gene = "".join(rng.choice(list("ACGT"), size=3 * res))
        prots.append(prot)
        genes.append(gene)
        nucs.append(oracle_np.protein2gene_alignment("r%d" % i, prot, gene))
    ids = ["r%d%s" % (i, " café" if i == 3 else "") for i in range(n)]     # the whole header line is the id (helpers.py:112)
    (tmp_path / "p.a2m").write_bytes("".join(">%s\n%s\n" % (ids[i], p) for i, p in enumerate(prots)).encode("utf-8"))
    gene_file = tmp_path / "genes.fna"
    # record 3's header carries a character outside ASCII, record 11 ends in a lone carriage return
    gene_file.write_bytes("".join(">%s\n%s%s" % (ids[i], s, "\r" if i == 11 else "\n")
                                  for i, s in enumerate(genes)).encode("utf-8"))
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "hmmalign").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s\n" % (tmp_path / "p.a2m"))
    (bindir / "esl-reformat").write_text("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
    for t in ("hmmalign", "esl-reformat"):
        os.chmod(bindir / t, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    prot_file = tmp_path / "prot.faa"
    prot_file.write_text(">unused\nMKV\n")
    # the batchers do cut differently here
    with open(gene_file, "rb") as fh:
        gene_cuts = [len(ids) for ids, _, _ in ingest_batches(fh)]
    assert gene_cuts != [5, 5, 5, 5, 3] and sum(gene_cuts) == n
    length, rows = sc.classify("db", fasta_input=str(gene_file), protein_fasta_input=str(prot_file), verbose=1,
                               output=str(tmp_path / "o"), long_out=True)
    assert length == L and [r[0] for r in rows] == ids
    for i, row in enumerate(rows):
        x, _ = oracle_np.convert_alignment(nucs[i])
        gu.rows_match(row, oracle_np.classify_seq(ids[i], x, taxonomy, tax_function, classifiers), rtol=RTOL[engine])
    capsys.readouterr()


def ingest_batches(fh):
    from stag_amd import ingest
    return ingest.fasta_batches(fh, batch_records=5, text_mode=True)


def test_long_rows_of_a_batch_equal_the_per_row_formatter():
    """Database.result_rows formats the long fields of a whole batch in one pass (every float through repr once, the
    lineage strings once per end node); result_row is the per-row statement of classify.py:116-122.  Same rows,
    also with filtered sequences in between and with paths of different lengths."""
    import bench_workload as bw
    from fake_engine import OracleDatabase
    from stag_amd import _lib
    from stag_amd.tables import pack_bool_rows
    wl = bw.make_db(11, 40, [3, 9, 27, 50], 0.1, name="t")
    db = OracleDatabase(wl.tables)
    rng = np.random.default_rng(1)
    X = rng.random((3000, wl.tables.n_features)) < 0.2
    res = db.classify_bits(pack_bool_rows(X, db.words_per_row))
    res.status[::97] = _lib.SEQ_FILTERED
    ids = ["r%d" % i for i in range(len(X))]
    want = [db.result_row(res, i, ids[i], True) for i in range(res.n) if res.status[i] == _lib.SEQ_OK]
    assert db.result_rows(res, ids, True) == want and len(want) > 2900
    assert db.result_rows(res, ids, False) == [w[:2] for w in want]


def test_paired_batches_stay_full_when_the_batchers_cut_unevenly(monkeypatch, tmp_path):
    """aligned_batches for a protein database: one batcher cutting at 10 records, the other at 9, used to alternate
    batches of 9 and 1; both sides are now topped up before a pair is cut, so every batch but the last is full."""
    from stag_amd import align as sa
    from stag_amd.tables import pack_text

    def fake_batches(sizes, tag):
        at = 0
        for m in sizes:
            ids = ["r%d" % i for i in range(at, at + m)]
            text, offs = pack_text(["%s%d" % (tag, i) for i in range(at, at + m)])
            at += m
            yield ids, text, offs

    class Proc:
        stdout = "prot"

    monkeypatch.setattr(sa, "_start_aligner", lambda *a: (Proc(), Proc()))
    monkeypatch.setattr(sa, "_finish_aligner", lambda *a: None)
    n = 95
    monkeypatch.setattr(sa.ingest, "fasta_batches",
                        lambda stream, **kw: fake_batches([10] * 9 + [5], "P") if stream == "prot" else
                        fake_batches([9] * 10 + [5], "G"))
    gene_file = tmp_path / "g.fna"
    gene_file.write_text(">x\nA\n")
    got = list(sa.aligned_batches(str(gene_file), "prot.faa", "hmm", False, 1, 1, size=10))
    sizes = [len(b[0]) for b in got]
    assert sum(sizes) == n and all(s >= 5 for s in sizes[:-1]) and len(sizes) <= 11
    at = 0
    for ids, ptext, poffs, (gtext, goffs), lt in got:
        assert lt is None
        for j, sid in enumerate(ids):
            assert sid == "r%d" % at
            assert bytes(ptext[poffs[j]:poffs[j + 1]]).decode() == "P%d" % at
            assert bytes(gtext[goffs[j]:goffs[j + 1]]).decode() == "G%d" % at
            at += 1
