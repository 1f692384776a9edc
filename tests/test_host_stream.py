"""stag_amd.stream: an A2M file cut into chunks, scanned by a pool of threads, classified batch by batch and written
as the reference's two-column output -- against the single-threaded reader and the per-row formatter, on the
oracle-backed engine (no GPU) and, with a GPU, on the kernels."""
import io
import os

import numpy as np
import pytest

from stag_amd import ingest, parallel, stream
from stag_amd.engine import Database


def make_file(tmp_path, rng, ncol, n, width=60, crlf=False, non_ascii_at=None):
    recs = []
    for i in range(n):
        s = np.full(ncol, "-", dtype="<U1")
        a, b = sorted(rng.integers(0, ncol + 1, size=2))
        if i % 11:
            s[a:b] = rng.choice(list("ACGT-N"), size=b - a, p=[0.22, 0.22, 0.22, 0.22, 0.08, 0.04])
        s = list(s)
        for _ in range(int(rng.integers(0, 3))):
            s.insert(int(rng.integers(0, len(s) + 1)), str(rng.choice(list("acgt"))))
        recs.append("".join(s))
    nl = "\r\n" if crlf else "\n"
    ids = ["read_%d some text" % i for i in range(n)]
    if non_ascii_at is not None:
        ids[non_ascii_at] = "réad %d" % non_ascii_at
    text = "".join(">%s%s%s%s" % (sid, nl, nl.join(r[j:j + width] for j in range(0, len(r), width)), nl)
                   for sid, r in zip(ids, recs))
    path = tmp_path / "reads.a2m"
    path.write_bytes(text.encode("utf-8"))
    return str(path), ids, recs


def test_chunk_starts_are_header_lines(tmp_path):
    rng = np.random.default_rng(1)
    path, ids, recs = make_file(tmp_path, rng, 90, 400, width=17)
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    for chunk in (1 << 16, 1000, 64):
        starts = stream.chunk_starts(buf, chunk)
        assert starts[0] == 0 and starts[-1] == len(buf) and starts == sorted(set(starts))
        for s in starts[1:-1]:
            assert buf[s] == ord(">") and buf[s - 1] == ord("\n")


@pytest.mark.parametrize("threads,chunk_bytes,kw", [(1, 1 << 20, {}), (4, 1 << 16, {}), (3, 1 << 16, {"crlf": True}),
                                                    (4, 1 << 16, {"non_ascii_at": 37}), (2, 1 << 16, {"width": 7})])
def test_parallel_batches_equal_the_single_threaded_reader(tmp_path, threads, chunk_bytes, kw):
    rng = np.random.default_rng(2)
    path, ids, recs = make_file(tmp_path, rng, 150, 3000, **kw)
    got_ids, got = [], []
    n_batches = 0
    for b in stream.a2m_file_batches(path, threads=threads, chunk_bytes=chunk_bytes):
        n_batches += 1
        assert len(b.text) % 16 == 0 and b.lead_trail.shape == (b.n, 2)
        got_ids += b.ids()
        got += ["-" * int(b.lead_trail[i, 0]) + bytes(b.text[b.offs[i]:b.offs[i + 1]]).decode() + "-" * int(b.lead_trail[i, 1])
                for i in range(b.n)]
    assert n_batches >= (1 if chunk_bytes > 1 << 19 else 4)
    assert got_ids == ids and got == recs
    with open(path, "rb") as fh:
        ref = [(i, t) for bids, text, offs in ingest.fasta_batches(fh)
               for i, t in zip(bids, [bytes(text[offs[k]:offs[k + 1]]).decode() for k in range(len(bids))])]
    assert [r[0] for r in ref] == ids and [r[1] for r in ref] == recs


def test_ranks_scan_only_their_own_chunks(tmp_path):
    rng = np.random.default_rng(3)
    path, ids, recs = make_file(tmp_path, rng, 120, 2000)
    whole = list(stream.a2m_file_batches(path, threads=2, chunk_bytes=1 << 16))
    seen = {}
    for rank in range(3):
        for b, batch in enumerate(stream.a2m_file_batches(path, threads=2, chunk_bytes=1 << 16, world=3, rank=rank)):
            assert (batch is None) == (b % 3 != rank)
            if batch is not None:
                seen[b] = batch.ids()
    assert [seen[b] for b in sorted(seen)] == [w.ids() for w in whole]


def test_file_to_two_column_output(engine, tmp_path, monkeypatch):
    """classify_a2m_file == rows of the per-row formatter over the whole file, filter included."""
    from test_gpu_parity import random_db
    rng = np.random.default_rng(4)
    ncol = 150
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [2, 3], [1, 2]], 0.1)
    from stag_amd.tables import pack_db, pack_text
    tables = pack_db(taxonomy, tax_function, classifiers, 5 * ncol)
    if engine == "oracle":
        from fake_engine import OracleDatabase as DB
    else:
        DB = Database
    path, ids, recs = make_file(tmp_path, rng, ncol, 2500, non_ascii_at=900)
    pool = parallel.DevicePool(lambda d: DB(tables, device=d), [0])
    try:
        out = tmp_path / "out.tsv"
        length, lines = stream.classify_a2m_file(pool, path, output=str(out), threads=3, min_perc_state=30.0,
                                                 chunk_bytes=1 << 16)
        db = pool.first
        text, offs = pack_text(recs)
        res = db.classify_a2m_text(text, offs, min_perc_state=30.0)
        want = db.result_rows(res, ids, long_out=False)
    finally:
        pool.close()
    got = out.read_text(encoding="utf-8").splitlines()
    assert got[0] == "sequence\ttaxonomy"
    assert [l.split("\t") for l in got[1:]] == want and lines == len(want) and 0 < len(want) < len(ids)
    assert length == 5 * ncol


def _db_file(tmp_path, rng, ncol):
    import stagdb_writer
    from test_gpu_parity import random_db
    taxonomy, tax_function, classifiers = random_db(rng, ncol, [[4], [2, 3], [1, 2]], 0.1)
    path = str(tmp_path / "db.stagDB")
    stagdb_writer.write_stagdb(path, taxonomy, tax_function, classifiers, hmm_text="HMMER3/f\nLENG %d\n" % ncol)
    return path


@pytest.mark.parametrize("long_out", [False, True])
def test_classify_a2m_gives_classifys_output_file(engine, tmp_path, monkeypatch, long_out):
    """--a2m FILE: the same output file, byte for byte, as classify() reading that alignment from the aligner's pipe."""
    import stagdb_writer
    if not stagdb_writer.available():
        pytest.skip("tests/csrc/libstagdb_writer.so not built")
    from stag_amd import classify as sc
    from test_gpu_configs import stand_in_aligner
    rng = np.random.default_rng(6)
    ncol = 120
    db_path = _db_file(tmp_path, rng, ncol)
    path, ids, recs = make_file(tmp_path, rng, ncol, 1500)
    stand_in_aligner(tmp_path, monkeypatch, "cat %s" % path)
    fasta = tmp_path / "in.fna"
    fasta.write_text(">x\nACGT\n")
    ref_out, got_out = tmp_path / "ref.tsv", tmp_path / "got.tsv"
    n_ref, rows = sc.classify(db_path, fasta_input=str(fasta), output=str(ref_out), long_out=long_out, verbose=1,
                              min_perc_state=25, cache=False)
    n_got, lines = sc.classify_a2m(db_path, path, output=str(got_out), long_out=long_out, verbose=1, min_perc_state=25,
                                   threads=3, cache=False)
    assert got_out.read_bytes() == ref_out.read_bytes()
    assert (n_got, lines) == (n_ref, len(rows)) and 0 < lines < len(ids)
    # and through the command line
    from stag_amd.__main__ import main
    cli_out = tmp_path / "cli.tsv"
    main(["classify", "-d", db_path, "--a2m", path, "-o", str(cli_out), "-v", "1", "-m", "25"] + (["-l"] if long_out else []))
    assert cli_out.read_bytes() == ref_out.read_bytes()


A2M_WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
import stag_amd.classify as sc
from stag_amd import stream
from fake_engine import OracleDatabase
sc.Database = OracleDatabase
stream.CHUNK_BYTES = 1 << 16       # several chunks, so that both ranks own some
seen = []
real = stream._scan_chunk
def spy(index, *a, **k):
    seen.append(index)
    return real(index, *a, **k)
stream._scan_chunk = spy
ret = sc.classify_a2m({db!r}, {a2m!r}, output={out!r} + ".rank" + sys.argv[1], verbose=1, min_perc_state=25, threads=2, cache=False)
json.dump([list(ret), sorted(seen)], open({out!r} + ".ret" + sys.argv[1], "w"))
dist.destroy_process_group()
'''


def test_classify_a2m_world2_gloo_ranks_scan_their_own_chunks(tmp_path):
    """Two ranks (gloo, CPU engine): each scans only the chunks it owns (it seeks; the other rank's chunks are never
    parsed), rows come back to rank 0 in file order, and rank 0's file is the single-process file."""
    import json
    import socket
    import subprocess
    import sys
    import stagdb_writer
    if not stagdb_writer.available():
        pytest.skip("tests/csrc/libstagdb_writer.so not built")
    from stag_amd import classify as sc
    from fake_engine import OracleDatabase
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(8)
    ncol = 120
    db_path = _db_file(tmp_path, rng, ncol)
    path, ids, recs = make_file(tmp_path, rng, ncol, 2500)
    single = tmp_path / "single.tsv"
    real_db, real_chunk = sc.Database, stream.CHUNK_BYTES
    sc.Database, stream.CHUNK_BYTES = OracleDatabase, 1 << 16
    try:
        n1, lines1 = sc.classify_a2m(db_path, path, output=str(single), verbose=1, min_perc_state=25, threads=2, cache=False)
    finally:
        sc.Database, stream.CHUNK_BYTES = real_db, real_chunk
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dist")
    code = A2M_WORKER.format(root=root, port=port, db=db_path, a2m=path, out=out)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=180)
        assert p.returncode == 0, err.decode()[-2000:]
    assert open(out + ".rank0").read() == single.read_text()
    assert not os.path.exists(out + ".rank1")
    (ret0, seen0), (ret1, seen1) = json.load(open(out + ".ret0")), json.load(open(out + ".ret1"))
    assert ret0 == [n1, lines1] and ret1[1] == 0
    assert seen0 and seen1 and all(b % 2 == 0 for b in seen0) and all(b % 2 == 1 for b in seen1)
    assert sorted(seen0 + seen1) == list(range(len(seen0) + len(seen1)))
