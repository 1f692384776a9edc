"""libstag_ingest.so (include/stag_ingest.h) against the reference's own readers.

tests/golden/ingest.json holds byte streams and what stag.helpers.read_fasta (helpers.py:107-121)
and stag.classify.alignment_reader (classify.py:16-20) returned for them (oracle/make_golden.py).
Every case is pushed through the native parsers at several chunk sizes, so record, line and
"\r\n" boundaries fall on chunk edges.  No GPU involved.
"""
import ctypes
import io
import os

import numpy as np
import pytest

from golden_util import load
from stag_amd import ingest
from stag_amd.classify import parse_onehot_line
from stag_amd.tables import pack_bool_rows, unpack_rows, words_per_row

G = load("ingest.json")


def test_library_exports_every_declared_symbol():
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "stag_ingest.h")).read()
    declared = set(re.findall(r"\b(stag_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(ingest.EXPORTS)
    L = ingest.lib()
    for name in declared:
        assert hasattr(L, name)
    assert ingest.RECORD.itemsize == 48


def records_of(batches):
    out = []
    for ids, text, offs in batches:
        assert len(text) % 16 == 0 and len(offs) == len(ids) + 1
        for i, sid in enumerate(ids):
            out.append([sid, bytes(text[offs[i]:offs[i + 1]]).decode("utf-8")])
    return out


@pytest.mark.parametrize("name", sorted(G["fasta"]))
@pytest.mark.parametrize("chunk", [1, 2, 3, 7, 64, 1 << 20])
def test_fasta_scan_matches_reference(name, chunk):
    case = G["fasta"][name]
    data = bytes.fromhex(case["hex"])
    got = records_of(ingest.fasta_batches(io.BytesIO(data), batch_records=1 << 16, chunk_bytes=chunk))
    assert got == case["records"]


@pytest.mark.parametrize("name", sorted(G["fasta_text"]))
@pytest.mark.parametrize("chunk", [1, 3, 64, 1 << 20])
def test_fasta_scan_text_mode_matches_reference(name, chunk):
    """The gene file of a protein database is read in text mode (align.py:150): '\r' ends a line."""
    case = G["fasta_text"][name]
    data = bytes.fromhex(case["hex"])
    got = records_of(ingest.fasta_batches(io.BytesIO(data), chunk_bytes=chunk, text_mode=True))
    assert got == case["records"]


@pytest.mark.parametrize("batch", [1, 2, 5, 299, 300, 301])
def test_fasta_scan_batch_limits(batch):
    case = G["fasta"]["many_wrapped"]
    data = bytes.fromhex(case["hex"])
    batches = list(ingest.fasta_batches(io.BytesIO(data), batch_records=batch, chunk_bytes=997))
    assert all(len(b[0]) <= batch for b in batches)
    assert records_of(batches) == case["records"]


def rows_of(path, L, chunk, batch_rows=7):
    W = words_per_row(L)
    out = []
    for ids, bits, n_fields in ingest.onehot_batches(path, L, W, batch_rows=batch_rows, chunk_bytes=chunk,
                                                     slow_line=parse_onehot_line):
        dense = unpack_rows(bits, L)
        for i, gid in enumerate(ids):
            out.append((gid, int(n_fields[i]), dense[i].astype(int).tolist() if n_fields[i] == L else None))
    return out


@pytest.mark.parametrize("name", sorted(G["onehot"]))
@pytest.mark.parametrize("chunk", [1, 2, 5, 64, 1 << 20])
def test_onehot_parse_matches_reference(tmp_path, name, chunk):
    case = G["onehot"][name]
    path = tmp_path / "in.tsv"
    path.write_bytes(bytes.fromhex(case["hex"]))
    if case["raises"]:
        with pytest.raises(ValueError):      # int() on a field that is not an integer
            rows_of(str(path), 3, chunk)
        return
    ref = case["rows"]
    for L in sorted({len(r[1]) for r in ref} | {3}):
        got = rows_of(str(path), L, chunk)
        assert [g[0] for g in got] == [r[0] for r in ref]
        assert [g[1] for g in got] == [len(r[1]) for r in ref]
        for g, r in zip(got, ref):
            if len(r[1]) == L:
                assert g[2] == [int(bool(v)) for v in r[1]]


def test_onehot_format_is_the_reference_S_line():
    rng = np.random.default_rng(5)
    for L in (0, 1, 5, 32, 33, 160, 7500):
        X = rng.random((9, L)) < 0.3
        ids = ["g%d" % i for i in range(8)] + ["sé q"]
        W = max(words_per_row(L), 4)
        bits = pack_bool_rows(X, W) if L else np.zeros((9, W), dtype=np.uint32)
        want = io.StringIO()
        for gid, x in zip(ids, X):          # classify.py:160-162
            print(gid, "\t".join(np.where(x, "1", "0")), sep="\t", file=want)
        assert ingest.format_onehot(ids, bits, L).decode("utf-8") == want.getvalue()


def test_roundtrip_format_then_parse(tmp_path):
    rng = np.random.default_rng(6)
    L = 7500
    X = rng.random((50, L)) < 0.2
    ids = ["read_%d" % i for i in range(50)]
    W = words_per_row(L)
    bits = pack_bool_rows(X, W)
    path = tmp_path / "s.tsv"
    path.write_bytes(ingest.format_onehot(ids, bits, L))
    got_ids, got_bits = [], []
    for i, b, nf in ingest.onehot_batches(str(path), L, W, batch_rows=16, chunk_bytes=100_000, slow_line=parse_onehot_line):
        assert (nf == L).all()
        got_ids += i
        got_bits.append(b)
    assert got_ids == ids
    assert (np.concatenate(got_bits) == bits).all()


def test_bad_arguments_are_rejected():
    L = ingest.lib()
    c = ctypes.c_int64(0)
    assert L.stag_onehot_parse(None, 0, 1, 5, 4, 1, None, None, ctypes.byref(c)) == -1
    assert L.stag_fasta_scan(None, 0, 1, 1, 1, None, 0, 0, None, None, ctypes.byref(c), ctypes.byref(c)) == -1


def test_fasta_scan_random_streams_equal_the_python_reader():
    """Seeded random byte streams (headers, wrapped lines, blank lines, '\\r', tabs, non-ASCII bytes,
    missing final newline) through the native scanner at random chunk and batch sizes against
    stag_amd.helpers.read_fasta, the port of helpers.py:107-121 that the golden vectors pin."""
    from stag_amd.helpers import read_fasta
    rng = np.random.default_rng(2024)
    pieces = [b">", b"\n", b"\n", b"\r\n", b"\r", b" ", b"\t", b"-", b"ACGT", b"acgu", b"N.", b"\xc3\xa9", b"\xc2\xa0",
              b"\xc2\x85", b"id_1", b"---A--", b">seq two\n", b"\x1c", b"\x0b", b"\xe2\x80\xa8", b"\x0c", b">\n"]
    for case in range(1500):
        m = int(rng.integers(0, 60))
        data = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=m))
        if case % 3 == 0:
            data = b">first\n" + data
        for text_mode in (False, True):
            if text_mode:
                want = [[i, s] for i, s in read_fasta(io.TextIOWrapper(io.BytesIO(data), encoding="utf-8", newline=None),
                                                      head_start=1, is_binary=False)]
            else:
                want = [[i, s] for i, s in read_fasta(io.BytesIO(data), head_start=1)]
            chunk = int(rng.choice([1, 2, 5, 17, 4096]))
            batch = int(rng.choice([1, 3, 1 << 16]))
            # sequence text travels one byte per CHARACTER (a character is one column for the reference):
            # a non-ASCII character becomes an ASCII stand-in of its class (tables.ascii_stand_in)
            from stag_amd.tables import ascii_stand_in
            want = [[i, ascii_stand_in(s)] for i, s in want]
            got = []
            for ids, text, offs in ingest.fasta_batches(io.BytesIO(data), batch_records=batch, chunk_bytes=chunk,
                                                        text_mode=text_mode):
                assert len(text) % 16 == 0      # (a re-parsed stretch may push a batch past batch_records)
                got += [[sid, bytes(text[offs[i]:offs[i + 1]])] for i, sid in enumerate(ids)]
            assert got == want, (case, text_mode, chunk, batch, data)


def test_onehot_parse_random_files_equal_the_per_line_parser(tmp_path):
    """Seeded random -s files (well-formed lines with occasional damage, and byte soup) through the
    native parser at random chunk and batch sizes against parse_onehot_line over text-mode lines
    (classify.py:16-20; pinned by the golden vectors): same ids, field counts, bits, and the same
    exception class when a field is not an integer."""
    rng = np.random.default_rng(4242)
    pieces = [b"0", b"1", b"\t", b"\t", b"\t", b"\n", b"\n", b"\r\n", b"\r", b"id", b" ", b"2", b"-1", b"10", b"01",
              b"\xc3\xa9", b"\xc2\xa0", b"x", b"\x0b", b"+1", b"1_0"]
    path = tmp_path / "s.tsv"
    for case in range(1200):
        if case % 2 == 0:
            lines, L0 = [], int(rng.integers(0, 8))
            for r in range(int(rng.integers(0, 6))):
                m = L0 if rng.random() < 0.8 else int(rng.integers(0, 8))
                line = b"\t".join([b"g%d" % r] + [bytes(rng.choice([b"0", b"1"])) for _ in range(m)])
                if rng.random() < 0.15:
                    line += pieces[int(rng.integers(0, len(pieces)))]
                lines.append(line + bytes(rng.choice([b"\n", b"\r\n", b"\r", b" \n"])))
            data = b"".join(lines)
            if rng.random() < 0.3 and data:
                data = data[:-1]
        else:
            data = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=int(rng.integers(0, 30))))
        path.write_bytes(data)
        try:
            with open(path, "r") as fh:
                want = [(gid, [bool(v) for v in x]) for gid, x in map(parse_onehot_line, fh)]
        except Exception as e:      # noqa: BLE001
            want = type(e).__name__
        for L in (0, 3, 5):
            chunk, rows = int(rng.choice([1, 2, 7, 4096])), int(rng.choice([1, 2, 7, 100]))
            try:
                got = rows_of(str(path), L, chunk, batch_rows=rows)
            except Exception as e:      # noqa: BLE001
                got = type(e).__name__
            if isinstance(want, str) or isinstance(got, str):
                assert got == want, (case, L, chunk, rows, data)
                continue
            assert [g[0] for g in got] == [w[0] for w in want], (case, L, chunk, rows, data)
            for g, w in zip(got, want):
                assert g[1] == len(w[1]) and (g[2] is None or g[2] == [int(v) for v in w[1]]), (case, L, chunk, rows, data)


def test_fasta_batches_in_the_two_buffer_ring_equal_plain_batches():
    """pinned=True hands out views of two buffers used in turn (page-locked when a GPU library is usable, plain
    arrays here): same batches as without, and a batch is still intact after the next one has been produced."""
    import io
    from stag_amd import ingest
    rng = np.random.default_rng(5)
    recs = []
    for i in range(700):
        seq = "".join(rng.choice(list("ACGT-acgt."), size=int(rng.integers(0, 300))))
        recs.append(">r%d x\n%s\n" % (i, "\n".join(seq[j:j + 50] for j in range(0, len(seq), 50))))
    data = "".join(recs).encode()
    plain = list(ingest.fasta_batches(io.BytesIO(data), batch_records=64, chunk_bytes=4096))
    prev = None
    count = 0
    for k, (ids, text, offs) in enumerate(ingest.fasta_batches(io.BytesIO(data), batch_records=64, chunk_bytes=4096, pinned=True)):
        assert ids == plain[k][0] and np.array_equal(offs, plain[k][2]) and np.array_equal(text, plain[k][1])
        assert len(text) % 16 == 0
        if prev is not None:
            assert np.array_equal(prev[0], plain[k - 1][1])      # the batch before is still what it was
        prev = (text, k)
        count += 1
    assert count == len(plain) and count > 5


def lines_of(path, chunk, batch_rows=7):
    """[(id, fields as text or None)] from the line finder that feeds the GPU parser."""
    out = []
    for ids, text, bb, be, raw in ingest.onehot_text_batches(path, batch_rows=batch_rows, chunk_bytes=chunk,
                                                             slow_line=parse_onehot_line):
        assert len(ids) == len(bb) == len(be) <= batch_rows
        for i, gid in enumerate(ids):
            out.append((gid, bytes(text[bb[i]:be[i]]).decode("utf-8"), raw(i)))
    return out


@pytest.mark.parametrize("name", sorted(G["onehot"]))
@pytest.mark.parametrize("chunk", [64, 100, 1 << 20])
def test_onehot_line_finder_matches_reference(tmp_path, name, chunk):
    """stag_onehot_lines: the ids are the reference's, and a line's body is its fields joined by tabs whenever the
    line is one the device parser may take (single-character fields); the slow path gets the line itself."""
    case = G["onehot"][name]
    data = bytes.fromhex(case["hex"])
    path = tmp_path / "in.tsv"
    path.write_bytes(data)
    with open(path, "r") as fh:
        ref_lines = list(fh)
    if case["raises"]:
        try:
            got = lines_of(str(path), chunk)
        except ValueError:       # (a flagged line whose field is not an integer: named by the slow path)
            return
    else:
        got = lines_of(str(path), chunk)
        assert [g[0] for g in got] == [r[0] for r in case["rows"]]
    assert [g[2].rstrip() for g in got] == [l.rstrip() for l in ref_lines]
    for gid, body, line in got:
        fields = line.rstrip().split("\t")[1:]
        if body:
            assert body == "\t".join(fields) and gid == line.rstrip().split("\t")[0]


def test_onehot_line_finder_random_files(tmp_path):
    rng = np.random.default_rng(77)
    pieces = [b"0", b"1", b"\t", b"\t", b"\n", b"\r\n", b"\r", b"id", b" ", b"2", b"\xc3\xa9", b"\xc2\xa0", b"x", b"\x0b"]
    path = tmp_path / "s.tsv"
    for case in range(600):
        data = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=int(rng.integers(0, 40))))
        path.write_bytes(data)
        with open(path, "r") as fh:
            ref_lines = list(fh)
        try:
            want_ids = [parse_onehot_line(l)[0] for l in ref_lines]
        except ValueError:
            want_ids = None
        for chunk in (64, 4096):
            try:
                got = lines_of(str(path), chunk, batch_rows=int(rng.choice([1, 3, 100])))
            except ValueError:       # a flagged line whose field is not an integer: the reference raises there too
                assert want_ids is None, (case, data)
                continue
            assert [g[2].rstrip() for g in got] == [l.rstrip() for l in ref_lines], (case, data)
            if want_ids is not None:
                assert [g[0] for g in got] == want_ids, (case, data)
            for gid, body, line in got:
                if body:
                    assert body == "\t".join(line.rstrip().split("\t")[1:]), (case, data)


def test_parsers_are_memory_safe_under_address_and_ub_sanitizers(tmp_path):
    """tests/csrc/sanitize_ingest.c drives the entry points of libstag_ingest with random and well-formed text in
    exactly sized heap blocks, chunked as the Python batchers chunk it; built with -fsanitize=address,undefined, any
    read or write outside a buffer aborts the run."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "sanitize_ingest")
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
                            "-I", os.path.join(root, "include"), "-o", exe,
                            os.path.join(root, "tests", "csrc", "sanitize_ingest.c"),
                            os.path.join(root, "stag_amd", "csrc", "stag_ingest.c")], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("sanitizer build not available: " + build.stderr[-300:])
    run = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    assert run.stdout.strip() == "ok 1500"


def test_onehot_line_finder_reader_thread(tmp_path):
    """The file is read a few chunks ahead by a pool of threads: lone-'\\r' line ends (no '\\n' to cut a chunk at), a
    last line without line end, an empty file; a generator that is dropped early ends its readers."""
    import threading
    path = tmp_path / "r.tsv"
    body = "\t".join("01"[i % 2] for i in range(9))
    for sep, last in (("\r", "\r"), ("\r", ""), ("\n", ""), ("\r\n", "\r\n")):
        lines = ["id%d\t%s" % (i, body) for i in range(50)]
        path.write_bytes((sep.join(lines) + last).encode())
        for chunk in (64, 100, 1 << 12):
            got = lines_of(str(path), chunk, batch_rows=3)
            assert [g[0] for g in got] == ["id%d" % i for i in range(50)], (sep, last, chunk)
            assert all(g[1] == body for g in got)
    path.write_bytes(b"")
    assert lines_of(str(path), 64) == []
    path.write_bytes(("\n".join("id%d\t%s" % (i, body) for i in range(5000)) + "\n").encode())
    before = threading.active_count()
    gen = ingest.onehot_text_batches(str(path), batch_rows=2, chunk_bytes=256, slow_line=parse_onehot_line)
    assert next(gen)[0] == ["id0", "id1"]
    assert threading.active_count() > before
    gen.close()
    assert threading.active_count() == before
    # something that cannot be read at an offset (a FIFO) goes through the single reader thread
    fifo = tmp_path / "fifo"
    os.mkfifo(fifo)
    data = path.read_bytes()
    writer = threading.Thread(target=lambda: open(fifo, "wb").write(data))
    writer.start()
    got = lines_of(str(fifo), 1 << 12, batch_rows=64)
    writer.join()
    assert [g[0] for g in got] == ["id%d" % i for i in range(5000)] and all(g[1] == body for g in got)


def test_onehot_line_finder_lines_far_longer_than_the_search_window(tmp_path):
    """Chunks are cut at the first line end behind each multiple of the chunk size, looked for 64 kB at a time: lines of
    300 kB in 1 MB chunks, and in chunks shorter than a line (every chunk is then one line)."""
    nf = 150_000
    body = "\t".join("01"[(i * 7) % 2] for i in range(nf))
    path = tmp_path / "wide.tsv"
    path.write_bytes("".join("g%d\t%s\n" % (i, body) for i in range(9)).encode())
    got = lines_of(str(path), 1 << 20, batch_rows=2)
    assert [g[0] for g in got] == ["g%d" % i for i in range(9)] and all(g[1] == body for g in got)
    got = lines_of(str(path), 1 << 18, batch_rows=2)
    assert [g[0] for g in got] == ["g%d" % i for i in range(9)] and all(g[1] == body for g in got)


@pytest.mark.parametrize("batch_records,chunk_bytes", [(1 << 16, 1 << 26), (97, 1000), (5, 333), (1, 64)])
def test_trimmed_records_put_back_together_are_the_records(batch_records, chunk_bytes):
    """fasta_batches(trim=True): each record without the run of '-' at its start and its end, and the two counts;
    '-' * lead + text + '-' * trail is the record read_fasta gives, whatever the chunk and batch sizes: all-gap
    records, records without gaps, wrapped lines whose breaks fall inside the runs, empty covered stretches."""
    rng = np.random.default_rng(11)
    recs = []
    for i in range(1200):
        ncol = int(rng.integers(1, 300))
        s = np.full(ncol, "-", dtype="<U1")
        a, b = sorted(rng.integers(0, ncol + 1, size=2))
        if i % 7:
            s[a:b] = rng.choice(list("ACGT-Nacg."), size=b - a)
        recs.append("".join(s))
    width = [60, 7, 1000][batch_records % 3]
    stream = "".join(">id%d x y\n%s\n" % (i, "\n".join(r[j:j + width] for j in range(0, len(r), width)))
                     for i, r in enumerate(recs)).encode()
    ids, got, dropped = [], [], 0
    for bids, text, offs, lt in ingest.fasta_batches(io.BytesIO(stream), batch_records=batch_records,
                                                     chunk_bytes=chunk_bytes, trim=True):
        assert lt.shape == (len(bids), 2) and lt.dtype == np.int32 and len(text) % 16 == 0
        for i in range(len(bids)):
            body = bytes(text[offs[i]:offs[i + 1]]).decode()
            assert not body.startswith("-") and not body.endswith("-")
            got.append("-" * int(lt[i, 0]) + body + "-" * int(lt[i, 1]))
            dropped += int(lt[i].sum())
        ids += bids
    assert got == recs and ids == ["id%d x y" % i for i in range(len(recs))]
    assert dropped > sum(map(len, recs)) // 3
    plain = records_of(ingest.fasta_batches(io.BytesIO(stream), batch_records=batch_records, chunk_bytes=chunk_bytes))
    assert [r[1] for r in plain] == recs


def test_wrapped_records_with_oddities_equal_the_python_reader():
    """The scanner copies plain lines 64 bytes at a time (copy_plain_lines, AVX2) and leaves everything else to its
    byte-wise code: long seeded streams of wrapped records -- line widths around 64, lines ending in blanks, tabs or
    '\\r\\n', a '\\r' or a character outside ASCII inside a line, blank lines, a '>' inside a line, lines longer than
    64 bytes, no newline at the end -- against stag_amd.helpers.read_fasta (helpers.py:107-121), at chunk sizes
    that cut lines anywhere; with trim, the records put back together."""
    from stag_amd.helpers import read_fasta
    from stag_amd.tables import ascii_stand_in
    rng = np.random.default_rng(77)
    odd = [b" ", b"\t", b"\r", b"\xc3\xa9", b"\xc2\xa0", b">", b"\x0b", b"  ", b"\xe2\x80\xa8"]
    for case in range(60):
        width = int(rng.choice([60, 61, 63, 64, 65, 80, 33, 200]))
        parts = []
        for i in range(int(rng.integers(1, 40))):
            ncol = int(rng.integers(1, 700))
            s = np.full(ncol, "-", dtype="<U1")
            a, b = sorted(rng.integers(0, ncol + 1, size=2))
            if i % 5:
                s[a:b] = rng.choice(list("ACGT-Nacg."), size=b - a)
            body = "".join(s).encode()
            lines = [body[j:j + width] for j in range(0, len(body), width)]
            for j in range(len(lines)):
                r = rng.random()
                if r < 0.04:       # something odd at the end of the line, inside it, or a blank line after it
                    lines[j] = lines[j] + odd[int(rng.integers(0, 4))]
                elif r < 0.08 and len(lines[j]) > 2:
                    at = int(rng.integers(1, len(lines[j])))
                    lines[j] = lines[j][:at] + odd[int(rng.integers(0, len(odd)))] + lines[j][at:]
                elif r < 0.10:
                    lines[j] = lines[j] + b"\n"
            eol = b"\r\n" if case % 7 == 3 else b"\n"
            parts.append(b">r%d some text\n" % i + eol.join(lines) + eol)
        data = b"".join(parts)
        if case % 4 == 1:
            data = data.rstrip(b"\r\n")
        want = [[i, ascii_stand_in(s)] for i, s in read_fasta(io.BytesIO(data), head_start=1)]
        for chunk in (int(rng.choice([100, 257, 1000])), 4096, 1 << 20):
            got = []
            for ids, text, offs in ingest.fasta_batches(io.BytesIO(data), batch_records=int(rng.choice([1, 7, 1 << 16])),
                                                        chunk_bytes=chunk):
                got += [[sid, bytes(text[offs[i]:offs[i + 1]])] for i, sid in enumerate(ids)]
            assert got == want, (case, chunk)
            got = []
            for ids, text, offs, lt in ingest.fasta_batches(io.BytesIO(data), batch_records=1 << 16, chunk_bytes=chunk, trim=True):
                got += [[sid, b"-" * int(lt[i, 0]) + bytes(text[offs[i]:offs[i + 1]]) + b"-" * int(lt[i, 1])]
                        for i, sid in enumerate(ids)]
            assert got == want, (case, chunk, "trim")
