"""k_parse_onehot (stag_parse_onehot_dev / stag_onehot_text_parse + _classify): the lines of a -s file parsed on the
GPU, against the reference's reader.

tests/golden/ingest.json holds byte streams and what stag.classify.alignment_reader (classify.py:16-20) returned
for them (oracle/make_golden.py).  Lines the kernel flags go through parse_onehot_line, the restatement of the
reference's per-field int() that the same golden vectors pin (tests/test_host_ingest.py), exactly as
classify() does it.  Bit-exact: ids, field counts and every bit of every row.
"""
import ctypes

import numpy as np
import pytest

from golden_util import load
from stag_amd import _lib, ingest
from stag_amd.classify import parse_onehot_line
from stag_amd.tables import pack_bool_rows, unpack_rows, words_per_row

pytestmark = pytest.mark.gpu
G = load("ingest.json")


def device_rows(path, L, chunk, batch_rows):
    """[(id, n_fields, bits or None)] of a file: lines found on the host, fields parsed by the kernel, flagged
    lines by parse_onehot_line."""
    import torch
    dev = torch.device("cuda", 0)
    W = words_per_row(L)
    out = []
    for ids, text, bb, be, raw in ingest.onehot_text_batches(path, batch_rows=batch_rows, chunk_bytes=chunk, slow_line=parse_onehot_line):
        n = len(ids)
        padded = np.zeros(len(text) + (-len(text) % 16) + 16, dtype=np.uint8)
        padded[:len(text)] = text
        d_text = torch.from_numpy(padded).to(dev)
        d_bb, d_be = torch.from_numpy(np.ascontiguousarray(bb)).to(dev), torch.from_numpy(np.ascontiguousarray(be)).to(dev)
        d_bits = torch.full((n, W), -1, dtype=torch.int32, device=dev)       # stale contents must not survive
        d_nf = torch.full((n,), -7, dtype=torch.int32, device=dev)
        d_flag = torch.full((n,), -7, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev)
        _lib.check(_lib.lib().stag_parse_onehot_dev(L, d_text.data_ptr(), len(padded), d_bb.data_ptr(), d_be.data_ptr(), n,
                                                    d_bits.data_ptr(), d_nf.data_ptr(), d_flag.data_ptr(), st.cuda_stream))
        torch.cuda.synchronize()
        bits = d_bits.cpu().numpy().view(np.uint32)
        nf, flag = d_nf.cpu().numpy(), d_flag.cpu().numpy()
        dense = unpack_rows(bits, L)
        for i, gid in enumerate(ids):
            if flag[i]:
                assert not bits[i].any()
                gid2, x = parse_onehot_line(raw(i))
                assert gid2 == gid
                out.append((gid, len(x), [int(bool(v)) for v in x] if len(x) == L else None))
            else:
                if nf[i] != L:
                    assert not bits[i].any()
                else:       # nothing set past the last feature
                    assert not (bits[i, (L + 31) // 32:]).any() and (L % 32 == 0 or bits[i, L // 32] >> (L % 32) == 0)
                out.append((gid, int(nf[i]), dense[i].astype(int).tolist() if nf[i] == L else None))
    return out


@pytest.mark.parametrize("name", sorted(G["onehot"]))
@pytest.mark.parametrize("chunk", [1 << 16, 1 << 20])
def test_device_parser_matches_reference(tmp_path, name, chunk):
    case = G["onehot"][name]
    path = tmp_path / "in.tsv"
    path.write_bytes(bytes.fromhex(case["hex"]))
    if case["raises"]:
        with pytest.raises(ValueError):      # int() on a field that is not an integer
            device_rows(str(path), 3, chunk, 7)
        return
    ref = case["rows"]
    for L in sorted({len(r[1]) for r in ref if len(r[1])} | {3}):
        got = device_rows(str(path), L, chunk, 7)
        assert [g[0] for g in got] == [r[0] for r in ref]
        assert [g[1] for g in got] == [len(r[1]) for r in ref]
        for g, r in zip(got, ref):
            if len(r[1]) == L:
                assert g[2] == [int(bool(v)) for v in r[1]]


def test_device_parser_random_files_equal_the_per_line_parser(tmp_path):
    """The seeded random -s files of tests/test_host_ingest.py (well-formed lines with occasional damage, byte
    soup, every line end) plus long lines whose fields cross the 64-byte lane boundaries at every offset."""
    rng = np.random.default_rng(4243)
    pieces = [b"0", b"1", b"\t", b"\t", b"\t", b"\n", b"\n", b"\r\n", b"\r", b"id", b" ", b"2", b"-1", b"10", b"01",
              b"\xc3\xa9", b"\xc2\xa0", b"x", b"\x0b", b"+1", b"1_0"]
    path = tmp_path / "s.tsv"
    for case in range(300):
        kind = case % 3
        if kind == 0:
            lines, L0 = [], int(rng.integers(1, 8))
            for r in range(int(rng.integers(0, 6))):
                m = L0 if rng.random() < 0.8 else int(rng.integers(0, 8))
                line = b"\t".join([b"g%d" % r] + [bytes(rng.choice([b"0", b"1"])) for _ in range(m)])
                if rng.random() < 0.15:
                    line += pieces[int(rng.integers(0, len(pieces)))]
                lines.append(line + bytes(rng.choice([b"\n", b"\r\n", b"\r", b" \n"])))
            data = b"".join(lines)
            if rng.random() < 0.3 and data:
                data = data[:-1]
            Ls = (L0, 3)
        elif kind == 1:
            data = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=int(rng.integers(0, 30))))
            Ls = (3, 5)
        else:
            # long lines: ids of every length shift the fields against the 4-byte and 64-byte grids
            L0 = int(rng.choice([31, 32, 33, 63, 64, 65, 127, 129, 2047, 2049, 4100, 7500]))
            lines = []
            for r in range(int(rng.integers(1, 5))):
                m = L0 if rng.random() < 0.7 else int(L0 + rng.integers(-3, 4))
                body = np.full(2 * m - 1, 9, dtype=np.uint8)
                body[0::2] = 48 + (rng.random(m) < 0.4)
                if rng.random() < 0.3:       # one damaged byte somewhere
                    body[int(rng.integers(0, len(body)))] = int(rng.choice([ord("2"), ord(" "), ord("\t"), ord("0")]))
                lines.append(b"i" * int(rng.integers(0, 9)) + b"\t" + body.tobytes() + bytes(rng.choice([b"\n", b"\r\n", b"  \n"])))
            data = b"".join(lines)
            Ls = (L0,)
        path.write_bytes(data)
        try:
            with open(path, "r") as fh:
                want = [(gid, [bool(v) for v in x]) for gid, x in map(parse_onehot_line, fh)]
        except Exception as e:      # noqa: BLE001
            want = type(e).__name__
        for L in Ls:
            try:
                got = device_rows(str(path), L, 1 << 16, int(rng.choice([1, 2, 7, 100])))
            except Exception as e:      # noqa: BLE001
                got = type(e).__name__
            if isinstance(want, str) or isinstance(got, str):
                assert got == want, (case, L, data[:200])
                continue
            assert [g[0] for g in got] == [w[0] for w in want], (case, L)
            assert [g[1] for g in got] == [len(w[1]) for w in want], (case, L)
            for g, w in zip(got, want):
                if len(w[1]) == L:
                    assert g[2] == [int(v) for v in w[1]], (case, L, g[0])


def test_text_entry_equals_packed_rows_entry(tmp_path):
    """Database.classify_onehot_text (text up, parsed on the GPU, flagged and wrong-length lines handled) against
    classify_bits on the rows the host parser makes of the same file: every output identical."""
    import bench_workload as bw
    from stag_amd.engine import Database
    wl = bw.make_db(11, 40, [3, 9, 27], 0.1, name="t")
    L = wl.tables.n_features
    W = words_per_row(L)
    rng = np.random.default_rng(8)
    n = 700
    X = rng.random((n, L)) < 0.2
    ids = ["r%d" % i for i in range(n)]
    blob = bytearray(ingest.format_onehot(ids, pack_bool_rows(X, W), L))
    lines = bytes(blob).split(b"\n")[:-1]
    lines[5] = lines[5][:-2]                         # one field short
    lines[9] = lines[9] + b"\t1"                     # one field long
    lines[17] = lines[17].replace(b"\t1\t", b"\t7\t", 1)      # int("7") is True for the reference
    lines[23] = lines[23].replace(b"\t0\t", b"\t 0\t", 1)     # int(" 0") is 0
    path = tmp_path / "s.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    with Database(wl.tables) as db:
        want_rows, want_nf, want_ids = [], [], []
        for i, b, nf in ingest.onehot_batches(str(path), L, W, batch_rows=256, slow_line=parse_onehot_line):
            want_ids += i
            want_rows.append(b)
            want_nf.append(nf)
        want_rows, want_nf = np.concatenate(want_rows), np.concatenate(want_nf)
        status = np.where(want_nf == L, _lib.SEQ_OK, _lib.SEQ_LENGTH_MISMATCH).astype(np.int32)
        ref = db.classify_bits(want_rows, status)
        at = 0
        for ids_b, text, bb, be, raw in ingest.onehot_text_batches(str(path), batch_rows=256, slow_line=parse_onehot_line):
            res, nf = db.classify_onehot_text(text, bb, be, parse_onehot_line, raw)
            m = len(ids_b)
            assert ids_b == want_ids[at:at + m]
            assert np.array_equal(nf, want_nf[at:at + m])
            for f in ("path_node", "path_prob", "path_len", "level_prob", "selected", "status", "n_aligned", "n_lead_single"):
                assert np.array_equal(getattr(res, f), getattr(ref, f)[at:at + m]), f
            at += m
        assert at == n
        assert (ref.status == _lib.SEQ_LENGTH_MISMATCH).sum() == 2 and (ref.status == _lib.SEQ_OK).sum() == n - 2


def test_parse_then_classify_protocol_is_checked():
    import bench_workload as bw
    from stag_amd.engine import Database
    wl = bw.make_db(12, 20, [2, 4], 0.1, name="t")
    with Database(wl.tables) as db:
        res = np.zeros(4, dtype=np.int32)
        out = _lib.Results()
        rc = _lib.lib().stag_onehot_text_classify(db.handle, 3, None, None, 0, None, ctypes.byref(out))
        assert rc != 0 and "bad argument" in _lib.lib().stag_last_error().decode()
        text = np.frombuffer(b"x\t0\t1\n" + b"\0" * 10, dtype=np.uint8)
        bb, be = np.array([2], dtype=np.int64), np.array([40], dtype=np.int64)      # body past the buffer
        rc = _lib.lib().stag_onehot_text_parse(db.handle, _lib.ptr(text), len(text), _lib.ptr(bb), _lib.ptr(be), 1,
                                               _lib.ptr(res), _lib.ptr(res))
        assert rc != 0 and "not inside the text buffer" in _lib.lib().stag_last_error().decode()


def test_device_parser_on_genome_wide_lines(tmp_path):
    """Lines of the concatenated genome database's width (L = 157,500: 315 kB per line, 77 steps of a wave), one of them
    damaged in its last field, one a field short: rows, counts and flags as the per-line parser gives them."""
    L = 157_500
    W = words_per_row(L)
    rng = np.random.default_rng(21)
    X = rng.random((6, L)) < 0.3
    blob = ingest.format_onehot(["g%d" % i for i in range(6)], pack_bool_rows(X, W), L)
    lines = blob.split(b"\n")[:-1]
    lines[2] = lines[2][:-1] + b"5"          # int("5") is True for the reference: the slow path sets that bit
    lines[4] = lines[4][:-2]                 # one field short
    path = tmp_path / "wide.tsv"
    path.write_bytes(b"\n".join(lines) + b"\n")
    got = device_rows(str(path), L, 1 << 22, 4)
    assert [g[0] for g in got] == ["g%d" % i for i in range(6)]
    assert [g[1] for g in got] == [L, L, L, L, L - 1, L]
    X[2, -1] = True
    for i in (0, 1, 2, 3, 5):
        assert got[i][2] == X[i].astype(int).tolist(), i
    assert got[4][2] is None
