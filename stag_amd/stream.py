"""Aligned text on disk -> two-column result lines, with every host core the job has.

The reference reads the aligner's output one line at a time in one Python thread (stag/helpers.py:107-121 under
stag/classify.py:155-158); at GPU rates that loop, not the GPU, sets the pace.  Here an A2M file (what
`hmmalign | esl-reformat a2m -` wrote, stag/align.py:127-141, kept on disk) is cut into chunks of whole records
and the chunks are scanned by a pool of threads -- stag_fasta_scan_trim releases the GIL, so they run side by side
-- into page-locked buffers; batches reach the GPU(s) in file order, without their flanking gap columns
(stag_classify_a2m_trimmed); ids stay bytes in the mapped file and the two-column lines (classify.py:182-190) are put
together in C (stag_rows_format_short).  No Python object per record anywhere on the path.

    a2m_file_batches   chunks of the file as trimmed record batches, scanned in parallel, yielded in file order
    classify_a2m_file  those batches through a Database (or a DevicePool of them) into an output file
"""
import collections
import ctypes
import io
import os
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, ingest

c_p = ctypes.c_void_p
CHUNK_BYTES = 48 << 20      # file bytes per chunk = per batch (about 31,000 amplicon reads of a 1,500-column alignment)


def default_threads():
    """Scanner threads: the cores this process may run on (a container's share, not the machine's), at most 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for var in ("STAG_THREADS", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            n = min(n, int(v)) if var == "OMP_NUM_THREADS" else int(v)
            break
    return max(1, min(n, 16))


def chunk_starts(buf, chunk_bytes):
    """Offsets at which chunks of about chunk_bytes begin, each at a header line ('>' at the start of a line);
    the first is 0, the last entry is len(buf)."""
    n = len(buf)
    starts, at = [0], 0
    while True:
        at += chunk_bytes
        if at >= n:
            break
        nxt = -1
        lo = at
        while lo < n:                                     # next "\n>" from `at` on, looked for a window at a time
            hi = min(n, lo + (1 << 20))
            hit = bytes(buf[lo:hi + 1]).find(b"\n>")
            if hit >= 0:
                nxt = lo + hit + 1
                break
            lo = hi
        if nxt < 0:
            break
        if nxt > starts[-1]:
            starts.append(nxt)
        at = max(at, nxt)
    starts.append(n)
    return starts


class Batch:
    """The records of one chunk: text / offs / lead_trail as stag_classify_a2m_trimmed takes them; ids as bytes
    of `id_buf` (id_begin, id_len) -- or, for a chunk that had to be parsed the slow way, as a list of str."""

    def __init__(self, index, text, offs, lead_trail, id_buf=None, id_begin=None, id_len=None, id_list=None):
        self.index, self.text, self.offs, self.lead_trail = index, text, offs, lead_trail
        self.id_buf, self.id_begin, self.id_len, self.id_list = id_buf, id_begin, id_len, id_list
        self.n = len(offs) - 1

    def ids(self):
        if self.id_list is None:
            mv = memoryview(self.id_buf)
            self.id_list = [str(mv[b:b + l], "utf-8") if l >= 0 else None
                            for b, l in zip(self.id_begin.tolist(), self.id_len.tolist())]
        return self.id_list

    def id_bytes(self):
        """(buffer, begin, len) of the ids for stag_rows_format_short."""
        if self.id_buf is None or (self.id_len < 0).any():     # (a record without header line prints as "None")
            enc = [("None" if s is None else s).encode("utf-8") for s in self.ids()]
            blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)
            lens = np.array([len(e) for e in enc], dtype=np.int64)
            begin = np.zeros(len(enc), dtype=np.int64)
            np.cumsum(lens[:-1], out=begin[1:])
            return blob, begin, lens
        return self.id_buf, self.id_begin, self.id_len


def _scan_chunk(index, view, at_start, text_buf, trim):
    """One chunk of whole records through stag_fasta_scan_trim (runs in a pool thread; the C call and the NumPy
    passes release the GIL)."""
    L = ingest.lib()
    max_records = int(np.count_nonzero(view == 62)) + 2          # '>' anywhere: an upper bound of the headers
    offs = np.zeros(max_records + 1, dtype=np.int64)
    recs = np.zeros(max_records, dtype=ingest.RECORD)
    lt = np.zeros(2 * max_records, dtype=np.int32)
    consumed, used = ctypes.c_int64(0), ctypes.c_int64(0)
    n = L.stag_fasta_scan_trim(view.ctypes.data_as(c_p), len(view), 1, int(at_start), max_records,
                               text_buf.ctypes.data_as(c_p), len(text_buf) - 16, 0, offs.ctypes.data_as(c_p),
                               recs.ctypes.data_as(c_p), ctypes.byref(consumed), ctypes.byref(used),
                               lt.ctypes.data_as(c_p) if trim else None)
    if n < 0 or consumed.value != len(view):
        raise ValueError("malformed FASTA/A2M file (a record does not start with '>')")
    recs = recs[:n]
    if n and (recs["flag"] == ingest.FALLBACK).any():
        # bytes outside ASCII or a lone '\r' somewhere in the chunk: the whole chunk goes the way of the reference's
        # reader (ingest.fasta_batches re-parses what it must with Python's str methods)
        ids, texts, offs_parts, lts, total = [], [], [np.zeros(1, dtype=np.int64)], [], 0
        for bids, btext, boffs, blt in ingest.fasta_batches(io.BytesIO(bytes(view)), trim=True):
            ids += bids
            texts.append(btext[:int(boffs[-1])])
            offs_parts.append(boffs[1:] + total)
            total += int(boffs[-1])
            lts.append(blt)
        text = np.concatenate(texts + [np.zeros(-total % 16 + 16, dtype=np.uint8)])
        lead_trail = np.concatenate(lts) if trim else None
        if not trim:        # put the gap columns back
            raise ValueError("untrimmed batches of a chunk with non-ASCII bytes are not supported")
        return Batch(index, text, np.concatenate(offs_parts), lead_trail, id_list=ids)
    total = used.value
    pad = -total % 16
    text_buf[total:total + pad] = 0
    return Batch(index, text_buf[:total + pad], offs[:n + 1], lt[:2 * n].reshape(n, 2) if trim else None,
                 id_buf=view, id_begin=recs["id_begin"].copy(), id_len=recs["id_len"].copy())


def a2m_file_batches(path, threads=None, chunk_bytes=None, trim=True, keep=2, world=1, rank=0):
    """Yields Batch objects for the chunks of an A2M / FASTA file, in file order, scanned by `threads` threads.
    A batch's text sits in one of a set of page-locked buffers and stays valid until `keep` further batches have
    been yielded.  Under a launcher with several ranks (world > 1) a rank scans only its own chunks (chunk b
    belongs to rank b % world) -- it seeks, it does not parse what the others take -- and yields None for the
    others, so that batch indices agree on every rank."""
    threads = threads or default_threads()
    size = os.path.getsize(path)
    if size == 0:
        return
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    starts = chunk_starts(buf, max(int(chunk_bytes or CHUNK_BYTES), 1 << 16))
    n_chunks = len(starts) - 1
    cap = max(b - a for a, b in zip(starts[:-1], starts[1:])) + 64
    n_buf = threads + keep + 1
    bufs = [None] * n_buf
    free, held, pending = collections.deque(range(n_buf)), collections.deque(), collections.deque()

    def text_buffer(i):
        if bufs[i] is None or len(bufs[i][1]) < cap:
            try:
                keep_alive = _lib.PinnedArray(cap, np.uint8)
                bufs[i] = (keep_alive, keep_alive.a)
            except Exception:
                bufs[i] = (None, np.empty(cap, dtype=np.uint8))
        return bufs[i][1]

    with ThreadPoolExecutor(threads, thread_name_prefix="stag-scan") as pool:
        nxt = 0
        while nxt < n_chunks or pending:
            while nxt < n_chunks and len(pending) < threads and (free or nxt % world != rank):
                if nxt % world != rank:
                    pending.append((nxt, None, None))
                else:
                    i = free.popleft()
                    view = buf[starts[nxt]:starts[nxt + 1]]
                    pending.append((nxt, i, pool.submit(_scan_chunk, nxt, view, nxt == 0, text_buffer(i), trim)))
                nxt += 1
            b, i, fut = pending.popleft()
            if fut is None:
                yield None
                continue
            batch = fut.result()
            held.append(i)
            while len(held) > keep + 1:
                free.append(held.popleft())
            yield batch
    del buf


DIRECT_KEYS = 8 << 20      # largest (end position, selected level) code space that is looked up in a table


class _ShortLines:
    """What a database has learnt about its two-column lines: the second column of a line (classify.py:182-184, the
    lineage down to the selected level) is fixed by where the sequence's path ends and which level was selected, so
    it is built once per (end position, selected level) code and kept -- all of them in one byte blob with offsets,
    which is what stag_rows_format_short indexes.  at[code] = index of the code's string (-1: not seen yet); for a
    database whose code space is larger than DIRECT_KEYS a dict does the same.  Not locked: it hangs on ONE Database
    (classify_a2m_file formats every batch with pool.first, whichever device computed it) and only the thread that
    writes the output touches it."""

    def __init__(self, space):
        self.space = space
        self.at = np.full(space, -1, dtype=np.int32) if space <= DIRECT_KEYS else None
        self.by_code = {}
        self.blob = bytearray(b"\0")     # (one byte in front, so that the buffer is never empty: a line's second column
        self._off = np.ones(1024, dtype=np.int64)   # may be the empty string, and even that needs an address to be
        self.n = 0                                  # copied from); string k = blob[_off[k] : _off[k + 1]]

    @property
    def off(self):
        """Offsets of the n strings known so far (n + 1 values; a view, no copy per batch)."""
        return self._off[:self.n + 1]

    def add(self, code, text):
        self.blob += text
        if self.n + 2 > len(self._off):
            self._off = np.concatenate([self._off, np.zeros(len(self._off), dtype=np.int64)])
        self.n += 1
        self._off[self.n] = len(self.blob)
        k = self.n - 1
        if self.at is not None:
            self.at[code] = k
        else:
            self.by_code[code] = k
        return k


def _lineage_keys(db, res):
    """key[i] = index of sequence i's second column in the database's _ShortLines (-1: no line for sequence i).
    A batch of known codes costs one table lookup per sequence; only codes that were never seen before are turned
    into strings (one Python call each), which after the first batches of a file is a few hundred per batch.
    (The first form of this sorted each batch's codes with np.unique and walked its distinct ones in Python.
    Measured in the build container on the same 32 synthetic batches of 31,000 sequences over 10,000 end nodes,
    same bytes out: 22 ms per batch in the steady state then, 10 ms now, 0.85 s against 0.53 s for all 32; the very
    first batch of a file is slower now, 126 ms against 96, because it adds its 20,000 strings one at a time.)"""
    per_node = len(db.keys) + 2
    space = (len(db.names) + 1) * per_node      # (names: one per node of the tree)
    cache = db.__dict__.get("_short_lines")
    if cache is None or cache.space != space:
        cache = db.__dict__["_short_lines"] = _ShortLines(space)
    key = np.full(res.n, -1, dtype=np.int32)
    rows = np.flatnonzero(res.status == _lib.SEQ_OK)
    if len(rows) == 0:
        return key, cache
    plen = res.path_len[rows].astype(np.int64)
    end = np.where(plen > 0, res.path_node[rows, np.maximum(plen - 1, 0)], -1).astype(np.int64)
    codes = (end + 1) * per_node + (res.selected[rows].astype(np.int64) + 1)

    def text_of(i):      # the second column of sequence i's line
        sel = int(res.selected[i])
        selected = db.keys[sel] if sel >= 0 else -1
        return ";".join(db.lineage(res, i)[0:int(selected) + 1]).encode("utf-8")

    if cache.at is not None and int(codes.min()) >= 0 and int(codes.max()) < space:
        idx = cache.at[codes]
        new = np.flatnonzero(idx < 0)
        if len(new):
            for j in new.tolist():               # (a code met twice in this batch is added once)
                c = int(codes[j])
                if cache.at[c] < 0:
                    cache.add(c, text_of(int(rows[j])))
            idx = cache.at[codes]
        key[rows] = idx
    else:
        # a code space too large for a table (or a code outside it, which db.lineage then complains about): a dict
        out = np.empty(len(rows), dtype=np.int32)
        for j, c in enumerate(codes.tolist()):
            k = cache.by_code.get(c)
            if k is None and cache.at is not None and 0 <= c < space and cache.at[c] >= 0:
                k = int(cache.at[c])
            if k is None:
                k = cache.add(c, text_of(int(rows[j])))
                cache.by_code[c] = k
            out[j] = k
        key[rows] = out
    return key, cache


def format_short_rows(db, res, batch):
    """The two-column output lines (classify.py:182-190, short form) of one batch, as bytes."""
    from .engine import raise_for_status
    if ((res.status != _lib.SEQ_OK) & (res.status != _lib.SEQ_FILTERED)).any():
        raise_for_status(res.status, batch.ids())
    key, cache = _lineage_keys(db, res)
    str_off = np.ascontiguousarray(cache.off)
    id_buf, id_begin, id_len = batch.id_bytes()
    id_begin = np.ascontiguousarray(id_begin, dtype=np.int64)
    id_len = np.ascontiguousarray(id_len, dtype=np.int64)
    used = key[key >= 0]
    longest = int((str_off[used + 1] - str_off[used]).max()) if len(used) else 0
    cap = int(id_len.clip(min=0).sum()) + batch.n * (longest + 2) + 16
    out = np.empty(cap, dtype=np.uint8)
    blob = np.frombuffer(cache.blob, dtype=np.uint8)     # a view: the bytearray cannot grow while it exists
    try:
        m = ingest.lib().stag_rows_format_short(id_buf.ctypes.data_as(c_p), id_begin.ctypes.data_as(c_p),
                                                id_len.ctypes.data_as(c_p), key.ctypes.data_as(c_p), batch.n,
                                                blob.ctypes.data_as(c_p), str_off.ctypes.data_as(c_p),
                                                out.ctypes.data_as(c_p), cap)
    finally:
        del blob
    if m < 0:
        raise RuntimeError("stag_rows_format_short: output buffer too small")
    return out[:m]


def classify_a2m_file(pool, path, output=None, threads=None, min_perc_state=0, chunk_bytes=None, header=True):
    """A2M file -> the reference's short output (header line `sequence<tab>taxonomy`, then one line per sequence
    that passed the filter, classify.py:172-190) in `output` (stdout when None; written to a temp file and moved
    into place as the reference does).  `pool` is a parallel.DevicePool (or anything with .first / .submit / .dbs):
    batches go to its devices in file order, two in flight per device.  Returns (alignment_length, n_lines)."""
    import shutil
    db = pool.first
    in_flight = 2 * len(pool.dbs) if len(pool.dbs) > 1 else 2
    sink = tempfile.NamedTemporaryFile(delete=False, mode="wb") if output else sys.stdout.buffer
    if output:
        os.chmod(sink.name, 0o644)
    pending = collections.deque()
    state = {"alignment_length": None, "lines": 0}

    def finish_oldest():
        batch, fut = pending.popleft()
        res = fut.result()
        passed = np.flatnonzero(res.status == _lib.SEQ_OK)
        if len(passed) and not state["alignment_length"]:
            state["alignment_length"] = 5 * int(res.n_char[passed[0]])
        rows = format_short_rows(db, res, batch)
        state["lines"] += len(passed)
        sink.write(memoryview(rows))

    try:
        if header:
            sink.write(b"sequence\ttaxonomy\n")
        for batch in a2m_file_batches(path, threads=threads, chunk_bytes=chunk_bytes, keep=in_flight + 1):
            if batch is None or batch.n == 0:
                continue
            pending.append((batch, pool.submit(lambda d, b=batch: d.classify_a2m_text(
                b.text, b.offs, min_perc_state=min_perc_state, lead_trail=b.lead_trail))))
            while len(pending) >= in_flight:
                finish_oldest()
        while pending:
            finish_oldest()
        if output:
            sink.flush()
            os.fsync(sink.fileno())
    finally:
        if output:
            sink.close()
        else:
            sink.flush()
    if output:
        shutil.move(sink.name, output)
    return state["alignment_length"], state["lines"]
