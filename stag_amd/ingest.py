"""Host ingestion around the GPU path: native parsers for the two text
formats that feed it and the writer for the one it emits, bound from libstag_ingest.so
(include/stag_ingest.h, plain C, no GPU).

  fasta_batches   stag/helpers.py:107-121 read_fasta over the `esl-reformat a2m -` pipe
                  (stag/align.py:139-147) -> (ids, text uint8, offs int64) batches ready for
                  stag_classify_a2m: no Python string per record
  onehot_batches  stag/classify.py:16-20 alignment_reader -> (ids, packed rows, n_fields) batches
  onehot_text_batches  the same lines found but not parsed: (ids, text, body ranges, raw lines) batches for the
                  GPU parser (engine.Database.classify_onehot_text)
  format_onehot   stag/classify.py:160-162, the -S lines

Records the C fast path cannot reproduce byte for byte are flagged by it and re-parsed here the
way the reference does (Python str methods), so the output is the reference's in every case.
"""
import ctypes
import io
import os
import queue
import stat
import threading

import numpy as np

from .helpers import read_fasta
from .tables import pack_text

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstag_ingest.so")
FALLBACK = 1

c_p = ctypes.c_void_p
RECORD = np.dtype([("id_begin", "<i8"), ("id_len", "<i8"), ("raw_begin", "<i8"), ("raw_end", "<i8"),
                   ("n_fields", "<i8"), ("flag", "<i4"), ("pad", "<i4")])

EXPORTS = {
    "stag_fasta_scan": (ctypes.c_int64, [c_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                         c_p, ctypes.c_int64, ctypes.c_int64, c_p, c_p, c_p, c_p]),
    "stag_fasta_scan_trim": (ctypes.c_int64, [c_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                              c_p, ctypes.c_int64, ctypes.c_int64, c_p, c_p, c_p, c_p, c_p]),
    "stag_onehot_parse": (ctypes.c_int64, [c_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int64, c_p, c_p, c_p]),
    "stag_onehot_lines": (ctypes.c_int64, [c_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, c_p, c_p, c_p, c_p]),
    "stag_onehot_format": (ctypes.c_int64, [c_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_p, c_p, c_p,
                                            ctypes.c_int64]),
    "stag_rows_format_short": (ctypes.c_int64, [c_p, c_p, c_p, c_p, ctypes.c_int64, c_p, c_p, c_p, ctypes.c_int64]),
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C stag_amd/csrc`" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(c_p)


def _chunks(stream, chunk_bytes):
    """(bytes, final) pieces of a binary stream; the last piece (possibly empty) has final=True."""
    cur = stream.read(chunk_bytes)
    while True:
        nxt = stream.read(chunk_bytes) if cur else b""
        yield cur, not nxt
        if not nxt:
            return
        cur = nxt


def _fill(stream, buf, have):
    """Reads into buf[have:] until it is full or the stream ends; returns (bytes held, at end)."""
    mv = memoryview(buf)
    while have < len(buf):
        got = stream.readinto(mv[have:])
        if not got:
            return have, True
        have += got
    return have, False


class _TextRing:
    """A few page-locked text buffers (stag_host_alloc) used in turn: the batches handed to the caller sit in
    some while the next is assembled in another, so the GPU's copy engine reads batches by DMA (the pipeline of
    stag_classify_a2m overlaps its copies only from such memory).  Without a usable GPU library the buffers
    are ordinary arrays."""

    def __init__(self, slots=2):
        self.bufs, self.turn, self.ok = [None] * max(slots, 2), 0, True

    def take(self, nbytes):
        self.turn = (self.turn + 1) % len(self.bufs)
        cur = self.bufs[self.turn]
        if cur is not None and len(cur[1]) >= nbytes:
            return cur[1][:nbytes]
        arr = None
        if self.ok:
            try:
                from . import _lib
                keep = _lib.PinnedArray(max(nbytes + (nbytes >> 2), 1 << 20), np.uint8)
                arr = (keep, keep.a)
            except Exception:
                self.ok = False
        if arr is None:
            arr = (None, np.empty(nbytes, dtype=np.uint8))
        self.bufs[self.turn] = arr
        return arr[1][:nbytes]


def fasta_batches(stream, batch_records=1 << 16, chunk_bytes=1 << 26, head_start=1, text_mode=False, pinned=False,
                  ring_slots=2, trim=False):
    """Yields (ids, text, offs): ids a list of str, record i's sequence = text[offs[i]:offs[i+1]]
    (uint8 array padded to a multiple of 16 bytes, int64 offsets).  `stream` is a binary file object
    (a pipe's stdout).  Same records, ids and strings as read_fasta(stream, head_start=1); with
    text_mode, as read_fasta(open(path), is_binary=False): universal newlines, so a lone '\r' also
    ends a line (the gene file of align.py:150).  batch_records is a target, not a bound: records
    that had to be re-parsed the slow way are all added to the batch that was being filled.
    pinned: the text arrays are views of ring_slots page-locked buffers used in turn (a batch stays valid until
    ring_slots - 1 further batches have been produced).
    trim: yields (ids, text, offs, lead_trail) instead -- each record without the run of '-' at its start and at
    its end, lead_trail int32 [n, 2] = how many were dropped (what Database.classify_a2m_text(lead_trail=...) and
    stag_classify_a2m_trimmed take: a gap column is what a row says anyway, so it need not travel to the GPU)."""
    L = lib()
    ring = _TextRing(ring_slots) if pinned else None
    lt_parts = []
    buf = bytearray(max(chunk_bytes, 1))     # read buffer: [0, have) holds unparsed bytes
    scratch = np.empty(len(buf), dtype=np.uint8)
    have, at_start, ended = 0, True, False
    ids, texts, offs_parts, total = [], [], [np.zeros(1, dtype=np.int64)], 0

    def flush():
        nonlocal ids, texts, offs_parts, total
        if ring is None:
            text = np.zeros(total + (-total % 16), dtype=np.uint8)
        else:
            text = ring.take(total + (-total % 16))
            text[total:] = 0
        at = 0
        for t in texts:
            text[at:at + len(t)] = t
            at += len(t)
        out = (ids, text, np.concatenate(offs_parts))
        if trim:
            out += (np.concatenate(lt_parts).reshape(-1, 2) if lt_parts else np.zeros((0, 2), dtype=np.int32),)
            lt_parts.clear()
        ids, texts, offs_parts, total = [], [], [np.zeros(1, dtype=np.int64)], 0
        return out

    while not ended:
        have, ended = _fill(stream, buf, have)
        if len(scratch) < len(buf):
            scratch = np.empty(len(buf), dtype=np.uint8)
        pos = 0
        while True:
            view = np.frombuffer(buf, dtype=np.uint8)[pos:have]
            room = batch_records - len(ids)
            offs = np.zeros(room + 1, dtype=np.int64)
            recs = np.zeros(max(room, 1), dtype=RECORD)
            consumed, used = ctypes.c_int64(0), ctypes.c_int64(0)
            lt = np.zeros(2 * max(room, 1), dtype=np.int32) if trim else None
            n = L.stag_fasta_scan_trim(_ptr(view) if len(view) else _ptr(scratch), len(view), int(ended), int(at_start),
                                       room, _ptr(scratch), len(scratch), 0, _ptr(offs), _ptr(recs), ctypes.byref(consumed),
                                       ctypes.byref(used), _ptr(lt) if trim else None)
            if n < 0:
                raise ValueError("malformed FASTA/A2M stream (a record does not start with '>')")
            recs = recs[:n]
            if n and (recs["flag"] == FALLBACK).any():
                # non-ASCII bytes in a line: let Python's str methods decide, as the reference does
                raw = bytes(view[int(recs["raw_begin"][0]):int(recs["raw_end"][-1])])
                if text_mode:
                    pairs = list(read_fasta(io.TextIOWrapper(io.BytesIO(raw), encoding="utf-8", newline=None),
                                            head_start=head_start, is_binary=False))
                else:
                    pairs = list(read_fasta(io.BytesIO(raw), head_start=head_start))
                t, o = pack_text([s for _, s in pairs])
                ids.extend(sid for sid, _ in pairs)
                texts.append(t[:int(o[-1])].copy())
                offs_parts.append(o[1:] + total)
                total += int(o[-1])
                if trim:
                    lt_parts.append(np.zeros(2 * len(pairs), dtype=np.int32))
            elif n:
                begins = (recs["id_begin"] + (pos - 1 + head_start)).tolist()
                ends = (recs["id_begin"] + recs["id_len"] + pos).tolist()
                # (lines before the first header form a record without id: sid is None, id_len < 0)
                ids.extend([bytes(buf[b:e]).decode("utf-8") if e >= b else None for b, e in zip(begins, ends)])
                texts.append(scratch[:used.value].copy())
                offs_parts.append(offs[1:n + 1] + total)
                total += used.value
                if trim:
                    lt_parts.append(lt[:2 * n])
            if consumed.value:
                at_start = False
            pos += consumed.value
            if len(ids) >= batch_records:
                yield flush()
            if not consumed.value:
                break             # the rest of the buffer is an incomplete record (or nothing)
        # keep the unparsed tail; a record longer than the buffer makes the buffer grow
        rest = have - pos
        if pos == 0 and have == len(buf) and not ended:
            buf = buf + bytearray(len(buf))      # a new object: NumPy views of the old one may be alive
        elif pos:
            view = None
            buf[:rest] = buf[pos:have]
        have = rest
    if ids:
        yield flush()


def onehot_batches(path, n_features, words_per_row, batch_rows=1 << 14, chunk_bytes=1 << 26, slow_line=None):
    """Yields (ids, bits uint32 [n, words_per_row], n_fields int64 [n]) for the lines of a -s file.
    Rows whose field count differs from n_features come back zero with their count.  Lines the
    fast path does not accept go through `slow_line(str) -> (id, bool array)` (the reference's
    int() per field)."""
    L = lib()
    pending = b""
    with open(path, "rb") as fh:
        for piece, final in _chunks(fh, chunk_bytes):
            buf = pending + piece
            pos = 0
            while True:
                view = np.frombuffer(buf, dtype=np.uint8)[pos:]
                bits = np.zeros((batch_rows, words_per_row), dtype=np.uint32)
                recs = np.zeros(batch_rows, dtype=RECORD)
                consumed = ctypes.c_int64(0)
                n = L.stag_onehot_parse(_ptr(view) if len(view) else _ptr(bits), len(view), int(final), n_features,
                                        words_per_row, batch_rows, _ptr(bits), _ptr(recs), ctypes.byref(consumed))
                if n < 0:
                    raise ValueError("stag_onehot_parse: bad arguments")
                if n == 0:
                    break
                recs, bits = recs[:n], bits[:n]
                begins = (recs["id_begin"] + pos).tolist()
                ends = (recs["id_begin"] + recs["id_len"] + pos).tolist()
                ids = [buf[b:e].decode("utf-8") for b, e in zip(begins, ends)]
                n_fields = recs["n_fields"].copy()
                for i in np.flatnonzero(recs["flag"] == FALLBACK):
                    line = buf[pos + int(recs["raw_begin"][i]):pos + int(recs["raw_end"][i])].decode("utf-8")
                    gid, x = slow_line(line)
                    ids[i] = gid
                    n_fields[i] = len(x)
                    bits[i] = 0
                    if len(x) == n_features:
                        from .tables import pack_bool_rows
                        bits[i] = pack_bool_rows(np.asarray(x, dtype=bool)[None, :], words_per_row)[0]
                yield ids, bits, n_fields
                pos += consumed.value
            pending = buf[pos:]


# Page-locked buffers that a finished reader hands back, for the next reader OF THE SAME THREAD: locking memory costs
# about as much as reading it once, which a short file (or a process that classifies file after file) would pay again
# and again.  (Per thread, because a thread starts its next reader only after it is done with the batches of the
# last one; another thread's reader could start while those are still being copied to the GPU.)
_PINNED_SPARE = {}            # (thread, nbytes) -> [(owner, array)]
_PINNED_SPARE_LOCK = threading.Lock()
_PINNED_SPARE_MAX = 1 << 30   # bytes kept at most


class _PinnedRing:
    """Page-locked byte buffers used in turn, allocated when first used (pageable arrays when the GPU library cannot
    give such memory)."""

    def __init__(self, slots, nbytes):
        self.bufs, self.turn, self.nbytes = [None] * max(slots, 2), 0, nbytes

    def next(self):
        self.turn = (self.turn + 1) % len(self.bufs)
        if self.bufs[self.turn] is None:
            with _PINNED_SPARE_LOCK:
                spare = _PINNED_SPARE.get((threading.get_ident(), self.nbytes))
                got = spare.pop() if spare else None
            if got is not None:
                self.bufs[self.turn] = got
                return got[1]
            try:
                from . import _lib
                keep = _lib.PinnedArray(self.nbytes, np.uint8)
                self.bufs[self.turn] = (keep, keep.a)
            except Exception:
                self.bufs[self.turn] = (None, np.empty(self.nbytes, dtype=np.uint8))
        return self.bufs[self.turn][1]

    def release(self):
        """The reader is done (and so is everything that read its batches): the page-locked buffers go to the spare
        list, up to _PINNED_SPARE_MAX bytes in all."""
        with _PINNED_SPARE_LOCK:
            held = sum(k[1] * len(v) for k, v in _PINNED_SPARE.items())
            for item in self.bufs:
                if item is not None and item[0] is not None and held + self.nbytes <= _PINNED_SPARE_MAX:
                    _PINNED_SPARE.setdefault((threading.get_ident(), self.nbytes), []).append(item)
                    held += self.nbytes
        self.bufs = [None] * len(self.bufs)


def _last_line_end(buf, have):
    """End (exclusive) of the last complete line of buf[:have] for a reader with universal newlines that may get
    more bytes: after the last '\n'; without any, after the last '\r' that is not the last byte (a '\n' could
    still follow that one).  0: no complete line."""
    for byte, limit in ((10, have), (13, have - 1)):
        hi, span = limit, 1 << 16
        while hi > 0:
            lo = max(0, hi - span)
            hit = np.flatnonzero(buf[lo:hi] == byte)
            if len(hit):
                return lo + int(hit[-1]) + 1
            hi, span = lo, span * 8
    return 0


class _LineChunks:
    """Reads a file in a thread of its own, one buffer of the ring ahead of the consumer: iterating yields
    (buf, have) with buf[:have] = whole lines only (the bytes after the last line end of a read open the next
    buffer).  File reads release the GIL, so they run beside the consumer's line finding, its GPU calls and its
    formatting."""

    def __init__(self, path, ring, cap):
        self.path, self.ring, self.cap = path, ring, cap
        self.q = queue.Queue(maxsize=1)
        self.stop = threading.Event()
        self.thread = threading.Thread(target=self._run, name="stag-onehot-reader", daemon=True)
        self.thread.start()

    def _put(self, item):
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.1)
                return
            except queue.Full:
                pass

    def _run(self):
        try:
            room = self.cap - 16          # (16 bytes stay free: the GPU's copy of a batch is padded)
            with open(self.path, "rb", buffering=0) as fh:
                tail, ended = b"", False
                while not ended and not self.stop.is_set():
                    buf = self.ring.next()
                    rest = len(tail)
                    buf[:rest] = np.frombuffer(tail, dtype=np.uint8)
                    have, ended = _fill(fh, buf[:room], rest)
                    cut = have if ended else _last_line_end(buf, have)
                    if cut == 0 and not ended:
                        raise ValueError("a line of %s is longer than the read buffer (%d bytes)" % (self.path, self.cap))
                    tail = buf[cut:have].tobytes()
                    self._put((buf, cut))
            self._put(None)
        except BaseException as e:      # noqa: BLE001  (handed to the consumer, which raises it)
            self._put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self.stop.set()
        while True:                      # a producer blocked in put() sees the flag; one waiting in read() ends with it
            try:
                self.q.get_nowait()
            except queue.Empty:
                break
        self.thread.join(timeout=5.0)


def _line_chunk_starts(mm, chunk_bytes):
    """Offsets at which chunks of about chunk_bytes begin, each right after a line end ('\n', '\r\n' or a lone
    '\r': the universal newlines of the reference's text-mode reader); first 0, last len(mm)."""
    n = len(mm)
    starts, at = [0], 0
    while True:
        at += chunk_bytes
        if at >= n:
            break
        nxt, lo = -1, at
        while lo < n:
            hi = min(n, lo + (1 << 16))
            w = np.asarray(mm[lo:hi])
            hit = np.flatnonzero((w == 10) | (w == 13))
            if len(hit):
                p = lo + int(hit[0])
                if mm[p] == 13 and p + 1 < n and mm[p + 1] == 10:
                    p += 1
                nxt = p + 1
                break
            lo = hi
        if nxt < 0 or nxt >= n:
            break
        if nxt > starts[-1]:
            starts.append(nxt)
        at = max(at, nxt)
    starts.append(n)
    return starts


def _read_and_find_lines(fd, begin, end, buf):
    """Pool thread: file bytes [begin, end) -- whole lines -- into buf (page-locked), then stag_onehot_lines over them.
    Both release the GIL.  Returns (recs, body_begin, body_end) of the chunk's lines."""
    L = lib()
    have, want = 0, end - begin
    mv = memoryview(buf)
    while have < want:
        got = os.preadv(fd, [mv[have:want]], begin + have)
        if got <= 0:
            raise OSError("short read: the file changed while it was being read")
        have += got
    parts, pos = [], 0
    while pos < want:
        room = 4096
        recs = np.zeros(room, dtype=RECORD)
        bb = np.zeros(room, dtype=np.int64)
        be = np.zeros(room, dtype=np.int64)
        consumed = ctypes.c_int64(0)
        view = buf[pos:want]
        n = L.stag_onehot_lines(_ptr(view), len(view), 1, room, _ptr(recs), _ptr(bb), _ptr(be), ctypes.byref(consumed))
        if n <= 0 or consumed.value <= 0:
            raise ValueError("stag_onehot_lines: bad arguments")
        recs = recs[:n]
        for f in ("id_begin", "raw_begin", "raw_end"):
            recs[f] += pos
        parts.append((recs, bb[:n] + pos, be[:n] + pos))
        pos += consumed.value
    if not parts:
        return np.zeros(0, dtype=RECORD), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))


def _reader_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    v = os.environ.get("STAG_THREADS", "")
    if v.isdigit() and int(v) > 0:
        n = int(v)
    return max(1, min(n, 8))       # (eight copies out of the page cache side by side outrun the PCIe link)


def onehot_text_batches(path, batch_rows=1 << 14, chunk_bytes=1 << 25, ring_slots=3, slow_line=None, threads=None):
    """Yields (ids, text, body_begin, body_end, raw) for the lines of a -s file WITHOUT parsing their fields:
    text is a uint8 view (of a page-locked buffer, valid until ring_slots - 1 further batches have been produced)
    that holds whole lines, line i's fields are text[body_begin[i]:body_end[i]] (see stag_onehot_lines), and
    raw(i) returns line i as the str the reference would iterate over (for the slow path).  A batch is the whole
    lines of one chunk of the file, at most batch_rows of them.  slow_line(str) -> (id, bool array) names the
    lines the line finder flags (they end in a non-ASCII byte, which may be white space for str.rstrip).
    A regular file is cut into chunks of whole lines that a pool of `threads` threads (default: the job's cores, at
    most 8) reads into page-locked buffers and finds the lines of, a few chunks ahead of the consumer; what comes
    out is in file order.  Anything else (a FIFO) is read by one thread, one chunk ahead."""
    try:
        st = os.stat(path)
        regular = stat.S_ISREG(st.st_mode)
    except OSError:
        regular = False
    if not regular:
        yield from _onehot_text_batches_stream(path, batch_rows, chunk_bytes, ring_slots, slow_line)
        return
    if st.st_size == 0:
        return
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or _reader_threads()
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    starts = _line_chunk_starts(mm, max(int(chunk_bytes), 64))
    del mm
    n_chunks = len(starts) - 1
    threads = max(1, min(threads, n_chunks))
    cap = max(b - a for a, b in zip(starts[:-1], starts[1:])) + 64       # (room for the GPU copy's padding)
    cap = (cap + (1 << 20) - 1) >> 20 << 20                              # (sizes that a later reader asks for again)
    ring = _PinnedRing(threads + ring_slots + 1, cap)
    n_buf = len(ring.bufs)
    fd = os.open(path, os.O_RDONLY)
    pool = ThreadPoolExecutor(threads, thread_name_prefix="stag-onehot-reader")
    free, held, pending = list(range(n_buf)), [], []
    try:
        nxt = 0
        while nxt < n_chunks or pending:
            while nxt < n_chunks and len(pending) < threads and free:
                i = free.pop(0)
                ring.turn = i - 1
                buf = ring.next()
                pending.append((i, buf, starts[nxt + 1] - starts[nxt],
                                pool.submit(_read_and_find_lines, fd, starts[nxt], starts[nxt + 1], buf)))
                nxt += 1
            i, buf, have, fut = pending.pop(0)
            recs_all, bb_all, be_all = fut.result()
            held.append(i)
            while len(held) > ring_slots:
                free.append(held.pop(0))
            for r0 in range(0, len(recs_all), batch_rows):
                recs = recs_all[r0:r0 + batch_rows]
                lo, hi = int(recs["raw_begin"][0]), int(recs["raw_end"][-1])
                text = buf[lo:hi]
                mv = memoryview(text)
                begins = (recs["id_begin"] - lo).tolist()
                ends = (recs["id_begin"] + recs["id_len"] - lo).tolist()
                ids = [str(mv[b:e], "utf-8") for b, e in zip(begins, ends)]

                def raw(i, mv=mv, rb=recs["raw_begin"] - lo, re_=recs["raw_end"] - lo):
                    return str(mv[int(rb[i]):int(re_[i])], "utf-8")

                for i in np.flatnonzero(recs["flag"] == FALLBACK):
                    ids[i] = slow_line(raw(int(i)))[0]
                yield ids, text, bb_all[r0:r0 + batch_rows] - lo, be_all[r0:r0 + batch_rows] - lo, raw
    finally:
        for _, _, _, fut in pending:
            fut.cancel()
        pool.shutdown(wait=True)
        os.close(fd)
        ring.release()


def _onehot_text_batches_stream(path, batch_rows, chunk_bytes, ring_slots, slow_line):
    """onehot_text_batches for something that cannot be read at an offset: one reader thread, one chunk ahead."""
    L = lib()
    cap = max(int(chunk_bytes), 64)
    chunks = _LineChunks(path, _PinnedRing(ring_slots + 2, cap), cap)
    try:
        for buf, have in chunks:
            pos = 0
            while pos < have:
                recs = np.zeros(batch_rows, dtype=RECORD)
                bb = np.zeros(batch_rows, dtype=np.int64)
                be = np.zeros(batch_rows, dtype=np.int64)
                consumed = ctypes.c_int64(0)
                view = buf[pos:have]
                # final: the chunk ends at a line end (or at the end of the file), nothing is left over
                n = L.stag_onehot_lines(_ptr(view), len(view), 1, batch_rows, _ptr(recs), _ptr(bb), _ptr(be),
                                        ctypes.byref(consumed))
                if n <= 0 or consumed.value <= 0:
                    raise ValueError("stag_onehot_lines: bad arguments")
                recs = recs[:n]
                text = view[:consumed.value]
                mv = memoryview(text)
                begins = recs["id_begin"].tolist()
                ends = (recs["id_begin"] + recs["id_len"]).tolist()
                ids = [str(mv[b:e], "utf-8") for b, e in zip(begins, ends)]

                def raw(i, mv=mv, rb=recs["raw_begin"], re_=recs["raw_end"]):
                    return str(mv[int(rb[i]):int(re_[i])], "utf-8")

                for i in np.flatnonzero(recs["flag"] == FALLBACK):
                    ids[i] = slow_line(raw(int(i)))[0]
                yield ids, text, bb[:n], be[:n], raw
                pos += consumed.value
    finally:
        chunks.close()


def format_onehot(ids, bits, n_features, as_view=False):
    """The -S lines (classify.py:160-162) for packed rows, as bytes (as_view: as a memoryview of the buffer they were
    written into, for a caller that only hands them to write(): the text is 2 bytes per feature, and one more copy
    of it costs more than producing it)."""
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    n = len(ids)
    enc = [s.encode("utf-8") for s in ids]
    id_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=id_off[1:])
    blob = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)
    cap = int(id_off[-1]) + n * (2 * n_features + 2)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    w = bits.shape[1] if bits.ndim == 2 else 0
    m = lib().stag_onehot_format(_ptr(bits), n, n_features, w, _ptr(blob), _ptr(id_off), _ptr(out), cap)
    if m < 0:
        raise RuntimeError("stag_onehot_format: buffer too small (%d)" % -m)
    return memoryview(out)[:m] if as_view else out[:m].tobytes()
