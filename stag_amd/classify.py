"""``classify``: drop-in for stag/classify.py:131-207 (zellerlab/stag 0.8.2) on MI355X.

Same signature, same files, same return value ``(alignment_length, rows)``.  What differs is
where the work happens: the per-sequence Python loop of the reference (classify.py:155-158 ->
classify_seq) becomes batches through the HIP kernels (stag_amd.engine.Database); only text
parsing and text formatting stay on the host.  ``threads`` is accepted and, as in the
reference (classify.py:94), only reaches ``cmalign --cpu``.
"""
import collections
import contextlib
import os
import shutil
import sys
import tempfile
import time

import numpy as np

from . import __version__ as tool_version
from . import _lib, align, ingest, parallel, stream, table_cache
from .databases import load_db, open_stagdb
from .engine import Database, raise_for_status
from .tables import pack_db

OUT_HEADER = ["sequence", "taxonomy", "full_taxonomy", "selected_level",
              "prob_from_classifiers", "prob_per_level", "n_aligned_characters"]
READER_BATCH = 1 << 14


def parse_onehot_line(ali_line):
    """One line of a -s file (classify.py:19-20): id, bool[L]; any non-zero integer is True."""
    gene_id, *fields = ali_line.rstrip().split("\t")
    n = len(fields)
    body = ali_line.rstrip()[len(gene_id) + 1:] if n else ""
    if len(body) == 2 * n - 1:                       # every field is a single character: no int() per field
        raw = np.frombuffer(body.encode("latin-1"), dtype=np.uint8)[0::2]
        if ((raw == 48) | (raw == 49)).all():
            return gene_id, raw == 49
    return gene_id, np.array(list(map(int, fields)), dtype=bool)


def alignment_reader(aligned_sequences):
    """classify.py:16-20."""
    with open(aligned_sequences, "r") as align_in:
        for ali_line in align_in:
            yield parse_onehot_line(ali_line)


def _in_flight(pool):
    """Batches handed to the devices before the oldest one's result is taken: one per device and one more being
    prepared for each.  (Also with a single device: its worker thread is inside the library, copying and computing,
    while this thread turns the previous batch into rows and the readers fetch the next; results are still taken
    in input order, so files, rows and the first error are those of the reference's one loop.)"""
    return 2 * len(pool.dbs)


def _first_things_first(batches, pending, finish_oldest):
    """The batches of a reader whose own failure (an identifier mismatch: sys.exit; a malformed line; an aligner that
    failed) must not overtake what it had already handed out: the batches still in flight are finished first -- one
    of them may hold an earlier failing sequence, which is where the reference's single loop stops."""
    it = iter(batches)
    while True:
        try:
            item = next(it)
        except StopIteration:
            return
        except BaseException:
            while pending:
                finish_oldest()
            raise
        yield item


def _classify_prealigned(pool, aligned_sequences, tagged, state):
    """-s path: rows arrive already one-hot encoded (any bit pattern).  The file is parsed in C
    (stag_amd.ingest.onehot_batches) straight into packed rows; lines it does not take literally
    go through parse_onehot_line.  One parser; the batches go to the pool's devices in input order.
    Under torchrun the batches of the stream are dealt round-robin to the ranks (parallel.owner_of_batch)."""
    world, rank = parallel.world_rank()
    db = pool.first
    L = db.n_features
    pending = collections.deque()

    def finish_oldest():
        b, ids, fut = pending.popleft()
        state["batch"] = b
        res = fut.result()
        tagged.append((b, db.result_rows(res[0] if isinstance(res, tuple) else res, ids, long_out=state["long_rows"])))

    if getattr(db, "classify_onehot_text", None) is not None:
        # the GPU parses the fields (k_parse_onehot): the host only finds the lines and their ids
        batches = ingest.onehot_text_batches(aligned_sequences, batch_rows=READER_BATCH, ring_slots=_in_flight(pool) + 2,
                                             slow_line=parse_onehot_line)
        for b, (ids, text, bb, be, raw) in enumerate(_first_things_first(batches, pending, finish_oldest)):
            if not state["alignment_length"]:
                state["alignment_length"] = len(parse_onehot_line(raw(0))[1])
            if parallel.owner_of_batch(b, world) != rank:
                continue
            pending.append((b, ids, pool.submit(lambda d, t=text, bb=bb, be=be, raw=raw:
                                                d.classify_onehot_text(t, bb, be, parse_onehot_line, raw))))
            while len(pending) >= _in_flight(pool):
                finish_oldest()
        while pending:
            finish_oldest()
        return
    batches = ingest.onehot_batches(aligned_sequences, L, db.words_per_row, batch_rows=READER_BATCH,
                                    slow_line=parse_onehot_line)
    for b, (ids, bits, n_fields) in enumerate(_first_things_first(batches, pending, finish_oldest)):
        if not state["alignment_length"]:
            state["alignment_length"] = int(n_fields[0])
        if parallel.owner_of_batch(b, world) != rank:
            continue
        status = np.where(n_fields == L, _lib.SEQ_OK, _lib.SEQ_LENGTH_MISMATCH).astype(np.int32)
        pending.append((b, ids, pool.submit(lambda d, bits=bits, status=status: d.classify_bits(bits, status))))
        while len(pending) >= _in_flight(pool):
            finish_oldest()
    while pending:
        finish_oldest()


def _report_backtranslation_failure(status, ids, text, offs, gene):
    """The message protein2gene_alignment writes (align.py:70-73) before the reference fails on
    that sequence -- only if it is the first failing sequence, as the reference stops there."""
    bad = np.flatnonzero((status != _lib.SEQ_OK) & (status != _lib.SEQ_FILTERED))
    if bad.size and status[bad[0]] == _lib.SEQ_BACKTRANSLATE:
        i = int(bad[0])
        prot = np.asarray(text[offs[i]:offs[i + 1]])
        n_res = int((prot != ord("-")).sum())
        sys.stderr.write("Error, length of genes/alignment is not correct")
        sys.stderr.write(" (protein: " + str(n_res * 3) + ", gene: " + str(int(gene[1][i + 1] - gene[1][i])) + ")\n")
        sys.stderr.write(" ID: " + ids[i] + "\n")


def classify(database, fasta_input=None, protein_fasta_input=None, verbose=3, threads=1, output=None,
             long_out=False, current_tool_version=tool_version,
             aligned_sequences=None, save_ali_to_file=None, min_perc_state=0, internal_call=False, device=None,
             packed_rows_out=None, concat_source=None, cache=True):
    """The reference's arguments, plus four of this build's own: `device` (GPU index),
    `packed_rows_out` (a list that receives (ids, packed rows) of every batch written to
    save_ali_to_file), `concat_source` (marker-gene rows to concatenate on the GPU instead of
    reading aligned_sequences; see classify_genome.concat_on_device) and `cache` (False: do not
    look the database up in, or add it to, the packed-table cache)."""
    t0 = time.time()
    cached = table_cache.load(database, protein_fasta_input, aligned_sequences) if cache else None
    if cached is not None:
        hmm_file_path, use_cmalign, tables, db_tool_version = cached
    else:
        loaded = load_db(database, protein_fasta_input=protein_fasta_input, aligned_sequences=aligned_sequences)
        hmm_file_path, use_cmalign, taxonomy, tax_function, classifiers, db_tool_version = loaded
        tables = pack_db(taxonomy, tax_function, classifiers)
        del classifiers
        if cache and table_cache.cache_dir() and os.path.isfile(database):
            with open_stagdb(database) as db_in:
                use_proteins = bool(db_in.read("align_protein")[0])
            table_cache.store(database, tables, hmm_file_path, use_cmalign, use_proteins, db_tool_version)
    world, rank = parallel.world_rank()
    # one process can drive several GPUs (STAG_DEVICES): tables replicated, batches dealt in input order
    devices = [device] if device is not None else (parallel.devices_from_env() if world == 1 else [parallel.local_device()])
    pool = parallel.DevicePool(lambda d: Database(tables, device=d), devices)
    db = pool.first
    time_after_loading = time.time()
    if verbose > 2:
        sys.stderr.write("Load database: " + str("{0:.2f}".format(time_after_loading - t0)) + " sec\n")

    # the long fields (14 float reprs per row) are formatted only when they will be printed (classify.py:182-184)
    state = {"alignment_length": None, "batch": 0, "long_rows": bool(long_out) and not internal_call}
    tagged = []          # (batch index, rows) of the batches this rank classified
    n_pass = n_not_pass = 0
    error = None
    try:
        if concat_source is not None:      # classify_genome.py:364-388 + :494 without the text file
            state["alignment_length"] = int(sum(concat_source["gene_len"]))
            if rank == 0 and len(concat_source["ids"]):
                res = db.classify_concat(concat_source["gene_len"], concat_source["gene_rows"], concat_source["src_row"])
                tagged.append((0, db.result_rows(res, concat_source["ids"], long_out=state["long_rows"])))
        elif aligned_sequences:
            _classify_prealigned(pool, aligned_sequences, tagged, state)
        else:
            batches = align.aligned_batches(fasta_input, protein_fasta_input, hmm_file_path, use_cmalign,
                                            threads, verbose, ring_slots=_in_flight(pool) + 2)
            alignment_out = open(save_ali_to_file, "wb") if save_ali_to_file and rank == 0 else contextlib.nullcontext()
            want_bits = bool(save_ali_to_file) and rank == 0
            pending = collections.deque()

            def run(d, text, offs, gene, lead_trail):
                if gene is None:
                    return d.classify_a2m_text(text, offs, min_perc_state=min_perc_state, want_bits=want_bits,
                                               lead_trail=lead_trail)
                # protein database: back-translation happens on the GPU (align.py:63-94)
                return d.classify_protein_a2m_text(text, offs, gene[0], gene[1], min_perc_state=min_perc_state,
                                                   want_bits=want_bits)

            def finish_oldest():
                """What the reference does after classifying the sequences of a batch, in input order."""
                nonlocal n_pass, n_not_pass
                b, mine, ids, text, offs, gene, fut = pending.popleft()
                state["batch"] = b
                res = fut.result()
                _report_backtranslation_failure(res.status, ids, text, offs, gene)
                failed = np.flatnonzero((res.status != _lib.SEQ_OK) & (res.status != _lib.SEQ_FILTERED))
                if failed.size:
                    # the reference stops at the first failing sequence; by then it has written the -S
                    # lines of the sequences before it (classify.py:158-162)
                    before = np.flatnonzero(res.status[:failed[0]] == _lib.SEQ_OK)
                    if want_bits and len(before):
                        alignment_out.write(ingest.format_onehot([ids[i] for i in before], res.bits[before], db.n_features, as_view=True))
                    raise_for_status(res.status, ids)
                passed = np.flatnonzero(res.status == _lib.SEQ_OK)
                n_pass += len(passed)
                n_not_pass += len(ids) - len(passed)
                if len(passed) and not state["alignment_length"]:
                    state["alignment_length"] = 5 * int(res.n_char[passed[0]])
                if mine:
                    tagged.append((b, db.result_rows(res, ids, long_out=state["long_rows"])))
                if want_bits and len(passed):      # classify.py:160-162
                    alignment_out.write(ingest.format_onehot([ids[i] for i in passed], res.bits[passed], db.n_features, as_view=True))
                    if packed_rows_out is not None:
                        packed_rows_out.append(([ids[i] for i in passed], res.bits[passed]))

            with alignment_out:
                for b, (ids, text, offs, gene, lead_trail) in enumerate(_first_things_first(batches, pending, finish_oldest)):
                    mine = parallel.owner_of_batch(b, world) == rank
                    if not mine and not want_bits and (state["alignment_length"] or pending):
                        continue
                    pending.append((b, mine, ids, text, offs, gene,
                                    pool.submit(lambda d, t=text, o=offs, g=gene, lt=lead_trail: run(d, t, o, g, lt))))
                    while len(pending) >= _in_flight(pool):
                        finish_oldest()
                while pending:
                    finish_oldest()
            if verbose > 3 and world == 1:
                print(f" Number of sequences that pass the filter: {n_pass}", file=sys.stderr)
                print(f" Number of sequences that do not pass the filter: {n_not_pass}", file=sys.stderr)
    except Exception as e:       # re-raised on every rank by gather_in_order
        error = e
    finally:
        pool.close()
    if error is not None:
        os.remove(hmm_file_path)
    list_to_print = parallel.gather_in_order(tagged, error, state["batch"])
    if rank != 0:
        os.remove(hmm_file_path)
        return state["alignment_length"], []

    if verbose > 2:
        sys.stderr.write("Classify sequences: " + str("{0:.2f}".format(time.time() - time_after_loading)) + " sec\n")

    os.remove(hmm_file_path)       # classify.py:169

    # ---- output, classify.py:172-207
    if output:
        outfile = tempfile.NamedTemporaryFile(delete=False, mode="w")
        os.chmod(outfile.name, 0o644)
    else:
        outfile = sys.stdout
    out_header = OUT_HEADER
    if not long_out or internal_call:
        out_header = out_header[:2]
        list_to_print = [item[:2] for item in list_to_print]     # (already two fields each, see long_rows)
    if not internal_call:
        try:
            print(*out_header, sep="\t", file=outfile)
            for line in list_to_print:
                print(*line, sep="\t", file=outfile)
            if output:
                try:
                    outfile.flush()
                    os.fsync(outfile.fileno())
                except Exception:
                    sys.stderr.write("[E::main] Error: failed to save the result\n")
                    sys.exit(1)
        finally:
            if output:
                outfile.close()      # the reference also closes sys.stdout here; we leave it open
            else:
                outfile.flush()
        if output:
            try:
                shutil.move(outfile.name, output)
            except Exception:
                sys.stderr.write("[E::main] Error: failed to save the profile\n")
                sys.stderr.write("[E::main] you can find the file here:\n" + outfile.name + "\n")
                sys.exit(1)
    return state["alignment_length"], list_to_print


def classify_a2m(database, a2m_file, output=None, long_out=False, verbose=3, min_perc_state=0, threads=None,
                 device=None, cache=True):
    """This build's own entry (no counterpart among the reference's commands; `stag classify --a2m FILE`): the
    classification path from an alignment that is already on disk -- the A2M text `hmmalign | esl-reformat a2m -`
    wrote (align.py:127-141), for a protein database the codon alignment -- so that the aligner, which stays the
    reference's external CPU step, can be run and timed apart from this path.  The file is scanned by a pool of
    threads (stag_amd.stream), batches go to the GPU(s) of STAG_DEVICES without their flanking gap columns, and
    the output is classify()'s: same header, same lines, same temp-file-then-move.  Under torchrun every rank
    scans only its own chunks of the file.  Returns (alignment_length, number of result lines)."""
    t0 = time.time()
    cached = table_cache.load(database, None, a2m_file) if cache else None
    if cached is not None:
        hmm_file_path, _, tables, _ = cached
    else:
        hmm_file_path, _, taxonomy, tax_function, classifiers, _ = load_db(database, aligned_sequences=a2m_file)
        tables = pack_db(taxonomy, tax_function, classifiers)
        del classifiers
    os.remove(hmm_file_path)        # (no aligner runs here)
    world, rank = parallel.world_rank()
    devices = [device] if device is not None else (parallel.devices_from_env() if world == 1 else [parallel.local_device()])
    pool = parallel.DevicePool(lambda d: Database(tables, device=d), devices)
    if verbose > 2:
        sys.stderr.write("Load database: " + str("{0:.2f}".format(time.time() - t0)) + " sec\n")
    t1 = time.time()
    error, tagged, state = None, [], {"alignment_length": None, "batch": 0}
    try:
        if world == 1 and not long_out:
            # two-column lines straight from the result arrays: no Python object per sequence
            result = stream.classify_a2m_file(pool, a2m_file, output=output, threads=threads, min_perc_state=min_perc_state)
            if verbose > 2:
                sys.stderr.write("Classify sequences: " + str("{0:.2f}".format(time.time() - t1)) + " sec\n")
            return result
        db = pool.first
        pending = collections.deque()

        def finish_oldest():
            b, batch, fut = pending.popleft()
            state["batch"] = b
            res = fut.result()
            passed = np.flatnonzero(res.status == _lib.SEQ_OK)
            if len(passed) and not state["alignment_length"]:
                state["alignment_length"] = 5 * int(res.n_char[passed[0]])
            tagged.append((b, db.result_rows(res, batch.ids(), long_out=long_out)))

        for b, batch in enumerate(stream.a2m_file_batches(a2m_file, threads=threads, keep=_in_flight(pool) + 2,
                                                          world=world, rank=rank)):
            if batch is None or batch.n == 0:
                continue
            pending.append((b, batch, pool.submit(lambda d, x=batch: d.classify_a2m_text(
                x.text, x.offs, min_perc_state=min_perc_state, lead_trail=x.lead_trail))))
            while len(pending) > _in_flight(pool):
                finish_oldest()
        while pending:
            finish_oldest()
    except Exception as e:       # re-raised on every rank by gather_in_order
        error = e
    finally:
        pool.close()
    rows = parallel.gather_in_order(tagged, error, state["batch"])
    if rank != 0:
        return state["alignment_length"], 0
    if verbose > 2:
        sys.stderr.write("Classify sequences: " + str("{0:.2f}".format(time.time() - t1)) + " sec\n")
    header = OUT_HEADER if long_out else OUT_HEADER[:2]
    sink = tempfile.NamedTemporaryFile(delete=False, mode="w") if output else sys.stdout
    if output:
        os.chmod(sink.name, 0o644)
    try:
        print(*header, sep="\t", file=sink)
        for line in rows:
            print(*line, sep="\t", file=sink)
        sink.flush()
        if output:
            os.fsync(sink.fileno())
    finally:
        if output:
            sink.close()
    if output:
        shutil.move(sink.name, output)
    return state["alignment_length"], len(rows)
