"""Alignment stream driver: mirror of stag/align.py:98-180 (zellerlab/stag 0.8.2) feeding the GPU.

hmmalign / cmalign and esl-reformat stay the reference's external CPU step.  What changes is
what happens to their output: instead of one-hot encoding every record in a Python loop
(align.py:37-58), records are collected into batches of aligned text that the HIP encode
kernel converts and filters on the GPU.  Protein databases back-translate each record to a
codon alignment first (align.py:63-94) through the C ABI (stag_backtranslate).
"""
import ctypes
import shlex
import subprocess
import sys

from . import _lib, ingest
from .helpers import is_tool, read_fasta

BATCH_RECORDS = 1 << 16      # records handed to the GPU at a time


def protein2gene_alignment(gene_id, protein_alignment, gene_sequence, check_length=True):
    """align.py:63-94 via libstag_hip's host entry point.  None when the lengths do not fit."""
    if not (protein_alignment.isascii() and gene_sequence.isascii()):
        return _protein2gene_alignment_py(gene_id, protein_alignment, gene_sequence, check_length)
    p = protein_alignment.encode("ascii")
    g = gene_sequence.encode("ascii")
    out = ctypes.create_string_buffer(3 * len(p) + 4)
    n = ctypes.c_int64(0)
    _lib.check(_lib.lib().stag_backtranslate(p, len(p), g, len(g), int(bool(check_length)), out, ctypes.byref(n)))
    if n.value < 0:
        sys.stderr.write("Error, length of genes/alignment is not correct")
        n_res = len(p) - p.count(b"-")
        sys.stderr.write(" (protein: " + str(n_res * 3) + ", gene: " + str(len(g)) + ")\n")
        sys.stderr.write(" ID: " + gene_id + "\n")
        return None
    return out.raw[:n.value].decode("ascii")


def _protein2gene_alignment_py(gene_id, protein_alignment, gene_sequence, check_length):
    """The same for strings with characters outside ASCII (str.isupper / islower decide, as in the reference)."""
    n_res = len(protein_alignment) - protein_alignment.count("-")
    if check_length and len(gene_sequence) not in (3 * n_res, 3 * n_res + 3):
        sys.stderr.write("Error, length of genes/alignment is not correct")
        sys.stderr.write(" (protein: " + str(n_res * 3) + ", gene: " + str(len(gene_sequence)) + ")\n")
        sys.stderr.write(" ID: " + gene_id + "\n")
        return None
    out, at = [], 0
    for ch in protein_alignment:
        if ch == "-":
            out.append("---")
        elif ch.isupper():
            out.append(gene_sequence[at:at + 3])
            at += 3
        elif ch.islower():
            at += 3
    return "".join(out)


def _start_aligner(seq_file, protein_file, hmm_file, use_cmalign, n_threads, verbose):
    """align.py:116-141: tool checks and the hmmalign/cmalign | esl-reformat pipe."""
    if use_cmalign and not is_tool("cmalign"):
        raise ValueError("[E::align] Error: cmalign is not in the path. Please install Infernal.")
    elif not is_tool("hmmalign"):
        raise ValueError("[E::align] Error: hmmalign is not in the path. Please install HMMER3.")
    if not is_tool("esl-reformat"):
        raise ValueError("[E::align] Error: esl-reformat is not in the path. Please install Easel.")

    aligner = f"cmalign --cpu {n_threads}" if use_cmalign else "hmmalign"
    seq_input = protein_file if protein_file else seq_file
    align_cmd = f"{aligner} {hmm_file} {seq_input}"
    if verbose > 4:
        print(f"Command used to align the sequences: {align_cmd}", file=sys.stderr)
    align_proc = subprocess.Popen(shlex.split(align_cmd), stdout=subprocess.PIPE)
    parse_proc = subprocess.Popen(shlex.split("esl-reformat a2m -"), stdin=align_proc.stdout, stdout=subprocess.PIPE)
    return align_proc, parse_proc


def _finish_aligner(align_proc, parse_proc):
    align_proc.stdout.close()
    if align_proc.wait():
        raise ValueError("[E::align] Error. hmmalign/cmalign failed.")
    parse_proc.stdout.close()
    if parse_proc.wait():
        raise ValueError("[E::align] Error. esl-reformat failed.")


def aligned_records(seq_file, protein_file, hmm_file, use_cmalign, n_threads, verbose):
    """The part of align_generator (align.py:98-175) before the one-hot conversion: yields
    (id, nucleotide-aligned string) and checks the tools' exit codes once the stream drains."""
    align_proc, parse_proc = _start_aligner(seq_file, protein_file, hmm_file, use_cmalign, n_threads, verbose)
    if protein_file:
        gene_fh = open(seq_file)
        stream = zip(read_fasta(parse_proc.stdout, head_start=1), read_fasta(gene_fh, is_binary=False, head_start=1))
    else:
        gene_fh = None
        stream = read_fasta(parse_proc.stdout, head_start=1)
    try:
        for item in stream:
            if protein_file:
                (pid, pseq), (gid, gseq) = item
                if pid != gid:
                    sys.stderr.write("[E::align] Error. protein and gene identifiers {} {} don't match.".format(pid, gid))
                    sys.exit(1)
                gseq = protein2gene_alignment(gid, pseq, gseq, check_length=True)
                if gseq is None:   # the reference goes on and fails in convert_alignment(None)
                    raise TypeError("'NoneType' object is not iterable")
            else:
                gid, gseq = item
            yield gid, gseq
    finally:
        if gene_fh:
            gene_fh.close()
    _finish_aligner(align_proc, parse_proc)


def aligned_batches(seq_file, protein_file, hmm_file, use_cmalign, n_threads, verbose, size=None, ring_slots=2):
    """Batches (ids, text uint8, offs int64, gene, lead_trail) of aligned records, parsed in C
    (stag_amd.ingest.fasta_batches: no Python string per record).  gene is None for nucleotide
    databases (text/offs is the A2M alignment, for Database.classify_a2m_text); for protein
    databases text/offs is the protein alignment and gene = (gene text, gene offs) of the same
    records (align.py:148-158), for Database.classify_protein_a2m_text, which back-translates
    on the GPU.  lead_trail (nucleotide databases; None for protein ones): the records come WITHOUT the run of
    '-' at either end and lead_trail [n, 2] says how long those were (ingest.fasta_batches(trim=True)): the gap
    columns around an amplicon read never travel to the GPU.  ring_slots: how many batches of a nucleotide database stay valid at a time (their text sits in a
    ring of page-locked buffers; the caller keeps ring_slots - 2 batches in flight)."""
    size = size or BATCH_RECORDS
    align_proc, parse_proc = _start_aligner(seq_file, protein_file, hmm_file, use_cmalign, n_threads, verbose)
    if not protein_file:
        for ids, text, offs, lead_trail in ingest.fasta_batches(parse_proc.stdout, batch_records=size, pinned=True,
                                                                ring_slots=ring_slots, trim=True):
            yield ids, text, offs, None, lead_trail
        _finish_aligner(align_proc, parse_proc)
        return
    # The two streams are paired record by record, as the reference's zip() does (align.py:148): the batchers cut
    # them independently (batch_records is a target, not a bound), so whatever one side has beyond the other's
    # batch waits for the other side's next batch.
    import numpy as np

    def take(side, stream):
        """More records for `side` = [ids, text, offs] from its stream; False when the stream has ended."""
        nxt = next(stream, None)
        if nxt is None:
            return False
        ids, text, offs = nxt
        if not side[0]:
            side[:] = [list(ids), text, offs]
        else:       # append behind the leftover
            used = int(side[2][-1])
            side[0] = side[0] + list(ids)
            side[1] = np.concatenate([side[1][:used], text])
            side[2] = np.concatenate([side[2], offs[1:] + used])
        return True

    def cut(side, m):
        """The first m records of `side` as (ids, text, offs); `side` keeps the rest."""
        ids, text, offs = side
        head = (ids[:m], text, offs[:m + 1])
        if m < len(ids):
            lo = int(offs[m])
            rest_text = text[lo:int(offs[-1])]
            pad = np.zeros(-len(rest_text) % 16, dtype=np.uint8)
            side[:] = [ids[m:], np.concatenate([rest_text, pad]), offs[m:] - lo]
        else:
            side[:] = [[], None, None]
        return head

    with open(seq_file, "rb") as gene_fh:
        genes = ingest.fasta_batches(gene_fh, batch_records=size, text_mode=True)
        prots = ingest.fasta_batches(parse_proc.stdout, batch_records=size)
        P, G = [[], None, None], [[], None, None]
        p_ended = g_ended = False
        while True:
            # both sides are topped up to a full batch before a pair of batches is cut: two batchers that cut at
            # different records would otherwise stay out of phase and every second batch would be a handful of
            # records that still pays for a whole chain of launches
            while not p_ended and len(P[0]) < size:
                p_ended = not take(P, prots)
            while not g_ended and len(G[0]) < size:
                g_ended = not take(G, genes)
            m = min(len(P[0]), len(G[0]))
            if m == 0:
                break                                  # zip() stops with the shorter stream
            clash = next((j for j in range(m) if P[0][j] != G[0][j]), None)
            if clash is not None:
                # the reference gets this far only if the pairs before it went through (align.py:149-159):
                # let them, then stop as it does
                bad_p, bad_g = P[0][clash], G[0][clash]
                if clash:
                    pids, ptext, poffs = cut(P, clash)
                    gids, gtext, goffs = cut(G, clash)
                    yield pids, ptext, poffs, (gtext, goffs), None
                sys.stderr.write("[E::align] Error. protein and gene identifiers {} {} don't match.".format(bad_p, bad_g))
                sys.exit(1)
            pids, ptext, poffs = cut(P, m)
            gids, gtext, goffs = cut(G, m)
            yield pids, ptext, poffs, (gtext, goffs), None
    _finish_aligner(align_proc, parse_proc)


def batched(records, size=BATCH_RECORDS):
    ids, seqs = [], []
    for gid, s in records:
        ids.append(gid)
        seqs.append(s)
        if len(ids) >= size:
            yield ids, seqs
            ids, seqs = [], []
    if ids:
        yield ids, seqs
