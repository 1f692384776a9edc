"""``classify_genome``: the post-alignment parts of stag/classify_genome.py (zellerlab/stag 0.8.2).

In scope: per-marker-gene ``classify(internal_call=True)``
(annotate_MGs, :307-341), merge_gene_predictions (:343-362), concat_alignments (:364-388) and
the final ``classify`` on the concatenated database (:494).  The reference forks a process
pool for the marker genes (:323); a HIP context must not be created before a fork, and the
GPU is one shared device anyway, so here the marker-gene databases run one after another in
this process, each as GPU batches.

Out of scope (they stay the reference's external CPU steps): gene calling with prodigal
(:54-96) and marker-gene detection with hmmsearch (:102-305).  Genomes therefore enter through
``-G`` (a JSON file {marker: [genes.fna, proteins.faa]}, :423-424), which is what those steps
produce.

The function names and arguments are the reference's (they are the interface its callers use);
the bodies are this build's own and are pinned by the reference's outputs in
tests/golden/concat.json and tests/golden/genome_db.json.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

from . import __version__ as tool_version
from .classify import classify
from .databases import load_genome_DB

NO_PROTEIN = "no_protein"      # a gene database aligned at the nucleotide level (classify_genome.py:449-450)


def _drop_last_token(name, sep):
    head, _, _ = name.rpartition(sep)
    return head if sep in name else name


def _genome_of(seq_id):
    """Genome a marker-gene sequence is filed under by merge_gene_predictions (classify_genome.py:352-355):
    the part before '##' without its last '_' token -- or its last '.' token when it has no '_'."""
    genome = seq_id.split("##")[0]
    return _drop_last_token(genome, "_" if "_" in genome else ".")


def _genome_of_ali_id(seq_id):
    """The same for concat_alignments (classify_genome.py:372-374), which picks the separator by looking at the
    WHOLE id, marker-gene name included: "plain.3##MG_4" stays "plain.3" there while merge_gene_predictions
    files it under "plain".  Kept as the reference has it (golden/concat.json holds that very case)."""
    return _drop_last_token(seq_id.split("##")[0], "_" if "_" in seq_id else ".")


def annotate_MGs(MGS, database_files, database_base_path, dir_ali, procs=2, packed=None):
    """classify_genome.py:307-341 without the fork pool: {sequence id: assigned lineage} over all marker
    genes.  `packed` (a dict, optional) receives per marker gene (alignment length, [(ids, packed rows), ...])
    of what went to dir_ali/<gene>, which is what concat_on_device works from."""
    todo = []
    for mg, (fna, faa) in MGS.items():
        if fna and os.path.isfile(fna) and os.path.getsize(fna) > 0:      # genes were found for this marker
            todo.append((mg, fna, None if faa in (None, "", NO_PROTEIN) else faa))
    if not todo:
        raise ValueError("No marker genes found!")
    absent = [mg for mg, _, _ in todo if not os.path.isfile(os.path.join(database_base_path, mg))]
    if absent:
        raise ValueError(f"Error: file for gene database {os.path.join(database_base_path, absent[0])} is missing")
    lineage_of = {}
    for mg, fna, faa in todo:
        rows_of_gene = []
        # cache=False: these databases sit in a fresh temp directory per run, a cache entry could never be hit again
        ali_len, short_rows = classify(os.path.join(database_base_path, mg), fasta_input=fna, protein_fasta_input=faa,
                                       save_ali_to_file=os.path.join(dir_ali, mg), internal_call=True, verbose=1,
                                       threads=procs, packed_rows_out=rows_of_gene, cache=False)
        lineage_of.update(short_rows)
        if packed is not None:
            packed[mg] = (ali_len, rows_of_gene)
    return lineage_of


def merge_gene_predictions(genome_files, mgs_list, all_classifications, verbose, threads, output, long_out,
                           keep_all_genes, full_genomes=True):
    """classify_genome.py:343-362: genes_predictions/<basename(genome)> with one line
    `sequence id <tab> marker gene <tab> lineage` per classified marker-gene sequence, in input order."""
    lines_of = {}
    for seq_id, lineage in all_classifications.items():
        seq_id = seq_id.rstrip()
        _, marker = seq_id.split("##")              # exactly one '##', as the reference's unpacking demands
        lines_of.setdefault(_genome_of(seq_id), []).append(f"{seq_id}\t{marker}\t{lineage}\n")
    target_dir = os.path.join(output, "genes_predictions")
    os.makedirs(target_dir, exist_ok=True)
    for genome, lines in lines_of.items():
        with open(os.path.join(target_dir, os.path.basename(genome)), "w") as fh:
            fh.writelines(lines)


def concat_alignments(genome_files, ali_dir, gene_order, ali_lengths, full_genomes=True):
    """classify_genome.py:364-388 as text (the fallback of concat_on_device): per genome, in order of first
    appearance, the 0/1 fields of its marker genes in gene_order; a gene the genome lacks is
    ali_lengths[gene] zeros, and of several lines for one (genome, gene) the last one counts.
    Returns the path of a temp file the caller removes."""
    blank = ["\t".join(["0"] * int(ali_lengths[mg])) for mg in gene_order]
    fields_of = {}                                   # genome -> the field text of each gene, in gene_order
    for slot, mg in enumerate(gene_order):
        path = os.path.join(ali_dir, mg)
        if not os.path.exists(path):
            continue
        with open(path) as fh:                       # text mode, as the reference reads it (universal newlines)
            for line in fh:
                seq_id, _, fields = line.strip().partition("\t")
                fields_of.setdefault(_genome_of_ali_id(seq_id), list(blank))[slot] = fields
    out = tempfile.NamedTemporaryFile(delete=False, mode="w")
    with out:
        os.chmod(out.name, 0o644)
        for genome, per_gene in fields_of.items():
            out.write("\t".join([genome] + per_gene) + "\n")
    return out.name


def concat_on_device(packed, gene_order, ali_lengths):
    """What concat_alignments (classify_genome.py:364-388) builds, as an index instead of text: the
    genomes in order of first appearance (genes in gene_order, sequences in file order) and, per
    genome and gene, which packed row is its alignment (the last one wins, as the dict assignment
    of :376 does; -1 = the gene is missing and is zero-filled).  Returns None when a gene's rows do
    not have the length the database announces (the text path then reproduces the reference's
    failure)."""
    slot_of_genome, src = {}, []
    gene_rows, gene_len = [], []
    for pos, mg in enumerate(gene_order):
        length = int(ali_lengths[mg])
        gene_len.append(length)
        n, batches = packed.get(mg, (None, []))
        if batches and n != length:
            return None
        rows = [b for _, b in batches]
        gene_rows.append(np.concatenate(rows) if rows else np.zeros((0, 4), dtype=np.uint32))
        r = 0
        for ids, _ in batches:
            for seq_id in ids:
                j = slot_of_genome.setdefault(_genome_of_ali_id(seq_id), len(slot_of_genome))
                if j == len(src):
                    src.append([-1] * len(gene_order))
                src[j][pos] = r
                r += 1
    return {"ids": list(slot_of_genome), "gene_len": gene_len, "gene_rows": gene_rows,
            "src_row": np.array(src, dtype=np.int32).reshape(len(src), len(gene_order))}


def store_marker_sequences(marker_sequences, outdir, genome_files_available=True):
    """classify_genome.py:391-410: MG_sequences/<marker>.fna and .faa -- moved there when this run produced
    them from genome files, symlinked when the user passed them with -G, an empty file when there is none --
    and marker_sequences[marker] re-pointed at them."""
    os.makedirs(outdir, exist_ok=True)
    place = shutil.move if genome_files_available else os.symlink
    for marker in list(marker_sequences):
        kept = []
        for source, ext in zip(marker_sequences[marker], (".fna", ".faa")):
            if ext == ".faa" and source == NO_PROTEIN:
                kept.append(NO_PROTEIN)             # nothing to store: the gene is aligned as nucleotides
                continue
            target = os.path.join(outdir, marker + ext)
            try:
                if source:
                    place(os.path.abspath(source), target)
                else:
                    open(target, "w").close()
            except Exception as e:
                raise ValueError(f"[E::main] Error: failed to save the marker gene sequences\n{e}")
            kept.append(target)
        marker_sequences[marker] = kept


def classify_genome(database, genome_files=None, marker_genes=None, verbose=None,
                    threads=1, output=None, long_out=False, keep_all_genes=False):
    """classify_genome.py:413-504 for the -G entry; same output tree (MG_sequences/, MG_ali/,
    genes_predictions/, genome_annotation)."""
    verbose = 3 if verbose is None else verbose
    if verbose > 2:
        sys.stderr.write("Unzip the database\n")
    database_files, temp_dir, gene_thresholds, gene_order, ali_lengths, concat_ali_stag_db = \
        load_genome_DB(database, tool_version, verbose)
    try:
        if marker_genes:
            with open(marker_genes[0]) as fh:
                MGS = json.load(fh)
        elif genome_files:
            raise NotImplementedError(
                "[E::main] gene calling (prodigal) and marker-gene detection (hmmsearch) are the reference's "
                "external CPU steps and are not part of this build: run them with the reference and pass the "
                "marker genes with -G")
        else:
            raise ValueError("[E::main] Error: no input")
        store_marker_sequences(MGS, os.path.join(output, "MG_sequences"), genome_files_available=bool(genome_files))
        if verbose > 2:
            sys.stderr.write("Taxonomically annotate single marker genes\n")
        align_dir = os.path.join(output, "MG_ali")
        os.makedirs(align_dir, exist_ok=True)
        packed = {}
        all_classifications = annotate_MGs(MGS, database_files, temp_dir, align_dir, procs=threads, packed=packed)
        input_files = genome_files if genome_files else marker_genes
        merge_gene_predictions(input_files, list(database_files), all_classifications, verbose, threads, output,
                               long_out, keep_all_genes, full_genomes=bool(genome_files))
        if verbose > 2:
            sys.stderr.write("Taxonomically annotate genomes\n")
        # the concatenated rows are assembled on the GPU from the rows the per-gene pass produced;
        # the text round trip of the reference (:364-388, 2 bytes per feature and genome) is the fallback
        source = concat_on_device(packed, gene_order, ali_lengths)
        if source is not None:
            classify(concat_ali_stag_db, aligned_sequences="<marker gene rows>", concat_source=source,
                     output=os.path.join(output, "genome_annotation"), long_out=long_out, verbose=verbose, cache=False)
        else:
            file_ali = concat_alignments(input_files, align_dir, gene_order, ali_lengths, full_genomes=bool(genome_files))
            try:
                classify(concat_ali_stag_db, aligned_sequences=file_ali, output=os.path.join(output, "genome_annotation"),
                         long_out=long_out, verbose=verbose, cache=False)
            finally:
                os.remove(file_ali)
    finally:
        shutil.rmtree(temp_dir, ignore_errors=True)
