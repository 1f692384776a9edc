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
"""
import json
import os
import pathlib
import shutil
import sys
import tempfile

from . import __version__ as tool_version
from .classify import classify
from .databases import load_genome_DB


def annotate_MGs(MGS, database_files, database_base_path, dir_ali, procs=2, packed=None):
    """classify_genome.py:307-341 without the fork pool: {sequence id: lineage}.  `packed` (a dict)
    receives per marker gene (alignment length, [(ids, packed rows), ...]) of what was written
    to dir_ali, for concat_on_device."""
    found_marker_genes = {
        mg: (fna, faa)
        for mg, (fna, faa) in MGS.items()
        if fna and pathlib.Path(fna).is_file() and pathlib.Path(fna).stat().st_size
    }
    if not found_marker_genes:
        raise ValueError("No marker genes found!")
    for mg in found_marker_genes:
        db = os.path.join(database_base_path, mg)
        if not os.path.isfile(db):
            raise ValueError(f"Error: file for gene database {db} is missing")
    d = dict()
    for mg, (fna, faa) in found_marker_genes.items():
        # "no_protein" marks a gene database aligned at the nucleotide level (:449-450)
        faa = None if faa in (None, "", "no_protein") else faa
        batches = []
        n, predictions = classify(os.path.join(database_base_path, mg), fasta_input=fna, protein_fasta_input=faa,
                                  save_ali_to_file=os.path.join(dir_ali, mg), internal_call=True, verbose=1,
                                  threads=procs, packed_rows_out=batches)
        if packed is not None:
            packed[mg] = (n, batches)
        d.update(predictions)
    return d


def _genome_of(seq_id):
    """classify_genome.py:352-355 (merge_gene_predictions): the part before '##' with its last '_'
    (or '.', when it has no '_') token dropped."""
    genome = seq_id.split("##")[0]
    sep = "_" if "_" in genome else "."
    toks = genome.split(sep)
    return sep.join(toks[:-1] if len(toks) > 1 else toks)


def _genome_of_ali_id(seq_id):
    """classify_genome.py:372-374 (concat_alignments): as above, but the separator is chosen from the
    WHOLE id, marker-gene name included -- "plain.3##MG_4" stays "plain.3" there while
    merge_gene_predictions files it under "plain".  Kept as the reference has it."""
    sep = "_" if "_" in seq_id else "."
    toks = seq_id.split("##")[0].split(sep)
    return sep.join(toks[:-1] if len(toks) > 1 else toks)


def merge_gene_predictions(genome_files, mgs_list, all_classifications, verbose, threads, output, long_out,
                           keep_all_genes, full_genomes=True):
    """classify_genome.py:343-362: writes genes_predictions/<basename(genome)>."""
    outdir = os.path.join(output, "genes_predictions")
    pathlib.Path(outdir).mkdir(exist_ok=True, parents=True)
    merged_predictions = dict()
    for marker_gene, lineage in all_classifications.items():
        genome, mg_id = marker_gene.rstrip().split("##")
        merged_predictions.setdefault(_genome_of(marker_gene.rstrip()), list()).append(
            "\t".join([marker_gene.rstrip(), mg_id, lineage]))
    for genome, predictions in merged_predictions.items():
        with open(os.path.join(outdir, os.path.basename(genome)), "w") as merged_out:
            print(*predictions, sep="\n", file=merged_out, flush=True)


def concat_alignments(genome_files, ali_dir, gene_order, ali_lengths, full_genomes=True):
    """classify_genome.py:364-388: one line per genome, marker genes in gene_order, a missing gene
    is ali_lengths[gene] zeros.  Returns the path of the temp file."""
    all_genes = dict()
    for pos, mg in enumerate(gene_order):
        mg_alignment_file = os.path.join(ali_dir, mg)
        if os.path.exists(mg_alignment_file):
            with open(mg_alignment_file) as align_in:
                for line in align_in:
                    genome, *alignment = line.strip().split("\t")
                    genome = _genome_of_ali_id(genome)
                    if genome not in all_genes:
                        all_genes[genome] = ["\t".join(["0"] * int(ali_lengths[g])) for g in gene_order]
                    all_genes[genome][pos] = "\t".join(alignment)
    with tempfile.NamedTemporaryFile(delete=False, mode="w") as concat_ali_f:
        os.chmod(concat_ali_f.name, 0o644)
        for genome, alignment in all_genes.items():
            print(genome, *alignment, sep="\t", file=concat_ali_f, flush=True)
    return concat_ali_f.name


def concat_on_device(packed, gene_order, ali_lengths):
    """What concat_alignments (classify_genome.py:364-388) builds, as an index instead of text: the
    genomes in order of first appearance (genes in gene_order, sequences in file order) and, per
    genome and gene, which packed row is its alignment (the last one wins, as the dict assignment
    of :378 does; -1 = the gene is missing and is zero-filled).  Returns None when a gene's rows do
    not have the length the database announces (the text path then reproduces the reference's
    failure)."""
    import numpy as np
    genomes, src = {}, []
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
                genome = _genome_of_ali_id(seq_id)
                j = genomes.setdefault(genome, len(genomes))
                if j == len(src):
                    src.append([-1] * len(gene_order))
                src[j][pos] = r
                r += 1
    return {"ids": list(genomes), "gene_len": gene_len, "gene_rows": gene_rows,
            "src_row": np.array(src, dtype=np.int32).reshape(len(src), len(gene_order))}


def store_marker_sequences(marker_sequences, outdir, genome_files_available=True):
    """classify_genome.py:391-410."""
    def store_seqs(source, target, copy_function):
        try:
            if not source:
                open(target, "w").close()
            else:
                copy_function(os.path.abspath(source), target)
        except Exception as e:
            raise ValueError(f"[E::main] Error: failed to save the marker gene sequences\n{e}")
        return target

    pathlib.Path(outdir).mkdir(parents=True, exist_ok=True)
    copy_function = shutil.move if genome_files_available else os.symlink
    for marker, (fna, faa) in marker_sequences.items():
        fna = store_seqs(fna, os.path.join(outdir, f"{marker}.fna"), copy_function)
        faa = store_seqs(faa if faa != "no_protein" else None, os.path.join(outdir, f"{marker}.faa"), copy_function) \
            if faa != "no_protein" else "no_protein"
        marker_sequences[marker] = [fna, faa]


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
        pathlib.Path(align_dir).mkdir(exist_ok=True, parents=True)
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
                     output=os.path.join(output, "genome_annotation"), long_out=long_out, verbose=verbose)
        else:
            file_ali = concat_alignments(input_files, align_dir, gene_order, ali_lengths, full_genomes=bool(genome_files))
            classify(concat_ali_stag_db, aligned_sequences=file_ali, output=os.path.join(output, "genome_annotation"),
                     long_out=long_out, verbose=verbose)
            os.remove(file_ali)
    finally:
        shutil.rmtree(temp_dir, ignore_errors=True)
