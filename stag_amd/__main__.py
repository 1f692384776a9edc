"""Command line: the ``classify`` and ``classify_genome`` commands of the reference's ``stag``
(stag/__main__.py:205-236, :393-422, :531-591) with the same flags.  The other commands of the
reference (train, align, create_db, ...) are not on the classification path and are not part
of this build."""
import argparse
import os
import pathlib
import shutil
import sys

from . import __version__ as tool_version
from .helpers import check_file_exists


def handle_error(error, help_f=None):
    sys.stderr.write("\n[E::main] Error: " + error + "\n\n")
    if help_f:
        help_f()
    sys.exit(1)


def print_menu_classify():
    sys.stderr.write("\nUsage: stag classify -d <database> [-i <fasta_seqs>|-s <aligned_file>] [options]\n\n"
                     "  -d  FILE  database created with create_db or train\n"
                     "  -i  FILE  sequences to taxonomically annotate (fasta format)\n"
                     "  -s  FILE  aligned sequences, can be provided instead of -i\n"
                     "  --a2m FILE  (this build) the aligner's A2M output kept on disk, instead of -i: scanned by all cores\n"
                     "  -p  FILE  protein sequences, corresponding to -i\n"
                     "  -S  FILE  save intermediate alignment file\n"
                     "  -o  FILE  output file name [stdout]\n"
                     "  -l        long output (all predictions and probabilities)\n"
                     "  -m  INT   threshold for the number of features per sequence (percentage) [0]\n"
                     "  -t  INT   number of threads (only cmalign uses them) [1]\n"
                     "  -v  INT   verbose level: 1=error, 2=warning, 3=message, 4+=debugging [3]\n\n")


def print_menu_classify_genome():
    sys.stderr.write("\nUsage: stag classify_genome -d <database> -o <output_dir> -G <markers.json> [options]\n\n"
                     "  -d  FILE  genome database created with train_genome\n"
                     "  -G  FILE  JSON {marker gene: [genes.fna, proteins.faa]} of identified marker genes\n"
                     "  -i/-D     genome fasta file(s): gene calling is the reference's CPU step, not part of this build\n"
                     "  -o  DIR   output directory\n"
                     "  -l        long output\n  -f        rewrite the output directory\n"
                     "  -v  INT   verbose level [3]\n  -t  INT   threads [1]\n\n")


def main(argv=None):
    from . import parallel
    created = parallel.init_from_env()      # torchrun: one process per GPU, rank 0 writes the output
    try:
        return _main(argv)
    finally:
        if created:
            parallel.shutdown()


def _main(argv=None):
    parser = argparse.ArgumentParser(prog="stag", add_help=False)
    parser.add_argument("command", choices=["classify", "classify_genome"])
    # flag -> (attribute the reference's code reads, type or None for a switch): stag/__main__.py:205-236
    flags = {
        "-d": ("database", str), "-i": ("fasta_input", str), "-p": ("protein_fasta_input", str),
        "-s": ("aligned_sequences", str), "-S": ("intermediate_al", str), "-o": ("output", str),
        "-D": ("dir_input", str), "-G": ("marker_genes", str), "-m": ("min_perc_state", int),
        "-l": ("long_out", None), "-f": ("force_rewrite", None), "-r": ("keep_all_genes", None),
    }
    for flag, (dest, kind) in flags.items():
        if kind is None:
            parser.add_argument(flag, dest=dest, action="store_true")
        else:
            parser.add_argument(flag, dest=dest, type=kind, default=None)
    parser.add_argument("--a2m", dest="aligned_a2m", type=str, default=None)      # this build's own flag
    parser.add_argument("-t", dest="threads", type=int, default=1)
    parser.add_argument("-v", dest="verbose", type=int, default=3, choices=[1, 2, 3, 4])
    parser.add_argument("--version", action="version",
                        version="%(prog)s {0} on python {1}".format(tool_version, sys.version.split()[0]))
    args = parser.parse_args(argv)

    if args.min_perc_state is None:
        args.min_perc_state = 0
    if args.threads < 1:
        handle_error("number of threads (-t) is less than 1", None)
    if args.min_perc_state < 0 or args.min_perc_state > 100:
        handle_error("-m should be between 0 and 100. It represents the percentage\n"
                     "of internal states covered by the sequence (i.e. the number of features).", None)

    error = ""
    if args.command == 'classify':
        if not args.fasta_input and not args.aligned_sequences and not args.aligned_a2m:
            error = "missing <fasta_seqs> (-i) or <aligned_seq> (-s)"
        elif not args.database:
            error = "missing <database> (-d)"
        if error:
            handle_error(error, print_menu_classify)
        if args.fasta_input:
            check_file_exists(args.fasta_input, isfasta=True)
        check_file_exists(args.database, isfasta=False)
        if args.protein_fasta_input:
            check_file_exists(args.protein_fasta_input, isfasta=True)
        if args.intermediate_al and os.path.isfile(args.intermediate_al):
            os.remove(args.intermediate_al)
        from . import classify
        if args.aligned_a2m:
            check_file_exists(args.aligned_a2m, isfasta=True)
            classify.classify_a2m(args.database, args.aligned_a2m, output=args.output, long_out=args.long_out,
                                  verbose=args.verbose, min_perc_state=args.min_perc_state)
            return None
        classify.classify(args.database, fasta_input=args.fasta_input, protein_fasta_input=args.protein_fasta_input,
                          verbose=args.verbose, threads=args.threads, output=args.output, long_out=args.long_out,
                          current_tool_version=tool_version, aligned_sequences=args.aligned_sequences,
                          save_ali_to_file=args.intermediate_al, min_perc_state=args.min_perc_state)

    if args.command == 'classify_genome':
        if not args.database:
            error = "missing <database> (-d)"
        elif not any((args.fasta_input, args.dir_input, args.marker_genes)):
            error = "you need to provide at least -i, -D, or -G."
        elif sum(map(bool, (args.fasta_input, args.dir_input, args.marker_genes))) != 1:
            error = "options -i, -D, and -G are mutually exclusive"
        elif args.dir_input and not os.path.isdir(args.dir_input):
            error = "-D is not a directory."
        elif args.marker_genes and not os.path.isfile(args.marker_genes):
            error = "-G is not a valid file."
        elif not args.output:
            error = "missing output directory (-o)"
        if error:
            handle_error(error, print_menu_classify_genome)
        marker_genes, list_files = list(), list()
        if args.fasta_input:
            check_file_exists(args.fasta_input, isfasta=True)
            list_files.append(args.fasta_input)
        elif args.marker_genes:
            marker_genes = [args.marker_genes]
        else:
            for f in os.listdir(args.dir_input):
                f = os.path.join(args.dir_input, f)
                try:
                    if os.path.isfile(f) and open(f).read(1)[0] == ">":
                        list_files.append(f)
                except Exception:
                    if args.verbose > 1:
                        sys.stderr.write("[W::main] Warning: Cannot open file: {}\n".format(f))
            if not list_files:
                handle_error("no fasta files found in the directory.", None)
        if os.path.isdir(args.output):
            if args.force_rewrite:
                shutil.rmtree(args.output)
            else:
                handle_error("output directory (-o {}) exists already.".format(args.output), None)
        try:
            pathlib.Path(args.output).mkdir(exist_ok=True, parents=True)
        except Exception:
            handle_error("creating the output directory (-o {}).".format(args.output), None)
        from . import classify_genome
        classify_genome.classify_genome(args.database, genome_files=list_files, marker_genes=marker_genes,
                                        verbose=args.verbose, threads=args.threads, output=args.output,
                                        long_out=args.long_out, keep_all_genes=args.keep_all_genes)
    return None


if __name__ == '__main__':
    main()
