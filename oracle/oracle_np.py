"""CPU oracle (NumPy / pure Python) for STAG's post-alignment classification path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stag_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker.  The product path is the HIP library and fails loudly
when that library is missing.

Each function restates one reference function (zellerlab/stag 0.8.2, paths relative to
``/root/reference``) and keeps the reference's cost structure on purpose (per-sequence
Python loops, one NumPy reduction per tree node), so that timing it gives a fair picture
of ``stag classify`` on one core.

Pinning: ``oracle/make_golden.py`` imports the reference's own ``stag.align`` and
``stag.classify`` in the build container and stores their outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this module against those vectors (bit-exact for
encodings and text fields, 1e-12 relative for probabilities).
"""
import numpy as np

TREE_ROOT = "tree_root"          # stag/taxonomy3.py:18
NO_NEG = "no_negative_examples"  # stag/databases.py:104

# stag/align.py:28-35 -- position of the set bit inside a 5-bit column block.
_BIT_OF = {"A": 4, "C": 3, "G": 2, "T": 1, "U": 1}


def convert_alignment(alignment):
    """stag/align.py:37-58.  A2M string -> (bool[5*n_char], percent of A/C/G/T/U columns).

    Lowercase characters (HMM insert states) are dropped; every other character is one
    column; columns that are not A/C/G/T/U set bit 0.  ``n_char == 0`` raises
    ZeroDivisionError exactly like the reference.
    """
    kept = [ch for ch in alignment if not ch.islower()]
    n_char = len(kept)
    bits = np.zeros(5 * n_char, dtype=bool)
    n_aligned = 0
    for col, ch in enumerate(kept):
        b = _BIT_OF.get(ch)
        if b is None:
            bits[5 * col] = True
        else:
            bits[5 * col + b] = True
            n_aligned += 1
    return bits, n_aligned / n_char * 100


def protein2gene_alignment(gene_id, protein_alignment, gene_sequence, check_length=True):
    """stag/align.py:63-94.  Protein A2M row + gene nucleotides -> codon alignment row.

    Returns None when the gene length is neither 3*residues nor 3*residues+3 (:66-73).
    '-' -> '---'; uppercase residue -> next 3 nt (a short tail yields fewer than 3);
    lowercase residue -> consume 3 nt, emit nothing; other characters emit nothing.
    """
    n_res = sum(1 for ch in protein_alignment if ch != "-")
    if check_length:
        if len(gene_sequence) != 3 * n_res and len(gene_sequence) - 3 != 3 * n_res:
            return None
    out, pos = [], 0
    for ch in protein_alignment:
        if ch == "-":
            out.append("---")
        elif ch.isupper():
            out.append(gene_sequence[pos:pos + 3])
            pos += 3
        elif ch.islower():
            pos += 3
    return "".join(out)


def parse_onehot_line(line):
    """stag/classify.py:16-20 (one line of the -s file): id, bool[L]; any non-zero int is True."""
    gene_id, *fields = line.rstrip().split("\t")
    return gene_id, np.array([int(f) for f in fields], dtype=bool)


def run_logistic_prediction(seq, coeff_raw):
    """stag/classify.py:22-27.  sigmoid(coeff_raw[0] + sum(coeff_raw[1:] * seq)) in fp64.

    The reference unpacks ``coeff_raw`` into a Python list and lets NumPy broadcast
    ``list * ndarray``; shapes that do not broadcast raise ValueError, and a
    ``"no_negative_examples"`` entry raises a TypeError subclass.  Both are kept.
    """
    if isinstance(coeff_raw, str):
        raise TypeError("classifier is the string %r" % coeff_raw)
    # The reference splits the row into a Python float and a Python *list* of L floats on every
    # call and lets NumPy turn that list back into an array.  That unpack is where its time
    # goes, so the port keeps it: timing this function is timing the reference.
    intercept, *coeff = coeff_raw
    z = (coeff * seq).sum() + intercept
    with np.errstate(over="ignore"):
        return 1 / (1 + np.exp(-z))


def find_best_score(test_seq, siblings, classifiers):
    """stag/classify.py:31-43.  Strict '>' over probabilities, stored child order."""
    best_score, best_taxa = -1, ""
    if len(siblings) == 1:
        best_score, best_taxa = 2, siblings[0]
    elif len(siblings) > 1:
        for sib in siblings:
            p = run_logistic_prediction(test_seq, classifiers[sib])
            if p > best_score:
                best_score, best_taxa = p, sib
    return best_taxa, best_score


def predict_path(test_seq, taxonomy, classifiers):
    """stag/classify.py:45-52, written as a loop instead of recursion."""
    tax, perc = [], []
    node = TREE_ROOT
    while taxonomy.get(node) is not None:
        node, p = find_best_score(test_seq, taxonomy[node], classifiers)
        if node:
            tax.append(node)
            perc.append(p)
    return tax, perc


def find_correct_level(perc, tax_function):
    """stag/classify.py:58-72.  Keys visited in *string* sort order; strict '>' from 0."""
    best_p, selected, texts = 0, -1, []
    x = np.asarray(perc)
    for level in sorted(tax_function):
        p = run_logistic_prediction(x, tax_function[level])
        if p > best_p:
            best_p, selected = p, level
        texts.append(level + ":" + str(p))
    return selected, texts


def find_n_aligned_characters(test_seq):
    """stag/classify.py:79-87.  Columns whose bit 0 is clear."""
    first = test_seq[0::5]
    return np.size(first) - np.sum(first)


def classify_seq(gene_id, test_seq, taxonomy, tax_function, classifiers):
    """stag/classify.py:94-124.  One sequence -> the 7 output fields."""
    n_aligned = find_n_aligned_characters(test_seq)
    tax, perc = predict_path(test_seq, taxonomy, classifiers)
    if perc[0] == 2:            # :106-107, Python int 1 on purpose (prints as "1")
        perc[0] = 1
    for i in range(len(perc)):  # :108-110
        if perc[i] == 2:
            perc[i] = perc[i - 1]
    selected, per_level = find_correct_level(perc, tax_function)
    return [gene_id,
            ";".join(tax[0:int(selected) + 1]),
            "/".join(tax),
            selected,
            "/".join(str(p) for p in perc),
            "/".join(per_level),
            str(n_aligned)]


OUT_HEADER = ["sequence", "taxonomy", "full_taxonomy", "selected_level",
              "prob_from_classifiers", "prob_per_level", "n_aligned_characters"]


def format_tsv(rows, long_out):
    """stag/classify.py:179-190: header + rows, first two columns unless long_out."""
    keep = len(OUT_HEADER) if long_out else 2
    lines = ["\t".join(OUT_HEADER[:keep])]
    for r in rows:
        lines.append("\t".join(str(f) for f in r[:keep]))
    return "\n".join(lines) + "\n"


def onehot_line(gene_id, bits):
    """-S output line, stag/classify.py:160-162 ('%.0f' of a bool array gives 0/1)."""
    return gene_id + "\t" + "\t".join("1" if b else "0" for b in bits)


def concat_alignments(ali_lines_by_gene, gene_order, ali_lengths):
    """stag/classify_genome.py:364-388 on in-memory lines: {gene: [line, ...]} -> [line, ...].

    The genome key is the id before '##' with its last '_' (or '.') token dropped; a
    genome's missing genes are ``ali_lengths[gene]`` zeros.
    """
    all_genes = {}
    for pos, mg in enumerate(gene_order):
        for line in ali_lines_by_gene.get(mg, ()):
            genome, *ali = line.strip().split("\t")
            sep = "_" if "_" in genome else "."
            toks = genome.split("##")[0].split(sep)
            genome = sep.join(toks[:-1] if len(toks) > 1 else toks)
            if genome not in all_genes:
                all_genes[genome] = ["\t".join(["0"] * int(ali_lengths[g])) for g in gene_order]
            all_genes[genome][pos] = "\t".join(ali)
    return [genome + "\t" + "\t".join(parts) for genome, parts in all_genes.items()]
