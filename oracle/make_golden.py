#!/opt/conda/bin/python3.9
"""Generate tests/golden/*.json by running the REFERENCE's own functions.

Build-container only (needs /root/reference).  Run with the interpreter that has a real
h5py so the reference imports unmodified:

    /opt/conda/bin/python3.9 oracle/make_golden.py

What is called from the reference (zellerlab/stag 0.8.2):
  stag.align.convert_alignment          (align.py:37-58)     -> encode.json
  stag.align.protein2gene_alignment     (align.py:63-94)     -> backtranslate.json
  stag.classify.classify_seq            (classify.py:94-124) -> classify_seq.json
  stag.classify.run_logistic_prediction (classify.py:22-27)  -> sklearn_parity.json
  stag.classify.classify                (classify.py:131-207, with load_db replaced by an
                                         in-memory DB; output formatting is the reference's)
                                                             -> tsv.json
  stag.classify_genome.concat_alignments (classify_genome.py:364-388) -> concat.json
  stag.databases.save_to_file           (databases.py:111-147) -> tiny.stagDB (+ tiny_db.json)
  stag.create_db.train_all_classifiers_nonmp (create_db.py:127-145) on the bundled test sequences,
  stag.alignment.load_alignment_from_file, stag.taxonomy3.Taxonomy        -> c1.stagDB, c1.json
  stag.databases.save_to_file + the tar layout of train_genome.py:46-60,
  stag.classify_genome.concat_alignments / merge_gene_predictions       -> genome3.stagDB, genome_db.json
  stag.helpers.read_fasta               (helpers.py:107-121) and
  stag.classify.alignment_reader        (classify.py:16-20)  -> ingest.json

Only inputs and the reference's outputs are stored; no reference source enters the repo.
Floats are stored through repr(), which round-trips float64 exactly.
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, "/root/reference")
import stag.align as ref_align            # noqa: E402
import stag.classify as ref_classify      # noqa: E402
import stag.classify_genome as ref_genome  # noqa: E402
import stag.databases as ref_db           # noqa: E402
import stag.helpers as ref_helpers        # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as fh:
        json.dump(obj, fh, indent=0, separators=(",", ":"))
    print("wrote", name)


def bits_str(a):
    return "".join("1" if b else "0" for b in a)


# ---------------------------------------------------------------- encode (a1)
def gen_encode():
    rng = np.random.default_rng(1)
    cases = ["ACGTU", "acgtu", "-", ".", "N", "ACGTUacgtu-.N*RYKMSWBDHVn", "-----", "A", "a",
             "", "ac.gt", "AC-GT..NNacgtACGU--", "*", "ACGT" * 40 + "acg" + "-" * 7,
             "U" * 64, "-" * 63 + "A", "ac" * 5 + "G", "AcCgGtTuU-", "XZ@[`{|}~ 09azAZ",
             # characters outside ASCII: str.islower() decides (align.py:47), everything else is a column
             "AC\u00e9GT\u00c9-\u00df\u4e2dN\u00aa\u01c5G", "\u00e9\u00e8"]
    alphabet = list("ACGTU-.N") + list("acgtn") + list("RYKM*")
    p = np.array([6, 6, 6, 6, 2, 8, 1, 1] + [0.4] * 5 + [0.2] * 5, dtype=float)
    p /= p.sum()
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 300, 1031):
        cases.append("".join(rng.choice(alphabet, size=n, p=p)))
    out = []
    for s in cases:
        try:
            with contextlib.redirect_stderr(io.StringIO()):
                bits, pct = ref_align.convert_alignment(s, 1, as_numpy=True)
            out.append({"a2m": s, "bits": bits_str(bits), "pct": repr(float(pct))})
        except ZeroDivisionError:
            out.append({"a2m": s, "error": "ZeroDivisionError"})
    dump("encode.json", out)


# ---------------------------------------------------------------- back-translation (a2)
def gen_backtranslate():
    cases = [
        ("MKV", "ATGAAAGTT"),
        ("MKV", "ATGAAAGTTTAA"),           # stop codon
        ("MKV", "ATGAAAGT"),               # wrong length -> None
        ("M-KV", "ATGAAAGTT"),
        ("-M-K-V-", "ATGAAAGTTTGA"),
        ("MkV", "ATGAAAGTT"),              # lowercase residue consumed, not emitted
        ("M\u00e9K\u00c9V\u4e2d", "ATGAAAGTTCCCGGGTTT"),   # outside ASCII: isupper / islower decide (align.py:82-86)
        ("MK", "AT\u00e9AAA"),             # the gene's characters are copied as they are
        ("mkv", "ATGAAAGTT"),
        ("--", ""),
        ("M.KV", "ATGAAAGTTCCC"),          # '.' counts as a residue for the length check, emits nothing
        ("MK*", "ATGAAATAA"),              # '*' is neither upper nor lower
        ("MKV", "atgaaagtt"),              # gene kept "as is" (lowercase nt are dropped later by a1)
        ("M-kKv-V", "ATGCCCAAAGGGGTTTAG"),
        ("", ""),
    ]
    out = []
    for prot, gene in cases:
        for check in (True, False):
            with contextlib.redirect_stderr(io.StringIO()):
                r = ref_align.protein2gene_alignment("id1", prot, gene, check_length=check)
            out.append({"protein": prot, "gene": gene, "check_length": check, "result": r})
    dump("backtranslate.json", out)


# ---------------------------------------------------------------- synthetic DBs for a5-a11
def rand_onehot(rng, ncol, kind):
    x = np.zeros(5 * ncol, dtype=bool)
    if kind == "onehot":
        sym = rng.integers(0, 5, size=ncol)
        x[5 * np.arange(ncol) + sym] = True
    elif kind == "zerofill":       # some columns carry no bit (classify_genome.py:375)
        sym = rng.integers(0, 5, size=ncol)
        keep = rng.random(ncol) < 0.6
        x[(5 * np.arange(ncol) + sym)[keep]] = True
    elif kind == "anybits":        # arbitrary -s input
        x = rng.random(5 * ncol) < 0.3
    elif kind == "allzero":
        pass
    return x


def make_tree(rng, fanouts, ncol, density=0.3, scale=2.0, single_kind="string"):
    """fanouts: list per level of candidate fan-outs to draw from."""
    L = 5 * ncol
    taxonomy = {"tree_root": []}
    classifiers = {}
    frontier = ["tree_root"]
    for lvl, choices in enumerate(fanouts):
        nxt = []
        for parent in frontier:
            k = int(rng.choice(choices))
            kids = ["L%d_%s_%d" % (lvl, parent[-6:].replace("tree_r", "r"), j) for j in range(k)]
            kids = ["%s|%d" % (kid, len(taxonomy) + j) for j, kid in enumerate(kids)]
            taxonomy[parent] = kids
            for kid in kids:
                taxonomy[kid] = []
                w = rng.normal(0, scale, size=1 + L) * (rng.random(1 + L) < density)
                w[0] = rng.normal(-1, 1)
                if k == 1 and single_kind == "string":
                    classifiers[kid] = "no_negative_examples"
                else:
                    classifiers[kid] = w
            nxt.extend(kids)
        frontier = nxt
    return taxonomy, classifiers


def db_to_json(taxonomy, tax_function, classifiers):
    return {
        "taxonomy": taxonomy,
        "tax_function": {k: [repr(float(v)) for v in row] for k, row in tax_function.items()},
        "classifiers": {k: (v if isinstance(v, str) else [repr(float(x)) for x in v])
                        for k, v in classifiers.items()},
    }


def run_case(name, taxonomy, tax_function, classifiers, seqs, note=""):
    rows = []
    for i, x in enumerate(seqs):
        try:
            with np.errstate(over="ignore"):
                r = ref_classify.classify_seq("seq%d" % i, x, taxonomy, tax_function, classifiers, 1, 1)
            rows.append({"row": r})
        except Exception as e:  # the exception class is part of the contract
            rows.append({"error": [c.__name__ for c in type(e).__mro__]})
    case = db_to_json(taxonomy, tax_function, classifiers)
    case.update({"name": name, "note": note, "L": int(len(seqs[0])),
                 "seqs": [bits_str(x) for x in seqs], "expected": rows})
    return case


def gen_classify_seq():
    cases = []
    # A: generic random tree, depth 3, fan-outs 1..5, one-hot inputs
    rng = np.random.default_rng(11)
    ncol = 24
    tax, cls = make_tree(rng, [[3, 4], [1, 2, 3, 5], [1, 2, 4]], ncol)
    tf = {str(l): rng.normal(0, 1, size=4) for l in (-1, 0, 1, 2)}
    seqs = [rand_onehot(rng, ncol, "onehot") for _ in range(40)]
    cases.append(run_case("A_random_depth3", tax, tf, cls, seqs))

    # B: zero-filled / arbitrary / all-zero inputs on another tree; leaves at different depths
    rng = np.random.default_rng(12)
    ncol = 31
    tax, cls = make_tree(rng, [[2, 3], [0, 2, 3], [0, 1, 2], [2]], ncol, density=0.15)
    D = 4
    tf = {str(l): rng.normal(0, 1.5, size=1 + D) for l in (-1, 0, 1, 2, 3)}
    seqs = ([rand_onehot(rng, ncol, "zerofill") for _ in range(15)] +
            [rand_onehot(rng, ncol, "anybits") for _ in range(15)] +
            [rand_onehot(rng, ncol, "allzero")])
    # paths shorter than D make coeff*perc fail to broadcast -> ValueError (classify.py:25)
    cases.append(run_case("B_ragged_depth_inputs", tax, tf, cls, seqs,
                          "paths of length != n_tf raise ValueError unless one length is 1"))

    # C: exact ties (duplicated rows), fp64 saturation to 1.0, single-child root
    rng = np.random.default_rng(13)
    ncol = 12
    L = 5 * ncol
    taxonomy = {"tree_root": ["only"], "only": ["a", "b", "c", "d"],
                "a": ["a1", "a2"], "b": ["b1", "b2", "b3"], "c": ["c1"], "d": [],
                "a1": [], "a2": [], "b1": [], "b2": [], "b3": [], "c1": []}
    w_dup = rng.normal(0, 1, size=1 + L)
    w_big = np.concatenate([[50.0], np.abs(rng.normal(0, 1, size=L))])      # sigmoid == 1.0
    w_big2 = np.concatenate([[80.0], np.abs(rng.normal(0, 1, size=L))])     # also exactly 1.0
    classifiers = {
        "only": "no_negative_examples",
        "a": w_dup, "b": w_dup.copy(), "c": rng.normal(0, 1, size=1 + L), "d": rng.normal(0, 1, size=1 + L),
        "a1": w_big, "a2": w_big2,
        "b1": rng.normal(0, 1, size=1 + L), "b2": w_dup.copy(), "b3": w_dup.copy(),
        "c1": "no_negative_examples",
    }
    classifiers["c"][0] += 3.0
    tf = {str(l): rng.normal(0, 1, size=4) for l in (-1, 0, 1, 2)}
    seqs = [rand_onehot(rng, ncol, "onehot") for _ in range(30)]
    cases.append(run_case("C_ties_saturation_single_root", taxonomy, tf, classifiers, seqs,
                          "a==b rows identical: first stored child wins; a1/a2 both saturate to 1.0"))

    # D: 12 levels -> keys '10','11' sort before '2' (classify.py:62); chain of single children inside
    rng = np.random.default_rng(14)
    ncol = 10
    tax, cls = make_tree(rng, [[2], [2], [1], [2], [1], [1], [2], [2], [1], [2], [2], [2]], ncol,
                         density=0.4, scale=1.0)
    D = 12
    tf = {str(l): rng.normal(0, 1, size=1 + D) for l in range(-1, D)}
    seqs = [rand_onehot(rng, ncol, "onehot") for _ in range(12)]
    cases.append(run_case("D_twelve_levels_string_sort", tax, tf, cls, seqs))

    # E: huge negative logits (exp overflow -> prob 0.0), rank-cut never above 0 -> selected -1 (int)
    rng = np.random.default_rng(15)
    ncol = 8
    L = 5 * ncol
    taxonomy = {"tree_root": ["x", "y"], "x": ["x1", "x2"], "y": ["y1", "y2"],
                "x1": [], "x2": [], "y1": [], "y2": []}
    classifiers = {k: np.concatenate([[-800.0], -np.abs(rng.normal(0, 1, size=L))])
                   for k in ("x", "y", "x1", "x2", "y1", "y2")}
    classifiers["y1"] = rng.normal(0, 1, size=1 + L)
    tf = {"-1": np.array([-900.0, 0.0, 0.0]), "0": np.array([-900.0, 1.0, 1.0]),
          "1": np.array([-1000.0, 1.0, 1.0])}
    seqs = [rand_onehot(rng, ncol, "onehot") for _ in range(6)]
    cases.append(run_case("E_overflow_selected_minus1", taxonomy, tf, classifiers, seqs,
                          "all probabilities 0.0: first child wins (0.0 > -1); selected_level stays int -1"))

    # F: error contracts: missing classifier (KeyError), string classifier evaluated, empty root
    rng = np.random.default_rng(16)
    ncol = 6
    L = 5 * ncol
    seqs = [rand_onehot(rng, ncol, "onehot") for _ in range(3)]
    tf = {"-1": rng.normal(size=3), "0": rng.normal(size=3), "1": rng.normal(size=3)}
    taxonomy = {"tree_root": ["p", "q"], "p": ["p1", "p2"], "q": [], "p1": [], "p2": []}
    good = {k: rng.normal(0, 1, size=1 + L) for k in ("p", "q", "p1", "p2")}
    good["p"][0] += 30
    missing = dict(good)
    del missing["p2"]
    cases.append(run_case("F1_missing_classifier", taxonomy, tf, missing, seqs))
    strrow = dict(good)
    strrow["p1"] = "no_negative_examples"
    cases.append(run_case("F2_string_classifier_evaluated", taxonomy, tf, strrow, seqs))
    cases.append(run_case("F3_root_without_children", {"tree_root": []}, tf, {}, seqs))
    cases.append(run_case("F4_root_missing", {"other": ["z"]}, tf, {}, seqs))
    # child that is not itself a taxonomy key: walk stops there
    tax5 = {"tree_root": ["p", "q"], "p": ["p1", "p2"], "q": []}
    cases.append(run_case("F5_child_not_a_key", tax5, {k: v[:3] for k, v in tf.items()}, good, seqs))
    # rank-cut with a single coefficient (1+1): broadcasts against any path length
    tf1 = {"0": np.array([0.3, -1.2]), "-1": np.array([-0.1, 0.7])}
    cases.append(run_case("F6_rankcut_len1_broadcast", taxonomy, tf1, good, seqs))
    dump("classify_seq.json", cases)


# ---------------------------------------------------------------- sklearn parity (north_star 1e-9)
def gen_sklearn():
    from sklearn.linear_model import LogisticRegression
    rng = np.random.default_rng(21)
    ncol = 40
    L = 5 * ncol
    n = 120
    proto = rng.integers(0, 5, size=(2, ncol))
    X = np.zeros((n, L), dtype=bool)
    y = []
    for i in range(n):
        c = i % 2
        sym = proto[c].copy()
        mut = rng.random(ncol) < 0.15
        sym[mut] = rng.integers(0, 5, size=mut.sum())
        X[i, 5 * np.arange(ncol) + sym] = True
        y.append("yes" if c else "no")
    # create_db.py:132 (penalty l1, liblinear), classes sorted -> column 1 is "yes" (create_db.py:124)
    clf = LogisticRegression(random_state=0, penalty="l1", solver="liblinear")
    clf.fit(X, np.array(y))
    row = np.append(clf.intercept_, clf.coef_)       # databases.py:140
    proba = clf.predict_proba(X)[:, 1]
    ref = [float(ref_classify.run_logistic_prediction(X[i], row)) for i in range(n)]
    dump("sklearn_parity.json", {
        "L": L, "row": [repr(float(v)) for v in row],
        "X": [bits_str(x) for x in X],
        "predict_proba": [repr(float(v)) for v in proba],
        "run_logistic_prediction": [repr(v) for v in ref],
        "sklearn_version": __import__("sklearn").__version__,
    })


# ---------------------------------------------------------------- classify() TSV text (a12)
def gen_tsv():
    rng = np.random.default_rng(31)
    ncol = 16
    tax, cls = make_tree(rng, [[2, 3], [1, 2, 3], [2]], ncol)
    tf = {str(l): rng.normal(0, 1, size=4) for l in (-1, 0, 1, 2)}
    seqs = [rand_onehot(rng, ncol, "onehot") for _ in range(8)]
    ali_lines = ["read_%d\t%s" % (i, "\t".join("1" if b else "0" for b in x)) for i, x in enumerate(seqs)]
    out = {"db": db_to_json(tax, tf, cls), "L": 5 * ncol, "ali_lines": ali_lines, "runs": {}}

    def fake_load_db(path, protein_fasta_input=None, aligned_sequences=None, dir_output=None):
        hmm = tempfile.NamedTemporaryFile(delete=False, mode="w")
        hmm.write("NA")
        hmm.close()
        return hmm.name, False, tax, tf, cls, "0.8.2"

    real = ref_classify.load_db
    ref_classify.load_db = fake_load_db
    try:
        with tempfile.TemporaryDirectory() as tmp:
            ali = os.path.join(tmp, "ali.tsv")
            with open(ali, "w") as fh:
                fh.write("\n".join(ali_lines) + "\n")
            for tag, kw in (("short", {}), ("long", {"long_out": True}),
                            ("internal", {"internal_call": True, "long_out": True})):
                res = os.path.join(tmp, "out_" + tag)
                ret = ref_classify.classify("db", aligned_sequences=ali, output=res, verbose=1, **kw)
                text = open(res).read() if os.path.exists(res) else None
                out["runs"][tag] = {"kwargs": kw, "return_len": ret[0], "return_rows": ret[1], "file": text}
            class _Keep(io.StringIO):     # classify() closes sys.stdout (`with outfile:`)
                def close(self):
                    self.kept = self.getvalue()
                    super().close()
            buf = _Keep()
            with contextlib.redirect_stdout(buf):
                ref_classify.classify("db", aligned_sequences=ali, verbose=1)
            out["runs"]["stdout_short"] = {"kwargs": {}, "file": buf.kept}
    finally:
        ref_classify.load_db = real
    # the -S line format (classify.py:160-162)
    x = seqs[0]
    buf = io.StringIO()
    print("read_0", *np.char.mod('%.0f', x), sep="\t", file=buf)
    out["save_ali_line"] = buf.getvalue()
    dump("tsv.json", out)


# ---------------------------------------------------------------- concat_alignments (a15)
def gen_concat():
    rng = np.random.default_rng(41)
    gene_order = ["COG0012", "COG0016", "COG0018", "MG_4"]
    ali_lengths = {"COG0012": "10", "COG0016": "15", "COG0018": "5", "MG_4": "5"}
    files = {
        "COG0012": ["dir/genomeA.fna_12##COG0012", "dir/genomeB.fna_7##COG0012", "plain.3##COG0012"],
        "COG0016": ["dir/genomeA.fna_40##COG0016", "solo##COG0016"],
        # COG0018 file does not exist
        # a marker-gene name with '_': concat_alignments (classify_genome.py:372) picks the separator from the
        # WHOLE id, merge_gene_predictions (:353) from the part before '##' -- they disagree on "plain.3"
        "MG_4": ["plain.3##MG_4", "dotted.name.9##MG_4", "dir/genomeA.fna_12##MG_4", "solo##MG_4"],
    }
    with tempfile.TemporaryDirectory() as tmp:
        stored = {}
        for mg, ids in files.items():
            lines = []
            for gid in ids:
                bits = (rng.random(int(ali_lengths[mg])) < 0.4).astype(int)
                lines.append(gid + "\t" + "\t".join(map(str, bits)))
            stored[mg] = lines
            with open(os.path.join(tmp, mg), "w") as fh:
                fh.write("\n".join(lines) + "\n")
        with contextlib.redirect_stdout(io.StringIO()):
            path = ref_genome.concat_alignments(None, tmp, gene_order, ali_lengths)
        text = open(path).read()
        os.remove(path)
    # merge_gene_predictions (classify_genome.py:343-362) on ids of the same kinds
    preds = {"dir/genomeA.fna_12##COG0012": "Bacteria;Firmicutes", "dir/genomeA.fna_40##COG0016": "Bacteria",
             "plain.3##COG0012": "", "plain.3##MG_4": "Archaea", "dotted.name.9##MG_4": "Bacteria;X",
             "solo##MG_4": "Bacteria"}
    with tempfile.TemporaryDirectory() as tmp:
        with contextlib.redirect_stdout(io.StringIO()):
            ref_genome.merge_gene_predictions(None, [], preds, 1, 1, tmp, False, False)
        merged = {name: open(os.path.join(tmp, "genes_predictions", name)).read()
                  for name in sorted(os.listdir(os.path.join(tmp, "genes_predictions")))}
    dump("concat.json", {"gene_order": gene_order, "ali_lengths": ali_lengths,
                         "ali_files": stored, "concatenated": text,
                         "predictions": preds, "merged_files": merged})


# ---------------------------------------------------------------- stagDB written by the reference writer
class _Model:   # what save_to_file reads from an sklearn model: intercept_ and coef_
    def __init__(self, row):
        self.intercept_ = np.array([row[0]])
        self.coef_ = np.array([row[1:]])


def gen_stagdb():
    from stag.taxonomy3 import Taxonomy
    rng = np.random.default_rng(51)
    ncol = 29                                    # LENG 29 of stag/test/gene.hmm
    L = 5 * ncol
    with tempfile.TemporaryDirectory() as tmp:
        taxfile = os.path.join(tmp, "tax.tsv")
        lines = ["g1\tBacteria;Firmicutes;Bacilli", "g2\tBacteria;Firmicutes;Clostridia",
                 "g3\tBacteria;Bacteroidetes;Bacteroidia", "g4\tBacteria;Proteobacteria;Alpha",
                 "g5\tBacteria;Proteobacteria;Gamma", "g6\tBacteria;Proteobacteria;Beta/Delta"]
        with open(taxfile, "w") as fh:
            fh.write("\n".join(lines) + "\n")
        full = Taxonomy(taxfile)
        full.load_from_file()
        nodes = [n for n, _ in full.get_all_nodes(get_root=False)]
        classifiers = {}
        for n in nodes:
            sibs = full.find_node(n) if hasattr(full, "find_node") else None
            row = rng.normal(0, 1.5, size=1 + L) * (rng.random(1 + L) < 0.2)
            row[0] = rng.normal()
            classifiers[n] = _Model(row)
        classifiers["Bacteroidia"] = "no_negative_examples"   # single child (databases.py:145)
        classifiers["Bacteria"] = "no_negative_examples"
        tf = {l: _Model(rng.normal(0, 1, size=4)) for l in (-1, 0, 1, 2)}
        path = os.path.join(OUT, "tiny.stagDB")
        hmm = "/root/reference/stag/test/gene.hmm"
        ref_db.save_to_file(classifiers, full, tf, False, path, hmm_file_path=hmm, protein_fasta_input=None)
        meta = {
            "taxonomy": {n: list(full[n].children.keys()) for n, _ in full.get_all_nodes(get_root=True)},
            "tax_function": {str(k): [repr(float(v)) for v in np.append(m.intercept_, m.coef_)]
                             for k, m in tf.items()},
            "classifiers": {k: (m if isinstance(m, str) else
                                [repr(float(v)) for v in np.append(m.intercept_, m.coef_)])
                            for k, m in classifiers.items()},
            "use_cmalign": False, "align_protein": False, "tool_version": ref_db.tool_version,
            "hmm_sha1": __import__("hashlib").sha1(open(hmm, "rb").read()).hexdigest(),
            "hmm_len": os.path.getsize(hmm),
        }
        dump("tiny_db.json", meta)
        print("wrote tiny.stagDB", os.path.getsize(path), "bytes")


# ---------------------------------------------------------------- C1 substitute: the bundled test sequences
def gen_c1():
    """BASELINE config 1 / SURVEY 8(c)6.  The reference's own known-answer test (stag_test_short.py:197-280)
    trains on stag/test/{sequences.fasta, sequences.taxonomy, gene.hmm}, classifies the same 50 sequences and
    wants every predicted lineage to equal the true one.  hmmalign is not installed here, so the alignment
    is a fixed one: each sequence left-justified in the HMM's 29 columns.  Everything after the alignment is
    the reference's code: convert_alignment, load_alignment_from_file, Taxonomy, ensure_geneset_consistency,
    train_all_classifiers_nonmp (create_db.py:127-145: LogisticRegression(l1, liblinear) per node), save_to_file,
    classify_seq.  learn_taxonomy_selection_function (create_db.py:278-408) does not run under this
    container's pandas 2.x, so the rank-cut rows are fixed ones that favour the deepest level;
    the fixture says so."""
    import logging
    import stag.create_db as ref_create
    from stag.alignment import load_alignment_from_file
    from stag.taxonomy3 import Taxonomy
    logging.disable(logging.CRITICAL)
    test_dir = "/root/reference/stag/test"
    ncol = 29
    with open(os.path.join(test_dir, "sequences.fasta"), "rb") as fh:
        seqs = list(ref_helpers.read_fasta(fh, head_start=1))
    truth = dict(line.rstrip("\n").split("\t") for line in open(os.path.join(test_dir, "sequences.taxonomy")))
    aligned = [(sid, s.upper() + "-" * (ncol - len(s))) for sid, s in seqs]
    assert all(len(a) == ncol for _, a in aligned)
    with tempfile.TemporaryDirectory() as tmp:
        ali_file = os.path.join(tmp, "ali.tsv")
        with open(ali_file, "w") as fh:
            for sid, a in aligned:
                x, _ = ref_align.convert_alignment(a, 1, as_numpy=True)
                fh.write(sid + "\t" + "\t".join("1" if b else "0" for b in x) + "\n")
        alignment = load_alignment_from_file(ali_file)          # drops duplicated rows, as training does
        full = Taxonomy(os.path.join(test_dir, "sequences.taxonomy"))
        full.load_from_file()
        full.ensure_geneset_consistency(list(alignment.index.values))
        with contextlib.redirect_stdout(io.StringIO()):
            classifiers = ref_create.train_all_classifiers_nonmp(alignment, full, "l1", "liblinear")
        tf_note = "fixed rows (the reference's rank-cut training does not run under pandas 2.x here)"
        tf = {-1: _Model(np.array([-4.0, 0.0, 0.0])), 0: _Model(np.array([-2.0, 1.0, 0.0])),
              1: _Model(np.array([-1.0, 1.5, 2.5]))}
        path = os.path.join(OUT, "c1.stagDB")
        ref_db.save_to_file(classifiers, full, tf, False, path,
                            hmm_file_path=os.path.join(test_dir, "gene.hmm"), protein_fasta_input=None)
        taxonomy = {n: list(full[n].children.keys()) for n, _ in full.get_all_nodes(get_root=True)}
        cls = {k: (m if isinstance(m, str) else np.append(m.intercept_, m.coef_)) for k, m in classifiers.items()}
        tfd = {str(k): np.append(m.intercept_, m.coef_) for k, m in tf.items()}
        rows = []
        for sid, a in aligned:
            x, _ = ref_align.convert_alignment(a, 1, as_numpy=True)
            rows.append([str(v) for v in ref_classify.classify_seq(sid, x, taxonomy, tfd, cls, 1, 1)])
    n_right = sum(r[1] == truth[r[0]] for r in rows)
    dump("c1.json", {"ids": [sid for sid, _ in aligned], "sequences": [s for _, s in seqs],
                     "aligned": [a for _, a in aligned], "true_lineage": [truth[sid] for sid, _ in aligned],
                     "reference_rows": rows, "rank_cut": tf_note, "n_nodes": len(taxonomy),
                     "n_trained": sum(not isinstance(m, str) for m in classifiers.values())})
    print("wrote c1.stagDB", os.path.getsize(path), "bytes;", n_right, "of", len(rows), "lineages equal the taxonomy file")


# ---------------------------------------------------------------- genome database (a14 layout)
def gen_genome_db():
    """A genome .stagDB in the layout of train_genome.py:46-60 -- a gztar of one HDF5 per marker gene (written
    by the reference's save_to_file), threshold_file.tsv (its order is the concatenation order),
    hmm_lengths_file.tsv and concatenated_genes_STAG_database.HDF5 -- plus what the reference's
    concat_alignments + classify_seq give for a set of marker-gene alignments (genome_db.json)."""
    import tarfile
    from stag.taxonomy3 import Taxonomy
    rng = np.random.default_rng(414)
    ncols = {"COG0012": 12, "COG0016": 9, "COG0018": 15}
    order = ["COG0016", "COG0012", "COG0018"]                  # threshold-file order, not alphabetical
    lineages = ["g1\tBacteria;Firmicutes;Bacilli", "g2\tBacteria;Firmicutes;Clostridia",
                "g3\tBacteria;Bacteroidetes;Bacteroidia", "g4\tBacteria;Proteobacteria;Alpha",
                "g5\tBacteria;Proteobacteria;Gamma", "g6\tArchaea;Euryarchaeota;Halobacteria",
                "g7\tArchaea;Euryarchaeota;Methanococci"]
    dbs = {}
    with tempfile.TemporaryDirectory() as tmp:
        taxfile = os.path.join(tmp, "tax.tsv")
        with open(taxfile, "w") as fh:
            fh.write("\n".join(lineages) + "\n")

        def write_db(name, L, seed_scale):
            full = Taxonomy(taxfile)
            full.load_from_file()
            classifiers = {}
            for n, sibs in full.get_all_nodes(get_root=False):
                row = rng.normal(0, seed_scale, size=1 + L) * (rng.random(1 + L) < 0.25)
                row[0] = rng.normal()
                classifiers[n] = _Model(row) if sibs else "no_negative_examples"
            tf = {l: _Model(rng.normal(0, 1, size=4)) for l in (-1, 0, 1, 2)}
            path = os.path.join(tmp, name)
            ref_db.save_to_file(classifiers, full, tf, False, path, hmm_file_path="/root/reference/stag/test/gene.hmm",
                                protein_fasta_input=None)
            taxonomy = {n: list(full[n].children.keys()) for n, _ in full.get_all_nodes(get_root=True)}
            cls = {k: (m if isinstance(m, str) else np.append(m.intercept_, m.coef_)) for k, m in classifiers.items()}
            tfd = {str(k): np.append(m.intercept_, m.coef_) for k, m in tf.items()}
            dbs[name] = (taxonomy, tfd, cls)
            return path

        gene_paths = [write_db(mg, 5 * ncols[mg], 1.5) for mg in order]
        concat_path = write_db("concat", 5 * sum(ncols.values()), 0.8)
        thr = os.path.join(tmp, "thr.tsv")
        with open(thr, "w") as fh:
            fh.write("".join("%s\t%d\n" % (mg, 20 + i) for i, mg in enumerate(order)))
        lens = os.path.join(tmp, "lens.tsv")
        with open(lens, "w") as fh:
            fh.write("".join("%s\t%d\n" % (mg, 5 * ncols[mg]) for mg in order))
        out = os.path.join(OUT, "genome3.stagDB")
        with tarfile.open(out, "w:gz") as tar:                 # train_genome.py:47-60
            for p in gene_paths:
                tar.add(p, os.path.basename(p))
            for source, target in zip((thr, lens, concat_path), ("threshold_file.tsv", "hmm_lengths_file.tsv",
                                                                 "concatenated_genes_STAG_database.HDF5")):
                tar.add(source, target)
        # marker genes of four genomes: gA has every gene (COG0016 twice: the later line wins, :376), gB lacks
        # COG0018, gC has only COG0018, "solo.7" has no '_' (split on '.')
        have = {"COG0016": ["gA_2", "gB_9", "gA_7"], "COG0012": ["gA_1", "gB_4", "solo.7"], "COG0018": ["gC_3", "gA_5"]}
        a2m, per_gene_rows, ali_dir = {}, {}, os.path.join(tmp, "MG_ali")
        os.makedirs(ali_dir)
        for mg in order:
            recs = []
            for sid in have[mg]:
                s = "".join(rng.choice(list("ACGT-"), size=ncols[mg], p=[0.22, 0.22, 0.22, 0.22, 0.12]))
                at = int(rng.integers(0, ncols[mg]))
                recs.append((sid + "##" + mg, s[:at] + "ac"[: int(rng.integers(0, 3))] + s[at:]))
            a2m[mg] = recs
            taxonomy, tfd, cls = dbs[mg]
            rows = []
            with open(os.path.join(ali_dir, mg), "w") as fh:
                for sid, s in recs:
                    x, _ = ref_align.convert_alignment(s, 1, as_numpy=True)
                    rows.append([str(v) for v in ref_classify.classify_seq(sid, x, taxonomy, tfd, cls, 1, 1)])
                    fh.write(sid + "\t" + "\t".join("1" if b else "0" for b in x) + "\n")
            per_gene_rows[mg] = rows
        with contextlib.redirect_stdout(io.StringIO()):
            cpath = ref_genome.concat_alignments(None, ali_dir, order, {mg: str(5 * ncols[mg]) for mg in order})
        taxonomy, tfd, cls = dbs["concat"]
        genome_rows = []
        for gid, x in ref_classify.alignment_reader(cpath):
            genome_rows.append([str(v) for v in ref_classify.classify_seq(gid, x, taxonomy, tfd, cls, 1, 1)])
        concat_text = open(cpath).read()
        os.remove(cpath)
        preds = {r[0]: r[1] for mg in order for r in per_gene_rows[mg]}
        with contextlib.redirect_stdout(io.StringIO()):
            ref_genome.merge_gene_predictions(None, order, preds, 1, 1, tmp, False, False)
        merged = {name: open(os.path.join(tmp, "genes_predictions", name)).read()
                  for name in sorted(os.listdir(os.path.join(tmp, "genes_predictions")))}
    dump("genome_db.json", {"gene_order": order, "ncols": ncols, "a2m": a2m, "gene_rows": per_gene_rows,
                            "concatenated": concat_text, "genome_rows": genome_rows, "merged_files": merged})
    print("wrote genome3.stagDB", os.path.getsize(out), "bytes")


def gen_ingest():
    """read_fasta over byte streams and alignment_reader over -s files, with the awkward cases:
    wrapped lines, trailing blanks, CRLF, headers without lines, text before the first header,
    no newline at the end, non-ASCII whitespace, fields that are not plain 0/1."""
    rng = np.random.default_rng(77)
    fasta_cases = {
        "plain": b">a\nACGT\n>b desc here\nAC\nGT\n",
        "wrapped_blank_tail": b">r1  \nAC-gt  \n..TT\t\n\n>r2\nA\n",
        "crlf": b">x\r\nAC\r\nGT\r\n>y\r\nTT\r\n",
        "no_final_newline": b">a\nAC\n>b\nGG",
        "header_without_lines": b">a\n>b\nAC\n>c\n>d\n",
        "text_before_header": b"ACGT\nTT\n>a\nGG\n",
        "empty_line_record": b">a\n\n>b\nAC\n",
        "only_header": b">a\n",
        "empty": b"",
        "inner_spaces": b">a b\tc\nAC GT\n  leading\n",
        "gt_inside": b">a\nAC>GT\n>b\n>\nAA\n",
        "nbsp_tail": ">a\nAC\u00a0\n>b\u00a0\nGT\n".encode("utf-8"),
        "unicode_id": ">s\u00e9q 1\nAC\n".encode("utf-8"),
        "ctrl_ws": b">a\nAC\x1c\x1f\n\x0bGT\x0c\n",
    }
    big = []
    for i in range(300):
        seq = "".join(rng.choice(list("ACGT-acgt."), size=int(rng.integers(0, 200))))
        lines = [seq[j:j + 60] for j in range(0, len(seq), 60)]
        big.append(">read%d some text\n%s\n" % (i, "\n".join(lines)) if lines else ">read%d\n" % i)
    fasta_cases["many_wrapped"] = "".join(big).encode("ascii")
    fasta = {}
    for name, data in fasta_cases.items():
        got = list(ref_helpers.read_fasta(io.BytesIO(data), head_start=1))
        fasta[name] = {"hex": data.hex(), "records": [[sid, seq] for sid, seq in got]}

    # the gene file of align.py:150 is read in text mode: universal newlines
    fasta_text = {}
    text_cases = dict(fasta_cases)
    text_cases["lone_cr"] = b">a\rAC\rGT\n>b\rTT\r"
    text_cases["cr_before_header"] = b">a\nAC\r>b\nGG\n"
    text_cases["cr_inside_header"] = b">a x\rAC\n>b\nGG\n"
    for name, data in text_cases.items():
        with tempfile.NamedTemporaryFile("wb", suffix=".fa", delete=False) as fh:
            fh.write(data)
        with open(fh.name) as tfh:
            got = list(ref_helpers.read_fasta(tfh, is_binary=False, head_start=1))
        os.remove(fh.name)
        fasta_text[name] = {"hex": data.hex(), "records": [[sid, seq] for sid, seq in got]}

    onehot_cases = {
        "plain": "a\t0\t1\t1\t0\t0\nb\t1\t0\t0\t0\t0\n",
        "crlf_cr": "a\t0\t1\r\nb\t1\t0\rc\t1\t1\n",
        "no_final_newline": "a\t0\t1\nb\t1\t1",
        "trailing_tabs": "a\t0\t1\t\t\nb\t1\t0 \n",
        "no_fields": "justid\n\n\t\n",
        "empty_id": "\t1\t0\n",
        "wide_ints": "a\t2\t-1\t0\t+1\t 1\n",
        "ragged": "a\t0\t1\t1\nb\t1\nc\t1\t0\t1\t1\t0\t1\n",
        "nbsp_tail": "a\t0\t1\u00a0\n",
        "underscore": "a\t1_0\t0\n",
    }
    x = rng.random((40, 35)) < 0.4
    onehot_cases["block_40x35"] = "".join("g%d\t%s\n" % (i, "\t".join("1" if v else "0" for v in row))
                                          for i, row in enumerate(x))
    onehot = {}
    for name, text in onehot_cases.items():
        with tempfile.NamedTemporaryFile("wb", suffix=".tsv", delete=False) as fh:
            fh.write(text.encode("utf-8"))
        try:
            rows = [[gid, [int(v) for v in arr]] for gid, arr in ref_classify.alignment_reader(fh.name)]
            err = None
        except Exception as e:      # int() rejects a field: the reference raises
            rows, err = None, type(e).__name__
        os.remove(fh.name)
        onehot[name] = {"hex": text.encode("utf-8").hex(), "rows": rows, "raises": err}
    dump("ingest.json", {"fasta": fasta, "fasta_text": fasta_text, "onehot": onehot})


if __name__ == "__main__":
    if len(sys.argv) > 1:          # only the named generators, e.g. `make_golden.py c1 genome_db`
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_c1()
    gen_genome_db()
    gen_ingest()
    gen_encode()
    gen_backtranslate()
    gen_classify_seq()
    gen_sklearn()
    gen_tsv()
    gen_concat()
    gen_stagdb()
