"""BASELINE.json configs[3] measured: `classify_genome -G` over synthetic genomes with a genome database of 30
marker genes (C3 width each) + the concatenated database (L = 157,500), both written here by the test-side .stagDB
writer (tests/stagdb_writer.py) in train_genome's tar layout.  The aligner is a stand-in that replays prepared A2M
text (hmmalign is the reference's external step and is not installed here).

    python tools/bench_genome.py [genomes=10000] [genes=30]      (on a GPU box, from the repository root)

Prints one JSON line: genomes/s and marker-gene sequences/s for the fused route (concatenated rows assembled on the
GPU from the per-gene rows) and for the reference's text route (MG_ali -> concat_alignments -> -s reader)."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("STAG_AMD_CACHE", "off")

import bench_workload as bw  # noqa: E402
import stagdb_writer  # noqa: E402
from stag_amd import classify_genome as scg  # noqa: E402
from test_gpu_shapes import dicts_of  # noqa: E402


def main():
    n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    n_genes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    gcol = 1050
    tmp = tempfile.mkdtemp(prefix="stag_genome_bench_")
    try:
        rng = np.random.default_rng(4)
        genes = ["COG%04d" % (12 + 7 * i) for i in range(n_genes)]
        t0 = time.perf_counter()
        gene_db = {}
        for i, g in enumerate(genes):
            wl = bw.make_db(400 + i, gcol, [3, 7, 16, 30, 60], 0.02, name=g)
            wl.tables.names[1:] = ["%s_n%d" % (g, j) for j in range(1, wl.tables.n_nodes)]
            gene_db[g] = os.path.join(tmp, g + ".stagDB")
            stagdb_writer.write_stagdb(gene_db[g], *dicts_of(wl.tables))
        cwl = bw.make_db(499, n_genes * gcol, [3, 6, 12, 24, 40, 80], 0.004, name="concat")
        concat = os.path.join(tmp, "concat.stagDB")
        stagdb_writer.write_stagdb(concat, *dicts_of(cwl.tables))
        genome_db = os.path.join(tmp, "gtdb_like.stagDB")
        stagdb_writer.write_genome_db(genome_db, gene_db, concat, {g: "60.0" for g in genes}, {g: str(5 * gcol) for g in genes})
        t_db = time.perf_counter() - t0
        present = rng.random((n_genomes, n_genes)) < 0.9
        present[:, 0] |= ~present.any(axis=1)
        mgs, n_seq = {}, int(present.sum())
        fna = {g: open(os.path.join(tmp, g + "_genes.fna"), "w") for g in genes}
        a2m = {g: open(os.path.join(tmp, g + ".fna.a2m"), "wb") for g in genes}
        for c0 in range(0, n_genomes, 2000):          # one pass over the genomes: a genome's genes are cut from one read
            m = min(2000, n_genomes - c0)
            text, offs, _ = bw.make_reads(cwl, seed=1000 + c0, n=m, window=n_genes * gcol, max_ins=0)
            whole = text[:offs[-1]].reshape(m, n_genes * gcol)
            for i, g in enumerate(genes):
                rows = np.flatnonzero(present[c0:c0 + m, i])
                ids = ["GENOME%06d_%d##%s" % (c0 + j, i + 1, g) for j in rows]
                fna[g].write("".join(">%s\nACGT\n" % s for s in ids))
                block = whole[rows, i * gcol:(i + 1) * gcol]
                a2m[g].write(b"".join(b">%s\n%s\n" % (s.encode(), block[r].tobytes()) for r, s in enumerate(ids)))
        for g in genes:
            fna[g].close()
            a2m[g].close()
            mgs[g] = [os.path.join(tmp, g + "_genes.fna"), "no_protein"]
        bindir = os.path.join(tmp, "bin")
        os.makedirs(bindir)
        with open(os.path.join(bindir, "hmmalign"), "w") as fh:
            fh.write("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat %s/$(basename $2).a2m\n" % tmp)
        with open(os.path.join(bindir, "esl-reformat"), "w") as fh:
            fh.write("#!/bin/sh\n[ $# -eq 0 ] && exit 0\ncat\n")
        for tool in ("hmmalign", "esl-reformat"):
            os.chmod(os.path.join(bindir, tool), 0o755)
        os.environ["PATH"] = bindir + os.pathsep + os.environ["PATH"]
        spec = os.path.join(tmp, "mgs.json")
        json.dump(mgs, open(spec, "w"))

        def run(tag, fused):
            out = os.path.join(tmp, "out_" + tag)
            real = scg.concat_on_device
            if not fused:
                scg.concat_on_device = lambda *a: None
            try:
                t1 = time.perf_counter()
                scg.classify_genome(genome_db, marker_genes=[spec], output=out, verbose=1, long_out=False)
                dt = time.perf_counter() - t1
            finally:
                scg.concat_on_device = real
            lines = open(os.path.join(out, "genome_annotation")).read()
            shutil.rmtree(out)
            return dt, lines

        run("warm", True)
        dt_fused, a = run("fused", True)
        dt_text, b = run("text", False)
        print(json.dumps({"workload": "classify_genome -G: %d genomes x %d marker genes of %d columns (p = 0.9 present), "
                                      "concatenated database L = %d" % (n_genomes, n_genes, gcol, 5 * n_genes * gcol),
                          "marker_gene_sequences": int(n_seq), "database_write_s": round(t_db, 1),
                          "fused": {"seconds": round(dt_fused, 2), "genomes_per_s": round(n_genomes / dt_fused, 1),
                                    "sequences_per_s": round(n_seq / dt_fused, 1)},
                          "text_route": {"seconds": round(dt_text, 2), "genomes_per_s": round(n_genomes / dt_text, 1)},
                          "same_genome_annotation": a == b, "genomes_annotated": a.count("\n") - 1}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
