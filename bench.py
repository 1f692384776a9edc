#!/usr/bin/env python3
"""bench.py -- throughput of the post-alignment classification path on MI355X.

Workload (BASELINE.json configs[1], "C2"): a synthetic 16S amplicon stagDB (ncol 1500 -> L 7500,
7 ranks, ~20k nodes, 2 % non-zero fp64 coefficients per row) and 1M synthetic 250-bp V4 reads,
pre-aligned (A2M text: '-' outside a 253-column window, 0-3 lowercase inserts).  One step = one
pass of aligned text -> one-hot rows -> level-synchronous walk -> rank cut over the whole
batch, inputs and outputs resident in HBM.  N GPUs: one process per GPU, each classifies its
own 1M-read shard against its own replica of the tables (weak scaling, no collective on the
data path; torch.distributed is used only for the barrier and the max-over-ranks time).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how roofline and cpu_baseline
are defined).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


PMC_FILE = os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")


def _load_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE
    doubled as the gfx950 guide prescribes, + WRITE_SIZE; separate passes).  The file is stamped with
    the hash of the kernel sources it was measured on: a build from other sources gets no traffic
    figure (null) rather than a stale one.  Only valid for the default workload."""
    from stag_amd import _lib
    try:
        with open(PMC_FILE) as fh:
            doc = json.load(fh)
        if doc.get("kernel_source_sha1") != _lib.kernel_source_sha1():
            return {}
        return {k: v["hbm_bytes_per_launch"] for k, v in doc["kernels"].items()}
    except Exception:
        return {}


TRAFFIC = {}
WORKLOADS = {
    "c2": "C2: 16S amplicon stagDB, 1M synthetic 250-bp V4 reads pre-aligned, per GPU",
    "c3": "C3 shape: marker-gene stagDB (ncol 1050 = 350 aa x 3), synthetic full-length genes as codon alignments, per GPU",
    "c5": "C5 shape: ~50k-node taxonomy at the 16S width, synthetic 250-bp reads pre-aligned, per GPU",
    "c3p": "C3 through the protein entry point: marker-gene stagDB (350 aa), synthetic (protein alignment, gene) pairs, "
           "back-translated on the GPU (align.py:63-94), per GPU",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--db-scale", type=float, default=1.0, help="scales the node count of the synthetic tree")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1500, help="reads timed through the NumPy port")
    ap.add_argument("--check", type=int, default=2000, help="reads compared with the C oracle after the run")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 is the bench line (BASELINE.json configs[1]); c3 / c5 reproduce the other shapes")
    ap.add_argument("--host-steps", type=int, default=3, help="timed host-to-host steps (0: skip that leg)")
    ap.add_argument("--onehot-lines", type=int, default=65536, help="lines of the -s reader leg (0: skip it)")
    ap.add_argument("--a2m-reads", type=int, default=1_000_000, help="reads of the A2M-file leg (0: skip it)")
    ap.add_argument("--threads", type=int, default=0, help="scanner threads of the A2M-file leg (0: the job's cores, at most 16)")
    args = ap.parse_args()
    if args.reads == 1_000_000 and args.db_scale == 1.0 and args.workload == "c2":
        TRAFFIC.update(_load_traffic())

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # STAG_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("STAG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import bench_workload as bw
    from stag_amd import _lib
    from stag_amd.engine import Database

    if args.workload == "c2":
        wl = bw.config_16s(seed=2, scale=args.db_scale)
    else:
        wl = bw.config_c5() if args.workload == "c5" else bw.config_c3()
    t = wl.tables
    n = args.reads
    protein = args.workload == "c3p"
    text, offs, _ = bw.make_reads(wl, seed=1000 + rank, n=n, **({"window": wl.ncol, "max_ins": 0 if protein else 3}
                                                                if args.workload in ("c3", "c3p") else {}))
    if protein:
        # (protein alignment, gene) pairs whose back-translation is `text`: 4 % of the codons are alignment gaps
        # ('-' in the protein, nothing in the gene), every second gene ends in a stop codon (align.py:66-69)
        rng = np.random.default_rng(77 + rank)
        naa = wl.ncol // 3
        nt = text[:offs[-1]].reshape(n, naa, 3)
        gap = rng.random((n, naa)) < 0.04
        nt[gap] = ord("-")
        prot = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)[rng.integers(0, 20, size=(n, naa))]
        prot[gap] = ord("-")
        stop = np.arange(n) % 2 == 1
        ext = np.concatenate([nt.reshape(n, 3 * naa), np.tile(np.frombuffer(b"TAA", dtype=np.uint8), (n, 1))], axis=1)
        mask = np.concatenate([np.repeat(~gap, 3, axis=1), np.repeat(stop[:, None], 3, axis=1)], axis=1)
        gene_text = ext[mask]
        gene_offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(mask.sum(axis=1), out=gene_offs[1:])
        prot_text = prot.reshape(-1)
        prot_offs = np.arange(n + 1, dtype=np.int64) * naa
        del ext, mask
    text = np.concatenate([text, np.zeros(-len(text) % 16, dtype=np.uint8)])    # the _dev ABI wants a 16-byte multiple
    db = Database(t, device=local)
    D, K, W = db.max_depth, db.n_keys, db.words_per_row
    dev = torch.device("cuda", local)
    d_text = torch.from_numpy(text).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    out = {
        "path_node": torch.empty((n, D), dtype=torch.int32, device=dev),
        "path_prob": torch.empty((n, D), dtype=torch.float64, device=dev),
        "path_len": torch.empty(n, dtype=torch.int32, device=dev),
        "n_lead_single": torch.empty(n, dtype=torch.int32, device=dev),
        "level_prob": torch.empty((n, K), dtype=torch.float64, device=dev),
        "selected": torch.empty(n, dtype=torch.int32, device=dev),
        "n_aligned": torch.empty(n, dtype=torch.int32, device=dev),
        "status": torch.empty(n, dtype=torch.int32, device=dev),
    }
    res = _lib.Results()
    for f, _ in _lib.Results._fields_:
        setattr(res, f, out[f].data_ptr())
    db.reserve(n, len(text))
    stream = torch.cuda.current_stream(dev)
    lib = _lib.lib()

    if protein:
        d_prot = torch.from_numpy(np.concatenate([prot_text, np.zeros(16, np.uint8)])).to(dev)
        d_prot_offs = torch.from_numpy(prot_offs).to(dev)
        d_gene = torch.from_numpy(np.concatenate([gene_text, np.zeros(16, np.uint8)])).to(dev)
        d_gene_offs = torch.from_numpy(gene_offs).to(dev)

        def step():
            _lib.check(lib.stag_classify_protein_a2m_dev(db.handle, d_prot.data_ptr(), int(prot_offs[-1]), d_prot_offs.data_ptr(),
                                                         d_gene.data_ptr(), d_gene_offs.data_ptr(), n, 0.0, ctypes.byref(res),
                                                         stream.cuda_stream))
    else:
        def step():
            _lib.check(lib.stag_classify_a2m_dev(db.handle, d_text.data_ptr(), len(text), d_offs.data_ptr(), n, 0.0,
                                                 ctypes.byref(res), stream.cuda_stream))

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt_min = dt
    if world > 1:     # the job's time is the slowest rank's; the fastest rank's is kept beside it to show skew
        tt = torch.tensor([dt, -dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_min = float(tt[0].item()), -float(tt[1].item())

    # per-kernel device time (HIP events on the launch stream, recorded inside the library)
    prof_steps = max(1, min(args.steps, 5))
    db.profile_enable(True)
    for _ in range(prof_steps):
        step()
    prof = db.profile_read()
    db.profile_enable(False)

    # ---- SURVEY 8(d)'s own boundary, measured: aligned text in (page-locked) host memory -> result arrays
    # in host memory through the pipelined host entry point (H2D of the text, kernels, D2H of the results
    # overlapped on three streams inside the library).  Never `value`.
    host_leg = None
    if args.host_steps > 0 and world == 1:
        h_text = _lib.PinnedArray(len(text), np.uint8)
        h_offs = _lib.PinnedArray(len(offs), np.int64)
        h_text.a[:] = text
        h_offs.a[:] = offs
        h_out = {"path_node": _lib.PinnedArray((n, D), np.int32), "path_prob": _lib.PinnedArray((n, D), np.float64),
                 "path_len": _lib.PinnedArray(n, np.int32), "n_lead_single": _lib.PinnedArray(n, np.int32),
                 "level_prob": _lib.PinnedArray((n, K), np.float64), "selected": _lib.PinnedArray(n, np.int32),
                 "n_aligned": _lib.PinnedArray(n, np.int32), "status": _lib.PinnedArray(n, np.int32)}
        h_res = _lib.Results()
        for f, _ in _lib.Results._fields_:
            setattr(h_res, f, _lib.ptr(h_out[f].a))

        def host_step():
            _lib.check(lib.stag_classify_a2m(db.handle, _lib.ptr(h_text.a), _lib.ptr(h_offs.a), n, 0.0,
                                             ctypes.byref(h_res), None, None))

        host_step()
        h0 = time.perf_counter()
        for _ in range(args.host_steps):
            host_step()
        hdt = (time.perf_counter() - h0) / args.host_steps
        up = float(offs[-1]) + 8.0 * (n + 1)
        down = float(sum(v.nbytes for v in h_out.values()))
        same = all(np.array_equal(h_out[f].a, out[f].cpu().numpy()) for f in h_out)
        host_leg = {"value": round(n / hdt, 1), "unit": "seqs/s", "ms_per_step": round(hdt * 1e3, 3),
                    "steps": args.host_steps, "h2d_bytes": int(up), "d2h_bytes": int(down),
                    "link_GBs": {"h2d": round(up / hdt / 1e9, 2), "d2h": round(down / hdt / 1e9, 2)},
                    "equals_device_resident_results": bool(same),
                    "how": "stag_classify_a2m on page-locked host buffers (stag_host_alloc): sub-batches of 65,536 reads, "
                           "text up / kernels / results down on three streams"}
        del h_text, h_offs, h_out

    # ---- the whole host side, measured: an A2M file (what `hmmalign | esl-reformat a2m -` writes: records wrapped
    # at 60 columns) in the page cache -> scanned by a pool of threads into trimmed records in page-locked buffers
    # (stag_amd.stream) -> GPU (stag_classify_a2m_trimmed) -> the reference's two-column output lines in a file.
    # And, from the same scan, SURVEY 8(d)'s boundary once more for the trimmed form the scanner hands over.
    a2m_leg = None
    if args.a2m_reads > 0 and world == 1:
        import tempfile
        from stag_amd import parallel, stream
        m = min(args.a2m_reads, n)
        lens = (offs[1:m + 1] - offs[:m]).astype(np.int64)
        a2m = tempfile.NamedTemporaryFile(suffix=".a2m", delete=False)
        order = []                                      # read index of every record of the file, in file order
        with a2m:
            for ln in np.unique(lens).tolist():         # (reads of one length wrap the same way: whole blocks at a time)
                idx = np.flatnonzero(lens == ln)
                for c0 in range(0, len(idx), 50_000):
                    part = idx[c0:c0 + 50_000]
                    body = text[(offs[part][:, None] + np.arange(ln)[None, :]).ravel()].reshape(len(part), ln)
                    head = np.frombuffer("".join(">read%08d\n" % i for i in part.tolist()).encode(), np.uint8).reshape(len(part), 14)
                    cols = [head]
                    for c in range(0, ln, 60):
                        cols += [body[:, c:c + 60], np.full((len(part), 1), 10, np.uint8)]
                    a2m.write(np.concatenate(cols, axis=1).tobytes())
                    order.append(part)
        order = np.concatenate(order)
        size = os.path.getsize(a2m.name)
        threads = args.threads or stream.default_threads()
        pool = parallel.DevicePool(lambda d_: db, [local], own=False)
        tsv = a2m.name + ".tsv"
        try:
            stream.classify_a2m_file(pool, a2m.name, output=tsv, threads=threads)        # warm: page cache, buffers
            a0 = time.perf_counter()
            alen, n_lines = stream.classify_a2m_file(pool, a2m.name, output=tsv, threads=threads)
            adt = time.perf_counter() - a0
            # the scan alone (all threads, nothing else)
            s0 = time.perf_counter()
            batches = [(b.text.copy(), b.offs, b.lead_trail) for b in stream.a2m_file_batches(a2m.name, threads=threads)]
            sdt = time.perf_counter() - s0
        finally:
            pool.close()
        # the lines are what the per-row formatter gives for the device-resident results of the same reads
        names = np.array(t.names, dtype=object)
        sel = out["selected"].cpu().numpy()
        pn = out["path_node"].cpu().numpy()
        want = {}
        for i in order[:2000].tolist() + order[-2000:].tolist():
            keep_to = int(db.keys[sel[i]]) + 1 if sel[i] >= 0 else 0
            want["read%08d" % i] = ";".join(names[pn[i, :keep_to]])
        with open(tsv) as fh:
            got_lines = fh.read().splitlines()
        got = dict(l.split("\t") for l in got_lines[1:] if l.split("\t")[0] in want)
        os.remove(tsv)
        os.remove(a2m.name)
        # the trimmed form through the pipelined host entry point, from page-locked buffers
        tt_text = np.concatenate([b[0][:int(b[1][-1])] for b in batches])
        tt_offs = np.concatenate([[0]] + [b[1][1:] + o for b, o in zip(batches, np.cumsum([0] + [int(b[1][-1]) for b in batches[:-1]]))])
        tt_lt = np.concatenate([b[2] for b in batches])
        del batches
        p_text = _lib.PinnedArray(len(tt_text) + 16, np.uint8)
        p_text.a[:len(tt_text)] = tt_text
        p_offs = _lib.PinnedArray(len(tt_offs), np.int64)
        p_offs.a[:] = tt_offs
        p_lt = _lib.PinnedArray(tt_lt.shape, np.int32)
        p_lt.a[:] = tt_lt
        t_out = {"path_node": _lib.PinnedArray((m, D), np.int32), "path_prob": _lib.PinnedArray((m, D), np.float64),
                 "path_len": _lib.PinnedArray(m, np.int32), "n_lead_single": _lib.PinnedArray(m, np.int32),
                 "level_prob": _lib.PinnedArray((m, K), np.float64), "selected": _lib.PinnedArray(m, np.int32),
                 "n_aligned": _lib.PinnedArray(m, np.int32), "status": _lib.PinnedArray(m, np.int32)}
        t_res = _lib.Results()
        for f, _ in _lib.Results._fields_:
            setattr(t_res, f, _lib.ptr(t_out[f].a))

        def trimmed_step():
            _lib.check(lib.stag_classify_a2m_trimmed(db.handle, _lib.ptr(p_text.a), _lib.ptr(p_offs.a), _lib.ptr(p_lt.a), m, 0.0,
                                                     ctypes.byref(t_res), None, None))

        trimmed_step()
        h0 = time.perf_counter()
        for _ in range(max(args.host_steps, 1)):
            trimmed_step()
        tdt = (time.perf_counter() - h0) / max(args.host_steps, 1)
        same = all(np.array_equal(t_out[f].a, out[f][:m].cpu().numpy()[order]) for f in t_out)
        up = float(tt_offs[-1]) + 16.0 * m
        a2m_leg = {"value": round(m / adt, 1), "unit": "seqs/s", "reads": m, "threads": threads, "file_bytes": int(size),
                   "text_GBs": round(size / adt / 1e9, 2), "lines_written": int(n_lines),
                   "lines_equal_device_resident_results": bool(got == want and len(got) == len(want)),
                   "scan_only": {"value": round(m / sdt, 1), "text_GBs": round(size / sdt / 1e9, 2)},
                   "how": "stream.classify_a2m_file: A2M file (60 columns per line) in the page cache, chunks scanned by %d "
                          "threads (stag_fasta_scan_trim) into page-locked buffers, stag_classify_a2m_trimmed, two-column "
                          "lines put together in C (stag_rows_format_short), output file written" % threads,
                   "host_to_host_trimmed": {
                       "value": round(m / tdt, 1), "unit": "seqs/s", "ms_per_step": round(tdt * 1e3, 3),
                       "h2d_bytes": int(up), "bytes_per_read_up": round(up / m, 1),
                       "equals_device_resident_results": bool(same),
                       "how": "stag_classify_a2m_trimmed on the scanner's output in page-locked memory (records without "
                              "their flanking gap columns + two counts each) -> result arrays in page-locked memory"}}
        del p_text, p_offs, p_lt, t_out, tt_text

    # ---- the -s reader (classify.py:16-20) on the GPU, measured: a file of one-hot lines (what -S writes: 2 bytes
    # per feature) -> lines found on the host -> fields parsed by k_parse_onehot -> walk -> host results
    onehot_leg = None
    if args.onehot_lines > 0 and world == 1:
        import tempfile
        from stag_amd import ingest
        from stag_amd.classify import parse_onehot_line
        m = min(args.onehot_lines, n)
        sub_text = np.concatenate([text[:int(offs[m])], np.zeros(-int(offs[m]) % 16, dtype=np.uint8)])
        rows = db.classify_a2m_text(sub_text, offs[:m + 1], want_bits=True).bits
        with tempfile.NamedTemporaryFile(suffix=".tsv", delete=False) as fh:
            fh.write(ingest.format_onehot(["read%d" % i for i in range(m)], rows, wl.tables.n_features))
            tsv = fh.name
        size = os.path.getsize(tsv)

        from stag_amd import parallel as par
        opool = par.DevicePool(lambda d_: db, [local], own=False)

        def onehot_pass():     # as classify() drives it: one batch inside the library while the next is being read
            got, pending = [], []
            for ids, tx, bb, be, raw in ingest.onehot_text_batches(tsv, slow_line=parse_onehot_line, ring_slots=4):
                pending.append(opool.submit(lambda d_, a=(tx, bb, be, raw): d_.classify_onehot_text(a[0], a[1], a[2], parse_onehot_line, a[3])))
                while len(pending) >= 2:
                    got.append(pending.pop(0).result()[0])
            got += [f.result()[0] for f in pending]
            return got

        onehot_pass()
        o0 = time.perf_counter()
        got = onehot_pass()
        odt = time.perf_counter() - o0
        opool.close()
        os.remove(tsv)
        same = all(np.array_equal(np.concatenate([getattr(g, f) for g in got]), out[f][:m].cpu().numpy())
                   for f in ("path_node", "path_prob", "selected", "n_aligned", "status"))
        onehot_leg = {"value": round(m / odt, 1), "unit": "seqs/s", "lines": m, "file_bytes": int(size),
                      "text_GBs": round(size / odt / 1e9, 2), "equals_device_resident_results": bool(same),
                      "how": "ingest.onehot_text_batches (32 MB chunks of whole lines read into page-locked buffers and their lines found "
                             "by a pool of up to 8 threads) + Database.classify_onehot_text (k_parse_onehot + the walk); "
                             "file in the page cache"}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_ok = int((out["status"] == 0).sum().item())
    # ---- algorithmic bytes: B_seq = A + ncol + R
    A = (float(prot_offs[-1]) + float(gene_offs[-1])) / n if protein else float(offs[-1]) / n      # input bytes per read
    ncol = wl.ncol
    R = D * 12 + K * 8 + 12
    b_seq = A + ncol + R
    total_ms = sum(v[0] for v in prof.values())
    per_step = {k: v[0] / prof_steps for k, v in prof.items()}
    # `roofline` always describes k_walk_level (the class with the most device time flips between
    # two classes that are within 1 % of each other); k_encode_a2m and the whole path are carried beside it in
    # every run.  Algorithmic bytes per launch are each kernel's own minimum in SURVEY
    # 8d's terms -- encode reads the text (A) and writes the encoded form (ncol); the walk reads the encoded
    # form once (ncol) and writes the results (R), spread over its D launches -- and a kernel is never
    # credited with more bytes than the HBM counters saw it move (encode writes 944 B of packed bits, not
    # 1,500 one-byte codes).
    def kernel_line(cls, kname, algorithmic, launches_per_step):
        ms, launches = prof[cls]
        avg_ms = ms / max(launches, 1)
        traffic = TRAFFIC.get(kname)
        credited = min(algorithmic, traffic) if traffic else algorithmic
        achieved = credited / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        return {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                "launches_per_step": launches_per_step, "algorithmic_bytes_per_launch": int(algorithmic),
                "credited_bytes_per_launch": int(credited)}

    roofline = kernel_line("walk_level", "k_walk_level", n * (ncol + R) / D, D)
    roofline["encode"] = kernel_line("encode", "k_encode_a2m", n * (A + ncol), 1)
    roofline["path"] = {   # whole step priced as one pass: B_seq * n over the device time of ALL its kernels
        "bytes_per_seq": round(b_seq, 1),
        "device_ms_per_step": round(total_ms / prof_steps, 4),
        "achieved": round(b_seq * n / (total_ms / prof_steps * 1e-3) / 1e9, 2),
        "frac": round(b_seq * n / (total_ms / prof_steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
    }
    roofline["kernel_ms_per_step"] = {k: round(v, 4) for k, v in per_step.items()}

    # ---- parity spot check against the C oracle on the first reads of rank 0's shard
    parity = None
    if args.check > 0:
        from oracle import oracle_c
        m = min(args.check, n)
        odb = oracle_c.DenseDB.from_packed(t)
        X = np.zeros((m, t.n_features), dtype=np.uint8)
        ok_len = np.zeros(m, dtype=bool)
        for i in range(m):
            b, n_char, _ = oracle_c.encode_a2m(bytes(text[offs[i]:offs[i + 1]]).decode("latin-1"))
            ok_len[i] = 5 * n_char == t.n_features
            if ok_len[i]:
                X[i] = b
        o = odb.classify(X)
        g_node = out["path_node"][:m].cpu().numpy()
        g_prob = out["path_prob"][:m].cpu().numpy()
        g_lp = out["level_prob"][:m].cpu().numpy()
        same_path = (g_node == o["path_node"]).all(axis=1)
        rel = np.abs(g_prob - o["path_prob"]) / np.maximum(np.abs(o["path_prob"]), 1e-300)
        rel_lp = np.abs(g_lp - o["level_prob"]) / np.maximum(np.abs(o["level_prob"]), 1e-300)
        parity = {"checked": m, "lineage_equal": int(same_path.sum()),
                  "max_rel_err_node_prob": float(rel[same_path].max()) if same_path.any() else None,
                  "max_rel_err_level_prob": float(rel_lp[same_path].max()) if same_path.any() else None,
                  "n_aligned_equal": bool((out["n_aligned"][:m].cpu().numpy() == o["n_aligned"]).all())}
        del X

    # ---- CPU baseline: the NumPy port (reference cost structure, 1 core) on a bounded sample
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle_np
        taxonomy = bw.taxonomy_dict(t)
        tax_function = bw.tax_function_dict(t)
        classifiers = bw.LazyClassifiers(t)
        sample = min(args.cpu_sample, n)
        strings = [bytes(text[offs[i]:offs[i + 1]]).decode("latin-1") for i in range(sample)]
        for s in strings[:20]:      # materialise the hot rows outside the timed region
            oracle_np.classify_seq("w", oracle_np.convert_alignment(s)[0], taxonomy, tax_function, classifiers)
        c0 = time.perf_counter()
        for i, s in enumerate(strings):
            x, _ = oracle_np.convert_alignment(s)
            oracle_np.classify_seq(str(i), x, taxonomy, tax_function, classifiers)
        cdt = time.perf_counter() - c0
        cpu = {"value": round(sample / cdt, 2), "unit": "seqs/s", "cores": 1, "kind": "port",
               "sample": "%d reads of the same batch through oracle/oracle_np.py (per-sequence NumPy, the "
                         "reference's cost structure; `stag classify -t N` is single-core on this path); "
                         "host has %d cores" % (sample, os.cpu_count())}

    # ---- "fair CPU" (SURVEY 8d ii): the C restatement (oracle/stag_oracle.c, dense rows, scalar code)
    # on all the host cores the box gives this job, same reads; reported beside cpu_baseline
    cpu_c = None
    if not args.no_cpu_baseline and world == 1 and args.check > 0:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle_c
        threads = max(1, min(16, os.cpu_count() or 1))
        per = 10000
        strings = [bytes(text[offs[i]:offs[i + 1]]).decode("latin-1") for i in range(min(n, threads * per))]

        def work(k):
            part = strings[k * per:(k + 1) * per]
            Xp = np.stack([oracle_c.encode_a2m(s_)[0] for s_ in part]) if part else np.zeros((0, t.n_features), np.uint8)
            return len(odb.classify(Xp)["status"]) if len(part) else 0

        c0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            done = sum(ex.map(work, range(threads)))
        cdt = time.perf_counter() - c0
        cpu_c = {"value": round(done / cdt, 1), "unit": "seqs/s", "cores": threads, "kind": "port",
                 "sample": "%d reads through oracle/stag_oracle.c (dense fp64 rows), %d threads" % (done, threads)}
    if args.check > 0:
        del odb

    # ---- what a plain device-to-device copy reaches on this GPU (read + write), for the roofline's context
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize(dev)
    c0 = time.perf_counter()
    for _ in range(5):
        dst.copy_(src)
    torch.cuda.synchronize(dev)
    roofline["measured_copy_GBs"] = round(5 * 2 * src.numel() / (time.perf_counter() - c0) / 1e9, 1)
    del src, dst

    value = n * world * args.steps / dt
    print(json.dumps({
        "metric": "classified seqs/sec post-hmmalign", "value": round(value, 1), "unit": "seqs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_step_fastest_rank": round(dt_min / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload],
                   "reads_per_gpu": n, "ncol": ncol, "L": t.n_features, "depth": D, "n_nodes": t.n_nodes,
                   "nnz_coefficients": t.nnz, "table_bytes": db.table_bytes, "classified_ok": n_ok,
                   "sharding": "reads split across ranks, tables replicated, no collective"},
        "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_c, "host_to_host": host_leg, "a2m_file": a2m_leg, "onehot_text": onehot_leg,
        "parity": parity,
    }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
