#!/usr/bin/env python3
"""Turns gpurun_out/final/ (written by tools/collect_profiles.sh on the GPU box) into the files
committed under profiles/: the bench line, the rocprofv3 kernel-stats summary and the HBM traffic
per launch (FETCH_SIZE doubled for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes,
plus WRITE_SIZE; both are reported by rocprofv3 in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
PMC_ONLY = "--pmc-only" in sys.argv      # on the GPU box, between the counter passes and the bench run of collect_profiles.sh
tag = args[0] if args else "r3"
SRC = os.path.join(ROOT, "gpurun_out", "final_" + tag)      # written by ONE run of tools/collect_profiles.sh <tag>
if os.path.exists(os.path.join(SRC, "run_id")):             # ... which names its own subdirectory there
    SRC = os.path.join(SRC, open(os.path.join(SRC, "run_id")).read().strip())
sys.path.insert(0, ROOT)
from stag_amd._lib import kernel_source_sha1  # noqa: E402


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "").replace("stagk::", "")
    return name.split("<")[0]


def newest(pattern):
    """The one file a run leaves (collect_profiles.sh starts from an empty directory; more than one file means that
    runs were mixed, and per-launch figures from mixed runs are wrong without looking wrong)."""
    found = glob.glob(pattern)
    if len(found) != 1:
        raise SystemExit("%d files match %s: expected the output of exactly one run" % (len(found), pattern))
    return found[0]


def counter(dirname, cname):
    f = newest(os.path.join(SRC, dirname, "*", "*_counter_collection.csv"))
    tot, disp = collections.Counter(), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cname:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


if not PMC_ONLY:
    shutil.copy(os.path.join(SRC, "bench.json"), os.path.join(DST, tag + "_bench_final.json"))
    stats = newest(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(DST, tag + "_kernel_stats.csv"))

fetch, nf = counter("pmc_fetch", "FETCH_SIZE")
write, nw = counter("pmc_write", "WRITE_SIZE")
# A counter pass is `bench.py --steps 1 --warmup 0`: the timed step and the event-profiled step = 2 passes over the
# batch, each with ONE encode launch and one walk launch per tree level (7 for the bench database).  Anything else
# means the totals and the dispatch counts are not of the same run, and dividing them would give a wrong number
# (round 2: 80 MB per walk launch instead of 272 MB).
for name, disp in (("FETCH_SIZE", nf), ("WRITE_SIZE", nw)):
    passes = disp.get("k_encode_a2m", 0)
    if passes != 2 or disp.get("k_walk_level", 0) != 7 * passes or disp.get("k_rankcut", 0) != passes:
        raise SystemExit("%s pass: %d encode, %d walk, %d rank-cut dispatches; expected 2, 14, 2"
                         % (name, passes, disp.get("k_walk_level", 0), disp.get("k_rankcut", 0)))
if nf != nw:
    raise SystemExit("the FETCH_SIZE and WRITE_SIZE passes saw different dispatches: %r vs %r" % (nf, nw))
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline --check 0` (two passes over the C2 batch: the timed step and the event-profiled "
               "step). Counter values are KiB as rocprofv3 reports them (x1024 -> bytes); FETCH_SIZE is doubled "
               "for gfx950 as MI355X_MICROARCH.md (HBM section) prescribes, which over-counts kernels whose "
               "reads are narrow. hbm_bytes_per_launch = (2*FETCH + WRITE) / dispatches.",
       "kernel_source_sha1": kernel_source_sha1(),   # bench.py reports traffic only for a build from these sources
       "kernels": {}}
for k in fetch:
    if not k.startswith("k_"):
        continue
    d = max(nf.get(k, 1), 1)
    f = fetch[k] * 1024 / d
    w = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    out["kernels"][k] = {"dispatches": d, "FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes": w,
                         "hbm_bytes_per_launch": 2 * f + w}
json.dump(out, open(os.path.join(DST, tag + "_pmc_traffic.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-16s %3d launches  %8.1f MB/launch" % (k, v["dispatches"], v["hbm_bytes_per_launch"] / 1e6))
