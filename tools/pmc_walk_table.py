#!/usr/bin/env python3
"""Per-launch SQ counters of the walk from gpurun_out/pmcw/ (tools/pmc_walk.sh)."""
import collections
import csv
import glob
import sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmcw"
rows = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])):
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("stagk::", ""))
    rows.setdefault(k, {})[r["Counter_Name"]] = rows.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS",
         "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
print("%-5s %-14s " % ("id", "kernel") + " ".join("%12s" % n.replace("SQ_", "")[:12] for n in names))
for (i, k), c in rows.items():
    if "walk" in k or "encode" in k:
        print("%-5d %-14s " % (i, k[:14]) + " ".join("%12.4g" % c.get(n, float("nan")) for n in names))
