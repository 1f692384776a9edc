# LDS pipe occupancy of the walk (one counter pass): instructions, active cycles, bank conflicts per dispatch.
set -e
R=$PWD; O=$R/gpurun_out/pmclds; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 > $O/log.txt 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-30:])
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = open(sys.argv[1] + "/lds.txt", "w")
for (d, name), c in by.items():
    if "walk" in name or "encode" in name:
        out.write("%5s %-30s %s\n" % (d, name, " ".join("%s=%.3g" % kv for kv in sorted(c.items()))))
PY
cat $O/lds.txt
