#!/bin/bash
# Runs on the GPU box: kernel trace (per-dispatch durations) of one bench step for a workload, to see each
# level's walk launch on its own.  usage: tools/per_level_trace.sh c2|c3|c5
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
WL=${1:-c2}
O=$R/gpurun_out/levels_$WL
rm -rf $O/trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 --a2m-reads 0 > $O/trace.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = open(sys.argv[1] + "/levels.txt", "w")
for r in rows[-40:]:
    name = r["Kernel_Name"].split("(")[0][-40:]
    out.write("%-42s %8.1f us  grid %s\n" % (name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
cat $O/levels.txt
