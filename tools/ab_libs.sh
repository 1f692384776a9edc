#!/bin/bash
# On the GPU box: the C2 and C3 bench lines of two builds of the library on the SAME box, A B A B, so that a difference
# between them is not the difference between two boxes.  Every GPU step runs under its own time limit and its exit
# status is checked on the spot: the first step that fails, faults or times out ends the script, nothing runs after it.
# usage: tools/ab_libs.sh <tag> <lib A> <lib B>
T=$1; A=$PWD/$2; B=$PWD/$3
O=gpurun_out/$T
mkdir -p $O || exit 1
Q="--no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 --a2m-reads 0"
run() {   # run <lib> <workload> <steps> <out>
  STAG_HIP_LIB=$1 timeout -k 10 100 python bench.py --workload $2 --steps $3 --warmup 5 $Q > $O/$4.json 2> $O/$4.err
}
run $A c2 30 c2_A1 || exit 1
run $B c2 30 c2_B1 || exit 1
run $A c3 15 c3_A1 || exit 1
run $B c3 15 c3_B1 || exit 1
run $A c2 30 c2_A2 || exit 1
run $B c2 30 c2_B2 || exit 1
run $A c3 15 c3_A2 || exit 1
run $B c3 15 c3_B2 || exit 1
python - <<P
import json, glob
for f in sorted(glob.glob("$O/c?_??.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"] / 1e6, 1), d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
P
