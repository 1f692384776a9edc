#!/bin/bash
# On the GPU box: parity tests, a fuzz sample, then the C2 and C3 bench lines of the build in the tree -> gpurun_out/<tag>/
# usage: tools/ab_bench.sh <tag>      (every GPU step under its own time limit and with its exit status checked on the
# spot: the first step that fails, faults or times out ends the script, nothing runs after it)
set -e
T=${1:-ab}
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_shapes.py -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -5 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 200 python -u tools/fuzz_gpu.py 400 700000 > $O/fuzz.log 2>&1 || { tail -3 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
Q="--no-cpu-baseline --host-steps 0 --onehot-lines 0 --a2m-reads 0"
timeout -k 10 150 python bench.py --steps 20 --warmup 3 $Q > $O/c2.json 2> $O/c2.err || { tail -3 $O/c2.err; exit 1; }
timeout -k 10 150 python bench.py --workload c3 --steps 10 --warmup 3 $Q > $O/c3.json 2> $O/c3.err || { tail -3 $O/c3.err; exit 1; }
python - <<P
import json
for w in ("c2", "c3"):
    d = json.load(open("$O/%s.json" % w))
    print(w, round(d["value"] / 1e6, 1), d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["parity"]["lineage_equal"])
P
