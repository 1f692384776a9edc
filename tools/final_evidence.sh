#!/bin/bash
# On the GPU box: what the round's claims rest on, from the build in the tree -> gpurun_out/r3final/.
#   part 1: smoke(), the whole GPU suite, the differential fuzz in its four modes (logs keep every seed)
#   part 2: profiles (tools/collect_profiles.sh r3) and the bench lines of the other BASELINE shapes
# Every GPU step runs under its own time limit and its exit status is checked on the spot: the first step that fails,
# faults or times out ends the script, nothing runs after it.
# usage: tools/final_evidence.sh 1|2
O=gpurun_out/r3final
mkdir -p $O || exit 1
Q="--host-steps 0 --onehot-lines 0 --a2m-reads 0"
if [ "$1" = 1 ]; then
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -3 $O/smoke.log; exit 1; }
  timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/gpu_suite.log 2>&1 || { tail -5 $O/gpu_suite.log; exit 1; }
  tail -1 $O/gpu_suite.log
  FUZZ_TRACE=$O/trace_auto.log timeout -k 10 300 python -u tools/fuzz_gpu.py 4000 100000 > $O/fuzz_auto.log 2>&1 || { tail -3 $O/fuzz_auto.log; exit 1; }
  tail -1 $O/fuzz_auto.log
  STAG_WALK=gather FUZZ_TRACE=$O/trace_gather.log timeout -k 10 200 python -u tools/fuzz_gpu.py 2000 10000 > $O/fuzz_gather.log 2>&1 || { tail -3 $O/fuzz_gather.log; exit 1; }
  tail -1 $O/fuzz_gather.log
  STAG_WALK=mfma FUZZ_TRACE=$O/trace_mfma.log timeout -k 10 200 python -u tools/fuzz_gpu.py 2000 200000 > $O/fuzz_mfma.log 2>&1 || { tail -3 $O/fuzz_mfma.log; exit 1; }
  tail -1 $O/fuzz_mfma.log
  FUZZ_TRACE=$O/trace_big.log timeout -k 10 300 python -u tools/fuzz_gpu.py 120 300000 big > $O/fuzz_big.log 2>&1 || { tail -3 $O/fuzz_big.log; exit 1; }
  tail -1 $O/fuzz_big.log
else
  bash tools/collect_profiles.sh r3 > $O/collect.log 2>&1 || { tail -5 $O/collect.log; exit 1; }
  timeout -k 10 150 python bench.py --workload c3 $Q > $O/bench_c3.json 2> $O/bench_c3.err || { tail -3 $O/bench_c3.err; exit 1; }
  timeout -k 10 150 python bench.py --workload c5 $Q > $O/bench_c5.json 2> $O/bench_c5.err || { tail -3 $O/bench_c5.err; exit 1; }
  STAG_WALK=mfma timeout -k 10 150 python bench.py --workload c5 $Q > $O/bench_c5_mfma.json 2> $O/bench_c5_mfma.err || { tail -3 $O/bench_c5_mfma.err; exit 1; }
  timeout -k 10 150 python bench.py --workload c3p $Q > $O/bench_c3p.json 2> $O/bench_c3p.err || { tail -3 $O/bench_c3p.err; exit 1; }
  timeout -k 10 300 python tools/bench_genome.py > $O/bench_genome.json 2> $O/bench_genome.err || { tail -3 $O/bench_genome.err; exit 1; }
  for f in c3 c5 c5_mfma c3p; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['parity'])"; done
  tail -c 400 $O/bench_genome.json
fi
