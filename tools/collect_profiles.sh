#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench, the rocprofv3 kernel trace of the same
# command, and the two HBM-traffic counter passes (separate runs, as the MI355X guide prescribes).
# Results land under gpurun_out/final/; tools/summarize_profiles.py turns them into profiles/*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 > $O/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 > $O/pmc_write.log 2>&1
cat $O/bench.json
