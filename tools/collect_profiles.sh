#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench, the rocprofv3 kernel trace of the same
# command, and the two HBM-traffic counter passes (separate runs, as the MI355X guide prescribes).
# usage: tools/collect_profiles.sh <tag>   -> gpurun_out/final_<tag>/<run id>/, a directory of this run alone, named in
# gpurun_out/final_<tag>/run_id (gpurun MERGES what every run leaves under gpurun_out/ into the local copy, and a summary
# read from a directory that an earlier run also wrote into once reported a third of the real traffic);
# tools/summarize_profiles.py <tag> reads run_id and turns that one directory into profiles/<tag>_*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r3}
RUN=$(date +%s)_$$
O=$R/gpurun_out/final_$TAG/$RUN      # a directory of this run alone, whatever older runs left beside it after gpurun's merge
mkdir -p $O
echo $RUN > $R/gpurun_out/final_$TAG/run_id
cd /tmp && export TMPDIR=/tmp
QUIET="--no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 --a2m-reads 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 $QUIET > $O/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 $QUIET > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 $QUIET > $O/pmc_write.log 2>&1
# the counters become profiles/<tag>_pmc_traffic.json of THIS tree before the bench line is taken, so that the line carries
# the traffic of the kernels it timed (bench.py prints traffic only when that file is stamped with these sources)
(cd $R && python3 tools/summarize_profiles.py $TAG --pmc-only > $O/pmc_summary.txt)
timeout -k 10 600 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
