cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 > gpurun_out/prof_b.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 > gpurun_out/pmc3.log 2>&1 || exit 1
