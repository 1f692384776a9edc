cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 > gpurun_out/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 > gpurun_out/pmc2.log 2>&1 || exit 1
ls gpurun_out/pmc1/*/ gpurun_out/pmc2/*/
