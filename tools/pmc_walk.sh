#!/bin/bash
# SQ counters per dispatch of one C2 step (one pass: eight SQ counters + GRBM_GUI_ACTIVE), with the kernel trace for the
# durations -> gpurun_out/pmcw/; tools/pmc_walk_table.py prints the walk's launches.
R=$PWD; O=$R/gpurun_out/pmcw; rm -rf $O; mkdir -p $O || exit 1; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 --a2m-reads 0 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
ls $O/*/ | head
