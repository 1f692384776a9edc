set -e
R=$PWD; O=$R/gpurun_out/pmcw; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --check 0 --host-steps 0 --onehot-lines 0 > $O/log.txt 2>&1
ls $O/*/ | head
