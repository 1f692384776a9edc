# Timing ablations of k_walk_level on the bench workload (results are wrong by construction; only the
# per-kernel times matter).  Build: for v in 1 2 3 4 5; do make -C stag_amd/csrc OUT=../lib/abl$v.so EXTRA=-DSTAG_ABL=$v ../lib/abl$v.so; done
#   1 no coefficient loops   2 no sigmoid   3 no row staging   4 no gap-constant gathers   5 tasks fetched, nothing else
set -e
for v in "$@"; do
  STAG_HIP_LIB=$PWD/stag_amd/lib/$v.so timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 0 > gpurun_out/$v.json 2> gpurun_out/$v.err
  python -c "
import json,sys;d=json.load(open('gpurun_out/$v.json'));print('$v',round(d['value']/1e6,1),d['roofline']['kernel_ms_per_step'])"
done
