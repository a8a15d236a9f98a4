#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/${1:-r03_h}; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "config4 or long_horizon or work_sharing or alm or stay_behind or full_size" > $OUT/tests.log 2>&1; tail -6 $OUT/tests.log
scripts/ab_bench.sh ${1:-r03_h} "4" 6
DEV=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so
for t in "resume_iters=0" "resume_iters=16" "resume_iters=32" "resume_iters=48"; do
  for rep in 1 2; do
    CILQR_AMD_LIB=$DEV CILQR_TUNE="$t" python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$t', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))"
  done
done
TIMELINE_OUT=$OUT/timeline_c4.npy python scripts/block_timeline.py 4 > $OUT/timeline_c4.json 2>$OUT/timeline_c4.err; python -c "
import json; d=json.load(open('$OUT/timeline_c4.json')); print(d['kernel_ms'], d['mean_resident_blocks'], d['resident_blocks_in_20_time_slices'])"
