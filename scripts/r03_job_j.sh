#!/bin/bash
# bench args x tune settings, development library, interleaved:  scripts/r03_job_j.sh TAG "args1|args2" "tune1" "tune2" ...
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/${1:-r03_j}; mkdir -p $OUT
DEV=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so
IFS='|' read -ra ARGS <<< "$2"
shift 2
for a in "${ARGS[@]}"; do
 for rep in 1 2; do
  for t in "$@"; do
    CILQR_AMD_LIB=${LIB:-$DEV} CILQR_TUNE="$t" python bench.py $a --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('[$a] $t', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a $OUT/tune.txt
  done
 done
done
