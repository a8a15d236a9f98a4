#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/${1:-r03_i}; mkdir -p $OUT
DEV=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so
shift
for t in "$@"; do
  for rep in 1 2; do
    CILQR_AMD_LIB=$DEV CILQR_TUNE="$t" python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$t', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))"
  done
done
