#!/bin/bash
# Runs ON THE GPU BOX: the development library under several CILQR_TUNE settings, interleaved.
#   scripts/tune_ab.sh TAG "configs" steps reps "tune1" "tune2" ...
TAG=$1; CFGS=$2; STEPS=$3; REPS=$4; shift 4
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for rep in $(seq $REPS); do
  for t in "$@"; do
    for c in $CFGS; do
      CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=$t timeout 300 python bench.py --config $c --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$t rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done
  done
done
