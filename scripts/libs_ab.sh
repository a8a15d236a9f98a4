#!/bin/bash
# Runs ON THE GPU BOX: the shipped library against several other builds of it (ab/lib<NAME>.so), interleaved.
#   scripts/libs_ab.sh TAG "configs" steps reps NAME1 NAME2 ...
TAG=$1; CFGS=$2; STEPS=$3; REPS=$4; shift 4
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for rep in $(seq $REPS); do
  for m in shipped "$@"; do
    for c in $CFGS; do
      if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$m.so; fi
      timeout 300 python bench.py --config $c --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$m rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done
  done
done
