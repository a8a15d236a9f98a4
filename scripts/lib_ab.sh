#!/bin/bash
# Runs ON THE GPU BOX: the shipped library against another build of it (ab/lib<NAME>.so), interleaved.
#   scripts/lib_ab.sh TAG NAME "configs" steps [reps]
TAG=${1:-lib_ab}; NAME=${2:-T1}; CFGS=${3:-"5 3"}; STEPS=${4:-8}; REPS=${5:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for rep in $(seq $REPS); do
  for m in shipped $NAME; do
    for c in $CFGS; do
      if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$NAME.so; fi
      timeout 300 python bench.py --config $c --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$m rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done
  done
done
