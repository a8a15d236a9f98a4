#!/bin/bash
# Runs ON THE GPU BOX: A/B of two builds of the library on the same box, interleaved.
#   [VARIANTS="A B C"] scripts/ab_bench.sh TAG "configs" steps     (ab/libA.so vs ab/libB.so [vs ab/libC.so ...])
TAG=${1:-ab}; CFGS=${2:-"5 3 2"}; STEPS=${3:-20}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for rep in 1 2 3; do
  for v in ${VARIANTS:-A B}; do
    for c in $CFGS; do
      CILQR_AMD_LIB=$ROOT/ab/lib$v.so python bench.py --config $c --steps $STEPS --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$v rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done
  done
done
