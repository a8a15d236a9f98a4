#!/bin/bash
# Runs ON THE GPU BOX: A/B of ab/libA.so vs ab/libB.so on arbitrary bench arguments, interleaved.
#   scripts/ab_args.sh TAG "bench args" ["bench args" ...]
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for ARGS in "$@"; do
  for rep in 1 2; do
    for v in A B; do
      CILQR_AMD_LIB=$ROOT/ab/lib$v.so python bench.py $ARGS --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$v rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done
  done
done
