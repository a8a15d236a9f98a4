#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/${1:-r03_k}; mkdir -p $OUT
for c in 3 5 2; do
 for rep in 1 2 3; do
  for v in A B; do
   for t in "resume_iters=0" "resume_iters=16" "resume_iters=32"; do
    [ $v = A ] && [ "$t" != "resume_iters=0" ] && continue
    CILQR_AMD_LIB=$ROOT/ab/lib$v.so CILQR_TUNE="$t" python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$v $t', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a $OUT/ab.txt
   done
  done
 done
done
