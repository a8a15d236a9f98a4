#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/${1:-r03_m}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q --durations=8 > $OUT/tests.log 2>&1; tail -14 $OUT/tests.log
scripts/ab_bench.sh ${1:-r03_m} "4 5 3 2" 8
for a in "--config 4 --batch 16384" "--config 4 --batch 4096" "--config 2 --horizon 80 --batch 8192"; do
 for rep in 1 2; do for v in A B; do
  CILQR_AMD_LIB=$ROOT/ab/lib$v.so python bench.py $a --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$v [$a]', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))"
 done; done
done
