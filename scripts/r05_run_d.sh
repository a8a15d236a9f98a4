#!/bin/bash
# GPU box: A/B of grouped-kernel variants (ab/lib*.so) against the shipped library, interleaved; then a few GPU tests on the shipped one
TAG=${1:-r05d}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "pairs or full_size or in_flight or closed_loop_in_one or sharded" > "$OUT/tests.log" 2>&1
tail -3 "$OUT/tests.log"
for rep in 1 2 3; do
  for m in shipped "$@"; do
    for c in 5 3; do for k in 1 3; do
      if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$m.so; fi
      timeout 300 python bench.py --config $c --in-flight $k --steps 16 --warmup 3 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$m rep$rep K$k', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done; done
  done
done
