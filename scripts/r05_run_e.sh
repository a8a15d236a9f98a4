#!/bin/bash
# GPU box: the pair sweep — parity first (quick subset, then everything), then A/B against ab/lib*.so
TAG=${1:-r05e}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "pairs or full_size_configs or solve_bitexact_with_trace" > "$OUT/tests_quick.log" 2>&1
tail -12 "$OUT/tests_quick.log"
for rep in 1 2; do
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
unset CILQR_AMD_LIB
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1
tail -5 "$OUT/tests.log"
