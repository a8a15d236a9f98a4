#!/bin/bash
# GPU box: parity subset, then the shipped library against ab/lib<NAME>.so on configs 5, 3, 4, one launch at a time and three in flight
TAG=${1:-r05m}; NAME=${2:-NODMA}; CFGS=${3:-"5 3 4"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q -k "long_horizons or config4_every_rank or pairs or full_size_configs or solve_bitexact_with_trace or sweeps_of_two or sliced" > "$OUT/tests_quick.log" 2>&1
tail -5 "$OUT/tests_quick.log"
for rep in 1 2; do
  for m in shipped $NAME; do
    for c in $CFGS; do for k in 1 3; do
      if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$m.so; fi
      timeout 300 python bench.py --config $c --in-flight $k --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$m rep$rep K$k', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done; done
  done
done
unset CILQR_AMD_LIB
