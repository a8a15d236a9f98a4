#!/bin/bash
# GPU box: sliced solves in the grouped kernel — parity subset, then A/B by CILQR_TUNE on configs 4, 3, 5
TAG=${1:-r05i}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q -k "long_horizons or config4_every_rank or pairs or full_size_configs or solve_bitexact_with_trace or sweeps_of_two" > "$OUT/tests_quick.log" 2>&1
tail -15 "$OUT/tests_quick.log"
CFGS=${CFGS:-"4 3 5"}
for rep in 1 2; do
  for t in "$@"; do
    for c in $CFGS; do for k in 1 3; do
      CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=$t timeout 300 python bench.py --config $c --in-flight $k --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('[$t] rep$rep K$k', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done; done
  done
done
