#!/bin/bash
# Runs ON THE GPU BOX: the grouped build (default) against the one-trajectory-per-wavefront build (development library,
# CILQR_TUNE=group=0) on the same box, interleaved.   scripts/grp_ab.sh TAG "configs" steps [reps] [notest]
TAG=${1:-grp_ab}; CFGS=${2:-"5 3"}; STEPS=${3:-8}; REPS=${4:-2}; MODES=${MODES:-"pairs single"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
if [ -z "${5:-}" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pairs" > "$OUT/tests.log" 2>&1; tail -3 "$OUT/tests.log"
fi
for rep in $(seq $REPS); do
  for m in $MODES; do
    for c in $CFGS; do
      if [ $m = single ]; then export CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=group=0; elif [ $m = triples ]; then export CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=group=3; elif [ $m = g1 ]; then export CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=group=4; elif [ $m = tune ]; then export CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=$TUNE; else unset CILQR_AMD_LIB CILQR_TUNE; fi
      timeout 300 python bench.py --config $c --steps $STEPS --warmup 2 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$m rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done
  done
done
