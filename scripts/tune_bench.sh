#!/bin/bash
# Runs ON THE GPU BOX: the development library (the production one does not read CILQR_TUNE) under different
# CILQR_TUNE settings, interleaved.
#   scripts/tune_bench.sh TAG "bench args" "tune1" "tune2" ...
TAG=$1; ARGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
for rep in 1 2; do
  for t in "$@"; do
      CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE="$t" python bench.py $ARGS --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('[$t] rep$rep', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/tune.txt"
  done
done
