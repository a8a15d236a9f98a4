#!/bin/bash
# round-3 GPU job C: the full GPU suite, then the libm-tolerance report (HIP results from the GPU)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/r03_c; mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=15 > $OUT/tests.log 2>&1; tail -30 $OUT/tests.log
python tests/libm_tolerance.py --gpu --configs 2 3 5 4 2alm 1 --threads 16 --out $OUT/r03_libm_tolerance.json > $OUT/libm.log 2>&1; tail -c 600 $OUT/libm.log
