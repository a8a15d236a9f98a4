#!/bin/bash
# round-3 evidence run ON THE GPU BOX: the GPU suite with its slowest tests, then everything profiles/ quotes
TAG=${1:-r03_v1}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=15 > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
scripts/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
ls $OUT | wc -l
