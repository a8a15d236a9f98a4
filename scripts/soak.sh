#!/bin/bash
# Soak run ON THE GPU BOX (end of a round, final library): the randomised parity tests with other seeds and many more
# trials than the suite runs, and the closed loop in one launch against the tick-by-tick loop and the oracle.
#   scripts/soak.sh TAG [fuzz trials per seed] [closed-loop cases per seed]
TAG=${1:-soak}; TRIALS=${2:-160}; CASES=${3:-40}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
for seed in 11 12 13; do
  CILQR_FUZZ_SEED=$seed CILQR_FUZZ_TRIALS=$TRIALS timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz_random_parameter_sets > $OUT/fuzz_seed$seed.log 2>&1
  echo "fuzz seed $seed trials $TRIALS: $(tail -1 $OUT/fuzz_seed$seed.log)" | tee -a $OUT/summary.txt
done
for seed in 1 2 3; do
  timeout 1500 python scripts/soak_closed_loop.py --cases $CASES --seed $seed --out $OUT/closed_loop_seed$seed.json > $OUT/closed_loop_seed$seed.log 2>&1
  echo "closed loop seed $seed: $(tail -1 $OUT/closed_loop_seed$seed.log)" | tee -a $OUT/summary.txt
done
