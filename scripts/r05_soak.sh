#!/bin/bash
# GPU box, end of round 5: stress of the grouped kernel on NaN-poisoned scratch, then the soak (fuzzed parameter sets, closed loops)
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
mkdir -p gpurun_out/r05_soak
STRESS_POISON=1 timeout 900 python scripts/stress_grouped.py ${1:-600} gpurun_out/r05_soak/stress_poisoned.json > gpurun_out/r05_soak/stress_poisoned.log 2>&1
tail -2 gpurun_out/r05_soak/stress_poisoned.log
bash scripts/soak.sh r05_soak/soak 120 30
