#!/bin/bash
# round-3 GPU job A: parity tests of the working tree, A/B against HEAD, timelines + phase accounting of config 4
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/r03_a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q --durations=15 > $OUT/tests.log 2>&1; tail -25 $OUT/tests.log
scripts/ab_bench.sh r03_a "4 3 5 2" 8
TIMELINE_OUT=$OUT/timeline_c4.npy python scripts/block_timeline.py 4 > $OUT/timeline_c4.json 2>$OUT/timeline_c4.err
TIMELINE_OUT=$OUT/timeline_c3.npy python scripts/block_timeline.py 3 > $OUT/timeline_c3.json 2>$OUT/timeline_c3.err
PHASE_OUT=$OUT/phase_c4.npy python scripts/phase_profile.py --config 4 > $OUT/phase_c4.json 2>$OUT/phase_c4.err
CILQR_AMD_LIB=$ROOT/ab/libA.so PHASE_OUT=$OUT/phase_c4_A.npy python scripts/phase_profile.py --config 4 > $OUT/phase_c4_A.json 2>$OUT/phase_c4_A.err
python - <<'PY'
import json
for f in ('timeline_c4','timeline_c3'):
    d=json.load(open(f'gpurun_out/r03_a/{f}.json')); print(f, d['kernel_ms'], d['mean_resident_blocks'], d['block_ms'])
for f in ('phase_c4','phase_c4_A'):
    d=json.load(open(f'gpurun_out/r03_a/{f}.json')); print(f, d['kernel_ms'], d['ref_scan_fallbacks'], d['cycles_per_trial_cost'], d['trial_cost_split_cycles_per_eval'], d['slowest_8'][:3])
PY
