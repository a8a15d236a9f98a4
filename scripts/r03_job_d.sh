#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/${1:-r03_d}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "detmath or stages or solve_bitexact or alm_stages or full_size" > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log
scripts/ab_bench.sh ${1:-r03_d} "5 3 4 2" 8
python bench.py --config 1 > $OUT/bench_config1.json 2>$OUT/bench_config1.err; python - <<PY
import json
b=json.loads([l for l in open('$OUT/bench_config1.json') if l.startswith('{')][-1]); print('config1', b['ms_per_step'], b['extra']['tick_split_ms'], b['extra'].get('closed_loop_ticks_bit_identical_to_det_oracle'))
PY
