#!/bin/bash
# round-3 GPU job B: parity tests (production + development library), the default bench line with its extras, A/B against
# HEAD, timeline + phase accounting of config 4
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
OUT=gpurun_out/r03_b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q --durations=15 > $OUT/tests.log 2>&1; tail -22 $OUT/tests.log
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
try:
    b=json.loads([l for l in open('gpurun_out/r03_b/bench_default.json') if l.startswith('{')][-1])
    print('headline', b['value'], b['ms_per_step'], b['roofline']['kernel_ms'])
    e=b['extra']
    for k in ('config2_latency','config4_sharded','closed_loop','config5_alm'):
        v=e.get(k); print(k, {kk: v[kk] for kk in v if kk in ('value','ms_per_step','kernel_ms','error','ego_ticks_per_s','ms_per_tick','cpu_check','iterations_per_ego_later_ticks_warm')} if v else v)
    print('ranks', e['ranks'], e['per_rank'])
    print('cpu', b.get('cpu_baseline',{}).get('value'), e.get('cpu_check'))
except Exception as ex: print('bench parse failed', ex)
PY
scripts/ab_bench.sh r03_b "4 3 5 2" 8
TIMELINE_OUT=$OUT/timeline_c4.npy python scripts/block_timeline.py 4 > $OUT/timeline_c4.json 2>$OUT/timeline_c4.err
PHASE_OUT=$OUT/phase_c4.npy python scripts/phase_profile.py --config 4 > $OUT/phase_c4.json 2>$OUT/phase_c4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_b/timeline_c4.json')); print('timeline_c4', d['kernel_ms'], d['mean_resident_blocks'], d['block_ms'], d['resident_blocks_in_20_time_slices'])
d=json.load(open('gpurun_out/r03_b/phase_c4.json')); print('phase_c4', d['kernel_ms'], d['ref_scan_fallbacks'], d['cycles_per_trial_cost'], d['trial_cost_split_cycles_per_eval'], d['slowest_8'][:3])
PY
