#!/bin/bash
# GPU box, round 6 first call: evidence at HEAD before csrc/ is touched — GPU suite in both XNACK modes (the lost-rows test no
# longer skips), smoke(), the default bench line
TAG=${1:-r06a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -rs > "$OUT/gpu_tests_xnack_off.log" 2>&1
tail -4 "$OUT/gpu_tests_xnack_off.log"
timeout 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=b['roofline']; e=b['extra']
print('headline', '%.4g it/s'%b['value'], 'ms/step %.3f'%b['ms_per_step'], 'kernel_ms one at a time %.3f'%r['kernel_ms'], 'value one at a time %.4g'%b['value_one_batch_at_a_time'], r['in_flight'])
for k in ('config2_latency','config3','config4_sharded','config5_alm'):
    x=e.get(k)
    print(k, x and {kk:x.get(kk) for kk in ('value','ms_per_step','kernel_ms','in_flight','error')})
print('closed_loop', e.get('closed_loop') and {kk:e['closed_loop'].get(kk) for kk in ('ms_per_tick','ego_ticks_per_s','error')})
print('cpu_check', e.get('cpu_check'))
print('cpu_baseline', b.get('cpu_baseline'))
PY
HSA_XNACK=1 timeout 1800 python -m pytest tests -m gpu -q --durations=5 -rs > "$OUT/gpu_tests_xnack_on.log" 2>&1
tail -3 "$OUT/gpu_tests_xnack_on.log"
