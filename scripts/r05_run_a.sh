#!/bin/bash
# GPU box: the in-flight feature — its test, the whole GPU suite, the default bench line, in-flight sweep on configs 3 and 5
TAG=${1:-r05a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/tests.log" 2>&1
tail -15 "$OUT/tests.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=b['roofline']; e=b['extra']
print('headline', '%.4g it/s'%b['value'], 'ms/step %.3f'%b['ms_per_step'], 'eff kernel_ms %.3f'%r['kernel_ms'], r['in_flight'])
for k in ('config2_latency','config3','config4_sharded','config5_alm'):
    x=e.get(k)
    print(k, x and {kk:x.get(kk) for kk in ('value','ms_per_step','kernel_ms','in_flight','error')})
print('closed_loop', e.get('closed_loop') and {kk:e['closed_loop'].get(kk) for kk in ('ms_per_tick','ego_ticks_per_s','error')})
print('cpu_check', e.get('cpu_check'))
PY
for c in 3 5; do for k in 1 2 3 4; do
    timeout 600 python bench.py --config $c --in-flight $k --steps 24 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_c${c}_k$k.json" 2> "$OUT/bench_c${c}_k$k.err"
    python - "$OUT/bench_c${c}_k$k.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=b['roofline']
print(b['config']['workload'], 'K', b['config']['batches_in_flight'], '%.4g it/s'%b['value'], 'ms/step %.3f'%b['ms_per_step'], 'eff kernel %.3f'%r['kernel_ms'], (r['in_flight'].get('sequential') or {}).get('kernel_ms'), r['in_flight'].get('buffer_sets_identical'))
PY
done; done
