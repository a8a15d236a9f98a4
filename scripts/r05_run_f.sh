#!/bin/bash
# GPU box: the pair sweep, final form — whole GPU suite (XNACK off), stress, default bench line, phase accounting, counters
TAG=${1:-r05f}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/tests.log" 2>&1
tail -6 "$OUT/tests.log"
timeout 400 python scripts/stress_grouped.py 200 > "$OUT/stress.json" 2> "$OUT/stress.err"; tail -c 400 "$OUT/stress.json"; tail -2 "$OUT/stress.err"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=b['roofline']; e=b['extra']
print('headline', '%.4g it/s'%b['value'], 'ms/step %.3f'%b['ms_per_step'], 'seq', r['in_flight'].get('sequential'))
for k in ('config2_latency','config3','config4_sharded','config5_alm'):
    x=e.get(k)
    print(k, x and {kk:x.get(kk) for kk in ('value','ms_per_step','error')}, x and (x.get('in_flight') or {}).get('sequential'))
for k in ('closed_loop','closed_loop_N30'):
    print(k, e.get(k) and {kk:e[k].get(kk) for kk in ('ms_per_tick','ego_ticks_per_s','value','error','cpu_check')})
print(e.get('cpu_check'))
PY
for c in 5 3; do
timeout 600 python scripts/phase_profile.py --config $c --group 2 > "$OUT/phase_config$c.json" 2> "$OUT/phase_config$c.err"
python - "$OUT/phase_config$c.json" <<'PY'
import json,sys
p=json.load(open(sys.argv[1])); print(p['workload'], p['kernel_ms'], {k:round(v) for k,v in p['cycles_per_iteration'].items()}, round(p['cycles_per_trial_cost']))
PY
done
bash scripts/r05_pmc_ab.sh ${TAG}_pmc "--config 5"
