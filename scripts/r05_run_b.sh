#!/bin/bash
# GPU box, round 5 run B: new build table + in-flight streams at high priority: whole GPU suite (XNACK off as the pool runs,
# then once with HSA_XNACK=1), the default bench line, the lost-rows reproduction with the real instruction stream.
TAG=${1:-r05b}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/tests_xnack_off.log" 2>&1
tail -4 "$OUT/tests_xnack_off.log"
HSA_XNACK=1 timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests_xnack_on.log" 2>&1
tail -2 "$OUT/tests_xnack_on.log"
for x in "" 1; do
  if [ -n "$x" ]; then export HSA_XNACK=1; else unset HSA_XNACK; fi
  CILQR_AMD_LIB=$ROOT/ab/libLR.so timeout 600 python scripts/lost_rows_repro.py 12 2>>"$OUT/lost_rows.err" | tee -a "$OUT/lost_rows.jsonl"
  timeout 600 python scripts/lost_rows_repro.py 12 2>>"$OUT/lost_rows.err" | tee -a "$OUT/lost_rows.jsonl"
done
unset HSA_XNACK
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=b['roofline']; e=b['extra']
print('headline', '%.4g it/s'%b['value'], 'ms/step %.3f'%b['ms_per_step'], r['in_flight'])
for k in ('config2_latency','config3','config4_sharded','config5_alm'):
    x=e.get(k)
    print(k, x and {kk:x.get(kk) for kk in ('value','ms_per_step','error')}, x and (x.get('in_flight') or {}).get('sequential'))
for k in ('closed_loop','closed_loop_N30'):
    print(k, e.get(k) and {kk:e[k].get(kk) for kk in ('ms_per_tick','ego_ticks_per_s','value','error','cpu_check')})
PY
for c in 3 5; do for k in 2 3 4; do
    timeout 600 python bench.py --config $c --in-flight $k --steps 24 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_c${c}_k$k.json" 2> "$OUT/bench_c${c}_k$k.err"
    python - "$OUT/bench_c${c}_k$k.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r=b['roofline']
print(b['config']['workload'], 'K', b['config']['batches_in_flight'], '%.4g it/s'%b['value'], 'ms/step %.3f'%b['ms_per_step'], (r['in_flight'].get('sequential') or {}).get('kernel_ms'), r['in_flight'].get('kernel_ms_of_one_launch_while_overlapped (last launch of each slot)'))
PY
done; done
timeout 300 python bench.py --config 1 > "$OUT/bench_config1.json" 2> "$OUT/bench_config1.err"; python - "$OUT/bench_config1.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('config1', b['ms_per_step'], b['extra']['tick_split_ms'], b.get('cpu_baseline',{}).get('solve_latency_ms_mean'))
PY
