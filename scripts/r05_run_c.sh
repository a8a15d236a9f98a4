#!/bin/bash
# GPU box, round 5 run C: the grouped kernel as a dispatcher over LDS-resident state — whole GPU suite, stress run, A/B against
# the previous commit's library (ab/libPREV.so), interleaved, one launch at a time and three in flight.
TAG=${1:-r05c}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/tests.log" 2>&1
tail -4 "$OUT/tests.log"
timeout 400 python scripts/stress_grouped.py 240 > "$OUT/stress.json" 2> "$OUT/stress.err"; tail -c 600 "$OUT/stress.json"; tail -3 "$OUT/stress.err"
for rep in 1 2 3; do
  for m in shipped PREV; do
    for c in 5 3; do for k in 1 3; do
      if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$m.so; fi
      timeout 300 python bench.py --config $c --in-flight $k --steps 16 --warmup 3 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('$m rep$rep K$k', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']))" | tee -a "$OUT/ab.txt"
    done; done
  done
done
unset CILQR_AMD_LIB
timeout 600 python scripts/phase_profile.py --config 5 --group 2 > "$OUT/phase_config5.json" 2> "$OUT/phase_config5.err"
python - "$OUT/phase_config5.json" <<'PY'
import json,sys
p=json.load(open(sys.argv[1])); print(p['workload'], p['kernel_ms'], {k:round(v) for k,v in p['cycles_per_iteration'].items()}, round(p['cycles_per_trial_cost']), p.get('trial_cost_split_cycles_per_eval'))
PY
