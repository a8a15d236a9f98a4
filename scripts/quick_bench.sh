#!/bin/bash
# Runs ON THE GPU BOX: the short loop used while tuning — parity tests that cover every kernel variant, then one
# bench line per configuration and the phase accounting.   scripts/quick_bench.sh TAG [notest]
TAG=${1:-quick}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ -z "${2:-}" ]; then
python -m pytest tests -m gpu -x -q -k "bitexact or transparent or statistics or closed_loop or horizon or edge" > "$OUT/tests.log" 2>&1
tail -3 "$OUT/tests.log"
fi
for c in 5 3 2 4; do
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"
    python - "$OUT/bench_config$c.json" <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(b['config']['workload'], '%.4g it/s'%b['value'], '%.3f ms'%b['roofline']['kernel_ms'])
PY
done
for c in 2 5; do python scripts/phase_profile.py --config $c > "$OUT/phase_config$c.json" 2>/dev/null; python - "$OUT/phase_config$c.json" <<'PY'
import json,sys
p=json.load(open(sys.argv[1])); print(p['workload'], p['kernel_ms'], {k:round(v) for k,v in p['cycles_per_iteration'].items()}, round(p['cycles_per_trial_cost']))
PY
done
