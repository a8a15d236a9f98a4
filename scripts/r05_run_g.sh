#!/bin/bash
# GPU box: the grouped kernel's long layout (horizons 64 ... 127) — parity, then A/B on configs[3]'s shard against k_solve's build
TAG=${1:-r05g}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q -k "long_horizons or long_horizon_builds or horizon_boundaries or config4_every_rank" > "$OUT/tests_quick.log" 2>&1
tail -15 "$OUT/tests_quick.log"
for rep in 1 2; do
  for t in "" "group_long=0"; do
    for k in 1 3; do
      CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=$t timeout 300 python bench.py --config 4 --in-flight $k --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>>"$OUT/err.log" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        b=json.loads(l); print('[$t] rep$rep K$k', b['config']['workload'], '%.5g it/s %.4f ms'%(b['value'], b['roofline']['kernel_ms']), b['roofline'].get('launch'))" | tee -a "$OUT/ab.txt"
    done
  done
done
