#!/bin/bash
# GPU box, round 6: everything that has been waiting for the pool to open, most valuable first, every step under its own timeout
# so that a later step cannot cost an earlier one its result.  (ab/ travels for this call: the pending kernel changes as libraries
# built from the branches pending/*, profiles/r06_pending/.)
#   1. evidence at HEAD: GPU suite (XNACK off), smoke(), default bench line, GPU suite (HSA_XNACK=1)        ~15 min
#   2. A/B, one launch at a time, config 5 + config 3: shipped / dist_less out of line / rows of 24 doubles    ~10 min
#   3. augmented Lagrangian in pairs: -k alm with both libraries swapped, then config 5 under ALM (1 024-row oracle check)
TAG=${1:-r06w}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash scripts/r06_run_a.sh $TAG
if [ -f ab/libDL.so ]; then
  timeout 1200 bash scripts/libs_ab.sh $TAG/ab "5 3" 12 3 DL R24 2>&1 | tail -40
  (cd "$OUT"; [ -d ab ] && cat ab/ab.txt | sort | awk '{k=$1" "$3; s[k]+=$(NF-1); n[k]++} END{for(k in s) printf "%s mean kernel_ms %.3f (n=%d)\n", k, s[k]/n[k], n[k]}' | sort) | tee "$OUT/ab_summary.txt"
fi
if [ -f ab/libALM.so ]; then
  CILQR_AMD_LIB=$ROOT/ab/libALM.so CILQR_AMD_LIB_DEV=$ROOT/ab/libALM_dev.so timeout 1200 python -m pytest tests -m gpu -q -k "alm" > "$OUT/alm_pairs_tests.log" 2>&1
  tail -5 "$OUT/alm_pairs_tests.log"
  for lib in shipped ALM; do
    if [ $lib = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/libALM.so; fi
    CILQR_BENCH_ALM=1 timeout 600 python bench.py --config 5 --in-flight 1 --steps 5 --warmup 1 --no-extras > "$OUT/bench_c5alm_$lib.json" 2> "$OUT/bench_c5alm_$lib.err"
    python - "$OUT/bench_c5alm_$lib.json" $lib <<'PY'
import json,sys
try:
    b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('c5 alm', sys.argv[2], '%.4g it/s'%b['value'], 'kernel_ms %.3f'%b['roofline']['kernel_ms'], b['roofline']['launch'], (b['extra'].get('cpu_check') or {}).get('bit_identical_to_det_oracle'))
except Exception as e:
    print('c5 alm', sys.argv[2], 'FAILED', e)
PY
  done
  unset CILQR_AMD_LIB
fi
