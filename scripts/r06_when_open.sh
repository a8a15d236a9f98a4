#!/bin/bash
# GPU box, round 6: everything that has been waiting for the pool to open, most valuable first, every step under its own timeout
# so that a later step cannot cost an earlier one its result.  (ab/ travels for this call: the pending kernel changes as libraries
# built from the branches pending/*, profiles/r06_pending/.)
#   1. evidence at HEAD: GPU suite (XNACK off), smoke(), default bench line, GPU suite (HSA_XNACK=1)        ~15 min
#   2. A/B, one launch at a time, config 5 + config 3: shipped / dist_less out of line / rows of 24 doubles    ~10 min
#   3. augmented Lagrangian in pairs / horizons above 127 (in the shipped library since): numbers out of step 1's bench line
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
# 3. round 6's kernels that have never run on a GPU (they are in the shipped library: ALM in pairs is opt-in, horizons 128 ... 255 have
#    no other build): their tests ran inside step 1 (test_alm_in_pairs_per_wavefront, test_horizons_above_127,
#    test_a_lost_hand_over_is_loud); here the numbers — config 5 under ALM, lone wavefronts vs pairs — are in bench.json
#    (extra.config5_alm / extra.config5_alm_pairs); a stress pass of the grouped kernel with ALM shapes follows if time allows
python - "$OUT/bench.json" <<'PY'
import json,sys
try:
    b=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); e=b['extra']
    for k in ('config5_alm','config5_alm_pairs'):
        x=e.get(k) or {}
        print(k, {kk:x.get(kk) for kk in ('value','kernel_ms','error')}, (x.get('cpu_check') or {}).get('bit_identical_to_det_oracle'), (x.get('launch') or {}))
except Exception as ex:
    print('no bench line', ex)
PY
# 4. stress of the grouped kernel incl. round 6's shapes (ALM lone vs pairs, horizons 128 ... 200), five minutes
timeout 420 python scripts/stress_grouped.py 300 "$OUT/stress_grouped.json" 2>&1 | tail -2
