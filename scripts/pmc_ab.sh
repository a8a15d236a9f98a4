#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one workload for the grouped build (default) and the one-trajectory-per-wavefront
# build (development library, CILQR_TUNE=group=0), one pass each.   scripts/pmc_ab.sh TAG "--config 5"
set -u
TAG=${1:-pmc_ab}; ARGS=${2:---config 5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for m in pairs single; do
  if [ $m = single ]; then export CILQR_AMD_LIB=$ROOT/toy-example-of-ilqr_amd/libcilqr_amd_dev.so CILQR_TUNE=group=0; else unset CILQR_AMD_LIB CILQR_TUNE; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
      --kernel-trace --output-format csv -d "$OUT/$m" -- $BENCH > "$OUT/$m.log" 2>&1
  python - "$OUT/$m" $m <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        if 'k_solve' in r['Kernel_Name']: acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
print(sys.argv[2], {k: '%.4g'%(sum(v.values())/len(v)) for k,v in acc.items()})
PY
done
