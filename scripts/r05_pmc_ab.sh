#!/bin/bash
# GPU box: SQ counters of one workload for the shipped library and ab/lib<NAME>.so, one pass each (one launch at a time).
#   scripts/r05_pmc_ab.sh TAG "--config 5" NAME [NAME2 ...]
set -u
TAG=${1:-pmc_ab}; ARGS=${2:---config 5}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py $ARGS --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for m in shipped "$@"; do
  if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$m.so; fi
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    name=$(echo $grp | tr ' ' '+')
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/${m}_$name" -- $BENCH > "$OUT/${m}_$name.log" 2>&1
    python - "$OUT/${m}_$name" $m <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        if 'k_solve' in r['Kernel_Name']: acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
print(sys.argv[2], {k: '%.5g'%(sum(v.values())/len(v)) for k,v in acc.items()})
PY
  done
done 2>&1 | tee "$OUT/summary.txt"
