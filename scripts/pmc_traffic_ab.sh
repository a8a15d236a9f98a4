#!/bin/bash
# Runs ON THE GPU BOX: HBM-side traffic counters (TCC_EA0_RDREQ / WRREQ family via FETCH_SIZE / WRITE_SIZE) of one workload
# for the shipped library and another build of it (ab/lib<NAME>.so).   scripts/pmc_traffic_ab.sh TAG NAME "--config 5"
set -u
TAG=${1:-pmc_traffic}; NAME=${2:-T1}; ARGS=${3:---config 5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for m in shipped $NAME; do
  if [ $m = shipped ]; then unset CILQR_AMD_LIB; else export CILQR_AMD_LIB=$ROOT/ab/lib$NAME.so; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do  # (one counter per pass: the two together abort rocprofv3 on this pool)
    timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/$m/$ctr" -- $BENCH > "$OUT/$m.$ctr.log" 2>&1
  done
  python - "$OUT/$m" $m <<'PY' | tee -a "$OUT/traffic.txt"
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(sys.argv[1]+'/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        if 'k_solve' in r['Kernel_Name']: acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
print(sys.argv[2], {k: '%.5g KiB/launch (raw counter units of 1 KiB)'%(sum(v.values())/len(v)) for k,v in acc.items()})
PY
done
