#!/bin/bash
# Runs ON THE GPU BOX: where do the wave-cycles of the solve kernel go?  Two SQ counter passes per workload.
#   scripts/stall_counters.sh TAG "--config 5" name
set -u
TAG=${1:-r02}; ARGS=${2:---config 5}; NAME=${3:-c5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F64"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/stall_${NAME}_$i" -- $BENCH > "$OUT/stall_${NAME}_$i.log" 2>&1
done
