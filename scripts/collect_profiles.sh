#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects everything profiles/ and DESIGN.md quote for one build.
#   scripts/collect_profiles.sh TAG      ->  gpurun_out/TAG/...
# rocprofv3 counter passes are separate runs with --kernel-trace only (no sys/hip/hsa trace next to --pmc).
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python
BENCH="$PY $ROOT/bench.py"

# 1. the benchmark line as the driver runs it (includes the CPU baseline leg)
$BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"

# 2. kernel trace + stats of the same command (without the CPU leg)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --no-cpu-baseline > "$OUT/stats.log" 2>&1

# 3. HBM traffic and SQ counters, one small group per pass
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
    name=$(echo $grp | tr ' ' '+')
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- $BENCH --steps 3 --warmup 1 --no-cpu-baseline \
        > "$OUT/pmc_$name.log" 2>&1
done

# 4. the other BASELINE configurations (parity-test cases, one bench line each)
for c in 3 4 5; do
    $BENCH --config $c --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"
done
# batch sweep of config 2 (how much of the chip one launch fills)
for b in 512 2048 4096 16384; do
    $BENCH --config 2 --batch $b --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_config2_B$b.json" 2>/dev/null
done

# 4b. four batches in flight (extra.pipelined)
$BENCH --streams 4 --steps 40 --warmup 3 --no-cpu-baseline > "$OUT/bench_pipelined.json" 2>/dev/null

# 5. in-kernel phase accounting
$PY $ROOT/scripts/phase_profile.py --config 2 > "$OUT/phase_config2.json" 2> "$OUT/phase_config2.err"
$PY $ROOT/scripts/phase_profile.py --config 3 --batch 8192 > "$OUT/phase_config3.json" 2> "$OUT/phase_config3.err"
ls -R "$OUT" | head -50
