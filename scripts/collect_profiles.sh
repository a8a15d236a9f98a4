#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects everything profiles/ and DESIGN.md quote for one build.
#   scripts/collect_profiles.sh TAG [quick]     ->  gpurun_out/TAG/...
# rocprofv3 counter passes are separate runs with --kernel-trace only (no sys/hip/hsa trace next to --pmc).
set -u
TAG=${1:-r02}
QUICK=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=python
BENCH="$PY $ROOT/bench.py"
ONLY="--no-cpu-baseline --no-extras"

# 1. the benchmark line as the driver runs it (headline = config 5, config 2 under extra, CPU baseline leg)
$BENCH > "$OUT/bench.json" 2> "$OUT/bench.err"

# 2. kernel trace + stats of the same command (headline workload only: every k_solve launch is the headline's).  Round 5: the
# default command keeps three batches in flight inside the handle, so the trace's per-kernel durations are those of OVERLAPPING
# launches (bench.py prints them as in_flight.kernel_ms_of_one_launch_while_overlapped); the second pass runs the same launches
# one at a time (--in-flight 1: what roofline.in_flight.sequential.kernel_ms of the default line measures)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_config5" -- $BENCH $ONLY > "$OUT/stats_config5.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_config5_seq" -- $BENCH --in-flight 1 $ONLY > "$OUT/stats_config5_seq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_config2" -- $BENCH --config 2 $ONLY > "$OUT/stats_config2.log" 2>&1

# 3. HBM traffic and SQ counters per workload, one small counter group per pass
wl_args() { case $1 in c5|c5alm) echo "--config 5";; c3) echo "--config 3";; c2) echo "--config 2";; c2b16k) echo "--config 2 --batch 16384";; c4) echo "--config 4";; esac; }
for wl in c5 c3 c2 c2b16k c4 c5alm; do
    if [ $wl = c5alm ]; then export CILQR_BENCH_ALM=1; else unset CILQR_BENCH_ALM; fi  # (the headline batch with solve_type alm)
    for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
        name=$(echo $grp | tr ' ' '+')
        rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_${wl}_$name" -- \
            $BENCH $(wl_args $wl) --in-flight 1 --steps 3 --warmup 1 $ONLY > "$OUT/pmc_${wl}_$name.log" 2>&1
    done
done
unset CILQR_BENCH_ALM

# 4. the other BASELINE configurations (one bench line each) and a batch sweep of config 2
for c in 2 3 4; do
    $BENCH --config $c --in-flight 1 --steps 5 --warmup 1 $ONLY > "$OUT/bench_config$c.json" 2> "$OUT/bench_config$c.err"
done
if [ -z "$QUICK" ]; then
for b in 512 2048 4096 16384; do
    $BENCH --config 2 --batch $b --in-flight 1 --steps 3 --warmup 1 $ONLY > "$OUT/bench_config2_B$b.json" 2>/dev/null
done
# 4b. batches in flight inside the handle: 1 .. 4 launch slots on the three large launches
for c in 5 3 4; do for k in 1 2 3 4; do
    $BENCH --config $c --in-flight $k --steps 24 --warmup 3 $ONLY > "$OUT/bench_inflight_c${c}_k$k.json" 2>/dev/null
done; done
fi

# 5. in-kernel phase accounting.  Configs 3 and 5 run the grouped build (two trajectories per wavefront, two wavefronts per
# SIMD: --group 2); next to it the one-trajectory build they ran before, accounted at two wavefronts per SIMD as well
# (CILQR_TUNE=prof2=1) and at one (the r01-r03 figures)
# (round 5, second half: configs[3]'s shard runs the grouped build's long layout too; the accounting of a sliced solve restarts
#  when it is resumed, so the grouped passes run with the slices switched off)
$PY $ROOT/scripts/phase_profile.py --config 2 > "$OUT/phase_config2.json" 2> "$OUT/phase_config2.err"
$PY $ROOT/scripts/phase_profile.py --config 4 > "$OUT/phase_config4_single_1wps.json" 2> "$OUT/phase_config4.err"
CILQR_TUNE=group_slice=0,group_slice_long=0 $PY $ROOT/scripts/phase_profile.py --config 4 --group 2 > "$OUT/phase_config4.json" 2>> "$OUT/phase_config4.err"
for c in 3 5; do
    CILQR_TUNE=group_slice=0,group_slice_long=0 $PY $ROOT/scripts/phase_profile.py --config $c --group 2 > "$OUT/phase_config$c.json" 2> "$OUT/phase_config$c.err"
    CILQR_TUNE=prof2=1 $PY $ROOT/scripts/phase_profile.py --config $c > "$OUT/phase_config${c}_single_2wps.json" 2>> "$OUT/phase_config$c.err"
    $PY $ROOT/scripts/phase_profile.py --config $c > "$OUT/phase_config${c}_single_1wps.json" 2>> "$OUT/phase_config$c.err"
done

# 6. the RCCL branch of bench.py on one rank (init_process_group("nccl") + the two all-reduces + barrier)
# (the default multi-GPU command: config 5 per rank as the headline, configs[3] and config 2 as extras)
CILQR_FORCE_DIST=1 $BENCH --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/bench_force_dist.json" 2> "$OUT/bench_force_dist.err"

# 7. BASELINE configs[0]: single ego, closed loop through the drop-in solve()
$BENCH --config 1 > "$OUT/bench_config1.json" 2> "$OUT/bench_config1.err"
ls "$OUT" | head -80

# 8. how the launches fill the chip: block timelines (raw records kept next to the summaries)
for c in 3 4 5; do
    TIMELINE_OUT="$OUT/timeline_c$c.npy" $PY $ROOT/scripts/block_timeline.py $c > "$OUT/timeline_config$c.json" 2> "$OUT/timeline_config$c.err"
done

# 8b. the same launches with one trajectory per wavefront (the r03 shape), for the timelines' before / after
for c in 3 4 5; do
    GROUP_MODE=0 $PY $ROOT/scripts/block_timeline.py $c > "$OUT/timeline_config${c}_single.json" 2>> "$OUT/timeline_config$c.err"
done
# 8c. sliced solves: kernel time, hand-overs and unfinished trajectories over time, by slice length (development library)
$PY $ROOT/scripts/slice_probe.py 3 "group_slice=0" "group_slice=8" "group_slice=16" "group_slice=32" > "$OUT/slices_config3.jsonl" 2> "$OUT/slices.err"
$PY $ROOT/scripts/slice_probe.py 4 "group_slice_long=0" "group_slice_long=8" "group_slice_long=12" "group_slice_long=24" > "$OUT/slices_config4.jsonl" 2>> "$OUT/slices.err"
$PY $ROOT/scripts/slice_probe.py 5 "group_slice=0" "group_slice=16" > "$OUT/slices_config5.jsonl" 2>> "$OUT/slices.err"

# 9. where the wave-cycles go (SQ wait / active counters), headline and the two 8192-trajectory launches
for a in "5 c5" "3 c3" "4 c4"; do set -- $a; $ROOT/scripts/stall_counters.sh $TAG "--config $1" $2 > /dev/null 2>&1; done

# 10. the N > 1 code path of bench.py rehearsed with two processes on this one GPU (statistics over gloo): not a measurement
CILQR_BENCH_ONE_DEVICE=1 CILQR_BENCH_BACKEND=gloo $PY -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29577 $ROOT/bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > "$OUT/two_rank.json" 2> "$OUT/two_rank.err"

# 11. the GPU test suite in both XNACK modes (this pool runs xnack-; HSA_XNACK=1 is the mode in which round 4's lost-store
# anomaly does not occur) — logs kept under profiles/
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > "$OUT/gpu_tests_xnack_off.log" 2>&1
HSA_XNACK=1 timeout 1800 python -m pytest tests -m gpu -q --durations=10 > "$OUT/gpu_tests_xnack_on.log" 2>&1
tail -n 2 "$OUT/gpu_tests_xnack_off.log"; tail -n 2 "$OUT/gpu_tests_xnack_on.log"
