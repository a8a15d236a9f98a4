#!/usr/bin/env python3
"""Turn gpurun_out/TAG (written by scripts/collect_profiles.sh on the GPU box) into the small, tracked
summaries under profiles/.   usage: scripts/summarize_profiles.py TAG [--current]

--current also rewrites profiles/pmc_current.json, the file bench.py reads `roofline.traffic` and
`roofline.valu_issue` from (keyed by workload name).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def last_json(path):
    if not os.path.exists(path) or not os.path.getsize(path):
        return None
    for line in reversed(open(path, errors="replace").read().strip().splitlines()):
        if line.startswith("{"):  # RCCL prints a version banner after the line on some runs
            try:
                return json.loads(line)
            except Exception:
                pass
    return None


def counters(pattern):
    """per-launch value of every counter in the pass (summed over the dispatch's rows, averaged over launches)"""
    out = {}
    for path in glob.glob(os.path.join(src, pattern, "*", "*_counter_collection.csv")):
        per = defaultdict(lambda: defaultdict(float))
        kern = None
        for r in csv.DictReader(open(path)):
            if "k_solve" in r["Kernel_Name"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                kern = r["Kernel_Name"]
        for name, d in per.items():
            out[name] = sum(d.values()) / len(d)
            out[name + "_launches"] = len(d)
        if kern:
            out["kernel"] = kern
    return out


bench = last_json(os.path.join(src, "bench.json"))
if bench:
    json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
for c in ("config5", "config5_seq", "config2"):
    stats = glob.glob(os.path.join(src, f"stats_{c}", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_{c}_kernel_stats.csv"))

# per-workload counters.  The workload name of each pass comes from the bench line the pass printed.
pmc_all = {}
for wl in ("c5", "c3", "c2", "c2b16k", "c4", "c5alm"):
    name = alg = iters = kernel_ms = None
    for log in glob.glob(os.path.join(src, f"pmc_{wl}_*.log")):
        for line in open(log, errors="replace"):
            if line.startswith('{"metric"'):
                b = json.loads(line)
                name = b["config"]["workload"]
                alg = b["roofline"]["algorithmic_bytes_per_launch"]
                iters = b["extra"]["iterations_per_step_rank0"]
    if name is None:
        continue
    f = counters(f"pmc_{wl}_FETCH_SIZE")
    w = counters(f"pmc_{wl}_WRITE_SIZE")
    sq = {}
    for pat in glob.glob(os.path.join(src, f"pmc_{wl}_SQ*")):
        if os.path.isdir(pat):
            sq.update({k: v for k, v in counters(os.path.basename(pat)).items() if not k.endswith("_launches")})
    if "FETCH_SIZE" not in f or "WRITE_SIZE" not in w:
        continue
    ent = {
        "workload": name,
        "kernel": f.get("kernel"),
        "source": f"profiles/{tag}_pmc.json: rocprofv3 --pmc <group> --kernel-trace -- python bench.py <workload> --steps 3 "
                  "--warmup 1 --no-cpu-baseline --no-extras, one counter group per pass (scripts/collect_profiles.sh); "
                  "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB = bytes that cross the L2 <-> fabric boundary, i.e. Infinity "
                  "Cache + HBM together, NOT HBM alone.  Calibrated with known-byte kernels in this solver's access widths "
                  "through the same passes (profiles/r03_traffic_calibration.json): FETCH_SIZE reports exactly half the "
                  "fetched bytes for coalesced 16-B AND 8-B reads (128-B requests tallied at 64 B) and one 64-B tally (one "
                  "128-B line) per 8-B read at the slab's 160-B stride; WRITE_SIZE reports the written bytes as they are "
                  "(1.00 coalesced, 1.04 for the rollout's 160-byte rows); a 64 MiB working set re-read eight times — served "
                  "by the 256 MiB Infinity Cache after the first pass — counts the same as a 1 GiB one, so cache hits there "
                  "ARE counted",
        "FETCH_SIZE_KiB_per_launch": f["FETCH_SIZE"],
        "WRITE_SIZE_KiB_per_launch": w["WRITE_SIZE"],
        "hbm_bytes_per_launch_raw": (f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0,
        "hbm_bytes_per_launch_corrected": (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0,
        "algorithmic_bytes_per_launch": alg,
        "iterations_per_launch": iters,
    }
    ent["traffic_over_algorithmic"] = ent["hbm_bytes_per_launch_corrected"] / alg if alg else None
    ent["hbm_bytes_per_iteration_corrected"] = ent["hbm_bytes_per_launch_corrected"] / iters if iters else None
    for k, v in sq.items():
        if k != "kernel":
            ent[k] = v
    pmc_all[name] = ent
if pmc_all:
    json.dump(pmc_all, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
    if "--current" in sys.argv:
        # what the passes were collected on: bench.py compares the fingerprint of the kernel sources with its own and flags
        # roofline.traffic_collected.stale when csrc/ has changed since (VERDICT r04 item 8 / task 10)
        import subprocess
        sys.path.insert(0, ROOT)
        import bench as _bench
        try:
            rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
            dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "toy-example-of-ilqr_amd/csrc", "include"],
                                        capture_output=True, text=True).stdout.strip())
        except Exception:
            rev, dirty = None, None
        sha = _bench.csrc_fingerprint()
        stamp = os.path.join(src, "_collected.json")  # written when the collection was LAUNCHED (scripts/r05_collect.sh): the tree may have moved on since
        if os.path.exists(stamp):
            st_ = json.load(open(stamp))
            rev, dirty, sha = st_.get("git", rev), st_.get("csrc_dirty_at_collection", dirty), st_.get("csrc_sha16", sha)
        cur = dict(pmc_all)
        # the machine code the passes ran (scripts/device_code_identity.py): bench.py compares the library it runs with this
        manifest_rel = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import device_code_identity as dci
            manifest_rel = f"profiles/{tag}_device_code.json"
            json.dump({"lib": "libcilqr_amd.so", "functions": dci.manifest(os.path.join(ROOT, "toy-example-of-ilqr_amd", "libcilqr_amd.so"))},
                      open(os.path.join(ROOT, manifest_rel), "w"), indent=0, sort_keys=True)
        except Exception as e:  # noqa: BLE001
            print("no device-code manifest:", e, file=sys.stderr)
            manifest_rel = None
        cur["_collected"] = {"git": rev, "csrc_dirty_at_collection": dirty, "csrc_sha16": sha, "in_flight": 1,
                             "device_code_manifest": manifest_rel, "tag": tag, "how": "scripts/collect_profiles.sh: one counter group per rocprofv3 pass, launches one at a time"}
        json.dump(cur, open(os.path.join(dst, "pmc_current.json"), "w"), indent=1)

other = {"note": "python bench.py --config C [--batch B] --steps 3..5 --warmup 1 --no-cpu-baseline --no-extras on one "
                 "MI355X: the other BASELINE configurations and a batch sweep of config 2"}
for path in sorted(glob.glob(os.path.join(src, "bench_config[2-5]*.json"))):
    b = last_json(path)
    if not b:
        continue
    e = b["extra"]
    other[b["config"]["workload"]] = {
        "it_per_s": round(b["value"]), "ms_per_step": round(b["ms_per_step"], 3), "solves_per_s": round(e["solves_per_s"]),
        "iters_per_solve": round(e["iterations_per_solve_mean"], 2), "converged": e["converged"],
        "max_lamb": e["max_lamb"], "max_iter": e["max_iter"], "hbm_frac": round(b["roofline"]["frac"], 5)}
json.dump(other, open(os.path.join(dst, f"{tag}_other_configs.json"), "w"), indent=1)

fl = {}
for c in (5, 3, 4):
    for k in (1, 2, 3, 4):
        b_ = last_json(os.path.join(src, f"bench_inflight_c{c}_k{k}.json"))
        if b_:
            fl.setdefault(b_["config"]["workload"], {})[f"in_flight_{k}"] = {
                "value": b_["value"], "ms_per_step": b_["ms_per_step"],
                "sequential_kernel_ms": (b_["roofline"]["in_flight"].get("sequential") or {}).get("kernel_ms"),
                "buffer_sets_identical": b_["roofline"]["in_flight"].get("buffer_sets_identical")}
if fl:
    json.dump({"command": "python bench.py --config C --in-flight K --steps 24 --warmup 3 --no-cpu-baseline --no-extras",
               "what": "cilqr_set_batches_in_flight(K): K launch slots inside ONE handle", "workloads": fl},
              open(os.path.join(dst, f"{tag}_in_flight.json"), "w"), indent=1)
for mode in ("off", "on"):
    p_ = os.path.join(src, f"gpu_tests_xnack_{mode}.log")
    if os.path.exists(p_):
        tail = open(p_, errors="replace").read().splitlines()[-25:]
        open(os.path.join(dst, f"{tag}_gpu_tests_xnack_{mode}.txt"), "w").write("\n".join(tail) + "\n")

b = last_json(os.path.join(src, "bench_pipelined.json"))
if b:
    pl = {"command": "python bench.py --config 2 --streams 4 --steps 40 --warmup 3 --no-cpu-baseline --no-extras",
          "sequential_value": b["value"], "pipelined": b["extra"]["pipelined"], "large_launches": {}}
    for c in (5, 3, 4):
        bc = last_json(os.path.join(src, f"bench_pipelined_c{c}.json"))
        if bc:
            pl["large_launches"][bc["config"]["workload"]] = {
                "command": f"python bench.py --config {c} --streams 3 --steps 24 --warmup 2 --no-cpu-baseline --no-extras",
                "one_batch_at_a_time": {"value": bc["value"], "ms_per_step": bc["ms_per_step"]},
                "three_batches_in_flight": bc["extra"]["pipelined"]}
    json.dump(pl, open(os.path.join(dst, f"{tag}_pipelined.json"), "w"), indent=1)

for c in (2, 3, 5, 4):
    for suffix in ("", "_single_2wps", "_single_1wps"):
        p = os.path.join(src, f"phase_config{c}{suffix}.json")
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_phase_config{c}{suffix}.json"))

# SQ wait / active counters of scripts/stall_counters.sh (per launch, averaged over launches)
for name in ("c5", "c4", "c3", "c2"):
    st = {}
    for d in sorted(glob.glob(os.path.join(src, f"stall_{name}_[0-9]"))):
        st.update({k: v for k, v in counters(os.path.basename(d)).items() if not k.endswith("_launches") and k != "kernel"})
    if st:
        json.dump(st, open(os.path.join(dst, f"{tag}_stall_{name}.json"), "w"), indent=1)

b = last_json(os.path.join(src, "bench_force_dist.json"))
if b:
    err = open(os.path.join(src, "bench_force_dist.err"), errors="replace").read().splitlines()[-15:]
    json.dump({"command": "CILQR_FORCE_DIST=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline",
               "what": "bench.py's multi-GPU branch on one rank: init_process_group('nccl') = RCCL, dist.barrier(), the SUM "
                       "and MAX all-reduces of the statistics, destroy_process_group",
               "bench_line": b, "stderr_tail": err}, open(os.path.join(dst, f"{tag}_force_dist.json"), "w"), indent=1)
# sliced solves of the grouped build by slice length (scripts/slice_probe.py): one JSON line per setting
sl = {}
for c in (3, 4, 5):
    p = os.path.join(src, f"slices_config{c}.jsonl")
    if os.path.exists(p) and os.path.getsize(p):
        sl[f"config{c}"] = [json.loads(l) for l in open(p) if l.startswith("{")]
if sl:
    json.dump({"what": "development library, CILQR_TUNE=group_slice / group_slice_long = iterations per slice (0: solves run whole); "
                       "one launch each with the block timeline on; parked = hand-overs (end of a slice + idle wavefronts at the tail); "
                       "unfinished_started_in_20_slices = trajectories started and not finished, in 20 equal slices of the launch",
               "by_workload": sl}, open(os.path.join(dst, f"{tag}_sliced_solves.json"), "w"), indent=1)

for c in (3, 4, 5):
    for suffix in ("", "_single"):
        p = os.path.join(src, f"timeline_config{c}{suffix}.json")
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_timeline_config{c}{suffix}.json"))
    # the raw per-trajectory records (start, end, block, XCC: what scripts/schedule_sim.py replays) stay where the collection
    # put them, gpurun_out/TAG/timeline_cN.npy — scratch, not tracked (round 6: profiles/ holds summaries only)
b = last_json(os.path.join(src, "two_rank.json"))
if b:
    json.dump({"command": "CILQR_BENCH_ONE_DEVICE=1 CILQR_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 "
                          "bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline",
               "what": "REHEARSAL of bench.py's N > 1 code path with two processes on the one-GPU box (both ranks on GPU 0, "
                       "statistics reduced over gloo): it shows the line a multi-GPU run prints — rank identities, per-rank "
                       "kernel times, guarded extras — not a measurement", "bench_line": b},
              open(os.path.join(dst, f"{tag}_two_rank_rehearsal.json"), "w"), indent=1)
if bench:
    e = bench.get("extra", {})
    if e.get("closed_loop"):
        json.dump({"command": "python bench.py (default command, extra.closed_loop)", **e["closed_loop"]},
                  open(os.path.join(dst, f"{tag}_closed_loop.json"), "w"), indent=1)
    if e.get("config5_alm"):
        json.dump({"command": "python bench.py (default command, extra.config5_alm)", **e["config5_alm"]},
                  open(os.path.join(dst, f"{tag}_alm.json"), "w"), indent=1)
b = last_json(os.path.join(src, "bench_config1.json"))
if b:
    json.dump(b, open(os.path.join(dst, f"{tag}_config1.json"), "w"), indent=1)

print(json.dumps({"bench_value": bench and bench["value"], "kernel_ms": bench and bench["roofline"]["kernel_ms"],
                  "pmc": {k: {"traffic": v["hbm_bytes_per_launch_corrected"], "ratio": v["traffic_over_algorithmic"],
                              "valu": v.get("SQ_INSTS_VALU")} for k, v in pmc_all.items()},
                  "other": {k: v["it_per_s"] for k, v in other.items() if k != "note"}}, indent=1))
