#!/usr/bin/env python3
"""Turn gpurun_out/TAG (written by scripts/collect_profiles.sh on the GPU box) into the small, tracked
summaries under profiles/.   usage: scripts/summarize_profiles.py TAG [--current]

--current also rewrites profiles/r01_pmc_config2.json, the file bench.py reads `roofline.traffic` from.
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


def counters(pattern):
    """per-launch value of every counter in the pass (summed over the dispatch's rows, averaged over launches)"""
    out = {}
    for path in glob.glob(os.path.join(src, pattern, "*", "*_counter_collection.csv")):
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(path)):
            if "k_solve" in r["Kernel_Name"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, d in per.items():
            out[name] = sum(d.values()) / len(d)
            out[name + "_launches"] = len(d)
    return out


bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, f"{tag}_config2_bench.json"), "w"), indent=1)

stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_config2_kernel_stats.csv"))

f = counters("pmc_FETCH_SIZE")
w = counters("pmc_WRITE_SIZE")
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
raw = (f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0
cor = (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0
pmc = {
    "workload": bench["config"]["workload"],
    "command": "rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline (scripts/collect_profiles.sh: one small counter group per pass, no other trace domain)",
    "kernel": "k_solve<false, 1, false, true, false, 1> (main + helper wavefront per trajectory)",
    "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request "
                 "for wide coalesced streams (MI355X_MICROARCH.md HBM section), so the read side is doubled as an upper "
                 "bound; this kernel's reads are mostly 8-byte gathers, for which the factor is uncalibrated",
    "FETCH_SIZE_KiB_per_launch": f["FETCH_SIZE"],
    "WRITE_SIZE_KiB_per_launch": w["WRITE_SIZE"],
    "hbm_bytes_per_launch_raw": raw,
    "hbm_bytes_per_launch_corrected": cor,
    "algorithmic_bytes_per_launch": alg,
    "note": "the excess over the algorithmic bytes is the scratch slab of the 20 trial trajectories (49 KB written per "
            "trajectory-iteration, mostly never read back)",
}
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_config2.json"), "w"), indent=1)
if "--current" in sys.argv:
    json.dump(pmc, open(os.path.join(dst, "r01_pmc_config2.json"), "w"), indent=1)

sq = {}
for pat in glob.glob(os.path.join(src, "pmc_SQ*")):
    if os.path.isdir(pat):
        sq.update({k: v for k, v in counters(os.path.basename(pat)).items() if not k.endswith("_launches")})
sq["note"] = ("per launch of k_solve on config 2 (1024 blocks x 2 wavefronts); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU "
              "count quad-cycles")
json.dump(sq, open(os.path.join(dst, f"{tag}_pmc_sq_config2.json"), "w"), indent=1)

other = {"note": "python bench.py --config C [--batch B] --steps 3 --warmup 1 --no-cpu-baseline on one MI355X; "
                 "parity-test configurations of BASELINE.json and a batch sweep of config 2, not the benchmark line"}
for path in sorted(glob.glob(os.path.join(src, "bench_config*.json"))):
    txt = open(path).read().strip()
    if not txt:
        continue
    b = json.loads(txt.splitlines()[-1])
    e = b["extra"]
    other[b["config"]["workload"]] = {
        "it_per_s": round(b["value"]), "ms_per_step": round(b["ms_per_step"], 3), "solves_per_s": round(e["solves_per_s"]),
        "iters_per_solve": round(e["iterations_per_solve_mean"], 2), "converged": e["converged"],
        "max_lamb": e["max_lamb"], "max_iter": e["max_iter"], "hbm_frac": round(b["roofline"]["frac"], 5)}
json.dump(other, open(os.path.join(dst, f"{tag}_other_configs.json"), "w"), indent=1)

pp = os.path.join(src, "bench_pipelined.json")
if os.path.exists(pp) and os.path.getsize(pp):
    b = json.loads(open(pp).read().strip().splitlines()[-1])
    json.dump({"command": "python bench.py --streams 4 --steps 40 --warmup 3 --no-cpu-baseline",
               "sequential_value": b["value"], "pipelined": b["extra"]["pipelined"]},
              open(os.path.join(dst, f"{tag}_pipelined.json"), "w"), indent=1)

for c in (2, 3):
    p = os.path.join(src, f"phase_config{c}.json")
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_phase_config{c}.json"))
print(json.dumps({"bench_value": bench["value"], "kernel_ms": bench["roofline"]["kernel_ms"], "traffic": cor, "sq": sq,
                  "other": {k: v["it_per_s"] for k, v in other.items() if k != "note"}}, indent=1))
