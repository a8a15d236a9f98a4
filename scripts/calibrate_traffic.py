#!/usr/bin/env python3
"""Condenses the counter passes of scripts/calibrate_traffic.sh: per calibration kernel and working-set size, the
counter value per launch next to the bytes the kernel is known to move.  usage: calibrate_traffic.py <dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
known = [json.loads(l) for l in open(os.path.join(src, "known_bytes.jsonl")) if l.startswith("{")]
KERNELS = ["cal_read16", "cal_read8", "cal_read8_s160", "cal_write8", "cal_write8_slab"]


def per_dispatch(pattern):
    out = defaultdict(dict)  # counter -> dispatch id -> (kernel, value)
    for path in glob.glob(os.path.join(src, pattern, "*", "*_counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(float))
        names = {}
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if not k.startswith("cal_"):
                continue
            d = int(r["Dispatch_Id"])
            acc[r["Counter_Name"]][d] += float(r["Counter_Value"])
            names[d] = k
        for cn, dd in acc.items():
            for d, v in dd.items():
                out[cn][d] = (names[d], v)
    return out


counters = {}
for d in glob.glob(os.path.join(src, "pmc_*")):
    if os.path.isdir(d):
        counters.update(per_dispatch(os.path.basename(d)))

report = {"what": "known-byte kernels in the solver's access widths through rocprofv3 --pmc (one counter group per pass); "
                  "FETCH_SIZE / WRITE_SIZE are reported in KiB; each kernel makes 8 passes over its working set, so 7/8 of "
                  "the traffic of the 64 MiB set can be served by the 256 MiB Infinity Cache, none of the 1 GiB set's",
          "sets": []}
for si, kb in enumerate(known):
    ent = {"working_set_bytes": kb["set_bytes"], "passes": kb["passes"], "kernels": {}}
    for ki, k in enumerate(KERNELS):
        rec = {"useful_bytes_per_launch": kb["useful_bytes"][k]}
        for cn, dd in counters.items():
            # dispatches come in launch order: per set 2 repeats x 5 kernels; take the second repeat (warm)
            ids = sorted(d for d, (kn, _) in dd.items() if kn == k)
            if len(ids) >= 2 * len(known):
                pick = ids[2 * si + 1]
                rec[cn] = dd[pick][1]
        if "FETCH_SIZE" in rec:
            rec["FETCH_SIZE_bytes_over_useful"] = rec["FETCH_SIZE"] * 1024.0 / rec["useful_bytes_per_launch"]
        if "WRITE_SIZE" in rec:
            rec["WRITE_SIZE_bytes_over_useful"] = rec["WRITE_SIZE"] * 1024.0 / rec["useful_bytes_per_launch"]
        ent["kernels"][k] = rec
    report["sets"].append(ent)
print(json.dumps(report, indent=1))
