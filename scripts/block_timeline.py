#!/usr/bin/env python3
"""Runs ON THE GPU BOX: how a launch fills the chip over time.  Records start / end of every block (100 MHz clock),
its index and its XCC, and prints the mean number of resident blocks, per XCC: blocks, busy time, finishing time.
usage: scripts/block_timeline.py [config] [batch]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-example-of-ilqr_amd")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
wl = {2: pkg.workloads.config2, 3: pkg.workloads.config3, 5: pkg.workloads.config5}.get(cfg)
wl = wl() if wl else pkg.workloads.config4(B=int(sys.argv[2]) if len(sys.argv) > 2 else 8192, N=100)
eng = pkg.BatchedCILQR(wl.params, wl.scenes)
if os.environ.get("GROUP_MODE"):
    eng.set_group_mode(int(os.environ["GROUP_MODE"]))
eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
eng.set_block_timeline(True)
eng.set_timing(True)
r = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
kms = eng.last_kernel_ms()
tl = eng.block_timeline(wl.B)
eng_parked = eng.resume_stats()
eng.close()
t0 = tl[:, 0].min()
st, en = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0  # microseconds
span = en.max()
sliced = bool((tl[:, 2] < 0).any())   # resumable solves: [2] = minus the busy time (ticks) instead of the block index
busy = np.where(tl[:, 2] < 0, -tl[:, 2] / 100.0, en - st) if sliced else (en - st)
out = {"workload": wl.name, "kernel_ms": kms, "span_ms_by_block_clock": span / 1e3,
       "resumable_solves": sliced, "parked": eng_parked,
       "mean_resident_blocks": float(busy.sum() / span),
       "block_ms": {"mean": float(busy.mean() / 1e3), "p50": float(np.median(busy) / 1e3), "max": float(busy.max() / 1e3)},
       "first_start_to_last_end_ms": {"mean": float((en - st).mean() / 1e3), "max": float((en - st).max() / 1e3)}}
if sliced:  # (per-XCC placement and restart gaps describe one block per trajectory: not defined here)
    if os.environ.get("TIMELINE_OUT"):
        np.save(os.environ["TIMELINE_OUT"], tl)
    print(json.dumps(out, indent=1))
    sys.exit(0)
per = []
for x in sorted(set(tl[:, 3].tolist())):
    m = tl[:, 3] == x
    per.append({"xcc": int(x), "blocks": int(m.sum()), "busy_block_ms": round(float((en[m] - st[m]).sum() / 1e3), 1),
                "last_start_ms": round(float(st[m].max() / 1e3), 2), "finish_ms": round(float(en[m].max() / 1e3), 2),
                "block_index_mod_8": sorted(set((tl[m, 2] % 8).tolist()))})
out["per_xcc"] = per
# resident blocks over time, in 20 slices
edges = np.linspace(0, span, 21)
occ = []
for a, b in zip(edges[:-1], edges[1:]):
    occ.append(round(float((np.clip(en, a, b) - np.clip(st, a, b)).sum() / (b - a))))
out["resident_blocks_in_20_time_slices"] = occ
# per XCC: how long after a block ends does the next one start?  (the k-th block to end frees the slot the
# (k + capacity)-th block to start takes; capacity = the largest number of blocks seen resident on the XCC)
gaps = []
for x in sorted(set(tl[:, 3].tolist())):
    m = tl[:, 3] == x
    s_sorted, e_sorted = np.sort(st[m]), np.sort(en[m])
    ev = np.concatenate([np.stack([s_sorted, np.ones_like(s_sorted)], 1), np.stack([e_sorted, -np.ones_like(e_sorted)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    cap = int(np.cumsum(ev[:, 1]).max())
    n = m.sum() - cap
    if n > 0:
        gaps.append(s_sorted[cap:cap + n] - e_sorted[:n])
    out.setdefault("capacity_per_xcc", []).append(cap)
if gaps:
    g = np.concatenate(gaps)
    out["restart_gap_us"] = {"mean": float(g.mean()), "p50": float(np.median(g)), "p90": float(np.percentile(g, 90)), "max": float(g.max())}
if os.environ.get("TIMELINE_OUT"):
    np.save(os.environ["TIMELINE_OUT"], tl)
print(json.dumps(out, indent=1))
