#!/usr/bin/env python3
"""Round-4 experiment (CPU only, test infrastructure: imports oracle/): how far is the FUSED flavour of the arithmetic —
explicit fma at four named groups of sites, oracle/cilqr_oracle.c ORC_FUSED — from the oracle's glibc-libm build, the
stand-in for what the reference binary links, next to the shipped (unfused, detmath) flavour?  Per workload: share of
trajectories whose u, x and J_final lie within 1e-5 of the libm build's.

    python scripts/fused_tolerance.py [--rows 1024] [--out profiles/r04_experiments/fused_flavour_tolerance.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cilqr_amd as pkg  # noqa: E402  (workload generators only: no GPU is touched)
from libm_tolerance import oracle_scenes, outside_band  # noqa: E402
from oracle import Oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1024)
ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
ap.add_argument("--out", default=None)
a = ap.parse_args()
W = pkg.workloads
cases = {"config2": W.config2(), "config3": W.config3(B=a.rows), "config5": W.config5(B_base=max(1, a.rows // 16)),
         "configs[3] rank-0 shard (N = 100)": W.config4(B=min(a.rows, 512), N=100)}
alm = W.config2()
cases["config2, augmented Lagrangian"] = W.Workload(alm.name + "_alm", [pkg.copy_params(q, solve_type=1) for q in alm.params],
                                                     alm.scenes, alm.x0, alm.scenario_id, alm.param_id, alm.tick)
rep = {"what": __doc__.strip().splitlines()[0], "rows_per_workload": a.rows, "workloads": {}}
for name, wl in cases.items():
    nb = min(a.rows, wl.B)
    args = (wl.params, oracle_scenes(wl), wl.x0[:nb], wl.scenario_id[:nb], wl.param_id[:nb], wl.tick[:nb])
    sol = {m: Oracle(m).solve_batch(*args, n_threads=a.threads) for m in ("libm", "det", "fused")}
    ent = {"trajectories": int(nb)}
    for m in ("det", "fused"):
        du, dx, dJ, bad = outside_band(sol[m], sol["libm"])
        gap = np.maximum(np.maximum(du, dx), dJ)
        ent[m + "_vs_libm"] = {"within_1e-5_frac": float(1.0 - bad.mean()), "outside": int(bad.sum()),
                               "median_gap": float(np.nanmedian(gap)), "max_gap_inside_band": float(np.nanmax(gap[~bad])) if (~bad).any() else None,
                               "iterations_equal_frac": float((sol[m]["res"]["iters"] == sol["libm"]["res"]["iters"]).mean())}
    du, dx, dJ, bad = outside_band(sol["fused"], sol["det"])
    ent["fused_vs_det"] = {"within_1e-5_frac": float(1.0 - bad.mean()), "outside": int(bad.sum()),
                           "bit_identical_frac": float(np.mean([np.array_equal(sol["fused"]["x"][b], sol["det"]["x"][b]) for b in range(nb)]))}
    rep["workloads"][name] = ent
    print(name, json.dumps(ent), flush=True)
if a.out:
    json.dump(rep, open(a.out, "w"), indent=1)
