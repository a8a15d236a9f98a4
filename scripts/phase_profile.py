#!/usr/bin/env python3
"""Phase breakdown of the fused solve kernel (in-kernel cycle counters), for DESIGN.md/profiles."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cilqr_amd as pkg

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--horizon", type=int, default=50)
ap.add_argument("--group", type=int, default=0, help="2: the grouped build (two trajectories per wavefront), which runs two wavefronts per SIMD")
args = ap.parse_args()
W = pkg.workloads
if args.config == 2:
    wl = W.config2(B=args.batch or 1024, N=args.horizon)
elif args.config == 3:
    wl = W.config3(B=args.batch or 8192, N=args.horizon)
elif args.config == 4:
    wl = W.config4(B=args.batch or 8192, N=100)
else:
    wl = W.config5(B_base=args.batch or 4096, N=args.horizon)
eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True)  # the cycle accounting lives in the development library
if args.group:
    eng.set_group_mode(args.group)
    eng.set_helper_mode(0)
ids = dict(scenario_id=wl.scenario_id, param_id=wl.param_id, tick=wl.tick)
eng.solve_batch(wl.x0, **ids)
eng.set_phase_profiling(True)
eng.set_timing(True)
out = eng.solve_batch(wl.x0, **ids)
ms = eng.last_kernel_ms()
cyc = eng.phase_cycles(wl.B)
names = ["init", "derivs", "backward", "rollout", "trial_cost", "accept", "total", "iters"]
res = out["res"]
tot = cyc[:, 6].astype(float)
rep = {"workload": wl.name, "kernel_ms": ms, "iters_sum": int(res["iters"].sum()), "iters_max": int(res["iters"].max()),
       "iters_mean": float(res["iters"].mean()), "trials_sum": int(res["ls_trials"].sum()),
       "cycles_total_max": float(tot.max()), "cycles_total_mean": float(tot.mean()),
       "clock_GHz_est": float(tot.max() / (ms * 1e-3) / 1e9),
       "share_of_mean_total": {n: float(cyc[:, i].mean() / tot.mean()) for i, n in enumerate(names[:6])},
       "cycles_per_iteration": {n: float(cyc[:, i].sum() / res["iters"].sum()) for i, n in enumerate(names[1:6], start=1)},
       "cycles_per_trial_cost": float(cyc[:, 4].sum() / max(1, res["ls_trials"].sum())),
       "trial_cost_split_cycles_per_eval": {n: float(cyc[:, i].sum() / max(1, cyc[:, 9].sum())) for n, i in (("ref_points", 10), ("stage_costs", 11), ("ordered_sum", 12))},
       "rollout_passes": {"first_trial_alone": int(cyc[:, 14].sum()), "all_20_at_once": int(cyc[:, 15].sum()),
                          "second_pass_after_rejected_first_trial": int(cyc[:, 16].sum()),
                          "slab_written_frac_of_line_searches": float((cyc[:, 15].sum() + cyc[:, 16].sum())
                                                                      / max(1, cyc[:, 14].sum() + cyc[:, 15].sum()))},
       "ref_scan_fallbacks": int(cyc[:, 8].sum()), "ref_sampled_proofs": int(cyc[:, 13].sum()), "trial_cost_evals": int(cyc[:, 9].sum()),
       "slowest": {"iters": int(res["iters"][tot.argmax()]), "trials": int(res["ls_trials"][tot.argmax()]),
                   "phases": {n: int(cyc[tot.argmax(), i]) for i, n in enumerate(names[:6])}}}
order = np.argsort(-tot)[:8]
rep["slowest_8"] = [{"b": int(b), "iters": int(res["iters"][b]), "trials": int(res["ls_trials"][b]), "cycles": int(tot[b]),
                     "trial_cost_cycles": int(cyc[b, 4]), "ref_points_cycles": int(cyc[b, 10]),
                     "ref_scan_fallbacks": int(cyc[b, 8])} for b in order]
rep["ref_scan_fallbacks_max_per_trajectory"] = int(cyc[:, 8].max())
if args.group:  # the grouped build books its own bookkeeping in slots 10-12
    rep["grouped_extra_cycles_per_iteration"] = {n: float(cyc[:, i].sum() / res["iters"].sum()) for n, i in
                                                 (("segment_setup", 10), ("back_in_solve", 11), ("state_store", 12))}
    passes = cyc[:, 14].sum() + cyc[:, 15].sum() + cyc[:, 16].sum()
    rep["rollout_steps_in_the_small_angle_form_frac"] = float(cyc[:, 13].sum() / max(1, passes * wl.N))
if os.environ.get("PHASE_OUT"):
    np.save(os.environ["PHASE_OUT"], cyc)
print(json.dumps(rep, indent=1))
