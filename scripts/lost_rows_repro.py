#!/usr/bin/env python3
"""Round 5, time-boxed: the REAL instruction stream that lost store data in round 4 (profiles/r04_experiments/
tiled_slab_lost_rows.txt), run again.  toy-example-of-ilqr_amd/libcilqr_amd_lostrows.so = the library built with -DCILQR_LOSTROWS_REPRO: the grouped rollout pass
with its slab stores inside waterfall loops (scripts/probes/lost_rows_failing_pass.s.gz is that function's code).  What the anomaly
changes are line-search costs, hence sometimes trajectories: the check is pairing invariance — lone wavefronts (k_solve, whose
rollout has no such loop) against pairs per wavefront, same library, same inputs.

   CILQR_AMD_LIB=toy-example-of-ilqr_amd/libcilqr_amd_lostrows.so [HSA_XNACK=1] python scripts/lost_rows_repro.py [launches]      (on the GPU box)
prints one JSON line: launches, launches with a mismatch, mismatching trajectories per launch."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cilqr_amd as pkg  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 12
wl = pkg.workloads.config3(B=4100)
eng = pkg.BatchedCILQR(wl.params, wl.scenes)
eng.set_helper_mode(0)
eng.set_group_mode(0)
lone = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
eng.set_group_mode(2)
bad = []
for rep in range(launches):
    r = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    neq = ~((r["x"].reshape(wl.B, -1).view(np.uint64) == lone["x"].reshape(wl.B, -1).view(np.uint64)).all(axis=1)
            & (r["res"]["J_final"].view(np.uint64) == lone["res"]["J_final"].view(np.uint64))
            & (r["res"]["ls_trials"] == lone["res"]["ls_trials"]))
    bad.append(int(neq.sum()))
eng.close()
print(json.dumps({"library": os.environ.get("CILQR_AMD_LIB", "shipped"), "HSA_XNACK": os.environ.get("HSA_XNACK"),
                  "workload": wl.name, "launches": launches, "launches_with_mismatch": int(sum(b > 0 for b in bad)),
                  "mismatching_trajectories_per_launch": bad}))
