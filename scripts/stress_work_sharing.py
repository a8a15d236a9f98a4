#!/usr/bin/env python3
"""Runs ON THE GPU BOX: hammer the work sharing between blocks and (round 3) the resumable solves.  Random horizons /
batch sizes / scenario mixes, every launch repeated with sharing on and off and with random slice lengths of the
resumable solves (same handle) and compared bit for bit with the first, unsliced and unshared one; counters checked.
usage: scripts/stress_work_sharing.py [seconds]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-example-of-ilqr_amd")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(20260928)
    t_end = time.time() + budget
    launches = announced = helped = parked = sliced_launches = 0
    while time.time() < t_end:
        N = int(rng.choice([64, 70, 76, 88, 100, 112, 127]))
        mixed = bool(rng.integers(0, 2))
        B = int(rng.choice([520, 700, 1100, 1600, 2300, 4100, 6200]))
        if mixed:
            wl = pkg.workloads.config4(B=B, N=min(N, 100), first=int(rng.integers(0, 50000)))
        else:
            wl = pkg.workloads.config3(B=B, first=int(rng.integers(0, 50000)))
            wl = pkg.workloads.Workload(wl.name, [pkg.copy_params(q, N=N, max_iter=int(rng.integers(5, 60))) for q in wl.params],
                                        wl.scenes, wl.x0, wl.scenario_id, wl.param_id, wl.tick)
        alm = rng.integers(0, 4) == 0  # every fourth launch with the augmented-Lagrangian solve type
        if alm:
            wl = pkg.workloads.Workload(wl.name + "_alm", [pkg.copy_params(q, solve_type=1) for q in wl.params], wl.scenes, wl.x0,
                                        wl.scenario_id, wl.param_id, wl.tick)
        eng = pkg.BatchedCILQR(wl.params, wl.scenes)
        if alm:
            eng.set_helper_mode(0)  # lone wavefronts whatever the batch: the builds that share work
        ref = None
        eng.set_resume_iters(0)
        eng.set_work_sharing(0)
        for rep in range(3):
            for mode in (0, 1, 0, 1) if rep == 0 else (1, 0, 1):
                if ref is not None:
                    eng.set_work_sharing(mode)
                    eng.set_resume_iters(int(rng.choice([0, 1, 3, 7, 16, 32, 90])))
                out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
                if mode == 1 and ref is not None:
                    st = eng.work_sharing_stats()
                    assert st["error"] == 0, (wl.name, st)
                    announced += st["announced"]
                    helped += st["helped"]
                if ref is not None:
                    n = eng.resume_stats()
                    parked += n
                    sliced_launches += int(n > 0)
                if ref is None:
                    ref = out
                else:
                    assert np.array_equal(ref["u"].view(np.uint64), out["u"].view(np.uint64)), (wl.name, mode, rep)
                    assert np.array_equal(ref["x"].view(np.uint64), out["x"].view(np.uint64)), (wl.name, mode, rep)
                    assert (ref["res"] == out["res"]).all(), (wl.name, mode, rep)
                launches += 1
        eng.close()
    print({"launches": launches, "searches_announced": announced, "trial_costs_delivered": helped,
           "launches_with_parked_solves": sliced_launches, "solves_parked": parked, "mismatches": 0})


if __name__ == "__main__":
    main()
