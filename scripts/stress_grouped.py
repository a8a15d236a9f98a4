#!/usr/bin/env python3
"""Runs ON THE GPU BOX: hammer the grouped build (k_solve_grp: two or three trajectories per wavefront, one rollout pass for
all, trajectories handed over between wavefronts at the launch's tail, sliced solves).  Random horizons up to 63 — and, round 5,
64 ... 127 (the long layout: both expansions and the gains streamed, config 4's scenario mix) —, batch sizes from a handful
to several rounds of the resident wavefronts, scenario / parameter mixes, iteration budgets, rollout policies, warm starts;
every launch is solved with one trajectory per wavefront first (k_solve) and then with 2 and 3 per wavefront, with and
without the tail hand-over (development library: CILQR_TUNE is read per handle), and compared bit for bit — u, x, every
result field, the decision trace.
usage: scripts/stress_grouped.py [seconds] [out.json]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-example-of-ilqr_amd")


def same(a, b):
    """equal as IEEE values, field by field (NaN == NaN positionally: a diverged solve's NaN cost may carry another payload —
    which operand of a commutative add the compiler put first decides whose NaN propagates)"""
    if a.dtype.names:
        return all(same(a[f], b[f]) for f in a.dtype.names)
    if a.dtype.kind == "f":
        return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def engine(wl, tune):
    if tune and os.environ.get("STRESS_POISON"):  # every launch on scratch pre-filled with NaN patterns (1) or zeros (2)
        tune = tune + ",poison=" + os.environ["STRESS_POISON"]
    if tune:
        os.environ["CILQR_TUNE"] = tune
    else:
        os.environ.pop("CILQR_TUNE", None)
    return pkg.BatchedCILQR(wl.params, wl.scenes, dev=True)  # (the tuning switches live in the development library)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(20260929)
    t_end = time.time() + budget
    launches = handed_over = waiting = 0
    shapes = []
    while time.time() < t_end:
        N = int(rng.choice([5, 12, 30, 41, 50, 50, 50, 63, 64, 77, 100, 100, 127, 128, 160, 200]))  # round 6: + four rows per lane
        B = int(rng.choice([1, 2, 3, 7, 64, 333, 1024, 2049, 4100, 6200, 9000, 17000]))
        if os.environ.get("STRESS_MAX_B"):  # (rehearsals on the CPU emulator: scripts/emu_rehearse.py)
            B = min(B, int(os.environ["STRESS_MAX_B"]))
        kind = int(rng.integers(0, 3))
        # round 6: a third of the launches under the augmented Lagrangian — lone wavefronts (k_solve) against pairs (opt-in)
        alm = N <= 127 and rng.integers(0, 3) == 0
        first = int(rng.integers(0, 50000))
        if N > 63:
            kind = 3
            B = min(B, 9000)
            wl = pkg.workloads.config4(B=B, N=N, first=first)
            if any(sc.obs.shape[1] < N + 1 for sc in wl.scenes):
                continue
        elif kind == 0:
            wl = pkg.workloads.config3(B=B, N=N, first=first)
        elif kind == 1:
            wl = pkg.workloads.config5(B_base=max(1, B // 16), N=N, first=first)
        else:
            wl = pkg.workloads.config2(B=B, N=N, first=first)
        over = {"solve_type": 1} if alm else {}
        if N > 127:
            over["max_iter"] = int(rng.choice([5, 17, 40]))
        if rng.integers(0, 3) == 0:
            over["max_iter"] = int(rng.choice([0, 1, 2, 5, 17, 40]))
        if rng.integers(0, 4) == 0:
            over["init_lamb"] = float(rng.choice([0.0, 2.0, 64.0]))
        if over:
            wl = pkg.workloads.Workload(wl.name, [pkg.copy_params(q, **over) for q in wl.params], wl.scenes, wl.x0, wl.scenario_id,
                                        wl.param_id, wl.tick)
        warm = None
        if rng.integers(0, 3) == 0:  # a warm start from some plausible controls (cs:163-180)
            warm = np.cumsum(rng.normal(0, 0.05, (wl.B, wl.N, 2)), axis=1) * np.array([1.0, 0.05])
        rollout = int(rng.choice([-1, -1, 0, 1]))
        ref = None
        slice_a, slice_b = int(rng.choice([1, 2, 3, 5, 8])), int(rng.choice([0, 13, 40]))
        tunes = ("group=0", "group=2", "group=2,group_steal=0", "group=2,group_pair_costs=0", "group=2,pair_sweep=0",
                 f"group=2,group_slice={slice_a},group_slice_long={slice_a},group_slice_window=300",
                 f"group=2,group_slice={slice_b},group_slice_long={slice_b}")
        if alm:
            tunes = ("group=0", "group=2")       # (no hand-overs, no slices under ALM: lone wavefronts vs pairs)
        elif N > 127:
            tunes = tunes[1:2] + tunes[2:3] + tunes[5:]   # (no lone build at these horizons: pairs against pairs without hand-overs / other slices)
        for tune in tunes:
            eng = engine(wl, tune)
            eng.set_helper_mode(0)
            eng.set_rollout_mode(rollout)
            out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, last_u=warm, trace_cap=48)
            info = eng.last_launch_info()
            want = {"group=0": 1}.get(tune, 2)
            if tune == "group=0" and N > 127:
                want = 2
            assert info["trajectories_per_wavefront"] == want, (tune, info)
            st = eng.work_sharing_stats()
            assert st["error"] == 0, (wl.name, tune, st)
            if want > 1 and "group_steal=0" not in tune:
                handed_over += eng.resume_stats()
                waiting += st["helpers"]
            eng.close()
            if ref is None:
                ref = out
            else:
                what = (wl.name, N, B, over, rollout, tune)
                if not same(ref["u"], out["u"]) and os.environ.get("STRESS_DIAG"):
                    bad = np.nonzero((ref["u"] != out["u"]).reshape(wl.B, -1).any(1))[0]
                    print("MISMATCH", what, "warm" if warm is not None else "cold", "trajectories", len(bad), bad[:12])
                    for b0 in bad[:3]:
                        tb = np.argwhere(ref["trace"][b0] != out["trace"][b0])
                        k0 = int(tb[0][0]) if len(tb) else -1
                        print("  b", int(b0), "iters", ref["res"]["iters"][b0], out["res"]["iters"][b0], "first trace diff at", k0,
                              ref["trace"][b0, k0] if k0 >= 0 else None, out["trace"][b0, k0] if k0 >= 0 else None)
                    continue
                assert same(ref["u"], out["u"]), what
                assert same(ref["x"], out["x"]), what
                assert same(ref["res"], out["res"]), what
                if not same(ref["trace"], out["trace"]):
                    bad = np.argwhere(ref["trace"] != out["trace"])
                    b0, k0 = bad[0]
                    raise AssertionError((what, "first", first, "traces differ at", int(b0), int(k0), ref["trace"][b0, k0], out["trace"][b0, k0],
                                          "n_bad", len(bad), "res", ref["res"][b0]))
            launches += 1
        shapes.append([N, B, kind])
    rep = {"launches": launches, "shapes": len(shapes), "trajectories_handed_over_at_the_tail": handed_over,
           "wavefronts_that_waited_for_work": waiting, "mismatches": 0}
    print(rep)
    if len(sys.argv) > 2:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
        json.dump(rep, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
