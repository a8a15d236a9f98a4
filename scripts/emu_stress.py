#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the hand-over protocol of the grouped kernel (tickets, claims, pushes, sliced solves, bounded waits) run on
the wave64 emulator (tests/emu/) under ADVERSARIAL scheduling — resident blocks visited in random order, stalled or given bursts
of turns (CILQR_EMU_SCHED_SEED) — over random launch shapes: batch size, horizon, resident blocks, slice length.  The same REAL
device code the GPU runs, in interleavings a GPU stress run only samples by luck.  Every trajectory must come back == oracle,
no bounded wait may expire, nothing may stay CILQR_END_NOT_SOLVED.

    python scripts/emu_stress.py [--cases 40] [--seed 1] [--lib tests/emu/_build/libcilqr_emu_dev.so] [--long]
prints one JSON line per case and a summary; exit 1 on any mismatch."""
import argparse
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASE = r"""
import json, os, sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import cilqr_amd as pkg
from oracle import Oracle, Scene
c = json.loads(sys.argv[2])
cfg = pkg.GlobalConfig.get_instance(c["scenario"]); sc = pkg.build_scenario(cfg, c["scenario"])
p = pkg.params_from_config(cfg, N=c["N"], solve_type=c.get("solve_type", 0), max_iter=c.get("max_iter", 100))
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True); eng.set_group_mode(2)
x0 = pkg.workloads.perturbed_starts(sc.ego_state, c["B"], c["x0_seed"])
ok_wait = True
scene = lambda t=0: Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, t)
if c.get("kind") == "loop":
    # the closed planning loop in one launch (egos change wavefronts between ticks and inside solves), against stateful oracles
    p = pkg.params_from_config(cfg, N=c["N"], use_last_solution=1, max_iter=c.get("max_iter", 100))
    eng.close(); eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True); eng.set_group_mode(2)
    B, N, T = c["B"], c["N"], c["ticks"]
    ptr = lambda a: a.ctypes.data
    dx0 = x0.copy(); tick = np.zeros(B, dtype=np.int32)
    u = np.zeros((B, N, 2)); x = np.zeros((B, N + 1, 4)); res = np.zeros(B, dtype=pkg.RESULT_DTYPE)
    states = np.zeros((B, T, 4)); its = np.zeros((T, B), dtype=np.int32)
    r = dict(c)
    try:
        eng.closed_loop_batch_device(B, T, ptr(dx0), 0, 0, ptr(tick), 0, ptr(u), ptr(x), ptr(res), ptr(states), ptr(its), 0)
        eng.wait()
        bad = []
        o = Oracle("det")
        for b in range(B):
            s = o.solver(p); s.reset(); xe = x0[b].copy()
            for t in range(T):
                rr = s.solve(xe, scene(t)); xe = rr["x"][1].copy()
                if not (np.array_equal(xe, states[b, t]) and int(rr["res"]["iters"]) == int(its[t, b])): bad.append(b); break
        r.update(ok=not bad and (tick == T).all(), mismatching=bad[:8], not_solved=int((res["end_reason"] == 4).sum()), parked=eng.resume_stats(),
                 launch_error=eng.work_sharing_stats()["error"], blocks=eng.last_launch_info()["blocks"], iters_max=int(its.max()))
        r["ok"] = bool(r["ok"]) and r["not_solved"] == 0 and r["launch_error"] == 0
    except RuntimeError as e:
        r.update(ok=False, error=str(e)[:200])
    print("CASE " + json.dumps(r)); sys.exit(0)
try:
    out = eng.solve_batch(x0)
except RuntimeError as e:
    ok_wait = False; out = None; err = str(e)
ref = Oracle("det").solve_batch(p, scene(), x0, n_threads=2)
r = dict(c)
if out is None:
    r.update(ok=False, error=err[:200])
else:
    bad = [int(b) for b in range(c["B"]) if not (np.array_equal(out["u"][b], ref["u"][b]) and np.array_equal(out["x"][b], ref["x"][b]) and out["res"]["iters"][b] == ref["res"]["iters"][b])]
    r.update(ok=not bad, mismatching=bad[:8], not_solved=int((out["res"]["end_reason"] == 4).sum()), parked=eng.resume_stats(),
             launch_error=eng.work_sharing_stats()["error"], blocks=eng.last_launch_info()["blocks"], iters_max=int(out["res"]["iters"].max()))
    r["ok"] = r["ok"] and r["not_solved"] == 0 and r["launch_error"] == 0
try:
    import ctypes
    pc = (ctypes.c_longlong * 16)(); ctypes.CDLL(str(pkg._lib.LIB_PATH)).cilqr_emu_probe_counts(pc)
    r["waits_with_a_place"], r["places_kept"], r["idle_with_two_places"] = int(pc[0]), int(pc[1]), int(pc[2])   # (tests/emu/build_emu.py PROBES)
except (OSError, AttributeError):
    pass
print("CASE " + json.dumps(r))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--lib", default=os.path.join(ROOT, "tests", "emu", "_build", "libcilqr_emu_dev.so"))
    ap.add_argument("--long", action="store_true", help="horizons above 63 too (the long layout: slower)")
    ap.add_argument("--focus", default="", help="'claims': tiny launches that maximise racing claims on the queue of parked trajectories (two or "
                    "three blocks, slices of one or two iterations, every end of slice a hand-over); 'places': shapes in which slots end up "
                    "holding a place in the queue (with --preempt)")
    ap.add_argument("--preempt", type=int, default=0, help="n > 0: a lane hands the processor back before one atomic operation in n (CILQR_EMU_PREEMPT): "
                    "other blocks run INSIDE the protocols' windows, e.g. between a push's reservation and the store of its entry")
    ap.add_argument("--replay", default="", help="a JSON list of case dictionaries (as printed) to run instead of random ones: the schedule is a function of "
                    "the case and of the library's code, so a case that reached a rare branch reaches it again")
    ap.add_argument("--hits-out", default="", dest="hits_out", help="write the cases that reached the claimed-place branch (probes of the emulator build) here")
    ap.add_argument("--kinds", default="solve", help="comma list of solve (barrier), alm (augmented Lagrangian in pairs), loop (closed loop in one launch)")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    bad = 0
    tot_parked = 0
    tot_places = [0, 0, 0]
    replay = json.load(open(a.replay)) if a.replay else None
    if replay is not None:
        a.cases = len(replay)
    hits = []
    for i in range(a.cases):
        N = rng.choice([20, 30, 37, 50] + ([70, 100, 130] if a.long else []))
        kind = rng.choice(a.kinds.split(","))
        c = {"kind": kind, "solve_type": 1 if kind == "alm" else 0, "ticks": rng.choice([2, 3]),
             "scenario": rng.choice(["three_bend", "two_straight", "two_borrow"]) if kind != "loop" else "three_straight", "N": N,
             "B": rng.choice([3, 5, 9, 16, 17, 24, 33, 48] if N <= 63 else [3, 5, 9, 12]), "x0_seed": rng.randrange(1 << 30),
             "max_iter": rng.choice([100, 100, 40]), "sched_seed": rng.randrange(1, 1 << 30),
             "blocks_per_cu": rng.choice([1, 2, 3, 4, 8]), "cus": rng.choice([1, 1, 2]),
             "slice": rng.choice([1, 2, 5, 16, 0]), "window": rng.choice([0, 50, 200, 1000]), "wait_model": "real"}
        if a.focus == "claims":
            c.update(N=rng.choice([12, 20]), B=rng.choice([5, 6, 7, 9, 12]), blocks_per_cu=rng.choice([2, 3]), cus=1, slice=rng.choice([1, 1, 2]),
                     window=1000, max_iter=rng.choice([20, 40]), kind="solve", solve_type=0, scenario=rng.choice(["three_bend", "two_borrow"]))
        if a.focus == "places":
            # slots that CLAIM a place in the queue beyond the pushes so far and keep it (GP_CLAIMED; an idle wavefront waiting with
            # a place): needs two takers racing for the last unclaimed push — use with --preempt — and pushes that then stop coming
            c.update(N=rng.choice([12, 20]), B=rng.choice([5, 7, 9, 12, 16, 20]), blocks_per_cu=rng.choice([3, 4, 6, 8]), cus=1,
                     slice=rng.choice([2, 3, 5]), window=rng.choice([0, 50, 1000]), max_iter=rng.choice([20, 40]), kind="solve", solve_type=0,
                     scenario=rng.choice(["three_bend", "two_borrow"]))
        if replay is not None:
            c = {k: v for k, v in replay[i].items() if k in c}
        env = dict(os.environ)
        env.update({"CILQR_AMD_LIB": a.lib, "CILQR_AMD_LIB_DEV": a.lib, "CILQR_EMU_SCHED_SEED": str(c["sched_seed"]), "CILQR_EMU_PREEMPT": str(a.preempt),
                    "CILQR_EMU_BLOCKS_PER_CU": str(c["blocks_per_cu"]), "CILQR_EMU_CUS": str(c["cus"]),
                    "CILQR_TUNE": "group_slice=%d,group_slice_long=%d,group_slice_window=%d" % (c["slice"], c["slice"], c["window"])})
        r = subprocess.run([sys.executable, "-c", CASE, ROOT, json.dumps(c)], capture_output=True, text=True, timeout=1800, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("CASE ")]
        if not line:
            print(json.dumps(dict(c, ok=False, crashed=r.stderr[-400:])), flush=True)
            bad += 1
            continue
        res = json.loads(line[-1][5:])
        tot_parked += res.get("parked", 0)
        tot_places[0] += res.get("waits_with_a_place", 0); tot_places[1] += res.get("places_kept", 0); tot_places[2] += res.get("idle_with_two_places", 0)
        print(json.dumps(res), flush=True)
        bad += 0 if res["ok"] else 1
        if res.get("waits_with_a_place") or res.get("places_kept"):
            hits.append(c)
    if a.hits_out:
        json.dump(hits, open(a.hits_out, "w"), indent=0)
    print(json.dumps({"cases": a.cases, "failed": bad, "hand_overs": tot_parked, "waits_with_a_place": tot_places[0], "places_kept": tot_places[1], "idle_with_two_places": tot_places[2]}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
