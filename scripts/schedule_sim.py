#!/usr/bin/env python3
"""What could a scheduler do for the 8192-trajectory launches?  Replays the MEASURED per-trajectory solve times of a
launch (block timeline: gpurun_out/TAG/timeline_cK.npy or an .npz of it (raw records are scratch, not tracked), written by scripts/block_timeline.py on the GPU box) through
list schedulers on the chip's 2048 resident-wavefront slots:

  fifo      pull order as measured (the persistent blocks of the library)                      -> reproduces the launch
  lpt       longest first with the lengths known in advance (clairvoyant; the bound a perfect predictor would give)
  sliced    resumable solves: every trajectory runs K iterations at a time, is parked and re-queued (VERDICT r02 item 1:
            "park x, u, lambda, status, ... every K iterations"); fresh trajectories first, then parked ones either in
            parking order or oldest (most iterations done) first; 8 us per park / resume
  by_class  longest expected class first (class = scenario of configs[3]): what per-class statistics of earlier launches
            would allow without remembering individual trajectories

A trajectory's measured time is spread over its iterations in proportion to (a + c x trials of the iteration), a and c
fitted to the launch itself; iterations and trials per iteration come from the CPU oracle's decision trace (TEST
INFRASTRUCTURE: this script imports oracle/, the library never does).  No GPU needed.

    python scripts/schedule_sim.py --config 3 --timeline gpurun_out/TAG/timeline_c3.npy [--out profiles/r03_schedule_sim_c3.json]
"""
import argparse
import heapq
import json
import os
import sys
from collections import deque
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
P = 2048  # resident wavefronts of the large-batch builds: 8 per CU x 256 CUs


def _traces(args):
    cfg, lo, hi = args
    import cilqr_amd as pkg
    from oracle import Oracle, Scene
    W = pkg.workloads
    wl = W.config3() if cfg == 3 else W.config4(B=8192)
    orc = Oracle("det")
    scenes = [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes]
    solvers = [orc.solver(p) for p in wl.params]
    out = []
    for b in range(lo, hi):
        s = solvers[wl.param_id[b]]
        s.reset()
        r = s.solve(wl.x0[b], scenes[wl.scenario_id[b]], tick=int(wl.tick[b]), trace_cap=128)
        out.append((b, r["trace"]["trials"].copy()))
    return out


def decision_traces(cfg, B=8192):
    with Pool(os.cpu_count() or 1) as pool:
        res = pool.map(_traces, [(cfg, i, min(i + 64, B)) for i in range(0, B, 64)])
    flat = sorted((r for c in res for r in c), key=lambda r: r[0])
    n = max(len(t) for _, t in flat)
    trials = np.full((B, n), -1, np.int32)
    for b, t in flat:
        trials[b, :len(t)] = t
    return trials


def fifo(order, dur):
    h = [0.0] * P
    heapq.heapify(h)
    end = 0.0
    for b in order:
        t = heapq.heappop(h) + dur[b]
        heapq.heappush(h, t)
        end = max(end, t)
    return end


def sliced(K, it_ms, iters, order, overhead_ms, oldest_first):
    B = len(iters)
    fresh, parked = deque(order), []
    pos = np.zeros(B, int)
    idle = []
    ev = [(0.0, s, -1) for s in range(P)]
    heapq.heapify(ev)
    end, seq = 0.0, 0

    def start(t, s):
        nonlocal end
        if fresh:
            b = fresh.popleft()
        elif parked:
            b = heapq.heappop(parked)[2]
        else:
            return False
        k2 = min(pos[b] + K, iters[b])
        t2 = t + overhead_ms + it_ms[b, pos[b]:k2].sum()
        pos[b] = k2
        end = max(end, t2)
        heapq.heappush(ev, (t2, s, b if k2 < iters[b] else -1))
        return True

    while ev:
        t, s, pb = heapq.heappop(ev)
        if pb >= 0:
            seq += 1
            heapq.heappush(parked, (-pos[pb] if oldest_first else seq, seq, pb))
        cand, idle = [s] + idle, []
        for s2 in cand:
            if not start(t, s2):
                idle.append(s2)
    assert (pos == iters).all()
    return end


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=[3, 4], required=True)
    ap.add_argument("--timeline", required=True)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    tl = np.load(a.timeline)
    tl = tl["start_end_block_xcc"] if hasattr(tl, "files") else tl
    t0 = tl[:, 0].min()
    start = (tl[:, 0] - t0) / 1e5                 # ms (100 MHz clock)
    dur = (tl[:, 1] - tl[:, 0]) / 1e5
    B = len(dur)
    trials = decision_traces(a.config, B)
    valid = trials >= 0
    iters = valid.sum(1)
    # time per iteration ~ a + c x trials, fitted on this launch
    X = np.stack([iters, np.where(valid, trials, 0).sum(1), np.ones(B)], 1).astype(float)
    coef = np.linalg.lstsq(X, dur, rcond=None)[0]
    w = np.where(valid, max(coef[0], 1e-6) + max(coef[1], 0.0) * trials, 0.0)
    it_ms = w / w.sum(1, keepdims=True) * dur[:, None]
    order = np.argsort(start, kind="stable")
    rep = {"workload": f"config {a.config}, {B} trajectories, measured solve times of one launch", "timeline": a.timeline,
           "slots": P, "measured_launch_ms": float((tl[:, 1].max() - t0) / 1e5),
           "work_over_slots_ms (W/P: perfect packing)": float(dur.sum() / P), "longest_solve_ms": float(dur.max()),
           "mean_solve_ms": float(dur.mean()), "fit_ms_per_iteration": float(coef[0]), "fit_ms_per_trial": float(coef[1]),
           "fifo_replay_ms": fifo(order, dur), "lpt_clairvoyant_ms": fifo(np.argsort(-dur), dur), "sliced": []}
    for K in (4, 8, 16, 32):
        for oldest in (False, True):
            e = sliced(K, it_ms, iters, order, 0.008, oldest)
            rep["sliced"].append({"iterations_per_slice": K, "parked_queue": "oldest first" if oldest else "parking order",
                                  "launch_ms": e, "mean_residency": float(dur.sum() / P / e)})
    if a.config == 4:
        scen = np.arange(B) % 4
        means = [dur[scen == s].mean() for s in range(4)]
        by = np.concatenate([np.nonzero(scen == s)[0] for s in np.argsort(means)[::-1]])
        rep["by_class_ms (scenario with the longest mean first)"] = fifo(by, dur)
        rep["class_mean_solve_ms"] = [float(m) for m in means]
    rep["residency_fifo"] = float(dur.sum() / P / rep["fifo_replay_ms"])
    rep["best_sliced_ms"] = min(s["launch_ms"] for s in rep["sliced"])
    txt = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
