#!/usr/bin/env python3
"""Benchmark of the batched CILQR solve path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path — one cilqr_solve_batch_device call, i.e. CILQRSolver::solve
for every trajectory of the batch — over one batch of synthetic input that is already resident in
HBM.  Workload at N = 1: BASELINE.json configs[1] (batch = 1024 synthetic straight-lane scenarios,
horizon 50).  With N GPUs every rank solves its own 1024-trajectory shard of a 1024*N batch (weak
scaling, no data-path collective; RCCL only reduces the statistics afterwards).

metric = iLQR iterations/s = (sum over trajectories of executed iterations of the loop at
/root/reference/src/cilqr_solver.cpp:110) * steps / wall time, whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration (default 2 = configs[1], the headline)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (0 = the configuration's own)")
    ap.add_argument("--horizon", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="> 1: also measure the same steps with that many batches in flight (one handle and HIP stream "
                         "each) and report it under extra.pipelined — never the headline value.  Off by default so that "
                         "every k_solve launch of the default command is a sequential one (rocprofv3 averages agree).")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def make_workload(pkg, cfg_id, per_gpu_batch, horizon, rank, world):
    wl = pkg.workloads
    if cfg_id == 2:
        B = per_gpu_batch or 1024
        return wl.config2(B=B, N=horizon or 50, first=rank * B), B
    if cfg_id == 3:
        B = per_gpu_batch or 8192
        return wl.config3(B=B, N=horizon or 50, first=rank * B), B
    if cfg_id == 4:
        B = per_gpu_batch or 8192
        return wl.config4(B=B, N=horizon or 100, first=rank * B), B
    Bb = per_gpu_batch or 4096
    return wl.config5(B_base=Bb, N=horizon or 50, first=rank * Bb), Bb * 16


def usable_cores():
    """host cores this process may actually use: affinity mask and cgroup CPU quota both count
    (the GPU box exposes 256 hardware threads but the container is capped by cpu.max)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    note = None
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                note = f"cgroup cpu.max = {quota} {period} caps the container at {q} of {n} hardware threads"
                n = q
    except Exception:
        pass
    return n, note


def cpu_baseline(pkg, wl, threads):
    """The oracle (CPU restatement of the reference path, glibc libm build, -O3 -ffp-contract=off)
    timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle, Scene
    orc = Oracle("libm")
    scenes = [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes]
    nb = min(wl.B, 1024)
    x0, sid, pid, tk = wl.x0[:nb], wl.scenario_id[:nb], wl.param_id[:nb], wl.tick[:nb]
    # single thread on a 128-trajectory sub-sample
    n1 = min(nb, 128)
    t = time.perf_counter()
    r1 = orc.solve_batch(wl.params, scenes, x0[:n1], sid[:n1], pid[:n1], tk[:n1], n_threads=1)
    t1 = time.perf_counter() - t
    one_core = float(r1["res"]["iters"].sum()) / t1
    # all threads: one OpenMP region over the sample tiled so that it holds ~20 s of single-core work
    # (enough solves per thread to amortise start-up and the long-tailed solve times)
    t = time.perf_counter()
    rs = orc.solve_batch(wl.params, scenes, x0, sid, pid, tk, n_threads=threads)
    t_first = time.perf_counter() - t
    work_one = float(rs["res"]["iters"].sum()) / one_core          # single-core seconds in one copy
    tiles = int(max(1, min(64, round(20.0 / max(work_one, 1e-3)))))
    X0, SID, PID, TK = (np.tile(v, (tiles, 1)) if v.ndim == 2 else np.tile(v, tiles) for v in (x0, sid, pid, tk))
    best = None
    for _ in range(3):
        t = time.perf_counter()
        r = orc.solve_batch(wl.params, scenes, X0, SID, PID, TK, n_threads=threads)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    its = float(r["res"]["iters"].sum())
    return {"value": its / best, "unit": "iLQR iterations/s", "cores": threads, "kind": "port",
            "sample": f"{nb} trajectories of {wl.name} x {tiles} copies, cold-start solves, OpenMP over trajectories, "
                      f"best of 3 passes (~{work_one * tiles:.0f} s of single-core work per pass)",
            "one_core_value": one_core, "math": "glibc libm", "flags": "-O3 -ffp-contract=off",
            "first_pass_value_untiled": float(rs["res"]["iters"].sum()) / t_first}, rs


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # called directly with --gpus N: re-launch as one process per GPU, the way the driver does
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29541"),
               os.path.abspath(__file__)] + sys.argv[1:]
        os.execvp(cmd[0], cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the CILQR solve path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("CILQR_FORCE_DIST") == "1":  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    import cilqr_amd as pkg

    wl, B = make_workload(pkg, args.config, args.batch, args.horizon, rank, world)
    N = wl.N
    eng = pkg.BatchedCILQR(wl.params, wl.scenes, device=local_rank)
    dev = torch.device("cuda", local_rank)
    d_x0 = torch.from_numpy(wl.x0).to(dev)
    d_sid = torch.from_numpy(wl.scenario_id).to(dev)
    d_pid = torch.from_numpy(wl.param_id).to(dev)
    d_tick = torch.from_numpy(wl.tick).to(dev)
    d_u = torch.empty((B, N, 2), dtype=torch.float64, device=dev)
    d_x = torch.empty((B, N + 1, 4), dtype=torch.float64, device=dev)
    d_res = torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        eng.solve_batch_device(B, d_x0.data_ptr(), d_sid.data_ptr(), d_pid.data_ptr(), d_tick.data_ptr(), 0,
                               d_u.data_ptr(), d_x.data_ptr(), d_res.data_ptr(), 0, 0, stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record(stream)
        step()
        ev1[i].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))

    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
    M_of = wl.M_of

    # Not the headline: the same K steps with several batches in flight (one handle + HIP stream each).  A 1024
    # batch leaves most wave slots idle while its slowest trajectories finish; independent batches on other
    # streams fill them.  Reported under extra.pipelined, next to the strictly sequential headline value.
    pipelined = None
    if world == 1 and args.streams > 1:
        S = args.streams
        engs = [eng] + [pkg.BatchedCILQR(wl.params, wl.scenes, device=local_rank) for _ in range(S - 1)]
        strs = [torch.cuda.Stream(dev) for _ in range(S)]
        outs = [(d_u, d_x, d_res)] + [(torch.empty_like(d_u), torch.empty_like(d_x), torch.zeros_like(d_res))
                                      for _ in range(S - 1)]

        def pstep(i):
            e, st, (ou, ox, orr) = engs[i % S], strs[i % S], outs[i % S]
            e.solve_batch_device(B, d_x0.data_ptr(), d_sid.data_ptr(), d_pid.data_ptr(), d_tick.data_ptr(), 0,
                                 ou.data_ptr(), ox.data_ptr(), orr.data_ptr(), 0, 0, st.cuda_stream)

        for i in range(S):
            pstep(i)
        torch.cuda.synchronize(dev)
        tp = time.perf_counter()
        for i in range(args.steps):
            pstep(i)
        torch.cuda.synchronize(dev)
        tp = time.perf_counter() - tp
        same = all(bool(torch.equal(o[1], d_x)) and bool(torch.equal(o[2], d_res)) for o in outs[1:])
        pipelined = {"streams": S, "value": float(res["iters"].sum()) * args.steps / tp,
                     "ms_per_step": tp / args.steps * 1e3, "results_identical_to_sequential": same}
        for e in engs[1:]:
            e.close()
    from importlib import import_module
    st_mod = import_module("toy-example-of-ilqr_amd.stats")
    stats, tmax = st_mod.reduce_stats(st_mod.local_stats(res, N, M_of), elapsed, dist, dev if dist is not None else None)
    total_iters, total_trials = stats[0], stats[1]
    value = total_iters * args.steps / tmax

    if rank == 0:
        my_iters = float(res["iters"].sum())
        alg_bytes_launch = float((res["iters"] * pkg.workloads.bytes_per_iteration(N, M_of)).sum())
        achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of this same command (rocprofv3 cannot be nested
        # inside the benchmark); recorded under profiles/ together with how it was collected
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_config2.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            if pmc.get("workload") == wl.name and world == 1:
                traffic = pmc["hbm_bytes_per_launch_corrected"]
                traffic_src = "profiles/r01_pmc_config2.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
        # the bound that actually applies: vector-instruction issue.  Wave instructions per launch from the SQ
        # counter pass (profiles/), 4 cycles of a SIMD each at best, against all SIMD-cycles of the live kernel time
        valu = None
        sq_path = os.path.join(ROOT, "profiles", "r01_pmc_sq_config2.json")
        if os.path.exists(sq_path) and traffic is not None:
            sq = json.load(open(sq_path))
            if "SQ_INSTS_VALU" in sq:
                simds, clock_hz = 256 * 4, 2.4e9
                valu = {"wave_instructions_per_launch": sq["SQ_INSTS_VALU"],
                        "frac_of_issue_slots": sq["SQ_INSTS_VALU"] * 4.0 / (simds * clock_hz * kernel_ms * 1e-3),
                        "source": "profiles/r01_pmc_sq_config2.json (SQ_INSTS_VALU), 256 CUs x 4 SIMDs, 4 cycles per "
                                  "wave64 instruction, 2.4 GHz"}
        out = {
            "metric": "iLQR iterations/sec (batch x horizon)", "value": value, "unit": "iLQR iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name, "baseline_config": args.config, "batch_per_gpu": B,
                       "global_batch": int(stats[8]), "horizon": N, "nx": 4, "nu": 2,
                       "parallelism": f"trajectory-sharded x{world}, one wavefront per trajectory"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_solve", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "valu_issue": valu,
                         "note": "the fused solve is FP64-VALU/latency bound, not HBM bound (DESIGN.md)"},
            "extra": {"iterations_per_step_rank0": my_iters, "iterations_per_solve_mean": my_iters / B,
                      "line_search_trials_per_step": float(total_trials),
                      "solves_per_s": stats[8] * args.steps / tmax,
                      "step_updates_per_s": value * N,
                      "converged": int(stats[2]), "max_lamb": int(stats[3]), "max_iter": int(stats[4]),
                      "nan_costs": int(stats[6]), "sum_J_final": float(stats[5]), "pipelined": pipelined},
        }
        if not args.no_cpu_baseline:
            cores, cores_note = usable_cores()
            threads = args.cpu_threads or cores
            cb, r = cpu_baseline(pkg, wl, threads)
            if cores_note:
                cb["cores_note"] = cores_note
            out["cpu_baseline"] = cb
            # parity of the run that was just timed (the oracle is only the checker here)
            nb = r["res"].shape[0]
            out["extra"]["cpu_check"] = {
                "trajectories": nb,
                "same_iterations": int((r["res"]["iters"] == res["iters"][:nb]).sum()),
                "max_abs_dJ_final": float(np.nanmax(np.abs(r["res"]["J_final"] - res["J_final"][:nb]))),
            }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
