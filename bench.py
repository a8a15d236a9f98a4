#!/usr/bin/env python3
"""Benchmark of the batched CILQR solve path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path — one cilqr_solve_batch_device call, i.e. CILQRSolver::solve
for every trajectory of the batch — over one batch of synthetic input that is already resident in
HBM.

Workload (BASELINE.json `configs`):
  * N = 1 (default): config 5 — 4096 perturbed three_bend starts x 16 barrier settings = 65 536 solves of
    horizon 50, the largest single-GPU configuration (north_star: ">= 10k trajectories x 50-step horizon on
    1 x MI355X").  Config 2 (1024 straight-lane trajectories, one launch = a latency measurement: its wall
    time is the slowest trajectory's) is timed in the same run and reported under `extra.config2_latency`,
    one rank's shard of BASELINE configs[3] (8192 mixed scenarios of horizon 100) under `extra.config4_sharded`.
  * N > 1 (default): the SAME workload per GPU as at N = 1 — every rank solves its own 4096 x 16 sweep (base
    starts first = rank * 4096 of one global batch), so that value(N) / value(1) is a weak-scaling efficiency
    whoever computes it; no data-path collective, RCCL only reduces the statistics afterwards.  BASELINE
    configs[3] — 65 536 mixed scenarios, horizon 100, sharded 8 x 8192 (8192 per rank) — is timed in the same
    run and reported under `extra.config4_sharded`, config 2's weak scaling (1024 per GPU) under
    `extra.config2_weak_scaling`.
  * --config 1: the reference's own case (scenario_two_straight, single ego, horizon 50) as a 10 Hz closed
    loop through the drop-in CILQRSolver.solve(), B = 1: per-tick latency next to the oracle on one core.
  * --config 2|3|4|5 selects any of them explicitly.
  * Also timed by the default command (never the headline): `extra.closed_loop` — 8192 egos x 40 ticks of the
    reference's planning loop (motion_planning.cpp:180-197) resident on the device, every tick warm-started from the
    previous one (cilqr_solver.cpp:163-180), and `extra.config5_alm` — the headline batch with the augmented-Lagrangian
    solve type (cilqr_solver.cpp:88-93).  Every extra is wrapped: a failing one is reported under its key and cannot
    lose the headline line.

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
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_current.json")  # written by scripts/summarize_profiles.py


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="BASELINE.json configuration (0 = default: 5 per GPU, with 2 (and 4 on several GPUs) as extras)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (0 = the configuration's own)")
    ap.add_argument("--horizon", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline workload (profiling passes: every k_solve launch is then the headline's)")
    ap.add_argument("--in-flight", type=int, default=3, dest="in_flight",
                    help="batches in flight inside the handle (cilqr_set_batches_in_flight): the K timed steps go round robin to "
                         "this many launch slots and output buffer sets, so that the next batch's blocks fill the tail of the "
                         "previous launch; 1 = one launch at a time (what rounds 1-4 reported).  The line states it.")
    ap.add_argument("--streams", type=int, default=1,
                    help="> 1: also measure the same steps with that many batches in flight (one handle and HIP stream "
                         "each) and report it under extra.pipelined — never the headline value.  Off by default so that "
                         "every k_solve launch of the default command is a sequential one (rocprofv3 averages agree).")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--only-extra", default="", dest="only_extra", help=argparse.SUPPRESS)  # (the child of isolated_extra below)
    ap.add_argument("--ticks", type=int, default=120, help="--config 1: closed-loop ticks")
    return ap.parse_args()


def make_workload(pkg, cfg_id, per_gpu_batch, horizon, rank):
    w, B = _make_workload(pkg, cfg_id, per_gpu_batch, horizon, rank)
    if os.environ.get("CILQR_BENCH_ALM") == "1":  # the same batch with the augmented-Lagrangian solve type (not a BASELINE config)
        w = pkg.workloads.Workload(w.name + "_alm", [pkg.copy_params(q, solve_type=1) for q in w.params], w.scenes, w.x0,
                                   w.scenario_id, w.param_id, w.tick)
    return w, B


# CILQR_BENCH_REHEARSAL=k: every default batch size divided by k and the closed loops cut to 3 ticks — so that the WHOLE default
# command (headline + every extra) can be walked through where a solve takes a tenth of a second (the CPU emulator of the test
# suite, scripts/emu_rehearse.py).  The line then says "rehearsal": k at top level and in `data`; it is not a measurement of
# anything and the driver never sets the variable.
REHEARSAL = max(1, int(os.environ.get("CILQR_BENCH_REHEARSAL", "1")))


def _make_workload(pkg, cfg_id, per_gpu_batch, horizon, rank):
    wl = pkg.workloads
    if cfg_id == 2:
        B = per_gpu_batch or max(4, 1024 // REHEARSAL)
        return wl.config2(B=B, N=horizon or 50, first=rank * B), B
    if cfg_id == 3:
        B = per_gpu_batch or max(4, 8192 // REHEARSAL)
        return wl.config3(B=B, N=horizon or 50, first=rank * B), B
    if cfg_id == 4:
        B = per_gpu_batch or max(4, 8192 // REHEARSAL)
        return wl.config4(B=B, N=horizon or 100, first=rank * B), B
    Bb = per_gpu_batch or max(1, 4096 // REHEARSAL)
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


def oracle_scenes(wl):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Scene
    return [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes]


def cpu_baseline(pkg, wl, threads):
    """The oracle (CPU restatement of the reference path, glibc libm build, -O3 -ffp-contract=off)
    timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    orc = Oracle("libm")
    scenes = oracle_scenes(wl)
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
            "sample": f"first {nb} trajectories of {wl.name} x {tiles} copies, cold-start solves, OpenMP over "
                      f"trajectories, best of 3 passes (~{work_one * tiles:.0f} s of single-core work per pass)",
            "one_core_value": one_core, "math": "glibc libm", "flags": "-O3 -ffp-contract=off",
            "first_pass_value_untiled": float(rs["res"]["iters"].sum()) / t_first}, rs


def parity_check(gpu_u, gpu_x, gpu_res, ref, tol=1e-5):
    """The run that was just timed against the libm oracle (the oracle is only the checker here).  The GPU
    path is bit-identical to the oracle's detmath build (tests); against glibc's libm the elementary functions
    differ by an ulp.  On inputs whose solve amplifies such a difference past 1e-5 the libm build does not
    reproduce itself either when x0 moves by one unit in the last place (profiles/r02_libm_tolerance.json,
    tests/test_gpu_parity.py::test_libm_gap_is_input_conditioning): 10 of config 2's 1024 straight-lane
    trajectories, none of configs 3 and 5."""
    nb = ref["res"].shape[0]
    du = np.abs(gpu_u[:nb] - ref["u"]).reshape(nb, -1).max(axis=1)
    dx = np.abs(gpu_x[:nb] - ref["x"]).reshape(nb, -1).max(axis=1)
    dJ = np.abs(gpu_res["J_final"][:nb] - ref["res"]["J_final"])
    bad = ~((du <= tol) & (dx <= tol) & (dJ <= tol))  # NaN counts as outside
    same_path = (gpu_res["iters"][:nb] == ref["res"]["iters"]) & (gpu_res["ls_trials"][:nb] == ref["res"]["ls_trials"])
    return {"trajectories": int(nb), "tolerance": tol,
            "within_1e-5_frac": float(1.0 - bad.mean()),
            "outside_1e-5": int(bad.sum()),
            "same_iterations": int((gpu_res["iters"][:nb] == ref["res"]["iters"]).sum()),
            "same_decision_path": int(same_path.sum()),
            "outside_1e-5_among_same_decision_path": int((bad & same_path).sum()),
            "max_abs_du_same_path": float(du[same_path].max()) if same_path.any() else None,
            "max_abs_dx_same_path": float(dx[same_path].max()) if same_path.any() else None,
            "max_abs_dJ_same_path": float(dJ[same_path].max()) if same_path.any() else None,
            "max_abs_dJ_final": float(np.nanmax(dJ))}


def det_identity(gpu_u, gpu_x, gpu_res, det):
    """bit-for-bit agreement of the launch that was timed with the oracle's detmath build (same arithmetic as the
    kernel: the parity claim proper; NaN patterns compared as bits)"""
    nb = det["res"].shape[0]
    same = ((gpu_u[:nb].reshape(nb, -1).view(np.uint64) == np.ascontiguousarray(det["u"]).reshape(nb, -1).view(np.uint64)).all(axis=1)
            & (gpu_x[:nb].reshape(nb, -1).view(np.uint64) == np.ascontiguousarray(det["x"]).reshape(nb, -1).view(np.uint64)).all(axis=1)
            & (gpu_res["iters"][:nb] == det["res"]["iters"]) & (gpu_res["ls_trials"][:nb] == det["res"]["ls_trials"])
            & (gpu_res["J_final"][:nb].view(np.uint64) == det["res"]["J_final"].view(np.uint64)))
    return {"trajectories": int(nb), "identical": int(same.sum())}


class GpuRun:
    """One workload resident in HBM + the timed loop over it."""

    def __init__(self, pkg, torch, wl, B, local_rank, group_mode=None):
        self.pkg, self.torch, self.wl, self.B = pkg, torch, wl, B
        N = self.N = wl.N
        self.dev = dev = torch.device("cuda", local_rank)
        self.eng = pkg.BatchedCILQR(wl.params, wl.scenes, device=local_rank)
        if group_mode is not None:
            self.eng.set_group_mode(group_mode)
        self.d_x0 = torch.from_numpy(wl.x0).to(dev)
        self.d_sid = torch.from_numpy(wl.scenario_id).to(dev)
        self.d_pid = torch.from_numpy(wl.param_id).to(dev)
        self.d_tick = torch.from_numpy(wl.tick).to(dev)
        self.d_u = torch.empty((B, N, 2), dtype=torch.float64, device=dev)
        self.d_x = torch.empty((B, N + 1, 4), dtype=torch.float64, device=dev)
        self.d_res = torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.current_stream(dev)

    def step(self, eng=None, outs=None, stream=None):
        e = eng or self.eng
        ou, ox, orr = outs or (self.d_u, self.d_x, self.d_res)
        st = stream or self.stream
        e.solve_batch_device(self.B, self.d_x0.data_ptr(), self.d_sid.data_ptr(), self.d_pid.data_ptr(),
                             self.d_tick.data_ptr(), 0, ou.data_ptr(), ox.data_ptr(), orr.data_ptr(), 0, 0,
                             st.cuda_stream)

    def timed(self, steps, warmup, barrier, in_flight=1, seq_steps=0):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides.
        in_flight = 1: one launch at a time; HIP events on the launch stream around every launch give the kernel's own
        duration.  in_flight > 1 (cilqr_set_batches_in_flight): the K steps go round robin to that many launch slots of
        the handle and output buffer sets; the launches overlap, so the kernel time of the region is taken with two
        events around ALL of it (first launch enqueued ... every slot joined), per-launch durations from the slots' own
        events, and `seq_steps` further launches are timed one at a time afterwards (not part of the timed region)."""
        torch = self.torch
        self.in_flight = K = max(1, int(in_flight))
        outs = [(self.d_u, self.d_x, self.d_res)]
        if K > 1:
            self.eng.set_batches_in_flight(K)
            self.eng.set_timing(True)
            outs += [(torch.empty_like(self.d_u), torch.empty_like(self.d_x), torch.zeros_like(self.d_res)) for _ in range(K - 1)]
        for i in range(max(warmup, K if K > 1 else 0)):
            self.step(outs=outs[i % K])
        if K > 1:
            self.eng.join_device(self.stream.cuda_stream)
        barrier()
        extra = {}
        if K == 1:
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
            t0 = time.perf_counter()
            for i in range(steps):
                ev0[i].record(self.stream)
                self.step()
                ev1[i].record(self.stream)
            barrier()
            elapsed = time.perf_counter() - t0
            kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(self.stream)
            for i in range(steps):
                self.step(outs=outs[i % K])
            self.eng.join_device(self.stream.cuda_stream)
            e1.record(self.stream)
            barrier()
            elapsed = time.perf_counter() - t0
            region_ms = float(e0.elapsed_time(e1))
            kernel_ms = region_ms / steps  # effective: the launches overlap
            per_launch = [self.eng.slot_kernel_ms(k) for k in range(K)]
            ident = all(bool(torch.equal(o[0], outs[0][0])) and bool(torch.equal(o[1], outs[0][1])) and bool(torch.equal(o[2], outs[0][2]))
                        for o in outs[1:])
            extra = {"in_flight": K, "steps": steps, "kernel_region_ms": region_ms,
                     "kernel_ms_of_one_launch_while_overlapped (last launch of each slot)": per_launch,
                     "mean_launches_overlapping": float(np.mean(per_launch)) * steps / region_ms if region_ms > 0 else None,
                     "buffer_sets_identical": ident}
            if seq_steps > 0:
                # the same launches one at a time (what rounds 1-4 timed): not part of the timed region above
                self.eng.set_batches_in_flight(1)
                ref = (torch.empty_like(self.d_u), torch.empty_like(self.d_x), torch.zeros_like(self.d_res))
                ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(seq_steps)]
                ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(seq_steps)]
                ts = time.perf_counter()
                for i in range(seq_steps):
                    ev0[i].record(self.stream)
                    self.step(outs=ref)
                    ev1[i].record(self.stream)
                barrier()
                ts = time.perf_counter() - ts
                extra["sequential"] = {"steps": seq_steps, "kernel_ms": float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)])),
                                       "ms_per_step": ts / seq_steps * 1e3,
                                       "results_identical_to_in_flight": bool(torch.equal(ref[0], outs[0][0])) and bool(torch.equal(ref[1], outs[0][1]))
                                       and bool(torch.equal(ref[2], outs[0][2]))}
        self.flight = extra
        res = np.frombuffer(self.d_res.cpu().numpy().tobytes(), dtype=self.pkg.RESULT_DTYPE)
        try:
            self.launch_info = self.eng.last_launch_info()
        except Exception:  # noqa: BLE001
            self.launch_info = None
        return elapsed, kernel_ms, res

    def pipelined(self, steps, S):
        """Not the headline: the same K steps with several batches in flight (one handle + HIP stream each)."""
        torch, pkg, wl = self.torch, self.pkg, self.wl
        engs = [self.eng] + [pkg.BatchedCILQR(wl.params, wl.scenes, device=self.dev.index) for _ in range(S - 1)]
        strs = [torch.cuda.Stream(self.dev) for _ in range(S)]
        outs = [(self.d_u, self.d_x, self.d_res)] + [
            (torch.empty_like(self.d_u), torch.empty_like(self.d_x), torch.zeros_like(self.d_res)) for _ in range(S - 1)]
        ref_x, ref_res = self.d_x.clone(), self.d_res.clone()
        for i in range(S):
            self.step(engs[i % S], outs[i % S], strs[i % S])
        torch.cuda.synchronize(self.dev)
        tp = time.perf_counter()
        for i in range(steps):
            self.step(engs[i % S], outs[i % S], strs[i % S])
        torch.cuda.synchronize(self.dev)
        tp = time.perf_counter() - tp
        same = all(bool(torch.equal(o[1], ref_x)) and bool(torch.equal(o[2], ref_res)) for o in outs)
        res = np.frombuffer(ref_res.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
        for e in engs[1:]:
            e.close()
        return {"streams": S, "value": float(res["iters"].sum()) * steps / tp,
                "ms_per_step": tp / steps * 1e3, "results_identical_to_sequential": same}

    def close(self):
        self.eng.close()


def counters_for(workload):
    """HBM bytes and wave-instruction counts per launch from the PMC passes of this same command on this
    workload (rocprofv3 cannot be nested inside the benchmark): profiles/pmc_current.json, keyed by workload
    name, written by scripts/summarize_profiles.py together with how they were collected."""
    if not os.path.exists(PMC_FILE):
        return None
    try:
        return json.load(open(PMC_FILE)).get(workload)
    except Exception:
        return None


def csrc_fingerprint():
    """sha256 over the kernel sources (csrc/*, include/cilqr_amd.h): what the counter passes were collected on vs. what runs"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "toy-example-of-ilqr_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".hpp", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(base, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "cilqr_amd.h"), "rb").read())
    return h.hexdigest()[:16]


_DEVICE_CODE_CACHE = {}


def device_code_unchanged(manifest_rel):
    """Are the INSTRUCTIONS of the library this process runs the ones the counter passes were collected on?  The source stamp
    (csrc_fingerprint) moves with any host-side edit of csrc/; this compares the machine code of every kernel and out-of-line
    device function inside libcilqr_amd.so with the manifest recorded at collection (scripts/device_code_identity.py).
    None when it cannot be answered (no manifest recorded, LLVM tools missing)."""
    if not manifest_rel:
        return None
    if manifest_rel in _DEVICE_CODE_CACHE:
        return _DEVICE_CODE_CACHE[manifest_rel]
    ans = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import device_code_identity as dci
        ref = json.load(open(os.path.join(ROOT, manifest_rel)))["functions"]
        ref = {k: sorted([list(x) for x in v]) for k, v in ref.items()}
        r = dci.compare(ref, dci.manifest(os.path.join(ROOT, "toy-example-of-ilqr_amd", "libcilqr_amd.so")))
        solve = [n for n in r["changed"] + r["missing"] if "k_solve" in n]
        ans = {"manifest": manifest_rel, "functions_same": r["same"], "functions_changed": len(r["changed"]),
               "functions_missing": len(r["missing"]), "functions_added": len(r["added"]),
               "functions_same_under_another_name (template parameters added)": len(r["renamed"]),
               "all_solve_kernels_unchanged": not solve, "solve_kernels_changed": [n[:60] for n in solve][:8],
               # the headline's kernel and its phases: k_solve_grp<50, 2, false, 1>
               "headline_kernel_unchanged": not any("k_solve_grpILi50ELi2ELb0E" in n for n in r["changed"] + r["missing"]),
               "everything_unchanged": not r["changed"] and not r["missing"]}
    except Exception as e:  # noqa: BLE001
        ans = {"error": f"{type(e).__name__}: {e}"[:200]}
    _DEVICE_CODE_CACHE[manifest_rel] = ans
    return ans


def roofline_block(pkg, wl, res, kernel_ms, world, launch_info=None, flight=None):
    """kernel_ms: the average duration of ONE launch with nothing else in flight (HIP events around every launch) — what
    rocprofv3's per-kernel average of `--in-flight 1` shows and what the counter passes of pmc_current.json ran as.  When the
    timed region kept several batches in flight, the region's effective per-launch time is reported under
    in_flight.effective (throughput, not a kernel duration)."""
    N, M_of = wl.N, wl.M_of
    alg_bytes_launch = float((res["iters"] * pkg.workloads.bytes_per_iteration(N, M_of)).sum())
    eff_ms = None
    if flight and flight.get("in_flight", 1) > 1:
        eff_ms = flight.get("kernel_region_ms", 0.0) / max(1, flight.get("steps", 1)) or None
        seq = (flight.get("sequential") or {}).get("kernel_ms")
        if seq:
            kernel_ms = seq
        flight = dict(flight)
        if eff_ms:
            flight["effective"] = {"kernel_ms": eff_ms, "achieved_GBs": alg_bytes_launch / (eff_ms * 1e-3) / 1e9,
                                   "frac": alg_bytes_launch / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "what": "kernel time of the whole timed region (two HIP events: first launch enqueued ... every "
                                           "launch slot joined) / launches; the launches overlap, so this is a throughput figure — "
                                           "rocprofv3's per-kernel average of the default command is the OVERLAPPED duration of one "
                                           "launch (kernel_ms_of_one_launch_while_overlapped), longer than either"}
    achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9
    pmc = counters_for(wl.name) if world == 1 else None
    traffic = traffic_src = valu = None
    traffic_meta = None
    if pmc:
        traffic = pmc.get("hbm_bytes_per_launch_corrected")
        traffic_src = pmc.get("source")
        try:
            meta = json.load(open(PMC_FILE)).get("_collected", {})
        except Exception:  # noqa: BLE001
            meta = {}
        now = csrc_fingerprint()
        traffic_meta = {"collected_at_git": meta.get("git"), "collected_on_csrc_sha16": meta.get("csrc_sha16"),
                        "this_run_csrc_sha16": now, "collected_with_in_flight": meta.get("in_flight", 1),
                        "stale": (meta.get("csrc_sha16") != now),
                        "device_code": device_code_unchanged(meta.get("device_code_manifest"))}
        if pmc.get("SQ_INSTS_VALU"):
            # the bound that actually applies: vector-instruction issue.  Wave instructions per launch from the
            # SQ counter pass, 4 cycles of a SIMD each at best, against all SIMD-cycles of the live kernel time
            simds, clock_hz = 256 * 4, 2.4e9
            valu = {"wave_instructions_per_launch": pmc["SQ_INSTS_VALU"],
                    "frac_of_issue_slots": pmc["SQ_INSTS_VALU"] * 4.0 / (simds * clock_hz * kernel_ms * 1e-3),
                    "source": "SQ_INSTS_VALU of the same file; 256 CUs x 4 SIMDs, 4 cycles per wave64 "
                              "instruction, 2.4 GHz (rocm-smi shows 2.36 GHz at ~900 W under this load)"}
            if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_WAVE_CYCLES"):
                # share of a resident wave's cycles in which it executes a vector instruction, and how many waves are
                # resident on average (SQ_WAVE_CYCLES counts quad-cycles; the chip holds 2048 waves of the
                # large-batch build, two per SIMD): SIMD VALU busy = share x resident waves / 1024
                valu["valu_active_share_of_wave_cycles"] = pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"]
                valu["mean_resident_waves"] = pmc["SQ_WAVE_CYCLES"] * 4.0 / (2.36e9 * kernel_ms * 1e-3)
            if pmc.get("SQ_THREAD_CYCLES_VALU") and pmc.get("SQ_ACTIVE_INST_VALU"):
                # how many of the 64 lanes of an issued vector instruction are live, averaged over the instructions'
                # cycles (both counters in quad-cycles: thread-cycles / (instruction-cycles x 64)).  The serial phases
                # pull it down: the lone first-trial rollout runs one (with trajectories in pairs: two) lanes of 64
                valu["valu_lane_occupancy"] = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0)
    # how much of the FP64 vector peak does the ALGORITHM's arithmetic amount to: SURVEY 8(d)'s dense count
    # N (965 + 168 M + T (106 + 34 M)) flops per iteration, T = line-search trials per iteration of THIS run
    iters_sum = float(res["iters"].sum())
    trials_per_iter = float(res["ls_trials"].sum()) / iters_sum if iters_sum else 0.0
    flops_launch = float((res["iters"] * N * (965.0 + 168.0 * M_of)).sum() + (res["ls_trials"] * N * (106.0 + 34.0 * M_of)).sum())
    useful = {"flops_per_launch": flops_launch, "trials_per_iteration": trials_per_iter,
              "fp64_useful_frac": flops_launch / (kernel_ms * 1e-3) / (FP64_VECTOR_PEAK_TFLOPS * 1e12),
              "definition": "SURVEY.md 8(d): N (965 + 168 M + T (106 + 34 M)) FP64 flops per iteration as the reference "
                            "computes them (dense 4x4 products, transcendentals NOT weighted), T from this run, over "
                            "the 78.6 TFLOP/s FP64 vector peak"}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_collected": traffic_meta,
            "traffic_is": "bytes crossing the L2 <-> fabric boundary per launch (Infinity Cache + HBM; the counters cannot "
                          "tell the two apart: profiles/r03_traffic_calibration.json), from the counter passes of the same "
                          "command — a static file, not this run",
            "kernel": ("k_solve_grp" if (launch_info or {}).get("trajectories_per_wavefront", 1) > 1 else "k_solve"),
            "launch": launch_info, "kernel_ms": kernel_ms,
            "kernel_ms_is": "average duration of a launch, one launch at a time (HIP events around every launch); with batches in "
                            "flight in the timed region: of the sequential leg timed right after it (in_flight.sequential), and "
                            "in_flight.effective has the region's own figure.  achieved / frac / valu_issue / fp64_useful all use "
                            "this duration — the mode the counter passes ran in",
            "in_flight": flight or {"in_flight": 1},
            "algorithmic_bytes_per_launch": alg_bytes_launch,
            "algorithmic_bytes_per_iteration": "16(6N+4) + 24M(N+1) (SURVEY.md 8(d))",
            "valu_issue": valu,
            "fp64_useful": useful,
            "note": "the fused solve is FP64-VALU/latency bound, not HBM bound (DESIGN.md)"}


def device_identity(torch, local_rank):
    """what identifies the GPU this rank drives: gathered from every rank so that the line shows N distinct devices"""
    pr = torch.cuda.get_device_properties(local_rank)
    bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xff,
                              getattr(pr, "pci_device_id", 0))
    return {"local_rank": int(local_rank), "device": pr.name, "arch": getattr(pr, "gcnArchName", ""),
            "pci_bus_id": bus, "uuid": str(getattr(pr, "uuid", "")), "compute_units": int(pr.multi_processor_count),
            "pid": os.getpid()}


def gather_objects(dist, obj, world):
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def closed_loop_run(pkg, torch, local_rank, rank, barrier, B=8192, ticks=40, N=50, check_egos=8):
    """thousands of egos x ticks with the plan fed back on the device.  One handle, scenario
    three_straight (the YAML that sets use_last_solution, config/scenario_three_straight.yaml:24; 8 obstacles), every
    ego its own perturbed start; per tick one cilqr_solve_batch_device warm-started from its own previous u buffer
    (cs:163-180) and one cilqr_advance_batch_device (ego <- x.row(1), obstacle window one tick on: mp:181,196-197).
    No host round trip inside the timed region.  The first `check_egos` egos' states after every tick are kept (on the
    device) for the comparison with stateful oracle solvers."""
    cfg = pkg.GlobalConfig.get_instance("three_straight")
    sc = pkg.build_scenario(cfg, "three_straight")
    p = pkg.params_from_config(cfg, N=N, use_last_solution=1)
    if REHEARSAL > 1:
        B, ticks, check_egos = max(4, B // REHEARSAL), 3, 2
    ticks = min(ticks, sc.obstacles.shape[1] - N - 1)
    dev = torch.device("cuda", local_rank)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xC11A00F2, first=rank * B)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), device=local_rank)
    d_u = torch.zeros((B, N, 2), dtype=torch.float64, device=dev)
    d_x = torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev)
    d_res = torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    d_its = torch.zeros((ticks, B), dtype=torch.int32, device=dev)
    d_hist = torch.zeros((ticks, check_egos, N + 1, 4), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    its_off = pkg.RESULT_DTYPE.fields["iters"][1]

    def loop(record):
        d_x0 = torch.from_numpy(x0).to(dev)
        d_tick = torch.zeros(B, dtype=torch.int32, device=dev)
        for t in range(ticks):
            eng.solve_batch_device(B, d_x0.data_ptr(), 0, 0, d_tick.data_ptr(), d_u.data_ptr() if t else 0, d_u.data_ptr(),
                                   d_x.data_ptr(), d_res.data_ptr(), 0, 0, st.cuda_stream)
            if record:  # device-side, stream-ordered copies: iterations of this tick, plans of the sampled egos
                d_its[t].copy_(d_res[:, its_off:its_off + 4].contiguous().view(torch.int32).view(B))
                d_hist[t].copy_(d_x[:check_egos])
            eng.advance_batch_device(B, d_x.data_ptr(), d_x0.data_ptr(), d_tick.data_ptr(), st.cuda_stream)

    loop(True)  # warm-up pass (also the recorded one: the timed pass below does nothing but solve and advance)
    barrier()
    t0 = time.perf_counter()
    loop(False)
    barrier()
    el_ticks = time.perf_counter() - t0
    # the same loop in ONE launch (cilqr_closed_loop_batch_device): every ego's ticks back to back on the block that picked
    # it up, no synchronisation between egos — the tick-by-tick loop ends every tick with its slowest solves on a mostly
    # idle chip, this one ends once
    d_states = torch.zeros((B, ticks, 4), dtype=torch.float64, device=dev)
    d_fits = torch.zeros((ticks, B), dtype=torch.int32, device=dev)

    def fused():
        d_x0 = torch.from_numpy(x0).to(dev)
        d_tick = torch.zeros(B, dtype=torch.int32, device=dev)
        eng.closed_loop_batch_device(B, ticks, d_x0.data_ptr(), 0, 0, d_tick.data_ptr(), 0, d_u.data_ptr(), d_x.data_ptr(),
                                     d_res.data_ptr(), d_states.data_ptr(), d_fits.data_ptr(), st.cuda_stream)

    fused()
    barrier()
    t0 = time.perf_counter()
    fused()
    barrier()
    el = time.perf_counter() - t0
    fused_same = bool((d_fits.cpu().numpy() == d_its.cpu().numpy()).all()) and bool(
        np.array_equal(d_states[:check_egos].cpu().numpy()[:, :, :], np.stack([h[:, 1] for h in d_hist.cpu().numpy()], 1)))
    its = d_its.cpu().numpy()
    hist = d_hist.cpu().numpy()
    eng.close()
    return {"elapsed": el, "elapsed_tick_by_tick": el_ticks, "fused_equals_tick_by_tick": fused_same,
            "iters_per_tick": its.sum(axis=1), "iters_total": float(its.sum()), "ticks": ticks, "B": B, "N": N,
            "params": p, "scenario": sc, "x0": x0, "hist": hist, "M": int(sc.obstacles.shape[0])}


def closed_loop_check(cl, check_egos=8):
    """the sampled egos against one stateful oracle solver each (detmath build: the comparison is with ==)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle, Scene
    orc = Oracle("det")
    sc, p = cl["scenario"], cl["params"]
    same = total = 0
    for b in range(min(check_egos, cl["hist"].shape[1])):
        s = orc.solver(p)
        s.reset()
        x = cl["x0"][b].copy()
        for t in range(cl["ticks"]):
            scene = Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, t)
            r = s.solve(x, scene)
            same += int(np.array_equal(r["x"], cl["hist"][t, b]))
            total += 1
            x = r["x"][1].copy()
    return {"sampled_egos": int(min(check_egos, cl["hist"].shape[1])), "ego_ticks_compared": total,
            "ego_ticks_bit_identical_to_det_oracle": same}


def config1_closed_loop(args):
    """BASELINE configs[0]: scenario_two_straight, single ego, horizon 50, as the reference runs it — one
    CILQRSolver instance, solve() once per 0.1 s tick, ego <- row 1 of the plan (motion_planning.cpp:178-197) —
    through the drop-in class (B = 1 per call, host buffers in and out: the PCIe-inclusive path)."""
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the CILQR solve path has no CPU fallback")
    import cilqr_amd as pkg
    cfg = pkg.GlobalConfig.get_instance("two_straight")
    sc = pkg.build_scenario(cfg, "two_straight")
    N = args.horizon or 50
    ticks = min(args.ticks, sc.routes.shape[1] - N - 1)
    solver = pkg.CILQRSolver(cfg, N=N)
    obs_full = sc.obstacles
    x0 = sc.ego_state.copy()
    lat, its, states, kms = [], [], [x0.copy()], []
    for t in range(ticks):
        t0 = time.perf_counter()
        u, x = solver.solve(x0, sc.lane, sc.target_velocity, obs_full[:, t:], sc.road_borders)
        lat.append(time.perf_counter() - t0)
        if t == 0:
            solver._engine.set_timing(True)  # HIP events around the kernel from the second tick on
        else:
            kms.append(solver._engine.last_kernel_ms())
        its.append(int(solver.last_result["iters"]))
        x0 = x[1].copy()
        states.append(x0.copy())
    lat, its, kms = np.array(lat), np.array(its), np.array(kms)
    out = {"metric": "iLQR iterations/sec (batch x horizon)", "value": float(its[1:].sum() / lat[1:].sum()),
           "unit": "iLQR iterations/s", "n_gpus": 1, "steps": int(ticks), "warmup": 1,
           "ms_per_step": float(lat[1:].mean() * 1e3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "scenario_two_straight.yaml values, noise-free obstacle routes",
           "config": {"workload": f"config1_two_straight_single_ego_N{N}_closed_loop", "baseline_config": 1,
                      "batch_per_gpu": 1, "horizon": N, "ticks": int(ticks),
                      "parallelism": "one ego, one wavefront (+ helper wavefront)"},
           "extra": {"us_per_iteration_gpu": float(lat[1:].sum() / its[1:].sum() * 1e6),
                     "solve_latency_ms": {"first_tick_incl_setup": float(lat[0] * 1e3), "mean": float(lat[1:].mean() * 1e3),
                                          "p50": float(np.median(lat[1:]) * 1e3), "max": float(lat[1:].max() * 1e3)},
                     "iterations_per_tick_mean": float(its.mean()), "iterations_total": int(its.sum()),
                     "tick_split_ms": {"wall_mean": float(lat[1:].mean() * 1e3), "kernel_mean (HIP events)": float(kms.mean()),
                                       "everything_else_mean (argument comparison, pinned staging, H2D, launch, D2H, "
                                       "stream synchronisation, Python binding)": float(lat[1:].mean() * 1e3 - kms.mean())},
                     "note": "latency of the drop-in solve(): scenario tables re-used across ticks when unchanged, "
                             "host buffers in/out, one stream synchronisation per tick"}}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle, Scene
        for mode in ("libm", "det"):
            orc = Oracle(mode)
            s = orc.solver(solver.params)
            s.reset()
            scene = Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, obs_full, sc.road_borders, sc.target_velocity)
            x0 = sc.ego_state.copy()
            tl, ti, same = [], [], 0
            for t in range(ticks):
                t0 = time.perf_counter()
                r = s.solve(x0, scene, tick=t)
                tl.append(time.perf_counter() - t0)
                ti.append(int(r["res"]["iters"]))
                x0 = r["x"][1].copy()
                same += int(np.array_equal(x0, states[t + 1]))
            tl, ti = np.array(tl), np.array(ti)
            if mode == "libm":
                out["cpu_baseline"] = {"value": float(ti[1:].sum() / tl[1:].sum()), "unit": "iLQR iterations/s", "cores": 1,
                                       "kind": "port", "sample": f"the same {ticks}-tick closed loop, one thread, glibc libm",
                                       "us_per_iteration": float(tl[1:].sum() / ti[1:].sum() * 1e6),
                                       "solve_latency_ms_mean": float(tl[1:].mean() * 1e3),
                                       "iterations_total": int(ti.sum())}
            else:
                out["extra"]["closed_loop_ticks_bit_identical_to_det_oracle"] = same
    print(json.dumps(out))


def main():
    args = parse()
    if args.config == 1:
        return config1_closed_loop(args)
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
    # Rehearsal switches (no multi-GPU node has been available): CILQR_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and
    # CILQR_BENCH_BACKEND=gloo reduces the statistics through host memory, so that the N > 1 code path of this file
    # can run with two processes on a one-GPU box.  Never set by the driver; the numbers of such a run mean nothing.
    if os.environ.get("CILQR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("CILQR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("CILQR_FORCE_DIST") == "1":  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import cilqr_amd as pkg
    from importlib import import_module
    st_mod = import_module("toy-example-of-ilqr_amd.stats")
    dev = torch.device("cuda", local_rank)
    red_dev = dev if (dist is not None and backend == "nccl") else None  # where the statistics are reduced

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # which GPU does each rank drive?  Gathered before anything is timed: a multi-GPU line must show N distinct devices
    ranks = gather_objects(dist, dict(rank=rank, **device_identity(torch, local_rank)), world)
    distinct = len({(r["pci_bus_id"], r["uuid"]) for r in ranks}) == len(ranks)
    if world > 1 and not distinct and os.environ.get("CILQR_BENCH_ONE_DEVICE") != "1":
        if rank == 0:
            print(json.dumps({"error": "ranks share a GPU: a scaling run needs one device per rank", "ranks": ranks}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.stdout.flush()
        os._exit(3)

    cfg_id = args.config or 5
    wl, B = make_workload(pkg, cfg_id, args.batch, args.horizon, rank)
    N = wl.N
    run = GpuRun(pkg, torch, wl, B, local_rank)
    elapsed, kernel_ms, res = run.timed(args.steps, args.warmup, barrier, in_flight=args.in_flight,
                                        seq_steps=max(3, args.steps // 4) if args.in_flight > 1 else 0)
    launch_info_main = getattr(run, "launch_info", None)
    flight_main = getattr(run, "flight", None) or None
    stats, tmax = st_mod.reduce_stats(st_mod.local_stats(res, N, wl.M_of), elapsed, dist, red_dev)
    value = stats[0] * args.steps / tmax
    # every rank's own clock and kernel time: a straggler rank (or a GPU that throttles) is visible in the line
    seq_main = (flight_main or {}).get("sequential") or {}
    per_rank = gather_objects(dist, {"rank": rank, "elapsed_s": elapsed, "kernel_ms": kernel_ms,
                                     "kernel_ms_one_batch_at_a_time": seq_main.get("kernel_ms", kernel_ms),
                                     "ms_per_step_one_batch_at_a_time": seq_main.get("ms_per_step", elapsed / args.steps * 1e3),
                                     "iterations_per_step": float(res["iters"].sum())}, world)
    gpu_u = gpu_x = None
    if rank == 0 and not args.no_cpu_baseline:
        nb = min(B, 1024)
        gpu_u, gpu_x = run.d_u[:nb].cpu().numpy(), run.d_x[:nb].cpu().numpy()
    pipelined = run.pipelined(args.steps, args.streams) if (world == 1 and args.streams > 1) else None
    run.close()

    # the other workloads of the default command: config 2 (1024 trajectories per GPU), a latency measurement, and —
    # on several GPUs — BASELINE configs[3] (8192 trajectories of horizon 100 per rank)
    def side_run(cfg, steps_side, note, alm=False, cpu_check_rows=0, in_flight=1, group_mode=None):
        wl_s, B_s = _make_workload(pkg, cfg, 0, 0, rank)
        if alm:
            wl_s = pkg.workloads.Workload(wl_s.name + "_alm", [pkg.copy_params(q, solve_type=1) for q in wl_s.params],
                                          wl_s.scenes, wl_s.x0, wl_s.scenario_id, wl_s.param_id, wl_s.tick)
        run_s = GpuRun(pkg, torch, wl_s, B_s, local_rank, group_mode=group_mode)
        el_s, kms_s, res_s = run_s.timed(steps_side, min(args.warmup, 2), barrier, in_flight=in_flight,
                                         seq_steps=max(3, steps_side // 3) if in_flight > 1 else 0)
        li_s = getattr(run_s, "launch_info", None)
        fl_s = getattr(run_s, "flight", None) or None
        st_s, tmax_s = st_mod.reduce_stats(st_mod.local_stats(res_s, wl_s.N, wl_s.M_of), el_s, dist, red_dev)
        chk = None
        if rank == 0 and cpu_check_rows and not args.no_cpu_baseline:
            nb = min(B_s, cpu_check_rows)
            chk = (run_s.d_u[:nb].cpu().numpy(), run_s.d_x[:nb].cpu().numpy(), nb)
        run_s.close()
        pr_s = gather_objects(dist, kms_s, world)
        if rank != 0:
            return None
        rl = roofline_block(pkg, wl_s, res_s, kms_s, world, li_s, fl_s)
        cpu_chk = None
        if chk is not None:  # the launch that was just timed against the oracle's libm build (checker only)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from oracle import Oracle
            nb = chk[2]
            ref = Oracle("libm").solve_batch(wl_s.params, oracle_scenes(wl_s), wl_s.x0[:nb], wl_s.scenario_id[:nb],
                                             wl_s.param_id[:nb], wl_s.tick[:nb], n_threads=args.cpu_threads or usable_cores()[0])
            cpu_chk = parity_check(chk[0], chk[1], res_s, ref)
            det = Oracle("det").solve_batch(wl_s.params, oracle_scenes(wl_s), wl_s.x0[:nb], wl_s.scenario_id[:nb],
                                            wl_s.param_id[:nb], wl_s.tick[:nb], n_threads=args.cpu_threads or usable_cores()[0])
            cpu_chk["bit_identical_to_det_oracle"] = det_identity(chk[0], chk[1], res_s, det)
            if wl_s.N >= 100:
                cpu_chk["note"] = ("horizon 100 amplifies a one-ulp difference of libm past 1e-5 on most of these solves; the "
                                   "libm build differs from ITSELF as much when x0 moves by one unit in the last place "
                                   "(DESIGN.md section 2, profiles/r03_libm_tolerance.json workload 4); the parity claim is "
                                   "the bit-identity with the oracle's detmath build in this object")
        return {"cpu_check": cpu_chk, "workload": wl_s.name, "baseline_config": cfg, "batch_per_gpu": B_s, "global_batch": int(st_s[8]),
                "horizon": wl_s.N, "steps": steps_side, "value": st_s[0] * steps_side / tmax_s, "unit": "iLQR iterations/s",
                "ms_per_step": tmax_s / steps_side * 1e3, "kernel_ms": rl["kernel_ms"],
                "kernel_ms_is": "one launch at a time (in_flight.effective.kernel_ms: the timed region's launches / its kernel time)",
                "in_flight": rl["in_flight"],
                "kernel_ms_min_max_over_ranks": [float(min(pr_s)), float(max(pr_s))],
                "iterations_per_launch_rank0": float(res_s["iters"].sum()),
                "slowest_trajectory_iterations": int(res_s["iters"].max()),
                "converged": int(st_s[2]), "max_lamb": int(st_s[3]), "max_iter": int(st_s[4]), "nan_costs": int(st_s[6]),
                "kernel": rl["kernel"], "launch": rl["launch"], "hbm_frac": rl["frac"], "traffic": rl["traffic"], "valu_issue": rl["valu_issue"], "fp64_useful": rl["fp64_useful"],
                "note": note}

    def closed_loop_extra(N_loop=50):
        cl = closed_loop_run(pkg, torch, local_rank, rank, barrier, N=N_loop)
        vec = np.array([cl["iters_total"], cl["B"] * cl["ticks"]], dtype=np.float64)
        red, tmax_c = st_mod.reduce_stats(vec, cl["elapsed"], dist, red_dev)
        _, tmax_t = st_mod.reduce_stats(vec, cl["elapsed_tick_by_tick"], dist, red_dev)
        if rank != 0:
            return None
        ipt = cl["iters_per_tick"]
        out_c = {"workload": f"closed_loop_three_straight_B{cl['B']}_N{cl['N']}_ticks{cl['ticks']}", "egos_per_gpu": cl["B"],
                 "ticks": cl["ticks"], "horizon": cl["N"], "obstacles": cl["M"],
                 "value": red[0] / tmax_c, "unit": "iLQR iterations/s", "ego_ticks_per_s": red[1] / tmax_c,
                 "ms_per_tick": tmax_c / cl["ticks"] * 1e3, "timed_region_s": tmax_c,
                 "tick_by_tick": {"value": red[0] / tmax_t, "ego_ticks_per_s": red[1] / tmax_t, "ms_per_tick": tmax_t / cl["ticks"] * 1e3,
                                  "what": "one cilqr_solve_batch_device + one cilqr_advance_batch_device per tick"},
                 "one_launch_equals_tick_by_tick (iterations of every ego-tick, sampled states)": cl["fused_equals_tick_by_tick"],
                 "iterations_per_ego_tick_mean": float(cl["iters_total"] / (cl["B"] * cl["ticks"])),
                 "iterations_per_ego_first_tick_cold": float(ipt[0] / cl["B"]),
                 "iterations_per_ego_later_ticks_warm": float(ipt[1:].sum() / (cl["B"] * max(1, cl["ticks"] - 1))),
                 "note": "value = cilqr_closed_loop_batch_device: the whole loop in one launch, every ego's ticks back to back "
                         "(solve, ego <- x.row(1), tick + 1, warm start from the plan just made: mp:180-197, cs:163-180), no "
                         "host round trip and no synchronisation between egos"}
        if not args.no_cpu_baseline:
            out_c["cpu_check"] = closed_loop_check(cl)
        return out_c

    # Every extra is wrapped: the headline line must survive a failing side measurement.  (The extras run the same code
    # on every rank, so a failure is the same failure everywhere and the ranks stay in step; a rank that failed alone
    # would leave the others in a collective until the process group times out.)
    def guarded(fn, *a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:  # noqa: BLE001 - reported in the line
            return {"error": f"{type(e).__name__}: {e}"[:500]} if rank == 0 else None

    # Round 6's kernels have never run on hardware (the GPU pool was closed all round; they are bit-exact on the wave64 emulator of
    # the CPU test suite).  A side measurement on such a kernel runs in a PROCESS OF ITS OWN with a time limit: a fault or a hang
    # there — which no `except` catches — costs that entry, not the line.
    ALM_PAIRS = ("the headline batch with solve_type alm IN PAIRS per wavefront (cilqr_set_group_mode(2): the grouped kernel's long "
                 "layout with dense rows; written without a GPU, bit-exact on the wave64 emulator of the CPU test suite)")
    if args.only_extra == "alm_pairs":
        r = guarded(side_run, 5, max(3, args.steps // 4), ALM_PAIRS, alm=True, cpu_check_rows=1024, group_mode=2)
        if rank == 0:
            print("ONLY-EXTRA " + json.dumps(r), flush=True)
        return

    def isolated_extra(name, limit_s):
        if world != 1:
            return {"skipped": "measured at N = 1 only (it runs in a process of its own)"} if rank == 0 else None
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--only-extra", name, "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--cpu-threads", str(args.cpu_threads)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s * (20 if REHEARSAL > 1 else 1))
        got = [ln for ln in r.stdout.splitlines() if ln.startswith("ONLY-EXTRA ")]
        if r.returncode != 0 or not got:
            return {"error": f"the child process ended with code {r.returncode}: " + (r.stderr or r.stdout)[-400:]}
        out_x = json.loads(got[-1][len("ONLY-EXTRA "):])
        if isinstance(out_x, dict):
            out_x["measured_in"] = "a process of its own (a kernel without hardware history cannot take the line down)"
        return out_x

    second = third = fourth = closed = closed30 = alm5 = alm5p = None
    if not args.no_extras and args.config == 0 and not args.batch and not args.horizon:
        third = guarded(side_run, 3, max(args.steps, 20), "BASELINE configs[2]: 8192 three_bend trajectories = two rounds of the "
                        "chip's trajectory slots; one launch at a time a third of it is tail (`in_flight.sequential`)",
                        in_flight=args.in_flight)
        second = guarded(side_run, 2, max(args.steps, 20), "one launch of 1024 trajectories occupies a quarter of the chip's "
                         "wave slots; its wall time is the slowest trajectory's (DESIGN.md)")
        fourth = guarded(side_run, 4, max(3, args.steps // 4), "BASELINE configs[3]: 65 536 mixed scenarios of horizon 100 "
                         "sharded 8 x 8192; every rank solves 8192 (the full configuration at 8 GPUs; at fewer, the "
                         "first ranks' shards)", cpu_check_rows=512, in_flight=args.in_flight)
        closed = guarded(closed_loop_extra)
        closed30 = guarded(closed_loop_extra, 30)  # the horizon the reference's own YAMLs use (config/scenario_*.yaml:5)
        alm5 = guarded(side_run, 5, max(3, args.steps // 4), "the headline batch with solve_type alm (cs:88-93, 253-261, "
                       "581-643): multipliers [B][N][8 + 2M] in HBM, kept by the handle across calls", alm=True,
                       cpu_check_rows=1024)
        # round 6: the same batch on the grouped kernel's ALM builds (two trajectories per wavefront; opt-in until measured here)
        alm5p = guarded(isolated_extra, "alm_pairs", 600)

    if rank == 0:
        my_iters = float(res["iters"].sum())
        out = {
            "metric": "iLQR iterations/sec (batch x horizon)", "value": value, "unit": "iLQR iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if REHEARSAL == 1 else f"REHEARSAL (batches / {REHEARSAL}): not a measurement",
            **({"rehearsal": REHEARSAL} if REHEARSAL > 1 else {}),
            # the two ways of running the same K steps, by name: `value` is the first — the timed region of this
            # command; the second is the sequential leg timed right after it (rounds 1-4 reported that one)
            "value_is": (f"value_batches_in_flight_{args.in_flight}" if args.in_flight > 1 else "value_one_batch_at_a_time"),
            f"value_batches_in_flight_{max(1, args.in_flight)}": value,
            "value_one_batch_at_a_time": float(sum(r["iterations_per_step"] for r in per_rank)
                                               / (max(r["ms_per_step_one_batch_at_a_time"] for r in per_rank) * 1e-3)),
            "kernel_ms_one_batch_at_a_time": float(max(r["kernel_ms_one_batch_at_a_time"] for r in per_rank)),
            "config": {"workload": wl.name, "baseline_config": cfg_id, "batch_per_gpu": B, "batches_in_flight": args.in_flight,
                       "global_batch": int(stats[8]), "horizon": N, "nx": 4, "nu": 2, "library": pkg.library_info(),
                       "parallelism": f"trajectory-sharded x{world}, "
                                      + {1: "one wavefront per trajectory", 2: "two trajectories per wavefront",
                                         3: "three trajectories per wavefront"}.get(
                                             (launch_info_main or {}).get("trajectories_per_wavefront", 1), "one wavefront per trajectory")},
            "roofline": roofline_block(pkg, wl, res, kernel_ms, world, launch_info_main, flight_main),
            "extra": {"iterations_per_step_rank0": my_iters, "iterations_per_solve_mean": my_iters / B,
                      "line_search_trials_per_step": float(stats[1]),
                      "solves_per_s": stats[8] * args.steps / tmax,
                      "step_updates_per_s": value * N,
                      "timed_region_s": tmax,
                      "converged": int(stats[2]), "max_lamb": int(stats[3]), "max_iter": int(stats[4]),
                      "nan_costs": int(stats[6]), "sum_J_final": float(stats[5]), "pipelined": pipelined,
                      ("config2_latency" if world == 1 else "config2_weak_scaling"): second,
                      "config3": third, "config4_sharded": fourth, "closed_loop": closed, "closed_loop_N30": closed30, "config5_alm": alm5, "config5_alm_pairs": alm5p,
                      "ranks": ranks, "distinct_devices": distinct, "per_rank": per_rank,
                      "kernel_ms_min_max_over_ranks": [float(min(r["kernel_ms"] for r in per_rank)),
                                                       float(max(r["kernel_ms"] for r in per_rank))]},
        }
        if cfg_id == 5:
            # convergence-vs-throughput view of the sweep: per barrier setting, over its 4096 solves
            pid = wl.param_id
            out["extra"]["per_setting"] = [
                {"obstacle_exp_q1": float(wl.params[s].obstacle_exp_q1), "obstacle_exp_q2": float(wl.params[s].obstacle_exp_q2),
                 "iterations_mean": float(res["iters"][pid == s].mean()),
                 "converged_frac": float((res["end_reason"][pid == s] == 0).mean()),
                 "J_final_median": float(np.nanmedian(res["J_final"][pid == s]))} for s in range(len(wl.params))]
        if not args.no_cpu_baseline:
            cores, cores_note = usable_cores()
            threads = args.cpu_threads or cores
            try:
                cb, r = cpu_baseline(pkg, wl, threads)
                if cores_note:
                    cb["cores_note"] = cores_note
                out["cpu_baseline"] = cb
                out["extra"]["cpu_check"] = parity_check(gpu_u, gpu_x, res, r)
                from oracle import Oracle  # checker only
                nb = r["res"].shape[0]
                det = Oracle("det").solve_batch(wl.params, oracle_scenes(wl), wl.x0[:nb], wl.scenario_id[:nb], wl.param_id[:nb],
                                                wl.tick[:nb], n_threads=threads)
                out["extra"]["cpu_check"]["bit_identical_to_det_oracle"] = det_identity(gpu_u, gpu_x, res, det)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        # RCCL keeps a version banner in the C library's stdout buffer and emits it at exit, after the JSON
        # line; leave without running the C exit handlers so that the line above stays the only output
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
