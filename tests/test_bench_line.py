"""CPU checks of bench.py's bookkeeping (no GPU, no solve): which duration the roofline object is computed from, and that the
stamp of the counter passes names the device code it was collected on."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _fake():
    import cilqr_amd as pkg
    N, B = 50, 64
    res = np.zeros(B, dtype=pkg.RESULT_DTYPE)
    res["iters"] = 20
    res["ls_trials"] = 60
    wl = types.SimpleNamespace(N=N, M_of=np.full(B, 3), name="config5_sweep_B4096x16_N50")
    return pkg, wl, res


def test_roofline_uses_the_one_at_a_time_duration():
    """ADVICE r05 / VERDICT r05 task 1: with batches in flight the roofline's achieved / frac come from the sequential leg (a
    clean kernel duration, the mode the counter passes ran in); the overlapped region's figure sits under in_flight.effective."""
    pkg, wl, res = _fake()
    alg = float((res["iters"] * pkg.workloads.bytes_per_iteration(wl.N, wl.M_of)).sum())
    flight = {"in_flight": 3, "steps": 20, "kernel_region_ms": 20 * 57.0, "sequential": {"steps": 5, "kernel_ms": 59.7, "ms_per_step": 59.9}}
    r = bench.roofline_block(pkg, wl, res, 57.0, 1, {"trajectories_per_wavefront": 2}, flight)
    assert r["kernel_ms"] == 59.7
    assert abs(r["achieved"] - alg / 59.7e-3 / 1e9) < 1e-9 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-15
    eff = r["in_flight"]["effective"]
    assert abs(eff["kernel_ms"] - 57.0) < 1e-12 and eff["frac"] > r["frac"]
    assert r["kernel"] == "k_solve_grp" and r["bound"] == "hbm" and r["unit"] == "GB/s"
    # one launch at a time: the duration handed in is the one used, nothing under `effective`
    r1 = bench.roofline_block(pkg, wl, res, 60.1, 1, None, None)
    assert r1["kernel_ms"] == 60.1 and r1["in_flight"] == {"in_flight": 1}
    json.dumps(r), json.dumps(r1)  # serialisable


def test_counter_stamp_names_the_device_code():
    """profiles/pmc_current.json carries, next to the source fingerprint, the manifest of the machine code it was collected on;
    the shipped library is compared with it function by function (host-side edits of csrc/ move the source stamp only)."""
    meta = json.load(open(bench.PMC_FILE))["_collected"]
    man = meta.get("device_code_manifest")
    assert man and os.path.exists(os.path.join(ROOT, man))
    lib = os.path.join(ROOT, "toy-example-of-ilqr_amd", "libcilqr_amd.so")
    if not (os.path.exists(lib) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")):
        import pytest
        pytest.skip("needs the built library and hipcc's LLVM tools")
    ans = bench.device_code_unchanged(man)
    assert "error" not in ans, ans
    assert ans["functions_same"] + ans["functions_changed"] + ans["functions_missing"] > 50
    # the flag bench.py prints must follow the comparison, whichever way it goes
    assert ans["everything_unchanged"] == (ans["functions_changed"] == 0 and ans["functions_missing"] == 0)
    pkg, wl, res = _fake()
    r = bench.roofline_block(pkg, wl, res, 59.7, 1, None, None)
    assert r["traffic_collected"]["device_code"] == ans
    assert r["traffic_collected"]["stale"] == (meta["csrc_sha16"] != bench.csrc_fingerprint())


def test_bench_command_walks_through_on_the_emulator():
    """bench.py end to end — workload in "device" memory, warm-up, the timed region with three batches in flight, the sequential leg,
    statistics, roofline object, the line — REHEARSED on the wave64 emulator of tests/emu/ through a numpy-backed stand-in for torch
    (scripts/emu_rehearse.py): the driver's contract keys are there and consistent.  Numbers of such a run mean nothing; what is
    checked is that the command the driver runs does not first meet its own code on the GPU box."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "emu_rehearse.py"), "--", sys.executable, os.path.join(ROOT, "bench.py"),
                        "--config", "3", "--batch", "6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "value_one_batch_at_a_time", "kernel_ms_one_batch_at_a_time"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["warmup"] == 1 and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert b["config"]["workload"] == "config3_bend_B6_N50" and b["config"]["batches_in_flight"] == 3 and "model" not in b["config"]
    rl = b["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
    assert rl["in_flight"]["buffer_sets_identical"] is True and rl["in_flight"]["sequential"]["results_identical_to_in_flight"] is True
    assert rl["kernel_ms"] == rl["in_flight"]["sequential"]["kernel_ms"] == b["kernel_ms_one_batch_at_a_time"]
    its = b["extra"]["iterations_per_step_rank0"]
    assert abs(b["value"] - its * b["steps"] / b["extra"]["timed_region_s"]) < 1e-6 * b["value"]
    assert abs(rl["algorithmic_bytes_per_launch"] - its * 8536) < 1e-6


def test_bench_multi_rank_path_walks_through_on_the_emulator():
    """bench.py --gpus 2 as the driver launches it (one process per device: RANK / LOCAL_RANK / WORLD_SIZE) — no multi-GPU node has
    been available in six rounds, so the N > 1 code path is REHEARSED: two processes, one emulated device each, the statistics
    exchanged by a file-based stand-in for torch.distributed.  Each rank solves its own shard (rows first = rank * B of one global
    batch), rank 0 prints ONE line: n_gpus 2, the global batch and iteration count of both shards, distinct devices, the slowest
    rank's clock.  (tests/test_distributed.py covers the reduction itself over gloo.)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "emu_rehearse.py"), "--ranks", "2", "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "3", "--batch", "5", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines            # rank 0 alone prints
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["config"]["global_batch"] == 10 and b["config"]["batch_per_gpu"] == 5
    e = b["extra"]
    assert e["distinct_devices"] is True and len(e["ranks"]) == 2 and len({d["pci_bus_id"] for d in e["ranks"]}) == 2
    assert [p["rank"] for p in e["per_rank"]] == [0, 1]
    # the two shards are different rows of one global batch: their iteration counts differ, the line carries the sum
    its = [p["iterations_per_step"] for p in e["per_rank"]]
    assert its[0] != its[1] and its[0] == e["iterations_per_step_rank0"]
    assert abs(b["value"] - sum(its) * b["steps"] / e["timed_region_s"]) < 1e-6 * b["value"]
    assert e["timed_region_s"] >= max(p["elapsed_s"] for p in e["per_rank"]) - 1e-9
    # ... and equal to what one process computes for the same rows (shard offsets: first = rank * B)
    sys.path.insert(0, ROOT)
    import cilqr_amd as pkg
    from oracle import Oracle, Scene
    for rank in (0, 1):
        wl = pkg.workloads.config3(B=5, N=50, first=rank * 5)
        ref = Oracle("det").solve_batch(wl.params, [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes],
                                       wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=2)
        assert float(ref["res"]["iters"].sum()) == its[rank], (rank, its)
