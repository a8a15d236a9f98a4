"""The kernels' wave64 LOGIC, run without a GPU.  TEST INFRASTRUCTURE (tests/emu/): the unmodified sources of
toy-example-of-ilqr_amd/csrc/ are compiled for the x86 host against a stand-in <hip/hip_runtime.h> and an emulator in which every
lane of every resident block is a fibre and every cross-lane operation (DPP, ds_bpermute, v_readlane, v_readfirstlane, ballot,
barriers) is resolved for the lanes that execute it together; raw buffers keep their range check, the buffer -> LDS DMA, the
persistent blocks, the spin waits and the hand-over protocol between blocks behave as on the device.  The resulting library has
the product's C-ABI and is driven through the product's own Python binding (CILQR_AMD_LIB names it, in a subprocess).

What this checks: that the SOURCE the GPU library is built from computes the reference's numbers — every output == the oracle's
detmath build, bit for bit — for the kernels that matter, on the build container, every round, whether or not a GPU can be had.
What it cannot check: what hipcc makes of the source, and the hardware (timing, LDS bank conflicts, the lost-store anomaly of
round 4): the `-m gpu` suite stays the parity gate.  The product never loads the emulator (test below)."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
EMU = ROOT / "tests" / "emu"

PRELUDE = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import cilqr_amd as pkg
from oracle import Oracle, Scene
assert "tests/emu/_build" in str(pkg._lib.LIB_PATH), pkg._lib.LIB_PATH
ORC = Oracle("det")
EMULIB = ctypes.CDLL(str(pkg._lib.LIB_PATH))
def scene_of(sc, tick=0): return Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, tick)
def scenes_of(wl): return [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes]
def bits(a): return np.ascontiguousarray(a).view(np.uint64)
def same(out, ref):
    ok = bool((bits(out["u"]) == bits(ref["u"])).all() and (bits(out["x"]) == bits(ref["x"])).all())
    for f in ("iters", "end_reason", "final_status", "ls_trials", "cost_evals"):
        ok = ok and bool((out["res"][f] == ref["res"][f]).all())
    return ok and bool((bits(out["res"]["J_final"]) == bits(ref["res"]["J_final"])).all() and (bits(out["res"]["J_init"]) == bits(ref["res"]["J_init"])).all())
def emu_stats():
    st = (ctypes.c_longlong * 8)(); EMULIB.cilqr_emu_stats(st)
    return dict(launches=st[0], lane_ops=st[1], groups=st[2], split_resolutions=st[3], partial_barriers=st[4], readlane_of_inactive_lane=st[5])
def scenario(name, N, **kw):
    cfg = pkg.GlobalConfig.get_instance(name); sc = pkg.build_scenario(cfg, name)
    return sc, pkg.params_from_config(cfg, N=N, **kw)
OUT = {}
"""


@pytest.fixture(scope="session")
def emu_libs(built):
    sys.path.insert(0, str(EMU))
    import build_emu
    if os.environ.get("CILQR_EMU_SANITIZE_ALL") == "1":   # every emulator test on the ASan + UBSan build (a one-off sweep, slow)
        lib = build_emu.build(dev=True, sanitize=True)
        return {"prod": lib, "dev": lib}
    if os.environ.get("CILQR_EMU_COVERAGE") == "1":   # scripts/emu_coverage.py: the same tests on the block-coverage build
        lib = build_emu.build(dev=True, coverage=True)
        return {"prod": lib, "dev": lib}
    return {"prod": build_emu.build(), "dev": build_emu.build(dev=True)}


def run(emu_libs, body, timeout=600, env=None, libs=None):
    libs = libs or emu_libs
    e = dict(os.environ)
    e.pop("CILQR_TUNE", None)
    e.update({"CILQR_AMD_LIB": str(libs["prod"]), "CILQR_AMD_LIB_DEV": str(libs["dev"])})
    e.update(env or {})
    if os.environ.get("CILQR_EMU_SANITIZE_ALL") == "1":
        import glob
        e.update({"LD_PRELOAD": sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))[-1],
                  "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=0:detect_stack_use_after_return=0:exitcode=0:log_path=" +
                                  str(ROOT / "scratch" / "asan_sweep"),
                  "UBSAN_OPTIONS": "print_stacktrace=1:log_path=" + str(ROOT / "scratch" / "ubsan_sweep")})
    r = subprocess.run([sys.executable, "-c", PRELUDE + body + "\nprint('EMU-RESULT ' + json.dumps(OUT))\n", str(ROOT)],
                       capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("EMU-RESULT ")][-1]
    return json.loads(line[len("EMU-RESULT "):])


def healthy(st):
    """the emulator's own bookkeeping: no barrier met by part of a wavefront while the rest waited elsewhere, no v_readlane of a
    lane that was not executing (either would mean the run did not follow the device's semantics)"""
    assert st["partial_barriers"] == 0 and st["readlane_of_inactive_lane"] == 0, st
    assert st["lane_ops"] > 100000, st


def test_the_product_never_touches_the_emulator(pkg):
    """tests/emu/ is loaded by this file alone: nothing under the package, include/, examples/, bench.py or __graft_entry__.py
    names it, the default library path is the gfx950 build, and the emulator library is not built by build()."""
    for p in list((ROOT / "toy-example-of-ilqr_amd").rglob("*.py")) + list((ROOT / "toy-example-of-ilqr_amd" / "csrc").iterdir()) + \
            list((ROOT / "include").iterdir()) + list((ROOT / "examples").glob("*.cpp")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]:
        if p.is_file() and p.suffix != ".so":
            t = p.read_text(errors="ignore")
            assert "tests/emu" not in t and "libcilqr_emu" not in t and "CILQR_EMULATED" not in t, p
    if not os.environ.get("CILQR_AMD_LIB"):
        assert pkg._lib.LIB_PATH == ROOT / "toy-example-of-ilqr_amd" / "libcilqr_amd.so"


def test_pairs_per_wavefront_with_hand_overs_and_slices(emu_libs):
    """k_solve_grp, short layout (the headline's kernel family): 40 three_bend solves of horizon 30 on 8 resident blocks — two
    rounds of trajectories, pair sweeps, paired costs, sliced solves, idle wavefronts taking trajectories over at the tail — and
    24 of horizon 50 (the compile-time build); every output and the whole decision trace == oracle."""
    r = run(emu_libs, r"""
for N, B in ((30, 40), (50, 24)):
    sc, p = scenario("three_bend", N)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc)); eng.set_group_mode(2)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0x5A0CE)
    out = eng.solve_batch(x0, trace_cap=128)
    ref = [ORC.solver(p) for _ in range(B)]
    whole = ORC.solve_batch(p, scene_of(sc), x0, n_threads=4)
    tr_ok = True
    for b in range(B):
        s = ref[b]; s.reset(); rr = s.solve(x0[b], scene_of(sc), trace_cap=128)
        n = int(out["res"]["trace_len"][b])
        tr_ok = tr_ok and n == len(rr["trace"]) and bool((out["trace"][b][:n] == rr["trace"]).all())
    OUT[str(N)] = dict(same=same(out, whole), traces=tr_ok, launch=eng.last_launch_info(), parked=eng.resume_stats(),
                       error=eng.work_sharing_stats()["error"], iters_max=int(out["res"]["iters"].max()))
    eng.close()
OUT["stats"] = emu_stats()
""")
    for N in ("30", "50"):
        assert r[N]["same"] and r[N]["traces"], r[N]
        assert r[N]["launch"]["trajectories_per_wavefront"] == 2 and r[N]["error"] == 0, r[N]
    assert r["30"]["parked"] > 0, r["30"]   # (trajectories did change wavefronts)
    healthy(r["stats"])


def test_long_layout_and_lone_builds(emu_libs):
    """horizon 100: the grouped kernel's long layout (rows streamed through LDS rings, gains ring filled by LDS-DMA behind a counted
    wait, two rows per lane) on the mixed scenarios of configs[3], and the lone / helper builds of k_solve on the same batch;
    straight-lane starts (RearCenter model) of config 2 through the helper build and in pairs."""
    r = run(emu_libs, r"""
wl = pkg.workloads.config4(B=8, N=100)
ref = ORC.solve_batch(wl.params, scenes_of(wl), wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=4)
for mode in (2, 0, -1):
    eng = pkg.BatchedCILQR(wl.params, wl.scenes); eng.set_group_mode(mode)
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    OUT["c4 mode %d" % mode] = dict(same=same(out, ref), launch=eng.last_launch_info())
    eng.close()
wl = pkg.workloads.config2(B=10, N=50)
ref = ORC.solve_batch(wl.params, scenes_of(wl), wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=4)
for mode in (2, -1):
    eng = pkg.BatchedCILQR(wl.params, wl.scenes); eng.set_group_mode(mode)
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    OUT["c2 mode %d" % mode] = dict(same=same(out, ref), launch=eng.last_launch_info())
    eng.close()
OUT["stats"] = emu_stats()
""", timeout=900)
    for k, v in r.items():
        if k != "stats":
            assert v["same"], (k, v)
    assert r["c4 mode 2"]["launch"]["trajectories_per_wavefront"] == 2 and r["c4 mode 0"]["launch"]["trajectories_per_wavefront"] == 1
    assert r["c2 mode -1"]["launch"]["threads_per_block"] == 128   # (the helper build)
    healthy(r["stats"])


def test_smoke_entry_point_on_the_emulator(emu_libs):
    """__graft_entry__.smoke() — what the driver runs on the GPU box before the bench — rehearsed with the library swapped."""
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "emu_rehearse.py"), "--", sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"],
                       capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "smoke ok: 8 solves" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


def test_serial_reference_chain_forced(emu_libs):
    """The serial first-local-minimum chain (cs:289-314 as written) is the fallback of the lane-parallel reference search and
    rarely runs; the development build forces it (a testing aid).  scripts/emu_mutants.py showed that no other emulator test
    reaches it: a planted off-by-one in the chain survived until this case was added."""
    r = run(emu_libs, r"""
sc, p = scenario("three_bend", 30)
x0 = pkg.workloads.perturbed_starts(sc.ego_state, 4, 0x5E21A1)
ref = ORC.solve_batch(p, scene_of(sc), x0, n_threads=2)
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True); eng.set_group_mode(0)
eng.set_debug_flags(pkg._lib.DBG_SERIAL_REF_SCAN)
OUT["same"] = same(eng.solve_batch(x0), ref); OUT["launch"] = eng.last_launch_info()
eng.close()
""")
    assert r["same"], r


def test_lone_long_horizon_builds_share_work_and_resume(emu_libs):
    """k_solve's two-row build with the expansion in global rows (what runs at horizons 64 ... 127 when pairs are switched off):
    four persistent blocks for 14 trajectories of the configs[3] mix — finished blocks cost open line-search trials of running
    ones (work sharing between blocks), solves run in slices and are parked / resumed by whoever is free (resumable solves).
    Whatever the slice length, == oracle; the counters say the machinery was used."""
    r = run(emu_libs, r"""
wl = pkg.workloads.config4(B=14, N=100)
ref = ORC.solve_batch(wl.params, scenes_of(wl), wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=4)
eng = pkg.BatchedCILQR(wl.params, wl.scenes); eng.set_group_mode(0); eng.set_helper_mode(0)
for res in (-1, 3, 0):
    eng.set_resume_iters(res)
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    OUT["slice %d" % res] = dict(same=same(out, ref), launch=eng.last_launch_info(), parked=eng.resume_stats(), share=eng.work_sharing_stats())
eng.close()
OUT["stats"] = emu_stats()
""", timeout=900, env={"CILQR_EMU_BLOCKS_PER_CU": "4"})
    healthy(r.pop("stats"))
    for k, v in r.items():
        assert v["same"] and v["launch"]["trajectories_per_wavefront"] == 1 and v["launch"]["blocks"] == 4 and v["share"]["error"] == 0, (k, v)
    assert r["slice 3"]["parked"] > 20 and r["slice 0"]["parked"] == 0 and r["slice 0"]["share"]["helped"] > 0, r


def test_augmented_lagrangian_lone_and_in_pairs(emu_libs):
    """solve_type alm (cs:88-93, 253-277, 377-378, 581-643): k_solve's builds (what the default dispatch runs) and — round 6, written
    while the GPU pool was closed, never run on a GPU — the grouped kernel's ALM builds, two trajectories per wavefront, chosen with
    cilqr_set_group_mode(2).  Horizons 30 and 100; a second call warm-started from the first one's plan continues from the
    multipliers the handle kept (hpp:106-112), against stateful oracle solvers."""
    r = run(emu_libs, r"""
for N, B in ((30, 8), (100, 4)):
    sc, p = scenario("three_bend", N, solve_type=1, use_last_solution=1)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xA11)
    refs = [ORC.solver(p) for _ in range(B)]
    first, second = [], []
    for b in range(B):
        refs[b].reset()
        a = refs[b].solve(x0[b], scene_of(sc))
        first.append(a)
        second.append(refs[b].solve(a["x"][1], scene_of(sc, 1)))
    stack = lambda rs: dict(u=np.stack([q["u"] for q in rs]), x=np.stack([q["x"] for q in rs]), res=np.array([q["res"] for q in rs]).reshape(-1))
    for mode in (-1, 2):
        eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc)); eng.set_group_mode(mode)
        o1 = eng.solve_batch(x0)
        info = eng.last_launch_info()
        o2 = eng.solve_batch(o1["x"][:, 1].copy(), tick=np.ones(B, dtype=np.int32), last_u=o1["u"])
        OUT["N%d mode %d" % (N, mode)] = dict(first=same(o1, stack(first)), second=same(o2, stack(second)), launch=info,
                                              iters=o1["res"]["iters"].tolist())
        eng.close()
OUT["stats"] = emu_stats()
""", timeout=900)
    for k, v in r.items():
        if k != "stats":
            assert v["first"] and v["second"], (k, v)
            assert v["launch"]["trajectories_per_wavefront"] == (2 if k.endswith("mode 2") else 1), (k, v)
    healthy(r["stats"])


def test_horizons_of_128_to_255(emu_libs):
    """Round 6 (VERDICT r05 task 9; cs:19): the long layout with four rows per lane — written without a GPU.  Horizons 128 (the
    first with four rows), 200 and 255 (the cap), both vehicle models, both solve types == oracle; the entry points that have no
    build at these horizons refuse."""
    r = run(emu_libs, r"""
for name, N in (("two_straight", 255), ("three_bend", 200), ("two_borrow", 128)):
    cfg = pkg.GlobalConfig.get_instance(name); sc = pkg.build_scenario(cfg, name)
    obs = np.concatenate([sc.obstacles, np.repeat(sc.obstacles[:, -1:, :], 120, axis=1)], axis=1)
    tab = pkg.SceneTable(sc.lane.x, sc.lane.y, sc.lane.yaw, obs, sc.road_borders, sc.target_velocity)
    scene = Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, obs, sc.road_borders, sc.target_velocity)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 3, 0x255)
    for st in (0, 1):
        p = pkg.params_from_config(cfg, N=N, solve_type=st, max_iter=20)
        eng = pkg.BatchedCILQR(p, tab)
        out = eng.solve_batch(x0)
        ref = ORC.solve_batch(p, scene, x0, n_threads=4)
        refuses = 0
        for call in (lambda: eng.total_cost(out["u"][:1], out["x"][:1]), lambda: eng.init_traj(x0[:1])):
            try: call()
            except pkg.CilqrError as e: refuses += int(e.code == -4)
        OUT["%s N=%d type %d" % (name, N, st)] = dict(same=same(out, ref), launch=eng.last_launch_info(), refuses=refuses,
                                                     iters=out["res"]["iters"].tolist())
        eng.close()
try:
    pkg.BatchedCILQR(pkg.params_from_config(cfg, N=256), tab); OUT["N=256 accepted"] = True
except pkg.CilqrError:
    OUT["N=256 accepted"] = False
OUT["stats"] = emu_stats()
""", timeout=1200)
    assert r.pop("N=256 accepted") is False
    healthy(r.pop("stats"))
    assert len(r) == 6
    for k, v in r.items():
        assert v["same"] and v["launch"]["trajectories_per_wavefront"] == 2 and v["refuses"] == 2, (k, v)


def test_closed_loop_in_one_launch(emu_libs):
    """cilqr_closed_loop_batch_device (mp:180-197 for every ego in one launch) through the device-pointer entry point — under the
    emulator "device memory" is host memory, so numpy buffers stand in.  Horizon 30: 6 egos x 4 ticks in pairs and on lone
    wavefronts; round 6 (never run on a GPU): the loop on the grouped kernel's LONG layout — horizon 100 in pairs (opt-in) and
    horizon 150 (four rows per lane, the only build there).  Ego states and iteration counts of every tick and the last tick's
    plan == stateful oracle solvers; under the augmented Lagrangian horizons above 127 refuse."""
    r = run(emu_libs, r"""
ptr = lambda a: a.ctypes.data
def loop(N, B, T, mode, st=0):
    sc, p = scenario("three_straight", N, use_last_solution=1, max_iter=40, solve_type=st)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xC10)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc)); eng.set_group_mode(mode)
    dx0 = x0.copy(); tick = np.zeros(B, dtype=np.int32)
    u = np.zeros((B, N, 2)); x = np.zeros((B, N + 1, 4)); res = np.zeros(B, dtype=pkg.RESULT_DTYPE)
    states = np.zeros((B, T, 4)); its = np.zeros((T, B), dtype=np.int32)
    try:
        eng.closed_loop_batch_device(B, T, ptr(dx0), 0, 0, ptr(tick), 0, ptr(u), ptr(x), ptr(res), ptr(states), ptr(its), 0)
        eng.wait()
    except pkg.CilqrError as e:
        eng.close(); return dict(refused=e.code)
    ok = True
    for b in range(B):
        s = ORC.solver(p); s.reset(); xe = x0[b].copy()
        for t in range(T):
            rr = s.solve(xe, scene_of(sc, t))
            xe = rr["x"][1].copy()
            ok = ok and bool((bits(xe) == bits(states[b, t])).all()) and int(rr["res"]["iters"]) == int(its[t, b])
        ok = ok and bool((bits(rr["x"]) == bits(x[b])).all() and (bits(rr["u"]) == bits(u[b])).all())
    out = dict(same=ok, launch=eng.last_launch_info(), ticks=tick.tolist())
    eng.close()
    return out
OUT["N30 pairs"] = loop(30, 6, 4, 2); OUT["N30 lone"] = loop(30, 6, 4, 0)
OUT["N100 pairs"] = loop(100, 5, 3, 2); OUT["N100 default"] = loop(100, 3, 2, -1)
OUT["N150"] = loop(150, 3, 2, -1); OUT["N150 alm"] = loop(150, 2, 2, -1, st=1)
OUT["stats"] = emu_stats()
""", timeout=1200)
    healthy(r.pop("stats"))
    assert r.pop("N150 alm") == {"refused": -4}
    for k, v in r.items():
        assert v["same"] and len(set(v["ticks"])) == 1, (k, v)
    assert r["N100 pairs"]["launch"]["trajectories_per_wavefront"] == 2 and r["N150"]["launch"]["trajectories_per_wavefront"] == 2
    assert r["N100 default"]["launch"]["trajectories_per_wavefront"] == 1   # (k_solve's GPU-proven loop builds stay the default)


def test_a_lost_hand_over_is_loud_under_emulation(emu_libs):
    """Round 6 (VERDICT r05 task 5), the CPU twin of tests/test_gpu_parity.py::test_a_lost_hand_over_is_loud:
    the development build's hand-over wait is forced to expire at once.  Exactly the trajectories in transit keep
    CILQR_END_NOT_SOLVED (pre-marked on the launch stream), every other one == oracle; cilqr_wait reports CILQR_ERR_DEVICE once;
    with three launches in flight a failure in the FIRST slot is still reported after two clean launches (the handle-wide latch)."""
    r = run(emu_libs, r"""
sc, p = scenario("three_bend", 30)
B, N = 40, 30
x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0x10057)
ref = ORC.solve_batch(p, scene_of(sc), x0, n_threads=4)
ptr = lambda a: a.ctypes.data
ids = (np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32))
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True); eng.set_group_mode(2)
def bufs():
    r_ = np.zeros(B, dtype=pkg.RESULT_DTYPE); r_["iters"] = 7; r_["J_final"] = 1.0   # garbage that LOOKS like results
    return np.zeros((B, N, 2)), np.zeros((B, N + 1, 4)), r_
def solve(o): eng.solve_batch_device(B, ptr(x0), ptr(ids[0]), ptr(ids[1]), ptr(ids[2]), 0, ptr(o[0]), ptr(o[1]), ptr(o[2]), 0, 0, 0)
def check(o):
    lost = o[2]["end_reason"] == 4
    ok = ~lost
    good = bool((bits(o[0][ok]) == bits(ref["u"][ok])).all() and (bits(o[1][ok]) == bits(ref["x"][ok])).all() and (o[2]["iters"][ok] == ref["res"]["iters"][ok]).all())
    marked = bool((o[2]["iters"][lost] == 0).all() and np.isnan(o[2]["J_final"][lost]).all())
    return dict(lost=int(lost.sum()), others_equal_oracle=good, lost_look_unsolved=marked)
def wait_fails():
    try: eng.wait(); return False
    except RuntimeError as e: return "bounded wait" in str(e) and "NOT_SOLVED" in str(e)
a = bufs(); solve(a); OUT["healthy"] = dict(wait_fails=wait_fails(), **check(a), error=eng.work_sharing_stats()["error"])
os.environ["CILQR_GRP_WAIT_SPINS"] = "1"
b = bufs(); solve(b)
OUT["forced"] = dict(error_shown=eng.work_sharing_stats()["error"], wait_fails=wait_fails(), again=wait_fails(), **check(b), parked=eng.resume_stats())
try:
    eng.solve_batch(x0); OUT["host_entry_raises"] = False
except RuntimeError as e:
    OUT["host_entry_raises"] = "bounded wait" in str(e)
del os.environ["CILQR_GRP_WAIT_SPINS"]
eng.set_batches_in_flight(3)
outs = [bufs() for _ in range(3)]
os.environ["CILQR_GRP_WAIT_SPINS"] = "1"; solve(outs[0]); del os.environ["CILQR_GRP_WAIT_SPINS"]
solve(outs[1]); solve(outs[2])
OUT["in_flight"] = dict(error_shown=eng.work_sharing_stats()["error"], wait_fails=wait_fails(), slots=[check(o) for o in outs])
c = bufs(); solve(c); OUT["after"] = dict(wait_fails=wait_fails(), **check(c))
eng.close()
""", timeout=900)
    assert r["healthy"] == dict(wait_fails=False, lost=0, others_equal_oracle=True, lost_look_unsolved=True, error=0), r["healthy"]
    f = r["forced"]
    assert f["error_shown"] != 0 and f["wait_fails"] and not f["again"], f
    assert 1 <= f["lost"] <= f["parked"] and f["others_equal_oracle"] and f["lost_look_unsolved"], f
    assert r["host_entry_raises"] is True
    g = r["in_flight"]
    assert g["error_shown"] != 0 and g["wait_fails"], g
    assert g["slots"][0]["lost"] >= 1 and g["slots"][1]["lost"] == 0 and g["slots"][2]["lost"] == 0, g
    assert all(s["others_equal_oracle"] for s in g["slots"]), g
    assert r["after"] == dict(wait_fails=False, lost=0, others_equal_oracle=True, lost_look_unsolved=True), r["after"]


def test_cpp_drop_in_and_sharded_solver_on_the_emulator(emu_libs, pkg, orc_det, scenarios, tmp_path):
    """The drop-in boundary end to end without a GPU: examples/headless_planner.cpp (host C++ -> include/cilqr_solver_shim.hpp ->
    C-ABI -> kernels) linked against the emulator library.  (1) The reference's planning loop (mp:180-197) for four ticks of
    two_straight and three_straight (use_last_solution) == the oracle driven the same way.  (2) cilqr_amd::ShardedSolver on an
    emulated box with three devices: --devices 1, 2, 3 — one handle and one host thread per device, contiguous blocks — give
    the same checksum over every output bit and the same statistics (results do not depend on the shard count)."""
    import numpy as np
    from conftest import oracle_scene
    exe = tmp_path / "headless_planner_emu"
    lib = pathlib.Path(emu_libs["prod"])
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", str(ROOT / "include"), str(ROOT / "examples" / "headless_planner.cpp"), "-L", str(lib.parent),
                    "-l" + lib.stem[3:], "-Wl,-rpath," + str(lib.parent), "-pthread", "-o", str(exe)], check=True)
    env = dict(os.environ, CILQR_EMU_DEVICES="3")
    for name in ("two_straight", "three_straight"):
        cfg, sc = scenarios[name]
        out = subprocess.run([str(exe), str(pkg.config.SCENARIO_DIR / f"{name}.json"), "4"], check=True, capture_output=True, text=True, env=env).stdout
        rows = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
        assert rows.shape == (4, 9)
        s = orc_det.solver(pkg.params_from_config(cfg))
        ego, t = sc.ego_state.copy(), 0.0
        for i in range(4):
            index = int(t / sc.delta_t)
            r = s.solve(ego, oracle_scene(sc, index))
            ego = r["x"][1].copy()
            assert rows[i, 0] == index and rows[i, 7] == r["res"]["iters"], (name, i)
            assert np.array_equal(rows[i, 1:5], ego) and np.array_equal(rows[i, 5:7], r["u"][0]), (name, i)
            t += sc.delta_t
    lines = {}
    for g in ("1", "2", "3"):
        r = subprocess.run([str(exe), str(pkg.config.SCENARIO_DIR / "three_bend.json"), "--batch", "21", "--horizon", "30", "--devices", g],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-1500:]
        lines[g] = r.stdout.strip().split()
    field = lambda ln, name: ln[ln.index(name) + 1]
    for g, ln in lines.items():
        assert field(ln, "devices") == g, ln
        for name in ("checksum", "iters", "ls_trials", "converged", "max_lamb", "max_iter", "not_solved", "sum_J_final"):
            assert field(ln, name) == field(lines["1"], name), (g, name, ln, lines["1"])
    assert int(field(lines["1"], "iters")) > 21 and field(lines["1"], "not_solved") == "0"


def test_kernels_under_address_and_ub_sanitizers(emu_libs):
    """There are no sanitizers for gfx950 code; on the emulator the kernel sources are host code.  The --sanitize build
    (AddressSanitizer + UBSan, fibre switches announced to ASan, everything beyond a launch's dynamic LDS poisoned, "device"
    buffers = instrumented heap blocks of exactly the caller's size) runs one launch of every kernel family through the
    device-pointer entry points: no out-of-bounds access of outputs, tables, scratch or LDS, no undefined behaviour — and the
    results are still the oracle's.  That the tool is live is shown first: a hipMemset 16 bytes past a 64-byte device buffer is
    reported.  (An out-of-bounds store injected into a kernel is reported with the kernel's file:line — tests/emu/README.md.)"""
    sys.path.insert(0, str(EMU))
    import build_emu
    import glob
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no shared AddressSanitizer runtime next to the ROCm clang")
    lib = build_emu.build(dev=True, sanitize=True)
    env = {"LD_PRELOAD": rt[-1], "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=0:detect_stack_use_after_return=0:exitcode=0",
           "UBSAN_OPTIONS": "print_stacktrace=1"}
    e = dict(os.environ, **env)
    probe = subprocess.run([sys.executable, "-c", "import ctypes,sys; l=ctypes.CDLL(sys.argv[1]); p=ctypes.c_void_p(); "
                            "l._Z9hipMallocPPvm(ctypes.byref(p), ctypes.c_size_t(64)); l._Z9hipMemsetPvim(p, 0, ctypes.c_size_t(80))", str(lib)],
                           capture_output=True, text=True, env=e)
    assert "heap-buffer-overflow" in probe.stderr and "64-byte region" in probe.stderr, probe.stderr[-800:]
    body = r"""
ptr = lambda a: a.ctypes.data
def dev_solve(p, tab, x0, mode, ids=None, helper=None):
    B, N = len(x0), (p[0] if isinstance(p, (list, tuple)) else p).N
    eng = pkg.BatchedCILQR(p, tab, dev=True); eng.set_group_mode(mode)
    if helper is not None: eng.set_helper_mode(helper)
    z = np.zeros(B, dtype=np.int32) if ids is None else None
    sid, pid, tk = (z, z, z) if ids is None else ids
    u = np.zeros((B, N, 2)); x = np.zeros((B, N + 1, 4)); res = np.zeros(B, dtype=pkg.RESULT_DTYPE)
    eng.solve_batch_device(B, ptr(x0), ptr(sid), ptr(pid), ptr(tk), 0, ptr(u), ptr(x), ptr(res), 0, 0, 0); eng.wait()
    info = eng.last_launch_info(); eng.close()
    return dict(u=u, x=x, res=res), info
for name, N, B, st, mode in (("three_bend", 30, 10, 0, 2), ("three_bend", 30, 3, 0, 0), ("two_straight", 50, 6, 0, 2), ("three_bend", 30, 6, 1, 2),
                             ("three_bend", 30, 3, 1, -1), ("two_borrow", 130, 3, 0, -1)):
    cfg = pkg.GlobalConfig.get_instance(name); sc = pkg.build_scenario(cfg, name)
    obs = np.concatenate([sc.obstacles, np.repeat(sc.obstacles[:, -1:, :], 20, axis=1)], axis=1)
    p = pkg.params_from_config(cfg, N=N, solve_type=st, max_iter=25)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0x5A9)
    out, info = dev_solve(p, pkg.SceneTable(sc.lane.x, sc.lane.y, sc.lane.yaw, obs, sc.road_borders, sc.target_velocity), x0, mode)
    ref = ORC.solve_batch(p, Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, obs, sc.road_borders, sc.target_velocity), x0, n_threads=2)
    OUT["%s N=%d type %d mode %d" % (name, N, st, mode)] = dict(same=same(out, ref), launch=info)
wl = pkg.workloads.config4(B=6, N=100)
ref = ORC.solve_batch(wl.params, scenes_of(wl), wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=2)
for mode, helper in ((2, None), (0, 0)):
    out, info = dev_solve(wl.params, wl.scenes, wl.x0, mode, ids=(wl.scenario_id, wl.param_id, wl.tick), helper=helper)
    OUT["config4 N=100 mode %d" % mode] = dict(same=same(out, ref), launch=info)
"""
    e2 = dict(os.environ)
    e2.pop("CILQR_TUNE", None)
    e2.update(env, CILQR_AMD_LIB=str(lib), CILQR_AMD_LIB_DEV=str(lib), CILQR_EMU_BLOCKS_PER_CU="2")
    r = subprocess.run([sys.executable, "-c", PRELUDE + body + "\nprint('EMU-RESULT ' + json.dumps(OUT))\n", str(ROOT)], capture_output=True,
                       text=True, timeout=1500, env=e2)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("EMU-RESULT ")][-1][len("EMU-RESULT "):])
    assert len(res) == 8
    for k, v in res.items():
        assert v["same"], (k, v)
    # ... and a PLANTED bug is found: one compilation unit rebuilt with k_mark_unsolved storing one record past the end of the
    # caller's result array — reported as a heap-buffer-overflow inside k_mark_unsolved, with the source line
    bug = build_emu.build(dev=True, sanitize=True, planted_bug=True)
    small = r"""
sc, p = scenario("three_bend", 30)
B = 6; x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 3)
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True); eng.set_group_mode(2)
z = np.zeros(B, dtype=np.int32); u = np.zeros((B, 30, 2)); x = np.zeros((B, 31, 4)); res = np.zeros(B, dtype=pkg.RESULT_DTYPE)
ptr = lambda a: a.ctypes.data
eng.solve_batch_device(B, ptr(x0), ptr(z), ptr(z), ptr(z), 0, ptr(u), ptr(x), ptr(res), 0, 0, 0); eng.wait()
"""
    e3 = dict(e2, CILQR_AMD_LIB=str(bug), CILQR_AMD_LIB_DEV=str(bug))
    r = subprocess.run([sys.executable, "-c", PRELUDE + small, str(ROOT)], capture_output=True, text=True, timeout=600, env=e3)
    assert "heap-buffer-overflow" in r.stderr and "k_mark_unsolved" in r.stderr and "cilqr_amd.hip:" in r.stderr, r.stderr[-3000:]


def test_lockstep_points_cover_every_hazard(emu_libs):
    """The emulator runs the lanes of a wavefront one after the other between two cross-lane operations.  Where lanes exchange data
    through memory inside such a stretch (legal on the device: a wavefront's LDS operations execute in order) the scratch copy of
    the sources gets a rendezvous — tests/emu/build_emu.py LOCKSTEP_POINTS.  The instrumented build traces every load and store of
    the kernels and reports words touched by two lanes of one wavefront in the same stretch with a store among them: with the
    listed points in place nothing is left, on the pair sweep, the long layout and the helper build."""
    sys.path.insert(0, str(EMU))
    import build_emu
    libs = {"prod": build_emu.build(hazards=True), "dev": build_emu.build(dev=True, hazards=True)}
    r = run(emu_libs, r"""
HZ = EMULIB
def hazards_of(fn):
    HZ.cilqr_emu_hazards_enable(1); fn(); HZ.cilqr_emu_hazards_enable(0)
    A = (ctypes.c_void_p * 64)(); Bv = (ctypes.c_void_p * 64)(); Nn = (ctypes.c_longlong * 64)()
    return HZ.cilqr_emu_hazards(A, Bv, Nn, 64)
sc, p = scenario("three_bend", 30)
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=True)
x0 = pkg.workloads.perturbed_starts(sc.ego_state, 6, 0x5A0CE)
eng.set_group_mode(2); OUT["pairs"] = hazards_of(lambda: eng.solve_batch(x0))
eng.set_group_mode(0); OUT["helper"] = hazards_of(lambda: eng.solve_batch(x0[:2]))
eng.close()
wl = pkg.workloads.config4(B=2, N=70)
eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True); eng.set_group_mode(0); eng.set_helper_mode(0); eng.set_work_sharing(0)
OUT["lone two rows"] = hazards_of(lambda: eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick))
eng.close()
wl = pkg.workloads.config4(B=4, N=70)
eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True); eng.set_group_mode(2)
OUT["long"] = hazards_of(lambda: eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick))
eng.close()
""", timeout=1500, libs=libs, env={"CILQR_TUNE": "group_steal=0,group_slice=0"})
    assert r == {"pairs": 0, "helper": 0, "long": 0, "lone two rows": 0}, r
    assert len(build_emu.LOCKSTEP_POINTS) <= 6   # (a list that grows means the kernels lean on lockstep more and more: look again)


def test_single_ego_tick_costs_one_launch_two_copies_and_one_wait(emu_libs):
    """VERDICT r05 task 6 (small batches / single ego), the part a CPU can see: what the HOST side of the drop-in class's solve()
    (include/cilqr_solver.hpp:37-41 -> cilqr_solve) asks of the HIP runtime per planning tick.  The emulator's runtime counts the
    calls (cilqr_emu_api_counts).  Steady state — the tables resident, the obstacle window one tick on (mp:194-196) —: ONE kernel
    launch, TWO asynchronous copies (the ego state and the scalars in, the plan out: ~2 KB together at N = 30), ONE wait, one event
    record; no allocation, no memset, no synchronous copy.  The first call uploads the tables (allocations, ~70 KB).  The latency
    of a tick beyond that is the kernel's serial chain over the horizon, not host work."""
    r = run(emu_libs, r"""
NAMES = ["launches", "malloc", "free", "memcpy_sync", "memcpy_async", "memset", "sync", "event_record", "stream_wait", "bytes_copied"]
def counts():
    a = (ctypes.c_longlong * 10)(); EMULIB.cilqr_emu_api_counts(a); return np.array(list(a))
cfg = pkg.GlobalConfig.get_instance("two_straight"); sc = pkg.build_scenario(cfg, "two_straight")
solver = pkg.CILQRSolver(cfg)
ref = ORC.solver(solver.params); ref.reset()
x0 = sc.ego_state.copy(); prev = counts(); ticks = []; equal = True
for t in range(5):
    u, x = solver.solve(x0, sc.lane, sc.target_velocity, sc.obstacles[:, t:], sc.road_borders)
    now = counts(); ticks.append(dict(zip(NAMES, (now - prev).tolist()))); prev = now
    o = ref.solve(x0, scene_of(sc), tick=t)
    equal = equal and bool((bits(u) == bits(o["u"])).all() and (bits(x) == bits(o["x"])).all())
    x0 = x[1].copy()
OUT["ticks"] = ticks; OUT["equal"] = equal
""")
    assert r["equal"]
    first, later = r["ticks"][0], r["ticks"][1:]
    assert first["malloc"] > 0 and first["bytes_copied"] > 20000, first
    for t in later:
        assert t == dict(launches=1, malloc=0, free=0, memcpy_sync=0, memcpy_async=2, memset=0, sync=1, event_record=1, stream_wait=0,
                         bytes_copied=t["bytes_copied"]), t
        assert t["bytes_copied"] < 4096, t


def test_places_held_in_the_queue_with_preemption_inside_the_protocol(emu_libs):
    """The hand-over protocol's rarest branch: a slot that CLAIMS a place in the queue of parked trajectories beyond the pushes so far
    keeps it (GP_CLAIMED: grp_take_parked's -2) and an idle wavefront waits WITH its place (grp_wait_for_work's `claim` arm; ADVICE
    r05 low is about this branch).  It needs two takers racing for the last unclaimed push — a window between two atomic operations.
    The emulator's adversarial scheduler can take the processor away from a lane BEFORE any atomic operation, for one visit or for
    hundreds (CILQR_EMU_PREEMPT), so other blocks run inside such windows; before that existed no emulator run had ever executed
    the branch (block coverage).  Counters planted in the emulator's scratch copy of the sources (build_emu.py PROBES, not in csrc/)
    tell whether a run got there.  Launches of scripts/emu_stress.py --focus places: every one == oracle, no bounded wait expired, no
    trajectory left marked NOT_SOLVED — AND the branch was taken."""
    e = dict(os.environ)
    e.pop("CILQR_TUNE", None)
    stress = [sys.executable, str(ROOT / "scripts" / "emu_stress.py"), "--focus", "places", "--preempt", "2", "--lib", str(emu_libs["dev"])]
    # (a) the seven launches of seed 506's hundred that entered the branch, replayed (tests/emu/places_cases.json: the schedule is a
    #     function of the case and of the code, so they enter it again — until the code changes: then the message below says how to
    #     find new ones); (b) forty fresh random launches
    r = subprocess.run(stress + ["--replay", str(EMU / "places_cases.json")], capture_output=True, text=True, timeout=1200, env=e)
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and last["failed"] == 0 and last["hand_overs"] > 50, r.stdout[-2000:] + r.stderr[-1000:]
    if os.environ.get("CILQR_EMU_SANITIZE_ALL") != "1":
        assert last["places_kept"] > 0 and last["waits_with_a_place"] > 0, \
            (last, "the recorded launches no longer reach the claimed-place branch (the schedule follows the code): record new ones with "
                   "scripts/emu_stress.py --focus places --preempt 2 --cases 100 --seed N --hits-out tests/emu/places_cases.json")
    r = subprocess.run(stress + ["--cases", "40", "--seed", "2"], capture_output=True, text=True, timeout=1200, env=e)
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and last["failed"] == 0 and last["hand_overs"] > 50, r.stdout[-2000:] + r.stderr[-1000:]
