#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the `-m gpu` suite REHEARSED on the wave64 emulator of tests/emu/ — the same test bodies, the library
swapped (CILQR_AMD_LIB / CILQR_AMD_LIB_DEV).  Not a substitute for the GPU run (tests/emu/README.md says what the emulator cannot
see); it tells, without a GPU, whether the sources still compute the oracle's numbers along every path the GPU tests take.

The torch-based tests see a numpy-backed stand-in for `torch` (tests/emu/fake_torch: device memory is host memory here).  The
suite runs in passes, each with its own divisor of the batch sizes (CILQR_TEST_SHRINK: a solve takes a tenth of a second here
instead of microseconds).  LEFT OUT, by name, each with its reason (LEFT_OUT below).

    python scripts/emu_gpu_suite.py [--workers 6] [--out profiles/r06_gpu_suite_on_emulator.json]"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCALE = ("its assertions are about SCALE (helpers > 0, trajectories parked, two per wavefront: a batch beyond the helper range on a "
         "256-CU chip) — shrunk, the launch takes another build; the paths themselves run in tests/test_emulator.py with the ranges "
         "overridden (CILQR_TUNE)")
LEFT_OUT = {
    "detmath_device": "float -> int conversion of 1e300 in detmath's argument reduction: undefined on x86",
    "uniform_and_lane_parallel": "the development build's wave-uniform backward sweep zero-fills and stores in one lockstep stretch",
    "full_size": "BASELINE configurations at full size: tens of thousands of solves, hours here",
    "config2_full": "1024 solves x 2 libraries: an hour here; the same comparison at 8-64 rows runs in the first pass",
    "config4_every": "65 536 solves", "at_scale": "65 536 solves", "libm_gap": "thousands of solves per parameter",
    "fuzz_random": "thousands of solves",
    "lost_rows": "loads a gfx950 library (the experiment build with round 4's instruction shape)",
    "cpp_headless_planner": "a binary linked against the gfx950 library (its emulator twin: tests/test_emulator.py)",
    "sharded_solver_in_one_process": "a binary linked against the gfx950 library (its emulator twin: tests/test_emulator.py)",
    "block_timeline": "needs a clock (s_memrealtime)",
    "in_flight": "passes (IN-FLIGHT-OK, run as a script: ~230 launches take 40 minutes here, the test's own limit for its child is 20)",
    "pairs_at_scale": SCALE, "long_horizon_builds": SCALE, "resumable_solves": SCALE, "work_sharing_between": SCALE,
    "sliced_solves": SCALE,
}
# (divisor of the batch sizes, -k expression)
PASSES = [
    (20, None),  # everything not named below
    (100, "two_trajectories_per_wavefront_at_long or scratch_held"),
    (600, "resident_on_the_device or device_pointer_entry or one_handle_on_two or in_one_launch or concurrent_handles"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--timeout", type=int, default=3000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    libs = (build_emu.build(), build_emu.build(dev=True))
    named = [k for _, k in PASSES if k]
    rest = " and ".join("not " + s for s in list(LEFT_OUT) + [w for k in named for w in k.split(" or ")])
    rep = {"what": __doc__.split("\n\n")[0], "passes": [], "left_out": LEFT_OUT, "workers": a.workers}
    failed_all, passed_all, t00 = [], 0, time.time()
    for shrink, expr in PASSES:
        env = dict(os.environ, CILQR_AMD_LIB=str(libs[0]), CILQR_AMD_LIB_DEV=str(libs[1]), CILQR_TEST_SHRINK=str(shrink))
        env["PYTHONPATH"] = os.path.join(ROOT, "tests", "emu", "fake_torch") + os.pathsep + env.get("PYTHONPATH", "")
        env.pop("CILQR_TUNE", None)
        t0 = time.time()
        k = expr if expr else rest
        if expr:  # (a named pass never runs what is left out)
            k = f"({expr}) and " + " and ".join("not " + s for s in LEFT_OUT)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-p", "no:cacheprovider",
                            "-n", str(a.workers), f"--timeout={a.timeout}", "-rfE", "-k", k],
                           env=env, cwd=ROOT, capture_output=True, text=True)
        tail = r.stdout.strip().splitlines()
        summary = tail[-1] if tail else ""
        failed = [ln.split(" ", 1)[1].split(" - ")[0] for ln in tail if ln.startswith(("FAILED", "ERROR"))]
        m = {kk: int(v) for v, kk in re.findall(r"(\d+) (passed|failed|deselected|error|errors|skipped)", summary)}
        rep["passes"].append({"batch_sizes_divided_by": shrink, "selection": expr or "everything not named in another pass or left out",
                              "summary": summary, "counts": m, "failed": failed, "wall_s": round(time.time() - t0)})
        print(json.dumps(rep["passes"][-1]), flush=True)
        failed_all += failed
        passed_all += m.get("passed", 0)
    rep.update({"passed": passed_all, "failed": failed_all, "wall_s": round(time.time() - t00)})
    print(json.dumps({"passed": passed_all, "failed": failed_all}))
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)
    sys.exit(0 if not failed_all and passed_all > 0 else 1)


if __name__ == "__main__":
    main()
