#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: the `-m gpu` suite REHEARSED on the wave64 emulator of tests/emu/ — the same test bodies, the library
swapped (CILQR_AMD_LIB / CILQR_AMD_LIB_DEV).  Not a substitute for the GPU run (tests/emu/README.md says what the emulator cannot
see); it tells, without a GPU, whether the sources still compute the oracle's numbers along every path the GPU tests take.

The torch-based tests see a numpy-backed stand-in for `torch` (tests/emu/fake_torch: device memory is host memory here).
Left out, by name: torch-based tests with fixed sizes in the thousands of solves, the binaries linked against the
gfx950 library, the BASELINE configurations at full size (tens of thousands of solves: hours here), block timelines (a clock), and
two testing aids the emulator does not model (float -> int conversion of 1e300 in detmath's argument reduction: undefined on x86;
the wave-uniform backward sweep of the development build, which zero-fills and stores in one lockstep stretch).

    python scripts/emu_gpu_suite.py [--workers 6] [--shrink 20] [--out profiles/r06_gpu_suite_on_emulator.json]"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ["detmath_device", "uniform_and_lane_parallel", "full_size", "config2_full", "config4_every", "at_scale", "libm_gap", "concurrent_handles",
        "resident_on_the_device", "device_pointer_entry", "one_handle_on_two", "in_one_launch", "in_flight", "lost_rows",
        "cpp_headless_planner", "sharded_solver_in_one_process", "block_timeline", "pairs_at_scale",
        "scratch_held", "fuzz_random", "long_horizon_builds", "resumable_solves", "work_sharing_between", "sliced_solves",
        "two_trajectories_per_wavefront_at_long"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--shrink", type=int, default=20)
    ap.add_argument("--timeout", type=int, default=1500)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    libs = (build_emu.build(), build_emu.build(dev=True))
    env = dict(os.environ, CILQR_AMD_LIB=str(libs[0]), CILQR_AMD_LIB_DEV=str(libs[1]), CILQR_TEST_SHRINK=str(a.shrink))
    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "emu", "fake_torch") + os.pathsep + env.get("PYTHONPATH", "")
    env.pop("CILQR_TUNE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-p", "no:cacheprovider",
                        "-n", str(a.workers), f"--timeout={a.timeout}", "-rfE", "-k", " and ".join("not " + s for s in SKIP)],
                       env=env, cwd=ROOT, capture_output=True, text=True)
    tail = r.stdout.strip().splitlines()
    summary = tail[-1] if tail else ""
    failed = [ln.split(" ", 1)[1].split(" - ")[0] for ln in tail if ln.startswith(("FAILED", "ERROR"))]
    m = {k: int(v) for v, k in re.findall(r"(\d+) (passed|failed|deselected|error|errors|skipped)", summary)}
    rep = {"what": __doc__.split("\n\n")[0], "summary": summary, "counts": m, "failed": failed, "left_out_by_name": SKIP,
           "batch_sizes_of_round6_tests_divided_by": a.shrink, "wall_s": round(time.time() - t0), "workers": a.workers}
    print(json.dumps(rep, indent=1))
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)
    sys.exit(0 if not failed and m.get("passed", 0) > 0 else 1)


if __name__ == "__main__":
    main()
