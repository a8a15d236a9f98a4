#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: which lines of the DEVICE sources do the emulator tests execute?  gfx950 code has no coverage tool; on the
wave64 emulator (tests/emu/) the same sources run as host code, built with -fsanitize-coverage=trace-pc-guard,pc-table: every
basic block of every template instantiation has a guard, hit flags are dumped per process, this script symbolises the blocks
(innermost inlined frame -> file:line) and merges.  A source line counts as executed when a block of it ran in ANY instantiation.

    python scripts/emu_coverage.py [--run] [--out profiles/r06_emulator_coverage.json]
--run: first execute tests/test_emulator.py (+ a short adversarial stress) on the coverage build into a fresh dump directory."""
import argparse
import collections
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COVDIR = os.path.join(ROOT, "scratch", "cov")
LIB = os.path.join(ROOT, "tests", "emu", "_build", "libcilqr_emu_dev_cov.so")
SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def run_tests():
    os.makedirs(COVDIR, exist_ok=True)
    for f in glob.glob(os.path.join(COVDIR, "*.cov")):
        os.remove(f)
    env = dict(os.environ, CILQR_EMU_COVERAGE="1", CILQR_EMU_COV_DIR=COVDIR)
    subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emulator.py"), "-q", "-p", "no:cacheprovider",
                    "-k", "not lockstep_points"], env=env, cwd=ROOT, check=False)
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "emu_stress.py"), "--cases", "10", "--seed", "3", "--lib", LIB], env=env,
                   cwd=ROOT, check=False, stdout=subprocess.DEVNULL)
    # ... and the GPU suite's own test bodies that fit the emulator's speed (stage-by-stage entry points, failure paths, edge cases,
    # the single-ego entry point): the same assertions, the library swapped
    env2 = dict(env, CILQR_AMD_LIB=LIB, CILQR_AMD_LIB_DEV=LIB, CILQR_TEST_SHRINK="600")
    env2["PYTHONPATH"] = os.path.join(ROOT, "tests", "emu", "fake_torch") + os.pathsep + env2.get("PYTHONPATH", "")  # (the torch-based ones)
    subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-p", "no:cacheprovider",
                    "--timeout=300", "-k", "stages_bitexact or backward_pass_failure or error_codes or api_contract or edge_no_obstacles or "
                    "edge_start or irregular_lane or single_ego_entry_keeps or solver_class_mirror or solve_yaml_start or "
                    "solve_ticks_and_warm or alm_state_follows or device_pointer_entry"], env=env2, cwd=ROOT, check=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--show", default="", help="print the unexecuted line ranges of this file (e.g. cilqr_group.hpp)")
    a = ap.parse_args()
    if a.run:
        run_tests()
    hit = {}
    for f in glob.glob(os.path.join(COVDIR, "*.cov")):
        for ln in open(f):
            pc, h = ln.split()
            hit[pc] = hit.get(pc, 0) | int(h)
    pcs = sorted(hit)
    assert pcs, "no coverage dumps under scratch/cov (run with --run)"
    out = subprocess.run([SYMBOLIZER, "--obj=" + LIB, "--functions=none", "--inlines"], input="\n".join("0x" + p for p in pcs), capture_output=True,
                         text=True, check=True).stdout.strip().split("\n\n")
    assert len(out) == len(pcs), (len(out), len(pcs))
    lines = collections.defaultdict(lambda: [0, 0])   # (file, line) -> [blocks, blocks hit]
    for pc, blk in zip(pcs, out):
        first = blk.split("\n")[0]           # innermost frame
        path, line = first.rsplit(":", 2)[0], first.rsplit(":", 2)[1]
        name = os.path.basename(path)
        if "/_gen/" not in path or not line.isdigit() or int(line) == 0:
            continue
        e = lines[(name, int(line))]
        e[0] += 1
        e[1] += hit[pc]
    rep = {"what": __doc__.split("\n\n")[0], "files": {}}
    for name in sorted({k[0] for k in lines}):
        ls = {ln: v for (n, ln), v in lines.items() if n == name}
        done = sorted(ln for ln, v in ls.items() if v[1] > 0)
        todo = sorted(ln for ln, v in ls.items() if v[1] == 0)
        ranges, start, prev = [], None, None
        for ln in todo:
            if start is None:
                start = prev = ln
            elif ln <= prev + 3:
                prev = ln
            else:
                ranges.append([start, prev]); start = prev = ln
        if start is not None:
            ranges.append([start, prev])
        rep["files"][name] = {"lines_with_code": len(ls), "lines_executed": len(done), "frac": round(len(done) / max(1, len(ls)), 4),
                              "unexecuted_line_ranges (scratch copy: +/- a few lines of csrc/)": ranges}
        print(f"{name:24s} {len(done):5d} / {len(ls):5d} lines executed ({100.0 * len(done) / max(1, len(ls)):.1f} %), {len(ranges)} unexecuted ranges")
        if a.show and a.show == name:
            for r in ranges:
                print("   ", r)
    tot = sum(v["lines_with_code"] for v in rep["files"].values())
    ex = sum(v["lines_executed"] for v in rep["files"].values())
    rep["total"] = {"lines_with_code": tot, "lines_executed": ex, "frac": round(ex / max(1, tot), 4), "blocks": len(pcs), "blocks_executed": sum(hit.values())}
    print("total", rep["total"])
    if a.out:
        json.dump(rep, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
