"""Sanitizer jobs: the CPU oracle and the GPU-free host C++ of the product are compiled with
AddressSanitizer + UndefinedBehaviorSanitizer (no recovery) and exercised; any finding aborts the program."""
import subprocess

import numpy as np

from conftest import ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1", "OMP_NUM_THREADS": "2",
       "PATH": "/usr/bin:/bin"}


def run(cmd, **kw):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    assert p.returncode == 0, (cmd, p.stdout[-1500:], p.stderr[-3000:])
    return p.stdout


def test_oracle_under_asan_ubsan(tmp_path):
    for mode, defs in (("libm", []), ("det", ["-DORC_DETMATH"])):
        exe = tmp_path / f"oracle_san_{mode}"
        run(["gcc", "-std=gnu11", "-ffp-contract=off", "-fopenmp", *SAN, *defs, "-I", str(ROOT / "oracle"),
             str(ROOT / "tests" / "sanitize" / "oracle_main.c"), str(ROOT / "oracle" / "cilqr_oracle.c"), "-lm", "-o", str(exe)])
        out = run([str(exe)], env=ENV)
        assert "SANITIZE-ORACLE-OK" in out, out


def test_host_cpp_under_asan_ubsan(pkg, tmp_path):
    exe = tmp_path / "host_san"
    run(["g++", "-std=c++17", *SAN, "-I", str(ROOT / "include"), str(ROOT / "tests" / "sanitize" / "host_main.cpp"),
         str(ROOT / "toy-example-of-ilqr_amd" / "csrc" / "scenario.cpp"), "-o", str(exe)])
    import json
    from conftest import reference_layout_yaml
    for name in ("three_bend", "two_straight"):
        flat = json.loads((pkg.config.SCENARIO_DIR / f"{name}.json").read_text())
        ypath = tmp_path / f"scenario_{name}.yaml"
        ypath.write_text(reference_layout_yaml(flat))
        out = run([str(exe), str(pkg.config.SCENARIO_DIR / f"{name}.json"), str(ypath)], env=ENV)
        assert "SANITIZE-HOST-OK" in out, out


def test_cpp_start_generator_matches_python(pkg):
    """cilqr_perturbed_starts (host C++) against workloads.perturbed_starts: x and y (uniform components) bit for
    bit, v and yaw (Box-Muller through the platform's log / cos) within 2 ulp; shards regenerate their own rows."""
    import ctypes as C
    lib = pkg._lib.load()
    base = np.array([-10.0, 1.0, 4.0, 0.0])
    for seed, first, B in ((0xC11A0003, 0, 4096), (0xC11A0005, 60000, 512)):
        out = np.empty((B, 4))
        pkg._lib.check(lib.cilqr_perturbed_starts(base.ctypes.data_as(C.c_void_p), B, seed, first, out.ctypes.data_as(C.c_void_p)),
                       "cilqr_perturbed_starts")
        ref = pkg.workloads.perturbed_starts(base, B, seed, first)
        assert np.array_equal(out[:, :2], ref[:, :2])
        for c in (2, 3):
            ulp = np.spacing(np.maximum(np.abs(ref[:, c] - base[c]), 1e-300))
            assert (np.abs(out[:, c] - ref[:, c]) <= 4 * np.maximum(ulp, np.spacing(np.abs(ref[:, c])))).all()
    a = np.empty((16, 4))
    b = np.empty((64, 4))
    lib.cilqr_perturbed_starts(base.ctypes.data_as(C.c_void_p), 16, 7, 32, a.ctypes.data_as(C.c_void_p))
    lib.cilqr_perturbed_starts(base.ctypes.data_as(C.c_void_p), 64, 7, 0, b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, b[32:48])
