"""In-tree build of libcilqr_amd.so (production) and libcilqr_amd_dev.so (the same sources with
-DCILQR_DEV_BUILD: + the testing-aid and cycle-accounting builds of the solve kernel, + CILQR_TUNE).  hipcc, gfx950,
-ffp-contract=off — parity with the CPU restatement depends on it.

The solve kernel has ~20 builds (template instantiations); they are compiled in CILQR_SOLVE_GROUPS groups
(csrc/cilqr_solve_inst.hip, -DCILQR_INST_GROUP=g) next to the host file, in parallel, and linked into one library."""
import concurrent.futures
import os
import pathlib
import shutil
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libcilqr_amd.so"
LIB_DEV = PKG / "libcilqr_amd_dev.so"
# NOT a product: the library with round 4's lost-store instruction shape compiled back in (-DCILQR_LOSTROWS_REPRO), kept next to
# the shipped ones so that tests/test_gpu_parity.py::test_lost_rows_shape_is_still_what_loses_rows runs wherever the suite runs
LIB_LOSTROWS = PKG / "libcilqr_amd_lostrows.so"
OBJ = PKG / "build"
GROUPS = 8

HIP_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result"]
# (--offload-compress would store the code objects zstd-compressed: 4.1 MB -> 1.7 MB on disk — but llvm-readelf cannot read what
#  llvm-objdump --offloading then extracts from a library of several bundles, i.e. scripts/kernel_metadata.py and the
#  disassembly scans of tests/test_cabi.py would go blind.  Not used.)

# The compiler the shipped kernels were validated with (bit-exact parity at scale, pairing invariance, the disassembly scans of
# tests/test_cabi.py).  Round 4 met a gfx950 anomaly that depends on the instruction shape the compiler emits (a 16-byte
# buffer store inside a waterfall loop: profiles/r04_experiments/tiled_slab_lost_rows.txt, mechanism open) in the grouped
# kernel's rollout pass; the guards are a disassembly scan of THIS compiler's output and GPU tests of ITS binaries.  A library
# built with any other compiler therefore does not run two trajectories per wavefront by default (cilqr_set_group_mode(2)
# still selects it explicitly): -DCILQR_COMPILER_VALIDATED is only passed when the versions match.
VALIDATED_HIPCC = ("HIP version: 7.2.26015-fc0010cf6a",
                   "AMD clang version 22.0.0git (https://github.com/RadeonOpenCompute/llvm-project roc-7.2.0 26014 "
                   "7b800a19466229b8479a78de19143dc33c3ab9b5)")


def compiler_identity(hipcc=None):
    out = subprocess.run([hipcc or hipcc_path(), "--version"], capture_output=True, text=True).stdout.splitlines()
    return tuple(ln.strip() for ln in out[:2])


def compiler_validated(hipcc=None):
    return compiler_identity(hipcc) == VALIDATED_HIPCC


def _newer(target, sources):
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(pathlib.Path(s).stat().st_mtime <= t for s in sources)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _deps():
    return [CSRC / "cilqr_amd.hip", CSRC / "cilqr_solve_inst.hip", CSRC / "scenario.cpp", CSRC / "cilqr_kernels.hpp",
            CSRC / "cilqr_device.hpp", CSRC / "cilqr_group.hpp", CSRC / "detmath.h", CSRC / "exports.map", ROOT / "include" / "cilqr_amd.h",
            pathlib.Path(__file__)]  # the flags live in this file


def _run(cmd, verbose):
    if verbose:
        print(" ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, cwd=str(CSRC))


def build_library(force=False, verbose=False, dev=False, out=None, jobs=None, extra_defs=()):
    """extra_defs: experiments only (e.g. ("-DCILQR_FUSED",) into ab/libF.so, profiles/r04_experiments)"""
    lib = pathlib.Path(out) if out else (LIB_DEV if dev else LIB)
    if not force and _newer(lib, _deps()):
        return lib
    hipcc = hipcc_path()
    tag = "dev" if dev else "prod"
    objdir = OBJ / (tag if out is None else tag + "_" + lib.stem)
    objdir.mkdir(parents=True, exist_ok=True)
    defs = (["-DCILQR_DEV_BUILD"] if dev else []) + list(extra_defs)
    if compiler_validated(hipcc):
        defs.append("-DCILQR_COMPILER_VALIDATED=1")
    else:
        print("build.py: hipcc is not the validated compiler (%s): the library will not pair trajectories per wavefront by "
              "default" % " / ".join(compiler_identity(hipcc)), flush=True)
    units = [(CSRC / "cilqr_amd.hip", objdir / "cilqr_amd.o", []), (CSRC / "scenario.cpp", objdir / "scenario.o", [])]
    units += [(CSRC / "cilqr_solve_inst.hip", objdir / f"solve_inst_{g}.o", [f"-DCILQR_INST_GROUP={g}"]) for g in range(GROUPS)]
    jobs = jobs or min(len(units), os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        futs = [ex.submit(_run, [hipcc] + HIP_FLAGS + defs + extra + ["-c", src, "-o", obj], verbose) for src, obj, extra in units]
        for f in futs:
            f.result()
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--strip-all", f"-Wl,--version-script={CSRC / 'exports.map'}"]
         + [obj for _, obj, _ in units] + ["-o", lib], verbose)
    return lib


def build_lostrows(force=False, verbose=False):
    """the anomaly-regression library (see LIB_LOSTROWS); nothing in the package loads it"""
    return build_library(force, verbose, out=LIB_LOSTROWS, extra_defs=("-DCILQR_LOSTROWS_REPRO",))


def build_examples(force=False, verbose=False):
    """host C++ driver on top of the C-ABI (plain g++, links libcilqr_amd.so)"""
    src = ROOT / "examples" / "headless_planner.cpp"
    exe = ROOT / "examples" / "headless_planner"
    deps = [src, ROOT / "include" / "cilqr_amd.h", ROOT / "include" / "cilqr_solver_shim.hpp",
            ROOT / "include" / "cilqr_config.hpp", LIB]
    if not force and _newer(exe, deps):
        return exe
    cmd = ["g++", "-std=c++17", "-O2", "-I", str(ROOT / "include"), str(src), "-L", str(PKG), "-lcilqr_amd",
           "-Wl,-rpath," + str(PKG), "-Wl,-rpath,$ORIGIN/../toy-example-of-ilqr_amd", "-o", str(exe)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return exe


def build_all(force=False, verbose=False):
    lib = build_library(force, verbose)
    build_library(force, verbose, dev=True)
    build_lostrows(force, verbose)
    build_examples(force, verbose)
    return lib
