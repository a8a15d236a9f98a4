"""In-tree build of libcilqr_amd.so (hipcc, gfx950).  Compiled with -ffp-contract=off — parity
with the CPU restatement depends on it."""
import os
import pathlib
import shutil
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libcilqr_amd.so"

HIP_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared",
             "-Wno-unused-result"]


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


def build_library(force=False, verbose=False):
    srcs = [CSRC / "cilqr_amd.hip", CSRC / "scenario.cpp"]
    deps = srcs + [CSRC / "cilqr_device.hpp", CSRC / "detmath.h", ROOT / "include" / "cilqr_amd.h",
                   pathlib.Path(__file__)]  # the flags live in this file
    if not force and _newer(LIB, deps):
        return LIB
    cmd = [hipcc_path()] + HIP_FLAGS + [str(s) for s in srcs] + ["-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return LIB


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
    build_examples(force, verbose)
    return lib
