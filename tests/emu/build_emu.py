"""TEST INFRASTRUCTURE ONLY — builds tests/emu/_build/libcilqr_emu[_dev].so: the kernel and host sources of
toy-example-of-ilqr_amd/csrc/ compiled as plain C++ for the x86 host against tests/emu/include/hip/hip_runtime.h and
tests/emu/emu_runtime.cpp (a wave64 emulator: tests/emu/README.md).  Same C-ABI as the product library; loaded by
tests/test_emulator.py through CILQR_AMD_LIB — never by the package on its own, never shipped.

The sources are used AS THEY ARE but for two mechanical rewrites made on a scratch copy (tests/emu/_gen/, not tracked):
  * `extern __shared__ double g_lds[];`  ->  the current block's emulated LDS,
  * gfx950 inline assembly: `v_fma_f64` -> __builtin_fma (the same correctly rounded operation), `s_waitcnt ...` and the empty
    register-pinning statements ("+v" / "+s" constraints) -> a compiler barrier.
Anything else that does not compile for the host fails the build: the emulator must follow the sources, not the other way round.

    python tests/emu/build_emu.py [--dev] [--csrc DIR] [--out PATH] [--force]
"""
import argparse
import concurrent.futures
import os
import pathlib
import re
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "toy-example-of-ilqr_amd" / "csrc"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
GROUPS = 8

# LOCKSTEP POINTS.  The emulator runs the lanes of a wavefront one after the other between two cross-lane operations; the device
# runs them in lockstep.  Where lanes of one wavefront pass data through LDS INSIDE such a stretch — legal on the device, whose
# LDS executes a wavefront's operations in order — the scratch copy gets an explicit rendezvous (a wave barrier: no effect on the
# arithmetic).  The list is not guessed: the instrumented build (--hazards, emu_runtime.cpp) reports every word that two lanes of
# one wavefront touch in the same stretch with at least one store; tests/test_emulator.py runs it and fails on any pair that is
# not covered here.  (file, text the rendezvous goes in FRONT of, occurrences expected, why)
LOCKSTEP_POINTS = [
    ("cilqr_group.hpp", "        double m1[4], m2[4];\n", 1,
     "backward_sweep_pair: every lane stores its element of the streamed trajectory's ring chunk, then reads other lanes' elements"),
    ("cilqr_device.hpp", "        double m1[4], m2[4];\n", 1,
     "backward_sweep_lanes with the expansion in global rows (k_solve's two-row builds): the same ring refill, lone sweep"),
    ("cilqr_kernels.hpp", "            // what the trajectory carries to its next segment\n", 1,
     "k_solve_grp: lane 0 stores the trajectory's scalars into GrpSt, which every lane has read as wave-uniform values in the same stretch"),
]

# PROBES.  Counters planted in the scratch copy (never in csrc/): how often did a run reach a branch that is too rare to assume?
# emu_runtime.cpp counts one per wavefront (lane 0); cilqr_emu_probe_counts() returns them in this order.  (file, text the probe
# goes in FRONT of, occurrences expected, name)
PROBES = [
    ("cilqr_group.hpp", "        h = (unsigned)claim;\n", 1, "waits_with_a_place"),   # grp_wait_for_work entered holding a place in the queue
    ("cilqr_group.hpp", "    *claim = (int)h;\n    return -2;\n", 1, "places_kept"),   # grp_take_parked: claimed beyond the pushes, entry not there after four looks
    # k_solve_grp, a wavefront with nothing to do: do BOTH its slots hold a place (the state ADVICE r05's low finding starts from)?
    ("cilqr_kernels.hpp", "            if (claim < 0 && !LOOP && slice) {\n", 1, "idle_with_two_places",
     "if (G == 2 && grp_state(g_lds, N, 0)->phase == GP_CLAIMED && grp_state(g_lds, N, 1)->phase == GP_CLAIMED) "),
]

ASM_RE = re.compile(r"__asm__\s*(?:volatile)?\s*\((?:[^()]|\([^()]*\))*\)\s*;")


def rewrite(text, name):
    out, n_asm = [], 0

    def asm(m):
        nonlocal n_asm
        n_asm += 1
        s = m.group(0)
        if "v_fma_f64 %0, %1, %2, %3" in s:
            ops = re.findall(r'"=?[vs]"\((\w+)\)', s)
            assert len(ops) == 4, (name, s)
            return f"{ops[0]} = __builtin_fma({ops[1]}, {ops[2]}, {ops[3]});"
        body = re.search(r'\(\s*"([^"]*)"', s).group(1)
        assert body == "" or body.startswith("s_waitcnt"), (name, s)   # nothing else is known to be a no-op here
        return "EMU_ASM_BARRIER();"

    text = ASM_RE.sub(asm, text)
    assert "__asm__" not in text, name
    text = text.replace("extern __shared__ double g_lds[];", "#define g_lds (emu::lds_base())")
    text = text.replace('"../../include/cilqr_amd.h"', '"cilqr_amd.h"')  # (the scratch copy sits elsewhere: found through -I)
    assert "__shared__" not in text, name
    for fname, anchor, count, _why in LOCKSTEP_POINTS:
        if fname == name:
            assert text.count(anchor) == count, (name, anchor, text.count(anchor))
            text = text.replace(anchor, "EMU_LOCKSTEP();\n" + anchor)
    for i, (fname, anchor, count, _name, *cond) in enumerate(PROBES):
        if fname == name:
            assert text.count(anchor) == count, (name, anchor, text.count(anchor))
            text = text.replace(anchor, (cond[0] if cond else "") + f"EMU_PROBE({i});\n" + anchor)
    return text, n_asm


def generate(csrc, gen):
    gen.mkdir(parents=True, exist_ok=True)
    total = 0
    for p in sorted(csrc.iterdir()):
        if p.suffix in (".hip", ".hpp", ".h", ".cpp"):
            t, n = rewrite(p.read_text(), p.name)
            total += n
            q = gen / p.name
            if not q.exists() or q.read_text() != t:
                q.write_text(t)
    return total


def build(dev=False, csrc=CSRC, out=None, force=False, verbose=False, opt="-O1", hazards=False, coverage=False, sanitize=False, planted_bug=False):
    csrc = pathlib.Path(csrc)
    tag = ("dev" if dev else "prod") + ("_hazards" if hazards else "") + ("_cov" if coverage else "") + ("_san" if sanitize else "") + ("" if csrc == CSRC else "_" + re.sub(r"\W", "_", str(csrc))[-40:])
    gen = HERE / "_gen" / tag
    objdir = HERE / "_build" / tag
    objdir.mkdir(parents=True, exist_ok=True)
    lib = pathlib.Path(out) if out else HERE / "_build" / ("libcilqr_emu" + ("_dev" if dev else "") + ("_hazards" if hazards else "") + ("_cov" if coverage else "") + ("_san" if sanitize else "") + ".so")
    if planted_bug:
        lib = lib.with_name(lib.stem + "_planted.so")
    generate(csrc, gen)
    deps = [d for d in gen.iterdir() if d.is_file()] + [HERE / "emu_runtime.cpp", HERE / "include" / "hip" / "hip_runtime.h", ROOT / "include" / "cilqr_amd.h",
                                  pathlib.Path(__file__)]
    if not force and lib.exists() and all(d.stat().st_mtime <= lib.stat().st_mtime for d in deps):
        return lib
    fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []
    flags = ["-x", "c++", "-std=c++17", opt, "-g1", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-everything",
             "-D__HIPCC__=1", "-D__HIP_DEVICE_COMPILE__=1", "-DCILQR_COMPILER_VALIDATED=1",
             "-I", str(HERE / "include"), "-I", str(gen), "-I", str(ROOT / "include")] + fma + (["-DCILQR_DEV_BUILD"] if dev else [])
    units = [(gen / "cilqr_amd.hip", objdir / "cilqr_amd.o", []), (gen / "scenario.cpp", objdir / "scenario.o", []),
             (HERE / "emu_runtime.cpp", objdir / "emu_runtime.o", [])]
    units += [(gen / "cilqr_solve_inst.hip", objdir / f"solve_inst_{g}.o", [f"-DCILQR_INST_GROUP={g}"]) for g in range(GROUPS)]

    # the lockstep-hazard detector (emu_runtime.cpp): every load and store of the KERNEL sources traced
    cov = ["-fsanitize-coverage=trace-pc-guard,trace-loads,trace-stores"] if hazards else []
    if coverage:  # basic-block coverage of the kernel sources (emu_runtime.cpp, scripts/emu_coverage.py)
        cov = ["-fsanitize-coverage=trace-pc-guard,pc-table", "-g"]

    # AddressSanitizer + UndefinedBehaviorSanitizer on the KERNEL sources (there are no sanitizers for gfx950 code): out-of-bounds
    # accesses of "device" buffers (hipMalloc = the instrumented malloc) and of the block's LDS (the emulator poisons what lies
    # beyond the launch's dynamic LDS size), signed overflow, bad shifts, misaligned or null accesses.  The library then needs
    # LD_PRELOAD of the ASan runtime (tests/test_emulator.py does that)
    san = ["-fsanitize=address,undefined", "-fno-sanitize=float-cast-overflow,function", "-fno-omit-frame-pointer", "-fsanitize-recover=all",
           "-shared-libsan", "-mllvm", "-asan-globals=false", "-DCILQR_EMU_SANITIZE=1"] if sanitize else []

    def cc(src, obj, extra):
        extra = extra + san
        if src.name != "emu_runtime.cpp":
            extra = extra + cov
        if not force and obj.exists() and all(d.stat().st_mtime <= obj.stat().st_mtime for d in deps):
            return
        cmd = [CLANG] + flags + extra + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if planted_bug:
        # the sanitizer build's self-check: ONE compilation unit rebuilt from a copy with an out-of-bounds store planted in
        # k_mark_unsolved (one record past the end of the caller's result array), linked with the other units as they are
        bug = gen / "planted" / "cilqr_amd.hip"
        bug.parent.mkdir(exist_ok=True)
        t = (gen / "cilqr_amd.hip").read_text()
        assert t.count("    if (b >= B) return;\n    cilqr_result r;") == 1
        t = t.replace("    if (b >= B) return;\n    cilqr_result r;", "    if (b > B) return;   /* PLANTED BUG */\n    cilqr_result r;")
        if not bug.exists() or bug.read_text() != t:
            bug.write_text(t)
        units[0] = (bug, objdir / "cilqr_amd_planted.o", ["-I", str(gen)])
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
        for f in [ex.submit(cc, *u) for u in units]:
            f.result()
    # (noinline device functions defined in the headers exist once per compilation unit — one code object each on the GPU; here the
    #  identical copies meet in one link)
    subprocess.run([CLANG, "-shared", "-fPIC", "-o", str(lib)] + [str(o) for _, o, _ in units] + ["-lm", "-lstdc++", "-ldl", "-Wl,--allow-multiple-definition"]
                   + (["-fsanitize=address,undefined", "-shared-libsan"] if sanitize else []), check=True)
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dev", action="store_true")
    ap.add_argument("--csrc", default=str(CSRC))
    ap.add_argument("--out")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-v", action="store_true")
    ap.add_argument("--opt", default="-O1")
    ap.add_argument("--hazards", action="store_true", help="instrumented build for the lockstep-hazard detector")
    ap.add_argument("--coverage", action="store_true", help="instrumented build for basic-block coverage of the kernel sources")
    ap.add_argument("--sanitize", action="store_true", help="AddressSanitizer + UBSan build of the kernel sources")
    a = ap.parse_args()
    print(build(a.dev, pathlib.Path(a.csrc), a.out, a.force, a.v, a.opt, a.hazards, a.coverage, a.sanitize))
