#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: run a command (bench.py, a torch-based GPU test) on the wave64 emulator — CILQR_AMD_LIB[_DEV] point at the
emulator libraries and a numpy-backed stand-in for `torch` (tests/emu/fake_torch) is first on PYTHONPATH.  A REHEARSAL of code paths
that otherwise only ever run on a GPU box; every number it prints is meaningless as a measurement.

    python scripts/emu_rehearse.py -- python bench.py --config 3 --batch 12 --steps 2 --warmup 1 --no-cpu-baseline
    python scripts/emu_rehearse.py -- python -m pytest tests/test_gpu_parity.py -m gpu -k in_flight
    python scripts/emu_rehearse.py --ranks 2 -- python bench.py --gpus 2 --config 3 --batch 6 --steps 2 --warmup 1 --no-cpu-baseline
      (N > 1: one process per emulated device, the stand-in's torch.distributed exchanges the statistics through files)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu  # noqa: E402

args = sys.argv[1:]
ranks = 1
if args and args[0] == "--ranks":   # N processes, one emulated device each, RANK / LOCAL_RANK / WORLD_SIZE set as torchrun sets them
    ranks, args = int(args[1]), args[2:]
if args and args[0] == "--":
    args = args[1:]
env = dict(os.environ, CILQR_AMD_LIB=str(build_emu.build()), CILQR_AMD_LIB_DEV=str(build_emu.build(dev=True)))
env["PYTHONPATH"] = os.path.join(ROOT, "tests", "emu", "fake_torch") + os.pathsep + env.get("PYTHONPATH", "")
env.setdefault("CILQR_TEST_SHRINK", "20")
if ranks == 1:
    sys.exit(subprocess.run(args, env=env, cwd=ROOT).returncode)
import tempfile
with tempfile.TemporaryDirectory() as d:
    procs = []
    for r in range(ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT="29599",
                 CILQR_FAKE_DIST_DIR=d, CILQR_EMU_DEVICES=str(ranks))
        procs.append(subprocess.Popen(args, env=e, cwd=ROOT))
    sys.exit(max(p.wait() for p in procs))
