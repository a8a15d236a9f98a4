#!/usr/bin/env python3
"""Runs ON THE GPU BOX: how many trajectories of the large-batch build are resident at once?  B copies of ONE
trajectory (identical work per block): the launch time steps up where a second round of blocks begins.
usage: scripts/occupancy_probe.py [N]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-example-of-ilqr_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
wl = pkg.workloads.config3(B=8)
params = [pkg.copy_params(q, N=N) for q in wl.params]
eng = pkg.BatchedCILQR(params, wl.scenes)
eng.set_helper_mode(0)
eng.set_timing(True)
out = {}
for B in [int(b) for b in sys.argv[2:]] or (512, 1024, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 4096, 6144, 8192):
    x0 = np.repeat(wl.x0[3:4], B, axis=0)
    eng.solve_batch(x0)
    ms = []
    for _ in range(3):
        r = eng.solve_batch(x0)
        ms.append(eng.last_kernel_ms())
    out[B] = {"kernel_ms": round(min(ms), 3), "iters_each": int(r["res"]["iters"][0])}
eng.close()
print(json.dumps(out))
