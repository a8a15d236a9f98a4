#!/usr/bin/env python3
"""Runs ON THE GPU BOX: does the order of the trajectories in the batch matter?  Workgroups go to the chip's eight
XCDs round-robin (block b -> XCD b mod 8), each XCD then works through its own share: a batch whose work per
trajectory correlates with b mod 8 (config 5: parameter set = b mod 16) loads the XCDs unevenly.
usage: scripts/order_probe.py [config]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-example-of-ilqr_amd")
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
wl = {2: pkg.workloads.config2, 3: pkg.workloads.config3, 5: pkg.workloads.config5}.get(cfg)
wl = wl() if wl else pkg.workloads.config4(B=8192, N=100)
eng = pkg.BatchedCILQR(wl.params, wl.scenes)
eng.set_timing(True)
B = wl.B
ident = np.arange(B)
rng = np.random.default_rng(1)
q, k = ident // 8, ident % 8
orders = {"as generated": ident, "random permutation": rng.permutation(B),
          "rotated within groups of 8 by q>>1": 8 * q + ((k + (q >> 1)) & 7)}
res0 = None
out = {}
for name, perm in orders.items():
    def take(a):
        return None if a is None else a[perm]
    args = (wl.x0[perm], take(wl.scenario_id), take(wl.param_id), take(wl.tick))
    r = eng.solve_batch(*args)
    ms = []
    for _ in range(4):
        r = eng.solve_batch(*args)
        ms.append(eng.last_kernel_ms())
    # same results, wherever a trajectory sits in the batch
    inv = np.empty(B, dtype=np.int64); inv[perm] = ident
    if res0 is None:
        res0 = r
    else:
        assert np.array_equal(r["x"][inv].view(np.uint64), res0["x"].view(np.uint64))
    out[name] = {"kernel_ms": round(min(ms), 3), "it_per_s": round(float(r["res"]["iters"].sum()) / min(ms) * 1e3)}
# work per XCD in the generated order
it = res0["res"]["iters"].astype(np.float64); tr = res0["res"]["ls_trials"].astype(np.float64)
w = 2.0 * it + tr
out["work_per_xcd_as_generated (2 it + trials), relative to mean"] = [round(float(w[ident % 8 == x].sum() / (w.sum() / 8)), 3) for x in range(8)]
eng.close()
print(json.dumps(out, indent=1))
