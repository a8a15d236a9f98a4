#!/usr/bin/env python3
"""GPU box: one workload under the development library's CILQR_TUNE settings (given as arguments): kernel time, hand-over
counters, resident slots over time.   scripts/slice_probe.py CONFIG "tune1" "tune2" ..."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("toy-example-of-ilqr_amd")
cfg = int(sys.argv[1])
wl = {2: pkg.workloads.config2, 3: pkg.workloads.config3, 5: pkg.workloads.config5}.get(cfg)
wl = wl() if wl else pkg.workloads.config4(B=8192, N=100)
for tune in sys.argv[2:]:
    os.environ["CILQR_TUNE"] = tune
    eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True)
    eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    eng.set_block_timeline(True)
    eng.set_timing(True)
    eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    kms = eng.last_kernel_ms()
    tl = eng.block_timeline(wl.B)
    st = {"parked": eng.resume_stats(), "sharing": eng.work_sharing_stats()}
    eng.close()
    t0 = tl[:, 0].min()
    s, e = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0
    span = e.max()
    edges = np.linspace(0, span, 21)
    res = [int(((s < edges[i + 1]) & (e > edges[i])).sum()) for i in range(20)]
    print(json.dumps({"tune": tune, "workload": wl.name, "kernel_ms": kms, **st, "unfinished_started_in_20_slices": res,
                      "last_start_ms": float(s.max() / 1e3)}))
