import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cilqr_amd as pkg, ctypes as C
wl = pkg.workloads.config4(B=int(sys.argv[1]) if len(sys.argv) > 1 else 8192)
eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True)
eng.set_timing(True)
for rep in range(3):
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    ms = eng.last_kernel_ms()
    st = eng.work_sharing_stats()
    w = (C.c_uint32 * 16)()
    print(os.environ.get("CILQR_TUNE"), "kernel_ms %.2f" % ms, st, "iters", int(out["res"]["iters"].sum()))
