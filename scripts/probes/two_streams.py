import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cilqr_amd as pkg
torch.cuda.init(); dev = torch.device("cuda", 0)
wl = pkg.workloads.config3(); B, N = wl.B, wl.N
d_x0 = torch.from_numpy(wl.x0).to(dev); d_sid = torch.from_numpy(wl.scenario_id).to(dev); d_pid = torch.from_numpy(wl.param_id).to(dev); d_tick = torch.from_numpy(wl.tick).to(dev)
pool = [torch.cuda.Stream(dev) for _ in range(8)]
def run(idx, steps=24):
    S = len(idx)
    engs = [pkg.BatchedCILQR(wl.params, wl.scenes) for _ in range(S)]
    outs = [(torch.empty((B, N, 2), dtype=torch.float64, device=dev), torch.empty((B, N + 1, 4), dtype=torch.float64, device=dev),
             torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)) for _ in range(S)]
    def step(i):
        e, o, s = engs[i % S], outs[i % S], pool[idx[i % S]]
        e.solve_batch_device(B, d_x0.data_ptr(), d_sid.data_ptr(), d_pid.data_ptr(), d_tick.data_ptr(), 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), 0, 0, s.cuda_stream)
    for i in range(2 * S): step(i)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(dev)
    t = (time.perf_counter() - t) / steps * 1e3
    for e in engs: e.close()
    return t
for idx in ([0], [0, 1], [0, 2], [1, 2], [0, 3], [0, 4], [0, 1, 2], [3, 4, 5], [0, 1, 2, 3]):
    print("streams", idx, "ms per batch %.3f" % run(idx), flush=True)
