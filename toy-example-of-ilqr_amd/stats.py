"""Per-rank statistics of a batch solve and their reduction across ranks.

The solve path shards by trajectory and has no exchange step, so the only collective in a
multi-GPU run is this reduction of ~10 counters (RCCL over xGMI on the GPUs; gloo in the CPU tests)
plus a MAX over the per-rank wall times."""
import numpy as np

from .workloads import bytes_per_iteration

FIELDS = ("iters", "ls_trials", "converged", "max_lamb", "max_iter", "sum_J_final", "nan_costs",
          "algorithmic_bytes", "trajectories", "cost_evals")


def local_stats(res, N, M_of):
    """res: structured array (RESULT_DTYPE) of this rank's trajectories; M_of: obstacles per trajectory."""
    return np.array([res["iters"].sum(), res["ls_trials"].sum(), (res["end_reason"] == 0).sum(),
                     (res["end_reason"] == 1).sum(), (res["end_reason"] == 2).sum(),
                     np.nansum(res["J_final"]), np.isnan(res["J_final"]).sum(),
                     float((res["iters"] * bytes_per_iteration(N, np.asarray(M_of))).sum()),
                     res.shape[0], res["cost_evals"].sum()], dtype=np.float64)


def reduce_stats(vec, elapsed, dist=None, device=None):
    """SUM of the counters and MAX of the elapsed time over all ranks (identity when dist is None)."""
    if dist is None:
        return np.asarray(vec, dtype=np.float64), float(elapsed)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.from_numpy(np.asarray(vec, dtype=np.float64)).to(device) if device is not None else torch.from_numpy(
        np.asarray(vec, dtype=np.float64).copy())
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s.cpu().numpy(), float(t.item())


def as_dict(vec):
    return {k: float(v) for k, v in zip(FIELDS, vec)}
