"""Synthetic batched workloads: the configurations BASELINE.json names.

Random numbers come from a counter-based generator (splitmix64 of seed + counter -> uniform ->
Box-Muller) so that any implementation (this file, C++, the oracle harness) can regenerate exactly
the same initial states from (seed, b).
"""
import numpy as np

from .config import GlobalConfig, copy_params, params_from_config
from .scenario import build_scenario
from .solver import SceneTable

_M64 = (1 << 64) - 1


def splitmix64(v):
    v = (v + 0x9E3779B97F4A7C15) & _M64
    v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & _M64
    return v ^ (v >> 31)


def uniform01(seed, counter):
    """(0, 1) from the top 53 bits of splitmix64(seed ^ splitmix64(counter))."""
    r = splitmix64((seed ^ splitmix64(counter & _M64)) & _M64)
    return ((r >> 11) + 0.5) / float(1 << 53)


def normal(seed, counter):
    u1 = uniform01(seed, 2 * counter)
    u2 = uniform01(seed, 2 * counter + 1)
    return float(np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2))


def perturbed_starts(base, B, seed, first=0):
    """x0_b = base + (U(-5,5), +-U(0.05,1.0), N(0,0.5), N(0,0.02)); |dy| >= 0.05 keeps the start off
    the singular reference line (SURVEY.md §7 hard part 1).  `first` = global index of row 0, so a
    shard of a larger batch regenerates exactly its own rows.  Counter of component i of row b: 8 b + i (the
    normal components consume counters 2 c and 2 c + 1 of their own stream).  Host C++ twin:
    cilqr_perturbed_starts (csrc/scenario.cpp)."""
    out = np.empty((B, 4))
    for i in range(B):
        b = first + i
        dx = -5.0 + 10.0 * uniform01(seed, 8 * b + 0)
        mag = 0.05 + 0.95 * uniform01(seed, 8 * b + 1)
        sgn = 1.0 if uniform01(seed, 8 * b + 2) < 0.5 else -1.0
        dv = 0.5 * normal(seed, 8 * b + 3)
        dyaw = 0.02 * normal(seed, 8 * b + 4)
        out[i] = (base[0] + dx, base[1] + sgn * mag, base[2] + dv, base[3] + dyaw)
    return out


class Workload:
    """params table + scenario tables + per-trajectory arrays of one benchmark configuration."""

    def __init__(self, name, params, scenes, x0, scenario_id=None, param_id=None, tick=None):
        self.name = name
        self.params = params
        self.scenes = scenes
        self.x0 = np.ascontiguousarray(x0)
        B = self.x0.shape[0]
        self.scenario_id = np.zeros(B, np.int32) if scenario_id is None else np.asarray(scenario_id, np.int32)
        self.param_id = np.zeros(B, np.int32) if param_id is None else np.asarray(param_id, np.int32)
        self.tick = np.zeros(B, np.int32) if tick is None else np.asarray(tick, np.int32)

    @property
    def B(self):
        return self.x0.shape[0]

    @property
    def N(self):
        return int(self.params[0].N)

    @property
    def M_of(self):
        return np.array([s.obs.shape[0] for s in self.scenes])[self.scenario_id]

    def shard(self, rank, world):
        """contiguous block of ceil(B/world) trajectories for `rank`."""
        per = -(-self.B // world)
        lo, hi = min(rank * per, self.B), min((rank + 1) * per, self.B)
        return Workload(self.name, self.params, self.scenes, self.x0[lo:hi], self.scenario_id[lo:hi],
                        self.param_id[lo:hi], self.tick[lo:hi])


def _scene(name):
    cfg = GlobalConfig.get_instance(name)
    return cfg, build_scenario(cfg, name)


def config1(N=50):
    """scenario_two_straight, single ego, N=50, the YAML start (ill-conditioned: ego on the line)."""
    cfg, sc = _scene("two_straight")
    p = params_from_config(cfg, N=N)
    return Workload("config1_two_straight_single", [p], [SceneTable.from_scenario(sc)], sc.ego_state[None])


def config2(B=1024, N=50, first=0, seed=0xC11A0002):
    """batch of synthetic straight-lane scenarios (two_straight geometry/params, 3 obstacles)."""
    cfg, sc = _scene("two_straight")
    p = params_from_config(cfg, N=N)
    x0 = perturbed_starts(sc.ego_state, B, seed, first)
    return Workload(f"config2_straight_B{B}_N{N}", [p], [SceneTable.from_scenario(sc)], x0)


def config3(B=8192, N=50, first=0, seed=0xC11A0003):
    """three_bend obstacle set replicated, perturbed initial states."""
    cfg, sc = _scene("three_bend")
    p = params_from_config(cfg, N=N)
    x0 = perturbed_starts(sc.ego_state, B, seed, first)
    return Workload(f"config3_bend_B{B}_N{N}", [p], [SceneTable.from_scenario(sc)], x0)


def config4(B=65536, N=100, first=0, seed=0xC11A0004):
    """mixed straight/bend scenarios, N=100: scenario alternates by (global index mod 4)."""
    names = ("two_straight", "three_bend", "two_borrow", "three_straight")
    params, scenes, egos = [], [], []
    for nm in names:
        cfg, sc = _scene(nm)
        params.append(params_from_config(cfg, N=N, use_last_solution=0))
        scenes.append(SceneTable.from_scenario(sc))
        egos.append(sc.ego_state)
    gid = first + np.arange(B)
    sid = (gid % 4).astype(np.int32)
    x0 = np.empty((B, 4))
    for k in range(4):
        rows = np.nonzero(sid == k)[0]
        if rows.size:
            # regenerate by global index so shards agree with the full batch
            for r in rows:
                x0[r] = perturbed_starts(egos[k], 1, seed, first + int(r))[0]
    return Workload(f"config4_mixed_B{B}_N{N}", params, scenes, x0, scenario_id=sid, param_id=sid)


SWEEP_Q1 = (2.75, 5.5, 11.0, 22.0)
SWEEP_Q2 = (2.875, 4.3125, 5.75, 8.625)


def config5(B_base=4096, N=50, first=0, seed=0xC11A0005):
    """barrier-weight sweep: 16 (obstacle_exp_q1, q2) settings per base start (three_bend)."""
    cfg, sc = _scene("three_bend")
    p0 = params_from_config(cfg, N=N)
    params = [copy_params(p0, obstacle_exp_q1=q1, obstacle_exp_q2=q2) for q1 in SWEEP_Q1 for q2 in SWEEP_Q2]
    base = perturbed_starts(sc.ego_state, B_base, seed, first)
    x0 = np.repeat(base, 16, axis=0)
    pid = np.tile(np.arange(16, dtype=np.int32), B_base)
    return Workload(f"config5_sweep_B{B_base}x16_N{N}", params, [SceneTable.from_scenario(sc)], x0, param_id=pid)


def bytes_per_iteration(N, M):
    """ALGORITHMIC bytes of one trajectory-iteration:
    read (x,u) and the obstacle block, write new (x,u)."""
    return 16 * (6 * N + 4) + 24 * M * (N + 1)
