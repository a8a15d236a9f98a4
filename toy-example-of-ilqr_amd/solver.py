"""Host-side mirror of the reference's solver interface on top of the C-ABI.

``CILQRSolver`` keeps the call shape of the reference class
(/root/reference/include/cilqr_solver.hpp:31-41: ctor from a config, ``solve(x0, ref_waypoints,
ref_velo, obs_preds, road_boaders) -> (u, x)``) for one ego vehicle; ``BatchedCILQR`` is the batch
form of the same call that the HIP kernels are built for.  All compute happens in
libcilqr_amd.so on the GPU — there is no CPU fallback here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (CilqrParams, CilqrScenarioDesc, RESULT_DTYPE, TRACE_DTYPE, check)
from .config import copy_params, params_from_config


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SceneTable:
    """Host arrays of one solve() argument set: lane samples, obstacle routes, borders, ref_velo."""

    def __init__(self, lane_x, lane_y, lane_yaw, obstacles, road_borders, ref_velo):
        self.lane_x = _f64(lane_x)
        self.lane_y = _f64(lane_y)
        self.lane_yaw = _f64(lane_yaw)
        obs = np.zeros((0, 1, 3)) if obstacles is None else _f64(obstacles)
        if obs.ndim != 3 or obs.shape[2] != 3:
            raise ValueError("obstacles must be [M][T][3]")
        self.obs = obs
        self.road_borders = _f64(road_borders)
        self.ref_velo = float(ref_velo)

    @classmethod
    def from_scenario(cls, sc):
        ln = sc.lane
        return cls(ln.x, ln.y, ln.yaw, sc.obstacles, sc.road_borders, sc.target_velocity)

    def desc(self):
        d = CilqrScenarioDesc()
        dp = C.POINTER(C.c_double)
        d.lane_x = self.lane_x.ctypes.data_as(dp)
        d.lane_y = self.lane_y.ctypes.data_as(dp)
        d.lane_yaw = self.lane_yaw.ctypes.data_as(dp)
        d.L = self.lane_x.shape[0]
        d.M = self.obs.shape[0]
        d.obs = self.obs.ctypes.data_as(dp) if self.obs.shape[0] else None
        d.T = self.obs.shape[1] if self.obs.shape[0] else 0
        d.road_borders[0] = float(self.road_borders[0])
        d.road_borders[1] = float(self.road_borders[1])
        d.ref_velo = self.ref_velo
        return d


class BatchedCILQR:
    """A device handle with its parameter table and scenario tables; batch entry points."""

    def __init__(self, params, scenes=None, device=0, dev=False):
        """dev=True loads libcilqr_amd_dev.so: the build that carries the testing aids (set_debug_flags) and the
        in-kernel cycle accounting (set_phase_profiling); same kernels, same results otherwise."""
        self._lib = _lib.load(dev)
        self.dev = bool(dev)
        self._h = C.c_void_p()
        self._check(self._lib.cilqr_create(int(device), C.byref(self._h)), "cilqr_create")
        self.device = int(device)
        self.set_params(params)
        if scenes is not None:  # solve_one() brings its scenario with every call
            self.set_scenarios(scenes)

    def _check(self, code, where):
        check(code, where, self._lib)

    # -- tables -------------------------------------------------------------------------------
    def set_params(self, params):
        plist = list(params) if isinstance(params, (list, tuple)) else [params]
        arr = (CilqrParams * len(plist))(*[copy_params(p) for p in plist])
        self._check(self._lib.cilqr_set_params(self._h, arr, len(plist)), "cilqr_set_params")
        self.params = plist
        self.N = int(plist[0].N)

    def set_scenarios(self, scenes):
        slist = list(scenes) if isinstance(scenes, (list, tuple)) else [scenes]
        self._scenes = slist  # keep host arrays alive during the upload
        arr = (CilqrScenarioDesc * len(slist))(*[s.desc() for s in slist])
        self._check(self._lib.cilqr_set_scenarios(self._h, arr, len(slist)), "cilqr_set_scenarios")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.cilqr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the path ------------------------------------------------------------------------------
    def solve_batch(self, x0, scenario_id=None, param_id=None, tick=None, last_u=None, trace_cap=0):
        """CILQRSolver::solve for B trajectories.  Returns dict(u, x, res[, trace])."""
        x0 = _f64(x0).reshape(-1, 4)
        B, N = x0.shape[0], self.N
        sid, pid, tk = _i32(scenario_id), _i32(param_id), _i32(tick)
        lu = None if last_u is None else _f64(last_u).reshape(B, N, 2)
        u = np.empty((B, N, 2))
        x = np.empty((B, N + 1, 4))
        res = np.zeros(B, dtype=RESULT_DTYPE)
        trace = np.zeros((B, trace_cap), dtype=TRACE_DTYPE) if trace_cap > 0 else None
        self._check(self._lib.cilqr_solve_batch(self._h, B, _p(x0), _p(sid), _p(pid), _p(tk), _p(lu), _p(u), _p(x),
                                          _p(res), _p(trace), int(trace_cap)), "cilqr_solve_batch")
        out = {"u": u, "x": x, "res": res}
        if trace is not None:
            out["trace"] = trace
        return out

    def solve_one(self, x0, scene, last_u=None):
        """cilqr_solve: CILQRSolver::solve for one ego with all arguments handed over (hpp:37-41); the tables
        stay resident in HBM between calls when they are unchanged.  Returns (u, x, res)."""
        N = self.N
        x0 = _f64(x0).reshape(4)
        lu = None if last_u is None else _f64(last_u).reshape(N, 2)
        u = np.empty((N, 2))
        x = np.empty((N + 1, 4))
        res = np.zeros(1, dtype=RESULT_DTYPE)
        d = scene.desc()
        self._check(self._lib.cilqr_solve(self._h, _p(x0), C.byref(d), _p(lu), _p(u), _p(x), _p(res)), "cilqr_solve")
        return u, x, res[0]

    def solve_cache_stats(self):
        up, re = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.cilqr_solve_cache_stats(self._h, C.byref(up), C.byref(re)), "cilqr_solve_cache_stats")
        return int(up.value), int(re.value)

    def solve_batch_device(self, B, d_x0, d_scenario_id, d_param_id, d_tick, d_last_u, d_u, d_x, d_res,
                           d_trace=0, trace_cap=0, stream=0):
        """Raw device-pointer form (ints): enqueue on `stream`, no synchronisation."""
        self._check(self._lib.cilqr_solve_batch_device(self._h, int(B), d_x0, d_scenario_id or None, d_param_id or None,
                                                 d_tick or None, d_last_u or None, d_u, d_x, d_res or None,
                                                 d_trace or None, int(trace_cap), stream or None),
              "cilqr_solve_batch_device")

    def advance_batch_device(self, B, d_x, d_x0, d_tick=0, stream=0):
        """ego_state = x.row(1) and tick += 1 for every trajectory, on the device (raw pointers as ints)"""
        self._check(self._lib.cilqr_advance_batch_device(self._h, int(B), d_x, d_x0, d_tick or None, stream or None),
              "cilqr_advance_batch_device")

    def closed_loop_batch_device(self, B, ticks, d_x0, d_scenario_id, d_param_id, d_tick, d_last_u, d_u, d_x, d_res,
                                 d_states=0, d_iters=0, stream=0):
        """`ticks` ticks of the planning loop for every ego in one launch (raw device pointers as ints): solve, ego <-
        x.row(1), tick += 1, warm start from the plan just made; d_x0 / d_tick are advanced in place."""
        self._check(self._lib.cilqr_closed_loop_batch_device(self._h, int(B), int(ticks), d_x0, d_scenario_id or None,
                                                             d_param_id or None, d_tick, d_last_u or None, d_u, d_x,
                                                             d_res or None, d_states or None, d_iters or None, stream or None),
                    "cilqr_closed_loop_batch_device")

    def set_batches_in_flight(self, k):
        """k launch slots (1 = default: launches of the handle are ordered): up to k solve_batch_device /
        closed_loop_batch_device calls in flight at once, each on an internal stream with scratch of its own, the tables
        shared; the caller's stream sees the results after join_device(stream) (or wait())."""
        self._check(self._lib.cilqr_set_batches_in_flight(self._h, int(k)), "cilqr_set_batches_in_flight")

    def join_device(self, stream=0):
        """make `stream` wait for every launch of this handle that is in flight"""
        self._check(self._lib.cilqr_join_device(self._h, stream or None), "cilqr_join_device")

    def wait(self):
        """block the host until every launch of this handle has finished"""
        self._check(self._lib.cilqr_wait(self._h), "cilqr_wait")

    def slot_kernel_ms(self, k):
        ms = C.c_float(0)
        self._check(self._lib.cilqr_slot_kernel_ms(self._h, int(k), C.byref(ms)), "cilqr_slot_kernel_ms")
        return float(ms.value)

    def set_timing(self, on=True):
        self._check(self._lib.cilqr_set_timing(self._h, 1 if on else 0), "cilqr_set_timing")

    def set_phase_profiling(self, on=True):
        self._check(self._lib.cilqr_set_phase_profiling(self._h, 1 if on else 0), "cilqr_set_phase_profiling")

    def phase_cycles(self, B):
        """[B][10]: cycles of init, derivatives, backward, rollout, trial cost, accept, total; then
        iterations, serial-ref-scan fallbacks, trials"""
        out = np.zeros((B, _lib.PROF_SLOTS), dtype=np.int64)
        self._check(self._lib.cilqr_get_phase_cycles(self._h, _p(out), int(B)), "cilqr_get_phase_cycles")
        return out

    def set_helper_mode(self, mode):
        """-1 automatic, 0 one wavefront per trajectory, 1 main + helper wavefront"""
        self._check(self._lib.cilqr_set_helper_mode(self._h, int(mode)), "cilqr_set_helper_mode")

    def last_launch_info(self):
        """shape of the most recent fused launch: trajectories per wavefront, grid blocks, threads per block, window samples"""
        out = (C.c_int32 * 4)()
        self._check(self._lib.cilqr_last_launch_info(self._h, out), "cilqr_last_launch_info")
        return {"trajectories_per_wavefront": int(out[0]), "blocks": int(out[1]), "threads_per_block": int(out[2]),
                "lane_window_samples": int(out[3])}

    def set_group_mode(self, mode):
        """-1 automatic (two per wavefront beyond the helper range, barrier mode, every horizon), 0 / 1 one trajectory per
        wavefront, 2 two per wavefront wherever that build can run"""
        self._check(self._lib.cilqr_set_group_mode(self._h, int(mode)), "cilqr_set_group_mode")

    def set_rollout_mode(self, mode):
        """-1 adaptive, 0 all 20 step sizes in one rollout pass, 1 the first trial alone first"""
        self._check(self._lib.cilqr_set_rollout_mode(self._h, int(mode)), "cilqr_set_rollout_mode")


    def set_block_timeline(self, on=True):
        self._check(self._lib.cilqr_set_block_timeline(self._h, 1 if on else 0), "cilqr_set_block_timeline")

    def block_timeline(self, B):
        """[B][4] int64: start, end (100 MHz ticks), block index, XCC of the block that solved each trajectory"""
        out = np.zeros((B, 4), dtype=np.int64)
        self._check(self._lib.cilqr_get_block_timeline(self._h, _p(out), int(B)), "cilqr_get_block_timeline")
        return out

    def set_work_sharing(self, mode):
        """1 (default): finished blocks cost line-search trials of the trajectories still being solved (horizons
        above 63, barrier mode, large batches); 0: off.  Same results either way."""
        self._check(self._lib.cilqr_set_work_sharing(self._h, int(mode)), "cilqr_set_work_sharing")

    def work_sharing_stats(self):
        """counters of the last launch that shared work: searches announced, trial costs delivered by other blocks,
        blocks that stayed to help, error flag"""
        out = (C.c_uint32 * 4)()
        self._check(self._lib.cilqr_work_sharing_stats(self._h, out), "cilqr_work_sharing_stats")
        return {"announced": int(out[0]), "helped": int(out[1]), "helpers": int(out[2]), "error": int(out[3])}

    def set_resume_iters(self, iters):
        """iterations per slice of the resumable / sliced solves (batches larger than the chip holds at once: a solve runs
        this many iterations at a time and is parked in between, so that long solves do not finish alone at the end of a
        launch); -1 = automatic (32 for lone wavefronts, 16 / 12 for trajectories in pairs), 0 = off.  Same results either way."""
        self._check(self._lib.cilqr_set_resume_iters(self._h, int(iters)), "cilqr_set_resume_iters")

    def resume_stats(self):
        """how many times a solve was parked in the last launch that ran resumable solves"""
        n = C.c_uint32(0)
        self._check(self._lib.cilqr_resume_stats(self._h, C.byref(n)), "cilqr_resume_stats")
        return int(n.value)

    def set_debug_flags(self, flags):
        self._check(self._lib.cilqr_set_debug_flags(self._h, int(flags)), "cilqr_set_debug_flags")

    def set_alm_state(self, mu=None, rho=None):
        B = (mu.shape[0] if mu is not None else np.asarray(rho).shape[0])
        mu = None if mu is None else _f64(mu)
        rho = None if rho is None else _f64(rho)
        self._check(self._lib.cilqr_set_alm_state(self._h, int(B), _p(mu), _p(rho)), "cilqr_set_alm_state")

    def get_alm_state(self, B):
        cols = C.c_int32(0)
        self._check(self._lib.cilqr_get_alm_state(self._h, int(B), None, None, None, C.byref(cols)), "cilqr_get_alm_state")
        mu = np.empty((B, self.N, cols.value))
        mun = np.empty((B, self.N, cols.value))
        rho = np.empty(B)
        self._check(self._lib.cilqr_get_alm_state(self._h, int(B), _p(mu), _p(mun), _p(rho), C.byref(cols)),
              "cilqr_get_alm_state")
        return mu, mun, rho

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.cilqr_last_kernel_ms(self._h, C.byref(ms)), "cilqr_last_kernel_ms")
        return float(ms.value)

    # -- pieces --------------------------------------------------------------------------------
    def init_traj(self, x0, param_id=None):
        x0 = _f64(x0).reshape(-1, 4)
        B = x0.shape[0]
        x = np.empty((B, self.N + 1, 4))
        self._check(self._lib.cilqr_init_traj_batch(self._h, B, _p(x0), _p(_i32(param_id)), _p(x)), "cilqr_init_traj_batch")
        return x

    def ref_points(self, x, scenario_id=None, param_id=None):
        x = _f64(x).reshape(-1, self.N + 1, 4)
        B = x.shape[0]
        ref = np.empty((B, self.N + 1, 3))
        idx = np.empty((B, self.N + 1), dtype=np.int32)
        self._check(self._lib.cilqr_ref_points_batch(self._h, B, _p(x), _p(_i32(scenario_id)), _p(_i32(param_id)),
                                               _p(ref), _p(idx)), "cilqr_ref_points_batch")
        return ref, idx

    def total_cost(self, u, x, scenario_id=None, param_id=None, tick=None):
        u = _f64(u).reshape(-1, self.N, 2)
        x = _f64(x).reshape(-1, self.N + 1, 4)
        B = x.shape[0]
        J = np.empty(B)
        self._check(self._lib.cilqr_total_cost_batch(self._h, B, _p(u), _p(x), _p(_i32(scenario_id)), _p(_i32(param_id)),
                                               _p(_i32(tick)), _p(J)), "cilqr_total_cost_batch")
        return J

    def forward_pass(self, u, x, d, K, scenario_id=None, param_id=None, tick=None, n_alpha=_lib.MAX_ALPHA_TRIALS):
        N = self.N
        u = _f64(u).reshape(-1, N, 2)
        x = _f64(x).reshape(-1, N + 1, 4)
        d = _f64(d).reshape(-1, N, 2)
        K = _f64(K).reshape(-1, N, 2, 4)
        B = x.shape[0]
        nu = np.empty((B, n_alpha, N, 2))
        nx = np.empty((B, n_alpha, N + 1, 4))
        J = np.empty((B, n_alpha))
        self._check(self._lib.cilqr_forward_pass_batch(self._h, B, _p(u), _p(x), _p(d), _p(K), _p(_i32(scenario_id)),
                                                 _p(_i32(param_id)), _p(_i32(tick)), int(n_alpha), _p(nu), _p(nx),
                                                 _p(J)), "cilqr_forward_pass_batch")
        return nu, nx, J

    def cost_derivatives(self, u, x, scenario_id=None, param_id=None, tick=None):
        N = self.N
        u = _f64(u).reshape(-1, N, 2)
        x = _f64(x).reshape(-1, N + 1, 4)
        B = x.shape[0]
        out = {"l_x": np.empty((B, N + 1, 4)), "l_u": np.empty((B, N, 2)), "l_xx": np.empty((B, N + 1, 4, 4)),
               "l_uu": np.empty((B, N, 2, 2)), "A": np.empty((B, N, 4, 4)), "B": np.empty((B, N, 4, 2))}
        self._check(self._lib.cilqr_cost_derivatives_batch(self._h, B, _p(u), _p(x), _p(_i32(scenario_id)),
                                                     _p(_i32(param_id)), _p(_i32(tick)), _p(out["l_x"]),
                                                     _p(out["l_u"]), _p(out["l_xx"]), _p(out["l_uu"]), _p(out["A"]),
                                                     _p(out["B"])), "cilqr_cost_derivatives_batch")
        return out

    def backward_pass(self, u, x, lamb, scenario_id=None, param_id=None, tick=None):
        N = self.N
        u = _f64(u).reshape(-1, N, 2)
        x = _f64(x).reshape(-1, N + 1, 4)
        B = x.shape[0]
        lamb = _f64(np.broadcast_to(np.asarray(lamb, dtype=np.float64), (B,)))
        d = np.empty((B, N, 2))
        K = np.empty((B, N, 2, 4))
        dV = np.empty((B, 2))
        status = np.empty(B, dtype=np.int32)
        self._check(self._lib.cilqr_backward_pass_batch(self._h, B, _p(u), _p(x), _p(lamb), _p(_i32(scenario_id)),
                                                  _p(_i32(param_id)), _p(_i32(tick)), _p(d), _p(K), _p(dV),
                                                  _p(status)), "cilqr_backward_pass_batch")
        return d, K, dV, status

    def detmath(self, func, x, y=None):
        x = _f64(x).ravel()
        y = None if y is None else _f64(y).ravel()
        out = np.empty_like(x)
        self._check(self._lib.cilqr_detmath_eval(self._h, int(func), _p(x), _p(y), x.shape[0], _p(out)), "cilqr_detmath_eval")
        return out


class CILQRSolver:
    """Drop-in counterpart of the reference class for ONE ego vehicle (B = 1 per call).

    ``CILQRSolver(config)`` then ``solve(x0, ref_waypoints, ref_velo, obs_preds, road_boaders)``
    returning ``(u[N][2], x[N+1][4])`` as in include/cilqr_solver.hpp:34-41.  State carried across
    calls as upstream: ``is_first_solve`` / ``last_solve_u`` for ``use_last_solution`` (cs:97-102,144).
    ``obs_preds`` is a list of objects with x/y/yaw arrays, or an array [M][>=N+1][3], holding the
    obstacle predictions from the current tick on (what utils::get_sub_routing_lines returns).
    """

    def __init__(self, config, device=0, **param_overrides):
        self.params = params_from_config(config, **param_overrides)
        self.device = device
        self.is_first_solve = True
        self.last_solve_u = None
        self.last_result = None
        self._engine = None

    def solve(self, x0, ref_waypoints, ref_velo, obs_preds, road_boaders):
        if isinstance(obs_preds, np.ndarray):
            obs = obs_preds
        elif len(obs_preds) == 0:
            obs = None
        else:
            obs = np.stack([np.stack([np.asarray(r.x), np.asarray(r.y), np.asarray(r.yaw)], axis=1)
                            for r in obs_preds])
        scene = SceneTable(ref_waypoints.x, ref_waypoints.y, ref_waypoints.yaw, obs, road_boaders, ref_velo)
        if self._engine is None:
            self._engine = BatchedCILQR(self.params, None, self.device)
        warm = (not self.is_first_solve) and bool(self.params.use_last_solution)
        u, x, res = self._engine.solve_one(x0, scene, last_u=self.last_solve_u if warm else None)
        self.is_first_solve = False
        self.last_solve_u = u.copy()
        self.last_result = res
        return u, x
