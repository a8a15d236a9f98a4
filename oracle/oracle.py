"""ctypes wrapper of the CPU oracle (oracle/cilqr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
shipped package (toy-example-of-ilqr_amd/) never does.  Two builds are wrapped:
``Oracle("det")``  -> liboracle_det.so  (elementary functions from csrc/detmath.h: bit-identical to
the HIP kernels) and ``Oracle("libm")`` -> liboracle_libm.so (glibc libm, as the reference uses); ``Oracle("fused")`` ->
liboracle_fused.so, the round-4 experiment (detmath + explicit fma at four named groups of sites); ``Oracle("tree")`` /
``Oracle("pkt")`` -> the round-6 experiment builds of the libm flavour (ORC_SUM4: another association of four-term sums).
"""
import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent

PARAM_FIELDS = [
    ("N", C.c_int32), ("max_iter", C.c_int32), ("solve_type", C.c_int32),
    ("reference_point", C.c_int32), ("use_last_solution", C.c_int32), ("reserved0", C.c_int32),
    ("dt", C.c_double),
    ("w_pos", C.c_double), ("w_vel", C.c_double), ("w_yaw", C.c_double), ("w_acc", C.c_double),
    ("w_stl", C.c_double),
    ("obstacle_exp_q1", C.c_double), ("obstacle_exp_q2", C.c_double),
    ("state_exp_q1", C.c_double), ("state_exp_q2", C.c_double),
    ("alm_rho_init", C.c_double), ("alm_gamma", C.c_double), ("max_rho", C.c_double),
    ("max_mu", C.c_double),
    ("init_lamb", C.c_double), ("lamb_decay", C.c_double), ("lamb_amplify", C.c_double),
    ("max_lamb", C.c_double),
    ("convergence_threshold", C.c_double), ("accept_step_threshold", C.c_double),
    ("wheelbase", C.c_double), ("width", C.c_double), ("length", C.c_double),
    ("velo_max", C.c_double), ("velo_min", C.c_double), ("yaw_lim", C.c_double),
    ("acc_max", C.c_double), ("acc_min", C.c_double), ("stl_lim", C.c_double),
    ("d_safe", C.c_double),
]


class OrcParams(C.Structure):
    _fields_ = PARAM_FIELDS


class OrcScene(C.Structure):
    _fields_ = [("lane_x", C.POINTER(C.c_double)), ("lane_y", C.POINTER(C.c_double)),
                ("lane_yaw", C.POINTER(C.c_double)), ("L", C.c_int32), ("M", C.c_int32),
                ("obs", C.POINTER(C.c_double)), ("T", C.c_int32), ("tick", C.c_int32),
                ("road_borders", C.c_double * 2), ("ref_velo", C.c_double)]


RESULT_DTYPE = np.dtype([("J_init", "<f8"), ("J_final", "<f8"), ("iters", "<i4"), ("end_reason", "<i4"),
                         ("final_status", "<i4"), ("ls_trials", "<i4"), ("cost_evals", "<i4"),
                         ("trace_len", "<i4")])
TRACE_DTYPE = np.dtype([("status", "<i4"), ("trials", "<i4"), ("accepted", "<i4"), ("alpha_idx", "<i4"),
                        ("lamb", "<f8"), ("new_J", "<f8")])
MARGIN_DTYPE = np.dtype([("ls", "<f8"), ("pd", "<f8"), ("ref", "<f8")])  # orc_margin_rec


def ensure_built():
    subprocess.run(["make", "-C", str(HERE), "all"], check=True, stdout=subprocess.DEVNULL)


def make_params(src):
    """OrcParams from a dict or from any ctypes struct with the same field names."""
    p = OrcParams()
    for name, _ in PARAM_FIELDS:
        v = src[name] if isinstance(src, dict) else getattr(src, name)
        setattr(p, name, v)
    return p


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Scene:
    """Host arrays + the orc_scene struct pointing into them."""

    def __init__(self, lane_x, lane_y, lane_yaw, obstacles, road_borders, ref_velo, tick=0):
        self.lane_x, self.lane_y, self.lane_yaw = _f64(lane_x), _f64(lane_y), _f64(lane_yaw)
        self.obs = np.zeros((0, 1, 3)) if obstacles is None else _f64(obstacles)
        self.road_borders = _f64(road_borders)
        self.ref_velo = float(ref_velo)
        self.tick = int(tick)

    def struct(self, tick=None):
        s = OrcScene()
        dp = C.POINTER(C.c_double)
        s.lane_x = self.lane_x.ctypes.data_as(dp)
        s.lane_y = self.lane_y.ctypes.data_as(dp)
        s.lane_yaw = self.lane_yaw.ctypes.data_as(dp)
        s.L = self.lane_x.shape[0]
        s.M = self.obs.shape[0]
        s.obs = self.obs.ctypes.data_as(dp) if s.M else None
        s.T = self.obs.shape[1] if s.M else 0
        s.tick = self.tick if tick is None else int(tick)
        s.road_borders[0] = float(self.road_borders[0])
        s.road_borders[1] = float(self.road_borders[1])
        s.ref_velo = self.ref_velo
        return s


class Oracle:
    def __init__(self, mode="det"):
        assert mode in ("det", "libm", "fused", "rec", "det!", "tree", "pkt")
        # round-4 experiment: CILQR_ORACLE_FUSED=1 makes "det" mean the fused-flavour build (detmath + explicit fma at four
        # named groups of sites), the checker of a device library compiled with -DCILQR_FUSED; "det!" = the detmath build
        # whatever the environment says (the test that compares the two flavours)
        import os
        if mode == "det" and os.environ.get("CILQR_ORACLE_FUSED") == "1":
            mode = "fused"
        if mode == "det!":
            mode = "det"
        path = HERE / f"liboracle_{mode}.so"
        if not path.exists():
            ensure_built()
        self.mode = mode
        lib = C.CDLL(str(path))
        V, I, D = C.c_void_p, C.c_int32, C.c_double
        sig = {
            "orc_math_mode": (C.c_int, []),
            "orc_create": (V, [C.POINTER(OrcParams)]),
            "orc_destroy": (None, [V]),
            "orc_reset": (None, [V]),
            "orc_solve": (C.c_int, [V, V, C.POINTER(OrcScene), V, V, V, V, I]),
            "orc_set_margin_buffer": (None, [V, V, I]),
            "orc_set_alm_state": (None, [V, V, D, I]),
            "orc_get_alm_next": (None, [V, V]),
            "orc_solve_batch": (C.c_int, [C.POINTER(OrcParams), I, C.POINTER(OrcScene), I, I, V, V, V, V, I, V, V, V]),
            "orc_kinematic_propagate": (None, [V, V, D, D, I, V]),
            "orc_model_derivatives": (None, [V, V, D, D, I, I, V, V]),
            "orc_front_rear_centers": (None, [V, D, I, V, V]),
            "orc_front_rear_center_derivatives": (None, [D, D, I, V, V]),
            "orc_ellipsoid_scales": (None, [V, D, V]),
            "orc_ellipsoid_safety_margin": (D, [V, V, V]),
            "orc_ellipsoid_safety_margin_derivatives": (None, [V, V, V, V]),
            "orc_exp_barrier": (D, [D, D, D]),
            "orc_exp_barrier_derivative_and_Hessian": (None, [D, V, I, D, D, V, V]),
            "orc_obstacle_constr": (None, [C.POINTER(OrcParams), V, V, V]),
            "orc_obstacle_constr_derivatives": (None, [C.POINTER(OrcParams), V, V, V, V]),
            "orc_ref_exact_points": (None, [V, I, C.POINTER(OrcScene), V, V]),
            "orc_const_velo_prediction": (None, [C.POINTER(OrcParams), V, V]),
            "orc_total_cost": (D, [V, V, V, C.POINTER(OrcScene)]),
            "orc_cost_derivatives": (None, [V, V, V, C.POINTER(OrcScene), V, V, V, V]),
            "orc_backward_pass": (C.c_int, [V, V, V, D, C.POINTER(OrcScene), V, V, V]),
            "orc_forward_pass": (None, [C.POINTER(OrcParams), V, V, V, V, D, V, V]),
            "orc_m_exp": (D, [D]), "orc_m_sin": (D, [D]), "orc_m_cos": (D, [D]), "orc_m_tan": (D, [D]),
            "orc_m_atan": (D, [D]), "orc_m_hypot": (D, [D, D]),
            "orc_m_vec": (None, [I, V, V, C.c_int64, V]),
            "orc_alm_item": (D, [D, D, D]),
            "orc_lagrangian_derivative_and_Hessian": (None, [D, V, I, D, D, V, V]),
        }
        if mode == "rec":
            sig["orc_record_math"] = (None, [V, C.c_long])
            sig["orc_record_count"] = (C.c_long, [])
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        self.lib = lib
        assert lib.orc_math_mode() == (0 if mode in ("libm", "rec", "tree", "pkt") else 1)
        lib.orc_fused.restype = C.c_int
        assert lib.orc_fused() == (1 if mode == "fused" else 0)
        # round-6 experiment builds (libm flavour): four-term inner products as adjacent / interleaved pairs (ORC_SUM4)
        lib.orc_sum4.restype = C.c_int
        assert lib.orc_sum4() == {"tree": 1, "pkt": 2}.get(mode, 0)

    # ---- solver object ----
    def solver(self, params):
        return OracleSolver(self, params)

    def solve_batch(self, params, scenes, x0, scene_id=None, param_id=None, tick=None, n_threads=1):
        """Fresh cold-start solves of B trajectories (OpenMP over trajectories)."""
        plist = [make_params(p) for p in (params if isinstance(params, (list, tuple)) else [params])]
        slist = scenes if isinstance(scenes, (list, tuple)) else [scenes]
        parr = (OrcParams * len(plist))(*plist)
        sarr = (OrcScene * len(slist))(*[s.struct() for s in slist])
        x0 = _f64(x0).reshape(-1, 4)
        B, N = x0.shape[0], plist[0].N
        sid = None if scene_id is None else np.ascontiguousarray(scene_id, dtype=np.int32)
        pid = None if param_id is None else np.ascontiguousarray(param_id, dtype=np.int32)
        tk = None if tick is None else np.ascontiguousarray(tick, dtype=np.int32)
        u = np.empty((B, N, 2))
        x = np.empty((B, N + 1, 4))
        res = np.zeros(B, dtype=RESULT_DTYPE)
        rc = self.lib.orc_solve_batch(parr, len(plist), sarr, len(slist), B, _p(x0), _p(sid), _p(pid), _p(tk),
                                      int(n_threads), _p(u), _p(x), _p(res))
        if rc != 0:
            raise RuntimeError(f"orc_solve_batch rc={rc}")
        return {"u": u, "x": x, "res": res}

    # ---- leaf functions ----
    def propagate(self, x, u, dt, wb, rp):
        x, u, out = _f64(x), _f64(u), np.empty(4)
        self.lib.orc_kinematic_propagate(_p(x), _p(u), dt, wb, rp, _p(out))
        return out

    def model_derivatives(self, x, u, dt, wb, N, rp):
        x, u = _f64(x), _f64(u)
        A, B = np.empty((N, 4, 4)), np.empty((N, 4, 2))
        self.lib.orc_model_derivatives(_p(x), _p(u), dt, wb, N, rp, _p(A), _p(B))
        return A, B

    def front_rear(self, state, wb, rp):
        f, r = np.empty(2), np.empty(2)
        self.lib.orc_front_rear_centers(_p(_f64(state)), wb, rp, _p(f), _p(r))
        return f, r

    def front_rear_derivatives(self, yaw, wb, rp):
        f, r = np.empty((4, 2)), np.empty((4, 2))
        self.lib.orc_front_rear_center_derivatives(yaw, wb, rp, _p(f), _p(r))
        return f, r

    def ellipsoid_scales(self, obs_attr, radius):
        ab = np.empty(2)
        self.lib.orc_ellipsoid_scales(_p(_f64(obs_attr)), radius, _p(ab))
        return ab

    def safety_margin(self, pnt, obs_state, ab):
        return self.lib.orc_ellipsoid_safety_margin(_p(_f64(pnt)), _p(_f64(obs_state)), _p(_f64(ab)))

    def safety_margin_derivatives(self, pnt, obs_state, ab):
        out = np.empty(2)
        self.lib.orc_ellipsoid_safety_margin_derivatives(_p(_f64(pnt)), _p(_f64(obs_state)), _p(_f64(ab)), _p(out))
        return out

    def exp_barrier(self, c, q1, q2):
        return self.lib.orc_exp_barrier(c, q1, q2)

    def exp_barrier_dH(self, c, c_dot, q1, q2):
        c_dot = _f64(c_dot)
        n = c_dot.shape[0]
        bd, bdd = np.empty(n), np.empty((n, n))
        self.lib.orc_exp_barrier_derivative_and_Hessian(c, _p(c_dot), n, q1, q2, _p(bd), _p(bdd))
        return bd, bdd

    def obstacle_constr(self, params, ego, obs):
        p = make_params(params)
        out = np.empty(2)
        self.lib.orc_obstacle_constr(C.byref(p), _p(_f64(ego)), _p(_f64(obs)), _p(out))
        return out

    def obstacle_constr_derivatives(self, params, ego, obs):
        p = make_params(params)
        f, r = np.empty(4), np.empty(4)
        self.lib.orc_obstacle_constr_derivatives(C.byref(p), _p(_f64(ego)), _p(_f64(obs)), _p(f), _p(r))
        return f, r

    def ref_points(self, x, scene, tick=None):
        x = _f64(x)
        rows = x.shape[0]
        ref, idx = np.empty((rows, 3)), np.empty(rows, dtype=np.int32)
        s = scene.struct(tick)
        self.lib.orc_ref_exact_points(_p(x), rows, C.byref(s), _p(ref), _p(idx))
        return ref, idx

    def const_velo_prediction(self, params, x0):
        p = make_params(params)
        x = np.empty((p.N + 1, 4))
        self.lib.orc_const_velo_prediction(C.byref(p), _p(_f64(x0)), _p(x))
        return x

    def forward_pass(self, params, u, x, d, K, alpha):
        p = make_params(params)
        nu, nx = np.empty((p.N, 2)), np.empty((p.N + 1, 4))
        self.lib.orc_forward_pass(C.byref(p), _p(_f64(u)), _p(_f64(x)), _p(_f64(d)), _p(_f64(K)), alpha, _p(nu), _p(nx))
        return nu, nx

    MATH_CODES = {"exp": 0, "sin": 1, "cos": 2, "tan": 3, "atan": 4, "hypot": 5}

    def math(self, name, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        y = None if y is None else np.ascontiguousarray(y, dtype=np.float64).ravel()
        out = np.empty_like(x)
        self.lib.orc_m_vec(self.MATH_CODES[name], _p(x), _p(y), x.shape[0], _p(out))
        return out

    def alm_item(self, c, rho, mu):
        return self.lib.orc_alm_item(float(c), float(rho), float(mu))

    def lagrangian_dH(self, c, c_dot, rho, mu):
        c_dot = _f64(c_dot)
        n = c_dot.shape[0]
        bd, bdd = np.empty(n), np.empty((n, n))
        self.lib.orc_lagrangian_derivative_and_Hessian(float(c), _p(c_dot), n, float(rho), float(mu), _p(bd), _p(bdd))
        return bd, bdd

    def record_math(self, cap):
        """liboracle_rec.so: start recording (function, x, y) of every elementary-function call; returns the buffer"""
        buf = np.zeros((cap, 3))
        self._rec_buf = buf
        self.lib.orc_record_math(_p(buf), cap)
        return buf

    def record_count(self):
        return int(self.lib.orc_record_count())


class OracleSolver:
    """Stateful counterpart of the reference's CILQRSolver instance."""

    def __init__(self, oracle, params):
        self.o = oracle
        self.params = make_params(params)
        self.N = self.params.N
        self.h = oracle.lib.orc_create(C.byref(self.params))

    def close(self):
        if self.h:
            self.o.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.o.lib.orc_reset(self.h)

    def solve(self, x0, scene, tick=None, trace_cap=256, margins=False):
        """margins=True also returns, per iteration, how close its discrete decisions came to flipping
        (orc_margin_rec: line-search verdicts, Cholesky pivots, lane-scan comparisons)."""
        N = self.N
        u, x = np.empty((N, 2)), np.empty((N + 1, 4))
        res = np.zeros(1, dtype=RESULT_DTYPE)
        trace = np.zeros(trace_cap, dtype=TRACE_DTYPE)
        mg = np.zeros(trace_cap, dtype=MARGIN_DTYPE) if margins else None
        s = scene.struct(tick)
        self.o.lib.orc_set_margin_buffer(self.h, _p(mg), trace_cap if margins else 0)
        rc = self.o.lib.orc_solve(self.h, _p(_f64(x0)), C.byref(s), _p(u), _p(x), _p(res), _p(trace), trace_cap)
        self.o.lib.orc_set_margin_buffer(self.h, None, 0)
        if rc != 0:
            raise RuntimeError(f"orc_solve rc={rc}")
        out = {"u": u, "x": x, "res": res[0], "trace": trace[:res[0]["trace_len"]]}
        if margins:
            out["margins"] = mg[:res[0]["trace_len"]]
        return out

    def set_alm_state(self, mu, rho):
        mu = _f64(mu)
        self.o.lib.orc_set_alm_state(self.h, _p(mu), float(rho), int(mu.shape[1]))

    def get_alm_next(self, cols):
        out = np.empty((self.N, cols))
        self.o.lib.orc_get_alm_next(self.h, _p(out))
        return out

    def total_cost(self, u, x, scene, tick=None):
        s = scene.struct(tick)
        return self.o.lib.orc_total_cost(self.h, _p(_f64(u)), _p(_f64(x)), C.byref(s))

    def cost_derivatives(self, u, x, scene, tick=None):
        N = self.N
        out = {"l_x": np.empty((N + 1, 4)), "l_u": np.empty((N, 2)), "l_xx": np.empty((N + 1, 4, 4)),
               "l_uu": np.empty((N, 2, 2))}
        s = scene.struct(tick)
        self.o.lib.orc_cost_derivatives(self.h, _p(_f64(u)), _p(_f64(x)), C.byref(s), _p(out["l_x"]), _p(out["l_u"]),
                                        _p(out["l_xx"]), _p(out["l_uu"]))
        return out

    def backward_pass(self, u, x, lamb, scene, tick=None):
        N = self.N
        d, K, dV = np.empty((N, 2)), np.empty((N, 2, 4)), np.empty(2)
        s = scene.struct(tick)
        st = self.o.lib.orc_backward_pass(self.h, _p(_f64(u)), _p(_f64(x)), float(lamb), C.byref(s), _p(d), _p(K), _p(dV))
        return d, K, dV, st
