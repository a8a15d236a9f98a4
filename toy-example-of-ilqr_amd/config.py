"""Scenario/tuning configuration with the reference's GlobalConfig key set.

Mirror of /root/reference/src/global_config.cpp:17-131: a flat ``"section/key" -> value`` map with
the same keys and defaults; ``get_config`` returns the type's zero value (and reports on stderr)
for a missing key, like the reference's typed getter.  Sources: the reference's own YAML layout
(parsed with PyYAML) or the flattened JSON shipped under ``scenarios/``.
"""
import json
import pathlib
import sys

from ._lib import CilqrParams

SCENARIO_DIR = pathlib.Path(__file__).resolve().parent / "scenarios"
BUILTIN_SCENARIOS = ("two_straight", "two_borrow", "three_straight", "three_bend")

_DEFAULTS = {  # global_config.cpp: .as<T>(default) call sites
    "lqr/alm_rho_init": 1.0, "lqr/alm_gamma": 0.0, "lqr/max_rho": 100.0, "lqr/max_mu": 1000.0,
    "vehicle/reference_point": "gravity_center",
    "visualization/show_reference_line": False, "visualization/show_obstacle_boundary": False,
}


def _flatten_yaml(doc):
    flat = {}
    for key in ("max_simulation_time", "delta_t"):
        if key in doc:
            flat[key] = doc[key]
    for sec in ("lqr", "iteration", "vehicle", "visualization"):
        for k, v in (doc.get(sec) or {}).items():
            flat[f"{sec}/{k}"] = v
    lane = doc.get("laneline") or {}
    if "reference" in lane:
        flat["laneline/reference/x"] = list(lane["reference"]["x"])
        flat["laneline/reference/y"] = list(lane["reference"]["y"])
    for k in ("border", "center_line"):
        if k in lane:
            flat[f"laneline/{k}"] = list(lane[k])
    if "initial_condition" in doc:
        flat["initial_condition"] = [list(r) for r in doc["initial_condition"]]
    return flat


class GlobalConfig:
    """Key/value view of one scenario file (not a singleton, unlike the reference)."""

    def __init__(self, values):
        self._map = dict(_DEFAULTS)
        self._map.update(values)

    @classmethod
    def get_instance(cls, path):
        """GlobalConfig::get_instance(path): accepts a YAML file in the reference's layout, a
        flattened JSON file, or the name of a built-in scenario ("two_straight", ...)."""
        p = pathlib.Path(str(path))
        if not p.exists() and str(path) in BUILTIN_SCENARIOS:
            p = SCENARIO_DIR / f"{path}.json"
        if p.suffix == ".json":
            return cls(json.loads(p.read_text()))
        import yaml
        return cls(_flatten_yaml(yaml.safe_load(p.read_text())))

    def has_key(self, key):
        return key in self._map

    def get_config(self, key, typ=None):
        if key not in self._map:
            print(f"Key not found: {key}", file=sys.stderr)
            return typ() if typ else None
        v = self._map[key]
        return typ(v) if typ in (int, float, bool, str) else v

    def with_overrides(self, **kv):
        m = dict(self._map)
        m.update({k.replace("__", "/"): v for k, v in kv.items()})
        return GlobalConfig(m)

    def as_dict(self):
        return dict(self._map)


def params_from_config(cfg, **overrides):
    """CILQRSolver::CILQRSolver(config) (src/cilqr_solver.cpp:17-83): config -> struct cilqr_params.
    ``overrides`` use field names of the struct (N=50, obstacle_exp_q1=..., ...)."""
    g = cfg.get_config
    p = CilqrParams()
    p.N = g("lqr/N", int)
    p.max_iter = g("iteration/max_iter", int)
    st = g("lqr/slove_type", str)
    p.solve_type = 1 if st == "alm" else 0  # anything else defaults to barrier (cs:36-41)
    p.reference_point = 0 if g("vehicle/reference_point", str) == "rear_center" else 1
    p.use_last_solution = 1 if g("lqr/use_last_solution", bool) else 0
    p.dt = g("delta_t", float)
    p.w_pos = g("lqr/w_pos", float)
    p.w_vel = g("lqr/w_vel", float)
    p.w_yaw = g("lqr/w_yaw", float)
    p.w_acc = g("lqr/w_acc", float)
    p.w_stl = g("lqr/w_stl", float)
    p.obstacle_exp_q1 = g("lqr/obstacle_exp_q1", float)
    p.obstacle_exp_q2 = g("lqr/obstacle_exp_q2", float)
    p.state_exp_q1 = g("lqr/state_exp_q1", float)
    p.state_exp_q2 = g("lqr/state_exp_q2", float)
    p.alm_rho_init = g("lqr/alm_rho_init", float)
    p.alm_gamma = g("lqr/alm_gamma", float)
    p.max_rho = g("lqr/max_rho", float)
    p.max_mu = g("lqr/max_mu", float)
    p.init_lamb = g("iteration/init_lamb", float)
    p.lamb_decay = g("iteration/lamb_decay", float)
    p.lamb_amplify = g("iteration/lamb_amplify", float)
    p.max_lamb = g("iteration/max_lamb", float)
    p.convergence_threshold = g("iteration/convergence_threshold", float)
    p.accept_step_threshold = g("iteration/accept_step_threshold", float)
    p.wheelbase = g("vehicle/wheelbase", float)
    p.width = g("vehicle/width", float)
    p.length = g("vehicle/length", float)
    p.velo_max = g("vehicle/velo_max", float)
    p.velo_min = g("vehicle/velo_min", float)
    p.yaw_lim = g("vehicle/yaw_lim", float)
    p.acc_max = g("vehicle/acc_max", float)
    p.acc_min = g("vehicle/acc_min", float)
    p.stl_lim = g("vehicle/stl_lim", float)
    p.d_safe = g("vehicle/d_safe", float)
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def params_to_dict(p):
    return {name: getattr(p, name) for name, _ in CilqrParams._fields_}


def copy_params(p, **overrides):
    q = CilqrParams()
    for name, _ in CilqrParams._fields_:
        setattr(q, name, getattr(p, name))
    for k, v in overrides.items():
        if not hasattr(q, k):
            raise KeyError(k)
        setattr(q, k, v)
    return q
