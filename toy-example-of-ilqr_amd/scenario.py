"""Construction of the solve path's inputs from a scenario config (host side, no GPU).

Mirrors what /root/reference/src/motion_planning.cpp:52-174 does before the planning loop:
ReferenceLine objects for borders and centre lines (utils.cpp:21-35), road_borders
(motion_planning.cpp:101-103) and the obstacle routes (motion_planning.cpp:121-173, noise-free).
The numerical work is done by the C++ host code in csrc/scenario.cpp through the C-ABI.
"""
import ctypes as C
import dataclasses

import numpy as np

from . import _lib


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


@dataclasses.dataclass
class ReferenceLine:
    """Fields of the reference's ReferenceLine that the solver reads (include/utils.hpp:32-51)."""
    x: np.ndarray
    y: np.ndarray
    yaw: np.ndarray
    longitude: np.ndarray
    delta_d: float
    delta_s: float = 0.1

    def size(self):
        return int(self.x.shape[0])

    @classmethod
    def build(cls, ref_x, ref_y, width=0.0, accuracy=0.1):
        lib = _lib.load()
        wx = np.ascontiguousarray(ref_x, dtype=np.float64)
        wy = np.ascontiguousarray(ref_y, dtype=np.float64)
        cnt = C.c_int32(0)
        _lib.check(lib.cilqr_reference_line_build(_dp(wx), _dp(wy), len(wx), float(width), float(accuracy),
                                                  None, None, None, None, 0, C.byref(cnt)),
                   "cilqr_reference_line_build")
        n = cnt.value
        x, y, yaw, s = (np.empty(n) for _ in range(4))
        _lib.check(lib.cilqr_reference_line_build(_dp(wx), _dp(wy), len(wx), float(width), float(accuracy),
                                                  _dp(x), _dp(y), _dp(yaw), _dp(s), n, C.byref(cnt)),
                   "cilqr_reference_line_build")
        return cls(x, y, yaw, s, float(width), float(accuracy))


@dataclasses.dataclass
class RoutingLine:
    """x, y, yaw per tick (include/utils.hpp:53-68)."""
    x: np.ndarray
    y: np.ndarray
    yaw: np.ndarray

    def as_array(self):
        return np.stack([self.x, self.y, self.yaw], axis=1)


@dataclasses.dataclass
class Scenario:
    """Everything solve() needs besides x0, for one scenario file."""
    name: str
    center_lines: list          # list[ReferenceLine]
    borders: list               # list[ReferenceLine]
    road_borders: np.ndarray    # (max, min) border offsets
    routes: np.ndarray          # [V][T][3]; row 0 is the ego's own route, rows 1.. are obstacles
    target_velocity: float
    initial_conditions: np.ndarray  # [V][4]
    delta_t: float
    max_simulation_time: float
    line_num: np.ndarray = None
    start_s: np.ndarray = None

    @property
    def lane(self):
        """center_lines[0]: the ref_waypoints argument of solve() (motion_planning.cpp:195)."""
        return self.center_lines[0]

    @property
    def obstacles(self):
        """obs_prediction = routing_lines[1:] (motion_planning.cpp:174), [M][T][3]."""
        return np.ascontiguousarray(self.routes[1:])

    @property
    def ego_state(self):
        return self.initial_conditions[0].copy()


def build_scenario(cfg, name="scenario", accuracy=0.1):
    lib = _lib.load()
    g = cfg.get_config
    ref_x = np.asarray(g("laneline/reference/x"), dtype=np.float64)
    ref_y = np.asarray(g("laneline/reference/y"), dtype=np.float64)
    border_w = list(g("laneline/border"))
    center_w = np.asarray(g("laneline/center_line"), dtype=np.float64)
    init = np.ascontiguousarray(g("initial_condition"), dtype=np.float64)
    dt = g("delta_t", float)
    tmax = g("max_simulation_time", float)
    borders = [ReferenceLine.build(ref_x, ref_y, w, accuracy) for w in border_w]
    centers = [ReferenceLine.build(ref_x, ref_y, w, accuracy) for w in center_w]
    sorted_w = sorted(border_w, reverse=True)
    road_borders = np.array([sorted_w[0], sorted_w[-1]], dtype=np.float64)
    V = init.shape[0]
    T = C.c_int32(0)
    _lib.check(lib.cilqr_build_routes(_dp(ref_x), _dp(ref_y), len(ref_x), _dp(center_w), len(center_w),
                                      float(accuracy), _dp(init), V, tmax, dt, None, 0, C.byref(T), None, None),
               "cilqr_build_routes")
    routes = np.zeros((V, T.value, 3))
    line_num = np.zeros(V, dtype=np.int32)
    start_s = np.zeros(V)
    _lib.check(lib.cilqr_build_routes(_dp(ref_x), _dp(ref_y), len(ref_x), _dp(center_w), len(center_w),
                                      float(accuracy), _dp(init), V, tmax, dt, _dp(routes), T.value,
                                      C.byref(T), _dp(line_num), _dp(start_s)),
               "cilqr_build_routes")
    return Scenario(name=name, center_lines=centers, borders=borders, road_borders=road_borders,
                    routes=routes, target_velocity=g("vehicle/target_velocity", float),
                    initial_conditions=init, delta_t=dt, max_simulation_time=tmax,
                    line_num=line_num, start_s=start_s)
