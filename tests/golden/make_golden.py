#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own Python modules (build container only).

Imports /root/reference/scripts/utils/{kinematic,constraint,cubic_spline}.py — the reference's
leaf functions, formula-identical to the CoG branches of src/utils.cpp:262-439 and to
src/cilqr_solver.cpp:316-324,692-699 — evaluates them on seeded random inputs and
stores inputs + outputs in tests/golden/leaf_vectors.npz.  Also runs the reference's Python
*variant* solver (scripts/2-cilqr-motionplanning.py main scenario) headless and stores its end state
(tests/golden/variant_vectors.npz), the rear-axle step of scripts/1-lqr-pathtracking.py
(tests/golden/rear_axle_vectors.npz) and samples the Python cubic spline on the four scenario
way-point sets (tests/golden/spline_vectors.npz).  Only data is written; no reference source is
copied.  Nothing here runs on the GPU box.
"""
import importlib.util
import io
import contextlib
import json
import os
import pathlib
import sys

import numpy as np

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, str(REF / "scripts" / "utils"))

import kinematic  # noqa: E402  (reference module)
import constraint  # noqa: E402  (reference module)
import cubic_spline  # noqa: E402  (reference module)


def leaf_vectors(rng):
    out = {}
    dt, wb = 0.1, 2.8
    # kinematic_propagate (kinematic.py:3-14)
    n = 200
    x = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-10, 10, n), rng.uniform(0, 15, n), rng.uniform(-3.2, 3.2, n)])
    u = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-0.6, 0.6, n)])
    out["prop_x"], out["prop_u"] = x, u
    out["prop_out"] = np.stack([kinematic.kinematic_propagate(x[i], u[i], dt, wb) for i in range(n)])
    out["dt"], out["wb"] = np.array(dt), np.array(wb)
    # get_kinematic_model_derivatives (kinematic.py:17-51): layout (dim x time) in, (4,4,N)/(4,2,N) out
    N = 30
    xs = np.column_stack([rng.uniform(-50, 50, N + 1), rng.uniform(-10, 10, N + 1), rng.uniform(0, 15, N + 1), rng.uniform(-3.2, 3.2, N + 1)])
    us = np.column_stack([rng.uniform(-4, 4, N), rng.uniform(-0.6, 0.6, N)])
    dfdx, dfdu = kinematic.get_kinematic_model_derivatives(xs.T.copy(), us.T.copy(), dt, wb, N)
    out["md_x"], out["md_u"] = xs, us
    out["md_A"] = np.transpose(dfdx, (2, 0, 1)).copy()
    out["md_B"] = np.transpose(dfdu, (2, 0, 1)).copy()
    # const_velo_prediction (kinematic.py:54-66)
    x0 = np.array([1.0, -2.0, 6.5, 0.3])
    out["cvp_x0"] = x0
    out["cvp_out"] = kinematic.const_velo_prediction(x0, 40, dt, wb).T.copy()
    # front / rear centres and derivatives (kinematic.py:78-104)
    n = 50
    st = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-10, 10, n), rng.uniform(0, 15, n), rng.uniform(-3.2, 3.2, n)])
    fr = [kinematic.get_vehicle_front_and_rear_centers(st[i, :2], st[i, 3], wb) for i in range(n)]
    out["fr_state"] = st
    out["fr_front"] = np.stack([f for f, _ in fr])
    out["fr_rear"] = np.stack([r for _, r in fr])
    frd = [kinematic.get_vehicle_front_and_rear_center_derivatives(st[i, 3], wb) for i in range(n)]
    out["frd_front"] = np.stack([f for f, _ in frd])  # (2,4) each: [point dim][state dim]
    out["frd_rear"] = np.stack([r for _, r in frd])
    # ellipsoid safety margin + derivatives (kinematic.py:114-147) with explicit (a, b)
    n = 200
    pnt = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n)])
    cen = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n)])
    th = rng.uniform(-3.2, 6.4, n)
    a = rng.uniform(2, 10, n)
    b = rng.uniform(1, 4, n)
    out["em_pnt"], out["em_cen"], out["em_theta"], out["em_a"], out["em_b"] = pnt, cen, th, a, b
    out["em_margin"] = np.array([kinematic.ellipsoid_safety_margin(pnt[i], cen[i], th[i], a[i], b[i]) for i in range(n)])
    out["em_grad"] = np.stack([kinematic.ellipsoid_safety_margin_derivatives(pnt[i], cen[i], th[i], a[i], b[i]) for i in range(n)])
    # exp barrier (constraint.py:8-20) and bound constraints (:23-29)
    n = 50
    cval = rng.uniform(-3, 1.5, n)
    q1 = rng.uniform(1, 8, n)
    q2 = rng.uniform(1, 8, n)
    cdot = rng.uniform(-2, 2, (n, 4))
    out["eb_c"], out["eb_q1"], out["eb_q2"], out["eb_cdot"] = cval, q1, q2, cdot
    out["eb_b"] = np.array([constraint.exp_barrier(cval[i], q1[i], q2[i]) for i in range(n)])
    bd = [constraint.exp_barrier_derivative_and_Hessian(cval[i], cdot[i], q1[i], q2[i]) for i in range(n)]
    out["eb_bdot"] = np.stack([v for v, _ in bd])
    out["eb_bddot"] = np.stack([m for _, m in bd])
    var, bnd = rng.uniform(-5, 5, n), rng.uniform(-5, 5, n)
    out["bc_var"], out["bc_bound"] = var, bnd
    out["bc_upper"] = np.array([constraint.get_bound_constr(var[i], bnd[i], "upper") for i in range(n)])
    out["bc_lower"] = np.array([constraint.get_bound_constr(var[i], bnd[i], "lower") for i in range(n)])
    # obstacle avoidance constraint + derivatives (constraint.py:43-64), Python scales (1 x d_safe)
    n = 100
    ego = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-5, 5, n), rng.uniform(0, 15, n), rng.uniform(-3.2, 3.2, n)])
    obs = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-5, 5, n), rng.uniform(0, 15, n), rng.uniform(-3.2, 3.2, n)])
    width, attr = 2.0, np.array([2.0, 4.5, 0.8])  # (obs width, obs length, d_safe)
    out["oc_ego"], out["oc_obs"], out["oc_width"], out["oc_attr"] = ego, obs, np.array(width), attr
    oc = [constraint.get_obstacle_avoidance_constr(ego[i], obs[i], wb, width, attr) for i in range(n)]
    out["oc_front"] = np.array([f for f, _ in oc])
    out["oc_rear"] = np.array([r for _, r in oc])
    ocd = [constraint.get_obstacle_avoidance_constr_derivatives(ego[i], obs[i], wb, width, attr) for i in range(n)]
    out["ocd_front"] = np.stack([f for f, _ in ocd])
    out["ocd_rear"] = np.stack([r for _, r in ocd])
    out["oc_ab"] = np.array(kinematic.get_ellipsoid_obstacle_scales(0.5 * width, attr[0], attr[1], attr[2]))
    return out


def spline_vectors():
    out = {}
    import yaml
    for path in sorted((REF / "config").glob("scenario_*.yaml")):
        name = path.stem.replace("scenario_", "")
        doc = yaml.safe_load(path.read_text())
        rx = [float(v) for v in doc["laneline"]["reference"]["x"]]
        ry = [float(v) for v in doc["laneline"]["reference"]["y"]]
        sp = cubic_spline.CubicSpline2D(rx, ry)
        s = np.linspace(0.0, sp.s[-1] * 0.999999, 400)
        pos = np.array([sp.calc_position(v) for v in s])
        yaw = np.array([sp.calc_yaw(v) for v in s])
        out[name + "_wx"], out[name + "_wy"] = np.array(rx), np.array(ry)
        out[name + "_s"], out[name + "_pos"], out[name + "_yaw"] = s, pos, yaw
        out[name + "_slen"] = np.array(sp.s[-1])
    return out


def variant_vectors():
    """The reference's Python *variant* CILQR (scripts/2-cilqr-motionplanning.py) on its own main()
    scenario: run its solve() to the end, then evaluate ITS get_total_cost / derivatives /
    backward_pass(lamb=0) / forward_pass on the initial, an intermediate and the final trajectory.
    With w_yaw = 0, road borders at +-1e9, CoG model and length' = length - 10*d_safe the C++ path's
    formulas reduce to the variant's, so these vectors pin the oracle's composite
    functions against an upstream implementation."""
    spec = importlib.util.spec_from_file_location("cilqr_variant", str(REF / "scripts" / "2-cilqr-motionplanning.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.path.insert(0, str(REF / "scripts"))
    spec.loader.exec_module(mod)
    HL, DT, WB = mod.HORIZON_LENGTH, mod.DT, mod.WB
    ego_state = [0., 0., 5.0, 0.]
    ref_waypoints = np.vstack((np.linspace(0, 50, 1000), np.linspace(0, 0, 1000)))
    ref_velo = np.array(6.0)
    attr = np.array([mod.WIDTH, mod.LENGTH, mod.SAFETY_BUFFER])
    pred1 = kinematic.const_velo_prediction([6.5, -0.2, 3.0, 0.], HL, DT, WB)
    pred2 = kinematic.const_velo_prediction([20, 4, 2.0, 0.], HL, DT, WB)
    attrs = np.stack((attr, attr), axis=0)
    preds = np.stack((pred1, pred2), axis=0)  # [2][4][N+1]
    planner = mod.CILQR()
    buf = io.StringIO()
    # capture intermediate trajectories by wrapping iter_step
    snaps = []
    orig = planner.iter_step

    def wrapped(u, x, J, lamb, *a):
        snaps.append((u.copy(), x.copy(), float(J), float(lamb)))
        return orig(u, x, J, lamb, *a)

    planner.iter_step = wrapped
    with contextlib.redirect_stdout(buf):
        opti_u, opti_x = planner.solve(ego_state, ref_waypoints, ref_velo, attrs, preds)
    out = {"stdout": np.array(buf.getvalue()), "N": np.array(HL), "dt": np.array(DT), "wb": np.array(WB),
           "ego": np.array(ego_state), "lane": ref_waypoints.T.copy(), "ref_velo": np.array(float(ref_velo)),
           "attr": attr, "obs": np.transpose(preds, (0, 2, 1)).copy(),  # [2][N+1][4] (x, y, v, yaw)
           "state_weight": planner.state_weight, "ctrl_weight": planner.ctrl_weight,
           "exp_q1": np.array(planner.exp_q1), "exp_q2": np.array(planner.exp_q2),
           "bounds": np.array([planner.acc_max, planner.acc_min, planner.stl_lim, planner.velo_max, planner.velo_min]),
           "width": np.array(planner.width), "length": np.array(planner.length),
           "final_u": opti_u.T.copy(), "final_x": opti_x.T.copy(), "n_snaps": np.array(len(snaps))}
    picks = {"init": snaps[0][:2], "mid": snaps[len(snaps) // 2][:2], "final": (opti_u, opti_x)}
    for tag, (u, x) in picks.items():
        J = planner.get_total_cost(u, x, ref_waypoints, ref_velo, attrs, preds)
        l_u, l_uu, l_x, l_xx, _ = planner.get_total_cost_derivatives_and_Hessians(u, x, ref_waypoints, ref_velo, attrs, preds)
        d, K, dV = planner.backward_pass(u, x, 0.0, ref_waypoints, ref_velo, attrs, preds)
        nu, nx = planner.forward_pass(u, x, d, K, 0.5)
        out[tag + "_u"], out[tag + "_x"] = u.T.copy(), x.T.copy()
        out[tag + "_J"] = np.array(J)
        out[tag + "_l_u"] = l_u.T.copy()
        out[tag + "_l_uu"] = np.transpose(l_uu, (2, 0, 1)).copy()
        out[tag + "_l_x"] = l_x.T.copy()
        out[tag + "_l_xx"] = np.transpose(l_xx, (2, 0, 1)).copy()
        out[tag + "_d"] = d.T.copy()
        out[tag + "_K"] = np.transpose(K, (2, 0, 1)).copy()
        out[tag + "_dV"] = np.array(dV)
        out[tag + "_fw_u"], out[tag + "_fw_x"] = nu.T.copy(), nx.T.copy()
    return out


def rear_axle_vectors():
    """The rear-axle step of the reference's path-tracking demo (scripts/1-lqr-pathtracking.py:134-140, `update`):
    the same model as the RearCenter branch of utils::kinematic_propagate (src/utils.cpp:266-272) — x, y, v identical
    expressions, yaw as `v / WB * tan(delta) * dt` there against `v * tan(delta) * dt / wb` in the C++ (<= 2 ulp apart).
    State order there is (x, y, yaw, v); stored here in the solver's order (x, y, v, yaw)."""
    spec = importlib.util.spec_from_file_location("lqr_pathtracking", str(REF / "scripts" / "1-lqr-pathtracking.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.path.insert(0, str(REF / "scripts"))
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(20250830)  # its own stream: the other fixtures regenerate unchanged
    n = 200
    x = np.column_stack([rng.uniform(-50, 50, n), rng.uniform(-10, 10, n), rng.uniform(0, 15, n), rng.uniform(-3.2, 3.2, n)])
    u = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-0.6, 0.6, n)])
    out = np.empty_like(x)
    for i in range(n):
        st = mod.VehicleState(x=float(x[i, 0]), y=float(x[i, 1]), yaw=float(x[i, 3]), v=float(x[i, 2]))
        st = mod.update(st, float(u[i, 0]), float(u[i, 1]))
        out[i] = (st.x, st.y, st.v, st.yaw)
    # a 60-step open-loop rollout through the same function (chained, as const_velo_prediction chains the C++ step)
    st = mod.VehicleState(x=1.0, y=-2.0, yaw=0.3, v=6.5)
    ctl = np.column_stack([rng.uniform(-2, 2, 60), rng.uniform(-0.3, 0.3, 60)])
    chain = []
    for a, d in ctl:
        st = mod.update(st, float(a), float(d))
        chain.append((st.x, st.y, st.v, st.yaw))
    return {"x": x, "u": u, "out": out, "dt": np.array(mod.dt), "wb": np.array(mod.WB),
            "chain_x0": np.array([1.0, -2.0, 6.5, 0.3]), "chain_u": ctl, "chain_out": np.array(chain)}


def main():
    np.savez_compressed(OUT / "rear_axle_vectors.npz", **rear_axle_vectors())
    print("rear_axle_vectors.npz written")
    rng = np.random.default_rng(20250829)
    np.savez_compressed(OUT / "leaf_vectors.npz", **leaf_vectors(rng))
    print("leaf_vectors.npz written")
    np.savez_compressed(OUT / "spline_vectors.npz", **spline_vectors())
    print("spline_vectors.npz written")
    vv = variant_vectors()
    np.savez_compressed(OUT / "variant_vectors.npz", **vv)
    print("variant_vectors.npz written;", str(vv["stdout"]).strip().splitlines()[-1], "J_final =", float(vv["final_J"]))


if __name__ == "__main__":
    main()
