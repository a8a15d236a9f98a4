"""Derived pins of the C++-specific semantics that nothing upstream runs (VERDICT r04 "What's missing" 2, task 6).

The reference's Python scripts pin the centre-of-gravity leaf functions and, since round 4, the RearCenter STEP
(tests/golden/rear_axle_vectors.npz from scripts/1-lqr-pathtracking.py).  The pieces below have no runnable upstream
twin; they are pinned here by what CAN be derived from pinned pieces or from the reference's own definitions:

 (a) RearCenter Jacobians (src/utils.cpp:313-322) = finite differences of the upstream-pinned RearCenter step — that
     branch has no beta-tilde quirk, so analytic and numeric must agree to ~1e-7;
 (b) the road-border terms of the cost expansion (src/cilqr_solver.cpp:527-533) = finite differences of the road-border
     terms of get_total_cost (cs:235-252), isolated by differencing against far-away borders;
 (c) lagrangian_derivative_and_Hessian (cs:701-713) against augmented_lagrangian_item (include/cilqr_solver.hpp:81-83):
     the gradient is the derivative of the item; the "Hessian" is b_dot c_dot^T — the reference's quirk, (c + mu/rho)
     times the Gauss-Newton Hessian rho c_dot c_dot^T — asserted as such; and the whole ALM cost against its expansion;
 (d) csrc/detmath.h against glibc: the largest difference in units in the last place per function, on >= 1e6 points
     per function INCLUDING the arguments real solves hand to them (recorded by liboracle_rec.so), with the bound stated.
"""
import numpy as np
import pytest

from conftest import GOLDEN, oracle_scene


def rollout(orc, p, x0, u):
    x = np.zeros((p.N + 1, 4))
    x[0] = x0
    for i in range(p.N):
        x[i + 1] = orc.propagate(x[i], u[i], p.dt, p.wheelbase, p.reference_point)
    return x


# ---------------------------------------------------------------------------------------------------------------------
# (a) RearCenter Jacobians
def test_rear_center_jacobians_are_the_derivative_of_the_upstream_pinned_step(orc_libm):
    g = np.load(GOLDEN / "rear_axle_vectors.npz")
    dt, wb = float(g["dt"]), float(g["wb"])
    X, U, OUT = g["x"], g["u"], g["out"]
    # the step itself is the pinned one (x, y, v to the bit; yaw within the re-association noted in the golden test)
    for i in range(0, 200, 17):
        got = orc_libm.propagate(X[i], U[i], dt, wb, 0)
        assert np.array_equal(got[:3], OUT[i][:3]) and abs(got[3] - OUT[i][3]) <= 4e-16 * max(1.0, abs(OUT[i][3]))
    worst = 0.0
    for i in range(200):
        x, u = X[i], U[i]
        if abs(u[1]) > 1.2:  # (tan's curvature makes a 1e-6 central difference worse than 1e-7 near +-pi/2)
            continue
        A, B = orc_libm.model_derivatives(np.stack([x, x]), u[None], dt, wb, 1, 0)
        A, B = A[0], B[0]
        for j in range(4):
            h = 1e-6 * max(1.0, abs(x[j]))
            xp, xm = x.copy(), x.copy()
            xp[j] += h
            xm[j] -= h
            fd = (orc_libm.propagate(xp, u, dt, wb, 0) - orc_libm.propagate(xm, u, dt, wb, 0)) / (xp[j] - xm[j])
            worst = max(worst, np.abs(A[:, j] - fd).max() / max(1.0, np.abs(fd).max()))
        for j in range(2):
            h = 1e-6
            up, um = u.copy(), u.copy()
            up[j] += h
            um[j] -= h
            fd = (orc_libm.propagate(x, up, dt, wb, 0) - orc_libm.propagate(x, um, dt, wb, 0)) / (up[j] - um[j])
            worst = max(worst, np.abs(B[:, j] - fd).max() / max(1.0, np.abs(fd).max()))
        # structure of ut:313-322: identity + five entries, B has two non-zeros
        assert A[0, 0] == 1 and A[1, 1] == 1 and A[2, 2] == 1 and A[3, 3] == 1 and A[2, 3] == 0 and A[3, 0] == 0
        assert B[2, 0] == dt and B[0, 0] == 0 and B[0, 1] == 0 and B[1, 1] == 0 and B[3, 0] == 0
    assert worst < 2e-7, worst


# ---------------------------------------------------------------------------------------------------------------------
# (b) road-border terms
def test_road_border_terms_of_the_expansion_are_the_gradient_of_the_cost_terms(pkg, orc_libm, scenarios):
    cfg, sc = scenarios["two_straight"]  # borders (5.4, -1.8), RearCenter
    p = pkg.params_from_config(cfg, N=20)
    near = oracle_scene(sc)
    from oracle import Scene
    far = Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, np.array([1e3, -1e3]), sc.target_velocity)
    s = orc_libm.solver(p)
    rng = np.random.default_rng(21)
    checked = 0
    for y0, yaw0 in ((3.6, 0.05), (-0.55, -0.04), (4.1, -0.02), (1.0, 0.0)):
        x0 = sc.ego_state + np.array([2.0, y0, 0.0, yaw0])  # rows off the line, some close to a border
        u = np.stack([rng.normal(0, 0.3, p.N), rng.normal(0, 0.01, p.N)], axis=1)
        x = rollout(orc_libm, p, x0, u)
        d_near = s.cost_derivatives(u, x, near)["l_x"]
        d_far = s.cost_derivatives(u, x, far)["l_x"]
        border_grad = d_near - d_far  # what cs:527-533 contribute (with borders 1e3 away the two exponentials vanish)
        assert np.abs(d_far - d_near)[0].max() == 0.0  # row 0 carries no barrier terms (cs:211 vs :217)
        for k in (1, 6, 13, p.N):
            for j in (0, 1):
                h = 1e-6
                xp, xm = x.copy(), x.copy()
                xp[k, j] += h
                xm[k, j] -= h
                fd = ((s.total_cost(u, xp, near) - s.total_cost(u, xp, far)) -
                      (s.total_cost(u, xm, near) - s.total_cost(u, xm, far))) / (2 * h)
                assert abs(fd - border_grad[k, j]) <= 2e-5 * max(1.0, abs(fd)), (y0, k, j, fd, border_grad[k, j])
                checked += abs(fd) > 1e-3
            # the gradient is radial (cs:527-529): along (dx, dy) / hypot, no v / yaw component
            assert border_grad[k, 2] == 0.0 and border_grad[k, 3] == 0.0
    assert checked >= 8  # (the terms were actually active on the rows that were differenced)


# ---------------------------------------------------------------------------------------------------------------------
# (c) ALM
def test_lagrangian_derivative_and_the_hessian_quirk(orc_libm):
    rng = np.random.default_rng(8)
    active = 0
    for _ in range(400):
        n = int(rng.choice([2, 4]))
        c0, rho, mu = rng.normal(0, 1.0), rng.uniform(0.5, 50.0), rng.uniform(0.0, 5.0)
        c_dot = rng.normal(0, 1.0, n)
        bd, bdd = orc_libm.lagrangian_dH(c0, c_dot, rho, mu)
        t = c0 + mu / rho
        if t > 0:
            active += 1
            # gradient of the item along z: c(z) = c0 + c_dot . z
            for i in range(n):
                h = 1e-6
                fd = (orc_libm.alm_item(c0 + c_dot[i] * h, rho, mu) - orc_libm.alm_item(c0 - c_dot[i] * h, rho, mu)) / (2 * h)
                if abs(t) > 1e-4:  # (the max() kink)
                    assert abs(fd - bd[i]) <= 1e-6 * max(1.0, abs(fd)), (c0, rho, mu, fd, bd[i])
            # the reference's "Hessian": b_dot c_dot^T, element by element in that association (cs:709-711)
            assert np.array_equal(bdd, np.outer(bd, c_dot))
            # ... which is (c + mu / rho) times the Gauss-Newton Hessian of the item, rho c_dot c_dot^T: the quirk
            gn = rho * np.outer(c_dot, c_dot)
            np.testing.assert_allclose(bdd, t * gn, rtol=1e-12, atol=1e-300)
        else:
            assert not bd.any() and not bdd.any()
            assert orc_libm.alm_item(c0, rho, mu) == 0.0
    assert 100 < active < 400


def test_alm_cost_expansion_is_the_gradient_of_the_alm_cost(pkg, orc_libm, scenarios):
    cfg, sc = scenarios["three_bend"]
    p = pkg.params_from_config(cfg, N=20, solve_type=1)
    scene = oracle_scene(sc)
    s = orc_libm.solver(p)
    rng = np.random.default_rng(5)
    cols = 8 + 2 * sc.obstacles.shape[0]
    mu = rng.uniform(0.0, 2.0, (p.N, cols))
    s.set_alm_state(mu, 3.0)
    x0 = sc.ego_state + np.array([1.0, 0.6, 3.0, 0.02])  # fast enough for the speed and obstacle constraints to bite
    u = np.stack([rng.normal(0.5, 1.5, p.N), rng.normal(0, 0.2, p.N)], axis=1)
    x = rollout(orc_libm, p, x0, u)
    d = s.cost_derivatives(u, x, scene)
    h = 1e-6
    big = 0
    for k in (1, 4, 9, p.N):
        for j in range(4):
            xp, xm = x.copy(), x.copy()
            xp[k, j] += h
            xm[k, j] -= h
            fd = (s.total_cost(u, xp, scene) - s.total_cost(u, xm, scene)) / (2 * h)
            assert abs(fd - d["l_x"][k, j]) <= 2e-5 * max(1.0, abs(fd)), (k, j, fd, d["l_x"][k, j])
            big += abs(fd) > 1.0
    for k in (0, 3, p.N - 1):
        for j in range(2):
            up, um = u.copy(), u.copy()
            up[k, j] += h
            um[k, j] -= h
            fd = (s.total_cost(up, x, scene) - s.total_cost(um, x, scene)) / (2 * h)
            assert abs(fd - d["l_u"][k, j]) <= 2e-5 * max(1.0, abs(fd)), (k, j, fd, d["l_u"][k, j])
    assert big >= 4


# ---------------------------------------------------------------------------------------------------------------------
# (d) detmath vs glibc, in units in the last place
def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a.view(np.int64) - b.view(np.int64)).astype(np.float64)
    d[(np.isnan(a) & np.isnan(b)) | (a == b)] = 0
    return d


# the bound this build states and holds, per function (units in the last place of glibc's result; glibc itself is within
# 1 ulp of the true value): what "within an ulp or two of libm" in DESIGN.md section 2 means, number by number
ULP_BOUND = {"exp": 1, "sin": 1, "cos": 1, "tan": 3, "atan": 1, "hypot": 1}
EXACT_FRAC = {"exp": 0.85, "sin": 0.85, "cos": 0.85, "tan": 0.5, "atan": 0.85, "hypot": 0.75}  # share of bit-equal results


@pytest.fixture(scope="module")
def solver_arguments(pkg, scenarios, built):
    """(function code, x, y) of every elementary-function call of 24 config-5 solves (three_bend, N = 50, the 16 barrier
    settings) and 8 two_straight solves (RearCenter, road borders) — liboracle_rec.so, single thread"""
    from oracle import Oracle
    rec = Oracle("rec")
    buf = rec.record_math(6_000_000)
    wl = pkg.workloads.config5(B_base=2, N=50)
    import libm_tolerance as lt
    rec.solve_batch(wl.params, lt.oracle_scenes(wl), wl.x0[:24], wl.scenario_id[:24], wl.param_id[:24], wl.tick[:24], n_threads=1)
    wl2 = pkg.workloads.config2(B=8, N=50)
    rec.solve_batch(wl2.params, lt.oracle_scenes(wl2), wl2.x0, n_threads=1)
    n = min(rec.record_count(), buf.shape[0])
    rec.lib.orc_record_math(None, 0)
    return buf[:n].copy()


@pytest.mark.parametrize("name,code,lo,hi", [("exp", 0, -60.0, 60.0), ("sin", 1, -40.0, 40.0), ("cos", 2, -40.0, 40.0),
                                             ("tan", 3, -1.45, 1.45), ("atan", 4, -50.0, 50.0), ("hypot", 5, -300.0, 300.0)])
def test_detmath_ulp_bound_on_a_million_points_and_on_the_solver_s_own_arguments(orc_det, orc_libm, solver_arguments, name, code, lo, hi):
    rng = np.random.default_rng(1000 + code)
    mine = solver_arguments[solver_arguments[:, 0] == code]
    assert mine.shape[0] > 1000, (name, mine.shape)  # the solves do call it
    x = np.concatenate([rng.uniform(lo, hi, 700_000), rng.normal(0, 1.0, 300_000) * (hi - lo) * 0.02, mine[:, 1]])
    y = None
    if name == "hypot":
        y = np.concatenate([rng.uniform(lo, hi, 700_000), rng.normal(0, 1.0, 300_000) * 3.0, mine[:, 2]])
    finite = np.isfinite(x) if y is None else (np.isfinite(x) & np.isfinite(y))
    x = x[finite]
    y = None if y is None else y[finite]
    assert x.shape[0] >= 1_000_000
    d = ulp_diff(orc_det.math(name, x, y), orc_libm.math(name, x, y))
    n_solver = mine.shape[0]
    worst_solver = d[-n_solver:].max() if n_solver else 0
    assert d.max() <= ULP_BOUND[name], (name, "max ulp", d.max(), "at", x[d.argmax()], "solver's own arguments:", worst_solver)
    # and the typical case is exact agreement
    print(name, "points", x.shape[0], "of which the solver's own", n_solver, "max ulp", d.max(), "(solver's own:", worst_solver,
          ") exact", round(float((d == 0).mean()), 4))
    assert (d == 0).mean() > EXACT_FRAC[name], (name, (d == 0).mean())
