"""Pin the CPU oracle against vectors produced by the reference's own Python modules
(tests/golden/make_golden.py imported /root/reference/scripts/utils/*.py and the variant solver).
These run for both oracle builds (libm and detmath)."""
import numpy as np
import pytest

from conftest import GOLDEN

RP_COG = 1


@pytest.fixture(scope="module")
def leaf():
    return dict(np.load(GOLDEN / "leaf_vectors.npz"))


@pytest.fixture(scope="module")
def variant():
    return dict(np.load(GOLDEN / "variant_vectors.npz"))


@pytest.fixture(params=["libm", "det"])
def orc(request, orc_libm, orc_det):
    return orc_libm if request.param == "libm" else orc_det


def close(a, b, rtol=1e-12, atol=1e-13):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_kinematic_propagate(orc, leaf):
    dt, wb = float(leaf["dt"]), float(leaf["wb"])
    got = np.stack([orc.propagate(x, u, dt, wb, RP_COG) for x, u in zip(leaf["prop_x"], leaf["prop_u"])])
    close(got, leaf["prop_out"])


def test_kinematic_propagate_rear_axle(request, orc):
    """RearCenter branch (utils.cpp:266-272) against the reference's own rear-axle step, scripts/1-lqr-pathtracking.py:134-140
    (`update`), imported headless by make_golden.py: x, y, v are the same expressions; yaw is `v / WB * tan(d) * dt` there
    and `v * tan(d) * dt / wb` in the C++ — the same value up to the rounding of three operations."""
    g = dict(np.load(GOLDEN / "rear_axle_vectors.npz"))
    dt, wb = float(g["dt"]), float(g["wb"])
    got = np.stack([orc.propagate(x, u, dt, wb, 0) for x, u in zip(g["x"], g["u"])])
    close(got, g["out"])
    libm = request.node.callspec.params["orc"] == "libm"
    if libm:  # the same libm on both sides: position and speed to the bit, yaw within the re-association
        assert np.array_equal(got[:, :3], g["out"][:, :3])
    # yaw' = yaw + increment: the two increments are <= 2 ulp (of the increment) apart, the sum rounds once more; measured in
    # units of the largest of the three magnitudes involved (a small yaw' after cancellation has a finer spacing of its own)
    inc = g["x"][:, 2] * np.tan(g["u"][:, 1]) * dt / wb
    unit = np.spacing(np.maximum(np.maximum(np.abs(g["x"][:, 3]), np.abs(inc)), np.abs(g["out"][:, 3])))
    err = np.abs(got[:, 3] - g["out"][:, 3]) / unit
    assert err.max() <= (2.0 if libm else 4.0), err.max()
    # chained, as const_velo_prediction / forward_pass chain it (cilqr_solver.cpp:182-197, 442-461)
    x = g["chain_x0"].copy()
    for u, want in zip(g["chain_u"], g["chain_out"]):
        x = orc.propagate(x, u, dt, wb, 0)
        close(x, want, rtol=1e-12, atol=1e-12)


def test_model_derivatives(orc, leaf):
    dt, wb = float(leaf["dt"]), float(leaf["wb"])
    N = leaf["md_u"].shape[0]
    A, B = orc.model_derivatives(leaf["md_x"], leaf["md_u"], dt, wb, N, RP_COG)
    close(A, leaf["md_A"])
    close(B, leaf["md_B"])


def test_const_velo_prediction(orc, leaf):
    p = dict(N=40, dt=float(leaf["dt"]), wheelbase=float(leaf["wb"]), reference_point=RP_COG)
    got = orc.const_velo_prediction(full_params(**p), leaf["cvp_x0"])
    close(got, leaf["cvp_out"])


def test_front_rear_centers_and_derivatives(orc, leaf):
    wb = float(leaf["wb"])
    for i, st in enumerate(leaf["fr_state"]):
        f, r = orc.front_rear(st, wb, RP_COG)
        close(f, leaf["fr_front"][i])
        close(r, leaf["fr_rear"][i])
        fd, rd = orc.front_rear_derivatives(st[3], wb, RP_COG)  # oracle: [state][point]; python: [point][state]
        close(fd.T, leaf["frd_front"][i])
        close(rd.T, leaf["frd_rear"][i])


def test_ellipsoid_margin_and_gradient(orc, leaf):
    for i in range(leaf["em_pnt"].shape[0]):
        obs = np.array([leaf["em_cen"][i, 0], leaf["em_cen"][i, 1], leaf["em_theta"][i]])
        ab = np.array([leaf["em_a"][i], leaf["em_b"][i]])
        close(orc.safety_margin(leaf["em_pnt"][i], obs, ab), leaf["em_margin"][i], rtol=1e-11, atol=1e-11)
        close(orc.safety_margin_derivatives(leaf["em_pnt"][i], obs, ab), leaf["em_grad"][i], rtol=1e-11, atol=1e-12)


def test_exp_barrier_and_bounds(orc, leaf):
    for i in range(leaf["eb_c"].shape[0]):
        c, q1, q2 = float(leaf["eb_c"][i]), float(leaf["eb_q1"][i]), float(leaf["eb_q2"][i])
        close(orc.exp_barrier(c, q1, q2), leaf["eb_b"][i])
        bd, bdd = orc.exp_barrier_dH(c, leaf["eb_cdot"][i], q1, q2)
        close(bd, leaf["eb_bdot"][i])
        close(bdd, leaf["eb_bddot"][i])
    close(leaf["bc_var"] - leaf["bc_bound"], leaf["bc_upper"], rtol=0, atol=0)
    close(leaf["bc_bound"] - leaf["bc_var"], leaf["bc_lower"], rtol=0, atol=0)


def full_params(**kw):
    base = dict(N=30, max_iter=100, solve_type=0, reference_point=1, use_last_solution=0, reserved0=0, dt=0.1,
                w_pos=1.0, w_vel=1.0, w_yaw=20.0, w_acc=0.5, w_stl=20.0, obstacle_exp_q1=5.5, obstacle_exp_q2=5.75,
                state_exp_q1=3.0, state_exp_q2=3.5, alm_rho_init=20.0, alm_gamma=0.0, max_rho=20.0, max_mu=120.0,
                init_lamb=0.0, lamb_decay=0.5, lamb_amplify=2.0, max_lamb=1000.0, convergence_threshold=0.01,
                accept_step_threshold=0.5, wheelbase=2.8, width=2.0, length=4.5, velo_max=15.0, velo_min=0.0,
                yaw_lim=1.57, acc_max=3.0, acc_min=-3.0, stl_lim=0.12, d_safe=1.0)
    base.update(kw)
    return base


def test_obstacle_constraint_chain(orc, leaf):
    """get_obstacle_avoidance_constr(+derivatives): the Python module uses a = L/2 + d_safe + r while
    the C++ path uses a = L/2 + 6 d_safe + r (utils.cpp:389); length' = L - 10 d_safe maps one onto the other."""
    width = float(leaf["oc_width"])
    ow, ol, ds = leaf["oc_attr"]
    assert ow == width
    p = full_params(wheelbase=float(leaf["wb"]), width=width, length=float(ol - 10 * ds), d_safe=float(ds))
    ab = orc.ellipsoid_scales([width, ol - 10 * ds, ds], 0.5 * width)
    close(ab, leaf["oc_ab"], rtol=1e-15)
    for i in range(leaf["oc_ego"].shape[0]):
        ego = leaf["oc_ego"][i]
        ob = leaf["oc_obs"][i][[0, 1, 3]]  # python obstacles are (x, y, v, yaw)
        c2 = orc.obstacle_constr(p, ego, ob)
        close(c2, [leaf["oc_front"][i], leaf["oc_rear"][i]], rtol=1e-10, atol=1e-10)
        f, r = orc.obstacle_constr_derivatives(p, ego, ob)
        close(f, leaf["ocd_front"][i], rtol=1e-10, atol=1e-11)
        close(r, leaf["ocd_rear"][i], rtol=1e-10, atol=1e-11)


# ---- composite functions against the reference's Python variant solver ------------------------
def variant_setup(orc, v):
    from oracle import Scene
    N = int(v["N"])
    acc_max, acc_min, stl_lim, velo_max, velo_min = v["bounds"]
    sw, cw = v["state_weight"], v["ctrl_weight"]
    assert sw[0, 0] == sw[1, 1] and sw[3, 3] == 0.0
    ow, ol, ds = v["attr"]
    p = full_params(N=N, dt=float(v["dt"]), wheelbase=float(v["wb"]), reference_point=1,
                    w_pos=float(sw[0, 0]), w_vel=float(sw[2, 2]), w_yaw=0.0, w_acc=float(cw[0, 0]), w_stl=float(cw[1, 1]),
                    obstacle_exp_q1=float(v["exp_q1"]), obstacle_exp_q2=float(v["exp_q2"]),
                    state_exp_q1=float(v["exp_q1"]), state_exp_q2=float(v["exp_q2"]),
                    width=float(v["width"]), length=float(ol - 10 * ds), d_safe=float(ds),
                    velo_max=float(velo_max), velo_min=float(velo_min), acc_max=float(acc_max), acc_min=float(acc_min),
                    stl_lim=float(stl_lim))
    lane = v["lane"]
    obs = v["obs"][:, :, [0, 1, 3]]
    # road borders at +-1e9: both lateral barriers evaluate to exp(-3.5e9) == 0 exactly
    scene = Scene(lane[:, 0], lane[:, 1], np.zeros(lane.shape[0]), obs, [1e9, -1e9], float(v["ref_velo"]))
    return orc.solver(p), scene, p


@pytest.mark.parametrize("tag", ["init", "mid", "final"])
def test_variant_total_cost(orc, variant, tag):
    s, scene, _ = variant_setup(orc, variant)
    J = s.total_cost(variant[tag + "_u"], variant[tag + "_x"], scene)
    close(J, float(variant[tag + "_J"]), rtol=1e-12)


def test_variant_known_answer(orc, variant):
    """the end-to-end number quoted in SURVEY.md §8(c)/BASELINE.md §2"""
    assert "Tolerance condition satisfied. 39" in str(variant["stdout"])
    s, scene, _ = variant_setup(orc, variant)
    J = s.total_cost(variant["final_u"], variant["final_x"], scene)
    close(J, 429.90589797075575, rtol=1e-12)


@pytest.mark.parametrize("tag", ["init", "mid", "final"])
def test_variant_cost_derivatives(orc, variant, tag):
    s, scene, _ = variant_setup(orc, variant)
    d = s.cost_derivatives(variant[tag + "_u"], variant[tag + "_x"], scene)
    # the variant adds the state barriers on row 0 as well (the C++ path starts at k = 1, cs:502)
    close(d["l_u"], variant[tag + "_l_u"], rtol=1e-10, atol=1e-10)
    close(d["l_uu"], variant[tag + "_l_uu"], rtol=1e-10, atol=1e-10)
    close(d["l_x"][1:], variant[tag + "_l_x"][1:], rtol=1e-9, atol=1e-9)
    close(d["l_xx"][1:], variant[tag + "_l_xx"][1:], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tag", ["mid", "final"])
def test_variant_backward_and_forward(orc, variant, tag):
    """backward_pass at lamb = 0 (where the variant's state-space regularisation vanishes) and
    forward_pass(alpha = 0.5).  Row 0 of l_x/l_xx never enters the sweep, so the k = 0 difference
    noted above does not matter."""
    s, scene, p = variant_setup(orc, variant)
    u, x = variant[tag + "_u"], variant[tag + "_x"]
    d, K, dV, st = s.backward_pass(u, x, 0.0, scene)
    assert st == 0
    close(d, variant[tag + "_d"], rtol=1e-7, atol=1e-9)
    close(K, variant[tag + "_K"], rtol=1e-7, atol=1e-9)
    close(dV[0] + dV[1], float(variant[tag + "_dV"]), rtol=1e-7, atol=1e-9)
    nu, nx = orc.forward_pass(p, u, x, variant[tag + "_d"], variant[tag + "_K"], 0.5)
    close(nu, variant[tag + "_fw_u"], rtol=1e-11, atol=1e-12)
    close(nx, variant[tag + "_fw_x"], rtol=1e-11, atol=1e-12)
