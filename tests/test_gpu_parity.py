"""Parity of the HIP path (through the C-ABI) against the CPU oracle.  Needs an MI355X.

Bar: BIT-EXACT against the oracle's detmath build (same elementary functions, same operation
order) for every stage and for whole solves including the per-iteration decision trace; within
1e-5 (the tolerance north_star states) against the libm build wherever the libm build's own result is
determined to 1e-5 by its input (test_libm_gap_is_input_conditioning).
"""
import numpy as np
import pytest

from conftest import oracle_scene

pytestmark = pytest.mark.gpu

TOL_LIBM = 1e-5  # north_star: trajectories and final cost within 1e-5 of the reference CPU solver


def rehearsal_size(B):
    """CILQR_TEST_SHRINK=k divides the batch sizes of the round-6 tests by k (never below 8): the same test bodies rehearsed on the
    wave64 emulator of tests/emu/, where a solve takes a tenth of a second instead of microseconds.  Unset on a GPU."""
    import os
    k = int(os.environ.get("CILQR_TEST_SHRINK", "1"))
    return max(8, B // k) if k > 1 else B


def eq_bits(a, b, what=""):
    """equal as IEEE values (+0 == -0, NaN == NaN positionally)"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    bad = ~((a == b) | both_nan)
    if bad.any():
        idx = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {a.size} differ; first at {tuple(idx)}: "
                             f"{a[tuple(idx)]!r} vs {b[tuple(idx)]!r} (max abs diff {np.nanmax(np.abs(a - b))})")


@pytest.fixture(scope="module")
def engines(pkg, scenarios):
    """one device engine per (scenario, N) as needed"""
    cache = {}

    def get(name, N, dev=False, **over):
        """dev=True: an engine on libcilqr_amd_dev.so (the testing aids and the cycle accounting live there)"""
        key = (name, N, dev, tuple(sorted(over.items())))
        if key not in cache:
            cfg, sc = scenarios[name]
            p = pkg.params_from_config(cfg, N=N, **over)
            eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc), dev=dev)
            cache[key] = (eng, p, sc)
        return cache[key]

    yield get
    for eng, _, _ in cache.values():
        eng.close()


def test_detmath_device_bitexact(pkg, orc_det, engines):
    eng, _, _ = engines("two_straight", 30)
    rng = np.random.default_rng(11)
    ranges = {0: (-745, 710), 1: (-1e4, 1e4), 2: (-1e4, 1e4), 3: (-1e4, 1e4), 4: (-1e3, 1e3)}
    names = {0: "exp", 1: "sin", 2: "cos", 3: "tan", 4: "atan"}
    for f, (lo, hi) in ranges.items():
        x = rng.uniform(lo, hi, 20000)
        x[:8] = [0.0, np.nan, np.inf, -np.inf, 1e300, -1e-310, 5e-324, -0.0]
        eq_bits(eng.detmath(f, x), orc_det.math(names[f], x), names[f])
    # the wave-uniform shortcuts (every lane of a wavefront below pi/4, resp. 7/16): whole arrays of small
    # arguments take them, arrays that straddle the thresholds take both paths inside one launch
    edge = np.array([0.785, np.nextafter(0.785, 0), np.nextafter(0.785, 1), np.pi / 4, 0.4375,
                     np.nextafter(0.4375, 0), np.nextafter(0.4375, 1), 0.0, -0.0, 5e-324, -5e-324, 1e-300])
    small = np.concatenate([rng.uniform(-0.78, 0.78, 8192), np.tile(np.concatenate([edge, -edge]), 4),
                            rng.uniform(-0.43, 0.43, 8192), rng.uniform(-0.9, 0.9, 4096)])
    for f, nm in ((1, "sin"), (2, "cos"), (3, "tan"), (4, "atan"), (8, "sin"), (9, "cos"), (10, "tan")):
        eq_bits(eng.detmath(f, small), orc_det.math(nm, small), f"{nm} (small arguments, code {f})")
    big = rng.uniform(-1e4, 1e4, 20000)
    for f, nm in ((8, "sin"), (9, "cos"), (10, "tan")):
        eq_bits(eng.detmath(f, big), orc_det.math(nm, big), f"{nm} (pinned coefficients)")
    x, y = rng.uniform(-300, 300, 20000), rng.uniform(-300, 300, 20000)
    eq_bits(eng.detmath(5, x, y), orc_det.math("hypot", x, y), "hypot")
    eq_bits(eng.detmath(6, x, y), x / y, "div")
    eq_bits(eng.detmath(7, x), np.sqrt(np.abs(x)), "sqrt")


def random_trajectories(pkg, orc, p, sc, B, seed, rough=0.0):
    """physically plausible (u, x): roll random smooth controls out from perturbed starts"""
    rng = np.random.default_rng(seed)
    N = p.N
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, seed)
    us = np.zeros((B, N, 2))
    xs = np.zeros((B, N + 1, 4))
    for b in range(B):
        acc = np.cumsum(rng.normal(0, 0.2, N)) * 0.3
        stl = np.cumsum(rng.normal(0, 0.01, N)) * 0.5
        us[b] = np.stack([acc, stl], axis=1)
        xs[b, 0] = x0[b]
        for i in range(N):
            xs[b, i + 1] = orc.propagate(xs[b, i], us[b, i], p.dt, p.wheelbase, p.reference_point)
        if rough:
            xs[b, 1:] += rng.normal(0, rough, (N, 4))
    return us, xs


@pytest.mark.parametrize("name,N", [("two_straight", 30), ("three_bend", 50), ("three_straight", 30), ("two_borrow", 100)])
def test_stages_bitexact(pkg, orc_det, engines, name, N):
    """init trajectory, ref points, total cost, cost derivatives + model Jacobians, backward pass,
    forward pass for all 20 trial step sizes: every number equal to the oracle's."""
    eng, p, sc = engines(name, N)
    scene = oracle_scene(sc)
    B = 24
    us, xs = random_trajectories(pkg, orc_det, p, sc, B, seed=N + len(name), rough=0.02)
    # init trajectory
    x0 = xs[:, 0].copy()
    xi = eng.init_traj(x0)
    for b in range(B):
        eq_bits(xi[b], orc_det.const_velo_prediction(p, x0[b]), "init_traj")
    # ref points
    ref, idx = eng.ref_points(xs)
    for b in range(B):
        r0, i0 = orc_det.ref_points(xs[b], scene)
        eq_bits(idx[b], i0, "ref idx")
        eq_bits(ref[b], r0, "ref pts")
    # cost
    J = eng.total_cost(us, xs)
    s = orc_det.solver(p)
    for b in range(B):
        eq_bits(J[b], s.total_cost(us[b], xs[b], scene), "total cost")
    # derivatives
    dv = eng.cost_derivatives(us, xs)
    for b in range(B):
        od = s.cost_derivatives(us[b], xs[b], scene)
        for k in ("l_x", "l_u", "l_xx", "l_uu"):
            eq_bits(dv[k][b], od[k], k)
        A, Bm = orc_det.model_derivatives(xs[b], us[b], p.dt, p.wheelbase, N, p.reference_point)
        eq_bits(dv["A"][b], A, "A")
        eq_bits(dv["B"][b], Bm, "B")
    # backward pass at several regularisation levels
    for lamb in (0.0, 2.0, 64.0):
        d, K, dV, st = eng.backward_pass(us, xs, lamb)
        for b in range(B):
            od, oK, odV, ost = s.backward_pass(us[b], xs[b], lamb, scene)
            assert st[b] == ost, ("bp status", b, lamb)
            eq_bits(d[b], od, "d")
            eq_bits(K[b], oK, "K")
            if ost == 0:
                eq_bits(dV[b], odV, "dV")
    # forward pass + cost for all alphas, using the gains of lamb = 2
    d, K, dV, st = eng.backward_pass(us, xs, 2.0)
    nu, nx, Jt = eng.forward_pass(us, xs, d, K)
    for b in range(0, B, 3):
        for a in range(20):
            ou, ox = orc_det.forward_pass(p, us[b], xs[b], d[b], K[b], 2.0 ** -a)
            eq_bits(nu[b, a], ou, "fw u")
            eq_bits(nx[b, a], ox, "fw x")
            if np.all(np.isfinite(ox)):
                eq_bits(Jt[b, a], s.total_cost(ou, ox, scene), "fw J")


def test_backward_pass_failure_matches(pkg, orc_det, engines):
    """non-PD Q_uu: status and the partially filled d, K must match (cs:415-420)."""
    eng, p, sc = engines("two_straight", 30, w_acc=-40.0)
    scene = oracle_scene(sc)
    us, xs = random_trajectories(pkg, orc_det, p, sc, 8, seed=3)
    d, K, dV, st = eng.backward_pass(us, xs, 0.0)
    s = orc_det.solver(p)
    assert (st == 2).any()
    for b in range(8):
        od, oK, odV, ost = s.backward_pass(us[b], xs[b], 0.0, scene)
        assert st[b] == ost
        eq_bits(d[b], od, "d (fail)")
        eq_bits(K[b], oK, "K (fail)")


def compare_solves(out, ref_list, what):
    for b, r in enumerate(ref_list):
        eq_bits(out["u"][b], r["u"], f"{what} u[{b}]")
        eq_bits(out["x"][b], r["x"], f"{what} x[{b}]")
        res, rr = out["res"][b], r["res"]
        for f in ("J_init", "J_final", "iters", "end_reason", "final_status", "ls_trials", "cost_evals"):
            same = (res[f] == rr[f]) or (np.isnan(res[f]) and np.isnan(rr[f]))
            assert same, (what, b, f, res[f], rr[f])
        tr, rt = out["trace"][b][:res["trace_len"]], r["trace"]
        assert len(tr) == len(rt)
        for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
            eq_bits(tr[f], rt[f], f"{what} trace.{f}[{b}]")


@pytest.mark.parametrize("name,N,B", [("two_straight", 50, 96), ("three_bend", 50, 96), ("two_borrow", 30, 48),
                                      ("three_straight", 30, 48), ("three_bend", 100, 32)])
def test_solve_bitexact_with_trace(pkg, orc_det, engines, name, N, B):
    """whole solves from perturbed starts: trajectories, costs, counters and the per-iteration
    decision trace (status / alpha / lambda / cost) identical to the oracle."""
    eng, p, sc = engines(name, N, use_last_solution=0)
    scene = oracle_scene(sc)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xBEEF + N)
    out = eng.solve_batch(x0, trace_cap=128)
    s = orc_det.solver(p)
    refs = []
    for b in range(B):
        s.reset()
        refs.append(s.solve(x0[b], scene))
    compare_solves(out, refs, f"{name}/N{N}")


def test_solve_yaml_start_bitexact(pkg, orc_det, engines):
    """BASELINE config 1: the unperturbed YAML start (ego exactly on the reference line — the
    ill-conditioned case of SURVEY.md §7) is reproduced exactly because host and device share their
    elementary functions."""
    for name in ("two_straight", "three_bend"):
        eng, p, sc = engines(name, 50, use_last_solution=0)
        scene = oracle_scene(sc)
        out = eng.solve_batch(sc.ego_state[None], trace_cap=128)
        s = orc_det.solver(p)
        compare_solves(out, [s.solve(sc.ego_state, scene)], name + " yaml start")


def test_solve_ticks_and_warm_start(pkg, orc_det, engines):
    """closed loop over a few ticks (obstacle predictions start at the current tick,
    utils.cpp:88-103) with use_last_solution (cs:97-102,163-180)."""
    eng, p, sc = engines("three_straight", 30)
    assert p.use_last_solution == 1
    s = orc_det.solver(p)
    s.reset()
    x0 = sc.ego_state.copy()
    last_u = None
    for tick in range(4):
        scene = oracle_scene(sc, tick)
        out = eng.solve_batch(x0[None], tick=[tick], last_u=None if last_u is None else last_u[None], trace_cap=128)
        compare_solves(out, [s.solve(x0, scene)], f"tick {tick}")
        last_u = out["u"][0]
        x0 = out["x"][0, 1].copy()


def test_batched_closed_loop_with_warm_starts(pkg, orc_det, scenarios):
    """The step after the path (motion_planning.cpp:180-197) for a whole batch: 24 egos, 40 ticks, every ego
    advances to row 1 of its own solution and warm-starts from it (cs:163-180); each tick's batch is compared
    with 24 stateful oracle solvers."""
    cfg, sc = scenarios["three_straight"]
    p = pkg.params_from_config(cfg, N=30)
    assert p.use_last_solution == 1
    tab = pkg.SceneTable.from_scenario(sc)
    B, ticks = 24, 40
    assert sc.routes.shape[1] >= ticks + p.N + 1
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 31337)
    eng = pkg.BatchedCILQR(p, tab)
    solvers = [orc_det.solver(p) for _ in range(B)]
    for s_ in solvers:
        s_.reset()
    last_u = None
    iters = 0
    for tick in range(ticks):
        scene = oracle_scene(sc, tick)
        out = eng.solve_batch(x0, tick=np.full(B, tick, np.int32), last_u=last_u, trace_cap=128)
        refs = [solvers[b].solve(x0[b], scene) for b in range(B)]
        compare_solves(out, refs, f"closed loop tick {tick}")
        last_u = out["u"].copy()
        x0 = out["x"][:, 1].copy()
        iters += int(out["res"]["iters"].sum())
    assert iters > ticks * B  # more than one iteration per solve on average: the loop did real work
    eng.close()


def test_solve_param_sweep_and_mixed_scenarios(pkg, orc_det):
    """param_id / scenario_id indirection (BASELINE configs 4 and 5) at reduced size."""
    wl = pkg.workloads.config5(B_base=4, N=30)
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, trace_cap=128)
    from oracle import Scene
    sc0 = wl.scenes[0]
    scene = Scene(sc0.lane_x, sc0.lane_y, sc0.lane_yaw, sc0.obs, sc0.road_borders, sc0.ref_velo)
    refs = []
    for b in range(wl.B):
        s = orc_det.solver(wl.params[wl.param_id[b]])
        refs.append(s.solve(wl.x0[b], scene))
    compare_solves(out, refs, "config5")
    eng.close()
    wl = pkg.workloads.config4(B=16, N=100)
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, trace_cap=128)
    refs = []
    for b in range(wl.B):
        t = wl.scenes[wl.scenario_id[b]]
        scene = Scene(t.lane_x, t.lane_y, t.lane_yaw, t.obs, t.road_borders, t.ref_velo)
        refs.append(orc_det.solver(wl.params[wl.param_id[b]]).solve(wl.x0[b], scene))
    compare_solves(out, refs, "config4")
    eng.close()


@pytest.mark.parametrize("cfg,rows", [("2", 1024), ("3", 2048), ("5", 4096), ("4", 1024), ("2alm", 1024)])
def test_libm_gap_is_input_conditioning(pkg, orc_det, cfg, rows):
    """Against the oracle built on glibc's libm (what the reference binary links), on BASELINE configs 2 (all 1024
    trajectories), 3 and 5 (first 2048 / 4096), configs[3]'s rank-0 shard (first 1024: horizon 100, four scenarios,
    three of them with the RearCenter-free CoG model and road borders) and config 2 with the augmented Lagrangian:
      * every trajectory whose REFERENCE result is determined to 1e-5 by its input — the libm build moves by
        <= 1e-5 when one component of x0 moves by one unit in the last place — is within 1e-5 on u, x and J_final;
      * every trajectory outside the 1e-5 band is one the libm build itself does not reproduce to 1e-5 under
        such a move, and the HIP result is no further from it than 1.5 x what that move does (measured: <= 0.96 x).
    (tests/libm_tolerance.py holds the diagnosis; VERDICT r01 asked for a near-tie criterion — the traces turn
    out to part by smooth amplification under mostly identical decisions, not at near-ties)"""
    import libm_tolerance as lt
    wl = lt.make_workload(pkg, cfg)
    sel = np.arange(rows)
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    hip = eng.solve_batch(wl.x0[sel], wl.scenario_id[sel], wl.param_id[sel], wl.tick[sel])
    eng.close()
    # the HIP path is the detmath oracle's twin here too
    twin = orc_det.solve_batch(wl.params, lt.oracle_scenes(wl), wl.x0[sel], wl.scenario_id[sel], wl.param_id[sel],
                               wl.tick[sel], n_threads=8)
    eq_bits(hip["x"], twin["x"], "hip vs detmath oracle")
    hip_full = {k: np.zeros((wl.B,) + hip[k].shape[1:], dtype=hip[k].dtype) for k in ("u", "x", "res")}
    for k in hip_full:
        hip_full[k][sel] = hip[k]
    rep = lt.analyse(wl, hip_full, threads=8, rows=sel, symmetric=(cfg == "4"))
    brief = {k: v for k, v in rep.items() if k != "records"}
    if cfg == "4":
        # Horizon 100 over these scenarios is a chaotic map of the input: more than half of the solves are not
        # reproduced to 1e-5 by the libm build ITSELF after a one-ulp move of x0 (profiles/r03_libm_tolerance.json:
        # 5124 of the shard's 8192; three_straight is the tame one).  What can hold does: no trajectory is further from
        # the libm result than twice what such a move does to one of the two builds (measured: <= 1.6 x).
        assert rep["every_trajectory_obeys gap <= max(1e-5, 2 x spread)"], brief
        assert rep["ill_conditioned (libm spread > 1e-5)"] > rows // 4, brief
        return
    assert rep["well_conditioned_outside_1e-5"] == 0, brief
    assert rep["outside_1e-5"] == rep["outside_1e-5_with_spread_gt_1e-5"], brief
    assert rep["max_gap_over_spread_outside"] is None or rep["max_gap_over_spread_outside"] <= 1.5, brief
    assert rep["within_1e-5_frac"] >= 0.99, brief
    if cfg in ("3", "5", "2alm"):  # the CoG-model bend scenario is well-conditioned throughout; so is config 2 under ALM
        assert rep["outside_1e-5"] == 0 and rep["max_gap"] < 1e-7, brief


def test_libm_gap_of_the_single_ego_closed_loop(pkg):
    """BASELINE configs[0] (scenario_two_straight, one ego, horizon 50, 120 ticks): the libm build's loop and the HIP
    path's loop part ways after a few ticks — the YAML start lies ON the reference line, where the lateral-constraint
    gradient is 0/0-degenerate — and the tick that does it is a solve the libm build does not
    reproduce itself under a one-ulp move of its input: every tick's solve, given the libm loop's own state, obeys the
    same two rules as the batches above."""
    import libm_tolerance as lt
    rep = lt.analyse_config1(pkg, threads=8, gpu=True)
    brief = {k: v for k, v in rep.items() if k != "records"}
    assert rep["well_conditioned_outside_1e-5"] == 0, brief
    assert rep["outside_1e-5"] == rep["outside_1e-5_with_spread_gt_1e-5"], brief
    assert rep["max_gap_over_spread_outside"] is None or rep["max_gap_over_spread_outside"] <= 1.5, brief
    cl = rep["closed_loop"]
    first = cl["first_tick_whose_entering_state_differs_by_more_than_1e-5"]
    if first is not None:  # the solve of the tick before is on record as ill-conditioned
        cause = cl["the_solve_that_caused_it (tick before, same input to both builds)"]
        assert cause is None or cause["libm_spread_under_1ulp_x0"] > 1e-5, brief


def test_batch_order_invariance(pkg, engines):
    """race proxy: permuting the batch permutes the results, bit for bit."""
    eng, p, sc = engines("three_bend", 50, use_last_solution=0)
    B = 64
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 77)
    a = eng.solve_batch(x0)
    perm = np.random.default_rng(1).permutation(B)
    b = eng.solve_batch(x0[perm])
    eq_bits(a["u"][perm], b["u"], "perm u")
    eq_bits(a["x"][perm], b["x"], "perm x")
    assert (a["res"][perm] == b["res"]).all()


def test_solver_class_mirror(pkg, orc_det, scenarios):
    """the drop-in class: CILQRSolver(config).solve(x0, ref_waypoints, ref_velo, obs_preds, road_boaders)"""
    cfg, sc = scenarios["two_straight"]
    solver = pkg.CILQRSolver(cfg)
    obs = [pkg.RoutingLine(r[:, 0], r[:, 1], r[:, 2]) for r in sc.obstacles]
    u, x = solver.solve(sc.ego_state, sc.lane, sc.target_velocity, obs, sc.road_borders)
    ref = orc_det.solver(solver.params).solve(sc.ego_state, oracle_scene(sc))
    eq_bits(u, ref["u"], "class u")
    eq_bits(x, ref["x"], "class x")


def test_error_codes(pkg, engines):
    eng, p, sc = engines("two_straight", 30)
    T = sc.routes.shape[1]
    with pytest.raises(pkg.CilqrError) as e:
        eng.solve_batch(sc.ego_state[None], tick=[T - 5])
    assert e.value.code == -2  # CILQR_ERR_OBSTACLE_HORIZON
    with pytest.raises(pkg.CilqrError) as e:
        eng.solve_batch(sc.ego_state[None], scenario_id=[3])
    assert e.value.code == -1


def test_api_contract_details(pkg, orc_det, scenarios):
    """Small print of the C-ABI: a trace buffer shorter than the solve keeps the first records and reports its
    own length; parameter sets of one handle must share N and the solve type; tables can be replaced between
    calls of the same handle; B = 1 works like any other batch."""
    cfg, sc = scenarios["two_straight"]
    p = pkg.params_from_config(cfg, N=30)
    tab = pkg.SceneTable.from_scenario(sc)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 5, 4711)
    eng = pkg.BatchedCILQR(p, tab)
    full = eng.solve_batch(x0, trace_cap=128)
    short = eng.solve_batch(x0, trace_cap=3)
    assert (full["res"]["iters"] > 3).any()
    for b in range(5):
        n = int(full["res"]["iters"][b])
        assert full["res"]["trace_len"][b] == n
        assert short["res"]["trace_len"][b] == min(n, 3)
        assert (short["trace"][b][:min(n, 3)] == full["trace"][b][:min(n, 3)]).all()
    eq_bits(short["x"], full["x"], "x with a short trace buffer")
    one = eng.solve_batch(x0[2:3], trace_cap=128)
    eq_bits(one["x"][0], full["x"][2], "B = 1")
    assert (one["res"] == full["res"][2:3]).all()
    # new parameter table and new scenario table on the same handle
    cfg2, sc2 = scenarios["three_bend"]
    p2 = pkg.params_from_config(cfg2, N=40)
    eng.set_params(p2)
    eng.set_scenarios(pkg.SceneTable.from_scenario(sc2))
    y0 = pkg.workloads.perturbed_starts(sc2.ego_state, 4, 4712)
    out = eng.solve_batch(y0, trace_cap=128)
    from oracle import Scene
    t2 = pkg.SceneTable.from_scenario(sc2)
    scene2 = Scene(t2.lane_x, t2.lane_y, t2.lane_yaw, t2.obs, t2.road_borders, t2.ref_velo)
    compare_solves(out, [orc_det.solver(p2).solve(x, scene2) for x in y0], "tables replaced")
    # mixed N / mixed solve type in one table are refused
    for bad in ([p2, pkg.copy_params(p2, N=41)], [p2, pkg.copy_params(p2, solve_type=1)]):
        with pytest.raises(pkg.CilqrError) as e:
            eng.set_params(bad)
        assert e.value.code == -1
    eng.close()


def test_serial_and_parallel_reference_search_agree(pkg, orc_det, engines):
    """the lane-parallel reference-point search (+ proof) and the serial chain of cs:289-314 give the
    same solves; with wild gains the proof must fail sometimes and the fallback must take over."""
    eng, p, sc = engines("three_bend", 50, dev=True, use_last_solution=0)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 64, 4242)
    a = eng.solve_batch(x0, trace_cap=64)
    eng.set_debug_flags(pkg._lib.DBG_SERIAL_REF_SCAN)
    b = eng.solve_batch(x0, trace_cap=64)
    eng.set_debug_flags(0)
    eq_bits(a["u"], b["u"], "u")
    eq_bits(a["x"], b["x"], "x")
    assert (a["res"] == b["res"]).all() and (a["trace"] == b["trace"]).all()
    # forward passes with exaggerated gains: trial trajectories that double back / leave the lane
    scene = oracle_scene(sc)
    us, xs = random_trajectories(pkg, orc_det, p, sc, 12, seed=8)
    d, K, dV, st = eng.backward_pass(us, xs, 0.0)
    K2, d2 = K * 6.0, d * 25.0
    nu, nx, Jt = eng.forward_pass(us, xs, d2, K2)
    s = orc_det.solver(p)
    checked = 0
    for b_ in range(12):
        for a_ in range(20):
            ou, ox = orc_det.forward_pass(p, us[b_], xs[b_], d2[b_], K2[b_], 2.0 ** -a_)
            eq_bits(nx[b_, a_], ox, "wild fw x")
            if np.all(np.isfinite(ox)) and np.abs(ox).max() < 1e6:
                eq_bits(Jt[b_, a_], s.total_cost(ou, ox, scene), "wild fw J")
                checked += 1
    assert checked > 100


def test_uniform_and_lane_parallel_backward_agree(pkg, orc_det, engines):
    """the two device formulations of backward_pass (wave-uniform / lane-parallel) are
    interchangeable bit for bit, on success and on failure."""
    for name, N, over in (("three_bend", 50, {}), ("two_straight", 30, {"w_acc": -40.0})):
        eng, p, sc = engines(name, N, dev=True, **over)
        us, xs = random_trajectories(pkg, orc_det, p, sc, 16, seed=21, rough=0.01)
        for lamb in (0.0, 8.0):
            a = eng.backward_pass(us, xs, lamb)
            eng.set_debug_flags(pkg._lib.DBG_UNIFORM_BACKWARD)
            b = eng.backward_pass(us, xs, lamb)
            eng.set_debug_flags(0)
            assert (a[3] == b[3]).all()
            eq_bits(a[0], b[0], "d")
            eq_bits(a[1], b[1], "K")
            ok = a[3] == 0
            eq_bits(a[2][ok], b[2][ok], "dV")


def test_cpp_headless_planner_closed_loop(pkg, orc_det, scenarios):
    """host C++ -> C-ABI -> HIP: the closed loop of motion_planning.cpp:180-197 for a few ticks,
    against the oracle driven the same way (three_straight exercises use_last_solution)."""
    import importlib
    import subprocess
    build = importlib.import_module("toy-example-of-ilqr_amd.build")
    exe = build.build_examples()
    for name in ("two_straight", "three_straight"):
        cfg, sc = scenarios[name]
        path = pkg.config.SCENARIO_DIR / f"{name}.json"
        out = subprocess.run([str(exe), str(path), "4"], check=True, capture_output=True, text=True).stdout
        rows = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
        assert rows.shape == (4, 9)
        p = pkg.params_from_config(cfg)
        s = orc_det.solver(p)
        ego = sc.ego_state.copy()
        t = 0.0
        for i in range(4):
            index = int(t / sc.delta_t)
            r = s.solve(ego, oracle_scene(sc, index))
            ego = r["x"][1].copy()
            assert rows[i, 0] == index and rows[i, 7] == r["res"]["iters"]
            eq_bits(rows[i, 1:5], ego, f"{name} tick {i} ego")
            eq_bits(rows[i, 5:7], r["u"][0], f"{name} tick {i} u0")
            eq_bits(rows[i, 8], r["res"]["J_final"], f"{name} tick {i} J")
            t += sc.delta_t


# ---- solve_type "alm" (cs:88-93, 253-277, 377-378, 581-643, 665-680) -----------------------------
def test_alm_stages_bitexact(pkg, orc_det, engines):
    """augmented-Lagrangian cost / derivatives (non-symmetric Hessian) / multiplier proposals /
    backward pass with given multipliers."""
    for name, N in (("two_straight", 30), ("three_bend", 50)):
        eng, p, sc = engines(name, N, solve_type=1, use_last_solution=0)
        scene = oracle_scene(sc)
        B = 12
        us, xs = random_trajectories(pkg, orc_det, p, sc, B, seed=5 + N, rough=0.02)
        Ccols = 8 + 2 * sc.obstacles.shape[0]
        rng = np.random.default_rng(N)
        mu = rng.uniform(0, 3, (B, N, Ccols)) * (rng.uniform(0, 1, (B, N, Ccols)) < 0.6)
        rho = rng.uniform(5, 25, B)
        eng.set_alm_state(mu, rho)
        J = eng.total_cost(us, xs)
        dv = eng.cost_derivatives(us, xs)
        d, K, dV, st = eng.backward_pass(us, xs, 1.0)
        _, mun, _ = eng.get_alm_state(B)
        import ctypes as C
        for b in range(B):
            s = orc_det.solver(p)
            # inject the same multipliers into the oracle instance through a zero-iteration solve hook:
            # the oracle exposes its state only through solve(), so drive it with its own setters
            orc_det.lib.orc_set_alm_state.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int32]
            orc_det.lib.orc_set_alm_state(s.h, mu[b].ctypes.data, float(rho[b]), Ccols)
            eq_bits(J[b], s.total_cost(us[b], xs[b], scene), "alm cost")
            od = s.cost_derivatives(us[b], xs[b], scene)
            for k in ("l_x", "l_u", "l_xx", "l_uu"):
                eq_bits(dv[k][b], od[k], "alm " + k)
            omun = np.zeros((N, Ccols))
            orc_det.lib.orc_get_alm_next.argtypes = [C.c_void_p, C.c_void_p]
            orc_det.lib.orc_get_alm_next(s.h, omun.ctypes.data)
            eq_bits(mun[b], omun, "alm mu_next")
            o_d, o_K, o_dV, o_st = s.backward_pass(us[b], xs[b], 1.0, scene)
            assert st[b] == o_st
            eq_bits(d[b], o_d, "alm d")
            eq_bits(K[b], o_K, "alm K")
            if o_st == 0:
                eq_bits(dV[b], o_dV, "alm dV")
        assert not np.array_equal(dv["l_xx"], np.swapaxes(dv["l_xx"], -1, -2))  # genuinely non-symmetric


@pytest.mark.parametrize("name,N,B", [("two_straight", 30, 32), ("three_bend", 50, 32)])
def test_alm_solve_bitexact_with_trace(pkg, orc_det, engines, name, N, B):
    eng, p, sc = engines(name, N, solve_type=1, use_last_solution=0)
    scene = oracle_scene(sc)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xA1 + N)
    out = eng.solve_batch(x0, trace_cap=128)
    refs = []
    for b in range(B):
        refs.append(orc_det.solver(p).solve(x0[b], scene))
    compare_solves(out, refs, f"alm {name}/N{N}")
    assert (out["res"]["iters"] > 1).any()


def test_alm_closed_loop_keeps_multipliers(pkg, orc_det, engines):
    """use_last_solution + alm: multipliers and rho survive from tick to tick (cs:88-93)."""
    eng, p, sc = engines("three_straight", 30, solve_type=1)
    assert p.use_last_solution == 1
    s = orc_det.solver(p)
    x0 = sc.ego_state.copy()
    last_u = None
    for tick in range(3):
        out = eng.solve_batch(x0[None], tick=[tick], last_u=None if last_u is None else last_u[None], trace_cap=128)
        compare_solves(out, [s.solve(x0, oracle_scene(sc, tick))], f"alm tick {tick}")
        last_u = out["u"][0]
        x0 = out["x"][0, 1].copy()


@pytest.mark.not_yet_run_on_hardware
@pytest.mark.parametrize("name,N,B", [("three_bend", 50, 600), ("two_borrow", 100, 200), ("three_straight", 30, 300)])
def test_alm_in_pairs_per_wavefront(pkg, orc_det, scenarios, name, N, B):
    """Round 6 (VERDICT r05 task 2): the augmented Lagrangian on the grouped kernel — two trajectories per wavefront, the long
    layout at every horizon, dense 32-double rows, rho per trajectory in GrpSt, multipliers in HBM — chosen with
    cilqr_set_group_mode(2) (the kernels were written while the GPU pool was closed and are bit-exact on the wave64 emulator,
    tests/test_emulator.py; the default dispatch stays on k_solve's builds until this test has passed on a GPU).  Cold solves with
    the decision trace, then a second call warm-started from the first one's plan, which continues from the multipliers the
    handle kept (hpp:106-112, cs:88-93): == stateful oracle solvers on a sample, == the lone-wavefront builds on every row."""
    cfg, sc = scenarios[name]
    B = rehearsal_size(B)
    p = pkg.params_from_config(cfg, N=N, solve_type=1, use_last_solution=1)
    tab = pkg.SceneTable.from_scenario(sc)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xA1A + N)
    outs = {}
    for mode in (2, 0):
        eng = pkg.BatchedCILQR(p, tab)
        eng.set_group_mode(mode)
        first = eng.solve_batch(x0, trace_cap=128)
        assert eng.last_launch_info()["trajectories_per_wavefront"] == (2 if mode == 2 else 1)
        second = eng.solve_batch(first["x"][:, 1].copy(), tick=np.ones(B, np.int32), last_u=first["u"], trace_cap=128)
        mu, _mu_next, rho = eng.get_alm_state(B)
        outs[mode] = (first, second, mu, rho)
        eng.close()
    for k in (0, 1):
        a, b = outs[2][k], outs[0][k]
        eq_bits(a["u"], b["u"], f"pairs vs lone, call {k}: u")
        eq_bits(a["x"], b["x"], f"pairs vs lone, call {k}: x")
        assert (a["res"] == b["res"]).all() and (a["trace"] == b["trace"]).all(), k
    eq_bits(outs[2][2], outs[0][2], "multipliers after two calls")
    eq_bits(outs[2][3], outs[0][3], "rho after two calls")
    for b in range(0, B, max(1, B // 24)):   # the oracle on a sample (its ALM solves are the slow ones)
        s = orc_det.solver(p)
        s.reset()
        r1 = s.solve(x0[b], oracle_scene(sc))
        compare_solves({k: v[b:b + 1] for k, v in outs[2][0].items()}, [r1], f"alm pairs {name} b={b} first")
        r2 = s.solve(r1["x"][1], oracle_scene(sc, 1))
        compare_solves({k: v[b:b + 1] for k, v in outs[2][1].items()}, [r2], f"alm pairs {name} b={b} second")


# ---- BASELINE.json configurations at (or near) full size ------------------------------------------
def test_config2_full_batch_bitexact(pkg, orc_det):
    """configs[1], the benchmark workload: all 1024 trajectories, every output field, against the oracle."""
    wl = pkg.workloads.config2()
    assert wl.B == 1024 and wl.N == 50
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    from oracle import Scene
    s0 = wl.scenes[0]
    scene = Scene(s0.lane_x, s0.lane_y, s0.lane_yaw, s0.obs, s0.road_borders, s0.ref_velo)
    ref = orc_det.solve_batch(wl.params, scene, wl.x0, n_threads=16)
    eq_bits(out["u"], ref["u"], "config2 u")
    eq_bits(out["x"], ref["x"], "config2 x")
    for f in ("J_init", "J_final", "iters", "end_reason", "final_status", "ls_trials", "cost_evals"):
        assert np.array_equal(out["res"][f], ref["res"][f]), f
    # the statistics bench.py prints
    assert int(out["res"]["iters"].sum()) == 21955 and int(out["res"]["ls_trials"].sum()) == 57348
    eng.close()


def test_full_size_configs_bitexact(pkg, orc_det):
    """BASELINE configs[2] and [4] at their full sizes ([3]: test_config4_every_rank_shard_and_stats), and the batch of
    configs[1] once more with the augmented-Lagrangian solve type: every trajectory, every output field, bit
    for bit against the oracle (OpenMP over the host cores the box grants)."""
    import os
    from oracle import Scene
    threads = max(1, min(16, os.cpu_count() or 1))
    alm2 = pkg.workloads.config2()
    alm2 = pkg.workloads.Workload("config2_alm", [pkg.copy_params(q, solve_type=1) for q in alm2.params], alm2.scenes, alm2.x0,
                                  alm2.scenario_id, alm2.param_id, alm2.tick)   # the benchmark batch, augmented Lagrangian
    alm3 = pkg.workloads.config3(B=3072)  # large enough for the lone-wavefront, two-per-SIMD kernels, ALM flavour
    alm3 = pkg.workloads.Workload("config3_alm_B3072", [pkg.copy_params(q, solve_type=1) for q in alm3.params], alm3.scenes,
                                  alm3.x0, alm3.scenario_id, alm3.param_id, alm3.tick)
    cases = (pkg.workloads.config3(), pkg.workloads.config5(), alm2, alm3)
    alm4 = pkg.workloads.config4(B=1100)  # two rows per lane, ALM, lone wavefronts two per SIMD (helper switched off below)
    alm4 = pkg.workloads.Workload("config4_alm_B1100_nohelper", [pkg.copy_params(q, solve_type=1) for q in alm4.params],
                                  alm4.scenes, alm4.x0, alm4.scenario_id, alm4.param_id, alm4.tick)
    mid = pkg.workloads.config3(B=2048)  # just above the helper range: lone wavefronts, two per SIMD, horizon at run time
    mid = pkg.workloads.Workload("config3_bend_B2048_N40", [pkg.copy_params(q, N=40) for q in mid.params], mid.scenes, mid.x0,
                                 mid.scenario_id, mid.param_id, mid.tick)
    # horizons above 63 in batches beyond the helper range: the builds that stream the cost expansion from global memory
    # (horizon at run time; the compile-time N = 100 build runs in test_config4_every_rank_shard_and_stats), both models
    long_b = pkg.workloads.config3(B=3400)
    long_b = pkg.workloads.Workload("config3_bend_B3400_N80", [pkg.copy_params(q, N=80) for q in long_b.params], long_b.scenes,
                                    long_b.x0, long_b.scenario_id, long_b.param_id, long_b.tick)
    long_s = pkg.workloads.config2(B=2600)
    long_s = pkg.workloads.Workload("config2_straight_B2600_N71", [pkg.copy_params(q, N=71) for q in long_s.params], long_s.scenes,
                                    long_s.x0, long_s.scenario_id, long_s.param_id, long_s.tick)
    cases = cases + (alm4, mid, long_b, long_s)
    for wl in cases:
        eng = pkg.BatchedCILQR(wl.params, wl.scenes)
        if wl.name.endswith("_nohelper"):
            eng.set_helper_mode(0)
        out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
        eng.close()
        scenes = [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs if s.obs.shape[0] else None, s.road_borders, s.ref_velo)
                  for s in wl.scenes]
        ref = orc_det.solve_batch(wl.params, scenes, wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=threads)
        eq_bits(out["u"], ref["u"], f"{wl.name} u")
        eq_bits(out["x"], ref["x"], f"{wl.name} x")
        for f in ("J_init", "J_final"):
            eq_bits(out["res"][f], ref["res"][f], f"{wl.name} {f}")
        for f in ("iters", "end_reason", "final_status", "ls_trials", "cost_evals"):
            assert np.array_equal(out["res"][f], ref["res"][f]), (wl.name, f)


def test_config4_every_rank_shard_and_stats(pkg, orc_det):
    """BASELINE configs[3] = 65 536 mixed scenarios, horizon 100, sharded 8 x 8192.  No 8-GPU node here, so the eight
    rank shards — generated exactly as bench.py generates them on rank r (first = r * 8192) — are solved one after
    the other on this GPU: every shard bit for bit against the oracle, the statistics vector bench.py all-reduces
    summed over the shards equal to the one of a single 65 536-trajectory launch, whose rows equal the shards' rows."""
    import importlib
    import os
    from oracle import Scene
    st = importlib.import_module("toy-example-of-ilqr_amd.stats")
    threads = max(1, min(16, os.cpu_count() or 1))
    per, world = 8192, 8
    full = pkg.workloads.config4(B=per * world, N=100)
    eng = pkg.BatchedCILQR(full.params, full.scenes)
    whole = eng.solve_batch(full.x0, full.scenario_id, full.param_id, full.tick)
    scenes = [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs if s.obs.shape[0] else None, s.road_borders, s.ref_velo)
              for s in full.scenes]
    total = np.zeros(len(st.FIELDS))
    for r in range(world):
        wl = pkg.workloads.config4(B=per, N=100, first=r * per)
        np.testing.assert_array_equal(wl.x0, full.x0[r * per:(r + 1) * per])
        out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
        ref = orc_det.solve_batch(wl.params, scenes, wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=threads)
        eq_bits(out["u"], ref["u"], f"shard {r} u")
        eq_bits(out["x"], ref["x"], f"shard {r} x")
        assert (out["res"] == ref["res"]).all() or all(
            np.array_equal(out["res"][f], ref["res"][f]) for f in ("iters", "end_reason", "final_status", "ls_trials", "cost_evals"))
        eq_bits(out["res"]["J_final"], ref["res"]["J_final"], f"shard {r} J_final")
        eq_bits(out["x"], whole["x"][r * per:(r + 1) * per], f"shard {r} vs the single launch")
        assert (out["res"] == whole["res"][r * per:(r + 1) * per]).all()
        total += st.local_stats(out["res"], wl.N, wl.M_of)
    eng.close()
    ref_total = st.local_stats(whole["res"], full.N, full.M_of)
    names = st.FIELDS
    for i, nm in enumerate(names):
        if nm == "sum_J_final":  # a floating-point sum: associates differently over shards
            assert abs(total[i] - ref_total[i]) <= 1e-9 * abs(ref_total[i]), (nm, total[i], ref_total[i])
        else:
            assert total[i] == ref_total[i], (nm, total[i], ref_total[i])
    assert total[names.index("trajectories")] == per * world


def test_config3_and_config5_properties_at_scale(pkg, orc_det):
    """configs[2] (8192 x three_bend) and configs[4] (barrier sweep): size-independent properties on the
    full batch + oracle parity on a strided sample."""
    from oracle import Scene
    for wl in (pkg.workloads.config3(), pkg.workloads.config5(B_base=512)):
        eng = pkg.BatchedCILQR(wl.params, wl.scenes)
        out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
        res = out["res"]
        assert np.isfinite(out["x"]).all() and np.isfinite(out["u"]).all()
        assert (res["J_final"] <= res["J_init"]).all()                       # accepted steps only ever lower the cost
        assert ((res["iters"] >= 1) & (res["iters"] <= wl.params[0].max_iter)).all()
        assert (res["cost_evals"] == 1 + res["iters"] + res["ls_trials"]).all()
        assert np.array_equal(out["x"][:, 0], wl.x0)                          # x[0] is the given state
        # recomputing the cost of the returned trajectory reproduces J_final (idempotence)
        sel = np.arange(0, wl.B, max(1, wl.B // 256))
        J2 = np.empty(sel.size)
        for pid in np.unique(wl.param_id[sel]):
            m = wl.param_id[sel] == pid
            J2[m] = eng.total_cost(out["u"][sel][m], out["x"][sel][m], wl.scenario_id[sel][m], wl.param_id[sel][m], wl.tick[sel][m])
        eq_bits(J2, res["J_final"][sel], "J_final recomputed")
        # shard invariance: solving a slice alone gives the same rows
        lo, hi = wl.B // 3, wl.B // 3 + 64
        part = eng.solve_batch(wl.x0[lo:hi], wl.scenario_id[lo:hi], wl.param_id[lo:hi], wl.tick[lo:hi])
        eq_bits(part["x"], out["x"][lo:hi], "slice x")
        # oracle parity on the sample
        s0 = wl.scenes[0]
        scene = Scene(s0.lane_x, s0.lane_y, s0.lane_yaw, s0.obs, s0.road_borders, s0.ref_velo)
        ref = orc_det.solve_batch(wl.params, scene, wl.x0[sel], None, wl.param_id[sel], None, n_threads=16)
        eq_bits(out["x"][sel], ref["x"], "sample x")
        eq_bits(out["res"]["J_final"][sel], ref["res"]["J_final"], "sample J")
        eng.close()


def test_helper_wavefront_and_rollout_modes_are_transparent(pkg, orc_det, engines):
    """one wavefront per trajectory vs main + helper wavefront, and the three line-search rollout policies (all 20
    step sizes in one pass / the first trial alone first / adaptive): identical results and decision traces, and
    identical to the oracle (barrier and ALM, one and two rows per lane, both vehicle models)."""
    for name, N, over in (("three_bend", 50, dict(use_last_solution=0)), ("two_straight", 100, dict(use_last_solution=0)),
                          ("two_straight", 50, dict(use_last_solution=0)),
                          ("three_bend", 30, dict(use_last_solution=0, solve_type=1))):
        eng, p, sc = engines(name, N, **over)
        x0 = pkg.workloads.perturbed_starts(sc.ego_state, 48, 1234 + N)
        scene = oracle_scene(sc)
        refs = [orc_det.solver(p).solve(x, scene) for x in x0]
        base = None
        for helper in (0, 1):
            for rollout in (0, 1, -1):
                eng.set_helper_mode(helper)
                eng.set_rollout_mode(rollout)
                b = eng.solve_batch(x0, trace_cap=128)
                what = f"{name} N={N} helper={helper} rollout={rollout}"
                compare_solves(b, refs, what)
                if base is None:
                    base = b
                eq_bits(base["u"], b["u"], what + " u")
                eq_bits(base["x"], b["x"], what + " x")
                assert (base["res"] == b["res"]).all(), what
                for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
                    eq_bits(base["trace"][f], b["trace"][f], what + " trace." + f)
        eng.set_helper_mode(-1)
        eng.set_rollout_mode(-1)
        res = base["res"]
        assert (res["ls_trials"] > res["iters"]).any()  # some multi-trial line searches were exercised
        assert (base["trace"]["trials"] == 20).any() and (base["trace"]["trials"] == 1).any()


def test_work_sharing_between_blocks_is_transparent(pkg, orc_det):
    """Horizon 100 beyond the helper range: blocks that are done cost line-search trials of the ones still running.
    With and without it: the same outputs, counters and decision traces, bit for bit, and equal to the oracle's on a
    sample; the launch's own counters show that searches were announced and served, and no hand-over timed out."""
    from oracle import Scene
    wl = pkg.workloads.config4(B=rehearsal_size(640), N=100)
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    eng.set_group_mode(0)  # (k_solve's two-rows-per-lane builds: by default such a batch now runs two trajectories per wavefront)
    out = {}
    for mode in (1, 0):
        eng.set_work_sharing(mode)
        out[mode] = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, trace_cap=104)
        if mode == 1:
            st = eng.work_sharing_stats()
    eng.close()
    assert st["error"] == 0, st
    assert st["announced"] > 0 and st["helped"] > 0 and 0 < st["helpers"] <= 640, st
    a, b = out[1], out[0]
    eq_bits(a["u"], b["u"], "u")
    eq_bits(a["x"], b["x"], "x")
    assert (a["res"] == b["res"]).all()
    for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
        eq_bits(a["trace"][f], b["trace"][f], "trace." + f)
    sel = np.arange(0, wl.B, 8)
    scenes = [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs if s.obs.shape[0] else None, s.road_borders, s.ref_velo)
              for s in wl.scenes]
    ref = orc_det.solve_batch(wl.params, scenes, wl.x0[sel], wl.scenario_id[sel], wl.param_id[sel], None, n_threads=16)
    eq_bits(a["x"][sel], ref["x"], "sample x")
    eq_bits(a["u"][sel], ref["u"], "sample u")
    for f in ("iters", "ls_trials", "end_reason", "final_status"):
        assert np.array_equal(a["res"][f][sel], ref["res"][f]), f
    eq_bits(a["res"]["J_final"][sel], ref["res"]["J_final"], "sample J_final")


@pytest.mark.parametrize("N,B", [(64, 1600), (75, 1600), (76, 1600), (96, 600), (127, 600)])
def test_long_horizon_builds_across_horizons(pkg, orc_det, N, B):
    """The lone-wavefront builds of horizons above 63 on either side of their switches: N = 75 / 76 (cost expansion in
    LDS / in global memory), N = 96 (helper range 1536 -> 512), N = 127 (largest horizon; ring of expansion rows at
    its longest), batches just beyond the helper range so that blocks help each other from the first finisher on.
    Against the helper-wavefront build (work sharing off) on every trajectory, against the oracle on a sample; warm
    start from a shifted previous solution included (last_u)."""
    from oracle import Scene
    B = rehearsal_size(B)
    wl = pkg.workloads.config3(B=B)
    if wl.scenes[0].obs.shape[1] < N + 1:
        pytest.skip("obstacle routes shorter than the horizon")
    params = [pkg.copy_params(q, N=N, max_iter=30) for q in wl.params]
    eng = pkg.BatchedCILQR(params, wl.scenes)
    eng.set_group_mode(0)  # (the lone-wavefront builds; pairs at these horizons: test_two_trajectories_per_wavefront_at_long_horizons)
    rng = np.random.default_rng(N)
    last_u = rng.normal(0.0, 0.05, size=(B, N, 2))
    out = {}
    for mode in (1, 0):
        eng.set_work_sharing(mode)
        out[mode] = (eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick),
                     eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, last_u=last_u))
        if mode == 1:
            st = eng.work_sharing_stats()
    eng.close()
    assert st["error"] == 0 and st["helpers"] > 0, st
    for a, b, what in ((out[1][0], out[0][0], "cold"), (out[1][1], out[0][1], "warm")):
        eq_bits(a["u"], b["u"], f"N={N} {what} u")
        eq_bits(a["x"], b["x"], f"N={N} {what} x")
        assert (a["res"] == b["res"]).all(), (N, what)
    sel = np.arange(0, B, 16)
    s0 = wl.scenes[0]
    scene = Scene(s0.lane_x, s0.lane_y, s0.lane_yaw, s0.obs, s0.road_borders, s0.ref_velo)
    ref = orc_det.solve_batch(params, scene, wl.x0[sel], None, wl.param_id[sel] if wl.param_id is not None else None, None,
                              n_threads=16)
    a = out[1][0]
    eq_bits(a["x"][sel], ref["x"], f"N={N} sample x")
    eq_bits(a["u"][sel], ref["u"], f"N={N} sample u")
    eq_bits(a["res"]["J_final"][sel], ref["res"]["J_final"], f"N={N} sample J_final")
    for f in ("iters", "ls_trials", "end_reason", "final_status"):
        assert np.array_equal(a["res"][f][sel], ref["res"][f]), (N, f)


def test_block_timeline_of_persistent_blocks(pkg):
    """The development aid behind DESIGN.md's "filling the chip": every trajectory gets one record (start < end, the
    block that solved it, its XCC); a large batch is solved by at most 8 blocks per CU that each pull several
    trajectories, a small one by one block per trajectory; results do not depend on the recording."""
    wl = pkg.workloads.config3(B=6000)
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    ref = eng.solve_batch(wl.x0)
    eng.set_block_timeline(True)
    # the grouped build first (round 4: two trajectories per wavefront): a record per trajectory, the block that FINISHED it
    # (a trajectory may change wavefronts at the launch's tail), never more than two of a block's trajectories at a time
    pairs = eng.solve_batch(wl.x0)
    tlp = eng.block_timeline(wl.B)
    eq_bits(ref["x"], pairs["x"], "x with the timeline on (pairs)")
    assert (tlp[:, 1] > tlp[:, 0]).all() and (tlp[:, 0] > 0).all() and tlp[:, 2].max() < 2048
    assert len(np.unique(tlp[:, 2])) <= 2048 and np.bincount(tlp[:, 2].astype(np.int64)).max() >= 2
    eng.set_group_mode(0)  # one trajectory per wavefront: the persistent blocks of k_solve
    out = eng.solve_batch(wl.x0)
    tl = eng.block_timeline(wl.B)
    small = eng.solve_batch(wl.x0[:300])
    tl_small = eng.block_timeline(300)
    eng.close()
    eq_bits(ref["x"], out["x"], "x with the timeline on")
    eq_bits(ref["x"][:300], small["x"], "x of the small batch")
    assert (tl[:, 1] > tl[:, 0]).all() and (tl[:, 0] > 0).all()
    assert ((tl[:, 3] >= 0) & (tl[:, 3] < 8)).all() and len(set(tl[:, 3].tolist())) == 8
    blocks = np.unique(tl[:, 2])
    assert 256 <= len(blocks) <= 2048 and blocks.max() < 2048      # persistent blocks: as many as the chip holds
    per_block = np.bincount(tl[:, 2].astype(np.int64))
    assert per_block.max() >= 2                                    # ... each solving several trajectories
    # a block's solves do not overlap in time
    order = np.lexsort((tl[:, 0], tl[:, 2]))
    same = tl[order][1:, 2] == tl[order][:-1, 2]
    assert (tl[order][1:, 0][same] >= tl[order][:-1, 1][same]).all()
    assert len(np.unique(tl_small[:, 2])) == 300                   # helper range: one block per trajectory


def test_rollout_policy_statistics(pkg, engines):
    """the adaptive policy's bookkeeping (in-kernel counters): every line search starts with exactly one rollout
    pass, second passes happen only after a rejected first trial, and on the benchmark-like batch the slab is
    written in a minority of the iterations."""
    eng, p, sc = engines("two_straight", 50, dev=True, use_last_solution=0)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 256, 0xC11A0002)
    eng.set_phase_profiling(True)
    for helper in (0, 1):
        eng.set_helper_mode(helper)
        out = eng.solve_batch(x0, trace_cap=128)
        cyc = eng.phase_cycles(256)
        first, allp, second = cyc[:, 14], cyc[:, 15], cyc[:, 16]
        tr = out["trace"]
        searched = np.array([(tr["status"][b][:out["res"]["trace_len"][b]] != 2).sum() for b in range(256)])
        assert np.array_equal(first + allp, searched)
        deeper = np.array([((tr["trials"][b][:out["res"]["trace_len"][b]] > 1)).sum() for b in range(256)])
        assert (second <= deeper).all() and (second <= first).all()
        assert (allp + second).sum() < 0.3 * searched.sum(), ((allp + second).sum(), searched.sum())
    eng.set_phase_profiling(False)
    eng.set_helper_mode(-1)


# ---- edge cases ----------------------------------------------------------------------------------
def solve_both(pkg, orc_det, p, scene_tab, x0, **kw):
    from oracle import Scene
    eng = pkg.BatchedCILQR(p, scene_tab)
    out = eng.solve_batch(x0, trace_cap=128, **kw)
    scene = Scene(scene_tab.lane_x, scene_tab.lane_y, scene_tab.lane_yaw, scene_tab.obs if scene_tab.obs.shape[0] else None,
                  scene_tab.road_borders, scene_tab.ref_velo)
    refs = [orc_det.solver(p).solve(x, scene) for x in x0]
    eng.close()
    return out, refs


def test_edge_no_obstacles_short_horizon_tiny_lane(pkg, orc_det, scenarios):
    cfg, sc = scenarios["two_straight"]
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 8, 31)
    # no obstacles at all (M = 0), barrier and alm
    for st in (0, 1):
        p = pkg.params_from_config(cfg, N=30, solve_type=st)
        tab = pkg.SceneTable(sc.lane.x, sc.lane.y, sc.lane.yaw, None, sc.road_borders, sc.target_velocity)
        out, refs = solve_both(pkg, orc_det, p, tab, x0)
        compare_solves(out, refs, f"M=0 st={st}")
    # minimal horizon
    p = pkg.params_from_config(cfg, N=2)
    out, refs = solve_both(pkg, orc_det, p, pkg.SceneTable.from_scenario(sc), x0)
    compare_solves(out, refs, "N=2")
    # a lane table with 1 and with 3 samples, and one that ends inside the horizon
    for L in (1, 3, 150):
        tab = pkg.SceneTable(sc.lane.x[:L], sc.lane.y[:L], sc.lane.yaw[:L], sc.obstacles, sc.road_borders, sc.target_velocity)
        p = pkg.params_from_config(cfg, N=30)
        out, refs = solve_both(pkg, orc_det, p, tab, x0[:4])
        compare_solves(out, refs, f"L={L}")
    # max_iter = 0: the initial trajectory comes back
    p = pkg.params_from_config(cfg, N=30, max_iter=0)
    out, refs = solve_both(pkg, orc_det, p, pkg.SceneTable.from_scenario(sc), x0[:4])
    compare_solves(out, refs, "max_iter=0")
    assert (out["res"]["iters"] == 0).all() and np.array_equal(out["res"]["J_init"], out["res"]["J_final"])
    np.testing.assert_array_equal(out["u"], 0.0)


def test_edge_start_on_the_reference_line_and_wild_states(pkg, orc_det, scenarios):
    """hypot == 0 on the lane gives a NaN road-border gradient upstream (cs:527-529);
    huge / non-finite starts must not hang and must match the oracle's non-finite pattern."""
    cfg, sc = scenarios["two_straight"]
    p = pkg.params_from_config(cfg, N=30)
    tab = pkg.SceneTable.from_scenario(sc)
    j = 117
    x0 = np.array([[sc.lane.x[j], sc.lane.y[j], 0.0, 0.0],      # exactly on a sample, standing still: every row has hypot == 0
                   [1e7, 3.0, 5.0, 0.1],                          # far beyond the lane table
                   [0.0, 0.0, 1e6, 0.0],                          # absurd speed
                   [0.0, 0.5, 8.0, np.nan]])                      # NaN yaw
    out, refs = solve_both(pkg, orc_det, p, tab, x0)
    compare_solves(out, refs, "degenerate starts")
    assert np.isnan(out["res"]["J_final"][3])


@pytest.mark.parametrize("N", [63, 64, 127])
def test_horizon_boundaries(pkg, orc_det, scenarios, N):
    """N + 1 = 64 is the last horizon with one row per lane, N = 64 the first with two, N = 127 the largest
    the library accepts; both vehicle models, with and without the helper wavefront."""
    for name in ("two_straight", "three_bend"):
        cfg, sc = scenarios[name]
        if sc.obstacles.shape[1] < N + 1:
            pytest.skip("obstacle routes shorter than the horizon")
        p = pkg.params_from_config(cfg, N=N, max_iter=12)
        tab = pkg.SceneTable.from_scenario(sc)
        x0 = pkg.workloads.perturbed_starts(sc.ego_state, 6, 1000 + N)
        out, refs = solve_both(pkg, orc_det, p, tab, x0)
        compare_solves(out, refs, f"{name} N={N}")
        eng = pkg.BatchedCILQR(p, tab)
        eng.set_helper_mode(0)
        out0 = eng.solve_batch(x0, trace_cap=128)
        eng.close()
        compare_solves(out0, refs, f"{name} N={N} (no helper)")
    with pytest.raises(Exception):
        pkg.BatchedCILQR(pkg.params_from_config(cfg, N=256), tab)


@pytest.mark.not_yet_run_on_hardware
@pytest.mark.parametrize("N,B", [(128, 40), (150, 300), (200, 300), (255, 24)])
def test_horizons_above_127(pkg, orc_det, scenarios, N, B):
    """Round 6 (VERDICT r05 task 9; cs:19: upstream's N is any int): horizons of 128 ... 255 — the grouped kernel's long layout
    with FOUR rows per lane, the only family of builds at these horizons, so every batch size runs in pairs.  Both vehicle
    models, both solve types, == oracle; obstacle routes extended with their last sample where they are shorter than N + 1
    (the same arrays on both sides).  What has no build here says CILQR_ERR_UNSUPPORTED."""
    from oracle import Scene
    B = rehearsal_size(B)
    for name in ("two_straight", "three_bend"):
        cfg, sc = scenarios[name]
        obs = np.concatenate([sc.obstacles, np.repeat(sc.obstacles[:, -1:, :], 120, axis=1)], axis=1)
        tab = pkg.SceneTable(sc.lane.x, sc.lane.y, sc.lane.yaw, obs, sc.road_borders, sc.target_velocity)
        scene = Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, obs, sc.road_borders, sc.target_velocity)
        x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 2000 + N)
        for st in (0, 1):
            if st == 1 and B > 64:
                continue   # (the augmented Lagrangian on a sample: its oracle solves take the longest)
            p = pkg.params_from_config(cfg, N=N, solve_type=st, max_iter=40)
            eng = pkg.BatchedCILQR(p, tab)
            out = eng.solve_batch(x0)
            assert eng.last_launch_info()["trajectories_per_wavefront"] == 2
            ref = orc_det.solve_batch(p, scene, x0, n_threads=8)
            eq_bits(out["u"], ref["u"], f"{name} N={N} type {st} u")
            eq_bits(out["x"], ref["x"], f"{name} N={N} type {st} x")
            for f in ("iters", "end_reason", "ls_trials", "cost_evals"):
                assert (out["res"][f] == ref["res"][f]).all(), (name, N, st, f)
            eq_bits(out["res"]["J_final"], ref["res"]["J_final"], "J_final")
            with pytest.raises(pkg.CilqrError):
                eng.total_cost(out["u"][:1], out["x"][:1])
            eng.close()


@pytest.mark.not_yet_run_on_hardware
def test_closed_loop_on_the_long_layout():
    """Round 6: cilqr_closed_loop_batch_device on the grouped kernel's long layout — horizon 100 in pairs (cilqr_set_group_mode(2);
    the default keeps k_solve's loop builds) and horizon 150 (four rows per lane: the only build) — 600 egos x 6 ticks, the egos'
    states and iteration counts of every tick == the tick-by-tick loop of solve + advance on the same handle, and == stateful
    oracle solvers on a sample."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _LONG_LOOP_SCRIPT, root], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "LONG-LOOP-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


_LONG_LOOP_SCRIPT = r"""
import os, sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import cilqr_amd as pkg
from oracle import Oracle, Scene
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
RES = pkg.RESULT_DTYPE
cfg = pkg.GlobalConfig.get_instance("three_straight"); sc = pkg.build_scenario(cfg, "three_straight")
SHRINK = int(os.environ.get("CILQR_TEST_SHRINK", "1"))   # (rehearsals on the emulator: tests/emu/)
for N, mode in ((100, 2), (150, -1)):
    B, T = max(8, 600 // SHRINK), (6 if SHRINK == 1 else 3)
    p = pkg.params_from_config(cfg, N=N, use_last_solution=1, max_iter=40)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0x100B + N)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc)); eng.set_group_mode(mode)
    z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device=dev)
    # one launch
    dx0, tick = to(x0), z(B, dt=torch.int32)
    u, x, res = z(B, N, 2), z(B, N + 1, 4), torch.zeros((B, RES.itemsize), dtype=torch.uint8, device=dev)
    states, its = z(B, T, 4), z(T, B, dt=torch.int32)
    eng.closed_loop_batch_device(B, T, dx0.data_ptr(), 0, 0, tick.data_ptr(), 0, u.data_ptr(), x.data_ptr(), res.data_ptr(),
                                 states.data_ptr(), its.data_ptr(), st)
    eng.wait()
    assert eng.last_launch_info()["trajectories_per_wavefront"] == 2
    # tick by tick on the same handle
    ex0, etick = to(x0), z(B, dt=torch.int32)
    eu, ex, eres = z(B, N, 2), z(B, N + 1, 4), torch.zeros((B, RES.itemsize), dtype=torch.uint8, device=dev)
    off = RES.fields["iters"][1]
    for t in range(T):
        eng.solve_batch_device(B, ex0.data_ptr(), 0, 0, etick.data_ptr(), eu.data_ptr() if t else 0, eu.data_ptr(), ex.data_ptr(), eres.data_ptr(), 0, 0, st)
        eng.advance_batch_device(B, ex.data_ptr(), ex0.data_ptr(), etick.data_ptr(), st)
        torch.cuda.synchronize(dev)
        it_t = np.frombuffer(eres.cpu().numpy().tobytes(), dtype=RES)["iters"]
        assert np.array_equal(it_t, its[t].cpu().numpy()), (N, t)
        assert np.array_equal(ex0.cpu().numpy(), states[:, t].cpu().numpy()), (N, t)
    assert np.array_equal(ex.cpu().numpy(), x.cpu().numpy()) and np.array_equal(eu.cpu().numpy(), u.cpu().numpy())
    o = Oracle("det"); hs = states.cpu().numpy()
    for b in range(0, B, max(1, B // 8)):
        s = o.solver(p); s.reset(); xe = x0[b].copy()
        for t in range(T):
            rr = s.solve(xe, Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, t))
            xe = rr["x"][1].copy()
            assert np.array_equal(xe, hs[b, t]), (N, b, t)
    eng.close()
print("LONG-LOOP-OK")
"""


_CONCURRENT_SCRIPT = r"""
import sys, numpy as np
import torch                      # first: the process then has ONE HIP runtime (torch's), as in bench.py
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import cilqr_amd as pkg
dev = torch.device("cuda", 0)
import os
wl = pkg.workloads.config2(B=max(12, 256 // int(os.environ.get("CILQR_TEST_SHRINK", "1"))))   # (shrunk in rehearsals on the CPU emulator)
N, B = wl.N, wl.B
ref_eng = pkg.BatchedCILQR(wl.params, wl.scenes)
ref = ref_eng.solve_batch(wl.x0)
ref_eng.close()
S = 3
engs = [pkg.BatchedCILQR(wl.params, wl.scenes) for _ in range(S)]
strs = [torch.cuda.Stream(dev) for _ in range(S)]
d_x0 = torch.from_numpy(wl.x0).to(dev)
outs = [(torch.empty((B, N, 2), dtype=torch.float64, device=dev), torch.empty((B, N + 1, 4), dtype=torch.float64, device=dev),
         torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)) for _ in range(S)]
torch.cuda.synchronize(dev)
for rep in range(4):
    for i in range(S):
        u, x, r = outs[i]
        engs[i].solve_batch_device(B, d_x0.data_ptr(), 0, 0, 0, 0, u.data_ptr(), x.data_ptr(), r.data_ptr(), 0, 0,
                                   strs[i].cuda_stream)
torch.cuda.synchronize(dev)
for u, x, r in outs:
    assert np.array_equal(u.cpu().numpy(), ref["u"]) and np.array_equal(x.cpu().numpy(), ref["x"])
    res = np.frombuffer(r.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
    assert (res == ref["res"]).all()
print("CONCURRENT-OK")
"""


def test_concurrent_handles_on_separate_streams():
    """Independent batches overlap when each has its own handle and stream: the results are
    those of running them one after the other.  Runs in its own process with torch imported first — torch
    ships its own HIP runtime, and a process that loaded the system one before cannot bring up a second."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CONCURRENT_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CONCURRENT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


_DEVICE_IDS_SCRIPT = r"""
import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import cilqr_amd as pkg
dev = torch.device("cuda", 0)
wl = pkg.workloads.config4(B=64, N=30)
N, B = wl.N, wl.B
eng = pkg.BatchedCILQR(wl.params, wl.scenes)
ref = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
T_of = np.array([s.obs.shape[1] for s in wl.scenes])[wl.scenario_id]
sid, pid, tk = wl.scenario_id.copy(), wl.param_id.copy(), wl.tick.copy()
bad = np.zeros(B, bool)
sid[3] = 4; sid[7] = -1; pid[11] = 9; pid[12] = -5; tk[20] = -1; bad[[3, 7, 11, 12, 20]] = True
tk[30] = T_of[30] - N           # route one sample too short: tick + N + 1 > T
bad[30] = True
tk[31] = T_of[31] - N - 1       # exactly long enough: solved
to = lambda a: torch.from_numpy(a).to(dev)
d = [to(wl.x0), to(sid), to(pid), to(tk)]
u = torch.zeros((B, N, 2), dtype=torch.float64, device=dev)
x = torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev)
r = torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
for helper in (0, 1):
    eng.set_helper_mode(helper)
    eng.solve_batch_device(B, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, u.data_ptr(), x.data_ptr(),
                           r.data_ptr(), 0, 0, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    res = np.frombuffer(r.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
    U, X = u.cpu().numpy(), x.cpu().numpy()
    assert (res["end_reason"][bad] == 3).all() and (res["iters"][bad] == 0).all(), res[bad]
    assert np.isnan(U[bad]).all() and np.isnan(X[bad]).all() and np.isnan(res["J_final"][bad]).all()
    ok = ~bad
    ok[31] = False
    assert np.array_equal(U[ok], ref["u"][ok]) and np.array_equal(X[ok], ref["x"][ok]) and (res[ok] == ref["res"][ok]).all()
    assert res["end_reason"][31] != 3 and np.isfinite(X[31]).all()
# a negative trace capacity is refused
try:
    eng.solve_batch_device(B, d[0].data_ptr(), 0, 0, 0, 0, u.data_ptr(), x.data_ptr(), r.data_ptr(), 0, -1, 0)
    raise SystemExit("negative trace_cap accepted")
except pkg.CilqrError as e:
    assert e.code == -1
print("DEVICE-IDS-OK")
"""


_DEVICE_LOOP_SCRIPT = r"""
import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import cilqr_amd as pkg
from oracle import Oracle, Scene
dev = torch.device("cuda", 0)
cfg = pkg.GlobalConfig.get_instance("three_straight")
sc = pkg.build_scenario(cfg, "three_straight")
p = pkg.params_from_config(cfg, N=30)
assert p.use_last_solution == 1
B, ticks = 48, 25
x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 424242)
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
N = p.N
d_x0 = torch.from_numpy(x0).to(dev)
d_tick = torch.zeros(B, dtype=torch.int32, device=dev)
d_u = torch.zeros((B, N, 2), dtype=torch.float64, device=dev)
d_x = torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev)
d_res = torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
hist = []
for t in range(ticks):  # no host round trip inside: solve -> advance -> solve ...
    eng.solve_batch_device(B, d_x0.data_ptr(), 0, 0, d_tick.data_ptr(), d_u.data_ptr() if t else 0, d_u.data_ptr(), d_x.data_ptr(),
                           d_res.data_ptr(), 0, 0, st)
    hist.append((d_u.clone(), d_x.clone(), d_res.clone()))   # device-side copies, stream-ordered
    eng.advance_batch_device(B, d_x.data_ptr(), d_x0.data_ptr(), d_tick.data_ptr(), st)
torch.cuda.synchronize(dev)
assert d_tick.cpu().numpy().tolist() == [ticks] * B
orc = Oracle("det")
solvers = [orc.solver(p) for _ in range(B)]
for s_ in solvers:
    s_.reset()
xs = x0.copy()
iters = 0
for t in range(ticks):
    scene = Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, t)
    U, X = hist[t][0].cpu().numpy(), hist[t][1].cpu().numpy()
    res = np.frombuffer(hist[t][2].cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
    for b in range(B):
        r = solvers[b].solve(xs[b], scene)
        assert np.array_equal(U[b], r["u"]) and np.array_equal(X[b], r["x"]), (t, b)
        assert res["iters"][b] == r["res"]["iters"] and res["J_final"][b] == r["res"]["J_final"], (t, b)
        xs[b] = r["x"][1]
    iters += int(res["iters"].sum())
assert iters > ticks * B
print("DEVICE-LOOP-OK", iters)
"""


def test_closed_loop_resident_on_the_device():
    """48 egos x 25 ticks of the reference's planning loop (mp:180-197) without a host round trip per tick:
    cilqr_solve_batch_device (warm-started from its own previous u buffer) + cilqr_advance_batch_device; every tick
    of every ego equals its own stateful oracle solver."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DEVICE_LOOP_SCRIPT, root], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DEVICE-LOOP-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_device_pointer_entry_checks_its_ids_on_the_device():
    """cilqr_solve_batch_device cannot validate scenario_id / param_id / tick on the host: trajectories with an id
    outside the tables, a negative tick or an obstacle route that ends before tick + N + 1 (upstream:
    std::out_of_range, ut:52-58) come back unsolved — NaN, end_reason BAD_INPUT — and do not disturb the others."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DEVICE_IDS_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DEVICE-IDS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_alm_state_follows_the_parameter_table(pkg, orc_det, scenarios):
    """The multiplier arrays are [B][N][8 + 2M]: replacing the parameter table of a live ALM handle by one with a
    longer horizon re-lays them out (the old arrays would have been indexed past their end); growing
    the batch keeps the rows that exist; a warm-started call continues from its multipliers."""
    cfg, sc = scenarios["three_bend"]
    tab = pkg.SceneTable.from_scenario(sc)
    scene = oracle_scene(sc)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 24, 777)
    p30 = pkg.params_from_config(cfg, N=30, solve_type=1, use_last_solution=1)
    eng = pkg.BatchedCILQR(p30, tab)
    out = eng.solve_batch(x0[:8], trace_cap=128)
    compare_solves(out, [orc_det.solver(p30).solve(x, scene) for x in x0[:8]], "alm N=30")
    # longer horizon, same handle, bigger batch than the slack of the old allocation would have covered
    p60 = pkg.params_from_config(cfg, N=60, solve_type=1, use_last_solution=1)
    eng.set_params(p60)
    out = eng.solve_batch(x0, trace_cap=128)
    compare_solves(out, [orc_det.solver(p60).solve(x, scene) for x in x0], "alm N=60 after N=30")
    mu, mun, rho = eng.get_alm_state(24)
    assert mu.shape == (24, 60, 8 + 2 * tab.obs.shape[0])
    eng.close()
    # growing the arrays keeps the rows that exist: a warm-started tick after the growth continues from them
    eng = pkg.BatchedCILQR(p30, tab)
    solvers = [orc_det.solver(p30) for _ in range(8)]
    for s_ in solvers:
        s_.reset()
    t0 = eng.solve_batch(x0[:8], trace_cap=128)
    compare_solves(t0, [solvers[b].solve(x0[b], scene) for b in range(8)], "alm tick 0")
    mu8, _, rho8 = eng.get_alm_state(8)
    from ctypes import c_void_p
    pkg._lib.check(eng._lib.cilqr_set_alm_state(eng._h, 200, None, None), "grow")  # 200 rows: beyond any slack
    mu200, _, rho200 = eng.get_alm_state(200)
    eq_bits(mu200[:8], mu8, "multipliers kept while growing")
    eq_bits(rho200[:8], rho8, "rho kept while growing")
    assert (mu200[8:] == 0).all() and (rho200[8:] == p30.alm_rho_init).all()
    x1 = t0["x"][:, 1].copy()
    t1 = eng.solve_batch(x1, tick=np.ones(8, np.int32), last_u=t0["u"], trace_cap=128)
    compare_solves(t1, [solvers[b].solve(x1[b], oracle_scene(sc, 1)) for b in range(8)], "alm tick 1 after growth")
    eng.close()


def test_single_ego_entry_keeps_tables_resident(pkg, orc_det, scenarios):
    """cilqr_solve (the drop-in solve() of one ego): a closed loop whose obstacle predictions are the tail of the
    routes from the current tick on re-uses the tables in HBM (one upload), any other change uploads again, and
    every tick equals the oracle's stateful solver."""
    cfg, sc = scenarios["three_straight"]
    solver = pkg.CILQRSolver(cfg, N=30)
    assert solver.params.use_last_solution == 1
    ref = orc_det.solver(solver.params)
    ref.reset()
    x0 = sc.ego_state.copy()
    obs = sc.obstacles
    ticks = 12
    for t in range(ticks):
        preds = [pkg.RoutingLine(r[t:, 0], r[t:, 1], r[t:, 2]) for r in obs]  # utils::get_sub_routing_lines
        u, x = solver.solve(x0, sc.lane, sc.target_velocity, preds, sc.road_borders)
        r = ref.solve(x0, oracle_scene(sc, t))
        eq_bits(u, r["u"], f"tick {t} u")
        eq_bits(x, r["x"], f"tick {t} x")
        x0 = x[1].copy()
    up, re = solver._engine.solve_cache_stats()
    assert (up, re) == (1, ticks - 1), (up, re)
    # a prediction window of fixed length (N + 1 rows from the current tick) is recognised too
    t = ticks
    u, x = solver.solve(x0, sc.lane, sc.target_velocity, obs[:, t:t + 31], sc.road_borders)
    r = ref.solve(x0, oracle_scene(sc, t))
    eq_bits(x, r["x"], "fixed window x")
    assert solver._engine.solve_cache_stats() == (1, ticks)
    # changed borders: uploaded again, still the oracle's result
    x0 = x[1].copy()
    borders = sc.road_borders + np.array([0.25, 0.0])
    u, x = solver.solve(x0, sc.lane, sc.target_velocity, obs[:, t + 1:], borders)
    sc2 = oracle_scene(sc, t + 1)
    sc2.road_borders = np.ascontiguousarray(borders)
    eq_bits(x, ref.solve(x0, sc2)["x"], "new borders x")
    assert solver._engine.solve_cache_stats() == (2, ticks)
    # too short a prediction is refused like upstream's out_of_range
    with pytest.raises(pkg.CilqrError) as e:
        solver.solve(x0, sc.lane, sc.target_velocity, obs[:, :20], sc.road_borders)
    assert e.value.code == -2


def test_irregular_lane_tables_reference_search(pkg, orc_det, scenarios):
    """The reference-point proof leans on a per-lane convexity certificate; on lane tables that cannot be
    certified (a sharp corner, jittered or very uneven sampling, a lane that doubles back, a gap) it must
    fall back to sampling / the serial chain and still reproduce the oracle bit for bit."""
    cfg, sc = scenarios["two_straight"]
    p = pkg.params_from_config(cfg, N=40)
    rng = np.random.default_rng(77)
    L = len(sc.lane.x)
    s = np.arange(L) * 0.1
    lanes = {}
    # sharp 60 degree corner 12 m ahead of the ego
    k = int(np.searchsorted(sc.lane.x, sc.ego_state[0] + 12.0))
    x, y = sc.lane.x.copy(), sc.lane.y.copy()
    t = s[k:] - s[k]
    x[k:] = x[k] + t * np.cos(np.pi / 3); y[k:] = y[k] + t * np.sin(np.pi / 3)
    yaw = sc.lane.yaw.copy(); yaw[k:] = np.pi / 3
    lanes["corner"] = (x, y, yaw)
    # centimetre jitter on every sample: second differences of the table are as large as the segments
    lanes["jitter"] = (sc.lane.x + rng.normal(0, 0.03, L), sc.lane.y + rng.normal(0, 0.03, L), sc.lane.yaw)
    # uneven sampling: segment lengths between 1 cm and 60 cm
    ds = rng.uniform(0.01, 0.6, L)
    lanes["uneven"] = (sc.lane.x[0] + np.cumsum(ds) - ds[0], sc.lane.y.copy(), sc.lane.yaw)
    # a hairpin: the lane comes back 3 m to the side of itself
    xa = sc.lane.x[:k + 200]
    xb = xa[::-1]
    lanes["hairpin"] = (np.concatenate([xa, xb]), np.concatenate([sc.lane.y[:k + 200], sc.lane.y[:k + 200] + 3.0]),
                        np.concatenate([sc.lane.yaw[:k + 200], sc.lane.yaw[:k + 200] + np.pi]))
    # a 5 m gap in the table
    keep = np.r_[0:k, k + 50:L]
    lanes["gap"] = (sc.lane.x[keep], sc.lane.y[keep], sc.lane.yaw[keep])
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, 12, 909)
    sampled = serial = 0
    for name, (lx, ly, lyaw) in lanes.items():
        tab = pkg.SceneTable(lx, ly, lyaw, sc.obstacles, sc.road_borders, sc.target_velocity)
        out, refs = solve_both(pkg, orc_det, p, tab, x0)
        compare_solves(out, refs, f"lane={name}")
        # the same through the instrumented kernel: which proof levels did this table need?
        eng = pkg.BatchedCILQR(p, tab, dev=True)
        eng.set_phase_profiling(True)
        out2 = eng.solve_batch(x0, trace_cap=128)
        cyc = eng.phase_cycles(len(x0))
        eng.close()
        compare_solves(out2, refs, f"lane={name} (instrumented)")
        sampled += int(cyc[:, 13].sum())
        serial += int(cyc[:, 8].sum())
    assert sampled > 0, "no table needed the sample-by-sample proof: the test does not exercise it"
    assert serial > 0, "no table needed the serial chain: the test does not exercise it"


def test_fuzz_random_parameter_sets(pkg, orc_det, scenarios):
    """randomised parameter sets (weights, barrier shapes, bounds, lambda schedule, horizon, vehicle
    model, solve type) on all four scenarios: whole solves incl. decision traces stay bit-exact."""
    from oracle import Scene
    import os
    # CILQR_FUZZ_SEED / CILQR_FUZZ_TRIALS widen the search for soak runs; the defaults are what the suite runs
    rng = np.random.default_rng(int(os.environ.get("CILQR_FUZZ_SEED", "20250829")))
    names = list(scenarios)
    total_iters = 0
    for trial in range(int(os.environ.get("CILQR_FUZZ_TRIALS", "28"))):
        name = names[trial % 4]
        cfg, sc = scenarios[name]
        N = int(rng.choice([3, 7, 20, 30, 45, 63, 64, 80, 110]))
        T = sc.routes.shape[1]
        over = dict(
            N=N, use_last_solution=0, solve_type=int(rng.random() < 0.3), reference_point=int(rng.random() < 0.5),
            w_pos=float(rng.uniform(0.2, 3)), w_vel=float(rng.uniform(0.2, 3)), w_yaw=float(rng.uniform(1, 40)),
            w_acc=float(rng.uniform(0.1, 2)), w_stl=float(rng.uniform(5, 60)),
            obstacle_exp_q1=float(rng.uniform(1, 12)), obstacle_exp_q2=float(rng.uniform(2, 9)),
            state_exp_q1=float(rng.uniform(1, 6)), state_exp_q2=float(rng.uniform(2, 6)),
            init_lamb=float(rng.choice([0.0, 0.0, 1.0, 20.0])), lamb_decay=float(rng.uniform(0.3, 0.9)),
            lamb_amplify=float(rng.uniform(1.5, 4)), max_lamb=float(rng.choice([100.0, 1000.0, 1e4])),
            convergence_threshold=float(rng.choice([1e-3, 1e-2, 0.1])), accept_step_threshold=float(rng.uniform(0.1, 0.8)),
            max_iter=int(rng.choice([3, 25, 100])), velo_max=float(rng.uniform(8, 16)), acc_max=float(rng.uniform(1.5, 4)),
            acc_min=-float(rng.uniform(1.5, 4)), stl_lim=float(rng.uniform(0.08, 0.4)), d_safe=float(rng.uniform(0.5, 1.2)),
            alm_rho_init=float(rng.uniform(1, 30)), alm_gamma=float(rng.choice([0.0, 0.5])), max_rho=float(rng.uniform(20, 80)),
            max_mu=float(rng.uniform(50, 200)))
        p = pkg.params_from_config(cfg, **over)
        tick = int(rng.integers(0, max(1, T - N - 1)))
        B = 10
        x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 7000 + trial)
        tab = pkg.SceneTable.from_scenario(sc)
        eng = pkg.BatchedCILQR(p, tab)
        eng.set_helper_mode(int(trial % 3) - 1)  # -1 auto, 0 off, 1 on
        out = eng.solve_batch(x0, tick=np.full(B, tick, np.int32), trace_cap=128)
        scene = Scene(tab.lane_x, tab.lane_y, tab.lane_yaw, tab.obs, tab.road_borders, tab.ref_velo, tick)
        refs = [orc_det.solver(p).solve(x, scene) for x in x0]
        compare_solves(out, refs, f"fuzz {trial} {name} N={N} st={over['solve_type']} rp={over['reference_point']}")
        total_iters += int(out["res"]["iters"].sum())
        eng.close()
    assert total_iters > 1000


# ---- round 3 -------------------------------------------------------------------------------------
def test_production_library_refuses_the_testing_aids(pkg, scenarios):
    """libcilqr_amd.so carries neither the DBG nor the PROF builds of the solve kernel (they live in
    libcilqr_amd_dev.so): asking for them is an error, not a silent no-op."""
    cfg, sc = scenarios["three_bend"]
    eng = pkg.BatchedCILQR(pkg.params_from_config(cfg, N=30), pkg.SceneTable.from_scenario(sc))
    for call in (lambda: eng.set_debug_flags(pkg._lib.DBG_SERIAL_REF_SCAN), lambda: eng.set_phase_profiling(True)):
        with pytest.raises(pkg.CilqrError) as e:
            call()
        assert e.value.code == pkg._lib.ERR_UNSUPPORTED
    eng.set_debug_flags(0)
    eng.set_phase_profiling(False)
    eng.close()


def test_rows_that_stay_behind_the_chain_are_proven_in_parallel(pkg, orc_det):
    """BASELINE configs[3] has solves (rank-0 shard: trajectories 1402 and 5317) whose trial trajectories slow down
    and swerve so that for a score of consecutive rows the nearest lane sample lies BEHIND the index an earlier row has
    reached: the chain of cs:289-314 stays put there.  The lane-parallel search takes the running maximum of its
    candidates and proves the stay with one comparison per row; before, each of these solves' ~300 trial costs went
    down the serial chain (45 of the launch's 52 ms).  Same bits as the oracle; the serial chain is now the exception."""
    wl = pkg.workloads.config4(B=8192)
    rows = np.array([1402, 5317, 121, 4607])
    sub = pkg.workloads.Workload("sub", wl.params, wl.scenes, wl.x0[rows], wl.scenario_id[rows], wl.param_id[rows], wl.tick[rows])
    eng = pkg.BatchedCILQR(sub.params, sub.scenes, dev=True)
    eng.set_phase_profiling(True)
    out = eng.solve_batch(sub.x0, sub.scenario_id, sub.param_id, sub.tick, trace_cap=128)
    cyc = eng.phase_cycles(len(rows))
    eng.close()
    scenes = [oracle_scene_tab(t) for t in sub.scenes]
    refs = []
    for i in range(len(rows)):
        s = orc_det.solver(sub.params[sub.param_id[i]])
        refs.append(s.solve(sub.x0[i], scenes[sub.scenario_id[i]], trace_cap=128))
    compare_solves(out, refs, "config 4 outliers (instrumented)")
    trials, fallbacks = cyc[:, 9], cyc[:, 8]
    assert trials[0] >= 300 and trials[1] >= 290
    assert (fallbacks[:2] * 8 <= trials[:2]).all(), (fallbacks.tolist(), trials.tolist())
    # and through the production library (persistent blocks need a big batch: the whole shard is covered by
    # test_config4_every_rank_shard_and_stats; here the four as a small batch)
    eng = pkg.BatchedCILQR(sub.params, sub.scenes)
    out2 = eng.solve_batch(sub.x0, sub.scenario_id, sub.param_id, sub.tick, trace_cap=128)
    eng.close()
    compare_solves(out2, refs, "config 4 outliers")


def oracle_scene_tab(tab, tick=0):
    from oracle import Scene
    return Scene(tab.lane_x, tab.lane_y, tab.lane_yaw, tab.obs, tab.road_borders, tab.ref_velo, tick)


_ONE_HANDLE_TWO_STREAMS_SCRIPT = r"""
import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import cilqr_amd as pkg
dev = torch.device("cuda", 0)
# large batches: persistent blocks, whose trajectory counter and scratch areas belong to the launch in flight
import os
B2 = max(24, 4096 // int(os.environ.get("CILQR_TEST_SHRINK", "1")))   # (shrunk in rehearsals on the CPU emulator)
wa, wb = pkg.workloads.config3(B=B2), pkg.workloads.config3(B=B2, first=4096)
N = wa.N
eng = pkg.BatchedCILQR(wa.params, wa.scenes)
refs = [eng.solve_batch(w.x0) for w in (wa, wb)]
strs = [torch.cuda.Stream(dev) for _ in range(2)]
d_x0 = [torch.from_numpy(w.x0).to(dev) for w in (wa, wb)]
outs = [(torch.empty((B2, N, 2), dtype=torch.float64, device=dev), torch.empty((B2, N + 1, 4), dtype=torch.float64, device=dev),
         torch.zeros((B2, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)) for _ in range(2)]
torch.cuda.synchronize(dev)
for rep in range(3):   # the SAME handle, alternating streams, no host synchronisation in between
    for i in range(2):
        u, x, r = outs[i]
        eng.solve_batch_device(B2, d_x0[i].data_ptr(), 0, 0, 0, 0, u.data_ptr(), x.data_ptr(), r.data_ptr(), 0, 0, strs[i].cuda_stream)
torch.cuda.synchronize(dev)
for (u, x, r), ref in zip(outs, refs):
    assert np.array_equal(u.cpu().numpy(), ref["u"]) and np.array_equal(x.cpu().numpy(), ref["x"])
    res = np.frombuffer(r.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
    assert (res == ref["res"]).all()
print("ONE-HANDLE-TWO-STREAMS-OK")
"""


def test_one_handle_on_two_streams_is_serialised():
    """the control words, scratch areas and work-sharing state of a handle belong to its launch in flight.
    A launch on another stream than the previous one now waits for it on the device (an event), so alternating
    streams on one handle gives the right results instead of corrupting the trajectory counter."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ONE_HANDLE_TWO_STREAMS_SCRIPT, root], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ONE-HANDLE-TWO-STREAMS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_single_ego_cache_hit_still_checks_the_obstacle_horizon(pkg, scenarios):
    """routes shorter than N + 1 that happen to match a prefix of the routes an earlier call uploaded were
    solved from the cached samples; upstream throws std::out_of_range (ut:52-58) whatever was uploaded before."""
    cfg, sc = scenarios["three_straight"]
    solver = pkg.CILQRSolver(cfg, N=30)
    x0 = sc.ego_state.copy()
    solver.solve(x0, sc.lane, sc.target_velocity, sc.obstacles, sc.road_borders)   # uploads the full routes
    for short in (sc.obstacles[:, :20], sc.obstacles[:, 5:25]):                     # a prefix / a sub-range of them
        with pytest.raises(pkg.CilqrError) as e:
            solver.solve(x0, sc.lane, sc.target_velocity, short, sc.road_borders)
        assert e.value.code == pkg._lib.ERR_OBSTACLE_HORIZON
    u, x = solver.solve(x0, sc.lane, sc.target_velocity, sc.obstacles[:, :31], sc.road_borders)  # exactly N + 1: fine
    assert np.isfinite(x).all()


def test_resumable_solves_are_transparent(pkg, orc_det):
    """Long horizons in batches larger than the chip's 2048 resident blocks run RESUMABLE solves: a slice of iterations,
    the state parked in HBM (x, u, lane indices, the scalars cs:110-141 carries), the solve queued and picked up again by
    whichever block is free — fresh trajectories first.  Whatever the slice length, with and without the work sharing
    between blocks, every output — decision traces included — is the one of the unsliced solve and of the oracle."""
    wl = pkg.workloads.config4(B=rehearsal_size(4096))
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    eng.set_group_mode(0)  # (k_solve's builds; the grouped build's sliced solves: test_sliced_solves_of_the_grouped_build)
    ids = (wl.scenario_id, wl.param_id, wl.tick)
    eng.set_resume_iters(0)
    whole = eng.solve_batch(wl.x0, *ids, trace_cap=128)
    assert eng.resume_stats() == 0
    for iters, share in ((8, 1), (32, 1), (5, 0), (100000, 1)):
        eng.set_resume_iters(iters)
        eng.set_work_sharing(share)
        # (without work sharing a batch of this size and horizon would get helper wavefronts, which do not slice their
        #  solves: lone wavefronts asked for — round 3 passed this case on the stale counter of the launch before)
        eng.set_helper_mode(-1 if share else 0)
        out = eng.solve_batch(wl.x0, *ids, trace_cap=128)
        parked = eng.resume_stats()
        for k in ("u", "x"):
            eq_bits(whole[k], out[k], f"{k} with {iters} iterations per slice, sharing {share}")
        assert (whole["res"] == out["res"]).all() and (whole["trace"] == out["trace"]).all(), (iters, share)
        if iters <= 32:
            assert parked > 4096 // 4, (iters, parked)   # most solves of this mix outlast a slice
        else:
            assert parked == 0
    eng.close()
    scenes = [oracle_scene_tab(t) for t in wl.scenes]
    rows = np.r_[0:24, 1402, 4081]
    refs = [orc_det.solver(wl.params[wl.param_id[b]]).solve(wl.x0[b], scenes[wl.scenario_id[b]], trace_cap=128) for b in rows]
    sub = {k: whole[k][rows] for k in ("u", "x", "res", "trace")}
    compare_solves(sub, refs, "resumable (config 4, first rows)")
    # a batch that fits the chip at once has nothing to reorder: its solves run whole
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    eng.set_group_mode(0)
    eng.set_resume_iters(8)
    small = eng.solve_batch(wl.x0[:1800], wl.scenario_id[:1800], wl.param_id[:1800], wl.tick[:1800])
    assert eng.resume_stats() == 0
    eq_bits(small["x"], whole["x"][:1800], "small batch")
    eng.close()


_FUSED_LOOP_SCRIPT = r"""
import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import cilqr_amd as pkg
from oracle import Oracle, Scene
dev = torch.device("cuda", 0)
cfg = pkg.GlobalConfig.get_instance("three_straight")
sc = pkg.build_scenario(cfg, "three_straight")
import os
SHR = int(os.environ.get("CILQR_TEST_SHRINK", "1"))   # (rehearsals on the CPU emulator divide the batch sizes and cut the ticks)
for N, B, ticks, alm in ((30, 300, 12, 0), (30, 3000, 12, 0), (50, 2600, 9, 0), (30, 200, 6, 1), (30, 2500, 5, 1), (70, 2200, 4, 0), (30, 64, 1, 0)):
    if SHR > 1:
        B, ticks = max(6, B // SHR), min(ticks, 3)
    p = pkg.params_from_config(cfg, N=N, use_last_solution=1, solve_type=alm)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 515151 + N)
    st = torch.cuda.current_stream(dev).cuda_stream
    def buffers():
        return (torch.from_numpy(x0).to(dev), torch.zeros(B, dtype=torch.int32, device=dev),
                torch.zeros((B, N, 2), dtype=torch.float64, device=dev), torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev),
                torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev))
    # tick by tick: one launch + one advance per tick
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
    d_x0, d_tick, d_u, d_x, d_res = buffers()
    states, iters = [], []
    for t in range(ticks):
        eng.solve_batch_device(B, d_x0.data_ptr(), 0, 0, d_tick.data_ptr(), d_u.data_ptr() if t else 0, d_u.data_ptr(), d_x.data_ptr(),
                               d_res.data_ptr(), 0, 0, st)
        eng.advance_batch_device(B, d_x.data_ptr(), d_x0.data_ptr(), d_tick.data_ptr(), st)
        torch.cuda.synchronize(dev)
        states.append(d_x0.cpu().numpy().copy())
        iters.append(np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)["iters"].copy())
    ref = (d_x0.cpu().numpy(), d_tick.cpu().numpy(), d_u.cpu().numpy(), d_x.cpu().numpy(), d_res.cpu().numpy())
    eng.close()
    # the same loop in one launch (a fresh handle: the ALM multipliers start from zero again)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
    f_x0, f_tick, f_u, f_x, f_res = buffers()
    f_states = torch.zeros((B, ticks, 4), dtype=torch.float64, device=dev)
    f_iters = torch.zeros((ticks, B), dtype=torch.int32, device=dev)
    eng.closed_loop_batch_device(B, ticks, f_x0.data_ptr(), 0, 0, f_tick.data_ptr(), 0, f_u.data_ptr(), f_x.data_ptr(), f_res.data_ptr(),
                                 f_states.data_ptr(), f_iters.data_ptr(), st)
    torch.cuda.synchronize(dev)
    got = (f_x0.cpu().numpy(), f_tick.cpu().numpy(), f_u.cpu().numpy(), f_x.cpu().numpy(), f_res.cpu().numpy())
    for a_, b_, nm in zip(ref, got, ("x0", "tick", "u", "x", "res")):
        assert np.array_equal(a_.view(np.uint8), b_.view(np.uint8)), (N, B, nm)
    assert np.array_equal(np.stack(states, 1).view(np.uint64), f_states.cpu().numpy().view(np.uint64)), (N, B, "states")
    assert np.array_equal(np.stack(iters, 0), f_iters.cpu().numpy()), (N, B, "iters")
    assert got[1].tolist() == [ticks] * B
    eng.close()
    # ... and both equal a stateful oracle solver, ego by ego (a few egos)
    orc = Oracle("det")
    fs = f_states.cpu().numpy()
    for b in range(4):
        s_ = orc.solver(p); s_.reset()
        xs = x0[b].copy()
        for t in range(ticks):
            r = s_.solve(xs, Scene(sc.lane.x, sc.lane.y, sc.lane.yaw, sc.obstacles, sc.road_borders, sc.target_velocity, t))
            xs = r["x"][1].copy()
            assert np.array_equal(xs, fs[b, t]), (N, B, b, t)
# an ego whose routes run out stops there: T = 200 samples, N = 30: ticks 0 .. 169 are solvable
p = pkg.params_from_config(cfg, N=30, use_last_solution=1)
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
B = 8
x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 7)
d_x0 = torch.from_numpy(x0).to(dev)
d_tick = torch.tensor([0, 160, 165, 168, 169, 170, 100, 0], dtype=torch.int32, device=dev)
d_u = torch.zeros((B, 30, 2), dtype=torch.float64, device=dev); d_x = torch.zeros((B, 31, 4), dtype=torch.float64, device=dev)
d_res = torch.zeros((B, pkg.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
eng.closed_loop_batch_device(B, 5, d_x0.data_ptr(), 0, 0, d_tick.data_ptr(), 0, d_u.data_ptr(), d_x.data_ptr(), d_res.data_ptr(), 0, 0,
                             torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize(dev)
res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=pkg.RESULT_DTYPE)
assert d_tick.cpu().numpy().tolist() == [5, 165, 170, 170, 170, 170, 105, 5], d_tick.cpu().numpy().tolist()
# (ego 2 starts at tick 165 and solves 165 .. 169: five good ticks; egos 3, 4 run into tick 170, ego 5 starts there)
assert (res["end_reason"][[3, 4, 5]] == 3).all() and (res["end_reason"][[0, 1, 2, 6, 7]] != 3).all()
print("FUSED-LOOP-OK")
"""


def test_closed_loop_in_one_launch_equals_the_tick_by_tick_loop():
    """cilqr_closed_loop_batch_device: every ego's ticks back to back on one block (solve, ego <- x.row(1), tick + 1, warm
    start from the plan just made: mp:180-197, cs:163-180) give, ego by ego, the numbers of the tick-by-tick loop of
    cilqr_solve_batch_device + cilqr_advance_batch_device — final states, ticks, last plans, results, the state after every
    tick and every tick's iteration count — for one block per ego (small batches, helper wavefronts), for persistent
    blocks, for the augmented Lagrangian; and of stateful oracle solvers.  An ego whose obstacle routes run out stops."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FUSED_LOOP_SCRIPT, root], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FUSED-LOOP-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_closed_loop_in_one_launch_follows_use_last_solution():
    """Found by scripts/soak_closed_loop.py: later ticks of the fused loop start warm only for egos whose parameter set has
    use_last_solution (cs:95-101); the others start cold every tick (fresh multipliers under "alm", cs:88-93), as the
    tick-by-tick loop with d_last_u = NULL and the oracle's stateful solver do.  A short seeded run of the soak script:
    random shapes, egos of several scenarios mixed in one launch, each from its own tick, warm start on and off."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_closed_loop.py"), "--cases", "24", "--seed", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SOAK-CLOSED-LOOP OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    cold = [l for l in r.stdout.splitlines() if l.endswith("ok") and l.split()[5] == "0"]
    assert len(cold) >= 3, "the seeded run no longer holds cold-start cases"



def test_single_ego_cache_under_random_scene_changes(pkg, orc_det, scenarios):
    """cilqr_solve keeps the tables of the previous call in HBM and re-uses them when the new arguments are the same or
    the obstacle predictions are a later window of the routes uploaded before.  A random walk over what a caller may do
    between two ticks — window one tick on, same window again, a jump back or ahead, routes cut to N + 1 samples, an
    obstacle moved, another lane, other borders or target speed, the other scenario — must give the oracle's numbers on
    every call, whatever the cache did."""
    from oracle import Scene
    rng = np.random.default_rng(31337)
    N = 30
    hits = 0
    for name in ("three_straight", "two_borrow"):
        cfg, sc = scenarios[name]
        other = scenarios["three_bend" if name == "three_straight" else "two_straight"][1]
        p = pkg.params_from_config(cfg, N=N, use_last_solution=0)
        eng = pkg.BatchedCILQR(p, None)
        T = sc.routes.shape[1]
        lane, obs_all, borders, velo, tick = sc.lane, sc.obstacles.copy(), np.array(sc.road_borders, float), float(sc.target_velocity), 0
        x0 = sc.ego_state.copy()
        for step in range(70):
            op = int(rng.integers(0, 10))
            if op <= 3:
                tick = min(tick + 1, T - N - 1)                       # the usual case: the window one tick on
            elif op == 4:
                pass                                                  # the very same arguments again
            elif op == 5:
                tick = int(rng.integers(0, T - N - 1))                # a jump
            elif op == 6:
                obs_all = obs_all.copy(); obs_all[int(rng.integers(0, len(obs_all))), :, 1] += 0.25   # an obstacle moved
            elif op == 7:
                lane = other.lane if lane is sc.lane else sc.lane     # another reference line
            elif op == 8:
                borders = borders + np.array([0.1, -0.1]); velo += 0.5
            cut = rng.random() < 0.3                                  # predictions cut to exactly N + 1 samples
            obs = obs_all[:, tick:tick + N + 1] if cut else obs_all[:, tick:]
            tab = pkg.SceneTable(lane.x, lane.y, lane.yaw, np.ascontiguousarray(obs), borders, velo)
            x = x0 + np.array([0.3 * rng.standard_normal(), 0.2 * rng.standard_normal(), 0.3 * rng.standard_normal(), 0.0])
            u, xx, res = eng.solve_one(x, tab)
            ref = orc_det.solver(p).solve(x, Scene(tab.lane_x, tab.lane_y, tab.lane_yaw, tab.obs, tab.road_borders, tab.ref_velo, 0))
            eq_bits(ref["u"], u, f"{name} step {step} op {op} cut {cut}: u")
            eq_bits(ref["x"], xx, f"{name} step {step} op {op} cut {cut}: x")
            assert int(res["iters"]) == int(ref["res"]["iters"]), (name, step, op)
        up, re = eng.solve_cache_stats()
        assert up + re == 70 and re >= 20 and up >= 10, (up, re)   # both paths taken
        hits += re
        eng.close()
    assert hits > 0


def test_trajectories_in_pairs_per_wavefront_are_transparent(pkg, orc_det, engines, scenarios):
    """Round 4: the grouped build (k_solve_grp, cilqr_set_group_mode) — two trajectories per wavefront whose line-search
    rollouts share one pass — against the one-trajectory-per-wavefront build and the oracle: trajectories, costs, counters
    and the whole decision trace, for every rollout policy, both vehicle models, warm starts with tick offsets, odd batch
    sizes (a wavefront left with one trajectory), mixed parameter sets and scenarios inside one launch, a zero iteration
    budget, and ids the device refuses."""
    for name, N, B in (("three_bend", 50, 97), ("two_straight", 50, 64), ("three_straight", 30, 33), ("two_borrow", 63, 21)):
        eng, p, sc = engines(name, N, use_last_solution=0)
        x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 4242 + N)
        scene = oracle_scene(sc)
        refs = [orc_det.solver(p).solve(x, scene) for x in x0]
        eng.set_helper_mode(0)
        eng.set_group_mode(0)
        base = eng.solve_batch(x0, trace_cap=128)
        compare_solves(base, refs, f"{name} N={N} one per wavefront")
        for gm, rollout in ((2, -1), (2, 0), (2, 1)):  # two trajectories per wavefront (three: measured slower in round 4, no longer built)
            eng.set_group_mode(gm)
            eng.set_rollout_mode(rollout)
            g = eng.solve_batch(x0, trace_cap=128)
            what = f"{name} N={N} {gm} per wavefront, rollout={rollout}"
            compare_solves(g, refs, what)
            eq_bits(base["u"], g["u"], what + " u")
            eq_bits(base["x"], g["x"], what + " x")
            assert (base["res"] == g["res"]).all(), what
            for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
                eq_bits(base["trace"][f], g["trace"][f], what + " trace." + f)
        # (whether a wavefront that ran dry took a trajectory over at the tail of such a small launch is a matter of timing —
        #  ADVICE r04 — and not asserted here; test_pairs_at_scale_equal_the_single_build sees ~1 000 hand-overs per launch)
        print(what, "trajectories handed over at the tail:", eng.resume_stats())
        eng.set_rollout_mode(-1)
        eng.set_helper_mode(-1)
        eng.set_group_mode(-1)
        assert (base["trace"]["trials"] == 1).any() and (base["trace"]["trials"] > 1).any()
        if name in ("three_bend", "two_straight"):
            assert (base["trace"]["trials"] == 20).any()  # failed searches: the second pass and the deep mode were exercised
    # warm starts and tick offsets (cs:163-180, ut:88-103): a short closed loop, every tick through the grouped build
    cfg, sc = scenarios["three_straight"]
    p = pkg.params_from_config(cfg, N=30)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
    eng.set_group_mode(2)
    eng.set_helper_mode(0)
    B = 13
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 77)
    solvers = [orc_det.solver(p) for _ in range(B)]
    for s_ in solvers:
        s_.reset()
    last_u = None
    for tick in range(6):
        scene = oracle_scene(sc, tick)
        out = eng.solve_batch(x0, tick=np.full(B, tick, np.int32), last_u=last_u, trace_cap=128)
        compare_solves(out, [solvers[b].solve(x0[b], scene) for b in range(B)], f"pairs, closed loop tick {tick}")
        last_u = out["u"].copy()
        x0 = out["x"][:, 1].copy()
    eng.close()
    # parameter sweep (16 barrier settings) and a zero iteration budget
    from oracle import Scene
    wl = pkg.workloads.config5(B_base=5, N=30)
    for max_iter in (None, 0, 3):
        params = wl.params if max_iter is None else [pkg.copy_params(q, max_iter=max_iter) for q in wl.params]
        eng = pkg.BatchedCILQR(params, wl.scenes)
        eng.set_group_mode(2)
        eng.set_helper_mode(0)
        out = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, trace_cap=128)
        sc0 = wl.scenes[0]
        scene = Scene(sc0.lane_x, sc0.lane_y, sc0.lane_yaw, sc0.obs, sc0.road_borders, sc0.ref_velo)
        refs = [orc_det.solver(params[wl.param_id[b]]).solve(wl.x0[b], scene) for b in range(wl.B)]
        compare_solves(out, refs, f"pairs, config5 sweep max_iter={max_iter}")
        eng.close()
    # both vehicle models in one launch: the lanes of one rollout pass belong to different models
    cfg, sc = scenarios["three_bend"]
    pa = pkg.params_from_config(cfg, N=40, use_last_solution=0)
    pb = pkg.copy_params(pa, reference_point=1 - pa.reference_point, wheelbase=2.5, dt=0.12)
    eng = pkg.BatchedCILQR([pa, pb], pkg.SceneTable.from_scenario(sc))
    eng.set_group_mode(2)
    eng.set_helper_mode(0)
    B = 30
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 909)
    pid = (np.arange(B) % 3 == 0).astype(np.int32)
    out = eng.solve_batch(x0, param_id=pid, trace_cap=128)
    scene = oracle_scene(sc)
    refs = [orc_det.solver([pa, pb][pid[b]]).solve(x0[b], scene) for b in range(B)]
    compare_solves(out, refs, "pairs, two vehicle models in one launch")
    eng.close()


def test_pairs_at_scale_equal_the_single_build(pkg):
    """8192 three_bend trajectories (BASELINE config 3) through both builds: every output field identical (the oracle
    comparison of the full batch is test_full_size_configs_bitexact, which runs the automatic choice = pairs)."""
    wl = pkg.workloads.config3()
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    outs = {}
    for mode in (0, -1):
        eng.set_group_mode(mode)
        outs[mode] = eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick, trace_cap=64)
    assert eng.resume_stats() > 0, "no trajectory changed wavefronts at the tail of an 8192-trajectory launch in pairs"
    a, b = outs[0], outs[-1]
    eq_bits(a["u"], b["u"], "u")
    eq_bits(a["x"], b["x"], "x")
    assert (a["res"] == b["res"]).all()
    for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
        eq_bits(a["trace"][f], b["trace"][f], "trace." + f)
    # which two trajectories share a wavefront is a matter of timing and differs from launch to launch (the first launch of a
    # process most of all): the bits must not.  (Round 4's first tiled slab lost store data of a deep pass depending on the
    # pairing — costs off in the ninth digit; lone wavefronts, helpers off, are the reference then.)
    eng.set_helper_mode(0)
    eng.set_group_mode(0)
    lone = eng.solve_batch(wl.x0[:4100], wl.scenario_id[:4100], wl.param_id[:4100], wl.tick[:4100])
    eng.set_group_mode(2)
    for rep in range(3):
        again = eng.solve_batch(wl.x0[:4100], wl.scenario_id[:4100], wl.param_id[:4100], wl.tick[:4100])
        eq_bits(lone["u"], again["u"], f"u, launch {rep}")
        assert (lone["res"] == again["res"]).all()
    eng.close()


@pytest.mark.gpu
def test_results_do_not_depend_on_what_the_scratch_held(pkg, monkeypatch):
    """The kernels' scratch (trial slab, first-trial buffers, gains, expansion rows) is written before it is read, every
    iteration: filling it with NaN patterns or zeros before each launch (development library, CILQR_TUNE=poison) changes no
    bit — for lone wavefronts, helper wavefronts and pairs, horizons 50 and 100."""
    import os
    W = pkg.workloads
    for wl, modes in ((W.config3(B=rehearsal_size(2100)), (0, 2)), (W.config2(B=rehearsal_size(700)), (0,)), (W.config4(B=rehearsal_size(600)), (0,))):
        outs = []
        for tune in ("", "poison=1", "poison=2"):
            if tune:
                monkeypatch.setenv("CILQR_TUNE", tune)
            else:
                monkeypatch.delenv("CILQR_TUNE", raising=False)
            eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True)  # (the switch is read when the handle is made)
            per_mode = []
            for mode in modes:
                eng.set_group_mode(mode)
                per_mode.append(eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick))
            outs.append(per_mode)
            eng.close()
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                eq_bits(a["u"], b["u"], wl.name + " u")
                eq_bits(a["x"], b["x"], wl.name + " x")
                assert (a["res"] == b["res"]).all()



_IN_FLIGHT_SCRIPT = r"""
import sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import cilqr_amd as pkg
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
RES = pkg.RESULT_DTYPE
def res_of(t):
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=RES)
def bufs(B, N):
    return (torch.zeros((B, N, 2), dtype=torch.float64, device=dev), torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev),
            torch.zeros((B, RES.itemsize), dtype=torch.uint8, device=dev))
def same(a, b):
    return bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1], b[1])) and bool(torch.equal(a[2], b[2]))

import os
SHR = int(os.environ.get("CILQR_TEST_SHRINK", "1"))   # (rehearsals on the CPU emulator divide the batch sizes)
for wl_of, what in ((lambda f: pkg.workloads.config3(B=max(12, 3000 // SHR), first=f), "config 3 geometry (pairs per wavefront, persistent blocks, hand-over at the tail)"),
                    (lambda f: pkg.workloads.config4(B=max(8, 2400 // SHR), N=100, first=f), "horizon 100 (work sharing between blocks, resumable solves)"),
                    (lambda f: pkg.workloads.config2(B=max(4, 700 // SHR), first=f), "helper wavefronts")):
    wls = [wl_of(f) for f in (0, 5000, 10000, 15000, 20000)]
    B, N = wls[0].B, wls[0].N
    eng = pkg.BatchedCILQR(wls[0].params, wls[0].scenes)
    ins = [(to(w.x0), to(w.scenario_id), to(w.param_id), to(w.tick)) for w in wls]
    def solve(i, out, last_u=0):
        eng.solve_batch_device(B, ins[i][0].data_ptr(), ins[i][1].data_ptr(), ins[i][2].data_ptr(), ins[i][3].data_ptr(), last_u,
                               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), 0, 0, st)
    # the reference: one launch at a time
    ref = [bufs(B, N) for _ in wls]
    for i in range(len(wls)):
        solve(i, ref[i])
    torch.cuda.synchronize(dev)
    assert res_of(ref[0][2])["iters"].sum() > B
    for K in (2, 3, 4):
        eng.set_batches_in_flight(K)
        out = [bufs(B, N) for _ in wls]
        for rep in range(2):
            for i in range(len(wls)):
                solve(i, out[i])
        eng.join_device(st)
        torch.cuda.synchronize(dev)
        for i in range(len(wls)):
            assert same(out[i], ref[i]), (what, K, i)
        # hazards: every call into the SAME output arrays (write after write: ordered, the last one stays) ...
        one = bufs(B, N)
        for i in range(len(wls)):
            solve(i, one)
        eng.wait()
        assert same(one, ref[-1]), (what, K, "same outputs")
        # ... and a chain of warm starts, each from the previous call's plan (read after write)
        chain = [bufs(B, N) for _ in range(3)]
        solve(0, chain[0])
        solve(0, chain[1], chain[0][0].data_ptr())
        solve(0, chain[2], chain[1][0].data_ptr())
        eng.wait()
        eng.set_batches_in_flight(1)
        seq = [bufs(B, N) for _ in range(3)]
        solve(0, seq[0])
        solve(0, seq[1], seq[0][0].data_ptr())
        solve(0, seq[2], seq[1][0].data_ptr())
        torch.cuda.synchronize(dev)
        for a, b in zip(chain, seq):
            assert same(a, b), (what, K, "warm-start chain")
    # launches in flight, then a host-buffer call and the step after the path: both join by themselves
    eng.set_batches_in_flight(3)
    out = [bufs(B, N) for _ in range(3)]
    for i in range(3):
        solve(i, out[i])
    host = eng.solve_batch(wls[3].x0, wls[3].scenario_id, wls[3].param_id, wls[3].tick)
    assert np.array_equal(host["u"], ref[3][0].cpu().numpy()) and (host["res"] == res_of(ref[3][2])).all()
    x0n = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    for i in range(3):
        solve(i, out[i])
    eng.advance_batch_device(B, out[2][1].data_ptr(), x0n.data_ptr(), 0, st)
    torch.cuda.synchronize(dev)
    assert bool(torch.equal(x0n, ref[2][1][:, 1, :]))
    for i in range(3):
        assert same(out[i], ref[i])
    eng.close()

# the closed loop in one launch, two fleets in flight
cfg = pkg.GlobalConfig.get_instance("three_straight")
sc = pkg.build_scenario(cfg, "three_straight")
p = pkg.params_from_config(cfg, N=30, use_last_solution=1)
B, ticks, N = 2500, 6, 30
eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
def fleet(seed):
    x0 = to(pkg.workloads.perturbed_starts(sc.ego_state, B, seed))
    tick = torch.zeros(B, dtype=torch.int32, device=dev)
    o = bufs(B, N)
    states = torch.zeros((B, ticks, 4), dtype=torch.float64, device=dev)
    return [x0, tick, o, states]
def run(f):
    eng.closed_loop_batch_device(B, ticks, f[0].data_ptr(), 0, 0, f[1].data_ptr(), 0, f[2][0].data_ptr(), f[2][1].data_ptr(),
                                 f[2][2].data_ptr(), f[3].data_ptr(), 0, st)
seq = [fleet(1), fleet(2), fleet(3)]
for f in seq:
    run(f)
torch.cuda.synchronize(dev)
eng.set_batches_in_flight(3)
par = [fleet(1), fleet(2), fleet(3)]
for f in par:
    run(f)
eng.wait()
for a, b in zip(seq, par):
    assert same(a[2], b[2]) and bool(torch.equal(a[3], b[3])) and bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1], b[1]))
eng.close()
print("IN-FLIGHT-OK")
"""


def test_batches_in_flight_inside_one_handle():
    """cilqr_set_batches_in_flight (round 5): k launch slots inside one handle — internal streams, scratch and control
    words per slot, tables shared.  Five different batches through 2, 3 and 4 slots equal the same batches one launch at
    a time, bit for bit, on the three launch shapes (pairs per wavefront with the hand-over at the tail; horizon 100 with
    work sharing and resumable solves; helper wavefronts); calls that share output arrays or chain warm starts are
    ordered by the library's own hazard tracking; host-buffer calls and cilqr_advance_batch_device join by themselves;
    the closed loop in one launch with three fleets in flight."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _IN_FLIGHT_SCRIPT, root], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "IN-FLIGHT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]



_LOST_HAND_OVER_SCRIPT = r"""
import os, sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import cilqr_amd as pkg
from oracle import Oracle, Scene
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
RES = pkg.RESULT_DTYPE
NOT_SOLVED = 4
res_of = lambda t: np.frombuffer(t.cpu().numpy().tobytes(), dtype=RES)
def bufs(B, N):  # (results start as garbage that LOOKS like a result: the mark must come from the launch, not from the caller)
    r = np.zeros(B, dtype=RES); r["iters"] = 7; r["end_reason"] = 0; r["J_final"] = 1.0
    return (torch.zeros((B, N, 2), dtype=torch.float64, device=dev), torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev),
            torch.from_numpy(np.frombuffer(r.tobytes(), dtype=np.uint8).reshape(B, RES.itemsize).copy()).to(dev))
wl = pkg.workloads.config3(B=max(40, 3000 // int(os.environ.get("CILQR_TEST_SHRINK", "1"))), N=30)   # (shrunk in emulator rehearsals)
B, N = wl.B, wl.N
scenes = [Scene(s.lane_x, s.lane_y, s.lane_yaw, s.obs, s.road_borders, s.ref_velo) for s in wl.scenes]
ref = Oracle("det").solve_batch(wl.params, scenes, wl.x0, wl.scenario_id, wl.param_id, wl.tick, n_threads=8)
ins = (to(wl.x0), to(wl.scenario_id), to(wl.param_id), to(wl.tick))
eng = pkg.BatchedCILQR(wl.params, wl.scenes, dev=True)   # development library: the only one with the forcing knob
eng.set_group_mode(2)
def solve(out):
    eng.solve_batch_device(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), ins[3].data_ptr(), 0,
                           out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), 0, 0, st)
def check(out, must_lose):
    r = res_of(out[2])
    lost = r["end_reason"] == NOT_SOLVED
    ok = ~lost
    # every trajectory is either the oracle's, bit for bit, or carries the mark (and nothing that looks like a result)
    assert np.array_equal(out[0].cpu().numpy()[ok], ref["u"][ok]) and np.array_equal(out[1].cpu().numpy()[ok], ref["x"][ok])
    for f in ("iters", "end_reason", "ls_trials"):
        assert (r[f][ok] == ref["res"][f][ok]).all(), f
    assert (r["J_final"][ok].view(np.uint64) == ref["res"]["J_final"][ok].view(np.uint64)).all()
    assert (r["iters"][lost] == 0).all() and np.isnan(r["J_final"][lost]).all() and np.isnan(r["J_init"][lost]).all()
    assert must_lose is None or bool(lost.any()) == must_lose, int(lost.sum())
    return int(lost.sum())
def wait_fails():
    try:
        eng.wait()
    except RuntimeError as e:
        assert "bounded wait" in str(e) and "NOT_SOLVED" in str(e), str(e)
        return True
    return False

# 1. a healthy launch: nothing marked, nothing latched
a = bufs(B, N); solve(a); assert not wait_fails(); check(a, False)
assert eng.work_sharing_stats()["error"] == 0
# 2. the hand-over wait forced to expire at once: whoever was in transit is marked, the launch is reported, the report clears the latch
#    (the expiry is certain — some wavefront runs dry first and gives up after one look; that a trajectory is pushed to the place it
#     left behind is a matter of timing on hardware: all but certain with hundreds of hand-overs per launch, so a few launches
#     are allowed for it.  On the emulator's schedule the first launch loses some.)
os.environ["CILQR_GRP_WAIT_SPINS"] = "1"
for attempt in range(6):
    b = bufs(B, N); solve(b)
    assert eng.work_sharing_stats()["error"] != 0          # (shown without clearing)
    assert wait_fails(); n_lost = check(b, None)
    assert not wait_fails()                                  # (reported once)
    parked = eng.resume_stats()
    assert n_lost <= parked, (n_lost, parked)
    if n_lost:
        break
assert n_lost >= 1, "six launches with the wait forced to expire and no trajectory was in transit"
# 3. host-buffer entry point: CILQR_ERR_DEVICE, the outputs still delivered with the marks in them
try:
    eng.solve_batch(wl.x0, wl.scenario_id, wl.param_id, wl.tick)
    raise SystemExit("cilqr_solve_batch did not report the expired wait")
except RuntimeError as e:
    assert "bounded wait" in str(e), str(e)
# 4. three launches in flight, the failure in the FIRST slot only — the last launch's control words are clean (ADVICE r05):
#    the latch still reports it, and the later launches' results are whole
del os.environ["CILQR_GRP_WAIT_SPINS"]
eng.set_batches_in_flight(3)
outs = [bufs(B, N) for _ in range(3)]
os.environ["CILQR_GRP_WAIT_SPINS"] = "1"
solve(outs[0])
del os.environ["CILQR_GRP_WAIT_SPINS"]
solve(outs[1]); solve(outs[2])
eng.join_device(st); torch.cuda.synchronize(dev)
assert eng.work_sharing_stats()["error"] != 0
assert wait_fails()
check(outs[0], None); check(outs[1], False); check(outs[2], False)   # (whether slot 0 LOST one is timing; that it expired is not)
# 5. and the handle is healthy afterwards
c = bufs(B, N); solve(c); assert not wait_fails(); check(c, False)
eng.close()
print("LOST-HAND-OVER-OK", n_lost, parked)
"""


@pytest.mark.not_yet_run_on_hardware
def test_a_lost_hand_over_is_loud():
    """Round 6 (VERDICT r05 task 5): a trajectory lost between wavefronts cannot be mistaken for a result.  Launches
    that hand trajectories over pre-mark every cilqr_result CILQR_END_NOT_SOLVED on the launch stream; the development
    library's hand-over wait is forced to expire (CILQR_GRP_WAIT_SPINS=1, read per launch): exactly the trajectories in transit
    keep the mark, all others == oracle; cilqr_wait and cilqr_solve_batch return CILQR_ERR_DEVICE once; with three launches in
    flight a failure in a slot that is not the last one is still reported (the latch), the other launches are whole."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("CILQR_GRP_WAIT_SPINS", None)
    r = subprocess.run([sys.executable, "-c", _LOST_HAND_OVER_SCRIPT, root], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "LOST-HAND-OVER-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]



def test_lost_rows_shape_is_still_what_loses_rows():
    """Round 5 (VERDICT r04 task 5): the real instruction stream of round 4's lost-store anomaly.  ab/libLR.so — the library
    built with -DCILQR_LOSTROWS_REPRO: the grouped rollout pass with its 16-byte slab stores inside waterfall loops — against
    the shipped library, pairs per wavefront against lone wavefronts (scripts/lost_rows_repro.py), with XNACK off (how this
    pool runs) and with HSA_XNACK=1.  (Round 6: the experiment library is libcilqr_amd_lostrows.so, built by build().)  ASSERTED: the shipped library is clean in both modes.  RECORDED (printed, and in
    profiles/r05_experiments/lost_rows_time_box.txt: 12 of 12 launches, ~290 of 4 100 trajectories with XNACK off, none with
    XNACK on): what the excluded shape does on this box — a hardware / firmware revision that stops losing rows shows up here.
    Never skipped: the experiment library is built on the spot when it is not there."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib
    b = importlib.import_module("toy-example-of-ilqr_amd.build")
    lib = str(b.build_lostrows())  # part of build(); rebuilt here (hipcc, ~1 min) when absent or older than the sources: never skipped
    seen = {}
    for which in ("shipped", "repro"):
        for xnack in (None, "1"):
            env = dict(os.environ)
            env.pop("HSA_XNACK", None)
            env.pop("CILQR_AMD_LIB", None)
            if xnack:
                env["HSA_XNACK"] = xnack
            if which == "repro":
                env["CILQR_AMD_LIB"] = lib
            r = subprocess.run([sys.executable, os.path.join(root, "scripts", "lost_rows_repro.py"), "2"], capture_output=True,
                               text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            seen[(which, xnack)] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print({f"{k[0]} xnack={k[1]}": v["mismatching_trajectories_per_launch"] for k, v in seen.items()})
    assert seen[("shipped", None)]["launches_with_mismatch"] == 0
    assert seen[("shipped", "1")]["launches_with_mismatch"] == 0



def test_sharded_solver_in_one_process(pkg, orc_det, scenarios):
    """cilqr_amd::ShardedSolver through examples/headless_planner --batch: the batch over G handles in one process (one host
    thread per shard), statistics summed on the host.  --devices 1 equals the plain cilqr_solve_batch call (and the oracle);
    G = 2, 3 and 5 shards — sharing the one GPU of this box: a rehearsal of the multi-GPU path — give the same checksum over
    every output bit: results do not depend on the shard count."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "headless_planner")
    cfgp = str(pkg.config.SCENARIO_DIR / "three_bend.json")
    B, N = 333, 30
    lines = {}
    for args in (["--devices", "1"], ["--devices", "2", "--share"], ["--devices", "3", "--share"], ["--devices", "5", "--share"]):
        r = subprocess.run([exe, cfgp, "--batch", str(B), "--horizon", str(N)] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        lines[" ".join(args)] = r.stdout.strip().split()
    one = lines["--devices 1"]
    field = lambda ln, name: ln[ln.index(name) + 1]
    for k, ln in lines.items():
        assert field(ln, "checksum") == field(one, "checksum"), (k, ln, one)
        for name in ("iters", "ls_trials", "converged", "max_lamb", "max_iter", "sum_J_final"):
            assert field(ln, name) == field(one, name), (k, name)
    assert [field(lines[k], "devices") for k in lines] == ["1", "2", "3", "5"]
    # the same batch through the Python binding (one handle) and the oracle
    cfg, sc = scenarios["three_bend"]
    p = pkg.params_from_config(cfg, N=N, use_last_solution=0)
    x0 = pkg.workloads.perturbed_starts(sc.ego_state, B, 0xC11A0B5)
    eng = pkg.BatchedCILQR(p, pkg.SceneTable.from_scenario(sc))
    out = eng.solve_batch(x0)
    eng.close()
    assert int(field(one, "iters")) == int(out["res"]["iters"].sum()) and int(field(one, "ls_trials")) == int(out["res"]["ls_trials"].sum())
    acc = 0.0
    for v in out["res"]["J_final"]:
        acc += float(v)
    assert float(field(one, "sum_J_final")) == acc
    ref = orc_det.solve_batch(p, oracle_scene(sc), x0, n_threads=4)
    eq_bits(out["x"], ref["x"], "x vs oracle")



def test_sweeps_of_two_trajectories_in_one_instruction_stream(pkg, orc_det, scenarios, monkeypatch):
    """Round 5: backward_sweep_pair — the backward sweeps (cs:383-440) of a wavefront's two trajectories on two 4 x 8 lane
    grids in ONE instruction stream, the second trajectory's Jacobians and expansion streamed from rows in global memory.
    Against round 4's turn (development library, CILQR_TUNE=pair_sweep=0: one sweep after the other) and the oracle: every
    output and the whole decision trace, on workloads whose solves include failed backward passes (a negative control
    weight: non-PD Q_uu, cs:415-420 — one trajectory of a pair fails while the other goes on), both vehicle models, odd
    batches, a horizon that is not a compile-time one, mixed parameter sets (different dt in the two halves)."""
    cases = []
    cfg, sc = scenarios["three_bend"]
    cases.append(("three_bend N=50", [pkg.params_from_config(cfg, N=50, use_last_solution=0)], sc, 301, 50))
    cases.append(("three_bend N=37, two parameter sets with different dt",
                  [pkg.params_from_config(cfg, N=37, use_last_solution=0), pkg.params_from_config(cfg, N=37, use_last_solution=0, dt=0.08)],
                  sc, 150, 37))
    cfg2, sc2 = scenarios["two_straight"]
    cases.append(("two_straight N=30, w_acc < 0 (non-PD Q_uu)", [pkg.params_from_config(cfg2, N=30, use_last_solution=0, w_acc=-40.0),
                                                               pkg.params_from_config(cfg2, N=30, use_last_solution=0)], sc2, 97, 30))
    bpf = 0
    for what, plist, scn, B, N in cases:
        x0 = pkg.workloads.perturbed_starts(scn.ego_state, B, 777 + N)
        pid = (np.arange(B) % len(plist)).astype(np.int32)
        outs = {}
        for tune in ("group=2", "group=2,pair_sweep=0"):
            monkeypatch.setenv("CILQR_TUNE", tune)
            eng = pkg.BatchedCILQR(plist, pkg.SceneTable.from_scenario(scn), dev=True)
            outs[tune] = eng.solve_batch(x0, param_id=pid, trace_cap=128)
            assert eng.last_launch_info()["trajectories_per_wavefront"] == 2
            eng.close()
        a, b = outs["group=2"], outs["group=2,pair_sweep=0"]
        eq_bits(a["u"], b["u"], what + " u")
        eq_bits(a["x"], b["x"], what + " x")
        assert (a["res"] == b["res"]).all(), what
        for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
            eq_bits(a["trace"][f], b["trace"][f], what + " trace." + f)
        scene = oracle_scene(scn)
        for bb in range(0, B, 7):
            s_ = orc_det.solver(plist[pid[bb]])
            r = s_.solve(x0[bb], scene)
            eq_bits(a["u"][bb], r["u"], f"{what} u[{bb}] vs oracle")
            eq_bits(a["x"][bb], r["x"], f"{what} x[{bb}] vs oracle")
            assert a["res"]["iters"][bb] == r["res"]["iters"] and a["res"]["J_final"][bb] == r["res"]["J_final"] or (
                np.isnan(a["res"]["J_final"][bb]) and np.isnan(r["res"]["J_final"]))
        for bb in range(B):
            bpf += int((a["trace"]["status"][bb][:a["res"]["trace_len"][bb]] == 2).sum())
    monkeypatch.delenv("CILQR_TUNE", raising=False)
    assert bpf > 0, "no backward pass failed: the second pass of a turn was not exercised"


@pytest.mark.parametrize("N,B", [(64, 203), (76, 150), (100, 301), (127, 97)])
def test_two_trajectories_per_wavefront_at_long_horizons(pkg, orc_det, N, B, monkeypatch):
    """Round 5: the grouped kernel's LONG layout — horizons of 64 ... 127, two rows per lane: both trajectories' expansions
    and Jacobians stream from rows in global memory through one ring per half of the wavefront (backward_sweep_pair<BOTH>),
    the gains reach the rollout pass through a ring of two eight-step chunks (rollout_group_long), shadow lanes store out
    of range.  Against k_solve's lone-wavefront builds (CILQR_TUNE=group_long=0) on every trajectory — outputs, results and
    the whole decision trace — and against the oracle on a sample; cold and warm starts; mixed scenarios (config 4's four,
    one of them with the rear-axle model), two parameter sets per scenario with different dt, a negative control weight on
    some rows (non-PD Q_uu: one half of a sweep fails while the other goes on), odd batches (a trajectory that sweeps alone)."""
    from oracle import Scene
    B = rehearsal_size(B)
    wl = pkg.workloads.config4(B=B, N=N)
    for s in wl.scenes:
        if s.obs.shape[1] < N + 1:
            pytest.skip("obstacle routes shorter than the horizon")
    params = []
    for q in wl.params:
        params.append(pkg.copy_params(q, max_iter=25))
        params.append(pkg.copy_params(q, max_iter=25, dt=0.08))
    params.append(pkg.copy_params(wl.params[0], max_iter=25, w_acc=-40.0))
    sid = wl.scenario_id
    pid = (2 * sid + (np.arange(B) // 4) % 2).astype(np.int32)
    pid[np.arange(B) % 13 == 5] = len(params) - 1
    rng = np.random.default_rng(N)
    last_u = rng.normal(0.0, 0.05, size=(B, N, 2))
    outs = {}
    for tune in ("group=2", "group=2,group_long=0"):
        monkeypatch.setenv("CILQR_TUNE", tune)
        eng = pkg.BatchedCILQR(params, wl.scenes, dev=True)
        outs[tune] = (eng.solve_batch(wl.x0, sid, pid, wl.tick, trace_cap=64),
                      eng.solve_batch(wl.x0, sid, pid, wl.tick, last_u=last_u, trace_cap=64))
        info = eng.last_launch_info()
        assert info["trajectories_per_wavefront"] == (2 if tune == "group=2" else 1), info
        eng.close()
    monkeypatch.delenv("CILQR_TUNE", raising=False)
    bpf = 0
    for k, what in ((0, "cold"), (1, "warm")):
        a, b = outs["group=2"][k], outs["group=2,group_long=0"][k]
        eq_bits(a["u"], b["u"], f"N={N} {what} u")
        eq_bits(a["x"], b["x"], f"N={N} {what} x")
        assert (a["res"] == b["res"]).all(), (N, what)
        for f in ("status", "trials", "accepted", "alpha_idx", "lamb", "new_J"):
            eq_bits(a["trace"][f], b["trace"][f], f"N={N} {what} trace.{f}")
        for bb in range(B):
            bpf += int((a["trace"]["status"][bb][:a["res"]["trace_len"][bb]] == 2).sum())
    assert bpf > 0, "no backward pass failed"
    a = outs["group=2"][0]
    for bb in range(0, B, 9):
        s0 = wl.scenes[sid[bb]]
        scene = Scene(s0.lane_x, s0.lane_y, s0.lane_yaw, s0.obs, s0.road_borders, s0.ref_velo)
        r = orc_det.solver(params[pid[bb]]).solve(wl.x0[bb], scene)
        eq_bits(a["u"][bb], r["u"], f"N={N} u[{bb}] vs oracle")
        eq_bits(a["x"][bb], r["x"], f"N={N} x[{bb}] vs oracle")
        assert a["res"]["iters"][bb] == r["res"]["iters"]
        assert a["res"]["J_final"][bb] == r["res"]["J_final"] or (np.isnan(a["res"]["J_final"][bb]) and np.isnan(r["res"]["J_final"]))


@pytest.mark.parametrize("cfg", ["3", "4"])
def test_sliced_solves_of_the_grouped_build(pkg, orc_det, cfg):
    """Round 5: the launches that run two trajectories per wavefront slice their solves (cilqr_set_resume_iters; automatic:
    16 / 12 iterations) once the last round of fresh trajectories is being handed out: at the end of a slice a trajectory is
    parked and queued, the slot takes the next one; places in the queue are CLAIMED (fetch-and-add), a slot whose place has no
    entry yet looks again every turn.  Whatever the slice length — 1 iteration (every trajectory changes slots after every
    iteration once the final round has begun), 5, the automatic one, none — every output and the whole decision trace are the
    ones of the unsliced launch, of lone wavefronts (group mode 0) and, on a sample, of the oracle; no wait expired."""
    wl = pkg.workloads.config3(B=rehearsal_size(5000)) if cfg == "3" else pkg.workloads.config4(B=rehearsal_size(4600))
    ids = (wl.scenario_id, wl.param_id, wl.tick)
    eng = pkg.BatchedCILQR(wl.params, wl.scenes)
    eng.set_resume_iters(0)
    if not pkg.library_info()["pairs_per_wavefront_by_default"]:
        eng.set_group_mode(2)  # (a library built with another compiler pairs only when asked: build.py VALIDATED_HIPCC)
    whole = eng.solve_batch(wl.x0, *ids, trace_cap=128)
    assert eng.last_launch_info()["trajectories_per_wavefront"] == 2
    for iters in (1, 5, -1):
        eng.set_resume_iters(iters)
        out = eng.solve_batch(wl.x0, *ids, trace_cap=128)
        parked = eng.resume_stats()
        st = eng.work_sharing_stats()
        assert st["error"] == 0, (iters, st)
        for k in ("u", "x"):
            eq_bits(whole[k], out[k], f"config {cfg}: {k} with {iters} iterations per slice")
        assert (whole["res"] == out["res"]).all() and (whole["trace"] == out["trace"]).all(), (cfg, iters)
        assert parked > (2000 if iters == 1 else 100), (cfg, iters, parked)
    eng.set_group_mode(0)
    eng.set_resume_iters(-1)
    lone = eng.solve_batch(wl.x0, *ids, trace_cap=128)
    eng.close()
    for k in ("u", "x"):
        eq_bits(whole[k], lone[k], f"config {cfg}: {k} against lone wavefronts")
    assert (whole["res"] == lone["res"]).all() and (whole["trace"] == lone["trace"]).all()
    scenes = [oracle_scene_tab(t) for t in wl.scenes]
    rows = np.r_[0:16, 1402, 4081]
    sid = wl.scenario_id if wl.scenario_id is not None else np.zeros(wl.B, np.int32)
    pid = wl.param_id if wl.param_id is not None else np.zeros(wl.B, np.int32)
    refs = [orc_det.solver(wl.params[pid[b]]).solve(wl.x0[b], scenes[sid[b]], trace_cap=128) for b in rows]
    sub = {k: whole[k][rows] for k in ("u", "x", "res", "trace")}
    compare_solves(sub, refs, f"sliced solves (config {cfg}, sample)")
