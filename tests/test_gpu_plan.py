"""GPU parity, graph level: linearization (block-tridiagonal normal equations), the batched
block-tridiagonal Cholesky, and the whole Gauss-Newton solve against the CPU oracle.

Stated tolerances (BASELINE.md parity gate): normal-equation entries and gradients relative 1e-9
of the block's largest entry; per-iteration graph error relative 1e-9; final trajectory absolute
1e-6; identical per-trajectory iteration counts and status codes."""
import os

import numpy as np
import pytest

from gpmp2_amd import problems

pytestmark = pytest.mark.gpu


def _handles(engine, oracle, p):
    return (engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data),
            oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data))


def _args(p):
    return p.start_conf, p.start_vel, p.end_conf, p.end_vel


@pytest.fixture(scope="module")
def small_wam():
    return problems.wam_restarts(B=5, total_step=12, obs_check_inter=3, opt="GN", sdf="40")


@pytest.mark.parametrize("inter", [0, 3])
def test_linearize_matches_oracle(engine, oracle, inter):
    p = problems.wam_restarts(B=4, total_step=9, obs_check_inter=inter, opt="GN", sdf="40")
    r, s, ro, so = _handles(engine, oracle, p)
    a = engine.linearize(r, s, p.setting, *_args(p), p.init)
    b = oracle.linearize(ro, so, p.setting, *_args(p), p.init)
    for x, y in zip(a[:3], b[:3]):
        scale = np.abs(y).max()
        np.testing.assert_allclose(x, y, atol=1e-9 * scale)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    np.testing.assert_allclose(engine.graph_error(r, s, p.setting, *_args(p), p.init), b[3], rtol=1e-9)


def test_linearize_with_limits_and_planar_sdf(engine, oracle):
    p = problems.arm3_planner()
    p.setting.setGaussNewton()
    rng = np.random.default_rng(0)
    traj = p.init + 0.3 * rng.normal(size=p.init.shape)      # pushes some joints over the limits
    r, s, ro, so = _handles(engine, oracle, p)
    a = engine.linearize(r, s, p.setting, *_args(p), traj)
    b = oracle.linearize(ro, so, p.setting, *_args(p), traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)


def test_linearize_point_robot_skip_first(engine, oracle):
    p = problems.point_robot_2d()
    r, s, ro, so = _handles(engine, oracle, p)
    a = engine.linearize(r, s, p.setting, *_args(p), p.init)
    b = oracle.linearize(ro, so, p.setting, *_args(p), p.init)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)


@pytest.mark.parametrize("n,nblk", [(14, 101), (4, 11), (6, 51), (1, 7), (15, 3), (10, 1)])
def test_block_tridiag_solve(engine, oracle, n, nblk):
    rng = np.random.default_rng(n * 100 + nblk)
    B = 3
    # SPD block-tridiagonal from a random banded Jacobian
    Hd, Ho = np.zeros((B, nblk, n, n)), np.zeros((B, max(nblk - 1, 0), n, n))
    for b in range(B):
        for i in range(nblk):
            A = rng.normal(size=(3 * n, 2 * n))
            Hd[b, i] += A[:, :n].T @ A[:, :n] + 1e-3 * np.eye(n)
            if i + 1 < nblk:
                Hd[b, i + 1] += A[:, n:].T @ A[:, n:]
                Ho[b, i] = A[:, n:].T @ A[:, :n]
    rhs = rng.normal(size=(B, nblk, n))
    x, ok = engine.block_tridiag_solve(Hd, Ho, rhs)
    xo, oko = oracle.block_tridiag_solve(Hd, Ho, rhs)
    assert list(ok) == [1] * B and list(oko) == [1] * B
    # independent dense check
    for b in range(B):
        H = np.zeros((nblk * n, nblk * n))
        for i in range(nblk):
            H[i * n:(i + 1) * n, i * n:(i + 1) * n] = Hd[b, i]
            if i + 1 < nblk:
                H[(i + 1) * n:(i + 2) * n, i * n:(i + 1) * n] = Ho[b, i]
                H[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] = Ho[b, i].T
        xd = np.linalg.solve(H, rhs[b].reshape(-1))
        cond = np.linalg.cond(H)
        np.testing.assert_allclose(x[b].reshape(-1), xd, atol=1e-13 * cond * np.abs(xd).max())
    np.testing.assert_allclose(x, xo, atol=1e-9 * np.abs(xo).max())


def test_block_tridiag_solve_flags_indefinite(engine):
    Hd = np.tile(np.eye(4), (2, 3, 1, 1))
    Hd[1, 1, 2, 2] = -1.0
    x, ok = engine.block_tridiag_solve(Hd, np.zeros((2, 2, 4, 4)), np.ones((2, 3, 4)))
    assert list(ok) == [1, 0]
    np.testing.assert_allclose(x[0], 1.0, atol=1e-14)


def _compare_solves(res, ref, max_iter):
    assert list(res["iters"]) == list(ref["iters"])
    assert list(res["status"]) == list(ref["status"])
    ta, tb = res["error_trace"], ref["error_trace"]
    assert np.array_equal(np.isnan(ta), np.isnan(tb))
    m = ~np.isnan(tb)
    np.testing.assert_allclose(ta[m], tb[m], rtol=1e-9)
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-9)
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


def test_gauss_newton_solve_matches_oracle(engine, oracle, small_wam):
    p = small_wam
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    _compare_solves(res, ref, p.setting.max_iter)
    # size-independent properties: the returned values never have a larger error than the start,
    # and the reported final error is the graph error of the returned values
    e0 = engine.graph_error(r, s, p.setting, *_args(p), p.init)
    ef = engine.graph_error(r, s, p.setting, *_args(p), res["traj"])
    assert np.all(ef < e0)
    np.testing.assert_allclose(ef, res["final_error"], rtol=1e-9)


def test_fixed_iteration_budget_matches_oracle(engine, oracle, small_wam):
    p = small_wam
    st = problems.wam_setting(12, 3, "GN")
    st.fixed_iterations = 2
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, st, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, st, *_args(p), p.init)
    assert list(res["iters"]) == [2] * p.B
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-9)


def test_planar_arm_with_limits_gn_matches_oracle(engine, oracle):
    p = problems.arm3_planner()
    p.setting.setGaussNewton()
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    _compare_solves(res, ref, p.setting.max_iter)


def test_point_robot_config1_matches_oracle(engine, oracle):
    p = problems.point_robot_2d()
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    _compare_solves(res, ref, p.setting.max_iter)


def test_plan_can_be_rerun_and_reused(engine, small_wam):
    p = small_wam
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    pl = engine.plan(r, s, p.setting, p.B)
    pl.set_problem(*_args(p), p.init)
    pl.optimize()
    a = pl.result()
    pl.optimize()                       # idempotent: same inputs, same answer
    b = pl.result()
    np.testing.assert_array_equal(a["traj"], b["traj"])
    assert list(a["iters"]) == list(b["iters"])
    pl.set_problem(p.start_conf, p.start_vel, p.end_conf + 0.05, p.end_vel, p.init)
    pl.optimize()
    c = pl.result()
    assert not np.allclose(a["traj"], c["traj"])


def _full_size_vs_oracle(engine, oracle, p):
    """every restart of a full-size batch against the oracle (all host threads): iteration counts, status, final
    error (1e-9 rel) and trajectory (1e-6 abs) -- SURVEY.md 8(d) parity gate"""
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init, nthreads=min(os.cpu_count() or 1, 64))
    assert list(res["iters"]) == list(ref["iters"])
    assert list(res["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-9)
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)
    return r, s, res


def test_full_size_headline_config_properties(engine, oracle):
    """BASELINE config 3 at full size (WAM, N=100, I=5, 200^3 SDF): ALL 64 restarts against the oracle, plus
    size-independent properties."""
    p = problems.wam_restarts(B=64)
    r, s, res = _full_size_vs_oracle(engine, oracle, p)
    e0 = engine.graph_error(r, s, p.setting, *_args(p), p.init)
    ef = engine.graph_error(r, s, p.setting, *_args(p), res["traj"])
    assert np.all(ef < e0)
    np.testing.assert_allclose(ef, res["final_error"], rtol=1e-9)
    assert np.all(res["iters"] >= 1)
    # start / end priors (sigma 1e-4) hold the end points
    np.testing.assert_allclose(res["traj"][:, 0, :7], p.start_conf, atol=1e-3)
    np.testing.assert_allclose(res["traj"][:, -1, :7], p.end_conf, atol=1e-3)


def test_crosslane_primitives(engine):
    """Pins the lane semantics of the gfx950 moves the tile solver is built on
    (v_permlane16_swap / v_permlane32_swap, DPP row_newbcast / row_ror)."""
    import ctypes as C
    from gpmp2_amd._capi import dptr
    v = np.random.default_rng(11).normal(size=64)
    out = np.zeros((8, 64))
    engine._ck(engine.lib.gpmp2mi_debug_crosslane(dptr(v), dptr(out)))
    rows = v.reshape(4, 16)
    for gsel in range(4):
        np.testing.assert_array_equal(out[gsel].reshape(4, 16), np.tile(rows[gsel], (4, 1)))
    np.testing.assert_array_equal(out[4].reshape(4, 16), np.repeat(rows[:, 5:6], 16, axis=1))
    np.testing.assert_array_equal(out[7].reshape(4, 16), np.repeat(rows[:, 13:14], 16, axis=1))
    np.testing.assert_allclose(out[5].reshape(4, 16), np.repeat(rows.sum(axis=1, keepdims=True), 16, axis=1), atol=1e-14)
    np.testing.assert_allclose(out[6].reshape(4, 16), np.tile(rows.sum(axis=0), (4, 1)), atol=1e-14)


@pytest.mark.parametrize("opt", ["LM", "DOGLEG"])
def test_lm_and_dogleg_match_oracle(engine, oracle, small_wam, opt):
    """LevenbergMarquardt (lambda0 = 100) and Dogleg (delta0 = 0.2) step control on device vs the
    oracle's restatement of GTSAM's semantics ('parity unpinned' w.r.t. GTSAM itself)."""
    p = small_wam
    st = problems.wam_setting(12, 3, opt)
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, st, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, st, *_args(p), p.init)
    _compare_solves(res, ref, st.max_iter)
    tr = res["error_trace"]
    for b in range(p.B):      # accepted steps never increase the error
        t = tr[b][~np.isnan(tr[b])]
        assert np.all(np.diff(t) <= 1e-9 * t[0])


def test_arm3_planner_config2_dogleg_with_limits(engine, oracle):
    """BASELINE config 2: Arm3PlannerExample through BatchTrajOptimize2DArm rules -- Dogleg, planar
    SDF, joint + velocity limits, I = 3."""
    p = problems.arm3_planner()
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    _compare_solves(res, ref, p.setting.max_iter)


def test_generic_path_reproduces_fused_gauss_newton(engine, small_wam, monkeypatch):
    """The trial-step machinery (assemble / solve_step / linearize / decide) run with GaussNewton must
    give exactly what the fused fast path gives."""
    p = small_wam
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    a = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    monkeypatch.setenv("GPMP2MI_GENERIC_GN", "1")
    b = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    assert list(a["iters"]) == list(b["iters"]) and list(a["status"]) == list(b["status"])
    np.testing.assert_allclose(a["traj"], b["traj"], atol=1e-9)
    np.testing.assert_allclose(a["final_error"], b["final_error"], rtol=1e-10)


@pytest.mark.parametrize("opt,B", [("LM", 16), ("DOGLEG", 16)])
def test_full_size_headline_config_lm_dogleg(engine, oracle, opt, B):
    """the optimizers the reference's WAM scripts use (matlab/WAMFactorGraphExample.m:158-166 LM,
    WAMPlannerExample.m:118 Dogleg) at the headline size, every restart against the oracle"""
    _full_size_vs_oracle(engine, oracle, problems.wam_restarts(B=B, opt=opt))


def update_beyond_budget_check(engine, oracle, p):
    """gpmp2mi_plan_update(iterations) with iterations > the plan's fixed_iterations (any value <= max_iter is
    legal): the pass arrays are sized for it, and the result is the oracle's warm-started run."""
    import copy
    p.setting.setGaussNewton()
    p.setting.fixed_iterations = 1
    r, s, ro, so = _handles(engine, oracle, p)
    pl = engine.plan(r, s, p.setting, p.B)
    pl.set_problem(*_args(p), p.init)
    pl.optimize()
    first = pl.result()
    assert list(first["iters"]) == [1] * p.B
    pl.update(iterations=5)
    got = pl.result()
    st = copy.copy(p.setting)
    st.fixed_iterations = 5
    ref = oracle.batch_optimize(ro, so, st, *_args(p), first["traj"])
    assert list(got["iters"]) == [5] * p.B
    np.testing.assert_allclose(got["traj"], ref["traj"], atol=1e-6)
    np.testing.assert_allclose(got["final_error"], ref["final_error"], rtol=1e-8)
    with pytest.raises(Exception):
        pl.update(iterations=p.setting.max_iter + 1)


def test_update_beyond_the_fixed_iteration_budget(engine, oracle):
    update_beyond_budget_check(engine, oracle, problems.wam_restarts(B=3, total_step=10, obs_check_inter=2, sdf="40"))


# ------------------------------------------------------------------ Pose2 (Lie) path, BASELINE config 5
def test_mobile_arm_factors_vs_oracle(engine, oracle, golden):
    import gpmp2_amd as g
    from helpers import num, vec
    d = golden["pose2_mobile_arm"]
    arm = g.Arm(2, d["a"], d["alpha"], d["d"])
    base = g.pose3(g.rot_yaw(num(d["base_T_arm_yaw"])), d["base_T_arm_xyz"])
    model = g.Pose2MobileArmModel(g.Pose2MobileArm(arm, base),
                                  [g.BodySphere(0, 0.1, (0.2, 0.1, 0.0)), g.BodySphere(1, 0.1, (-0.5, 0, 0)),
                                   g.BodySphere(2, 0.1, (-0.3, 0.1, 0.05)), g.BodySphere(2, 0.1, (0, 0, 0))])
    r, ro = engine.robot(model), oracle.robot(model)
    for c in d["cases"]:                                   # the reference's literal link poses
        poses, _ = engine.forward_kinematics(r, vec(c["q"]))
        for l in range(3):
            np.testing.assert_allclose(poses[0, l], g.pose3(g.rot_yaw(num(c["yaw"][l])), c["xyz"][l]), atol=d["tol"])
    q = np.random.default_rng(21).uniform(-2, 2, size=(64, 5))
    for fa, fb in ((engine.forward_kinematics, oracle.forward_kinematics), (engine.sphere_centers, oracle.sphere_centers)):
        a, b = fa(r, q), fb(ro, q)
        np.testing.assert_allclose(a[0], b[0], atol=1e-9)
        np.testing.assert_allclose(a[1], b[1], atol=1e-9)
    p = problems.mobile_arm_config5()
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    q = np.random.default_rng(22).uniform([-4, -4, -3, -2, -2], [4, 4, 3, 2, 2], size=(256, 5))
    (ea, ha), (eb, hb) = engine.obstacle_factor(r, s, 0.1, q), oracle.obstacle_factor(ro, so, 0.1, q)
    assert (eb > 0).sum() > 10
    np.testing.assert_allclose(ea, eb, atol=1e-9)
    np.testing.assert_allclose(ha, hb, atol=1e-8)


def test_config5_linearize_matches_oracle(engine, oracle):
    p = problems.mobile_arm_config5()
    rng = np.random.default_rng(23)
    traj = p.init + 0.2 * rng.normal(size=p.init.shape)     # rotate / shift the base so the Lie blocks are non-trivial
    r, s, ro, so = _handles(engine, oracle, p)
    a = engine.linearize(r, s, p.setting, *_args(p), traj)
    b = oracle.linearize(ro, so, p.setting, *_args(p), traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)


@pytest.mark.parametrize("opt", ["DOGLEG", "GN", "LM"])
def test_config5_mobile_arm_solve_matches_oracle(engine, oracle, opt):
    """MobileArm2FactorGraphExample: Pose2Vector states, GaussianProcessPriorPose2Vector, planar obstacle
    factor and VehicleDynamicsFactorPose2Vector on every state; Dogleg is the script's optimizer."""
    p = problems.mobile_arm_config5()
    {"DOGLEG": p.setting.setDogleg, "GN": p.setting.setGaussNewton, "LM": p.setting.setLM}[opt]()
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    _compare_solves(res, ref, p.setting.max_iter)


def test_config4_receding_horizon_windows(engine, oracle):
    """BASELINE config 4: windows warm-started from the solved trajectory, 3 fixed GN iterations."""
    base = problems.wam_restarts(B=1, total_step=20, obs_check_inter=3, sdf="40")
    r, s, ro, so = _handles(engine, oracle, base)
    sol = engine.batch_optimize(r, s, base.setting, *_args(base), base.init)["traj"][0]
    p = problems.wam_windows(sol, B=6, total_step=20, obs_check_inter=3, fixed_iterations=3, sdf="40")
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    assert list(res["iters"]) == [3] * 6 == list(ref["iters"])
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-9)


def test_lie_factor_level_entry_points(engine, oracle, golden):
    """GaussianProcessPriorPose2Vector / GaussianProcessInterpolatorPose2Vector and the GP-interpolated
    planar obstacle factor of a Pose2 mobile arm: reference known answers + oracle parity."""
    d = golden["gp_prior_pose2vector"]
    for c in d["zero_error_cases"]:
        err, _ = engine.gp_prior_factor(6, True, d["delta_t"], c["p1"], c["v1"], c["p2"], c["v2"])
        np.testing.assert_allclose(err[0], 0.0, atol=d["tol"])
    gi = golden["gp_interpolator_pose2vector"]
    for c in gi["cases"]:
        conf, _ = engine.gp_interpolate(6, True, None, gi["delta_t"], gi["tau"], c["p1"], c["v1"], c["p2"], c["v2"])
        np.testing.assert_allclose(conf[0], c["expect"], atol=gi["tol"])
    rng = np.random.default_rng(31)
    a = [rng.uniform(-2, 2, size=(40, 6)) for _ in range(4)]
    a[0][0], a[1][0], a[2][0], a[3][0] = (np.array(d["random_case"][k], dtype=float) for k in ("p1", "v1", "p2", "v2"))
    (ea, Ha), (eb, Hb) = engine.gp_prior_factor(6, True, 0.1, *a), oracle.gp_prior_factor(6, True, 0.1, *a)
    np.testing.assert_allclose(ea, eb, atol=1e-9)
    for k in range(4):
        np.testing.assert_allclose(Ha[k], Hb[k], atol=1e-8)
    for x, y in zip(engine.gp_interpolate(6, True, None, 0.1, 0.03, *a), oracle.gp_interpolate(6, True, None, 0.1, 0.03, *a)):
        np.testing.assert_allclose(x, y, atol=1e-9)
    p = problems.mobile_arm_config5()
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    q = rng.uniform([-3, -3, -3, -2, -2], [3, 3, 3, 2, 2], size=(200, 5))
    v = rng.normal(size=(200, 5))
    q2 = q + 0.1 * v + 0.02 * rng.normal(size=(200, 5))
    v2 = v + 0.1 * rng.normal(size=(200, 5))
    (ea, Ha), (eb, Hb) = (f(rr, ss, 0.1, None, 0.1, 0.04, q, v, q2, v2) for f, rr, ss in
                          ((engine.obstacle_gp_factor, r, s), (oracle.obstacle_gp_factor, ro, so)))
    assert (eb > 0).sum() > 5
    np.testing.assert_allclose(ea, eb, atol=1e-9)
    for k in range(4):
        np.testing.assert_allclose(Ha[k], Hb[k], atol=1e-8)


def test_mobile_arm_planner_with_gp_interpolation(engine, oracle):
    """BatchTrajOptimizePose2MobileArm2D-style graph: Pose2 mobile arm WITH GP-interpolated obstacle
    factors (obs_check_inter = 3), linearization and a Gauss-Newton / Dogleg solve vs the oracle."""
    p = problems.mobile_arm_config5()
    p.setting.set_obs_check_inter(3)
    p.setting.set_total_step(20)
    N = 20
    init = np.zeros((1, N + 1, 10))
    for i in range(N + 1):
        init[0, i] = p.init[0, 0] * (N - i) / N + p.init[0, -1] * i / N
    init[0, :, 5:] = (p.end_conf[0] - p.start_conf[0])[None, :] / 5.0
    rng = np.random.default_rng(41)
    traj = init + 0.1 * rng.normal(size=init.shape)
    r, s, ro, so = _handles(engine, oracle, p)
    a = engine.linearize(r, s, p.setting, *_args(p), traj)
    b = oracle.linearize(ro, so, p.setting, *_args(p), traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    for opt in ("GN", "DOGLEG"):
        {"DOGLEG": p.setting.setDogleg, "GN": p.setting.setGaussNewton}[opt]()
        res = engine.batch_optimize(r, s, p.setting, *_args(p), init)
        ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), init)
        _compare_solves(res, ref, p.setting.max_iter)


def test_pose2_mobile_base_robot(engine, oracle):
    import gpmp2_amd as g
    model = g.Pose2MobileBaseModel(g.Pose2MobileBase(), [g.BodySphere(0, 0.2, (0.1, 0.0, 0.0)), g.BodySphere(0, 0.2, (-0.1, 0.05, 0.0))])
    r, ro = engine.robot(model), oracle.robot(model)
    q = np.random.default_rng(51).uniform(-3, 3, size=(32, 3))
    for fa, fb in ((engine.forward_kinematics, oracle.forward_kinematics), (engine.sphere_centers, oracle.sphere_centers)):
        a, b = fa(r, q), fb(ro, q)
        np.testing.assert_allclose(a[0], b[0], atol=1e-9)
        np.testing.assert_allclose(a[1], b[1], atol=1e-9)


def test_replanner_fix_state_change_goal_update(engine, oracle):
    """WAMReplannerExample flow (matlab/WAMReplannerExample.m:102-126): batch solve, then
    fixConfigAndVel(5, ...), changeGoalConfigAndVel(...), update(); plus addStateEstimate with a full
    covariance and removeGoalConfigAndVel.  The oracle runs the same warm-started fixed-iteration
    Gauss-Newton with the same extra priors (exact iSAM2 parity is unpinned, see include/gpmp2mi.h)."""
    p = problems.wam_restarts(B=2, total_step=10, obs_check_inter=4, sdf="40")
    r, s, ro, so = _handles(engine, oracle, p)
    D = 7
    pl = engine.plan(r, s, p.setting, p.B)
    pl.set_problem(*_args(p), p.init)
    pl.optimize()
    first = pl.result()["traj"]
    # --- step 1: execute up to state 5, fix it, move the goal of trajectory 0
    new_goal = np.array([-0.6, 0.94, 0, 1.6, 0, -0.919, 1.55])
    pl.fix_state(0, 5, first[0, 5, :D], first[0, 5, D:])
    pl.change_goal(0, new_goal, np.zeros(D))
    rng = np.random.default_rng(61)
    A = rng.normal(size=(D, D))
    cov = 1e-4 * (A @ A.T + D * np.eye(D))
    est = first[1, 3, :D] + 0.01
    pl.add_state_estimate(1, 3, est, cov)                      # pose-only estimate on trajectory 1
    pl.remove_goal(1)
    pl.update(iterations=2)
    got = pl.result()
    st = problems.wam_setting(10, 4, "GN")
    st.fixed_iterations = 2
    w = 1.0 / st.conf_prior_sigma ** 2
    priors = [[dict(state=5, conf=first[0, 5, :D], Wc=w * np.eye(D), vel=first[0, 5, D:], Wv=w * np.eye(D))],
              [dict(state=3, conf=est, Wc=np.linalg.inv(cov))]]
    end = p.end_conf.copy()
    end[0] = new_goal
    ref = oracle.batch_optimize_xp(ro, so, st, p.start_conf, p.start_vel, end, p.end_vel, first, priors, [1, 0])
    assert list(got["iters"]) == [2, 2]
    np.testing.assert_allclose(got["traj"], ref["traj"], atol=1e-6)
    np.testing.assert_allclose(got["final_error"], ref["final_error"], rtol=1e-8)
    # the fixed state stayed put, the new goal is reached, the free end of trajectory 1 moved
    np.testing.assert_allclose(got["traj"][0, 5, :D], first[0, 5, :D], atol=1e-3)
    np.testing.assert_allclose(got["traj"][0, -1, :D], new_goal, atol=1e-3)
    # --- step 2: a second update continues from the new estimate
    pl.update(iterations=1)
    again = pl.result()
    st.fixed_iterations = 1
    ref2 = oracle.batch_optimize_xp(ro, so, st, p.start_conf, p.start_vel, end, p.end_vel, ref["traj"], priors, [1, 0])
    np.testing.assert_allclose(again["traj"], ref2["traj"], atol=1e-6)
    # --- clearing the priors and re-optimising from scratch reproduces the batch answer of the new goal
    pl.clear_state_priors(0)
    pl.clear_state_priors(1)
    pl.change_goal(1, p.end_conf[1], p.end_vel[1])
    pl.optimize()
    ref3 = oracle.batch_optimize(ro, so, p.setting, p.start_conf, p.start_vel, end, p.end_vel, p.init)
    np.testing.assert_allclose(pl.result()["traj"], ref3["traj"], atol=1e-6)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65])
@pytest.mark.parametrize("opt", ["GN", "LM"])
def test_every_tree_shape_of_the_cyclic_reduction(engine, oracle, N, opt):
    """trajectory lengths around every power of two: each one exercises a different shape of the
    elimination tree (levels fused into the assemble kernel, last partly filled group, final level)"""
    p = problems.wam_restarts(B=3, total_step=N, obs_check_inter=2, opt=opt, sdf="40", max_iter=6)
    r, s, ro, so = _handles(engine, oracle, p)
    res = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    _compare_solves(res, ref, p.setting.max_iter)


def test_edge_sizes_empty_long_and_many_spheres(engine, oracle):
    """edge cases: empty batches of evaluations, a single long trajectory (N = 600, beyond one wavefront of
    blocks per level), the largest sphere model the engine stages (96), a robot entirely outside the field"""
    import gpmp2_amd as g
    p = problems.wam_restarts(B=1, total_step=600, obs_check_inter=1, opt="GN", sdf="40", max_iter=4)
    r, s, ro, so = _handles(engine, oracle, p)
    # delta_t = total_time / 600 makes Q^-1 ~ 12 / delta_t^3 and the normal equations ill-conditioned: two
    # backward-stable Cholesky orders (cyclic reduction here, natural order in the oracle) then agree to
    # ~ cond * eps only -- measured 1e-10 at N = 100, 6e-9 at N = 300, 4e-7 at N = 600 (scripts/long_traj_cond.py).
    # The contract stays 1e-6 (identical iterations / status); should the trajectory exceed it, the oracle's own
    # 2-ulp sensitivity must explain it (tests/parity_bound.py).  The error of the FIRST linearization is free of any
    # amplification and is held to 1e-9.
    from parity_bound import check_contract
    rep = check_contract(engine, oracle, p, label="N = 600", final_error_rtol=1e-8)
    res, ref = rep["res"], rep["ref"]
    np.testing.assert_allclose(res["error_trace"][:, 0], ref["error_trace"][:, 0], rtol=1e-9)
    # M = 0 evaluations are a no-op at every factor-level entry point
    z7 = np.zeros((0, 7))
    assert engine.obstacle_factor(r, s, 0.2, z7)[0].shape == (0, 16)
    assert engine.sphere_centers(r, z7)[0].shape == (0, 16, 3)
    assert engine.forward_kinematics(r, z7)[0].shape == (0, 7, 4, 4)
    assert engine.gp_prior_factor(7, False, 0.1, z7, z7, z7, z7)[0].shape == (0, 14)
    assert engine.sdf_query(s, np.zeros((0, 3)))[0].shape == (0,)
    # 96 body spheres (GPMP2MI_MAX_SPHERES; the PR2 model has 65) on a 7-dof arm; one more is rejected
    wam = g.generateArm("WAMArm")
    rng = np.random.default_rng(77)
    sph = [g.BodySphere(int(rng.integers(0, 7)), 0.04, tuple(rng.uniform(-0.1, 0.1, size=3))) for _ in range(96)]
    big = g.ArmModel(wam.fk_model(), sph)
    rb, rbo = engine.robot(big), oracle.robot(big)
    q = rng.uniform(-1.5, 1.5, size=(20, 7))
    a, b = engine.obstacle_factor(rb, s, 0.2, q), oracle.obstacle_factor(rbo, so, 0.2, q)
    np.testing.assert_allclose(a[0], b[0], atol=1e-9)
    np.testing.assert_allclose(a[1], b[1], atol=1e-8)
    with pytest.raises(g.engine.Gpmp2miError):
        engine.robot(g.ArmModel(wam.fk_model(), sph + [g.BodySphere(0, 0.04, (0, 0, 0))]))
    # a robot far outside the field: SDFQueryOutOfRange is swallowed into zero error / zero Jacobian rows
    far = g.ArmModel(g.Arm(7, wam.fk_model().a, wam.fk_model().alpha, wam.fk_model().d, g.pose3(t=(50.0, 0.0, 0.0))), sph[:8])
    a, b = engine.obstacle_factor(engine.robot(far), s, 0.2, q), oracle.obstacle_factor(oracle.robot(far), so, 0.2, q)
    assert not a[0].any() and not a[1].any() and not b[0].any()


def test_two_plans_from_two_host_threads(engine, oracle):
    """independent plans driven concurrently from two host threads give the same answers as one after the other
    (per-plan device state, thread-local error state, no shared scratch)"""
    import threading
    pa = problems.wam_restarts(B=6, total_step=20, obs_check_inter=3, opt="GN", sdf="40")
    pbm = problems.wam_restarts(B=4, total_step=14, obs_check_inter=1, opt="LM", sdf="40")
    out = {}

    def run(key, p):
        r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
        pl = engine.plan(r, s, p.setting, p.B)
        pl.set_problem(*_args(p), p.init)
        for _ in range(5):
            pl.optimize()
        out[key] = pl.result()

    run("a_seq", pa)
    run("b_seq", pbm)
    ta, tb = threading.Thread(target=run, args=("a_par", pa)), threading.Thread(target=run, args=("b_par", pbm))
    ta.start(); tb.start(); ta.join(); tb.join()
    for k in ("a", "b"):
        np.testing.assert_array_equal(out[k + "_seq"]["iters"], out[k + "_par"]["iters"])
        np.testing.assert_array_equal(out[k + "_seq"]["traj"], out[k + "_par"]["traj"])


def test_reference_small_graph_solves_on_gpu(engine, golden):
    """the only solve-level expectations the reference's own tests hold
    (gp/tests/testGaussianProcessPriorLinear.cpp:140-202, kinematics/tests/testJointLimitFactorVector.cpp:67-158),
    through the planner on the GPU"""
    import gpmp2_amd as g
    from test_oracle_solve import gp_prior_graph_problem, joint_limit_graph_problem, lie_gp_prior_graph_problem
    field = np.full((3, 3, 3), 10.0)
    s = engine.sdf([-1, -1, -1], 1.0, field)

    def arm(dof):
        return engine.robot(g.ArmModel(g.Arm(dof, [1.0] * dof, [0.0] * dof, [0.0] * dof), []))

    st, sc, sv, ec, ev, init, o = gp_prior_graph_problem(golden)
    res = engine.batch_optimize(arm(3), s, st, sc, sv, ec, ev, init)
    np.testing.assert_allclose(res["traj"][0, 0, :3], o["p1"], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, 1, :3], o["p2"], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, 0, 3:], o["v1"], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, 1, 3:], o["v2"], atol=1e-6)
    assert res["final_error"][0] < 1e-6
    model, st, sc, sv, ec, ev, init, v1 = lie_gp_prior_graph_problem()   # testGaussianProcessPriorPose2Vector.cpp:147-200
    res = engine.batch_optimize(engine.robot(model), s, st, sc, sv, ec, ev, init)
    np.testing.assert_allclose(res["traj"][0, :, :6], [sc[0], ec[0]], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, :, 6:], [v1, v1], atol=1e-6)
    assert res["final_error"][0] < 1e-6
    r2 = arm(2)
    for conf, want in (([0.0, 0.0], [0.0, 0.0]), ([-10.0, -10.0], [-3.0, -8.0]), ([10.0, 10.0], [3.0, 8.0])):
        st, sc, sv, ec, ev, init = joint_limit_graph_problem(golden, conf)
        res = engine.batch_optimize(r2, s, st, sc, sv, ec, ev, init)
        np.testing.assert_allclose(res["traj"][0, :, :2], [want, want], atol=1e-6)


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("opt", ["GN", "LM"])
def test_every_block_width_of_the_one_tile_path(engine, oracle, D, opt):
    """planar arms with 1..7 joints: every instantiation of the one-tile kernels (block width n = 2 .. 14).  Round 3 found
    a select that hipcc miscompiled for n <= 8 only (cr_kernels.hip: schur_prod); the reference's own models stop at
    2, 3 and 7 joints, so the widths in between had no case."""
    import gpmp2_amd as g
    from gpmp2_amd import datasets
    from gpmp2_amd.settings import TrajOptimizerSetting
    from gpmp2_amd.trajutils import initArmTrajStraightLine
    from parity_bound import check_contract
    arm = g.Arm(D, [0.9 / D] * D, [0.0] * D, [0.0] * D)
    model = g.ArmModel(arm, [g.BodySphere(l, 0.05, (-0.45 / D, 0, 0)) for l in range(D)])
    d = datasets.generate2Ddataset("TwoObstaclesDataset")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    N, B = 21, 3
    st = TrajOptimizerSetting(D)
    st.set_total_step(N); st.set_total_time(3.0); st.set_obs_check_inter(2); st.set_cost_sigma(0.1); st.set_epsilon(0.2)
    st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.set_max_iter(12)
    {"GN": st.setGaussNewton, "LM": st.setLM}[opt]()
    if opt == "GN":
        # these short arms barely reach an obstacle: the cost is close to quadratic, Gauss-Newton is essentially exact
        # after one or two steps and the last step changes the error by rounding noise -- "converged" vs "rolled back"
        # would be a coin toss (it flipped with the summation order of two Schur complements); a fixed iteration count
        # tests the kernels, not the toss.  LM keeps its own stopping rule.
        st.fixed_iterations = 3
    rng = np.random.default_rng(40 + D)
    start = np.zeros((B, D))
    end = np.linspace(0.3, 0.9, D)[None] + 0.2 * rng.normal(size=(B, D))
    init = np.stack([initArmTrajStraightLine(start[b], end[b], N) for b in range(B)])
    z = np.zeros((B, D))
    p = problems.Problem(f"planar arm, {D} joints", model, [d.origin_x, d.origin_y], d.cell_size, field, st, start, z, end,
                         z.copy(), init)
    check_contract(engine, oracle, p, label=p.name, final_error_rtol=1e-8)


@pytest.mark.parametrize("case", ["wam3d", "planar3", "planar5"])
def test_every_linearization_form_of_fixed_base_arms(engine, oracle, monkeypatch, case):
    """Fixed-base arms have three forms of the linearization kernel: one wavefront per 64 points (GPMP2MI_LIN_SPLIT=1), two
    wavefronts that both walk the chain (2; the default above 256 trajectories) and four that share one walk through LDS
    (4: k_linearize_arm, the default up to 256).  All three against the oracle's normal equations, and against each other."""
    import gpmp2_amd as g
    from gpmp2_amd import datasets
    from gpmp2_amd.settings import TrajOptimizerSetting
    from gpmp2_amd.trajutils import initArmTrajStraightLine
    rng = np.random.default_rng(12)
    if case == "wam3d":
        p = problems.wam_restarts(B=3, total_step=11, obs_check_inter=4, opt="GN", sdf="40")
        traj = p.init + 0.05 * rng.normal(size=p.init.shape)
    else:
        D = int(case[-1])
        arm = g.Arm(D, [0.9 / D] * D, [0.0] * D, [0.0] * D)
        model = g.ArmModel(arm, [g.BodySphere(l, 0.05, (-0.45 / D * k, 0, 0)) for l in range(D) for k in (0, 1)])
        d = datasets.generate2Ddataset("TwoObstaclesDataset")
        field = datasets.signedDistanceField2D(d.map, d.cell_size)
        N, B = 13, 3
        st = TrajOptimizerSetting(D)
        st.set_total_step(N); st.set_total_time(3.0); st.set_obs_check_inter(3); st.set_cost_sigma(0.1); st.set_epsilon(0.3)
        st.set_conf_prior_model(1e-3); st.set_vel_prior_model(1e-3); st.set_Qc_model(np.eye(D)); st.setGaussNewton()
        start = np.zeros((B, D))
        end = np.linspace(0.3, 0.9, D)[None] + 0.2 * rng.normal(size=(B, D))
        init = np.stack([initArmTrajStraightLine(start[b], end[b], N) for b in range(B)])
        z = np.zeros((B, D))
        p = problems.Problem(case, model, [d.origin_x, d.origin_y], d.cell_size, field, st, start, z, end, z.copy(), init)
        traj = p.init + 0.1 * rng.normal(size=p.init.shape)
    r, s, ro, so = _handles(engine, oracle, p)
    ref = oracle.linearize(ro, so, p.setting, *_args(p), traj)
    assert np.abs(ref[2]).max() > 0 and (np.abs(ref[0]).reshape(p.B, -1).max(axis=1) > 0).all()
    got = {}
    for form in ("1", "2", "4"):
        monkeypatch.setenv("GPMP2MI_LIN_SPLIT", form)
        got[form] = engine.linearize(r, s, p.setting, *_args(p), traj)
        for x, y in zip(got[form][:3], ref[:3]):
            np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max(), err_msg=f"form {form}")
        np.testing.assert_allclose(got[form][3], ref[3], rtol=1e-9, err_msg=f"form {form}")
    for form in ("2", "4"):
        for x, y in zip(got[form][:3], got["1"][:3]):
            np.testing.assert_allclose(x, y, atol=1e-12 * np.abs(y).max())


def test_both_run_ahead_modes_of_the_gauss_newton_driver(engine, small_wam, monkeypatch):
    """The host enqueues either the whole next pass or only its linearization before it looks at a pass count
    (GPMP2MI_GN_LOOKAHEAD=pass / lin, api.hip: plan_run_impl): same kernels in the same order on the same data, so the
    results are bit-identical; fixed-iteration runs (closing error pass) included."""
    from copy import deepcopy
    p = small_wam
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    for fixed in (0, 2):
        st = deepcopy(p.setting)
        st.fixed_iterations = fixed
        res = {}
        for mode in ("pass", "lin"):
            monkeypatch.setenv("GPMP2MI_GN_LOOKAHEAD", mode)
            res[mode] = engine.batch_optimize(r, s, st, *_args(p), p.init)
        for k in ("traj", "iters", "status", "final_error"):
            np.testing.assert_array_equal(res["pass"][k], res["lin"][k])


@pytest.mark.parametrize("N,inter,fixed", [(100, 5, 0), (37, 2, 0), (64, 3, 3), (23, 4, 0), (16, 2, 2)])
def test_fused_finish_is_the_finish_kernel(engine, oracle, monkeypatch, N, inter, fixed):
    """Gauss-Newton fast path of fixed-base arms: levels 4, 2, 1 of the back-substitution and the retract run either in
    k_finish_step or at the head of the next pass's k_linearize_arm (GPMP2MI_FUSED_FINISH=0 / default; the two state
    buffers then swap roles every pass).  Same tiles, same arithmetic: the results are bit-identical -- trajectories,
    iteration counts, status (including the rolled-back ones, which return the buffer the last step started from) --
    and they meet the oracle.  Sizes: the headline's, N not a multiple of 8, the smallest sub-step count the fused form
    takes, a fixed-iteration run (closing error pass), the smallest N with a split back-substitution."""
    from copy import deepcopy
    p = problems.wam_restarts(B=6, total_step=N, obs_check_inter=inter, opt="GN", sdf="40")
    st = deepcopy(p.setting)
    st.fixed_iterations = fixed
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GPMP2MI_FUSED_FINISH", mode)
        res[mode] = engine.batch_optimize(r, s, st, *_args(p), p.init)
    for k in ("traj", "iters", "status", "final_error", "error_trace"):
        np.testing.assert_array_equal(res["0"][k], res["1"][k], err_msg=k)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    if fixed == 0:
        ref = oracle.batch_optimize(ro, so, st, *_args(p), p.init)
        assert list(res["1"]["iters"]) == list(ref["iters"]) and list(res["1"]["status"]) == list(ref["status"])
        np.testing.assert_allclose(res["1"]["traj"], ref["traj"], atol=1e-6)


def test_fused_finish_of_the_trial_step_path(engine, oracle, monkeypatch):
    """LM on a fixed-base arm: the trial point cur (+) delta and the step-control shares g.delta, |delta|^2, |g|^2 come either
    from k_finish_trial (per group of 8 blocks) or from the head of the trial linearization (per chunk of 64 evaluation
    points; GPMP2MI_FUSED_FINISH=0 / default).  The shares are summed in another grouping, so the two forms agree to
    rounding, not bit for bit; both meet the oracle."""
    p = problems.wam_restarts(B=6, total_step=37, obs_check_inter=3, opt="LM", sdf="40")
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GPMP2MI_FUSED_FINISH", mode)
        res[mode] = engine.batch_optimize(r, s, p.setting, *_args(p), p.init)
    assert list(res["0"]["iters"]) == list(res["1"]["iters"]) and list(res["0"]["status"]) == list(res["1"]["status"])
    np.testing.assert_allclose(res["0"]["traj"], res["1"]["traj"], atol=1e-9)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ref = oracle.batch_optimize(ro, so, p.setting, *_args(p), p.init)
    assert list(res["1"]["iters"]) == list(ref["iters"]) and list(res["1"]["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["1"]["traj"], ref["traj"], atol=1e-6)
