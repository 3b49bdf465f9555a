"""Whole-solve checks of the CPU oracle.  The reference pins NO full trajectory solve (SURVEY.md
appendix D last row: 'parity unpinned'), so the oracle's linearize -> block-tridiagonal Cholesky
path is cross-checked against an independent dense numpy least-squares solve of the same whitened
Jacobian, and the optimizer loop against its own invariants."""
import numpy as np
import pytest

import gpmp2_amd as g
from gpmp2_amd import problems


@pytest.fixture(scope="module")
def small_wam(oracle):
    p = problems.wam_restarts(B=3, total_step=12, obs_check_inter=3, opt="GN", sdf="40")
    return p, oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)


def test_normal_equations_match_dense_numpy(oracle, small_wam):
    p, r, s = small_wam
    b = 1
    args = (r, s, p.setting, p.start_conf[b:b + 1], p.start_vel[b:b + 1], p.end_conf[b:b + 1], p.end_vel[b:b + 1],
            p.init[b:b + 1])
    A, res = oracle.dense_linearize(*args)
    Hd, Ho, g, err = oracle.linearize(*args)
    assert err[0] == pytest.approx(0.5 * res @ res, rel=1e-12)
    n, nb = Hd.shape[2], Hd.shape[1]
    H = np.zeros((nb * n, nb * n))
    for i in range(nb):
        H[i * n:(i + 1) * n, i * n:(i + 1) * n] = Hd[0, i]
        if i + 1 < nb:
            H[(i + 1) * n:(i + 2) * n, i * n:(i + 1) * n] = Ho[0, i]
            H[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n] = Ho[0, i].T
    np.testing.assert_allclose(H, A.T @ A, rtol=1e-10, atol=1e-6 * np.abs(H).max() * 1e-6)
    np.testing.assert_allclose(g[0].reshape(-1), A.T @ res, rtol=1e-9, atol=1e-6)
    x, ok = oracle.block_tridiag_solve(Hd, Ho, -g)
    assert ok[0] == 1
    x_np = np.linalg.lstsq(A, -res, rcond=None)[0]
    np.testing.assert_allclose(x[0].reshape(-1), x_np, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("opt", ["GN", "LM", "DOGLEG"])
def test_optimizer_invariants(oracle, small_wam, opt):
    p, r, s = small_wam
    st = problems.wam_setting(12, 3, opt)
    res = oracle.batch_optimize(r, s, st, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    e0 = oracle.graph_error(r, s, st, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    ef = oracle.graph_error(r, s, st, p.start_conf, p.start_vel, p.end_conf, p.end_vel, res["traj"])
    np.testing.assert_allclose(res["error_trace"][:, 0], e0, rtol=1e-12)
    np.testing.assert_allclose(res["final_error"], ef, rtol=1e-12)
    assert np.all(ef < e0)                       # the no-increase guard holds
    assert np.all(res["iters"] >= 1) and np.all(res["iters"] <= st.max_iter)
    if opt != "GN":                              # LM / Dogleg never accept an increasing step
        tr = res["error_trace"]
        for b in range(p.B):
            t = tr[b][~np.isnan(tr[b])]
            assert np.all(np.diff(t) <= 1e-9 * t[0])


def test_fixed_iterations_budget(oracle, small_wam):
    p, r, s = small_wam
    st = problems.wam_setting(12, 3, "GN")
    st.fixed_iterations = 2
    res = oracle.batch_optimize(r, s, st, p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    assert list(res["iters"]) == [2] * p.B


def test_interpolate_arm_traj_known_answer(oracle, golden):
    # gpmp2/planner/tests/testTrajUtils.cpp:26-54: const velocity 10, dt 0.1, 4 interpolated steps
    Qc = 0.01 * np.eye(2)
    for k in range(1, 5):
        conf, vel = oracle.gp_interpolate(2, False, Qc, 0.1, 0.1 * k / 5, [0, 0], [10, 0], [1, 0], [10, 0])
        np.testing.assert_allclose(conf[0], [0.2 * k, 0], atol=1e-6)
        np.testing.assert_allclose(vel[0], [10, 0], atol=1e-6)


def test_collision_cost_uses_zero_epsilon(oracle, small_wam):
    p, r, s = small_wam
    c = oracle.collision_cost(r, s, 12, p.init)
    err, _ = oracle.obstacle_factor(r, s, 0.0, p.init[:, :, :7].reshape(-1, 7))
    np.testing.assert_allclose(c, err.reshape(p.B, -1).sum(axis=1), rtol=1e-12)


# ------------------------------------------------------------------ the reference's own small-graph solves
def _no_sphere_arm(dof):
    return g.ArmModel(g.Arm(dof, [1.0] * dof, [0.0] * dof, [0.0] * dof), [])


def gp_prior_graph_problem(golden):
    """testGaussianProcessPriorLinear.cpp:140-202: pose priors (sigma 1e-3) on both states + one GP prior, noisy
    initial values; Gauss-Newton must recover v1 = v2 = (1, 0, 0).  The planner always carries velocity priors on
    the end states; sigma 1e6 makes them vanish (weight 1e-12 against 1e5 of the GP prior)."""
    from gpmp2_amd.settings import TrajOptimizerSetting
    d = golden["gp_prior_linear"]
    o = d["optimization"]
    st = TrajOptimizerSetting(3)
    st.set_total_step(1)
    st.set_total_time(d["delta_t"])
    st.set_obs_check_inter(0)
    st.set_conf_prior_model(o["prior_sigma"])
    st.set_vel_prior_model(1e6)
    st.set_Qc_model(d["Qc_scale"] * np.eye(3))
    st.setGaussNewton()
    st.set_rel_thresh(1e-12)
    init = np.array([[o["p1init"] + o["v1init"], o["p2init"] + o["v2init"]]], dtype=float)
    z = np.zeros((1, 3))
    return st, np.array([o["p1"]], dtype=float), z, np.array([o["p2"]], dtype=float), z.copy(), init, o


def joint_limit_graph_problem(golden, conf):
    """testJointLimitFactorVector.cpp:67-158: limit factor (sigma 1e-3) + weak prior (sigma 1000) on one
    configuration; here on both states of a one-interval plan, which the GP prior leaves at rest."""
    from gpmp2_amd.settings import TrajOptimizerSetting
    d = golden["joint_limit"]
    st = TrajOptimizerSetting(2)
    st.set_total_step(1)
    st.set_total_time(1.0)
    st.set_obs_check_inter(0)
    st.set_conf_prior_model(1000.0)
    st.set_vel_prior_model(1.0)
    st.set_Qc_model(np.eye(2))
    st.set_flag_pos_limit(True)
    st.set_joint_pos_limits_down(d["down"])
    st.set_joint_pos_limits_up(d["up"])
    st.set_pos_limit_thresh(d["thresh"])
    st.set_pos_limit_model([0.001, 0.001])
    st.setGaussNewton()
    st.set_rel_thresh(1e-12)
    c = np.array([conf], dtype=float)
    z = np.zeros((1, 2))
    init = np.concatenate([c, z], axis=1)[:, None, :].repeat(2, axis=1)
    return st, c, z, c.copy(), z.copy(), init


def lie_gp_prior_graph_problem():
    """testGaussianProcessPriorPose2Vector.cpp:147-200: the same graph on Pose2Vector states (dof 6)"""
    from gpmp2_amd.settings import TrajOptimizerSetting
    st = TrajOptimizerSetting(6)
    st.set_total_step(1)
    st.set_total_time(0.1)
    st.set_obs_check_inter(0)
    st.set_conf_prior_model(0.001)
    st.set_vel_prior_model(1e6)
    st.set_Qc_model(0.01 * np.eye(6))
    st.setGaussNewton()
    st.set_rel_thresh(1e-12)
    pose1, pose2 = np.zeros(6), np.array([0.1, 0, 0, 0, 0.2, 0])
    v1, v2rnd = np.array([1.0, 0, 0, 0, 2, 0]), np.array([1.2, 0.3, 0.4, 1.0, 1.0, -1.0])
    init = np.array([[np.concatenate([pose1, v1]), np.concatenate([pose2, v2rnd])]])
    z = np.zeros((1, 6))
    model = g.ArmModel(g.Pose2MobileArm(g.Arm(3, [1.0] * 3, [0.0] * 3, [0.0] * 3)), [])
    return model, st, pose1[None], z, pose2[None], z.copy(), init, v1


def test_reference_small_graph_solves(oracle, golden):
    field = np.full((3, 3, 3), 10.0)
    st, sc, sv, ec, ev, init, o = gp_prior_graph_problem(golden)
    r, s = oracle.robot(_no_sphere_arm(3)), oracle.sdf([-1, -1, -1], 1.0, field)
    res = oracle.batch_optimize(r, s, st, sc, sv, ec, ev, init)
    np.testing.assert_allclose(res["traj"][0, 0, :3], o["p1"], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, 1, :3], o["p2"], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, 0, 3:], o["v1"], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, 1, 3:], o["v2"], atol=1e-6)
    assert res["final_error"][0] < 1e-6
    model, st, sc, sv, ec, ev, init, v1 = lie_gp_prior_graph_problem()
    res = oracle.batch_optimize(oracle.robot(model), s, st, sc, sv, ec, ev, init)
    np.testing.assert_allclose(res["traj"][0, :, :6], [sc[0], ec[0]], atol=1e-6)
    np.testing.assert_allclose(res["traj"][0, :, 6:], [v1, v1], atol=1e-6)
    assert res["final_error"][0] < 1e-6
    r2 = oracle.robot(_no_sphere_arm(2))
    for conf, want in (([0.0, 0.0], [0.0, 0.0]), ([-10.0, -10.0], [-3.0, -8.0]), ([10.0, 10.0], [3.0, 8.0])):
        st, sc, sv, ec, ev, init = joint_limit_graph_problem(golden, conf)
        res = oracle.batch_optimize(r2, s, st, sc, sv, ec, ev, init)
        np.testing.assert_allclose(res["traj"][0, :, :2], [want, want], atol=1e-6)


# ------------------------------------------------------------------ extra factors carried by a plan as data
def _numeric_gradient(oracle, ro, so, p, traj, h=1e-6):
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    g = np.zeros_like(traj)
    flat, gf = traj.reshape(-1), g.reshape(-1)
    for k in range(flat.size):
        keep = flat[k]
        flat[k] = keep + h
        ep = oracle.graph_error(ro, so, p.setting, *args, traj)[0]
        flat[k] = keep - h
        em = oracle.graph_error(ro, so, p.setting, *args, traj)[0]
        flat[k] = keep
        gf[k] = (ep - em) / (2 * h)
    return g


def test_goal_reach_graph_in_the_oracle(oracle):
    """matlab/Arm3GoalReachExample.m through the oracle: the GoalFactorArm rows are in the gradient (numeric check of
    d error / d x against J^T r) and the solve brings the end effector to the goal point with no end configuration"""
    from gpmp2_amd import problems
    p = problems.arm3_goal_reach()
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    traj = p.init + 0.05 * np.random.default_rng(8).normal(size=p.init.shape)
    _, _, g, _ = oracle.linearize(ro, so, p.setting, *args, traj)
    gn = _numeric_gradient(oracle, ro, so, p, traj.copy())
    np.testing.assert_allclose(g, gn.reshape(g.shape), rtol=2e-5, atol=2e-5 * np.abs(gn).max())
    res = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
    poses, _ = oracle.forward_kinematics(ro, res["traj"][0, -1, :3])
    np.testing.assert_allclose(poses[0, 2, :3, 3], [0.0, 1.1, 0.0], atol=1e-3)
    assert res["final_error"][0] < 0.1 * oracle.graph_error(ro, so, p.setting, *args, p.init)[0]


def test_workspace_and_self_collision_rows_in_the_oracle(oracle):
    """WAMWorkspaceConstraintsExample graph + SelfCollisionArm rows: gradient of the oracle's factor list against
    numeric differentiation of its graph error"""
    from gpmp2_amd import problems
    fk = lambda model, q: oracle.forward_kinematics(oracle.robot(model), q)[0][0, 6]
    p = problems.wam_workspace_constraints(fk, sdf="24", B=1)
    p.setting.self_collision = np.array([[0, 9, 0.6, 0.05], [5, 12, 0.4, 0.05]])
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    traj = p.init + 0.1 * np.random.default_rng(9).normal(size=p.init.shape)
    _, _, g, _ = oracle.linearize(ro, so, p.setting, *args, traj)
    gn = _numeric_gradient(oracle, ro, so, p, traj.copy(), h=1e-7)
    np.testing.assert_allclose(g, gn.reshape(g.shape), rtol=1e-4, atol=1e-4 * np.abs(gn).max())
