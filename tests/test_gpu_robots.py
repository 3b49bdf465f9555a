"""GPU parity for the two-arm / vertical-lift mobile robots (Pose2Mobile2Arms, Pose2MobileVetLinArm,
Pose2MobileVetLin2Arms): forward kinematics against the reference's known answers
(kinematics/tests/testPose2Mobile2Arms.cpp, testPose2MobileVetLinArm.cpp, testPose2MobileVetLin2Arms.cpp),
sphere centres / obstacle factors / whole plans against the oracle."""
import numpy as np
import pytest

from parity_bound import check_contract

import gpmp2_amd as g
from gpmp2_amd import problems
from gpmp2_amd.settings import TrajOptimizerSetting
from helpers import num, tree_robot_from_golden, vec

pytestmark = pytest.mark.gpu

KEYS = ["pose2_mobile_2arms", "pose2_mobile_vetlin_arm", "pose2_mobile_vetlin_2arms"]


@pytest.mark.parametrize("key", KEYS)
def test_tree_robot_fk_and_spheres(engine, oracle, golden, key):
    d = golden[key]
    model = tree_robot_from_golden(d, key)
    r, ro = engine.robot(model), oracle.robot(model)
    L = model.fk_model().nr_links()
    for c in d["cases"]:
        poses, _ = engine.forward_kinematics(r, vec(c["q"]))
        for l in range(L):
            np.testing.assert_allclose(poses[0, l], g.pose3(g.rot_yaw(num(c["yaw"][l])), c["xyz"][l]), atol=d["tol"])
    rng = np.random.default_rng(7)
    q = rng.uniform(-3, 3, size=(50, model.dof()))
    for fa, fb in ((engine.forward_kinematics, oracle.forward_kinematics), (engine.sphere_centers, oracle.sphere_centers)):
        a, b = fa(r, q), fb(ro, q)
        np.testing.assert_allclose(a[0], b[0], atol=1e-11)
        np.testing.assert_allclose(a[1], b[1], atol=1e-11)
    if key == "pose2_mobile_vetlin_2arms":                     # the reference's random base poses, :122-150
        m2 = tree_robot_from_golden(d, key, random_bases=True)
        a = engine.forward_kinematics(engine.robot(m2), vec(d["random"]["q"]))
        b = oracle.forward_kinematics(oracle.robot(m2), vec(d["random"]["q"]))
        np.testing.assert_allclose(a[0], b[0], atol=1e-10)
        np.testing.assert_allclose(a[1], b[1], atol=1e-10)
    if key == "pose2_mobile_vetlin_arm":
        d2 = dict(d)
        d2["reverse_linact"] = True
        m3 = tree_robot_from_golden(d2, key)
        a, b = engine.sphere_centers(engine.robot(m3), q), oracle.sphere_centers(oracle.robot(m3), q)
        np.testing.assert_allclose(a[0], b[0], atol=1e-11)
        np.testing.assert_allclose(a[1], b[1], atol=1e-11)


SCALE = 4.0   # the desk scene blown up to [-4, 4]^3 so that the metre-sized test robots live inside it


def _small_field():
    origin, cell, data = problems.small3d_sdf(40)
    return list(np.array(origin) * SCALE), cell * SCALE, data * SCALE


@pytest.mark.parametrize("key", KEYS)
def test_tree_robot_obstacle_factors(engine, oracle, golden, key):
    d = golden[key]
    model = tree_robot_from_golden(d, key)
    r, ro = engine.robot(model), oracle.robot(model)
    origin, cell, data = _small_field()
    s, so = engine.sdf(origin, cell, data), oracle.sdf(origin, cell, data)
    rng = np.random.default_rng(11)
    D = model.dof()
    q1 = rng.uniform(-1.0, 1.0, size=(40, D))
    q2 = q1 + 0.2 * rng.normal(size=q1.shape)
    v1, v2 = rng.normal(size=q1.shape), rng.normal(size=q1.shape)
    a, b = engine.obstacle_factor(r, s, 0.8, q1), oracle.obstacle_factor(ro, so, 0.8, q1)
    np.testing.assert_allclose(a[0], b[0], atol=1e-9)
    np.testing.assert_allclose(a[1], b[1], atol=1e-8)
    assert (b[0] > 0).any()
    a = engine.obstacle_gp_factor(r, s, 0.8, None, 0.5, 0.2, q1, v1, q2, v2)
    b = oracle.obstacle_gp_factor(ro, so, 0.8, None, 0.5, 0.2, q1, v1, q2, v2)
    np.testing.assert_allclose(a[0], b[0], atol=1e-9)
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_allclose(x, y, atol=1e-8)


def _tree_problem(model, N=12, inter=2, opt="GN"):
    D = model.dof()
    origin, cell, data = _small_field()
    st = TrajOptimizerSetting(D)
    st.set_total_step(N)
    st.set_total_time(3.0)
    st.set_obs_check_inter(inter)
    st.set_cost_sigma(0.2)
    st.set_epsilon(0.6)
    st.set_conf_prior_model(1e-3)
    st.set_vel_prior_model(1e-3)
    st.set_Qc_model(np.eye(D))
    {"GN": st.setGaussNewton, "LM": st.setLM, "DOGLEG": st.setDogleg}[opt]()
    st.set_max_iter(30)
    st.set_rel_thresh(1e-4)
    start = np.zeros(D)
    start[:3] = [-1.5, -1.0, 0.3]
    end = np.zeros(D)
    end[:3] = [1.5, 1.2, -0.4]
    end[3:] = np.linspace(0.3, 0.9, D - 3)
    init = np.zeros((1, N + 1, 2 * D))
    for i in range(N + 1):
        init[0, i, :D] = start * (N - i) / N + end * i / N
    init[0, :, D:] = (end - start)[None, :] / 3.0
    z = np.zeros((1, D))
    return problems.Problem("tree", model, origin, cell, data, st, start[None], z.copy(), end[None], z.copy(), init)


@pytest.mark.parametrize("key,opt", [("pose2_mobile_2arms", "GN"), ("pose2_mobile_2arms", "DOGLEG"),
                                     ("pose2_mobile_vetlin_arm", "GN"), ("pose2_mobile_vetlin_arm", "LM")])
def test_tree_robot_plans(engine, oracle, golden, key, opt):
    """whole plans (dof 7 and 6): linearization and solve vs the oracle, incl. GP-interpolated obstacle
    factors on the Lie path"""
    model = tree_robot_from_golden(golden[key], key)
    p = _tree_problem(model, opt=opt)
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    rng = np.random.default_rng(19)
    traj = p.init + 0.05 * rng.normal(size=p.init.shape)
    a = engine.linearize(r, s, p.setting, *args, traj)
    b = oracle.linearize(ro, so, p.setting, *args, traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    res = engine.batch_optimize(r, s, p.setting, *args, p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
    assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-8)
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


def test_vetlin_2arms_plans_through_the_dense_path(engine, oracle, golden):
    """dof 8 > 7: blocks of 16 do not fit one tile, the plan runs on the dense block path"""
    model = tree_robot_from_golden(golden["pose2_mobile_vetlin_2arms"], "pose2_mobile_vetlin_2arms")
    p = _tree_problem(model)
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    ro, so = oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    res = engine.batch_optimize(r, s, p.setting, *args, p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
    assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


@pytest.mark.parametrize("arm_dof", [1, 4])
def test_mobile_arm_other_sizes(engine, oracle, arm_dof):
    """Pose2MobileArm with 1 and 4 arm joints (dof 4 and 7, the widest the block solver takes)"""
    arm = g.Arm(arm_dof, [0.4] * arm_dof, [0.0, np.pi / 2, 0.0, -np.pi / 2][:arm_dof], [0.1] * arm_dof)
    fk = g.Pose2MobileArm(arm, g.pose3(g.rot_yaw(0.3), (0.2, 0.0, 0.5)))
    sph = [g.BodySphere(l, 0.12, (-0.1, 0.0, 0.0)) for l in range(fk.nr_links())] + [g.BodySphere(0, 0.3, (0.0, 0.0, 0.2))]
    model = g.Pose2MobileArmModel(fk, sph)
    r, ro = engine.robot(model), oracle.robot(model)
    q = np.random.default_rng(3).uniform(-2, 2, size=(20, model.dof()))
    for fa, fb in ((engine.forward_kinematics, oracle.forward_kinematics), (engine.sphere_centers, oracle.sphere_centers)):
        a, b = fa(r, q), fb(ro, q)
        np.testing.assert_allclose(a[0], b[0], atol=1e-11)
        np.testing.assert_allclose(a[1], b[1], atol=1e-11)
    p = _tree_problem(model, N=10, inter=2, opt="GN")
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    res = engine.batch_optimize(r, s, p.setting, *args, p.init)
    ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
    assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


# ------------------------------------------------------------------ 8 <= dof <= 11: the dense block path
def _wide_models():
    wam = g.generateArm("WAMArm")
    arm7 = wam.fk_model()
    sph = wam.spheres
    mob = g.Pose2MobileArm(g.Arm(7, arm7.a, arm7.alpha, arm7.d), g.pose3(g.rot_yaw(0.2), (0.1, 0.0, 0.4)))
    lift = g.Pose2MobileVetLinArm(g.Arm(7, arm7.a, arm7.alpha, arm7.d), g.pose3(t=(0.0, 0.0, 0.3)), g.pose3(g.rot_yaw(-0.3), (0.2, 0.0, 0.2)))
    a3 = g.Arm(3, [0.5, 0.4, 0.3], [0.0, np.pi / 2, 0.0], [0.1, 0.0, 0.05])
    two = g.Pose2Mobile2Arms(a3, a3, g.pose3(g.rot_yaw(0.7), (0.3, 0.2, 0.5)), g.pose3(g.rot_yaw(-0.7), (0.3, -0.2, 0.5)))
    arm8 = g.Arm(8, [0.3] * 8, [0.0, np.pi / 2, 0.0, -np.pi / 2, 0.0, np.pi / 2, 0.0, 0.0], [0.1] * 8)

    def shift(spheres, k):
        return [g.BodySphere(s.link_id + k, s.radius, s.center) for s in spheres]

    def simple(fk):
        return [g.BodySphere(l, 0.15, (-0.05, 0.0, 0.0)) for l in range(fk.nr_links())]

    return {
        "arm8 (dof 8)": g.ArmModel(arm8, simple(arm8)),
        "2arms 3+3 (dof 9)": g.RobotModel(two, simple(two)) if hasattr(g, "RobotModel") else g.ArmModel(two, simple(two)),
        "mobile WAM (dof 10)": g.ArmModel(mob, [g.BodySphere(0, 0.3, (0, 0, 0.2))] + shift(sph, 1)),
        "lift WAM (dof 11)": g.ArmModel(lift, [g.BodySphere(0, 0.3, (0, 0, 0.2)), g.BodySphere(1, 0.2, (0, 0, 0))] + shift(sph, 2)),
    }


@pytest.mark.parametrize("name", ["arm8 (dof 8)", "2arms 3+3 (dof 9)", "mobile WAM (dof 10)", "lift WAM (dof 11)"])
def test_wide_robot_linearize_and_plans(engine, oracle, name):
    """robots whose blocks exceed one 16x16 tile: dense normal equations (2x2 tiles per block) and the
    dense block-Cholesky trial-step path, GN / LM / Dogleg, against the oracle"""
    model = _wide_models()[name]
    D = model.dof()
    p = _tree_problem(model, N=10, inter=2, opt="GN") if model.kind >= 2 else None
    if p is None:                                              # fixed-base arm: joint-space endpoints
        p = _tree_problem(model, N=10, inter=2, opt="GN")
        p.start_conf[0, :] = 0.1
        p.end_conf[0, :] = np.linspace(0.3, 0.9, D)
        for i in range(11):
            p.init[0, i, :D] = p.start_conf[0] * (10 - i) / 10 + p.end_conf[0] * i / 10
        p.init[0, :, D:] = (p.end_conf[0] - p.start_conf[0])[None, :] / 3.0
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    rng = np.random.default_rng(29)
    traj = p.init + 0.05 * rng.normal(size=p.init.shape)
    a = engine.linearize(r, s, p.setting, *args, traj)
    b = oracle.linearize(ro, so, p.setting, *args, traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    for opt in ("GN", "LM", "DOGLEG"):
        {"GN": p.setting.setGaussNewton, "LM": p.setting.setLM, "DOGLEG": p.setting.setDogleg}[opt]()
        res = engine.batch_optimize(r, s, p.setting, *args, p.init)
        ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
        assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"]), opt
        np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-8)
        np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


@pytest.mark.parametrize("name,N", [("mobile WAM (dof 10)", 20), ("mobile WAM (dof 10)", 35), ("lift WAM (dof 11)", 17),
                                    ("2arms 3+3 (dof 9)", 24)])
def test_wide_long_trajectories(engine, oracle, name, N):
    """total_step >= 16: forward levels 2 and 4 as chip-wide launches (k_cr_level_wide) and, for GN / LM, the
    back-substitution levels 4, 2, 1 + step + trial point in k_finish_trial_wide (groups of 8 blocks, the last one
    ragged for these N); Dogleg keeps the one-kernel tail.  Three trajectories per plan with different starts so that
    they leave the loop at different passes."""
    model = _wide_models()[name]
    D = model.dof()
    p = _tree_problem(model, N=N, inter=2, opt="GN")
    B = 3
    rng = np.random.default_rng(41)
    start = np.repeat(p.start_conf, B, 0)
    end = np.repeat(p.end_conf, B, 0)
    start[1:, 3:] += 0.2 * rng.normal(size=(B - 1, D - 3))
    end[1:, :2] += 0.3 * rng.normal(size=(B - 1, 2))
    init = np.zeros((B, N + 1, 2 * D))
    for b in range(B):
        for i in range(N + 1):
            init[b, i, :D] = start[b] * (N - i) / N + end[b] * i / N
        init[b, :, D:] = (end[b] - start[b])[None, :] / 3.0
    z = np.zeros((B, D))
    args = (start, z, end, z)
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    q = problems.Problem(name, p.model, p.sdf_origin, p.sdf_cell, p.sdf_data, p.setting, start, z, end, z.copy(), init)
    for opt in ("GN", "LM", "DOGLEG"):
        {"GN": p.setting.setGaussNewton, "LM": p.setting.setLM, "DOGLEG": p.setting.setDogleg}[opt]()
        # 1e-6 for every trajectory; the one LM trajectory (N = 35) on a flat valley of its cost that round 2 gave a 1e-5
        # gate is admitted only inside the oracle's own 2-ulp sensitivity (tests/parity_bound.py)
        rep = check_contract(engine, oracle, q, label=f"{name} N={N} {opt}", final_error_rtol=1e-8)
        assert rep["over"].size <= 1


def test_wide_dense_fallback_agrees(engine, oracle, monkeypatch):
    """GPMP2MI_WIDE_DENSE=1: the independent dense block-Cholesky implementation of the 8..11-dof solve"""
    model = _wide_models()["mobile WAM (dof 10)"]
    p = _tree_problem(model, N=10, inter=2, opt="LM")
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
    tiles = engine.batch_optimize(r, s, p.setting, *args, p.init)
    monkeypatch.setenv("GPMP2MI_WIDE_DENSE", "1")
    dense = engine.batch_optimize(r, s, p.setting, *args, p.init)
    for res in (tiles, dense):
        assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
        np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)
    np.testing.assert_allclose(tiles["traj"], dense["traj"], atol=1e-7)


@pytest.mark.parametrize("N", [10, 21])
def test_wide_robot_replanning(engine, oracle, N):
    """fix_state / change_goal / update on a 10-dof mobile manipulator (2x2-tile blocks carry the extra priors too);
    total_step 21 runs the chip-wide levels and the split tail"""
    model = _wide_models()["mobile WAM (dof 10)"]
    p = _tree_problem(model, N=N, inter=2, opt="GN")
    D = model.dof()
    r, ro = engine.robot(p.model), oracle.robot(p.model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    pl = engine.plan(r, s, p.setting, 1)
    pl.set_problem(p.start_conf, p.start_vel, p.end_conf, p.end_vel, p.init)
    pl.optimize()
    first = pl.result()["traj"]
    goal2 = p.end_conf[0].copy()
    goal2[:2] += [0.3, -0.2]
    goal2[3:] *= 0.5
    pl.fix_state(0, 4, first[0, 4, :D], first[0, 4, D:])
    pl.change_goal(0, goal2, np.zeros(D))
    pl.update(iterations=2)
    got = pl.result()
    st = p.setting
    st.setGaussNewton()
    st.fixed_iterations = 2
    w = 1.0 / st.conf_prior_sigma ** 2
    priors = [[dict(state=4, conf=first[0, 4, :D], Wc=w * np.eye(D), vel=first[0, 4, D:], Wv=w * np.eye(D))]]
    ref = oracle.batch_optimize_xp(ro, so, st, p.start_conf, p.start_vel, goal2[None], p.end_vel, first, priors, [1])
    np.testing.assert_allclose(got["traj"], ref["traj"], atol=1e-6)
    np.testing.assert_allclose(got["traj"][0, 4], first[0, 4], atol=1e-3)


def test_wide_robot_update_beyond_the_fixed_iteration_budget(engine, oracle):
    """plan created with fixed_iterations = 1, update(iterations = 5) on the 2x2-tile path (trial-step driver)"""
    from test_gpu_plan import update_beyond_budget_check
    update_beyond_budget_check(engine, oracle, _tree_problem(_wide_models()["mobile WAM (dof 10)"], N=10, inter=2, opt="GN"))


# ------------------------------------------------------------------ 12 <= dof <= 18: the PR2 model (dense block path)
def test_pr2_model_plans(engine, oracle):
    """generateMobileArm('PR2') (matlab/+gpmp2/generateMobileArm.m:244-349): SE(2) base + lift + two 7-joint arms,
    dof 18, 65 spheres -- the robot BatchTrajOptimizePose2MobileVetLin2Arms (planner/BatchTrajOptimizer.cpp:118-128) is
    instantiated for.  Kinematics, obstacle factors, normal equations (3x3-tile export) and GN / LM / Dogleg plans
    (dense blocks, cyclic reduction with one launch per level) against the oracle."""
    model = g.generateMobileArm("PR2")
    assert model.dof() == 18 and model.nr_body_spheres() == 65
    r, ro = engine.robot(model), oracle.robot(model)
    rng = np.random.default_rng(41)
    q = rng.uniform(-1.0, 1.0, size=(32, 18))
    for fa, fb in ((engine.forward_kinematics, oracle.forward_kinematics), (engine.sphere_centers, oracle.sphere_centers)):
        a, b = fa(r, q), fb(ro, q)
        np.testing.assert_allclose(a[0], b[0], atol=1e-9)
        np.testing.assert_allclose(a[1], b[1], atol=1e-9)
    p = _tree_problem(model, N=8, inter=1, opt="GN")
    p.end_conf[0, 3] = 0.2                                       # lift
    p.end_conf[0, 4:] = np.tile(np.linspace(0.2, 0.8, 7), 2) * np.r_[np.ones(7), -np.ones(7)]
    for i in range(9):
        p.init[0, i, :18] = p.start_conf[0] * (8 - i) / 8 + p.end_conf[0] * i / 8
    p.init[0, :, 18:] = (p.end_conf[0] - p.start_conf[0])[None, :] / 3.0
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    (ea, ha), (eb, hb) = engine.obstacle_factor(r, s, 0.6, p.init[0, :, :18]), oracle.obstacle_factor(ro, so, 0.6, p.init[0, :, :18])
    assert (eb > 0).sum() > 5
    np.testing.assert_allclose(ea, eb, atol=1e-9)
    np.testing.assert_allclose(ha, hb, atol=1e-8)
    args = (p.start_conf, p.start_vel, p.end_conf, p.end_vel)
    traj = p.init + 0.05 * rng.normal(size=p.init.shape)
    a = engine.linearize(r, s, p.setting, *args, traj)
    b = oracle.linearize(ro, so, p.setting, *args, traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    for opt in ("GN", "LM", "DOGLEG"):
        {"GN": p.setting.setGaussNewton, "LM": p.setting.setLM, "DOGLEG": p.setting.setDogleg}[opt]()
        res = engine.batch_optimize(r, s, p.setting, *args, p.init)
        ref = oracle.batch_optimize(ro, so, p.setting, *args, p.init)
        assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"]), opt
        np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-8)
        np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


def test_pr2_ragged_tree(engine, oracle):
    """PR2, total_step = 13 (14 blocks: every level of the dense cyclic reduction has a block without a right
    neighbour), two trajectories that stop at different passes, LM"""
    model = g.generateMobileArm("PR2")
    N, B = 13, 2
    p = _tree_problem(model, N=N, inter=1, opt="LM")
    start, end = np.repeat(p.start_conf, B, 0), np.repeat(p.end_conf, B, 0)
    end[:, 3] = 0.2
    end[0, 4:] = np.tile(np.linspace(0.2, 0.8, 7), 2) * np.r_[np.ones(7), -np.ones(7)]
    end[1, 4:] = np.tile(np.linspace(-0.3, 0.5, 7), 2)
    init = np.zeros((B, N + 1, 36))
    for b in range(B):
        for i in range(N + 1):
            init[b, i, :18] = start[b] * (N - i) / N + end[b] * i / N
        init[b, :, 18:] = (end[b] - start[b])[None, :] / 3.0
    z = np.zeros((B, 18))
    args = (start, z, end, z)
    r, ro = engine.robot(model), oracle.robot(model)
    s, so = engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    res = engine.batch_optimize(r, s, p.setting, *args, init)
    ref = oracle.batch_optimize(ro, so, p.setting, *args, init)
    assert list(res["iters"]) == list(ref["iters"]) and list(res["status"]) == list(ref["status"])
    np.testing.assert_allclose(res["final_error"], ref["final_error"], rtol=1e-8)
    np.testing.assert_allclose(res["traj"], ref["traj"], atol=1e-6)


@pytest.mark.parametrize("which", ["wam (one tile)", "mobile WAM (2x2 tiles)", "PR2 (dense blocks)"])
def test_bad_pivot_stops_one_trajectory_only(engine, which):
    """a trajectory whose normal equations have no Cholesky factor (here: a NaN state, so every pivot test fails) ends
    with GPMP2MI_TRAJ_NOT_SPD -- gtsam::IndeterminantLinearSystemException in the reference -- on every solver path
    (fused GN kernels, chip-wide levels + split tail of the 2x2-tile path, dense cyclic reduction), and the other
    trajectories of the batch finish exactly as they do without it"""
    from gpmp2_amd import engine as E
    if which.startswith("wam"):
        model, N = g.generateArm("WAMArm"), 24
    elif which.startswith("mobile"):
        model, N = _wide_models()["mobile WAM (dof 10)"], 20
    else:
        model, N = g.generateMobileArm("PR2"), 9
    D = model.dof()
    p = _tree_problem(model, N=N, inter=1, opt="GN")
    if model.kind < 2:                                          # fixed-base arm: joint-space endpoints
        p.start_conf[0, :] = 0.1
        p.end_conf[0, :] = np.linspace(0.3, 0.9, D)
    B = 3
    start, end = np.repeat(p.start_conf, B, 0), np.repeat(p.end_conf, B, 0)
    end[2, -1] += 0.2
    init = np.zeros((B, N + 1, 2 * D))
    for b in range(B):
        for i in range(N + 1):
            init[b, i, :D] = start[b] * (N - i) / N + end[b] * i / N
        init[b, :, D:] = (end[b] - start[b])[None, :] / 3.0
    z = np.zeros((B, D))
    r, s = engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data)
    for opt in ("GN", "LM"):
        {"GN": p.setting.setGaussNewton, "LM": p.setting.setLM}[opt]()
        good = engine.batch_optimize(r, s, p.setting, start, z, end, z, init)
        assert E.TRAJ_NOT_SPD not in list(good["status"])
        broken = init.copy()
        broken[1, N // 2, D - 1] = np.nan
        res = engine.batch_optimize(r, s, p.setting, start, z, end, z, broken)
        if opt == "GN":
            assert res["status"][1] == E.TRAJ_NOT_SPD, list(res["status"])
        else:   # LM takes a failed factorisation as a rejected step: lambda grows to its bound, the state stays
            assert res["status"][1] in (E.TRAJ_MAX_ITER, E.TRAJ_CONVERGED, E.TRAJ_NOT_SPD), list(res["status"])
        for b in (0, 2):
            assert res["status"][b] == good["status"][b] and res["iters"][b] == good["iters"][b]
            np.testing.assert_array_equal(res["traj"][b], good["traj"][b])
