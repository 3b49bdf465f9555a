"""Extra factors carried by a plan as data (gpmp2mi_graph_opts): GoalFactorArm in place of the end-conf prior,
GaussianPriorWorkspace{Orientation,Pose}Arm on ranges of states, SelfCollisionArm pairs -- the hand-built graphs of
matlab/Arm3GoalReachExample.m:95-110 and matlab/WAMWorkspaceConstraintsExample.m:85-105 solved on the GPU and checked
against the oracle's factor-list restatement (normal equations, graph error, whole solves).  The reference holds
no solve-level fixture for these graphs: parity unpinned beyond the factor-level known answers
(tests/test_oracle_known_answers.py pins the factors themselves)."""
import copy

import numpy as np
import pytest

from gpmp2_amd import problems
from parity_bound import check_contract

pytestmark = pytest.mark.gpu


def _handles(engine, oracle, p):
    return (engine.robot(p.model), engine.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data),
            oracle.robot(p.model), oracle.sdf(p.sdf_origin, p.sdf_cell, p.sdf_data))


def _args(p):
    return p.start_conf, p.start_vel, p.end_conf, p.end_vel


def _check_linearize(engine, oracle, p, traj):
    r, s, ro, so = _handles(engine, oracle, p)
    a = engine.linearize(r, s, p.setting, *_args(p), traj)
    b = oracle.linearize(ro, so, p.setting, *_args(p), traj)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_allclose(x, y, atol=1e-9 * np.abs(y).max())
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    np.testing.assert_allclose(engine.graph_error(r, s, p.setting, *_args(p), traj), b[3], rtol=1e-9)


def _check_solve(engine, oracle, p):
    """the SURVEY 8(d) contract (tests/parity_bound.py): identical control flow, trajectory 1e-6 -- a trajectory above it
    only inside the oracle's own 2-ulp sensitivity --, final error = the oracle's E at the returned values to 1e-8"""
    return check_contract(engine, oracle, p, label=p.name, final_error_rtol=1e-8)["res"]


def test_arm3_goal_reach_example(engine, oracle):
    p = problems.arm3_goal_reach()
    rng = np.random.default_rng(3)
    _check_linearize(engine, oracle, p, p.init + 0.1 * rng.normal(size=p.init.shape))
    res = _check_solve(engine, oracle, p)
    assert res["iters"][0] >= 2
    # the end effector reaches the goal point although no end configuration was given
    poses, _ = engine.forward_kinematics(engine.robot(p.model), res["traj"][0, -1, :3])
    np.testing.assert_allclose(poses[0, 2, :3, 3], [0.0, 1.1, 0.0], atol=1e-3)
    # without the goal factor the same plan stays at the (zero) end-conf prior: the factor is really in the solve
    q = copy.deepcopy(p)
    q.setting.workspace_factors = []
    q.setting.end_conf_prior_off = False
    r, s, _, _ = _handles(engine, oracle, q)
    other = engine.batch_optimize(r, s, q.setting, *_args(q), q.init)
    assert np.abs(other["traj"][0, -1, :3]).max() < 1e-3


@pytest.mark.parametrize("opt", ["LM", "GN", "DOGLEG"])
def test_wam_workspace_constraints_example(engine, oracle, opt):
    eng_fk = lambda model, q: engine.forward_kinematics(engine.robot(model), q)[0][0, 6]
    p = problems.wam_workspace_constraints(eng_fk, sdf="40", B=3)
    {"LM": p.setting.setLM, "GN": p.setting.setGaussNewton, "DOGLEG": p.setting.setDogleg}[opt]()
    rng = np.random.default_rng(4)
    _check_linearize(engine, oracle, p, p.init + 0.05 * rng.normal(size=p.init.shape))
    res = _check_solve(engine, oracle, p)
    if opt != "LM":      # plain GN / Dogleg are parity cases; the script's optimizer (LM, lambda0 = 1000) is the one that
        return           # has to reach the goal
    # the end-effector pose prior (sigma 1e-4) is met, the orientation prior (1e-2) keeps the tool level on the way
    r = engine.robot(p.model)
    des = p.setting.workspace_factors[1]["des_pose"]
    poses, _ = engine.forward_kinematics(r, res["traj"][0, -1, :7])
    np.testing.assert_allclose(poses[0, 6], des, atol=5e-3)
    mid, _ = engine.forward_kinematics(r, res["traj"][0, 5, :7])
    assert np.abs(mid[0, 6, :3, :3] - p.setting.workspace_factors[0]["des_pose"][:3, :3]).max() < 0.2


def test_wam_workspace_constraints_full_size_field(engine, oracle):
    """the same graph in the 200^3 field of the headline workload (the script's 300^3 field is the uncropped one)"""
    eng_fk = lambda model, q: engine.forward_kinematics(engine.robot(model), q)[0][0, 6]
    _check_solve(engine, oracle, problems.wam_workspace_constraints(eng_fk, sdf="synth200", B=2))


def test_self_collision_pairs_in_a_plan(engine, oracle):
    """SelfCollisionArm rows (sphere A, sphere B, epsilon, sigma) on every support state of a WAM plan"""
    p = problems.wam_restarts(B=3, total_step=10, obs_check_inter=2, sdf="40", opt="LM")
    p.setting.self_collision = np.array([[0, 9, 0.6, 0.05], [1, 15, 0.5, 0.1], [5, 12, 0.4, 0.05]])
    p.setting.self_collision_states = (1, 9)
    rng = np.random.default_rng(6)
    traj = p.init + 0.2 * rng.normal(size=p.init.shape)
    ro = oracle.robot(p.model)
    e, _ = oracle.self_collision_factor(ro, p.setting.self_collision, traj[0, :, :7])
    assert (e > 0).sum() >= 3                                  # the pairs are active on this trajectory
    _check_linearize(engine, oracle, p, traj)
    _check_solve(engine, oracle, p)


@pytest.mark.parametrize("opt", ["GN", "LM"])
def test_workspace_and_self_collision_factors_on_a_mobile_arm(engine, oracle, opt):
    """the extra factors are documented for <Arm> robots (kinematics/GaussianPriorWorkspacePose.h:53-70 is templated on
    the robot; the reference instantiates it for Arm only) but a plan accepts them for Pose2 mobile manipulators too:
    their Jacobians then go through the Pose2 chart of the base like every other factor of the Lie path.  Pinned
    against the oracle's factor list (parity unpinned by the reference: no fixture exists)."""
    p = problems.mobile_arm_config5()
    {"GN": p.setting.setGaussNewton, "LM": p.setting.setLM}[opt]()
    p.setting.set_obs_check_inter(2)
    p.setting.set_max_iter(30)
    p.setting.setOptimizationNoIncrase(True)
    N = p.setting.total_step
    des = np.eye(4)
    des[:3, 3] = [0.9, 0.55, 0.0]
    p.setting.add_workspace_prior(0, 2, des, 0.01, N // 2)             # the arm tip passes a way point half-way
    p.setting.self_collision = np.array([[0, 9, 0.3, 0.05], [2, 8, 0.25, 0.1]])
    p.setting.self_collision_states = (1, N - 1)
    rng = np.random.default_rng(8)
    _check_linearize(engine, oracle, p, p.init + 0.1 * rng.normal(size=p.init.shape))
    _check_solve(engine, oracle, p)
