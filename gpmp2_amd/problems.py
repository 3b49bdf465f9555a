"""Deterministic problem generators for the BASELINE.json configs (SURVEY.md section 8d and
appendix C).  Pure numpy; used identically by tests/, bench.py and __graft_entry__.smoke()."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import datasets, robots
from .settings import TrajOptimizerSetting
from .trajutils import initArmTrajStraightLine


@dataclass
class Problem:
    name: str
    model: robots.RobotModel
    sdf_origin: list
    sdf_cell: float
    sdf_data: np.ndarray        # [nz][ny][nx] or [ny][nx]
    setting: TrajOptimizerSetting
    start_conf: np.ndarray      # [B][D]
    start_vel: np.ndarray
    end_conf: np.ndarray
    end_vel: np.ndarray
    init: np.ndarray            # [B][N+1][2D]

    @property
    def B(self):
        return self.init.shape[0]


_SDF_CACHE = {}


def synth200_sdf():
    """'Synth200' (SURVEY.md 8d): 200^3 central crop of WAMDeskDataset, origin (-1,-1,-1), cell 0.01,
    field = cell * (EDT(free) - EDT(occupied)), returned as [z][y][x] fp64 (64 MB)."""
    if "synth200" not in _SDF_CACHE:
        d = datasets.generate3Ddataset("Synth200")
        f = datasets.signedDistanceField3D(d.map, d.cell_size)
        _SDF_CACHE["synth200"] = ([d.origin_x, d.origin_y, d.origin_z], d.cell_size, datasets.sdf3_zyx(f))
    return _SDF_CACHE["synth200"]


def small3d_sdf(n=40):
    """Down-scaled desk scene (n^3 voxels, cell 2/n, same extent [-1,1]^3) for fast CPU tests."""
    key = ("small", n)
    if key not in _SDF_CACHE:
        m = np.zeros((n, n, n))
        k = n / 200.0
        for pos, size in datasets._WAM_DESK:
            lo = [int(round((pos[a] - 50 - (size[a] - 1) // 2) * k)) for a in range(3)]
            hi = [int(round((pos[a] - 50 + (size[a] - 1) // 2) * k)) + 1 for a in range(3)]
            sl = tuple(slice(max(lo[a], 0), min(max(hi[a], lo[a] + 1), n)) for a in range(3))
            m[sl] = 1.0
        cell = 2.0 / n
        f = datasets.signedDistanceField3D(m, cell)
        _SDF_CACHE[key] = ([-1.0, -1.0, -1.0], cell, datasets.sdf3_zyx(f))
    return _SDF_CACHE[key]


WAM_START = np.array([-0.8, -1.70, 1.64, 1.29, 1.1, -0.106, 2.2])
WAM_END = np.array([-0.0, 0.94, 0, 1.6, 0, -0.919, 1.55])


def wam_setting(total_step=100, obs_check_inter=5, opt="GN", max_iter=50):
    """matlab/WAMFactorGraphExample.m:43-60 with BASELINE's N=100, I=5."""
    s = TrajOptimizerSetting(7)
    s.set_total_step(total_step)
    s.set_total_time(2.0)
    s.set_obs_check_inter(obs_check_inter)
    s.set_cost_sigma(0.02)
    s.set_epsilon(0.2)
    s.set_conf_prior_model(0.0001)
    s.set_vel_prior_model(0.0001)
    s.set_Qc_model(np.eye(7))
    s.set_max_iter(max_iter)
    s.set_rel_thresh(1e-2)
    {"GN": s.setGaussNewton, "LM": s.setLM, "DOGLEG": s.setDogleg}[opt.upper()]()
    return s


def wam_restarts(B=64, total_step=100, obs_check_inter=5, opt="GN", sdf="synth200", max_iter=50):
    """BASELINE config 3: B random-init restarts of the WAM problem.  restart 0 = straight line;
    restart b >= 1 = straight line + A_b sin(pi i / N), A_b ~ N(0, 0.5^2 I_7) from
    default_rng(1234 + b); velocities unchanged."""
    model = robots.generateArm("WAMArm")
    origin, cell, data = synth200_sdf() if sdf == "synth200" else small3d_sdf(int(sdf))
    s = wam_setting(total_step, obs_check_inter, opt, max_iter)
    N = total_step
    base = initArmTrajStraightLine(WAM_START, WAM_END, N)
    init = np.repeat(base[None], B, axis=0)
    bump = np.sin(math.pi * np.arange(N + 1) / N)
    for b in range(1, B):
        A = np.random.default_rng(1234 + b).normal(0.0, 0.5, size=7)
        init[b, :, :7] += bump[:, None] * A[None, :]
    z = np.zeros((B, 7))
    return Problem("wam_restarts", model, origin, cell, data, s, np.repeat(WAM_START[None], B, 0), z.copy(),
                   np.repeat(WAM_END[None], B, 0), z.copy(), init)


def wam_windows(solution, B=1024, total_step=100, obs_check_inter=5, fixed_iterations=3, sdf="synth200"):
    """BASELINE config 4: B receding-horizon windows warm-started from `solution` ([N+1][14], the
    converged restart-0 trajectory): start = x*_k + N(0, 0.05^2), k = w mod (N+1);
    goal = end_conf + N(0, 0.1^2); rng(4321 + w); fixed GN iteration budget."""
    p = wam_restarts(1, total_step, obs_check_inter, "GN", sdf)
    p.name = "wam_windows"
    N = total_step
    sc, ec = np.zeros((B, 7)), np.zeros((B, 7))
    for w in range(B):
        rng = np.random.default_rng(4321 + w)
        sc[w] = solution[w % (N + 1), :7] + rng.normal(0.0, 0.05, size=7)
        ec[w] = WAM_END + rng.normal(0.0, 0.1, size=7)
    p.start_conf, p.end_conf = sc, ec
    p.start_vel, p.end_vel = np.zeros((B, 7)), np.zeros((B, 7))
    p.init = np.repeat(np.asarray(solution)[None], B, axis=0).copy()
    p.setting.fixed_iterations = fixed_iterations
    return p


def arm3_planner(obs_check_inter=3):
    """BASELINE config 2: matlab/Arm3PlannerExample.m (3-link planar arm, 2-D SDF, Dogleg,
    joint + velocity limits) with I = 3 as BASELINE asks."""
    model = robots.generateArm("SimpleThreeLinksArm")
    d = datasets.generate2Ddataset("TwoObstaclesDataset")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    s = TrajOptimizerSetting(3)
    s.set_total_step(50)
    s.set_total_time(5.0)
    s.set_epsilon(0.2)
    s.set_cost_sigma(0.1)
    s.set_obs_check_inter(obs_check_inter)
    s.set_conf_prior_model(0.0001)
    s.set_vel_prior_model(0.0001)
    s.set_Qc_model(np.eye(3))
    s.set_flag_pos_limit(True)
    s.set_flag_vel_limit(True)
    s.set_joint_pos_limits_down([-1000.0, -1000.0, 0.0])
    s.set_joint_pos_limits_up([1000.0, 1000.0, 0.0])
    s.set_pos_limit_thresh([0.001, 0.001, 0.001])
    s.set_pos_limit_model([0.001, 0.001, 0.001])
    s.set_vel_limits([1.0, 1.0, 1.0])
    s.set_vel_limit_thresh([0.01, 0.01, 0.01])
    s.set_vel_limit_model([0.1, 0.1, 0.1])
    s.setDogleg()
    start, end = np.zeros(3), np.array([0.9, math.pi / 2 - 0.9, 0.0])
    init = initArmTrajStraightLine(start, end, 50)[None]
    z = np.zeros((1, 3))
    return Problem("arm3_planner", model, [d.origin_x, d.origin_y], d.cell_size, field, s, start[None], z.copy(),
                   end[None], z.copy(), init)


def point_robot_2d():
    """BASELINE config 1: matlab/PointRobot2DFactorGraphExample.m (hand-built graph, GaussNewton,
    obstacle factors only for i > 0, init velocity avg_vel = (end/N)/delta_t)."""
    model = robots.generatePointRobot(1.5)
    d = datasets.generate2Ddataset("MultiObstacleDataset")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    N = 10
    s = TrajOptimizerSetting(2)
    s.set_total_step(N)
    s.set_total_time(5.0)
    s.set_obs_check_inter(4)
    s.set_cost_sigma(0.3)
    s.set_epsilon(2.0)
    s.set_conf_prior_model(0.0001)
    s.set_vel_prior_model(0.0001)
    s.set_Qc_model(np.eye(2))
    s.setGaussNewton()
    s.set_max_iter(100)          # gtsam GaussNewtonParams defaults (the script calls GTSAM directly)
    s.set_rel_thresh(1e-5)
    s.obs_skip_first_state = True
    start, end = np.array([-15.0, -8.0]), np.array([17.0, 14.0])
    dt = 5.0 / N
    avg_vel = (end / N) / dt
    init = np.zeros((1, N + 1, 4))
    for i in range(N + 1):
        init[0, i, :2] = start * (N - i) / N + end * i / N
        init[0, i, 2:] = avg_vel
    z = np.zeros((1, 2))
    return Problem("point_robot_2d", model, [d.origin_x, d.origin_y], d.cell_size, field, s, start[None], z.copy(),
                   end[None], z.copy(), init)


def mobile_arm_config5():
    """BASELINE config 5: matlab/MobileArm2FactorGraphExample.m -- SE(2) base + 2-link arm (dof 5),
    N = 50, no GP interpolation, 2-D MobileMap1 SDF, planar obstacle factor on every state,
    VehicleDynamicsFactorPose2Vector (sigma 1e-3) on every state, GaussianProcessPriorPose2Vector,
    Dogleg with GTSAM defaults (deltaInitial 1.0, 100 iterations, relative tol 1e-5, no rollback)."""
    model = robots.generateMobileArm("SimpleTwoLinksArm")
    d = datasets.generate2Ddataset("MobileMap1")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    N = 50
    s = TrajOptimizerSetting(5)
    s.set_total_step(N)
    s.set_total_time(5.0)
    s.set_obs_check_inter(0)
    s.set_cost_sigma(0.1)
    s.set_epsilon(0.1)
    s.set_conf_prior_model(0.0001)
    s.set_vel_prior_model(0.0001)
    s.set_Qc_model(np.eye(5))
    s.setDogleg()
    s.dogleg_delta_initial = 1.0
    s.set_max_iter(100)
    s.set_rel_thresh(1e-5)
    s.setOptimizationNoIncrase(False)
    s.vehicle_dynamics_sigma = 0.001
    start = np.array([-1.0, 0.0, math.pi / 2, 0.0, 0.0])
    end = np.array([1.0, 0.0, math.pi / 2, 0.0, 0.0])
    dt = 5.0 / N
    avg_vel = np.concatenate([end[:3] - start[:3], (end[3:] / N)]) / dt
    init = np.zeros((1, N + 1, 10))
    for i in range(N + 1):
        init[0, i, :5] = start * (N - i) / N + end * i / N
        init[0, i, 5:] = avg_vel
    z = np.zeros((1, 5))
    return Problem("mobile_arm_config5", model, [d.origin_x, d.origin_y], d.cell_size, field, s, start[None], z.copy(),
                   end[None], z.copy(), init)


def arm3_goal_reach():
    """matlab/Arm3GoalReachExample.m: 3-link planar arm, TwoObstaclesDataset, N = 10, 4 interpolated checks,
    hand-built graph: start priors, GoalFactorArm on x_N INSTEAD of the end-conf prior (:105-109), end-velocity
    prior, GP priors, planar obstacle factors for i > 0; Dogleg with GTSAM defaults (:151-158)."""
    model = robots.generateArm("SimpleThreeLinksArm")
    d = datasets.generate2Ddataset("TwoObstaclesDataset")
    field = datasets.signedDistanceField2D(d.map, d.cell_size)
    N = 10
    s = TrajOptimizerSetting(3)
    s.set_total_step(N)
    s.set_total_time(5.0)
    s.set_obs_check_inter(4)
    s.set_cost_sigma(0.1)
    s.set_epsilon(0.1)
    s.set_conf_prior_model(0.0001)
    s.set_vel_prior_model(0.0001)
    s.set_Qc_model(np.eye(3))
    s.setDogleg()
    s.dogleg_delta_initial = 1.0           # gtsam DoglegParams defaults (the script calls GTSAM directly)
    s.set_max_iter(100)
    s.set_rel_thresh(1e-5)
    s.setOptimizationNoIncrase(False)
    s.obs_skip_first_state = True
    s.end_conf_prior_off = True
    s.add_goal_factor_arm(link=2, dest_point=(0.0, 1.1, 0.0), sigma=0.0001)
    start = np.zeros(3)
    init = np.zeros((1, N + 1, 6))        # end_conf_init = 0, avg_vel = 0 (:70-75)
    z = np.zeros((1, 3))
    return Problem("arm3_goal_reach", model, [d.origin_x, d.origin_y], d.cell_size, field, s, start[None], z.copy(),
                   np.zeros((1, 3)), z.copy(), init)


def wam_workspace_constraints(fk_pose, sdf="synth200", B=1):
    """matlab/WAMWorkspaceConstraintsExample.m: WAM, N = 10, 5 interpolated checks, Qc = 0.1 I, obstacle sigma 0.005,
    epsilon 0.15; hand-built graph: start priors, GaussianPriorWorkspaceOrientationArm (sigma 1e-2, the start pose's
    end-effector orientation) on every interior state, GaussianPriorWorkspacePoseArm (sigma 1e-4, the end
    configuration's end-effector pose) on x_N instead of the end-conf prior, end-velocity prior; LM, lambda0 = 1000,
    GTSAM defaults (:131-140).  fk_pose(model, conf) -> 4x4 pose of the last link (the caller's FK: the engine's on
    the GPU, the oracle's in CPU tests).  The script's 300^3 WAMDeskDataset field is replaced by `sdf`."""
    model = robots.generateArm("WAMArm")
    origin, cell, data = synth200_sdf() if sdf == "synth200" else small3d_sdf(int(sdf))
    N = 10
    start = np.array([-0.0, 0.94, 0, 1.6, 0, -0.919, 1.55])
    end = np.array([-0.8, -1.70, 1.64, 1.29, 1.1, -0.106, 2.2])
    s = TrajOptimizerSetting(7)
    s.set_total_step(N)
    s.set_total_time(2.0)
    s.set_obs_check_inter(5)
    s.set_cost_sigma(0.005)
    s.set_epsilon(0.15)
    s.set_conf_prior_model(1e-4)
    s.set_vel_prior_model(1e-4)
    s.set_Qc_model(0.1 * np.eye(7))
    s.setLM()
    s.lm_lambda_initial = 1000.0
    s.set_max_iter(100)
    s.set_rel_thresh(1e-5)
    s.setOptimizationNoIncrase(False)
    s.obs_skip_first_state = True
    s.end_conf_prior_off = True
    traj_orien = np.eye(4)
    traj_orien[:3, :3] = fk_pose(model, start)[:3, :3]
    s.add_workspace_prior(1, 6, traj_orien, 1e-2, 1, N - 1)
    s.add_workspace_prior(2, 6, fk_pose(model, end), 1e-4, N)
    base = initArmTrajStraightLine(start, start, N)      # the script starts from the start configuration everywhere
    init = np.repeat(base[None], B, axis=0)
    for b in range(1, B):
        init[b, :, :7] += 0.05 * np.random.default_rng(99 + b).normal(size=(N + 1, 7)) * np.sin(math.pi * np.arange(N + 1) / N)[:, None]
    z = np.zeros((B, 7))
    return Problem("wam_workspace_constraints", model, origin, cell, data, s, np.repeat(start[None], B, 0), z.copy(),
                   np.repeat(end[None], B, 0), z.copy(), init)
