"""Host-side trajectory utilities (gpmp2/planner/TrajUtils.cpp) on flat [N+1][2D] arrays."""
from __future__ import annotations

import numpy as np


def initArmTrajStraightLine(init_conf, end_conf, total_step: int) -> np.ndarray:
    """gpmp2::initArmTrajStraightLine gpmp2/planner/TrajUtils.cpp:25-50 -> [N+1][2D].
    NB the velocity is (end - init) / total_step, NOT divided by delta_t (TrajUtils.cpp:45)."""
    a = np.asarray(init_conf, dtype=np.float64).reshape(-1)
    b = np.asarray(end_conf, dtype=np.float64).reshape(-1)
    D, N = a.size, int(total_step)
    out = np.zeros((N + 1, 2 * D))
    for i in range(N + 1):
        if i == 0:
            out[i, :D] = a
        elif i == N:
            out[i, :D] = b
        else:
            r = float(i) / float(N)
            out[i, :D] = r * b + (1.0 - r) * a
    out[:, D:] = (b - a) / float(N)
    return out


def values_from_traj(traj: np.ndarray):
    """flat trajectory -> {('x', i): conf, ('v', i): vel}, the gtsam::Values key convention
    (Symbol('x', i) / Symbol('v', i), gpmp2/planner/BatchTrajOptimizer.h:39-41)."""
    D = traj.shape[1] // 2
    out = {}
    for i in range(traj.shape[0]):
        out[("x", i)] = traj[i, :D].copy()
        out[("v", i)] = traj[i, D:].copy()
    return out


def traj_from_values(values, total_step: int) -> np.ndarray:
    D = np.asarray(values[("x", 0)]).size
    out = np.zeros((total_step + 1, 2 * D))
    for i in range(total_step + 1):
        out[i, :D] = values[("x", i)]
        out[i, D:] = values[("v", i)]
    return out


# ---- Pose2 (gtsam::Pose2 semantics; host side of the init utilities only) ------------------
def _wrap(theta):
    return float(np.arctan2(np.sin(theta), np.cos(theta)))


def pose2_between(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    dx, dy = b[0] - a[0], b[1] - a[1]
    return np.array([c * dx + s * dy, -s * dx + c * dy, _wrap(b[2] - a[2])])


def pose2_compose(a, b):
    c, s = np.cos(a[2]), np.sin(a[2])
    return np.array([a[0] + c * b[0] - s * b[1], a[1] + s * b[0] + c * b[1], _wrap(a[2] + b[2])])


def pose2_logmap(p):
    w = _wrap(p[2])
    if abs(w) < 1e-10:
        return np.array([p[0], p[1], w])
    c, s = np.cos(p[2]), np.sin(p[2])
    det = (c - 1.0) ** 2 + s * s
    ux, uy = c * p[0] + s * p[1] - p[0], -s * p[0] + c * p[1] - p[1]
    return np.array([(w / det) * -uy, (w / det) * ux, w])


def pose2_expmap(v):
    w = v[2]
    if abs(w) < 1e-10:
        return np.array([v[0], v[1], v[2]])
    c, s = np.cos(w), np.sin(w)
    ox, oy = -v[1], v[0]
    rx, ry = c * ox - s * oy, s * ox + c * oy
    return np.array([(ox - rx) / w, (oy - ry) / w, _wrap(w)])


def pose2_interpolate(a, b, t):
    """gtsam::interpolate<Pose2>(X, Y, t) = X * Expmap(t * Logmap(between(X, Y)))"""
    return pose2_compose(a, pose2_expmap(t * pose2_logmap(pose2_between(a, b))))


def initPose2VectorTrajStraightLine(init_pose, init_conf, end_pose, end_conf, total_step: int) -> np.ndarray:
    """gpmp2::initPose2VectorTrajStraightLine gpmp2/planner/TrajUtils.cpp:53-73 -> [N+1][2D] with
    states [x, y, theta, q...]; the velocity is the plain coordinate difference / total_step."""
    p0, p1 = np.asarray(init_pose, dtype=np.float64).reshape(3), np.asarray(end_pose, dtype=np.float64).reshape(3)
    q0, q1 = np.asarray(init_conf, dtype=np.float64).reshape(-1), np.asarray(end_conf, dtype=np.float64).reshape(-1)
    D, N = 3 + q0.size, int(total_step)
    out = np.zeros((N + 1, 2 * D))
    avg_vel = np.concatenate([p1 - p0, q1 - q0]) / float(N)
    for i in range(N + 1):
        r = float(i) / float(N)
        out[i, :3] = pose2_interpolate(p0, p1, r)
        out[i, 3:D] = (1.0 - r) * q0 + r * q1
        out[i, D:] = avg_vel
    return out


def initPose2TrajStraightLine(init_pose, end_pose, total_step: int) -> np.ndarray:
    """gpmp2::initPose2TrajStraightLine gpmp2/planner/TrajUtils.cpp:76-93"""
    return initPose2VectorTrajStraightLine(init_pose, [], end_pose, [], total_step)


def _interpolate(opt_values, Qc_model, delta_t, inter_step, start_index, end_index, lie):
    from .planner import _eng
    as_values = isinstance(opt_values, dict)
    if as_values:
        total_step = max(k[1] for k in opt_values if k[0] == "x")
        traj = traj_from_values(opt_values, total_step)
    else:
        traj = np.ascontiguousarray(opt_values, dtype=np.float64)
    D = traj.shape[-1] // 2
    out = _eng().interpolate_traj(D, lie, Qc_model, delta_t, inter_step, traj[None], start_index, end_index)[0]
    return values_from_traj(out) if as_values else out


def interpolateArmTraj(opt_values, Qc_model, delta_t, inter_step, start_index=0, end_index=None):
    """gpmp2::interpolateArmTraj, both overloads (gpmp2/planner/TrajUtils.cpp:96-197); runs on the GPU"""
    return _interpolate(opt_values, Qc_model, delta_t, inter_step, start_index, end_index, False)


def interpolatePose2MobileArmTraj(opt_values, Qc_model, delta_t, inter_step, start_index, end_index):
    """gpmp2::interpolatePose2MobileArmTraj (gpmp2/planner/TrajUtils.cpp:200-236); runs on the GPU"""
    return _interpolate(opt_values, Qc_model, delta_t, inter_step, start_index, end_index, True)


def interpolatePose2Traj(opt_values, Qc_model, delta_t, inter_step, start_index, end_index):
    """gpmp2::interpolatePose2Traj (gpmp2/planner/TrajUtils.cpp:239-275); runs on the GPU"""
    return _interpolate(opt_values, Qc_model, delta_t, inter_step, start_index, end_index, True)
