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
