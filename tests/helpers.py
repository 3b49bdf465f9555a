"""Shared helpers for tests: literal parsing for golden fixtures, model builders, numeric Jacobians,
and the BASELINE problem generators (kept in gpmp2_amd.problems so bench.py uses the same)."""
from __future__ import annotations

import math

import numpy as np

import gpmp2_amd as g


def num(x):
    """golden literals may be strings such as "pi/4" or "pi*10"."""
    if isinstance(x, str):
        return float(eval(x, {"__builtins__": {}}, {"pi": math.pi}))
    return float(x)


def vec(xs):
    return np.array([num(x) for x in xs], dtype=np.float64)


def arm_from_golden(d):
    base = g.pose3(t=d["arm"]["base_xyz"])
    arm = g.Arm(d["arm"]["dof"], d["arm"]["a"], d["arm"]["alpha"], d["arm"]["d"], base)
    sph = [g.BodySphere(int(s[0]), s[1], (s[2], s[3], s[4])) for s in d["spheres"]]
    return g.ArmModel(arm, sph)


def numeric_jacobian(f, x, h=1e-6):
    """central difference, the same scheme as gtsam::numericalDerivative11 at delta = h."""
    x = np.asarray(x, dtype=np.float64)
    f0 = np.asarray(f(x))
    J = np.zeros(f0.shape + (x.size,))
    for k in range(x.size):
        dx = np.zeros_like(x)
        dx[k] = h
        J[..., k] = (np.asarray(f(x + dx)) - np.asarray(f(x - dx))) / (2 * h)
    return J


def sdf_to_err(sdf, eps):
    e = eps - np.asarray(sdf, dtype=np.float64)
    return np.where(e > 0.0, e, 0.0)


def rot_ypr(y, p, r):
    """gtsam::Rot3::Ypr(y, p, r) = Rz(y) Ry(p) Rx(r)"""
    cy, sy, cp, sp, cr, sr = math.cos(y), math.sin(y), math.cos(p), math.sin(p), math.cos(r), math.sin(r)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def _gold_pose(x):
    if x == "identity":
        return np.eye(4)
    return g.pose3(g.rot_yaw(num(x["yaw"])), x["xyz"])


def tree_robot_from_golden(d, key, random_bases=False, spheres=True):
    """Pose2Mobile2Arms / Pose2MobileVetLinArm / Pose2MobileVetLin2Arms of the reference's FK tests
    (two planar 2-link arms), with a few body spheres on every link for the sphere-level checks."""
    arm = g.Arm(2, d["a"], d["alpha"], d["d"])
    if key == "pose2_mobile_2arms":
        fk = g.Pose2Mobile2Arms(arm, arm, _gold_pose(d["base1"]), _gold_pose(d["base2"]))
    elif key == "pose2_mobile_vetlin_arm":
        fk = g.Pose2MobileVetLinArm(arm, _gold_pose(d["base_T_torso"]), _gold_pose(d["torso_T_arm"]), d["reverse_linact"])
    else:
        if random_bases:
            rd = d["random"]
            T = [g.pose3(rot_ypr(*rd[k + "_ypr"]), rd[k + "_xyz"]) for k in ("torso", "base1", "base2")]
            fk = g.Pose2MobileVetLin2Arms(arm, arm, T[0], T[1], T[2], False)
        else:
            fk = g.Pose2MobileVetLin2Arms(arm, arm, _gold_pose(d["base_T_torso"]), _gold_pose(d["torso_T_arm1"]),
                                          _gold_pose(d["torso_T_arm2"]), d["reverse_linact"])
    sph = []
    if spheres:
        for l in range(fk.nr_links()):
            sph.append(g.BodySphere(l, 0.1, (-0.3 + 0.1 * l, 0.1, 0.05 * l)))
            if l % 2 == 0:
                sph.append(g.BodySphere(l, 0.15, (0.0, 0.0, 0.0)))
    return g.ArmModel(fk, sph)   # every *Model name is the same RobotModel class
