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
