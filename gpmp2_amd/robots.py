"""Robot descriptions: host-side mirror of gpmp2/kinematics (Arm, PointRobot, Pose2MobileArm,
BodySphere, RobotModel) as plain data, plus the model tables of the reference toolbox.

Classes keep the reference's names and constructor argument order so that code written against
the wrapped API (gpmp2.h:95-214) reads the same:
    Arm(dof, a, alpha, d [, base_pose [, theta_bias]])      gpmp2/kinematics/Arm.h:48-59
    BodySphere(link_id, radius, center)                      gpmp2/kinematics/RobotModel.h:20-27
    ArmModel(arm, spheres)                                   gpmp2/kinematics/ArmModel.h
    PointRobot(dof, nr_links), PointRobotModel(pr, spheres)  gpmp2/kinematics/PointRobot.h
    Pose2MobileArm(arm [, base_T_arm])                       gpmp2/kinematics/Pose2MobileArm.h
Model tables: matlab/+gpmp2/generateArm.m:20-117, matlab/+gpmp2/generateMobileArm.m:20-51,
matlab/PointRobot2DFactorGraphExample.m:40-46.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

ROBOT_ARM, ROBOT_POINT, ROBOT_POSE2_MOBILE_BASE, ROBOT_POSE2_MOBILE_ARM = 0, 1, 2, 3
ROBOT_POSE2_MOBILE_2ARMS, ROBOT_POSE2_MOBILE_VETLIN_ARM, ROBOT_POSE2_MOBILE_VETLIN_2ARMS = 4, 5, 6


def pose3(R=None, t=(0.0, 0.0, 0.0)) -> np.ndarray:
    """4x4 homogeneous matrix (stand-in for gtsam.Pose3(Rot3, Point3))."""
    T = np.eye(4)
    if R is not None:
        T[:3, :3] = np.asarray(R, dtype=np.float64)
    T[:3, 3] = np.asarray(t, dtype=np.float64)
    return T


def rot_yaw(yaw: float) -> np.ndarray:
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


@dataclass
class BodySphere:
    link_id: int
    radius: float
    center: Sequence[float]


class Arm:
    def __init__(self, dof, a, alpha, d, base_pose=None, theta_bias=None):
        self._dof = int(dof)
        self.a = np.asarray(a, dtype=np.float64).reshape(-1).copy()
        self.alpha = np.asarray(alpha, dtype=np.float64).reshape(-1).copy()
        self.d = np.asarray(d, dtype=np.float64).reshape(-1).copy()
        self.base_pose = np.eye(4) if base_pose is None else np.asarray(base_pose, dtype=np.float64).copy()
        self.theta_bias = (np.zeros(self._dof) if theta_bias is None
                           else np.asarray(theta_bias, dtype=np.float64).reshape(-1).copy())
        for v in (self.a, self.alpha, self.d, self.theta_bias):
            if v.size != self._dof:
                raise ValueError("[Arm] DH parameter vector dim does not fit dof")

    def dof(self):
        return self._dof

    def nr_links(self):
        return self._dof


class PointRobot:
    def __init__(self, dof=2, nr_links=1):
        if dof != 2 or nr_links != 1:
            raise ValueError("[PointRobot] only the planar single-link point robot is supported")
        self._dof, self._nr_links = dof, nr_links

    def dof(self):
        return self._dof

    def nr_links(self):
        return self._nr_links


class Pose2MobileBase:
    def dof(self):
        return 3

    def nr_links(self):
        return 1


class Pose2MobileArm:
    def __init__(self, arm: Arm, base_T_arm=None):
        self.arm = arm
        self.base_T_arm = np.eye(4) if base_T_arm is None else np.asarray(base_T_arm, dtype=np.float64).copy()

    def dof(self):
        return self.arm.dof() + 3

    def nr_links(self):
        return self.arm.dof() + 1


def _T(x):
    return np.eye(4) if x is None else np.asarray(x, dtype=np.float64).reshape(4, 4).copy()


class Pose2Mobile2Arms:
    """gpmp2::Pose2Mobile2Arms (kinematics/Pose2Mobile2Arms.cpp:18-108): state [x, y, theta, q_arm1, q_arm2];
    links: vehicle base, arm-1 links, arm-2 links"""

    def __init__(self, arm1: Arm, arm2: Arm, base_T_arm1=None, base_T_arm2=None):
        self.arm1, self.arm2 = arm1, arm2
        self.base_T_arm1, self.base_T_arm2 = _T(base_T_arm1), _T(base_T_arm2)

    def dof(self):
        return self.arm1.dof() + self.arm2.dof() + 3

    def nr_links(self):
        return self.arm1.dof() + self.arm2.dof() + 1


class Pose2MobileVetLinArm:
    """gpmp2::Pose2MobileVetLinArm (kinematics/Pose2MobileVetLinArm.cpp:18-108): state
    [x, y, theta, lift, q_arm]; links: vehicle base, torso, arm links"""

    def __init__(self, arm: Arm, base_T_torso=None, torso_T_arm=None, reverse_linact=False):
        self.arm = arm
        self.base_T_torso, self.torso_T_arm = _T(base_T_torso), _T(torso_T_arm)
        self.reverse_linact = bool(reverse_linact)

    def dof(self):
        return self.arm.dof() + 4

    def nr_links(self):
        return self.arm.dof() + 2


class Pose2MobileVetLin2Arms:
    """gpmp2::Pose2MobileVetLin2Arms (kinematics/Pose2MobileVetLin2Arms.cpp:20-114): state
    [x, y, theta, lift, q_arm1, q_arm2]; links: vehicle base, torso, arm-1 links, arm-2 links"""

    def __init__(self, arm1: Arm, arm2: Arm, base_T_torso=None, torso_T_arm1=None, torso_T_arm2=None,
                 reverse_linact=False):
        self.arm1, self.arm2 = arm1, arm2
        self.base_T_torso, self.torso_T_arm1, self.torso_T_arm2 = _T(base_T_torso), _T(torso_T_arm1), _T(torso_T_arm2)
        self.reverse_linact = bool(reverse_linact)

    def dof(self):
        return self.arm1.dof() + self.arm2.dof() + 4

    def nr_links(self):
        return self.arm1.dof() + self.arm2.dof() + 2


class RobotModel:
    """FK model + body spheres -> flat description consumed by the C ABI."""

    def __init__(self, fk_model, body_spheres: List[BodySphere]):
        self.fk = fk_model
        self.spheres = list(body_spheres)
        for s in self.spheres:
            if not (0 <= s.link_id < fk_model.nr_links()):
                raise ValueError("[RobotModel] sphere link id out of range")

    def fk_model(self):
        return self.fk

    def dof(self):
        return self.fk.dof()

    def nr_body_spheres(self):
        return len(self.spheres)

    def sphere_radius(self, i):
        return self.spheres[i].radius

    @property
    def kind(self):
        if isinstance(self.fk, Arm):
            return ROBOT_ARM
        if isinstance(self.fk, PointRobot):
            return ROBOT_POINT
        if isinstance(self.fk, Pose2MobileBase):
            return ROBOT_POSE2_MOBILE_BASE
        if isinstance(self.fk, Pose2MobileArm):
            return ROBOT_POSE2_MOBILE_ARM
        if isinstance(self.fk, Pose2Mobile2Arms):
            return ROBOT_POSE2_MOBILE_2ARMS
        if isinstance(self.fk, Pose2MobileVetLinArm):
            return ROBOT_POSE2_MOBILE_VETLIN_ARM
        if isinstance(self.fk, Pose2MobileVetLin2Arms):
            return ROBOT_POSE2_MOBILE_VETLIN_2ARMS
        raise TypeError("unknown FK model")

    def flat(self):
        """dict of contiguous numpy arrays in the layout of gpmp2mi_robot_desc."""
        fk = self.fk
        eye = np.eye(4)
        arms, base, base2, base3, rev = [], eye, eye, eye, False
        if isinstance(fk, Arm):
            arms, base = [fk], fk.base_pose
        elif isinstance(fk, Pose2MobileArm):
            arms, base = [fk.arm], fk.base_T_arm
        elif isinstance(fk, Pose2Mobile2Arms):
            arms, base, base2 = [fk.arm1, fk.arm2], fk.base_T_arm1, fk.base_T_arm2
        elif isinstance(fk, Pose2MobileVetLinArm):
            arms, base, base2, rev = [fk.arm], fk.base_T_torso, fk.torso_T_arm, fk.reverse_linact
        elif isinstance(fk, Pose2MobileVetLin2Arms):
            arms, base, base2, base3, rev = ([fk.arm1, fk.arm2], fk.base_T_torso, fk.torso_T_arm1, fk.torso_T_arm2,
                                             fk.reverse_linact)
        ad = sum(a.dof() for a in arms)
        z = np.zeros(1)

        def cat(name):
            return np.ascontiguousarray(np.concatenate([np.asarray(getattr(a, name), dtype=np.float64) for a in arms])
                                        if arms else z)

        return dict(
            kind=self.kind, dof=fk.dof(), arm_dof=ad,
            a=cat("a"), alpha=cat("alpha"), d=cat("d"), theta_bias=cat("theta_bias"),
            base_pose=np.ascontiguousarray(base, dtype=np.float64).reshape(16),
            sphere_link=np.ascontiguousarray([s.link_id for s in self.spheres], dtype=np.int32),
            sphere_radius=np.ascontiguousarray([s.radius for s in self.spheres], dtype=np.float64),
            sphere_center=np.ascontiguousarray([list(s.center) for s in self.spheres], dtype=np.float64).reshape(-1),
            arm2_dof=arms[1].dof() if len(arms) == 2 else 0,
            base_pose2=np.ascontiguousarray(base2, dtype=np.float64).reshape(16),
            base_pose3=np.ascontiguousarray(base3, dtype=np.float64).reshape(16),
            reverse_linact=int(rev),
        )


ArmModel = RobotModel
PointRobotModel = RobotModel
Pose2MobileBaseModel = RobotModel
Pose2MobileArmModel = RobotModel
Pose2Mobile2ArmsModel = RobotModel
Pose2MobileVetLinArmModel = RobotModel
Pose2MobileVetLin2ArmsModel = RobotModel


def _spheres(rows):
    return [BodySphere(int(r[0]), float(r[4]), (float(r[1]), float(r[2]), float(r[3]))) for r in rows]


def generateArm(arm_str: str, base_pose=None) -> RobotModel:
    """matlab/+gpmp2/generateArm.m (rows are [link x y z r])."""
    if arm_str == "SimpleTwoLinksArm":
        arm = Arm(2, [0.5, 0.5], [0, 0], [0, 0], base_pose)
        rows = [[0, x, 0, 0, 0.01] for x in (-0.5, -0.4, -0.3, -0.2, -0.1)] + \
               [[1, x, 0, 0, 0.01] for x in (-0.5, -0.4, -0.3, -0.2, -0.1, 0.0)]
    elif arm_str == "SimpleThreeLinksArm":
        arm = Arm(3, [0.5, 0.5, 0.5], [0, 0, 0], [0, 0, 0], base_pose)
        rows = [[0, x, 0, 0, 0.01] for x in (-0.5, -0.4, -0.3, -0.2, -0.1)] + \
               [[1, x, 0, 0, 0.01] for x in (-0.5, -0.4, -0.3, -0.2, -0.1)] + \
               [[2, x, 0, 0, 0.01] for x in (-0.5, -0.4, -0.3, -0.2, -0.1, 0.0)]
    elif arm_str == "WAMArm":
        pi = math.pi
        arm = Arm(7, [0, 0, 0.045, -0.045, 0, 0, 0],
                  [-pi / 2, pi / 2, -pi / 2, pi / 2, -pi / 2, pi / 2, 0],
                  [0, 0, 0.55, 0, 0.3, 0, 0.06], base_pose, [0] * 7)
        rows = [[0, 0.0, 0.0, 0.0, 0.15],
                [1, 0.0, 0.0, 0.2, 0.06], [1, 0.0, 0.0, 0.3, 0.06], [1, 0.0, 0.0, 0.4, 0.06],
                [1, 0.0, 0.0, 0.5, 0.06],
                [2, 0.0, 0.0, 0.0, 0.06],
                [3, 0.0, 0.0, 0.1, 0.06], [3, 0.0, 0.0, 0.2, 0.06], [3, 0.0, 0.0, 0.3, 0.06],
                [5, 0.0, 0.0, 0.1, 0.06],
                [6, 0.1, -0.025, 0.08, 0.04], [6, 0.1, 0.025, 0.08, 0.04], [6, -0.1, 0, 0.08, 0.04],
                [6, 0.15, -0.025, 0.13, 0.04], [6, 0.15, 0.025, 0.13, 0.04], [6, -0.15, 0, 0.13, 0.04]]
    else:
        raise ValueError("No such arm exist")
    return ArmModel(arm, _spheres(rows))


def generateMobileArm(name: str, base_T_arm=None) -> RobotModel:
    """matlab/+gpmp2/generateMobileArm.m:20-51."""
    if name == "PR2":
        return _generate_pr2()
    if name != "SimpleTwoLinksArm":
        raise ValueError("No such mobile arm exist")
    arm = Arm(2, [0.3, 0.3], [0, 0], [0, 0])
    marm = Pose2MobileArm(arm, base_T_arm)
    rows = [[0, -0.1, 0, 0, 0.12], [0, 0.0, 0, 0, 0.12], [0, 0.1, 0, 0, 0.12],
            [1, -0.3, 0, 0, 0.05], [1, -0.2, 0, 0, 0.05], [1, -0.1, 0, 0, 0.05],
            [2, -0.3, 0, 0, 0.05], [2, -0.2, 0, 0, 0.05], [2, -0.1, 0, 0, 0.05], [2, 0.0, 0, 0, 0.05]]
    return Pose2MobileArmModel(marm, _spheres(rows))


def _generate_pr2() -> RobotModel:
    """matlab/+gpmp2/generateMobileArm.m:244-349 'PR2': SE(2) base + vertical lift torso + two 7-joint arms (dof 18),
    65 body spheres (data table restated: link, x, y, z, radius)."""
    hp = 1.5708
    arm = Arm(7, [0.1, 0, 0, 0, 0, 0, 0], [-hp, hp, -hp, hp, -hp, hp, 0], [0, 0, 0.4, 0, 0.321, 0, 0], None,
              [0, hp, 0, 0, 0, 0, 0])
    marm = Pose2MobileVetLin2Arms(arm, arm, pose3(t=(-0.05, 0.0, 0.790675)), pose3(t=(0.0, 0.188, 0.0)),
                                  pose3(t=(0.0, -0.188, 0.0)), False)
    base = [[0, 0.0, 0.0, 0.13, 0.17], [0, 0.23, 0.0, 0.13, 0.17], [0, -0.23, 0.0, 0.13, 0.17], [0, 0.23, 0.23, 0.13, 0.17],
            [0, 0.0, 0.23, 0.13, 0.17], [0, 0.0, -0.23, 0.13, 0.17], [0, 0.23, -0.23, 0.13, 0.17],
            [0, -0.23, -0.23, 0.13, 0.17], [0, -0.23, 0.23, 0.13, 0.17],
            [0, -0.27, 0.0, 0.38, 0.08], [0, -0.27, 0.16, 0.38, 0.08], [0, -0.27, -0.16, 0.38, 0.08],
            [0, -0.27, 0.0, 0.54, 0.08], [0, -0.27, 0.14, 0.54, 0.08], [0, -0.27, -0.14, 0.54, 0.08]]
    torso = [[1, -0.11, 0.0, 0.1, 0.25], [1, -0.09, -0.12, -0.34, 0.2], [1, -0.09, 0.12, -0.34, 0.2],
             [1, -0.02, 0.0, 0.37, 0.17]]

    def one_arm(l0):   # l0 = link id of the arm's first link (2: left, 9: right)
        return [[l0, -0.01, 0.0, 0.0, 0.18],
                [l0 + 2, 0.015, 0.22, -0.0, 0.11], [l0 + 2, 0.035, 0.14, -0.0, 0.08], [l0 + 2, 0.035, 0.0725, -0.0, 0.08],
                [l0 + 2, 0.0, 0.0, -0.0, 0.105],
                [l0 + 4, -0.005, 0.321 - 0.13, -0.0, 0.075], [l0 + 4, 0.01, 0.321 - 0.2, -0.025, 0.055],
                [l0 + 4, 0.01, 0.321 - 0.2, 0.025, 0.055], [l0 + 4, 0.015, 0.321 - 0.265, -0.0275, 0.05],
                [l0 + 4, 0.015, 0.321 - 0.265, 0.0275, 0.05], [l0 + 4, 0.005, 0.321 - 0.32, -0.0225, 0.05],
                [l0 + 4, 0.005, 0.321 - 0.32, 0.0225, 0.05],
                [l0 + 6, 0, -0.0175, 0.0725, 0.04], [l0 + 6, 0, 0.0175, 0.0725, 0.04], [l0 + 6, 0, 0, 0.0925, 0.04],
                [l0 + 6, 0, 0.036, 0.11, 0.04], [l0 + 6, 0, 0.027, 0.155, 0.035], [l0 + 6, 0, 0.009, 0.18, 0.03],
                [l0 + 6, 0, 0.0095, 0.205, 0.02],
                [l0 + 6, 0, -0.036, 0.11, 0.04], [l0 + 6, 0, -0.027, 0.155, 0.035], [l0 + 6, 0, -0.009, 0.18, 0.03],
                [l0 + 6, 0, -0.0095, 0.205, 0.02]]

    rows = base + torso + one_arm(2) + one_arm(9)
    assert len(rows) == 65
    return RobotModel(marm, _spheres(rows))


def generatePointRobot(radius=1.5) -> RobotModel:
    """matlab/PointRobot2DFactorGraphExample.m:40-46: PointRobot(2,1), one sphere r = 1.5."""
    return PointRobotModel(PointRobot(2, 1), [BodySphere(0, radius, (0.0, 0.0, 0.0))])
