"""Reference-named factor classes over the HIP factor-level entry points (gpmp2.h:60-530).

Constructor arguments follow the wrapped classes of the reference (keys first, then the model objects);
noise models are plain sigmas / covariance matrices since GTSAM is not part of this package.
``evaluateError(x...)`` returns the UNWHITENED error like the reference; ``evaluateError(x..., jacobians=True)``
additionally returns the Jacobians (the ``boost::optional<Matrix&> H`` arguments of the C++ signatures).
Every evaluation runs on the GPU (one call = a batch of one); use ``gpmp2_amd.engine.Engine`` directly for
large batches.
"""
from __future__ import annotations

import numpy as np

from .planner import _eng, _robot_handle


def _one(x):
    return np.asarray(x, dtype=np.float64).reshape(1, -1)


class _Keyed:
    def keys(self):
        return list(self._keys)


class ObstacleSDFFactorArm(_Keyed):
    """gpmp2::ObstacleSDFFactorArm  gpmp2/obstacle/ObstacleSDFFactor.h:27-100, -inl.h:18-56"""

    def __init__(self, poseKey, arm, sdf, cost_sigma, epsilon):
        self._keys, self.arm_, self.sdf_, self.cost_sigma_, self.epsilon_ = (poseKey,), arm, sdf, cost_sigma, epsilon

    def evaluateError(self, conf, jacobians=False):
        err, H = _eng().obstacle_factor(_robot_handle(self.arm_), self.sdf_.handle(), self.epsilon_, _one(conf))
        return (err[0], H[0]) if jacobians else err[0]


class ObstaclePlanarSDFFactorArm(ObstacleSDFFactorArm):
    """gpmp2::ObstaclePlanarSDFFactorArm  gpmp2/obstacle/ObstaclePlanarSDFFactor.h:27-98"""


ObstaclePlanarSDFFactorPointRobot = ObstaclePlanarSDFFactorArm
ObstaclePlanarSDFFactorPose2MobileArm = ObstaclePlanarSDFFactorArm
ObstacleSDFFactorPose2MobileArm = ObstacleSDFFactorArm
ObstacleSDFFactorPose2Mobile2Arms = ObstacleSDFFactorArm
ObstacleSDFFactorPose2MobileVetLinArm = ObstacleSDFFactorArm
ObstacleSDFFactorPose2MobileVetLin2Arms = ObstacleSDFFactorArm


class ObstacleSDFFactorGPArm(_Keyed):
    """gpmp2::ObstacleSDFFactorGPArm  gpmp2/obstacle/ObstacleSDFFactorGP.h:29-120, -inl.h:18-76"""

    def __init__(self, pose1Key, vel1Key, pose2Key, vel2Key, arm, sdf, cost_sigma, epsilon, Qc_model, delta_t, tau):
        self._keys = (pose1Key, vel1Key, pose2Key, vel2Key)
        self.arm_, self.sdf_, self.cost_sigma_, self.epsilon_ = arm, sdf, cost_sigma, epsilon
        self.Qc_, self.delta_t_, self.tau_ = Qc_model, delta_t, tau

    def evaluateError(self, conf1, vel1, conf2, vel2, jacobians=False):
        err, H = _eng().obstacle_gp_factor(_robot_handle(self.arm_), self.sdf_.handle(), self.epsilon_, self.Qc_,
                                           self.delta_t_, self.tau_, _one(conf1), _one(vel1), _one(conf2), _one(vel2))
        return (err[0],) + tuple(h[0] for h in H) if jacobians else err[0]


class ObstaclePlanarSDFFactorGPArm(ObstacleSDFFactorGPArm):
    """gpmp2::ObstaclePlanarSDFFactorGPArm  gpmp2/obstacle/ObstaclePlanarSDFFactorGP.h:29-118"""


ObstaclePlanarSDFFactorGPPointRobot = ObstaclePlanarSDFFactorGPArm
ObstaclePlanarSDFFactorGPPose2MobileArm = ObstaclePlanarSDFFactorGPArm
ObstacleSDFFactorGPPose2MobileArm = ObstacleSDFFactorGPArm


class GaussianProcessPriorLinear(_Keyed):
    """gpmp2::GaussianProcessPriorLinear  gpmp2/gp/GaussianProcessPriorLinear.h:25-120"""
    _lie = False

    def __init__(self, key1, key2, key3, key4, delta, Qc_model):
        self._keys, self.delta_t_, self.Qc_ = (key1, key2, key3, key4), delta, np.asarray(Qc_model, dtype=np.float64)
        self.dof_ = self.Qc_.shape[0]

    def evaluateError(self, pose1, vel1, pose2, vel2, jacobians=False):
        err, H = _eng().gp_prior_factor(self.dof_, self._lie, self.delta_t_, _one(pose1), _one(vel1), _one(pose2), _one(vel2))
        return (err[0],) + tuple(h[0] for h in H) if jacobians else err[0]


class GaussianProcessPriorPose2Vector(GaussianProcessPriorLinear):
    """gpmp2::GaussianProcessPriorPose2Vector = GaussianProcessPriorLie<Pose2Vector>  gpmp2/gp/GaussianProcessPriorLie.h:27-128"""
    _lie = True


class GaussianProcessInterpolatorLinear:
    """gpmp2::GaussianProcessInterpolatorLinear  gpmp2/gp/GaussianProcessInterpolatorLinear.h:25-122"""
    _lie = False

    def __init__(self, Qc_model, delta_t, tau):
        self.Qc_, self.delta_t_, self.tau_ = np.asarray(Qc_model, dtype=np.float64), delta_t, tau
        self.dof_ = self.Qc_.shape[0]

    def _call(self, p1, v1, p2, v2):
        return _eng().gp_interpolate(self.dof_, self._lie, self.Qc_, self.delta_t_, self.tau_, _one(p1), _one(v1), _one(p2), _one(v2))

    def interpolatePose(self, pose1, vel1, pose2, vel2):
        return self._call(pose1, vel1, pose2, vel2)[0][0]

    def interpolateVelocity(self, pose1, vel1, pose2, vel2):
        return self._call(pose1, vel1, pose2, vel2)[1][0]


class GaussianProcessInterpolatorPose2Vector(GaussianProcessInterpolatorLinear):
    """gpmp2::GaussianProcessInterpolatorPose2Vector  gpmp2/gp/GaussianProcessInterpolatorLie.h:27-146"""
    _lie = True


class JointLimitFactorVector(_Keyed):
    """gpmp2::JointLimitFactorVector  gpmp2/kinematics/JointLimitFactorVector.h:25-100"""

    def __init__(self, key, cost_model, down_limit, up_limit, limit_thresh):
        self._keys = (key,)
        self.down_, self.up_, self.thresh_ = (np.asarray(a, dtype=np.float64).reshape(-1) for a in (down_limit, up_limit, limit_thresh))
        if not (self.down_.size == self.up_.size == self.thresh_.size):
            raise RuntimeError("[JointLimitFactorVector] ERROR: limit vector dim does not fit.")

    def evaluateError(self, conf, jacobians=False):
        err, Hd = _eng().joint_limit_factor(self.down_, self.up_, self.thresh_, _one(conf))
        return (err[0], np.diag(Hd[0])) if jacobians else err[0]


class VelocityLimitFactorVector(JointLimitFactorVector):
    """gpmp2::VelocityLimitFactorVector  gpmp2/kinematics/VelocityLimitFactorVector.h:25-98: limits are +-vel_limit"""

    def __init__(self, key, cost_model, vel_limit, limit_thresh):
        v = np.asarray(vel_limit, dtype=np.float64).reshape(-1)
        super().__init__(key, cost_model, -v, v, limit_thresh)


class GoalFactorArm(_Keyed):
    """gpmp2::GoalFactorArm  gpmp2/kinematics/GoalFactorArm.h:24-100 (arm: an Arm or an ArmModel)"""

    def __init__(self, poseKey, cost_model, arm, dest_point):
        from .robots import Arm, ArmModel
        self._keys = (poseKey,)
        self.model_ = ArmModel(arm, []) if isinstance(arm, Arm) else arm
        self.dest_ = np.asarray(dest_point, dtype=np.float64).reshape(3)

    def evaluateError(self, conf, jacobians=False):
        err, H = _eng().goal_factor_arm(_robot_handle(self.model_), self.dest_, _one(conf))
        return (err[0], H[0]) if jacobians else err[0]


class _WorkspacePrior(_Keyed):
    _mode = 0

    def __init__(self, poseKey, robot, joint, des, cost_model=None):
        self._keys, self.robot_, self.joint_ = (poseKey,), robot, int(joint)
        self.des_ = self._as_pose(des)

    def evaluateError(self, conf, jacobians=False):
        err, H = _eng().workspace_prior_factor(_robot_handle(self.robot_), self._mode, self.joint_, self.des_, _one(conf))
        return (err[0], H[0]) if jacobians else err[0]


class GaussianPriorWorkspacePositionArm(_WorkspacePrior):
    """gpmp2::GaussianPriorWorkspacePositionArm  gpmp2/kinematics/GaussianPriorWorkspacePosition.h:24-90; des = Point3"""
    _mode = 0

    @staticmethod
    def _as_pose(p):
        T = np.eye(4)
        T[:3, 3] = np.asarray(p, dtype=np.float64).reshape(3)
        return T


class GaussianPriorWorkspaceOrientationArm(_WorkspacePrior):
    """gpmp2::GaussianPriorWorkspaceOrientationArm  gpmp2/kinematics/GaussianPriorWorkspaceOrientation.h:24-92; des = 3x3 rotation"""
    _mode = 1

    @staticmethod
    def _as_pose(R):
        T = np.eye(4)
        T[:3, :3] = np.asarray(R, dtype=np.float64).reshape(3, 3)
        return T


class GaussianPriorWorkspacePoseArm(_WorkspacePrior):
    """gpmp2::GaussianPriorWorkspacePoseArm  gpmp2/kinematics/GaussianPriorWorkspacePose.h:24-93; des = 4x4 pose"""
    _mode = 2

    @staticmethod
    def _as_pose(T):
        return np.asarray(T, dtype=np.float64).reshape(4, 4).copy()


class SelfCollisionArm(_Keyed):
    """gpmp2::SelfCollisionArm  gpmp2/obstacle/SelfCollision.h:27-140; data [n][4] = (sphere A, sphere B, epsilon, sigma)"""

    def __init__(self, poseKey, robot, data):
        self._keys, self.robot_, self.data_ = (poseKey,), robot, np.asarray(data, dtype=np.float64).reshape(-1, 4)

    def evaluateError(self, conf, jacobians=False):
        err, H = _eng().self_collision_factor(_robot_handle(self.robot_), self.data_, _one(conf))
        return (err[0], H[0]) if jacobians else err[0]


class VehicleDynamicsFactorPose2Vector(_Keyed):
    """gpmp2::VehicleDynamicsFactorPose2Vector  gpmp2/dynamics/VehicleDynamicsFactorPose2Vector.h:24-100 (Lie form: the
    body-frame lateral velocity v(1)); VehicleDynamicsFactorPose2 is the 3-dof case of the same"""
    _lie = True

    def __init__(self, poseKey, velKey, cost_sigma):
        self._keys, self.cost_sigma_ = (poseKey, velKey), cost_sigma

    def evaluateError(self, conf, vel, jacobians=False):
        err, Hp, Hv = _eng().vehicle_dynamics_factor(self._lie, _one(conf), _one(vel))
        return (err, Hp[0][None], Hv[0][None]) if jacobians else err


VehicleDynamicsFactorPose2 = VehicleDynamicsFactorPose2Vector


class VehicleDynamicsFactorVector(VehicleDynamicsFactorPose2Vector):
    """gpmp2::VehicleDynamicsFactorVector  gpmp2/dynamics/VehicleDynamicsFactorVector.h:24-98 (vector form:
    v_y cos(theta) - v_x sin(theta) with world-frame velocities)"""
    _lie = False
