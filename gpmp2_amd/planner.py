"""Reference-shaped planner interface over the HIP engine (gpmp2.h:440-470, 738-935).

Same names, argument order and error behaviour as the reference's wrapped classes and functions for
the hot path; trajectories travel as the ``{('x', i): conf, ('v', i): vel}`` stand-in for
``gtsam::Values`` (see trajutils.values_from_traj) or as flat ``[N+1][2D]`` arrays -- both are
accepted, and the same kind is returned.  Everything computes on the GPU through
libgpmp2mi.so; without a device the calls raise ``Gpmp2miError`` (there is no CPU path).
"""
from __future__ import annotations

import numpy as np

from . import engine as _engine
from .settings import TrajOptimizerSetting
from .trajutils import traj_from_values, values_from_traj

_ENGINE = None


def _eng():
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = _engine.Engine()
    return _ENGINE


class SDFQueryOutOfRange(RuntimeError):
    """gpmp2::SDFQueryOutOfRange (gpmp2/obstacle/SDFexception.h:15-24)"""

    def __init__(self):
        super().__init__("Querying SDF out of range")


class _DeviceSdf:
    _handle = None

    def handle(self):
        if self._handle is None:
            self._handle = self._upload()
        return self._handle

    def getSignedDistance(self, point):
        dist, _, inr = _eng().sdf_query(self.handle(), np.asarray(point, dtype=np.float64).reshape(1, -1))
        if not inr[0]:
            raise SDFQueryOutOfRange()
        return float(dist[0])


class SignedDistanceField(_DeviceSdf):
    """gpmp2::SignedDistanceField (gpmp2/obstacle/SignedDistanceField.h:28-72): origin, cell size and
    field_z layers of [field_rows][field_cols] (row = y, col = x)."""

    def __init__(self, origin, cell_size, field_rows, field_cols, field_z):
        self.origin_ = np.asarray(origin, dtype=np.float64).reshape(3)
        self.cell_size_ = float(cell_size)
        self.field_rows_, self.field_cols_, self.field_z_ = int(field_rows), int(field_cols), int(field_z)
        self.data_ = np.zeros((self.field_z_, self.field_rows_, self.field_cols_))

    def initFieldData(self, z_idx, field_layer):
        layer = np.asarray(field_layer, dtype=np.float64)
        if z_idx >= self.field_z_:
            raise RuntimeError("[SignedDistanceField] matrix layer out of index")
        if layer.shape != (self.field_rows_, self.field_cols_):
            raise RuntimeError("[SignedDistanceField] matrix size does not fit")
        self.data_[z_idx] = layer
        self._handle = None

    def origin(self):
        return self.origin_

    def x_count(self):
        return self.field_cols_

    def y_count(self):
        return self.field_rows_

    def z_count(self):
        return self.field_z_

    def cell_size(self):
        return self.cell_size_

    def raw_data(self):
        return self.data_

    def _upload(self):
        return _eng().sdf(self.origin_, self.cell_size_, self.data_)


class PlanarSDF(_DeviceSdf):
    """gpmp2::PlanarSDF (gpmp2/obstacle/PlanarSDF.h:28-60): data [rows = y][cols = x]"""

    def __init__(self, origin, cell_size, data):
        self.origin_ = np.asarray(origin, dtype=np.float64).reshape(2)
        self.cell_size_ = float(cell_size)
        self.data_ = np.ascontiguousarray(data, dtype=np.float64)

    def origin(self):
        return self.origin_

    def x_count(self):
        return self.data_.shape[1]

    def y_count(self):
        return self.data_.shape[0]

    def cell_size(self):
        return self.cell_size_

    def raw_data(self):
        return self.data_

    def _upload(self):
        return _eng().sdf(self.origin_, self.cell_size_, self.data_)


def signedDistanceField2D(ground_truth_map, cell_size):
    """matlab/+gpmp2/signedDistanceField2D.m:14-34 on the GPU: map [rows = y][cols = x] -> field, same shape"""
    return _eng().sdf_field_from_occupancy(np.asarray(ground_truth_map, dtype=np.float64), cell_size)


def signedDistanceField3D(ground_truth_map, cell_size):
    """matlab/+gpmp2/signedDistanceField3D.m:14-34 on the GPU: map and field in the dataset's own index
    order (the transform is symmetric in the axes, so no transpose is needed)"""
    return _eng().sdf_field_from_occupancy(np.asarray(ground_truth_map, dtype=np.float64), cell_size)


def readSDFvolfile(filename_pre):
    """gpmp2::readSDFvolfile (gpmp2/utils/fileUtils.cpp:17-62) -> SignedDistanceField"""
    h = _eng().sdf_read_vol(filename_pre)
    f = _eng().sdf_field(h)
    sdf = SignedDistanceField(f["origin"], f["cell_size"], f["data"].shape[1], f["data"].shape[2], f["data"].shape[0])
    sdf.data_ = f["data"]
    sdf._handle = h
    return sdf


def _robot_handle(model):
    h = getattr(model, "_gpmp2mi_handle", None)
    if h is None:
        h = _eng().robot(model)
        model._gpmp2mi_handle = h
    return h


def _flat(values, setting):
    if isinstance(values, dict):
        return traj_from_values(values, setting.total_step), True
    return np.ascontiguousarray(values, dtype=np.float64), False


def _check_dims(model, setting, *vecs):
    D = model.dof()
    if setting.dof != D:
        raise ValueError("[TrajOptimizerSetting] dof does not fit the robot")
    for v in vecs:
        if np.asarray(v).size != D:
            raise ValueError("[BatchTrajOptimize] vector dim does not fit dof")


def _batch(model, sdf, start_conf, start_vel, end_conf, end_vel, init_values, setting):
    """gpmp2::internal::BatchTrajOptimize gpmp2/planner/BatchTrajOptimizer-inl.h:21-84 + optimize()"""
    _check_dims(model, setting, start_conf, start_vel, end_conf, end_vel)
    init, as_values = _flat(init_values, setting)
    res = _eng().batch_optimize(_robot_handle(model), sdf.handle(), setting, start_conf, start_vel, end_conf, end_vel,
                                init[None])
    if res["status"][0] == _engine.TRAJ_NOT_SPD:
        raise RuntimeError("IndeterminantLinearSystemException")
    traj = res["traj"][0]
    return values_from_traj(traj) if as_values else traj


# gpmp2/planner/BatchTrajOptimizer.h:43-73 -- one engine path for every instantiation
BatchTrajOptimize2DArm = _batch
BatchTrajOptimize3DArm = _batch
BatchTrajOptimizePose2MobileArm2D = _batch
BatchTrajOptimizePose2MobileArm = _batch
BatchTrajOptimizePose2Mobile2Arms = _batch
BatchTrajOptimizePose2MobileVetLinArm = _batch
BatchTrajOptimizePose2MobileVetLin2Arms = _batch
BatchTrajOptimizePointRobot2D = _batch       # the PointRobot graphs of the matlab examples


def _collision_cost(model, sdf, result, setting):
    """gpmp2::internal::CollisionCost gpmp2/planner/BatchTrajOptimizer-inl.h:87-100"""
    traj, _ = _flat(result, setting)
    return float(_eng().collision_cost(_robot_handle(model), sdf.handle(), setting.total_step, traj[None])[0])


CollisionCost2DArm = _collision_cost
CollisionCost3DArm = _collision_cost
CollisionCostPose2MobileBase2D = _collision_cost
CollisionCostPose2MobileBase = _collision_cost
CollisionCostPose2MobileArm2D = _collision_cost
CollisionCostPose2MobileArm = _collision_cost
CollisionCostPose2Mobile2Arms = _collision_cost
CollisionCostPose2MobileVetLinArm = _collision_cost
CollisionCostPose2MobileVetLin2Arms = _collision_cost


class _ISAM2TrajOptimizer:
    """gpmp2::internal::ISAM2TrajOptimizer (gpmp2/planner/ISAM2TrajOptimizer.h:58-137).  Same call
    sequence; update() is one full relinearise + solve of the resident plan (gpmp2mi_plan_update)
    instead of an iSAM2 partial update."""

    def __init__(self, arm, sdf, setting: TrajOptimizerSetting):
        if setting.dof != arm.dof():
            raise ValueError("[TrajOptimizerSetting] dof does not fit the robot")
        self.setting_, self.arm_, self.sdf_ = setting, arm, sdf
        self.plan_ = _eng().plan(_robot_handle(arm), sdf.handle(), setting, 1)
        self.problem_ = None
        self.opt_values_ = None
        self.as_values_ = True

    def initFactorGraph(self, start_conf, start_vel, goal_conf, goal_vel):
        _check_dims(self.arm_, self.setting_, start_conf, start_vel, goal_conf, goal_vel)
        self.problem_ = [np.asarray(v, dtype=np.float64).reshape(1, -1) for v in (start_conf, start_vel, goal_conf, goal_vel)]

    def initValues(self, init_values):
        if self.problem_ is None:
            raise RuntimeError("[ISAM2TrajOptimizer] initFactorGraph must come first")
        init, self.as_values_ = _flat(init_values, self.setting_)
        self.plan_.set_problem(*self.problem_, init[None])
        self.opt_values_ = init

    def update(self):
        self.plan_.update(1)
        res = self.plan_.result()
        if res["status"][0] == _engine.TRAJ_NOT_SPD:
            raise RuntimeError("IndeterminantLinearSystemException")
        self.opt_values_ = res["traj"][0]

    def changeGoalConfigAndVel(self, goal_conf, goal_vel):
        self.plan_.change_goal(0, goal_conf, goal_vel)

    def removeGoalConfigAndVel(self):
        self.plan_.remove_goal(0)

    def fixConfigAndVel(self, state_idx, conf_fix, vel_fix):
        self.plan_.fix_state(0, state_idx, conf_fix, vel_fix)

    def addPoseEstimate(self, state_idx, pose, pose_cov):
        self.plan_.add_state_estimate(0, state_idx, pose, pose_cov)

    def addStateEstimate(self, state_idx, pose, pose_cov, vel, vel_cov):
        self.plan_.add_state_estimate(0, state_idx, pose, pose_cov, vel, vel_cov)

    def values(self):
        return values_from_traj(self.opt_values_) if self.as_values_ else self.opt_values_


class ISAM2TrajOptimizer2DArm(_ISAM2TrajOptimizer):
    """gpmp2/planner/ISAM2TrajOptimizer.h:143-147"""


class ISAM2TrajOptimizer3DArm(_ISAM2TrajOptimizer):
    """gpmp2/planner/ISAM2TrajOptimizer.h:150-154"""


class ISAM2TrajOptimizerPose2MobileArm2D(_ISAM2TrajOptimizer):
    """gpmp2/planner/ISAM2TrajOptimizer.h:157-162"""


class ISAM2TrajOptimizerPose2MobileArm(_ISAM2TrajOptimizer):
    """gpmp2/planner/ISAM2TrajOptimizer.h:165-170"""


class ISAM2TrajOptimizerPose2MobileVetLin2Arms(_ISAM2TrajOptimizer):
    """gpmp2/planner/ISAM2TrajOptimizer.h:173-178"""
