"""gpmp2_amd -- MI355X-native GPMP2 linearize-and-solve engine (host-side Python mirror).

The compute path is the HIP library gpmp2_amd/csrc/libgpmp2mi.so behind the C ABI of
include/gpmp2mi.h; this package is the reference-shaped façade over it (robots, SDFs, settings,
BatchTrajOptimize*, factor evaluateError) used by tests and bench.py.
"""
from .datasets import generate2Ddataset, generate3Ddataset, sdf3_zyx  # noqa: F401
from .robots import (Arm, ArmModel, BodySphere, PointRobot, PointRobotModel, Pose2Mobile2Arms,  # noqa: F401
                     Pose2Mobile2ArmsModel, Pose2MobileArm, Pose2MobileArmModel, Pose2MobileBase,
                     Pose2MobileBaseModel, Pose2MobileVetLin2Arms, Pose2MobileVetLin2ArmsModel,
                     Pose2MobileVetLinArm, Pose2MobileVetLinArmModel, generateArm, generateMobileArm,
                     generatePointRobot, pose3, rot_yaw)
from .factors import (GaussianPriorWorkspaceOrientationArm, GaussianPriorWorkspacePoseArm,  # noqa: F401
                      GaussianPriorWorkspacePositionArm, GaussianProcessInterpolatorLinear,
                      GaussianProcessInterpolatorPose2Vector, GaussianProcessPriorLinear,
                      GaussianProcessPriorPose2Vector, GoalFactorArm, JointLimitFactorVector,
                      ObstaclePlanarSDFFactorArm, ObstaclePlanarSDFFactorGPArm, ObstaclePlanarSDFFactorGPPointRobot,
                      ObstaclePlanarSDFFactorPointRobot, ObstacleSDFFactorArm, ObstacleSDFFactorGPArm,
                      SelfCollisionArm, VehicleDynamicsFactorPose2, VehicleDynamicsFactorPose2Vector,
                      VehicleDynamicsFactorVector, VelocityLimitFactorVector)
from .planner import (BatchTrajOptimize2DArm, BatchTrajOptimize3DArm, BatchTrajOptimizePose2Mobile2Arms,  # noqa: F401
                      BatchTrajOptimizePose2MobileArm, BatchTrajOptimizePose2MobileArm2D,
                      BatchTrajOptimizePose2MobileVetLin2Arms, BatchTrajOptimizePose2MobileVetLinArm,
                      CollisionCostPose2Mobile2Arms, CollisionCostPose2MobileVetLin2Arms,
                      CollisionCostPose2MobileBase, CollisionCostPose2MobileBase2D, CollisionCostPose2MobileVetLinArm, ISAM2TrajOptimizerPose2MobileVetLin2Arms, CollisionCost2DArm, CollisionCost3DArm,
                      CollisionCostPose2MobileArm, CollisionCostPose2MobileArm2D, ISAM2TrajOptimizer2DArm,
                      ISAM2TrajOptimizer3DArm, ISAM2TrajOptimizerPose2MobileArm, ISAM2TrajOptimizerPose2MobileArm2D,
                      PlanarSDF, SDFQueryOutOfRange, SignedDistanceField, readSDFvolfile, signedDistanceField2D,
                      signedDistanceField3D)
from .settings import TrajOptimizerSetting  # noqa: F401
from .trajutils import (initArmTrajStraightLine, initPose2TrajStraightLine, initPose2VectorTrajStraightLine,  # noqa: F401
                        interpolateArmTraj, interpolatePose2MobileArmTraj, interpolatePose2Traj, traj_from_values,
                        values_from_traj)

__all__ = [n for n in dir() if not n.startswith("_")]
