/*
 * gpmp2mi.h -- C ABI of the MI355X-native GPMP2 linearize-and-solve engine.
 *
 * This header is the drop-in boundary.  Every entry point is plain C:
 * opaque handles, POD structs, caller-owned flat `double` buffers, `int` status returns, no
 * exceptions, no torch / gtsam / Eigen types.  Each declaration cites the reference interface
 * (path:line relative to the ori-drs/gpmp2 tree) it replaces.
 *
 * Conventions
 *   D  = robot dof, N = total_step (N+1 support states), I = obs_check_inter, S = #body spheres,
 *   B  = number of independent trajectories in a batch.
 *   A trajectory is a flat [N+1][2*D] array, state i = [x_i (D) ; v_i (D)]
 *   (cf. the flat layout precedent gpmp2/utils/OpenRAVEutils.cpp:35-39; keys Symbol('x',i),
 *   Symbol('v',i) of gpmp2/planner/BatchTrajOptimizer.h:39-41 map to row i).
 *   For Pose2-based robots x_i = [x, y, theta, q_arm...] (gpmp2/geometry/Pose2Vector.h:26-73).
 *   All matrices are row-major unless stated otherwise.  Everything is IEEE fp64.
 *
 * Memory spaces: entry points ending in `_dev` take HIP device pointers and a hipStream_t passed
 * as `void*`; all others take host pointers and synchronise before returning.
 *
 * The library has NO CPU fallback: every compute entry point runs hand-written gfx950 HIP
 * kernels and returns GPMP2MI_ERR_NO_DEVICE when no GPU is usable.
 */
#ifndef GPMP2MI_H
#define GPMP2MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMP2MI_VERSION 100
#define GPMP2MI_MAX_DOF 18      /* largest total dof a plan is instantiated for (csrc/common.h MAXD): the PR2 model */
#define GPMP2MI_MAX_SPHERES 96  /* largest sphere model staged on chip (PR2: 65) */

/* ---- status codes (replace the C++ exceptions of SURVEY.md section 8b "Error convention") --- */
enum {
  GPMP2MI_OK = 0,
  GPMP2MI_ERR_INVALID = 1,      /* bad argument: null pointer, dof/dimension mismatch
                                   (std::runtime_error in kinematics/JointLimitFactorVector.h:52-56,
                                    kinematics/VelocityLimitFactorVector.h:51-56) */
  GPMP2MI_ERR_NO_DEVICE = 2,    /* no usable HIP device / kernel image */
  GPMP2MI_ERR_HIP = 3,          /* a HIP runtime call failed (see gpmp2mi_last_error) */
  GPMP2MI_ERR_UNSUPPORTED = 4,  /* combination not instantiated (dof > GPMP2MI_MAX_DOF, ...) */
  GPMP2MI_ERR_ALLOC = 5,
  GPMP2MI_ERR_TIMEOUT = 6       /* a pass did not finish within GPMP2MI_WAIT_TIMEOUT_MS (default 5 s): the plan is
                                   POISONED -- every later call on it returns this code, gpmp2mi_plan_destroy neither
                                   waits for its stream nor recycles its memory (it is leaked on purpose: a hung kernel
                                   would hang the wait, a late one would write into recycled memory) */
};

/* per-trajectory status written by the optimizers */
enum {
  GPMP2MI_TRAJ_CONVERGED = 0,      /* gtsam::checkConvergence fired */
  GPMP2MI_TRAJ_MAX_ITER = 1,       /* stopped by max_iter */
  GPMP2MI_TRAJ_ROLLED_BACK = 2,    /* final step increased the error; previous values returned
                                      (planner/BatchTrajOptimizer.cpp:297-307) */
  GPMP2MI_TRAJ_NOT_SPD = 3,        /* a Cholesky pivot was <= 0 or NaN
                                      (gtsam::IndeterminantLinearSystemException) */
  GPMP2MI_TRAJ_ALREADY_OPTIMAL = 4 /* initial error <= errorTol (BatchTrajOptimizer.cpp:250-255) */
};

/* ---- robots: gpmp2/kinematics ----------------------------------------------------------- */
enum {
  GPMP2MI_ROBOT_ARM = 0,               /* gpmp2::ArmModel          kinematics/Arm.h:27-146 */
  GPMP2MI_ROBOT_POINT = 1,             /* gpmp2::PointRobotModel   kinematics/PointRobot.cpp:15-49 */
  GPMP2MI_ROBOT_POSE2_MOBILE_BASE = 2, /* gpmp2::Pose2MobileBaseModel kinematics/Pose2MobileBase.cpp:20-55 */
  GPMP2MI_ROBOT_POSE2_MOBILE_ARM = 3,  /* gpmp2::Pose2MobileArmModel  kinematics/Pose2MobileArm.cpp:30-108 */
  GPMP2MI_ROBOT_POSE2_MOBILE_2ARMS = 4,        /* gpmp2::Pose2Mobile2ArmsModel kinematics/Pose2Mobile2Arms.cpp:32-108 */
  GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_ARM = 5,   /* gpmp2::Pose2MobileVetLinArmModel kinematics/Pose2MobileVetLinArm.cpp:31-108 */
  GPMP2MI_ROBOT_POSE2_MOBILE_VETLIN_2ARMS = 6  /* gpmp2::Pose2MobileVetLin2ArmsModel kinematics/Pose2MobileVetLin2Arms.cpp:36-114 */
};

/* POD description of RobotModel<FK> = FK + BodySphereVector (kinematics/RobotModel.h:20-90). */
typedef struct gpmp2mi_robot_desc {
  int kind;                   /* GPMP2MI_ROBOT_* */
  int dof;                    /* total dof (POINT: 2; MOBILE_BASE: 3; MOBILE_ARM / 2ARMS: 3 + arm_dof;
                                 VETLIN_*: 4 + arm_dof, state [x, y, theta, lift, q...]) */
  int arm_dof;                /* number of DH joints, both arms together (0 for POINT / MOBILE_BASE);
                                 the DH arrays list arm 1 first, then arm 2 */
  const double* a;            /* [arm_dof] DH a        (Arm ctor, kinematics/Arm.cpp:15-28) */
  const double* alpha;        /* [arm_dof] DH alpha */
  const double* d;            /* [arm_dof] DH d */
  const double* theta_bias;   /* [arm_dof] or NULL (= 0) */
  double base_pose[16];       /* row-major 4x4.  ARM: pose of the arm base in the world.
                                 MOBILE_ARM: base_T_arm (Pose2MobileArm ctor).  Others: ignored. */
  int nr_spheres;             /* S */
  const int* sphere_link;     /* [S] link id (BodySphere::link_id) */
  const double* sphere_radius;/* [S] */
  const double* sphere_center;/* [S][3] centre in the link frame */
  /* two-arm / vertical-lift robots (zero / identity otherwise).  Link order: vehicle base,
   * [torso], arm-1 links, arm-2 links. */
  int arm2_dof;               /* DH joints of the second arm (the last arm2_dof of arm_dof) */
  double base_pose2[16];      /* 2ARMS: base_T_arm2 (base_pose = base_T_arm1).
                                 VETLIN_ARM: torso_T_arm (base_pose = base_T_torso).
                                 VETLIN_2ARMS: torso_T_arm1 (base_pose = base_T_torso) */
  double base_pose3[16];      /* VETLIN_2ARMS: torso_T_arm2 */
  int reverse_linact;         /* VETLIN_*: lift moves the torso down (liftBasePose3, mobileBaseUtils.cpp:51-82) */
} gpmp2mi_robot_desc;

typedef struct gpmp2mi_robot gpmp2mi_robot;
int gpmp2mi_robot_create(const gpmp2mi_robot_desc* desc, gpmp2mi_robot** out);
void gpmp2mi_robot_destroy(gpmp2mi_robot* r);
int gpmp2mi_robot_dof(const gpmp2mi_robot* r);
int gpmp2mi_robot_nr_links(const gpmp2mi_robot* r);
int gpmp2mi_robot_nr_spheres(const gpmp2mi_robot* r);

/* ---- signed distance fields: gpmp2/obstacle ---------------------------------------------- */
enum {
  GPMP2MI_SDF_LAYOUT_ZYX = 0,   /* voxels[(z*ny + y)*nx + x]  (x = column index fastest)        */
  GPMP2MI_SDF_LAYOUT_GTSAM = 1  /* voxels[(z*nx + x)*ny + y]  = std::vector<Matrix> with column-
                                   major Eigen slices, data_[z](row=y, col=x)
                                   (obstacle/SignedDistanceField.h:50,170-172)                   */
};
/* 3-D: gpmp2::SignedDistanceField(origin, cell_size, data) obstacle/SignedDistanceField.h:57-60.
 * 2-D: gpmp2::PlanarSDF(origin, cell_size, data)           obstacle/PlanarSDF.h:44-46 (nz = 1).
 * nx = field_cols_, ny = field_rows_, nz = field_z_.  The voxels are copied to the device and
 * re-laid-out there; the caller's buffer may be freed after the call returns. */
typedef struct gpmp2mi_sdf gpmp2mi_sdf;
int gpmp2mi_sdf_create(int dim, const double origin[3], double cell_size, int nx, int ny, int nz,
                       const double* voxels, int layout, gpmp2mi_sdf** out);
void gpmp2mi_sdf_destroy(gpmp2mi_sdf* s);

/* Signed distance field from an occupancy grid, on the device
 * (matlab/+gpmp2/signedDistanceField3D.m:16-34, signedDistanceField2D.m:16-34,
 * gpmp2_python/utils/signedDistanceField3D.py:22-42): cells with occ > 0.75 are obstacles;
 * field = (EDT to the obstacle set - EDT to the free set) * cell_size with exact Euclidean
 * distances; a grid without obstacles (or without free space) gives the constant 1000.
 * occ, field: host [nz][ny][nx] (nz ignored when dim = 2). */
int gpmp2mi_sdf_field_from_occupancy(int dim, int nx, int ny, int nz, const double* occ,
                                     double cell_size, double* field);
/* same, straight into a field handle (no host round trip of the field); occ in `layout` */
int gpmp2mi_sdf_create_from_occupancy(int dim, const double origin[3], double cell_size, int nx,
                                      int ny, int nz, const double* occ, int layout,
                                      gpmp2mi_sdf** out);
/* geometry and voxel data ([nz][ny][nx]) of a handle; any output pointer may be NULL */
int gpmp2mi_sdf_get_field(const gpmp2mi_sdf* s, int* dim, int* nx, int* ny, int* nz,
                          double origin[3], double* cell_size, double* field);
/* gpmp2::readSDFvolfile gpmp2/utils/fileUtils.cpp:17-62: "<pre>.vol.head" (cols rows z, origin
 * xyz, resolution) + "<pre>.vol.data" (text, x outermost, then y, then z) */
int gpmp2mi_sdf_read_vol(const char* filename_pre, gpmp2mi_sdf** out);

/* SignedDistanceField::getSignedDistance(point, g) obstacle/SignedDistanceField.h:93-99 and
 * PlanarSDF::getSignedDistance obstacle/PlanarSDF.h:61-68, batched over M points.
 * points [M][dim]; dist [M]; grad [M][dim] or NULL; in_range [M] or NULL
 * (0 where the reference throws SDFQueryOutOfRange; dist/grad are then 0). */
int gpmp2mi_sdf_query(const gpmp2mi_sdf* s, int M, const double* points, double* dist,
                      double* grad, int* in_range);

/* ---- settings: POD mirror of gpmp2::TrajOptimizerSetting planner/TrajOptimizerSetting.h:17-100 */
enum { GPMP2MI_OPT_GAUSS_NEWTON = 0, GPMP2MI_OPT_LM = 1, GPMP2MI_OPT_DOGLEG = 2 };

typedef struct gpmp2mi_settings {
  int dof;
  int total_step;                  /* N */
  double total_time;
  double conf_prior_sigma;         /* conf_prior_model = Isotropic::Sigma(dof, .) */
  double vel_prior_sigma;          /* vel_prior_model */
  int flag_pos_limit;
  int flag_vel_limit;
  const double* joint_pos_limits_up;   /* [dof] (Pose2 robots: first 3 entries ignored, see
                                          kinematics/JointLimitFactorPose2Vector.h:66-91) */
  const double* joint_pos_limits_down; /* [dof] */
  const double* vel_limits;            /* [dof] */
  const double* pos_limit_thresh;      /* [dof] */
  const double* vel_limit_thresh;      /* [dof] */
  const double* pos_limit_sigmas;      /* [dof] pos_limit_model = Diagonal::Sigmas */
  const double* vel_limit_sigmas;      /* [dof] */
  double epsilon;
  double cost_sigma;
  int obs_check_inter;             /* I */
  const double* Qc;                /* [dof][dof] covariance of Qc_model; NULL = identity */
  int opt_type;                    /* GPMP2MI_OPT_* */
  int verbosity;                   /* 0 = None, 1 = Error (per-iteration errors to stdout) */
  int final_iter_no_increase;
  double rel_thresh;
  int max_iter;
} gpmp2mi_settings;

/* Fill with the defaults of TrajOptimizerSetting(size_t) planner/TrajOptimizerSetting.cpp:32-56
 * (pointer members are left NULL: limits default to +-1e6 / thresh 1e-3 / sigma 1e-3, Qc = I). */
void gpmp2mi_settings_default(gpmp2mi_settings* s, int dof);

/* Knobs that are NOT in TrajOptimizerSetting but differ between BatchTrajOptimize and the
 * hand-built graphs of the example scripts or are hard-coded GTSAM
 * parameters in gpmp2::optimize (planner/BatchTrajOptimizer.cpp:219-234). */
#define GPMP2MI_WORKSPACE_POSITION 0
#define GPMP2MI_WORKSPACE_ORIENTATION 1
#define GPMP2MI_WORKSPACE_POSE 2
#define GPMP2MI_MAX_WORKSPACE_FACTORS 4
#define GPMP2MI_MAX_SELF_COLLISION_PAIRS 16
/* A workspace factor carried by a plan: GaussianPriorWorkspace{Position,Orientation,Pose}<Arm>
 * (kinematics/GaussianPriorWorkspacePosition.h:52-67, ...Orientation.h:52-69, ...Pose.h:53-70) or GoalFactorArm
 * (kinematics/GoalFactorArm.h:58-77 = POSITION on the last link) on the support states first_state..last_state,
 * isotropic noise `sigma`, as the hand-built graphs of matlab/Arm3GoalReachExample.m:95-110 and
 * matlab/WAMWorkspaceConstraintsExample.m:85-105 add them. */
typedef struct gpmp2mi_workspace_factor {
  int mode;                     /* GPMP2MI_WORKSPACE_POSITION / _ORIENTATION / _POSE */
  int link;                     /* link index of the FK model (GoalFactorArm: arm dof - 1) */
  int first_state, last_state;  /* inclusive range of support states */
  double sigma;
  double des_pose[16];          /* row-major 4x4 (POSITION uses the translation, ORIENTATION the rotation) */
} gpmp2mi_workspace_factor;

typedef struct gpmp2mi_graph_opts {
  int obs_skip_first_state;      /* 1: unary obstacle factors only for i>0
                                    (matlab/WAMFactorGraphExample.m:126-138); default 0 */
  double vehicle_dynamics_sigma; /* >0: add VehicleDynamicsFactorPose2Vector on every state
                                    (matlab/MobileArm2FactorGraphExample.m:122-126); default 0 */
  double lm_lambda_initial;      /* default 100 (BatchTrajOptimizer.cpp:226) */
  double lm_lambda_factor;       /* default 10   (gtsam LevenbergMarquardtParams) */
  double lm_lambda_upper;        /* default 1e5 */
  double lm_lambda_lower;        /* default 0 */
  double lm_min_model_fidelity;  /* default 1e-3 */
  double dogleg_delta_initial;   /* default 0.2 (BatchTrajOptimizer.cpp:222) */
  double abs_error_tol;          /* default 1e-5 (gtsam NonlinearOptimizerParams) */
  double error_tol;              /* default 0 */
  int fixed_iterations;          /* >0: run exactly this many iterations, no convergence test
                                    (receding-horizon budget, BASELINE config 4); default 0 */
  /* ---- extra factors of hand-built graphs, as data (gpmp2::optimize takes any NonlinearFactorGraph,
   * planner/BatchTrajOptimizer.h:206-208); all default to none */
  int end_conf_prior_off;        /* 1: no PriorFactor on x_N (a goal / workspace factor takes its place,
                                    matlab/Arm3GoalReachExample.m:107); the prior on v_N stays */
  int n_workspace;               /* <= GPMP2MI_MAX_WORKSPACE_FACTORS */
  gpmp2mi_workspace_factor workspace[GPMP2MI_MAX_WORKSPACE_FACTORS];
  int n_self_collision;          /* SelfCollision<Arm> (obstacle/SelfCollision.h:66-128) on the support states
                                    self_collision_first..last; <= GPMP2MI_MAX_SELF_COLLISION_PAIRS rows */
  int self_collision_first, self_collision_last;
  double self_collision[GPMP2MI_MAX_SELF_COLLISION_PAIRS][4];  /* sphere A, sphere B, epsilon, sigma */
} gpmp2mi_graph_opts;
void gpmp2mi_graph_opts_default(gpmp2mi_graph_opts* o);

/* ---- the planner: a resident batch of B trajectory problems ------------------------------- */
/* Replaces gpmp2::BatchTrajOptimize{2DArm,3DArm,Pose2MobileArm2D,Pose2MobileArm}
 * (planner/BatchTrajOptimizer.h:43-73; graph rules planner/BatchTrajOptimizer-inl.h:21-84;
 * optimizer loop planner/BatchTrajOptimizer.cpp:212-308) for B independent problems sharing one
 * robot, one SDF and one setting.  The plan owns all device workspace; nothing is allocated in
 * the optimize call, so it can be enqueued repeatedly (receding horizon). */
typedef struct gpmp2mi_plan gpmp2mi_plan;
int gpmp2mi_plan_create(const gpmp2mi_robot* robot, const gpmp2mi_sdf* sdf,
                        const gpmp2mi_settings* setting, const gpmp2mi_graph_opts* opts /*NULL ok*/,
                        int B, gpmp2mi_plan** out);
void gpmp2mi_plan_destroy(gpmp2mi_plan* p);

/* start/end priors (PriorFactor on x_0,v_0,x_N,v_N; BatchTrajOptimizer-inl.h:41-48) and the
 * initial values, host pointers: start_conf,start_vel,end_conf,end_vel [B][D]; init [B][N+1][2D] */
int gpmp2mi_plan_set_problem(gpmp2mi_plan* p, const double* start_conf, const double* start_vel,
                             const double* end_conf, const double* end_vel, const double* init);
/* same with device pointers; copies are enqueued on `stream` (hipStream_t) */
int gpmp2mi_plan_set_problem_dev(gpmp2mi_plan* p, const double* start_conf, const double* start_vel,
                                 const double* end_conf, const double* end_vel, const double* init,
                                 void* stream);

/* Run gpmp2::optimize on every trajectory of the batch.  All work is enqueued on `stream`
 * (hipStream_t; NULL = the default stream) and stays on the device; the host only follows the
 * per-pass active counts (pinned flags, no copies) to know when to stop enqueueing passes, and the call
 * returns once the stream has drained.  Re-running after set_problem re-optimises. */
int gpmp2mi_plan_optimize(gpmp2mi_plan* p, void* stream);

/* Results.  traj [B][N+1][2D]; iters [B] (GTSAM `iterations()`); final_error [B] (graph error of
 * the returned values); status [B] (GPMP2MI_TRAJ_*); error_trace [B][max_iter+1] (error before
 * iteration k, entry 0 = initial error; unused entries NaN).  Any pointer may be NULL. */
int gpmp2mi_plan_get_result(gpmp2mi_plan* p, double* traj, int* iters, double* final_error,
                            int* status, double* error_trace);
int gpmp2mi_plan_get_result_dev(gpmp2mi_plan* p, double* traj, int* iters, double* final_error,
                                int* status, void* stream);
/* device pointer to the resident [B][N+1][2D] result (valid until the plan is destroyed) */
const double* gpmp2mi_plan_traj_dev(const gpmp2mi_plan* p);

/* ---- incremental replanning ---------------------------------------
 * The role of gpmp2::ISAM2TrajOptimizer{2DArm,3DArm,Pose2MobileArm...}
 * (planner/ISAM2TrajOptimizer.h:57-171, planner/ISAM2TrajOptimizer-inl.h:16-195; usage
 * matlab/WAMReplannerExample.m:102-126) on the same resident plan: the chain graph is re-solved warm
 * from the current estimate with extra per-state priors.  One gpmp2mi_plan_update(p, 1, ...) is one
 * relinearise-and-solve Gauss-Newton step of the WHOLE chain -- iSAM2 would relinearise only the
 * variables whose delta exceeds relinearizeThreshold (1e-3, -inl.h:20-21); exact iSAM2 parity is
 * unpinned (no reference test exercises it, planner/tests/testISAM2TrajOptimizer.cpp:24-67).
 * `b` selects the trajectory of the batch; up to GPMP2MI_MAX_STATE_PRIORS priors per trajectory. */
#define GPMP2MI_MAX_STATE_PRIORS 8
/* fixConfigAndVel(state_idx, conf, vel): tight priors (conf_prior_model / vel_prior_model)  -inl.h:159-169 */
int gpmp2mi_plan_fix_state(gpmp2mi_plan* p, int b, int state_idx, const double* conf, const double* vel);
/* addPoseEstimate / addStateEstimate: Gaussian priors with full covariance [D][D]; vel / vel_cov may be
 * NULL (pose only)  -inl.h:172-195 */
int gpmp2mi_plan_add_state_estimate(gpmp2mi_plan* p, int b, int state_idx, const double* conf,
                                    const double* conf_cov, const double* vel, const double* vel_cov);
/* changeGoalConfigAndVel / removeGoalConfigAndVel  -inl.h:118-156 */
int gpmp2mi_plan_change_goal(gpmp2mi_plan* p, int b, const double* goal_conf, const double* goal_vel);
int gpmp2mi_plan_remove_goal(gpmp2mi_plan* p, int b);
int gpmp2mi_plan_clear_state_priors(gpmp2mi_plan* p, int b);
/* update(): `iterations` Gauss-Newton steps warm-started from the current estimate (the result of the
 * previous optimize / update; the initial values if there is none).  Results via gpmp2mi_plan_get_result. */
int gpmp2mi_plan_update(gpmp2mi_plan* p, int iterations, void* stream);

/* NonlinearFactorGraph::error(values) of the plan's graph for arbitrary trajectories
 * (host pointers; traj [B][N+1][2D] -> err [B]); uses the plan's start/end priors. */
int gpmp2mi_plan_graph_error(gpmp2mi_plan* p, const double* traj, double* err);

/* One linearization of the plan's graph at `traj` (host pointers) exported as the block-
 * tridiagonal normal equations  H delta = -g  with block size n = 2D and ordering z_i=[x_i;v_i]:
 *   Hdiag [B][N+1][n][n]  (full symmetric blocks), Hoff [B][N][n][n] (block (i+1,i)),
 *   g [B][N+1][n] (= J^T Sigma^-1 r), err [B].   Any output may be NULL. */
int gpmp2mi_plan_linearize(gpmp2mi_plan* p, const double* traj, double* Hdiag, double* Hoff,
                           double* g, double* err);

/* One-shot convenience wrapper with host buffers: create plan, set problem, optimize, fetch. */
int gpmp2mi_batch_optimize(const gpmp2mi_robot* robot, const gpmp2mi_sdf* sdf,
                           const gpmp2mi_settings* setting, const gpmp2mi_graph_opts* opts, int B,
                           const double* start_conf, const double* start_vel,
                           const double* end_conf, const double* end_vel, const double* init,
                           double* traj_out, int* iters, double* final_error, int* status);

/* gpmp2::CollisionCost{2DArm,3DArm,...} planner/BatchTrajOptimizer-inl.h:87-100:
 * sum over all states of the unary obstacle error with epsilon = 0.  traj [B][N+1][2D] host. */
int gpmp2mi_collision_cost(const gpmp2mi_robot* robot, const gpmp2mi_sdf* sdf, int total_step,
                           int B, const double* traj, double* cost);

/* ---- factor-level entry points (the GTSAM plug-in contract: evaluateError(x..., H...)) -----
 * All batched over M independent evaluations, host pointers, Jacobian outputs may be NULL. */

/* ForwardKinematics::forwardKinematics(jp, none, jpx, none, J_jpx_jp)
 * kinematics/Arm.cpp:31-143, PointRobot.cpp:15-49, Pose2MobileArm.cpp:30-108.
 * conf [M][D] -> poses [M][L][16] (row-major 4x4), J_pose [M][L][6][D] (GTSAM Pose3 tangent
 * order [omega; v], body frame). */
int gpmp2mi_forward_kinematics(const gpmp2mi_robot* r, int M, const double* conf, double* poses,
                               double* J_pose);

/* RobotModel::sphereCenters kinematics/RobotModel-inl.h:12-40.
 * conf [M][D] -> centers [M][S][3], J [M][S][3][D]. */
int gpmp2mi_sphere_centers(const gpmp2mi_robot* r, int M, const double* conf, double* centers,
                           double* J);

/* ObstacleSDFFactor / ObstaclePlanarSDFFactor ::evaluateError
 * obstacle/ObstacleSDFFactor-inl.h:18-56, obstacle/ObstaclePlanarSDFFactor-inl.h:18-58.
 * conf [M][D] -> err [M][S] (unwhitened), H1 [M][S][D]. */
int gpmp2mi_obstacle_factor(const gpmp2mi_robot* r, const gpmp2mi_sdf* s, double epsilon, int M,
                            const double* conf, double* err, double* H1);

/* ObstacleSDFFactorGP / ObstaclePlanarSDFFactorGP ::evaluateError
 * obstacle/ObstacleSDFFactorGP-inl.h:18-76, obstacle/ObstaclePlanarSDFFactorGP-inl.h:19-79
 * with GaussianProcessInterpolatorLinear (gp/GaussianProcessInterpolatorLinear.h:48-96) for
 * vector-space robots and GaussianProcessInterpolatorPose2Vector for Pose2 robots.
 * conf1,vel1,conf2,vel2 [M][D]; Qc [D][D] or NULL (= I) -> err [M][S], H1..H4 [M][S][D]. */
int gpmp2mi_obstacle_gp_factor(const gpmp2mi_robot* r, const gpmp2mi_sdf* s, double epsilon,
                               const double* Qc, double delta_t, double tau, int M,
                               const double* conf1, const double* vel1, const double* conf2,
                               const double* vel2, double* err, double* H1, double* H2,
                               double* H3, double* H4);

/* GaussianProcessPriorLinear::evaluateError gp/GaussianProcessPriorLinear.h:57-83 (lie = 0) and
 * GaussianProcessPriorPose2Vector gp/GaussianProcessPriorLie.h:61-86 (lie = 1, first three
 * coordinates are a Pose2).  -> err [M][2D]; H1..H4 [M][2D][D]. */
int gpmp2mi_gp_prior_factor(int dof, int lie, double delta_t, int M, const double* conf1,
                            const double* vel1, const double* conf2, const double* vel2,
                            double* err, double* H1, double* H2, double* H3, double* H4);

/* GaussianProcessInterpolatorLinear::interpolatePose / interpolateVelocity
 * gp/GaussianProcessInterpolatorLinear.h:62-122 (used by interpolateArmTraj,
 * planner/TrajUtils.cpp:96-197).  -> conf [M][D], vel [M][D]. */
int gpmp2mi_gp_interpolate(int dof, int lie, const double* Qc, double delta_t, double tau, int M,
                           const double* conf1, const double* vel1, const double* conf2,
                           const double* vel2, double* conf, double* vel);

/* gpmp2::interpolateArmTraj (both overloads) / interpolatePose2MobileArmTraj
 * planner/TrajUtils.cpp:96-236: up-sample B trajectories with inter_step GP-interpolated states
 * inside every interval of [start_index, end_index] (0 <= start_index < end_index <= total_step;
 * the no-range overload of the reference is start_index = 0, end_index = total_step).
 * traj [B][total_step+1][2D] -> out [B][(end_index - start_index)*(inter_step+1) + 1][2D].
 * Qc is accepted for interface parity; Lambda and Psi do not depend on it (gp/GPutils.h:44-59). */
int gpmp2mi_interpolate_traj(int dof, int lie, const double* Qc, double delta_t, int inter_step,
                             int B, int total_step, int start_index, int end_index,
                             const double* traj, double* out);
/* same on device pointers, enqueued on `stream` (e.g. traj = gpmp2mi_plan_traj_dev(plan)) */
int gpmp2mi_interpolate_traj_dev(int dof, int lie, double delta_t, int inter_step, int B,
                                 int total_step, int start_index, int end_index,
                                 const double* traj, double* out, void* stream);

/* GaussianPriorWorkspace{Position,Orientation,Pose}::evaluateError
 * kinematics/GaussianPriorWorkspacePosition.h:52-67, ...Orientation.h:52-69, ...Pose.h:53-70:
 * prior on the world pose of link `joint`.  des_pose: row-major 4x4 (position mode uses its
 * translation, orientation mode its rotation).  -> err [M][3|3|6] (pose: [omega; u] of
 * Pose3::Logmap(des^-1 * pose)), H [M][rows][D] (may be NULL).  Rot3 / Pose3 log maps follow
 * GTSAM 4.0 (upstream; pinned by the reference's known answers). */
int gpmp2mi_workspace_prior_factor(const gpmp2mi_robot* r, int mode, int joint,
                                   const double des_pose[16], int M, const double* conf,
                                   double* err, double* H);
/* GoalFactorArm::evaluateError kinematics/GoalFactorArm.h:58-77: end-effector position minus
 * dest_point (= the position prior on the last link).  -> err [M][3], H [M][3][D] */
int gpmp2mi_goal_factor_arm(const gpmp2mi_robot* r, const double dest_point[3], int M,
                            const double* conf, double* err, double* H);
/* SelfCollision::evaluateError obstacle/SelfCollision.h:66-128.  data [n_pairs][4] = (sphere A id,
 * sphere B id, epsilon, sigma) with ids in the order of the robot description; hinge on
 * radius_A + radius_B + epsilon - |c_A - c_B|.  -> err [M][n_pairs], H [M][n_pairs][D] */
int gpmp2mi_self_collision_factor(const gpmp2mi_robot* r, int n_pairs, const double* data, int M,
                                  const double* conf, double* err, double* H);

/* VehicleDynamicsFactorPose2 / Pose2Vector (lie = 1) and VehicleDynamicsFactorVector (lie = 0)
 * dynamics/VehicleDynamics.h:19-40, dynamics/VehicleDynamicsFactorPose2Vector.h:55-78,
 * dynamics/VehicleDynamicsFactorVector.h:53-78: sliding velocity of an SE(2) base.  conf, vel [M][D] with
 * D >= 3 (the first three coordinates are x, y, theta).  -> err [M], Hp [M][D], Hv [M][D] (may be NULL). */
int gpmp2mi_vehicle_dynamics_factor(int dof, int lie, int M, const double* conf, const double* vel,
                                    double* err, double* Hp, double* Hv);

/* JointLimitFactorVector / VelocityLimitFactorVector ::evaluateError
 * kinematics/JointLimitFactorVector.h:62-79, kinematics/VelocityLimitFactorVector.h:62-79.
 * x [M][D] -> err [M][D], Hdiag [M][D] (the diagonal of the Jacobian). */
int gpmp2mi_joint_limit_factor(int dof, const double* down, const double* up, const double* thresh,
                               int M, const double* x, double* err, double* Hdiag);

/* Batched block-tridiagonal SPD solve  H x = b  (the replacement for GTSAM's sparse Cholesky,
 * planner/BatchTrajOptimizer.cpp:240-286 -> GaussianFactorGraph::optimize).
 * Hdiag [B][nblk][n][n], Hoff [B][nblk-1][n][n] (block (i+1,i)), b [B][nblk][n] -> x, ok [B]. */
int gpmp2mi_block_tridiag_solve(int B, int nblk, int n, const double* Hdiag, const double* Hoff,
                                const double* b, double* x, int* ok);

/* ---- misc ---------------------------------------------------------------------------------- */
const char* gpmp2mi_last_error(void);  /* thread-local message of the last failing call */
int gpmp2mi_device_count(void);
int gpmp2mi_version(void);
/* Per-kernel timing of the last gpmp2mi_plan_optimize when enabled (HIP events on the plan's
 * stream): names[i] / ms[i] / launches[i] for i < *n.  Used by bench.py for the roofline line. */
int gpmp2mi_plan_enable_timing(gpmp2mi_plan* p, int enable);
int gpmp2mi_plan_get_timing(gpmp2mi_plan* p, int* n, const char** names, double* ms, int* launches);
/* Diagnostic builds (-DG2_STAMPS) only: 64 raw s_memtime stamps of trajectory b's last solve step. */
int gpmp2mi_plan_debug_stamps(gpmp2mi_plan* p, int b, unsigned long long* out64);
/* Diagnostic: scalars of trajectory b's last LM / Dogleg trial step, out17 = {g.delta, |delta|^2, g.g, g^T H g,
 * g.dx_n, |dx_n|^2, model decrease q, |step|, zero-step flag, -, ..., [16] = current lambda / trust radius}. */
int gpmp2mi_plan_debug_scalars(gpmp2mi_plan* p, int b, double* out17);
/* Test hook, host only (no GPU needed): the wall-clock-bounded spin the pass driver uses on its device-mapped
 * pass flags, run on a caller-owned flag: returns GPMP2MI_OK with *value = *flag once *flag >= 0, or
 * GPMP2MI_ERR_TIMEOUT (gpmp2mi_last_error set) after timeout_ms.  The driver's own limit is 5 s
 * (GPMP2MI_WAIT_TIMEOUT_MS overrides). */
int gpmp2mi_debug_wait_flag(const int* flag, int timeout_ms, int* value);
/* Test hook (works without a GPU: all zeros then): arena chunks / pass-flag buffers owned by live plans, the pooled
 * ones, and the plans leaked because they were poisoned (GPMP2MI_ERR_TIMEOUT).  Any pointer may be NULL. */
int gpmp2mi_debug_resource_counts(long* live_chunks, long* pooled_chunks, long* live_flagbufs, long* pooled_flagbufs,
                                  long* leaked_plans);
/* Test hooks: a one-thread kernel that occupies `stream` until gpmp2mi_debug_stall_release(token) -- or, whatever
 * happens, until max_ms (<= 10000) of device wall clock have passed -- so that the pass driver's timeout path can be
 * driven on a real stream.  release() sets the flag, waits for that stream and frees the token. */
int gpmp2mi_debug_stall_begin(void* stream, int max_ms, void** token);
int gpmp2mi_debug_stream_create(void** stream);   /* a non-blocking stream of the HIP runtime the library uses */
int gpmp2mi_debug_stream_destroy(void* stream);
int gpmp2mi_debug_stall_release(void* token);
/* Diagnostic: lane semantics of the wave-level moves the solver relies on (tests/test_gpu_plan.py). */
int gpmp2mi_debug_crosslane(const double* in64, double* out512);
/* Diagnostic: raw device-to-host copy of a solver hand-over buffer of the plan (0: diagonal tiles [B][N+1][256],
 * 1: factor tiles [B][N+1][3][256], 2: pending Schur tiles [B][groups][256], 3: level-4 couplings [B][groups][256]). */
int gpmp2mi_plan_debug_read(gpmp2mi_plan* p, int which, double* out, long count);

#ifdef __cplusplus
}
#endif
#endif /* GPMP2MI_H */
