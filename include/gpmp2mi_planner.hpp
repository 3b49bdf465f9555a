// gpmp2mi_planner.hpp -- header-only C++ host facade over the C ABI (include/gpmp2mi.h) that keeps the
// call shapes of gpmp2/planner so existing C++ callers can switch with a namespace change:
//
//   gpmp2::Arm, gpmp2::BodySphere, gpmp2::ArmModel      gpmp2/kinematics/Arm.h:48-59, RobotModel.h:20-90
//   gpmp2::SignedDistanceField, gpmp2::PlanarSDF        gpmp2/obstacle/SignedDistanceField.h:57-81, PlanarSDF.h:44-46
//   gpmp2::TrajOptimizerSetting                         gpmp2/planner/TrajOptimizerSetting.h:17-100
//   gpmp2::BatchTrajOptimize3DArm / 2DArm               gpmp2/planner/BatchTrajOptimizer.h:43-56
//   gpmp2::CollisionCost3DArm / 2DArm                   gpmp2/planner/BatchTrajOptimizer.h:135-147
//   gpmp2::Pose2MobileArm / Pose2MobileArmModel, BatchTrajOptimizePose2MobileArm(2D)   gpmp2/planner/BatchTrajOptimizer.h:57-73
//   gpmp2::Obstacle(Planar)SDFFactorArm / ...Pose2MobileArm, GoalFactorArm, GaussianPriorWorkspacePoseArm, SelfCollisionArm
//   gpmp2::interpolateArmTraj / interpolatePose2MobileArmTraj  gpmp2/planner/TrajUtils.cpp:96-236
//   gpmp2::ISAM2TrajOptimizer2DArm / 3DArm              gpmp2/planner/ISAM2TrajOptimizer.h:143-156
//   gpmp2::initArmTrajStraightLine                      gpmp2/planner/TrajUtils.cpp:25-50
//
// The reference passes gtsam::Values / gtsam::Vector / gtsam::Pose3.  GTSAM, Boost and Eigen are not part of
// this repository, so the facade uses plain std::vector containers (`Trajectory`, state i = [x_i; v_i]) and,
// where <gtsam/nonlinear/Values.h> is on the include path, also provides converters to and from
// gtsam::Values with the reference's key convention Symbol('x', i) / Symbol('v', i).
// Errors: the C ABI's status codes are rethrown as std::runtime_error, matching the reference's use of
// exceptions.
#pragma once
#include <array>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "gpmp2mi.h"

#if defined(__has_include)
#if __has_include(<gtsam/nonlinear/Values.h>) && __has_include(<gtsam/inference/Symbol.h>)
#include <gtsam/inference/Symbol.h>
#include <gtsam/nonlinear/Values.h>
#define GPMP2MI_HAVE_GTSAM 1
#endif
#endif

namespace gpmp2mi {

using Vector = std::vector<double>;

inline void check(int rc, const char* what) {
  if (rc != GPMP2MI_OK)
    throw std::runtime_error(std::string("[gpmp2mi] ") + what + ": " + gpmp2mi_last_error() + " (code " +
                             std::to_string(rc) + ")");
}

/// 4x4 homogeneous transform, row-major (stand-in for gtsam::Pose3)
struct Pose3 {
  std::array<double, 16> m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  static Pose3 Translation(double x, double y, double z) {
    Pose3 p;
    p.m[3] = x;
    p.m[7] = y;
    p.m[11] = z;
    return p;
  }
};

/// body sphere: attached link id, radius, centre in the link frame
struct BodySphere {
  std::size_t link_id;
  double radius;
  std::array<double, 3> center;
  BodySphere(std::size_t id, double r, const std::array<double, 3>& c) : link_id(id), radius(r), center(c) {}
};
using BodySphereVector = std::vector<BodySphere>;

/// DH arm, same constructor argument order as gpmp2::Arm
class Arm {
 public:
  Arm(std::size_t dof, const Vector& a, const Vector& alpha, const Vector& d, const Pose3& base_pose = Pose3(),
      const Vector& theta_bias = Vector())
      : dof_(dof), a_(a), alpha_(alpha), d_(d), base_(base_pose),
        bias_(theta_bias.empty() ? Vector(dof, 0.0) : theta_bias) {
    if (a.size() != dof || alpha.size() != dof || d.size() != dof || bias_.size() != dof)
      throw std::runtime_error("[Arm] DH parameter vector dim does not fit dof");
  }
  std::size_t dof() const { return dof_; }
  std::size_t nr_links() const { return dof_; }
  const Vector& a() const { return a_; }
  const Vector& alpha() const { return alpha_; }
  const Vector& d() const { return d_; }
  const Vector& theta_bias() const { return bias_; }
  const Pose3& base_pose() const { return base_; }

 private:
  std::size_t dof_;
  Vector a_, alpha_, d_;
  Pose3 base_;
  Vector bias_;
};

/// RobotModel<Arm>: owns the device-side robot handle
class ArmModel {
 public:
  ArmModel(const Arm& arm, const BodySphereVector& spheres) : arm_(arm), spheres_(spheres) {
    gpmp2mi_robot_desc d{};
    d.kind = GPMP2MI_ROBOT_ARM;
    d.dof = d.arm_dof = static_cast<int>(arm.dof());
    d.a = arm_.a().data();
    d.alpha = arm_.alpha().data();
    d.d = arm_.d().data();
    d.theta_bias = arm_.theta_bias().data();
    for (int i = 0; i < 16; i++) d.base_pose[i] = arm_.base_pose().m[i];
    std::vector<int> link;
    Vector radius, center;
    for (const auto& s : spheres_) {
      link.push_back(static_cast<int>(s.link_id));
      radius.push_back(s.radius);
      center.insert(center.end(), s.center.begin(), s.center.end());
    }
    d.nr_spheres = static_cast<int>(spheres_.size());
    d.sphere_link = link.data();
    d.sphere_radius = radius.data();
    d.sphere_center = center.data();
    check(gpmp2mi_robot_create(&d, &h_), "gpmp2mi_robot_create");
  }
  ArmModel(const ArmModel&) = delete;
  ArmModel& operator=(const ArmModel&) = delete;
  ~ArmModel() { gpmp2mi_robot_destroy(h_); }
  std::size_t dof() const { return arm_.dof(); }
  std::size_t nr_body_spheres() const { return spheres_.size(); }
  double sphere_radius(std::size_t i) const { return spheres_[i].radius; }
  const Arm& fk_model() const { return arm_; }
  const gpmp2mi_robot* handle() const { return h_; }

  /// RobotModel::sphereCentersMat -> [S][3]
  Vector sphereCenters(const Vector& conf) const {
    Vector out(3 * spheres_.size());
    check(gpmp2mi_sphere_centers(h_, 1, conf.data(), out.data(), nullptr), "gpmp2mi_sphere_centers");
    return out;
  }

 private:
  Arm arm_;
  BodySphereVector spheres_;
  gpmp2mi_robot* h_ = nullptr;
};

/// gpmp2::Pose2MobileArm  gpmp2/kinematics/Pose2MobileArm.h:22-70 (state [x, y, theta, q...])
class Pose2MobileArm {
 public:
  explicit Pose2MobileArm(const Arm& arm, const Pose3& base_T_arm = Pose3()) : arm_(arm), base_T_arm_(base_T_arm) {}
  std::size_t dof() const { return arm_.dof() + 3; }
  std::size_t nr_links() const { return arm_.dof() + 1; }
  const Arm& arm() const { return arm_; }
  const Pose3& base_T_arm() const { return base_T_arm_; }

 private:
  Arm arm_;
  Pose3 base_T_arm_;
};

/// RobotModel<Pose2MobileArm>: owns the device-side robot handle; sphere link 0 = vehicle base
class Pose2MobileArmModel {
 public:
  Pose2MobileArmModel(const Pose2MobileArm& marm, const BodySphereVector& spheres) : marm_(marm), spheres_(spheres) {
    gpmp2mi_robot_desc d{};
    d.kind = GPMP2MI_ROBOT_POSE2_MOBILE_ARM;
    d.dof = static_cast<int>(marm.dof());
    d.arm_dof = static_cast<int>(marm.arm().dof());
    d.a = marm_.arm().a().data();
    d.alpha = marm_.arm().alpha().data();
    d.d = marm_.arm().d().data();
    d.theta_bias = marm_.arm().theta_bias().data();
    for (int i = 0; i < 16; i++) {
      d.base_pose[i] = marm_.base_T_arm().m[i];
      d.base_pose2[i] = d.base_pose3[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    std::vector<int> link;
    Vector radius, center;
    for (const auto& sp : spheres_) {
      link.push_back(static_cast<int>(sp.link_id));
      radius.push_back(sp.radius);
      center.insert(center.end(), sp.center.begin(), sp.center.end());
    }
    d.nr_spheres = static_cast<int>(spheres_.size());
    d.sphere_link = link.data();
    d.sphere_radius = radius.data();
    d.sphere_center = center.data();
    check(gpmp2mi_robot_create(&d, &h_), "gpmp2mi_robot_create");
  }
  Pose2MobileArmModel(const Pose2MobileArmModel&) = delete;
  Pose2MobileArmModel& operator=(const Pose2MobileArmModel&) = delete;
  ~Pose2MobileArmModel() { gpmp2mi_robot_destroy(h_); }
  std::size_t dof() const { return marm_.dof(); }
  std::size_t nr_body_spheres() const { return spheres_.size(); }
  const Pose2MobileArm& fk_model() const { return marm_; }
  const gpmp2mi_robot* handle() const { return h_; }

 private:
  Pose2MobileArm marm_;
  BodySphereVector spheres_;
  gpmp2mi_robot* h_ = nullptr;
};

/// 3-D signed distance field; data in the reference's own storage order: z slices of column-major
/// (row = y, col = x) matrices, i.e. voxels[(z * cols + x) * rows + y]
class SignedDistanceField {
 public:
  SignedDistanceField(const std::array<double, 3>& origin, double cell_size, std::size_t field_rows,
                      std::size_t field_cols, std::size_t field_z, const Vector& column_major_slices) {
    if (column_major_slices.size() != field_rows * field_cols * field_z)
      throw std::runtime_error("[SignedDistanceField] data size does not match the field dimensions");
    check(gpmp2mi_sdf_create(3, origin.data(), cell_size, static_cast<int>(field_cols), static_cast<int>(field_rows),
                             static_cast<int>(field_z), column_major_slices.data(), GPMP2MI_SDF_LAYOUT_GTSAM, &h_),
          "gpmp2mi_sdf_create");
  }
  SignedDistanceField(const SignedDistanceField&) = delete;
  SignedDistanceField& operator=(const SignedDistanceField&) = delete;
  ~SignedDistanceField() { gpmp2mi_sdf_destroy(h_); }
  /// getSignedDistance(point, gradient); returns false where the reference throws SDFQueryOutOfRange
  bool getSignedDistance(const std::array<double, 3>& p, double& dist, std::array<double, 3>* grad = nullptr) const {
    int in = 0;
    check(gpmp2mi_sdf_query(h_, 1, p.data(), &dist, grad ? grad->data() : nullptr, &in), "gpmp2mi_sdf_query");
    return in != 0;
  }
  const gpmp2mi_sdf* handle() const { return h_; }

 private:
  gpmp2mi_sdf* h_ = nullptr;
};

/// 2-D signed distance field, column-major (row = y, col = x) like the gtsam::Matrix it replaces
class PlanarSDF {
 public:
  PlanarSDF(const std::array<double, 2>& origin, double cell_size, std::size_t field_rows, std::size_t field_cols,
            const Vector& column_major) {
    if (column_major.size() != field_rows * field_cols)
      throw std::runtime_error("[PlanarSDF] data size does not match the field dimensions");
    const double o[3] = {origin[0], origin[1], 0.0};
    check(gpmp2mi_sdf_create(2, o, cell_size, static_cast<int>(field_cols), static_cast<int>(field_rows), 1,
                             column_major.data(), GPMP2MI_SDF_LAYOUT_GTSAM, &h_),
          "gpmp2mi_sdf_create");
  }
  PlanarSDF(const PlanarSDF&) = delete;
  PlanarSDF& operator=(const PlanarSDF&) = delete;
  ~PlanarSDF() { gpmp2mi_sdf_destroy(h_); }
  const gpmp2mi_sdf* handle() const { return h_; }

 private:
  gpmp2mi_sdf* h_ = nullptr;
};

/// general setting of all trajectory optimizers -- same public fields and setters as the reference
struct TrajOptimizerSetting {
  enum IterationType { GaussNewton = GPMP2MI_OPT_GAUSS_NEWTON, LM = GPMP2MI_OPT_LM, Dogleg = GPMP2MI_OPT_DOGLEG };
  enum VerbosityLevel { None, Error };
  std::size_t dof;
  std::size_t total_step = 10;
  double total_time = 1.0;
  double conf_prior_sigma = 0.0001, vel_prior_sigma = 0.0001;
  bool flag_pos_limit = false, flag_vel_limit = false;
  Vector joint_pos_limits_up, joint_pos_limits_down, vel_limits, pos_limit_thresh, vel_limit_thresh;
  Vector pos_limit_sigmas, vel_limit_sigmas;
  double epsilon = 0.2, cost_sigma = 0.1;
  std::size_t obs_check_inter = 5;
  Vector Qc;  // dof x dof row-major; empty = identity (noiseModel::Unit)
  IterationType opt_type = Dogleg;
  VerbosityLevel opt_verbosity = None;
  bool final_iter_no_increase = true;
  double rel_thresh = 1e-2;
  std::size_t max_iter = 50;

  explicit TrajOptimizerSetting(std::size_t system_dof)
      : dof(system_dof), joint_pos_limits_up(system_dof, 1e6), joint_pos_limits_down(system_dof, -1e6),
        vel_limits(system_dof, 1e6), pos_limit_thresh(system_dof, 0.001), vel_limit_thresh(system_dof, 0.001),
        pos_limit_sigmas(system_dof, 0.001), vel_limit_sigmas(system_dof, 0.001) {}

  void set_total_step(std::size_t step) { total_step = step; }
  void set_total_time(double time) { total_time = time; }
  void set_conf_prior_model(double sigma) { conf_prior_sigma = sigma; }
  void set_vel_prior_model(double sigma) { vel_prior_sigma = sigma; }
  void set_flag_pos_limit(bool flag) { flag_pos_limit = flag; }
  void set_flag_vel_limit(bool flag) { flag_vel_limit = flag; }
  void set_joint_pos_limits_up(const Vector& v) { joint_pos_limits_up = v; }
  void set_joint_pos_limits_down(const Vector& v) { joint_pos_limits_down = v; }
  void set_vel_limits(const Vector& v) { vel_limits = v; }
  void set_pos_limit_thresh(const Vector& v) { pos_limit_thresh = v; }
  void set_vel_limit_thresh(const Vector& v) { vel_limit_thresh = v; }
  void set_pos_limit_model(const Vector& v) { pos_limit_sigmas = v; }
  void set_vel_limit_model(const Vector& v) { vel_limit_sigmas = v; }
  void set_epsilon(double eps) { epsilon = eps; }
  void set_cost_sigma(double sigma) { cost_sigma = sigma; }
  void set_obs_check_inter(std::size_t inter) { obs_check_inter = inter; }
  void set_Qc_model(const Vector& Qc_row_major) { Qc = Qc_row_major; }
  void setGaussNewton() { opt_type = GaussNewton; }
  void setLM() { opt_type = LM; }
  void setDogleg() { opt_type = Dogleg; }
  void set_rel_thresh(double thresh) { rel_thresh = thresh; }
  void set_max_iter(std::size_t iter) { max_iter = iter; }
  void setVerbosityNone() { opt_verbosity = None; }
  void setVerbosityError() { opt_verbosity = Error; }
  void setOptimizationNoIncrase(bool flag) { final_iter_no_increase = flag; }

  gpmp2mi_settings c_struct() const {
    auto fits = [&](const Vector& v, const char* n) {
      if (v.size() != dof) throw std::runtime_error(std::string("[TrajOptimizerSetting] ") + n + " dim does not fit dof");
      return v.data();
    };
    gpmp2mi_settings s;
    gpmp2mi_settings_default(&s, static_cast<int>(dof));
    s.total_step = static_cast<int>(total_step);
    s.total_time = total_time;
    s.conf_prior_sigma = conf_prior_sigma;
    s.vel_prior_sigma = vel_prior_sigma;
    s.flag_pos_limit = flag_pos_limit;
    s.flag_vel_limit = flag_vel_limit;
    s.joint_pos_limits_up = fits(joint_pos_limits_up, "joint_pos_limits_up");
    s.joint_pos_limits_down = fits(joint_pos_limits_down, "joint_pos_limits_down");
    s.vel_limits = fits(vel_limits, "vel_limits");
    s.pos_limit_thresh = fits(pos_limit_thresh, "pos_limit_thresh");
    s.vel_limit_thresh = fits(vel_limit_thresh, "vel_limit_thresh");
    s.pos_limit_sigmas = fits(pos_limit_sigmas, "pos_limit_model");
    s.vel_limit_sigmas = fits(vel_limit_sigmas, "vel_limit_model");
    s.epsilon = epsilon;
    s.cost_sigma = cost_sigma;
    s.obs_check_inter = static_cast<int>(obs_check_inter);
    if (!Qc.empty()) {
      if (Qc.size() != dof * dof) throw std::runtime_error("[TrajOptimizerSetting] Qc dim does not fit dof");
      s.Qc = Qc.data();
    }
    s.opt_type = opt_type;
    s.verbosity = opt_verbosity;
    s.final_iter_no_increase = final_iter_no_increase;
    s.rel_thresh = rel_thresh;
    s.max_iter = static_cast<int>(max_iter);
    return s;
  }
};

/// flat stand-in for gtsam::Values: state(i) = [x_i ; v_i]
struct Trajectory {
  std::size_t dof = 0, total_step = 0;
  Vector data;  // [total_step + 1][2 * dof]
  Trajectory() {}
  Trajectory(std::size_t dof_, std::size_t total_step_) : dof(dof_), total_step(total_step_), data((total_step_ + 1) * 2 * dof_, 0.0) {}
  double* x(std::size_t i) { return &data[i * 2 * dof]; }
  double* v(std::size_t i) { return &data[i * 2 * dof + dof]; }
  const double* x(std::size_t i) const { return &data[i * 2 * dof]; }
  const double* v(std::size_t i) const { return &data[i * 2 * dof + dof]; }
};

/// gpmp2::initArmTrajStraightLine (velocity = (end - init) / total_step, TrajUtils.cpp:45)
inline Trajectory initArmTrajStraightLine(const Vector& init_conf, const Vector& end_conf, std::size_t total_step) {
  const std::size_t D = init_conf.size();
  Trajectory t(D, total_step);
  for (std::size_t i = 0; i <= total_step; i++)
    for (std::size_t k = 0; k < D; k++) {
      const double r = static_cast<double>(i) / static_cast<double>(total_step);
      t.x(i)[k] = (i == 0) ? init_conf[k] : (i == total_step) ? end_conf[k] : r * end_conf[k] + (1.0 - r) * init_conf[k];
      t.v(i)[k] = (end_conf[k] - init_conf[k]) / static_cast<double>(total_step);
    }
  return t;
}

namespace internal {
inline Trajectory BatchTrajOptimize(const gpmp2mi_robot* robot, const gpmp2mi_sdf* sdf, std::size_t dof,
                                    const Vector& start_conf, const Vector& start_vel, const Vector& end_conf,
                                    const Vector& end_vel, const Trajectory& init_values,
                                    const TrajOptimizerSetting& setting, int* iterations, double* final_error) {
  if (init_values.dof != dof || init_values.total_step != setting.total_step)
    throw std::runtime_error("[BatchTrajOptimize] init_values do not match dof / total_step");
  const gpmp2mi_settings s = setting.c_struct();
  Trajectory out(dof, setting.total_step);
  int status = 0;
  check(gpmp2mi_batch_optimize(robot, sdf, &s, nullptr, 1, start_conf.data(), start_vel.data(), end_conf.data(),
                               end_vel.data(), init_values.data.data(), out.data.data(), iterations, final_error,
                               &status),
        "gpmp2mi_batch_optimize");
  if (status == GPMP2MI_TRAJ_NOT_SPD) throw std::runtime_error("[gpmp2mi] IndeterminantLinearSystemException");
  return out;
}
}  // namespace internal

/// gpmp2::BatchTrajOptimize3DArm  gpmp2/planner/BatchTrajOptimizer.cpp:53-63
inline Trajectory BatchTrajOptimize3DArm(const ArmModel& arm, const SignedDistanceField& sdf, const Vector& start_conf,
                                         const Vector& start_vel, const Vector& end_conf, const Vector& end_vel,
                                         const Trajectory& init_values, const TrajOptimizerSetting& setting,
                                         int* iterations = nullptr, double* final_error = nullptr) {
  return internal::BatchTrajOptimize(arm.handle(), sdf.handle(), arm.dof(), start_conf, start_vel, end_conf, end_vel,
                                     init_values, setting, iterations, final_error);
}
/// gpmp2::BatchTrajOptimize2DArm  gpmp2/planner/BatchTrajOptimizer.cpp:40-50
inline Trajectory BatchTrajOptimize2DArm(const ArmModel& arm, const PlanarSDF& sdf, const Vector& start_conf,
                                         const Vector& start_vel, const Vector& end_conf, const Vector& end_vel,
                                         const Trajectory& init_values, const TrajOptimizerSetting& setting,
                                         int* iterations = nullptr, double* final_error = nullptr) {
  return internal::BatchTrajOptimize(arm.handle(), sdf.handle(), arm.dof(), start_conf, start_vel, end_conf, end_vel,
                                     init_values, setting, iterations, final_error);
}
/// gpmp2::BatchTrajOptimizePose2MobileArm  gpmp2/planner/BatchTrajOptimizer.cpp:79-89
inline Trajectory BatchTrajOptimizePose2MobileArm(const Pose2MobileArmModel& marm, const SignedDistanceField& sdf,
                                                  const Vector& start_conf, const Vector& start_vel, const Vector& end_conf,
                                                  const Vector& end_vel, const Trajectory& init_values,
                                                  const TrajOptimizerSetting& setting, int* iterations = nullptr,
                                                  double* final_error = nullptr) {
  return internal::BatchTrajOptimize(marm.handle(), sdf.handle(), marm.dof(), start_conf, start_vel, end_conf, end_vel,
                                     init_values, setting, iterations, final_error);
}
/// gpmp2::BatchTrajOptimizePose2MobileArm2D  gpmp2/planner/BatchTrajOptimizer.cpp:66-76
inline Trajectory BatchTrajOptimizePose2MobileArm2D(const Pose2MobileArmModel& marm, const PlanarSDF& sdf,
                                                    const Vector& start_conf, const Vector& start_vel,
                                                    const Vector& end_conf, const Vector& end_vel,
                                                    const Trajectory& init_values, const TrajOptimizerSetting& setting,
                                                    int* iterations = nullptr, double* final_error = nullptr) {
  return internal::BatchTrajOptimize(marm.handle(), sdf.handle(), marm.dof(), start_conf, start_vel, end_conf, end_vel,
                                     init_values, setting, iterations, final_error);
}

// ---- factors: evaluateError(x..., H...) of the reference's NoiseModelFactors, one evaluation per call ----------
namespace internal {
template <class ROBOT, class SDF>
class ObstacleSDFFactor {  // gpmp2/obstacle/ObstacleSDFFactor.h:27-100, ObstaclePlanarSDFFactor.h:27-98
 public:
  ObstacleSDFFactor(std::size_t /*poseKey*/, const ROBOT& robot, const SDF& sdf, double /*cost_sigma*/, double epsilon)
      : robot_(robot), sdf_(sdf), epsilon_(epsilon) {}
  /// unwhitened error [nr_body_spheres]; H1 (optional) row-major [nr_body_spheres][dof]
  Vector evaluateError(const Vector& conf, Vector* H1 = nullptr) const {
    if (conf.size() != robot_.dof()) throw std::runtime_error("[ObstacleSDFFactor] conf dim does not fit dof");
    Vector err(robot_.nr_body_spheres());
    if (H1) H1->assign(err.size() * robot_.dof(), 0.0);
    check(gpmp2mi_obstacle_factor(robot_.handle(), sdf_.handle(), epsilon_, 1, conf.data(), err.data(),
                                  H1 ? H1->data() : nullptr),
          "gpmp2mi_obstacle_factor");
    return err;
  }

 private:
  const ROBOT& robot_;
  const SDF& sdf_;
  double epsilon_;
};
}  // namespace internal
typedef internal::ObstacleSDFFactor<ArmModel, SignedDistanceField> ObstacleSDFFactorArm;
typedef internal::ObstacleSDFFactor<ArmModel, PlanarSDF> ObstaclePlanarSDFFactorArm;
typedef internal::ObstacleSDFFactor<Pose2MobileArmModel, SignedDistanceField> ObstacleSDFFactorPose2MobileArm;
typedef internal::ObstacleSDFFactor<Pose2MobileArmModel, PlanarSDF> ObstaclePlanarSDFFactorPose2MobileArm;

/// gpmp2::GoalFactorArm  gpmp2/kinematics/GoalFactorArm.h:24-100
class GoalFactorArm {
 public:
  GoalFactorArm(std::size_t /*poseKey*/, const ArmModel& arm, const std::array<double, 3>& dest_point)
      : arm_(arm), dest_(dest_point) {}
  Vector evaluateError(const Vector& conf, Vector* H1 = nullptr) const {
    Vector err(3);
    if (H1) H1->assign(3 * arm_.dof(), 0.0);
    check(gpmp2mi_goal_factor_arm(arm_.handle(), dest_.data(), 1, conf.data(), err.data(), H1 ? H1->data() : nullptr),
          "gpmp2mi_goal_factor_arm");
    return err;
  }

 private:
  const ArmModel& arm_;
  std::array<double, 3> dest_;
};

/// gpmp2::GaussianPriorWorkspacePoseArm  gpmp2/kinematics/GaussianPriorWorkspacePose.h:24-93
class GaussianPriorWorkspacePoseArm {
 public:
  GaussianPriorWorkspacePoseArm(std::size_t /*poseKey*/, const ArmModel& arm, int joint, const Pose3& des_pose)
      : arm_(arm), joint_(joint), des_(des_pose) {}
  Vector evaluateError(const Vector& conf, Vector* H1 = nullptr) const {
    Vector err(6);
    if (H1) H1->assign(6 * arm_.dof(), 0.0);
    check(gpmp2mi_workspace_prior_factor(arm_.handle(), GPMP2MI_WORKSPACE_POSE, joint_, des_.m.data(), 1, conf.data(),
                                         err.data(), H1 ? H1->data() : nullptr),
          "gpmp2mi_workspace_prior_factor");
    return err;
  }

 private:
  const ArmModel& arm_;
  int joint_;
  Pose3 des_;
};

/// gpmp2::SelfCollisionArm  gpmp2/obstacle/SelfCollision.h:27-140; data rows = (sphere A, sphere B, epsilon, sigma)
class SelfCollisionArm {
 public:
  SelfCollisionArm(std::size_t /*poseKey*/, const ArmModel& arm, const Vector& data_row_major) : arm_(arm), data_(data_row_major) {
    if (data_.size() % 4) throw std::runtime_error("[SelfCollision] data must have 4 columns");
  }
  Vector evaluateError(const Vector& conf, Vector* H = nullptr) const {
    const int np = static_cast<int>(data_.size() / 4);
    Vector err(np);
    if (H) H->assign(np * arm_.dof(), 0.0);
    check(gpmp2mi_self_collision_factor(arm_.handle(), np, data_.data(), 1, conf.data(), err.data(), H ? H->data() : nullptr),
          "gpmp2mi_self_collision_factor");
    return err;
  }

 private:
  const ArmModel& arm_;
  Vector data_;
};

/// gpmp2::CollisionCost3DArm / 2DArm  gpmp2/planner/BatchTrajOptimizer-inl.h:87-100
inline double CollisionCost3DArm(const ArmModel& arm, const SignedDistanceField& sdf, const Trajectory& result,
                                 const TrajOptimizerSetting&) {
  double c = 0;
  check(gpmp2mi_collision_cost(arm.handle(), sdf.handle(), static_cast<int>(result.total_step), 1, result.data.data(), &c),
        "gpmp2mi_collision_cost");
  return c;
}
inline double CollisionCost2DArm(const ArmModel& arm, const PlanarSDF& sdf, const Trajectory& result,
                                 const TrajOptimizerSetting&) {
  double c = 0;
  check(gpmp2mi_collision_cost(arm.handle(), sdf.handle(), static_cast<int>(result.total_step), 1, result.data.data(), &c),
        "gpmp2mi_collision_cost");
  return c;
}

namespace internal {
inline Trajectory interpolateTraj(const Trajectory& opt_values, const Vector& Qc, double delta_t, std::size_t inter_step,
                                  std::size_t start_index, std::size_t end_index, bool lie) {
  if (!Qc.empty() && Qc.size() != opt_values.dof * opt_values.dof)
    throw std::runtime_error("[interpolateArmTraj] Qc dim does not fit dof");
  if (start_index >= end_index || end_index > opt_values.total_step)
    throw std::runtime_error("[interpolateArmTraj] need start_index < end_index <= total_step");
  Trajectory out(opt_values.dof, (end_index - start_index) * (inter_step + 1));
  check(gpmp2mi_interpolate_traj(static_cast<int>(opt_values.dof), lie ? 1 : 0, Qc.empty() ? nullptr : Qc.data(), delta_t,
                                 static_cast<int>(inter_step), 1, static_cast<int>(opt_values.total_step),
                                 static_cast<int>(start_index), static_cast<int>(end_index), opt_values.data.data(),
                                 out.data.data()),
        "gpmp2mi_interpolate_traj");
  return out;
}
}  // namespace internal
/// gpmp2::interpolateArmTraj  gpmp2/planner/TrajUtils.cpp:96-159 (Qc row-major [dof][dof], may be empty)
inline Trajectory interpolateArmTraj(const Trajectory& opt_values, const Vector& Qc, double delta_t, std::size_t inter_step) {
  return internal::interpolateTraj(opt_values, Qc, delta_t, inter_step, 0, opt_values.total_step, false);
}
/// gpmp2::interpolateArmTraj with a state range  gpmp2/planner/TrajUtils.cpp:162-197
inline Trajectory interpolateArmTraj(const Trajectory& opt_values, const Vector& Qc, double delta_t, std::size_t inter_step,
                                     std::size_t start_index, std::size_t end_index) {
  return internal::interpolateTraj(opt_values, Qc, delta_t, inter_step, start_index, end_index, false);
}
/// gpmp2::interpolatePose2MobileArmTraj  gpmp2/planner/TrajUtils.cpp:200-236 (states [x, y, theta, q...])
inline Trajectory interpolatePose2MobileArmTraj(const Trajectory& opt_values, const Vector& Qc, double delta_t,
                                                std::size_t inter_step, std::size_t start_index, std::size_t end_index) {
  return internal::interpolateTraj(opt_values, Qc, delta_t, inter_step, start_index, end_index, true);
}

namespace internal {
/// gpmp2::internal::ISAM2TrajOptimizer  gpmp2/planner/ISAM2TrajOptimizer.h:58-137
/// Same call sequence as the reference (initFactorGraph, initValues, update, then the replanning
/// interface and update again); each update() is one full relinearise + solve of the resident plan
/// (gpmp2mi_plan_update), not an iSAM2 partial update -- see include/gpmp2mi.h.
template <class ROBOT, class SDF>
class ISAM2TrajOptimizer {
 public:
  ISAM2TrajOptimizer(const ROBOT& arm, const SDF& sdf, const TrajOptimizerSetting& setting)
      : dof_(arm.dof()), setting_(setting) {
    const gpmp2mi_settings s = setting_.c_struct();
    check(gpmp2mi_plan_create(arm.handle(), sdf.handle(), &s, nullptr, 1, &plan_), "gpmp2mi_plan_create");
  }
  ~ISAM2TrajOptimizer() {
    if (plan_) gpmp2mi_plan_destroy(plan_);
  }
  ISAM2TrajOptimizer(const ISAM2TrajOptimizer&) = delete;
  ISAM2TrajOptimizer& operator=(const ISAM2TrajOptimizer&) = delete;

  /// ISAM2TrajOptimizer-inl.h:28-84
  void initFactorGraph(const Vector& start_conf, const Vector& start_vel, const Vector& goal_conf, const Vector& goal_vel) {
    start_conf_ = start_conf, start_vel_ = start_vel, goal_conf_ = goal_conf, goal_vel_ = goal_vel;
    fits(start_conf), fits(start_vel), fits(goal_conf), fits(goal_vel);
    have_graph_ = true;
  }
  /// ISAM2TrajOptimizer-inl.h:90-96
  void initValues(const Trajectory& init_values) {
    if (!have_graph_) throw std::runtime_error("[ISAM2TrajOptimizer] initFactorGraph must come first");
    if (init_values.dof != dof_ || init_values.total_step != setting_.total_step)
      throw std::runtime_error("[ISAM2TrajOptimizer] init_values do not match dof / total_step");
    check(gpmp2mi_plan_set_problem(plan_, start_conf_.data(), start_vel_.data(), goal_conf_.data(), goal_vel_.data(),
                                   init_values.data.data()),
          "gpmp2mi_plan_set_problem");
    opt_values_ = init_values;
  }
  /// ISAM2TrajOptimizer-inl.h:102-114
  void update() {
    check(gpmp2mi_plan_update(plan_, 1, nullptr), "gpmp2mi_plan_update");
    int status = 0;
    check(gpmp2mi_plan_get_result(plan_, opt_values_.data.data(), nullptr, nullptr, &status, nullptr),
          "gpmp2mi_plan_get_result");
    if (status == GPMP2MI_TRAJ_NOT_SPD) throw std::runtime_error("[gpmp2mi] IndeterminantLinearSystemException");
  }
  /// ISAM2TrajOptimizer-inl.h:120-141
  void changeGoalConfigAndVel(const Vector& goal_conf, const Vector& goal_vel) {
    fits(goal_conf), fits(goal_vel);
    check(gpmp2mi_plan_change_goal(plan_, 0, goal_conf.data(), goal_vel.data()), "gpmp2mi_plan_change_goal");
  }
  /// ISAM2TrajOptimizer-inl.h:147-153
  void removeGoalConfigAndVel() { check(gpmp2mi_plan_remove_goal(plan_, 0), "gpmp2mi_plan_remove_goal"); }
  /// ISAM2TrajOptimizer-inl.h:159-168
  void fixConfigAndVel(std::size_t state_idx, const Vector& conf_fix, const Vector& vel_fix) {
    fits(conf_fix), fits(vel_fix);
    check(gpmp2mi_plan_fix_state(plan_, 0, static_cast<int>(state_idx), conf_fix.data(), vel_fix.data()),
          "gpmp2mi_plan_fix_state");
  }
  /// ISAM2TrajOptimizer-inl.h:174-180; pose_cov row-major [dof][dof]
  void addPoseEstimate(std::size_t state_idx, const Vector& pose, const Vector& pose_cov) {
    fits(pose);
    if (pose_cov.size() != dof_ * dof_) throw std::runtime_error("[ISAM2TrajOptimizer] covariance dim does not fit dof");
    check(gpmp2mi_plan_add_state_estimate(plan_, 0, static_cast<int>(state_idx), pose.data(), pose_cov.data(), nullptr,
                                          nullptr),
          "gpmp2mi_plan_add_state_estimate");
  }
  /// ISAM2TrajOptimizer-inl.h:186-195
  void addStateEstimate(std::size_t state_idx, const Vector& pose, const Vector& pose_cov, const Vector& vel,
                        const Vector& vel_cov) {
    fits(pose), fits(vel);
    if (pose_cov.size() != dof_ * dof_ || vel_cov.size() != dof_ * dof_)
      throw std::runtime_error("[ISAM2TrajOptimizer] covariance dim does not fit dof");
    check(gpmp2mi_plan_add_state_estimate(plan_, 0, static_cast<int>(state_idx), pose.data(), pose_cov.data(),
                                          vel.data(), vel_cov.data()),
          "gpmp2mi_plan_add_state_estimate");
  }
  const Trajectory& values() const { return opt_values_; }

 private:
  void fits(const Vector& v) const {
    if (v.size() != dof_) throw std::runtime_error("[ISAM2TrajOptimizer] vector dim does not fit dof");
  }
  std::size_t dof_;
  TrajOptimizerSetting setting_;
  gpmp2mi_plan* plan_ = nullptr;
  Vector start_conf_, start_vel_, goal_conf_, goal_vel_;
  Trajectory opt_values_;
  bool have_graph_ = false;
};
}  // namespace internal
/// gpmp2/planner/ISAM2TrajOptimizer.h:143-156
typedef internal::ISAM2TrajOptimizer<ArmModel, PlanarSDF> ISAM2TrajOptimizer2DArm;
typedef internal::ISAM2TrajOptimizer<ArmModel, SignedDistanceField> ISAM2TrajOptimizer3DArm;

#ifdef GPMP2MI_HAVE_GTSAM
/// gtsam::Values (keys Symbol('x', i) / Symbol('v', i), gpmp2/planner/BatchTrajOptimizer.h:39-41) <-> Trajectory
inline Trajectory fromValues(const gtsam::Values& values, std::size_t dof, std::size_t total_step) {
  Trajectory t(dof, total_step);
  for (std::size_t i = 0; i <= total_step; i++) {
    const gtsam::Vector x = values.at<gtsam::Vector>(gtsam::Symbol('x', i));
    const gtsam::Vector v = values.at<gtsam::Vector>(gtsam::Symbol('v', i));
    for (std::size_t k = 0; k < dof; k++) {
      t.x(i)[k] = x(k);
      t.v(i)[k] = v(k);
    }
  }
  return t;
}
inline gtsam::Values toValues(const Trajectory& t) {
  gtsam::Values values;
  for (std::size_t i = 0; i <= t.total_step; i++) {
    gtsam::Vector x(t.dof), v(t.dof);
    for (std::size_t k = 0; k < t.dof; k++) {
      x(k) = t.x(i)[k];
      v(k) = t.v(i)[k];
    }
    values.insert(gtsam::Symbol('x', i), x);
    values.insert(gtsam::Symbol('v', i), v);
  }
  return values;
}
#endif

}  // namespace gpmp2mi
