// oracle_core.h -- CPU restatement of the GPMP2 hot path (TEST INFRASTRUCTURE ONLY).
//
// This directory is the parity oracle of SURVEY.md section 8(c): a plain, dependency-free fp64
// restatement of the reference algorithm, written from the reference sources cited at every
// function.  It is NOT part of the product: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load it.  The shipped library (gpmp2_amd/csrc) never links,
// imports or calls anything in here.
//
// Parity status: factor arithmetic is PINNED by the reference's known-answer tests (tests/golden,
// SURVEY.md appendix D).  Whole-trajectory solves, LM / Dogleg step control and anything that
// lives in un-vendored GTSAM are "parity unpinned" (restated from upstream semantics, SURVEY.md
// appendix B); they are cross-checked against an independent dense numpy solve only.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- tiny dense matrix (row major)
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
  static Mat identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; i++) m(i, i) = 1.0;
    return m;
  }
};
Mat matmul(const Mat& A, const Mat& B);
Mat transpose(const Mat& A);
Mat inverse(const Mat& A);          // Gauss-Jordan, partial pivoting
Mat chol_upper(const Mat& W);       // R with R^T R = W
Mat operator+(const Mat& A, const Mat& B);
Mat operator-(const Mat& A, const Mat& B);
Mat operator*(double s, const Mat& A);

// ---------------------------------------------------------------- robot / sdf descriptions
enum RobotKind { ARM = 0, POINT = 1, MOBILE_BASE = 2, MOBILE_ARM = 3, MOBILE_2ARMS = 4, MOBILE_VETLIN_ARM = 5, MOBILE_VETLIN_2ARMS = 6 };

struct Robot {
  int kind = ARM, dof = 0, arm_dof = 0;
  std::vector<double> a, alpha, d, bias;
  double base[16];  // row-major 4x4
  int arm2_dof = 0;       // two-arm robots: the last arm2_dof DH joints belong to arm 2
  double base2[16], base3[16];
  bool reverse_linact = false;
  bool has_lift() const { return kind == MOBILE_VETLIN_ARM || kind == MOBILE_VETLIN_2ARMS; }
  std::vector<int> sph_link;
  std::vector<double> sph_r, sph_c;  // radius [S], centre [S][3]
  int nr_links() const {
    return kind == ARM ? arm_dof : kind == POINT ? 1 : kind == MOBILE_BASE ? 1 : arm_dof + 1 + (has_lift() ? 1 : 0);
  }
  int nr_spheres() const { return (int)sph_r.size(); }
  bool is_lie() const { return kind >= MOBILE_BASE; }
};

struct Sdf {
  int dim = 3;
  double origin[3] = {0, 0, 0};
  double cell = 1.0;
  int nx = 0, ny = 0, nz = 1;  // cols, rows, z
  std::vector<double> v;       // [(z*ny + y)*nx + x]
  double at(int row, int col, int z) const { return v[((size_t)z * ny + row) * nx + col]; }
};

// signed field of an occupancy grid (matlab/+gpmp2/signedDistanceField3D.m:16-34): occ, field [nz][ny][nx]
void sdf_from_occupancy(int nx, int ny, int nz, const double* occ, double cell, double* field);

// ---------------------------------------------------------------- GP (gpmp2/gp/GPutils.h)
Mat calcQ(const Mat& Qc, double tau);
Mat calcQ_inv(const Mat& Qc, double tau);
Mat calcPhi(int dof, double tau);
Mat calcLambda(const Mat& Qc, double delta_t, double tau);
Mat calcPsi(const Mat& Qc, double delta_t, double tau);

// ---------------------------------------------------------------- Pose2 helpers (GTSAM semantics)
struct Pose2 {
  double x = 0, y = 0, th = 0;
};
Pose2 pose2_compose(const Pose2& a, const Pose2& b);
Pose2 pose2_inverse(const Pose2& a);
Pose2 pose2_between(const Pose2& a, const Pose2& b);
void pose2_logmap(const Pose2& p, double v[3]);
Pose2 pose2_expmap(const double v[3]);
Pose2 pose2_retract(const Pose2& p, const double v[3]);  // GTSAM default (non-SLOW) chart
void pose2_adjoint(const Pose2& p, double A[9]);
void pose2_expmap_derivative(const double v[3], double H[9]);
void pose2_logmap_derivative(const Pose2& p, double H[9]);

// ---------------------------------------------------------------- kinematics
// poses [L][16]; Jpose [L][6][dof] (may be null)
void forward_kinematics(const Robot& R, const double* conf, double* poses, double* Jpose);
// centers [S][3]; J [S][3][dof] (may be null)
void sphere_centers(const Robot& R, const double* conf, double* centers, double* J);

// ---------------------------------------------------------------- sdf + hinge
// returns false when the reference throws SDFQueryOutOfRange
bool sdf_query(const Sdf& s, const double* p, double* dist, double* grad /*dim, may be null*/);
double hinge_obstacle(const Sdf& s, const double* p, double eps, double* Hp /*dim or null*/);
double hinge_limit(double p, double lo, double hi, double th, double* H);

// ---------------------------------------------------------------- factors (unwhitened)
void obstacle_factor(const Robot& R, const Sdf& s, double eps, const double* conf, double* err,
                     double* H1 /*[S][dof] or null*/);
struct GPInterp {
  int dof = 0;
  bool lie = false;
  double delta_t = 0, tau = 0;
  Mat Qc, Lambda, Psi;
  GPInterp() {}
  GPInterp(int dof, bool lie, const Mat& Qc, double delta_t, double tau);
  // conf [dof]; H1..H4 [dof][dof] or null
  void interpolate_pose(const double* c1, const double* v1, const double* c2, const double* v2,
                        double* conf, Mat* H1, Mat* H2, Mat* H3, Mat* H4) const;
  void interpolate_velocity(const double* c1, const double* v1, const double* c2,
                            const double* v2, double* vel) const;
};
void obstacle_gp_factor(const Robot& R, const Sdf& s, double eps, const GPInterp& gp,
                        const double* c1, const double* v1, const double* c2, const double* v2,
                        double* err, double* H1, double* H2, double* H3, double* H4);
// err [2 dof]; H1..H4 [2 dof][dof] or null
void gp_prior_factor(int dof, bool lie, double delta_t, const double* c1, const double* v1,
                     const double* c2, const double* v2, double* err, Mat* H1, Mat* H2, Mat* H3,
                     Mat* H4);

// ---------------------------------------------------------------- workspace / self-collision factors
// Rot3 / Pose3 log maps with GTSAM 4.0 semantics (upstream, restated; pinned through the known answers of
// kinematics/tests/testGaussianPriorWorkspace{Pose,Orientation}.cpp)
void rot3_logmap(const double R[9], double w[3]);
void rot3_logmap_derivative(const double w[3], double H[9]);
void pose3_logmap(const double R[9], const double t[3], double xi[6]);
void pose3_logmap_derivative(const double R[9], const double t[3], double H[36]);
enum WorkspaceMode { WS_POSITION = 0, WS_ORIENTATION = 1, WS_POSE = 2 };
// GaussianPriorWorkspace{Position,Orientation,Pose}::evaluateError (kinematics/GaussianPriorWorkspacePose.h:53-70
// and siblings); des = row-major 4x4; err [3|3|6]; H [rows][dof] or null
void workspace_prior_factor(const Robot& R, int mode, int joint, const double des[16], const double* conf,
                            double* err, double* H);
// SelfCollision::evaluateError obstacle/SelfCollision.h:66-128; data [n][4] = (sphere A, sphere B, eps, sigma)
void self_collision_factor(const Robot& R, int n_pairs, const double* data, const double* conf, double* err,
                           double* H /*[n][dof] or null*/);

// ---------------------------------------------------------------- settings / graph / optimizer
struct Settings {
  int dof = 0, total_step = 10;
  double total_time = 1.0, conf_prior_sigma = 1e-4, vel_prior_sigma = 1e-4;
  bool flag_pos_limit = false, flag_vel_limit = false;
  std::vector<double> pos_up, pos_down, vel_limits, pos_thresh, vel_thresh, pos_sigmas, vel_sigmas;
  double epsilon = 0.2, cost_sigma = 0.1;
  int obs_check_inter = 5;
  Mat Qc;
  int opt_type = 2;  // 0 GN, 1 LM, 2 Dogleg
  int verbosity = 0;
  bool final_iter_no_increase = true;
  double rel_thresh = 1e-2;
  int max_iter = 50;
  // graph opts
  bool obs_skip_first = false;
  double vehicle_dynamics_sigma = 0;
  double lm_lambda_initial = 100, lm_lambda_factor = 10, lm_lambda_upper = 1e5,
         lm_lambda_lower = 0, lm_min_model_fidelity = 1e-3;
  double dogleg_delta_initial = 0.2, abs_error_tol = 1e-5, error_tol = 0;
  int fixed_iterations = 0;
  // extra factors of hand-built graphs (matlab/Arm3GoalReachExample.m:95-110,
  // matlab/WAMWorkspaceConstraintsExample.m:85-105, obstacle/SelfCollision.h:66-128)
  bool end_conf_prior_off = false;
  struct WorkspaceFactor {
    int mode = 0, link = 0, first_state = 0, last_state = 0;
    double sigma = 1.0;
    double des[16] = {0};
  };
  std::vector<WorkspaceFactor> workspace;
  std::vector<double> self_collision;  // [n][4]
  int self_collision_first = 0, self_collision_last = 0;
  // replanner state (planner/ISAM2TrajOptimizer-inl.h:118-195)
  bool goal_on = true;
  struct StatePrior {
    int state = 0;
    bool has_vel = false;
    std::vector<double> conf, vel;  // targets
    Mat Wc, Wv;                     // information matrices
  };
  std::vector<StatePrior> state_priors;
};

// one whitened factor touching states [s0, s0 + ns) with m rows: A [m][ns * 2 dof], b [m] = r
struct LinFactor {
  int s0 = 0, ns = 1, m = 0;
  std::vector<double> A, r;
};

struct Problem {
  const Robot* robot = nullptr;
  const Sdf* sdf = nullptr;
  Settings set;
  std::vector<double> start_conf, start_vel, end_conf, end_vel;
  // derived
  double delta_t = 0;
  std::vector<GPInterp> interp;  // per sub-step j=1..I
  Mat Qinv, Rgp;                 // GP prior information and its upper Cholesky factor
  void prepare();
  int n() const { return 2 * set.dof; }
  int nstates() const { return set.total_step + 1; }
  // whitened linearization at traj ([N+1][2 dof]); if factors == null only the error is computed
  double linearize(const double* traj, std::vector<LinFactor>* factors) const;
  double error(const double* traj) const { return linearize(traj, nullptr); }
  void retract(const double* traj, const double* delta, double* out) const;
};

struct NormalEq {  // block tridiagonal
  int nblk = 0, n = 0;
  std::vector<double> D, O, g;  // D [nblk][n][n], O [nblk-1][n][n] (block (i+1,i)), g [nblk][n]
  void assemble(const std::vector<LinFactor>& f, int nblk, int n);
  // solves (H + lambda I) x = -g ; returns false if a pivot is not positive
  bool solve(double lambda, double* x) const;
  double quad(const double* x) const;      // g^T x + 0.5 x^T H x
  void times(const double* x, double* y) const;  // y = H x
};

struct OptResult {
  int iterations = 0, status = 0;
  double final_error = 0;
  std::vector<double> trace;  // error before each iteration; [0] = initial
};
// gpmp2::optimize (planner/BatchTrajOptimizer.cpp:212-308) with GTSAM GN / LM / Dogleg semantics
OptResult optimize(const Problem& P, const double* init, double* out);
// test probe of the Dogleg trial points (see oracle_core.cpp); buf = nullptr switches it off
void set_dogleg_probe(double* buf, int cap_rows);
int dogleg_probe_rows();

}  // namespace orc
