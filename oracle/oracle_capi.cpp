// oracle_capi.cpp -- extern "C" surface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
// Mirrors include/gpmp2mi.h one-to-one with the prefix `orc_` so that parity tests call the
// oracle and the HIP library with identical arguments.  Never linked into the product.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/gpmp2mi.h"
#include "oracle_core.h"

using namespace orc;

static void fill_robot(const gpmp2mi_robot_desc* d, Robot& R) {
  R.kind = d->kind;
  R.dof = d->dof;
  R.arm_dof = d->arm_dof;
  R.a.assign(d->a ? d->a : nullptr, d->a ? d->a + d->arm_dof : nullptr);
  R.alpha.assign(d->alpha ? d->alpha : nullptr, d->alpha ? d->alpha + d->arm_dof : nullptr);
  R.d.assign(d->d ? d->d : nullptr, d->d ? d->d + d->arm_dof : nullptr);
  if (d->theta_bias) R.bias.assign(d->theta_bias, d->theta_bias + d->arm_dof);
  else R.bias.assign(d->arm_dof, 0.0);
  std::memcpy(R.base, d->base_pose, sizeof(R.base));
  R.arm2_dof = d->arm2_dof;
  std::memcpy(R.base2, d->base_pose2, sizeof(R.base2));
  std::memcpy(R.base3, d->base_pose3, sizeof(R.base3));
  R.reverse_linact = d->reverse_linact != 0;
  R.sph_link.assign(d->sphere_link, d->sphere_link + d->nr_spheres);
  R.sph_r.assign(d->sphere_radius, d->sphere_radius + d->nr_spheres);
  R.sph_c.assign(d->sphere_center, d->sphere_center + 3 * d->nr_spheres);
}

static void fill_settings(const gpmp2mi_settings* s, const gpmp2mi_graph_opts* o, Settings& S) {
  const int d = s->dof;
  S.dof = d;
  S.total_step = s->total_step;
  S.total_time = s->total_time;
  S.conf_prior_sigma = s->conf_prior_sigma;
  S.vel_prior_sigma = s->vel_prior_sigma;
  S.flag_pos_limit = s->flag_pos_limit != 0;
  S.flag_vel_limit = s->flag_vel_limit != 0;
  auto cp = [&](const double* p, std::vector<double>& v) {
    if (p) v.assign(p, p + d);
  };
  cp(s->joint_pos_limits_up, S.pos_up);
  cp(s->joint_pos_limits_down, S.pos_down);
  cp(s->vel_limits, S.vel_limits);
  cp(s->pos_limit_thresh, S.pos_thresh);
  cp(s->vel_limit_thresh, S.vel_thresh);
  cp(s->pos_limit_sigmas, S.pos_sigmas);
  cp(s->vel_limit_sigmas, S.vel_sigmas);
  S.epsilon = s->epsilon;
  S.cost_sigma = s->cost_sigma;
  S.obs_check_inter = s->obs_check_inter;
  S.Qc = Mat::identity(d);
  if (s->Qc)
    for (int i = 0; i < d * d; i++) S.Qc.a[i] = s->Qc[i];
  S.opt_type = s->opt_type;
  S.verbosity = s->verbosity;
  S.final_iter_no_increase = s->final_iter_no_increase != 0;
  S.rel_thresh = s->rel_thresh;
  S.max_iter = s->max_iter;
  if (o) {
    S.obs_skip_first = o->obs_skip_first_state != 0;
    S.vehicle_dynamics_sigma = o->vehicle_dynamics_sigma;
    S.lm_lambda_initial = o->lm_lambda_initial;
    S.lm_lambda_factor = o->lm_lambda_factor;
    S.lm_lambda_upper = o->lm_lambda_upper;
    S.lm_lambda_lower = o->lm_lambda_lower;
    S.lm_min_model_fidelity = o->lm_min_model_fidelity;
    S.dogleg_delta_initial = o->dogleg_delta_initial;
    S.abs_error_tol = o->abs_error_tol;
    S.error_tol = o->error_tol;
    S.fixed_iterations = o->fixed_iterations;
    S.end_conf_prior_off = o->end_conf_prior_off != 0;
    for (int k = 0; k < o->n_workspace; k++) {
      Settings::WorkspaceFactor w;
      w.mode = o->workspace[k].mode;
      w.link = o->workspace[k].link;
      w.first_state = o->workspace[k].first_state;
      w.last_state = o->workspace[k].last_state;
      w.sigma = o->workspace[k].sigma;
      for (int t = 0; t < 16; t++) w.des[t] = o->workspace[k].des_pose[t];
      S.workspace.push_back(w);
    }
    for (int k = 0; k < o->n_self_collision; k++)
      for (int t = 0; t < 4; t++) S.self_collision.push_back(o->self_collision[k][t]);
    S.self_collision_first = o->self_collision_first;
    S.self_collision_last = o->self_collision_last;
  }
}

static Problem make_problem(const Robot* R, const Sdf* sdf, const gpmp2mi_settings* s,
                            const gpmp2mi_graph_opts* o, const double* sc, const double* sv,
                            const double* ec, const double* ev) {
  Problem P;
  P.robot = R;
  P.sdf = sdf;
  fill_settings(s, o, P.set);
  const int d = s->dof;
  P.start_conf.assign(sc, sc + d);
  P.start_vel.assign(sv, sv + d);
  P.end_conf.assign(ec, ec + d);
  P.end_vel.assign(ev, ev + d);
  P.prepare();
  return P;
}

extern "C" {

int orc_robot_create(const gpmp2mi_robot_desc* d, void** out) {
  Robot* R = new Robot();
  fill_robot(d, *R);
  *out = R;
  return 0;
}
void orc_robot_destroy(void* r) { delete (Robot*)r; }

int orc_sdf_create(int dim, const double* origin, double cell, int nx, int ny, int nz,
                   const double* vox, int layout, void** out) {
  Sdf* s = new Sdf();
  s->dim = dim;
  for (int i = 0; i < 3; i++) s->origin[i] = (i < dim) ? origin[i] : 0.0;
  s->cell = cell;
  s->nx = nx;
  s->ny = ny;
  s->nz = (dim == 3) ? nz : 1;
  s->v.resize((size_t)nx * ny * s->nz);
  for (int z = 0; z < s->nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        const size_t src = (layout == GPMP2MI_SDF_LAYOUT_ZYX) ? ((size_t)z * ny + y) * nx + x
                                                                : ((size_t)z * nx + x) * ny + y;
        s->v[((size_t)z * ny + y) * nx + x] = vox[src];
      }
  *out = s;
  return 0;
}
void orc_sdf_destroy(void* s) { delete (Sdf*)s; }

int orc_sdf_field_from_occupancy(int dim, int nx, int ny, int nz, const double* occ, double cell, double* field) {
  sdf_from_occupancy(nx, ny, dim == 3 ? nz : 1, occ, cell, field);
  return 0;
}

int orc_sdf_query(const void* s_, int M, const double* pts, double* dist, double* grad, int* inr) {
  const Sdf& s = *(const Sdf*)s_;
  for (int m = 0; m < M; m++) {
    double d = 0, g[3] = {0, 0, 0};
    const bool ok = sdf_query(s, pts + (size_t)m * s.dim, &d, g);
    dist[m] = ok ? d : 0.0;
    if (grad)
      for (int i = 0; i < s.dim; i++) grad[(size_t)m * s.dim + i] = ok ? g[i] : 0.0;
    if (inr) inr[m] = ok ? 1 : 0;
  }
  return 0;
}

int orc_forward_kinematics(const void* r, int M, const double* conf, double* poses, double* J) {
  const Robot& R = *(const Robot*)r;
  const int L = R.nr_links(), D = R.dof;
  for (int m = 0; m < M; m++)
    forward_kinematics(R, conf + (size_t)m * D, poses + (size_t)m * L * 16,
                       J ? J + (size_t)m * L * 6 * D : nullptr);
  return 0;
}

int orc_sphere_centers(const void* r, int M, const double* conf, double* c, double* J) {
  const Robot& R = *(const Robot*)r;
  const int S = R.nr_spheres(), D = R.dof;
  for (int m = 0; m < M; m++)
    sphere_centers(R, conf + (size_t)m * D, c + (size_t)m * S * 3,
                   J ? J + (size_t)m * S * 3 * D : nullptr);
  return 0;
}

int orc_workspace_prior_factor(const void* r, int mode, int joint, const double* des, int M, const double* conf,
                               double* err, double* H) {
  const Robot& R = *(const Robot*)r;
  const int rows = mode == WS_POSE ? 6 : 3, D = R.dof;
  for (int m = 0; m < M; m++)
    workspace_prior_factor(R, mode, joint, des, conf + (size_t)m * D, err + (size_t)m * rows,
                           H ? H + (size_t)m * rows * D : nullptr);
  return 0;
}

int orc_self_collision_factor(const void* r, int n_pairs, const double* data, int M, const double* conf, double* err,
                              double* H) {
  const Robot& R = *(const Robot*)r;
  for (int m = 0; m < M; m++)
    self_collision_factor(R, n_pairs, data, conf + (size_t)m * R.dof, err + (size_t)m * n_pairs,
                          H ? H + (size_t)m * n_pairs * R.dof : nullptr);
  return 0;
}

// simple2DVehicleDynamicsPose2 / ...Vector3  dynamics/VehicleDynamics.h:19-40
int orc_vehicle_dynamics_factor(int D, int lie, int M, const double* conf, const double* vel, double* err, double* Hp,
                                double* Hv) {
  for (int m = 0; m < M; m++) {
    const double* p = conf + (size_t)m * D;
    const double* v = vel + (size_t)m * D;
    double hp[3] = {0, 0, 0}, hv[3] = {0, 1, 0}, e = v[1];
    if (!lie) {
      hp[2] = -(v[1] * std::sin(p[2]) + v[0] * std::cos(p[2]));
      hv[0] = -std::sin(p[2]);
      hv[1] = std::cos(p[2]);
      e = v[1] * std::cos(p[2]) - v[0] * std::sin(p[2]);
    }
    err[m] = e;
    for (int k = 0; k < D; k++) {
      if (Hp) Hp[(size_t)m * D + k] = k < 3 ? hp[k] : 0.0;
      if (Hv) Hv[(size_t)m * D + k] = k < 3 ? hv[k] : 0.0;
    }
  }
  return 0;
}

int orc_obstacle_factor(const void* r, const void* s, double eps, int M, const double* conf,
                        double* err, double* H1) {
  const Robot& R = *(const Robot*)r;
  const int S = R.nr_spheres(), D = R.dof;
  for (int m = 0; m < M; m++)
    obstacle_factor(R, *(const Sdf*)s, eps, conf + (size_t)m * D, err + (size_t)m * S,
                    H1 ? H1 + (size_t)m * S * D : nullptr);
  return 0;
}

int orc_obstacle_gp_factor(const void* r, const void* s, double eps, const double* Qc, double dt,
                           double tau, int M, const double* c1, const double* v1,
                           const double* c2, const double* v2, double* err, double* H1,
                           double* H2, double* H3, double* H4) {
  const Robot& R = *(const Robot*)r;
  const int S = R.nr_spheres(), D = R.dof;
  Mat Q = Mat::identity(D);
  if (Qc)
    for (int i = 0; i < D * D; i++) Q.a[i] = Qc[i];
  GPInterp gp(D, R.is_lie(), Q, dt, tau);
  for (int m = 0; m < M; m++) {
    const size_t o = (size_t)m * D, oh = (size_t)m * S * D;
    obstacle_gp_factor(R, *(const Sdf*)s, eps, gp, c1 + o, v1 + o, c2 + o, v2 + o,
                       err + (size_t)m * S, H1 ? H1 + oh : nullptr, H2 ? H2 + oh : nullptr,
                       H3 ? H3 + oh : nullptr, H4 ? H4 + oh : nullptr);
  }
  return 0;
}

int orc_gp_prior_factor(int D, int lie, double dt, int M, const double* c1, const double* v1,
                        const double* c2, const double* v2, double* err, double* H1, double* H2,
                        double* H3, double* H4) {
  for (int m = 0; m < M; m++) {
    const size_t o = (size_t)m * D;
    Mat A, B, C, E;
    const bool jac = H1 || H2 || H3 || H4;
    gp_prior_factor(D, lie != 0, dt, c1 + o, v1 + o, c2 + o, v2 + o, err + (size_t)m * 2 * D,
                    jac ? &A : nullptr, jac ? &B : nullptr, jac ? &C : nullptr, jac ? &E : nullptr);
    const size_t oh = (size_t)m * 2 * D * D;
    if (H1) std::memcpy(H1 + oh, A.a.data(), sizeof(double) * 2 * D * D);
    if (H2) std::memcpy(H2 + oh, B.a.data(), sizeof(double) * 2 * D * D);
    if (H3) std::memcpy(H3 + oh, C.a.data(), sizeof(double) * 2 * D * D);
    if (H4) std::memcpy(H4 + oh, E.a.data(), sizeof(double) * 2 * D * D);
  }
  return 0;
}

int orc_gp_interpolate(int D, int lie, const double* Qc, double dt, double tau, int M,
                       const double* c1, const double* v1, const double* c2, const double* v2,
                       double* conf, double* vel) {
  Mat Q = Mat::identity(D);
  if (Qc)
    for (int i = 0; i < D * D; i++) Q.a[i] = Qc[i];
  GPInterp gp(D, lie != 0, Q, dt, tau);
  for (int m = 0; m < M; m++) {
    const size_t o = (size_t)m * D;
    if (conf) gp.interpolate_pose(c1 + o, v1 + o, c2 + o, v2 + o, conf + o, nullptr, nullptr, nullptr, nullptr);
    if (vel) gp.interpolate_velocity(c1 + o, v1 + o, c2 + o, v2 + o, vel + o);
  }
  return 0;
}

// interpolateArmTraj / interpolatePose2MobileArmTraj  gpmp2/planner/TrajUtils.cpp:162-236
// traj [B][N+1][2D] -> out [B][(end-start)*(inter+1)+1][2D]
int orc_interpolate_traj(int D, int lie, const double* Qc, double dt, int inter, int B, int N, int start,
                         int end, const double* traj, double* out) {
  Mat Q = Mat::identity(D);
  if (Qc)
    for (int i = 0; i < D * D; i++) Q.a[i] = Qc[i];
  const double inter_dt = dt / static_cast<double>(inter + 1);
  std::vector<GPInterp> gp;
  for (int j = 1; j <= inter; j++) gp.emplace_back(D, lie != 0, Q, dt, static_cast<double>(j) * inter_dt);
  const size_t Mo = (size_t)(end - start) * (inter + 1) + 1;
  for (int b = 0; b < B; b++) {
    const double* t = traj + (size_t)b * (N + 1) * 2 * D;
    double* o = out + (size_t)b * Mo * 2 * D;
    size_t ri = 0;
    for (int i = start; i < end; i++) {
      std::memcpy(o + ri * 2 * D, t + (size_t)i * 2 * D, sizeof(double) * 2 * D);
      const double *c1 = t + (size_t)i * 2 * D, *v1 = c1 + D, *c2 = c1 + 2 * D, *v2 = c2 + D;
      for (int j = 1; j <= inter; j++) {
        ri++;
        gp[j - 1].interpolate_pose(c1, v1, c2, v2, o + ri * 2 * D, nullptr, nullptr, nullptr, nullptr);
        gp[j - 1].interpolate_velocity(c1, v1, c2, v2, o + ri * 2 * D + D);
      }
      ri++;
    }
    std::memcpy(o + ri * 2 * D, t + (size_t)end * 2 * D, sizeof(double) * 2 * D);
  }
  return 0;
}

// Jacobians of interpolatePose (H1..H4 [M][D][D]) -- used to pin the Lie interpolator
int orc_gp_interpolate_jac(int D, int lie, const double* Qc, double dt, double tau, int M,
                           const double* c1, const double* v1, const double* c2,
                           const double* v2, double* H1, double* H2, double* H3, double* H4) {
  Mat Q = Mat::identity(D);
  if (Qc)
    for (int i = 0; i < D * D; i++) Q.a[i] = Qc[i];
  GPInterp gp(D, lie != 0, Q, dt, tau);
  std::vector<double> conf(D);
  for (int m = 0; m < M; m++) {
    const size_t o = (size_t)m * D, oh = (size_t)m * D * D;
    Mat A, B, C, E;
    gp.interpolate_pose(c1 + o, v1 + o, c2 + o, v2 + o, conf.data(), &A, &B, &C, &E);
    std::memcpy(H1 + oh, A.a.data(), sizeof(double) * D * D);
    std::memcpy(H2 + oh, B.a.data(), sizeof(double) * D * D);
    std::memcpy(H3 + oh, C.a.data(), sizeof(double) * D * D);
    std::memcpy(H4 + oh, E.a.data(), sizeof(double) * D * D);
  }
  return 0;
}

// Lambda, Psi [2D][2D] of gp/GPutils.h:49-59 for a general Qc
int orc_gp_matrices(int D, const double* Qc, double dt, double tau, double* Lambda, double* Psi) {
  Mat Q = Mat::identity(D);
  if (Qc)
    for (int i = 0; i < D * D; i++) Q.a[i] = Qc[i];
  const Mat L = calcLambda(Q, dt, tau), P = calcPsi(Q, dt, tau);
  std::memcpy(Lambda, L.a.data(), sizeof(double) * 4 * D * D);
  std::memcpy(Psi, P.a.data(), sizeof(double) * 4 * D * D);
  return 0;
}

int orc_joint_limit_factor(int D, const double* down, const double* up, const double* th, int M,
                           const double* x, double* err, double* Hd) {
  for (int m = 0; m < M; m++)
    for (int k = 0; k < D; k++) {
      double H;
      err[(size_t)m * D + k] = hinge_limit(x[(size_t)m * D + k], down[k], up[k], th[k], &H);
      if (Hd) Hd[(size_t)m * D + k] = H;
    }
  return 0;
}

int orc_graph_error(const void* r, const void* s, const gpmp2mi_settings* set,
                    const gpmp2mi_graph_opts* o, int B, const double* sc, const double* sv,
                    const double* ec, const double* ev, const double* traj, double* err) {
  const int d = set->dof;
  const size_t m = (size_t)(set->total_step + 1) * 2 * d;
  for (int b = 0; b < B; b++) {
    Problem P = make_problem((const Robot*)r, (const Sdf*)s, set, o, sc + (size_t)b * d,
                             sv + (size_t)b * d, ec + (size_t)b * d, ev + (size_t)b * d);
    err[b] = P.error(traj + b * m);
  }
  return 0;
}

int orc_linearize(const void* r, const void* s, const gpmp2mi_settings* set,
                  const gpmp2mi_graph_opts* o, int B, const double* sc, const double* sv,
                  const double* ec, const double* ev, const double* traj, double* Hdiag,
                  double* Hoff, double* g, double* err) {
  const int d = set->dof, n = 2 * d, nb = set->total_step + 1;
  const size_t m = (size_t)nb * n;
  for (int b = 0; b < B; b++) {
    Problem P = make_problem((const Robot*)r, (const Sdf*)s, set, o, sc + (size_t)b * d,
                             sv + (size_t)b * d, ec + (size_t)b * d, ev + (size_t)b * d);
    std::vector<LinFactor> F;
    const double e = P.linearize(traj + b * m, &F);
    NormalEq ne;
    ne.assemble(F, nb, n);
    if (Hdiag) std::memcpy(Hdiag + (size_t)b * nb * n * n, ne.D.data(), sizeof(double) * nb * n * n);
    if (Hoff) std::memcpy(Hoff + (size_t)b * (nb - 1) * n * n, ne.O.data(), sizeof(double) * (nb - 1) * n * n);
    if (g) std::memcpy(g + (size_t)b * m, ne.g.data(), sizeof(double) * m);
    if (err) err[b] = e;
  }
  return 0;
}

// dense whitened Jacobian A [rows][(N+1) 2D] and residual r [rows] of ONE trajectory, for an
// independent numpy solve.  Call with A == NULL to get the row count.
int orc_dense_linearize(const void* r, const void* s, const gpmp2mi_settings* set,
                        const gpmp2mi_graph_opts* o, const double* sc, const double* sv,
                        const double* ec, const double* ev, const double* traj, double* A,
                        double* res, int* rows) {
  Problem P = make_problem((const Robot*)r, (const Sdf*)s, set, o, sc, sv, ec, ev);
  std::vector<LinFactor> F;
  P.linearize(traj, &F);
  int R = 0;
  for (auto& f : F) R += f.m;
  *rows = R;
  if (!A) return 0;
  const int n = P.n(), W = P.nstates() * n;
  std::memset(A, 0, sizeof(double) * (size_t)R * W);
  int row = 0;
  for (auto& f : F)
    for (int i = 0; i < f.m; i++, row++) {
      res[row] = f.r[i];
      for (int c = 0; c < f.ns * n; c++) A[(size_t)row * W + f.s0 * n + c] = f.A[(size_t)i * f.ns * n + c];
    }
  return 0;
}

int orc_block_tridiag_solve(int B, int nblk, int n, const double* Hd, const double* Ho,
                            const double* b, double* x, int* ok) {
  for (int t = 0; t < B; t++) {
    NormalEq ne;
    ne.nblk = nblk;
    ne.n = n;
    ne.D.assign(Hd + (size_t)t * nblk * n * n, Hd + (size_t)(t + 1) * nblk * n * n);
    ne.O.assign(Ho + (size_t)t * (nblk - 1) * n * n, Ho + (size_t)(t + 1) * (nblk - 1) * n * n);
    ne.g.resize((size_t)nblk * n);
    for (int i = 0; i < nblk * n; i++) ne.g[i] = -b[(size_t)t * nblk * n + i];  // solve() uses -g
    const bool good = ne.solve(0.0, x + (size_t)t * nblk * n);
    if (ok) ok[t] = good ? 1 : 0;
  }
  return 0;
}

int orc_batch_optimize(const void* r, const void* s, const gpmp2mi_settings* set,
                       const gpmp2mi_graph_opts* o, int B, const double* sc, const double* sv,
                       const double* ec, const double* ev, const double* init, double* out,
                       int* iters, double* final_err, int* status, double* trace, int nthreads) {
  const int d = set->dof;
  const size_t m = (size_t)(set->total_step + 1) * 2 * d;
  const int tl = set->max_iter + 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
  for (int b = 0; b < B; b++) {
    Problem P = make_problem((const Robot*)r, (const Sdf*)s, set, o, sc + (size_t)b * d,
                             sv + (size_t)b * d, ec + (size_t)b * d, ev + (size_t)b * d);
    OptResult res = optimize(P, init + b * m, out + b * m);
    if (iters) iters[b] = res.iterations;
    if (final_err) final_err[b] = res.final_error;
    if (status) status[b] = res.status;
    if (trace)
      for (int k = 0; k < tl; k++)
        trace[(size_t)b * tl + k] = k < (int)res.trace.size() ? res.trace[k]
                                                              : std::numeric_limits<double>::quiet_NaN();
  }
  return 0;
}

// batch optimize with the replanner's extra state priors and goal switch (mirrors
// gpmp2mi_plan_fix_state / add_state_estimate / remove_goal + gpmp2mi_plan_update)
int orc_batch_optimize_xp(const void* r, const void* s, const gpmp2mi_settings* set, const gpmp2mi_graph_opts* o,
                          int B, const double* sc, const double* sv, const double* ec, const double* ev,
                          const double* init, const int* xp_n, const int* xp_state, const int* xp_has_vel,
                          const double* xp_target, const double* xp_info, const int* goal_on, double* out,
                          int* iters, double* final_err, int* status) {
  const int d = set->dof, XP = GPMP2MI_MAX_STATE_PRIORS;
  const size_t m = (size_t)(set->total_step + 1) * 2 * d;
  for (int b = 0; b < B; b++) {
    Problem P = make_problem((const Robot*)r, (const Sdf*)s, set, o, sc + (size_t)b * d, sv + (size_t)b * d,
                             ec + (size_t)b * d, ev + (size_t)b * d);
    P.set.goal_on = goal_on ? goal_on[b] != 0 : true;
    for (int e = 0; xp_n && e < xp_n[b]; e++) {
      const size_t xe = (size_t)b * XP + e;
      Settings::StatePrior sp;
      sp.state = xp_state[xe];
      sp.has_vel = xp_has_vel[xe] != 0;
      sp.conf.assign(xp_target + xe * 2 * d, xp_target + xe * 2 * d + d);
      sp.vel.assign(xp_target + xe * 2 * d + d, xp_target + xe * 2 * d + 2 * d);
      sp.Wc = Mat(d, d);
      sp.Wv = Mat(d, d);
      for (int k = 0; k < d * d; k++) {
        sp.Wc.a[k] = xp_info[xe * 2 * d * d + k];
        sp.Wv.a[k] = xp_info[xe * 2 * d * d + d * d + k];
      }
      P.set.state_priors.push_back(sp);
    }
    OptResult res = optimize(P, init + b * m, out + b * m);
    if (iters) iters[b] = res.iterations;
    if (final_err) final_err[b] = res.final_error;
    if (status) status[b] = res.status;
  }
  return 0;
}

int orc_set_dogleg_probe(double* buf, int cap_rows) {
  set_dogleg_probe(buf, cap_rows);
  return 0;
}
int orc_dogleg_probe_rows() { return dogleg_probe_rows(); }

int orc_collision_cost(const void* r, const void* s, int total_step, int B, const double* traj,
                       double* cost) {
  // internal::CollisionCost  planner/BatchTrajOptimizer-inl.h:87-100 (epsilon = 0)
  const Robot& R = *(const Robot*)r;
  const int d = R.dof, S = R.nr_spheres();
  std::vector<double> e(S);
  for (int b = 0; b < B; b++) {
    double c = 0;
    for (int i = 0; i <= total_step; i++) {
      obstacle_factor(R, *(const Sdf*)s, 0.0, traj + ((size_t)b * (total_step + 1) + i) * 2 * d, e.data(), nullptr);
      for (double x : e) c += x;
    }
    cost[b] = c;
  }
  return 0;
}

// Pose2 helpers exposed for pinning against the reference's Lie tests
int orc_pose2_expmap(const double* v, double* p) {
  const Pose2 q = pose2_expmap(v);
  p[0] = q.x; p[1] = q.y; p[2] = q.th;
  return 0;
}
int orc_pose2_logmap(const double* p, double* v) {
  pose2_logmap(Pose2{p[0], p[1], p[2]}, v);
  return 0;
}
int orc_retract(const void* r, int nstates, const double* traj, const double* delta, double* out) {
  Problem P;
  P.robot = (const Robot*)r;
  P.set.dof = P.robot->dof;
  P.set.total_step = nstates - 1;
  P.retract(traj, delta, out);
  return 0;
}

}  // extern "C"
