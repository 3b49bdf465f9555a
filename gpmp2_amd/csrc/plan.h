// plan.h -- device-resident state of a batch of trajectory problems (gpmp2mi_plan) and the
// launchers of the fused hot-path kernels in plan_kernels.hip.
#pragma once
#include "common.h"

namespace g2 {

constexpr int MAXI = 16;   // max obs_check_inter
constexpr int TILE = 16;
constexpr int XP_MAX = GPMP2MI_MAX_STATE_PRIORS;  // extra per-state priors per trajectory (replanning)   // block-tridiagonal tile edge (n = 2*dof <= 16 in the MFMA solver)

// Uniform parameters of a plan; lives in HBM, read through scalar loads.
struct PlanParams {
  int B, N, I, P, Ppad, D, n, NG, REC, Npad, GPREC;
  int RECS, GPS;                       // record strides in HBM and LDS: REC / GPREC rounded up to even (16-B pieces)
  int max_pass;
  int obs_skip_first, flag_pos_limit, flag_vel_limit, opt_type, max_iter, no_increase, fixed_iters,
      lie;
  int lin_split;                       // 2: k_linearize splits the spheres of a point over 2 wavefronts (fixed-base arms)
  int end_conf_prior_off;              // 1: no PriorFactor on x_N (a goal / workspace factor stands in)
  int wide;                            // 2 dof > 15: blocks wider than one tile (2x2-tile kernels of wide_cr.h, or the dense path of dense_kernels.hip)
  int split_back;                      // GN: back-substitution levels 1, 2 and the retract run in k_finish_step
  int spart_groups;                    // shares per trajectory in pb.spart: workgroups of k_finish_trial(_wide), or the chunks of
                                       // k_linearize_arm when it applies the trial step itself (fuse_finish)
  int wide_h0;                         // wide blocks: first forward level k_solve_step_wide runs itself (levels below: k_cr_level_wide)
  int fuse_finish;                     // GN fast path: levels 4, 2, 1 of the back-substitution and the retract run at the head of the
                                       // NEXT pass's k_linearize_arm (no k_finish_step); the state buffers cur / last swap roles every pass
  double eps, obs_w, delta_t;          // obs_w = 1 / cost_sigma^2
  double conf_prior_w, vel_prior_w;    // 1 / sigma^2
  double vdyn_w;                       // 1 / dynamics_sigma^2 or 0
  double rel_thresh, abs_tol, err_tol;
  double lm_lambda0, lm_factor, lm_upper, lm_lower, lm_min_fidelity, dl_delta0;
  alignas(16) GpCoef coef[MAXI];
  // per sub-step, for the assembler (staged into LDS): [0..3] Psi_r Psi_c, [4..7] Lam_r Lam_c, [8..11] Lam_r Psi_c,
  // [12..15] Psi_r Lam_c at index ar * 2 + ac (a = 0 conf / 1 vel; Lam = (l11, l12), Psi = (p11, p12)); [16..19]
  // l11 l12 p11 p12; rest padding
  alignas(16) double coefq[MAXI][24];
  double Winv[4];                      // B(delta_t): 12/dt^3, -6/dt^2, -6/dt^2, 4/dt
  double Qc_inv[MAXD * MAXD];
  double pos_lo[MAXD], pos_hi[MAXD], pos_th[MAXD], pos_w[MAXD];
  double vel_lim[MAXD], vel_th[MAXD], vel_w[MAXD];
  // GP prior constant Hessian blocks (n x n row-major, ld = n): KA = Phi^T W Phi, KB = W,
  // KO = -Phi^T W (block (i, i+1))
  double KA[4 * MAXD * MAXD], KB[4 * MAXD * MAXD], KO[4 * MAXD * MAXD];
};

// Extra factors a plan carries as data (gpmp2mi_graph_opts): workspace priors / goal factor and self collision on
// ranges of support states.  They are unary in x_i, so their J^T J / sigma^2, J^T r / sigma^2, r^T r / sigma^2 are
// added to the record of the state's unary evaluation point after k_linearize (launch_extra_factors); the solver
// kernels never see them separately.
struct PlanExtras {
  int n_ws, n_sc, sc_first, sc_last;
  int ws_mode[GPMP2MI_MAX_WORKSPACE_FACTORS], ws_link[GPMP2MI_MAX_WORKSPACE_FACTORS];
  int ws_first[GPMP2MI_MAX_WORKSPACE_FACTORS], ws_last[GPMP2MI_MAX_WORKSPACE_FACTORS];
  double ws_w[GPMP2MI_MAX_WORKSPACE_FACTORS];                       // 1 / sigma^2
  double sc_w[GPMP2MI_MAX_SELF_COLLISION_PAIRS];                    // 1 / sigma^2
  // device workspace, M = B (N + 1) states
  double* des;       // [n_ws][16]
  double* sc_data;   // [n_sc][4]
  double* radius;    // [S] caller's sphere order
  double* poses;     // [M][L][16]
  double* Jp;        // [M][L][6][D]
  double* ws_err;    // [n_ws][M][6]
  double* ws_H;      // [n_ws][M][6][D]
  double* cen;       // [M][S][3]
  double* Jc;        // [M][S][3][D]
  double* sc_err;    // [M][n_sc]
  double* sc_H;      // [M][n_sc][D]
};

// Per-plan device buffers.
struct PlanBuffers {
  PlanParams* params;      // device copy
  double* start_conf;      // [B][D]
  double* start_vel;
  double* end_conf;
  double* end_vel;
  double* cur;             // [B][N+1][2D]  opt->values()
  double* last;            // [B][N+1][2D]  last_values
  double* trial;           // [B][N+1][2D]  LM / Dogleg trial point
  double* init;            // [B][N+1][2D]  pristine initial values (optimize() can be re-run)
  double* result;          // [B][N+1][2D]
  double* rec;             // [B][Ppad][RECS] point-major records: G (packed upper), g, e [, M1..M4]   (buffer 0)
  double* rec2;            // second buffer for trial linearizations (LM / Dogleg)
  double* gpu;             // [B][Npad][GPS] per interval end state: GP prior u = W r (n), r^T W r [, J1, J3]
  double* gpu2;
  double* tiles;           // [B][N+1][256] diagonal tile S_i = [D_i | -g_i] of every even block (updated in place)
  double* fac;             // [B][N+1][3][256] factor tiles Wl, Wr (both carry y), V = R^-T
  double* pend;            // [B][ceil((N+1)/4)][256] what block 4q+4 still owes to blocks 4q+2, 4q+3 (k_assemble -> cr_forward)
  double* coup;            // [B][ceil((N+1)/4)][256] level-4 coupling between blocks 4q and 4q+4 (k_assemble -> cr_forward)
  double* delta;           // [B][N+1][2D]
  double* gvec;            // [B][N+1][16] gradient g_i = J^T Sigma^-1 r of the current linearization
  double* htiles;          // [B][N+1][2][256] un-eliminated D_i and H_{i,i+1} (Dogleg: g^T H g)
  double* hgpart;          // [B][Npad] per-block share of g^T H g
  double* scal;            // [B][16] per-trajectory scalars of the current trial step (see SC_*)
  // extra priors (gpmp2mi_plan_fix_state / add_state_estimate) and goal switch (remove_goal)
  int* xp_n;               // [B] number of extra priors
  int* xp_state;           // [B][XP_MAX] state index
  int* xp_has_vel;         // [B][XP_MAX] 1 if the entry also constrains the velocity
  double* xp_target;       // [B][XP_MAX][2D] conf, vel
  double* xp_info;         // [B][XP_MAX][2][D*D] information matrices (conf, vel)
  int* goal_on;            // [B] 0 after removeGoalConfigAndVel
  // dense block-tridiagonal system and its factors (wide path only; n = 2 dof)
  double* wHd;             // [B][N+1][n][n]   diagonal blocks, then their upper Cholesky factors R_i
  double* wHo;             // [B][N][n][n]     block (i+1, i) (read only)
  double* wg;              // [B][N+1][n]      gradient (read only)
  double* wWl;             // [B][N+1][n][n]   dense cyclic reduction: W_l = R^-T C_l of every eliminated block
  double* wWr;             // [B][N+1][n][n]   ... W_r
  double* wy;              // [B][N+1][n]      ... y = R^-T b
  double* wrb;             // [B][N+1][n]      ... running right-hand side of the blocks still in the tree
  double* wx;              // [B][N+1][n]      ... solution
  double* xg;              // [B][N+1][16 or 32] step of the blocks the solve kernel back-substitutes itself (split path)
  int* stepped;            // [B] pass + 1 of the last pass in which the trajectory took a step (split path); trial-step
                           // path: 1 when k_solve_step left a factorisation for k_finish_trial
  double* spart;           // [B][ceil((N+1)/4)][3] per-group shares of g.delta, |delta|^2, |g|^2 (k_finish_trial)
  int* which;              // [B] record buffer (0: rec/gpu, 1: rec2/gpu2) holding the linearization at `cur`
  // per-trajectory scalars
  double* cur_err;         // error at `cur`
  double* prev_err;        // currentError of gpmp2::optimize
  double* last_err;
  double* final_err;
  double* lambda;          // LM lambda / Dogleg delta
  double* trace;           // [B][max_iter+1]
  int* iters;
  int* status;
  int* active;             // 1 while the trajectory is still iterating
  int* phase;              // optimizer-specific sub-state
  int* notspd;             // [B] set when a level-1 pivot (k_assemble) was not positive
  double* epart;           // [B][Npad] per-block share of the graph error (k_assemble)
  int* n_active;           // [max_pass] trajectories that iterated in each pass
  int* done;               // [max_pass] workgroups of the pass-closing kernel that have finished
  int* host_flags;         // [max_pass] pinned, device-mapped: n_active[pass] once the pass is complete, else -1
  unsigned long long* stamps;  // [B][64] s_memtime stamps (diagnostic builds only)
};

// scalars of a trial step (PlanBuffers::scal)
enum { SC_GD = 0, SC_DD, SC_GG, SC_GHG, SC_GN, SC_NN, SC_Q, SC_XNORM, SC_ZERO_STEP, SC_COUNT = 16 };

// record buffers of trajectory b: sel = 0 -> the linearization at `cur`, 1 -> the other one
__host__ __device__ inline double* rec_of(const PlanBuffers& pb, int which_b, int sel) {
  return (which_b ^ sel) ? pb.rec2 : pb.rec;
}
__host__ __device__ inline double* gpu_of(const PlanBuffers& pb, int which_b, int sel) {
  return (which_b ^ sel) ? pb.gpu2 : pb.gpu;
}

int launch_linearize(const RobotDev& hrobot, const RobotDev* robot, const SdfDev& sdf,
                     const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                     const int* active, hipStream_t st, double* dst = nullptr, int pass = 0, bool trial = false);
int launch_extra_accumulate(const PlanParams& hp, const PlanBuffers& pb, const PlanExtras& ex, int L, int S, int bufsel,
                            const int* active, hipStream_t st);
int launch_set_mode(const PlanBuffers& pb, int opt_type, int fixed_iters, hipStream_t st);
int launch_error_parts(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel, const int* active,
                       hipStream_t st);
int launch_plan_reset(const PlanParams& hp, const PlanBuffers& pb, const double* start, hipStream_t st);
int launch_assemble(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                    const int* active, hipStream_t st);
int launch_solve_step(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_ghg(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_decide(const PlanParams& hp, const PlanBuffers& pb, int pass, bool init, hipStream_t st);
int launch_finalize_unfinished(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_debug_crosslane(const double* in, double* out, hipStream_t st);
int launch_gn_step_cr(const PlanParams& hp, const PlanBuffers& pb, int pass, hipStream_t st);
int launch_finish_step(const PlanParams& hp, const PlanBuffers& pb, int pass, hipStream_t st);
int launch_finish_trial(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_solve_dense(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
// wide_cr.h (8 <= dof <= 11 on 2x2 tiles)
int launch_assemble_wide(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                         const int* active, hipStream_t st);
int launch_ghg_wide(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_solve_step_wide(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_finish_trial_wide(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st);
int launch_cr_level_wide(const PlanParams& hp, const PlanBuffers& pb, int h, hipStream_t st);
int launch_error_reduce(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                        double* err, hipStream_t st);
int launch_export_normal_eq(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                            double* Hdiag, double* Hoff, double* g, hipStream_t st, const int* active = nullptr);
int launch_block_tridiag_solve(int B, int nblk, int n, const double* Hd, const double* Ho,
                               const double* b, double* x, int* ok, double* scratch, hipStream_t st);

}  // namespace g2
