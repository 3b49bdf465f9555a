// assembler.h -- builds the block-tridiagonal tiles of one support state from the per-point records
// of k_linearize.  Used by k_assemble (solver input) and k_export_normal_eq (parity tests).
//
// Vector-space robots: Kronecker structure -- a point with interpolation
// scalars c contributes (c c^T) (x) G to the 2x2-block window and c (x) g to the gradient, and the
// GP prior Hessian blocks are constants.  Pose2 robots (Lie path, config 5): the GP prior Jacobians
// depend on the states (gp/GaussianProcessPriorLie.h:61-86); k_linearize stores the two 3x3 pose
// blocks J1 = Hlog*Hcomp1*Hinv, J3 = Hlog per interval and the blocks are formed here.
#pragma once
#include "device_math.h"
#include "plan.h"
#include "tiles.h"

namespace g2 {

// diagnostic build only: phase stamps of build_tiles for block i == 1 (slots 24.. of the trajectory's stamp row)
#ifdef G2_STAMPS
#define G2_BSTAMP(k) do { if (i == 1 && lane == 0 && row0 == 0 && col0 == 0 && pb.iters[b] == G2_STAMP_ITER) pb.stamps[(size_t)b * 64 + 24 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G2_BSTAMP(k) do {} while (0)
#endif

constexpr int GP_EXTRA_LIE = 18;  // J1 (9) + J3 (9) appended to the GP record of Pose2 robots

template <int D, bool LIE>
struct Assembler {
  static constexpr int n = 2 * D, NG = D * (D + 1) / 2, RECP = NG + D + 1;  // per-point record
  static constexpr int GPN = n + 1 + GP_EXTRA_LIE;
  static constexpr int CQ = 24;            // doubles per sub-step in the staged weight table (PlanParams::coefq)
  static constexpr int PT_EXTRA_LIE = 36;  // M1..M4 (3x3 pose blocks of the interpolation Jacobians)
  static constexpr int RECMAX = RECP + PT_EXTRA_LIE;

  // LDS image of one interval (dynamic shared memory, sized at launch from I and the record
  // lengths): pts[jj][RECS] for jj = 0..I (I = unary of the end state), then the GP record
  // u = Q^-1 r (n), r^T u, [J1 (9), J3 (9)] padded to GPS, then the weight table of the I sub-steps (CQ doubles
  // each, PlanParams::coefq: the four 2x2 products of the interpolation scalars + the scalars themselves)
  struct Slot {
    double* base;
    int rec;   // P.RECS (record stride, even)
    int npts;  // I + 1
    int gps;   // P.GPS
    __device__ __forceinline__ double* pt(int jj) const { return base + jj * rec; }
    __device__ __forceinline__ const double* gpr() const { return base + npts * rec; }
    __device__ __forceinline__ double* gpw() const { return base + npts * rec; }
    __device__ __forceinline__ const double* coef(int jj) const { return base + npts * rec + gps + CQ * jj; }
  };
  __host__ __device__ static int slot_doubles(int I, int RECS, int GPS) { return (I + 1) * RECS + GPS + CQ * I; }
  __device__ __forceinline__ Slot make_slot(double* smem, int which) const {
    return Slot{smem + which * slot_doubles(P.I, P.RECS, P.GPS), P.RECS, P.I + 1, P.GPS};
  }

  const PlanParams& P;
  const PlanBuffers& pb;
  const double* rec;
  const double* gpu;
  int b, lane, c, g;   // c / g: this lane's global column / first global row (tile offsets included)
  int rhscol;          // column that carries the right-hand side (-g_i)
  int row0, col0;      // offset of this 16x16 tile inside a wider block
  // per-lane static decode of its 4 rows
  int tri[4];
  bool valid[4];   // rho < n && c < n
  int a_row[4], k_row[4], a_col, k_col;

  // row0 / col0: offset of the 16x16 tile inside a wider block (blocks with 2 dof > 15 are exported as
  // 2x2 tiles by k_export_normal_eq); rhscol_: global column of the right-hand side
  __device__ Assembler(const PlanParams& P_, const PlanBuffers& pb_, const double* rec_, const double* gpu_,
                       int b_, int lane_, int row0_ = 0, int col0_ = 0, int rhscol_ = RHSCOL)
      : P(P_), pb(pb_), rec(rec_), gpu(gpu_), b(b_), lane(lane_), c(col0_ + (lane_ & 15)), g(row0_ + (lane_ >> 4)),
        rhscol(rhscol_), row0(row0_), col0(col0_) {
    a_col = c >= D;
    k_col = c - a_col * D;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int rho = g + 4 * k;
      valid[k] = rho < n && c < n;
      a_row[k] = rho >= D;
      k_row[k] = rho - a_row[k] * D;
      const int lo = min(k_row[k], k_col), hi = max(k_row[k], k_col);
      tri[k] = valid[k] ? lo * D - (lo * (lo - 1)) / 2 + (hi - lo) : 0;
    }
  }

  // global -> registers -> LDS for the two intervals a block needs (iv and iv + 1; interval 0 is
  // only the unary point of state 0, intervals beyond N read as zeros).  The records are point-major
  // (rec[b][p][RECS], gpu[b][i][GPS]), so an interval is ONE contiguous run of (I + 1) * RECS doubles: every
  // lane moves 16-B pieces, fully coalesced.  All loads of both intervals are issued before the first
  // LDS store so the wavefront pays one memory latency; NLD2 bounds the per-lane load count (checked on
  // the host).  The weight table of the sub-steps rides along into the slot (LDS reads later instead of scalar
  // loads and per-entry products inside the accumulation loops).
  static constexpr int NLD2 = (D <= 7) ? 6 : (D <= 11) ? 9 : 14;
  // count = 1 stages interval iv only (kernels whose wavefronts share their slots)
  __device__ __forceinline__ void stage2(int iv, const Slot& s0, const Slot& s1, int count = 2) const {
    const int I = P.I, RECS = P.RECS, GPS = P.GPS;
    double2 val[2][NLD2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
      if (w >= count) break;
      const int ivw = iv + w;
      const int npt = (ivw == 0) ? 1 : I + 1;
      const int nd2 = (npt * RECS) >> 1, ng2 = GPS >> 1, nc2 = (CQ / 2) * I;
      const int p0 = (ivw == 0) ? 0 : 1 + (ivw - 1) * (I + 1);
      const bool in_range = ivw <= P.N;
      const double2* src = reinterpret_cast<const double2*>(rec + ((size_t)b * P.Ppad + p0) * RECS);
      const double2* gsrc = reinterpret_cast<const double2*>(gpu + ((size_t)b * P.Npad + ivw) * GPS);
      const double2* csrc = reinterpret_cast<const double2*>(P.coefq);
#pragma unroll
      for (int m = 0; m < NLD2; m++) {
        const int v = lane + 64 * m;
        double2 x = {0.0, 0.0};
        if (v < nd2) {
          if (in_range) x = src[v];
        } else if (v < nd2 + ng2) {
          if (in_range && ivw > 0) x = gsrc[v - nd2];
        } else if (v < nd2 + ng2 + nc2) {
          x = csrc[v - nd2 - ng2];
        }
        val[w][m] = x;
      }
    }
#pragma unroll
    for (int w = 0; w < 2; w++) {
      if (w >= count) break;
      const Slot& s = w ? s1 : s0;
      const int ivw = iv + w;
      const int npt = (ivw == 0) ? 1 : I + 1;
      const int nd2 = (npt * RECS) >> 1, ng2 = GPS >> 1, nc2 = (CQ / 2) * I;
      double2* dpt = reinterpret_cast<double2*>(s.pt((ivw == 0) ? I : 0));
      double2* dgp = reinterpret_cast<double2*>(s.gpw());
#pragma unroll
      for (int m = 0; m < NLD2; m++) {
        const int v = lane + 64 * m;
        if (v < nd2) dpt[v] = val[w][m];
        else if (v < nd2 + ng2 + nc2) dgp[v - nd2] = val[w][m];
      }
    }
  }

  // (Hint^T g)[kr]
  __device__ __forceinline__ static double hint_vec(const double* pt, const double* M, double s, int kr) {
    const double* gp_ = pt + NG;
    if (kr >= 3) return s * gp_[kr];
    return M[0 * 3 + kr] * gp_[0] + M[1 * 3 + kr] * gp_[1] + M[2 * 3 + kr] * gp_[2];
  }

  // ---- Pose2 robots: interpolated obstacle factors through the matrix cores.
  // With Hint_k = diag(M_k (3x3), s_k I) the contribution of one interpolated point to the block rows / columns of a
  // support state is the congruence E_L^T G E_R, E = [Hint_x | Hint_v] (D x n) of the state's role in the interval
  // (role 0: first state -> M1, M2, Lambda scalars; role 1: second state -> M3, M4, Psi scalars).  E and G are
  // formed in the tile layout straight from the point's LDS record (rows = configuration coordinate) and the two
  // products run as v_mfma_f64_16x16x4 chains that accumulate over the sub-steps -- no per-entry case analysis.
  static constexpr int KT = (D + 15) / 16;  // tiles along the contracted (configuration) dimension
  // E tile: rows a = 16 rt + (lane >> 4) + 4 k, columns j = cbase + (lane & 15) of the state's [x; v] block
  __device__ __forceinline__ Tile hint_tile(const double* M, const double* cf, int role, int rt, int cbase) const {
    const int j = cbase + (lane & 15), av = j >= D, kk = j - av * D;
    const double* Mk = M + (role * 2 + av) * 9;
    const double sk = cf[role * 2 + av];
    Tile T;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a = 16 * rt + (lane >> 4) + 4 * k;
      const bool pose = a < 3 && kk < 3;
      const double m = Mk[pose ? a * 3 + kk : 0];
      const double v = pose ? m : ((a == kk && a >= 3) ? sk : 0.0);
      T.r[k] = (a < D && j < n) ? v : 0.0;
    }
    return T;
  }
  // G tile (rt, ct) of the point's packed J^T J / sigma^2
  __device__ __forceinline__ Tile g_tile(const double* pt, int rt, int ct) const {
    const int a2 = 16 * ct + (lane & 15);
    Tile T;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a = 16 * rt + (lane >> 4) + 4 * k;
      const bool in = a < D && a2 < D;
      const int lo = in ? min(a, a2) : 0, hi = in ? max(a, a2) : 0;
      const double v = pt[lo * D - (lo * (lo - 1)) / 2 + (hi - lo)];
      T.r[k] = in ? v : 0.0;
    }
    return T;
  }
  // acc += A^T B over the rows of tile row `rt` that hold configuration coordinates
  __device__ __forceinline__ static void mfma_atb_acc(const Tile& A, const Tile& B, int rt, v4d& acc) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (16 * rt + 4 * k >= D) continue;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A.r[k], B.r[k], acc, 0, 0, 0);
    }
  }
  // The GP prior of a Pose2 robot the same way: its Hessian blocks are quadrant-weighted congruences X^T Qc^-1 Y with
  // X, Y = [diag(J (3x3), s I) | I] (D x n): J = J3, s = +1 for the second state of the interval (Bm), J = J1, s = -1
  // for the first (A); the 2 x 2 weights of Q^-1(delta_t) and Phi are applied per quadrant afterwards (build_tiles).
  __device__ __forceinline__ Tile gp_tile(const double* J, double s, int rt, int cbase) const {
    const int j = cbase + (lane & 15), av = j >= D, kk = j - av * D;
    Tile T;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a = 16 * rt + (lane >> 4) + 4 * k;
      const bool pose = !av && a < 3 && kk < 3;
      const double m = J[pose ? a * 3 + kk : 0];
      const double v = pose ? m : ((a == kk && (av || a >= 3)) ? (av ? 1.0 : s) : 0.0);
      T.r[k] = (a < D && j < n) ? v : 0.0;
    }
    return T;
  }
  __device__ __forceinline__ Tile qc_tile(int rt, int ct) const {
    const int a2 = 16 * ct + (lane & 15);
    Tile T;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int a = 16 * rt + (lane >> 4) + 4 * k;
      const bool in = a < D && a2 < D;
      const double v = P.Qc_inv[in ? a * D + a2 : 0];
      T.r[k] = in ? v : 0.0;
    }
    return T;
  }
  // acc += E_L^T G E_R for one output tile; G[rb][ra], EL[ra], ER[rb] = tiles along the configuration dimension
  __device__ __forceinline__ static void congruence(const Tile (&G)[KT][KT], const Tile (&EL)[KT], const Tile (&ER)[KT],
                                                    v4d& acc) {
#pragma unroll
    for (int ra = 0; ra < KT; ra++) {
      v4d ge = {0.0, 0.0, 0.0, 0.0};   // (G E_R) rows of tile row ra; G is symmetric: G[rb][ra]^T = G[ra][rb]
#pragma unroll
      for (int rb = 0; rb < KT; rb++) mfma_atb_acc(G[rb][ra], ER[rb], rb, ge);
      Tile GE;
#pragma unroll
      for (int k = 0; k < 4; k++) GE.r[k] = ge[k];
      mfma_atb_acc(EL[ra], GE, ra, acc);
    }
  }
  bool skip_interp = false;   // kernels that add the interpolated factors of all tiles of a wide block in one go
  __device__ __forceinline__ void lie_interp(const Slot& si, const Slot& sn, bool has_prev, bool has_next, bool want_c,
                                             double (&dk)[4], double (&hrk)[4], double (&hlk)[4]) const {
    if (skip_interp) return;
    v4d ad = {dk[0], dk[1], dk[2], dk[3]}, ar_ = {hrk[0], hrk[1], hrk[2], hrk[3]}, al = {hlk[0], hlk[1], hlk[2], hlk[3]};
    // one interpolated point: rows take the role of state i in the interval, the coupling's columns the other role
    auto point = [&](const double* pt, const double* cf, int role, v4d& acc_d, v4d& acc_c) {
      const double* M = pt + RECP;
      Tile G[KT][KT], EL[KT], ER[KT];
#pragma unroll
      for (int ra = 0; ra < KT; ra++) {
#pragma unroll
        for (int rb = 0; rb < KT; rb++) G[rb][ra] = g_tile(pt, rb, ra);
        EL[ra] = hint_tile(M, cf, role, ra, row0);
        ER[ra] = hint_tile(M, cf, role, ra, col0);
      }
      congruence(G, EL, ER, acc_d);
      if (want_c) {
#pragma unroll
        for (int ra = 0; ra < KT; ra++) ER[ra] = hint_tile(M, cf, 1 - role, ra, col0);
        congruence(G, EL, ER, acc_c);
      }
    };
#pragma unroll 1
    for (int jj = 0; jj < P.I; jj++) {
      const double* cf = si.coef(jj) + 16;       // l11 l12 p11 p12
      if (has_prev) point(si.pt(jj), cf, 1, ad, al);   // interval i: state i second; H_{i,i-1}: columns of the first state
      if (has_next) point(sn.pt(jj), cf, 0, ad, ar_);  // interval i + 1: state i first; H_{i,i+1}: columns of the second
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      dk[k] = ad[k];
      hrk[k] = ar_[k];
      hlk[k] = al[k];
    }
  }

  // Tiles of block i: S = [D_i | -g_i in column RHSCOL], Cl = H_{i,i-1}, Cr = H_{i,i+1}; returns this
  // lane's share of the block's graph-error contribution (to be wave-summed; not yet halved).
  // si = slot of interval i (its unary point is state i), sn = slot of interval i+1; z = state i.
  // want_c: also build the couplings (only odd blocks and the export / Dogleg paths need them).
  //
  // Two phases.  (1) Row-owner lanes: lane r < 16 owns row row0 + r of the tile and forms that row's diagonal
  // additions (priors, limits, dynamics, replanner priors), its gradient entry and its error share -- every
  // global load this needs is issued once, at the top, by all owner lanes together.  (2) All lanes form the matrix
  // entries from the LDS records; the owners' values are then dropped into the diagonal and the rhs column with
  // wave shuffles.
  __device__ __forceinline__ double build_tiles(int i, const Slot& si, const Slot& sn, const double* z, Tile& S,
                                                Tile& Cl, Tile& Cr, bool want_c) const {
    const int I = P.I, N = P.N;
    const bool has_prev = i > 0, has_next = i < N;
    constexpr bool lie = LIE;
    const double dt = P.delta_t, w0 = P.Winv[0], w1 = P.Winv[1], w3 = P.Winv[3];
    const double* J3i = si.gpr() + n + 1 + 9;   // of interval i   (state i is the second state)
    const double* J1n = sn.gpr() + n + 1;       // of interval i+1 (state i is the first state)
    const double* J1i = si.gpr() + n + 1;
    const double* J3n = sn.gpr() + n + 1 + 9;

    G2_BSTAMP(0);
    // ---------------------------------------------------------------- phase 1: row owners
    const bool need_diag = (row0 == col0), need_rhs = (rhscol >= col0 && rhscol < col0 + 16);
    const int orow = row0 + lane;                       // the row this lane owns (lanes 0..15)
    const bool owner = lane < 16 && orow < n && (need_diag || need_rhs);
    const int nxp = pb.xp_n[b];
    double o_dd = 0.0, o_gg = 0.0, o_ee = 0.0;
    if (owner) {
      const int rho = orow, ar = rho >= D, kr = rho - ar * D;
      // PriorFactor on x_0, v_0, x_N, v_N; the one on x_N can be switched off (a goal factor stands in)
      const bool prior_here = (i == 0 || (i == N && pb.goal_on[b] && (ar || !P.end_conf_prior_off)));
      const double zz = z[rho];
      double z0 = 0.0, z1 = 0.0, z2 = 0.0;
      if (lie) { z0 = z[0]; z1 = z[1]; z2 = z[2]; }
      double dd = 0.0, gg = 0.0, ee = 0.0;
      if (prior_here) {  // PriorFactor on x_0, v_0, x_N, v_N  (BatchTrajOptimizer-inl.h:41-48)
        const double* tg = (i == 0) ? (ar ? pb.start_vel : pb.start_conf) : (ar ? pb.end_vel : pb.end_conf);
        tg += (size_t)b * D;
        const double w = ar ? P.vel_prior_w : P.conf_prior_w;
        double dz = zz - tg[kr];
        if (lie && !ar && kr < 3) {
          // gtsam 4.0 PriorFactor<Pose2Vector>: error = -Local(x, prior), H = I
          const P2 bt = pose2_between(P2{z0, z1, z2}, P2{tg[0], tg[1], tg[2]});
          dz = -(kr == 0 ? bt.x : kr == 1 ? bt.y : bt.th);
        }
        dd += w;
        gg += w * dz;
        ee += w * dz * dz;
      }
      double Hh;
      if (!ar && P.flag_pos_limit && !(lie && kr < 3)) {  // JointLimitFactorPose2Vector skips the base
        const double e = hinge_limit(zz, P.pos_lo[kr], P.pos_hi[kr], P.pos_th[kr], Hh);
        dd += P.pos_w[kr] * Hh * Hh;
        gg += P.pos_w[kr] * Hh * e;
        ee += P.pos_w[kr] * e * e;
      }
      if (ar && P.flag_vel_limit) {
        const double e = hinge_limit(zz, -P.vel_lim[kr], P.vel_lim[kr], P.vel_th[kr], Hh);
        dd += P.vel_w[kr] * Hh * Hh;
        gg += P.vel_w[kr] * Hh * e;
        ee += P.vel_w[kr] * e * e;
      }
      if (ar && kr == 1 && P.vdyn_w > 0.0) {  // VehicleDynamicsFactorPose2Vector: r = v(1)
        dd += P.vdyn_w;
        gg += P.vdyn_w * zz;
        ee += P.vdyn_w * zz * zz;
      }
      // extra per-state priors of the replanner (fixConfigAndVel / addStateEstimate,
      // planner/ISAM2TrajOptimizer-inl.h:159-195): full information matrices
      for (int e = 0; e < nxp; e++) {
        const size_t xe = (size_t)b * XP_MAX + e;
        if (pb.xp_state[xe] != i || (ar && !pb.xp_has_vel[xe])) continue;
        const double* Wm = pb.xp_info + (xe * 2 + ar) * D * D + (size_t)kr * D;  // row kr
        const double* tg = pb.xp_target + xe * n + ar * D;
        double wr = 0.0, rk = 0.0;
        for (int cc = 0; cc < D; cc++) {
          double rc = z[ar * D + cc] - tg[cc];
          if (lie && !ar && cc < 3) {
            const P2 bt = pose2_between(P2{z0, z1, z2}, P2{tg[0], tg[1], tg[2]});
            rc = -(cc == 0 ? bt.x : cc == 1 ? bt.y : bt.th);
          }
          wr = fma(Wm[cc], rc, wr);
          if (cc == kr) rk = rc;
        }
        dd += Wm[kr];
        gg += wr;
        ee += wr * rk;
      }
      if (need_rhs) {
        if (!ar) gg += si.pt(I)[NG + kr];
        for (int jj = 0; jj < I; jj++) {
          const double* cf = si.coef(jj) + 16;     // l11 l12 p11 p12
          const double cp = ar ? cf[3] : cf[2], cl = ar ? cf[1] : cf[0];
          if (lie) {
            if (has_prev) gg += hint_vec(si.pt(jj), si.pt(jj) + RECP + (ar ? 27 : 18), cp, kr);
            if (has_next) gg += hint_vec(sn.pt(jj), sn.pt(jj) + RECP + (ar ? 9 : 0), cl, kr);
            continue;
          }
          if (has_prev) gg = fma(cp, si.pt(jj)[NG + kr], gg);
          if (has_next) gg = fma(cl, sn.pt(jj)[NG + kr], gg);
        }
        if (!lie) {
          // GP prior gradient: + Phi^T u_{i+1} - u_i
          if (has_next) gg += ar ? (dt * sn.gpr()[kr] + sn.gpr()[D + kr]) : sn.gpr()[kr];
          if (has_prev) gg -= si.gpr()[rho];
        } else {
          // + A_{i+1}^T u_{i+1} + Bm_i^T u_i
          if (has_next) {
            if (ar) gg += -dt * sn.gpr()[kr] - sn.gpr()[D + kr];
            else if (kr < 3) gg += J1n[0 * 3 + kr] * sn.gpr()[0] + J1n[1 * 3 + kr] * sn.gpr()[1] + J1n[2 * 3 + kr] * sn.gpr()[2];
            else gg -= sn.gpr()[kr];
          }
          if (has_prev) {
            if (ar) gg += si.gpr()[D + kr];
            else if (kr < 3) gg += J3i[0 * 3 + kr] * si.gpr()[0] + J3i[1 * 3 + kr] * si.gpr()[1] + J3i[2 * 3 + kr] * si.gpr()[2];
            else gg += si.gpr()[kr];
          }
        }
      }
      o_dd = dd;
      o_gg = gg;
      o_ee = need_diag ? ee : 0.0;
    }

    G2_BSTAMP(1);
    // ---------------------------------------------------------------- phase 2: matrix entries
    double dk[4] = {0, 0, 0, 0}, hrk[4] = {0, 0, 0, 0}, hlk[4] = {0, 0, 0, 0};
    const int ac = a_col, kc = k_col;
    if constexpr (!lie) {
      // Vector-space robots: branch-free per entry.  Lanes / rows outside the n x n block compute on clamped
      // indices and are zeroed at the end, so there is no per-entry exec masking.  Entry (rho, c) of
      //   D_i      = KB + KA + [xx] G_unary + sum_jj Psi_r Psi_c G_jj(interval i) + Lam_r Lam_c G_jj(interval i+1)
      //   H_{i,i+1} = KO   + sum_jj Lam_r Psi_c G_jj(interval i+1),   H_{i,i-1} = KO^T + sum_jj Psi_r Lam_c G_jj(interval i)
      // with the 2x2 weight products read from the staged table (coefq) at [quad * 4 + ar * 2 + ac].
      const int cc_ = min(c, n - 1);
      const double fp = has_prev ? 1.0 : 0.0, fn = has_next ? 1.0 : 0.0;
      int wq[4];
      const double* un = si.pt(I);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int rr_ = min(g + 4 * k, n - 1);
        wq[k] = a_row[k] * 2 + ac;
        double d = fp * P.KB[rr_ * n + cc_] + fn * P.KA[rr_ * n + cc_];
        const double u = un[tri[k]];
        d += (wq[k] == 0) ? u : 0.0;            // unary obstacle factor at state i: the xx quadrant
        dk[k] = d;
        if (want_c) {
          hrk[k] = fn * P.KO[rr_ * n + cc_];
          hlk[k] = fp * P.KO[cc_ * n + rr_];
        }
      }
      G2_BSTAMP(2);
      if (has_prev && has_next) {
#pragma unroll 1
        for (int jj = 0; jj < I; jj++) {
          const double* cq = si.coef(jj);
          const double* pp_ = si.pt(jj);
          const double* pn_ = sn.pt(jj);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const double Gp = pp_[tri[k]], Gn = pn_[tri[k]];
            dk[k] = fma(cq[wq[k]], Gp, dk[k]);
            dk[k] = fma(cq[4 + wq[k]], Gn, dk[k]);
            if (want_c) {
              hrk[k] = fma(cq[8 + wq[k]], Gn, hrk[k]);
              hlk[k] = fma(cq[12 + wq[k]], Gp, hlk[k]);
            }
          }
        }
      } else {
        for (int jj = 0; jj < I; jj++) {
          const double* cq = si.coef(jj);
          const double* pp_ = si.pt(jj);
          const double* pn_ = sn.pt(jj);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (has_prev) {
              const double Gp = pp_[tri[k]];
              dk[k] = fma(cq[wq[k]], Gp, dk[k]);
              if (want_c) hlk[k] = fma(cq[12 + wq[k]], Gp, hlk[k]);
            }
            if (has_next) {
              const double Gn = pn_[tri[k]];
              dk[k] = fma(cq[4 + wq[k]], Gn, dk[k]);
              if (want_c) hrk[k] = fma(cq[8 + wq[k]], Gn, hrk[k]);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        dk[k] = valid[k] ? dk[k] : 0.0;
        hrk[k] = valid[k] ? hrk[k] : 0.0;
        hlk[k] = valid[k] ? hlk[k] : 0.0;
      }
    } else {
    // GP prior blocks: A = d r / d z_first = [[J1, -dt I],[0, -I]],  Bm = d r / d z_second = [[J3, 0],[0, I]];
    // the four congruences with Qc^-1 on the matrix cores, the quadrant weights of W = Q^-1 (x) Qc^-1 per entry
    v4d xdp = {0.0, 0.0, 0.0, 0.0}, xl = xdp, xdn = xdp, xr = xdp;
    {
      Tile Qt[KT][KT], EL[KT], ER[KT];
#pragma unroll
      for (int ra = 0; ra < KT; ra++)
#pragma unroll
        for (int rb = 0; rb < KT; rb++) Qt[rb][ra] = qc_tile(rb, ra);
      if (has_prev) {  // Bm^T W Bm of interval i ; H_{i,i-1} = Bm^T W A of interval i
#pragma unroll
        for (int ra = 0; ra < KT; ra++) { EL[ra] = gp_tile(J3i, 1.0, ra, row0); ER[ra] = gp_tile(J3i, 1.0, ra, col0); }
        congruence(Qt, EL, ER, xdp);
        if (want_c) {
#pragma unroll
          for (int ra = 0; ra < KT; ra++) ER[ra] = gp_tile(J1i, -1.0, ra, col0);
          congruence(Qt, EL, ER, xl);
        }
      }
      if (has_next) {  // A^T W A of interval i+1 ; H_{i,i+1} = A^T W Bm of interval i+1
#pragma unroll
        for (int ra = 0; ra < KT; ra++) { EL[ra] = gp_tile(J1n, -1.0, ra, row0); ER[ra] = gp_tile(J1n, -1.0, ra, col0); }
        congruence(Qt, EL, ER, xdn);
        if (want_c) {
#pragma unroll
          for (int ra = 0; ra < KT; ra++) ER[ra] = gp_tile(J3n, 1.0, ra, col0);
          congruence(Qt, EL, ER, xr);
        }
      }
    }
    const double cAxv = -(dt * w0 + w1), cAvv = dt * dt * w0 + 2.0 * dt * w1 + w3, cOvv = -(dt * w1 + w3);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (!valid[k]) continue;
      const int ar = a_row[k];
      double d = (ar ? (ac ? w3 : w1) : (ac ? w1 : w0)) * xdp[k] + (ar ? (ac ? cAvv : cAxv) : (ac ? cAxv : w0)) * xdn[k];
      const double hl = (ar ? (ac ? cOvv : w1) : (ac ? cAxv : w0)) * xl[k];
      const double hr = (ar ? (ac ? cOvv : cAxv) : (ac ? w1 : w0)) * xr[k];
      if (!ar && !ac) d += si.pt(I)[tri[k]];  // unary obstacle factor at state i
      dk[k] = d;
      hrk[k] = hr;
      hlk[k] = hl;
    }
    // interpolated obstacle factors of interval i (state i is the second state) and interval i + 1 (first state):
    // congruences E_L^T G E_R on the matrix cores, see lie_interp
    lie_interp(si, sn, has_prev, has_next, want_c, dk, hrk, hlk);
    }
    G2_BSTAMP(3);
    for (int e = 0; e < nxp; e++) {
      const size_t xe = (size_t)b * XP_MAX + e;
      if (pb.xp_state[xe] != i) continue;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // the diagonal entries arrive with the owners' values below
        if (!valid[k] || a_row[k] != ac || (ac && !pb.xp_has_vel[xe]) || (g + 4 * k) == c) continue;
        dk[k] += pb.xp_info[(xe * 2 + ac) * D * D + (size_t)k_row[k] * D + kc];
      }
    }
    // this block's share of the graph error: unary point of state i, the interpolated points and the
    // GP prior of the interval ending at i, plus the owners' prior / limit / dynamics terms of state i
    double err_acc = o_ee;
    if (lane <= I && (lane == I || has_prev)) err_acc += si.pt(lane)[NG + D];
    if (lane == 63 && has_prev) err_acc += si.gpr()[n];
    // owners' values -> diagonal and rhs column (-g_i in column rhscol)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int rho = g + 4 * k, src = rho - row0;   // src = (lane >> 4) + 4 k, always a lane in 0..15
      const double dsh = __shfl(o_dd, src, 64), gsh = __shfl(o_gg, src, 64);
      double v = dk[k];
      if (rho < n && c == rho) v += dsh;
      if (rho < n && c == rhscol) v = -gsh;
      S.r[k] = v;
      Cr.r[k] = hrk[k];
      Cl.r[k] = hlk[k];
    }
    G2_BSTAMP(4);
    return err_acc;
  }
};

}  // namespace g2
