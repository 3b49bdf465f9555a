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

constexpr int GP_EXTRA_LIE = 18;  // J1 (9) + J3 (9) appended to the GP record of Pose2 robots

template <int D, bool LIE>
struct Assembler {
  static constexpr int n = 2 * D, NG = D * (D + 1) / 2, RECP = NG + D + 1;  // per-point record
  static constexpr int GPN = n + 1 + GP_EXTRA_LIE;
  static constexpr int PT_EXTRA_LIE = 36;  // M1..M4 (3x3 pose blocks of the interpolation Jacobians)
  static constexpr int RECMAX = RECP + PT_EXTRA_LIE;

  // LDS image of one interval (dynamic shared memory, sized at launch from I and the record
  // lengths): pts[jj][REC] for jj = 0..I (I = unary of the end state), then the GP record
  // u = Q^-1 r (n), r^T u, [J1 (9), J3 (9)]
  struct Slot {
    double* base;
    int rec;   // P.REC
    int npts;  // I + 1
    __device__ __forceinline__ double* pt(int jj) const { return base + jj * rec; }
    __device__ __forceinline__ const double* gpr() const { return base + npts * rec; }
    __device__ __forceinline__ double* gpw() const { return base + npts * rec; }
  };
  __host__ __device__ static int slot_doubles(int I, int REC, int GPREC) { return (I + 1) * REC + GPREC; }
  __device__ __forceinline__ Slot make_slot(double* smem, int which) const {
    return Slot{smem + which * slot_doubles(P.I, P.REC, P.GPREC), P.REC, P.I + 1};
  }

  const PlanParams& P;
  const PlanBuffers& pb;
  const double* rec;
  const double* gpu;
  int b, lane, c, g;   // c / g: this lane's global column / first global row (tile offsets included)
  int rhscol;          // column that carries the right-hand side (-g_i)
  // per-lane static decode of its 4 rows
  int tri[4];
  bool valid[4];   // rho < n && c < n
  int a_row[4], k_row[4], a_col, k_col;

  // row0 / col0: offset of the 16x16 tile inside a wider block (blocks with 2 dof > 15 are exported as
  // 2x2 tiles by k_export_normal_eq); rhscol_: global column of the right-hand side
  __device__ Assembler(const PlanParams& P_, const PlanBuffers& pb_, const double* rec_, const double* gpu_,
                       int b_, int lane_, int row0 = 0, int col0 = 0, int rhscol_ = RHSCOL)
      : P(P_), pb(pb_), rec(rec_), gpu(gpu_), b(b_), lane(lane_), c(col0 + (lane_ & 15)), g(row0 + (lane_ >> 4)),
        rhscol(rhscol_) {
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
  // only the unary point of state 0, intervals beyond N read as zeros).  All loads of both intervals
  // are issued before the first LDS store so the wavefront pays one memory latency, not one per
  // 64 values; NLD bounds the per-lane load count (checked on the host).
  static constexpr int NLD = (D <= 7) ? 10 : 16;
  // count = 1 stages interval iv only (kernels whose wavefronts share their slots)
  __device__ __forceinline__ void stage2(int iv, const Slot& s0, const Slot& s1, int count = 2) const {
    const int I = P.I;
    const double* rb = rec + (size_t)b * P.REC * P.Ppad;
    const double* gb = gpu + (size_t)b * P.GPREC * P.Npad;
    double val[2][NLD];
#pragma unroll
    for (int w = 0; w < 2; w++) {
      if (w >= count) break;
      const int ivw = iv + w;
      const int npt = (ivw == 0) ? 1 : I + 1;
      const int nv = P.REC * npt;
      const int p0 = (ivw == 0) ? 0 : 1 + (ivw - 1) * (I + 1);
      const bool in_range = ivw <= P.N;
      const float inv_npt = 1.0f / (float)npt;
#pragma unroll
      for (int m = 0; m < NLD; m++) {
        const int v = lane + 64 * m;
        double x = 0.0;
        if (in_range && v < nv + P.GPREC) {
          if (v < nv) {
            const int k = (int)(((float)v + 0.5f) * inv_npt), jj = v - k * npt;  // exact for v < 2^20
            x = rb[(size_t)k * P.Ppad + p0 + jj];
          } else if (ivw > 0) {
            x = gb[(size_t)(v - nv) * P.Npad + ivw];
          }
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
      const int nv = P.REC * npt;
      const float inv_npt = 1.0f / (float)npt;
#pragma unroll
      for (int m = 0; m < NLD; m++) {
        const int v = lane + 64 * m;
        if (v < nv) {
          const int k = (int)(((float)v + 0.5f) * inv_npt), jj = v - k * npt;
          s.pt((ivw == 0) ? I : jj)[k] = val[w][m];
        } else if (v < nv + P.GPREC) {
          s.gpw()[v - nv] = val[w][m];
        }
      }
    }
  }

  // sum_{a,b} L[a][kr] Qc^-1[a][b] R[b][kc] where L / R are either the identity (ML == nullptr) or
  // the DxD block-diagonal Jacobian diag(M (3x3, row-major), sign * I)
  __device__ __forceinline__ double lie_quad(const double* ML, double sL, const double* MR, double sR, int kr,
                                             int kc) const {
    double acc = 0.0;
    const int a0 = (ML && kr < 3) ? 0 : kr, a1 = (ML && kr < 3) ? 3 : kr + 1;
    const int b0 = (MR && kc < 3) ? 0 : kc, b1 = (MR && kc < 3) ? 3 : kc + 1;
    for (int a = a0; a < a1; a++) {
      const double la = !ML ? 1.0 : (kr < 3 ? ML[a * 3 + kr] : sL);
      for (int bb = b0; bb < b1; bb++) {
        const double rb_ = !MR ? 1.0 : (kc < 3 ? MR[bb * 3 + kc] : sR);
        acc = fma(la * rb_, P.Qc_inv[a * D + bb], acc);
      }
    }
    return acc;
  }

  // packed upper-triangular lookup of the point's G = J^T J / sigma^2
  __device__ __forceinline__ static double Gat(const double* pt, int a, int bb) {
    const int lo = min(a, bb), hi = max(a, bb);
    return pt[lo * D - (lo * (lo - 1)) / 2 + (hi - lo)];
  }
  // (Hint_L^T G Hint_R)[kr][kc] for Hint = diag(M (3x3), s I): Pose2 interpolated obstacle factor
  __device__ __forceinline__ static double hint_quad(const double* pt, const double* ML, double sL,
                                                     const double* MR, double sR, int kr, int kc) {
    double acc = 0.0;
    const int a0 = kr < 3 ? 0 : kr, a1 = kr < 3 ? 3 : kr + 1;
    const int b0 = kc < 3 ? 0 : kc, b1 = kc < 3 ? 3 : kc + 1;
    for (int a = a0; a < a1; a++) {
      const double la = kr < 3 ? ML[a * 3 + kr] : sL;
      for (int bb = b0; bb < b1; bb++) acc = fma(la * (kc < 3 ? MR[bb * 3 + kc] : sR), Gat(pt, a, bb), acc);
    }
    return acc;
  }
  // (Hint^T g)[kr]
  __device__ __forceinline__ static double hint_vec(const double* pt, const double* M, double s, int kr) {
    const double* gp_ = pt + NG;
    if (kr >= 3) return s * gp_[kr];
    return M[0 * 3 + kr] * gp_[0] + M[1 * 3 + kr] * gp_[1] + M[2 * 3 + kr] * gp_[2];
  }

  // Tiles of block i: S = [D_i | -g_i in column RHSCOL], Cl = H_{i,i-1}, Cr = H_{i,i+1}; returns this
  // lane's share of the block's graph-error contribution (to be wave-summed; not yet halved).
  // si = slot of interval i (its unary point is state i), sn = slot of interval i+1; z = state i.
  // want_c: also build the couplings (only odd blocks and the export / Dogleg paths need them).
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
    double dk[4] = {0, 0, 0, 0}, hrk[4] = {0, 0, 0, 0}, hlk[4] = {0, 0, 0, 0};
    const int ac = a_col, kc = k_col;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (!valid[k]) continue;
      const int ar = a_row[k], kr = k_row[k], rho = g + 4 * k;
      double d = 0.0, hr = 0.0, hl = 0.0;
      if (!lie) {
        // constant GP prior blocks: KB = Q^-1 (state i second), KA = Phi^T Q^-1 Phi (state i first),
        // KO = -Phi^T Q^-1 = H_{i,i+1}
        d = (has_prev ? P.KB[rho * n + c] : 0.0) + (has_next ? P.KA[rho * n + c] : 0.0);
        if (want_c) {
          hr = has_next ? P.KO[rho * n + c] : 0.0;
          hl = has_prev ? P.KO[c * n + rho] : 0.0;
        }
      } else {
        // A = d r / d z_first = [[J1, -dt I],[0, -I]],  Bm = d r / d z_second = [[J3, 0],[0, I]]
        const double cAxv = -(dt * w0 + w1), cAvv = dt * dt * w0 + 2.0 * dt * w1 + w3, cOvv = -(dt * w1 + w3);
        if (has_prev) {  // Bm^T W Bm of interval i ; H_{i,i-1} = Bm^T W A of interval i
          const double* L = ar ? nullptr : J3i;
          d += (ar ? (ac ? w3 : w1) : (ac ? w1 : w0)) * lie_quad(L, 1.0, ac ? nullptr : J3i, 1.0, kr, kc);
          if (want_c) hl = (ar ? (ac ? cOvv : w1) : (ac ? cAxv : w0)) * lie_quad(L, 1.0, ac ? nullptr : J1i, -1.0, kr, kc);
        }
        if (has_next) {  // A^T W A of interval i+1 ; H_{i,i+1} = A^T W Bm of interval i+1
          const double* L = ar ? nullptr : J1n;
          d += (ar ? (ac ? cAvv : cAxv) : (ac ? cAxv : w0)) * lie_quad(L, -1.0, ac ? nullptr : J1n, -1.0, kr, kc);
          if (want_c) hr = (ar ? (ac ? cOvv : cAxv) : (ac ? w1 : w0)) * lie_quad(L, -1.0, ac ? nullptr : J3n, 1.0, kr, kc);
        }
      }
      if (!ar && !ac) d += si.pt(I)[tri[k]];  // unary obstacle factor at state i
      dk[k] = d;
      hrk[k] = hr;
      hlk[k] = hl;
    }
    // interpolated obstacle factors: interval i (state i second) and interval i+1 (state i first)
    for (int jj = 0; jj < I; jj++) {
      const GpCoef cf = P.coef[jj];
      const double w1c = ac ? cf.l12 : cf.l11, w2c = ac ? cf.p12 : cf.p11;
      const double* pp_ = si.pt(jj);
      const double* pn_ = sn.pt(jj);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!valid[k]) continue;
        const int ar = a_row[k], kr = k_row[k];
        const double w1r = ar ? cf.l12 : cf.l11, w2r = ar ? cf.p12 : cf.p11;
        if (lie) {
          // Hint_k = diag(M_k, s_k I), k = (x_first, v_first, x_second, v_second)
          if (has_prev) {
            const double* M = pp_ + RECP;
            dk[k] += hint_quad(pp_, M + (ar ? 27 : 18), w2r, M + (ac ? 27 : 18), w2c, kr, kc);
            if (want_c) hlk[k] += hint_quad(pp_, M + (ar ? 27 : 18), w2r, M + (ac ? 9 : 0), w1c, kr, kc);
          }
          if (has_next) {
            const double* M = pn_ + RECP;
            dk[k] += hint_quad(pn_, M + (ar ? 9 : 0), w1r, M + (ac ? 9 : 0), w1c, kr, kc);
            if (want_c) hrk[k] += hint_quad(pn_, M + (ar ? 9 : 0), w1r, M + (ac ? 27 : 18), w2c, kr, kc);
          }
          continue;
        }
        const int t = tri[k];
        if (has_prev) {
          const double Gp = pp_[t];
          dk[k] = fma(w2r * w2c, Gp, dk[k]);
          if (want_c) hlk[k] = fma(w2r * w1c, Gp, hlk[k]);  // rows: state i (second), cols: state i-1 (first)
        }
        if (has_next) {
          const double Gn = pn_[t];
          dk[k] = fma(w1r * w1c, Gn, dk[k]);
          if (want_c) hrk[k] = fma(w1r * w2c, Gn, hrk[k]);  // rows: state i (first), cols: state i+1 (second)
        }
      }
    }
    const int nxp = pb.xp_n[b];
    for (int e = 0; e < nxp; e++) {
      const size_t xe = (size_t)b * XP_MAX + e;
      if (pb.xp_state[xe] != i) continue;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // the diagonal entries are added with the other diagonal terms below
        if (!valid[k] || a_row[k] != ac || (ac && !pb.xp_has_vel[xe]) || (g + 4 * k) == c) continue;
        dk[k] += pb.xp_info[(xe * 2 + ac) * D * D + (size_t)k_row[k] * D + kc];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      S.r[k] = dk[k];
      Cr.r[k] = hrk[k];
      Cl.r[k] = hlk[k];
    }
    // this block's share of the graph error: unary point of state i, the interpolated points and the
    // GP prior of the interval ending at i, plus (below) the prior / limit / dynamics terms of state i
    double err_acc = 0.0;
    if (lane <= I && (lane == I || has_prev)) err_acc = si.pt(lane)[NG + D];
    if (lane == 63 && has_prev) err_acc += si.gpr()[n];
    // diagonal terms (priors, limits, dynamics) and the gradient column (-g_i in column RHSCOL)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int rho = g + 4 * k;
      if (rho >= n) continue;
      const bool on_diag = (c == rho), on_rhs = (c == rhscol);
      if (!on_diag && !on_rhs) continue;
      const int ar = a_row[k], kr = k_row[k];
      double dd = 0.0, gg = 0.0, ee = 0.0;
      const double zz = z[rho];
      if (i == 0 || (i == N && pb.goal_on[b])) {  // PriorFactor on x_0, v_0, x_N, v_N  (BatchTrajOptimizer-inl.h:41-48)
        const double* tg = (i == 0) ? (ar ? pb.start_vel : pb.start_conf) : (ar ? pb.end_vel : pb.end_conf);
        tg += (size_t)b * D;
        const double w = ar ? P.vel_prior_w : P.conf_prior_w;
        double dz = zz - tg[kr];
        if (lie && !ar && kr < 3) {
          // gtsam 4.0 PriorFactor<Pose2Vector>: error = -Local(x, prior), H = I
          const P2 bt = pose2_between(P2{z[0], z[1], z[2]}, P2{tg[0], tg[1], tg[2]});
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
            const P2 bt = pose2_between(P2{z[0], z[1], z[2]}, P2{tg[0], tg[1], tg[2]});
            rc = -(cc == 0 ? bt.x : cc == 1 ? bt.y : bt.th);
          }
          wr = fma(Wm[cc], rc, wr);
          if (cc == kr) rk = rc;
        }
        dd += Wm[kr];
        gg += wr;
        ee += wr * rk;
      }
      if (on_diag) {
        S.r[k] += dd;
        err_acc += ee;
      }
      if (on_rhs) {
        if (!ar) gg += si.pt(I)[NG + kr];
        for (int jj = 0; jj < I; jj++) {
          const GpCoef cf = P.coef[jj];
          if (lie) {
            if (has_prev) gg += hint_vec(si.pt(jj), si.pt(jj) + RECP + (ar ? 27 : 18), ar ? cf.p12 : cf.p11, kr);
            if (has_next) gg += hint_vec(sn.pt(jj), sn.pt(jj) + RECP + (ar ? 9 : 0), ar ? cf.l12 : cf.l11, kr);
            continue;
          }
          if (has_prev) gg = fma(ar ? cf.p12 : cf.p11, si.pt(jj)[NG + kr], gg);
          if (has_next) gg = fma(ar ? cf.l12 : cf.l11, sn.pt(jj)[NG + kr], gg);
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
        S.r[k] = -gg;
      }
    }
    return err_acc;
  }
};

}  // namespace g2
