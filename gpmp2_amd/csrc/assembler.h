// assembler.h -- builds block-tridiagonal tiles from the per-point records of k_linearize.
#pragma once
#include "device_math.h"
#include "plan.h"
#include "tiles.h"

namespace g2 {

// =============================================================================== assembly
// Builds, for block i of trajectory b, the diagonal tile D_i, the coupling tile
// [H_{i,i+1} | -g_i] and (optionally) nothing else, from the point records staged in LDS.
// Kronecker structure: a point with interpolation scalars c contributes
// (c c^T) (x) G to the 2x2-block window and c (x) g to the gradient.
template <int D>
struct Assembler {
  static constexpr int n = 2 * D, NG = D * (D + 1) / 2, RECP = NG + D + 1;  // per-point record
  static constexpr int NROUND = (RECP * (MAXI + 1) + n + 1 + 63) / 64;

  // LDS image of one interval: pts[jj][RECP] for jj = 0..I (I = unary of the end state), then
  // the GP vector u (n) and energy
  struct Slot {
    double pts[MAXI + 1][RECP];
    double gp[n + 1];
  };

  const PlanParams& P;
  const PlanBuffers& pb;
  const double* rec;
  const double* gpu;
  int b, lane, c, g;
  // per-lane static decode of its 4 rows
  int tri[4];
  bool valid[4];   // rho < n && c < n
  int a_row[4], k_row[4], a_col, k_col;
  double KA[4], KB[4], KO[4], KOt[4];

  __device__ Assembler(const PlanParams& P_, const PlanBuffers& pb_, const double* rec_, const double* gpu_,
                       int b_, int lane_)
      : P(P_), pb(pb_), rec(rec_), gpu(gpu_), b(b_), lane(lane_), c(lane_ & 15), g(lane_ >> 4) {
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
      KA[k] = valid[k] ? P.KA[rho * n + c] : 0.0;
      KB[k] = valid[k] ? P.KB[rho * n + c] : 0.0;
      KO[k] = valid[k] ? P.KO[rho * n + c] : 0.0;
      KOt[k] = valid[k] ? P.KO[c * n + rho] : 0.0;
    }
  }

  // global -> LDS for interval `iv` in one go (no software pipelining; used by k_assemble where
  // every wavefront handles a single block).  Intervals beyond N read as zeros.
  __device__ __forceinline__ void stage(int iv, Slot& s) const {
    const int I = P.I;
    const int npt = (iv == 0) ? 1 : I + 1;
    const int nv = RECP * npt;
    const int p0 = (iv == 0) ? 0 : 1 + (iv - 1) * (I + 1);
    const double* rb = rec + (size_t)b * P.REC * P.Ppad;
    const double* gb = gpu + (size_t)b * (n + 1) * P.Npad;
    const bool in_range = iv <= P.N;
    for (int v = lane; v < nv + n + 1; v += 64) {
      if (v < nv) {
        const int k = v / npt, jj = v - k * npt;
        s.pts[(iv == 0) ? I : jj][k] = in_range ? rb[(size_t)k * P.Ppad + p0 + jj] : 0.0;
      } else {
        s.gp[v - nv] = (in_range && iv > 0) ? gb[(size_t)(v - nv) * P.Npad + iv] : 0.0;
      }
    }
  }

  // global -> registers for interval `iv` (1..N); interval 0 is just the unary point of state 0
  __device__ __forceinline__ void prefetch(int iv, double (&pf)[NROUND]) const {
    const int I = P.I;
    const int npt = (iv == 0) ? 1 : I + 1;
    const int nv = RECP * npt;
    const int p0 = (iv == 0) ? 0 : 1 + (iv - 1) * (I + 1);
    const double* rb = rec + (size_t)b * P.REC * P.Ppad;
    const double* gb = gpu + (size_t)b * (n + 1) * P.Npad;
#pragma unroll
    for (int m = 0; m < NROUND; m++) {
      const int v = lane + 64 * m;
      double x = 0.0;
      if (iv <= P.N) {
        if (v < nv) {
          const int k = v / npt, jj = v - k * npt;
          x = rb[(size_t)k * P.Ppad + p0 + jj];
        } else if (iv > 0 && v < nv + n + 1) {
          x = gb[(size_t)(v - nv) * P.Npad + iv];
        }
      }
      pf[m] = x;
    }
  }

  __device__ __forceinline__ void commit(int iv, const double (&pf)[NROUND], Slot& s) const {
    const int I = P.I;
    const int npt = (iv == 0) ? 1 : I + 1;
    const int nv = RECP * npt;
#pragma unroll
    for (int m = 0; m < NROUND; m++) {
      const int v = lane + 64 * m;
      if (v < nv) {
        const int k = v / npt, jj = v - k * npt;
        s.pts[(iv == 0) ? I : jj][k] = pf[m];
      } else if (v < nv + n + 1) {
        s.gp[v - nv] = pf[m];
      }
    }
  }

  // si = slot of interval i (its unary point is state i), sn = slot of interval i+1.
  // zi[k] = z_i[rho_k] (state value of this lane's rows).  Outputs the two tiles.
  __device__ __forceinline__ void build(int i, const Slot& si, const Slot& sn, const double (&zi)[4],
                                        Tile& Dt, Tile& Wt) const {
    const int I = P.I, N = P.N;
    const bool has_prev = i > 0, has_next = i < N;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double d = 0.0, h = 0.0;
      if (valid[k]) {
        const int ar = a_row[k], ac = a_col, t = tri[k];
        d = (has_prev ? KB[k] : 0.0) + (has_next ? KA[k] : 0.0);
        h = has_next ? KO[k] : 0.0;
        if (!ar && !ac) d += si.pts[I][t];  // unary obstacle factor at state i
        for (int jj = 0; jj < I; jj++) {
          const GpCoef cf = P.coef[jj];
          if (has_prev) {
            const double w2r = ar ? cf.p12 : cf.p11, w2c = ac ? cf.p12 : cf.p11;
            d = fma(w2r * w2c, si.pts[jj][t], d);
          }
          if (has_next) {
            const double w1r = ar ? cf.l12 : cf.l11, w1c = ac ? cf.l12 : cf.l11;
            const double w2c = ac ? cf.p12 : cf.p11;
            const double Gn = sn.pts[jj][t];
            d = fma(w1r * w1c, Gn, d);
            h = fma(w1r * w2c, Gn, h);
          }
        }
      }
      Dt.r[k] = d;
      Wt.r[k] = h;
    }
    // diagonal terms and the gradient column
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int rho = g + 4 * k;
      if (rho >= n) continue;
      const int ar = a_row[k], kr = k_row[k];
      const bool on_diag = (c == rho), on_rhs = (c == RHSCOL);
      if (!on_diag && !on_rhs) continue;
      double dd = 0.0, gg = 0.0;
      const double z = zi[k];
      if (i == 0 || i == N) {
        const double* tg = (i == 0) ? (ar ? pb.start_vel : pb.start_conf) : (ar ? pb.end_vel : pb.end_conf);
        const double w = ar ? P.vel_prior_w : P.conf_prior_w;
        dd += w;
        gg += w * (z - tg[(size_t)b * D + kr]);
      }
      double Hh;
      if (!ar && P.flag_pos_limit) {
        const double e = hinge_limit(z, P.pos_lo[kr], P.pos_hi[kr], P.pos_th[kr], Hh);
        dd += P.pos_w[kr] * Hh * Hh;
        gg += P.pos_w[kr] * Hh * e;
      }
      if (ar && P.flag_vel_limit) {
        const double e = hinge_limit(z, -P.vel_lim[kr], P.vel_lim[kr], P.vel_th[kr], Hh);
        dd += P.vel_w[kr] * Hh * Hh;
        gg += P.vel_w[kr] * Hh * e;
      }
      if (ar && kr == 1 && P.vdyn_w > 0.0) {
        dd += P.vdyn_w;
        gg += P.vdyn_w * z;
      }
      if (on_diag) Dt.r[k] += dd;
      if (on_rhs) {
        // obstacle gradients
        if (!ar) gg += si.pts[I][NG + kr];
        for (int jj = 0; jj < I; jj++) {
          const GpCoef cf = P.coef[jj];
          if (has_prev) gg = fma(ar ? cf.p12 : cf.p11, si.pts[jj][NG + kr], gg);
          if (has_next) gg = fma(ar ? cf.l12 : cf.l11, sn.pts[jj][NG + kr], gg);
        }
        // GP prior gradient: + Phi^T u_{i+1} - u_i
        if (has_next) gg += ar ? (P.delta_t * sn.gp[kr] + sn.gp[D + kr]) : sn.gp[kr];
        if (has_prev) gg -= si.gp[rho];
        Wt.r[k] = -gg;
      }
    }
  }
};

}  // namespace g2
