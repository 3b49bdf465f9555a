// plan_kernels.hip -- the fused hot path:
//
//   k_linearize : one lane per evaluation point (support state or GP-interpolated sub-step) of
//                 every trajectory.  interpolate -> FK -> sphere centres -> packed-cell SDF
//                 lookup -> hinge -> per-point  G = J^T J / sigma^2 (DxD packed), g = J^T r /
//                 sigma^2, e = r^T r / sigma^2, plus the GP-prior residual of each interval.
//                 (ObstacleSDFFactor / ObstacleSDFFactorGP / GaussianProcessPriorLinear
//                 evaluateError + NoiseModelFactor::linearize + WhitenSystem.)
//   k_gn_step   : one wavefront per trajectory.  Reduces the graph error, applies the
//                 gpmp2::optimize / gtsam::checkConvergence control flow, assembles the block-
//                 tridiagonal normal equations on the fly with the Kronecker weights of
//                 SURVEY.md appendix A.6, factorises them with a register-resident block
//                 Cholesky (16x16 fp64 tiles in the v_mfma_f64_16x16x4 accumulator layout, Schur
//                 updates on MFMA), back-substitutes and retracts.
//                 (GaussianFactorGraph::optimize + GaussNewtonOptimizer::iterate +
//                 planner/BatchTrajOptimizer.cpp:273-307.)
#include "device_math.h"
#include "dispatch.h"
#include "plan.h"

#include <type_traits>

namespace g2 {

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// compile-time loop
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// =============================================================================== linearize
template <int KIND, int AD, int SDIM>
__global__ __launch_bounds__(64) void k_linearize(const RobotDev* __restrict__ Rg, SdfDev sdf,
                                                   const PlanParams* __restrict__ pp,
                                                   const double* __restrict__ traj,
                                                   double* __restrict__ rec, double* __restrict__ gpu,
                                                   const int* __restrict__ active) {
  using K = Kin<KIND, AD>;
  constexpr int D = K::DOF, n = 2 * D, NG = D * (D + 1) / 2;
  const PlanParams& P = *pp;
  const int nchunk = P.Ppad / 64;
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  if (active && !active[b]) return;
  __shared__ RobotDev R;
  stage_robot(&R, Rg);
  const int p = chunk * 64 + threadIdx.x;
  if (p >= P.P) return;
  const int N = P.N, I = P.I;
  int i = 0, j = I;
  if (p > 0) {
    const int t = p - 1;
    i = 1 + t / (I + 1);
    j = t - (i - 1) * (I + 1);
  }
  const bool unary = (j == I);
  const double* z1 = traj + ((size_t)b * (N + 1) + i) * n;          // state i
  const double* z0 = (i > 0) ? z1 - n : z1;                          // state i-1 (only used if i > 0)
  double x0[D], v0[D], x1[D], v1[D], q[D];
#pragma unroll
  for (int k = 0; k < D; k++) {
    x1[k] = z1[k];
    v1[k] = z1[D + k];
    x0[k] = (i > 0) ? z0[k] : 0.0;
    v0[k] = (i > 0) ? z0[D + k] : 0.0;
  }
  if (unary) {
#pragma unroll
    for (int k = 0; k < D; k++) q[k] = x1[k];
  } else {
    const GpCoef c = P.coef[j];
#pragma unroll
    for (int k = 0; k < D; k++) q[k] = c.l11 * x0[k] + c.l12 * v0[k] + c.p11 * x1[k] + c.p12 * v1[k];
  }

  double G[NG], gv[D], e = 0.0;
#pragma unroll
  for (int k = 0; k < NG; k++) G[k] = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) gv[k] = 0.0;

  if (!(P.obs_skip_first && p == 0)) {
    const double eps = P.eps;
    K::for_each_sphere(R, q, [&](int s, const double (&pt)[3], const double (&Jc)[D][3], int) {
      double hx, hy, hz;
      const double r = hinge_obstacle<SDIM>(sdf, pt[0], pt[1], pt[2], R.sph_r[s] + eps, hx, hy, hz);
      if (hx == 0.0 && hy == 0.0 && hz == 0.0 && r == 0.0) return;  // inactive hinge: zero row
      double Jr[D];
#pragma unroll
      for (int k = 0; k < D; k++)
        Jr[k] = hx * Jc[k][0] + hy * Jc[k][1] + (SDIM == 3 ? hz * Jc[k][2] : 0.0);
      e += r * r;
      int t = 0;
#pragma unroll
      for (int k = 0; k < D; k++) {
        gv[k] += Jr[k] * r;
#pragma unroll
        for (int k2 = k; k2 < D; k2++) G[t++] += Jr[k] * Jr[k2];
      }
    });
  }
  const double w = P.obs_w;
  double* rb = rec + (size_t)b * P.REC * P.Ppad + p;
#pragma unroll
  for (int k = 0; k < NG; k++) rb[(size_t)k * P.Ppad] = G[k] * w;
#pragma unroll
  for (int k = 0; k < D; k++) rb[(size_t)(NG + k) * P.Ppad] = gv[k] * w;
  rb[(size_t)(NG + D) * P.Ppad] = e * w;

  // GaussianProcessPriorLinear of the interval ending at state i: r = Phi z_{i-1} - z_i,
  // u = Q^-1 r (Q^-1 = B(dt) (x) Qc^-1), energy r^T u.   gp/GaussianProcessPriorLinear.h:57-83
  if (unary && i > 0) {
    double rx[D], rv[D], sx[D], sv[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
      rx[k] = x0[k] + P.delta_t * v0[k] - x1[k];
      rv[k] = v0[k] - v1[k];
    }
#pragma unroll
    for (int k = 0; k < D; k++) {
      double ax = 0, av = 0;
#pragma unroll
      for (int m = 0; m < D; m++) {
        ax += P.Qc_inv[k * D + m] * rx[m];
        av += P.Qc_inv[k * D + m] * rv[m];
      }
      sx[k] = ax;
      sv[k] = av;
    }
    double* gb = gpu + (size_t)b * (n + 1) * P.Npad + i;
    double en = 0.0;
#pragma unroll
    for (int k = 0; k < D; k++) {
      const double ux = P.Winv[0] * sx[k] + P.Winv[1] * sv[k];
      const double uv = P.Winv[2] * sx[k] + P.Winv[3] * sv[k];
      gb[(size_t)k * P.Npad] = ux;
      gb[(size_t)(D + k) * P.Npad] = uv;
      en += rx[k] * ux + rv[k] * uv;
    }
    gb[(size_t)n * P.Npad] = en;
  }
}

int launch_linearize(const RobotDev& h, const RobotDev* robot, const SdfDev& sdf, const PlanParams& hp,
                     const PlanBuffers& pb, const double* traj, double* rec, double* gpu,
                     const int* active, hipStream_t st) {
  const dim3 grid(hp.B * (hp.Ppad / 64)), block(64);
  if (sdf.dim == 3) {
    G2_DISPATCH_ROBOT(h.kind, h.arm_dof, (k_linearize<KIND_, AD_, 3><<<grid, block, 0, st>>>(robot, sdf, pb.params, traj, rec, gpu, active)));
  } else {
    G2_DISPATCH_ROBOT(h.kind, h.arm_dof, (k_linearize<KIND_, AD_, 2><<<grid, block, 0, st>>>(robot, sdf, pb.params, traj, rec, gpu, active)));
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== error terms
// prior + limit + vehicle-dynamics error of one trajectory (0.5 * whitened squared residuals),
// wave-reduced.  PriorFactor (planner/BatchTrajOptimizer-inl.h:41-48), JointLimitFactorVector,
// VelocityLimitFactorVector (:50-59), VehicleDynamicsFactor (dynamics/VehicleDynamics.h:19-27).
__device__ __forceinline__ double misc_error(const PlanParams& P, const PlanBuffers& pb, int b,
                                             const double* __restrict__ tr, int lane) {
  const int D = P.D, n = P.n, N = P.N;
  double acc = 0.0;
  for (int idx = lane; idx < (N + 1) * n; idx += 64) {
    const int i = idx / n, rho = idx - i * n;
    const int a = rho >= D, k = rho - a * D;
    const double z = tr[idx];
    if (i == 0 || i == N) {
      const double* tg = (i == 0) ? (a ? pb.start_vel : pb.start_conf) : (a ? pb.end_vel : pb.end_conf);
      const double d = z - tg[(size_t)b * D + k];
      acc += (a ? P.vel_prior_w : P.conf_prior_w) * d * d;
    }
    double H;
    if (!a && P.flag_pos_limit) {
      const double e = hinge_limit(z, P.pos_lo[k], P.pos_hi[k], P.pos_th[k], H);
      acc += P.pos_w[k] * e * e;
    }
    if (a && P.flag_vel_limit) {
      const double e = hinge_limit(z, -P.vel_lim[k], P.vel_lim[k], P.vel_th[k], H);
      acc += P.vel_w[k] * e * e;
    }
    if (a && k == 1 && P.vdyn_w > 0.0) acc += P.vdyn_w * z * z;
  }
  return wave_sum(acc);
}

// total graph error of trajectory b from its point records: 0.5 * (sum e_p + sum gp energy + misc)
__device__ __forceinline__ double total_error(const PlanParams& P, const PlanBuffers& pb, int b,
                                              const double* __restrict__ tr,
                                              const double* __restrict__ rec,
                                              const double* __restrict__ gpu, int lane) {
  const double* eb = rec + ((size_t)b * P.REC + (P.NG + P.D)) * P.Ppad;
  double acc = 0.0;
  for (int p = lane; p < P.P; p += 64) acc += eb[p];
  const double* gb = gpu + ((size_t)b * (P.n + 1) + P.n) * P.Npad;
  for (int i = 1 + lane; i <= P.N; i += 64) acc += gb[i];
  return 0.5 * (wave_sum(acc) + misc_error(P, pb, b, tr, lane));
}

__global__ __launch_bounds__(64) void k_error_reduce(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                      const double* __restrict__ traj,
                                                      const double* __restrict__ rec,
                                                      const double* __restrict__ gpu,
                                                      double* __restrict__ err) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x, lane = threadIdx.x;
  const double e = total_error(P, pb, b, traj + (size_t)b * (P.N + 1) * P.n, rec, gpu, lane);
  if (lane == 0) err[b] = e;
}

int launch_error_reduce(const PlanParams& hp, const PlanBuffers& pb, const double* traj, const double* rec,
                        const double* gpu, double* err, hipStream_t st) {
  k_error_reduce<<<dim3(hp.B), dim3(64), 0, st>>>(pb.params, pb, traj, rec, gpu, err);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== 16x16 tiles
// A tile is a 16x16 fp64 matrix spread over one wavefront in the accumulator layout of
// v_mfma_f64_16x16x4_f64: lane l holds column c = l & 15 and rows rho = (l >> 4) + 4 k in r[k].
struct Tile {
  double r[4];
};

// T = A^T B.  With both operands in the tile layout, k-chunk `k` of the MFMA takes register k of
// each operand (the chunk's internal k index l>>4 then addresses row (l>>4)+4k of both), so no
// lane movement is needed.
__device__ __forceinline__ Tile tile_atb(const Tile& A, const Tile& B) {
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A.r[k], B.r[k], acc, 0, 0, 0);
  Tile T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = acc[k];
  return T;
}

constexpr int RHSCOL = 15;  // column of the coupling tile that carries the right-hand side

// Eliminate the n leading pivots of the SPD tile S while applying the same row operations to
// the coupling tile W (= [H | b]) and to V (initialised to identity by the caller).  On return
//   W <- R^-T [H | b],  V <- R^-T   (R = upper Cholesky factor of S),  S is destroyed.
// Returns false when a pivot is not positive (gtsam::IndeterminantLinearSystemException).
template <int n>
__device__ __forceinline__ bool tile_eliminate(Tile& S, Tile& W, Tile& V, int lane) {
  const int c = lane & 15, g = lane >> 4;
  double piv_of_row[4] = {1.0, 1.0, 1.0, 1.0};
  bool ok = true;
  static_for<0, n>([&](auto jc) {
    constexpr int j = decltype(jc)::value, gj = j & 3, rj = j >> 2;
    const int src = gj * 16 + c;
    const double rowS = __shfl(S.r[rj], src, 64);
    const double rowW = __shfl(W.r[rj], src, 64);
    const double rowV = __shfl(V.r[rj], src, 64);
    const double piv = readlane_d(S.r[rj], gj * 16 + j);
    ok = ok && (piv > 0.0);
    const double inv = fast_rcp(piv);
    if (g == gj) piv_of_row[rj] = piv;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double m = __shfl(S.r[k], g * 16 + j, 64);  // S[rho][j], rho = g + 4k
      const int rho = g + 4 * k;
      if (rho > j) {
        const double f = m * inv;
        S.r[k] = fma(-f, rowS, S.r[k]);
        W.r[k] = fma(-f, rowW, W.r[k]);
        V.r[k] = fma(-f, rowV, V.r[k]);
      }
    }
  });
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double s = 1.0 / sqrt(piv_of_row[k]);
    W.r[k] *= s;
    V.r[k] *= s;
  }
  return ok;
}

// x_i = V^T (y - W x_next) with y in column RHSCOL of W.  xn = x_next[c] (replicated over the four
// lane groups, 0 for c >= n).  Returns x_i[c] in the same replicated form.
template <int n>
__device__ __forceinline__ double tile_backsolve(const Tile& W, const Tile& V, double xn, int lane) {
  const int c = lane & 15;
  const double coef = (c == RHSCOL) ? -1.0 : ((c < n) ? xn : 0.0);
  double t[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double v = W.r[k] * coef;  // sum over c of W[rho][c] x[c] - y[rho]
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
    t[k] = -v;  // t[rho] = y - W x, rho = g + 4k, same in all 16 lanes of the group
  }
  double x = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) x = fma(V.r[k], t[k], x);
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

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
  double KA[4], KB[4], KO[4];

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

// =============================================================================== chain solve
// Forward elimination + back substitution of one trajectory's block-tridiagonal system.
// `next_block(i, Dt, Wt)` fills the tiles of block i.  Writes delta [nblk][n].
template <int n, class BlockSrc>
__device__ __forceinline__ bool chain_solve(int nblk, BlockSrc&& next_block, double* __restrict__ fac,
                                            double* __restrict__ delta, int lane) {
  const int c = lane & 15, g = lane >> 4;
  Tile Wprev;
  bool ok = true;
  for (int i = 0; i < nblk; i++) {
    Tile S, W, V;
    next_block(i, S, W);
    if (i > 0) {
      const Tile T = tile_atb(Wprev, Wprev);  // [W^T W , W^T y]
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool row_ok = (g + 4 * k) < n;  // padding rows must stay exactly zero
        if (row_ok && c < n) S.r[k] -= T.r[k];
        if (row_ok && c == RHSCOL) W.r[k] -= T.r[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) V.r[k] = (g + 4 * k == c) ? 1.0 : 0.0;
    ok = tile_eliminate<n>(S, W, V, lane) && ok;
    double* f = fac + (size_t)i * 512;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      f[k * 64 + lane] = V.r[k];
      f[256 + k * 64 + lane] = W.r[k];
    }
    Wprev = W;
  }
  double xn = 0.0;
  for (int i = nblk - 1; i >= 0; i--) {
    Tile W, V;
    const double* f = fac + (size_t)i * 512;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      V.r[k] = f[k * 64 + lane];
      W.r[k] = f[256 + k * 64 + lane];
    }
    if (i == nblk - 1) {
      // no coupling beyond the last block: zero the H part, keep the rhs column
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (c != RHSCOL) W.r[k] = 0.0;
    }
    xn = tile_backsolve<n>(W, V, xn, lane);
    if (g == 0 && c < n) delta[(size_t)i * n + c] = xn;
  }
  return ok;
}

// =============================================================================== GN step
__device__ __forceinline__ bool check_convergence(double rel, double abs_, double err_tol, double cur,
                                                  double nw) {
  if (nw <= err_tol) return true;
  const double abs_dec = cur - nw;
  const double rel_dec = abs_dec / cur;
  return (rel != 0.0 && rel_dec <= rel) || (abs_dec <= abs_);
}

template <int D>
__global__ __launch_bounds__(64) void k_gn_step(const PlanParams* __restrict__ pp, PlanBuffers pb, int pass) {
  constexpr int n = 2 * D;
  using Asm = Assembler<D>;
  const PlanParams& P = *pp;
  const int b = blockIdx.x, lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  if (!pb.active[b]) return;
  const int N = P.N;
  const size_t tsz = (size_t)(N + 1) * n;
  double* cur = pb.cur + b * tsz;
  double* last = pb.last + b * tsz;
  double* result = pb.result + b * tsz;

  // ---- graph error at `cur` and the gpmp2::optimize control flow
  const double new_err = total_error(P, pb, b, cur, pb.rec, pb.gpu, lane);
  int decision = 0;  // 0 iterate, 1 stop(result = cur), 2 stop(result = last)
  if (lane == 0) {
    const int it = pb.iters[b];
    double* tr = pb.trace + (size_t)b * (P.max_iter + 1);
    if (it <= P.max_iter) tr[it] = new_err;
    if (pass == 0) {
      pb.prev_err[b] = new_err;
      if (P.fixed_iters > 0) decision = 0;
      else if (new_err <= P.err_tol) { decision = 1; pb.status[b] = GPMP2MI_TRAJ_ALREADY_OPTIMAL; }
      else if (P.max_iter <= 0) { decision = 1; pb.status[b] = GPMP2MI_TRAJ_MAX_ITER; }
    } else if (P.fixed_iters > 0) {
      if (it >= P.fixed_iters) { decision = 1; pb.status[b] = GPMP2MI_TRAJ_MAX_ITER; }
    } else {
      const double prev = pb.prev_err[b];
      const bool conv = check_convergence(P.rel_thresh, P.abs_tol, P.err_tol, prev, new_err);
      if (it < P.max_iter && !conv) {
        pb.prev_err[b] = new_err;
      } else if (new_err > prev && P.no_increase) {
        decision = 2;
        pb.status[b] = GPMP2MI_TRAJ_ROLLED_BACK;
        pb.final_err[b] = prev;
      } else {
        decision = 1;
        pb.status[b] = conv ? GPMP2MI_TRAJ_CONVERGED : GPMP2MI_TRAJ_MAX_ITER;
      }
    }
    if (decision == 1) pb.final_err[b] = new_err;
    pb.cur_err[b] = new_err;
  }
  decision = __shfl(decision, 0, 64);
  if (decision != 0) {
    const double* src = (decision == 2) ? last : cur;
    for (size_t k = lane; k < tsz; k += 64) result[k] = src[k];
    if (lane == 0) pb.active[b] = 0;
    return;
  }

  // ---- one Gauss-Newton iteration: last = cur ; solve ; cur += delta
  __shared__ typename Asm::Slot slots[2];
  Asm as(P, pb, pb.rec, pb.gpu, b, lane);
  double pf[Asm::NROUND];
  as.prefetch(0, pf);
  as.commit(0, pf, slots[0]);
  as.prefetch(1, pf);
  as.commit(1, pf, slots[1]);
  as.prefetch(2, pf);
  __syncthreads();
  double znext[4];
#pragma unroll
  for (int k = 0; k < 4; k++) znext[k] = (g + 4 * k < n) ? cur[g + 4 * k] : 0.0;

  double* fac = pb.fac + (size_t)b * (N + 1) * 512;
  double* delta = pb.delta + b * tsz;
  const bool ok = chain_solve<n>(
      N + 1,
      [&](int i, Tile& Dt, Tile& Wt) {
        double zi[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          zi[k] = znext[k];
          znext[k] = (i < N && g + 4 * k < n) ? cur[(size_t)(i + 1) * n + g + 4 * k] : 0.0;
        }
        as.build(i, slots[i & 1], slots[(i + 1) & 1], zi, Dt, Wt);
        __syncthreads();
        // interval i's slot is free now: commit interval i+2, start fetching i+3
        as.commit(i + 2, pf, slots[i & 1]);
        as.prefetch(i + 3, pf);
        __syncthreads();
      },
      fac, delta, lane);
  __syncthreads();  // make the wave's own delta stores visible to its loads below
  if (!ok) {
    for (size_t k = lane; k < tsz; k += 64) result[k] = cur[k];
    if (lane == 0) {
      pb.status[b] = GPMP2MI_TRAJ_NOT_SPD;
      pb.final_err[b] = pb.cur_err[b];
      pb.active[b] = 0;
    }
    return;
  }
  for (size_t k = lane; k < tsz; k += 64) {
    const double v = cur[k];
    last[k] = v;
    cur[k] = v + delta[k];  // Values::retract for vector-valued states
  }
  if (lane == 0) {
    pb.last_err[b] = new_err;
    pb.iters[b] += 1;
    atomicAdd(pb.n_active, 1);
  }
}

int launch_gn_step(const PlanParams& hp, const PlanBuffers& pb, int pass, hipStream_t st) {
  const dim3 grid(hp.B), block(64);
  switch (hp.D) {
#define G2_STEP_CASE(DD) \
  case DD: k_gn_step<DD><<<grid, block, 0, st>>>(pb.params, pb, pass); break;
    G2_STEP_CASE(1) G2_STEP_CASE(2) G2_STEP_CASE(3) G2_STEP_CASE(4) G2_STEP_CASE(5) G2_STEP_CASE(6) G2_STEP_CASE(7)
#undef G2_STEP_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// reset the per-trajectory optimizer state before a run: cur = init is copied by the host
__global__ void k_plan_reset(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  pb.iters[b] = 0;
  pb.status[b] = GPMP2MI_TRAJ_MAX_ITER;
  pb.active[b] = 1;
  pb.phase[b] = 0;
  pb.cur_err[b] = pb.prev_err[b] = pb.last_err[b] = pb.final_err[b] = 0.0;
  pb.lambda[b] = (P.opt_type == GPMP2MI_OPT_DOGLEG) ? P.dl_delta0 : P.lm_lambda0;
  double* tr = pb.trace + (size_t)b * (P.max_iter + 1);
  for (int k = 0; k <= P.max_iter; k++) tr[k] = __longlong_as_double(0x7ff8000000000000LL);
}

int launch_plan_reset(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  k_plan_reset<<<dim3((hp.B + 63) / 64), dim3(64), 0, st>>>(pb.params, pb);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== export H, g
template <int D>
__global__ __launch_bounds__(64) void k_export_normal_eq(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                          const double* __restrict__ traj,
                                                          double* __restrict__ Hd, double* __restrict__ Ho,
                                                          double* __restrict__ gout) {
  constexpr int n = 2 * D;
  using Asm = Assembler<D>;
  const PlanParams& P = *pp;
  const int b = blockIdx.x, lane = threadIdx.x, c = lane & 15, g = lane >> 4, N = P.N;
  const double* tr = traj + (size_t)b * (N + 1) * n;
  __shared__ typename Asm::Slot slots[2];
  Asm as(P, pb, pb.rec, pb.gpu, b, lane);
  double pf[Asm::NROUND];
  as.prefetch(0, pf);
  as.commit(0, pf, slots[0]);
  as.prefetch(1, pf);
  as.commit(1, pf, slots[1]);
  __syncthreads();
  for (int i = 0; i <= N; i++) {
    double zi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) zi[k] = (g + 4 * k < n) ? tr[(size_t)i * n + g + 4 * k] : 0.0;
    Tile Dt, Wt;
    as.build(i, slots[i & 1], slots[(i + 1) & 1], zi, Dt, Wt);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int rho = g + 4 * k;
      if (rho < n && c < n) {
        if (Hd) Hd[(((size_t)b * (N + 1) + i) * n + rho) * n + c] = Dt.r[k];
        // H_{i,i+1}[rho][c] is block (i, i+1); the ABI exports block (i+1, i) = its transpose
        if (Ho && i < N) Ho[(((size_t)b * N + i) * n + c) * n + rho] = Wt.r[k];
      }
      if (rho < n && c == RHSCOL && gout) gout[((size_t)b * (N + 1) + i) * n + rho] = -Wt.r[k];
    }
    as.prefetch(i + 2, pf);
    as.commit(i + 2, pf, slots[i & 1]);
    __syncthreads();
  }
}

int launch_export_normal_eq(const PlanParams& hp, const PlanBuffers& pb, const double* traj, double* Hd,
                            double* Ho, double* g, hipStream_t st) {
  const dim3 grid(hp.B), block(64);
  switch (hp.D) {
#define G2_EXP_CASE(DD) \
  case DD: k_export_normal_eq<DD><<<grid, block, 0, st>>>(pb.params, pb, traj, Hd, Ho, g); break;
    G2_EXP_CASE(1) G2_EXP_CASE(2) G2_EXP_CASE(3) G2_EXP_CASE(4) G2_EXP_CASE(5) G2_EXP_CASE(6) G2_EXP_CASE(7)
#undef G2_EXP_CASE
    default:
      set_error("block solver is instantiated for dof <= 7");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== generic solve
// gpmp2mi_block_tridiag_solve: dense blocks in, x out; same chain solver.
template <int n>
__global__ __launch_bounds__(64) void k_block_tridiag_solve(int nblk, const double* __restrict__ Hd,
                                                             const double* __restrict__ Ho,
                                                             const double* __restrict__ rhs,
                                                             double* __restrict__ x, int* __restrict__ okf,
                                                             double* __restrict__ scratch) {
  const int b = blockIdx.x, lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const double* D_ = Hd + (size_t)b * nblk * n * n;
  const double* O_ = Ho + (size_t)b * (nblk - 1) * n * n;
  const double* r_ = rhs + (size_t)b * nblk * n;
  const bool ok = chain_solve<n>(
      nblk,
      [&](int i, Tile& Dt, Tile& Wt) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int rho = g + 4 * k;
          double d = 0.0, h = 0.0;
          if (rho < n && c < n) {
            d = D_[((size_t)i * n + rho) * n + c];
            if (i + 1 < nblk) h = O_[((size_t)i * n + c) * n + rho];  // block (i,i+1) = (i+1,i)^T
          }
          if (rho < n && c == RHSCOL) h = r_[(size_t)i * n + rho];
          Dt.r[k] = d;
          Wt.r[k] = h;
        }
      },
      scratch + (size_t)b * nblk * 512, x + (size_t)b * nblk * n, lane);
  if (lane == 0 && okf) okf[b] = ok ? 1 : 0;
}

int launch_block_tridiag_solve(int B, int nblk, int n, const double* Hd, const double* Ho, const double* b,
                               double* x, int* ok, double* scratch, hipStream_t st) {
  const dim3 grid(B), block(64);
  switch (n) {
#define G2_SOLVE_CASE(NN) \
  case NN: k_block_tridiag_solve<NN><<<grid, block, 0, st>>>(nblk, Hd, Ho, b, x, ok, scratch); break;
    G2_SOLVE_CASE(1) G2_SOLVE_CASE(2) G2_SOLVE_CASE(3) G2_SOLVE_CASE(4) G2_SOLVE_CASE(5) G2_SOLVE_CASE(6)
    G2_SOLVE_CASE(7) G2_SOLVE_CASE(8) G2_SOLVE_CASE(9) G2_SOLVE_CASE(10) G2_SOLVE_CASE(11) G2_SOLVE_CASE(12)
    G2_SOLVE_CASE(13) G2_SOLVE_CASE(14) G2_SOLVE_CASE(15)
#undef G2_SOLVE_CASE
    default:
      set_error("block size must be 1..15");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

}  // namespace g2
