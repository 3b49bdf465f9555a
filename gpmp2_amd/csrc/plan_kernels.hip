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
#include "tiles.h"
#include "assembler.h"
#include "plan_device.h"

namespace g2 {

// =============================================================================== linearize
#ifdef G2_STAMPS
#define G2_LSTAMP(k) do { if (chunk == 1 && threadIdx.x == 0 && pb.iters[b] == G2_STAMP_ITER) pb.stamps[(size_t)b * 64 + 48 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G2_LSTAMP(k) do {} while (0)
#endif
// NSPLIT = 1: one wavefront per 64 evaluation points.  NSPLIT = 2 (fixed-base arms): a workgroup of two wavefronts per
// 64 points -- both walk the kinematic chain (replicated) but each visits only the body spheres s % 2 == its index,
// so a point's 16 serial lookup / Jacobian steps become 8; the partial records are summed through LDS (w0 += w1),
// wavefront 0 stores the record while wavefront 1 evaluates the GP prior.  Splitting over LANES cannot work (lanes with
// different sphere subsets diverge and take turns); splitting over four wavefronts needs <= 168 VGPRs for all
// workgroups to be resident and spills (measured: 34.8 us against 18.1 us for two and 22.2 us for one at 64
// trajectories).  Register budget: 2 wavefronts per SIMD (<= 256 VGPRs) for arms -- at 1 024 trajectories that alone
// takes the unsplit kernel from 86.7 to 74.2 us, the split one to 69.3 us.
template <int KIND, int AD, int AD2, int SDIM, int NSPLIT>
__global__ __launch_bounds__(64 * NSPLIT, KIND == GPMP2MI_ROBOT_ARM ? 2 : 1) void k_linearize(const RobotDev* __restrict__ Rg, SdfDev sdf,
                                                            const PlanParams* __restrict__ pp,
                                                            PlanBuffers pb, const double* __restrict__ traj,
                                                            int bufsel, const int* __restrict__ active) {
  using K = Kin<KIND, AD, AD2>;
  constexpr int D = K::DOF, n = 2 * D, NG = D * (D + 1) / 2;
  const PlanParams& P = *pp;
  const int nchunk = P.Ppad / 64;
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  if (active && !active[b]) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#ifdef G2_WGTIMES
  if (threadIdx.x == 0) pb.stamps[(size_t)blockIdx.x * 2] = wall_clock64();
#endif
  double* __restrict__ rec = rec_of(pb, pb.which[b], bufsel);
  double* __restrict__ gpu = gpu_of(pb, pb.which[b], bufsel);
  G2_LSTAMP(0);
  // robot model -> LDS: the global loads are issued first and committed after the state loads and
  // the GP interpolation below, so their latency overlaps
  __shared__ RobotDev R;
  constexpr int NT = 64 * NSPLIT, RN = sizeof(RobotDev) / 4, RPT = (RN + NT - 1) / NT;
  int rtmp[RPT];
#pragma unroll
  for (int u = 0; u < RPT; u++) {
    const int idx = threadIdx.x + NT * u;
    rtmp[u] = idx < RN ? reinterpret_cast<const int*>(Rg)[idx] : 0;
  }
  const int p_raw = chunk * 64 + lane;
  const int p = min(p_raw, P.P - 1);  // tail lanes shadow the last point until the barrier below
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
    if constexpr (K::BASE == 3) {
      // GaussianProcessInterpolatorPose2Vector: the configuration now, the pose blocks of its four Jacobians when the
      // record is stored (36 doubles that would otherwise stay live across the whole sphere loop)
      lie_interpolate<D>(c, x0, v0, x1, v1, q, nullptr);
    } else {
#pragma unroll
      for (int k = 0; k < D; k++) q[k] = c.l11 * x0[k] + c.l12 * v0[k] + c.p11 * x1[k] + c.p12 * v1[k];
    }
  }

#pragma unroll
  for (int u = 0; u < RPT; u++) {
    const int idx = threadIdx.x + NT * u;
    if (idx < RN) reinterpret_cast<int*>(&R)[idx] = rtmp[u];
  }
  __syncthreads();
  if (NSPLIT == 1 && p_raw >= P.P) return;   // (split form: tail lanes keep shadowing the last point, stores are predicated)
  G2_LSTAMP(1);
  double G[NG], gv[D], e = 0.0;
#pragma unroll
  for (int k = 0; k < NG; k++) G[k] = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) gv[k] = 0.0;

  if (!(P.obs_skip_first && p == 0)) {
    const double eps = P.eps;
    double hx, hy, hz, r;
    auto accumulate = [&](const double (&Jc)[D][3], auto nc) {
      // columns >= NC of this sphere's Jacobian are zero, and so are the columns [NB, FIRST) of the OTHER arm of a
      // two-arm robot: those entries of g and G are never touched (the arm-A x arm-B block of G stays a compile-time
      // zero and takes no registers)
      constexpr int NC = decltype(nc)::value, FIRST = decltype(nc)::first, NB = K::NB;
      auto live = [](int k) { return !(k >= NB && k < FIRST); };
      double Jr[NC];
#pragma unroll
      for (int k = 0; k < NC; k++)
        Jr[k] = hx * Jc[k][0] + hy * Jc[k][1] + (SDIM == 3 ? hz * Jc[k][2] : 0.0);
      e += r * r;
#pragma unroll
      for (int k = 0; k < NC; k++) {
        if (!live(k)) continue;
        gv[k] += Jr[k] * r;
#pragma unroll
        for (int k2 = k; k2 < NC; k2++)
          if (live(k2)) G[k * D - (k * (k - 1)) / 2 + (k2 - k)] += Jr[k] * Jr[k2];
      }
    };
    K::visit_spheres(
        R, q,
        [&](int s, const double (&pt)[3]) {
          if (s < 11) G2_LSTAMP(2 + s);
          r = hinge_obstacle<SDIM>(sdf, pt[0], pt[1], pt[2], R.sph_r[s] + eps, hx, hy, hz);
          // inactive hinge (or out of the field): zero residual row, nothing to accumulate --
          // and the sphere's Jacobian is never formed
          return !(hx == 0.0 && hy == 0.0 && hz == 0.0 && r == 0.0);
        },
        [&](int, const double (&)[3], const double (&Jc)[D][3], auto nc) { accumulate(Jc, nc); }, wv, NSPLIT);
  }
  if constexpr (NSPLIT > 1) {
    // partial records -> wavefront 0 through LDS: w0 += w1
    static_assert(NSPLIT == 2, "two wavefronts per point set");
    constexpr int RV = NG + D + 1;
    __shared__ double part[RV][64];
    if (wv == 1) {
#pragma unroll
      for (int k = 0; k < NG; k++) part[k][lane] = G[k];
#pragma unroll
      for (int k = 0; k < D; k++) part[NG + k][lane] = gv[k];
      part[NG + D][lane] = e;
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int k = 0; k < NG; k++) G[k] += part[k][lane];
#pragma unroll
      for (int k = 0; k < D; k++) gv[k] += part[NG + k][lane];
      e += part[NG + D][lane];
    }
  }
  const bool store_ok = (NSPLIT == 1) || (p_raw < P.P);
  G2_LSTAMP(13);
  const double w = P.obs_w;
  // point-major record: this lane's REC values are one contiguous run, stored in 16-B pieces (split form: wavefront 0
  // holds the sums; the last wavefront takes the GP prior below, so the two tails run side by side)
  if (NSPLIT == 1 || wv == 0) {
    constexpr int RECL = NG + D + 1 + (K::BASE == 3 ? 36 : 0);
    double rv[RECL + 1];
#pragma unroll
    for (int k = 0; k < NG; k++) rv[k] = G[k] * w;
#pragma unroll
    for (int k = 0; k < D; k++) rv[NG + k] = gv[k] * w;
    rv[NG + D] = e * w;
    if constexpr (K::BASE == 3) {  // pose blocks of the four interpolation Jacobians
      double Mlie[4][9];
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int t = 0; t < 9; t++) Mlie[m][t] = 0.0;
      if (!unary) {
        double a0[D], b0[D], a1[D], b1[D], qq[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
          a1[k] = z1[k];
          b1[k] = z1[D + k];
          a0[k] = z0[k];
          b0[k] = z0[D + k];
        }
        lie_interpolate<D>(P.coef[j], a0, b0, a1, b1, qq, Mlie);
      }
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int t = 0; t < 9; t++) rv[NG + D + 1 + m * 9 + t] = Mlie[m][t];
    }
    rv[RECL] = 0.0;
    double2* rb = reinterpret_cast<double2*>(rec + ((size_t)b * P.Ppad + p) * P.RECS);
    const int nst = P.RECS >> 1;   // REC <= RECL: mobile robots without interpolation store the short record
#pragma unroll
    for (int k = 0; k < (RECL + 1) / 2; k++)
      if (k < nst && store_ok) rb[k] = double2{rv[2 * k], rv[2 * k + 1]};
  }

  G2_LSTAMP(14);
  // GP prior of the interval ending at state i.  Vector spaces: GaussianProcessPriorLinear
  // (gp/GaussianProcessPriorLinear.h:57-83) r = Phi z_{i-1} - z_i.  Pose2 robots:
  // GaussianProcessPriorLie<Pose2Vector> (gp/GaussianProcessPriorLie.h:61-86)
  // r = [Log(x1^-1 x2) - v1 dt ; v2 - v1] plus the pose blocks of its Jacobians.
  // Both: u = Q^-1 r (Q^-1 = B(dt) (x) Qc^-1), energy r^T u.
  if (unary && i > 0 && store_ok && (NSPLIT == 1 || wv == NSPLIT - 1)) {
    double rx[D], rv[D], sx[D], sv[D];
    double* gb = gpu + ((size_t)b * P.Npad + i) * P.GPS;
    if constexpr (NSPLIT > 1) {   // the states were not kept in registers across the sphere loop: fetch them again
#pragma unroll
      for (int k = 0; k < D; k++) {
        x1[k] = z1[k];
        v1[k] = z1[D + k];
        x0[k] = z0[k];
        v0[k] = z0[D + k];
      }
    }
    if constexpr (K::BASE == 3) {
      const P2 p1{x0[0], x0[1], x0[2]}, p2{x1[0], x1[1], x1[2]};
      const P2 bt = pose2_between(p1, p2);
      double lg[3], Hinv[9], Hc1[9], Hlog[9], T[9], J1[9];
      pose2_logmap(bt, lg);
      pose2_adjoint(p1, Hinv);                  // Inverse: H = -Ad(p1)
      pose2_adjoint(pose2_inverse(p2), Hc1);    // Compose(a, b): H1 = Ad(b^-1)
      pose2_logmap_derivative(bt, Hlog);
      mat3_mul(Hlog, Hc1, T);
      mat3_mul(T, Hinv, J1);
#pragma unroll
      for (int k = 0; k < 9; k++) {
        gb[n + 1 + k] = -J1[k];
        gb[n + 1 + 9 + k] = Hlog[k];
      }
#pragma unroll
      for (int k = 0; k < D; k++) {
        const double r = (k < 3) ? lg[k] : (x1[k] - x0[k]);
        rx[k] = r - v0[k] * P.delta_t;
        rv[k] = v1[k] - v0[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < D; k++) {
        rx[k] = x0[k] + P.delta_t * v0[k] - x1[k];
        rv[k] = v0[k] - v1[k];
      }
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
    double en = 0.0;
#pragma unroll
    for (int k = 0; k < D; k++) {
      const double ux = P.Winv[0] * sx[k] + P.Winv[1] * sv[k];
      const double uv = P.Winv[2] * sx[k] + P.Winv[3] * sv[k];
      gb[k] = ux;
      gb[D + k] = uv;
      en += rx[k] * ux + rv[k] * uv;
    }
    gb[n] = en;
  }
  G2_LSTAMP(15);
#ifdef G2_WGTIMES
  if (threadIdx.x == 0) pb.stamps[(size_t)blockIdx.x * 2 + 1] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// k_linearize for fixed-base arms, round 3: NW wavefronts per 64 evaluation points that SHARE one walk of the
// kinematic chain instead of replicating it.
//   phase 1  every wavefront interpolates the point's configuration (replicated: 28 loads, 28 FMAs) and takes the
//            sin / cos of the joints j % NW == its index -> LDS
//   phase 2  wavefront 0 walks the chain once (AD dependent frame advances) and leaves, per link, the columns c0, c2
//            and the origin t of its frame in LDS (9 doubles per lane and link; c1 = c2 x c0 is recomputed where a
//            sphere centre needs it).  Axis and origin of joint k are c2 and t of frame k - 1 (the base frame for k = 0)
//   phase 3  wavefront w visits the spheres s % NW == w: centre from its link's frame, SDF lookup, hinge, and for the
//            active lanes the Jacobian columns z_k x (p - o_k) of the joints below the link, accumulated as before
//   phase 4  partial records summed over the wavefronts in a fixed tree order through LDS (the frames' bytes), wavefront
//            0 stores the record while the last wavefront evaluates the GP prior
// Against the two-wavefront split above (both wavefronts walk the chain and keep all joint axes in registers, 252
// VGPRs, two wavefronts per SIMD) the dependent chain of a wavefront is sin/cos of two joints + 4 spheres instead of
// 7 joints + 8 spheres, and 168 VGPRs leave room for three wavefronts per SIMD: 640 workgroups x 4 wavefronts of the
// 64-restart batch are resident at once.
// Fused finish (Gauss-Newton fast path, `dst` != nullptr): the kernel first APPLIES the step the previous pass solved.  The
// step kernel left the solution of the blocks that are multiples of 8 (pb.xg); every workgroup here back-substitutes levels
// 4, 2, 1 for the 12 - 18 states its 64 points touch -- only the blocks those states need, found with bit masks over a window
// of <= 40 blocks; what k_finish_step did chip-wide in a launch of its own (7 - 8 us) --, adds the step to the states it
// reads from `traj` (the buffer of the previous pass, which nobody writes during this kernel), keeps the new states in
// LDS for its own points and writes those whose unary point lies in its chunk to `dst`.  The two state buffers of a
// plan (cur / last) swap roles from pass to pass, so `last` is simply the buffer the step started from.
template <int AD, int SDIM, int NW>
__global__ __launch_bounds__(64 * NW, 3) void k_linearize_arm(const RobotDev* __restrict__ Rg, SdfDev sdf,
                                                               const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                               const double* __restrict__ traj, int bufsel,
                                                               const int* __restrict__ active, double* __restrict__ dst,
                                                               int pass, int trial) {
  static_assert(NW == 2 || NW == 4, "tree reduction below");
  constexpr int D = AD, n = 2 * D, NG = D * (D + 1) / 2, RV = NG + D + 1;
  constexpr int FR = 9;                                  // doubles per lane and link: c0, c2, t
  constexpr int ROWS_F = FR * AD, ROWS_SC = 2 * AD;
  constexpr int ROWS_P = (NW / 2) * RV;                  // partial records: NW / 2 buffers
  constexpr int ROWS = (ROWS_F + ROWS_SC > ROWS_P) ? ROWS_F + ROWS_SC : ROWS_P;
  const PlanParams& P = *pp;
  const int nchunk = P.Ppad / 64;
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  if (active && !active[b]) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double* __restrict__ rec = rec_of(pb, pb.which[b], bufsel);
  double* __restrict__ gpu = gpu_of(pb, pb.which[b], bufsel);
  G2_LSTAMP(0);
  __shared__ RobotDev R;
  __shared__ double buf[ROWS][64];                       // [0, ROWS_F) frames, [ROWS_F, ROWS_F + ROWS_SC) sin / cos; later the partial records
  constexpr int NT = 64 * NW, RN = sizeof(RobotDev) / 4, RPT = (RN + NT - 1) / NT;
  int rtmp[RPT];
#pragma unroll
  for (int u = 0; u < RPT; u++) {
    const int idx = threadIdx.x + NT * u;
    rtmp[u] = idx < RN ? reinterpret_cast<const int*>(Rg)[idx] : 0;
  }
  const int p_raw = chunk * 64 + lane;
  const int p = min(p_raw, P.P - 1);  // tail lanes shadow the last point, their stores are predicated
  const int N = P.N, I = P.I;
  int i = 0, j = I;
  if (p > 0) {
    const int t = p - 1;
    i = 1 + t / (I + 1);
    j = t - (i - 1) * (I + 1);
  }
  const bool unary = (j == I);
  // ---- the states this workgroup reads: [s0, s1]; through LDS (zn), with the pending step applied in the fused form
  constexpr int FXS = 40, ZNS = 24;
  double (*fx)[16] = reinterpret_cast<double (*)[16]>(&buf[0][0]);   // step of the blocks w0 .. w0 + 39 (buf is not in use yet)
  static_assert(FXS * 16 <= ROWS * 64, "the step window lives in the frame buffer");
  __shared__ double zn[ZNS][n];       // states s0 .. s0 + 23
  const int p_lo = chunk * 64, p_hi = min(p_lo + 63, P.P - 1);
  auto state_of = [&](int pt) { return pt == 0 ? 0 : 1 + (pt - 1) / (I + 1); };
  const int s1 = state_of(p_hi), s0 = max(0, state_of(p_lo) - 1), ns = s1 - s0 + 1;   // ns <= ZNS: launch_linearize checks
  const bool apply = dst != nullptr && pb.stepped[b] == pass;
  if (apply) {
    const double* fac = pb.fac + (size_t)b * (N + 1) * 3 * TILE_DBL;
    const double* xg = pb.xg + (size_t)b * (N + 1) * 16;
    const int w0 = s0 & ~7, c = lane & 15, g = lane >> 4;
    using u64 = unsigned long long;
    auto bits = [](int lo, int hi) -> u64 { return (hi < lo) ? 0ull : ((~0ull >> (63 - (hi - lo))) << lo); };   // [lo, hi], hi <= 63
    const u64 valid = bits(0, min(N - w0, FXS - 1)), inr = bits(s0 - w0, s1 - w0);
    const u64 L1 = 0xAAAAAAAAAAAAAAAAull, L2 = 0x4444444444444444ull, L4 = 0x1010101010101010ull, L8 = 0x0101010101010101ull;
    // a block of level h needs its neighbours at distance h, which belong to higher levels
    const u64 need1 = inr & L1 & valid, nb1 = (need1 << 1) | (need1 >> 1);
    const u64 need2 = (inr | nb1) & L2 & valid, nb2 = (need2 << 2) | (need2 >> 2);
    const u64 need4 = (inr | nb1 | nb2) & L4 & valid, nb4 = (need4 << 4) | (need4 >> 4);
    const u64 need8 = (inr | nb1 | nb2 | nb4) & L8 & valid;
    // multiples of 8: solved by the step kernel
    for (u64 m = need8; m; m &= m - 1) {
      const int k = __builtin_ctzll(m);
      if (wv == ((k >> 3) & (NW - 1)) && lane < 16) fx[k][lane] = xg[(size_t)(w0 + k) * 16 + lane];
    }
    __syncthreads();
    // Task t of a level (the t-th needed block) belongs to wavefront t % NW.  (Requesting the factor tiles ahead -- all twelve
    // of a wavefront at once, or one level ahead -- was slower: 21.0 / 20.2 against 19.5 us; 2 560 wavefronts x 21 KB.)
    auto level = [&](u64 need, int h) {
      int t = 0;
      for (u64 m = need; m; m &= m - 1, t++) {
        if ((t & (NW - 1)) != wv) continue;
        const int k = __builtin_ctzll(m), jb = w0 + k;
        const double* f = fac + (size_t)jb * 3 * TILE_DBL;
        const Tile Wl = tile_load_rows<n>(f, lane), Wr = tile_load_rows<n>(f + TILE_DBL, lane);
        const Tile V = load_v<n>(f + 2 * TILE_DBL, h, N, lane);
        const double xl = (k - h >= 0) ? fx[k - h][c] : 0.0;          // (k - h < 0 cannot happen: w0 is a multiple of 8)
        const double xr = (jb + h <= N) ? fx[k + h][c] : 0.0;
        const double x = cr_backsolve<n>(Wl, Wr, V, xl, xr, lane);
        if (g == 0) fx[k][c] = (c < n) ? x : 0.0;
      }
      __syncthreads();
    };
    level(need4, 4);
    level(need2, 2);
    level(need1, 1);
  }
  // trial-step path (`trial`): dst is the trial point, the step itself goes to pb.delta, and the workgroup leaves its share
  // of g.delta, |delta|^2, |g|^2 over the states it owns in pb.spart for k_decide (what k_finish_trial did per group of 8)
  double s_gd = 0.0, s_dd = 0.0, s_gg = 0.0;
  for (int e = threadIdx.x; e < ns * n; e += 64 * NW) {
    const int t = e / n, rho = e - t * n, st = s0 + t;
    const size_t k = ((size_t)b * (N + 1) + st) * n + rho;
    double z = traj[k];
    const double x = apply ? fx[st - (s0 & ~7)][rho] : 0.0;
    z += x;                                       // Values::retract of a vector-valued state
    zn[t][rho] = z;
    const int pu = st * (I + 1);                  // the state's unary evaluation point: its owner writes the state
    if (dst != nullptr && pu >= p_lo && pu <= p_lo + 63) {
      dst[k] = z;
      if (trial && apply) {
        const double gk = pb.gvec[((size_t)b * (N + 1) + st) * 16 + rho];
        pb.delta[k] = x;
        s_gd = fma(gk, x, s_gd);
        s_dd = fma(x, x, s_dd);
        s_gg = fma(gk, gk, s_gg);
      }
    }
  }
  if (trial && apply) {   // fixed order: lanes (wave_sum), then wavefronts 0 .. NW - 1
    __shared__ double psum[NW][3];
    s_gd = wave_sum(s_gd);
    s_dd = wave_sum(s_dd);
    s_gg = wave_sum(s_gg);
    if (lane == 0) {
      psum[wv][0] = s_gd;
      psum[wv][1] = s_dd;
      psum[wv][2] = s_gg;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      double a = 0.0;
#pragma unroll
      for (int w = 0; w < NW; w++) a += psum[w][threadIdx.x];
      pb.spart[((size_t)b * nchunk + chunk) * 3 + threadIdx.x] = a;
    }
  }
  __syncthreads();
  const double* z1 = &zn[i - s0][0];                                 // state i
  const double* z0 = (i > 0) ? z1 - n : z1;                          // state i-1 (only used if i > 0)
  {
    double q[D];
    if (unary) {
#pragma unroll
      for (int k = 0; k < D; k++) q[k] = z1[k];
    } else {
      const GpCoef c = P.coef[j];
#pragma unroll
      for (int k = 0; k < D; k++) q[k] = c.l11 * z0[k] + c.l12 * z0[D + k] + c.p11 * z1[k] + c.p12 * z1[D + k];
    }
    // sin / cos of this wavefront's joints (the joint bias straight from the model in HBM: a uniform scalar load)
#pragma unroll
    for (int k = 0; k < AD; k++) {
      if (k % NW != wv) continue;
      double sn, cs;
      sincos(q[k] + Rg->bias[k], &sn, &cs);
      buf[ROWS_F + 2 * k][lane] = sn;
      buf[ROWS_F + 2 * k + 1][lane] = cs;
    }
  }
#pragma unroll
  for (int u = 0; u < RPT; u++) {
    const int idx = threadIdx.x + NT * u;
    if (idx < RN) reinterpret_cast<int*>(&R)[idx] = rtmp[u];
  }
  __syncthreads();
  G2_LSTAMP(1);
  if (wv == 0) {
    Frame F;
    frame_from_3x4(R.base, F);  // world_T_base
    static_for<0, AD>([&](auto jc) {
      constexpr int k = decltype(jc)::value;
      const double sn = buf[ROWS_F + 2 * k][lane], cs = buf[ROWS_F + 2 * k + 1][lane];
      dh_advance_sc(F, sn, cs, R.a[k], R.d[k], R.ca[k], R.sa[k]);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        buf[FR * k + t][lane] = F.c0[t];
        buf[FR * k + 3 + t][lane] = F.c2[t];
        buf[FR * k + 6 + t][lane] = F.t[t];
      }
    });
  }
  __syncthreads();
  G2_LSTAMP(2);
  double G[NG], gv[D], e = 0.0;
#pragma unroll
  for (int k = 0; k < NG; k++) G[k] = 0.0;
#pragma unroll
  for (int k = 0; k < D; k++) gv[k] = 0.0;
  if (!(P.obs_skip_first && p == 0)) {
    const double eps = P.eps;
    static_for<0, AD>([&](auto jc) {
      constexpr int L = decltype(jc)::value, NC = L + 1;     // link L: the joints 0 .. L move it
      for (int s = R.link_first[L]; s < R.link_first[L + 1]; s++) {
        if (s % NW != wv) continue;
        double c0[3], c2[3], o[3], pt[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          c0[t] = buf[FR * L + t][lane];
          c2[t] = buf[FR * L + 3 + t][lane];
          o[t] = buf[FR * L + 6 + t][lane];
        }
        const double c1[3] = {c2[1] * c0[2] - c2[2] * c0[1], c2[2] * c0[0] - c2[0] * c0[2], c2[0] * c0[1] - c2[1] * c0[0]};
        const double cx = R.sph_c[3 * s], cy = R.sph_c[3 * s + 1], cz = R.sph_c[3 * s + 2];
#pragma unroll
        for (int t = 0; t < 3; t++) pt[t] = o[t] + c0[t] * cx + c1[t] * cy + c2[t] * cz;
        double hx, hy, hz;
        const double r = hinge_obstacle<SDIM>(sdf, pt[0], pt[1], pt[2], R.sph_r[s] + eps, hx, hy, hz);
        // inactive hinge (or out of the field): zero residual row, nothing to accumulate
        if (hx == 0.0 && hy == 0.0 && hz == 0.0 && r == 0.0) continue;
        double Jr[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
          double zx, zy, zz, ox, oy, oz;
          if (k == 0) {   // joint 0 sits in the base frame: R.base rows are (c0 c1 c2 t) per coordinate
            zx = R.base[2]; zy = R.base[6]; zz = R.base[10];
            ox = R.base[3]; oy = R.base[7]; oz = R.base[11];
          } else {
            zx = buf[FR * (k - 1) + 3][lane]; zy = buf[FR * (k - 1) + 4][lane]; zz = buf[FR * (k - 1) + 5][lane];
            ox = buf[FR * (k - 1) + 6][lane]; oy = buf[FR * (k - 1) + 7][lane]; oz = buf[FR * (k - 1) + 8][lane];
          }
          const double rx = pt[0] - ox, ry = pt[1] - oy, rz = pt[2] - oz;
          const double Jx = zy * rz - zz * ry, Jy = zz * rx - zx * rz, Jz = zx * ry - zy * rx;   // z_k x (p - o_k)
          Jr[k] = hx * Jx + hy * Jy + (SDIM == 3 ? hz * Jz : 0.0);
        }
        e += r * r;
#pragma unroll
        for (int k = 0; k < NC; k++) {
          gv[k] += Jr[k] * r;
#pragma unroll
          for (int k2 = k; k2 < NC; k2++) G[k * D - (k * (k - 1)) / 2 + (k2 - k)] += Jr[k] * Jr[k2];
        }
      }
    });
  }
  G2_LSTAMP(3);
  // partial records -> wavefront 0, fixed order: (w0 + w2) + (w1 + w3)
  __syncthreads();   // every wavefront is done with the frames: their bytes now take the partial records
  auto put = [&](int slot) {
#pragma unroll
    for (int k = 0; k < NG; k++) buf[slot * RV + k][lane] = G[k];
#pragma unroll
    for (int k = 0; k < D; k++) buf[slot * RV + NG + k][lane] = gv[k];
    buf[slot * RV + NG + D][lane] = e;
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int k = 0; k < NG; k++) G[k] += buf[slot * RV + k][lane];
#pragma unroll
    for (int k = 0; k < D; k++) gv[k] += buf[slot * RV + NG + k][lane];
    e += buf[slot * RV + NG + D][lane];
  };
  if constexpr (NW == 4) {
    if (wv >= 2) put(wv - 2);
    __syncthreads();
    if (wv < 2) add(wv);
    if (wv == 1) put(1);      // (slot 1 was read by wavefront 1 alone)
    __syncthreads();
    if (wv == 0) add(1);
  } else {
    if (wv == 1) put(0);
    __syncthreads();
    if (wv == 0) add(0);
  }
  const bool store_ok = p_raw < P.P;
  G2_LSTAMP(13);
  if (wv == 0) {
    // point-major record: this lane's REC values are one contiguous run, stored in 16-B pieces
    const double w = P.obs_w;
    constexpr int RECL = NG + D + 1;
    double rv[RECL + 1];
#pragma unroll
    for (int k = 0; k < NG; k++) rv[k] = G[k] * w;
#pragma unroll
    for (int k = 0; k < D; k++) rv[NG + k] = gv[k] * w;
    rv[NG + D] = e * w;
    rv[RECL] = 0.0;
    double2* rb = reinterpret_cast<double2*>(rec + ((size_t)b * P.Ppad + p) * P.RECS);
#pragma unroll
    for (int k = 0; k < (RECL + 1) / 2; k++)
      if (store_ok) rb[k] = double2{rv[2 * k], rv[2 * k + 1]};
  }
  G2_LSTAMP(14);
  // GP prior of the interval ending at state i: GaussianProcessPriorLinear (gp/GaussianProcessPriorLinear.h:57-83),
  // r = Phi z_{i-1} - z_i, u = Q^-1 r (Q^-1 = B(dt) (x) Qc^-1), energy r^T u -- as in k_linearize above
  if (unary && i > 0 && store_ok && wv == NW - 1) {
    double rx[D], rv[D], sx[D], sv[D];
    double* gb = gpu + ((size_t)b * P.Npad + i) * P.GPS;
#pragma unroll
    for (int k = 0; k < D; k++) {
      rx[k] = z0[k] + P.delta_t * z0[D + k] - z1[k];
      rv[k] = z0[D + k] - z1[D + k];
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
    double en = 0.0;
#pragma unroll
    for (int k = 0; k < D; k++) {
      const double ux = P.Winv[0] * sx[k] + P.Winv[1] * sv[k];
      const double uv = P.Winv[2] * sx[k] + P.Winv[3] * sv[k];
      gb[k] = ux;
      gb[D + k] = uv;
      en += rx[k] * ux + rv[k] * uv;
    }
    gb[n] = en;
  }
  G2_LSTAMP(15);
}

// dst / pass: fused finish of the Gauss-Newton fast path (k_linearize_arm; only with hp.fuse_finish): apply the step of
// pass - 1 to the states in `traj` and write the new states to `dst`; dst = nullptr: linearize `traj` as it is
int launch_linearize(const RobotDev& h, const RobotDev* robot, const SdfDev& sdf, const PlanParams& hp,
                     const PlanBuffers& pb, const double* traj, int bufsel, const int* active,
                     hipStream_t st, double* dst, int pass, bool trial) {
  if (dst != nullptr && !(hp.fuse_finish && hp.lin_split == 4 && h.kind == GPMP2MI_ROBOT_ARM)) {
    set_error("fused finish asked of a plan that was not set up for it");
    return GPMP2MI_ERR_INVALID;
  }
  if (hp.lin_split == 4 && 63 / (hp.I + 1) + 3 > 24) {   // states per chunk kept in LDS by k_linearize_arm (ZNS)
    set_error("the four-wavefront linearization needs obs_check_inter >= 2");
    return GPMP2MI_ERR_INVALID;
  }
  // One lane per evaluation point; fixed-base arms split the spheres of a point over the two wavefronts of a
  // workgroup (NSPLIT = 2, see the kernel).  A variant that kept 4-8 SDF cells in flight per lane was no faster
  // (DESIGN.md section 4).
  const dim3 grid(hp.B * (hp.Ppad / 64));
  if (hp.lin_split == 4 && h.kind == GPMP2MI_ROBOT_ARM) {
    const dim3 block(256);
    if (sdf.dim == 3) {
      G2_DISPATCH_ROBOT_ARM_ONLY(h.arm_dof, (k_linearize_arm<AD_, 3, 4><<<grid, block, 0, st>>>(robot, sdf, pb.params, pb, traj, bufsel, active, dst, pass, trial ? 1 : 0)));
    } else {
      G2_DISPATCH_ROBOT_ARM_ONLY(h.arm_dof, (k_linearize_arm<AD_, 2, 4><<<grid, block, 0, st>>>(robot, sdf, pb.params, pb, traj, bufsel, active, dst, pass, trial ? 1 : 0)));
    }
  } else if (hp.lin_split == 2 && h.kind == GPMP2MI_ROBOT_ARM) {
    const dim3 block(128);
    if (sdf.dim == 3) {
      G2_DISPATCH_ROBOT_ARM_ONLY(h.arm_dof, (k_linearize<GPMP2MI_ROBOT_ARM, AD_, 0, 3, 2><<<grid, block, 0, st>>>(robot, sdf, pb.params, pb, traj, bufsel, active)));
    } else {
      G2_DISPATCH_ROBOT_ARM_ONLY(h.arm_dof, (k_linearize<GPMP2MI_ROBOT_ARM, AD_, 0, 2, 2><<<grid, block, 0, st>>>(robot, sdf, pb.params, pb, traj, bufsel, active)));
    }
  } else {
    const dim3 block(64);
    if (sdf.dim == 3) {
      G2_DISPATCH_ROBOT_H(h, (k_linearize<KIND_, AD_, AD2_, 3, 1><<<grid, block, 0, st>>>(robot, sdf, pb.params, pb, traj, bufsel, active)));
    } else {
      G2_DISPATCH_ROBOT_H(h, (k_linearize<KIND_, AD_, AD2_, 2, 1><<<grid, block, 0, st>>>(robot, sdf, pb.params, pb, traj, bufsel, active)));
    }
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== extra factors
// One lane per (trajectory, support state): adds the whitened normal-equation terms of the workspace priors /
// goal factor / self-collision rows of that state to the record of its unary evaluation point
// (p = i (I + 1)): G += H^T H / sigma^2, g += H^T r / sigma^2, e += r^T r / sigma^2.  The residuals and Jacobians come
// from the factor kernels (k_fk + k_workspace_prior, k_sphere_centers + k_self_collision) run on the states.
__global__ __launch_bounds__(64) void k_extra_accumulate(const PlanParams* __restrict__ pp, PlanBuffers pb, PlanExtras ex,
                                                          int bufsel, const int* __restrict__ active) {
  const PlanParams& P = *pp;
  const int N = P.N, D = P.D, M = P.B * (N + 1);
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int b = m / (N + 1), i = m - b * (N + 1);
  if (active && !active[b]) return;
  double* rec = rec_of(pb, pb.which[b], bufsel) + ((size_t)b * P.Ppad + (size_t)i * (P.I + 1)) * P.RECS;
  auto add_rows = [&](const double* err, const double* H, int rows, const double* w, double wu) {
    // rows x D Jacobian H (row-major), residual err; weight w[r] per row (or the uniform wu when w == nullptr)
    double e = 0.0;
    for (int r = 0; r < rows; r++) e += (w ? w[r] : wu) * err[r] * err[r];
    rec[P.NG + D] += e;
    for (int k = 0; k < D; k++) {
      double g = 0.0;
      for (int r = 0; r < rows; r++) g += (w ? w[r] : wu) * H[r * D + k] * err[r];
      rec[P.NG + k] += g;
      for (int k2 = k; k2 < D; k2++) {
        double a = 0.0;
        for (int r = 0; r < rows; r++) a += (w ? w[r] : wu) * H[r * D + k] * H[r * D + k2];
        rec[k * D - (k * (k - 1)) / 2 + (k2 - k)] += a;
      }
    }
  };
  for (int f = 0; f < ex.n_ws; f++) {
    if (i < ex.ws_first[f] || i > ex.ws_last[f]) continue;
    const int rows = ex.ws_mode[f] == GPMP2MI_WORKSPACE_POSE ? 6 : 3;
    // (k_workspace_prior packs `rows` per state; every factor owns a slice sized for 6)
    add_rows(ex.ws_err + (size_t)f * M * 6 + (size_t)m * rows, ex.ws_H + ((size_t)f * M * 6 + (size_t)m * rows) * D, rows,
             nullptr, ex.ws_w[f]);
  }
  if (ex.n_sc > 0 && i >= ex.sc_first && i <= ex.sc_last)
    add_rows(ex.sc_err + (size_t)m * ex.n_sc, ex.sc_H + (size_t)m * ex.n_sc * D, ex.n_sc, ex.sc_w, 0.0);
}

int launch_extra_accumulate(const PlanParams& hp, const PlanBuffers& pb, const PlanExtras& ex, int L, int S, int bufsel,
                            const int* active, hipStream_t st) {
  const int M = hp.B * (hp.N + 1);
  k_extra_accumulate<<<dim3((M + 63) / 64), dim3(64), 0, st>>>(pb.params, pb, ex, bufsel, active);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

__global__ __launch_bounds__(64) void k_error_reduce(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                      const double* __restrict__ traj, int bufsel,
                                                      double* __restrict__ err) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x, lane = threadIdx.x;
  const double e = total_error(P, pb, b, traj + (size_t)b * (P.N + 1) * P.n, rec_of(pb, pb.which[b], bufsel),
                               gpu_of(pb, pb.which[b], bufsel), lane);
  if (lane == 0) err[b] = e;
}

int launch_error_reduce(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                        double* err, hipStream_t st) {
  k_error_reduce<<<dim3(hp.B), dim3(64), 0, st>>>(pb.params, pb, traj, bufsel, err);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}


// The closing pass of a run with a fixed number of iterations only evaluates the error of the final values (every
// trajectory stops in the step kernel before it factorises anything): instead of k_assemble, which would build and
// eliminate all blocks for nothing, this kernel leaves the graph error of each active trajectory where the step kernel
// looks for it -- the whole sum in the share of block 0, zeros in the others.
__global__ __launch_bounds__(256) void k_error_parts(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                      const double* __restrict__ traj, int bufsel,
                                                      const int* __restrict__ active) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (active && !active[b]) return;
  __shared__ double red[4];
  const double part = total_error_partial(P, pb, b, traj + (size_t)b * (P.N + 1) * P.n, rec_of(pb, pb.which[b], bufsel),
                                          gpu_of(pb, pb.which[b], bufsel), tid, 256);
  const double ws = wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = ws;
  __syncthreads();
  const double e = 0.5 * (((red[0] + red[1]) + red[2]) + red[3]);   // the fixed-order sum of k_decide
  for (int i = tid; i <= P.N; i += 256) pb.epart[(size_t)b * P.Npad + i] = (i == 0) ? e : 0.0;
}

int launch_error_parts(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel, const int* active,
                       hipStream_t st) {
  k_error_parts<<<dim3(hp.B), dim3(256), 0, st>>>(pb.params, pb, traj, bufsel, active);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// gpmp2mi_plan_update switches the resident parameter block to `iterations` fixed Gauss-Newton steps and back: two
// words, written in stream order by this kernel instead of re-uploading the block from pageable host memory twice
__global__ void k_set_mode(PlanParams* pp, int opt_type, int fixed_iters) {
  pp->opt_type = opt_type;
  pp->fixed_iters = fixed_iters;
}
int launch_set_mode(const PlanBuffers& pb, int opt_type, int fixed_iters, hipStream_t st) {
  k_set_mode<<<dim3(1), dim3(1), 0, st>>>(pb.params, opt_type, fixed_iters);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// reset the optimizer state before a run and load the starting values: cur = start (no separate copy command in
// the stream); grid-stride over the flat index ranges so that no thread writes a long serial run
__global__ __launch_bounds__(256) void k_plan_reset(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                    const double* __restrict__ start) {
  const PlanParams& P = *pp;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const size_t B = P.B;
  if (start) {
    const size_t tot = B * (size_t)(P.N + 1) * P.n;
    for (size_t k = tid; k < tot; k += nth) pb.cur[k] = start[k];
  }
  for (size_t k = tid; k < (size_t)P.max_pass; k += nth) pb.n_active[k] = pb.done[k] = 0;
  for (size_t k = tid; k < B * SC_COUNT; k += nth) pb.scal[k] = 0.0;
  for (size_t k = tid; k < B * (size_t)(P.max_iter + 1); k += nth) pb.trace[k] = __longlong_as_double(0x7ff8000000000000LL);
  for (size_t b = tid; b < B; b += nth) {
    pb.iters[b] = 0;
    pb.status[b] = GPMP2MI_TRAJ_MAX_ITER;
    pb.active[b] = 1;
    pb.phase[b] = 0;
    pb.which[b] = 0;
    pb.stepped[b] = 0;
    pb.notspd[b] = 0;
    pb.cur_err[b] = pb.prev_err[b] = pb.last_err[b] = pb.final_err[b] = 0.0;
    pb.lambda[b] = (P.opt_type == GPMP2MI_OPT_DOGLEG) ? P.dl_delta0 : P.lm_lambda0;
  }
}

int launch_plan_reset(const PlanParams& hp, const PlanBuffers& pb, const double* start, hipStream_t st) {
  const size_t tot = (size_t)hp.B * (hp.N + 1) * hp.n;
  const int blocks = (int)std::min<size_t>(1024, std::max<size_t>(1, (tot + 1023) / 1024));
  k_plan_reset<<<dim3(blocks), dim3(256), 0, st>>>(pb.params, pb, start);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== step control
// One wavefront per trajectory.  `init`: error of the initial values + the early exits of
// gpmp2::optimize (planner/BatchTrajOptimizer.cpp:248-268).  Otherwise: the trial point produced by
// k_solve_step has been linearized into the spare record buffer; compute its graph error and apply
//   GaussNewtonOptimizer::iterate      (always accept)
//   LevenbergMarquardtOptimizer::tryLambda  (model fidelity test, lambda *= / /= 10, give up at 1e5)
//   DoglegOptimizerImpl::Iterate(ONE_STEP_PER_ITERATION)  (gain ratio rho, trust radius update)
// followed by the do/while of gpmp2::optimize (checkConvergence, max_iter, no-increase rollback).
// GTSAM semantics restated from upstream.
__device__ __forceinline__ void decide_body(const PlanParams& P, const PlanBuffers& pb, int pass, int init) {
  // 4 wavefronts: all of them sum the graph error of the point in question (fixed-order block reduction),
  // the first thread takes the decision, all four wavefronts then move the trajectories
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const bool w0 = tid < 64;
  __shared__ int dec[2];
  __shared__ double red[4];
  if (!pb.active[b]) return;
  const int N = P.N, n = P.n;
  const size_t tsz = (size_t)(N + 1) * n;
  double* cur = pb.cur + b * tsz;
  double* last = pb.last + b * tsz;
  double* trial = pb.trial + b * tsz;
  double* result = pb.result + b * tsz;
  double* sc = pb.scal + (size_t)b * SC_COUNT;
  double* tr = pb.trace + (size_t)b * (P.max_iter + 1);
  const int wh = pb.which[b];
  // action: 0 keep iterating, 1 finish with cur, 2 finish with last; accept: copy trial -> cur
  int action = 0, accept = 0;

  // The trajectory moves at the end (last = cur, cur = trial, result = ...) read cur and trial; requested here, they
  // arrive while the error is being summed instead of one dependent load-store pair after another behind the decision.
  constexpr int PF = 10;   // prefetched elements per thread and array (covers (N + 1) n <= 2560)
  double pf_cur[PF], pf_trial[PF];
#pragma unroll
  for (int m = 0; m < PF; m++) {
    const size_t k = tid + (size_t)m * 256;
    pf_cur[m] = (k < tsz) ? cur[k] : 0.0;
    pf_trial[m] = (!init && k < tsz) ? trial[k] : 0.0;
  }

  const bool failed = !init && pb.notspd[b] != 0;
  double err_sum = 0.0;
  if (!failed) {
    const double part = init ? total_error_partial(P, pb, b, cur, rec_of(pb, wh, 0), gpu_of(pb, wh, 0), tid, blockDim.x)
                             : total_error_partial(P, pb, b, trial, rec_of(pb, wh, 1), gpu_of(pb, wh, 1), tid, blockDim.x);
    const double ws = wave_sum(part);
    if (lane == 0) red[tid >> 6] = ws;
  }
  __syncthreads();
  if (!failed) err_sum = 0.5 * (((red[0] + red[1]) + red[2]) + red[3]);
  if (w0 && init) {
    const double err = err_sum;
    if (lane == 0) {
      pb.cur_err[b] = pb.prev_err[b] = err;
      tr[0] = err;
      if (P.fixed_iters == 0 && err <= P.err_tol) { action = 1; pb.status[b] = GPMP2MI_TRAJ_ALREADY_OPTIMAL; }
      else if (P.fixed_iters == 0 && P.max_iter <= 0) { action = 1; pb.status[b] = GPMP2MI_TRAJ_MAX_ITER; }
      if (action) pb.final_err[b] = err;
    }
  } else if (w0) {
    const double new_err = err_sum;
    // LM / GN split form: g.delta, |delta|^2, |g|^2 arrive as per-group shares of k_finish_trial (fixed-order wave sums)
    double sp_gd = 0.0, sp_dd = 0.0, sp_gg = 0.0;
    if (P.split_back && P.opt_type == GPMP2MI_OPT_LM && !failed) {
      const int groups = P.spart_groups;
      const double* sp = pb.spart + (size_t)b * groups * 3;
      for (int qq = lane; qq < groups; qq += 64) {
        sp_gd += sp[3 * qq];
        sp_dd += sp[3 * qq + 1];
        sp_gg += sp[3 * qq + 2];
      }
      sp_gd = wave_sum(sp_gd);
      sp_dd = wave_sum(sp_dd);
      sp_gg = wave_sum(sp_gg);
    }
    if (lane == 0) {
      pb.notspd[b] = 0;
      const double cur_err = pb.cur_err[b];
      bool iterate_done = false;   // GTSAM iterate() returned
      bool moved = false;          // ... with new values
      double err_after = cur_err;
      if (P.opt_type == GPMP2MI_OPT_GAUSS_NEWTON) {
        if (failed) { action = 1; pb.status[b] = GPMP2MI_TRAJ_NOT_SPD; pb.final_err[b] = cur_err; }
        else { iterate_done = moved = true; err_after = new_err; }
      } else if (P.opt_type == GPMP2MI_OPT_LM) {
        double lambda = pb.lambda[b];
        bool step_ok = false, stop = false;
        if (!failed) {
          const double old_lin = cur_err;
          if (P.split_back) {   // per-group shares of k_finish_trial, summed by wavefront 0 above
            sc[SC_GD] = sp_gd;
            sc[SC_DD] = sp_dd;
            sc[SC_GG] = sp_gg;
          }
          const double lin_change = -(0.5 * sc[SC_GD] - 0.5 * lambda * sc[SC_DD]);
          if (lin_change >= 0) {
            const double cost_change = cur_err - new_err;
            if (lin_change > 2.220446049250313e-16 * old_lin) step_ok = (cost_change / lin_change) > P.lm_min_fidelity;
            if (fabs(cost_change) < P.rel_thresh * cur_err) stop = true;
          }
        }
        if (step_ok) {
          iterate_done = moved = true;
          err_after = new_err;
          lambda = fmax(P.lm_lower, lambda / P.lm_factor);
        } else if (!stop) {
          lambda *= P.lm_factor;
          if (lambda >= P.lm_upper) iterate_done = true;  // give up: state unchanged
        } else {
          iterate_done = true;                            // relative cost change tiny: state unchanged
        }
        pb.lambda[b] = lambda;
      } else {  // Dogleg
        if (failed) { action = 1; pb.status[b] = GPMP2MI_TRAJ_NOT_SPD; pb.final_err[b] = cur_err; }
        else {
          double Delta = pb.lambda[b];
          const double f_error = cur_err, M_error = cur_err, new_M = M_error + sc[SC_Q];
          const double rho = (fabs(f_error - new_err) < 1e-15 || fabs(M_error - new_M) < 1e-15)
                                 ? 0.5 : (f_error - new_err) / (M_error - new_M);
          if (rho >= 0.75) { Delta = fmax(Delta, 3.0 * sc[SC_XNORM]); iterate_done = moved = true; err_after = new_err; }
          else if (rho >= 0.25) { iterate_done = moved = true; err_after = new_err; }
          else if (rho >= 0.0) { if (Delta > 1e-5) Delta = 0.5 * Delta; iterate_done = moved = true; err_after = new_err; }
          else if (Delta > 1e-5) { Delta *= 0.5; pb.phase[b] = 1; }           // retry, same linearization
          else { iterate_done = true; err_after = cur_err; }                  // zero step
          pb.lambda[b] = Delta;
          if (iterate_done) pb.phase[b] = 0;
        }
      }
      if (iterate_done) {
        const bool counted = moved || P.opt_type == GPMP2MI_OPT_DOGLEG;  // LM give-up does not count
        const int it = pb.iters[b] + (counted ? 1 : 0);
        pb.iters[b] = it;
        if (moved) accept = 1;
        // trace = error after every call to iterate() (an LM call that gives up repeats the value);
        // LM keeps its call counter in `phase`, which only Dogleg uses otherwise
        const int call = (P.opt_type == GPMP2MI_OPT_LM) ? ++pb.phase[b] : it;
        if (call <= P.max_iter) tr[call] = err_after;
        const double prev = pb.prev_err[b];
        if (P.fixed_iters > 0) {
          if (it >= P.fixed_iters || !counted) { action = 1; pb.status[b] = GPMP2MI_TRAJ_MAX_ITER; pb.final_err[b] = err_after; }
        } else {
          const bool conv = check_convergence(P.rel_thresh, P.abs_tol, P.err_tol, prev, err_after);
          if (it < P.max_iter && !conv) {
            pb.prev_err[b] = err_after;
          } else if (err_after > prev && P.no_increase) {
            action = 2;  // the values before this iterate
            pb.status[b] = GPMP2MI_TRAJ_ROLLED_BACK;
            pb.final_err[b] = prev;
          } else {
            action = 1;
            pb.status[b] = conv ? GPMP2MI_TRAJ_CONVERGED : GPMP2MI_TRAJ_MAX_ITER;
            pb.final_err[b] = err_after;
          }
        }
        pb.cur_err[b] = err_after;
      }
    }
  }
  if (tid == 0) {
    dec[0] = action;
    dec[1] = accept;
  }
  __syncthreads();
  action = dec[0];
  accept = dec[1];
  // (pf_cur / pf_trial hold the first PF * 256 elements; longer trajectories finish with plain loads)
  if (accept) {
    if (action == 2) {
      // rollback: the result is the pre-step `cur`; nothing else reads cur afterwards
#pragma unroll
      for (int m = 0; m < PF; m++) {
        const size_t k = tid + (size_t)m * 256;
        if (k < tsz) result[k] = pf_cur[m];
      }
      for (size_t k = tid + (size_t)PF * 256; k < tsz; k += 256) result[k] = cur[k];
    } else {
#pragma unroll
      for (int m = 0; m < PF; m++) {
        const size_t k = tid + (size_t)m * 256;
        if (k < tsz) {
          last[k] = pf_cur[m];
          cur[k] = pf_trial[m];
          if (action == 1) result[k] = pf_trial[m];
        }
      }
      for (size_t k = tid + (size_t)PF * 256; k < tsz; k += 256) {
        const double t = trial[k];
        last[k] = cur[k];
        cur[k] = t;
        if (action == 1) result[k] = t;
      }
      if (tid == 0) pb.which[b] = wh ^ 1;  // the trial linearization is now the one at cur
    }
  } else if (action == 1) {
#pragma unroll
    for (int m = 0; m < PF; m++) {
      const size_t k = tid + (size_t)m * 256;
      if (k < tsz) result[k] = pf_cur[m];
    }
    for (size_t k = tid + (size_t)PF * 256; k < tsz; k += 256) result[k] = cur[k];
  } else if (action == 2) {
    for (size_t k = tid; k < tsz; k += blockDim.x) result[k] = last[k];
  }
  if (tid == 0) {
    if (action != 0) pb.active[b] = 0;
    else atomicAdd(pb.n_active + pass, 1);
  }
}
__global__ __launch_bounds__(256) void k_decide(const PlanParams* __restrict__ pp, PlanBuffers pb, int pass, int init) {
  decide_body(*pp, pb, pass, init);
  if (threadIdx.x == 0) publish_pass_count(pb, pass);
}

int launch_decide(const PlanParams& hp, const PlanBuffers& pb, int pass, bool init, hipStream_t st) {
  k_decide<<<dim3(hp.B), dim3(256), 0, st>>>(pb.params, pb, pass, init ? 1 : 0);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// Trajectories that are still iterating when the trial-step driver has spent its pass budget: finish them with
// their current values (status MAX_ITER) so that `result` is never stale.  One workgroup per trajectory.
__global__ __launch_bounds__(256) void k_finalize_unfinished(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x;
  if (!pb.active[b]) return;
  const size_t tsz = (size_t)(P.N + 1) * P.n;
  for (size_t k = threadIdx.x; k < tsz; k += blockDim.x) pb.result[b * tsz + k] = pb.cur[b * tsz + k];
  if (threadIdx.x == 0) {
    pb.status[b] = GPMP2MI_TRAJ_MAX_ITER;
    pb.final_err[b] = pb.cur_err[b];
    pb.active[b] = 0;
  }
}
int launch_finalize_unfinished(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  k_finalize_unfinished<<<dim3(hp.B), dim3(256), 0, st>>>(pb.params, pb);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== export H, g
template <int D, bool LIE>
__global__ __launch_bounds__(64) void k_export_normal_eq(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                          const double* __restrict__ traj, int bufsel,
                                                          double* __restrict__ Hd, double* __restrict__ Ho,
                                                          double* __restrict__ gout, const int* __restrict__ active) {
  constexpr int n = 2 * D;
  using Asm = Assembler<D, LIE>;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int b = blockIdx.x / (N + 1), i = blockIdx.x - b * (N + 1);
  // optimizer use (wide path): finished trajectories and Dogleg retries keep their last system
  if (active && (!active[b] || (P.opt_type == GPMP2MI_OPT_DOGLEG && pb.phase[b] != 0))) return;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  extern __shared__ __attribute__((aligned(16))) double asm_smem[];
  Asm as(P, pb, rec_of(pb, pb.which[b], bufsel), gpu_of(pb, pb.which[b], bufsel), b, lane);
  const typename Asm::Slot slot0 = as.make_slot(asm_smem, 0), slot1 = as.make_slot(asm_smem, 1);
  as.stage2(i, slot0, slot1);
  __syncthreads();
  // blocks wider than one tile (2 dof > 15) are walked as 2x2 (3x3 for 2 dof > 31) tiles; the right-hand side
  // rides in the last column of the tile grid
  constexpr int T = (n <= 15) ? 1 : (n <= 31) ? 2 : 3, RC = 16 * T - 1;
  const double* zi = traj + ((size_t)b * (N + 1) + i) * n;
  for (int ti = 0; ti < T; ti++)
    for (int tj = 0; tj < T; tj++) {
      Asm at(P, pb, rec_of(pb, pb.which[b], bufsel), gpu_of(pb, pb.which[b], bufsel), b, lane, 16 * ti, 16 * tj, RC);
      Tile S, Cl, Cr;
      at.build_tiles(i, slot0, slot1, zi, S, Cl, Cr, true);
      const int cc = 16 * tj + c;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int rho = 16 * ti + g + 4 * k;
        if (rho < n && cc < n) {
          if (Hd) Hd[(((size_t)b * (N + 1) + i) * n + rho) * n + cc] = S.r[k];
          // the ABI exports block (i+1, i) = H_{i,i+1}^T
          if (Ho && i < N) Ho[(((size_t)b * N + i) * n + cc) * n + rho] = Cr.r[k];
        }
        if (rho < n && cc == RC && gout) gout[((size_t)b * (N + 1) + i) * n + rho] = -S.r[k];
      }
    }
}

int launch_export_normal_eq(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                            double* Hd, double* Ho, double* g, hipStream_t st, const int* active) {
  const dim3 grid(hp.B * (hp.N + 1)), block(64);
  const size_t shmem = 2 * (size_t)((hp.I + 1) * hp.RECS + hp.GPS + 24 * hp.I) * sizeof(double);
  switch (hp.D) {
#define G2_EXP_CASE(DD) \
  case DD:                                                                                          \
    if (hp.lie) k_export_normal_eq<DD, true><<<grid, block, shmem, st>>>(pb.params, pb, traj, bufsel, Hd, Ho, g, active); \
    else k_export_normal_eq<DD, false><<<grid, block, shmem, st>>>(pb.params, pb, traj, bufsel, Hd, Ho, g, active);       \
    break;
    G2_EXP_CASE(1) G2_EXP_CASE(2) G2_EXP_CASE(3) G2_EXP_CASE(4) G2_EXP_CASE(5) G2_EXP_CASE(6) G2_EXP_CASE(7)
    G2_EXP_CASE(8) G2_EXP_CASE(9) G2_EXP_CASE(10) G2_EXP_CASE(11) G2_EXP_CASE(17) G2_EXP_CASE(18)
#undef G2_EXP_CASE
    default:
      set_error("normal equations are instantiated for dof <= 11 and 17, 18");
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
