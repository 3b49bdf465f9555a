// factor_kernels.hip -- factor-level kernels behind the evaluateError-style C ABI entry points
// (include/gpmp2mi.h "factor-level entry points").  One lane per evaluation; these exist for
// parity testing and for callers that embed single factors; the planner's hot loop uses the
// fused kernels in plan_kernels.hip.
#include "device_math.h"
#include "dispatch.h"
#include "launch.h"

namespace g2 {

// --------------------------------------------------------------------------- SDF packing
// plain [nz][ny][nx] -> cells [nz][ny][nx][2^dim]; neighbour indices clamp at the upper faces
// (their weights are exactly 0 there).
__global__ void k_sdf_pack(int dim, int nx, int ny, int nz, const double* __restrict__ plain,
                           double* __restrict__ cells) {
  const size_t n = (size_t)nx * ny * nz;
  const int nc = dim == 3 ? 8 : 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / ((size_t)nx * ny));
    const int x1 = min(x + 1, nx - 1), y1 = min(y + 1, ny - 1), z1 = min(z + 1, nz - 1);
    for (int c = 0; c < nc; c++) {
      const int xx = (c & 1) ? x1 : x, yy = (c & 2) ? y1 : y, zz = (c & 4) ? z1 : z;
      cells[i * nc + c] = plain[((size_t)zz * ny + yy) * nx + xx];
    }
  }
}

int launch_sdf_pack(const SdfDev& s, double* cells, hipStream_t st) {
  const size_t n = (size_t)s.nx * s.ny * s.nz;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(k_sdf_pack, dim3(grid), dim3(256), 0, st, s.dim, s.nx, s.ny, s.nz, s.plain, cells);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// --------------------------------------------------------------------------- SDF query
__global__ void k_sdf_query(SdfDev s, int M, const double* __restrict__ pts, double* __restrict__ dist,
                            double* __restrict__ grad, int* __restrict__ inr) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double d = 0, gx = 0, gy = 0, gz = 0;
  bool ok;
  if (s.dim == 3) ok = sdf3_lookup(s, pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], d, gx, gy, gz);
  else ok = sdf2_lookup(s, pts[2 * m], pts[2 * m + 1], d, gx, gy);
  if (!ok) d = gx = gy = gz = 0.0;
  dist[m] = d;
  if (grad) {
    grad[(size_t)m * s.dim + 0] = gx;
    grad[(size_t)m * s.dim + 1] = gy;
    if (s.dim == 3) grad[(size_t)m * 3 + 2] = gz;
  }
  if (inr) inr[m] = ok ? 1 : 0;
}

int launch_sdf_query(const SdfDev& s, int M, const double* pts, double* dist, double* grad, int* inr,
                     hipStream_t st) {
  hipLaunchKernelGGL(k_sdf_query, dim3((M + 63) / 64), dim3(64), 0, st, s, M, pts, dist, grad, inr);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// --------------------------------------------------------------------------- sphere centres
template <int KIND, int AD, int AD2>
__global__ void k_sphere_centers(const RobotDev* __restrict__ Rg, int M, const double* __restrict__ conf, int ld,
                                 double* __restrict__ centers, double* __restrict__ J) {
  using K = Kin<KIND, AD, AD2>;
  constexpr int D = K::DOF;
  __shared__ RobotDev R;
  stage_robot(&R, Rg);
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double q[D];
#pragma unroll
  for (int k = 0; k < D; k++) q[k] = conf[(size_t)m * ld + k];   // ld: D, or 2 D for the states of a trajectory
  const int S = R.nr_spheres;
  K::for_each_sphere(R, q, [&](int s, const double (&p)[3], const double (&Jc)[D][3], int) {
    const int so = R.sph_orig[s];
#pragma unroll
    for (int i = 0; i < 3; i++) centers[((size_t)m * S + so) * 3 + i] = p[i];
    if (J)
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < D; k++) J[(((size_t)m * S + so) * 3 + i) * D + k] = Jc[k][i];
  });
}

// --------------------------------------------------------------------------- link poses + body Jacobians
// ForwardKinematics::forwardKinematics(jp, none, jpx, none, J_jpx_jp): poses [L][16], J [L][6][D]
// in GTSAM Pose3 tangent order [omega; v] (body frame)  kinematics/Arm.cpp:105-115.
template <int KIND, int AD, int AD2>
__global__ void k_fk(const RobotDev* __restrict__ Rg, int M, const double* __restrict__ conf, int ld,
                     double* __restrict__ poses, double* __restrict__ Jp) {
  using K = Kin<KIND, AD, AD2>;
  constexpr int D = K::DOF, L = K::NLINKS;
  __shared__ RobotDev R;
  stage_robot(&R, Rg);
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double q[D];
#pragma unroll
  for (int k = 0; k < D; k++) q[k] = conf[(size_t)m * ld + k];
  double* P = poses + (size_t)m * L * 16;
  double* Jm = Jp ? Jp + (size_t)m * L * 6 * D : nullptr;
  auto put_pose = [&](int l, const Frame& F) {
    double* T = P + l * 16;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      T[i * 4 + 0] = F.c0[i];
      T[i * 4 + 1] = F.c1[i];
      T[i * 4 + 2] = F.c2[i];
      T[i * 4 + 3] = F.t[i];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
  };
  // body twist of link frame F for a world-frame rotation axis w through point o (or a pure
  // translation along w when `trans`)
  auto put_col = [&](int l, int k, const Frame& F, const double* w, const double* o, bool trans) {
    if (!Jm) return;
    double om[3] = {0, 0, 0}, v[3];
    if (trans) {
      v[0] = w[0]; v[1] = w[1]; v[2] = w[2];
    } else {
      om[0] = w[0]; om[1] = w[1]; om[2] = w[2];
      const double rx = F.t[0] - o[0], ry = F.t[1] - o[1], rz = F.t[2] - o[2];
      v[0] = w[1] * rz - w[2] * ry;
      v[1] = w[2] * rx - w[0] * rz;
      v[2] = w[0] * ry - w[1] * rx;
    }
    double* Jl = Jm + (size_t)l * 6 * D;
    Jl[0 * D + k] = F.c0[0] * om[0] + F.c0[1] * om[1] + F.c0[2] * om[2];
    Jl[1 * D + k] = F.c1[0] * om[0] + F.c1[1] * om[1] + F.c1[2] * om[2];
    Jl[2 * D + k] = F.c2[0] * om[0] + F.c2[1] * om[1] + F.c2[2] * om[2];
    Jl[3 * D + k] = F.c0[0] * v[0] + F.c0[1] * v[1] + F.c0[2] * v[2];
    Jl[4 * D + k] = F.c1[0] * v[0] + F.c1[1] * v[1] + F.c1[2] * v[2];
    Jl[5 * D + k] = F.c2[0] * v[0] + F.c2[1] * v[1] + F.c2[2] * v[2];
  };
  if (Jm)
    for (int i = 0; i < L * 6 * D; i++) Jm[i] = 0.0;

  if constexpr (KIND == GPMP2MI_ROBOT_POINT) {
    Frame F;
    F.c0[0] = 1; F.c0[1] = 0; F.c0[2] = 0; F.c1[0] = 0; F.c1[1] = 1; F.c1[2] = 0;
    F.c2[0] = 0; F.c2[1] = 0; F.c2[2] = 1; F.t[0] = q[0]; F.t[1] = q[1]; F.t[2] = 0;
    put_pose(0, F);
    const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0};
    put_col(0, 0, F, ex, nullptr, true);
    put_col(0, 1, F, ey, nullptr, true);
  } else {
    typename K::Axes A;
    const double ez[3] = {0, 0, 1};
    K::walk_links(R, q, A, [&](int link, const Frame& F, auto tag) {
      constexpr int NC = decltype(tag)::value, FIRST = decltype(tag)::first;
      put_pose(link, F);
      if constexpr (K::MOBILE) {
        put_col(link, 0, F, A.bx, nullptr, true);
        put_col(link, 1, F, A.by, nullptr, true);
        put_col(link, 2, F, ez, A.vt, false);
        if constexpr (K::LIFT == 1 && NC > 3) {
          const double lz[3] = {0, 0, A.lift_sign};
          put_col(link, 3, F, lz, nullptr, true);
        }
      }
#pragma unroll
      for (int k = FIRST; k < NC; k++) put_col(link, k, F, A.zax[k - K::NB], A.org[k - K::NB], false);
    });
  }
}

// --------------------------------------------------------------------------- obstacle factors
// ObstacleSDFFactor / ObstaclePlanarSDFFactor ::evaluateError
template <int KIND, int AD, int AD2, int SDIM>
__global__ void k_obstacle(const RobotDev* __restrict__ Rg, SdfDev sdf, double eps, int M,
                           const double* __restrict__ conf, double* __restrict__ err,
                           double* __restrict__ H1) {
  using K = Kin<KIND, AD, AD2>;
  constexpr int D = K::DOF;
  __shared__ RobotDev R;
  stage_robot(&R, Rg);
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double q[D];
#pragma unroll
  for (int k = 0; k < D; k++) q[k] = conf[(size_t)m * D + k];
  const int S = R.nr_spheres;
  K::for_each_sphere(R, q, [&](int s, const double (&p)[3], const double (&Jc)[D][3], int) {
    double hx, hy, hz;
    const double e = hinge_obstacle<SDIM>(sdf, p[0], p[1], p[2], R.sph_r[s] + eps, hx, hy, hz);
    const int so = R.sph_orig[s];
    err[(size_t)m * S + so] = e;
    if (H1)
#pragma unroll
      for (int k = 0; k < D; k++)
        H1[((size_t)m * S + so) * D + k] = hx * Jc[k][0] + hy * Jc[k][1] + (SDIM == 3 ? hz * Jc[k][2] : 0.0);
  });
}

// ObstacleSDFFactorGP / ObstaclePlanarSDFFactorGP ::evaluateError with
// GaussianProcessInterpolatorLinear (vector-space robots): conf = l11 c1 + l12 v1 + p11 c2 + p12 v2
// and H_k = Jerr_conf * (scalar_k I)   (gp/GaussianProcessInterpolatorLinear.h:62-96).
template <int KIND, int AD, int AD2, int SDIM>
__global__ void k_obstacle_gp(const RobotDev* __restrict__ Rg, SdfDev sdf, double eps, GpCoef gc, int M,
                              const double* __restrict__ c1, const double* __restrict__ v1,
                              const double* __restrict__ c2, const double* __restrict__ v2,
                              double* __restrict__ err, double* __restrict__ H1, double* __restrict__ H2,
                              double* __restrict__ H3, double* __restrict__ H4) {
  using K = Kin<KIND, AD, AD2>;
  constexpr int D = K::DOF;
  __shared__ RobotDev R;
  stage_robot(&R, Rg);
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double q[D], Ml[4][9];
  if constexpr (K::BASE == 3) {
    // GaussianProcessInterpolatorPose2Vector  gp/GaussianProcessInterpolatorLie.h:64-100
    double x0[D], w0[D], x1[D], w1[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
      const size_t o = (size_t)m * D + k;
      x0[k] = c1[o]; w0[k] = v1[o]; x1[k] = c2[o]; w1[k] = v2[o];
    }
    lie_interpolate<D>(gc, x0, w0, x1, w1, q, Ml);
  } else {
#pragma unroll
    for (int k = 0; k < D; k++) {
      const size_t o = (size_t)m * D + k;
      q[k] = gc.l11 * c1[o] + gc.l12 * v1[o] + gc.p11 * c2[o] + gc.p12 * v2[o];
    }
  }
  const int S = R.nr_spheres;
  const double sc[4] = {gc.l11, gc.l12, gc.p11, gc.p12};
  double* Hs[4] = {H1, H2, H3, H4};
  K::for_each_sphere(R, q, [&](int s, const double (&p)[3], const double (&Jc)[D][3], int) {
    double hx, hy, hz;
    const double e = hinge_obstacle<SDIM>(sdf, p[0], p[1], p[2], R.sph_r[s] + eps, hx, hy, hz);
    const int so = R.sph_orig[s];
    err[(size_t)m * S + so] = e;
    if (!H1) return;
    double Jr[D];
#pragma unroll
    for (int k = 0; k < D; k++) Jr[k] = hx * Jc[k][0] + hy * Jc[k][1] + (SDIM == 3 ? hz * Jc[k][2] : 0.0);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int k = 0; k < D; k++) {
        double v = Jr[k] * sc[t];
        if constexpr (K::BASE == 3) {
          if (k < 3) v = Jr[0] * Ml[t][0 * 3 + k] + Jr[1] * Ml[t][1 * 3 + k] + Jr[2] * Ml[t][2 * 3 + k];
        }
        Hs[t][((size_t)m * S + so) * D + k] = v;
      }
  });
}

// --------------------------------------------------------------------------- small vector factors
// GaussianProcessPriorLinear::evaluateError  gp/GaussianProcessPriorLinear.h:57-83
__global__ void k_gp_prior_linear(int D, double dt, int M, const double* __restrict__ c1,
                                  const double* __restrict__ v1, const double* __restrict__ c2,
                                  const double* __restrict__ v2, double* __restrict__ err,
                                  double* __restrict__ H1, double* __restrict__ H2,
                                  double* __restrict__ H3, double* __restrict__ H4) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  for (int k = 0; k < D; k++) {
    const size_t o = (size_t)m * D + k;
    err[(size_t)m * 2 * D + k] = c1[o] + dt * v1[o] - c2[o];
    err[(size_t)m * 2 * D + D + k] = v1[o] - v2[o];
  }
  if (!H1) return;
  const size_t hb = (size_t)m * 2 * D * D;
  for (int i = 0; i < 2 * D * D; i++) H1[hb + i] = H2[hb + i] = H3[hb + i] = H4[hb + i] = 0.0;
  for (int k = 0; k < D; k++) {
    H1[hb + (size_t)k * D + k] = 1.0;
    H2[hb + (size_t)k * D + k] = dt;
    H2[hb + (size_t)(D + k) * D + k] = 1.0;
    H3[hb + (size_t)k * D + k] = -1.0;
    H4[hb + (size_t)(D + k) * D + k] = -1.0;
  }
}

// GaussianProcessInterpolatorLinear::interpolatePose / interpolateVelocity
__global__ void k_gp_interp_linear(int D, GpCoef gc, int M, const double* __restrict__ c1,
                                   const double* __restrict__ v1, const double* __restrict__ c2,
                                   const double* __restrict__ v2, double* __restrict__ conf,
                                   double* __restrict__ vel) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  for (int k = 0; k < D; k++) {
    const size_t o = (size_t)m * D + k;
    if (conf) conf[o] = gc.l11 * c1[o] + gc.l12 * v1[o] + gc.p11 * c2[o] + gc.p12 * v2[o];
    if (vel) vel[o] = gc.l21 * c1[o] + gc.l22 * v1[o] + gc.p21 * c2[o] + gc.p22 * v2[o];
  }
}

// GaussianProcessPriorLie<Pose2Vector>::evaluateError  gp/GaussianProcessPriorLie.h:61-86
// states [x, y, theta, q...]; err [2D]; H1..H4 [2D][D]
template <int D>
__global__ void k_gp_prior_lie(double dt, int M, const double* __restrict__ c1, const double* __restrict__ v1,
                               const double* __restrict__ c2, const double* __restrict__ v2,
                               double* __restrict__ err, double* __restrict__ H1, double* __restrict__ H2,
                               double* __restrict__ H3, double* __restrict__ H4) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const double* x0 = c1 + (size_t)m * D;
  const double* x1 = c2 + (size_t)m * D;
  const P2 p1{x0[0], x0[1], x0[2]}, p2{x1[0], x1[1], x1[2]};
  const P2 bt = pose2_between(p1, p2);
  double lg[3], Hinv[9], Hc1[9], Hlog[9], T[9], J1[9];
  pose2_logmap(bt, lg);
  pose2_adjoint(p1, Hinv);
  pose2_adjoint(pose2_inverse(p2), Hc1);
  pose2_logmap_derivative(bt, Hlog);
  mat3_mul(Hlog, Hc1, T);
  mat3_mul(T, Hinv, J1);
  for (int k = 0; k < D; k++) {
    const double r = (k < 3) ? lg[k] : (x1[k] - x0[k]);
    err[(size_t)m * 2 * D + k] = r - v1[(size_t)m * D + k] * dt;
    err[(size_t)m * 2 * D + D + k] = v2[(size_t)m * D + k] - v1[(size_t)m * D + k];
  }
  if (!H1) return;
  const size_t hb = (size_t)m * 2 * D * D;
  for (int i = 0; i < 2 * D * D; i++) H1[hb + i] = H2[hb + i] = H3[hb + i] = H4[hb + i] = 0.0;
  for (int r = 0; r < D; r++)
    for (int c = 0; c < D; c++) {
      double a = 0.0, b = 0.0;
      if (r < 3 && c < 3) { a = -J1[r * 3 + c]; b = Hlog[r * 3 + c]; }
      else if (r == c) { a = -1.0; b = 1.0; }
      H1[hb + (size_t)r * D + c] = a;
      H3[hb + (size_t)r * D + c] = b;
    }
  for (int k = 0; k < D; k++) {
    H2[hb + (size_t)k * D + k] = -dt;
    H2[hb + (size_t)(D + k) * D + k] = -1.0;
    H4[hb + (size_t)(D + k) * D + k] = 1.0;
  }
}

// GaussianProcessInterpolatorLie<Pose2Vector>::interpolatePose / interpolateVelocity
// gp/GaussianProcessInterpolatorLie.h:64-146
template <int D>
__global__ void k_gp_interp_lie(GpCoef gc, int M, const double* __restrict__ c1, const double* __restrict__ v1,
                                const double* __restrict__ c2, const double* __restrict__ v2,
                                double* __restrict__ conf, double* __restrict__ vel) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double x0[D], w0[D], x1[D], w1[D], q[D];
  for (int k = 0; k < D; k++) {
    const size_t o = (size_t)m * D + k;
    x0[k] = c1[o]; w0[k] = v1[o]; x1[k] = c2[o]; w1[k] = v2[o];
  }
  if (conf) {
    lie_interpolate<D>(gc, x0, w0, x1, w1, q, nullptr);
    for (int k = 0; k < D; k++) conf[(size_t)m * D + k] = q[k];
  }
  if (vel) {
    const P2 bt = pose2_between(P2{x0[0], x0[1], x0[2]}, P2{x1[0], x1[1], x1[2]});
    double lg[3];
    pose2_logmap(bt, lg);
    for (int k = 0; k < D; k++) {
      const double r = (k < 3) ? lg[k] : (x1[k] - x0[k]);
      vel[(size_t)m * D + k] = gc.l22 * w0[k] + gc.p21 * r + gc.p22 * w1[k];
    }
  }
}

// Lambda / Psi scalars of one sub-step (gpmp2/gp/GPutils.h:44-59 with Qc factored out); device twin of
// the host gp_coef in api.hip so that a densify launch needs no coefficient upload
__device__ __forceinline__ GpCoef gp_coef_dev(double dt, double tau) {
  const double a0 = tau * tau * tau / 3.0, a1 = 0.5 * tau * tau, r = dt - tau;
  const double w0 = 12.0 / (dt * dt * dt), w1 = -6.0 / (dt * dt), w3 = 4.0 / dt;
  // T = A(tau) Phi(dt - tau)^T ;  Psi = T Q^-1(dt)
  const double t0 = a0 + a1 * r, t1 = a1, t2 = a1 + tau * r, t3 = tau;
  GpCoef c;
  c.p11 = t0 * w0 + t1 * w1;
  c.p12 = t0 * w1 + t1 * w3;
  c.p21 = t2 * w0 + t3 * w1;
  c.p22 = t2 * w1 + t3 * w3;
  c.l11 = 1.0 - c.p11;
  c.l12 = tau - (c.p11 * dt + c.p12);
  c.l21 = -c.p21;
  c.l22 = 1.0 - (c.p21 * dt + c.p22);
  return c;
}

// interpolateArmTraj / interpolatePose2MobileArmTraj  gpmp2/planner/TrajUtils.cpp:162-236
// one thread per output state: support state i copies, the inter_step states after it interpolate
// between i and i + 1 at tau = j * delta_t / (inter_step + 1).  traj [B][N+1][2D] -> out [B][Mo][2D]
template <int D, bool LIE>
__global__ void k_interpolate_traj(double dt, int inter, int B, int N, int start, int Mo,
                                   const double* __restrict__ traj, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * Mo) return;
  const int b = t / Mo, m = t % Mo;
  const int seg = m / (inter + 1), j = m % (inter + 1);
  const double* s0 = traj + ((size_t)b * (N + 1) + start + seg) * 2 * D;
  double* o = out + (size_t)t * 2 * D;
  if (j == 0) {
#pragma unroll
    for (int k = 0; k < 2 * D; k++) o[k] = s0[k];
    return;
  }
  const double* s1 = s0 + 2 * D;
  const GpCoef gc = gp_coef_dev(dt, (double)j * (dt / (double)(inter + 1)));
  double x0[D], w0[D], x1[D], w1[D];
#pragma unroll
  for (int k = 0; k < D; k++) { x0[k] = s0[k]; w0[k] = s0[D + k]; x1[k] = s1[k]; w1[k] = s1[D + k]; }
  if constexpr (LIE) {
    double q[D], lg[3];
    lie_interpolate<D>(gc, x0, w0, x1, w1, q, nullptr);
    pose2_logmap(pose2_between(P2{x0[0], x0[1], x0[2]}, P2{x1[0], x1[1], x1[2]}), lg);
#pragma unroll
    for (int k = 0; k < D; k++) {
      const double r = (k < 3) ? lg[k] : (x1[k] - x0[k]);
      o[k] = q[k];
      o[D + k] = gc.l22 * w0[k] + gc.p21 * r + gc.p22 * w1[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < D; k++) {
      o[k] = gc.l11 * x0[k] + gc.l12 * w0[k] + gc.p11 * x1[k] + gc.p12 * w1[k];
      o[D + k] = gc.l21 * x0[k] + gc.l22 * w0[k] + gc.p21 * x1[k] + gc.p22 * w1[k];
    }
  }
}

// --------------------------------------------------------------------------- SO(3) / SE(3) logs
// GTSAM 4.0 semantics (upstream, restated): SO3::Logmap / LogmapDerivative, Pose3::Logmap /
// LogmapDerivative with Barfoot's Q matrix.  Used by the workspace priors below.
__device__ __forceinline__ void skew3(const double* w, double* W) {
  W[0] = 0; W[1] = -w[2]; W[2] = w[1];
  W[3] = w[2]; W[4] = 0; W[5] = -w[0];
  W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}
__device__ __forceinline__ void rot3_logmap(const double* R, double* w) {
  const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  constexpr double kPi = 3.14159265358979323846;
  if (fabs(tr + 1.0) < 1e-10) {
    if (fabs(R33 + 1.0) > 1e-10) {
      const double k = kPi / sqrt(2.0 + 2.0 * R33);
      w[0] = k * R13; w[1] = k * R23; w[2] = k * (1.0 + R33);
    } else if (fabs(R22 + 1.0) > 1e-10) {
      const double k = kPi / sqrt(2.0 + 2.0 * R22);
      w[0] = k * R12; w[1] = k * (1.0 + R22); w[2] = k * R32;
    } else {
      const double k = kPi / sqrt(2.0 + 2.0 * R11);
      w[0] = k * (1.0 + R11); w[1] = k * R21; w[2] = k * R31;
    }
  } else {
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-7) {
      const double theta = acos((tr - 1.0) / 2.0);
      magnitude = theta / (2.0 * sin(theta));
    } else {
      magnitude = 0.5 - tr_3 * tr_3 / 12.0;
    }
    w[0] = magnitude * (R32 - R23);
    w[1] = magnitude * (R13 - R31);
    w[2] = magnitude * (R21 - R12);
  }
}
__device__ __forceinline__ void rot3_logmap_derivative(const double* w, double* H) {
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  for (int i = 0; i < 9; i++) H[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (theta2 <= 2.220446049250313e-16) return;
  const double theta = sqrt(theta2);
  double W[9], WW[9];
  skew3(w, W);
  mat3_mul(W, W, WW);
  const double k = 1.0 / (theta * theta) - (1.0 + cos(theta)) / (2.0 * theta * sin(theta));
  for (int i = 0; i < 9; i++) H[i] += 0.5 * W[i] + k * WW[i];
}
__device__ __forceinline__ void pose3_logmap(const double* R, const double* T, double* xi) {
  double w[3];
  rot3_logmap(R, w);
  const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (t < 1e-10) {
    xi[3] = T[0]; xi[4] = T[1]; xi[5] = T[2];
    return;
  }
  const double wn[3] = {w[0] / t, w[1] / t, w[2] / t};
  double W[9], WT[3], WWT[3];
  skew3(wn, W);
  const double Tan = tan(0.5 * t);
  for (int i = 0; i < 3; i++) WT[i] = W[i * 3] * T[0] + W[i * 3 + 1] * T[1] + W[i * 3 + 2] * T[2];
  for (int i = 0; i < 3; i++) WWT[i] = W[i * 3] * WT[0] + W[i * 3 + 1] * WT[1] + W[i * 3 + 2] * WT[2];
  for (int i = 0; i < 3; i++) xi[3 + i] = T[i] - (0.5 * t) * WT[i] + (1.0 - t / (2.0 * Tan)) * WWT[i];
}
__device__ __forceinline__ void pose3_logmap_derivative(const double* R, const double* T, double* H) {
  double xi[6], Jw[9], Q[9], Q2[9];
  pose3_logmap(R, T, xi);
  rot3_logmap_derivative(xi, Jw);
  {
    double V[9], W[9], WV[9], VW[9], WVW[9], WW[9], WWV[9], VWW[9], WVWW[9], WWVW[9];
    skew3(xi + 3, V);
    skew3(xi, W);
    mat3_mul(W, V, WV);
    mat3_mul(V, W, VW);
    mat3_mul(WV, W, WVW);
    mat3_mul(W, W, WW);
    mat3_mul(WW, V, WWV);
    mat3_mul(VW, W, VWW);
    mat3_mul(WVW, W, WVWW);
    mat3_mul(W, WVW, WWVW);
    const double phi = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    double c1, c2, c3;
    if (fabs(phi) > 1e-5) {
      const double sn = sin(phi), c = cos(phi);
      const double phi2 = phi * phi, phi3 = phi2 * phi, phi4 = phi3 * phi, phi5 = phi4 * phi;
      c1 = (phi - sn) / phi3;
      c2 = (1.0 - phi2 / 2.0 - c) / phi4;
      c3 = -0.5 * ((1.0 - phi2 / 2.0 - c) / phi4 - 3.0 * (phi - sn - phi3 / 6.0) / phi5);
    } else {
      c1 = 1.0 / 6.0;
      c2 = 1.0 / 24.0;
      c3 = -0.5 * (1.0 / 24.0 + 3.0 / 120.0);
    }
    for (int i = 0; i < 9; i++)
      Q[i] = -0.5 * V[i] + c1 * (WV[i] + VW[i] - WVW[i]) + c2 * (WWV[i] + VWW[i] - 3.0 * WVW[i]) + c3 * (WVWW[i] + WWVW[i]);
  }
  mat3_mul(Jw, Q, Q2);
  double Q3[9];
  mat3_mul(Q2, Jw, Q3);
  for (int i = 0; i < 36; i++) H[i] = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      H[i * 6 + j] = Jw[i * 3 + j];
      H[(3 + i) * 6 + 3 + j] = Jw[i * 3 + j];
      H[(3 + i) * 6 + j] = -Q3[i * 3 + j];
    }
}

// GaussianPriorWorkspace{Position,Orientation,Pose}::evaluateError and GoalFactorArm::evaluateError
// (kinematics/GaussianPriorWorkspacePosition.h:52-67, ...Orientation.h:52-69, ...Pose.h:53-70,
// kinematics/GoalFactorArm.h:58-77) on the link poses / pose Jacobians k_fk produced:
// poses [M][L][16], Jp [M][L][6][D]; err [M][rows], H [M][rows][D] (may be null)
__global__ void k_workspace_prior(int mode, int joint, int L, int D, int M, const double* __restrict__ des,
                                  const double* __restrict__ poses, const double* __restrict__ Jp,
                                  double* __restrict__ err, double* __restrict__ H) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const double* T = poses + ((size_t)m * L + joint) * 16;
  const double* J6 = Jp ? Jp + ((size_t)m * L + joint) * 6 * D : nullptr;
  const int rows = mode == 2 ? 6 : 3;
  double* e = err + (size_t)m * rows;
  double* Hm = H ? H + (size_t)m * rows * D : nullptr;
  double Rm[9], t[3], Rd[9], td[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      Rm[i * 3 + j] = T[i * 4 + j];
      Rd[i * 3 + j] = des[i * 4 + j];
    }
    t[i] = T[i * 4 + 3];
    td[i] = des[i * 4 + 3];
  }
  if (mode == 0) {
    for (int i = 0; i < 3; i++) e[i] = t[i] - td[i];
    if (Hm)  // Pose3::translation(H) = [0 R]
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < D; k++) {
          double a = 0;
          for (int c = 0; c < 3; c++) a += Rm[i * 3 + c] * J6[(3 + c) * D + k];
          Hm[i * D + k] = a;
        }
    return;
  }
  double Rrel[9], trel[3];  // between(des, pose)
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      double a = 0;
      for (int c = 0; c < 3; c++) a += Rd[c * 3 + i] * Rm[c * 3 + j];
      Rrel[i * 3 + j] = a;
    }
    trel[i] = Rd[i] * (t[0] - td[0]) + Rd[3 + i] * (t[1] - td[1]) + Rd[6 + i] * (t[2] - td[2]);
  }
  if (mode == 1) {
    double w[3], Her[9];
    rot3_logmap(Rrel, w);
    for (int i = 0; i < 3; i++) e[i] = w[i];
    if (Hm) {
      rot3_logmap_derivative(w, Her);  // Pose3::rotation(H) = [I 0]
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < D; k++) {
          double a = 0;
          for (int c = 0; c < 3; c++) a += Her[i * 3 + c] * J6[c * D + k];
          Hm[i * D + k] = a;
        }
    }
    return;
  }
  double xi[6];
  pose3_logmap(Rrel, trel, xi);
  for (int i = 0; i < 6; i++) e[i] = xi[i];
  if (Hm) {
    double Hep[36];
    pose3_logmap_derivative(Rrel, trel, Hep);
    for (int i = 0; i < 6; i++)
      for (int k = 0; k < D; k++) {
        double a = 0;
        for (int c = 0; c < 6; c++) a += Hep[i * 6 + c] * J6[c * D + k];
        Hm[i * D + k] = a;
      }
  }
}

// SelfCollision::evaluateError obstacle/SelfCollision.h:66-128 on the sphere centres / Jacobians
// k_sphere_centers produced: c [M][S][3], Jc [M][S][3][D]; data [n][4]; radius [S] (caller's order)
__global__ void k_self_collision(int n, int S, int D, int M, const double* __restrict__ data,
                                 const double* __restrict__ radius, const double* __restrict__ c,
                                 const double* __restrict__ Jc, double* __restrict__ err, double* __restrict__ H) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * n) return;
  const int m = t / n, i = t - m * n;
  const int a = (int)data[i * 4 + 0], b = (int)data[i * 4 + 1];
  const double eps = radius[a] + radius[b] + data[i * 4 + 2];
  const double* ca = c + ((size_t)m * S + a) * 3;
  const double* cb = c + ((size_t)m * S + b) * 3;
  const double dx = ca[0] - cb[0], dy = ca[1] - cb[1], dz = ca[2] - cb[2];
  const double dist = sqrt(dx * dx + dy * dy + dz * dz);
  const bool active = !(dist > eps);
  err[t] = active ? eps - dist : 0.0;
  if (!H) return;
  double* Hr = H + (size_t)t * D;
  const double nn[3] = {dx / dist, dy / dist, dz / dist};
  const double* Ja = Jc + ((size_t)m * S + a) * 3 * D;
  const double* Jb = Jc + ((size_t)m * S + b) * 3 * D;
  for (int k = 0; k < D; k++) {
    double v = 0;
    for (int q = 0; q < 3; q++) v += -nn[q] * Ja[q * D + k] + nn[q] * Jb[q * D + k];
    Hr[k] = active ? v : 0.0;
  }
}

// simple2DVehicleDynamicsPose2 / ...Vector3  dynamics/VehicleDynamics.h:19-40
__global__ void k_vehicle_dynamics(int D, int lie, int M, const double* __restrict__ conf, const double* __restrict__ vel,
                                   double* __restrict__ err, double* __restrict__ Hp, double* __restrict__ Hv) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const double* p = conf + (size_t)m * D;
  const double* v = vel + (size_t)m * D;
  double hp[3] = {0.0, 0.0, 0.0}, hv[3] = {0.0, 1.0, 0.0}, e = v[1];
  if (!lie) {
    double sn, cs;
    sincos(p[2], &sn, &cs);
    hp[2] = -(v[1] * sn + v[0] * cs);
    hv[0] = -sn;
    hv[1] = cs;
    e = v[1] * cs - v[0] * sn;
  }
  err[m] = e;
  for (int k = 0; k < D; k++) {
    if (Hp) Hp[(size_t)m * D + k] = (k < 3) ? hp[k] : 0.0;
    if (Hv) Hv[(size_t)m * D + k] = (k < 3) ? hv[k] : 0.0;
  }
}

// JointLimitFactorVector / VelocityLimitFactorVector ::evaluateError
__global__ void k_joint_limit(int D, const double* __restrict__ down, const double* __restrict__ up,
                              const double* __restrict__ th, int M, const double* __restrict__ x,
                              double* __restrict__ err, double* __restrict__ Hd) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  for (int k = 0; k < D; k++) {
    double H;
    err[(size_t)m * D + k] = hinge_limit(x[(size_t)m * D + k], down[k], up[k], th[k], H);
    if (Hd) Hd[(size_t)m * D + k] = H;
  }
}

// --------------------------------------------------------------------------- launchers
#define G2_GRID(M) dim3(((M) + 63) / 64), dim3(64), 0, st

int launch_sphere_centers(const RobotDev& h, const RobotDev* R, int M, const double* conf, double* c,
                          double* J, hipStream_t st, int ld) {
  if (ld <= 0) ld = h.dof;
  G2_DISPATCH_ROBOT_H(h, (k_sphere_centers<KIND_, AD_, AD2_><<<G2_GRID(M)>>>(R, M, conf, ld, c, J)));
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_fk(const RobotDev& h, const RobotDev* R, int M, const double* conf, double* poses, double* J,
              hipStream_t st, int ld) {
  if (ld <= 0) ld = h.dof;
  G2_DISPATCH_ROBOT_H(h, (k_fk<KIND_, AD_, AD2_><<<G2_GRID(M)>>>(R, M, conf, ld, poses, J)));
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_obstacle(const RobotDev& h, const RobotDev* R, const SdfDev& s, double eps, int M,
                    const double* conf, double* err, double* H1, hipStream_t st) {
  if (s.dim == 3) {
    G2_DISPATCH_ROBOT_H(h, (k_obstacle<KIND_, AD_, AD2_, 3><<<G2_GRID(M)>>>(R, s, eps, M, conf, err, H1)));
  } else {
    G2_DISPATCH_ROBOT_H(h, (k_obstacle<KIND_, AD_, AD2_, 2><<<G2_GRID(M)>>>(R, s, eps, M, conf, err, H1)));
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_obstacle_gp(const RobotDev& h, const RobotDev* R, const SdfDev& s, double eps, const GpCoef& gc,
                       int M, const double* c1, const double* v1, const double* c2, const double* v2,
                       double* err, double* H1, double* H2, double* H3, double* H4, hipStream_t st) {
  if (s.dim == 3) {
    G2_DISPATCH_ROBOT_H(h, (k_obstacle_gp<KIND_, AD_, AD2_, 3><<<G2_GRID(M)>>>(R, s, eps, gc, M, c1, v1, c2, v2, err, H1, H2, H3, H4)));
  } else {
    G2_DISPATCH_ROBOT_H(h, (k_obstacle_gp<KIND_, AD_, AD2_, 2><<<G2_GRID(M)>>>(R, s, eps, gc, M, c1, v1, c2, v2, err, H1, H2, H3, H4)));
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_gp_prior_linear(int D, double dt, int M, const double* c1, const double* v1, const double* c2,
                           const double* v2, double* err, double* H1, double* H2, double* H3, double* H4,
                           hipStream_t st) {
  k_gp_prior_linear<<<G2_GRID(M)>>>(D, dt, M, c1, v1, c2, v2, err, H1, H2, H3, H4);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_gp_interp_linear(int D, const GpCoef& gc, int M, const double* c1, const double* v1,
                            const double* c2, const double* v2, double* conf, double* vel, hipStream_t st) {
  k_gp_interp_linear<<<G2_GRID(M)>>>(D, gc, M, c1, v1, c2, v2, conf, vel);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_interpolate_traj(int D, bool lie, double dt, int inter, int B, int N, int start, int Mo,
                            const double* traj, double* out, hipStream_t st) {
  const int M = B * Mo;
  if (lie && D < 3) { set_error("Pose2Vector dof must be 3..10"); return GPMP2MI_ERR_UNSUPPORTED; }
  switch (D) {
#define G2_IT(DD) case DD: \
    if (lie) { if constexpr (DD >= 3) k_interpolate_traj<DD, true><<<G2_GRID(M)>>>(dt, inter, B, N, start, Mo, traj, out); } \
    else k_interpolate_traj<DD, false><<<G2_GRID(M)>>>(dt, inter, B, N, start, Mo, traj, out); \
    break;
    G2_IT(1) G2_IT(2) G2_IT(3) G2_IT(4) G2_IT(5) G2_IT(6) G2_IT(7) G2_IT(8) G2_IT(9) G2_IT(10)
#undef G2_IT
    default: set_error("dof must be 1..10"); return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_vehicle_dynamics(int D, int lie, int M, const double* conf, const double* vel, double* err, double* Hp,
                            double* Hv, hipStream_t st) {
  k_vehicle_dynamics<<<G2_GRID(M)>>>(D, lie, M, conf, vel, err, Hp, Hv);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_workspace_prior(int mode, int joint, int L, int D, int M, const double* des, const double* poses,
                           const double* Jp, double* err, double* H, hipStream_t st) {
  k_workspace_prior<<<G2_GRID(M)>>>(mode, joint, L, D, M, des, poses, Jp, err, H);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_self_collision(int n, int S, int D, int M, const double* data, const double* radius, const double* c,
                          const double* Jc, double* err, double* H, hipStream_t st) {
  k_self_collision<<<G2_GRID(M * n)>>>(n, S, D, M, data, radius, c, Jc, err, H);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_gp_prior_lie(int D, double dt, int M, const double* c1, const double* v1, const double* c2,
                        const double* v2, double* err, double* H1, double* H2, double* H3, double* H4,
                        hipStream_t st) {
  switch (D) {
#define G2_GPL(DD) case DD: k_gp_prior_lie<DD><<<G2_GRID(M)>>>(dt, M, c1, v1, c2, v2, err, H1, H2, H3, H4); break;
    G2_GPL(3) G2_GPL(4) G2_GPL(5) G2_GPL(6) G2_GPL(7) G2_GPL(8) G2_GPL(9) G2_GPL(10)
#undef G2_GPL
    default: set_error("Pose2Vector dof must be 3..10"); return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_gp_interp_lie(int D, const GpCoef& gc, int M, const double* c1, const double* v1, const double* c2,
                         const double* v2, double* conf, double* vel, hipStream_t st) {
  switch (D) {
#define G2_GIL(DD) case DD: k_gp_interp_lie<DD><<<G2_GRID(M)>>>(gc, M, c1, v1, c2, v2, conf, vel); break;
    G2_GIL(3) G2_GIL(4) G2_GIL(5) G2_GIL(6) G2_GIL(7) G2_GIL(8) G2_GIL(9) G2_GIL(10)
#undef G2_GIL
    default: set_error("Pose2Vector dof must be 3..10"); return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_joint_limit(int D, const double* down, const double* up, const double* th, int M,
                       const double* x, double* err, double* Hd, hipStream_t st) {
  k_joint_limit<<<G2_GRID(M)>>>(D, down, up, th, M, x, err, Hd);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

}  // namespace g2
