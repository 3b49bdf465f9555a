// dense_kernels.hip -- solve step for robots whose blocks do not fit one 16x16 tile (8 <= dof <= 11).
//
// Same role as k_solve_step (cr_kernels.hip) on the trial-step path: solve the current linearization,
// form the trial point and the step-control scalars of GN / LM / Dogleg.  The system arrives as dense
// blocks from k_export_normal_eq (2x2 tiles per block): D_i in wHd, block (i+1, i) in wHo, gradient in wg.
// One workgroup per trajectory runs a block Cholesky in natural order with the blocks in LDS: the
// augmented matrix [S_i | -g_i | H_{i,i+1}] is reduced row by row, which leaves R_i, y_i = R_i^-T b_i and
// W_i = R_i^-T H_{i,i+1} in place; S_{i+1} = D_{i+1} - W_i^T W_i.  Back-substitution
// x_i = R_i^-1 (y_i - W_i x_{i+1}).  A placeholder for a wide-tile cyclic reduction: O(N) dependent
// blocks instead of O(log N) levels, about 1 ms per solve at N = 100.
#include <hip/hip_runtime.h>

#include "common.h"
#include "device_math.h"
#include "plan.h"
#include "tiles.h"

namespace g2 {

constexpr int DENSE_THREADS = 256;
constexpr int DENSE_WAVES = DENSE_THREADS / 64;

__device__ __forceinline__ double dense_block_sum(double v, double* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < DENSE_WAVES; k++) t += red[k];
  return t;
}

__global__ __launch_bounds__(DENSE_THREADS) void k_solve_dense(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!pb.active[b]) return;
  const int N = P.N, n = P.n, D = P.D;
  const int AW = 2 * n + 1;  // augmented width: [S (n) | rhs (1) | H_{i,i+1} (n)]
  const size_t tsz = (size_t)(N + 1) * n;
  const double* cur = pb.cur + b * tsz;
  double* trial = pb.trial + b * tsz;
  double* delta = pb.delta + b * tsz;
  double* sc = pb.scal + (size_t)b * SC_COUNT;
  double* Hd = pb.wHd + (size_t)b * (N + 1) * n * n;
  double* Ho = pb.wHo + (size_t)b * N * n * n;
  const double* gv = pb.wg + (size_t)b * tsz;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;                       // [(N+1)][n]  y, then x
  double* A = xs + tsz;                    // [n][AW]     augmented block
  double* W = A + (size_t)n * AW;          // [n][n]      W_{i-1}
  double* red = W + (size_t)n * n;         // [DENSE_WAVES]
  int* flags = reinterpret_cast<int*>(red + DENSE_WAVES);
  const bool dogleg = P.opt_type == GPMP2MI_OPT_DOGLEG;
  const bool resolve = !(dogleg && pb.phase[b] != 0);
  const int ty = tid >> 4, tx = tid & 15;
  if (tid == 0) flags[0] = 0;
  __syncthreads();
  if (resolve) {
    if (dogleg) {
      // g^T H g from the untouched blocks: sum_i g_i^T D_i g_i + 2 g_{i+1}^T H_{i+1,i} g_i
      double acc = 0.0;
      for (size_t e = tid; e < (size_t)(N + 1) * n * n; e += DENSE_THREADS) {
        const int i = (int)(e / (n * n)), r = (int)((e / n) % n), c = (int)(e % n);
        acc = fma(gv[(size_t)i * n + r] * Hd[e], gv[(size_t)i * n + c], acc);
      }
      for (size_t e = tid; e < (size_t)N * n * n; e += DENSE_THREADS) {
        const int i = (int)(e / (n * n)), r = (int)((e / n) % n), c = (int)(e % n);
        acc = fma(2.0 * gv[(size_t)(i + 1) * n + r] * Ho[e], gv[(size_t)i * n + c], acc);
      }
      acc = dense_block_sum(acc, red, tid);
      if (tid == 0) sc[SC_GHG] = acc;
    }
    const double lam = (P.opt_type == GPMP2MI_OPT_LM) ? pb.lambda[b] : 0.0;
    // ---- forward: block Cholesky in natural order
    for (int i = 0; i <= N; i++) {
      for (int r = ty; r < n; r += 16)
        for (int c = tx; c < AW; c += 16) {
          double v;
          if (c < n) {
            v = Hd[((size_t)i * n + r) * n + c] + ((r == c) ? lam : 0.0);
            if (i > 0) {  // Schur complement of the previous block
              double s = 0.0;
              for (int k = 0; k < n; k++) s = fma(W[k * n + r], W[k * n + c], s);
              v -= s;
            }
          } else if (c == n) {
            v = -gv[(size_t)i * n + r];
            if (i > 0) {
              double s = 0.0;
              for (int k = 0; k < n; k++) s = fma(W[k * n + r], xs[(size_t)(i - 1) * n + k], s);
              v -= s;
            }
          } else {
            v = (i < N) ? Ho[((size_t)i * n + (c - n - 1)) * n + r] : 0.0;  // H_{i,i+1} = block (i+1, i)^T
          }
          A[r * AW + c] = v;
        }
      __syncthreads();
      // right-looking elimination on the 16 x 16 thread grid (ty: rows, tx: column strips); row k stays
      // unscaled (R[k][c] = A[k][c] / sqrt(p_k) is applied when the block is stored): one barrier per pivot
      for (int k = 0; k < n; k++) {
        const double piv = A[k * AW + k];
        if (!(piv > 0.0)) {  // uniform: every thread reads the same LDS value
          if (tid == 0) pb.notspd[b] = 1;  // k_decide consumes and clears it
          return;
        }
        const double ipiv = 1.0 / piv;
        for (int j = k + 1 + ty; j < n; j += 16) {
          const double m = A[k * AW + j] * ipiv;
          for (int c = tx; c < AW; c += 16)
            if (c >= j) A[j * AW + c] = fma(-m, A[k * AW + c], A[j * AW + c]);
        }
        __syncthreads();
      }
      // keep R_i (upper), W_i for the back-substitution; y_i in xs; W_i also stays in LDS for block i+1
      for (int r = ty; r < n; r += 16) {
        const double isq = 1.0 / sqrt(A[r * AW + r]);
        for (int c = tx; c < n; c += 16) {
          Hd[((size_t)i * n + r) * n + c] = (c >= r) ? A[r * AW + c] * isq : 0.0;
          const double w = A[r * AW + n + 1 + c] * isq;
          W[r * n + c] = w;
          if (i < N) Ho[((size_t)i * n + r) * n + c] = w;
        }
        if (tx == 0) xs[(size_t)i * n + r] = A[r * AW + n] * isq;
      }
      __syncthreads();
    }
    // ---- backward: x_i = R_i^-1 (y_i - W_i x_{i+1}); thread r carries t_r, one barrier per unknown
    double* xk = W;  // [n] scratch for the unknown being broadcast (W is free now); R_i is staged in A
    for (int i = N; i >= 0; i--) {
      for (int e = tid; e < n * n; e += DENSE_THREADS) A[e] = Hd[(size_t)i * n * n + e];
      double t = 0.0;
      if (tid < n) {
        t = xs[(size_t)i * n + tid];
        if (i < N) {
          const double* Wi = Ho + (size_t)i * n * n;
          for (int c = 0; c < n; c++) t = fma(-Wi[tid * n + c], xs[(size_t)(i + 1) * n + c], t);
        }
      }
      __syncthreads();
      for (int k = n - 1; k >= 0; k--) {
        if (tid == k) xk[k] = t / A[k * n + k];
        __syncthreads();
        if (tid < k) t = fma(-A[tid * n + k], xk[k], t);
      }
      if (tid < n) xs[(size_t)i * n + tid] = xk[tid];
      __syncthreads();
    }
    double gd = 0.0, dd = 0.0, gg = 0.0;
    for (size_t k = tid; k < tsz; k += DENSE_THREADS) {
      const double x = xs[k], gk = gv[k];
      delta[k] = x;
      gd = fma(gk, x, gd);
      dd = fma(x, x, dd);
      gg = fma(gk, gk, gg);
    }
    gd = dense_block_sum(gd, red, tid);
    dd = dense_block_sum(dd, red, tid);
    gg = dense_block_sum(gg, red, tid);
    if (tid == 0) {
      sc[SC_GD] = gd;
      sc[SC_DD] = dd;
      sc[SC_GG] = gg;
      sc[SC_GN] = gd;
      sc[SC_NN] = dd;
    }
    __syncthreads();
  }
  if (!dogleg) {
    for (size_t k = tid; k < tsz; k += DENSE_THREADS) {
      const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
      const double* zs = cur + (size_t)i * n;
      const double* dz = xs + (size_t)i * n;
      trial[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
    }
    return;
  }
  // ---- Powell dogleg point for trust radius pb.lambda[b]  (same blend as k_solve_step)
  const double Delta = pb.lambda[b];
  const double gg = sc[SC_GG], gHg = sc[SC_GHG], gn = sc[SC_GN], nn = sc[SC_NN];
  const double step = -gg / gHg;  // dx_u = step * g   (optimizeGradientSearch)
  const double uu = step * step * gg, un = step * gn;
  const double DeltaSq = Delta * Delta;
  double cu, cn, q;  // dx_d = cu * g + cn * dx_n
  if (DeltaSq < uu) {
    const double k = sqrt(DeltaSq / uu);
    cu = k * step;
    cn = 0.0;
    q = cu * gg + 0.5 * cu * cu * gHg;
  } else if (DeltaSq < nn) {
    const double a = uu - 2. * un + nn, bq = 2. * (un - uu), cq = uu - Delta * Delta;
    const double sq = sqrt(bq * bq - 4 * a * cq);
    const double tau1 = (-bq + sq) / (2. * a), tau2 = (-bq - sq) / (2. * a);
    const double tau = (0.0 <= tau1 && tau1 <= 1.0) ? tau1 : tau2;
    cu = (1. - tau) * step;
    cn = tau;
    q = cu * gg + cn * gn + 0.5 * (cu * cu * gHg - 2.0 * cu * cn * gg - cn * cn * gn);
  } else {
    cu = 0.0;
    cn = 1.0;
    q = 0.5 * gn;
  }
  double xn = 0.0;
  __syncthreads();
  for (size_t k = tid; k < tsz; k += DENSE_THREADS) {
    const double x = cu * gv[k] + cn * delta[k];
    xs[k] = x;
    xn = fma(x, x, xn);
  }
  __syncthreads();
  for (size_t k = tid; k < tsz; k += DENSE_THREADS) {
    const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
    const double* zs = cur + (size_t)i * n;
    const double* dz = xs + (size_t)i * n;
    trial[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
  }
  xn = dense_block_sum(xn, red, tid);
  if (tid == 0) {
    sc[SC_Q] = q;
    sc[SC_XNORM] = sqrt(xn);
  }
}

int launch_solve_dense(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const int n = hp.n;
  const size_t shmem = ((size_t)(hp.N + 1) * n + (size_t)n * (2 * n + 1) + (size_t)n * n + DENSE_WAVES + 2) * sizeof(double);
  if (shmem > 150 * 1024) {
    set_error("total_step too large for the LDS-resident dense solve");
    return GPMP2MI_ERR_UNSUPPORTED;
  }
  k_solve_dense<<<dim3(hp.B), dim3(DENSE_THREADS), shmem, st>>>(pb.params, pb);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

}  // namespace g2
