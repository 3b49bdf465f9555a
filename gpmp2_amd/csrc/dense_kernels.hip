// dense_kernels.hip -- solve step for robots whose blocks do not fit the tile kernels (12 <= dof <= 18, the PR2
// model; also 8 <= dof <= 11 with GPMP2MI_WIDE_DENSE=1 as an independent implementation of the 2x2-tile path).
//
// Same role as k_solve_step (cr_kernels.hip) on the trial-step path: solve the current linearization,
// form the trial point and the step-control scalars of GN / LM / Dogleg.  The system arrives as dense
// blocks from k_export_normal_eq: D_i in wHd, block (i+1, i) in wHo, gradient in wg.  The solve is the same
// cyclic reduction as the tile kernels with one launch per level (k_dense_cr_level forward, k_dense_cr_back
// backward) and a per-trajectory tail (k_dense_tail).
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

// =============================================================================== dense blocks, cyclic reduction
// The same tree as the tile kernels (cr_kernels.hip), for blocks of any width n <= 36, with the blocks in LDS and the
// levels spread over the chip: one workgroup per task, one launch per level.  Forward level h: every block that is
// a multiple of h absorbs the Schur complements of its neighbours j -+ h/2 (eliminated one level below); odd
// multiples are then eliminated (E task: [S | b | C_l | C_r] -> R, y = R^-T b, W_l = R^-T C_l, W_r = R^-T C_r), even
// multiples store their updated block (U task).  Backward level h: x_j = R_j^-1 (y_j - W_l x_{j-h} - W_r x_{j+h}).
// N = 50, n = 36 (PR2): 7 + 7 launches of <= 25 tasks per trajectory instead of 51 dependent blocks in one workgroup.
__device__ __forceinline__ int dense_hfinal(int N) {
  int h = 1;
  while (h <= N) h <<= 1;
  return h;
}

__global__ __launch_bounds__(DENSE_THREADS) void k_dense_ghg(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!pb.active[b] || pb.phase[b] != 0) return;
  const int N = P.N, n = P.n;
  const double* Hd = pb.wHd + (size_t)b * (N + 1) * n * n;
  const double* Ho = pb.wHo + (size_t)b * N * n * n;
  const double* gv = pb.wg + (size_t)b * (N + 1) * n;
  __shared__ double red[DENSE_WAVES];
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
  if (tid == 0) pb.scal[(size_t)b * SC_COUNT + SC_GHG] = acc;
}

// Thread (ty, tx) of the 16 x 16 grid owns rows ty + 16 a (a < 3) and columns tx + 16 b (b < 7) of the augmented
// block [S | b | C_l | C_r] (n <= 36) and keeps its 21 entries in registers from the first load to the
// final store: the Schur complements are register-blocked products over LDS copies of the neighbour's factors, the
// elimination passes only the pivot row through LDS (double-buffered: one barrier per pivot).
#ifndef G2_DENSE_GRID
#define G2_DENSE_GRID 16
#endif
constexpr int LVG = G2_DENSE_GRID, LVS = (LVG == 32) ? 5 : (LVG == 16) ? 4 : 3;     // thread grid side of k_dense_cr_level
constexpr int LVL_THREADS = LVG * LVG;
constexpr int DENSE_NR = (36 + LVG - 1) / LVG, DENSE_NC = (3 * 36 + 1 + LVG - 1) / LVG;   // 3 x 7 (measured: a 32 x 32 grid
// pays more for its 16-wavefront barriers than it gains, 448 vs 348 us per solve; one wavefront per block 550 us)
__global__ __launch_bounds__(LVL_THREADS) void k_dense_cr_level(const PlanParams* __restrict__ pp, PlanBuffers pb, int h,
                                                                 int final) {
  const PlanParams& P = *pp;
  const int N = P.N, n = P.n, tid = threadIdx.x;
  const int countE = final ? 1 : ((N / h) + 1) / 2;
  const int countU = (final || h == 1) ? 0 : (N / (2 * h)) + 1;   // nothing to absorb at level 1
  const int per = countE + countU;
  const int b = blockIdx.x / per, idx = blockIdx.x - b * per;
  if (!pb.active[b]) return;
  if (P.opt_type == GPMP2MI_OPT_DOGLEG && pb.phase[b] != 0) return;
  const bool elim = idx < countE;
  const int j = elim ? (final ? 0 : h * (2 * idx + 1)) : 2 * h * (idx - countE);
  const int hh = h >> 1, nn = n * n;
  const int AW = 3 * n + 1;  // [S (n) | b (1) | C_l (n) | C_r (n)]
  double* __restrict__ Hd = pb.wHd + (size_t)b * (N + 1) * nn;
  const double* __restrict__ Ho = pb.wHo + (size_t)b * N * nn;
  double* __restrict__ Wl = pb.wWl + (size_t)b * (N + 1) * nn;
  double* __restrict__ Wr = pb.wWr + (size_t)b * (N + 1) * nn;
  double* __restrict__ yv = pb.wy + (size_t)b * (N + 1) * n;
  double* __restrict__ rb = pb.wrb + (size_t)b * (N + 1) * n;
  const double* __restrict__ gv = pb.wg + (size_t)b * (N + 1) * n;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* W1 = smem;                    // [n][n] neighbour factor that multiplies from the left (transposed)
  double* W2 = W1 + nn;                 // [n][n] its other coupling
  double* yn = W2 + nn;                 // [n]    its y
  double* rowbuf = yn + n;              // [2][16 * DENSE_NC] pivot row (double-buffered, padded)
  double* diag = rowbuf + 2 * LVG * DENSE_NC;   // [n]
  const int ty = tid >> LVS, tx = tid & (LVG - 1);
  // a block is first touched at level 1 (odd) or 2 (even): LM damping and -g enter there
  const bool first = (h == 1) || (h == 2 && !(j & 1));
  const double lam = (first && P.opt_type == GPMP2MI_OPT_LM) ? pb.lambda[b] : 0.0;
  const bool want_l = elim && !final && j - h >= 0, want_r = elim && !final && j + h <= N;
  double reg[DENSE_NR][DENSE_NC];
#pragma unroll
  for (int a = 0; a < DENSE_NR; a++)
#pragma unroll
    for (int q = 0; q < DENSE_NC; q++) {
      // one unconditional load per entry from a selected (always valid) address: loads behind data-dependent
      // branches would be waited for one by one
      const int r = ty + LVG * a, c = tx + LVG * q;
      const bool in = r < n && c < AW;
      const int rc = min(r, n - 1);
      const double* src = gv;
      bool use = false;
      double sign = 1.0, add = 0.0;
      if (c < n) { src = Hd + (size_t)j * nn + rc * n + c; use = true; add = (r == c) ? lam : 0.0; }
      else if (c == n) { src = (first ? gv : rb) + (size_t)j * n + rc; use = true; sign = first ? -1.0 : 1.0; }
      else if (h == 1 && c <= 2 * n) { if (want_l) { src = Ho + (size_t)(j - 1) * nn + rc * n + (c - n - 1); use = true; } }   // H_{j,j-1} = block (j, j-1)
      else if (h == 1 && c < AW) { if (want_r) { src = Ho + (size_t)j * nn + (c - 2 * n - 1) * n + rc; use = true; } }          // H_{j,j+1} = block (j+1, j)^T
      const double x = *src;
      reg[a][q] = (use && in) ? fma(sign, x, add) : 0.0;
    }
  // Schur complements of the two neighbours eliminated one level below
  for (int side = 0; side < 2 && h > 1; side++) {
    const int jn = side ? j + hh : j - hh;
    if (jn < 0 || jn > N) continue;
    // left neighbour: its W_r couples to j (W1), its W_l to j - h (W2); right neighbour: W_l to j, W_r to j + h
    const double* __restrict__ g1 = (side ? Wl : Wr) + (size_t)jn * nn;
    const double* __restrict__ g2 = (side ? Wr : Wl) + (size_t)jn * nn;
    const bool want_c = side ? want_r : want_l;
    constexpr int NLD = (36 * 36 + LVL_THREADS - 1) / LVL_THREADS;
    double t1[NLD], t2[NLD];
#pragma unroll
    for (int m = 0; m < NLD; m++) {
      const int e = tid + LVL_THREADS * m;
      t1[m] = g1[min(e, nn - 1)];
      t2[m] = want_c ? g2[min(e, nn - 1)] : 0.0;
    }
    const double ty_ = yv[(size_t)jn * n + min(tid, n - 1)];
    __syncthreads();   // the previous side's products are done with W1 / W2
#pragma unroll
    for (int m = 0; m < NLD; m++) {
      const int e = tid + LVL_THREADS * m;
      if (e < nn) { W1[e] = t1[m]; W2[e] = t2[m]; }
    }
    if (tid < n) yn[tid] = ty_;
    __syncthreads();
    // column source of every owned column: S part -> W1 column, rhs -> y, this side's coupling -> W2 column
    const double* src[DENSE_NC];
    int stride[DENSE_NC];
    bool on[DENSE_NC];
    const int coff = side ? 2 * n + 1 : n + 1;
#pragma unroll
    for (int q = 0; q < DENSE_NC; q++) {
      const int c = tx + LVG * q;
      on[q] = true;
      if (c < n) { src[q] = W1 + c; stride[q] = n; }
      else if (c == n) { src[q] = yn; stride[q] = 1; }
      else if (want_c && c >= coff && c < coff + n) { src[q] = W2 + (c - coff); stride[q] = n; }
      else { src[q] = yn; stride[q] = 0; on[q] = false; }
    }
    int rr[DENSE_NR];
#pragma unroll
    for (int a = 0; a < DENSE_NR; a++) rr[a] = min(ty + LVG * a, n - 1);
    double acc[DENSE_NR][DENSE_NC];
#pragma unroll
    for (int a = 0; a < DENSE_NR; a++)
#pragma unroll
      for (int q = 0; q < DENSE_NC; q++) acc[a][q] = 0.0;
#pragma unroll 2
    for (int k = 0; k < n; k++) {
      double w[DENSE_NR], v[DENSE_NC];
#pragma unroll
      for (int a = 0; a < DENSE_NR; a++) w[a] = W1[k * n + rr[a]];
#pragma unroll
      for (int q = 0; q < DENSE_NC; q++) v[q] = src[q][k * stride[q]];
#pragma unroll
      for (int a = 0; a < DENSE_NR; a++)
#pragma unroll
        for (int q = 0; q < DENSE_NC; q++) acc[a][q] = fma(w[a], v[q], acc[a][q]);
    }
#pragma unroll
    for (int a = 0; a < DENSE_NR; a++)
#pragma unroll
      for (int q = 0; q < DENSE_NC; q++)
        if (on[q] && ty + LVG * a < n) reg[a][q] -= acc[a][q];
  }
  if (!elim) {
#pragma unroll
    for (int a = 0; a < DENSE_NR; a++)
#pragma unroll
      for (int q = 0; q < DENSE_NC; q++) {
        const int r = ty + LVG * a, c = tx + LVG * q;
        if (r < n && c < n) Hd[(size_t)j * nn + r * n + c] = reg[a][q];
        if (r < n && c == n) rb[(size_t)j * n + r] = reg[a][q];
      }
    return;
  }
  // elimination: the owners of row k publish it, everybody reads the pivot, its own columns of the row and the
  // multipliers of its own rows (A[k][r], upper triangle) from that copy; entries below the diagonal of S are
  // never read again, so they are updated along without a test
  constexpr int RB = LVG * DENSE_NC;   // padded row buffer: columns >= AW hold don't-care values
  if (ty == 0) {
#pragma unroll
    for (int q = 0; q < DENSE_NC; q++) rowbuf[tx + LVG * q] = reg[0][q];
  }
  __syncthreads();
  for (int k = 0; k < n; k++) {
    const double* row = rowbuf + (k & 1) * RB;
    const double piv = row[k];
    if (!(piv > 0.0)) {  // uniform: every thread reads the same LDS value
      if (tid == 0) pb.notspd[b] = 1;  // k_decide consumes and clears it
      return;
    }
    const double ipiv = 1.0 / piv;
    double rk[DENSE_NC], mk[DENSE_NR];
#pragma unroll
    for (int q = 0; q < DENSE_NC; q++) rk[q] = row[tx + LVG * q];
#pragma unroll
    for (int a = 0; a < DENSE_NR; a++) mk[a] = row[min(ty + LVG * a, n - 1)];
    double* nxt = rowbuf + ((k + 1) & 1) * RB;
#pragma unroll
    for (int a = 0; a < DENSE_NR; a++) {
      const int r = ty + LVG * a;
      if (r > k && r < n) {
        const double m = mk[a] * ipiv;
#pragma unroll
        for (int q = 0; q < DENSE_NC; q++) reg[a][q] = fma(-m, rk[q], reg[a][q]);
        if (r == k + 1) {
#pragma unroll
          for (int q = 0; q < DENSE_NC; q++) nxt[tx + LVG * q] = reg[a][q];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < DENSE_NR; a++)
#pragma unroll
    for (int q = 0; q < DENSE_NC; q++)
      if (ty + LVG * a < n && ty + LVG * a == tx + LVG * q) diag[ty + LVG * a] = reg[a][q];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < DENSE_NR; a++) {
    const int r = ty + LVG * a;
    if (r >= n) continue;
    const double isq = 1.0 / sqrt(diag[r]);
#pragma unroll
    for (int q = 0; q < DENSE_NC; q++) {
      const int c = tx + LVG * q;
      const double v = reg[a][q] * isq;
      if (c < n) Hd[(size_t)j * nn + r * n + c] = (c >= r) ? v : 0.0;
      else if (c == n) yv[(size_t)j * n + r] = v;
      else if (c <= 2 * n) Wl[(size_t)j * nn + r * n + (c - n - 1)] = v;
      else if (c < AW) Wr[(size_t)j * nn + r * n + (c - 2 * n - 1)] = v;
    }
  }
}

// one back-substitution level: one wavefront per block, R_j staged in LDS, lane r carries t_r
__global__ __launch_bounds__(64) void k_dense_cr_back(const PlanParams* __restrict__ pp, PlanBuffers pb, int h, int final) {
  const PlanParams& P = *pp;
  const int N = P.N, n = P.n, lane = threadIdx.x;
  const int count = final ? 1 : ((N / h) + 1) / 2;
  const int b = blockIdx.x / count, idx = blockIdx.x - b * count;
  if (!pb.active[b] || pb.notspd[b]) return;
  if (P.opt_type == GPMP2MI_OPT_DOGLEG && pb.phase[b] != 0) return;
  const int j = final ? 0 : h * (2 * idx + 1), nn = n * n;
  const double* R = pb.wHd + ((size_t)b * (N + 1) + j) * nn;
  const double* Wl = pb.wWl + ((size_t)b * (N + 1) + j) * nn;
  const double* Wr = pb.wWr + ((size_t)b * (N + 1) + j) * nn;
  double* x = pb.wx + (size_t)b * (N + 1) * n;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* M = smem;          // [n][n] W_l, W_r, then R
  double* xn = M + nn;       // [n] neighbour solution / the unknown being broadcast
  constexpr int NLDB = (36 * 36 + 63) / 64;   // 21 loads per lane in flight (n <= 36)
  const bool has_l = !final && j - h >= 0, has_r = !final && j + h <= N;
  double t = (lane < n) ? pb.wy[((size_t)b * (N + 1) + j) * n + lane] : 0.0;
  for (int side = 0; side < 2; side++) {
    if (!(side ? has_r : has_l)) continue;
    const double* Wg = side ? Wr : Wl;
    const int jn = side ? j + h : j - h;
    double tm[NLDB];
#pragma unroll
    for (int m = 0; m < NLDB; m++) tm[m] = (lane + 64 * m < nn) ? Wg[lane + 64 * m] : 0.0;
    const double xv = (lane < n) ? x[(size_t)jn * n + lane] : 0.0;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NLDB; m++)
      if (lane + 64 * m < nn) M[lane + 64 * m] = tm[m];
    if (lane < n) xn[lane] = xv;
    __syncthreads();
    if (lane < n)
      for (int c = 0; c < n; c++) t = fma(-M[lane * n + c], xn[c], t);
  }
  {
    double tm[NLDB];
#pragma unroll
    for (int m = 0; m < NLDB; m++) tm[m] = (lane + 64 * m < nn) ? R[lane + 64 * m] : 0.0;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NLDB; m++)
      if (lane + 64 * m < nn) M[lane + 64 * m] = tm[m];
  }
  __syncthreads();
  for (int k = n - 1; k >= 0; k--) {
    if (lane == k) xn[k] = t / M[k * n + k];
    __syncthreads();
    if (lane < k) t = fma(-M[lane * n + k], xn[k], t);
  }
  if (lane < n) x[(size_t)j * n + lane] = xn[lane];
}

// step, step-control sums and the trial point from the solution the back-substitution levels left in wx
__global__ __launch_bounds__(DENSE_THREADS) void k_dense_tail(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  const PlanParams& P = *pp;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!pb.active[b]) return;
  const int N = P.N, n = P.n, D = P.D;
  const size_t tsz = (size_t)(N + 1) * n;
  const double* cur = pb.cur + b * tsz;
  double* trial = pb.trial + b * tsz;
  double* delta = pb.delta + b * tsz;
  double* sc = pb.scal + (size_t)b * SC_COUNT;
  const double* gv = pb.wg + (size_t)b * tsz;
  const double* xg = pb.wx + (size_t)b * tsz;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;                       // [(N+1)][n]
  double* red = xs + tsz;                  // [DENSE_WAVES]
  const bool dogleg = P.opt_type == GPMP2MI_OPT_DOGLEG;
  const bool resolve = !(dogleg && pb.phase[b] != 0);
  if (resolve) {
    if (pb.notspd[b]) return;              // bad pivot somewhere in the tree: k_decide consumes and clears the flag
    double gd = 0.0, dd = 0.0, gg = 0.0;
    for (size_t k = tid; k < tsz; k += DENSE_THREADS) {
      const double x = xg[k], gk = gv[k];
      xs[k] = x;
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

// cyclic-reduction form of the dense solve: g^T H g (Dogleg), forward levels, backward levels, tail
int launch_solve_dense(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const int n = hp.n, N = hp.N;
  const size_t sh_level = (2 * (size_t)n * n + n + 2 * LVG * DENSE_NC + n) * sizeof(double);
  if (n > 36) {
    set_error("dense block solver: blocks wider than 36 are not instantiated");
    return GPMP2MI_ERR_UNSUPPORTED;
  }
  const size_t sh_back = ((size_t)n * n + n) * sizeof(double);
  const size_t sh_tail = ((size_t)(N + 1) * n + DENSE_WAVES) * sizeof(double);
  if (sh_level > 150 * 1024 || sh_tail > 150 * 1024) {
    set_error("block / trajectory too large for the LDS-resident dense solve");
    return GPMP2MI_ERR_UNSUPPORTED;
  }
  int hfinal = 1;
  while (hfinal <= N) hfinal <<= 1;
  if (hp.opt_type == GPMP2MI_OPT_DOGLEG) k_dense_ghg<<<dim3(hp.B), dim3(DENSE_THREADS), 0, st>>>(pb.params, pb);
  for (int h = 1; h <= hfinal; h <<= 1) {
    const int final = h == hfinal;
    const int countE = final ? 1 : ((N / h) + 1) / 2, countU = (final || h == 1) ? 0 : (N / (2 * h)) + 1;
    k_dense_cr_level<<<dim3(hp.B * (countE + countU)), dim3(LVL_THREADS), sh_level, st>>>(pb.params, pb, h, final);
  }
  for (int h = hfinal; h >= 1; h >>= 1) {
    const int final = h == hfinal;
    const int count = final ? 1 : ((N / h) + 1) / 2;
    k_dense_cr_back<<<dim3(hp.B * count), dim3(64), sh_back, st>>>(pb.params, pb, h, final);
  }
  k_dense_tail<<<dim3(hp.B), dim3(DENSE_THREADS), sh_tail, st>>>(pb.params, pb);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

}  // namespace g2
