// wide_cr.h -- cyclic reduction with blocks wider than one tile (8 <= dof <= 11, n = 2 dof <= 22).
//
// Included at the end of cr_kernels.hip (inside namespace g2).  A block is a 32 x 32 matrix held as
// 2 x 2 MFMA-layout tiles (WTile); element (row, col) lives in tile (row >> 4, col >> 4).  The right-hand
// side rides in column 31.  Everything mirrors the one-tile kernels: k_assemble_wide forms the block of
// one support state (four Assembler passes with tile offsets; the interpolated factors of Pose2 robots as
// E^T G E congruences on the matrix cores) and eliminates the odd blocks (level 1); k_cr_level_wide runs forward
// levels 2 and 4 chip-wide; k_solve_step_wide runs the levels above, the back-substitution down to the multiples
// of 8 and (Dogleg) the whole trial-step tail; k_finish_trial_wide finishes the back-substitution, the step and the
// trial point chip-wide for GN / LM (GN runs through the trial-step driver).
#pragma once

struct WTile {
  Tile t[2][2];
};
constexpr int WTILE_DBL = 4 * TILE_DBL;  // 1024 doubles
constexpr int WRHS = 31;                 // column that carries the right-hand side
constexpr int WX = 32;                   // stride of a block's solution in LDS / gvec

__device__ __forceinline__ WTile wtile_load(const double* p, int lane) {
  WTile W;
#pragma unroll
  for (int q = 0; q < 4; q++) W.t[q >> 1][q & 1] = tile_load(p + q * TILE_DBL, lane);
  return W;
}
__device__ __forceinline__ void wtile_store(double* p, const WTile& W, int lane) {
#pragma unroll
  for (int q = 0; q < 4; q++) tile_store(p + q * TILE_DBL, W.t[q >> 1][q & 1], lane);
}
// the same for blocks whose rows >= n are structurally zero (diagonal and factor blocks of the reduction): the
// padding rows are neither written nor read
template <int n>
__device__ __forceinline__ WTile wtile_load_rows(const double* p, int lane) {
  WTile W;
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (16 * (q >> 1) + 4 * k >= n) { W.t[q >> 1][q & 1].r[k] = 0.0; continue; }
      W.t[q >> 1][q & 1].r[k] = (16 * (q >> 1) + (lane >> 4) + 4 * k < n) ? p[q * TILE_DBL + k * 64 + lane] : 0.0;
    }
  return W;
}
template <int n>
__device__ __forceinline__ void wtile_store_rows(double* p, const WTile& W, int lane) {
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (16 * (q >> 1) + 4 * k >= n) continue;
      if (16 * (q >> 1) + (lane >> 4) + 4 * k < n) p[q * TILE_DBL + k * 64 + lane] = W.t[q >> 1][q & 1].r[k];
    }
}
__device__ __forceinline__ WTile wtile_zero() {
  WTile W;
#pragma unroll
  for (int q = 0; q < 4; q++) W.t[q >> 1][q & 1] = tile_zero();
  return W;
}
__device__ __forceinline__ WTile wtile_identity(int lane) {
  const int c = lane & 15, g = lane >> 4;
  WTile W = wtile_zero();
#pragma unroll
  for (int k = 0; k < 4; k++) W.t[0][0].r[k] = W.t[1][1].r[k] = (g + 4 * k == c) ? 1.0 : 0.0;
  return W;
}

// (A^T B)(i, j) = sum_k A(k, i)^T B(k, j) over the two tile rows
// (nrows: rows >= nrows of both operands are zero, their k-chunks are skipped)
template <int nrows = 32>
__device__ __forceinline__ Tile wtile_atb_ij(const WTile& A, const WTile& B, int i, int j) {
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kt = 0; kt < 2; kt++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (16 * kt + 4 * k >= nrows) continue;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A.t[kt][i].r[k], B.t[kt][j].r[k], acc, 0, 0, 0);
    }
  Tile T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = acc[k];
  return T;
}

// S -= A^T A restricted to real rows (< n) and to the matrix + rhs columns
template <int n>
__device__ __forceinline__ void wschur_sub(WTile& S, const WTile& A, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      if (16 * ti >= n) continue;
      const Tile T = wtile_atb_ij<n>(A, A, ti, tj);
      const int col = 16 * tj + c;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if ((16 * ti + g + 4 * k) < n && (col < n || col == WRHS)) S.t[ti][tj].r[k] -= T.r[k];
    }
}
// -(A^T B) restricted to the n x n matrix part
template <int n>
__device__ __forceinline__ WTile wcoupling(const WTile& A, const WTile& B, int lane) {
  const int c = lane & 15, g = lane >> 4;
  WTile C = wtile_zero();
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      if (16 * ti >= n || 16 * tj >= n) continue;
      const Tile T = wtile_atb_ij<n>(A, B, ti, tj);
#pragma unroll
      for (int k = 0; k < 4; k++)
        C.t[ti][tj].r[k] = ((16 * ti + g + 4 * k) < n && (16 * tj + c) < n) ? -T.r[k] : 0.0;
    }
  return C;
}

// Eliminate the n pivots of S = [S | b] (rhs in column WRHS) and apply the row operations to the two coupling
// blocks and to V (identity on entry); same contract as tile_eliminate3.
template <int n>
__device__ __forceinline__ bool wtile_eliminate3(WTile& S, WTile& Cl, WTile& Cr, WTile& V, int lane) {
  const int c = lane & 15, g = lane >> 4;
  double piv_of_row[2][4];
#pragma unroll
  for (int q = 0; q < 8; q++) piv_of_row[q >> 2][q & 3] = 1.0;
  bool ok = true;
  static_for<0, n>([&](auto jc) {
    constexpr int j = decltype(jc)::value, tp = j >> 4, jl = j & 15, gj = jl & 3, rj = jl >> 2;
    const int src = gj * 16 + c;
    double rowS[2], rowL[2], rowR[2], rowV[2];
#pragma unroll
    for (int tc = 0; tc < 2; tc++) {
      rowS[tc] = __shfl(S.t[tp][tc].r[rj], src, 64);
      rowL[tc] = __shfl(Cl.t[tp][tc].r[rj], src, 64);
      rowR[tc] = __shfl(Cr.t[tp][tc].r[rj], src, 64);
      rowV[tc] = __shfl(V.t[tp][tc].r[rj], src, 64);
    }
    const double piv = readlane_d(S.t[tp][tp].r[rj], gj * 16 + jl);
    ok = ok && (piv > 0.0);
    const double inv = fast_rcp(piv);
    if (g == gj) piv_of_row[tp][rj] = piv;
#pragma unroll
    for (int tr = tp; tr < 2; tr++) {
      if (16 * tr >= n) continue;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (tr == tp && 4 * k + 3 <= jl) continue;  // rows of this register are all at or above the pivot
        if (16 * tr + 4 * k >= n) continue;         // ... or all padding
        const double m = bcast_in_row<jl>(S.t[tr][tp].r[k]);
        const double f = (16 * tr + g + 4 * k > j) ? m * inv : 0.0;
#pragma unroll
        for (int tc = 0; tc < 2; tc++) {
          S.t[tr][tc].r[k] = fma(-f, rowS[tc], S.t[tr][tc].r[k]);
          Cl.t[tr][tc].r[k] = fma(-f, rowL[tc], Cl.t[tr][tc].r[k]);
          Cr.t[tr][tc].r[k] = fma(-f, rowR[tc], Cr.t[tr][tc].r[k]);
          V.t[tr][tc].r[k] = fma(-f, rowV[tc], V.t[tr][tc].r[k]);
        }
      }
    }
  });
#pragma unroll
  for (int tr = 0; tr < 2; tr++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (16 * tr + 4 * k >= n) continue;
      const double s = 1.0 / sqrt(piv_of_row[tr][k]);
      const double y = S.t[tr][1].r[k] * s;  // meaningful in the rhs column only
#pragma unroll
      for (int tc = 0; tc < 2; tc++) {
        const bool rhs = (tc == 1 && c == 15);
        Cl.t[tr][tc].r[k] = rhs ? y : Cl.t[tr][tc].r[k] * s;
        Cr.t[tr][tc].r[k] = rhs ? y : Cr.t[tr][tc].r[k] * s;
        V.t[tr][tc].r[k] *= s;
      }
    }
  return ok;
}

// ---- column form (tiles.h "CR elimination, column form") on 2 x 2 tiles.
// Column operations on the stacked pair [S ; Vt]: per pivot ONE row of S is moved across the row groups (two tile
// columns: 4 ds_bpermute instead of 16), and every register that holds a row at or below the pivot (S) or at or above
// it (Vt, upper triangular) gets one DPP-fused multiply-add per tile column right of the pivot.
template <int jl>
__device__ __forceinline__ void fmac_col2(double& d0, double& d1, double nf0, double nf1) {
  // d1 += bcast_jl(d0) * nf1 ; d0 += bcast_jl(d0) * nf0   (column jl of d0 itself is not changed: nf0 is zero there)
  asm volatile("s_nop 1\n\t"
               "v_fmac_f64_dpp %1, %0, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
               : "+v"(d0), "+v"(d1) : "v"(nf0), "v"(nf1), "n"(jl));
}
template <int jl>
__device__ __forceinline__ void fmac_col1(double& d, double nf) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(nf), "n"(jl));
}

template <int n>
__device__ __forceinline__ bool wtile_eliminate_col(WTile& S, WTile& Vt, int lane) {
  const int c = lane & 15;
  static_for<0, n>([&](auto jc) {
    constexpr int j = decltype(jc)::value, tp = j >> 4, jl = j & 15, gj = jl & 3, rj = jl >> 2;
    const int src = gj * 16 + c;
    const double piv = readlane_d(S.t[tp][tp].r[rj], gj * 16 + jl);
    const double ninv = -fast_rcp(piv);
    // -f_c for the tile columns that hold columns right of the pivot
    double nf[2] = {0.0, 0.0};
#pragma unroll
    for (int tc = tp; tc < 2; tc++) {
      if (16 * tc >= n) continue;
      const double rowS = __shfl(S.t[tp][tc].r[rj], src, 64);
      const int col = 16 * tc + c;
      nf[tc] = (col > j && col < n) ? rowS * ninv : 0.0;
    }
    static_for<0, 8>([&](auto qc) {
      constexpr int tr = decltype(qc)::value >> 2, k = decltype(qc)::value & 3;
      if constexpr (16 * tr + 4 * k < n) {
        // S: rows at or below the pivot's register; Vt: rows at or above it
        constexpr bool in_s = (tr > tp) || (tr == tp && k >= rj);
        constexpr bool in_v = (tr < tp) || (tr == tp && k <= rj);
        if constexpr (tp == 0 && n > 16) {
          if constexpr (in_s) fmac_col2<jl>(S.t[tr][0].r[k], S.t[tr][1].r[k], nf[0], nf[1]);
          if constexpr (in_v) fmac_col2<jl>(Vt.t[tr][0].r[k], Vt.t[tr][1].r[k], nf[0], nf[1]);
        } else if constexpr (tp == 0) {
          if constexpr (in_s) fmac_col1<jl>(S.t[tr][0].r[k], nf[0]);
          if constexpr (in_v) fmac_col1<jl>(Vt.t[tr][0].r[k], nf[0]);
        } else {   // pivot in the second tile column: only that column has entries right of it
          if constexpr (in_s) fmac_col1<jl>(S.t[tr][1].r[k], nf[1]);
          if constexpr (in_v) fmac_col1<jl>(Vt.t[tr][1].r[k], nf[1]);
        }
      }
    });
  });
  // pivots = diagonal of S: column 16 tc + c needs S[col][col], held by row group c & 3, register c >> 2 of tile (tc, tc)
  bool ok = true;
#pragma unroll
  for (int tc = 0; tc < 2; tc++) {
    if (16 * tc >= n) continue;
    double pv = 1.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (16 * tc + 4 * k >= n) continue;
      const double dgn = __shfl(S.t[tc][tc].r[k], (c & 3) * 16 + c, 64);
      pv = ((c >> 2) == k && 16 * tc + c < n) ? dgn : pv;
    }
    ok = ok && __all(pv > 0.0);
    const double rs = fast_rsqrt(pv);
#pragma unroll
    for (int tr = 0; tr <= tc; tr++)     // Vt is upper triangular: tile (1, 0) stays zero
#pragma unroll
      for (int k = 0; k < 4; k++) Vt.t[tr][tc].r[k] *= rs;
  }
  return ok;
}

// One elimination task on 2 x 2 tiles; same contract as tile_eliminate_cv: Cl <- W_l, Cr <- W_r (y in column WRHS),
// Vt = R^-1.
template <int n>
__device__ __forceinline__ bool wtile_eliminate_cv(WTile& S, WTile& Cl, WTile& Cr, WTile& Vt, int lane) {
  const int c = lane & 15, g = lane >> 4;
  Vt = wtile_zero();
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      Vt.t[ti][ti].r[k] = (g + 4 * k == c && 16 * ti + c < n) ? 1.0 : 0.0;
      Cl.t[ti][1].r[k] = (c == 15) ? S.t[ti][1].r[k] : Cl.t[ti][1].r[k];   // right-hand side: column WRHS = 31
      Cr.t[ti][1].r[k] = (c == 15) ? S.t[ti][1].r[k] : Cr.t[ti][1].r[k];
    }
  const bool ok = wtile_eliminate_col<n>(S, Vt, lane);
  WTile Wl = wtile_zero(), Wr = wtile_zero();
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      if (16 * ti >= n) continue;
      Wl.t[ti][tj] = wtile_atb_ij<n>(Vt, Cl, ti, tj);
      Wr.t[ti][tj] = wtile_atb_ij<n>(Vt, Cr, ti, tj);
    }
  Cl = Wl;
  Cr = Wr;
  return ok;
}

// V = Vt^T to memory (see tile_store_transposed): V(16 tc + c, 16 tr + g + 4 k) = Vt(16 tr + g + 4 k, 16 tc + c)
template <int n>
__device__ __forceinline__ void wtile_store_transposed(double* __restrict__ p, const WTile& Vt, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int tr = 0; tr < 2; tr++)
#pragma unroll
    for (int tc = 0; tc < 2; tc++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (16 * tr + 4 * k >= n || 16 * tc >= n) continue;
        if (16 * tc + c < n && 16 * tr + g + 4 * k < n) p[(2 * tc + tr) * TILE_DBL + c * 16 + g + 4 * k] = Vt.t[tr][tc].r[k];
      }
}

// x_j = V^T (y - Wl x_l - Wr x_r); xl[tc] / xr[tc] = neighbour solutions at column 16 tc + c; returns x[tc]
template <int n>
__device__ __forceinline__ void wcr_backsolve(const WTile& Wl, const WTile& Wr, const WTile& V, const double (&xl)[2],
                                              const double (&xr)[2], int lane, double (&x)[2]) {
  const int c = lane & 15;
  double t[2][4];
#pragma unroll
  for (int tr = 0; tr < 2; tr++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double acc = 0.0;
      if (16 * tr + 4 * k >= n) { t[tr][k] = 0.0; continue; }
#pragma unroll
      for (int tc = 0; tc < 2; tc++) {
        const int col = 16 * tc + c;
        double v;
        if (col == WRHS) v = -Wl.t[tr][tc].r[k];  // -y (same in both blocks)
        else v = (col < n) ? fma(Wl.t[tr][tc].r[k], xl[tc], Wr.t[tr][tc].r[k] * xr[tc]) : 0.0;
        acc += row_sum16(v);
      }
      t[tr][k] = -acc;
    }
#pragma unroll
  for (int tc = 0; tc < 2; tc++) {
    double a = 0.0;
#pragma unroll
    for (int tr = 0; tr < 2; tr++)
#pragma unroll
      for (int k = 0; k < 4; k++) a = fma(V.t[tr][tc].r[k], t[tr][k], a);
    x[tc] = sum_rows(a);
  }
}

// Interpolated obstacle factors of a Pose2 robot for all four tiles of a wide block at once (the one-tile form
// Assembler::lie_interp would rebuild G and the E tiles for every output tile): per point one G tile (dof <= 16),
// the E tiles of both column halves, G E once per half, then the 2 x 2 outer products E_L^T (G E_R).
template <int D, bool LIE>
__device__ __forceinline__ void wide_lie_interp(const Assembler<D, LIE>& as, const typename Assembler<D, LIE>::Slot& si,
                                                const typename Assembler<D, LIE>::Slot& sn, bool has_prev, bool has_next,
                                                bool want_c, WTile& S, WTile& Cl, WTile& Cr) {
  using Asm = Assembler<D, LIE>;
  static_assert(D <= 16, "one tile along the configuration dimension");
  v4d aS[2][2], aL[2][2], aR[2][2];
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      aS[q >> 1][q & 1][k] = S.t[q >> 1][q & 1].r[k];
      aL[q >> 1][q & 1][k] = Cl.t[q >> 1][q & 1].r[k];
      aR[q >> 1][q & 1][k] = Cr.t[q >> 1][q & 1].r[k];
    }
  auto ge_of = [&](const Tile& G, const Tile& E) {
    v4d ge = {0.0, 0.0, 0.0, 0.0};
    Asm::mfma_atb_acc(G, E, 0, ge);
    Tile T;
#pragma unroll
    for (int k = 0; k < 4; k++) T.r[k] = ge[k];
    return T;
  };
  auto point = [&](const double* pt, const double* cf, int role, v4d (&accD)[2][2], v4d (&accC)[2][2]) {
    const double* M = pt + Asm::RECP;
    const Tile G = as.g_tile(pt, 0, 0);
    Tile Er[2], GE[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      Er[h] = as.hint_tile(M, cf, role, 0, 16 * h);
      GE[h] = ge_of(G, Er[h]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) Asm::mfma_atb_acc(Er[q >> 1], GE[q & 1], 0, accD[q >> 1][q & 1]);
    if (want_c) {
#pragma unroll
      for (int h = 0; h < 2; h++) GE[h] = ge_of(G, as.hint_tile(M, cf, 1 - role, 0, 16 * h));
#pragma unroll
      for (int q = 0; q < 4; q++) Asm::mfma_atb_acc(Er[q >> 1], GE[q & 1], 0, accC[q >> 1][q & 1]);
    }
  };
#pragma unroll 1
  for (int jj = 0; jj < as.P.I; jj++) {
    const double* cf = si.coef(jj) + 16;
    if (has_prev) point(si.pt(jj), cf, 1, aS, aL);
    if (has_next) point(sn.pt(jj), cf, 0, aS, aR);
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      S.t[q >> 1][q & 1].r[k] = aS[q >> 1][q & 1][k];
      Cl.t[q >> 1][q & 1].r[k] = aL[q >> 1][q & 1][k];
      Cr.t[q >> 1][q & 1].r[k] = aR[q >> 1][q & 1][k];
    }
}

// =============================================================================== assemble (wide)
template <int D, bool LIE>
__global__ __launch_bounds__(64, 2) void k_assemble_wide(const PlanParams* __restrict__ pp, PlanBuffers pb,
                                                       const double* __restrict__ traj, int bufsel,
                                                       const int* __restrict__ active) {
  constexpr int n = 2 * D;
  using Asm = Assembler<D, LIE>;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int b = blockIdx.x / (N + 1), i = blockIdx.x - b * (N + 1);
  if (active && !active[b]) return;
  if (P.opt_type == GPMP2MI_OPT_DOGLEG && active && pb.phase[b] != 0) return;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  extern __shared__ __attribute__((aligned(16))) double asm_smem[];
  const double* rec = rec_of(pb, pb.which[b], bufsel);
  const double* gpu = gpu_of(pb, pb.which[b], bufsel);
  Asm as(P, pb, rec, gpu, b, lane);
  const typename Asm::Slot slot0 = as.make_slot(asm_smem, 0), slot1 = as.make_slot(asm_smem, 1);
  as.stage2(i, slot0, slot1);
  __syncthreads();
  const bool odd = (i & 1) != 0;
  const bool want_c = odd || P.opt_type == GPMP2MI_OPT_DOGLEG;
  const double* zi = traj + ((size_t)b * (N + 1) + i) * n;
  WTile S, Cl, Cr;
  static_for<0, 4>([&](auto qc) {  // forced unrolling: the four tiles must stay in registers
    constexpr int ti = decltype(qc)::value >> 1, tj = decltype(qc)::value & 1;
    Asm at(P, pb, rec, gpu, b, lane, 16 * ti, 16 * tj, WRHS);
    at.skip_interp = LIE;
    at.build_tiles(i, slot0, slot1, zi, S.t[ti][tj], Cl.t[ti][tj], Cr.t[ti][tj], want_c);
  });
  if constexpr (LIE) wide_lie_interp<D, LIE>(as, slot0, slot1, i > 0, i < N, want_c, S, Cl, Cr);
  // gradient g_i (the rhs column holds -g_i)
  if (c == 15) {
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
      for (int k = 0; k < 4; k++) pb.gvec[((size_t)b * (N + 1) + i) * WX + 16 * ti + g + 4 * k] = -S.t[ti][1].r[k];
  }
  if (P.opt_type == GPMP2MI_OPT_DOGLEG) {  // un-eliminated blocks for g^T H g
    double* ht = pb.htiles + ((size_t)b * (N + 1) + i) * 2 * WTILE_DBL;
    wtile_store(ht, S, lane);
    wtile_store(ht + WTILE_DBL, Cr, lane);
  }
  if (P.opt_type == GPMP2MI_OPT_LM) {
    const double lam = pb.lambda[b];
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (g + 4 * k == c && 16 * ti + c < n) S.t[ti][ti].r[k] += lam;
  }
  if (!odd) {
    wtile_store_rows<n>(pb.tiles + ((size_t)b * (N + 1) + i) * WTILE_DBL, S, lane);
  } else {
    WTile V;   // Vt = R^-1; stored transposed, as V
    const bool ok = wtile_eliminate_cv<n>(S, Cl, Cr, V, lane);
    double* f = pb.fac + ((size_t)b * (N + 1) + i) * 3 * WTILE_DBL;
    wtile_store_rows<n>(f, Cl, lane);
    wtile_store_rows<n>(f + WTILE_DBL, Cr, lane);
    wtile_store_transposed<n>(f + 2 * WTILE_DBL, V, lane);
    if (!ok && lane == 0) pb.notspd[b] = 1;
  }
}

int launch_assemble_wide(const PlanParams& hp, const PlanBuffers& pb, const double* traj, int bufsel,
                         const int* active, hipStream_t st) {
  const dim3 grid(hp.B * (hp.N + 1)), block(64);
  const size_t shmem = 2 * (size_t)((hp.I + 1) * hp.RECS + hp.GPS + 24 * hp.I) * sizeof(double);
  switch (hp.D) {
#define G2_ASMW_CASE(DD) \
  case DD:                                                                                      \
    if (hp.lie) k_assemble_wide<DD, true><<<grid, block, shmem, st>>>(pb.params, pb, traj, bufsel, active); \
    else k_assemble_wide<DD, false><<<grid, block, shmem, st>>>(pb.params, pb, traj, bufsel, active);       \
    break;
    G2_ASMW_CASE(8) G2_ASMW_CASE(9) G2_ASMW_CASE(10) G2_ASMW_CASE(11)
#undef G2_ASMW_CASE
    default:
      set_error("wide blocks are instantiated for 8 <= dof <= 11");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// g^T H g share of block i from the blocks saved by k_assemble_wide (Dogleg)
template <int D>
__global__ __launch_bounds__(64) void k_ghg_wide(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int b = blockIdx.x / (N + 1), i = blockIdx.x - b * (N + 1);
  if (!pb.active[b] || pb.phase[b] != 0) return;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const double* ht = pb.htiles + ((size_t)b * (N + 1) + i) * 2 * WTILE_DBL;
  const WTile Dt = wtile_load(ht, lane), Ht = wtile_load(ht + WTILE_DBL, lane);
  const double* gv = pb.gvec + ((size_t)b * (N + 1) + i) * WX;
  double acc = 0.0;
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      const int col = 16 * tj + c;
      const double gi_c = (col < n) ? gv[col] : 0.0;
      const double gn_c = (col < n && i < N) ? gv[WX + col] : 0.0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int rho = 16 * ti + g + 4 * k;
        const double gi_r = (rho < n) ? gv[rho] : 0.0;
        if (col < n) acc += gi_r * (Dt.t[ti][tj].r[k] * gi_c + 2.0 * Ht.t[ti][tj].r[k] * gn_c);
      }
    }
  acc = wave_sum(acc);
  if (lane == 0) pb.hgpart[(size_t)b * P.Npad + i] = acc;
}

int launch_ghg_wide(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const dim3 grid(hp.B * (hp.N + 1)), block(64);
  switch (hp.D) {
#define G2_GHGW_CASE(DD) \
  case DD: k_ghg_wide<DD><<<grid, block, 0, st>>>(pb.params, pb); break;
    G2_GHGW_CASE(8) G2_GHGW_CASE(9) G2_GHGW_CASE(10) G2_GHGW_CASE(11)
#undef G2_GHGW_CASE
    default:
      set_error("wide blocks are instantiated for 8 <= dof <= 11");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

// =============================================================================== solve step (wide)
constexpr int WCR_WAVES = 8;  // 16 tiles of state per elimination: 2 wavefronts per SIMD keep 256 VGPRs each

// One task of forward level h: idx < countE eliminates the idx-th odd multiple of h (E task), the others bring the
// diagonal block of an even multiple up to date (U task).
template <int n>
__device__ __forceinline__ bool wcr_task(const PlanBuffers& pb, int b, int N, int h, int idx, int countE, bool final,
                                         int lane) {
  double* tiles = pb.tiles + (size_t)b * (N + 1) * WTILE_DBL;
  double* fac = pb.fac + (size_t)b * (N + 1) * 3 * WTILE_DBL;
  const int hh = h >> 1;
  const bool elim = idx < countE;
  const int j = elim ? (final ? 0 : h * (2 * idx + 1)) : 2 * h * (idx - countE);
  WTile S = wtile_load_rows<n>(tiles + (size_t)j * WTILE_DBL, lane);
  WTile Cl = wtile_zero(), Cr = wtile_zero();
  const int jm = j - hh, jp = j + hh;
  if (jm >= 0) {
    const WTile Wr = wtile_load_rows<n>(fac + ((size_t)jm * 3 + 1) * WTILE_DBL, lane);
    wschur_sub<n>(S, Wr, lane);
    if (elim && !final) {
      const WTile Wl = wtile_load_rows<n>(fac + (size_t)jm * 3 * WTILE_DBL, lane);
      Cl = wcoupling<n>(Wr, Wl, lane);  // rows j, cols j - h
    }
  }
  if (jp <= N) {
    const WTile Wl = wtile_load_rows<n>(fac + (size_t)jp * 3 * WTILE_DBL, lane);
    wschur_sub<n>(S, Wl, lane);
    if (elim && !final && j + h <= N) {
      const WTile Wr = wtile_load_rows<n>(fac + ((size_t)jp * 3 + 1) * WTILE_DBL, lane);
      Cr = wcoupling<n>(Wl, Wr, lane);  // rows j, cols j + h
    }
  }
  if (!elim) {
    wtile_store_rows<n>(tiles + (size_t)j * WTILE_DBL, S, lane);
    return true;
  }
  WTile V;
  const bool ok = wtile_eliminate_cv<n>(S, Cl, Cr, V, lane);
  double* f = fac + (size_t)j * 3 * WTILE_DBL;
  wtile_store_rows<n>(f, Cl, lane);
  wtile_store_rows<n>(f + WTILE_DBL, Cr, lane);
  wtile_store_transposed<n>(f + 2 * WTILE_DBL, V, lane);
  return ok;
}

// levels h0 .. of the reduction inside the trajectory's workgroup (levels below h0 ran chip-wide, k_cr_level_wide)
template <int n>
__device__ __forceinline__ bool wcr_forward(const PlanBuffers& pb, int b, int N, int tid, int h0) {
  const int w = tid >> 6, lane = tid & 63;
  bool ok = true;
  int hfinal = 1;
  while (hfinal <= N) hfinal <<= 1;
  for (int h = h0; h <= hfinal; h <<= 1) {
    const bool final = (h == hfinal);
    const int countE = final ? 1 : ((N / h) + 1) / 2;
    const int countU = final ? 0 : (N / (2 * h)) + 1;
    for (int idx = w; idx < countE + countU; idx += WCR_WAVES) ok = wcr_task<n>(pb, b, N, h, idx, countE, final, lane) && ok;
    __syncthreads();
  }
  return ok;
}

// One forward level (never the final one) spread over the chip: one wavefront per task.  The first levels of a
// 100-state trajectory are 51 and 26 tasks -- 7 and 4 rounds of the 8 wavefronts of k_solve_step_wide.
template <int D>
__global__ __launch_bounds__(64, 2) void k_cr_level_wide(const PlanParams* __restrict__ pp, PlanBuffers pb, int h) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int countE = ((N / h) + 1) / 2, countU = (N / (2 * h)) + 1, per = countE + countU;
  const int b = blockIdx.x / per, idx = blockIdx.x - b * per;
  if (!pb.active[b]) return;
  if (P.opt_type == GPMP2MI_OPT_DOGLEG && pb.phase[b] != 0) return;
  const bool ok = wcr_task<n>(pb, b, N, h, idx, countE, false, threadIdx.x);
  if (!ok && threadIdx.x == 0) pb.notspd[b] = 1;
}

int launch_cr_level_wide(const PlanParams& hp, const PlanBuffers& pb, int h, hipStream_t st) {
  const int countE = ((hp.N / h) + 1) / 2, countU = (hp.N / (2 * h)) + 1;
  const dim3 grid(hp.B * (countE + countU)), block(64);
  switch (hp.D) {
#define G2_CRLW_CASE(DD) \
  case DD: k_cr_level_wide<DD><<<grid, block, 0, st>>>(pb.params, pb, h); break;
    G2_CRLW_CASE(8) G2_CRLW_CASE(9) G2_CRLW_CASE(10) G2_CRLW_CASE(11)
#undef G2_CRLW_CASE
    default:
      set_error("wide blocks are instantiated for 8 <= dof <= 11");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

template <int n>
__device__ __forceinline__ void wcr_backward(const PlanBuffers& pb, int b, int N, int tid, double* xs, int hmin = 1) {
  const int w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const double* fac = pb.fac + (size_t)b * (N + 1) * 3 * WTILE_DBL;
  int hfinal = 1;
  while (hfinal <= N) hfinal <<= 1;
  for (int h = hfinal; h >= hmin; h >>= 1) {
    const bool final = (h == hfinal);
    const int count = final ? 1 : ((N / h) + 1) / 2;
    for (int idx = w; idx < count; idx += WCR_WAVES) {
      const int j = final ? 0 : h * (2 * idx + 1);
      const double* f = fac + (size_t)j * 3 * WTILE_DBL;
      const WTile Wl = wtile_load_rows<n>(f, lane), Wr = wtile_load_rows<n>(f + WTILE_DBL, lane), V = wtile_load_rows<n>(f + 2 * WTILE_DBL, lane);
      const int jl = j - h, jr = j + h;
      double xl[2], xr[2], x[2];
#pragma unroll
      for (int tc = 0; tc < 2; tc++) {
        xl[tc] = (!final && jl >= 0) ? xs[jl * WX + 16 * tc + c] : 0.0;
        xr[tc] = (!final && jr <= N) ? xs[jr * WX + 16 * tc + c] : 0.0;
      }
      wcr_backsolve<n>(Wl, Wr, V, xl, xr, lane, x);
      if (g == 0) {
#pragma unroll
        for (int tc = 0; tc < 2; tc++) xs[j * WX + 16 * tc + c] = (16 * tc + c < n) ? x[tc] : 0.0;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ double wblock_sum(double v, double* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int k = 0; k < WCR_WAVES; k++) t += red[k];
  return t;
}

// k_solve_step for wide blocks: same contract (delta, trial point, step-control scalars)
template <int D>
__global__ __launch_bounds__(64 * WCR_WAVES) void k_solve_step_wide(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (!pb.active[b]) return;
  const int N = P.N;
  const size_t tsz = (size_t)(N + 1) * n;
  const double* cur = pb.cur + b * tsz;
  double* trial = pb.trial + b * tsz;
  double* delta = pb.delta + b * tsz;
  double* sc = pb.scal + (size_t)b * SC_COUNT;
  const double* gv = pb.gvec + (size_t)b * (N + 1) * WX;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xs = smem;
  double* red = smem + (size_t)(N + 1) * WX;
  int* flags = reinterpret_cast<int*>(red + WCR_WAVES);
  const bool dogleg = P.opt_type == GPMP2MI_OPT_DOGLEG;
  const bool resolve = !(dogleg && pb.phase[b] != 0);
  if (tid == 0) {
    flags[1] = 0;
    pb.stepped[b] = 0;   // set again once the factorisation has succeeded (split form)
  }
  __syncthreads();
  if (resolve) {
    const bool ok = wcr_forward<n>(pb, b, N, tid, P.wide_h0);
    if ((!ok && (tid & 63) == 0) || (tid == 0 && pb.notspd[b])) flags[1] = 1;
    __syncthreads();
    if (flags[1]) {
      if (tid == 0) pb.notspd[b] = 1;  // k_decide consumes and clears it
      return;
    }
    if (P.split_back && !dogleg) {
      // LM / GN: only the blocks that are multiples of 8 are back-substituted here; levels 4, 2, 1, the step, the
      // trial point and the step-control sums follow chip-wide in k_finish_trial_wide
      wcr_backward<n>(pb, b, N, tid, xs, 8);
      double* xg = pb.xg + (size_t)b * (N + 1) * WX;
      for (int k = tid; k < (N / 8 + 1) * WX; k += blockDim.x) {
        const size_t o = (size_t)(k / WX) * 8 * WX + (k % WX);
        xg[o] = xs[o];
      }
      if (tid == 0) pb.stepped[b] = 1;
      return;
    }
    wcr_backward<n>(pb, b, N, tid, xs);
    double gd = 0.0, dd = 0.0, gg = 0.0;
    for (size_t k = tid; k < tsz; k += blockDim.x) {
      const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
      const double x = xs[i * WX + rho], gk = gv[i * WX + rho];
      delta[k] = x;
      gd = fma(gk, x, gd);
      dd = fma(x, x, dd);
      gg = fma(gk, gk, gg);
    }
    gd = wblock_sum(gd, red, tid);
    dd = wblock_sum(dd, red, tid);
    gg = wblock_sum(gg, red, tid);
    if (tid == 0) {
      sc[SC_GD] = gd;
      sc[SC_DD] = dd;
      sc[SC_GG] = gg;
      sc[SC_GN] = gd;
      sc[SC_NN] = dd;
    }
    if (dogleg) {
      double acc = 0.0;
      for (int i = tid; i <= N; i += blockDim.x) acc += pb.hgpart[(size_t)b * P.Npad + i];
      acc = wblock_sum(acc, red, tid);
      if (tid == 0) sc[SC_GHG] = acc;
    }
    __syncthreads();
  }
  if (!dogleg) {
    for (size_t k = tid; k < tsz; k += blockDim.x) {
      const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
      const double* zs = cur + (size_t)i * n;
      const double* dz = xs + i * WX;
      trial[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
    }
    return;
  }
  // ---- Powell dogleg point for trust radius pb.lambda[b]  (same blend as k_solve_step)
  const double Delta = pb.lambda[b];
  const double gg = sc[SC_GG], gHg = sc[SC_GHG], gn = sc[SC_GN], nn = sc[SC_NN];
  const double step = -gg / gHg;
  const double uu = step * step * gg, un = step * gn;
  const double DeltaSq = Delta * Delta;
  double cu, cn, q;
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
  for (size_t k = tid; k < tsz; k += blockDim.x) {
    const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
    const double x = cu * gv[i * WX + rho] + cn * delta[k];
    xs[i * WX + rho] = x;
    xn = fma(x, x, xn);
  }
  __syncthreads();
  for (size_t k = tid; k < tsz; k += blockDim.x) {
    const int i = (int)(k / n), rho = (int)(k - (size_t)i * n);
    const double* zs = cur + (size_t)i * n;
    const double* dz = xs + i * WX;
    trial[k] = (rho < D) ? retract_coord(P.lie != 0, rho, zs, dz) : zs[rho] + dz[rho];
  }
  xn = wblock_sum(xn, red, tid);
  if (tid == 0) {
    sc[SC_Q] = q;
    sc[SC_XNORM] = sqrt(xn);
  }
}

// Chip-wide tail of an LM / GN trial step for wide blocks: one workgroup of 8 wavefronts per (trajectory, blocks
// 8q .. 8q+7).  Block 8q+4 is back-substituted from x_{8q}, x_{8q+8} (level 4), then 8q+2 / 8q+6 (level 2), then the odd
// blocks (level 1); every wavefront then writes the step and the trial point cur (+) delta of its own state and the
// workgroup leaves its share of g.delta, |delta|^2, |g|^2 in spart (k_decide sums them in group order).
template <int D>
__global__ __launch_bounds__(512) void k_finish_trial_wide(const PlanParams* __restrict__ pp, PlanBuffers pb) {
  constexpr int n = 2 * D;
  const PlanParams& P = *pp;
  const int N = P.N;
  const int groups = (N + 8) / 8;
  const int b = blockIdx.x / groups, q = blockIdx.x - b * groups;
  if (!pb.active[b] || pb.stepped[b] != 1) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int i = 8 * q + wv;
  const bool live = i <= N;
  __shared__ double xl_[9][WX];
  __shared__ double psum[8][3];
  const double* xg = pb.xg + (size_t)b * (N + 1) * WX;
  const double* f = pb.fac + ((size_t)b * (N + 1) + i) * 3 * WTILE_DBL;
  const bool has_block = live && wv != 0;
  WTile Wl = wtile_zero(), Wr = wtile_zero(), V = wtile_zero();
  if (has_block) {
    Wl = wtile_load_rows<n>(f, lane);
    Wr = wtile_load_rows<n>(f + WTILE_DBL, lane);
    V = wtile_load_rows<n>(f + 2 * WTILE_DBL, lane);
  }
  if (wv == 0 && lane < WX) xl_[0][lane] = xg[(size_t)(8 * q) * WX + lane];
  if (wv == 1 && lane < WX) xl_[8][lane] = (8 * q + 8 <= N) ? xg[(size_t)(8 * q + 8) * WX + lane] : 0.0;
  __syncthreads();
  auto solve = [&](int h) {
    const int jl = i - h, jr = i + h;
    double xl[2], xr[2], x[2];
#pragma unroll
    for (int tc = 0; tc < 2; tc++) {
      xl[tc] = (jl >= 0) ? xl_[jl - 8 * q][16 * tc + c] : 0.0;
      xr[tc] = (jr <= N) ? xl_[jr - 8 * q][16 * tc + c] : 0.0;
    }
    wcr_backsolve<n>(Wl, Wr, V, xl, xr, lane, x);
    if (g == 0) {
#pragma unroll
      for (int tc = 0; tc < 2; tc++) xl_[wv][16 * tc + c] = (16 * tc + c < n) ? x[tc] : 0.0;
    }
  };
  if (wv == 4 && live) solve(4);
  __syncthreads();
  if ((wv == 2 || wv == 6) && live) solve(2);
  __syncthreads();
  if ((wv & 1) && live) solve(1);
  __syncthreads();
  double gd = 0.0, dd = 0.0, gg = 0.0;
  if (live && lane < n) {
    const size_t k = ((size_t)b * (N + 1) + i) * n + lane;
    const double* zs = pb.cur + ((size_t)b * (N + 1) + i) * n;
    const double x = xl_[wv][lane], gk = pb.gvec[((size_t)b * (N + 1) + i) * WX + lane];
    pb.delta[k] = x;
    pb.trial[k] = (lane < D) ? retract_coord(P.lie != 0, lane, zs, xl_[wv]) : zs[lane] + x;
    gd = gk * x;
    dd = x * x;
    gg = gk * gk;
  }
  gd = wave_sum(gd);
  dd = wave_sum(dd);
  gg = wave_sum(gg);
  if (lane == 0) {
    psum[wv][0] = gd;
    psum[wv][1] = dd;
    psum[wv][2] = gg;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int t = threadIdx.x;
    double a = 0.0;
    for (int w = 0; w < 8; w++) a += psum[w][t];
    pb.spart[((size_t)b * groups + q) * 3 + t] = a;
  }
}

int launch_finish_trial_wide(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const dim3 grid(hp.B * ((hp.N + 8) / 8)), block(512);
  switch (hp.D) {
#define G2_FTW_CASE(DD) \
  case DD: k_finish_trial_wide<DD><<<grid, block, 0, st>>>(pb.params, pb); break;
    G2_FTW_CASE(8) G2_FTW_CASE(9) G2_FTW_CASE(10) G2_FTW_CASE(11)
#undef G2_FTW_CASE
    default:
      set_error("wide blocks are instantiated for 8 <= dof <= 11");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

int launch_solve_step_wide(const PlanParams& hp, const PlanBuffers& pb, hipStream_t st) {
  const dim3 grid(hp.B), block(64 * WCR_WAVES);
  const size_t shmem = ((size_t)(hp.N + 1) * WX + WCR_WAVES + 2) * sizeof(double);
  if (shmem > 150 * 1024) {
    set_error("total_step too large for the LDS-resident solution buffer");
    return GPMP2MI_ERR_UNSUPPORTED;
  }
  switch (hp.D) {
#define G2_SSW_CASE(DD) \
  case DD: k_solve_step_wide<DD><<<grid, block, shmem, st>>>(pb.params, pb); break;
    G2_SSW_CASE(8) G2_SSW_CASE(9) G2_SSW_CASE(10) G2_SSW_CASE(11)
#undef G2_SSW_CASE
    default:
      set_error("wide blocks are instantiated for 8 <= dof <= 11");
      return GPMP2MI_ERR_UNSUPPORTED;
  }
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}
