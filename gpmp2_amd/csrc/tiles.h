// tiles.h -- wavefront-level 16x16 fp64 tile primitives (MFMA accumulator layout) shared by the
// block-tridiagonal solvers.
#pragma once
#include <type_traits>

#include "device_math.h"

namespace g2 {

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
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
#ifdef G2_RCP_TWO_NEWTON
  r = fma(fma(-x, r, 1.0), r, r);
#endif
  return r;
}

// 1 / sqrt(x): v_rsq_f64 seed (~2^-23 relative) + two Newton steps (error below 1 ulp before the final
// roundings) instead of a full-precision sqrt followed by a full-precision division (~35 vs 9 instructions;
// four of them close every tile elimination).
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double t = x * y;
    const double e = fma(-t, y, 1.0);
    y = fma(0.5 * y, e, y);
  }
  return y;
}

// =============================================================================== cross-lane moves
// All VALU (no LDS): gfx950 v_permlane16_swap / v_permlane32_swap for moves between the four
// 16-lane rows of a wavefront, DPP row_newbcast / row_ror for moves inside a row.
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// value held by row G (lanes 16G .. 16G+15), broadcast to all four rows (same column)
template <int G>
__device__ __forceinline__ unsigned bcast_row_u32(unsigned x) {
  const u2v a = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // [r0,r0,r2,r2], [r1,r1,r3,r3]
  const unsigned y = (G & 1) ? a[1] : a[0];
  const u2v b = __builtin_amdgcn_permlane32_swap(y, y, false, false);  // [lo,lo], [hi,hi]
  return (G & 2) ? b[1] : b[0];
}
template <int G>
__device__ __forceinline__ double bcast_row(double v) {
  return __hiloint2double((int)bcast_row_u32<G>((unsigned)__double2hiint(v)),
                          (int)bcast_row_u32<G>((unsigned)__double2loint(v)));
}
// value held by lane J of each row, broadcast inside that row (DPP row_newbcast:J)
// (mov_dpp, not update_dpp(0, ...): with all rows and banks enabled every lane is written, so there is no "old" value
// to keep -- update_dpp with old = 0 costs a v_mov 0 per dword in front of every DPP move)
template <int J>
__device__ __forceinline__ double bcast_in_row(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + J, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + J, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int R>
__device__ __forceinline__ double dpp_row_ror(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x120 + R, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x120 + R, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of each row, result in every lane of the row
__device__ __forceinline__ double row_sum16_dpp(double v) {
  v += dpp_row_ror<8>(v);
  v += dpp_row_ror<4>(v);
  v += dpp_row_ror<2>(v);
  v += dpp_row_ror<1>(v);
  return v;
}
// sum over the four rows (same column), result in every row
__device__ __forceinline__ double sum_rows(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  u2v a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u2v b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double t = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  lo = (unsigned)__double2loint(t);
  hi = (unsigned)__double2hiint(t);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
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

constexpr int TILE_DBL = 256;  // doubles per tile (4 registers x 64 lanes)

__device__ __forceinline__ Tile tile_load(const double* p, int lane) {
  Tile T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = p[k * 64 + lane];
  return T;
}
__device__ __forceinline__ void tile_store(double* p, const Tile& T, int lane) {
#pragma unroll
  for (int k = 0; k < 4; k++) p[k * 64 + lane] = T.r[k];
}
// the same for tiles whose rows >= n are structurally zero (factor and diagonal tiles of the cyclic reduction): the
// padding rows are neither written nor read, a load leaves zeros there
template <int n>
__device__ __forceinline__ Tile tile_load_rows(const double* p, int lane) {
  Tile T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = ((lane >> 4) + 4 * k < n) ? p[k * 64 + lane] : 0.0;
  return T;
}
template <int n>
__device__ __forceinline__ void tile_store_rows(double* p, const Tile& T, int lane) {
#pragma unroll
  for (int k = 0; k < 4; k++)
    if ((lane >> 4) + 4 * k < n) p[k * 64 + lane] = T.r[k];
}
__device__ __forceinline__ Tile tile_zero() {
  Tile T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = 0.0;
  return T;
}

// =============================================================================== CR elimination
// Eliminate the n pivots of S = [S | b] (rhs in column RHSCOL) and apply the row operations to
// the two coupling tiles and to V (identity on entry).  On return
//   Cl <- R^-T Cl, Cr <- R^-T Cr (both with y = R^-T b copied into column RHSCOL), V <- R^-T.
#ifndef G2_M_DPP
#define G2_M_DPP 1
#endif
// measured on MI355X (profiles/r01_cr_variants.txt): pivot-row broadcast through ds_bpermute beats
// the permlane-swap form (48k vs 80k cycles at the widest level); the in-row multiplier broadcast
// is DPP row_newbcast either way.
#ifndef G2_ROW_SWAP
#define G2_ROW_SWAP 0
#endif
#ifndef G2_SUM_DPP
#define G2_SUM_DPP 1
#endif
template <int n>
__device__ __forceinline__ bool tile_eliminate3(Tile& S, Tile& Cl, Tile& Cr, Tile& V, int lane) {
  const int c = lane & 15, g = lane >> 4;
  double piv_of_row[4] = {1.0, 1.0, 1.0, 1.0};
  bool ok = true;
  // pivot row j of the four tiles, broadcast to all four 16-lane rows (same column)
  auto fetch_row = [&](auto jc, double& rS, double& rL, double& rR, double& rV) {
    constexpr int j = decltype(jc)::value, gj = j & 3, rj = j >> 2;
#if G2_ROW_SWAP
    rS = bcast_row<gj>(S.r[rj]);
    rL = bcast_row<gj>(Cl.r[rj]);
    rR = bcast_row<gj>(Cr.r[rj]);
    rV = bcast_row<gj>(V.r[rj]);
#else
    const int src = gj * 16 + c;
    rS = __shfl(S.r[rj], src, 64);
    rL = __shfl(Cl.r[rj], src, 64);
    rR = __shfl(Cr.r[rj], src, 64);
    rV = __shfl(V.r[rj], src, 64);
#endif
  };
  double rowS, rowL, rowR, rowV;
  fetch_row(std::integral_constant<int, 0>{}, rowS, rowL, rowR, rowV);
  static_for<0, n>([&](auto jc) {
    constexpr int j = decltype(jc)::value, gj = j & 3, rj = j >> 2;
    const double piv = readlane_d(S.r[rj], gj * 16 + j);
    ok = ok && (piv > 0.0);
    const double inv = fast_rcp(piv);
    auto update = [&](int k) {
#if G2_M_DPP
      const double m = bcast_in_row<j>(S.r[k]);
#else
      const double m = __shfl(S.r[k], g * 16 + j, 64);
#endif
      // rows at or above the pivot get a zero multiplier instead of a divergent branch; registers whose four
      // rows are all below the pivot (4 k > j) need no test
      const double f = (4 * k > j || g + 4 * k > j) ? m * inv : 0.0;
      S.r[k] = fma(-f, rowS, S.r[k]);
      Cl.r[k] = fma(-f, rowL, Cl.r[k]);
      Cr.r[k] = fma(-f, rowR, Cr.r[k]);
      V.r[k] = fma(-f, rowV, V.r[k]);
    };
    // look-ahead: the register that holds the NEXT pivot row is updated first and its broadcast is issued right
    // away, so the cross-lane latency overlaps with the updates of the remaining registers
    constexpr int k1 = (j + 1) >> 2;
    double nS = 0.0, nL = 0.0, nR = 0.0, nV = 0.0;
    if constexpr (j + 1 < n) {
      if constexpr (4 * k1 + 3 > j) update(k1);
      fetch_row(std::integral_constant<int, j + 1>{}, nS, nL, nR, nV);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (4 * k + 3 <= j) continue;            // rows g + 4k <= j for every g: nothing below the pivot here
      if (j + 1 < n && k == k1) continue;      // done above
      update(k);
    }
    rowS = nS; rowL = nL; rowR = nR; rowV = nV;
  });
  // the pivots are what is left on the diagonal (row j is final once pivot j has been applied): one gather per
  // register instead of two selects per pivot inside the loop
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int rho = g + 4 * k;
    const double d = __shfl(S.r[k], g * 16 + (rho & 15), 64);
    piv_of_row[k] = (rho < n) ? d : 1.0;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double s = fast_rsqrt(piv_of_row[k]);
    const double y = S.r[k] * s;
    Cl.r[k] = (c == RHSCOL) ? y : Cl.r[k] * s;
    Cr.r[k] = (c == RHSCOL) ? y : Cr.r[k] * s;
    V.r[k] *= s;
  }
  return ok;
}

// =============================================================================== CR elimination, column form
// The elimination the cyclic reduction uses since round 3 (tile_eliminate3 above stays as the measured baseline of
// scripts/probes/elim_probe.hip and as an independent implementation for the tests).
//
// Row operations on [S | C_l | C_r | V] move the pivot ROW of four tiles across the four 16-lane row groups of the
// accumulator layout (8 ds_bpermute per pivot) and update up to 16 registers per pivot.  Column operations need
// neither: with S = L D L^T,
//     [S ; I] L^-T = [L D ; L^-T]
// i.e. applying to the stacked pair [S ; Vt] (Vt = I on entry) the column operations that zero row j of S right of
// the diagonal -- column c -= column j * f_c, f_c = S'[j][c] / S'[j][j] -- leaves Vt = L^-T, and after scaling column
// c by 1 / sqrt(pivot c), Vt = R^-1 = V^T (R the upper Cholesky factor).  Per pivot this takes
//   * ONE cross-group broadcast (row j of S only: the multipliers f_c are per COLUMN, i.e. per lane),
//   * column j of every register broadcast inside its 16-lane row by DPP row_newbcast FUSED into the multiply-add
//     (v_fmac_f64_dpp: same issue cost as a plain v_fma_f64, scripts/probes/issue_cost_probe.hip),
//   * exactly 5 such multiply-adds: S only has rows at or below the pivot's register left to update (rows above hold a
//     zero in column j), Vt is upper triangular and only its rows at or above the pivot's register have one.
// The factor tiles then come off the matrix cores: W = R^-T C = V C = Vt^T C = tile_atb(Vt, C), with the right-hand
// side riding in column RHSCOL of both coupling tiles.  The back-substitution wants V in the tile layout (x = V^T t): Vt is
// written to memory transposed (tile_store_transposed).
// Measured (elim_probe, one wave per SIMD / 16 waves per CU): 4.0 k -> 3.3 k and 8.8 k -> 5.8 k cycles per elimination.
template <int j, int rj, int idx>
__device__ __forceinline__ double& elim_reg(Tile& S, Tile& Vt) {
  // the 5 registers pivot j touches, the one holding the NEXT pivot row first: S.r[rj .. 3], then Vt.r[0 .. rj]
  constexpr int ns = 4 - rj;
  if constexpr (idx < ns) {
    constexpr int first = ((j + 1) >> 2) < 4 ? ((j + 1) >> 2) : rj;   // register of row j + 1
    constexpr int k = (idx == 0) ? first : ((rj + idx - 1 >= first) ? rj + idx : rj + idx - 1);
    return S.r[k];
  } else {
    return Vt.r[idx - ns];
  }
}

template <int n>
__device__ __forceinline__ bool tile_eliminate_col(Tile& S, Tile& Vt, int lane) {
  const int c = lane & 15;
  static_for<0, n>([&](auto jc) {
    constexpr int j = decltype(jc)::value, gj = j & 3, rj = j >> 2;
    const double rowS = __shfl(S.r[rj], gj * 16 + c, 64);       // S'[j][c] in every row group
    const double piv = readlane_d(S.r[rj], gj * 16 + j);
    const double ninv = -fast_rcp(piv);
    const double nf = (c > j && c < n) ? rowS * ninv : 0.0;    // -f_c; columns <= j and the padding stay as they are
    double &r0 = elim_reg<j, rj, 0>(S, Vt), &r1 = elim_reg<j, rj, 1>(S, Vt), &r2 = elim_reg<j, rj, 2>(S, Vt),
           &r3 = elim_reg<j, rj, 3>(S, Vt), &r4 = elim_reg<j, rj, 4>(S, Vt);
    double t0 = r0, t1 = r1, t2 = r2, t3 = r3, t4 = r4;
    // (s_nop 1: a DPP source written by the preceding vector instruction needs two wait states; the five registers are
    // distinct, so one pad in front of the group covers it)
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %2, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %3, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %4, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
        : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4)
        : "v"(nf), "n"(j));
    r0 = t0; r1 = t1; r2 = t2; r3 = t3; r4 = t4;
  });
  // the pivots are what is left on the diagonal (column c is final once pivot c - 1 has been applied): column c needs
  // S[c][c], held by row group c & 3 in register c >> 2
  double pv = 1.0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (4 * k >= n) continue;
    const double dgn = __shfl(S.r[k], (c & 3) * 16 + c, 64);
    pv = ((c >> 2) == k && c < n) ? dgn : pv;
  }
  const bool ok = __all(pv > 0.0);   // false for a non-positive or NaN pivot (gtsam::IndeterminantLinearSystemException)
  const double rs = fast_rsqrt(pv);
#pragma unroll
  for (int k = 0; k < 4; k++) Vt.r[k] *= rs;
  return ok;
}

// One elimination task of the cyclic reduction.  On entry S = [S | b] (right-hand side in column RHSCOL), Cl / Cr the
// n x n couplings.  On return Cl <- W_l = R^-T C_l, Cr <- W_r = R^-T C_r, both with y = R^-T b in column RHSCOL, and
// Vt = R^-1 (= V^T); S is destroyed.  Returns false when a pivot is not positive.
template <int n>
__device__ __forceinline__ bool tile_eliminate_cv(Tile& S, Tile& Cl, Tile& Cr, Tile& Vt, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    Vt.r[k] = (g + 4 * k == c && c < n) ? 1.0 : 0.0;
    Cl.r[k] = (c == RHSCOL) ? S.r[k] : Cl.r[k];
    Cr.r[k] = (c == RHSCOL) ? S.r[k] : Cr.r[k];
  }
  const bool ok = tile_eliminate_col<n>(S, Vt, lane);
  Cl = tile_atb(Vt, Cl);
  Cr = tile_atb(Vt, Cr);
  return ok;
}

// V = Vt^T from a row-major Vt tile in memory: lane (g, c), register k <- Vt[c][g + 4 k].  Rows c >= n of Vt are not
// stored (tile_store_rows) and read as zeros; so do the rows >= n of V (columns >= n of Vt are zero).  A 16-lane group
// reads one 32-B sector of 16 different rows per register.  Used where the WRITER is the latency-critical side (the
// per-trajectory step kernels store Vt with plain row stores and their back-substitution levels load it this way);
// k_assemble, whose factors the chip-wide finish kernels read, stores V itself (tile_store_transposed).
template <int n>
__device__ __forceinline__ Tile tile_load_transposed(const double* __restrict__ p, int lane) {
  const int c = lane & 15, g = lane >> 4;
  Tile T;
#pragma unroll
  for (int k = 0; k < 4; k++) T.r[k] = (c < n && 4 * k < n) ? p[c * 16 + g + 4 * k] : 0.0;
  return T;
}

// V = Vt^T to memory: lane (g, c), register k holds Vt[g + 4 k][c] and writes it to V[c][g + 4 k].  The four lanes that
// share (c, k) fill one aligned 32-B sector, so every store instruction writes 16 full sectors; stores are off the
// critical path of the elimination, while the back-substitution -- which wants V in the tile layout (x = V^T t) -- keeps
// its plain coalesced tile loads.  (The first form of round 3 stored Vt and loaded it transposed: 16 cache lines per
// load instruction in the latency-bound finish kernel, 6.8 -> 7.3 us.)  Rows c >= n of V and the columns >= n of its
// rows are never written: they keep the zeros the plan's arena was created with.
template <int n>
__device__ __forceinline__ void tile_store_transposed(double* __restrict__ p, const Tile& Vt, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (c < n && g + 4 * k < n) p[c * 16 + g + 4 * k] = Vt.r[k];
}

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

// ---------------------------------------------------------------------------------------------------------------
// back-substitution of one block of the cyclic reduction (cr_kernels.hip: cr_backward, k_finish_step; plan_kernels.hip:
// the fused finish of k_linearize_arm)
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row_sum16(double v) {
#if G2_SUM_DPP
  return row_sum16_dpp(v);
#else
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
#endif
}

// x_j = V^T (y - Wl x_l - Wr x_r); xl / xr = neighbour solutions at this lane's column
template <int n>
__device__ __forceinline__ double cr_backsolve(const Tile& Wl, const Tile& Wr, const Tile& V, double xl,
                                               double xr, int lane) {
  const int c = lane & 15;
  double t[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double v;
    if (c == RHSCOL) v = -Wl.r[k];                     // -y (same in both tiles)
    else v = (c < n) ? fma(Wl.r[k], xl, Wr.r[k] * xr) : 0.0;
    t[k] = -row_sum16(v);
  }
  double x = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) x = fma(V.r[k], t[k], x);
#if G2_SUM_DPP
  return sum_rows(x);
#else
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
#endif
}

// The third factor tile of a block eliminated at level h: k_assemble (level 1, and level 2 when N >= 2) stores V, the
// levels the step kernels run themselves store Vt (see tile_load_transposed)
template <int n>
__device__ __forceinline__ Tile load_v(const double* p, int h, int N, int lane) {
  const int h0 = (N >= 2) ? 4 : 2;   // first level of cr_forward
  return (h >= h0) ? tile_load_transposed<n>(p, lane) : tile_load_rows<n>(p, lane);
}

}  // namespace g2
