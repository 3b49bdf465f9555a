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

}  // namespace g2
