// rtile.h -- the tile elimination of the cyclic reduction in a row-per-register layout ("R layout").
//
// tiles.h keeps every 16x16 matrix in the accumulator layout of v_mfma_f64_16x16x4 (lane = column + 16 * (row & 3),
// register = row >> 2): right for the Schur products, wrong for Gaussian elimination -- every pivot step needs the
// pivot ROW of four tiles moved across the four 16-lane groups (ds_bpermute) and the multiplier COLUMN moved inside
// them, and updates registers of which a growing share holds rows above the pivot.
//
// Here, for the 14 (n) pivot steps only, lane (q, c) = (lane >> 4, lane & 15) holds column c of
//     s[rho] = S[rho][c]        the SPD block with the right-hand side in column RHSCOL; the same in all four q
//     x[rho] = X_q[rho][c]      the companion tile of lane group q:  0: C_l,  1: C_r  (both carry a copy of the
//                               right-hand side in column RHSCOL),  2: V (identity on entry),  3: idle
// one register per ROW.  A pivot step is then
//     piv = s[j] at column j           one v_mov_b64_dpp row_newbcast:j (the pivot row is this lane's own register)
//     m   = s[rho] at column j         one v_mov_b64_dpp per row below the pivot (S is symmetric, replicated per group)
//     s[rho] -= m (s[j] / piv),  x[rho] -= m (x[j] / piv)
// with no cross-group traffic, no rows above the pivot touched and no LDS-pipe instruction in the dependent chain.
// The tiles enter through a wave-private LDS scratch (written in the accumulator layout, read back one row per
// register) and leave straight to memory: the factor tiles are plain row-major 16x16 there, which both layouts
// address directly.
#pragma once
#include "tiles.h"

namespace g2 {

// value held by column J of each 16-lane row, broadcast inside that row: ONE 64-bit DPP move (row_newbcast is the DPP
// control the double-precision ALU accepts)
template <int J>
__device__ __forceinline__ double bcast_col(double v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + J, 0xF, 0xF, false);
}

constexpr int RSCRATCH_DBL = 3 * TILE_DBL;   // wave-private LDS scratch of one elimination (S, C_l, C_r)

template <int n>
struct RBlock {
  double s[n], x[n];
};

// accumulator-layout tiles -> R layout through the wave's LDS scratch.  The right-hand side (column RHSCOL of S) is
// copied into column RHSCOL of both coupling tiles on the way, so that the row operations leave y = R^-T b there.
template <int n>
__device__ __forceinline__ void rblock_load(RBlock<n>& R, const Tile& S, const Tile& Cl, const Tile& Cr,
                                            double* __restrict__ scratch, int lane) {
  const int c = lane & 15, q = lane >> 4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (4 * k >= n) continue;
    scratch[k * 64 + lane] = S.r[k];
    scratch[TILE_DBL + k * 64 + lane] = (c == RHSCOL) ? S.r[k] : Cl.r[k];
    scratch[2 * TILE_DBL + k * 64 + lane] = (c == RHSCOL) ? S.r[k] : Cr.r[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const double* xs = scratch + (q < 2 ? (1 + q) * TILE_DBL : 0);   // groups 2, 3 read S again and overwrite it below
#pragma unroll
  for (int rho = 0; rho < n; rho++) {
    R.s[rho] = scratch[rho * 16 + c];
    const double v = xs[rho * 16 + c];
    R.x[rho] = (q < 2) ? v : ((c == rho) ? 1.0 : 0.0);
  }
}

// Eliminates the n pivots.  On return x[rho] holds row rho of W_l = R^-T C_l (q = 0), W_r = R^-T C_r (q = 1), both with
// y = R^-T b in column RHSCOL, and of V = R^-T (q = 2).  Returns false when a pivot is not positive.
template <int n>
__device__ __forceinline__ bool rblock_eliminate(RBlock<n>& R, int lane) {
  const int c = lane & 15;
  double pv = 1.0;   // column c collects pivot c
  bool ok = true;
  static_for<0, n>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double piv = bcast_col<j>(R.s[j]);
    ok = ok && (piv > 0.0);
    pv = (c == j) ? piv : pv;
    const double ninv = -fast_rcp(piv);
    const double ps = R.s[j] * ninv, px = R.x[j] * ninv;
    static_for<j + 1, n>([&](auto rc) {
      constexpr int rho = decltype(rc)::value;
      const double m = bcast_col<j>(R.s[rho]);
      R.x[rho] = fma(m, px, R.x[rho]);
      R.s[rho] = fma(m, ps, R.s[rho]);
    });
  });
  const double rs = fast_rsqrt(pv);   // column c: 1 / sqrt(pivot c)
  static_for<0, n>([&](auto rc) {
    constexpr int rho = decltype(rc)::value;
    R.x[rho] *= bcast_col<rho>(rs);
  });
  return ok;
}

// rows of W_l, W_r, V -> three consecutive row-major tiles at f (global or LDS); `count` tiles (2: W_l, W_r only)
template <int n>
__device__ __forceinline__ void rblock_store(const RBlock<n>& R, double* __restrict__ f, int lane, int count = 3) {
  const int c = lane & 15, q = lane >> 4;
  if (q >= count) return;
  double* p = f + q * TILE_DBL + c;
#pragma unroll
  for (int rho = 0; rho < n; rho++) p[rho * 16] = R.x[rho];
}

}  // namespace g2
