// elim_probe.hip -- cycles and results of the tile elimination variants, in isolation.
//   A  tiles.h tile_eliminate3 (accumulator layout, ds_bpermute row broadcasts)      -- the round-2 form
//   B  rtile.h rblock_* (row-per-register layout through an LDS scratch, 64-bit DPP column broadcasts)
//   C  as B with the broadcast fused into the multiply-add (v_fmac_f64_dpp, inline asm)
//   D  one lane per column of [S | C_l | C_r | V], multipliers through v_readlane -> SGPR operands
//   E  tiles.h tile_eliminate_col: column operations on [S ; Vt] with v_fmac_f64_dpp, W = V C on the matrix cores (shipped)
//   E2 as E with two pivots per step
// build: hipcc -O3 --offload-arch=gfx950 -I gpmp2_amd/csrc scripts/probes/elim_probe.hip -o /tmp/elim_probe
// run:   /tmp/elim_probe [waves_per_block] [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rtile.h"

using namespace g2;
constexpr int n = 14;

template <int n_>
__device__ __forceinline__ bool rblock_eliminate_fused(RBlock<n_>& R, int lane) {
  const int c = lane & 15;
  double pv = 1.0;
  bool ok = true;
  static_for<0, n_>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double piv = bcast_col<j>(R.s[j]);
    ok = ok && (piv > 0.0);
    pv = (c == j) ? piv : pv;
    const double ninv = -fast_rcp(piv);
    const double ps = R.s[j] * ninv, px = R.x[j] * ninv;
    static_for<j + 1, n_>([&](auto rc) {
      constexpr int rho = decltype(rc)::value;
      // x[rho] += bcast_j(s[rho]) * px ; s[rho] += bcast_j(s[rho]) * ps   (x first: it reads the old s[rho])
      double xr = R.x[rho], sr = R.s[rho];
      const double pxx = px, pss = ps;
      asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f64_dpp %1, %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                   : "+v"(xr), "+v"(sr)
                   : "v"(pxx), "v"(pss), "n"(j));
      R.x[rho] = xr;
      R.s[rho] = sr;
    });
  });
  const double rs = fast_rsqrt(pv);
  static_for<0, n_>([&](auto rc) {
    constexpr int rho = decltype(rc)::value;
    R.x[rho] *= bcast_col<rho>(rs);
  });
  return ok;
}

// D: lane (t, c): column c of tile t (0: S, 1: C_l, 2: C_r, 3: V), a[rho] = row rho
template <int n_>
struct CBlock {
  double a[n_];
};
template <int n_>
__device__ __forceinline__ void cblock_load(CBlock<n_>& R, const Tile& S, const Tile& Cl, const Tile& Cr, double* scratch, int lane) {
  const int c = lane & 15, t = lane >> 4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    scratch[k * 64 + lane] = S.r[k];
    scratch[TILE_DBL + k * 64 + lane] = (c == RHSCOL) ? S.r[k] : Cl.r[k];
    scratch[2 * TILE_DBL + k * 64 + lane] = (c == RHSCOL) ? S.r[k] : Cr.r[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const double* xs = scratch + (t < 3 ? t : 0) * TILE_DBL;
#pragma unroll
  for (int rho = 0; rho < n_; rho++) {
    const double v = xs[rho * 16 + c];
    R.a[rho] = (t < 3) ? v : ((c == rho) ? 1.0 : 0.0);
  }
}
template <int n_>
__device__ __forceinline__ bool cblock_eliminate(CBlock<n_>& R, int lane) {
  double pv = 1.0;
  bool ok = true;
  static_for<0, n_>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double piv = readlane_d(R.a[j], j);           // S[j][j]: lane (0, j)
    ok = ok && (piv > 0.0);
    pv = (lane == j) ? piv : pv;
    const double ninv = -fast_rcp(piv);
    const double prow = R.a[j] * ninv;
    static_for<j + 1, n_>([&](auto rc) {
      constexpr int rho = decltype(rc)::value;
      const double m = readlane_d(R.a[j], rho);         // S[j][rho] = S[rho][j]
      R.a[rho] = fma(m, prow, R.a[rho]);
    });
  });
  const double rs = fast_rsqrt(pv);   // lane j < n: 1 / sqrt(pivot j)
  static_for<0, n_>([&](auto rc) {
    constexpr int rho = decltype(rc)::value;
    R.a[rho] *= readlane_d(rs, rho);
  });
  return ok;
}

// E: the shipped column form, tiles.h tile_eliminate_col (column operations on [S ; Vt], W = V C by two tile products)
// E2: as E, two pivots per step.  Both multiplier rows are formed from the state BEFORE the pair (row j and row j + 1
// broadcast together, the 2x2 pivot block read in one go), so the cross-lane and reciprocal latencies are paid n / 2
// times; the two rank-1 column updates then run back to back.
template <int n_>
__device__ __forceinline__ bool tile_eliminate_col2(Tile& S, Tile& Vt, int lane) {
  static_assert(n_ % 2 == 0, "pairs of pivots");
  const int c = lane & 15;
  double pv = 1.0;
  bool ok = true;
  static_for<0, n_ / 2>([&](auto jc) {
    constexpr int j = 2 * decltype(jc)::value, j1 = j + 1, gj = j & 3, rj = j >> 2, gj1 = j1 & 3, rj1 = j1 >> 2;
#ifdef PROBE_SWAP
    const double row0 = bcast_row<gj>(S.r[rj]), row1 = bcast_row<gj1>(S.r[rj1]);
#else
    const double row0 = __shfl(S.r[rj], gj * 16 + c, 64);       // S'[j][c]
    const double row1 = __shfl(S.r[rj1], gj1 * 16 + c, 64);     // S'[j+1][c]   (before pivot j)
#endif
    const double a = readlane_d(S.r[rj], gj * 16 + j);
    const double b = readlane_d(S.r[rj1], gj1 * 16 + j);        // S'[j+1][j]
    const double d = readlane_d(S.r[rj1], gj1 * 16 + j1);
    const double inv1 = fast_rcp(a);
    const double bi = b * inv1;
    const double p2 = fma(-b, bi, d);                           // second pivot: d - b^2 / a
    const double inv2 = fast_rcp(p2);
    const double f = row0 * inv1;                               // multipliers of pivot j
    const double r1 = fma(-b, f, row1);                         // row j + 1 after pivot j
    const double nf = (c > j && c < n_) ? -f : 0.0;
    const double nf2 = (c > j1 && c < n_) ? -(r1 * inv2) : 0.0;
    static_for<0, 4>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k >= rj) {
        double t = S.r[k];
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                     : "+v"(t) : "v"(nf), "v"(nf2), "n"(j), "n"(j1));
        S.r[k] = t;
      }
      if constexpr (k <= rj1) {
        double t = Vt.r[k];
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                     : "+v"(t) : "v"(nf), "v"(nf2), "n"(j), "n"(j1));
        Vt.r[k] = t;
      }
    });
  });
  // the pivots are what is left on the diagonal: column c needs S[c][c] (row group c & 3, register c >> 2)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double dgn = __shfl(S.r[k], (c & 3) * 16 + c, 64);
    pv = ((c >> 2) == k && c < n_) ? dgn : pv;
  }
  ok = !(pv <= 0.0) && (pv == pv);
  ok = __all(ok);
  const double rs = fast_rsqrt(pv);
#pragma unroll
  for (int k = 0; k < 4; k++) Vt.r[k] *= rs;
  return ok;
}

template <int variant>
__global__ __launch_bounds__(1024) void k_probe(int reps, const double* __restrict__ in, double* __restrict__ out,
                                                long long* __restrict__ cycles) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * (blockDim.x >> 6) + w;
  double* scratch = smem + (size_t)w * RSCRATCH_DBL;
  const double* p = in + (size_t)wave * 3 * TILE_DBL;
  const Tile S0 = tile_load(p, lane), L0 = tile_load(p + TILE_DBL, lane), R0 = tile_load(p + 2 * TILE_DBL, lane);
  double* o = out + (size_t)wave * 3 * TILE_DBL;
  bool ok = true;
  long long t0 = 0, t1 = 0, tacc = 0, tr = 0;
  for (int rep = 0; rep < reps + 1; rep++) {
    if (rep == 1) t0 = __builtin_amdgcn_s_memtime();
    Tile S = S0, Cl = L0, Cr = R0;
    tr = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int k = 0; k < 4; k++) asm volatile("" : "+v"(S.r[k]), "+v"(Cl.r[k]), "+v"(Cr.r[k]));
    if constexpr (variant == 0) {
      Tile V;
#pragma unroll
      for (int k = 0; k < 4; k++) V.r[k] = (g + 4 * k == c) ? 1.0 : 0.0;
      ok = tile_eliminate3<n>(S, Cl, Cr, V, lane) && ok;
      asm volatile("" : "+v"(Cl.r[0]), "+v"(Cr.r[0]), "+v"(V.r[0]));
      if (rep) tacc += __builtin_amdgcn_s_memtime() - tr;
      tile_store_rows<n>(o, Cl, lane);
      tile_store_rows<n>(o + TILE_DBL, Cr, lane);
      tile_store_rows<n>(o + 2 * TILE_DBL, V, lane);
    } else if constexpr (variant == 1 || variant == 2) {
      RBlock<n> R;
      rblock_load<n>(R, S, Cl, Cr, scratch, lane);
      ok = (variant == 1 ? rblock_eliminate<n>(R, lane) : rblock_eliminate_fused<n>(R, lane)) && ok;
      rblock_store<n>(R, o, lane);
    } else if constexpr (variant == 4 || variant == 5) {
      Tile Vt;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        Vt.r[k] = (g + 4 * k == c) ? 1.0 : 0.0;
        Cl.r[k] = (c == RHSCOL) ? S.r[k] : Cl.r[k];
        Cr.r[k] = (c == RHSCOL) ? S.r[k] : Cr.r[k];
      }
      ok = (variant == 4 ? tile_eliminate_col<n>(S, Vt, lane) : tile_eliminate_col2<n>(S, Vt, lane)) && ok;
      asm volatile("" : "+v"(Vt.r[0]), "+v"(Vt.r[3]));
      if (rep) tacc += __builtin_amdgcn_s_memtime() - tr;
      const Tile Wl = tile_atb(Vt, Cl), Wr = tile_atb(Vt, Cr);
      tile_store_rows<n>(o, Wl, lane);
      tile_store_rows<n>(o + TILE_DBL, Wr, lane);
      tile_store_rows<n>(o + 2 * TILE_DBL, Vt, lane);
    } else {
      CBlock<n> R;
      cblock_load<n>(R, S, Cl, Cr, scratch, lane);
      ok = cblock_eliminate<n>(R, lane) && ok;
      if (g >= 1) {
#pragma unroll
        for (int rho = 0; rho < n; rho++) o[(g - 1) * TILE_DBL + rho * 16 + c] = R.a[rho];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[wave] = ok ? ((t1 - t0) / reps) * 100000LL + tacc / reps : -1;
}

int main(int argc, char** argv) {
  const int wpb = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 20;
  const int blocks = 256, waves = blocks * wpb;
  std::vector<double> in((size_t)waves * 3 * 256, 0.0);
  srand(7);
  auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
  std::vector<double> ref((size_t)waves * 3 * 256, 0.0);
  for (int w = 0; w < waves; w++) {
    double A[n][n], Sm[n][n], b[n], Cl[n][n], Cr[n][n];
    for (int i = 0; i < n; i++) {
      b[i] = rnd();
      for (int j = 0; j < n; j++) { A[i][j] = rnd(); Cl[i][j] = rnd(); Cr[i][j] = rnd(); }
    }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double v = (i == j) ? 0.5 : 0.0;
        for (int k = 0; k < n; k++) v += A[i][k] * A[j][k];
        Sm[i][j] = v;
      }
    double* p = &in[(size_t)w * 768];
    for (int i = 0; i < n; i++) {
      for (int j = 0; j < n; j++) { p[i * 16 + j] = Sm[i][j]; p[256 + i * 16 + j] = Cl[i][j]; p[512 + i * 16 + j] = Cr[i][j]; }
      p[i * 16 + 15] = b[i];
    }
    // reference: R^T R = S (upper R), W = R^-T [Cl | Cr | b | I]
    double R[n][n] = {};
    for (int j = 0; j < n; j++) {
      double d = Sm[j][j];
      for (int k = 0; k < j; k++) d -= R[k][j] * R[k][j];
      R[j][j] = std::sqrt(d);
      for (int i = j + 1; i < n; i++) {
        double v = Sm[j][i];
        for (int k = 0; k < j; k++) v -= R[k][j] * R[k][i];
        R[j][i] = v / R[j][j];
      }
    }
    double* q = &ref[(size_t)w * 768];
    auto fwd = [&](auto get, double* dst, int col) {   // solves R^T y = rhs column
      double y[n];
      for (int i = 0; i < n; i++) {
        double v = get(i);
        for (int k = 0; k < i; k++) v -= R[k][i] * y[k];
        y[i] = v / R[i][i];
      }
      for (int i = 0; i < n; i++) dst[i * 16 + col] = y[i];
    };
    for (int j = 0; j < n; j++) {
      fwd([&](int i) { return Cl[i][j]; }, q, j);
      fwd([&](int i) { return Cr[i][j]; }, q + 256, j);
      fwd([&](int i) { return i == j ? 1.0 : 0.0; }, q + 512, j);
    }
    fwd([&](int i) { return b[i]; }, q, 15);
    fwd([&](int i) { return b[i]; }, q + 256, 15);
  }
  double *din, *dout;
  long long* dcyc;
  (void)hipMalloc(&din, in.size() * 8);
  (void)hipMalloc(&dout, in.size() * 8);
  (void)hipMalloc(&dcyc, waves * 8);
  (void)hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  const char* names[6] = {"A tile_eliminate3 (bpermute)", "B R-layout, mov_b64_dpp", "C R-layout, fmac_f64_dpp", "D column lanes, readlane",
                          "E column ops on [S;Vt] + MFMA", "E2 two pivots per step"};
  for (int v = 0; v < 6; v++) {
    (void)hipMemset(dout, 0, in.size() * 8);
    const size_t shmem = (size_t)wpb * RSCRATCH_DBL * 8;
    for (int it = 0; it < 2; it++) {
      if (v == 0) k_probe<0><<<blocks, 64 * wpb, shmem>>>(reps, din, dout, dcyc);
      if (v == 1) k_probe<1><<<blocks, 64 * wpb, shmem>>>(reps, din, dout, dcyc);
      if (v == 2) k_probe<2><<<blocks, 64 * wpb, shmem>>>(reps, din, dout, dcyc);
      if (v == 3) k_probe<3><<<blocks, 64 * wpb, shmem>>>(reps, din, dout, dcyc);
      if (v == 4) k_probe<4><<<blocks, 64 * wpb, shmem>>>(reps, din, dout, dcyc);
      if (v == 5) k_probe<5><<<blocks, 64 * wpb, shmem>>>(reps, din, dout, dcyc);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("variant %d failed\n", v); return 1; }
    std::vector<double> out(in.size());
    std::vector<long long> cyc(waves);
    (void)hipMemcpy(out.data(), dout, in.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(cyc.data(), dcyc, waves * 8, hipMemcpyDeviceToHost);
    double maxerr = 0.0;
    for (int w = 0; w < waves; w++)
      for (int t = 0; t < 3; t++)
        for (int i = 0; i < n; i++)
          for (int j = 0; j < 16; j++) {
            if (j == 14 || (t == 2 && j == 15)) continue;
            const size_t o = (size_t)w * 768 + t * 256 + i * 16 + j;
            if (v >= 4 && t == 2) {   // E stores Vt = V^T
              if (j >= n) continue;
              maxerr = std::fmax(maxerr, std::fabs(out[(size_t)w * 768 + 512 + j * 16 + i] - ref[o]) / (1.0 + std::fabs(ref[o])));
              continue;
            }
            maxerr = std::fmax(maxerr, std::fabs(out[o] - ref[o]) / (1.0 + std::fabs(ref[o])));
          }
    long long lo = 1LL << 60, hi = 0, sum = 0, esum = 0;
    for (auto x : cyc) { esum += x % 100000LL; x /= 100000LL; lo = std::min(lo, x); hi = std::max(hi, x); sum += x; }
    printf("%-32s waves/block %2d: cycles per elimination (incl. conversion + stores) min %lld mean %lld max %lld; elimination alone %lld; max rel err %.2e\n",
           names[v], wpb, lo, sum / waves, hi, esum / waves, maxerr);
  }
  return 0;
}
