// sdf_kernels.hip -- signed distance field construction from an occupancy grid on the device.
//
// Replaces matlab/+gpmp2/signedDistanceField{2D,3D}.m:16-34 and
// gpmp2_python/gpmp2_python/utils/signedDistanceField{2D,3D}.py (bwdist / scipy
// distance_transform_edt): field = (EDT to the obstacle set - EDT to the free set) * cell_size.
//
// Exact Euclidean distance transform in integer arithmetic: squared distances are int32, the
// transform is separable, and every axis pass is   out[i] = min_j in[j] + (i - j)^2   with the
// line staged in LDS: the contiguous axis by an outward search that stops as soon as k^2 >= best
// (lines without any finite value are skipped), the strided axes by a linear-time lower-envelope
// scan (k_edt_strided_scan; the outward-search form remains as the fallback for very long axes).  The two target
// sets (obstacle cells, free cells) are two int32 volumes processed by the same launches
// (blockIdx.z).  HBM-bound integer work: 3 passes x 2 volumes x (4 B read + 4 B write) per cell.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"
#include "launch.h"

namespace g2 {

constexpr int EDT_INF = 0x3fffffff;

// a = squared distance seed to the obstacle set (0 on obstacle cells), b = to the free set
__global__ void k_edt_seed(size_t n, const double* __restrict__ occ, int* __restrict__ a, int* __restrict__ b) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const bool obst = occ[i] > 0.75;  // "regularize unknown area to open area", signedDistanceField3D.m:16
    a[i] = obst ? 0 : EDT_INF;
    b[i] = obst ? EDT_INF : 0;
  }
}

__device__ __forceinline__ int edt_line_min(const int* __restrict__ line, int stride, int len, int i) {
  int best = line[(size_t)i * stride];
  for (int k = 1; k < len; k++) {
    const int kk = k * k;
    if (kk >= best) break;
    const int lo = i - k, hi = i + k;
    if (lo < 0 && hi >= len) break;
    if (lo >= 0) best = min(best, line[(size_t)lo * stride] + kk);
    if (hi < len) best = min(best, line[(size_t)hi * stride] + kk);
  }
  return best;
}

// A line without any finite value stays all-INF (typical for the first pass over a sparse map: most
// rows contain no obstacle cell); the per-line flag spares its cells the full-length search.
// pass along the contiguous axis: a block stages `rows` consecutive lines of length len
__global__ void k_edt_x(int len, size_t nrows, int rows, int* __restrict__ va, int* __restrict__ vb) {
  extern __shared__ int edt_lds[];
  int* v = blockIdx.z ? vb : va;
  const size_t r0 = (size_t)blockIdx.x * rows;
  const int nr = (int)min((size_t)rows, nrows - r0);
  const int cnt = nr * len;
  int* g = v + r0 * len;
  int* has = edt_lds + rows * len;  // [rows] line contains a finite value
  for (int r = threadIdx.x; r < nr; r += blockDim.x) has[r] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    const int x = g[e];
    edt_lds[e] = x;
    if (x < EDT_INF) has[e / len] = 1;  // every writer stores the same value
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    const int r = e / len, x = e - r * len;
    if (has[r]) g[e] = edt_line_min(edt_lds + r * len, 1, len, x);
  }
}

// pass along a strided axis.  Memory is [outer][len][inner]; a block stages the tile
// [len][W inner positions] (coalesced along inner) and thread (w, grp) produces every
// (blockDim.x / W)-th output of column w.
template <int W>
__global__ void k_edt_strided(int len, size_t inner, int* __restrict__ va, int* __restrict__ vb) {
  extern __shared__ int edt_lds[];
  int* v = blockIdx.z ? vb : va;
  const size_t i0 = (size_t)blockIdx.x * W;
  int* g = v + (size_t)blockIdx.y * len * inner + i0;
  const int w = threadIdx.x % W, grp = threadIdx.x / W, ngrp = blockDim.x / W;
  const bool live = i0 + w < inner;
  int* has = edt_lds + len * W;  // [W] column contains a finite value
  if (threadIdx.x < W) has[threadIdx.x] = 0;
  __syncthreads();
  for (int j = grp; j < len; j += ngrp) {
    const int x = live ? g[(size_t)j * inner + w] : EDT_INF;
    edt_lds[j * W + w] = x;
    if (x < EDT_INF) has[w] = 1;
  }
  __syncthreads();
  if (!live || !has[w]) return;
  for (int j = grp; j < len; j += ngrp) g[(size_t)j * inner + w] = edt_line_min(edt_lds + w, W, len, j);
}

// Linear-time form of the strided pass (Meijster, Roerdink & Hesselink 2000: lower envelope of the
// parabolas x -> (x - i)^2 + g[i]) in exact integer arithmetic: the same minimum as edt_line_min in
// O(len) per line instead of O(len * distance).  The [len][W] tile and the envelope stack
// (s | t << 16 per entry) live in LDS; W threads each scan one line, all threads of the block stage the
// tile (coalesced along the inner axis); results go straight back to global memory, W consecutive
// values per store.  Needs len < 32768.
template <int W>
__global__ void k_edt_strided_scan(int len, size_t inner, int* __restrict__ va, int* __restrict__ vb) {
  extern __shared__ int edt_lds[];
  int* gl = edt_lds;                      // [len][W] input values
  int* stk = edt_lds + (size_t)len * W;   // [len][W] envelope stack
  int* v = blockIdx.z ? vb : va;
  const size_t i0 = (size_t)blockIdx.x * W;
  int* g = v + (size_t)blockIdx.y * len * inner + i0;
  const int w = threadIdx.x % W, grp = threadIdx.x / W, ngrp = blockDim.x / W;
  const bool live = i0 + w < inner;
  for (int j = grp; j < len; j += ngrp) gl[j * W + w] = live ? g[(size_t)j * inner + w] : EDT_INF;
  __syncthreads();
  if (threadIdx.x >= W || !live) return;
  auto G = [&](int i) { return gl[i * W + w]; };
  auto F = [&](int x, int i) { return (long long)(x - i) * (x - i) + G(i); };
  int q = -1;
  for (int u = 0; u < len; u++) {
    if (G(u) >= EDT_INF) continue;  // an empty cell contributes no parabola
    while (q >= 0) {
      const int e = stk[q * W + w], sq = e & 0xffff, tq = e >> 16;
      if (F(tq, sq) > F(tq, u)) q--;
      else break;
    }
    if (q < 0) {
      q = 0;
      stk[w] = u;  // s = u, valid from t = 0
    } else {
      const int sq = stk[q * W + w] & 0xffff;
      // first x at which the parabola of u lies below the one of sq (u > sq; non-negative after the pops)
      const long long num = (long long)u * u - (long long)sq * sq + G(u) - G(sq);
      const long long wpos = 1 + num / (2LL * (u - sq));
      if (wpos < len) {
        q++;
        stk[q * W + w] = u | ((int)wpos << 16);
      }
    }
  }
  if (q < 0) return;  // no finite value on this line: it stays all-INF
  for (int u = len - 1; u >= 0; u--) {
    const int e = stk[q * W + w], sq = e & 0xffff, tq = e >> 16;
    const long long val = F(u, sq);
    g[(size_t)u * inner + w] = val < EDT_INF ? (int)val : EDT_INF;
    if (u == tq) q--;
  }
}

// field = (map_dist - inv_map_dist) * cell_size; a volume without obstacles (or without free
// space) becomes the constant 1000 (signedDistanceField3D.m:30-33)
__global__ void k_edt_finish(size_t n, const int* __restrict__ a, const int* __restrict__ b, double cell,
                             double* __restrict__ field) {
  const bool degenerate = (a[0] == 0 ? b[0] : a[0]) >= EDT_INF;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double map_dist = sqrt((double)a[i]), inv_map_dist = sqrt((double)b[i]);
    field[i] = degenerate ? 1000.0 : (map_dist - inv_map_dist) * cell;
  }
}

// occ, field: [nz][ny][nx] device pointers; wa, wb: int32 scratch of the same element count
int launch_sdf_from_occupancy(int nx, int ny, int nz, const double* occ, double cell, int* wa, int* wb,
                              double* field, hipStream_t st) {
  const size_t n = (size_t)nx * ny * nz;
  const int sweep = (int)std::min<size_t>((n + 255) / 256, 256 * 16);
  k_edt_seed<<<dim3(sweep), dim3(256), 0, st>>>(n, occ, wa, wb);
  constexpr size_t kLdsBudget = 144 * 1024;
  auto strided = [&](int len, size_t inner, size_t outer) -> int {
    if (len < 2) return GPMP2MI_OK;
    if (len < 32768 && (size_t)len * 32 * 2 * sizeof(int) <= kLdsBudget) {  // linear-time scan, 32 lines per block
      k_edt_strided_scan<32><<<dim3((unsigned)((inner + 31) / 32), (unsigned)outer, 2), dim3(256),
                               (size_t)len * 32 * 2 * sizeof(int), st>>>(len, inner, wa, wb);
    } else if ((size_t)len * 64 * sizeof(int) <= kLdsBudget) {
      k_edt_strided<64><<<dim3((unsigned)((inner + 63) / 64), (unsigned)outer, 2), dim3(256),
                          ((size_t)len * 64 + 64) * sizeof(int), st>>>(len, inner, wa, wb);
    } else if ((size_t)len * 8 * sizeof(int) <= kLdsBudget) {
      k_edt_strided<8><<<dim3((unsigned)((inner + 7) / 8), (unsigned)outer, 2), dim3(256),
                         ((size_t)len * 8 + 8) * sizeof(int), st>>>(len, inner, wa, wb);
    } else {
      set_error("occupancy grid axis too long for the LDS-staged distance transform (max 4608 cells)");
      return GPMP2MI_ERR_UNSUPPORTED;
    }
    return GPMP2MI_OK;
  };
  if (nx >= 2) {
    if ((size_t)nx * sizeof(int) > kLdsBudget) {
      set_error("occupancy grid axis too long for the LDS-staged distance transform");
      return GPMP2MI_ERR_UNSUPPORTED;
    }
    const size_t nrows = (size_t)ny * nz;
    const int rows = (int)std::max<size_t>(1, std::min<size_t>(nrows, 2048 / nx));
    k_edt_x<<<dim3((unsigned)((nrows + rows - 1) / rows), 1, 2), dim3(256), ((size_t)rows * nx + rows) * sizeof(int), st>>>(
        nx, nrows, rows, wa, wb);
  }
  G2_TRY(strided(ny, (size_t)nx, (size_t)nz));
  G2_TRY(strided(nz, (size_t)nx * ny, 1));
  k_edt_finish<<<dim3(sweep), dim3(256), 0, st>>>(n, wa, wb, cell, field);
  G2_HIP(hipGetLastError());
  return GPMP2MI_OK;
}

}  // namespace g2
