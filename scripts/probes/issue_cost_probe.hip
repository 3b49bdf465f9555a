// issue_cost_probe.hip -- measured issue cost (shader cycles per wave-instruction) of the instructions the tile
// elimination is built from, for one wave alone on a SIMD and for 2 / 4 waves per SIMD.  Each test is a loop of
// UNROLL copies of one instruction on independent (or, "dep", one dependent) register chains, timed with s_memtime.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/issue_cost_probe.hip -o build/probes/issue_cost_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int ITERS = 200;

#define OPS8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int T>
__device__ __forceinline__ void body(double (&a)[8], double (&b)[8], double (&c)[8], int lane, double* lds) {
  if constexpr (T == 0) {          // v_fma_f64, 8 independent chains
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 1) {   // v_fma_f64, one dependent chain
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[0]) : "v"(b[i]), "v"(c[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 2) {   // v_mul_f64
#define X(i) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 3) {   // v_mov_b32_dpp row_newbcast
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(((int*)&a[i])[0]) : "v"(((int*)&b[i])[0]));
    OPS8(X)
#undef X
  } else if constexpr (T == 4) {   // v_mov_b64_dpp row_newbcast
#define X(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 5) {   // v_fmac_f64_dpp row_newbcast
#define X(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 6) {   // ds_bpermute_b32 (+ one wait per 8)
    int idx = (lane ^ 16) * 4;
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(((int*)&a[i])[0]) : "v"(idx), "v"(((int*)&b[i])[0]));
    OPS8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if constexpr (T == 7) {   // v_readlane_b32 -> sgpr
    int s;
#define X(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(((int*)&b[i])[0])); ((int*)&a[i])[1] = s;
    OPS8(X)
#undef X
  } else if constexpr (T == 8) {   // v_cndmask_b32
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(((int*)&a[i])[0]) : "v"(((int*)&b[i])[0]), "v"(((int*)&c[i])[0]) : "vcc");
    OPS8(X)
#undef X
  } else if constexpr (T == 9) {   // v_rcp_f64
#define X(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(b[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 10) {  // v_rsq_f64
#define X(i) asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(b[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 11) {  // v_mfma_f64_16x16x4, 2 independent accumulators, 8 instructions
    v4d acc0 = {a[0], a[1], a[2], a[3]}, acc1 = {a[4], a[5], a[6], a[7]};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[i], c[i], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[i + 4], c[i + 4], acc1, 0, 0, 0);
    }
    a[0] = acc0[0]; a[1] = acc0[1]; a[2] = acc0[2]; a[3] = acc0[3];
    a[4] = acc1[0]; a[5] = acc1[1]; a[6] = acc1[2]; a[7] = acc1[3];
  } else if constexpr (T == 12) {  // v_mfma_f64_16x16x4, one dependent accumulator, 8 instructions
    v4d acc0 = {a[0], a[1], a[2], a[3]};
#pragma unroll
    for (int i = 0; i < 8; i++) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[i], c[i], acc0, 0, 0, 0);
    a[0] = acc0[0]; a[1] = acc0[1]; a[2] = acc0[2]; a[3] = acc0[3];
  } else if constexpr (T == 13) {  // v_permlane32_swap
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&a[i])[0]), "+v"(((int*)&b[i])[0]));
    OPS8(X)
#undef X
  } else if constexpr (T == 14) {  // v_add_f64
#define X(i) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 15) {  // v_fma_f32 (reference point)
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(((float*)&a[i])[0]) : "v"(((float*)&b[i])[0]), "v"(((float*)&c[i])[0]));
    OPS8(X)
#undef X
  } else if constexpr (T == 16) {  // ds_read_b64 (+ wait per 8)
#define X(i) a[i] = lds[lane + 64 * i];
    OPS8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if constexpr (T == 17) {  // ds_write_b64
#define X(i) lds[lane + 64 * i] = b[i];
    OPS8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  } else if constexpr (T == 18) {  // v_pk_fma_f32 (packed: 128 fp32 FMA per instruction)
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
    OPS8(X)
#undef X
  } else if constexpr (T == 19) {  // v_fma_f64 with an SGPR-pair operand
    double s = __builtin_bit_cast(double, ((long)__builtin_amdgcn_readfirstlane(((int*)&b[0])[1]) << 32));
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "s"(s), "v"(c[i]));
    OPS8(X)
#undef X
  }
}

template <int T>
__global__ __launch_bounds__(1024) void k_cost(const double* in, double* out, long long* cycles) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a[8], b[8], c[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a[i] = in[threadIdx.x + 64 * i];
    b[i] = in[threadIdx.x + 64 * (i + 8)] + 1.5;
    c[i] = in[threadIdx.x + 64 * (i + 16)] * 1e-3;
  }
  double* mylds = lds + (size_t)w * 512;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) body<T>(a, b, c, lane, mylds);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
}

template <int T>
void run(const char* name, const double* din, double* dout, long long* dcyc) {
  printf("%-44s", name);
  for (int wpb : {4, 8, 16}) {   // 1, 2, 4 waves per SIMD (one block per CU)
    k_cost<T><<<256, 64 * wpb, wpb * 512 * 8>>>(din, dout, dcyc);
    k_cost<T><<<256, 64 * wpb, wpb * 512 * 8>>>(din, dout, dcyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf(" FAILED\n"); return; }
    std::vector<long long> cyc(256 * wpb);
    (void)hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    std::sort(cyc.begin(), cyc.end());
    const double per = cyc[cyc.size() / 2] / (double)(ITERS * 8);
    // per = wall cycles per instruction of ONE wave; SIMD cycles per instruction = per / waves-per-SIMD
    printf("  %dw/SIMD: %6.1f cyc/instr/wave (%5.1f SIMD-cyc)", wpb / 4, per, per / (wpb / 4));
  }
  printf("\n");
}

int main() {
  double *din, *dout;
  long long* dcyc;
  (void)hipMalloc(&din, 1024 * 24 * 8 + 4096);
  (void)hipMalloc(&dout, 256 * 1024 * 8);
  (void)hipMalloc(&dcyc, 256 * 16 * 8);
  std::vector<double> h(1024 * 24 + 512);
  for (auto& x : h) x = rand() / (double)RAND_MAX + 0.5;
  (void)hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  run<0>("v_fma_f64 (8 independent chains)", din, dout, dcyc);
  run<1>("v_fma_f64 (one dependent chain)", din, dout, dcyc);
  run<2>("v_mul_f64", din, dout, dcyc);
  run<14>("v_add_f64", din, dout, dcyc);
  run<19>("v_fma_f64 with SGPR operand", din, dout, dcyc);
  run<15>("v_fma_f32", din, dout, dcyc);
  run<18>("v_pk_fma_f32", din, dout, dcyc);
  run<3>("v_mov_b32_dpp row_newbcast", din, dout, dcyc);
  run<4>("v_mov_b64_dpp row_newbcast", din, dout, dcyc);
  run<5>("v_fmac_f64_dpp row_newbcast", din, dout, dcyc);
  run<6>("ds_bpermute_b32 (wait per 8)", din, dout, dcyc);
  run<7>("v_readlane_b32 + v_mov from sgpr", din, dout, dcyc);
  run<8>("v_cndmask_b32", din, dout, dcyc);
  run<13>("v_permlane32_swap", din, dout, dcyc);
  run<9>("v_rcp_f64", din, dout, dcyc);
  run<10>("v_rsq_f64", din, dout, dcyc);
  run<11>("v_mfma_f64_16x16x4 (2 accumulators)", din, dout, dcyc);
  run<12>("v_mfma_f64_16x16x4 (dependent)", din, dout, dcyc);
  run<16>("ds_read_b64 (wait per 8)", din, dout, dcyc);
  run<17>("ds_write_b64 (wait per 8)", din, dout, dcyc);
  return 0;
}
