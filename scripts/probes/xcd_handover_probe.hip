// Does data written by one kernel stay in the writing XCD's L2 for the next kernel?  Kernel W: workgroup i writes a
// pointer chain into its 64-KB chunk.  Kernel R(shift): workgroup i chases the chain of chunk (i + shift) % n with one
// lane (dependent 128-B-line loads).  shift 0 / 8 / 16: same XCD under round-robin placement (same CU slot or not),
// shift 1 / 3: another XCD.  Prints the mean load-to-load latency per variant (device time / steps).
// build: hipcc --offload-arch=gfx950 -O2 scripts/probes/xcd_handover_probe.hip -o gpurun_out/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int LINES = 512, STRIDE = 32;  // 512 lines of 128 B = 64 KB per chunk
__global__ void kw(unsigned* buf, int salt) {
  unsigned* c = buf + (size_t)blockIdx.x * LINES * STRIDE;
  for (int j = threadIdx.x; j < LINES; j += blockDim.x) c[(size_t)j * STRIDE] = (unsigned)((j * 37 + 11 + salt) % LINES);
}
__global__ void kr(const unsigned* buf, int shift, int steps, unsigned* out, long long* cyc) {
  const int n = gridDim.x;
  const unsigned* c = buf + (size_t)((blockIdx.x + shift) % n) * LINES * STRIDE;
  if (threadIdx.x == 0) {
    unsigned cur = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) cur = c[(size_t)cur * STRIDE];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x] = cur;
    cyc[blockIdx.x] = t1 - t0;
  }
}
int main() {
  const int n = 256, steps = 256;
  unsigned *buf, *out;
  long long* cyc;
  hipMalloc(&buf, (size_t)n * LINES * STRIDE * 4);
  hipMalloc(&out, n * 4);
  hipMalloc(&cyc, n * 8);
  std::vector<long long> h(n);
  const int shifts[] = {0, 8, 16, 1, 3, 0, 1};
  for (int rep = 0; rep < 2; rep++)
    for (int sh : shifts) {
      hipLaunchKernelGGL(kw, dim3(n), dim3(256), 0, 0, buf, rep * 7 + sh);
      hipLaunchKernelGGL(kr, dim3(n), dim3(64), 0, 0, buf, sh, steps, out, cyc);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, n * 8, hipMemcpyDeviceToHost);
      double m = 0;
      for (auto v : h) m += (double)v;
      printf("rep %d shift %2d: %.0f cycles per dependent load (mean over %d workgroups)\n", rep, sh, m / n / steps, n);
    }
  // the same chain re-read by a second R kernel without a writer in between (L2 / MALL warm from the reader itself)
  hipLaunchKernelGGL(kr, dim3(n), dim3(64), 0, 0, buf, 0, steps, out, cyc);
  hipLaunchKernelGGL(kr, dim3(n), dim3(64), 0, 0, buf, 0, steps, out, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), cyc, n * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (auto v : h) m += (double)v;
  printf("re-read by the same workgroups, previous kernel a reader: %.0f cycles per load\n", m / n / steps);
  return 0;
}
