// tools/bench/mfma_clock.hip — what bounds v_mfma_f64_16x16x4_f64 on MI355X: cycles per instruction or the clock under load?
// Every wave runs a loop of NACC independent MFMAs and reads BOTH counters around it: s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz).  Printed per configuration: shader cycles per MFMA per SIMD, the shader clock the two counters
// imply, and the chip's TFLOP/s — for runs of ~0.1 ms, ~2 ms and ~50 ms (a power or thermal limit shows as a clock that sags with
// the length of the run; an issue limit as a cycle count that does not depend on it).  Development measurement (r03 verdict item 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* stamps) {
  double4_t acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0;
  }
}
template <int NACC> void run(int blocks, int threads, int iters) {
  const size_t waves = (size_t)blocks * threads / 64;
  double* out; long long* st;
  hipMalloc(&out, 8 * (size_t)blocks * threads); hipMalloc(&st, 16 * waves);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 200, st);     // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, st);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(2 * waves);
  hipMemcpy(h.data(), st, 16 * waves, hipMemcpyDeviceToHost);
  std::vector<double> cyc(waves), mhz(waves);
  for (size_t w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; mhz[w] = 100.0 * (double)h[2 * w] / (double)std::max(1LL, h[2 * w + 1]); }
  std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
  const double n_mfma = (double)iters * NACC;
  const int wps = std::max(1, (int)(waves / 1024));     // waves per SIMD when the grid fills the chip (1024 SIMDs)
  printf("NACC %2d grid %5d x %3d (%d wave(s) per SIMD) iters %7d: %7.3f ms | s_memtime ticks per MFMA per wave: median %.1f | ticks/100MHz-tick -> %.0f MHz (min %.0f max %.0f) | %.2f TFLOP/s\n",
         NACC, blocks, threads, wps, iters, ms, cyc[waves / 2] / n_mfma, mhz[waves / 2], mhz.front(), mhz.back(),
         n_mfma * 2048.0 * (double)waves / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(st);
}
int main() {
  for (int iters : {500, 10000, 250000}) { run<4>(256, 256, iters); run<4>(512, 256, iters); run<4>(1024, 256, iters); }
  run<4>(1, 64, 20000);           // one wave on one SIMD: the instruction alone
  run<1>(1, 64, 20000);           // dependent chain
  return 0;
}
