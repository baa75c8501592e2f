// tools/bench/mfma_rate.hip — issue rate of v_mfma_f64_16x16x4_f64 on one SIMD / the whole chip (development measurement).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc) {
  double4_t acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> void run(int blocks, int threads) {
  double* out; long long* cyc; hipMalloc(&out, 8 * 1024 * 2048); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * NACC;
  printf("NACC %d blocks %4d threads %4d: %.1f ticks per MFMA per wave, %.3f ms, %.2f TFLOP/s\n", NACC, blocks, threads, c / n_mfma, ms,
         n_mfma * 2048.0 * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(1, 64); run<2>(1, 64); run<4>(1, 64); run<8>(1, 64);
  run<4>(1, 256); run<4>(256, 256); run<4>(512, 256); run<4>(768, 256); run<4>(1024, 256); run<4>(2048, 512); run<8>(1024, 256); run<16>(256, 256); run<16>(512, 256);
  return 0;
}
