// tools/bench/fma_rate.hip — FP64 rates on gfx950: v_fma_f64 (VALU), v_mfma_f64_4x4x4_4b, v_mfma_f64_16x16x4 (development measurement).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
  double acc[16];
  for (int q = 0; q < 16; ++q) acc[q] = q;
  const double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = fma(acc[q], a, b);
  }
  double s = 0;
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_m4(double* out, int iters) {
  double acc[8];
  for (int q = 0; q < 8; ++q) acc[q] = 0;
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
  }
  double s = 0;
  for (int q = 0; q < 8; ++q) s += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_m16(double* out, int iters) {
  double4_t acc[4];
  for (int q = 0; q < 4; ++q) acc[q] = double4_t{0, 0, 0, 0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  double s = 0;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int blocks, double flops_per_thread_iter, int iters) {
  double* out; hipMalloc(&out, 8 * 4096 * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-8s blocks %5d: %.3f ms  %.2f TFLOP/s\n", name, blocks, ms, flops_per_thread_iter * iters * blocks * 256.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int blocks : {1, 256, 1024, 2048}) {
    run("fma", k_fma, blocks, 16 * 2.0, 4000);
    run("mfma4x4", k_m4, blocks, 8 * 512.0 / 64, 4000);      // 4 blocks of 4x4x4: 512 flops per wave instruction
    run("mfma16", k_m16, blocks, 4 * 2048.0 / 64, 2000);
  }
  return 0;
}
