// tools/bench/l2_persist.hip — does a per-XCD L2 (4 MiB) keep a read-only working set from one launch to the next?
// The C2 CG product reads the same blocks in every launch with the same grid (work-group b -> XCD b % 8, observed).  If clean
// lines survive the kernel boundary, a form whose per-XCD share fits the L2 is served there instead of by the fabric.
// Each work-group reads the same contiguous chunk in every launch (16 B per lane, 18 loads in flight like a BSR slot); between
// two reads an optional "vector" kernel writes 480 KB (what k_uni_v does between two products).  Reported per size: us per read
// launch from HIP events over 200 back-to-back launches; run under `rocprofv3 --pmc FETCH_SIZE` for the fabric bytes.
// build: hipcc -O3 --offload-arch=gfx950 l2_persist.hip -o l2_persist ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// work-group b reads pairs [b * per_wg, (b + 1) * per_wg)
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, double* sink, int per_wg) {
  const double2* p = a + (size_t)blockIdx.x * per_wg;
  double s = 0;
  for (int i = threadIdx.x; i < per_wg; i += 256 * 6) {
    double2 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { const int j = i + k * 256; v[k] = j < per_wg ? p[j] : double2{0, 0}; }
#pragma unroll
    for (int k = 0; k < 6; ++k) s += v[k].x + v[k].y;
  }
  if (s == 1.2345e-300) *sink = s;
}
__global__ void k_vec(double* __restrict__ v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = v[i] * 0.999 + 1.0;
}

int main(int argc, char** argv) {
  const int reps = 200;
  const int n_wg = argc > 1 ? std::atoi(argv[1]) : 512;
  double2* a;
  double *sink, *vec;
  const size_t max_bytes = (size_t)96 << 20;
  CK(hipMalloc(&a, max_bytes)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&vec, 60000 * 8));
  CK(hipMemset(a, 0, max_bytes)); CK(hipMemset(vec, 0, 60000 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::printf("work-groups %d (256 lanes); us per read launch, read-only stream | with a 480 KB vector kernel between reads (its own time subtracted)\n", n_wg);
  // the vector kernel alone
  float ms_vec = 0;
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_vec, dim3(235), dim3(256), 0, 0, vec, 60000);
    hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_vec, e0, e1));
  }
  const double us_vec = 1e3 * ms_vec / reps;
  std::printf("vector kernel alone: %.2f us per launch\n", us_vec);
  const double sizes_mb[] = {2, 4, 8, 12, 15.4, 20, 24, 26, 28, 32, 40, 48, 64, 96};
  for (double mb : sizes_mb) {
    const int per_wg = (int)(mb * 1048576.0 / 16.0 / n_wg);
    float ms0 = 0, ms1 = 0;
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(n_wg), dim3(256), 0, 0, a, sink, per_wg);
      hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
    }
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_read, dim3(n_wg), dim3(256), 0, 0, a, sink, per_wg);
        hipLaunchKernelGGL(k_vec, dim3(235), dim3(256), 0, 0, vec, 60000);
      }
      hipEventRecord(e1, 0); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
    }
    const double us0 = 1e3 * ms0 / reps, us1 = 1e3 * ms1 / reps - us_vec;
    std::printf("%6.1f MB (%5.2f MB per XCD): %6.2f us = %6.2f TB/s | %6.2f us = %6.2f TB/s\n", mb, mb / 8, us0, mb * 1.048576 / us0, us1,
                mb * 1.048576 / us1);
  }
  return 0;
}
