// tools/bench/gemm_lds_bench.hip — development bench for the LDS-staged Schur-update kernel (r03): C[lower 64x64 tiles] -= A A^T on
// one large job, checked against k_front_gemm<64> of pgo_front_kernels.hip on the same input, then timed.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../posegraph-ceres_amd/csrc -o gemm_lds_bench gemm_lds_bench.hip
#include "../../posegraph-ceres_amd/csrc/pgo_front_kernels.hip"

#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 192;
  const int n = K + R, ld = n + 2;
  std::vector<double> h((size_t)(n + 1) * ld);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  double *F, *F2;
  hipMalloc(&F, h.size() * sizeof(double));
  hipMalloc(&F2, h.size() * sizeof(double));
  pgo::FrontJob J{0, ld, K, K + R, K, K + R, 0, K, 0};
  std::vector<int> wg_job, wg_tile;
  const int nt = (R + 63) / 64;
  for (int ti = 0; ti < nt; ++ti) for (int tj = 0; tj <= ti; ++tj) { wg_job.push_back(0); wg_tile.push_back((ti << 16) | tj); }
  pgo::FrontJob* dJ; int *dj, *dt;
  hipMalloc(&dJ, sizeof J); hipMalloc(&dj, wg_job.size() * 4); hipMalloc(&dt, wg_tile.size() * 4);
  hipMemcpy(dJ, &J, sizeof J, hipMemcpyHostToDevice);
  hipMemcpy(dj, wg_job.data(), wg_job.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dt, wg_tile.data(), wg_tile.size() * 4, hipMemcpyHostToDevice);
  pgo::FrontPlan p{};
  p.jobs = dJ; p.wg_job = dj; p.wg_tile = dt;
  // correctness: one launch each from the same input
  hipMemcpy(F, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  hipMemcpy(F2, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  p.Fval = F;
  hipLaunchKernelGGL(pgo::k_front_gemm<64>, dim3((unsigned)wg_job.size()), dim3(256), 0, 0, p, 0);
  p.Fval = F2;
  hipLaunchKernelGGL(pgo::k_front_gemm_lds, dim3((unsigned)wg_job.size()), dim3(256), 0, 0, p, 0);
  std::vector<double> a(h.size()), b(h.size());
  hipMemcpy(a.data(), F, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), F2, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (int i = K; i < K + R; ++i) for (int j = K; j <= i; ++j) { md = std::fmax(md, std::fabs(a[(size_t)i * ld + j] - b[(size_t)i * ld + j])); mx = std::fmax(mx, std::fabs(a[(size_t)i * ld + j])); }
  std::printf("R %d K %d: max |direct - lds| = %.3e (max |C| %.3e)\n", R, K, md, mx);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 10;
  for (int which = 0; which < 2; ++which) {
    float ms = 0;
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        if (which == 0) hipLaunchKernelGGL(pgo::k_front_gemm<64>, dim3((unsigned)wg_job.size()), dim3(256), 0, 0, p, 0);
        else hipLaunchKernelGGL(pgo::k_front_gemm_lds, dim3((unsigned)wg_job.size()), dim3(256), 0, 0, p, 0);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)wg_job.size() * 2.0 * 64 * 64 * K;
    std::printf("  %s: %zu tiles, %.1f us per launch, %.2f TFLOP/s\n", which ? "lds   " : "direct", wg_job.size(), 1e3 * ms / reps, flops * reps / (ms * 1e-3) / 1e12);
  }
  return 0;
}
