// tools/bench/gemm_bench.hip — the Schur-update kernel of the multifrontal solver (k_front_gemm of pgo_front_kernels.hip, the
// translation unit is included as it is) on ONE large job in isolation: C[R x R, lower tiles] -= A[R x K] A^T.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../posegraph-ceres_amd/csrc -o gemm_bench gemm_bench.hip
// usage: gemm_bench [R] [K]
#include "../../posegraph-ceres_amd/csrc/pgo_front_kernels.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 192;
  const int n = K + R, ld = n + 2;
  std::vector<double> h((size_t)(n + 1) * ld);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  double* F;
  hipMalloc(&F, h.size() * sizeof(double));
  hipMemcpy(F, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  for (int T : {64, 32}) {
    pgo::FrontJob J{0, ld, K, K + R, K, K + R, 0, K, 0};
    std::vector<int> wg_job, wg_tile;
    const int nt = (R + T - 1) / T;
    for (int ti = 0; ti < nt; ++ti) for (int tj = 0; tj <= ti; ++tj) { wg_job.push_back(0); wg_tile.push_back((ti << 16) | tj); }
    pgo::FrontJob* dJ; int *dj, *dt;
    hipMalloc(&dJ, sizeof J); hipMalloc(&dj, wg_job.size() * 4); hipMalloc(&dt, wg_tile.size() * 4);
    hipMemcpy(dJ, &J, sizeof J, hipMemcpyHostToDevice);
    hipMemcpy(dj, wg_job.data(), wg_job.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dt, wg_tile.data(), wg_tile.size() * 4, hipMemcpyHostToDevice);
    pgo::FrontPlan p{};
    p.jobs = dJ; p.wg_job = dj; p.wg_tile = dt; p.Fval = F;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        if (T == 64) hipLaunchKernelGGL(pgo::k_front_gemm<64>, dim3((unsigned)wg_job.size()), dim3(256), 0, 0, p, 0);
        else hipLaunchKernelGGL(pgo::k_front_gemm<32>, dim3((unsigned)wg_job.size()), dim3(256), 0, 0, p, 0);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wg_job.size() * 2.0 * T * T * K;
    std::printf("tile %d: R %d K %d, %zu tiles, %.1f us per launch, %.2f TFLOP/s (tiles computed in full)\n", T, R, K, wg_job.size(), 1e3 * ms / reps, flops * reps / (ms * 1e-3) / 1e12);
    hipFree(dJ); hipFree(dj); hipFree(dt);
  }
  return 0;
}
