// tools/bench/two_streams.hip — does this stack run small kernels of SEVERAL streams concurrently?  T host threads, one
// non-blocking stream each, K dependent launches of a small kernel per thread; prints microseconds per launch per thread.
// build: hipcc --offload-arch=gfx950 -O3 -o two_streams two_streams.hip -lpthread      usage: two_streams [K] [blocks]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_small(double* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * 1.0000001 + 1e-9;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 2000, blocks = argc > 2 ? atoi(argv[2]) : 512;
  for (int T : {1, 2, 4, 8}) {
    std::vector<hipStream_t> st(T);
    std::vector<double*> buf(T);
    for (int t = 0; t < T; ++t) {
      hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
      hipMalloc(&buf[t], sizeof(double) * blocks * 128);
      hipMemsetAsync(buf[t], 0, sizeof(double) * blocks * 128, st[t]);
      hipStreamSynchronize(st[t]);
    }
    for (int mode = 0; mode < 2; ++mode) {   // 0: one host thread per stream, 1: ONE host thread feeding all streams round robin
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
          th.emplace_back([&, t] {
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(128), 0, st[t], buf[t], blocks * 128);
            hipStreamSynchronize(st[t]);
          });
        for (auto& x : th) x.join();
      } else {
        for (int k = 0; k < K; ++k)
          for (int t = 0; t < T; ++t) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(128), 0, st[t], buf[t], blocks * 128);
        for (int t = 0; t < T; ++t) hipStreamSynchronize(st[t]);
      }
      const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
      std::printf("%d streams, %s: %.2f us per launch per stream, %.2f us per launch overall\n", T, mode ? "one feeder thread" : "thread per stream", us / K, us / K / T);
    }
    for (int t = 0; t < T; ++t) { hipFree(buf[t]); hipStreamDestroy(st[t]); }
  }
  return 0;
}
