// tools/bench/phase_barrier.hip — what does a grid-wide phase boundary cost INSIDE one launch on MI355X, against a kernel
// boundary?  P phases of G work-groups each in one grid (G x P tickets, ticket order); a work-group of phase p waits until all G
// work-groups of phase p - 1 have added themselves to that phase's counter (one lane polls, one wave acquires / releases at
// agent scope — the hand-over of k_sfront_factor / k_front_stages), touches `bytes` of a buffer (read-modify-write, so that the
// fences have something to write back and invalidate), and adds itself to its own phase's counter.  Compared with P launches
// of G work-groups doing the same touch.   build: hipcc --offload-arch=gfx950 -O3 -o phase_barrier phase_barrier.hip
// usage: phase_barrier [G] [P] [doubles per work-group]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_phases(double* buf, unsigned long long* cnt, int G, int P, int per, int poll_mode) {
  __shared__ int tk;
  if (threadIdx.x == 0) tk = (int)atomicAdd(cnt + P, 1ull);
  __syncthreads();
  const int phase = tk / G, w = tk - phase * G;
  if (phase > 0 && threadIdx.x == 0) {
    const unsigned long long target = (unsigned long long)G;
    while (__hip_atomic_load(cnt + phase - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (poll_mode == 1) __builtin_amdgcn_s_sleep(1); else if (poll_mode == 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  double* b = buf + (size_t)w * per;
  for (int i = threadIdx.x; i < per; i += 256) b[i] = b[i] * 1.0000001 + 1e-9;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    atomicAdd(cnt + phase, 1ull);
  }
}
__global__ __launch_bounds__(256) void k_one(double* buf, int per) {
  double* b = buf + (size_t)blockIdx.x * per;
  for (int i = threadIdx.x; i < per; i += 256) b[i] = b[i] * 1.0000001 + 1e-9;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 703, P = argc > 2 ? atoi(argv[2]) : 64, per = argc > 3 ? atoi(argv[3]) : 4096;
  double* buf; unsigned long long* cnt;
  (void)hipMalloc(&buf, sizeof(double) * (size_t)G * per);
  (void)hipMalloc(&cnt, sizeof(unsigned long long) * (P + 1));
  (void)hipMemset(buf, 0, sizeof(double) * (size_t)G * per);
  hipStream_t s; (void)hipStreamCreate(&s);
  for (int poll : {1, 8, 32}) {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipMemsetAsync(cnt, 0, sizeof(unsigned long long) * (P + 1), s);
      (void)hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_phases, dim3(G * P), dim3(256), 0, s, buf, cnt, G, P, per, poll);
      (void)hipStreamSynchronize(s);
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    std::printf("one launch, %d phases x %d work-groups x %d doubles, s_sleep(%d): %.2f us per phase\n", P, G, per, poll, 1e6 * best / P);
  }
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < P; ++p) hipLaunchKernelGGL(k_one, dim3(G), dim3(256), 0, s, buf, per);
    (void)hipStreamSynchronize(s);
    best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
  std::printf("%d launches of %d work-groups: %.2f us per launch\n", P, G, 1e6 * best / P);
  return 0;
}
