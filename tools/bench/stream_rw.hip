// tools/bench/stream_rw.hip — what the HBM of this box sustains for the access shapes the linearisation has: streaming reads,
// streaming writes, a copy, and the block store pattern of the BSR tiles (14 x 16 B per lane at 1 KiB stride).  HIP events, 20
// repetitions after 3 warm-ups, buffers larger than the 256 MiB Infinity Cache.
// build: hipcc -O3 --offload-arch=gfx950 stream_rw.hip -o stream_rw ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_read(const double2* __restrict__ a, double* sink, size_t n) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e-300) *sink = s;
}
__global__ void k_write(double2* __restrict__ a, size_t n, double x) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = double2{x, x + 1.0};
}
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// one lane per slot: 14 pairs at 64-pair stride inside an 18 KiB tile (the packed BSR slot), NT = nontemporal stores
template <bool READ, bool WRITE, bool NT>
__global__ __launch_bounds__(256) void k_tile(const double2* __restrict__ a, double2* __restrict__ b, int n_slots, double* sink) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_slots) return;
  const size_t base = (size_t)(t >> 6) * 1152 + (t & 63);
  double2 v[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) v[k] = READ ? a[base + (size_t)k * 64] : double2{(double)t, (double)k};
  if (WRITE) {
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      if (NT) { __builtin_nontemporal_store(v[k].x, &b[base + (size_t)k * 64].x); __builtin_nontemporal_store(v[k].y, &b[base + (size_t)k * 64].y); }
      else b[base + (size_t)k * 64] = v[k];
    }
  } else {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 14; ++k) s += v[k].x + v[k].y;
    if (s == 1.2345e-300) *sink = s;
  }
}

int main() {
  const int n_slots = 1400000;                            // the stored blocks of BASELINE configs[3]
  const size_t tile_pairs = (size_t)((n_slots + 63) / 64) * 1152;
  const size_t n = tile_pairs;                            // pairs (16 B) per buffer: 403 MB
  double2 *a, *b;
  double* sink;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms / 20 * 1e3;
  };
  for (int grid : {1024, 2048, 4096, 16384}) {
    const double tr = time([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, sink, n); });
    const double tw = time([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, 1.0); });
    const double tc = time([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    std::printf("grid %5d x 256, %.0f MB per buffer: read %.1f us %.2f TB/s | write %.1f us %.2f TB/s | copy %.1f us %.2f TB/s (read + write)\n", grid, n * 16 / 1e6,
                tr, n * 16 / tr / 1e6, tw, n * 16 / tw / 1e6, tc, 2.0 * n * 16 / tc / 1e6);
  }
  const dim3 g((n_slots + 255) / 256), blk(256);
  const double bytes = (double)n_slots * 224;
  const double t1 = time([&] { hipLaunchKernelGGL((k_tile<true, false, false>), g, blk, 0, 0, a, b, n_slots, sink); });
  const double t2 = time([&] { hipLaunchKernelGGL((k_tile<false, true, false>), g, blk, 0, 0, a, b, n_slots, sink); });
  const double t3 = time([&] { hipLaunchKernelGGL((k_tile<false, true, true>), g, blk, 0, 0, a, b, n_slots, sink); });
  const double t4 = time([&] { hipLaunchKernelGGL((k_tile<true, true, false>), g, blk, 0, 0, a, b, n_slots, sink); });
  const double t5 = time([&] { hipLaunchKernelGGL((k_tile<true, true, true>), g, blk, 0, 0, a, b, n_slots, sink); });
  std::printf("tile pattern, %d slots x 224 B: read %.1f us %.2f TB/s | write %.1f us %.2f TB/s | write nt %.1f us %.2f TB/s | copy %.1f us %.2f TB/s | copy nt %.1f us %.2f TB/s\n",
              n_slots, t1, bytes / t1 / 1e6, t2, bytes / t2 / 1e6, t3, bytes / t3 / 1e6, t4, 2 * bytes / t4 / 1e6, t5, 2 * bytes / t5 / 1e6);
  return 0;
}
