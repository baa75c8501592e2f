// pgo_candidates.hip — loop-closure candidate search on the GPU.
//
// Role of generate_edges_from_trajectory_origion.cpp:58-111 (getCandidatesIndex / isInSearchRange): for every frame
// k >= 1 the candidate list is [k-1] followed by every earlier frame i < k - gap whose FLOAT32 squared distance
// ((dx*dx + dy*dy) + dz*dz, each operation rounded, no fused multiply-add) to frame k is <= radius^2, in ascending
// i.  The result is index data, so parity with the CPU generator is bit-exact (tests/test_gpu_candidates.py).
//
// One lane per frame k; the earlier frames stream through LDS in tiles of 256 positions (each position is read from
// HBM once per workgroup), the lane scans the tile in ascending order.  Two passes over the same code: count, then
// fill at the offsets of the exclusive scan — deterministic output order, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/pgo.h"

namespace {

constexpr int CAND_BLOCK = 256;

template <bool FILL>
__global__ __launch_bounds__(CAND_BLOCK) void k_candidates(const float* __restrict__ xyz, int n, float r2, int gap,
                                                           int* __restrict__ counts, const long long* __restrict__ row_ptr,
                                                           int* __restrict__ indices) {
  __shared__ float sx[CAND_BLOCK], sy[CAND_BLOCK], sz[CAND_BLOCK];
  const int tid = threadIdx.x;
  const int k = blockIdx.x * CAND_BLOCK + tid;
  const bool live = k >= 1 && k < n;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (live) { cx = xyz[3 * (size_t)k]; cy = xyz[3 * (size_t)k + 1]; cz = xyz[3 * (size_t)k + 2]; }
  const int hi = live ? k - gap : 0;                                   // candidates are i < hi
  const int block_hi = min(n, blockIdx.x * CAND_BLOCK + CAND_BLOCK) - 1 - gap;   // largest hi inside this workgroup
  long long w = 0;
  int c = 0;
  if (live) {
    if (FILL) { w = row_ptr[k]; indices[w++] = k - 1; }
    c = 1;
  }
  for (int i0 = 0; i0 < block_hi; i0 += CAND_BLOCK) {
    const int i = i0 + tid;
    if (i < n) { sx[tid] = xyz[3 * (size_t)i]; sy[tid] = xyz[3 * (size_t)i + 1]; sz[tid] = xyz[3 * (size_t)i + 2]; }
    __syncthreads();
    const int jn = min(CAND_BLOCK, hi - i0);
    for (int j = 0; j < jn; ++j) {
      const float dx = __fsub_rn(sx[j], cx), dy = __fsub_rn(sy[j], cy), dz = __fsub_rn(sz[j], cz);
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      if (!(d > r2)) {
        if (FILL) indices[w++] = i0 + j;
        ++c;
      }
    }
    __syncthreads();
  }
  if (!FILL && k < n) counts[k] = live ? c : 0;
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
};

}  // namespace

int pgo_candidates_set_error(int code, const char* msg);   // pgo_problem.cpp

#define CAND_TRY(expr)                                                              \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) return pgo_candidates_set_error(PGO_ERR_HIP, hipGetErrorString(e_)); \
  } while (0)

extern "C" int pgo_generate_candidates(const float* xyz, int n, float search_radius, int gap, long long* row_ptr, int* indices,
                                       long long capacity, double* kernel_ms) {
  if (!xyz || n < 0 || gap < 0 || !row_ptr) return pgo_candidates_set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_generate_candidates");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return pgo_candidates_set_error(PGO_ERR_NO_DEVICE, "no HIP device: the candidate search has no CPU fallback");
  }
  row_ptr[0] = 0;
  if (n == 0) return PGO_OK;
  const float r2 = search_radius * search_radius;   // float product, as the reference computes it
  Buf d_xyz, d_cnt, d_ptr, d_idx;
  CAND_TRY(d_xyz.alloc(sizeof(float) * 3 * (size_t)n));
  CAND_TRY(d_cnt.alloc(sizeof(int) * (size_t)n));
  CAND_TRY(hipMemcpy(d_xyz.p, xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
  const int grid = (n + CAND_BLOCK - 1) / CAND_BLOCK;
  hipEvent_t e0, e1;
  CAND_TRY(hipEventCreate(&e0));
  CAND_TRY(hipEventCreate(&e1));
  CAND_TRY(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_candidates<false>, dim3(grid), dim3(CAND_BLOCK), 0, 0, (const float*)d_xyz.p, n, r2, gap, (int*)d_cnt.p,
                     (const long long*)nullptr, (int*)nullptr);
  CAND_TRY(hipEventRecord(e1, 0));
  std::vector<int> cnt(n);
  CAND_TRY(hipMemcpy(cnt.data(), d_cnt.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
  float ms_count = 0.f, ms_fill = 0.f;
  CAND_TRY(hipEventSynchronize(e1));
  CAND_TRY(hipEventElapsedTime(&ms_count, e0, e1));
  for (int k = 0; k < n; ++k) row_ptr[k + 1] = row_ptr[k] + cnt[k];
  if (indices) {
    if (capacity < row_ptr[n]) {
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      return pgo_candidates_set_error(PGO_ERR_INVALID_ARGUMENT, "indices capacity smaller than the candidate count (call with indices = NULL first)");
    }
    CAND_TRY(d_ptr.alloc(sizeof(long long) * ((size_t)n + 1)));
    CAND_TRY(d_idx.alloc(sizeof(int) * (size_t)row_ptr[n]));
    CAND_TRY(hipMemcpy(d_ptr.p, row_ptr, sizeof(long long) * ((size_t)n + 1), hipMemcpyHostToDevice));
    CAND_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_candidates<true>, dim3(grid), dim3(CAND_BLOCK), 0, 0, (const float*)d_xyz.p, n, r2, gap, (int*)nullptr,
                       (const long long*)d_ptr.p, (int*)d_idx.p);
    CAND_TRY(hipEventRecord(e1, 0));
    CAND_TRY(hipMemcpy(indices, d_idx.p, sizeof(int) * (size_t)row_ptr[n], hipMemcpyDeviceToHost));
    CAND_TRY(hipEventSynchronize(e1));
    CAND_TRY(hipEventElapsedTime(&ms_fill, e0, e1));
  }
  CAND_TRY(hipGetLastError());
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (kernel_ms) *kernel_ms = (double)ms_count + (double)ms_fill;
  return PGO_OK;
}
