// pgo_wave.h — wave / work-group reductions shared by the kernel files (gfx950: 64-lane waves, DPP crossbar).
#pragma once
#include <hip/hip_runtime.h>

namespace pgo {
namespace {

// Wave-wide reductions on the DPP crossbar (row_shr 1/2/4/8 inside each 16-lane row, then
// row_bcast:15 / row_bcast:31 across rows): pure VALU, no LDS round trips, total lands in lane 63
// and is broadcast with v_readlane.  (A __shfl_down ladder compiles to ds_bpermute + s_waitcnt per
// step: measured ~2 us for the 7-value CG prologue)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shifted(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_shifted<0x111, 0xf>(v);
  v += dpp_shifted<0x112, 0xf>(v);
  v += dpp_shifted<0x114, 0xf>(v);
  v += dpp_shifted<0x118, 0xf>(v);
  v += dpp_shifted<0x142, 0xa>(v);
  v += dpp_shifted<0x143, 0xc>(v);
  return lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {  // v >= 0 everywhere it is used
  v = fmax(v, dpp_shifted<0x111, 0xf>(v));
  v = fmax(v, dpp_shifted<0x112, 0xf>(v));
  v = fmax(v, dpp_shifted<0x114, 0xf>(v));
  v = fmax(v, dpp_shifted<0x118, 0xf>(v));
  v = fmax(v, dpp_shifted<0x142, 0xa>(v));
  v = fmax(v, dpp_shifted<0x143, 0xc>(v));
  return lane63(v);
}

// Sum NV values over the workgroup in a fixed order; every thread gets the totals.
// scratch: >= NV * (blockDim/64) doubles of LDS.  Ends with a barrier so scratch can be reused.
template <int NV>
__device__ __forceinline__ void block_sum_w(double (&v)[NV], double* scratch, int nw);
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
  block_sum_w<NV>(v, scratch, (blockDim.x + 63) >> 6);
}
// nw: the waves that take part (the others of the work-group must have left the kernel: a barrier only counts live waves)
template <int NV>
__device__ __forceinline__ void block_sum_w(double (&v)[NV], double* scratch, int nw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double s = wave_sum(v[k]);  // uniform across the wave
    if (lane == 0) scratch[wave * NV + k] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += scratch[w * NV + k];
    v[k] = s;
  }
  __syncthreads();
}

// Strided sum of an array of partials by the whole workgroup (fixed order -> deterministic).
__device__ __forceinline__ double partial_sum(const double* p, int n) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  return s;
}

}  // namespace
}  // namespace pgo
