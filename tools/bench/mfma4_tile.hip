// tools/bench/mfma4_tile.hip — a 16 x 16 x 4 product U V^T out of four v_mfma_f64_4x4x4_4b_f64 with the A operand rotated by
// DPP row_ror:4: determines which row block each accumulator holds (development check for pgo_front_kernels.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ double ror4(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], 0x124, 0xf, 0xf, false);
  u.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], 0x124, 0xf, 0xf, false);
  return u.d;
}
__global__ void k(const double* U, const double* V, double* out) {
  const int l = threadIdx.x, c = l & 15, kk = l >> 4;
  double a = U[c * 4 + kk], b = V[c * 4 + kk];
  double acc[4];
  for (int t = 0; t < 4; ++t) { acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); a = ror4(a); }
  for (int t = 0; t < 4; ++t) out[64 * t + l] = acc[t];
}
int main() {
  double hU[64], hV[64], ho[256];
  for (int i = 0; i < 64; ++i) { hU[i] = sin(1.0 + i); hV[i] = cos(2.0 + 3 * i); }
  double *dU, *dV, *dO; hipMalloc(&dU, 512); hipMalloc(&dV, 512); hipMalloc(&dO, 2048);
  hipMemcpy(dU, hU, 512, hipMemcpyHostToDevice); hipMemcpy(dV, hV, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dU, dV, dO);
  hipMemcpy(ho, dO, 2048, hipMemcpyDeviceToHost);
  for (int dir = -1; dir <= 1; dir += 2) {
    double err = 0;
    for (int t = 0; t < 4; ++t) for (int l = 0; l < 64; ++l) {
      const int c = l & 15, i = l >> 4, cb = c >> 2;
      const int row = 4 * (((cb + dir * t) % 4 + 4) % 4) + i;
      double s = 0; for (int kk = 0; kk < 4; ++kk) s += hU[row * 4 + kk] * hV[c * 4 + kk];
      err = fmax(err, fabs(s - ho[64 * t + l]));
    }
    printf("row block = (col block %+d * t) mod 4: err %.3e\n", dir, err);
  }
  return 0;
}
