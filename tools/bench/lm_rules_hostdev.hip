// Development check: pgo_lm_rules.h must give the same bits on the host and on the device (the two LM drivers share it).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../posegraph-ceres_amd/csrc lm_rules_hostdev.hip -o lm_rules_hostdev
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "pgo_lm_rules.h"
struct Out { double cube, oneminus, radius, rd; };
__host__ __device__ inline Out one(double cc, double mc, double radius) {
  Out o;
  o.rd = cc / mc;
  const double t = 2.0 * o.rd - 1.0;
  o.cube = pgo::lm_cube(t);
  o.oneminus = 1.0 - o.cube;
  o.radius = radius / fmax(1.0 / 3.0, o.oneminus);
  return o;
}
__global__ void k(const double* cc, const double* mc, const double* r, Out* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = one(cc[i], mc[i], r[i]);
}
int main() {
  const int n = 1 << 20;
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  std::vector<double> cc(n), mc(n), r(n);
  for (int i = 0; i < n; ++i) { mc[i] = 1.0 + 100.0 * u(rng); cc[i] = mc[i] * (u(rng) * 1.2); r[i] = 1e4 * (1.0 + u(rng)); }
  double *dcc, *dmc, *dr; Out* dout;
  hipMalloc(&dcc, n * 8); hipMalloc(&dmc, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dout, n * sizeof(Out));
  hipMemcpy(dcc, cc.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dmc, mc.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dr, r.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dcc, dmc, dr, dout, n);
  std::vector<Out> out(n);
  hipMemcpy(out.data(), dout, n * sizeof(Out), hipMemcpyDeviceToHost);
  int bad[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const Out h = one(cc[i], mc[i], r[i]);
    if (memcmp(&h.rd, &out[i].rd, 8)) ++bad[0];
    if (memcmp(&h.cube, &out[i].cube, 8)) { if (bad[1]++ < 3) printf("cube: t=%a host %a dev %a\n", 2.0 * h.rd - 1.0, h.cube, out[i].cube); }
    if (memcmp(&h.oneminus, &out[i].oneminus, 8)) ++bad[2];
    if (memcmp(&h.radius, &out[i].radius, 8)) { if (bad[3]++ < 3) printf("radius: %a / %a host %a dev %a\n", r[i], fmax(1.0 / 3.0, h.oneminus), h.radius, out[i].radius); }
  }
  printf("mismatches of %d: rd %d cube %d 1-cube %d radius %d\n", n, bad[0], bad[1], bad[2], bad[3]);
  return 0;
}
