#include <hip/hip_runtime.h>
enum { FRONT_NB = 48 };
constexpr int LDW = FRONT_NB + 2;
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) double2 lds_double2;

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return h + h;
}
__device__ __noinline__ bool potrf_wave(lds_double* DL, lds_double* colbuf) {
  const int lane = threadIdx.x & 63;
  const int i = lane < FRONT_NB ? lane : FRONT_NB - 1;
  double a[FRONT_NB];
#pragma unroll
  for (int j = 0; j < FRONT_NB; j += 2) {
    const double2 v = double2{DL[i * LDW + j], DL[i * LDW + j + 1]};
    a[j] = j <= i ? v.x : 0.0;
    a[j + 1] = j + 1 <= i ? v.y : 0.0;
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < FRONT_NB; ++k) {
    const double d = readlane_d(a[k], k);
    bad |= !(d > 0.0);
    const double rs = rsqrt_nr(d);
    const double l = a[k] * rs;
    a[k] = lane == k ? rs : l;      // the diagonal of the stored factor holds 1 / L_kk (what the inverse needs)
    lds_double* cb = colbuf + (k & 1) * 64;
    cb[lane] = l;
#pragma unroll
    for (int jb = (k + 1) & ~1; jb < FRONT_NB; jb += 2) {
      const double2 c = double2{cb[jb], cb[jb + 1]};
      if (jb > k) a[jb] = fma(-l, c.x, a[jb]);
      a[jb + 1] = fma(-l, c.y, a[jb + 1]);
    }
  }
  if (lane < FRONT_NB) {
#pragma unroll
    for (int j = 0; j < FRONT_NB; j += 2) { DL[i * LDW + j] = a[j]; DL[i * LDW + j + 1] = a[j + 1]; }
  }
  return bad;
}
__device__ __noinline__ void inverse_wave(const lds_double* DL, lds_double* Wl) {
  const int lane = threadIdx.x & 63;
  double a[FRONT_NB];
#pragma unroll
  for (int m = 0; m < FRONT_NB; ++m) {
    const double e = m == lane ? 1.0 : 0.0;
    const double w = (m == 0 ? e : a[m] + e) * DL[m * LDW + m];
    a[m] = w;
#pragma unroll
    for (int r = m + 1; r < FRONT_NB; ++r) {
      const double l = DL[r * LDW + m];
      a[r] = m == 0 ? -l * w : fma(-l, w, a[r]);
    }
  }
  if (lane < FRONT_NB) {
#pragma unroll
    for (int r = 0; r < FRONT_NB; ++r) Wl[r * LDW + lane] = a[r];
  }
}

template <int MODE>
__global__ __launch_bounds__(64) void kt(const double* in, double* out, int* flags, long long* cycles) {
  __shared__ double DL[FRONT_NB * LDW];
  __shared__ double Wl[FRONT_NB * LDW];
  __shared__ double colbuf[128];
  const int lane = threadIdx.x;
  for (int e = lane; e < 48*48; e += 64) { DL[(e/48)*LDW + e%48] = in[e]; Wl[(e/48)*LDW + e%48] = 0.0; }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  bool bad = false;
  if (MODE != 2) bad = potrf_wave((lds_double*)DL, (lds_double*)colbuf);
  const long long t1 = __builtin_readcyclecounter();
  if (MODE != 1) inverse_wave((lds_double*)DL, (lds_double*)Wl);
  const long long t2 = __builtin_readcyclecounter();
  if (bad) flags[0] = 1;
  if (lane == 0 && blockIdx.x == 0) { cycles[0] = t1 - t0; cycles[1] = t2 - t1; }
  for (int e = lane; e < 48*48; e += 64) { out[e] = Wl[(e/48)*LDW + e%48]; out[2304+e] = DL[(e/48)*LDW + e%48]; }
}

#include <cstdio>
#include <vector>
#include <cmath>
int main() {
  std::vector<double> A(2304), B(2304);
  for (int i = 0; i < 48; ++i) for (int j = 0; j < 48; ++j) B[i*48+j] = std::sin(1.0 + i*7 + j*3) ;
  for (int i = 0; i < 48; ++i) for (int j = 0; j < 48; ++j) { double s = 0; for (int k = 0; k < 48; ++k) s += B[i*48+k]*B[j*48+k]; A[i*48+j] = s + (i==j ? 48.0 : 0.0); }
  double *din, *dout; int* df; long long* dc;
  hipMalloc(&din, 2304*8); hipMalloc(&dout, 2*2304*8); hipMalloc(&df, 4); hipMalloc(&dc, 16);
  hipMemcpy(din, A.data(), 2304*8, hipMemcpyHostToDevice); hipMemset(df, 0, 4);
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int it = 0; it < 200; ++it) {
        if (mode == 0) kt<0><<<1, 64>>>(din, dout, df, dc);
        else if (mode == 1) kt<1><<<1, 64>>>(din, dout, df, dc);
        else kt<2><<<1, 64>>>(din, dout, df, dc);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc[2]; hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %.2f us per launch; cycles potrf %lld inverse %lld (100 MHz counter: x10 ns)\n", mode, 1e3*ms/200, cyc[0], cyc[1]);
  }
  // check W L = I
  kt<0><<<1, 64>>>(din, dout, df, dc);
  std::vector<double> O(2*2304); hipMemcpy(O.data(), dout, 2*2304*8, hipMemcpyDeviceToHost);
  double err = 0, errl = 0;
  // L has 1/diag on the diagonal
  for (int i = 0; i < 48; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) { double lik = k==i ? 1.0/O[2304+i*48+k] : O[2304+i*48+k]; double ljk = k==j ? 1.0/O[2304+j*48+k] : O[2304+j*48+k]; s += lik*ljk; } errl = fmax(errl, fabs(s - A[i*48+j])); }
  for (int i = 0; i < 48; ++i) for (int j = 0; j < 48; ++j) { double s = 0; for (int k = 0; k < 48; ++k) { double lkj = k==j ? 1.0/O[2304+k*48+j] : (k>j ? O[2304+k*48+j] : 0.0); s += O[i*48+k]*lkj; } err = fmax(err, fabs(s - (i==j))); }
  printf("|LL^T - A| %.3e  |W L - I| %.3e\n", errl, err);
  return 0;
}
