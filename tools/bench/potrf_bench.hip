// tools/bench/potrf_bench.hip — micro-benchmark of the diagonal-block wave of pgo_front_kernels.hip (the function text is
// pasted from there by tools/bench/make_potrf_bench.py); prints cycles of one call and checks L L^T = A, W_bb L_bb = I.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
enum { FRONT_NB = 48 };
typedef double double4_t __attribute__((ext_vector_type(4)));
#ifndef PGO_FRONT_MFMA4
#define PGO_FRONT_MFMA4 0
#endif
#if PGO_FRONT_MFMA4
__device__ __forceinline__ double ror4(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0x124, 0xf, 0xf, true);
  u.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0x124, 0xf, 0xf, true);
  return u.d;
}
struct Rot4 { double v[4]; };
__device__ __forceinline__ Rot4 rot4(double a) {
  Rot4 r;
  r.v[0] = a;
  r.v[1] = ror4(a);
  r.v[2] = ror4(r.v[1]);
  r.v[3] = ror4(r.v[2]);
  return r;
}
__device__ __forceinline__ void mma16(double4_t& acc, const Rot4& a, double b) {
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[0], b, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[1], b, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[2], b, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[3], b, acc[3], 0, 0, 0);
}
__device__ __forceinline__ void mma16(double4_t& acc, double a, double b) { mma16(acc, rot4(a), b); }
__device__ __forceinline__ double4_t unrot(const double4_t& acc) {
  const int cb = (threadIdx.x & 15) >> 2;
  double4_t o;
  o[0] = cb == 0 ? acc[0] : cb == 1 ? acc[1] : cb == 2 ? acc[2] : acc[3];
  o[1] = cb == 0 ? acc[3] : cb == 1 ? acc[0] : cb == 2 ? acc[1] : acc[2];
  o[2] = cb == 0 ? acc[2] : cb == 1 ? acc[3] : cb == 2 ? acc[0] : acc[1];
  o[3] = cb == 0 ? acc[1] : cb == 1 ? acc[2] : cb == 2 ? acc[3] : acc[0];
  return o;
}
#else
typedef double Rot4;
__device__ __forceinline__ Rot4 rot4(double a) { return a; }
__device__ __forceinline__ void mma16(double4_t& acc, double a, double b) {
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ double4_t unrot(const double4_t& acc) { return acc; }
#endif

constexpr int LDW = FRONT_NB + 2;   // LDS row stride (doubles): rows stay 16-byte aligned, 16 lanes x b64/b128 conflict-free

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

// 1/sqrt(d): v_rsq_f64 seed + two coupled Newton (Goldschmidt) steps, returned as h = 1 / (2 sqrt(d)) so the caller folds the
// doubling into an operand that is ready early ((a + a) * h): six dependent operations on the pivot chain.  The result is used
// both for the diagonal and for scaling the column, so the factor is self-consistent to an ulp or two
// (tools/bench/potrf_bench.hip: |L L^T - A| at the 1e-14 level for a 48 x 48 block of norm ~100).
__device__ __forceinline__ double half_rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return h;
}

// ---- the 48 x 48 diagonal block, one wave ------------------------------------------------------------------------------
// Blocked by 16 columns with the matrix resident in LDS (DL, row-major, stride LDW):
//   per block  load its 16 columns, lane i = row i (16 registers);
//              16 scalar steps: pivot by v_readlane, 1/sqrt by v_rsq_f64 + Newton; the next pivot is formed in its own lane
//              from that lane's multiplier (the pivot chain never waits for LDS or a second broadcast), the next column is
//              updated through a v_readlane, the other columns of the block through an LDS broadcast of column k;
//              store the block column; update the trailing 16 x 16 tiles on the matrix cores (K = 16).
//   then       the three 16 x 16 diagonal blocks are inverted, lane = (block, column), right-looking.
// Result: DL = L with 1 / L_kk on the diagonal, Wd[16 b + r][c] = (L_bb^-1)[r][c].  The consumers (TRSM below, backward
// substitution) work with M = [[W00 0 0], [L10 W11 0], [L20 L21 W22]] blockwise; the 48 x 48 inverse is never formed.
// Returns true when a pivot was not positive.  Kept out of line with typed LDS pointers (inlined, the unrolled code drove
// the register allocator into thousands of spills).
typedef __attribute__((address_space(3))) double lds_double;
constexpr int LDWD = 18;

#ifndef FRONT_DIAG_READLANE
#define FRONT_DIAG_READLANE 1
#endif
__device__ __noinline__ bool diag_block_wave(lds_double* DL, lds_double* Wd, lds_double* cbuf) {
  const int lane = threadIdx.x & 63, li = lane & 15, g4 = lane >> 4;
  const int i = lane < FRONT_NB ? lane : FRONT_NB - 1;
  bool bad = false;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const int c0 = 16 * b;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double v = DL[i * LDW + c0 + j];
      a[j] = c0 + j <= i ? v : 0.0;
    }
    double dn = a[0];     // the next pivot, valid in the lane that owns it
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double d = readlane_d(dn, c0 + k);
      bad |= !(d > 0.0);
      const double a2 = a[k] + a[k];
      const double h = half_rsqrt_nr(d);
      const double l = a2 * h;
      a[k] = lane == c0 + k ? h + h : l;
      if (k < 15) {
        dn = fma(-l, l, a[k + 1]);            // lane c0 + k + 1: its own l is the multiplier of its diagonal entry
        const double ln = readlane_d(l, c0 + k + 1);
        a[k + 1] = fma(-l, ln, a[k + 1]);
        if (k < 14) {
#if FRONT_DIAG_READLANE
          // the other columns of the block: L[c0 + j][k] straight out of lane c0 + j (v_readlane, no trip through the LDS)
#pragma unroll
          for (int j = k + 2; j < 16; ++j) a[j] = fma(-l, readlane_d(l, c0 + j), a[j]);
#else
          lds_double* cb = cbuf + (k & 1) * 64;
          cb[lane] = l;
#pragma unroll
          for (int j = k + 2; j < 16; ++j) a[j] = fma(-l, cb[c0 + j], a[j]);
#endif
        }
      }
    }
    if (lane < FRONT_NB) {
#pragma unroll
      for (int j = 0; j < 16; ++j) DL[i * LDW + c0 + j] = a[j];
    }
    if (b < 2) {
#pragma unroll
      for (int qa = b + 1; qa < 3; ++qa) {
        Rot4 fa[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { const double t = DL[(16 * qa + li) * LDW + c0 + 4 * s4 + g4]; fa[s4] = rot4(t); }
#pragma unroll
        for (int qb = b + 1; qb <= qa; ++qb) {
          double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) { const double t = DL[(16 * qb + li) * LDW + c0 + 4 * s4 + g4]; mma16(acc, fa[s4], t); }
          const double4_t res = unrot(acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) DL[(16 * qa + g4 + 4 * r) * LDW + 16 * qb + li] -= res[r];
        }
      }
    }
  }
  // inverses of the diagonal 16 x 16 blocks: lane = (block, column)
  {
    const int base = 16 * (i >> 4), j = i & 15;
    double w[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const double e = m == j ? 1.0 : 0.0;
      const double wv = (m == 0 ? e : w[m] + e) * DL[(base + m) * LDW + base + m];
      w[m] = wv;
#pragma unroll
      for (int r = m + 1; r < 16; ++r) {
        const double l = DL[(base + r) * LDW + base + m];
        w[r] = m == 0 ? -l * wv : fma(-l, wv, w[r]);
      }
    }
    if (lane < FRONT_NB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Wd[(base + r) * LDWD + j] = w[r];
    }
  }
  return bad;
}

__global__ __launch_bounds__(64) void kt(const double* in, double* out, int* flags, long long* cycles) {
  __shared__ double DL[FRONT_NB * LDW];
  __shared__ double Wd[FRONT_NB * LDWD];
  __shared__ double cbuf[128];
  const int lane = threadIdx.x;
  for (int e = lane; e < 48*48; e += 64) DL[(e/48)*LDW + e%48] = in[e];
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  const bool bad = diag_block_wave((lds_double*)DL, (lds_double*)Wd, (lds_double*)cbuf);
  const long long t1 = __builtin_readcyclecounter();
  if (bad) flags[0] = 1;
  if (lane == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
  for (int e = lane; e < 48*48; e += 64) { out[2304+e] = DL[(e/48)*LDW + e%48]; }
  for (int e = lane; e < 48*16; e += 64) { out[e] = Wd[(e/16)*LDWD + e%16]; }
}
int main() {
  std::vector<double> A(2304), B(2304);
  for (int i = 0; i < 48; ++i) for (int j = 0; j < 48; ++j) B[i*48+j] = std::sin(1.0 + i*7 + j*3);
  for (int i = 0; i < 48; ++i) for (int j = 0; j < 48; ++j) { double s = 0; for (int k = 0; k < 48; ++k) s += B[i*48+k]*B[j*48+k]; A[i*48+j] = s + (i==j ? 48.0 : 0.0); }
  double *din, *dout; int* df; long long* dc;
  hipMalloc(&din, 2304*8); hipMalloc(&dout, 2*2304*8); hipMalloc(&df, 4); hipMalloc(&dc, 16);
  hipMemcpy(din, A.data(), 2304*8, hipMemcpyHostToDevice); hipMemset(df, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int it = 0; it < 200; ++it) kt<<<1, 64>>>(din, dout, df, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  printf("%.2f us per launch; diag_block_wave %lld ticks of s_memtime\n", 1e3*ms/200, cyc);
  std::vector<double> O(2*2304); hipMemcpy(O.data(), dout, 2*2304*8, hipMemcpyDeviceToHost);
  auto L = [&](int i, int k) { return k == i ? 1.0 / O[2304+i*48+k] : (k < i ? O[2304+i*48+k] : 0.0); };
  double errl = 0, errw = 0;
  for (int i = 0; i < 48; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += L(i,k)*L(j,k); errl = fmax(errl, fabs(s - A[i*48+j])); }
  for (int b = 0; b < 3; ++b) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += O[(16*b+i)*16+k]*L(16*b+k,16*b+j); errw = fmax(errw, fabs(s - (i==j))); }
  printf("|LL^T - A| %.3e  |W_bb L_bb - I| %.3e\n", errl, errw);
  return 0;
}
