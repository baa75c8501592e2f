// pgo_front_kernels.hip — numeric phase of the multifrontal GPU Cholesky (pgo_front.h) for gfx950.
//
//   k_front_scatter      BSR blocks of (H~ + D^2) and the right-hand side S g -> the fronts (one lane per scalar)
//   k_front_extend_add   parent tile (8 x 8 poses) gathers the children's update matrices, children in list order
//   k_front_panel        one 48-column panel of every front of a level: per 64-row tile one workgroup
//                          (a) left-looking sums for its rows and for the 48 x 48 diagonal block     v_mfma_f64_16x16x4_f64
//                          (b) wave 0: Cholesky of the diagonal block + explicit inverse W, registers + LDS broadcasts
//                          (c) TRSM  L_i = P_i W^T                                                     v_mfma_f64_16x16x4_f64
//   k_front_gemm         right-looking update behind an outer panel / Schur update  C -= A B^T        v_mfma_f64_16x16x4_f64
//   k_front_bwd_gemv     backward substitution, t = y_c - L21^T x_r, 64 columns per workgroup
//   k_front_bwd_block    backward substitution, x_c = L11^-T t in 192-column blocks through the stored W blocks
//
// MFMA operand mapping (v_mfma_f64_16x16x4_f64, guide cdna_hip_programming.md §3): lane l supplies A[i = l & 15][k = l >> 4]
// and B[k = l >> 4][j = l & 15]; it receives D[row = (l >> 4) + 4 reg][col = l & 15], reg = 0..3.  Every product here is
// U V^T over row-major 16-row panels U, V, so the A and the B operand of a lane are the same kind of load: 16 bytes of row
// (l & 15).  The four lane groups g = l >> 4 split every run of 8 consecutive k: group g loads k = 8s + 2g, 8s + 2g + 1 as
// one double2; the .x halves feed one MFMA, the .y halves the next (the k order inside a sum is free as long as A and B
// agree).
#include "pgo_front.h"

#include <algorithm>
#include <cstdlib>

namespace pgo {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// A wave's global stores are complete (acknowledged by the XCD's L2) once its vmcnt has drained.  __syncthreads() alone does
// not wait for that on gfx950 (a work-group-scope release omits vmcnt(0) outside tgsplit mode), so a single-wave agent-scope
// release behind a barrier would only cover that wave's OWN stores: every wave drains first, then the barrier, then one
// buffer_wbl2 + flag (r02 advisor finding on the single-launch publishes).
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the 16 x 16 x 4 product, two ways -------------------------------------------------------------------------------------
// Measured on MI355X (tools/bench/fma_rate.hip, mfma_rate.hip): v_mfma_f64_16x16x4_f64 retires 2048 flops per ~61 ns per wave
// (32 TFLOP/s with one wave per SIMD, 46 saturated); v_mfma_f64_4x4x4_4b_f64 512 flops per ~8 ns (64-67 TFLOP/s) — the small
// form is the faster FP64 matrix instruction of this chip in isolation.  It multiplies the four DIAGONAL 4 x 4 blocks of a
// 16 x 16 tile (lane l: A/B operand = row / column (l & 15), k = l >> 4; result row block = column block = (l & 15) >> 2, row
// inside the block = l >> 4 — tools/bench/mfma4_layout.hip); rotating the A operand by four lanes inside every row of 16
// (DPP row_ror:4) shifts the row blocks, so four instructions with A rotated 0..3 times cover the tile: accumulator t of lane
// (i, c) holds C[4 ((c / 4 - t) mod 4) + i][c] (tools/bench/mfma4_tile.hip), and `unrot` restores the register layout of the
// 16x16x4 instruction.  In THESE kernels the small form measured 5-8 % slower end to end (C2 factorisation 5.4 -> 5.8 ms,
// C5 32.6 -> 35.1 ms: they are bound by operand delivery and dependent latency, not by the matrix pipe), so the large form
// is the default; PGO_FRONT_MFMA4=1 at compile time selects the other.
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

__global__ __launch_bounds__(256) void k_front_scatter(DeviceGraph g, FrontPlan p) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long na = (long long)p.n_ablk * 36;
  if (t < na) {
    const int a = (int)(t / 36), e = (int)(t - 36LL * a);
    double s = 0.0;
    for (int q = p.ablk_ptr[a]; q < p.ablk_ptr[a + 1]; ++q) { const int slot = p.ablk_slot[q]; s += bsr_elem(g, slot, g.slot_side[slot], e); }
    const FrontDesc& D = p.fronts[p.ablk_front[a]];
    if (D.pad) return;                 // mixed plan: a small front builds itself in LDS (k_sfront_factor)
    const int pos = p.ablk_pos[a], bi = pos >> 16, bj = pos & 0xffff;
    p.Fval[D.fbase + (size_t)(6 * bi + e / 6) * D.ld + 6 * bj + e % 6] = s;
  } else if (t < na + 6LL * p.n) {
    const int u = (int)(t - na), j = u / 6, k = u - 6 * j;
    const FrontDesc& D = p.fronts[p.col_front[j]];
    if (D.pad) return;
    const int n = 6 * (D.c + D.r);
    const size_t io = 6 * (size_t)p.perm[j] + k;
    const double b = g.scale[io] * g.grad[io];   // right-hand side S g (as pgo_direct_kernels forward_rhs); the step tail reads cg_b
    g.cg_b[io] = b;
    p.Fval[D.fbase + (size_t)n * D.ld + 6 * (j - D.first) + k] = b;
  }
}

constexpr int ASM_T = 6 * FRONT_ASM_TP;   // 48 scalars per tile side

__device__ __forceinline__ void front_extend_add_body(const FrontPlan& p, int w, double (*acc)[ASM_T + 1]) {
  const int* rec = p.asm_tile + 8 * (size_t)w;
  const int ti = rec[1] >> 16, tj = rec[1] & 0xffff, cb = rec[2], ce = rec[3];
  const long long pbase = ((long long)rec[5] << 32) | (unsigned)rec[4];
  const int pld = rec[6], np = rec[7] & 0xfffff, ntp = rec[7] >> 20;
  const bool rhs_tile = ti == ntp;
  const int tid = threadIdx.x;
  for (int e = tid; e < ASM_T * (ASM_T + 1); e += 256) (&acc[0][0])[e] = 0.0;
  __syncthreads();
  for (int ci = cb; ci < ce; ++ci) {
    const int* cr = p.asm_contrib + 8 * (size_t)ci;
    const long long cbase = ((long long)cr[1] << 32) | (unsigned)cr[0];
    const int cld = cr[2], cc = cr[3], cr_rows = cr[7];
    const int ks = cr[4] >> 16, ke = cr[4] & 0xffff, ms = cr[5] >> 16, me = cr[5] & 0xffff;
    const int* rel = p.rel + cr[6];
    const int nrow = rhs_tile ? 1 : 6 * (ke - ks), ncol = 6 * (me - ms);
    const double* Fc = p.Fval + cbase;
    for (int e = tid; e < nrow * ncol; e += 256) {
      const int lr = e / ncol, lc = e - lr * ncol;
      const int m = ms + lc / 6, b = lc % 6;
      int srow, drow;
      if (rhs_tile) { srow = 6 * (cc + cr_rows); drow = 0; }
      else {
        const int k = ks + lr / 6, a = lr % 6;
        if (k < m) continue;
        srow = 6 * (cc + k) + a;
        drow = 6 * (rel[k] - FRONT_ASM_TP * ti) + a;
      }
      acc[drow][6 * (rel[m] - FRONT_ASM_TP * tj) + b] += Fc[(size_t)srow * cld + 6 * (cc + m) + b];
    }
    __syncthreads();
  }
  double* Fp = p.Fval + pbase;
  const int row0 = rhs_tile ? np : ASM_T * ti, col0 = ASM_T * tj;
  const int nrow = rhs_tile ? 1 : min(ASM_T, np - row0), ncol = min(ASM_T, np - col0);
  for (int e = tid; e < nrow * ASM_T; e += 256) {
    const int lr = e / ASM_T, lc = e - lr * ASM_T;
    if (lc < ncol) Fp[(size_t)(row0 + lr) * pld + col0 + lc] += acc[lr][lc];
  }
}
__global__ __launch_bounds__(256) void k_front_extend_add(FrontPlan p, int wg_begin) {
  if (p.halt && *p.halt) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double acc[ASM_T][ASM_T + 1];
  front_extend_add_body(p, wg_begin + blockIdx.x, acc);
}

// ---- the diagonal block ------------------------------------------------------------------------------------------------
// ---- 48 x 48 Cholesky + inverse by ONE wave ---------------------------------------------------------------------------
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

// One 48-column panel step (see FrontJob).  320 lanes = 5 waves: waves 0..3 own rows [16 w, 16 w + 16) of the 64-row tile
// (sums and TRSM on the matrix cores), wave 4 factorises the diagonal block in between.
__device__ __forceinline__ void front_panel_body(const FrontPlan& p, const FrontJob& J, int tile, int* flags, double* DL, double* Wd, double* cbuf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g4 = lane >> 4;
  // M (row-major 48 x 48: the 16 x 16 inverses on the diagonal, L below) goes to memory for the backward substitution: by all
  // five waves of tile 0 BEHIND the second barrier (written by the diagonal-block wave alone in front of it, it was 4 us of every
  // panel step's critical path)
  auto store_m = [&]() {
    double* Wg = p.Winv + J.wbase;
    for (int e = threadIdx.x; e < FRONT_NB * FRONT_NB; e += 320) {
      const int r = e / FRONT_NB, c = e - r * FRONT_NB;
      Wg[e] = (r >> 4) == (c >> 4) ? Wd[r * LDWD + (c & 15)] : (r > c ? DL[r * LDW + c] : 0.0);
    }
  };
  if (wave == 4) {
    __syncthreads();
    const bool bad = diag_block_wave((lds_double*)DL, (lds_double*)Wd, (lds_double*)cbuf);
    if (tile == 0 && bad && lane == 0) atomicOr(&flags[2], 1);
    __syncthreads();
    if (tile == 0) store_m();
    return;
  }
  double* F = p.Fval + J.fbase;
  const int ld = J.ld, k0 = J.k0, nb = J.klen;
  const int rbase = J.r0 + FRONT_TILE * tile + 16 * wave;      // first row of this wave
  const bool wave_on = rbase < J.r1;
  const int orow = min(rbase + li, J.r1 - 1);                   // own row of this lane (clamped)
  // ---- (a) left-looking sums: S_P^T tiles (columns of the panel x own rows) and this wave's share of the D tiles ----
  // D tiles (qa, qb), qa >= qb: wave 0: (0,0) (1,0); wave 1: (1,1) (2,0); wave 2: (2,1); wave 3: (2,2)
  const int dqa0 = wave == 0 ? 0 : wave == 1 ? 1 : 2, dqb0 = wave == 0 ? 0 : wave == 1 ? 1 : wave == 2 ? 1 : 2;
  const int dqa1 = wave == 0 ? 1 : 2, dqb1 = 0;
  const bool two_d = wave < 2;
  double4_t sp[3], sd[2];
#pragma unroll
  for (int q = 0; q < 3; ++q) sp[q] = double4_t{0.0, 0.0, 0.0, 0.0};
  sd[0] = double4_t{0.0, 0.0, 0.0, 0.0};
  sd[1] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int ksum = k0 - J.c0;   // 0, 48, 96 or 144: three 8-wide steps per trip, all loads of a trip issued first
  const double* Bk[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) Bk[q] = F + (size_t)(k0 + min(16 * q + li, nb - 1)) * ld + J.c0 + 2 * g4;
  // first the sums of the diagonal block alone: wave 4 can start on it while the row sums below are still being formed
  for (int kc = 0; kc < ksum; kc += 24) {
    double2 bk[3][3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int q = 0; q < 3; ++q) bk[u][q] = *reinterpret_cast<const double2*>(Bk[q] + kc + 8 * u);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      // D tiles (wave-uniform branches: selecting the operands by index sent the fragment array to scratch memory)
      if (wave == 0) {
        const Rot4 rx0 = rot4(bk[u][0].x), ry0 = rot4(bk[u][0].y), rx1 = rot4(bk[u][1].x), ry1 = rot4(bk[u][1].y);
        mma16(sd[0], rx0, bk[u][0].x); mma16(sd[0], ry0, bk[u][0].y);
        mma16(sd[1], rx1, bk[u][0].x); mma16(sd[1], ry1, bk[u][0].y);
      } else if (wave == 1) {
        const Rot4 rx1 = rot4(bk[u][1].x), ry1 = rot4(bk[u][1].y), rx2 = rot4(bk[u][2].x), ry2 = rot4(bk[u][2].y);
        mma16(sd[0], rx1, bk[u][1].x); mma16(sd[0], ry1, bk[u][1].y);
        mma16(sd[1], rx2, bk[u][0].x); mma16(sd[1], ry2, bk[u][0].y);
      } else if (wave == 2) {
        const Rot4 rx2 = rot4(bk[u][2].x), ry2 = rot4(bk[u][2].y);
        mma16(sd[0], rx2, bk[u][1].x); mma16(sd[0], ry2, bk[u][1].y);
      } else {
        const Rot4 rx2 = rot4(bk[u][2].x), ry2 = rot4(bk[u][2].y);
        mma16(sd[0], rx2, bk[u][2].x); mma16(sd[0], ry2, bk[u][2].y);
      }
    }
  }
  sd[0] = unrot(sd[0]);
  sd[1] = unrot(sd[1]);
  // D = F[kb, kb] - S_D into LDS (identity beyond nb); loads from clamped (always valid) addresses, masked afterwards
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int qa = u == 0 ? dqa0 : dqa1, qb = u == 0 ? dqb0 : dqb1;
    double c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      c[r] = F[(size_t)(k0 + min(16 * qa + g4 + 4 * r, nb - 1)) * ld + k0 + min(16 * qb + li, nb - 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * qa + g4 + 4 * r, col = 16 * qb + li;
      const double v = (row < nb && col < nb) ? c[r] - sd[u][r] : (row == col ? 1.0 : 0.0);
      if (u == 0 || two_d) DL[row * LDW + col] = v;
    }
  }
  __syncthreads();
  // ---- (a') the row sums S_P^T (columns of the panel x own rows), beside wave 4's factorisation of the diagonal block ----
  {
    const double* Ao = F + (size_t)orow * ld + J.c0 + 2 * g4;
    for (int kc = 0; kc < ksum; kc += 24) {
      double2 ao[3], bk[3][3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        ao[u] = *reinterpret_cast<const double2*>(Ao + kc + 8 * u);
#pragma unroll
        for (int q = 0; q < 3; ++q) bk[u][q] = *reinterpret_cast<const double2*>(Bk[q] + kc + 8 * u);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        Rot4 rx[3], ry[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { rx[q] = rot4(bk[u][q].x); ry[q] = rot4(bk[u][q].y); }
#pragma unroll
        for (int q = 0; q < 3; ++q) mma16(sp[q], rx[q], ao[u].x);
#pragma unroll
        for (int q = 0; q < 3; ++q) mma16(sp[q], ry[q], ao[u].y);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) sp[q] = unrot(sp[q]);
  // P^T = F[own rows, kb]^T - S_P^T, kept in registers in the layout every later product consumes as its B operand:
  // lane (li, g4) holds P[own row li][column 16 q + g4 + 4 r]
  double4_t pt[3];
  {
    double c[12];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[4 * q + r] = F[(size_t)orow * ld + k0 + min(16 * q + g4 + 4 * r, nb - 1)];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[q][r] = (16 * q + g4 + 4 * r < nb) ? c[4 * q + r] - sp[q][r] : 0.0;
  }
  // ---- (b) wave 4: the diagonal block ----
  __syncthreads();
  if (tile == 0) store_m();
  if (!wave_on) return;
  // ---- (c) TRSM by 16-column blocks, everything kept transposed (lane (li, g4), register r = entry [own row li][16 q +
  // g4 + 4 r]): Y0^T = W00 P0^T;  Y1^T = W11 (P1^T - L10 Y0^T);  Y2^T = W22 (P2^T - L20 Y0^T - L21 Y1^T) ----
  auto mul16 = [&](const double* Mrow, int stride, const double4_t& x, double4_t acc) {
    // acc += M[16 x 16 block, rows li] * x^T-operand; Mrow points at M[block row 0][block column 0]
    double4_t ra = double4_t{0.0, 0.0, 0.0, 0.0}, rb = double4_t{0.0, 0.0, 0.0, 0.0};
    mma16(ra, Mrow[li * stride + g4], x[0]);
    mma16(rb, Mrow[li * stride + 4 + g4], x[1]);
    mma16(ra, Mrow[li * stride + 8 + g4], x[2]);
    mma16(rb, Mrow[li * stride + 12 + g4], x[3]);
    return acc + unrot(ra + rb);
  };
  const double4_t zero4 = double4_t{0.0, 0.0, 0.0, 0.0};
  double4_t y[3];
  y[0] = mul16(Wd, LDWD, pt[0], zero4);
  double4_t s1 = mul16(DL + 16 * LDW, LDW, y[0], zero4);
  y[1] = mul16(Wd + 16 * LDWD, LDWD, pt[1] - s1, zero4);
  double4_t s2 = mul16(DL + 32 * LDW, LDW, y[0], zero4);
  s2 = mul16(DL + 32 * LDW + 16, LDW, y[1], s2);
  y[2] = mul16(Wd + 32 * LDWD, LDWD, pt[2] - s2, zero4);
  if (rbase + li < J.r1) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * q + g4 + 4 * r;
        if (col < nb) F[(size_t)orow * ld + k0 + col] = y[q][r];
      }
  }
}
__global__ __launch_bounds__(320) void k_front_panel(FrontPlan p, int wg_begin, int* flags) {
  if (p.halt && *p.halt) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double DL[FRONT_NB * LDW];
  __shared__ double Wd[FRONT_NB * LDWD];
  __shared__ double cbuf[128];
  const int wgi = wg_begin + blockIdx.x;
  front_panel_body(p, p.jobs[p.wg_job[wgi]], p.wg_tile[wgi] >> 16, flags, DL, Wd, cbuf);
}

// C tile of 64 x 64 per workgroup; wave w owns rows [16 w, 16 w + 16) x 64 columns (four 16 x 16 tiles).  C -= A B^T.
// K runs in steps of 16 through two register buffers (the loads of the next step are in flight while the matrix cores
// work on the current one); every fragment is loaded from a clamped, always valid address and out-of-range k is masked in
// the A fragment afterwards, so no load sits behind a branch.  Tiles entirely above the diagonal are computed and dropped
// in the epilogue (only the workgroups that straddle the diagonal have any).
// Launches with few tiles (the top of the tree) use 32 x 32 tiles, one 16 x 16 tile per wave: a wave's K loop is bound by
// the matrix pipe (~61 ns per instruction), so a quarter of the tile per wave is a quarter of the time, and the extra
// workgroups are free on a mostly idle chip.
// Measured in isolation (tools/bench/gemm_bench.hip, one 4096 x 4096 x K job): 25 TFLOP/s at K = 192 (the outer-panel
// updates), 32-33 at K >= 768 (Schur updates of wide fronts) = the rate of one wave per SIMD issuing v_mfma_f64_16x16x4
// back to back (61 ns each).  A 128 x 128 tile with 64 x 32 per wave (6 fragment loads per 16 instructions instead of 5 per
// 8, prologue and read-modify-write epilogue amortised over four times the tile) measured the same 25 / 33: operand
// delivery is not what bounds this kernel, the matrix pipe's issue rate is; it was not kept.
template <int NQ>
struct GemmFrag { double2 a[2], b[NQ][2]; };

template <int TILE>
__device__ __forceinline__ void front_gemm_body(const FrontPlan& p, const FrontJob& J, int tt) {
  constexpr int NQ = TILE == 64 ? 4 : 1;
  const int ti = tt >> 16, tj = tt & 0xffff;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g4 = lane >> 4;
  const int wrow0 = J.r0 + TILE * ti + (TILE == 64 ? 16 * wave : 16 * (wave >> 1));
  const int col0 = J.c0 + TILE * tj + (TILE == 64 ? 0 : 16 * (wave & 1));
  double* F = p.Fval + J.fbase;
  const int ld = J.ld;
  if (wrow0 >= J.r1 || col0 >= J.c1) return;
  const int wrow_last = min(wrow0 + 15, J.r1 - 1);
  if (col0 > wrow_last) return;
  const double* Ap = F + (size_t)min(wrow0 + li, J.r1 - 1) * ld + J.k0;
  const double* Bp[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) Bp[q] = F + (size_t)min(col0 + 16 * q + li, J.c1 - 1) * ld + J.k0;
  double4_t acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int klen = J.klen, kmax = klen - 2;
  auto load = [&](int kc, GemmFrag<NQ>& f) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = kc + 8 * h + 2 * g4, kk = min(k, kmax);
      f.a[h] = *reinterpret_cast<const double2*>(Ap + kk);
#pragma unroll
      for (int q = 0; q < NQ; ++q) f.b[q][h] = *reinterpret_cast<const double2*>(Bp[q] + kk);
      if (k > kmax) f.a[h] = double2{0.0, 0.0};
    }
  };
  auto compute = [&](const GemmFrag<NQ>& f) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const Rot4 rx = rot4(f.a[h].x), ry = rot4(f.a[h].y);
#pragma unroll
      for (int q = 0; q < NQ; ++q) mma16(acc[q], rx, f.b[q][h].x);
#pragma unroll
      for (int q = 0; q < NQ; ++q) mma16(acc[q], ry, f.b[q][h].y);
    }
  };
  GemmFrag<NQ> f0, f1;
  load(0, f0);
  for (int kc = 0; kc < klen; kc += 32) {
    if (kc + 16 < klen) load(kc + 16, f1);
    compute(f0);
    if (kc + 16 >= klen) break;
    if (kc + 32 < klen) load(kc + 32, f0);
    compute(f1);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = unrot(acc[q]);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = col0 + 16 * q + li;
    if (c >= J.c1 || col0 + 16 * q > wrow_last) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wrow0 + g4 + 4 * r;
      if (row >= J.r1) continue;
      double* dst = F + (size_t)row * ld + c;
      *dst -= acc[q][r];
    }
  }
}
template <int TILE>
__global__ __launch_bounds__(256) void k_front_gemm(FrontPlan p, int wg_begin) {
  if (p.halt && *p.halt) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  const int wgi = wg_begin + blockIdx.x;
  front_gemm_body<TILE>(p, p.jobs[p.wg_job[wgi]], p.wg_tile[wgi]);
}
// (kept for tools/bench/gemm_lds_bench.hip, which times it beside its successor: the 64 x 64 tiles of the solver go through
// k_front_gemm_lds below since r03)

// ---- LDS-staged Schur / outer-panel update (r03) -----------------------------------------------------------------------------
// Same job as front_gemm_body<64> — C[64 x 64 tile] -= A[64 x K] B[64 x K]^T, lower tiles — with the operands going global -> LDS
// once per work-group (the register kernel above fetches every A row for one wave and every B row for each of the four: 320 row
// segments per k-chunk against 128 here) and 32 x 32 per wave (2 x 2 MFMA tiles, four independent accumulators; wave w: rows
// 32 (w >> 1), columns 32 (w & 1)).  K runs in chunks of GL_KC through two LDS buffers: the next chunk's 8 doubles per lane are
// in flight in registers while the matrix cores work on the current one.  The planes are stored k-major ([k][row], GL_LD doubles
// per k) so that a fragment read — lane (li, g4) takes row li of k = 4 s + g4 — is four runs of sixteen consecutive doubles.
// ~90 VGPRs and 34 KB of LDS: four work-groups per CU, i.e. the matrix pipe always has several waves to issue from
// (tools/bench/mfma_rate.hip: 31.5 TFLOP/s with one wave per SIMD, 41 with two, 44 with four).
constexpr int GL_KC = 16, GL_LD = 64 + 4;
__device__ __forceinline__ void front_gemm_lds_body(const FrontPlan& p, const FrontJob& J, int tt, double (*As)[GL_KC][GL_LD], double (*Bs)[GL_KC][GL_LD]) {
  const int ti = tt >> 16, tj = tt & 0xffff;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g4 = lane >> 4;
  const int row0 = J.r0 + 64 * ti, col0 = J.c0 + 64 * tj;
  double* F = p.Fval + J.fbase;
  const int ld = J.ld, klen = J.klen;
  // global -> registers: lane t fetches 4 consecutive k of row (t >> 2) of the A panel and of the B panel
  const int lrow = tid >> 2, lseg = tid & 3;
  const double* Ag = F + (size_t)min(row0 + lrow, J.r1 - 1) * ld + J.k0 + 4 * lseg;
  const double* Bg = F + (size_t)min(col0 + lrow, J.c1 - 1) * ld + J.k0 + 4 * lseg;
  double2 ra[2], rb[2];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = kc + 4 * lseg + 2 * h;
      const int kk = min(k, klen - 2);                       // (klen is even: 192 or 6 c)
      ra[h] = *reinterpret_cast<const double2*>(Ag + (kk - 4 * lseg));
      rb[h] = *reinterpret_cast<const double2*>(Bg + (kk - 4 * lseg));
      if (k >= klen) { ra[h] = double2{0.0, 0.0}; rb[h] = double2{0.0, 0.0}; }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      As[buf][4 * lseg + 2 * h][lrow] = ra[h].x; As[buf][4 * lseg + 2 * h + 1][lrow] = ra[h].y;
      Bs[buf][4 * lseg + 2 * h][lrow] = rb[h].x; Bs[buf][4 * lseg + 2 * h + 1][lrow] = rb[h].y;
    }
  };
  const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  fetch(0);
  // the C entries this lane updates, requested now: the epilogue then only subtracts and stores (their latency used to sit at the
  // end of every tile, where no other load hides it)
  double cv[2][2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(row0 + wr + 16 * a + g4 + 4 * r, J.r1 - 1), c = min(col0 + wc + 16 * b + li, J.c1 - 1);
        cv[a][b][r] = F[(size_t)row * ld + c];
      }
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int kc = 0; kc < klen; kc += GL_KC) {
    const bool more = kc + GL_KC < klen;
    if (more) fetch(kc + GL_KC);
#pragma unroll
    for (int s4 = 0; s4 < GL_KC / 4; ++s4) {
      const double a0 = As[buf][4 * s4 + g4][wr + li], a1 = As[buf][4 * s4 + g4][wr + 16 + li];
      const double b0 = Bs[buf][4 * s4 + g4][wc + li], b1 = Bs[buf][4 * s4 + g4][wc + 16 + li];
      mma16(acc[0][0], a0, b0);
      mma16(acc[0][1], a0, b1);
      mma16(acc[1][0], a1, b0);
      mma16(acc[1][1], a1, b1);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const double4_t v = unrot(acc[a][b]);
      const int R0 = row0 + wr + 16 * a, C0 = col0 + wc + 16 * b;
      if (C0 > min(R0 + 15, J.r1 - 1)) continue;            // a 16 x 16 block entirely above the diagonal
      const int c = C0 + li;
      if (c >= J.c1) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = R0 + g4 + 4 * r;
        if (row >= J.r1) continue;
        F[(size_t)row * ld + c] = cv[a][b][r] - v[r];
      }
    }
}
__global__ __launch_bounds__(256) void k_front_gemm_lds(FrontPlan p, int wg_begin) {
  if (p.halt && *p.halt) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double As[2][GL_KC][GL_LD], Bs[2][GL_KC][GL_LD];
  const int wgi = wg_begin + blockIdx.x;
  front_gemm_lds_body(p, p.jobs[p.wg_job[wgi]], p.wg_tile[wgi], As, Bs);
}

#ifndef FRONT_POLL_SLEEP
#define FRONT_POLL_SLEEP 8
#endif
// ---- the single-launch form (FrontStages, pgo_front.h): every work-group of the launch schedule in one grid ------------------
__global__ __launch_bounds__(320) void k_front_stages(DeviceGraph g, FrontPlan p, FrontStages fs) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  constexpr int SMEM_PANEL = FRONT_NB * LDW + FRONT_NB * LDWD + 128, SMEM_GEMM = 4 * GL_KC * GL_LD;   // panel step / LDS-staged update
  __shared__ double smem[SMEM_PANEL > SMEM_GEMM ? SMEM_PANEL : SMEM_GEMM];
  __shared__ int tk;
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int n_stages_slot = fs.n_stages;
    tk = (int)(atomicAdd(p.st_count + n_stages_slot, 1ull) - fs.ticket_base);
  }
  __syncthreads();
  if (fs.stamps && tid == 0) fs.stamps[3 * (size_t)tk] = (long long)__builtin_amdgcn_s_memrealtime();
  const int kind = p.st_table[2 * (size_t)tk] & 3, w = p.st_table[2 * (size_t)tk] >> 2, stage = p.st_table[2 * (size_t)tk + 1];
  // the job descriptor does not depend on the stages waited for: fetched ahead of the wait (two dependent loads off the chain)
  FrontJob J{};
  int tt = 0;
  if (kind != 0) { J = p.jobs[p.wg_job[w]]; tt = p.wg_tile[w]; }
  if (tid == 0) {
    int spins = 0;
    for (int q = p.st_pred_ptr[stage]; q < p.st_pred_ptr[stage + 1]; ++q) {
      const int ps = p.st_pred[q];
      const unsigned long long target = (unsigned long long)p.st_need[ps] * fs.epoch;
      while (__hip_atomic_load(p.st_count + ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > fs.max_spins || ((spins & 255) == 0 && (__hip_atomic_load(&g.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))) { atomicOr(&g.flags[2], 2); break; }   // (once one wait has run out nobody waits long: the factorisation is going to be repeated)
        __builtin_amdgcn_s_sleep(FRONT_POLL_SLEEP);       // (polling more rarely, or rarely while far from complete, changes nothing)
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // one wave acquires, the barrier orders the others behind it (k_sfront_factor)
    if (fs.stamps) fs.stamps[3 * (size_t)tk + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  if (kind != 1 && tid >= 256) return;                     // only a panel step has work for the fifth wave
  if (kind == 1) front_panel_body(p, J, tt >> 16, g.flags, smem, smem + FRONT_NB * LDW, smem + FRONT_NB * LDW + FRONT_NB * LDWD);
  else if (kind == 0) front_extend_add_body(p, w, reinterpret_cast<double (*)[ASM_T + 1]>(smem));
  else if (kind == 2) front_gemm_lds_body(p, J, tt, reinterpret_cast<double (*)[GL_KC][GL_LD]>(smem), reinterpret_cast<double (*)[GL_KC][GL_LD]>(smem + 2 * GL_KC * GL_LD));
  else front_gemm_body<32>(p, J, tt);
  drain_stores();          // every wave's own stores acknowledged by the L2 before the barrier: wave 0's write-back below then covers them all
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    atomicAdd(p.st_count + stage, 1ull);
    if (fs.stamps) fs.stamps[3 * (size_t)tk + 2] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}


// ---- backward substitution --------------------------------------------------------------------------------------------
// Phase A: t = y_c - L21^T x_r for 64 columns of one front; 512 lanes = 64 columns x 8 row groups.  t is parked in p.x at the
// front's own columns (their x is written by phase B).  Dynamic LDS: xr[r6] | red[512].
constexpr int BWD_T = 512;
__global__ __launch_bounds__(BWD_T) void k_front_bwd_gemv(FrontPlan p, int wg_begin) {
  if (p.halt && *p.halt) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  extern __shared__ double sh[];
  const int wgi = wg_begin + blockIdx.x;
  const FrontDesc D = p.fronts[p.bwd_front[wgi]];
  const int c6 = 6 * D.c, r6 = 6 * D.r, n = c6 + r6, tid = threadIdx.x;
  double* xr = sh;
  double* red = sh + r6;
  const double* F = p.Fval + D.fbase;
  const int ld = D.ld;
  for (int i = tid; i < r6; i += BWD_T) xr[i] = p.x[6 * (size_t)p.idx[D.idx_begin + i / 6] + i % 6];
  __syncthreads();
  const int jl = tid & 63, rg = tid >> 6, j = 64 * p.bwd_chunk[wgi] + jl;
  constexpr int NG = BWD_T / 64;
  double s0 = 0.0, s1 = 0.0;
  if (j < c6) {
    const double* col = F + (size_t)c6 * ld + j;
    int i = rg;
    for (; i + NG < r6; i += 2 * NG) {
      s0 = fma(col[(size_t)i * ld], xr[i], s0);
      s1 = fma(col[(size_t)(i + NG) * ld], xr[i + NG], s1);
    }
    if (i < r6) s0 = fma(col[(size_t)i * ld], xr[i], s0);
  }
  red[tid] = s0 + s1;
  __syncthreads();
  if (rg == 0 && j < c6) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < NG; ++q) tot += red[64 * q + jl];
    p.x[6 * (size_t)D.first + j] = F[(size_t)n * ld + j] - tot;
  }
}

// Phase B: x_c = L11^-T t in 192-column blocks, last block first (schedule: pgo_front.cpp).  One workgroup per
// (source block, target chunk): t_chunk -= L[source rows, chunk columns]^T x_source; the workgroup of the chunk right
// below the source then solves the chunk's own diagonal block panel by panel (x_k = W_k^T t_k, then
// t[j] -= sum_a L[k0 + a][j] x_k[a] for the chunk's columns j < k0) and publishes x.
__global__ __launch_bounds__(BWD_T) void k_front_bwd_block(DeviceGraph g, FrontPlan p, int wg_begin) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double xs[FRONT_NBO];
  __shared__ double tv[FRONT_NBO];
  __shared__ double red[BWD_T];
  __shared__ double Ms[2 * FRONT_NB * (FRONT_NB + 1)];      // M of the current panel | of the next one
  const int wgi = wg_begin + blockIdx.x;
  const FrontDesc D = p.fronts[p.bwdb_front[wgi]];
  const int code = p.bwdb_chunk[wgi], src = code >> 16, ch = code & 0xffff;
  const int c6 = 6 * D.c, tid = threadIdx.x;
  const int c0 = FRONT_NBO * ch, c1 = min(c0 + FRONT_NBO, c6), ncol = c1 - c0;
  const bool solve_only = src == ch, solver = solve_only || ch == src - 1;
  double* xb = p.x + 6 * (size_t)D.first;
  const double* F = p.Fval + D.fbase;
  const int ld = D.ld;
  const int jl = tid & 63, rg = tid >> 6;
  constexpr int NG = BWD_T / 64;
  if (tid < FRONT_NBO) tv[tid] = tid < ncol ? xb[c0 + tid] : 0.0;
  if (!solve_only) {
    const int s0 = FRONT_NBO * src, nsrc = min(s0 + FRONT_NBO, c6) - s0;
    if (tid < nsrc) xs[tid] = xb[s0 + tid];
    __syncthreads();
    for (int jc = 0; jc < ncol; jc += 64) {
      const int j = jc + jl;
      double sa = 0.0, sb = 0.0;
      if (j < ncol) {
        const double* col = F + (size_t)s0 * ld + c0 + j;
        int a = rg;
        for (; a + NG < nsrc; a += 2 * NG) {
          sa = fma(col[(size_t)a * ld], xs[a], sa);
          sb = fma(col[(size_t)(a + NG) * ld], xs[a + NG], sb);
        }
        if (a < nsrc) sa = fma(col[(size_t)a * ld], xs[a], sa);
      }
      red[tid] = sa + sb;
      __syncthreads();
      if (rg == 0 && j < ncol) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < NG; ++q) tot += red[64 * q + jl];
        tv[j] -= tot;
      }
      __syncthreads();
    }
  } else {
    __syncthreads();
  }
  if (!solver) {
    if (tid < ncol) xb[c0 + tid] = tv[tid];
    return;
  }
  // The panels of the chunk, last first.  M of a panel and the L rows its update needs do not depend on x: M of the NEXT panel is
  // fetched (registers, then the other LDS buffer) and this panel's L entries are loaded while wave 0 runs the three dependent
  // stages of the current one — two memory round trips per panel off the chain (21 -> ~13 us per four-panel block step).
  const int pn_first = (c1 - 1) / FRONT_NB, pn_last = c0 / FRONT_NB;
  constexpr int MPT = (FRONT_NB * FRONT_NB + BWD_T - 1) / BWD_T;    // entries of M per lane
  {
    const double* M = p.Winv + D.wbase + (size_t)pn_first * FRONT_NB * FRONT_NB;
    for (int e = tid; e < FRONT_NB * FRONT_NB; e += BWD_T) Ms[(e / FRONT_NB) * (FRONT_NB + 1) + e % FRONT_NB] = M[e];
  }
  __syncthreads();
  int cur = 0;
  for (int pn = pn_first; pn >= pn_last; --pn, cur ^= 1) {
    const int k0 = pn * FRONT_NB, nb = min((int)FRONT_NB, c6 - k0), kl = k0 - c0;
    const double* Mc = Ms + cur * (FRONT_NB * (FRONT_NB + 1));
    double mnext[MPT];
    if (pn > pn_last) {
      const double* M = p.Winv + D.wbase + (size_t)(pn - 1) * FRONT_NB * FRONT_NB;
#pragma unroll
      for (int u = 0; u < MPT; ++u) { const int e = tid + u * BWD_T; mnext[u] = e < FRONT_NB * FRONT_NB ? M[e] : 0.0; }
    }
    // L[k0 + a][c0 + j] for the update below: lane = (column j, third of the panel's rows), 16 independent loads each
    const int uj = tid % 160, upart = tid / 160;
    double lv[16];
    if (kl > 0 && uj < kl && upart < 3) {
      const double* col = F + (size_t)k0 * ld + c0 + uj;
#pragma unroll
      for (int a2 = 0; a2 < 16; ++a2) {
        const int row = 16 * upart + a2;
        const double v = col[(size_t)min(row, nb - 1) * ld];
        lv[a2] = row < nb ? v : 0.0;
      }
    }
    // x_k = L_kk^-T t_k through M = [[W00 0 0], [L10 W11 0], [L20 L21 W22]] (16 x 16 blocks), last block first:
    // x_b = W_bb^T t_b, then t_a -= L_ba^T x_b for the blocks a < b; the three dependent stages run on wave 0 alone (wave-level
    // ordering only, no workgroup barriers inside).
    if (tid < 64) {
      double* tp = tv + kl;
#pragma unroll
      for (int bb = 2; bb >= 0; --bb) {
        // the diagonal blocks of M hold the full 16 x 16 inverse (zeros above its diagonal): fixed 16-term sums
        double sx = 0.0;
        if ((tid >> 4) == bb) {
#pragma unroll
          for (int bq = 0; bq < 16; ++bq) sx = fma(Mc[(16 * bb + bq) * (FRONT_NB + 1) + tid], tp[16 * bb + bq], sx);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((tid >> 4) == bb) tp[tid] = sx;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (tid < 16 * bb) {
          double su = 0.0;
#pragma unroll
          for (int bq = 0; bq < 16; ++bq) su = fma(Mc[(16 * bb + bq) * (FRONT_NB + 1) + tid], tp[16 * bb + bq], su);
          tp[tid] -= su;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (tid < nb) {
        const int col = D.first + (k0 + tid) / 6, kk = (k0 + tid) % 6;
        xb[k0 + tid] = tp[tid];
        g.cg_x[6 * (size_t)p.perm[col] + kk] = tp[tid];
      }
    }
    if (pn > pn_last) {     // M of the next panel into the other buffer (nobody reads that one now)
      double* Mn = Ms + (cur ^ 1) * (FRONT_NB * (FRONT_NB + 1));
#pragma unroll
      for (int u = 0; u < MPT; ++u) { const int e = tid + u * BWD_T; if (e < FRONT_NB * FRONT_NB) Mn[(e / FRONT_NB) * (FRONT_NB + 1) + e % FRONT_NB] = mnext[u]; }
    }
    __syncthreads();
    // t[j] -= sum_a L[k0 + a][c0 + j] x_k[a] for the chunk's columns before the panel (at most 144)
    if (kl > 0) {
      double s0 = 0.0;
      if (uj < kl && upart < 3) {
#pragma unroll
        for (int a2 = 0; a2 < 16; ++a2) s0 = fma(lv[a2], tv[kl + 16 * upart + a2], s0);
      }
      red[tid] = s0;
      __syncthreads();
      if (tid < kl) tv[tid] -= red[tid] + red[tid + 160] + red[tid + 320];
      __syncthreads();
    }
  }
}


// ---- small fronts: the whole front in LDS, one launch per tree level (pgo_front.h) -------------------------------------------
constexpr int SF_T = 256;
#ifndef SF_EA_KG
#define SF_EA_KG 8
#define SF_EA_NU 4
#endif
#ifndef SF_POLL_SLEEP
#define SF_POLL_SLEEP 1
#endif

__global__ __launch_bounds__(SF_T) void k_sfront_factor(DeviceGraph g, FrontPlan p, SFrontPlan sp, int front_begin, int dbg, SFrontSync sy) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  extern __shared__ double F[];     // (n + 1) x ld, ld = n + 1 (n is a multiple of 6: the stride is odd)
  const int tid = threadIdx.x;
  int fi = blockIdx.x;
  if (sy.done) {                    // single-launch form: fronts in ticket order (SFrontSync)
    __shared__ int ticket;
    if (tid == 0) ticket = (int)(atomicAdd(reinterpret_cast<unsigned*>(sy.done + p.nf), 1u) - sy.ticket_base);
    __syncthreads();
    fi = ticket;
  }
  const int f = sp.list ? sp.list[front_begin + fi] : front_begin + fi;
  const FrontDesc D = p.fronts[f];
  const SFront S = sp.sf[f];
  const int c6 = 6 * D.c, r6 = 6 * D.r, n = c6 + r6, ld = n + 1;
  auto stamp = [&](int k) { if (sy.stamps && tid == 0) sy.stamps[6 * (size_t)f + k] = (long long)__builtin_amdgcn_s_memrealtime(); };
  stamp(0);
  // original entries H~ + D^2 (lower blocks; a diagonal block comes whole) and the right-hand side S g: fetched first, the
  // LDS is cleared while the loads are in flight
  // the children's descriptors too (two dependent loads each), parked in LDS for the extend-add below
  constexpr int KMAX = 24;
  __shared__ int kc_ubase[KMAX], kc_ucnt[KMAX];
  const int nkids = D.child_end - D.child_begin;
  if (tid < nkids && tid < KMAX) {
    const SFront C = sp.sf[p.child[D.child_begin + tid]];
    kc_ubase[tid] = C.ubase; kc_ucnt[tid] = C.ucnt;
  }
  constexpr int OMAX = 6;            // 256 lanes x 6 entries: fronts of up to 42 original blocks in one pass
  double ov[OMAX];
  int op[OMAX];
  const int no = (S.ablk_end - S.ablk_begin) * 36;
#pragma unroll
  for (int u = 0; u < OMAX; ++u) {
    const int e = tid + u * SF_T;
    ov[u] = 0.0;
    op[u] = -1;
    if (e < no) {
      const int a = S.ablk_begin + e / 36, k = e % 36;
      const int src = sp.osrc[a];
      double s = 0.0;
      if (src >= 0) s = bsr_elem(g, src & 0x0fffffff, src >> 28, k);
      else for (int q = p.ablk_ptr[a]; q < p.ablk_ptr[a + 1]; ++q) { const int slot = p.ablk_slot[q]; s += bsr_elem(g, slot, g.slot_side[slot], k); }
      const int pos = p.ablk_pos[a], bi = pos >> 16, bj = pos & 0xffff;
      ov[u] = s;
      op[u] = (6 * bi + k / 6) * ld + 6 * bj + k % 6;
    }
  }
  double rhs = 0.0;
  if (tid < c6) {
    const size_t io = 6 * (size_t)p.perm[D.first + tid / 6] + tid % 6;
    rhs = g.scale[io] * g.grad[io];
    g.cg_b[io] = rhs;
  }
  for (int e = tid; e < (n + 1) * ld; e += SF_T) F[e] = 0.0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < OMAX; ++u) if (op[u] >= 0) F[op[u]] = ov[u];
  for (int e = tid + OMAX * SF_T; e < no; e += SF_T) {       // (fronts with more original blocks than one pass holds)
    const int a = S.ablk_begin + e / 36, k = e % 36;
    double s = 0.0;
    for (int q = p.ablk_ptr[a]; q < p.ablk_ptr[a + 1]; ++q) { const int slot = p.ablk_slot[q]; s += bsr_elem(g, slot, g.slot_side[slot], k); }
    const int pos = p.ablk_pos[a], bi = pos >> 16, bj = pos & 0xffff;
    F[(6 * bi + k / 6) * ld + 6 * bj + k % 6] = s;
  }
  if (tid < c6) F[n * ld + tid] = rhs;
  __syncthreads();
  if (sy.done && nkids > 0) {
    // single-launch form: the children's update matrices are published by their workgroups (flag after the data, agent scope)
    // (one lane polls, one child after the other: every poll is a trip to memory, and hundreds of waiting fronts polling all
    // their children at once slow down the very stores they wait for)
    if (tid == 0) {
      int spins = 0;
      for (int k = 0; k < nkids; ++k) {
        const int* flag = sy.done + p.child[D.child_begin + k];
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sy.epoch) {
          if (++spins > sy.max_spins || ((spins & 255) == 0 && (__hip_atomic_load(&g.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))) { atomicOr(&g.flags[2], 2); break; }   // (once one wait has run out nobody waits long: the factorisation is going to be repeated)
          __builtin_amdgcn_s_sleep(SF_POLL_SLEEP);
        }
      }
      // ONE wave acquires at agent scope (the invalidation acts on this CU's vector cache and this XCD's L2, which the four
      // waves share); the barrier orders the others behind it.  With an acquire fence in every wave and a release fence in
      // every wave before the flag, a tree level's hand-over took ~12 us; this way ~5 (KITTI-00: 231 -> 174 us per factorisation).
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  stamp(1);
  // extend-add, gathered per row of this front: an 8-lane group walks the list of child rows that land in its row (child
  // order: fixed summation order, no atomics; no two groups share a row, so one barrier serves all children)
  if (!(dbg & 2)) {
    // child by child (fixed order, a barrier between two children: their entries may meet), a child's packed entries side by
    // side over the 256 lanes, eight per lane with all loads issued before the first add
    // four children at a time: the loads of all four are in flight together (a child's update matrix was written by another
    // workgroup of the previous launch, usually on another XCD: every first touch is a trip to memory), then they are added
    // one child after the other
    constexpr int KG = SF_EA_KG, NU = SF_EA_NU;       // children per batch, 256-lane passes per child held in registers
    for (int k0 = 0; k0 < nkids; k0 += KG) {
      double v[KG][NU];
      int pos[KG][NU];
      int ub[KG], uc[KG];
#pragma unroll
      for (int kk = 0; kk < KG; ++kk) {
        const int k = k0 + kk;
        ub[kk] = 0; uc[kk] = 0;
        if (k < nkids) {
          if (k < KMAX) { ub[kk] = kc_ubase[k]; uc[kk] = kc_ucnt[k]; }
          else { const SFront C = sp.sf[p.child[D.child_begin + k]]; ub[kk] = C.ubase; uc[kk] = C.ucnt; }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {       // clamped index, always loaded, masked afterwards: no load sits behind a branch
          const int e0 = u * SF_T + tid, e = ub[kk] + min(e0, max(uc[kk] - 1, 0));
          v[kk][u] = sp.Uval[e];
          const int q = sp.upos[e];
          pos[kk][u] = e0 < uc[kk] ? q : -1;
        }
      }
#pragma unroll
      for (int kk = 0; kk < KG; ++kk) {
        if (k0 + kk >= nkids) break;
#pragma unroll
        for (int u = 0; u < NU; ++u) if (pos[kk][u] >= 0) F[pos[kk][u]] += v[kk][u];
        for (int e = NU * SF_T + tid; e < uc[kk]; e += SF_T) F[sp.upos[ub[kk] + e]] += sp.Uval[ub[kk] + e];   // (children beyond NU x 256 packed entries)
        __syncthreads();
      }
    }
  }
  stamp(2);
  // right-looking Cholesky of the c6 own columns, one pose (6 columns) per step: the 6 x 6 pivot block is factorised by every
  // lane for itself (registers), the rows below are scaled one per lane, then the trailing lower triangle is updated
  bool bad = false;
  for (int kb = 0; kb < ((dbg & 1) ? 0 : c6); kb += 6) {
    double L[21], inv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = F[(kb + i) * ld + kb + j];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double d = L[c * (c + 1) / 2 + c];
#pragma unroll
      for (int k = 0; k < c; ++k) d = fma(-L[c * (c + 1) / 2 + k], L[c * (c + 1) / 2 + k], d);
      if (!(d > 0.0)) { bad = true; d = 1.0; }
      const double h = half_rsqrt_nr(d);
      inv[c] = h + h;
      L[c * (c + 1) / 2 + c] = d * inv[c];
#pragma unroll
      for (int i = c + 1; i < 6; ++i) {
        double s = L[i * (i + 1) / 2 + c];
#pragma unroll
        for (int k = 0; k < c; ++k) s = fma(-L[i * (i + 1) / 2 + k], L[c * (c + 1) / 2 + k], s);
        L[i * (i + 1) / 2 + c] = s * inv[c];
      }
    }
    const int below = n + 1 - (kb + 6);            // rows kb + 6 .. n (row n: the right-hand side -> y)
    for (int t = tid; t < below; t += SF_T) {
      double* row = F + (kb + 6 + t) * ld + kb;
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s = row[c];
#pragma unroll
        for (int k = 0; k < c; ++k) s = fma(-x[k], L[c * (c + 1) / 2 + k], s);
        x[c] = s * inv[c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) row[c] = x[c];
    }
    __syncthreads();
    // the factor of the pivot block itself (read back by the write-out below; nothing in the trailing update reads it).  AFTER the
    // barrier: every lane reads the unfactorised block at the top of the step, and a wave that is scheduled late must still find it
    if (tid == SF_T - 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) F[(kb + i) * ld + kb + j] = L[i * (i + 1) / 2 + j];
    }
    const int mt = n - (kb + 6);                   // trailing columns; rows: mt + 1 (with the right-hand side)
    for (int ii = tid >> 3; ii <= mt; ii += SF_T >> 3) {       // 8 lanes share a row, strided over its columns
      const double* a = F + (kb + 6 + ii) * ld + kb;
      const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
      const int jmax = ii < mt ? ii : mt - 1;
      double* frow = F + (kb + 6 + ii) * ld + kb + 6;
      int jj = tid & 7;
      for (; jj + 24 <= jmax; jj += 32) {                      // four entries per trip: 28 LDS reads in flight, then the arithmetic
        double b[4][6], fv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* bp = F + (kb + 6 + jj + 8 * u) * ld + kb;
#pragma unroll
          for (int c = 0; c < 6; ++c) b[u][c] = bp[c];
          fv[u] = frow[jj + 8 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          frow[jj + 8 * u] = fv[u] - (a0 * b[u][0] + a1 * b[u][1] + a2 * b[u][2] + a3 * b[u][3] + a4 * b[u][4] + a5 * b[u][5]);
      }
      for (; jj <= jmax; jj += 8) {
        const double* b = F + (kb + 6 + jj) * ld + kb;
        frow[jj] -= a0 * b[0] + a1 * b[1] + a2 * b[2] + a3 * b[3] + a4 * b[4] + a5 * b[5];
      }
    }
    __syncthreads();
  }
  if (bad && tid == 0) atomicOr(&g.flags[2], 1);
  stamp(3);
  // the update matrix first (the parent's launch is next in line), then the L panel (rows 0 .. n, the last one is y)
  if (dbg & 4) return;
  if (S.to_fval) {
    // mixed plan: the parent is a regular front, its extend-add reads this update matrix from Fval in the regular layout
    // (whole 6 x 6 blocks: the upper halves of the diagonal blocks are mirrored)
    double* Fv = p.Fval + D.fbase;
    for (int e = tid; e < (r6 + 1) * r6; e += SF_T) {
      const int i = e / r6, j = e - i * r6;
      Fv[(size_t)(c6 + i) * D.ld + c6 + j] = (j <= i) ? F[(c6 + i) * ld + c6 + j] : F[(c6 + j) * ld + c6 + i];
    }
  } else if (S.ucnt > 0) {
    double* Ug = sp.Uval + S.ubase;
    for (int ii = tid >> 3; ii <= r6; ii += SF_T >> 3) {
      const int e0 = ii < r6 ? ii * (ii + 1) / 2 : r6 * (r6 + 1) / 2;
      const int jmax = ii < r6 ? ii : r6 - 1;
      const double* row = F + (c6 + ii) * ld + c6;
      for (int jj = tid & 7; jj <= jmax; jj += 8) Ug[e0 + jj] = row[jj];
    }
  }
  if (sy.done) {                    // publish: every wave waits for the L2's acknowledgement of its own stores, the barrier collects them, ONE wave writes the L2 back, then the flag
    drain_stores();
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(sy.done + f, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  stamp(4);
  double* Lg = sp.Lval + S.lbase;
  for (int i = tid / 64; i <= n; i += SF_T / 64)
    for (int j = tid & 63; j < c6; j += 64) Lg[i * c6 + j] = F[i * ld + j];
  if (!sy.done) return;
  // single-launch form: W = L11^-1 here as well (k_sfront_invert's block recurrence on the L11 that is still in LDS), in the
  // shadow of the ancestors' factorisations: only the root's inverse is left on the critical path, and one launch less
  {
    const int lw = c6 + 1;
    double* Ws = F + (n + 1) * ld;
    double* Ts = Ws + c6 * lw;
    for (int e = tid; e < c6 * lw; e += SF_T) Ws[e] = 0.0;
    __syncthreads();
    for (int ib = 0; ib < c6; ib += 6) {
      for (int e = tid; e < 6 * ib; e += SF_T) {
        const int r = e / ib, c = e - r * ib;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const double* lrow = F + (ib + r) * ld;
        int k = c - c % 6;
        for (; k + 3 < ib; k += 4) {
          const double l0 = lrow[k], l1 = lrow[k + 1], l2 = lrow[k + 2], l3 = lrow[k + 3];
          const double w0 = Ws[k * lw + c], w1 = Ws[(k + 1) * lw + c], w2 = Ws[(k + 2) * lw + c], w3 = Ws[(k + 3) * lw + c];
          s0 = fma(l0, w0, s0); s1 = fma(l1, w1, s1); s2 = fma(l2, w2, s2); s3 = fma(l3, w3, s3);
        }
        for (; k < ib; ++k) s0 = fma(lrow[k], Ws[k * lw + c], s0);
        Ts[r * lw + c] = (s0 + s1) + (s2 + s3);
      }
      if (tid >= SF_T - 6) {
        const int q = tid - (SF_T - 6);
        double w[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double s = r == q ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < r; ++k) s = k >= q ? fma(-F[(ib + r) * ld + ib + k], w[k], s) : s;
          w[r] = r >= q ? s / F[(ib + r) * ld + ib + r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) Ws[(ib + r) * lw + ib + q] = w[r];
      }
      __syncthreads();
      for (int e = tid; e < 6 * ib; e += SF_T) {
        const int r = e / ib, c = e - r * ib;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) s = fma(Ws[(ib + r) * lw + ib + m], Ts[m * lw + c], s);
        Ws[(ib + r) * lw + c] = -s;
      }
      __syncthreads();
    }
    double* Wg = sp.Wval + S.wbase;
    for (int e = tid; e < c6 * c6; e += SF_T) { const int i = e / c6, j = e - i * c6; Wg[e] = Ws[i * lw + j]; }
  }
  stamp(5);
}

// once per topology: where every packed update entry of every front goes in its parent's LDS front
__global__ __launch_bounds__(SF_T) void k_sfront_upos(FrontPlan p, SFrontPlan sp) {
  const int f = sp.list ? sp.list[blockIdx.x] : blockIdx.x, tid = threadIdx.x;
  const FrontDesc D = p.fronts[f];
  const SFront S = sp.sf[f];
  if (D.parent < 0 || S.ucnt == 0) return;
  const FrontDesc Pd = p.fronts[D.parent];
  const int rc6 = 6 * D.r, tri = rc6 * (rc6 + 1) / 2, pn = 6 * (Pd.c + Pd.r), pld = pn + 1;
  const int* pcol = sp.urel + S.urel;
  for (int e = tid; e < S.ucnt; e += SF_T) {
    int i, j;
    if (e >= tri) { i = rc6; j = e - tri; }
    else {
      i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while (i * (i + 1) / 2 > e) --i;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      j = e - i * (i + 1) / 2;
    }
    sp.upos[S.ubase + e] = (i < rc6 ? pcol[i] : pn) * pld + pcol[j];
  }
}

// W = L11^-1 of every front at once (nothing of the factorisation waits for it), by 6 x 6 blocks: block row i of W needs the
// block rows above it, W_ij = -W_ii sum_{k = j}^{i - 1} L_ik W_kj, all its entries side by side (two barriers per block row).
__global__ __launch_bounds__(SF_T) void k_sfront_invert(FrontPlan p, SFrontPlan sp, int front_begin) {
  if (p.halt && *p.halt) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  extern __shared__ double sh[];     // L11: c6 x ld | W: c6 x ld | T: 6 x ld
  const int f = sp.list ? sp.list[front_begin + blockIdx.x] : front_begin + blockIdx.x, tid = threadIdx.x;
  const FrontDesc D = p.fronts[f];
  const SFront S = sp.sf[f];
  const int c6 = 6 * D.c, ld = c6 + 1;
  double* Ls = sh;
  double* Ws = sh + c6 * ld;
  double* Ts = Ws + c6 * ld;
  const double* Lg = sp.Lval + S.lbase;
  for (int e = tid; e < c6 * c6; e += SF_T) { const int i = e / c6, j = e - i * c6; Ls[i * ld + j] = j <= i ? Lg[e] : 0.0; Ws[i * ld + j] = 0.0; }
  __syncthreads();
  for (int ib = 0; ib < c6; ib += 6) {
    // T = sum_k L[ib.., k] W[k, 0 .. ib): entry (r, c), c < ib
    for (int e = tid; e < 6 * ib; e += SF_T) {
      const int r = e / ib, c = e - r * ib;
      // (W[k][c] = 0 for k above c's block; four independent partial sums: the loop is bound by LDS latency, not by arithmetic)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const double* lrow = Ls + (ib + r) * ld;
      int k = c - c % 6;
      for (; k + 3 < ib; k += 4) {
        const double l0 = lrow[k], l1 = lrow[k + 1], l2 = lrow[k + 2], l3 = lrow[k + 3];
        const double w0 = Ws[k * ld + c], w1 = Ws[(k + 1) * ld + c], w2 = Ws[(k + 2) * ld + c], w3 = Ws[(k + 3) * ld + c];
        s0 = fma(l0, w0, s0); s1 = fma(l1, w1, s1); s2 = fma(l2, w2, s2); s3 = fma(l3, w3, s3);
      }
      for (; k < ib; ++k) s0 = fma(lrow[k], Ws[k * ld + c], s0);
      Ts[r * ld + c] = (s0 + s1) + (s2 + s3);
    }
    // the diagonal block of W: inverse of the 6 x 6 lower triangle, one lane per column (the last six lanes: the first ones
    // are busy with T)
    if (tid >= SF_T - 6) {
      const int q = tid - (SF_T - 6);
      double w[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double s = r == q ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) s = k >= q ? fma(-Ls[(ib + r) * ld + ib + k], w[k], s) : s;
        w[r] = r >= q ? s / Ls[(ib + r) * ld + ib + r] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) Ws[(ib + r) * ld + ib + q] = w[r];
    }
    __syncthreads();
    for (int e = tid; e < 6 * ib; e += SF_T) {
      const int r = e / ib, c = e - r * ib;
      double s = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) s = fma(Ws[(ib + r) * ld + ib + m], Ts[m * ld + c], s);
      Ws[(ib + r) * ld + c] = -s;
    }
    __syncthreads();
  }
  double* Wg = sp.Wval + S.wbase;
  for (int e = tid; e < c6 * c6; e += SF_T) { const int i = e / c6, j = e - i * c6; Wg[e] = Ws[i * ld + j]; }
}

// backward substitution of one level (parents first): t = y_c - L21^T x_r, x_c = W^T t
__global__ __launch_bounds__(SF_T) void k_sfront_bwd(DeviceGraph g, FrontPlan p, SFrontPlan sp, int front_begin, SFrontSync sy) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double xr[SFRONT_MAX], tv[SFRONT_MAX], red[SF_T];
  const int tid = threadIdx.x;
  int fi = blockIdx.x;
  if (sy.done) {                    // single-launch form: the root first, a front waits for its parent (tickets as in k_sfront_factor)
    __shared__ int ticket;
    if (tid == 0) ticket = (int)(atomicAdd(reinterpret_cast<unsigned*>(sy.done + p.nf), 1u) - sy.ticket_base);
    __syncthreads();
    fi = p.nf - 1 - ticket;
  }
  const int f = sp.list ? sp.list[front_begin + fi] : front_begin + fi;
  const FrontDesc D = p.fronts[f];
  const SFront S = sp.sf[f];
  const int c6 = 6 * D.c, r6 = 6 * D.r, n = c6 + r6;
  const double* Lg = sp.Lval + S.lbase;
  const double* Wg = sp.Wval + S.wbase;
  const int parts = SF_T / c6, j = tid % c6, part = tid / c6;    // c6 <= 96: at least two lanes per column
  // this lane's share of column j of L21 and of column j of W: fetched before x_r is known (the panel was written by another
  // workgroup, usually on another XCD: a first touch is a trip to memory; twelve values each cover every front up to 96)
  constexpr int PB = 12;
  double lv[PB], wv[PB];
  const bool on = part < parts;
#pragma unroll
  for (int u = 0; u < PB; ++u) {
    const int i = part + u * parts, iw = j + part + u * parts;
    lv[u] = (on && i < r6) ? Lg[(size_t)(c6 + i) * c6 + j] : 0.0;
    wv[u] = (on && iw < c6) ? Wg[(size_t)iw * c6 + j] : 0.0;
  }
  const double yj = tid < c6 ? Lg[(size_t)n * c6 + tid] : 0.0;
  if (sy.done && D.parent >= 0) {   // x of every ancestor is published once the parent's flag is up (it waited for its own parent)
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(sy.done + D.parent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sy.epoch) {
        if (++spins > sy.max_spins || ((spins & 255) == 0 && (__hip_atomic_load(&g.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))) { atomicOr(&g.flags[2], 2); break; }   // (once one wait has run out nobody waits long: the factorisation is going to be repeated)
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // one wave acquires, the barrier orders the others behind it (k_sfront_factor)
    }
    __syncthreads();
  }
  for (int i = tid; i < r6; i += SF_T) xr[i] = p.x[6 * (size_t)p.idx[D.idx_begin + i / 6] + i % 6];
  __syncthreads();
  double s = 0.0;
  if (on) {
#pragma unroll
    for (int u = 0; u < PB; ++u) { const int i = part + u * parts; if (i < r6) s = fma(lv[u], xr[i], s); }
    for (int i = part + PB * parts; i < r6; i += parts) s += Lg[(size_t)(c6 + i) * c6 + j] * xr[i];
  }
  red[tid] = s;
  __syncthreads();
  if (tid < c6) {
    double tot = 0.0;
    for (int q = 0; q < parts; ++q) tot += red[q * c6 + tid];
    tv[tid] = yj - tot;
  }
  __syncthreads();
  s = 0.0;
  if (on) {                                                       // (W^T t)_j = sum_{i >= j} W[i][j] t[i]
#pragma unroll
    for (int u = 0; u < PB; ++u) { const int i = j + part + u * parts; if (i < c6) s = fma(wv[u], tv[i], s); }
    for (int i = j + part + PB * parts; i < c6; i += parts) s += Wg[(size_t)i * c6 + j] * tv[i];
  }
  red[tid] = s;
  __syncthreads();
  if (tid < c6) {
    double x = 0.0;
    for (int q = 0; q < parts; ++q) x += red[q * c6 + tid];
    const int col = D.first + tid / 6;
    p.x[6 * (size_t)col + tid % 6] = x;
    g.cg_x[6 * (size_t)p.perm[col] + tid % 6] = x;
  }
  if (sy.done) {
    drain_stores();
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(sy.done + f, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

static void sfront_attributes() {
  static bool attr_set = false;
  if (attr_set) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sfront_factor), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(((size_t)(SFRONT_MAX + 1) * (SFRONT_MAX + 1) + (SFRONT_MAX + 6) * (SFRONT_MAX + 1)) * sizeof(double)));   // front + W and scratch of its inversion (single-launch form)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sfront_invert), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((2 * (size_t)SFRONT_MAX + 6) * (SFRONT_MAX + 1) * sizeof(double)));
  attr_set = true;
}
#ifdef PGO_ABLATE
static const int sf_dbg = PGO_ABLATE;   // timing ablations of k_sfront_factor, a compile-time constant (results are wrong with any bit set)
#else
static const int sf_dbg = 0;
#endif

// the kernels that see only the plan gate on FrontPlan::halt = the first word of the device-resident LM state (null without it)
static inline FrontPlan with_halt(const FrontPlan& p, const DeviceGraph& g) {
  FrontPlan q = p;
  q.halt = reinterpret_cast<const int*>(g.lm);
  return q;
}

void launch_front_factor(const DeviceGraph& g, const FrontPlan& p_in, const FrontSymbolic& sym, hipStream_t s, const SFrontPlan* sp,
                         const FrontStages* stages) {
  const FrontPlan p = with_halt(p_in, g);
  (void)hipMemsetAsync(p.Fval, 0, (size_t)sym.fval_size * sizeof(double), s);
  const long long nt = (long long)p.n_ablk * 36 + 6LL * p.n;
  hipLaunchKernelGGL(k_front_scatter, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, g, p);
  if (stages && p.st_table && !sym.mixed) {
    hipLaunchKernelGGL(k_front_stages, dim3(stages->n_tickets), dim3(320), 0, s, g, p, *stages);
    return;
  }
  if (sym.mixed && sp) {
    // the fronts whose whole subtree is small: level by level in LDS, before the rounds of the others (their subtree roots
    // leave their update matrices in Fval for the regular extend-add), and their inverses for the backward levels
    sfront_attributes();
    const size_t lds = (size_t)(SFRONT_MAX + 1) * (SFRONT_MAX + 1) * sizeof(double);
    for (size_t l = 0; l + 1 < sym.slevel_ptr.size(); ++l) {
      const int cnt = sym.slevel_ptr[l + 1] - sym.slevel_ptr[l];
      if (cnt > 0) hipLaunchKernelGGL(k_sfront_factor, dim3(cnt), dim3(SF_T), lds, s, g, p, *sp, sym.slevel_ptr[l], sf_dbg, SFrontSync{nullptr, 0u, 0, 0, nullptr});
    }
    hipLaunchKernelGGL(k_sfront_invert, dim3(sym.n_small), dim3(SF_T), (2 * (size_t)SFRONT_MAX + 6) * (SFRONT_MAX + 1) * sizeof(double), s, p, *sp, 0);
  }
  for (const FrontLaunch& La : sym.launches) {
    if (La.n_wg <= 0) continue;
    if (La.type == FrontLaunch::ASM) hipLaunchKernelGGL(k_front_extend_add, dim3(La.n_wg), dim3(256), 0, s, p, La.wg_begin);
    else if (La.type == FrontLaunch::PANEL) hipLaunchKernelGGL(k_front_panel, dim3(La.n_wg), dim3(320), 0, s, p, La.wg_begin, g.flags);
    else if (La.tile == 32) hipLaunchKernelGGL(k_front_gemm<32>, dim3(La.n_wg), dim3(256), 0, s, p, La.wg_begin);
    else hipLaunchKernelGGL(k_front_gemm_lds, dim3(La.n_wg), dim3(256), 0, s, p, La.wg_begin);
  }
}

void launch_front_solve(const DeviceGraph& g, const FrontPlan& p_in, const FrontSymbolic& sym, hipStream_t s, const SFrontPlan* sp) {
  const FrontPlan p = with_halt(p_in, g);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_bwd_gemv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_set = true;
  }
  for (const FrontBwdLaunch& La : sym.bwd_launches) {
    if (La.n_wg <= 0) continue;
    if (La.kind == 0) hipLaunchKernelGGL(k_front_bwd_gemv, dim3(La.n_wg), dim3(BWD_T), (size_t)La.lds_bytes, s, p, La.wg_begin);
    else hipLaunchKernelGGL(k_front_bwd_block, dim3(La.n_wg), dim3(BWD_T), 0, s, g, p, La.wg_begin);
  }
  if (sym.mixed && sp)
    for (int l = (int)sym.slevel_ptr.size() - 2; l >= 0; --l) {
      const int cnt = sym.slevel_ptr[l + 1] - sym.slevel_ptr[l];
      if (cnt > 0) hipLaunchKernelGGL(k_sfront_bwd, dim3(cnt), dim3(SF_T), 0, s, g, p, *sp, sym.slevel_ptr[l], SFrontSync{nullptr, 0u, 0, 0, nullptr});
    }
}

void launch_sfront_prepare(const FrontPlan& p, const SFrontPlan& sp, const FrontSymbolic& sym, hipStream_t s) {
  hipLaunchKernelGGL(k_sfront_upos, dim3(sym.mixed ? sym.n_small : sym.nf), dim3(SF_T), 0, s, p, sp);
}

void launch_sfront_factor(const DeviceGraph& g, const FrontPlan& p_in, const SFrontPlan& sp, const FrontSymbolic& sym, hipStream_t s,
                          const SFrontSync* fused) {
  const FrontPlan p = with_halt(p_in, g);
  sfront_attributes();
  const size_t lds = (size_t)(sym.max_front + 1) * (sym.max_front + 1) * sizeof(double);
  if (fused && fused->done) {
    // (the front, then room for W and the six-row scratch of its inversion)
    size_t lds_inv = 0;
    for (const FrontDesc& D : sym.fronts) {
      const size_t c6 = 6 * (size_t)D.c, n = c6 + 6 * (size_t)D.r;
      lds_inv = std::max(lds_inv, ((n + 1) * (n + 1) + (c6 + 6) * (c6 + 1)) * sizeof(double));
    }
    hipLaunchKernelGGL(k_sfront_factor, dim3(sym.nf), dim3(SF_T), lds_inv, s, g, p, sp, 0, sf_dbg, *fused);
    return;
  } else {
    for (const FrontLevel& L : sym.levels)
      if (L.front_end > L.front_begin)
        hipLaunchKernelGGL(k_sfront_factor, dim3(L.front_end - L.front_begin), dim3(SF_T), lds, s, g, p, sp, L.front_begin, sf_dbg, SFrontSync{nullptr, 0u, 0, 0, nullptr});
  }
  // the inverses W = L11^-1 of all fronts in one launch.  (Forming them level by level on a side stream, in the shadow of the
  // upper levels, was measured: the event hand-overs between the streams cost more than the launch — KITTI-00 0.51 vs 0.43 ms
  // per LM iteration.)
  int c6max = 0;
  for (const FrontDesc& D : sym.fronts) c6max = std::max(c6max, 6 * D.c);
  hipLaunchKernelGGL(k_sfront_invert, dim3(sym.nf), dim3(SF_T), (2 * (size_t)c6max + 6) * (c6max + 1) * sizeof(double), s, p, sp, 0);
}

void launch_sfront_solve(const DeviceGraph& g, const FrontPlan& p_in, const SFrontPlan& sp, const FrontSymbolic& sym, hipStream_t s,
                         const SFrontSync* fused) {
  const FrontPlan p = with_halt(p_in, g);
  if (fused && fused->done) {
    hipLaunchKernelGGL(k_sfront_bwd, dim3(sym.nf), dim3(SF_T), 0, s, g, p, sp, 0, *fused);
    return;
  }
  for (int l = (int)sym.levels.size() - 1; l >= 0; --l) {
    const FrontLevel& L = sym.levels[l];
    if (L.front_end > L.front_begin) hipLaunchKernelGGL(k_sfront_bwd, dim3(L.front_end - L.front_begin), dim3(SF_T), 0, s, g, p, sp, L.front_begin, SFrontSync{nullptr, 0u, 0, 0, nullptr});
  }
}

}  // namespace pgo
