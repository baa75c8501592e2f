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

__global__ __launch_bounds__(256) void k_front_scatter(DeviceGraph g, FrontPlan p) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long na = (long long)p.n_ablk * 36;
  if (t < na) {
    const int a = (int)(t / 36), e = (int)(t - 36LL * a);
    double s = 0.0;
    for (int q = p.ablk_ptr[a]; q < p.ablk_ptr[a + 1]; ++q) s += g.bsr_val[bsr_index(p.ablk_slot[q], e)];
    const FrontDesc& D = p.fronts[p.ablk_front[a]];
    const int pos = p.ablk_pos[a], bi = pos >> 16, bj = pos & 0xffff;
    p.Fval[D.fbase + (size_t)(6 * bi + e / 6) * D.ld + 6 * bj + e % 6] = s;
  } else if (t < na + 6LL * p.n) {
    const int u = (int)(t - na), j = u / 6, k = u - 6 * j;
    const FrontDesc& D = p.fronts[p.col_front[j]];
    const int n = 6 * (D.c + D.r);
    const size_t io = 6 * (size_t)p.perm[j] + k;
    const double b = g.scale[io] * g.grad[io];   // right-hand side S g (as pgo_direct_kernels forward_rhs); the step tail reads cg_b
    g.cg_b[io] = b;
    p.Fval[D.fbase + (size_t)n * D.ld + 6 * (j - D.first) + k] = b;
  }
}

constexpr int ASM_T = 6 * FRONT_ASM_TP;   // 48 scalars per tile side

__global__ __launch_bounds__(256) void k_front_extend_add(FrontPlan p, int front_begin, int front_end) {
  __shared__ double acc[ASM_T][ASM_T + 1];
  // front of this workgroup
  int lo = front_begin, hi = front_end - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.fronts[mid].asm_wg_begin <= wg) lo = mid; else hi = mid - 1;
  }
  const FrontDesc P = p.fronts[lo];
  const int t = wg - P.asm_wg_begin;
  const int ti = t / P.ntp, tj = t - ti * P.ntp;
  const bool rhs_tile = ti == P.ntp;
  if (!rhs_tile && tj > ti) return;
  const int tid = threadIdx.x;
  // children that touch this tile
  bool any = false;
  for (int ci = P.child_begin; ci < P.child_end && !any; ++ci) {
    const FrontDesc& C = p.fronts[p.child[ci]];
    const int* cs = p.cstart + C.cs_begin;
    any = cs[tj + 1] > cs[tj] && (rhs_tile || cs[ti + 1] > cs[ti]);
  }
  if (!any) return;
  for (int e = tid; e < ASM_T * (ASM_T + 1); e += 256) (&acc[0][0])[e] = 0.0;
  __syncthreads();
  const int np = 6 * (P.c + P.r);
  for (int ci = P.child_begin; ci < P.child_end; ++ci) {
    const FrontDesc C = p.fronts[p.child[ci]];
    const int* cs = p.cstart + C.cs_begin;
    const int ms = cs[tj], me = cs[tj + 1];
    if (me <= ms) continue;
    const int ks = rhs_tile ? C.r : cs[ti], ke = rhs_tile ? C.r + 1 : cs[ti + 1];
    if (ke <= ks) continue;
    const int* rel = p.rel + C.rel_begin;
    const int nrow = rhs_tile ? 1 : 6 * (ke - ks), ncol = 6 * (me - ms);
    const double* Fc = p.Fval + C.fbase;
    for (int e = tid; e < nrow * ncol; e += 256) {
      const int lr = e / ncol, lc = e - lr * ncol;
      const int m = ms + lc / 6, b = lc % 6;
      int srow, drow;
      if (rhs_tile) { srow = 6 * (C.c + C.r); drow = 0; }
      else {
        const int k = ks + lr / 6, a = lr % 6;
        if (k < m) continue;
        srow = 6 * (C.c + k) + a;
        drow = 6 * (rel[k] - FRONT_ASM_TP * ti) + a;
      }
      acc[drow][6 * (rel[m] - FRONT_ASM_TP * tj) + b] += Fc[(size_t)srow * C.ld + 6 * (C.c + m) + b];
    }
    __syncthreads();
  }
  double* Fp = p.Fval + P.fbase;
  const int row0 = rhs_tile ? np : ASM_T * ti, col0 = ASM_T * tj;
  const int nrow = rhs_tile ? 1 : min(ASM_T, np - row0), ncol = min(ASM_T, np - col0);
  for (int e = tid; e < nrow * ASM_T; e += 256) {
    const int lr = e / ASM_T, lc = e - lr * ASM_T;
    if (lc < ncol) Fp[(size_t)(row0 + lr) * P.ld + col0 + lc] += acc[lr][lc];
  }
}

// ---- 48 x 48 Cholesky + inverse by ONE wave ---------------------------------------------------------------------------
constexpr int LDW = FRONT_NB + 2;   // LDS row stride (doubles): rows stay 16-byte aligned, 16 lanes x b64/b128 conflict-free

__device__ __forceinline__ double readlane_d(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}

// 1/sqrt(d): v_rsq_f64 seed + coupled Newton (Goldschmidt) steps; the result is used both for the diagonal (d * rs) and
// for scaling the column, so the factor is self-consistent to an ulp or two
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

// 48 x 48 Cholesky by one wave.  DL: on entry the lower triangle of D (row-major, stride LDW); on exit L with 1 / L_kk on the
// diagonal (nothing needs L_kk itself; the inverse needs its reciprocal).  Lane i < 48 owns row i in registers (compile-time
// indices only); column k travels through LDS and is read back as broadcasts.  Returns true when a pivot was not positive.
// Kept out of line (and the LDS pointers typed as such): inlined into the panel kernel the two unrolled phases drove the
// register allocator into thousands of spills.
typedef __attribute__((address_space(3))) double lds_double;

__device__ __noinline__ bool potrf_wave(lds_double* DL, lds_double* colbuf) {
  const int lane = threadIdx.x & 63;
  const int i = lane < FRONT_NB ? lane : FRONT_NB - 1;
  double a[FRONT_NB];
#pragma unroll
  for (int j = 0; j < FRONT_NB; ++j) {
    const double v = DL[i * LDW + j];
    a[j] = j <= i ? v : 0.0;
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < FRONT_NB; ++k) {
    const double d = readlane_d(a[k], k);
    bad |= !(d > 0.0);
    const double rs = rsqrt_nr(d);
    const double l = a[k] * rs;
    a[k] = lane == k ? rs : l;
    lds_double* cb = colbuf + (k & 1) * 64;
    cb[lane] = l;
#pragma unroll
    for (int j = k + 1; j < FRONT_NB; ++j) a[j] = fma(-l, cb[j], a[j]);
  }
  if (lane < FRONT_NB) {
#pragma unroll
    for (int j = 0; j < FRONT_NB; ++j) DL[i * LDW + j] = a[j];
  }
  return bad;
}

// W = L^-1 by one wave, right-looking: lane j owns column j of W; once w[m] is final every later row receives its
// contribution (independent FMAs; column m of L is read as broadcasts).
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

// One 48-column panel step (see FrontJob).  320 lanes = 5 waves: waves 0..3 own rows [16 w, 16 w + 16) of the 64-row tile
// (sums and TRSM on the matrix cores), wave 4 factorises the diagonal block in between.
__global__ __launch_bounds__(320) void k_front_panel(FrontPlan p, int wg_begin, int* flags) {
  __shared__ double DL[FRONT_NB * LDW];
  __shared__ double Wl[FRONT_NB * LDW];
  __shared__ double colbuf[128];
  const int wgi = wg_begin + blockIdx.x;
  const FrontJob J = p.jobs[p.wg_job[wgi]];
  const int tile = p.wg_tile[wgi] >> 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g4 = lane >> 4;
  if (wave == 4) {
    __syncthreads();
    const bool bad = potrf_wave((lds_double*)DL, (lds_double*)colbuf);
    inverse_wave((lds_double*)DL, (lds_double*)Wl);
    if (tile == 0) {
      if (bad && lane == 0) atomicOr(&flags[2], 1);
      double* Wg = p.Winv + J.wbase;
      for (int e = lane; e < FRONT_NB * FRONT_NB; e += 64) Wg[e] = Wl[(e / FRONT_NB) * LDW + e % FRONT_NB];
    }
    __syncthreads();
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
  const int ksum = k0 - J.c0;   // 0, 48, 96 or 144
  {
    const double* Ao = F + (size_t)orow * ld + J.c0 + 2 * g4;
    const double* Bk[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) Bk[q] = F + (size_t)(k0 + min(16 * q + li, nb - 1)) * ld + J.c0 + 2 * g4;
    for (int kc = 0; kc < ksum; kc += 8) {
      const double2 ao = *reinterpret_cast<const double2*>(Ao + kc);
      double2 bk[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) bk[q] = *reinterpret_cast<const double2*>(Bk[q] + kc);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        sp[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(bk[q].x, ao.x, sp[q], 0, 0, 0);
        sp[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(bk[q].y, ao.y, sp[q], 0, 0, 0);
      }
      // D tiles: the operand of tile row qa is bk[qa], of tile column qb is bk[qb]
      const double2 da0 = dqa0 == 0 ? bk[0] : dqa0 == 1 ? bk[1] : bk[2];
      const double2 db0 = dqb0 == 0 ? bk[0] : dqb0 == 1 ? bk[1] : bk[2];
      sd[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(da0.x, db0.x, sd[0], 0, 0, 0);
      sd[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(da0.y, db0.y, sd[0], 0, 0, 0);
      const double2 da1 = dqa1 == 1 ? bk[1] : bk[2];
      sd[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(da1.x, bk[0].x, sd[1], 0, 0, 0);
      sd[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(da1.y, bk[0].y, sd[1], 0, 0, 0);
    }
  }
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
  // P^T = F[own rows, kb]^T - S_P^T, kept in registers in the layout the TRSM consumes as its A operand:
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
  __syncthreads();
  // ---- (b) wave 4: Cholesky + inverse of the diagonal block ----
  __syncthreads();
  if (!wave_on) return;
  // ---- (c) TRSM: out[a][b] = sum_m P[a][m] W[b][m], W lower triangular: tile qb needs m < 16 (qb + 1) ----
  double4_t out[3];
#pragma unroll
  for (int qb = 0; qb < 3; ++qb) {
    out[qb] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q <= qb; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double w = Wl[(16 * qb + li) * LDW + 16 * q + 4 * r + g4];
        out[qb] = __builtin_amdgcn_mfma_f64_16x16x4f64(pt[q][r], w, out[qb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qb = 0; qb < 3; ++qb) {
    const int col = 16 * qb + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase + g4 + 4 * r;
      if (row < J.r1 && col < nb) F[(size_t)row * ld + k0 + col] = out[qb][r];
    }
  }
}

// C tile of 64 x 64 per workgroup; wave w owns rows [16 w, 16 w + 16) x 64 columns (four MFMA tiles).  C -= A B^T, K in
// register-prefetched steps of 16.
__global__ __launch_bounds__(256) void k_front_gemm(FrontPlan p, int wg_begin) {
  const int wgi = wg_begin + blockIdx.x;
  const FrontJob J = p.jobs[p.wg_job[wgi]];
  const int tt = p.wg_tile[wgi], ti = tt >> 16, tj = tt & 0xffff;
  const int row0 = J.r0 + FRONT_TILE * ti, col0 = J.c0 + FRONT_TILE * tj;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g4 = lane >> 4;
  double* F = p.Fval + J.fbase;
  const int ld = J.ld;
  const int wrow0 = row0 + 16 * wave;
  if (wrow0 >= J.r1) return;
  const int wrow_last = min(wrow0 + 15, J.r1 - 1);
  const int arow = min(wrow0 + li, J.r1 - 1);
  const double* Ap = F + (size_t)arow * ld + J.k0 + 2 * g4;
  const double* Bp[4];
  bool tile_on[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = col0 + 16 * q;
    tile_on[q] = c < J.c1 && c <= wrow_last;
    const int brow = min(c + li, J.c1 - 1);
    Bp[q] = F + (size_t)brow * ld + J.k0 + 2 * g4;
  }
  if (!tile_on[0]) return;
  double4_t acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int klen = J.klen;
  double2 a0, a1, b0[4], b1[4];
  auto load = [&](int kc, double2& x0, double2& x1, double2* y0, double2* y1) {
    // addresses clamped into the row (always valid), out-of-range k masked to zero afterwards: no branches around loads
    const int k0c = min(kc, klen - 2 - 2 * g4), k1c = min(kc + 8, klen - 2 - 2 * g4);
    const bool v0 = kc + 2 * g4 < klen, v1 = kc + 8 + 2 * g4 < klen;
    x0 = *reinterpret_cast<const double2*>(Ap + k0c);
    x1 = *reinterpret_cast<const double2*>(Ap + k1c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      y0[q] = *reinterpret_cast<const double2*>(Bp[q] + k0c);
      y1[q] = *reinterpret_cast<const double2*>(Bp[q] + k1c);
    }
    if (!v0) x0 = double2{0.0, 0.0};
    if (!v1) x1 = double2{0.0, 0.0};
  };
  load(0, a0, a1, b0, b1);
  for (int kc = 0; kc < klen; kc += 16) {
    double2 na0 = a0, na1 = a1, nb0[4], nb1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { nb0[q] = b0[q]; nb1[q] = b1[q]; }
    if (kc + 16 < klen) load(kc + 16, na0, na1, nb0, nb1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (tile_on[q]) {
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0[q].x, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0[q].y, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1[q].x, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1[q].y, acc[q], 0, 0, 0);
      }
    }
    a0 = na0; a1 = na1;
#pragma unroll
    for (int q = 0; q < 4; ++q) { b0[q] = nb0[q]; b1[q] = nb1[q]; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!tile_on[q]) continue;
    const int c = col0 + 16 * q + li;
    if (c >= J.c1) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wrow0 + g4 + 4 * r;
      if (row >= J.r1) continue;
      double* dst = F + (size_t)row * ld + c;
      *dst -= acc[q][r];
    }
  }
}

// ---- backward substitution --------------------------------------------------------------------------------------------
// Phase A: t = y_c - L21^T x_r for 64 columns of one front; 512 lanes = 64 columns x 8 row groups.  t is parked in p.x at the
// front's own columns (their x is written by phase B).  Dynamic LDS: xr[r6] | red[512].
constexpr int BWD_T = 512;
__global__ __launch_bounds__(BWD_T) void k_front_bwd_gemv(FrontPlan p, int wg_begin) {
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
  __shared__ double xs[FRONT_NBO];
  __shared__ double tv[FRONT_NBO];
  __shared__ double red[BWD_T];
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
  if (tid < ncol) tv[tid] = xb[c0 + tid];
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
  for (int pn = (c1 - 1) / FRONT_NB; pn >= c0 / FRONT_NB; --pn) {
    const int k0 = pn * FRONT_NB, nb = min((int)FRONT_NB, c6 - k0), kl = k0 - c0;
    const double* W = p.Winv + D.wbase + (size_t)pn * FRONT_NB * FRONT_NB;
    {
      double s = 0.0;
      if (jl < nb) for (int b = jl + rg; b < nb; b += NG) s = fma(W[b * FRONT_NB + jl], tv[kl + b], s);
      red[tid] = s;
      __syncthreads();
      if (rg == 0 && jl < nb) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < NG; ++q) tot += red[64 * q + jl];
        tv[kl + jl] = tot;
        const int col = D.first + (k0 + jl) / 6, kk = (k0 + jl) % 6;
        xb[k0 + jl] = tot;
        g.cg_x[6 * (size_t)p.perm[col] + kk] = tot;
      }
      __syncthreads();
    }
    for (int jc = 0; jc < kl; jc += 64) {
      const int j = jc + jl;
      double s = 0.0;
      if (j < kl) for (int a = rg; a < nb; a += NG) s = fma(F[(size_t)(k0 + a) * ld + c0 + j], tv[kl + a], s);
      red[tid] = s;
      __syncthreads();
      if (rg == 0 && j < kl) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < NG; ++q) tot += red[64 * q + jl];
        tv[j] -= tot;
      }
      __syncthreads();
    }
  }
}

}  // namespace

void launch_front_factor(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s) {
  (void)hipMemsetAsync(p.Fval, 0, (size_t)sym.fval_size * sizeof(double), s);
  const long long nt = (long long)p.n_ablk * 36 + 6LL * p.n;
  hipLaunchKernelGGL(k_front_scatter, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, g, p);
  for (const FrontLevel& L : sym.levels) {
    if (L.asm_wg > 0) hipLaunchKernelGGL(k_front_extend_add, dim3(L.asm_wg), dim3(256), 0, s, p, L.asm_front_begin, L.front_end);
    for (int li = L.launch_begin; li < L.launch_end; ++li) {
      const FrontLaunch& La = sym.launches[li];
      if (La.n_wg <= 0) continue;
      if (La.type == FrontLaunch::PANEL) hipLaunchKernelGGL(k_front_panel, dim3(La.n_wg), dim3(320), 0, s, p, La.wg_begin, g.flags);
      else hipLaunchKernelGGL(k_front_gemm, dim3(La.n_wg), dim3(256), 0, s, p, La.wg_begin);
    }
  }
}

void launch_front_solve(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_bwd_gemv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_set = true;
  }
  for (int l = sym.n_levels - 1; l >= 0; --l) {
    const FrontLevel& L = sym.levels[l];
    size_t lds_a = 0;
    for (int q = L.front_begin; q < L.front_end; ++q) lds_a = std::max(lds_a, (size_t)(6 * sym.fronts[q].r + BWD_T) * sizeof(double));
    if (L.bwd_wg > 0) hipLaunchKernelGGL(k_front_bwd_gemv, dim3(L.bwd_wg), dim3(BWD_T), lds_a, s, p, L.bwd_wg_begin);
    for (int st = 0; st < L.bwd_steps; ++st) {
      const int b = sym.bwd_step_ptr[L.bwd_step_begin + st], e = sym.bwd_step_ptr[L.bwd_step_begin + st + 1];
      if (e > b) hipLaunchKernelGGL(k_front_bwd_block, dim3(e - b), dim3(BWD_T), 0, s, g, p, b);
    }
  }
}

}  // namespace pgo
