// pgo_front_kernels.hip — numeric phase of the multifrontal GPU Cholesky (pgo_front.h) for gfx950.
//
//   k_front_scatter      BSR blocks of (H~ + D^2) and the right-hand side S g -> the fronts (one lane per scalar)
//   k_front_extend_add   parent tile (8 x 8 poses) gathers the children's update matrices, children in list order
//   k_front_potrf        48 x 48 diagonal block: Cholesky + explicit inverse W = L_kk^-1, one workgroup
//   k_front_gemm<TRSM>   panel rows  X <- X W^T                       v_mfma_f64_16x16x4_f64
//   k_front_gemm<UPDATE> trailing / Schur update  C -= A B^T          v_mfma_f64_16x16x4_f64
//   k_front_bwd          backward substitution of one front, one workgroup
//
// MFMA operand mapping (v_mfma_f64_16x16x4_f64, guide cdna_hip_programming.md §3): lane l supplies A[i = l & 15][k = l >> 4]
// and B[k = l >> 4][j = l & 15]; it receives D[row = (l >> 4) + 4 reg][col = l & 15], reg = 0..3.  Both GEMMs here are
// C = A B^T over row-major panels, so the A and the B operand of a lane are the same kind of load: 16 bytes of row
// (l & 15) of a 16-row panel.  The four lane groups l >> 4 split every run of 8 consecutive k: group g loads k = 8s + 2g,
// 8s + 2g + 1 as one double2; the .x halves feed one MFMA, the .y halves the next (the k order inside a product sum is
// free as long as A and B agree).
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

// first index k in [0, n) with a[k] >= v (a ascending)
__device__ __forceinline__ int lower_bound_dev(const int* a, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

constexpr int ASM_T = 6 * FRONT_ASM_TP;   // 48 scalars per tile side

__global__ __launch_bounds__(256) void k_front_extend_add(FrontPlan p, int front_begin, int front_end) {
  __shared__ double acc[ASM_T][ASM_T + 1];
  __shared__ int touched;
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
  for (int e = tid; e < ASM_T * (ASM_T + 1); e += 256) (&acc[0][0])[e] = 0.0;
  if (tid == 0) touched = 0;
  __syncthreads();
  const int np = 6 * (P.c + P.r);
  for (int ci = P.child_begin; ci < P.child_end; ++ci) {
    const FrontDesc C = p.fronts[p.child[ci]];
    const int* rel = p.rel + C.rel_begin;
    const int ms = lower_bound_dev(rel, C.r, FRONT_ASM_TP * tj), me = lower_bound_dev(rel, C.r, FRONT_ASM_TP * (tj + 1));
    if (me <= ms) continue;
    int ks, ke;
    if (rhs_tile) { ks = C.r; ke = C.r + 1; }
    else { ks = lower_bound_dev(rel, C.r, FRONT_ASM_TP * ti); ke = lower_bound_dev(rel, C.r, FRONT_ASM_TP * (ti + 1)); }
    if (ke <= ks) continue;
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
    if (tid == 0) touched = 1;
    __syncthreads();
  }
  __syncthreads();
  if (!touched) return;
  double* Fp = p.Fval + P.fbase;
  const int row0 = rhs_tile ? np : ASM_T * ti, col0 = ASM_T * tj;
  const int nrow = rhs_tile ? 1 : min(ASM_T, np - row0), ncol = min(ASM_T, np - col0);
  for (int e = tid; e < nrow * ASM_T; e += 256) {
    const int lr = e / ASM_T, lc = e - lr * ASM_T;
    if (lc < ncol) Fp[(size_t)(row0 + lr) * P.ld + col0 + lc] += acc[lr][lc];
  }
}

// job of workgroup `wg` inside a launch: last job with wg_begin <= wg
__device__ __forceinline__ int find_job(const FrontJob* jobs, int job_begin, int job_end, int wg) {
  int lo = job_begin, hi = job_end - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].wg_begin <= wg) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Cholesky of the diagonal block and its explicit inverse.  One workgroup of 256 lanes, the block in LDS.
__global__ __launch_bounds__(256) void k_front_potrf(FrontPlan p, int job_begin, int* flags) {
  __shared__ double D[FRONT_NB][FRONT_NB + 1];
  __shared__ double Wl[FRONT_NB][FRONT_NB + 1];
  __shared__ double rdiag[FRONT_NB];
  const FrontJob J = p.jobs[job_begin + blockIdx.x];
  const int nb = J.klen, tid = threadIdx.x;
  double* A = p.Fval + J.fbase + (size_t)J.k0 * J.ld + J.k0;
  for (int e = tid; e < FRONT_NB * FRONT_NB; e += 256) {
    const int i = e / FRONT_NB, j = e - i * FRONT_NB;
    D[i][j] = (i < nb && j <= i) ? A[(size_t)i * J.ld + j] : 0.0;
    Wl[i][j] = 0.0;
  }
  __syncthreads();
  bool bad = false;
  for (int k = 0; k < nb; ++k) {
    const double d = D[k][k];
    if (!(d > 0.0)) bad = true;
    const double rs = 1.0 / sqrt(d);
    __syncthreads();
    if (tid == 0) { D[k][k] = d * rs; rdiag[k] = rs; }
    if (tid > 0 && tid < nb - k) D[k + tid][k] *= rs;
    __syncthreads();
    const int w = nb - k - 1;
    for (int e = tid; e < w * w; e += 256) {
      const int i = e / w, j = e - i * w;
      if (j <= i) D[k + 1 + i][k + 1 + j] -= D[k + 1 + i][k] * D[k + 1 + j][k];
    }
    __syncthreads();
  }
  if (bad && tid == 0) atomicOr(&flags[2], 1);
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, j = e - i * nb;
    if (j <= i) A[(size_t)i * J.ld + j] = D[i][j];
  }
  // W = L^-1, row by row: wave w owns the columns 16w .. 16w+15, four lanes per column split the sum
  const int wave = tid >> 6, lane = tid & 63;
  if (wave < 3) {
    const int j = 16 * wave + (lane & 15), sub = lane >> 4;
    for (int i = 0; i < nb; ++i) {
      double s = 0.0;
      if (j <= i && j < nb) for (int m = j + sub; m < i; m += 4) s += D[i][m] * Wl[m][j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (sub == 0 && j <= i && j < nb) Wl[i][j] = ((i == j ? 1.0 : 0.0) - s) * rdiag[i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  double* Wg = p.Winv + J.wbase;
  for (int e = tid; e < FRONT_NB * FRONT_NB; e += 256) Wg[e] = Wl[e / FRONT_NB][e % FRONT_NB];
}

// C tile of 64 x 64 per workgroup; wave w owns rows [16w, 16w + 16) x 64 columns (four MFMA tiles).
// TRSM: C = A W^T written over A (each wave reads and writes its own 16 rows only); UPDATE: C -= A B^T.
template <bool TRSM>
__global__ __launch_bounds__(256) void k_front_gemm(FrontPlan p, int job_begin, int job_end) {
  const int wg = blockIdx.x;
  const FrontJob J = p.jobs[find_job(p.jobs, job_begin, job_end, wg)];
  const int t = wg - J.wg_begin;
  const int ti = t / J.ntc, tj = t - ti * J.ntc;
  const int row0 = J.r0 + FRONT_TILE * ti, col0 = J.c0 + FRONT_TILE * tj;
  const int row_end = min(row0 + FRONT_TILE, J.r1);
  if (!TRSM && row_end - 1 < col0) return;   // entirely above the diagonal
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g4 = lane >> 4;
  double* F = p.Fval + J.fbase;
  const int ld = J.ld;
  const int wrow0 = row0 + 16 * wave;
  if (wrow0 >= J.r1) return;
  const int arow = min(wrow0 + li, J.r1 - 1);
  const double* Ap = F + (size_t)arow * ld + J.k0 + 2 * g4;
  const double* Bp[4];
  bool tile_on[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = col0 + 16 * q;
    tile_on[q] = c < J.c1 && (TRSM || c <= min(wrow0 + 15, J.r1 - 1));
    const int brow = min(c + li, J.c1 - 1);
    Bp[q] = TRSM ? p.Winv + J.wbase + (size_t)(brow - J.c0) * FRONT_NB + 2 * g4 : F + (size_t)brow * ld + J.k0 + 2 * g4;
  }
  double4_t acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int klen = J.klen;
  for (int kc = 0; kc < klen; kc += 8) {
    const bool v = kc + 2 * g4 < klen;
    double2 a = v ? *reinterpret_cast<const double2*>(Ap + kc) : double2{0.0, 0.0};
    double2 b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = (v && tile_on[q]) ? *reinterpret_cast<const double2*>(Bp[q] + kc) : double2{0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (tile_on[q]) {
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b[q].x, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b[q].y, acc[q], 0, 0, 0);
      }
    }
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
      if (TRSM) *dst = acc[q][r];
      else *dst -= acc[q][r];
    }
  }
}

// Backward substitution of one front: x_c = L11^-T (y_c - L21^T x_r).  One workgroup of BWD_T lanes; dynamic LDS:
// t[c6] | xr[r6] | red[BWD_T]
constexpr int BWD_T = 512;
__global__ __launch_bounds__(BWD_T) void k_front_bwd(DeviceGraph g, FrontPlan p, int front_begin) {
  extern __shared__ double sh[];
  const FrontDesc D = p.fronts[front_begin + blockIdx.x];
  const int c6 = 6 * D.c, r6 = 6 * D.r, n = c6 + r6, tid = threadIdx.x;
  double* tv = sh;
  double* xr = sh + c6;
  double* red = xr + r6;
  const double* F = p.Fval + D.fbase;
  const int ld = D.ld;
  for (int i = tid; i < r6; i += BWD_T) xr[i] = p.x[6 * (size_t)p.idx[D.idx_begin + i / 6] + i % 6];
  for (int j = tid; j < c6; j += BWD_T) tv[j] = F[(size_t)n * ld + j];
  __syncthreads();
  const int jl = tid & 63, rg = tid >> 6;
  constexpr int NG = BWD_T / 64;
  // t -= L21^T x_r : lanes along the columns, NG row groups
  for (int jc = 0; jc < c6; jc += 64) {
    const int j = jc + jl;
    double s = 0.0;
    if (j < c6) for (int i = rg; i < r6; i += NG) s += F[(size_t)(c6 + i) * ld + j] * xr[i];
    red[tid] = s;
    __syncthreads();
    if (rg == 0 && j < c6) {
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < NG; ++q) tot += red[64 * q + jl];
      tv[j] -= tot;
    }
    __syncthreads();
  }
  const int npanels = (c6 + FRONT_NB - 1) / FRONT_NB;
  for (int pn = npanels - 1; pn >= 0; --pn) {
    const int k0 = pn * FRONT_NB, nb = min((int)FRONT_NB, c6 - k0);
    const double* W = p.Winv + D.wbase + (size_t)pn * FRONT_NB * FRONT_NB;
    // xs = W^T t_k : lane a, row groups over b
    {
      double s = 0.0;
      if (jl < nb) for (int b = jl + rg; b < nb; b += NG) s += W[b * FRONT_NB + jl] * tv[k0 + b];
      red[tid] = s;
      __syncthreads();
      if (rg == 0 && jl < nb) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < NG; ++q) tot += red[64 * q + jl];
        tv[k0 + jl] = tot;
        const int col = D.first + (k0 + jl) / 6, kk = (k0 + jl) % 6;
        p.x[6 * (size_t)col + kk] = tot;
        g.cg_x[6 * (size_t)p.perm[col] + kk] = tot;
      }
      __syncthreads();
    }
    // t[j] -= sum_a L[k0 + a][j] xs[a] for j < k0
    for (int jc = 0; jc < k0; jc += 64) {
      const int j = jc + jl;
      double s = 0.0;
      if (j < k0) for (int a = rg; a < nb; a += NG) s += F[(size_t)(k0 + a) * ld + j] * tv[k0 + a];
      red[tid] = s;
      __syncthreads();
      if (rg == 0 && j < k0) {
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
      if (La.type == FrontLaunch::POTRF) hipLaunchKernelGGL(k_front_potrf, dim3(La.n_wg), dim3(256), 0, s, p, La.job_begin, g.flags);
      else if (La.type == FrontLaunch::TRSM) hipLaunchKernelGGL(k_front_gemm<true>, dim3(La.n_wg), dim3(256), 0, s, p, La.job_begin, La.job_end);
      else hipLaunchKernelGGL(k_front_gemm<false>, dim3(La.n_wg), dim3(256), 0, s, p, La.job_begin, La.job_end);
    }
  }
}

void launch_front_solve(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_front_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_set = true;
  }
  for (int l = sym.n_levels - 1; l >= 0; --l) {
    const FrontLevel& L = sym.levels[l];
    size_t lds = 0;
    for (int q = L.front_begin; q < L.front_end; ++q) lds = std::max(lds, (size_t)(6 * (sym.fronts[q].c + sym.fronts[q].r) + BWD_T) * sizeof(double));
    hipLaunchKernelGGL(k_front_bwd, dim3(L.front_end - L.front_begin), dim3(BWD_T), lds, s, g, p, L.front_begin);
  }
}

}  // namespace pgo
