// pgo_coarse.hip — kernels of the PCG's coarse level (pgo_coarse.h).  FP64, gfx950.  None of this is bandwidth work: a coarse space of
// a few hundred aggregates is a few megabytes; what matters is a FIXED summation order everywhere (bit-reproducible run to run, no FP64
// atomics) and few dependent round trips per launch.
#include "pgo_coarse.h"
#include "pgo_wave.h"

namespace pgo {

namespace {

// poses of aggregate a: one rank — [a agg, (a + 1) agg); several — the a-th aggregate of the device numbering, in which rank k's rows start
// at k * rows_per (pgo_internal.h pose_int) and its aggregates at k * per_rank
__device__ __forceinline__ void agg_range(const DeviceGraph& g, const CoarsePlan& c, int a, int& v0, int& v1) {
  const int k = a / c.per_rank, l = a - k * c.per_rank;
  const int base = g.world > 1 ? k * g.rows_per : 0;
  // (a rank's segment ends at its last real row: the padding rows behind it have no slots and their vector entries are never written)
  const int lim = g.world > 1 ? c.rank_end[k] : g.N;
  v0 = min(lim, base + l * c.agg);
  v1 = min(lim, v0 + c.agg);
}
__device__ __forceinline__ int agg_of(const DeviceGraph& g, const CoarsePlan& c, int v) {
  if (g.world == 1) return v / c.agg;
  const int k = v / g.rows_per;
  return k * c.per_rank + (v - k * g.rows_per) / c.agg;
}

// P~ of the aggregate's poses: the aggregate's centre (mean position, fixed-order sum), then per pose the 6 x 6 block
//   [ I   2 [e_j x d] ]      d = p - c      (column 3 + j: the rotation e_j about the centre moves the pose by 2 e_j x d)
//   [ 0       I      ]
// with row r divided by the Jacobi scale of that component and the rows of constant blocks zero.
__global__ __launch_bounds__(256) void k_coarse_basis(DeviceGraph g, CoarsePlan c) {
  __shared__ double scratch[16];
  const int a = blockIdx.x, tid = threadIdx.x;
  int v0, v1;
  agg_range(g, c, a, v0, v1);
  if (v1 <= v0) return;
  double ctr[3] = {0.0, 0.0, 0.0};
  for (int v = v0 + tid; v < v1; v += 256) {
    const double* p = g.pose_x + (size_t)POSE_STRIDE * v;
    ctr[0] += p[0]; ctr[1] += p[1]; ctr[2] += p[2];
  }
  block_sum<3>(ctr, scratch);
  const double inv = 1.0 / (double)(v1 - v0);
  for (int v = v0 + tid; v < v1; v += 256) {
    const double* p = g.pose_x + (size_t)POSE_STRIDE * v;
    const double d[3] = {p[0] - ctr[0] * inv, p[1] - ctr[1] * inv, p[2] - ctr[2] * inv};
    double Pv[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) Pv[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) Pv[7 * k] = 1.0;
    Pv[6 * 1 + 3] = -2.0 * d[2]; Pv[6 * 2 + 3] = 2.0 * d[1];
    Pv[6 * 0 + 4] = 2.0 * d[2];  Pv[6 * 2 + 4] = -2.0 * d[0];
    Pv[6 * 0 + 5] = -2.0 * d[1]; Pv[6 * 1 + 5] = 2.0 * d[0];
    const uint8_t cm = g.cmask[v];
    double* o = c.Pt + (size_t)36 * v;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const bool cst = (r < 3) ? (cm & 1) : (cm & 2);
      const double sc = g.scale[6 * (size_t)v + r];
#pragma unroll
      for (int q = 0; q < 6; ++q) o[6 * r + q] = cst ? 0.0 : Pv[6 * r + q] / sc;
    }
  }
}

// Row panel a of the Galerkin matrix: sum over the incidence slots (i, j) of the aggregate's rows of P~_i' B_ij P~_j into
// Ac[6 a .., 6 a(j) ..] — the diagonal slot holds H~_ii + D^2 after the damping, every edge has a slot in each of its rows, so the
// panel is complete without a transpose.  64 slots per pass, one per lane (its 6 x 6 product in registers: one round trip of loads
// per pass); the 36 entry lanes then add the pass's products into the LDS panel in slot order: a fixed order, no atomics.
// NT lanes (64 or 256: as many as the LDS left by the panel allows) = NT slots per pass: the products are latency (slot words -> blocks and the
// two P~ -> 6 x 6 x 6 x 2 multiply-adds), so four waves of them at once are four times the rate; the adding-up stays with 36 lanes of wave 0.
template <int NT>
__global__ __launch_bounds__(NT) void k_coarse_galerkin(DeviceGraph g, CoarsePlan c) {
  extern __shared__ double lds[];      // panel[6][npad] | stage[NT][37]
  double* panel = lds;
  double* stage = lds + (size_t)6 * c.npad;
  __shared__ int scol[NT];
  const int a = c.a_lo + blockIdx.x, lane = threadIdx.x;
  int v0, v1;
  agg_range(g, c, a, v0, v1);
  for (int i = lane; i < 6 * c.npad; i += NT) panel[i] = 0.0;
  const int p_ent = lane / 6, q_ent = lane - 6 * p_ent;      // the entry (of a 6 x 6 product) lanes 0 .. 35 add up
  double own = 0.0;
  const int t_begin = v1 > v0 ? g.row_slot_begin[v0] : 0, t_end = v1 > v0 ? g.row_slot_begin[v1 - 1] + g.row_slot_cnt[v1 - 1] : 0;
  __syncthreads();
  for (int t0 = t_begin; t0 < t_end; t0 += NT) {
    const int t = t0 + lane;
    int col = -1;
    double C[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) C[k] = 0.0;
    if (t < t_end) {
      col = g.slot_col[t];
      const uint8_t side = g.slot_side[t];
      if (col >= 0 && side != SIDE_PAD) {
        const int row = g.slot_row[t];
        double Pj[36];
        const double* pj = c.Pt + (size_t)36 * col;
#pragma unroll
        for (int k = 0; k < 36; ++k) Pj[k] = pj[k];
        const double* pi = c.Pt + (size_t)36 * row;
        double Bf[36];
        bsr_block_full(g, t, side, Bf);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double T[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const double b = Bf[6 * r + k];
#pragma unroll
            for (int q = 0; q < 6; ++q) T[q] += b * Pj[6 * k + q];
          }
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            const double pr = pi[6 * r + p];
#pragma unroll
            for (int q = 0; q < 6; ++q) C[6 * p + q] += pr * T[q];
          }
        }
      } else {
        col = -1;
      }
    }
    // (the slot's lane works out where its product goes — the division by the aggregate size does not belong in the serial loop below)
    scol[lane] = col < 0 ? -1 : 6 * agg_of(g, c, col);
#pragma unroll
    for (int k = 0; k < 36; ++k) stage[lane * 37 + k] = C[k];
    __syncthreads();
    if (lane < 36) {
      // the pass's products in slot order; the ones for the aggregate's OWN diagonal block (most of them: the diagonal slots, the odometry
      // edges) are added up in a register, the others go to their columns of the LDS panel: a third of the dependent LDS round trips
      const int own_t = 6 * a;
      for (int s4 = 0; s4 < NT; s4 += 4) {
        int tj[4];
        double sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { tj[u] = scol[s4 + u]; sv[u] = stage[(s4 + u) * 37 + lane]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (tj[u] == own_t) own += sv[u];
          else if (tj[u] >= 0) panel[(size_t)p_ent * c.npad + tj[u] + q_ent] += sv[u];
        }
      }
    }
    __syncthreads();
  }
  if (lane < 36) panel[(size_t)p_ent * c.npad + 6 * a + q_ent] += own;      // (nothing else was added there)
  __syncthreads();
  for (int i = lane; i < 6 * c.npad; i += NT) {
    const int p = i / c.npad, j = i - p * c.npad;
    double v = panel[i];
    if (j == 6 * a + p && !(v > 0.0)) v = 1.0;          // an aggregate of constant blocks only, or an empty one behind a rank's last row: identity row
    c.Ac[(size_t)(6 * a + p) * c.npad + j] = v;
  }
}
// identity on the padding rows / columns (cdim .. npad)
__global__ void k_coarse_pad(CoarsePlan c) {
  const int i = c.cdim + blockIdx.x, j = threadIdx.x + 256 * blockIdx.y;
  if (i >= c.npad || j >= c.npad) return;
  c.Ac[(size_t)i * c.npad + j] = i == j ? 1.0 : 0.0;
  if (j < c.cdim) c.Ac[(size_t)j * c.npad + i] = 0.0;
}

// ---- explicit inverse: block Gauss-Jordan without pivoting (the matrix is symmetric positive definite), 32 x 32 pivot blocks, ONE launch
// per step, from one copy of the matrix into the other (src -> dst: nothing a work-group reads is written by the launch).  Step k, with K
// the k-th block of rows / columns, W = A_KK^-1 and R = W A_K: (the pivot rows times W):
//   rows i not in K:  A_i: -= A_iK R (columns not in K),  A_iK = -A_iK W;      rows K:  A_K: = R (columns not in K),  A_KK = W
// A work-group owns a 32-row strip x 64 GJ_NT columns (a wave GJ_NT 16-column tiles).  Every work-group inverts the pivot block for
// itself — its 1024 entries in registers, four per lane; per elimination step the pivot row and column go through LDS buffers (by step
// parity) behind ONE barrier — and forms R for its columns on the matrix cores from the pivot rows (v_mfma_f64_16x16x4: A operand = W from
// LDS, B operand = the pivot rows' entries in the layout the instruction wants); R's accumulators ARE the B operand of the update
// (D layout row (l >> 4) + 4 r = k index 4 s + (l >> 4) of step s = r: pgo_front_kernels.hip uses the same identity), A operand = the
// strip's negated column panel from LDS, accumulator preloaded with the strip's old entries.  Redundant arithmetic instead of a
// hand-over between launches: until r06's last day a step was two launches of 16 x 16 blocks (pivot + row panel | update: 59 steps x
// (10.3 + 8.1) us = 1.08 ms per LM iteration at BASELINE configs[1] with aggregates of 64) that handed the inverse and two panels over
// through memory.
typedef double cg_double4 __attribute__((ext_vector_type(4)));
constexpr int PB = 32, PBL = PB + 2;
// GJ_NT = 16-column tiles per wave: a work-group owns 32 rows x 64 GJ_NT columns.  FP64 MFMA is ~61 ns per instruction per SIMD: with four
// tiles a wave spends 7.8 us on its 128 products (19.9 us per step at BASELINE configs[1], aggregates of 64), with two 3.9 (17.2 us) — as
// long as the work-groups of a step still fit the chip at once, one per compute unit (450 work-groups of one tile per wave: 2.76 ms per LM iteration against 2.66 with 240 of two): the launcher picks the smallest that does.
__device__ __forceinline__ double fast_rcp(double d) {       // v_rcp_f64 + two Newton steps: full precision without the division sequence
  double x = __builtin_amdgcn_rcp(d);
  double e = fma(-d, x, 1.0);
  x = fma(x, e, x);
  e = fma(-d, x, 1.0);
  return fma(x, e, x);
}
template <int GJ_NT>
__global__ __launch_bounds__(256) void k_coarse_gj_step(CoarsePlan c, int k, const double* __restrict__ src, double* __restrict__ dst) {
  __shared__ __attribute__((aligned(16))) double Wl[PB][PBL], Cb[PB][PBL];
  __shared__ double rowb[2][2 * PB], colb[2][PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g4 = lane >> 4;
  const int n = c.npad, K0 = PB * k, I0 = PB * blockIdx.x, J0 = 64 * GJ_NT * blockIdx.y + 16 * GJ_NT * wave;
  const bool pivot_strip = I0 == K0;
  // ---- everything the step reads, requested at once ----
  double av[8][GJ_NT];          // av[s][t] = A[K0 + 4 s + g4][column li of tile t]: the pivot rows, B-operand layout
  cg_double4 cv[2][GJ_NT];      // cv[h][t][r] = A[I0 + 16 h + g4 + 4 r][column li of tile t]: the strip's old entries, accumulator layout
  bool on[GJ_NT], piv[GJ_NT];
#pragma unroll
  for (int t = 0; t < GJ_NT; ++t) {
    const int col0 = J0 + 16 * t;
    on[t] = col0 < n;
    piv[t] = col0 >= K0 && col0 < K0 + PB;
    const bool ld = on[t] && !piv[t];
    const int cc = ld ? col0 + li : 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) av[s][t] = ld ? src[(size_t)(K0 + 4 * s + g4) * n + cc] : 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[h][t][r] = (ld && !pivot_strip) ? src[(size_t)(I0 + 16 * h + g4 + 4 * r) * n + cc] : 0.0;
  }
  const int gi = tid >> 3, gj0 = tid & 7;       // pivot block: this lane holds the entries (gi, gj0 + 8 t), t = 0 .. 3, of M and of W
  double m[4], w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { m[t] = src[(size_t)(K0 + gi) * n + K0 + gj0 + 8 * t]; w[t] = gi == gj0 + 8 * t ? 1.0 : 0.0; }
  for (int e = tid; e < PB * PB; e += 256) {
    const int r = e / PB, q = e - PB * r;
    Cb[r][q] = -src[(size_t)(I0 + r) * n + K0 + q];
  }
  // ---- W = M^-1 ----
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int par = p & 1;
    if (gi == p) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { rowb[par][gj0 + 8 * t] = m[t]; rowb[par][PB + gj0 + 8 * t] = w[t]; }
    }
    if (gj0 == (p & 7)) colb[par][gi] = m[p >> 3];
    __syncthreads();
    const double ip = fast_rcp(rowb[par][p]);
    const double f = colb[par][gi] * ip;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double mp = rowb[par][gj0 + 8 * t], wp = rowb[par][PB + gj0 + 8 * t];
      if (gi == p) { m[t] = mp * ip; w[t] = wp * ip; }
      else { m[t] = fma(-f, mp, m[t]); w[t] = fma(-f, wp, w[t]); }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) Wl[gi][gj0 + 8 * t] = w[t];
  __syncthreads();
  // ---- R = W A_K: for this wave's columns (pivot columns: R := W, the strip's old entries := 0 — the same update then gives -A_iK W) ----
  double wa[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s = 0; s < 8; ++s) wa[h][s] = Wl[16 * h + li][4 * s + g4];
  cg_double4 R[2][GJ_NT];
#pragma unroll
  for (int t = 0; t < GJ_NT; ++t) {
    if (piv[t]) {
      const int q0 = J0 + 16 * t - K0;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) R[h][t][r] = Wl[16 * h + g4 + 4 * r][q0 + li];
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        cg_double4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[h][s], av[s][t], acc, 0, 0, 0);
        R[h][t] = acc;
      }
    }
  }
  // ---- the strip: A_i: - A_iK R (pivot strip: R itself) ----
  if (!pivot_strip) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 8; ++s) wa[h][s] = Cb[16 * h + li][4 * s + g4];
  }
#pragma unroll
  for (int t = 0; t < GJ_NT; ++t) {
    if (!on[t]) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cg_double4 acc = cv[h][t];
      if (pivot_strip) acc = R[h][t];
      else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[h][s], R[s >> 2][t][s & 3], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)(I0 + 16 * h + g4 + 4 * r) * n + J0 + 16 * t + li] = acc[r];
    }
  }
}

// rc = P~' vec over the aggregate's poses (lane = pose, block sum in fixed order).  fold_seq >= 0: one more work-group does what k_pipe_fold
// does behind the CG launch `fold_seq` (pgo_kernels.hip: this rank's per-work-group partial triples -> its three sums in the exchange
// buffer, same order) — a launch of its own per CG iteration otherwise (4.4 of 27 us at BASELINE configs[1])
__global__ __launch_bounds__(256) void k_coarse_restrict(DeviceGraph g, CoarsePlan c, const double* vec, int fold_seq) {
  __shared__ double scratch[32];
  if ((int)blockIdx.x == c.a_hi - c.a_lo) {
    if (g.cg->done) return;
    double t3[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < g.n_wg; i += 256) { t3[0] += g.part_rz[i]; t3[1] += g.part_q[i]; t3[2] += g.part_rr[i]; }
    block_sum<3>(t3, scratch);
    if (threadIdx.x == 0) {
      double* pp = g.pipe_buf[(fold_seq & 1) ^ 1] + (size_t)g.rank * g.pipe_seg + (size_t)g.rows_per * 6;
      pp[0] = t3[0]; pp[1] = t3[1]; pp[2] = t3[2];
    }
    return;
  }
  const int a = c.a_lo + blockIdx.x, tid = threadIdx.x;
  int v0, v1;
  agg_range(g, c, a, v0, v1);
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int v = v0 + tid; v < v1; v += 256) {
    const double* pv = c.Pt + (size_t)36 * v;
    const double* w = vec + 6 * (size_t)v;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double wr = w[r];
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q] += pv[6 * r + q] * wr;
    }
  }
  block_sum<6>(acc, scratch);
  if (tid < 6) c.rc[6 * a + tid] = acc[tid];
}
// xc = (Ainv rc) for the aggregate's six rows, then out += P~ xc for its poses
__global__ __launch_bounds__(256) void k_coarse_correct(DeviceGraph g, CoarsePlan c, double* out, int out_seg, double* out2) {
  __shared__ double scratch[32];
  __shared__ double xc[6];
  const int a = c.a_lo + blockIdx.x, tid = threadIdx.x;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int j = tid; j < c.cdim; j += 256) {
    const double r = c.rc[j];
#pragma unroll
    for (int p = 0; p < 6; ++p) acc[p] += c.Ainv[(size_t)(6 * a + p) * c.npad + j] * r;
  }
  block_sum<6>(acc, scratch);
  if (tid < 6) xc[tid] = acc[tid];
  __syncthreads();
  int v0, v1;
  agg_range(g, c, a, v0, v1);
  for (int v = v0 + tid; v < v1; v += 256) {
    const double* pv = c.Pt + (size_t)36 * v;
    const int k = g.world > 1 ? v / g.rows_per : 0;
    double* o = out + (g.world > 1 ? (size_t)k * out_seg + (size_t)(v - k * g.rows_per) * 6 : 6 * (size_t)v);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += pv[6 * r + q] * xc[q];
      o[r] += s;
      if (out2) out2[6 * (size_t)v + r] += s;
    }
  }
}

}  // namespace

void launch_coarse_galerkin(const DeviceGraph& g, const CoarsePlan& c, hipStream_t s) {
  static bool attr_set = false;
  const size_t lds_max = 160 * 1024 - 2048;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_coarse_galerkin<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_coarse_galerkin<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_coarse_basis, dim3(c.n_agg), dim3(256), 0, s, g, c);
  if (c.a_hi <= c.a_lo) return;
  const size_t lds256 = ((size_t)6 * c.npad + 256 * 37) * sizeof(double) + 256 * sizeof(int), lds64 = ((size_t)6 * c.npad + 64 * 37) * sizeof(double);
  if (lds256 <= lds_max) hipLaunchKernelGGL(k_coarse_galerkin<256>, dim3(c.a_hi - c.a_lo), dim3(256), lds256 - 256 * sizeof(int), s, g, c);
  else hipLaunchKernelGGL(k_coarse_galerkin<64>, dim3(c.a_hi - c.a_lo), dim3(64), lds64, s, g, c);
}
void launch_coarse_invert(const CoarsePlan& c, hipStream_t s) {
  if (c.npad > c.cdim) hipLaunchKernelGGL(k_coarse_pad, dim3(c.npad - c.cdim, (c.npad + 255) / 256), dim3(256), 0, s, c);
  // Ac -> Ac2 -> Ac ...: the inverse ends up in c.Ainv (lm_begin: whichever copy the last step writes)
  const int steps = c.npad / PB;
  for (int k = 0; k < steps; ++k) {
    const double* src = (k & 1) ? c.Ac2 : c.Ac;
    double* dst = (k & 1) ? c.Ac : c.Ac2;
    auto wgs = [&](int nt) { return steps * ((c.npad + 64 * nt - 1) / (64 * nt)); };
    if (wgs(1) <= 256) hipLaunchKernelGGL(k_coarse_gj_step<1>, dim3(steps, (c.npad + 63) / 64), dim3(256), 0, s, c, k, src, dst);
    else if (wgs(2) <= 256) hipLaunchKernelGGL(k_coarse_gj_step<2>, dim3(steps, (c.npad + 127) / 128), dim3(256), 0, s, c, k, src, dst);
    else hipLaunchKernelGGL(k_coarse_gj_step<4>, dim3(steps, (c.npad + 255) / 256), dim3(256), 0, s, c, k, src, dst);
  }
}
int coarse_pivot_block() { return PB; }
void launch_coarse_restrict(const DeviceGraph& g, const CoarsePlan& c, const double* vec, hipStream_t s, int fold_seq) {
  hipLaunchKernelGGL(k_coarse_restrict, dim3(c.a_hi - c.a_lo + (fold_seq >= 0 ? 1 : 0)), dim3(256), 0, s, g, c, vec, fold_seq);
}
void launch_coarse_correct(const DeviceGraph& g, const CoarsePlan& c, double* out, int out_seg, double* out2, hipStream_t s) {
  if (c.a_hi > c.a_lo) hipLaunchKernelGGL(k_coarse_correct, dim3(c.a_hi - c.a_lo), dim3(256), 0, s, g, c, out, out_seg, out2);
}

}  // namespace pgo
