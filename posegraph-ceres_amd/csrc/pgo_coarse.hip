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

// ---- explicit inverse, in place: block Gauss-Jordan without pivoting (the matrix is symmetric positive definite), 16 x 16 blocks.
// Step k, with K = the k-th block of rows / columns:   W = A_KK^-1;   R = W A_K:   (pivot kernel, one work-group)
//   rows i not in K:  A_i: -= A_iK R (columns not in K),  A_iK = -A_iK W;      rows K:  A_K: = R (columns not in K),  A_KK = W   (update kernel)
constexpr int GJ = 16;
// (every work-group of the launch inverts the pivot block for itself — sixteen elimination steps in LDS — and forms ITS 256 columns of
// R; work-group 0 also leaves W for the update kernel)
__global__ __launch_bounds__(256) void k_coarse_gj_pivot(CoarsePlan c, int k) {
  __shared__ double M[GJ][GJ + 1], W[GJ][GJ + 1];
  const int tid = threadIdx.x, i = tid / GJ, j = tid % GJ, n = c.npad, K0 = GJ * k;
  const int col = 256 * blockIdx.x + tid;
  double av[GJ];
#pragma unroll
  for (int m = 0; m < GJ; ++m) av[m] = c.Ac[(size_t)(K0 + m) * n + min(col, n - 1)];
  M[i][j] = c.Ac[(size_t)(K0 + i) * n + K0 + j];
  W[i][j] = i == j ? 1.0 : 0.0;
  __syncthreads();
  for (int p = 0; p < GJ; ++p) {
    const double ip = 1.0 / M[p][p];
    const double f = M[i][p] * ip;
    const double mp = M[p][j], wp = W[p][j];
    __syncthreads();
    if (i == p) { M[i][j] = mp * ip; W[i][j] = wp * ip; }
    else { M[i][j] -= f * mp; W[i][j] -= f * wp; }
    __syncthreads();
  }
  if (blockIdx.x == 0) c.piv[i * GJ + j] = W[i][j];
  if (col < n) {
#pragma unroll
    for (int ii = 0; ii < GJ; ++ii) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < GJ; ++m) acc += W[ii][m] * av[m];
      c.row[(size_t)ii * n + col] = acc;
    }
    // the OLD column panel A_:K of this work-group's 256 rows, for the update kernel: there the work-group that holds block K of a strip
    // rewrites A_iK while the strip's other work-groups still want the old one — they read this copy instead
    const double2* src = reinterpret_cast<const double2*>(c.Ac + (size_t)col * n + K0);
    double2* dstp = reinterpret_cast<double2*>(c.row + (size_t)GJ * n + (size_t)col * GJ);
#pragma unroll
    for (int m = 0; m < GJ / 2; ++m) dstp[m] = src[m];
  }
}
// grid (strips of 16 rows, chunks of 256 columns): the chunk's piece of the pivot row panel is staged in LDS once per work-group
__global__ __launch_bounds__(256) void k_coarse_gj_update(CoarsePlan c, int k) {
  __shared__ double Cb[GJ][GJ + 1], W[GJ][GJ + 1], Rs[GJ][256 + 1];
  const int tid = threadIdx.x, ii = tid / GJ, jj = tid % GJ, n = c.npad, K0 = GJ * k, I0 = GJ * blockIdx.x, J0 = 256 * blockIdx.y;
  const int ncol = min(256, n - J0);
  W[ii][jj] = c.piv[ii * GJ + jj];
  Cb[ii][jj] = c.row[(size_t)GJ * n + (size_t)(I0 + ii) * GJ + jj];      // the OLD A_iK of this strip (k_coarse_gj_pivot's copy: the chunk that holds block K rewrites the matrix's)
#pragma unroll
  for (int m = 0; m < GJ; ++m) if (tid < ncol) Rs[m][tid] = c.row[(size_t)m * n + J0 + tid];
  __syncthreads();
  const bool pivot_strip = (int)blockIdx.x == k;
  for (int u = 0; u * GJ < ncol; ++u) {
    const int jl = GJ * u + jj, j = J0 + jl;
    if (J0 + GJ * u == K0) continue;                        // block K of this strip: below
    double* dst = c.Ac + (size_t)(I0 + ii) * n + j;
    if (pivot_strip) { *dst = Rs[ii][jl]; continue; }
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < GJ; ++m) acc += Cb[ii][m] * Rs[m][jl];
    *dst -= acc;
  }
  if (K0 >= J0 && K0 < J0 + 256) {                          // this chunk holds block K: A_KK = W, A_iK = -A_iK W
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < GJ; ++m) acc += Cb[ii][m] * W[m][jj];
    c.Ac[(size_t)(I0 + ii) * n + K0 + jj] = pivot_strip ? W[ii][jj] : -acc;
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
    for (int p = 0; p < 6; ++p) acc[p] += c.Ac[(size_t)(6 * a + p) * c.npad + j] * r;
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
  for (int k = 0; k < c.npad / GJ; ++k) {
    hipLaunchKernelGGL(k_coarse_gj_pivot, dim3((c.npad + 255) / 256), dim3(256), 0, s, c, k);
    hipLaunchKernelGGL(k_coarse_gj_update, dim3(c.npad / GJ, (c.npad + 255) / 256), dim3(256), 0, s, c, k);
  }
}
void launch_coarse_restrict(const DeviceGraph& g, const CoarsePlan& c, const double* vec, hipStream_t s, int fold_seq) {
  hipLaunchKernelGGL(k_coarse_restrict, dim3(c.a_hi - c.a_lo + (fold_seq >= 0 ? 1 : 0)), dim3(256), 0, s, g, c, vec, fold_seq);
}
void launch_coarse_correct(const DeviceGraph& g, const CoarsePlan& c, double* out, int out_seg, double* out2, hipStream_t s) {
  if (c.a_hi > c.a_lo) hipLaunchKernelGGL(k_coarse_correct, dim3(c.a_hi - c.a_lo), dim3(256), 0, s, g, c, out, out_seg, out2);
}

}  // namespace pgo
