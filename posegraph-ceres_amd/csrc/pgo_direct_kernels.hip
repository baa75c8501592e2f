// pgo_direct_kernels.hip — numeric phase of the GPU block-sparse Cholesky (pgo_direct.h): level-scheduled
// left-looking factorisation over 6x6 blocks and the two triangular solves.
//
// One wave per block column.  Ten 6-lane groups work on ten blocks of the column at a time; lane r of a group
// owns row r of its block in registers: it gathers the block from the BSR slots, subtracts its L_ik L_jk^T
// updates (fixed order), and — once the diagonal block of the column has been factored by group 0 and
// published through LDS — finishes its row with the triangular solve against L_jj.  Nothing produced inside a
// launch is re-read from global memory inside the same wave, and a column only reads columns of lower levels.
// At these sizes (10^4..10^5 blocks) the schedule is launch-latency bound, so the elimination-tree levels
// that hold at most 8 columns are folded into ONE single-workgroup launch (barrier between levels).
// What bounds a column is a chain of dependent round trips (index -> index -> value) and FP64 sqrt/div latency, not
// bandwidth; the kernels below therefore run the independent chains of a column on different waves of one workgroup:
//   k_chol_level3     COLUMN levels: diagonal block | first ten off-diagonal blocks | forward partial sums
//   k_chol_assemble4  SPLIT/PANEL assembly: four waves share a block's update list
//   k_chol_panel      PANEL chains: wave 0 factorises L_jj while waves 1..7 walk the forward row list and the sub-diagonal blocks
//   k_bwd_level4      backward levels: four waves share a column's blocks, six-lane finish
//   k_bwd_tail        backward tail: x of the tail and its column descriptors in LDS
// (the one-wave forms k_chol_level, k_chol_assemble, k_bwd_level were removed in r03 with the switches that selected them;
//  k_chol_split is the single-launch SPLIT step, used while the step has at most one workgroup per CU.)
#include "pgo_direct.h"

#include <cstdlib>

namespace pgo {
namespace {

constexpr int FUSED_WAVES = 8;

// row `r` of block `b`, 6 doubles
__device__ __forceinline__ void load_row(const double* Lval, int b, int r, double* out) {
  const double2* p = reinterpret_cast<const double2*>(Lval + 36 * (size_t)b + 6 * r);
  const double2 a = p[0], c = p[1], d = p[2];
  out[0] = a.x; out[1] = a.y; out[2] = c.x; out[3] = c.y; out[4] = d.x; out[5] = d.y;
}

// v -= sum over the update pairs q = q_begin + sub, + stride, ... < q_end of (row r of L[upd_a[q]]) * L[upd_b[q]]^T
__device__ __forceinline__ void subtract_pairs(const DirectPlan& p, int q_begin, int q_end, int r, int sub, int stride, double* v) {
  // two pairs per trip, all loads of both issued before the arithmetic (the walk is latency-bound: each pair is two
  // dependent round trips — index, then 48 + 288 bytes of L)
  int q = q_begin + sub;
  for (; q + stride < q_end; q += 2 * stride) {
    const int ia0 = p.upd_a[q], ib0 = p.upd_b[q], ia1 = p.upd_a[q + stride], ib1 = p.upd_b[q + stride];
    double a0[6], a1[6];
    load_row(p.Lval, ia0, r, a0);
    load_row(p.Lval, ia1, r, a1);
    const double2* B0 = reinterpret_cast<const double2*>(p.Lval + 36 * (size_t)ib0);
    const double2* B1 = reinterpret_cast<const double2*>(p.Lval + 36 * (size_t)ib1);
    double2 b0[18], b1[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) { b0[k] = B0[k]; b1[k] = B1[k]; }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double s0 = a0[0] * b0[3 * c].x + a0[1] * b0[3 * c].y + a0[2] * b0[3 * c + 1].x + a0[3] * b0[3 * c + 1].y +
                        a0[4] * b0[3 * c + 2].x + a0[5] * b0[3 * c + 2].y;
      const double s1 = a1[0] * b1[3 * c].x + a1[1] * b1[3 * c].y + a1[2] * b1[3 * c + 1].x + a1[3] * b1[3 * c + 1].y +
                        a1[4] * b1[3 * c + 2].x + a1[5] * b1[3 * c + 2].y;
      v[c] -= s0;
      v[c] -= s1;
    }
  }
  if (q < q_end) {
    double a[6];
    load_row(p.Lval, p.upd_a[q], r, a);
    const double2* B = reinterpret_cast<const double2*>(p.Lval + 36 * (size_t)p.upd_b[q]);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double2 x0 = B[3 * c], x1 = B[3 * c + 1], x2 = B[3 * c + 2];
      v[c] -= a[0] * x0.x + a[1] * x0.y + a[2] * x1.x + a[3] * x1.y + a[4] * x2.x + a[5] * x2.y;
    }
  }
}

// Partial assembly of row r of block `bi`: BSR sources (only when sub == 0) minus the update pairs up to q_end
// (upd_ptr[bi+1] = all of them; upd_split[bi] = those of the columns before the block's panel);
// the `stride` partial results are summed by the caller.
__device__ __forceinline__ void assemble_row(const DeviceGraph& g, const DirectPlan& p, int bi, int r, int sub, int stride, double* v, int q_end) {
#pragma unroll
  for (int c = 0; c < 6; ++c) v[c] = 0.0;
  if (sub == 0) {
    for (int s = p.asrc_ptr[bi]; s < p.asrc_ptr[bi + 1]; ++s) {
      const int slot = p.asrc_slot[s];
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] += bsr_elem(g, slot, g.slot_side[slot], 6 * r + c);
    }
  }
  subtract_pairs(p, p.upd_ptr[bi], q_end, r, sub, stride, v);
}
__device__ __forceinline__ void assemble_row(const DeviceGraph& g, const DirectPlan& p, int bi, int r, int sub, int stride, double* v) {
  assemble_row(g, p, bi, r, sub, stride, v, p.upd_ptr[bi + 1]);
}

// forward-substitution pieces (defined below): the forward solve L y = P (S g) is fused into the factorisation — as soon
// as L_jj exists, y_j = L_jj^-1 (b_j - sum_k L_jk y_k) needs only columns of earlier levels
__device__ __forceinline__ void forward_partial(const DirectPlan& p, int j, int sub, int nsub, double* sh);
__device__ __forceinline__ double forward_rhs(const DeviceGraph& g, int old);
__device__ __forceinline__ void forward_finish(const DirectPlan& p, int j, double b, int nsub, const double* sh_all, const double* Ljj);
// the ten partial copies of a 6x6 block in sh[10][36] -> their sum (fixed order) in every lane of the wave
__device__ __forceinline__ void sum_partials(double* sh, double* out, const double* base) {
  const int lane = threadIdx.x & 63;
  if (lane < 36) {
    double s = base ? base[lane] : 0.0;
#pragma unroll
    for (int gq = 0; gq < 10; ++gq) s += sh[gq * 36 + lane];
    sh[lane] = s;          // entry `lane` of copy 0 is read by this lane only
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < 36; ++k) out[k] = sh[k];
}

// Factorises column j with one wave.  sh: 360 doubles of LDS private to the wave (10 groups x 6 rows x 6).
// The ten 6-lane groups share the work of a chunk of up to ten blocks: with fewer blocks than groups the
// update list of each block is split over several groups and the partial rows are summed through LDS
// (fixed order), so the long lists of the separator columns near the root do not serialise on 6 lanes.
__device__ void factor_column(const DeviceGraph& g, const DirectPlan& p, int j, double* sh) {
  const int lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  const bool in_grp = grp < 10;
  const int b0 = p.col_ptr[j], nblk = p.col_ptr[j + 1] - b0;
  const int old = p.perm[j];                    // static data of the fused forward step, fetched behind the assembly
  // ---- diagonal block: its update list is split over all ten groups ----
  if (in_grp) {
    double v[6];
    assemble_row(g, p, b0, r, grp, 10, v);
#pragma unroll
    for (int c = 0; c < 6; ++c) sh[(grp * 6 + r) * 6 + c] = v[c];
  }
  const double bj = forward_rhs(g, old);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double Ljj[36];
  sum_partials(sh, Ljj, nullptr);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double d = Ljj[7 * c];
#pragma unroll
    for (int k = 0; k < c; ++k) d -= Ljj[6 * c + k] * Ljj[6 * c + k];
    if (!(d > 0.0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    Ljj[7 * c] = d;
    const double inv = 1.0 / d;
#pragma unroll
    for (int i = c + 1; i < 6; ++i) {
      double s = Ljj[6 * i + c];
#pragma unroll
      for (int k = 0; k < c; ++k) s -= Ljj[6 * i + k] * Ljj[6 * c + k];
      Ljj[6 * i + c] = s * inv;
    }
#pragma unroll
    for (int i = 0; i < c; ++i) Ljj[6 * i + c] = 0.0;
  }
  if (!ok && lane == 0) atomicOr(&g.flags[2], 1);
  if (lane < 36) p.Lval[36 * (size_t)b0 + lane] = Ljj[lane];
  // ---- off-diagonal blocks in chunks of up to ten: row r of L_ij = (row r of V_ij) * L_jj^-T ----
  for (int t0 = 1; t0 < nblk; t0 += 10) {
    const int bc = min(10, nblk - t0);
    const int gpb = 10 / bc;                       // groups per block
    const int my = grp / gpb, sub = grp - my * gpb;
    const bool active = in_grp && my < bc;
    double v[6];
    if (active) {
      assemble_row(g, p, b0 + t0 + my, r, sub, gpb, v);
      if (gpb > 1) {
#pragma unroll
        for (int c = 0; c < 6; ++c) sh[(grp * 6 + r) * 6 + c] = v[c];
      }
    }
    if (gpb > 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (active && sub == 0) {
        for (int q = 1; q < gpb; ++q) {
#pragma unroll
          for (int c = 0; c < 6; ++c) v[c] += sh[((grp + q) * 6 + r) * 6 + c];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (active && sub == 0) {
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s = v[c];
#pragma unroll
        for (int k = 0; k < c; ++k) s -= x[k] * Ljj[6 * c + k];
        x[c] = s / Ljj[7 * c];
      }
      double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)(b0 + t0 + my) + 6 * r);
      o[0] = double2{x[0], x[1]}; o[1] = double2{x[2], x[3]}; o[2] = double2{x[4], x[5]};
    }
  }
  // fused forward substitution for this column (row j of L lives in columns of earlier levels)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  forward_partial(p, j, 0, 1, sh);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  forward_finish(p, j, bj, 1, sh, Ljj);
}

// In-register Cholesky of a 6x6 block held (row-major, 36 doubles) by every lane; returns false on a non-positive pivot.
__device__ __forceinline__ bool chol6_inplace(double* Ljj) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double d = Ljj[7 * c];
#pragma unroll
    for (int k = 0; k < c; ++k) d -= Ljj[6 * c + k] * Ljj[6 * c + k];
    if (!(d > 0.0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    Ljj[7 * c] = d;
    const double inv = 1.0 / d;
#pragma unroll
    for (int i = c + 1; i < 6; ++i) {
      double s = Ljj[6 * i + c];
#pragma unroll
      for (int k = 0; k < c; ++k) s -= Ljj[6 * i + k] * Ljj[6 * c + k];
      Ljj[6 * i + c] = s * inv;
    }
#pragma unroll
    for (int i = 0; i < c; ++i) Ljj[6 * i + c] = 0.0;
  }
  return ok;
}

// ---- SPLIT levels (dense separators): phase 1, four waves per BLOCK: V = A - sum_q L[upd_a] L[upd_b]^T, the update list shared by
// forty 6-lane groups, partial rows summed through LDS in a fixed order.  A diagonal block is factorised on the spot (L_jj) and the
// forward step of its column shared by the four waves too; an off-diagonal block is left as V in Lval for phase 2. ----
constexpr int ASM_WAVES = 4;
__global__ __launch_bounds__(64 * ASM_WAVES) void k_chol_assemble4(DeviceGraph g, DirectPlan p, int blk_begin) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[ASM_WAVES][360];
  __shared__ double shf[ASM_WAVES][64];
  __shared__ double Ld[36];
  const int bi = p.split_blk[blk_begin + blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  if (grp < 10) {
    double v[6];
    assemble_row(g, p, bi, r, wave * 10 + grp, 10 * ASM_WAVES, v, p.upd_split[bi]);
#pragma unroll
    for (int c = 0; c < 6; ++c) sh[wave][(grp * 6 + r) * 6 + c] = v[c];
  }
  __syncthreads();
  const bool diagonal = p.split_diag[blk_begin + blockIdx.x] != 0;
  if (wave == 0 && lane < 36) {        // fixed order: wave by wave, group by group
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < ASM_WAVES; ++w)
#pragma unroll
      for (int gq = 0; gq < 10; ++gq) s += sh[w][gq * 36 + lane];
    if (!diagonal) p.Lval[36 * (size_t)bi + lane] = s;
    else Ld[lane] = s;
  }
  if (!diagonal) return;
  const int j = p.blk_row[bi];
  double bj = 0.0;
  if (wave == 0) bj = forward_rhs(g, p.perm[j]);
  forward_partial(p, j, wave, ASM_WAVES, shf[wave]);
  __syncthreads();
  if (wave == 0) {
    double Ljj[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) Ljj[k] = Ld[k];
    const bool ok = chol6_inplace(Ljj);
    if (!ok && lane == 0) atomicOr(&g.flags[2], 1);
    if (lane < 36) p.Lval[36 * (size_t)bi + lane] = Ljj[lane];
    forward_finish(p, j, bj, ASM_WAVES, shf[0], Ljj);
  }
}

// SPLIT step as ONE launch: as k_chol_assemble4, but a sub-diagonal block does not leave V behind for a scaling launch —
// its workgroup waits (flag = this factorisation's epoch, polled by one lane, bounded) until the column's diagonal block
// has been published by ITS workgroup, then applies L_jj^-T itself.  The launcher uses this only while every workgroup of
// the step is resident at once (<= 256 blocks: one per CU, launch_direct_factor), so a waiting workgroup cannot keep its producer off the chip;
// a wait that runs out sets the failure flag instead of hanging.
__global__ __launch_bounds__(64 * ASM_WAVES) void k_chol_split(DeviceGraph g, DirectPlan p, int blk_begin, int epoch, int max_spins) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[ASM_WAVES][360];
  __shared__ double shf[ASM_WAVES][64];
  __shared__ double Ld[36];
  const int bi = p.split_blk[blk_begin + blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  if (grp < 10) {
    double v[6];
    assemble_row(g, p, bi, r, wave * 10 + grp, 10 * ASM_WAVES, v, p.upd_split[bi]);
#pragma unroll
    for (int c = 0; c < 6; ++c) sh[wave][(grp * 6 + r) * 6 + c] = v[c];
  }
  __syncthreads();
  const bool diagonal = p.split_diag[blk_begin + blockIdx.x] != 0;
  if (wave == 0) {
    if (lane < 36) {        // fixed order: wave by wave, group by group
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < ASM_WAVES; ++w)
#pragma unroll
        for (int gq = 0; gq < 10; ++gq) s += sh[w][gq * 36 + lane];
      Ld[lane] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!diagonal) {
    if (wave != 0) return;
    const int dblk = p.split_dblk[blk_begin + blockIdx.x];
    if (lane == 0) {
      int spins = 0;
      while (__hip_atomic_load(&p.col_flag[dblk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        if (++spins > max_spins || ((spins & 255) == 0 && (__hip_atomic_load(&g.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))) { atomicOr(&g.flags[2], 2); break; }   // (once one wait has run out nobody waits long: the factorisation is going to be repeated)
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) {
      const double* L = p.Lval + 36 * (size_t)dblk;     // lower triangular L_jj, published before the flag
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s = Ld[6 * lane + c];
#pragma unroll
        for (int k = 0; k < c; ++k) s -= x[k] * L[6 * c + k];
        x[c] = s / L[7 * c];
      }
      double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)bi + 6 * lane);
      o[0] = double2{x[0], x[1]}; o[1] = double2{x[2], x[3]}; o[2] = double2{x[4], x[5]};
    }
    return;
  }
  const int j = p.blk_row[bi];
  double bj = 0.0;
  double Ljj[36];
  if (wave == 0) {          // publish L_jj first: the column's other workgroups are waiting for it
#pragma unroll
    for (int k = 0; k < 36; ++k) Ljj[k] = Ld[k];
    const bool ok = chol6_inplace(Ljj);
    if (!ok && lane == 0) atomicOr(&g.flags[2], 1);
    if (lane < 36) p.Lval[36 * (size_t)bi + lane] = Ljj[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(&p.col_flag[bi], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bj = forward_rhs(g, p.perm[j]);
  }
  forward_partial(p, j, wave, ASM_WAVES, shf[wave]);
  __syncthreads();
  if (wave == 0) forward_finish(p, j, bj, ASM_WAVES, shf[0], Ljj);
}

// phase 2, one 6-lane group per sub-diagonal block of the level (ten per wave, lane = row): L_ij = V_ij L_jj^-T
__global__ __launch_bounds__(64) void k_chol_scale(DeviceGraph g, DirectPlan p, int sub_begin, int sub_end) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  const int lane = threadIdx.x;
  const int grp = lane / 6, r = lane - 6 * grp;
  const int q = sub_begin + blockIdx.x * 10 + grp;
  if (grp >= 10 || q >= sub_end) return;
  const int bi = p.split_sub[q];
  const double* L = p.Lval + 36 * (size_t)p.split_sub_diag[q];   // L_jj of the block's column (lower triangular)
  double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)bi + 6 * r);
  const double2 a = o[0], b = o[1], c2 = o[2];
  const double v[6] = {a.x, a.y, b.x, b.y, c2.x, c2.y};
  double x[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double s = v[c];
#pragma unroll
    for (int k = 0; k < c; ++k) s -= x[k] * L[6 * c + k];
    x[c] = s / L[7 * c];
  }
  o[0] = double2{x[0], x[1]}; o[1] = double2{x[2], x[3]}; o[2] = double2{x[4], x[5]};
}

// ---- PANEL steps, phase 2: one workgroup per chain of `width` columns; the contributions of every column before the
// panel are already in Lval (phase 1 = k_chol_assemble over all blocks of the panel).  Column by column, two phases:
// wave 0 adds the in-panel pairs of the diagonal block and factorises it WHILE waves 1..7 walk the row list of the fused
// forward step and bring the sub-diagonal blocks up to date (their few in-panel pairs; one 6-lane group per block) —
// none of that needs L_jj; after the barrier wave 0 finishes the forward step and the others apply L_jj^-T. ----
__global__ __launch_bounds__(64 * FUSED_WAVES) void k_chol_panel(DeviceGraph g, DirectPlan p, int cols_begin, int width) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[360];
  __shared__ double Ld[36];
  __shared__ double shf[FUSED_WAVES][64];
  constexpr int W = FUSED_WAVES - 1;            // waves on the forward step and the sub-diagonal blocks
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  for (int i = 0; i < width; ++i) {
    const int j = p.panel_cols[cols_begin + blockIdx.x * width + i];
    const int b0 = p.col_ptr[j], nblk = p.col_ptr[j + 1] - b0;
    double bj = 0.0;
    double v[6] = {0, 0, 0, 0, 0, 0};
    int t = 1 + (wave - 1) * 10 + grp;
    const bool first = wave > 0 && grp < 10 && t < nblk;
    if (wave == 0) {
      bj = forward_rhs(g, p.perm[j]);
      if (grp < 10) {
        double d[6] = {0, 0, 0, 0, 0, 0};
        subtract_pairs(p, p.upd_split[b0], p.upd_ptr[b0 + 1], r, grp, 10, d);
#pragma unroll
        for (int c = 0; c < 6; ++c) sh[(grp * 6 + r) * 6 + c] = d[c];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double Ljj[36];
      sum_partials(sh, Ljj, p.Lval + 36 * (size_t)b0);
      const bool ok = chol6_inplace(Ljj);
      if (!ok && lane == 0) atomicOr(&g.flags[2], 1);
      if (lane < 36) { p.Lval[36 * (size_t)b0 + lane] = Ljj[lane]; Ld[lane] = Ljj[lane]; }
    } else {
      forward_partial(p, j, wave - 1, W, shf[wave]);
      if (first) {
        const double2* o = reinterpret_cast<const double2*>(p.Lval + 36 * (size_t)(b0 + t) + 6 * r);
        const double2 a = o[0], b = o[1], c2 = o[2];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c2.x; v[5] = c2.y;
        subtract_pairs(p, p.upd_split[b0 + t], p.upd_ptr[b0 + t + 1], r, 0, 1, v);
      }
    }
    __syncthreads();
    if (wave == 0) {
      forward_finish(p, j, bj, W, shf[1], Ld);
    } else if (grp < 10) {
      for (bool pre = first; t < nblk; t += 10 * W, pre = false) {
        const int bi = b0 + t;
        double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)bi + 6 * r);
        if (!pre) {
          const double2 a = o[0], b = o[1], c2 = o[2];
          v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c2.x; v[5] = c2.y;
          subtract_pairs(p, p.upd_split[bi], p.upd_ptr[bi + 1], r, 0, 1, v);
        }
        double x[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double s = v[c];
#pragma unroll
          for (int k = 0; k < c; ++k) s -= x[k] * Ld[6 * c + k];
          x[c] = s / Ld[7 * c];
        }
        o[0] = double2{x[0], x[1]}; o[1] = double2{x[2], x[3]}; o[2] = double2{x[4], x[5]};
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

// The same column by three cooperating waves: the diagonal block (gather, pair walk, 6x6 Cholesky), the first ten
// off-diagonal blocks (gather, pair walk) and the partial sums of the fused forward step do not depend on each other, so
// their chains of dependent round trips run side by side; after one barrier L_jj is in LDS and the off-diagonal rows are
// scaled while the forward step finishes.  Arithmetic and summation order are those of factor_column.
__device__ void factor_column_roles(const DeviceGraph& g, const DirectPlan& p, int j, int role, double* shd, double* sho,
                                    double* shf, double* Ld) {
  const int lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  const bool in_grp = grp < 10;
  const int b0 = p.col_ptr[j], nblk = p.col_ptr[j + 1] - b0;
  double v[6] = {0, 0, 0, 0, 0, 0};
  bool active = false;
  int my = 0, sub = 0;
  double bj = 0.0;
  if (role == 0) {
    if (in_grp) {
      double d[6];
      assemble_row(g, p, b0, r, grp, 10, d);
#pragma unroll
      for (int c = 0; c < 6; ++c) shd[(grp * 6 + r) * 6 + c] = d[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double Ljj[36];
    sum_partials(shd, Ljj, nullptr);
    const bool ok = chol6_inplace(Ljj);
    if (!ok && lane == 0) atomicOr(&g.flags[2], 1);
    if (lane < 36) { p.Lval[36 * (size_t)b0 + lane] = Ljj[lane]; Ld[lane] = Ljj[lane]; }
  } else if (role == 1) {
    if (nblk > 1) {
      const int bc = min(10, nblk - 1);
      const int gpb = 10 / bc;
      my = grp / gpb; sub = grp - my * gpb;
      active = in_grp && my < bc;
      if (active) {
        assemble_row(g, p, b0 + 1 + my, r, sub, gpb, v);
        if (gpb > 1) {
#pragma unroll
          for (int c = 0; c < 6; ++c) sho[(grp * 6 + r) * 6 + c] = v[c];
        }
      }
      if (gpb > 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (active && sub == 0) {
          for (int q = 1; q < gpb; ++q) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += sho[((grp + q) * 6 + r) * 6 + c];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
    bj = forward_rhs(g, p.perm[j]);
    forward_partial(p, j, 0, 1, shf);
  }
  __syncthreads();
  if (role == 1) {
    if (active && sub == 0) {
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s = v[c];
#pragma unroll
        for (int k = 0; k < c; ++k) s -= x[k] * Ld[6 * c + k];
        x[c] = s / Ld[7 * c];
      }
      double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)(b0 + 1 + my) + 6 * r);
      o[0] = double2{x[0], x[1]}; o[1] = double2{x[2], x[3]}; o[2] = double2{x[4], x[5]};
    }
    for (int t0 = 11; t0 < nblk; t0 += 10) {     // columns with more than ten off-diagonal blocks: the rest, as factor_column
      const int bc = min(10, nblk - t0);
      const int gpb = 10 / bc;
      my = grp / gpb; sub = grp - my * gpb;
      active = in_grp && my < bc;
      if (active) {
        assemble_row(g, p, b0 + t0 + my, r, sub, gpb, v);
        if (gpb > 1) {
#pragma unroll
          for (int c = 0; c < 6; ++c) sho[(grp * 6 + r) * 6 + c] = v[c];
        }
      }
      if (gpb > 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (active && sub == 0) {
          for (int q = 1; q < gpb; ++q) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] += sho[((grp + q) * 6 + r) * 6 + c];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (active && sub == 0) {
        double x[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double s = v[c];
#pragma unroll
          for (int k = 0; k < c; ++k) s -= x[k] * Ld[6 * c + k];
          x[c] = s / Ld[7 * c];
        }
        double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)(b0 + t0 + my) + 6 * r);
        o[0] = double2{x[0], x[1]}; o[1] = double2{x[2], x[3]}; o[2] = double2{x[4], x[5]};
      }
    }
  } else if (role == 2) {
    forward_finish(p, j, bj, 1, shf, Ld);
  }
}

__global__ __launch_bounds__(192) void k_chol_level3(DeviceGraph g, DirectPlan p, int level) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double shd[360], sho[360], shf[64], Ld[36];
  const int j = p.level_cols[p.level_ptr[level] + blockIdx.x];
  factor_column_roles(g, p, j, threadIdx.x >> 6, shd, sho, shf, Ld);
}

// ---- wide levels (the union of a batched solve: tens of thousands of light columns per level) -------------------------------
// One 6-lane GROUP per column, ten columns per wave, forty per workgroup: a level of n columns keeps 30 x fewer waves busy
// than k_chol_level3 (three waves per column — the right shape when a level has a few hundred columns and latency is all that
// matters).  Lane r of a group owns row r of every block of its column: assembly (the whole update list, in list order),
// the 6 x 6 factor redundantly in the six lanes (rows exchanged through LDS), scaling, and the fused forward step.
constexpr int GRP_WAVES = 4;
__global__ __launch_bounds__(64 * GRP_WAVES) void k_chol_level_grp(DeviceGraph g, DirectPlan p, int level) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[GRP_WAVES * 10 * 36];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  const int c0 = p.level_ptr[level], nc = p.level_ptr[level + 1] - c0;
  const int col = (blockIdx.x * GRP_WAVES + wave) * 10 + grp;
  const bool on = grp < 10 && col < nc;
  double* my = sh + (wave * 10 + (grp < 10 ? grp : 0)) * 36;
  int j = 0, b0 = 0, nblk = 0, old = 0;
  double v[6];
  if (on) {
    j = p.level_cols[c0 + col];
    b0 = p.col_ptr[j];
    nblk = p.col_ptr[j + 1] - b0;
    old = p.perm[j];
    assemble_row(g, p, b0, r, 0, 1, v);
#pragma unroll
    for (int c = 0; c < 6; ++c) my[6 * r + c] = v[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  double Ljj[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) Ljj[k] = on ? my[k] : (k % 7 == 0 ? 1.0 : 0.0);
  const bool ok = chol6_inplace(Ljj);
  if (on) {
    if (!ok && r == 0) atomicOr(&g.flags[2], 1);
    double2* o = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)b0 + 6 * r);
    o[0] = double2{Ljj[6 * r], Ljj[6 * r + 1]}; o[1] = double2{Ljj[6 * r + 2], Ljj[6 * r + 3]}; o[2] = double2{Ljj[6 * r + 4], Ljj[6 * r + 5]};
    // off-diagonal blocks: row r of L_ij = (row r of V_ij) L_jj^-T
    for (int t = 1; t < nblk; ++t) {
      assemble_row(g, p, b0 + t, r, 0, 1, v);
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double t2 = v[c];
#pragma unroll
        for (int k = 0; k < c; ++k) t2 -= x[k] * Ljj[6 * c + k];
        x[c] = t2 / Ljj[7 * c];
      }
      double2* ob = reinterpret_cast<double2*>(p.Lval + 36 * (size_t)(b0 + t) + 6 * r);
      ob[0] = double2{x[0], x[1]}; ob[1] = double2{x[2], x[3]}; ob[2] = double2{x[4], x[5]};
    }
    // fused forward step: component r of b_j - sum_k L_jk y_k
    const double b = g.scale[6 * (size_t)old + r] * g.grad[6 * (size_t)old + r];
    g.cg_b[6 * (size_t)old + r] = b;
    double acc = 0.0;
    for (int q = p.rowl_ptr[j]; q < p.rowl_ptr[j + 1]; ++q) {
      double a[6];
      load_row(p.Lval, p.rowl_blk[q], r, a);
      const double* yk = p.y + 6 * (size_t)p.rowl_col[q];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc += a[c] * yk[c];
    }
    my[r] = b - acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (on) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double t2 = my[i];
#pragma unroll
      for (int k = 0; k < i; ++k) t2 -= Ljj[6 * i + k] * y[k];
      y[i] = t2 / Ljj[7 * i];
    }
    double out = y[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) out = r == k ? y[k] : out;
    p.y[6 * (size_t)j + r] = out;
  }
}

// backward level, one group per column: lane c owns component c of y_j - sum_i L_ij^T x_i
__global__ __launch_bounds__(64 * GRP_WAVES) void k_bwd_level_grp(DeviceGraph g, DirectPlan p, int level) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[GRP_WAVES * 10 * 6];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / 6, c = lane - 6 * grp;
  const int c0 = p.level_ptr[level], nc = p.level_ptr[level + 1] - c0;
  const int col = (blockIdx.x * GRP_WAVES + wave) * 10 + grp;
  const bool on = grp < 10 && col < nc;
  double* my = sh + (wave * 10 + (grp < 10 ? grp : 0)) * 6;
  int j = 0, b0 = 0;
  if (on) {
    j = p.level_cols[c0 + col];
    b0 = p.col_ptr[j];
    const int nblk = p.col_ptr[j + 1] - b0;
    double acc = 0.0;
    for (int t = 1; t < nblk; ++t) {
      const double* B = p.Lval + 36 * (size_t)(b0 + t);
      const double* xi = p.y + 6 * (size_t)p.blk_row[b0 + t];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += B[6 * k + c] * xi[k];
    }
    my[c] = p.y[6 * (size_t)j + c] - acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (on) {
    const double* L = p.Lval + 36 * (size_t)b0;
    double x[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      double t2 = my[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) t2 -= L[6 * k + i] * x[k];
      x[i] = t2 / L[7 * i];
    }
    double out = x[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) out = c == k ? x[k] : out;
    p.y[6 * (size_t)j + c] = out;
    g.cg_x[6 * (size_t)p.perm[j] + c] = out;
  }
}

__global__ __launch_bounds__(64 * FUSED_WAVES) void k_chol_tail(DeviceGraph g, DirectPlan p, int from_level, int to_level) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[FUSED_WAVES][360];
  const int wave = threadIdx.x >> 6;
  for (int l = from_level; l < to_level; ++l) {
    const int c0 = p.level_ptr[l], nc = p.level_ptr[l + 1] - c0;
    if (wave < nc) factor_column(g, p, p.level_cols[c0 + wave], sh[wave]);
    __threadfence_block();
    __syncthreads();
  }
}

// ---- forward solve  L y = P (S g):  y_j = L_jj^-1 (b_j - sum_k L_jk y_k) ----
// The row list of column j is shared by `nsub` waves x ten 6-lane groups; partial sums land in sh[64] of each wave
// (sh_all = the first of the nsub consecutive per-wave arrays), the finishing lane adds them in a fixed order.
__device__ __forceinline__ void forward_partial(const DirectPlan& p, int j, int sub, int nsub, double* sh) {
  const int lane = threadIdx.x & 63;
  const int grp = lane / 6, r = lane - 6 * grp;
  double acc = 0.0;
  if (grp < 10) {
    for (int q = p.rowl_ptr[j] + sub * 10 + grp; q < p.rowl_ptr[j + 1]; q += 10 * nsub) {
      double a[6];
      load_row(p.Lval, p.rowl_blk[q], r, a);
      const double* yk = p.y + 6 * (size_t)p.rowl_col[q];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc += a[c] * yk[c];
    }
    sh[lane] = acc;
  }
}
// component `lane` (lanes 0..5) of the right-hand side row of old pose `old`: b = S g; also written to cg_b
__device__ __forceinline__ double forward_rhs(const DeviceGraph& g, int old) {
  const int lane = threadIdx.x & 63;
  const int i = lane < 6 ? lane : 0;
  const double b = g.scale[6 * (size_t)old + i] * g.grad[6 * (size_t)old + i];
  if (lane < 6) g.cg_b[6 * (size_t)old + lane] = b;
  return b;
}
// called by a whole wave: lanes 0..5 add the partial sums of one component each (fixed order), the 6x6 triangular solve
// runs redundantly in every lane, lanes 0..5 store
__device__ __forceinline__ void forward_finish(const DirectPlan& p, int j, double b, int nsub, const double* sh_all, const double* L) {
  const int lane = threadIdx.x & 63;
  const int i = lane < 6 ? lane : 0;
  double s = 0.0;
  for (int w = 0; w < nsub; ++w) {                 // ten reads in flight, then the adds in the fixed order
    double t[10];
#pragma unroll
    for (int gq = 0; gq < 10; ++gq) t[gq] = sh_all[64 * w + 6 * gq + i];
#pragma unroll
    for (int gq = 0; gq < 10; ++gq) s += t[gq];
  }
  const double mine = b - s;
  double rhs[6], y[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) rhs[k] = __shfl(mine, k);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double t = rhs[r];
#pragma unroll
    for (int k = 0; k < r; ++k) t -= L[6 * r + k] * y[k];
    y[r] = t / L[7 * r];
  }
  double out = y[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) out = lane == k ? y[k] : out;
  if (lane < 6) p.y[6 * (size_t)j + lane] = out;
}

// ---- backward solve  L^T x = y:  x_j = L_jj^-T (y_j - sum_{i in struct(j)} L_ij^T x_i) ----
__device__ __forceinline__ void backward_partial(const DirectPlan& p, int j, int sub, int nsub, double* sh) {
  const int lane = threadIdx.x & 63;
  const int grp = lane / 6, c = lane - 6 * grp;   // lane owns COLUMN c of L_ij (component c of L_ij^T x_i)
  const int b0 = p.col_ptr[j], nblk = p.col_ptr[j + 1] - b0;
  double acc = 0.0;
  if (grp < 10) {
    for (int t = 1 + sub * 10 + grp; t < nblk; t += 10 * nsub) {
      const double* B = p.Lval + 36 * (size_t)(b0 + t);
      const double* xi = p.y + 6 * (size_t)p.blk_row[b0 + t];
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += B[6 * k + c] * xi[k];
    }
    sh[lane] = acc;
  }
}
__device__ __forceinline__ void backward_finish(const DeviceGraph& g, const DirectPlan& p, int j, int nsub, const double* sh_all) {
  const int b0 = p.col_ptr[j];
  double rhs[6], x[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = 0.0;
    for (int w = 0; w < nsub; ++w)
      for (int gq = 0; gq < 10; ++gq) s += sh_all[64 * w + 6 * gq + i];
    rhs[i] = p.y[6 * (size_t)j + i] - s;
  }
  const double* L = p.Lval + 36 * (size_t)b0;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = rhs[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
    x[i] = s / L[7 * i];
  }
  const int old = p.perm[j];
#pragma unroll
  for (int i = 0; i < 6; ++i) { p.y[6 * (size_t)j + i] = x[i]; g.cg_x[6 * (size_t)old + i] = x[i]; }
}

// Backward level: four waves per column (forty 6-lane groups on the column's blocks) and the six-lane finish of the tail,
// its inputs (y_j, L_jj, the permutation) fetched beside the partial sums.
constexpr int BWD_WAVES = 4;
__global__ __launch_bounds__(64 * BWD_WAVES) void k_bwd_level4(DeviceGraph g, DirectPlan p, int level) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[BWD_WAVES][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = p.level_cols[p.level_ptr[level] + blockIdx.x];
  backward_partial(p, j, wave, BWD_WAVES, sh[wave]);
  double yj = 0.0, Ld[21];
  int old = 0;
  if (wave == 0) {
    old = p.perm[j];
    yj = p.y[6 * (size_t)j + (lane < 6 ? lane : 0)];
    const double* L = p.Lval + 36 * (size_t)p.col_ptr[j];
#pragma unroll
    for (int i = 0, q = 0; i < 6; ++i)
#pragma unroll
      for (int k = i; k < 6; ++k) Ld[q++] = L[6 * k + i];     // row i of L^T from the diagonal on
  }
  __syncthreads();
  if (wave != 0) return;
  const int i = lane < 6 ? lane : 0;
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < BWD_WAVES; ++w) {
    double t[10];
#pragma unroll
    for (int gq = 0; gq < 10; ++gq) t[gq] = sh[w][6 * gq + i];
#pragma unroll
    for (int gq = 0; gq < 10; ++gq) s += t[gq];
  }
  const double my_rhs = yj - s;
  double rhs[6], x[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) rhs[k] = __shfl(my_rhs, k);
  int q = 21;
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    q -= 6 - r;                                  // Ld[q] = L[7r], Ld[q + (k - r)] = L[6k + r]
    double t = rhs[r];
#pragma unroll
    for (int k = r + 1; k < 6; ++k) t -= Ld[q + (k - r)] * x[k];
    x[r] = t / Ld[q];
  }
  double out = x[0];
#pragma unroll
  for (int k = 1; k < 6; ++k) out = lane == k ? x[k] : out;
  if (lane < 6) {
    p.y[6 * (size_t)j + lane] = out;
    g.cg_x[6 * (size_t)old + lane] = out;
  }
}
// Fused tail of the backward solve (levels with <= FUSED_WAVES columns): the FUSED_WAVES waves of the single workgroup are
// divided among the columns of the level, so the long lists of a dense separator chain (one column per level) are walked 80-wide.
// The rows of a tail column are its ancestors, i.e. tail columns too: with XLDS the x of the whole tail stays in LDS
// (indexed by position in level_cols), so the level-to-level dependency never leaves the CU; everything else a level
// reads (L, y, the indices) is final before the launch and is fetched ahead of the barrier.
constexpr int BWD_TAIL_LDS_COLS = 896;
template <bool XLDS>
__global__ __launch_bounds__(64 * FUSED_WAVES) void k_bwd_tail(DeviceGraph g, DirectPlan p, int from_level) {
  if (lm_halted(g)) return;   // device-resident LM: a sequence enqueued ahead of a halt (pgo_kernels.h LmDev)
  __shared__ double sh[FUSED_WAVES][64];
  __shared__ double xs[XLDS ? 6 * BWD_TAIL_LDS_COLS : 6];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grp = lane / 6, c = lane - 6 * grp;      // lane owns COLUMN c of L_ij (component c of L_ij^T x_i)
  const int tail_begin = p.level_ptr[from_level];
  // with XLDS the column descriptors and level bounds of the whole tail are staged once, so a level's index chain
  // (level -> column -> first block) is three LDS reads instead of three dependent trips to memory
  __shared__ int tj[XLDS ? BWD_TAIL_LDS_COLS : 1], tb0[XLDS ? BWD_TAIL_LDS_COLS : 1], tnb[XLDS ? BWD_TAIL_LDS_COLS : 1];
  __shared__ int tl[XLDS ? BWD_TAIL_LDS_COLS + 1 : 1];
  if (XLDS) {
    for (int q = threadIdx.x; q < p.n - tail_begin; q += blockDim.x) {
      const int jq = p.level_cols[tail_begin + q];
      const int b = p.col_ptr[jq];
      tj[q] = jq; tb0[q] = b; tnb[q] = p.col_ptr[jq + 1] - b;
    }
    for (int q = threadIdx.x; q <= p.n_levels - from_level; q += blockDim.x) tl[q] = p.level_ptr[from_level + q];
    __syncthreads();
  }
  for (int l = p.n_levels - 1; l >= from_level; --l) {
    const int c0 = XLDS ? tl[l - from_level] : p.level_ptr[l];
    const int nc = (XLDS ? tl[l - from_level + 1] : p.level_ptr[l + 1]) - c0;
    const int nsub = max(1, FUSED_WAVES / nc);
    const int col = wave / nsub, sub = wave - col * nsub;
    const bool mine = col < nc, finisher = mine && sub == 0;
    int j = 0, b0 = 0, nblk = 0;
    if (mine) {
      if (XLDS) { const int q = c0 + col - tail_begin; j = tj[q]; b0 = tb0[q]; nblk = tnb[q]; }
      else { j = p.level_cols[c0 + col]; b0 = p.col_ptr[j]; nblk = p.col_ptr[j + 1] - b0; }
    }
    double acc = 0.0;
    if (mine && grp < 10) {
      for (int t = 1 + sub * 10 + grp; t < nblk; t += 10 * nsub) {
        const double* B = p.Lval + 36 * (size_t)(b0 + t);
        double bk[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) bk[k] = B[6 * k + c];
        const double* xi = XLDS ? xs + 6 * (p.blk_lpos[b0 + t] - tail_begin) : p.y + 6 * (size_t)p.blk_row[b0 + t];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += bk[k] * xi[k];
      }
    }
    {   // the wave's ten group sums per component, folded in registers (fixed order); six values per wave go to LDS
      const double mine = grp < 10 ? acc : 0.0;
      const int comp = lane < 6 ? lane : 0;
      double fold = 0.0;
#pragma unroll
      for (int gq = 0; gq < 10; ++gq) fold += __shfl(mine, 6 * gq + comp);
      if (lane < 6) sh[wave][lane] = fold;
    }
    // the finisher's inputs do not depend on this level: fetched beside the partial sums, ahead of the barrier
    double yj = 0.0, Ld[21];
    int old = 0;
    if (finisher) {
      old = p.perm[j];
      yj = p.y[6 * (size_t)j + (lane < 6 ? lane : 0)];
      const double* L = p.Lval + 36 * (size_t)b0;
#pragma unroll
      for (int i = 0, q = 0; i < 6; ++i)
#pragma unroll
        for (int k = i; k < 6; ++k) Ld[q++] = L[6 * k + i];     // row i of L^T from the diagonal on
    }
    __syncthreads();
    if (finisher) {
      const int i = lane < 6 ? lane : 0;
      double s = 0.0;
      {                                              // the column's waves, in order (nsub <= FUSED_WAVES)
        double t[FUSED_WAVES];
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; ++w) t[w] = w < nsub ? sh[wave + w][i] : 0.0;
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; ++w) s += t[w];
      }
      const double my_rhs = yj - s;
      double rhs[6], x[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) rhs[k] = __shfl(my_rhs, k);
      int q = 21;
#pragma unroll
      for (int r = 5; r >= 0; --r) {
        q -= 6 - r;                                  // Ld[q] = L[7r], Ld[q + (k - r)] = L[6k + r]
        double t = rhs[r];
#pragma unroll
        for (int k = r + 1; k < 6; ++k) t -= Ld[q + (k - r)] * x[k];
        x[r] = t / Ld[q];
      }
      double out = x[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) out = lane == k ? x[k] : out;
      if (lane < 6) {
        p.y[6 * (size_t)j + lane] = out;
        g.cg_x[6 * (size_t)old + lane] = out;
        if (XLDS) xs[6 * (c0 + col - tail_begin) + lane] = out;
      }
    }
    if (!XLDS) __threadfence_block();
    __syncthreads();
  }
}

}  // namespace

void launch_direct_factor(const DeviceGraph& g, const DirectPlan& p, const DirectSymbolic& sym, hipStream_t s, int epoch) {
  // single-launch SPLIT steps: by default only while the step has at most one workgroup per CU (every workgroup resident
  // from the start; chain-like graphs: KITTI-00 replay -56 us per LM iteration; raising the limit to 1024 gained nothing on
  // dense separators: the wait costs what the launch did)
  static const int fuse_split_max = 256;
  // (block assembly with four waves per block — KITTI-00 dense 10.7 -> ~8 us per launch —, three waves per column in the COLUMN
  // levels — 1-4 us per level of KITTI-00 — and four per column in the backward levels were switches in r01 / r02
  // (PGO_DIRECT_ASM4 / _ROLES / _BWD4); the one-wave forms they kept alive went with them in r03)
  for (const DirectStep& st : sym.steps) {
    if (st.type == DirectStep::COLUMN) {
      const int nc = sym.level_ptr[st.level_begin + 1] - sym.level_ptr[st.level_begin];
      // wide levels (batched solves): one 6-lane group per column from PGO_DIRECT_GROUPS columns on (default 4096)
      static const int grp_min = 4096;
      if (nc >= grp_min) hipLaunchKernelGGL(k_chol_level_grp, dim3((nc + 10 * GRP_WAVES - 1) / (10 * GRP_WAVES)), dim3(64 * GRP_WAVES), 0, s, g, p, st.level_begin);
      else hipLaunchKernelGGL(k_chol_level3, dim3(nc), dim3(192), 0, s, g, p, st.level_begin);
    } else if (st.type == DirectStep::FUSED) {
      hipLaunchKernelGGL(k_chol_tail, dim3(1), dim3(64 * FUSED_WAVES), 0, s, g, p, st.level_begin, st.level_end);
    } else if (st.type == DirectStep::PANEL) {
      hipLaunchKernelGGL(k_chol_assemble4, dim3(st.blk_end - st.blk_begin), dim3(64 * ASM_WAVES), 0, s, g, p, st.blk_begin);
      hipLaunchKernelGGL(k_chol_panel, dim3(st.sub_end), dim3(64 * FUSED_WAVES), 0, s, g, p, st.sub_begin, st.level_end - st.level_begin);
    } else if (epoch > 0 && st.blk_end - st.blk_begin <= fuse_split_max) {
      // bounded wait (PGO_WAIT_SPINS, default 2^22 polls ~ 1 s): when it runs out — the workgroups of the step were
      // not all resident, e.g. on a partitioned or shared GPU — the solve is flagged and the LM driver repeats the
      // factorisation in the two-launch form (epoch 0) and keeps to it for this problem
      const char* spins_env = getenv("PGO_WAIT_SPINS");     // (read per call: the tests change it within one process)
      const int max_spins = spins_env ? atoi(spins_env) : (1 << 22);
      hipLaunchKernelGGL(k_chol_split, dim3(st.blk_end - st.blk_begin), dim3(64 * ASM_WAVES), 0, s, g, p, st.blk_begin, epoch, max_spins);
    } else {
      hipLaunchKernelGGL(k_chol_assemble4, dim3(st.blk_end - st.blk_begin), dim3(64 * ASM_WAVES), 0, s, g, p, st.blk_begin);
      const int nsub = st.sub_end - st.sub_begin;
      if (nsub > 0) hipLaunchKernelGGL(k_chol_scale, dim3((nsub + 9) / 10), dim3(64), 0, s, g, p, st.sub_begin, st.sub_end);
    }
  }
}

// backward solve only: the forward substitution runs inside the factorisation kernels
void launch_direct_solve(const DeviceGraph& g, const DirectPlan& p, const int* level_ptr_host, int fused_from_level, hipStream_t s) {
  if (fused_from_level < p.n_levels) {
    if (p.n - level_ptr_host[fused_from_level] <= BWD_TAIL_LDS_COLS)
      hipLaunchKernelGGL(k_bwd_tail<true>, dim3(1), dim3(64 * FUSED_WAVES), 0, s, g, p, fused_from_level);
    else
      hipLaunchKernelGGL(k_bwd_tail<false>, dim3(1), dim3(64 * FUSED_WAVES), 0, s, g, p, fused_from_level);
  }
  for (int l = fused_from_level - 1; l >= 0; --l) {
    const int nc = level_ptr_host[l + 1] - level_ptr_host[l];
    static const int grp_min = 4096;
    if (nc >= grp_min) hipLaunchKernelGGL(k_bwd_level_grp, dim3((nc + 10 * GRP_WAVES - 1) / (10 * GRP_WAVES)), dim3(64 * GRP_WAVES), 0, s, g, p, l);
    else hipLaunchKernelGGL(k_bwd_level4, dim3(nc), dim3(64 * BWD_WAVES), 0, s, g, p, l);
  }
}

}  // namespace pgo

// ---- development stress test of the exchange transports (tools/comm_stress.py) ----
#include "pgo_comm.h"
namespace pgo {
namespace {
__global__ void k_stress_fill(double* buf, size_t seg, int rank, int it) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < seg) buf[(size_t)rank * seg + i] = 1000.0 * it + 10.0 * rank + (double)(i % 7);
}
__global__ void k_stress_check(const double* buf, size_t seg, int world, int it, int* bad) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= seg * world) return;
  const int r = (int)(i / seg);
  const double want = 1000.0 * it + 10.0 * r + (double)((i - (size_t)r * seg) % 7);
  if (buf[i] != want) atomicAdd(bad, 1);
}
}  // namespace
int comm_stress(Comm* c, int iters, size_t seg, hipStream_t s, int* mismatches) {
  double* buf = nullptr;
  int* bad = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), c->world * seg * sizeof(double)) != hipSuccess) return -1;
  if (hipMalloc(reinterpret_cast<void**>(&bad), sizeof(int)) != hipSuccess) return -1;
  (void)hipMemsetAsync(bad, 0, sizeof(int), s);
  (void)hipMemsetAsync(buf, 0, c->world * seg * sizeof(double), s);
  const char* what = "";
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(k_stress_fill, dim3((seg + 255) / 256), dim3(256), 0, s, buf, seg, c->rank, it);
    if (c->all_gather(buf, seg, s, &what) != 0) return -2;
    hipLaunchKernelGGL(k_stress_check, dim3((seg * c->world + 255) / 256), dim3(256), 0, s, buf, seg, c->world, it, bad);
    if ((it & 63) == 63) (void)hipStreamSynchronize(s);
  }
  (void)hipStreamSynchronize(s);
  (void)hipMemcpy(mismatches, bad, sizeof(int), hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  (void)hipFree(bad);
  return 0;
}
}  // namespace pgo
