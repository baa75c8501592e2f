// pgo_front.h — supernodal MULTIFRONTAL Cholesky of (H~ + D^2) on the GPU with FP64 MFMA dense fronts.
//
// Role in the reference: the numeric factorisation CHOLMOD (supernodal) performs behind
// `options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY` (finial.cpp:536).  pgo_direct.* serves the chain-like
// graphs (KITTI-00: a trajectory plus a few hundred chords) with enumerated 6x6 block pairs; mesh-like graphs
// (Manhattan 10 k, sphere x10) have separators of hundreds of poses, and this solver treats them as dense algebra:
//
//   host, once per topology  : nested-dissection ordering (shared with pgo_direct), block symbolic factorisation,
//                              fundamental supernodes, relaxed amalgamation (flops are cheap here, launches and
//                              dependent steps are not), postorder renumbering, one dense FRONT per supernode
//                              [F11 . ; F21 F22] over its c own poses + r update poses, tree levels, launch schedule;
//   device, every LM iteration: scatter the BSR blocks + the right-hand side into the fronts, then per tree level
//                              (leaves first): extend-add of the children's update matrices (gather per parent tile,
//                              fixed child order: deterministic, no atomics), blocked right-looking factorisation of
//                              the c own columns (48-wide panels: POTRF + explicit inverse of the diagonal block in
//                              one workgroup; TRSM and all updates as v_mfma_f64_16x16x4_f64 GEMMs; two-level blocking,
//                              192-wide outer panels, so the trailing updates run with K = 192), Schur update
//                              U = F22 - L21 L21^T with K = 6c in one GEMM launch.
//                              The right-hand side rides along as ONE EXTRA ROW of every front (Cholesky of the
//                              bordered matrix [[A b],[b' .]] leaves y = L^-1 b in that row), so the forward
//                              substitution costs nothing of its own; the backward substitution walks the tree
//                              top-down, one workgroup per front.
//
// All fronts are row-major, leading dimension = front size + 2 (the RHS row is row n of an (n+1) x n matrix; ld is even so
// every row starts 16-byte aligned).  Only the lower triangle is meaningful.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "pgo_kernels.h"

namespace pgo {

enum { FRONT_NB = 48, FRONT_NBO = 192, FRONT_TILE = 64, FRONT_ASM_TP = 8 };

// One unit of dense work on a front.  POTRF: the diagonal block [k0, k0+klen); TRSM: rows [r0, r1) of the panel
// [k0, k0+klen) times W^T (W = inverse of the panel's diagonal block); GEMM: C[r0:r1, c0:c1] -= F[r0:r1, k0:k0+klen) *
// F[c0:c1, k0:k0+klen)^T, tiles entirely above the diagonal skipped.
struct FrontJob {
  long long fbase;   // offset of the front in Fval (doubles)
  int ld;
  int r0, r1, c0, c1;
  int k0, klen;
  int wbase;         // offset of the panel's W in Winv (doubles)
  int wg_begin;      // first workgroup of the job inside its launch
  int ntc;           // GEMM: tiles per tile row
};

struct FrontLaunch {
  enum Type { POTRF = 0, TRSM = 1, GEMM = 2 };
  int type, job_begin, job_end, n_wg;
};

struct FrontLevel {
  int front_begin, front_end;     // fronts are numbered level by level
  int launch_begin, launch_end;
  int asm_front_begin;            // fronts [asm_front_begin, front_end) have children (sorted last inside the level)
  int asm_wg;                     // grid of the extend-add launch
  int max_threads_bwd;            // (unused on the device; statistics)
};

// Per-front descriptor on the device.
struct FrontDesc {
  long long fbase;
  int first;        // first own column (new numbering); own columns are [first, first + c)
  int c, r;         // own poses, update poses
  int ld;
  int idx_begin;    // update rows: idx[idx_begin .. idx_begin + r)   (new numbering, ascending)
  int child_begin, child_end;   // children: child[child_begin .. child_end)
  int rel_begin;    // as a child: rel[rel_begin .. rel_begin + r) = position (pose units) of each update row in the parent's front
  int wbase;        // W of panel p at Winv + wbase + p * FRONT_NB * FRONT_NB
  int asm_wg_begin; // first workgroup of this front in its level's extend-add launch
  int ntp;          // pose tiles of FRONT_ASM_TP per side
  int parent;
};

struct FrontPlan {
  int n, nf;
  const int* perm;          // [n] new -> old
  const FrontDesc* fronts;  // [nf]
  const int* idx;
  const int* child;
  const int* rel;
  const int* col_front;     // [n] front owning each column (new numbering)
  const int* ablk_ptr;      // [n_ablk+1] BSR slots summed into one 6x6 block of a front
  const int* ablk_slot;
  const int* ablk_front;    // [n_ablk]
  const int* ablk_pos;      // [n_ablk] (bi << 16) | bj, pose units inside the front
  int n_ablk;
  const FrontJob* jobs;
  double* Fval;
  double* Winv;
  double* x;                // [6n] solution, new numbering
};

struct FrontSymbolic {
  int n = 0, nf = 0, n_levels = 0;
  std::vector<int> perm, iperm;
  std::vector<FrontDesc> fronts;
  std::vector<int> idx, child, rel, col_front;
  std::vector<int> ablk_ptr, ablk_slot, ablk_front, ablk_pos;
  std::vector<FrontJob> jobs;
  std::vector<FrontLaunch> launches;
  std::vector<FrontLevel> levels;
  long long fval_size = 0;   // doubles
  long long winv_size = 0;
  long long factor_blocks = 0;   // 6x6 blocks of L held by the fronts (incl. the amalgamation's explicit zeros)
  double flops = 0;          // factorisation flops of the dense fronts
  double est_us = 0;         // rough time estimate of factor + solve (launch floor + flops), microseconds
  int max_front = 0;         // largest front dimension (scalars)
  int n_launches = 0;
};

// Host analysis.  Returns false when the fronts would not fit the memory budget (bytes).
bool front_analyze(int N, const std::vector<int>& ia, const std::vector<int>& ib, int n_slots,
                   const std::vector<int>& slot_row, const std::vector<int>& slot_col,
                   const std::vector<uint8_t>& slot_side, long long max_bytes, FrontSymbolic* out);

// Device launches: factorisation (includes the forward substitution) and backward substitution into g.cg_x.
// flags[2] is set when a pivot is not positive.
void launch_front_factor(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s);
void launch_front_solve(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s);

}  // namespace pgo
