// pgo_direct.h — exact linear solve of (H~ + D^2) x = S g on the GPU: block-sparse Cholesky over 6x6 pose
// blocks with a nested-dissection ordering and an elimination-tree level schedule.  This is the path behind
// ceres::SPARSE_NORMAL_CHOLESKY (finial.cpp:536; role of SparseNormalCholeskySolver + CHOLMOD in the
// reference).  The symbolic phase runs once per topology on the host; the numeric
// factorisation and both triangular solves run on the device every LM iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "pgo_kernels.h"

namespace pgo {

// Device view of the symbolic structure (all indices in the PERMUTED numbering unless noted).
struct DirectPlan {
  int n;            // block columns (= poses)
  int nb;           // blocks of L (diagonal + strictly lower)
  int n_levels;
  const int* perm;        // [n]  new -> old pose index
  const int* col_ptr;     // [n+1] first block of a column is its diagonal block
  const int* blk_row;     // [nb] row of each block
  const int* asrc_ptr;    // [nb+1] BSR slots whose values initialise the block (summed)
  const int* asrc_slot;
  const int* upd_ptr;     // [nb+1] update pairs: block -= L[upd_a] * L[upd_b]^T
  const int* upd_a;
  const int* upd_b;
  const int* level_ptr;   // [n_levels+1] columns by elimination-tree level
  const int* level_cols;  // [n]
  const int* rowl_ptr;    // [n+1] forward solve: blocks (j,k), k < j, of row j ...
  const int* rowl_blk;    //       ... block index
  const int* rowl_col;    //       ... column k
  const int* split_blk;       // blocks of the "split" levels (heavy levels: one wave per BLOCK assembles, see DirectStep)
  const uint8_t* split_diag;  // ... 1 when the block is the diagonal block of its column
  const int* split_sub;       // sub-diagonal blocks of the split levels ...
  const int* split_sub_diag;  // ... and the diagonal block of their column
  const int* upd_split;       // [nb] PANEL steps: update pairs [upd_ptr, upd_split) come from columns before the panel
  const int* panel_cols;      // PANEL steps: chains of columns, chain c of a step at [begin + c*width, begin + (c+1)*width)
  const int* blk_lpos;        // [nb] position in level_cols of the block's ROW (backward tail: x of the tail lives in LDS)
  const int* split_dblk;      // per split_blk entry: the diagonal block of the block's column (fused SPLIT steps wait on it)
  int* col_flag;              // [nb] indexed by diagonal block: epoch of the factorisation that last published L_jj
  double* Lval;           // [nb][36] row-major blocks of the factor
  double* y;              // [6n] permuted work vector
};

// Launch schedule of the factorisation.  A level of the elimination tree is processed either
//   COLUMN : one wave per block column does everything (assemble, 6x6 Cholesky, scale) — many light columns;
//   FUSED  : a run of consecutive light levels with <= 8 columns each in ONE single-workgroup launch (chain-like tops);
//   SPLIT  : heavy level (long update lists: dense separators) — one wave per BLOCK assembles V = A - sum L L^T across
//            the whole GPU (the diagonal block is factorised on the spot), then one 6-lane group per block scales;
//   PANEL  : `width` consecutive heavy levels whose columns form parent chains (a dense separator): the contributions
//            of all columns BEFORE the panel are assembled one wave per block for every block of the panel at once,
//            then one workgroup per chain finishes its columns in order (the few in-panel pairs, 6x6 Cholesky, scaling)
//            — two launches for `width` levels instead of two per level, and one critical path instead of `width`.
struct DirectStep {
  enum Type { COLUMN = 0, FUSED = 1, SPLIT = 2, PANEL = 3 };
  int type, level_begin, level_end;   // levels [begin, end)
  int blk_begin, blk_end;             // SPLIT: range in split_blk
  int sub_begin, sub_end;             // SPLIT: range in split_sub;  PANEL: sub_begin = offset in panel_cols, sub_end = chains
};

struct DirectSymbolic {
  // host copies (kept for tests / statistics)
  std::vector<int> perm, iperm, col_ptr, blk_row, asrc_ptr, asrc_slot, upd_ptr, upd_a, upd_b, level_ptr, level_cols,
      rowl_ptr, rowl_blk, rowl_col;
  int n = 0, nb = 0, n_levels = 0;
  long long n_pairs = 0;
  int fused_from_level = 0;  // first level of the suffix whose levels hold <= 8 columns (forward/backward solves fuse it)
  std::vector<DirectStep> steps;   // factorisation schedule
  std::vector<int> split_blk, split_sub, split_sub_diag, upd_split, panel_cols, blk_lpos, split_dblk;
  std::vector<uint8_t> split_diag;
  double flops = 0;
  double est_steps = 0;      // critical-path length of the schedule in update-pair steps (cost model)
  bool hybrid = false;       // est_steps above the always-direct budget: the LM driver chooses per iteration between this
                             // factorisation and PCG run to exact_r_tolerance (whichever the last iterations made cheaper)
};

// Nested-dissection ordering of the pose graph (perm[new] = old); shared with the multifrontal solver (pgo_front.h).
// is_point (optional, [N]): 1 for the 3-D point blocks of a pose / landmark problem — they are eliminated first (the Schur
// complement onto the poses), the poses by nested dissection of the reduced graph
bool nested_dissection_order(int N, const std::vector<int>& ia, const std::vector<int>& ib, std::vector<int>* perm,
                             const std::vector<uint8_t>* is_point = nullptr);

// Host symbolic analysis.  slot_* describe the incidence-slot BSR (pgo_problem.cpp prepare()).
// Returns false when the factorisation would be impractical (caller falls back to the iterative path).
bool direct_analyze(int N, const std::vector<int>& ia, const std::vector<int>& ib, int n_slots,
                    const std::vector<int>& slot_row, const std::vector<int>& slot_col,
                    const std::vector<uint8_t>& slot_side, const std::vector<int>& row_slot_begin,
                    DirectSymbolic* out, const std::vector<uint8_t>* is_point = nullptr);

// Device launches.  flags[2] is set when a pivot is not positive.
// epoch > 0 (a new value per factorisation): SPLIT steps run as ONE launch, sub-diagonal blocks waiting in-kernel for their
// column's L_jj; 0 (stream capture: the argument would be frozen): assemble + scale launches.
void launch_direct_factor(const DeviceGraph& g, const DirectPlan& p, const DirectSymbolic& sym, hipStream_t s, int epoch = 0);
void launch_direct_solve(const DeviceGraph& g, const DirectPlan& p, const int* level_ptr_host, int fused_from_level, hipStream_t s);

}  // namespace pgo
