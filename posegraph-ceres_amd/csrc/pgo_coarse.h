// pgo_coarse.h — r06: a COARSE LEVEL for the block-Jacobi PCG (measured in the oracle first: oracle/pgo_oracle.cpp pcg_solve, CoarseSpace;
// EXPERIMENTS.md r06).  M^-1 = M_J^-1 + P (P' A P)^-1 P' with M_J the cluster Jacobi, A = H~ + D^2 and P an aggregation coarse space:
// aggregates of `agg` consecutive poses of the trajectory, six rigid-body modes each (translation t, rotation w about the aggregate's
// centre: dp_i = t + 2 w x (p_i - c), dtheta_i = w in the tangent coordinates of Plus), expressed in the Jacobi-scaled unknowns.  The
// block Jacobi alone misses the long-wavelength correction a pose graph needs from dead reckoning (BASELINE configs[1]: truncated PCG
// ends 11 % above the exact path's cost); with the coarse level the same forcing term ends 5.6 % BELOW it, at the same CG work.
//   per LM iteration:  k_coarse_basis (P~ per pose) -> k_coarse_galerkin (P~' A P~, one work-group per aggregate row panel, fixed
//                      summation order) -> explicit inverse by block Gauss-Jordan (32 x 32 pivot blocks, one launch per block, between two copies of the matrix)
//   per application:   k_coarse_restrict (rc = P~' w, one work-group per aggregate) -> k_coarse_correct (xc = Ainv rc for the
//                      aggregate's six rows, out += P~ xc for its poses)
// Incidence-slot storage, the one-launch pipelined CG iteration (k_pipe_cg) with the correction between the launches.  Several ranks (row
// shards): every rank forms the row panels of ITS aggregates and restricts over ITS rows; the panels (per LM iteration) and the restricted
// vector (per application: 6 doubles per aggregate) are all-gathered, the inverse is replicated — bit-identical on every rank.
#pragma once
#include "pgo_kernels.h"

namespace pgo {

struct CoarsePlan {
  int agg;          // poses per aggregate
  int n_agg;        // aggregates (several ranks: per_rank aggregates for every rank's segment of rows_per poses — aggregates never straddle ranks;
                    // the ones behind a rank's last row are empty: identity rows of the coarse matrix)
  int per_rank;     // aggregates per rank segment (one rank: n_agg)
  int a_lo, a_hi;   // the aggregates of THIS rank's rows: it forms their row panels of the Galerkin matrix and their entries of a restriction
  int cdim;         // 6 * n_agg
  int npad;         // cdim rounded up to a multiple of the pivot block (32): order of the stored matrix (identity on the padding)
  double* Pt;       // [N][36] P~ of every pose, row-major (row = fine component, column = mode)
  double* Ac;       // [npad][npad] Galerkin matrix (several ranks: all-gathered row panels)
  double* Ac2;      // [npad][npad] the other copy: a step of the inversion reads one and writes the other
  double* Ainv;     // = Ac or Ac2: where the last step leaves the inverse
  double* rc;       // [npad] restricted vector
  const int* rank_end;   // several ranks: [world] one past the last REAL row of every rank's segment (device numbering; the rows behind it pad the segment)
};

// setup in two halves around the all-gather of the Galerkin matrix's row panels (several ranks; one rank calls them back to back)
void launch_coarse_galerkin(const DeviceGraph& g, const CoarsePlan& c, hipStream_t s);     // P~ of every pose, the row panels of this rank's aggregates
void launch_coarse_invert(const CoarsePlan& c, hipStream_t s);                            // identity on empty / padding rows, explicit inverse into c.Ainv
int coarse_pivot_block();                                                                 // 32: npad is a multiple of it
// application in two halves around the all-gather of the restricted vector:
// rc of this rank's aggregates = P~' vec; fold_seq >= 0: the launch also folds the partial sums of the CG launch `fold_seq` (k_pipe_fold's job)
void launch_coarse_restrict(const DeviceGraph& g, const CoarsePlan& c, const double* vec, hipStream_t s, int fold_seq = -1);
// out += P~ (Ainv rc) for the poses of this rank's aggregates; out is an exchange buffer (pose v of rank k at k * out_seg + (v - k rows_per) * 6; one
// rank: 6 v), out2 (may be null) a flat vector
void launch_coarse_correct(const DeviceGraph& g, const CoarsePlan& c, double* out, int out_seg, double* out2, hipStream_t s);

}  // namespace pgo
