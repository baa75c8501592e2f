// pgo_coarse.h — r06: a COARSE LEVEL for the block-Jacobi PCG (measured in the oracle first: oracle/pgo_oracle.cpp pcg_solve, CoarseSpace;
// EXPERIMENTS.md r06).  M^-1 = M_J^-1 + P (P' A P)^-1 P' with M_J the cluster Jacobi, A = H~ + D^2 and P an aggregation coarse space:
// aggregates of `agg` consecutive poses of the trajectory, six rigid-body modes each (translation t, rotation w about the aggregate's
// centre: dp_i = t + 2 w x (p_i - c), dtheta_i = w in the tangent coordinates of Plus), expressed in the Jacobi-scaled unknowns.  The
// block Jacobi alone misses the long-wavelength correction a pose graph needs from dead reckoning (BASELINE configs[1]: truncated PCG
// ends 11 % above the exact path's cost); with the coarse level the same forcing term ends 5.6 % BELOW it, at the same CG work.
//   per LM iteration:  k_coarse_basis (P~ per pose) -> k_coarse_galerkin (P~' A P~, one work-group per aggregate row panel, fixed
//                      summation order) -> explicit inverse by block Gauss-Jordan (16 x 16 pivot blocks, two launches per block)
//   per application:   k_coarse_restrict (rc = P~' w, one work-group per aggregate) -> k_coarse_correct (xc = Ainv rc for the
//                      aggregate's six rows, out += P~ xc for its poses)
// One rank, incidence-slot storage, the one-launch pipelined CG iteration (k_pipe_cg) with the correction between the launches.
#pragma once
#include "pgo_kernels.h"

namespace pgo {

struct CoarsePlan {
  int agg;          // poses per aggregate
  int n_agg;        // aggregates
  int cdim;         // 6 * n_agg
  int npad;         // cdim rounded up to a multiple of 16: order of the stored matrix (identity on the padding)
  double* Pt;       // [N][36] P~ of every pose, row-major (row = fine component, column = mode)
  double* Ac;       // [npad][npad] Galerkin matrix, then its inverse
  double* piv;      // [16][16] inverse of the current pivot block
  double* row;      // [16][npad] pivot row panel of the current step
  double* rc;       // [npad] restricted vector
};

void launch_coarse_setup(const DeviceGraph& g, const CoarsePlan& c, hipStream_t s);
// out[6 v + r] += (P (P' A P)^-1 P' vec)[6 v + r] for every pose (out2 likewise when not null)
void launch_coarse_apply(const DeviceGraph& g, const CoarsePlan& c, const double* vec, double* out, double* out2, hipStream_t s);

}  // namespace pgo
