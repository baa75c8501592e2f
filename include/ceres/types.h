// ceres/types.h — enums of the Ceres public API that the pose-graph path touches.
// Part of the header-only `namespace ceres` facade over the C ABI in include/pgo.h (see ceres.h).
#ifndef PGO_CERES_TYPES_H_
#define PGO_CERES_TYPES_H_

namespace ceres {

enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };

// Same enumerator order as Ceres 1.13: SPARSE_NORMAL_CHOLESKY == 2.
enum LinearSolverType {
  DENSE_NORMAL_CHOLESKY,
  DENSE_QR,
  SPARSE_NORMAL_CHOLESKY,
  DENSE_SCHUR,
  SPARSE_SCHUR,
  ITERATIVE_SCHUR,
  CGNR
};

enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL };
enum MinimizerType { LINE_SEARCH, TRUST_REGION };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

inline const char* TerminationTypeToString(TerminationType t) {
  switch (t) {
    case CONVERGENCE: return "CONVERGENCE";
    case NO_CONVERGENCE: return "NO_CONVERGENCE";
    case FAILURE: return "FAILURE";
    case USER_SUCCESS: return "USER_SUCCESS";
    case USER_FAILURE: return "USER_FAILURE";
  }
  return "UNKNOWN";
}

}  // namespace ceres
#endif
