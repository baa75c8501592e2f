// ceres/ceres.h — umbrella header of the `namespace ceres` facade (#include <ceres/ceres.h>, finial.cpp:16).
//
// Header-only: compiles into the caller's translation unit and forwards to the C ABI of libpgo_hip.so
// (include/pgo.h).  It provides exactly the Ceres surface pose_graph_ceres_plus_finial.cpp imports
// (SURVEY.md §8b): Problem, AddResidualBlock, SetParameterization, SetParameterBlockConstant, Solve,
// Solver::Options / Summary (FullReport, IsSolutionUsable), CostFunction, SizedCostFunction,
// AutoDiffCostFunction + Jet, HuberLoss, EigenQuaternionParameterization.  Build: -I<repo>/include,
// link -lpgo_hip.  See INTEGRATION.md.
#ifndef PGO_CERES_CERES_H_
#define PGO_CERES_CERES_H_
#include "ceres/autodiff_cost_function.h"
#include "ceres/cost_function.h"
#include "ceres/jet.h"
#include "ceres/local_parameterization.h"
#include "ceres/loss_function.h"
#include "ceres/problem.h"
#include "ceres/sized_cost_function.h"
#include "ceres/solver.h"
#include "ceres/types.h"
#endif
