// pgo_lm_rules.h — the trust-region rules of Ceres 1.13's TrustRegionMinimizer + LevenbergMarquardtStrategy (SURVEY.md
// Appendix A.6 steps 4-7; the reference reaches them through ceres::Solve, finial.cpp:538), written ONCE for both places that
// apply them: the device (last work-group of the step tail, pgo_kernels.hip — the default since r03) and the host driver
// (several ranks, batched solve, PGO_NO_PIPELINE=1).  Only explicit IEEE operations and fma: host and device produce the same
// bits, so a solve takes the same decisions whichever side decides (tests/test_gpu_pipeline.py compares the two traces).
#pragma once
#include <math.h>

#include "pgo_kernels.h"

#if defined(__HIPCC__)
#define PGO_RULE_HD __host__ __device__
#else
#define PGO_RULE_HD
#endif

namespace pgo {

// What a trial step hands to the rules.
struct LmStepIn {
  double cand_cost, model_change, step_norm_sq, x_norm_sq;
  int cg_iterations, cg_status, linearize_bad, pad;
};
enum LmOutcome {
  LM_OUT_INVALID = 0,        // HandleInvalidStep: radius halved, record written (unsuccessful)
  LM_OUT_INVALID_FAIL = 1,   // ... and max_num_consecutive_invalid_steps reached: FAILURE, no record
  LM_OUT_PARAM_TOL = 2,      // ParameterToleranceReached on the candidate: CONVERGENCE, the step is NOT applied, no record
  LM_OUT_FUNC_TOL = 3,       // FunctionToleranceReached on the candidate: likewise
  LM_OUT_ACCEPT = 4,
  LM_OUT_REJECT = 5
};

// t^3 rounded once (double-double product, explicit fma): what pow(t, 3) of a correctly rounding libm returns.
// Contraction must stay off: hipcc's default (-ffp-contract=fast) fuses the final `q + ...` with the product that formed q and
// counts q's rounding error twice (measured on gfx950: 26 % of the results one ulp off the host's).
PGO_RULE_HD inline double lm_cube(double t) {
#pragma clang fp contract(off)
  const double p = t * t;
  const double e = fma(t, t, -p);
  const double q = p * t;
  const double eq = fma(p, t, -q);
  return q + fma(e, t, eq);
}

// One pass of the TrustRegionMinimizer loop body behind ComputeCandidatePointAndEvaluateCost.  `nx` is the record of the new
// iteration (valid for INVALID / ACCEPT / REJECT; its trust_region_radius is the radius AFTER the update, which is what
// Ceres logs); `term_value` the number the termination message quotes.
PGO_RULE_HD inline LmOutcome lm_decide(LmCore& L, const LmTolerances& o, const LmStepIn& sc, LmRecord& nx, double& term_value) {
#pragma clang fp contract(off)
  nx.iteration = L.iteration + 1;
  nx.step_is_successful = 0;
  nx.linear_solver_iterations = sc.cg_iterations;
  nx.reserved = 0;
  nx.cost = L.x_cost;
  nx.cost_change = 0.0;
  nx.gradient_max_norm = L.gmax;
  nx.step_norm = 0.0;
  nx.relative_decrease = 0.0;
  L.reuse_diagonal = 1;
  const bool lin_ok = (sc.cg_status != 2) && isfinite(sc.model_change) && !sc.linearize_bad;
  const bool step_valid = lin_ok && sc.model_change > 0.0;
  if (!step_valid) {
    ++L.num_consecutive_invalid;
    if (L.num_consecutive_invalid >= o.max_consecutive_invalid) {
      L.iteration = nx.iteration;
      return LM_OUT_INVALID_FAIL;
    }
    L.radius = L.radius * 0.5;
    L.iteration = nx.iteration;
    nx.trust_region_radius = L.radius;
    return LM_OUT_INVALID;
  }
  L.num_consecutive_invalid = 0;
  nx.step_norm = sqrt(sc.step_norm_sq);
  L.x_norm = sqrt(sc.x_norm_sq);
  const double step_size_tolerance = o.parameter_tolerance * (L.x_norm + o.parameter_tolerance);
  if (nx.step_norm <= step_size_tolerance) {
    term_value = nx.step_norm / (L.x_norm + o.parameter_tolerance);
    return LM_OUT_PARAM_TOL;
  }
  nx.cost_change = L.x_cost - sc.cand_cost;
  if (fabs(nx.cost_change) <= o.function_tolerance * L.x_cost) {
    term_value = fabs(nx.cost_change) / L.x_cost;
    return LM_OUT_FUNC_TOL;
  }
  nx.relative_decrease = nx.cost_change / sc.model_change;
  LmOutcome out;
  if (nx.relative_decrease > o.min_relative_decrease) {
    // LevenbergMarquardtStrategy::StepAccepted
    L.x_cost = sc.cand_cost;
    nx.step_is_successful = 1;
    nx.cost = L.x_cost;
    const double t = 2.0 * nx.relative_decrease - 1.0;
    L.radius = L.radius / fmax(1.0 / 3.0, 1.0 - lm_cube(t));
    L.radius = fmin(o.max_radius, L.radius);
    L.decrease_factor = 2.0;
    L.reuse_diagonal = 0;
    out = LM_OUT_ACCEPT;
  } else {
    // StepRejected
    nx.cost = sc.cand_cost;
    L.radius = L.radius / L.decrease_factor;
    L.decrease_factor = L.decrease_factor * 2.0;
    out = LM_OUT_REJECT;
  }
  L.iteration = nx.iteration;
  nx.trust_region_radius = L.radius;
  return out;
}

}  // namespace pgo
