// pgo_report.cpp — Summary::FullReport / IsSolutionUsable (finial.cpp:541-543) in Ceres 1.13's layout; host-side text only.
#include "pgo_internal.h"



// =================================================================================================
// C ABI (include/pgo.h)
// =================================================================================================
extern "C" {

int pgo_summary_is_solution_usable(const pgo_solver_summary* s) {
  return s && (s->termination_type == PGO_CONVERGENCE || s->termination_type == PGO_NO_CONVERGENCE) ? 1 : 0;
}

size_t pgo_summary_full_report(const pgo_solver_summary* s, const pgo_iteration_record* rec, int n_rec, char* buffer, size_t capacity) {
  std::string r;
  char line[512];
  auto add = [&](const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    r += line;
  };
  static const char* term[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
  // Layout of Ceres 1.13 Solver::Summary::FullReport (solver.cc): two columns Original / Reduced, Given / Used, the same
  // row labels and widths, so that parsers of the reference's stdout (finial.cpp:541) keep working; what Ceres has no row
  // for (factorisation statistics, device) follows in a block of its own.
  add("\nSolver Summary (v %d.%d.%d-hip-gfx950)\n\n", PGO_VERSION / 100, (PGO_VERSION / 10) % 10, PGO_VERSION % 10);
  add("%45s    %21s\n", "Original", "Reduced");
  add("Parameter blocks    % 25d% 25d\n", 2 * s->num_poses, s->num_parameter_blocks_reduced);
  add("Parameters          % 25d% 25d\n", 7 * s->num_poses, s->num_parameters_reduced);
  add("Effective parameters% 25d% 25d\n", 6 * s->num_poses, s->num_effective_parameters_reduced);
  add("Residual blocks     % 25d% 25d\n", s->num_edges, s->num_edges);
  add("Residual            % 25d% 25d\n", 6 * s->num_edges, 6 * s->num_edges);
  add("\nMinimizer                 %19s\n", "TRUST_REGION");
  add("\nSparse linear algebra library %15s\n", "HIP_GFX950");
  add("Trust region strategy     %19s\n", "LEVENBERG_MARQUARDT");
  add("\n%45s    %21s\n", "Given", "Used");
  const bool exact = s->linear_solver_used != 1;
  add("Linear solver       %25s%25s\n", exact ? "SPARSE_NORMAL_CHOLESKY" : "CGNR",
      s->linear_solver_used == 2 ? "CGNR" : exact ? "SPARSE_NORMAL_CHOLESKY" : "CGNR");
  if (!exact || s->linear_solver_used == 2) add("Preconditioner      %25s%25s\n", "JACOBI", "JACOBI");
  add("Threads             % 25d% 25d\n", 1, 1);
  add("Linear solver threads % 23d% 25d\n", 1, 1);
  if (exact) add("Linear solver ordering %22s% 25d\n", "AUTOMATIC", s->num_parameter_blocks_reduced);
  add("\nCost:\n");
  add("Initial        % 30e\n", s->initial_cost);
  add("Final          % 30e\n", s->final_cost);
  add("Change         % 30e\n", s->initial_cost - s->final_cost);
  add("\nMinimizer iterations         % 16d\n", s->num_iterations);
  add("Successful steps             % 16d\n", s->num_successful_steps);
  add("Unsuccessful steps           % 16d\n", s->num_unsuccessful_steps);
  add("\nTime (in seconds):\n");
  add("Preprocessor        %25.4f\n", s->setup_time_in_seconds);
  add("\n  Residual evaluation %23.4f\n", s->residual_evaluation_time_in_seconds);
  add("  Jacobian evaluation %23.4f\n", s->jacobian_evaluation_time_in_seconds);
  add("  Linear solver       %23.4f\n", s->linear_solver_time_in_seconds);
  add("Minimizer           %25.4f\n", s->total_time_in_seconds);
  add("\nPostprocessor       %25.4f\n", 0.0);
  add("Total               %25.4f\n", s->total_time_in_seconds + s->setup_time_in_seconds);
  static const char* ls[] = {"GPU factorisation", "block-Jacobi PCG (Q-tolerance eta)", "PCG to exact_r_tolerance (factorisation declined)",
                             "GPU factorisation or PCG to exact_r_tolerance, chosen per iteration"};
  add("\nGPU path (no Ceres counterpart):\n");
  add("Compute device              HIP gfx950 (FP64)\n");
  add("Linear solves served by     %s\n", ls[(s->linear_solver_used >= 0 && s->linear_solver_used <= 3) ? s->linear_solver_used : 1]);
  add("Linear solver iterations     % 16d\n", s->num_linear_solver_iterations);
  if (s->factor_nnz_blocks > 0) {
    add("Factorisation               %s\n", s->factor_kind == 3 ? "supernodal multifrontal, fronts in LDS" : s->factor_kind == 2 ? "supernodal multifrontal, FP64 MFMA fronts" : "enumerated 6x6 block pairs, nested dissection");
    add("Factor blocks / levels       % 16d / %d\n", s->factor_nnz_blocks, s->factor_levels);
    add("Factorisations               % 16d\n", s->num_factorizations);
  }
  add("\n");
  if (rec && n_rec > 0) {
    add("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter\n");
    for (int i = 0; i < n_rec; ++i)
      add("%4d %.6e %10.2e %10.2e %9.2e %10.2e %9.2e %8d\n", rec[i].iteration, rec[i].cost, rec[i].cost_change,
          rec[i].gradient_max_norm, rec[i].step_norm, rec[i].relative_decrease, rec[i].trust_region_radius,
          rec[i].linear_solver_iterations);
    add("\n");
  }
  const int t = (s->termination_type >= 0 && s->termination_type <= 2) ? s->termination_type : 2;
  add("Termination: %24s (%s)\n", term[t], s->message);   // "Termination:   %25s (%s)" in Ceres, same tokens
  if (buffer && capacity) {
    const size_t n = std::min(capacity - 1, r.size());
    memcpy(buffer, r.data(), n);
    buffer[n] = 0;
  }
  return r.size() + 1;
}

// ---- evaluation entry points -------------------------------------------------------------------

}  // extern "C"
