// ceres/solver.h — ceres::Solver::Options / Summary and ceres::Solve (finial.cpp:531-544) forwarding to
// pgo_solve() of include/pgo.h.
//
// What Solve() does: every residual block must be an SE(3) between factor, i.e. a 6-residual cost over
// blocks {3,4,3,4} whose residual is  L * [ R(q_a)^T (p_b - p_a) - p_hat ; 2 vec(q_hat (x) conj(q_a^-1 q_b)) ]
// (PLUS/include/PoseGraph3dError.h:21-54).  The functor's measurement and sqrt-information are private, so
// they are RECOVERED by probing CostFunction::Evaluate: with q_a = identity, p_a = 0 the
// residual is affine in p_b and linear in q_b; 9 residual-only evaluations give L, p_hat, q_hat, and the
// recovered factor is validated against Evaluate at random poses before it is trusted.  Anything that does
// not validate is reported as FAILURE in the summary ("unsupported cost function"), loudly: there is no
// host-side generic NLLS fallback.
#ifndef PGO_CERES_SOLVER_H_
#define PGO_CERES_SOLVER_H_

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "ceres/problem.h"
#include "ceres/types.h"
#include "pgo.h"

namespace ceres {

struct IterationSummary {
  int iteration;
  bool step_is_successful;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int linear_solver_iterations;
};

class Solver {
 public:
  struct Options {
    Options() {
      pgo_solver_options d;
      pgo_solver_options_init(&d);
      minimizer_type = TRUST_REGION;
      trust_region_strategy_type = LEVENBERG_MARQUARDT;
      max_num_iterations = d.max_num_iterations;
      max_solver_time_in_seconds = 1e9;
      num_threads = 1;
      num_linear_solver_threads = 1;
      initial_trust_region_radius = d.initial_trust_region_radius;
      max_trust_region_radius = d.max_trust_region_radius;
      min_trust_region_radius = d.min_trust_region_radius;
      min_relative_decrease = d.min_relative_decrease;
      min_lm_diagonal = d.min_lm_diagonal;
      max_lm_diagonal = d.max_lm_diagonal;
      max_num_consecutive_invalid_steps = d.max_num_consecutive_invalid_steps;
      function_tolerance = d.function_tolerance;
      gradient_tolerance = d.gradient_tolerance;
      parameter_tolerance = d.parameter_tolerance;
      linear_solver_type = SPARSE_NORMAL_CHOLESKY;
      preconditioner_type = JACOBI;
      min_linear_solver_iterations = d.min_linear_solver_iterations;
      max_linear_solver_iterations = d.max_linear_solver_iterations;
      eta = d.eta;
      jacobi_scaling = d.jacobi_scaling != 0;
      use_nonmonotonic_steps = false;
      use_inner_iterations = false;
      minimizer_progress_to_stdout = false;
      update_state_every_iteration = false;
    }
    MinimizerType minimizer_type;
    TrustRegionStrategyType trust_region_strategy_type;
    int max_num_iterations;
    double max_solver_time_in_seconds;
    int num_threads, num_linear_solver_threads;
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius, min_relative_decrease;
    double min_lm_diagonal, max_lm_diagonal;
    int max_num_consecutive_invalid_steps;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    LinearSolverType linear_solver_type;
    PreconditionerType preconditioner_type;
    int min_linear_solver_iterations, max_linear_solver_iterations;
    double eta;
    bool jacobi_scaling, use_nonmonotonic_steps, use_inner_iterations, minimizer_progress_to_stdout, update_state_every_iteration;
    bool IsValid(std::string* error) const {
      if (minimizer_type != TRUST_REGION || trust_region_strategy_type != LEVENBERG_MARQUARDT || use_nonmonotonic_steps || use_inner_iterations) {
        if (error) *error = "the MI355X pose-graph path implements TRUST_REGION / LEVENBERG_MARQUARDT, monotonic steps, no inner iterations";
        return false;
      }
      return true;
    }
  };

  struct Summary {
    Summary() : termination_type(FAILURE), initial_cost(-1), final_cost(-1), num_successful_steps(-1), num_unsuccessful_steps(-1),
                total_time_in_seconds(-1), num_parameter_blocks(0), num_residual_blocks(0) {
      std::memset(&raw, 0, sizeof raw);
      message = "ceres::Solve was not called.";
    }
    // A brief one line description of the state of the solver after termination.
    std::string BriefReport() const {
      char b[512];
      std::snprintf(b, sizeof b, "Ceres Solver Report (pgo/gfx950): Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
                    (int)iterations.size(), initial_cost, final_cost, TerminationTypeToString(termination_type));
      return b;
    }
    // finial.cpp:541
    std::string FullReport() const {
      std::vector<pgo_iteration_record> rec(iterations.size());
      for (size_t i = 0; i < iterations.size(); ++i) {
        std::memset(&rec[i], 0, sizeof rec[i]);
        rec[i].iteration = iterations[i].iteration; rec[i].step_is_successful = iterations[i].step_is_successful;
        rec[i].linear_solver_iterations = iterations[i].linear_solver_iterations; rec[i].cost = iterations[i].cost;
        rec[i].cost_change = iterations[i].cost_change; rec[i].gradient_max_norm = iterations[i].gradient_max_norm;
        rec[i].step_norm = iterations[i].step_norm; rec[i].relative_decrease = iterations[i].relative_decrease;
        rec[i].trust_region_radius = iterations[i].trust_region_radius;
      }
      pgo_solver_summary s = raw;
      std::snprintf(s.message, sizeof s.message, "%s", message.c_str());
      s.termination_type = termination_type == CONVERGENCE ? PGO_CONVERGENCE : termination_type == NO_CONVERGENCE ? PGO_NO_CONVERGENCE : PGO_FAILURE;
      const size_t n = pgo_summary_full_report(&s, rec.empty() ? 0 : &rec[0], (int)rec.size(), 0, 0);
      std::string out(n, '\0');
      pgo_summary_full_report(&s, rec.empty() ? 0 : &rec[0], (int)rec.size(), &out[0], n);
      out.resize(std::strlen(out.c_str()));
      return out;
    }
    // finial.cpp:543
    bool IsSolutionUsable() const {
      return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE || termination_type == USER_SUCCESS;
    }
    TerminationType termination_type;
    std::string message;
    double initial_cost, final_cost;
    int num_successful_steps, num_unsuccessful_steps;
    double total_time_in_seconds;
    int num_parameter_blocks, num_residual_blocks;
    std::vector<IterationSummary> iterations;
    pgo_solver_summary raw;
  };

  virtual ~Solver() {}
  virtual void Solve(const Options& options, Problem* problem, Summary* summary);
};

namespace internal {

// host restatement of the between-factor residual for validating recovered factors (xyzw quaternions)
inline void QuatMul(const double* a, const double* b, double* r) {
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
inline void BetweenResidual(const double* pa, const double* qa, const double* pb, const double* qb, const double* mp,
                            const double* mq, const double* L, double* r) {
  const double u[3] = {-qa[0], -qa[1], -qa[2]}, w = qa[3], d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  double ud[3] = {2 * (u[1] * d[2] - u[2] * d[1]), 2 * (u[2] * d[0] - u[0] * d[2]), 2 * (u[0] * d[1] - u[1] * d[0])};
  const double c[3] = {u[1] * ud[2] - u[2] * ud[1], u[2] * ud[0] - u[0] * ud[2], u[0] * ud[1] - u[1] * ud[0]};
  double e[6];
  for (int i = 0; i < 3; ++i) e[i] = d[i] + w * ud[i] + c[i] - mp[i];
  const double qbc[4] = {-qb[0], -qb[1], -qb[2], qb[3]};
  double t[4], dq[4];
  QuatMul(qbc, qa, t);
  QuatMul(mq, t, dq);
  e[3] = 2 * dq[0]; e[4] = 2 * dq[1]; e[5] = 2 * dq[2];
  for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += L[6 * i + j] * e[j]; r[i] = s; }
}

struct RecoveredFactor { double p[3], q[4], L[36]; };

inline double Det3(const double m[3][3]) {
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// Probes cost->Evaluate (residuals only) and recovers (p_hat, q_hat, L).  Returns false when the cost is
// not an SE(3) between factor of the reference's form.
inline bool RecoverBetweenFactor(const CostFunction* cost, RecoveredFactor* out) {
  const std::vector<int32>& sz = cost->parameter_block_sizes();
  if (cost->num_residuals() != 6 || sz.size() != 4 || sz[0] != 3 || sz[1] != 4 || sz[2] != 3 || sz[3] != 4) return false;
  const double pa[3] = {0, 0, 0}, qa[4] = {0, 0, 0, 1};
  double pb[3] = {0, 0, 0}, qb[4] = {0, 0, 0, 0};
  const double* blocks[4] = {pa, qa, pb, qb};
  double c0[6], r[6];
  if (!cost->Evaluate(blocks, c0, 0)) return false;                 // q_b = 0: r = -L_p p_hat
  double Lp[6][3], Y[6][4];
  for (int k = 0; k < 3; ++k) {                                      // translation columns of L
    pb[k] = 1.0;
    if (!cost->Evaluate(blocks, r, 0)) return false;
    for (int i = 0; i < 6; ++i) Lp[i][k] = r[i] - c0[i];
    pb[k] = 0.0;
  }
  for (int k = 0; k < 4; ++k) {                                      // Y = L_r E(q_hat)
    qb[k] = 1.0;
    if (!cost->Evaluate(blocks, r, 0)) return false;
    for (int i = 0; i < 6; ++i) Y[i][k] = r[i] - c0[i];
    qb[k] = 0.0;
  }
  // p_hat from the normal equations of L_p p = -c0
  double A[3][3], b[3];
  for (int i = 0; i < 3; ++i) {
    b[i] = 0;
    for (int j = 0; j < 3; ++j) { A[i][j] = 0; for (int k = 0; k < 6; ++k) A[i][j] += Lp[k][i] * Lp[k][j]; }
    for (int k = 0; k < 6; ++k) b[i] -= Lp[k][i] * c0[k];
  }
  const double detA = Det3(A);
  if (!(std::fabs(detA) > 0)) return false;
  for (int c = 0; c < 3; ++c) {
    double M[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = (j == c) ? b[i] : A[i][j];
    out->p[c] = Det3(M) / detA;
  }
  // q_hat = null vector of Y (e_q vanishes at q_b = q_hat): column of adj(Y'Y) with the largest norm
  double G[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { G[i][j] = 0; for (int k = 0; k < 6; ++k) G[i][j] += Y[k][i] * Y[k][j]; }
  double adj[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double m[3][3];
      int ri = 0;
      for (int a = 0; a < 4; ++a) {
        if (a == i) continue;
        int ci = 0;
        for (int c = 0; c < 4; ++c) { if (c == j) continue; m[ri][ci++] = G[a][c]; }
        ++ri;
      }
      adj[j][i] = (((i + j) & 1) ? -1.0 : 1.0) * Det3(m);
    }
  int best = 0;
  double bn = -1;
  for (int j = 0; j < 4; ++j) { double n = 0; for (int i = 0; i < 4; ++i) n += adj[i][j] * adj[i][j]; if (n > bn) { bn = n; best = j; } }
  if (!(bn > 0)) return false;
  const double inv = 1.0 / std::sqrt(bn);
  for (int i = 0; i < 4; ++i) out->q[i] = adj[i][best] * inv;
  // E = 2 * Lmat(q)[0:3,:] * diag(-1,-1,-1,1);  L_r = Y E^T / 4
  const double x = out->q[0], y = out->q[1], z = out->q[2], w = out->q[3];
  const double Lm[3][4] = {{w, -z, y, x}, {z, w, -x, y}, {-y, x, w, z}};
  double E[3][4];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) E[i][j] = 2.0 * Lm[i][j] * (j < 3 ? -1.0 : 1.0);
  for (int i = 0; i < 6; ++i) {
    for (int k = 0; k < 3; ++k) out->L[6 * i + k] = Lp[i][k];
    for (int k = 0; k < 3; ++k) { double s = 0; for (int j = 0; j < 4; ++j) s += Y[i][j] * E[k][j]; out->L[6 * i + 3 + k] = s / 4.0; }
  }
  // validate at pseudo-random (non-special) poses
  unsigned seed = 12345u;
  for (int trial = 0; trial < 3; ++trial) {
    double v[14];
    for (int i = 0; i < 14; ++i) { seed = seed * 1664525u + 1013904223u; v[i] = ((seed >> 8) & 0xffff) / 32768.0 - 1.0; }
    for (int o = 3; o <= 10; o += 7) { double n = std::sqrt(v[o] * v[o] + v[o + 1] * v[o + 1] + v[o + 2] * v[o + 2] + v[o + 3] * v[o + 3]); for (int i = 0; i < 4; ++i) v[o + i] /= n; }
    const double* bl[4] = {v, v + 3, v + 7, v + 10};
    double want[6], got[6];
    if (!cost->Evaluate(bl, want, 0)) return false;
    BetweenResidual(v, v + 3, v + 7, v + 10, out->p, out->q, out->L, got);
    double scale = 1.0, err = 0.0;
    for (int i = 0; i < 6; ++i) { scale = std::fmax(scale, std::fabs(want[i])); err = std::fmax(err, std::fabs(want[i] - got[i])); }
    if (!(err <= 1e-9 * scale)) return false;
  }
  return true;
}

inline void Fail(Solver::Summary* s, const std::string& why) {
  s->termination_type = FAILURE;
  s->message = why;
}

// ---- the MotionEstimate problem (REF/src/MotionEstimate.cc:71-129): residual blocks of
// AutoDiffCostFunction<ReprojectionError3Dto2D, 2, 4, 3, 3> over ONE quaternion block, ONE translation block and constant
// 3-D points.  The functor hides the pixel and reads the intrinsics from a config singleton (MotionEstimate.h:52-58): probing
// Evaluate recovers  r = (fx x/z + c0, fy y/z + c1)  with  c0 = cx - u, c1 = cy - v.
struct RecoveredReprojection { double fx, fy, c0, c1; };

inline void ReprojResidual(const RecoveredReprojection& k, const double* q, const double* t, const double* p, double* r) {
  const double u[3] = {q[0], q[1], q[2]}, w = q[3];
  const double uv[3] = {2 * (u[1] * p[2] - u[2] * p[1]), 2 * (u[2] * p[0] - u[0] * p[2]), 2 * (u[0] * p[1] - u[1] * p[0])};
  const double c[3] = {u[1] * uv[2] - u[2] * uv[1], u[2] * uv[0] - u[0] * uv[2], u[0] * uv[1] - u[1] * uv[0]};
  const double x = p[0] + w * uv[0] + c[0] + t[0], y = p[1] + w * uv[1] + c[1] + t[1], z = p[2] + w * uv[2] + c[2] + t[2];
  r[0] = (k.fx * x) / z + k.c0;
  r[1] = (k.fy * y) / z + k.c1;
}

inline bool RecoverReprojection(const CostFunction* cost, RecoveredReprojection* out) {
  const std::vector<int32>& sz = cost->parameter_block_sizes();
  if (cost->num_residuals() != 2 || sz.size() != 3 || sz[0] != 4 || sz[1] != 3 || sz[2] != 3) return false;
  const double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
  double p[3] = {0, 0, 1};
  const double* blocks[3] = {q, t, p};
  double r0[2], r[2];
  if (!cost->Evaluate(blocks, r0, 0)) return false;
  out->c0 = r0[0]; out->c1 = r0[1];
  p[0] = 1.0;
  if (!cost->Evaluate(blocks, r, 0)) return false;
  out->fx = r[0] - r0[0];
  if (std::fabs(r[1] - r0[1]) > 1e-9 * (1.0 + std::fabs(r0[1]))) return false;
  p[0] = 0.0; p[1] = 1.0;
  if (!cost->Evaluate(blocks, r, 0)) return false;
  out->fy = r[1] - r0[1];
  if (std::fabs(r[0] - r0[0]) > 1e-9 * (1.0 + std::fabs(r0[0]))) return false;
  unsigned seed = 54321u;
  for (int trial = 0; trial < 3; ++trial) {
    double v[10];
    for (int i = 0; i < 10; ++i) { seed = seed * 1664525u + 1013904223u; v[i] = ((seed >> 8) & 0xffff) / 32768.0 - 1.0; }
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    for (int i = 0; i < 4; ++i) v[i] /= n;
    v[9] = 6.0 + v[9];                                        // keep the point in front of the camera
    const double* bl[3] = {v, v + 4, v + 7};
    double want[2], got[2];
    if (!cost->Evaluate(bl, want, 0)) return false;
    ReprojResidual(*out, v, v + 4, v + 7, got);
    for (int i = 0; i < 2; ++i) if (!(std::fabs(want[i] - got[i]) <= 1e-9 * (1.0 + std::fabs(want[i])))) return false;
  }
  return true;
}

inline int LossKind(const LossFunction* loss, double* a) {
  *a = 1.0;
  if (!loss) return PGO_LOSS_TRIVIAL;
  if (const HuberLoss* h = dynamic_cast<const HuberLoss*>(loss)) { *a = h->a(); return PGO_LOSS_HUBER; }
  if (const SoftLOneLoss* h = dynamic_cast<const SoftLOneLoss*>(loss)) { *a = h->a(); return PGO_LOSS_SOFT_L_ONE; }
  if (const CauchyLoss* h = dynamic_cast<const CauchyLoss*>(loss)) { *a = h->a(); return PGO_LOSS_CAUCHY; }
  if (const ArctanLoss* h = dynamic_cast<const ArctanLoss*>(loss)) { *a = h->a(); return PGO_LOSS_ARCTAN; }
  if (const SwitchableConstraintLoss* h = dynamic_cast<const SwitchableConstraintLoss*>(loss)) { *a = h->a(); return PGO_LOSS_SWITCHABLE; }
  if (dynamic_cast<const TrivialLoss*>(loss)) return PGO_LOSS_TRIVIAL;
  return -1;
}

// ceres::Solve for the MotionEstimate problem -> pgo_reproj_solve_batch with one problem (the batched entry point is what a
// caller with many candidate pairs should use directly).
inline void SolveReprojection(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  const std::vector<ResidualBlock*>& rbs = problem->residual_blocks();
  double* q = rbs[0]->blocks[0];
  double* t = rbs[0]->blocks[1];
  const LossFunction* loss = rbs[0]->loss;
  std::vector<double> points(3 * rbs.size()), obs(2 * rbs.size());
  RecoveredReprojection k0 = {0, 0, 0, 0};
  for (size_t i = 0; i < rbs.size(); ++i) {
    const ResidualBlock* rb = rbs[i];
    RecoveredReprojection k;
    if (!RecoverReprojection(rb->cost, &k))
      return Fail(summary, "unsupported cost function: residual block is neither an SE(3) between factor nor a 3D-to-2D reprojection factor");
    if (rb->blocks[0] != q || rb->blocks[1] != t) return Fail(summary, "unsupported: reprojection blocks must share one quaternion and one translation block");
    if (rb->loss != loss) {   // one instance per AddResidualBlock is the usual Ceres pattern: equal kind and scale is what matters
      double a0 = 0.0, a1 = 0.0;
      const int k0 = LossKind(loss, &a0), k1 = LossKind(rb->loss, &a1);
      if (k0 < 0 || k0 != k1 || a0 != a1) return Fail(summary, "unsupported: residual blocks use LossFunctions of different kind or scale");
    }
    if (!problem->IsParameterBlockConstant(rb->blocks[2])) return Fail(summary, "unsupported: the 3-D points of the reprojection problem must be constant (MotionEstimate.cc:111-114)");
    if (i == 0) k0 = k;
    if (std::fabs(k.fx - k0.fx) > 1e-9 * std::fabs(k0.fx) || std::fabs(k.fy - k0.fy) > 1e-9 * std::fabs(k0.fy))
      return Fail(summary, "unsupported: reprojection blocks with different focal lengths");
    for (int c = 0; c < 3; ++c) points[3 * i + c] = rb->blocks[2][c];
    obs[2 * i] = -k.c0;      // with cx = cy = 0 the pixel is -(cx - u)
    obs[2 * i + 1] = -k.c1;
  }
  if (!dynamic_cast<const EigenQuaternionParameterization*>(problem->GetParameterization(q)))
    return Fail(summary, "unsupported: the quaternion block must use EigenQuaternionParameterization");
  if (problem->GetParameterization(t)) return Fail(summary, "unsupported: the translation block must be Euclidean");
  pgo_reproj_options o;
  pgo_reproj_options_init(&o);
  o.loss_kind = LossKind(loss, &o.loss_a);
  if (o.loss_kind < 0) return Fail(summary, "unsupported LossFunction (TrivialLoss, HuberLoss, SoftLOneLoss, CauchyLoss, ArctanLoss or NULL)");
  o.max_num_iterations = options.max_num_iterations;
  o.q_constant = problem->IsParameterBlockConstant(q) ? 1 : 0;
  o.t_constant = problem->IsParameterBlockConstant(t) ? 1 : 0;
  o.jacobi_scaling = options.jacobi_scaling ? 1 : 0;
  o.max_num_consecutive_invalid_steps = options.max_num_consecutive_invalid_steps;
  o.function_tolerance = options.function_tolerance; o.gradient_tolerance = options.gradient_tolerance;
  o.parameter_tolerance = options.parameter_tolerance;
  o.initial_trust_region_radius = options.initial_trust_region_radius; o.max_trust_region_radius = options.max_trust_region_radius;
  o.min_trust_region_radius = options.min_trust_region_radius; o.min_relative_decrease = options.min_relative_decrease;
  o.min_lm_diagonal = options.min_lm_diagonal; o.max_lm_diagonal = options.max_lm_diagonal;
  const long long ptr[2] = {0, (long long)rbs.size()};
  const double intr[4] = {k0.fx, k0.fy, 0.0, 0.0};
  pgo_reproj_summary rs;
  double ms = 0.0;
  if (pgo_reproj_solve_batch(1, ptr, &points[0], &obs[0], intr, q, t, &o, &rs, &ms) < 0) return Fail(summary, pgo_last_error());
  summary->termination_type = rs.termination_type == PGO_CONVERGENCE ? CONVERGENCE : rs.termination_type == PGO_NO_CONVERGENCE ? NO_CONVERGENCE : FAILURE;
  static const char* const kReason[] = {"", "Function tolerance reached.", "Parameter tolerance reached.", "Gradient tolerance reached.",
                                        "Minimum trust region radius reached.", "Maximum number of iterations reached.",
                                        "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps."};
  summary->message = kReason[rs.reason >= 1 && rs.reason <= 6 ? rs.reason : 0];
  summary->initial_cost = rs.initial_cost;
  summary->final_cost = rs.final_cost;
  summary->num_successful_steps = rs.num_successful_steps;
  summary->num_unsuccessful_steps = rs.num_unsuccessful_steps;
  summary->total_time_in_seconds = ms * 1e-3;
  summary->raw.termination_type = rs.termination_type; summary->raw.reason = rs.reason; summary->raw.num_iterations = rs.num_iterations;
  summary->raw.num_successful_steps = rs.num_successful_steps; summary->raw.num_unsuccessful_steps = rs.num_unsuccessful_steps;
  summary->raw.initial_cost = rs.initial_cost; summary->raw.final_cost = rs.final_cost; summary->raw.total_time_in_seconds = ms * 1e-3;
  summary->raw.num_edges = rs.num_points;
}

}  // namespace internal

inline void Solver::Solve(const Options& options, Problem* problem, Summary* summary) {
  *summary = Summary();
  std::string err;
  if (!options.IsValid(&err)) return internal::Fail(summary, err);
  summary->num_parameter_blocks = problem->NumParameterBlocks();
  summary->num_residual_blocks = problem->NumResidualBlocks();
  pgo_problem* P = pgo_problem_create();
  if (!P) return internal::Fail(summary, "pgo_problem_create failed");
  struct Guard { pgo_problem* p; ~Guard() { pgo_problem_destroy(p); } } guard = {P};

  const std::vector<internal::ResidualBlock*>& rbs = problem->residual_blocks();
  if (!rbs.empty() && rbs[0]->cost->num_residuals() == 2 && rbs[0]->blocks.size() == 3)
    return internal::SolveReprojection(options, problem, summary);      // MotionEstimate.cc problem
  const LossFunction* loss = rbs.empty() ? 0 : rbs[0]->loss;
  std::vector<int> ia(rbs.size()), ib(rbs.size());
  std::vector<double> t_be(7 * rbs.size()), sqrt_info(36 * rbs.size());
  for (size_t e = 0; e < rbs.size(); ++e) {
    const internal::ResidualBlock* rb = rbs[e];
    if (rb->loss != loss) {   // one instance per AddResidualBlock is the usual Ceres pattern: equal kind and scale is what matters
      double a0 = 0.0, a1 = 0.0;
      const int k0 = internal::LossKind(loss, &a0), k1 = internal::LossKind(rb->loss, &a1);
      if (k0 < 0 || k0 != k1 || a0 != a1)
        return internal::Fail(summary, "unsupported: residual blocks use LossFunctions of different kind or scale");
    }
    internal::RecoveredFactor f;
    if (!internal::RecoverBetweenFactor(rb->cost, &f))
      return internal::Fail(summary, "unsupported cost function: residual block is not an SE(3) between factor "
                                     "(6 residuals over blocks {3,4,3,4}); the GPU path has no generic-cost fallback");
    for (int side = 0; side < 2; ++side) {
      double* q = rb->blocks[2 * side + 1];
      if (!dynamic_cast<const EigenQuaternionParameterization*>(problem->GetParameterization(q)))
        return internal::Fail(summary, "unsupported: quaternion blocks must use EigenQuaternionParameterization");
      if (problem->GetParameterization(rb->blocks[2 * side]))
        return internal::Fail(summary, "unsupported: translation blocks must be Euclidean");
    }
    const int a = pgo_problem_add_pose(P, rb->blocks[0], rb->blocks[1]);
    const int b = pgo_problem_add_pose(P, rb->blocks[2], rb->blocks[3]);
    if (a < 0 || b < 0) return internal::Fail(summary, std::string("unsupported parameter-block pairing: ") + pgo_last_error());
    ia[e] = a; ib[e] = b;
    for (int i = 0; i < 3; ++i) t_be[7 * e + i] = f.p[i];
    for (int i = 0; i < 4; ++i) t_be[7 * e + 3 + i] = f.q[i];
    for (int i = 0; i < 36; ++i) sqrt_info[36 * e + i] = f.L[i];
  }
  // the reference always passes identity information (finial.cpp:217-218,276-277): recovered matrices within
  // rounding of I are snapped to it so that the identity fast path of the kernels is taken
  bool identity = true;
  for (size_t e = 0; e < rbs.size() && identity; ++e)
    for (int i = 0; i < 36; ++i) if (std::fabs(sqrt_info[36 * e + i] - ((i % 7 == 0) ? 1.0 : 0.0)) > 1e-13) { identity = false; break; }
  if (!rbs.empty() && pgo_problem_add_se3_between_batch(P, (int)rbs.size(), &ia[0], &ib[0], &t_be[0], identity ? 0 : &sqrt_info[0]) < 0)
    return internal::Fail(summary, pgo_last_error());
  if (loss) {
    int kind = -1;
    double a = 1.0;
    if (const HuberLoss* h = dynamic_cast<const HuberLoss*>(loss)) { kind = PGO_LOSS_HUBER; a = h->a(); }
    else if (const SoftLOneLoss* h = dynamic_cast<const SoftLOneLoss*>(loss)) { kind = PGO_LOSS_SOFT_L_ONE; a = h->a(); }
    else if (const CauchyLoss* h = dynamic_cast<const CauchyLoss*>(loss)) { kind = PGO_LOSS_CAUCHY; a = h->a(); }
    else if (const ArctanLoss* h = dynamic_cast<const ArctanLoss*>(loss)) { kind = PGO_LOSS_ARCTAN; a = h->a(); }
    else if (dynamic_cast<const TrivialLoss*>(loss)) kind = PGO_LOSS_TRIVIAL;
    if (kind < 0) return internal::Fail(summary, "unsupported LossFunction (TrivialLoss, HuberLoss, SoftLOneLoss, CauchyLoss, ArctanLoss or NULL)");
    if (kind != PGO_LOSS_TRIVIAL && pgo_problem_set_loss(P, kind, a) < 0) return internal::Fail(summary, pgo_last_error());
  }
  for (std::set<double*>::const_iterator it = problem->constant_blocks().begin(); it != problem->constant_blocks().end(); ++it)
    if (pgo_problem_set_parameter_block_constant(P, *it) < 0) { /* constant block that appears in no residual: nothing to do */ }

  if (pgo_version() != PGO_VERSION)    // the library writes sizeof(pgo_solver_summary) bytes of ITS header: a mismatch would overrun summary->raw
    return internal::Fail(summary, "libpgo_hip.so and include/pgo.h disagree on PGO_VERSION (rebuild against the header of the library in use)");
  pgo_solver_options o;
  pgo_solver_options_init(&o);
  o.max_num_iterations = options.max_num_iterations;
  o.linear_solver_type = (options.linear_solver_type == CGNR || options.linear_solver_type == ITERATIVE_SCHUR) ? PGO_BLOCK_JACOBI_PCG : PGO_SPARSE_NORMAL_CHOLESKY;
  o.jacobi_scaling = options.jacobi_scaling ? 1 : 0;
  o.max_linear_solver_iterations = options.max_linear_solver_iterations;
  o.min_linear_solver_iterations = options.min_linear_solver_iterations;
  o.max_num_consecutive_invalid_steps = options.max_num_consecutive_invalid_steps;
  o.function_tolerance = options.function_tolerance;
  o.gradient_tolerance = options.gradient_tolerance;
  o.parameter_tolerance = options.parameter_tolerance;
  o.initial_trust_region_radius = options.initial_trust_region_radius;
  o.max_trust_region_radius = options.max_trust_region_radius;
  o.min_trust_region_radius = options.min_trust_region_radius;
  o.min_relative_decrease = options.min_relative_decrease;
  o.min_lm_diagonal = options.min_lm_diagonal;
  o.max_lm_diagonal = options.max_lm_diagonal;
  o.eta = options.eta;
  o.pcg_cluster_poses = options.preconditioner_type == CLUSTER_JACOBI ? 4 : options.preconditioner_type == CLUSTER_TRIDIAGONAL ? 2 : 1;
  // CLUSTER_TRIDIAGONAL, the strongest preconditioner a Ceres caller can name: the library's strongest — 2-pose cluster Jacobi plus the
  // aggregation coarse level (include/pgo.h pcg_coarse_aggregate; one rank)
  if (o.linear_solver_type == PGO_BLOCK_JACOBI_PCG && options.preconditioner_type == CLUSTER_TRIDIAGONAL) o.pcg_coarse_aggregate = 64;

  std::vector<pgo_iteration_record> rec((size_t)options.max_num_iterations + 2 < 100000 ? options.max_num_iterations + 2 : 100000);
  if (pgo_solve(P, &o, &summary->raw, &rec[0], (int)rec.size()) < 0) return internal::Fail(summary, pgo_last_error());
  const pgo_solver_summary& r = summary->raw;
  summary->termination_type = r.termination_type == PGO_CONVERGENCE ? CONVERGENCE : r.termination_type == PGO_NO_CONVERGENCE ? NO_CONVERGENCE : FAILURE;
  summary->message = r.message;
  summary->initial_cost = r.initial_cost;
  summary->final_cost = r.final_cost;
  summary->num_successful_steps = r.num_successful_steps;
  summary->num_unsuccessful_steps = r.num_unsuccessful_steps;
  summary->total_time_in_seconds = r.total_time_in_seconds + r.setup_time_in_seconds;
  const int n = r.num_iterations < (int)rec.size() ? r.num_iterations : (int)rec.size();
  for (int i = 0; i < n; ++i) {
    IterationSummary it;
    it.iteration = rec[i].iteration; it.step_is_successful = rec[i].step_is_successful != 0; it.cost = rec[i].cost;
    it.cost_change = rec[i].cost_change; it.gradient_max_norm = rec[i].gradient_max_norm; it.step_norm = rec[i].step_norm;
    it.relative_decrease = rec[i].relative_decrease; it.trust_region_radius = rec[i].trust_region_radius;
    it.linear_solver_iterations = rec[i].linear_solver_iterations;
    summary->iterations.push_back(it);
    if (options.minimizer_progress_to_stdout)
      std::printf("%4d % .6e % .2e % .2e % .2e % .2e % .2e %6d\n", it.iteration, it.cost, it.cost_change, it.gradient_max_norm,
                  it.step_norm, it.relative_decrease, it.trust_region_radius, it.linear_solver_iterations);
  }
}

// Helper function which avoids going through the interface (finial.cpp:539).
inline void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  Solver solver;
  solver.Solve(options, problem, summary);
}

}  // namespace ceres
#endif
