// tools/facade_selftest.cpp — host-only checks of the `namespace ceres` facade (run by tests/test_facade.py):
// factor recovery by probing, AutoDiff Jacobians, ownership (each object deleted exactly once), and the loud
// failure of ceres::Solve without a GPU.  Prints "OK <name>" lines; exits non-zero on the first failure.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ceres/ceres.h>

static int g_cost_deleted = 0, g_loss_deleted = 0, g_lp_deleted = 0;
#define CHECK_OR_DIE(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static unsigned g_seed = 7u;
static double rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffffff) / double(0x800000) - 1.0; }

struct Term {  // same residual as the reference functor, parameters kept private like PoseGraph3dErrorTerm
  double p[3], q[4], L[36];
  template <typename T> bool operator()(const T* const pa, const T* const qa, const T* const pb, const T* const qb, T* r) const {
    const T u[3] = {-qa[0], -qa[1], -qa[2]}, w = qa[3], d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    T uv[3] = {u[1] * d[2] - u[2] * d[1], u[2] * d[0] - u[0] * d[2], u[0] * d[1] - u[1] * d[0]};
    for (int i = 0; i < 3; ++i) uv[i] = uv[i] + uv[i];
    const T c[3] = {u[1] * uv[2] - u[2] * uv[1], u[2] * uv[0] - u[0] * uv[2], u[0] * uv[1] - u[1] * uv[0]};
    T e[6];
    for (int i = 0; i < 3; ++i) e[i] = d[i] + w * uv[i] + c[i] - T(p[i]);
    // q_hat * conj(conj(q_a) * q_b)
    const T ai[4] = {-qa[0], -qa[1], -qa[2], qa[3]};
    T ab[4];
    ab[3] = ai[3] * qb[3] - ai[0] * qb[0] - ai[1] * qb[1] - ai[2] * qb[2];
    ab[0] = ai[3] * qb[0] + ai[0] * qb[3] + ai[1] * qb[2] - ai[2] * qb[1];
    ab[1] = ai[3] * qb[1] + ai[1] * qb[3] + ai[2] * qb[0] - ai[0] * qb[2];
    ab[2] = ai[3] * qb[2] + ai[2] * qb[3] + ai[0] * qb[1] - ai[1] * qb[0];
    const T cj[4] = {-ab[0], -ab[1], -ab[2], ab[3]};
    const T m[4] = {T(q[0]), T(q[1]), T(q[2]), T(q[3])};
    e[3] = T(2.0) * (m[3] * cj[0] + m[0] * cj[3] + m[1] * cj[2] - m[2] * cj[1]);
    e[4] = T(2.0) * (m[3] * cj[1] + m[1] * cj[3] + m[2] * cj[0] - m[0] * cj[2]);
    e[5] = T(2.0) * (m[3] * cj[2] + m[2] * cj[3] + m[0] * cj[1] - m[1] * cj[0]);
    for (int i = 0; i < 6; ++i) { T s = T(0.0); for (int j = 0; j < 6; ++j) s = s + T(L[6 * i + j]) * e[j]; r[i] = s; }
    return true;
  }
};
struct CountingCost : ceres::AutoDiffCostFunction<Term, 6, 3, 4, 3, 4> {
  explicit CountingCost(Term* t) : ceres::AutoDiffCostFunction<Term, 6, 3, 4, 3, 4>(t) {}
  ~CountingCost() { ++g_cost_deleted; }
};
struct CountingLoss : ceres::HuberLoss { CountingLoss() : ceres::HuberLoss(1.0) {} ~CountingLoss() { ++g_loss_deleted; } };
struct CountingLp : ceres::EigenQuaternionParameterization { ~CountingLp() { ++g_lp_deleted; } };
struct NotBetween { template <typename T> bool operator()(const T* const a, const T* const, const T* const, const T* const, T* r) const {
  for (int i = 0; i < 6; ++i) r[i] = a[0] * a[0] + T(double(i)); return true; } };

// the MotionEstimate functor (REF/include/MotionEstimate.h:34-91): intrinsics and pixel hidden inside, Eigen's q*p + t
struct Reproj {
  double fx, fy, cx, cy, u, v;
  template <typename T> bool operator()(const T* const q, const T* const t, const T* const p, T* r) const {
    const T uu[3] = {q[0], q[1], q[2]}, w = q[3];
    T uv[3] = {uu[1] * p[2] - uu[2] * p[1], uu[2] * p[0] - uu[0] * p[2], uu[0] * p[1] - uu[1] * p[0]};
    for (int i = 0; i < 3; ++i) uv[i] = uv[i] + uv[i];
    const T c[3] = {uu[1] * uv[2] - uu[2] * uv[1], uu[2] * uv[0] - uu[0] * uv[2], uu[0] * uv[1] - uu[1] * uv[0]};
    const T x = p[0] + w * uv[0] + c[0] + t[0], y = p[1] + w * uv[1] + c[1] + t[1], z = p[2] + w * uv[2] + c[2] + t[2];
    r[0] = (T(fx) * x) / z + T(cx) - T(u);
    r[1] = (T(fy) * y) / z + T(cy) - T(v);
    return true;
  }
};

static Term* RandomTerm(bool identity) {
  Term* t = new Term;
  double n = 0;
  for (int i = 0; i < 3; ++i) t->p[i] = 2 * rnd();
  for (int i = 0; i < 4; ++i) { t->q[i] = rnd(); n += t->q[i] * t->q[i]; }
  for (int i = 0; i < 4; ++i) t->q[i] /= std::sqrt(n);
  for (int i = 0; i < 36; ++i) t->L[i] = identity ? (i % 7 == 0 ? 1.0 : 0.0) : ((i / 6 >= i % 6) ? 0.5 * rnd() + (i % 7 == 0 ? 2.0 : 0.0) : 0.0);
  return t;
}

int main() {
  // ---- recovery of (p_hat, q_hat, L) from an opaque cost function ----
  for (int trial = 0; trial < 50; ++trial) {
    Term* t = RandomTerm(trial % 5 == 0);
    ceres::AutoDiffCostFunction<Term, 6, 3, 4, 3, 4> cost(t);
    ceres::internal::RecoveredFactor f;
    CHECK_OR_DIE(ceres::internal::RecoverBetweenFactor(&cost, &f));
    for (int k = 0; k < 5; ++k) {
      double v[14], want[6], got[6];
      for (int i = 0; i < 14; ++i) v[i] = rnd();
      const double* b[4] = {v, v + 3, v + 7, v + 10};
      cost.Evaluate(b, want, 0);
      ceres::internal::BetweenResidual(v, v + 3, v + 7, v + 10, f.p, f.q, f.L, got);
      for (int i = 0; i < 6; ++i) CHECK_OR_DIE(std::fabs(want[i] - got[i]) < 1e-10 * (1 + std::fabs(want[i])));
    }
    for (int i = 0; i < 3; ++i) CHECK_OR_DIE(std::fabs(f.p[i] - t->p[i]) < 1e-10);
    const double sgn = (f.q[0] * t->q[0] + f.q[1] * t->q[1] + f.q[2] * t->q[2] + f.q[3] * t->q[3]) > 0 ? 1.0 : -1.0;
    for (int i = 0; i < 4; ++i) CHECK_OR_DIE(std::fabs(sgn * f.q[i] - t->q[i]) < 1e-10);
  }
  {
    ceres::AutoDiffCostFunction<NotBetween, 6, 3, 4, 3, 4> bad(new NotBetween);
    ceres::internal::RecoveredFactor f;
    CHECK_OR_DIE(!ceres::internal::RecoverBetweenFactor(&bad, &f));
  }
  std::printf("OK recover\n");

  // ---- AutoDiff Jacobians against central differences ----
  {
    Term* t = RandomTerm(false);
    ceres::AutoDiffCostFunction<Term, 6, 3, 4, 3, 4> cost(t);
    double v[14], r[6], J0[18], J1[24], J2[18], J3[24];
    for (int i = 0; i < 14; ++i) v[i] = rnd();
    const double* b[4] = {v, v + 3, v + 7, v + 10};
    double* J[4] = {J0, J1, J2, J3};
    CHECK_OR_DIE(cost.Evaluate(b, r, J));
    const int off[4] = {0, 3, 7, 10}, sz[4] = {3, 4, 3, 4};
    for (int blk = 0; blk < 4; ++blk)
      for (int c = 0; c < sz[blk]; ++c) {
        double rp[6], rm[6];
        const double h = 1e-6, keep = v[off[blk] + c];
        v[off[blk] + c] = keep + h; cost.Evaluate(b, rp, 0);
        v[off[blk] + c] = keep - h; cost.Evaluate(b, rm, 0);
        v[off[blk] + c] = keep;
        for (int i = 0; i < 6; ++i) CHECK_OR_DIE(std::fabs((rp[i] - rm[i]) / (2 * h) - J[blk][i * sz[blk] + c]) < 1e-6);
      }
  }
  std::printf("OK autodiff\n");

  // ---- EigenQuaternionParameterization ----
  {
    ceres::EigenQuaternionParameterization lp;
    double q[4] = {0.1, -0.2, 0.3, 0.9}, n = std::sqrt(0.01 + 0.04 + 0.09 + 0.81), d[3] = {0.02, -0.01, 0.03}, out[4], J[12];
    for (int i = 0; i < 4; ++i) q[i] /= n;
    lp.Plus(q, d, out);
    CHECK_OR_DIE(std::fabs(out[0] * out[0] + out[1] * out[1] + out[2] * out[2] + out[3] * out[3] - 1) < 1e-14);
    lp.ComputeJacobian(q, J);
    for (int c = 0; c < 3; ++c) {
      double dp[3] = {0, 0, 0}, op[4], om[4];
      dp[c] = 1e-6; lp.Plus(q, dp, op); dp[c] = -1e-6; lp.Plus(q, dp, om);
      for (int i = 0; i < 4; ++i) CHECK_OR_DIE(std::fabs((op[i] - om[i]) / 2e-6 - J[3 * i + c]) < 1e-8);
    }
    const double zero[3] = {0, 0, 0};
    lp.Plus(q, zero, out);
    for (int i = 0; i < 4; ++i) CHECK_OR_DIE(out[i] == q[i]);
    CHECK_OR_DIE(lp.GlobalSize() == 4 && lp.LocalSize() == 3);
  }
  std::printf("OK parameterization\n");

  // ---- Problem bookkeeping, ownership, and Solve without a GPU ----
  {
    double poses[3][7] = {{0, 0, 0, 0, 0, 0, 1}, {1, 0, 0, 0, 0, 0, 1}, {2, 0, 0, 0, 0, 0, 1}};
    ceres::Solver::Summary summary;
    {
      ceres::Problem problem;
      ceres::LossFunction* loss = new CountingLoss;
      ceres::LocalParameterization* lp = new CountingLp;
      for (int e = 0; e < 2; ++e) {
        ceres::CostFunction* c = new CountingCost(RandomTerm(true));
        problem.AddResidualBlock(c, loss, poses[e + 1], poses[e + 1] + 3, poses[e], poses[e] + 3);
        problem.SetParameterization(poses[e + 1] + 3, lp);   // same pointer set repeatedly: legal (finial.cpp:519-522)
        problem.SetParameterization(poses[e] + 3, lp);
      }
      problem.SetParameterBlockConstant(poses[0]);
      problem.SetParameterBlockConstant(poses[0] + 3);
      CHECK_OR_DIE(problem.NumParameterBlocks() == 6 && problem.NumResidualBlocks() == 2);
      CHECK_OR_DIE(problem.NumParameters() == 21 && problem.NumResiduals() == 12);
      CHECK_OR_DIE(problem.IsParameterBlockConstant(poses[0]) && !problem.IsParameterBlockConstant(poses[1]));
      ceres::Solver::Options options;
      options.max_num_iterations = 1000;
      options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
      CHECK_OR_DIE(ceres::SPARSE_NORMAL_CHOLESKY == 2);
      ceres::Solve(options, &problem, &summary);
      if (pgo_device_count() == 0) {
        CHECK_OR_DIE(summary.termination_type == ceres::FAILURE && !summary.IsSolutionUsable());
        CHECK_OR_DIE(summary.message.find("no CPU fallback") != std::string::npos);
        CHECK_OR_DIE(poses[1][0] == 1.0 && poses[2][0] == 2.0);   // parameters untouched on failure
      } else {
        CHECK_OR_DIE(summary.IsSolutionUsable());
      }
      CHECK_OR_DIE(!summary.FullReport().empty() && !summary.BriefReport().empty());
    }
    CHECK_OR_DIE(g_cost_deleted == 2 && g_loss_deleted == 1 && g_lp_deleted == 1);
  }
  std::printf("OK problem\n");

  // ---- one LossFunction instance per residual block (the usual Ceres pattern): accepted when kind and scale agree,
  // refused when they differ ----
  {
    for (int differ = 0; differ < 2; ++differ) {
      double poses[3][7] = {{0, 0, 0, 0, 0, 0, 1}, {1, 0, 0, 0, 0, 0, 1}, {2, 0, 0, 0, 0, 0, 1}};
      ceres::Problem problem;
      ceres::LocalParameterization* lp = new ceres::EigenQuaternionParameterization;
      for (int e = 0; e < 2; ++e) {
        ceres::LossFunction* loss = new ceres::HuberLoss(differ && e == 1 ? 2.0 : 1.0);
        problem.AddResidualBlock(new ceres::AutoDiffCostFunction<Term, 6, 3, 4, 3, 4>(RandomTerm(true)), loss, poses[e + 1],
                                 poses[e + 1] + 3, poses[e], poses[e] + 3);
        problem.SetParameterization(poses[e + 1] + 3, lp);
        problem.SetParameterization(poses[e] + 3, lp);
      }
      problem.SetParameterBlockConstant(poses[0]);
      problem.SetParameterBlockConstant(poses[0] + 3);
      ceres::Solver::Options options;
      ceres::Solver::Summary summary;
      ceres::Solve(options, &problem, &summary);
      if (differ) CHECK_OR_DIE(summary.termination_type == ceres::FAILURE && summary.message.find("different kind or scale") != std::string::npos);
      else if (pgo_device_count() > 0) CHECK_OR_DIE(summary.IsSolutionUsable());
    }
  }
  std::printf("OK per-block loss instances\n");

  // ---- the MotionEstimate problem (MotionEstimate.cc:71-129) through ceres::Problem / ceres::Solve ----
  {
    const double fx = 718.856, fy = 718.856, cx = 607.1928, cy = 185.2157;
    const double t_true[3] = {0.3, -0.1, 0.9};
    const int n = 120;
    static double pts[120][3];
    double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
    ceres::Problem problem;
    ceres::LossFunction* loss = new ceres::HuberLoss(1.0);
    ceres::LocalParameterization* lp = new ceres::EigenQuaternionParameterization;
    ceres::CostFunction* first = 0;
    for (int i = 0; i < n; ++i) {
      pts[i][0] = 10 * rnd(); pts[i][1] = 3 * rnd(); pts[i][2] = 20 + 15 * rnd();
      const double X = pts[i][0] + t_true[0], Y = pts[i][1] + t_true[1], Z = pts[i][2] + t_true[2];
      Reproj* f = new Reproj;
      f->fx = fx; f->fy = fy; f->cx = cx; f->cy = cy;
      f->u = fx * X / Z + cx + 0.3 * rnd(); f->v = fy * Y / Z + cy + 0.3 * rnd();
      ceres::CostFunction* c = new ceres::AutoDiffCostFunction<Reproj, 2, 4, 3, 3>(f);
      if (!first) first = c;
      problem.AddResidualBlock(c, loss, q, t, pts[i]);
      problem.SetParameterization(q, lp);
    }
    problem.SetParameterBlockConstant(q);
    for (int i = 0; i < n; ++i) problem.SetParameterBlockConstant(pts[i]);
    ceres::internal::RecoveredReprojection k;
    CHECK_OR_DIE(ceres::internal::RecoverReprojection(first, &k));
    CHECK_OR_DIE(std::fabs(k.fx - fx) < 1e-9 && std::fabs(k.fy - fy) < 1e-9);
    ceres::Solver::Options options;
    options.max_num_iterations = 1000;
    options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
    if (pgo_device_count() == 0) {
      CHECK_OR_DIE(summary.termination_type == ceres::FAILURE && summary.message.find("no CPU fallback") != std::string::npos);
      CHECK_OR_DIE(t[0] == 0.0 && t[1] == 0.0 && t[2] == 0.0);
    } else {
      CHECK_OR_DIE(summary.termination_type == ceres::CONVERGENCE && summary.final_cost < summary.initial_cost);
      for (int i = 0; i < 3; ++i) CHECK_OR_DIE(std::fabs(t[i] - t_true[i]) < 0.05);
      CHECK_OR_DIE(q[0] == 0.0 && q[3] == 1.0);
      std::printf("   t = %.4f %.4f %.4f  cost %.3f -> %.3f\n", t[0], t[1], t[2], summary.initial_cost, summary.final_cost);
    }
  }
  std::printf("OK motion_estimate\n");
  return 0;
}
