// tools/reference_tu_smoke.cpp — compile (and, on a GPU box, run) the REFERENCE's own residual header against the
// `namespace ceres` facade of include/ceres/: SURVEY.md §7.2 #3.  PoseGraph3dError.h instantiates
// Eigen::Quaternion<ceres::Jet<double, 14>> (PoseGraph3dError.h:21-54), which needs the Eigen::NumTraits specialisation in
// include/ceres/jet.h — this target is where that meets a compiler.  The reference headers are NOT copied: they are
// included from where they lie (-I$(REF_INC) in tools/Makefile).  Needs Eigen 3; this image has none, so
// `make -C tools reference_tu_smoke` prints SKIPPED here.
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <cstdio>

#include "ceres/ceres.h"
#include "PoseGraph3dError.h"   // the reference's header, unchanged (includes its types.h)

using namespace POSE_GRAPH;

int main() {
  // finial.cpp:491-528 in miniature: three poses on a line, two odometry edges and one chord
  MapOfPoses poses;
  for (int i = 0; i < 3; ++i) {
    Pose3d p;
    p.p = Eigen::Vector3d(1.1 * i, 0.05 * i, 0.0);
    p.q = Eigen::Quaterniond::Identity();
    poses[i] = p;
  }
  VectorOfEdges edges;
  const int pairs[3][2] = {{0, 1}, {1, 2}, {0, 2}};
  for (const auto& pr : pairs) {
    Edge3d e;
    e.id_begin = pr[0];
    e.id_end = pr[1];
    e.t_be.p = Eigen::Vector3d(1.0 * (pr[1] - pr[0]), 0.0, 0.0);
    e.t_be.q = Eigen::Quaterniond::Identity();
    e.information = Eigen::Matrix<double, 6, 6>::Identity();
    edges.push_back(e);
  }
  ceres::Problem problem;
  ceres::LossFunction* loss_function = new ceres::HuberLoss(1.0);
  ceres::LocalParameterization* quaternion_local_parameterization = new ceres::EigenQuaternionParameterization;
  for (const Edge3d& constraint : edges) {
    Pose3d& a = poses[constraint.id_begin];
    Pose3d& b = poses[constraint.id_end];
    const Eigen::Matrix<double, 6, 6> sqrt_information = constraint.information.llt().matrixL();
    ceres::CostFunction* cost_function = PoseGraph3dErrorTerm::Create(constraint.t_be, sqrt_information);
    problem.AddResidualBlock(cost_function, loss_function, a.p.data(), a.q.coeffs().data(), b.p.data(), b.q.coeffs().data());
    problem.SetParameterization(a.q.coeffs().data(), quaternion_local_parameterization);
    problem.SetParameterization(b.q.coeffs().data(), quaternion_local_parameterization);
  }
  problem.SetParameterBlockConstant(poses.begin()->second.p.data());
  problem.SetParameterBlockConstant(poses.begin()->second.q.coeffs().data());
  ceres::Solver::Options options;
  options.max_num_iterations = 50;
  options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);
  std::printf("%s\nusable %d\n", summary.FullReport().c_str(), summary.IsSolutionUsable() ? 1 : 0);
  return summary.IsSolutionUsable() && poses[2].p.x() > 1.9 && poses[2].p.x() < 2.1 ? 0 : 1;
}
