// tools/ceres_baseline.cpp — CPU baseline against REAL Ceres (SURVEY.md §8d "CPU baseline, same run").
// Built only where Ceres + Eigen are installed (tools/Makefile target `ceres_baseline`, guarded by the presence of
// ceres/ceres.h); this image has neither, so the target reports "skipped" here and bench.py times the in-repo
// restatement instead.  Reads a g2o text file (VERTEX_SE3:QUAT / EDGE_SE3:QUAT, as datasets.write_g2o writes), builds the
// problem the way finial.cpp:491-528 does (one HuberLoss(1.0), one EigenQuaternionParameterization, first pose constant) and
// times ceres::Solve with the reference's options (finial.cpp:534-536) for num_threads = 1 and = hardware concurrency.
// usage: ceres_baseline graph.g2o [max_iterations]
#include <ceres/ceres.h>

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Vertex { Eigen::Vector3d p; Eigen::Quaterniond q; };
struct Constraint { int a, b; Eigen::Vector3d t; Eigen::Quaterniond r; Eigen::Matrix<double, 6, 6> sqrt_info; };

// residual of one relative-pose constraint: rotated translation error, then twice the vector part of the rotation error
struct BetweenError {
  BetweenError(const Constraint& c) : c_(c) {}
  template <typename T>
  bool operator()(const T* pa, const T* qa, const T* pb, const T* qb, T* res) const {
    Eigen::Map<const Eigen::Matrix<T, 3, 1>> ta(pa), tb(pb);
    Eigen::Map<const Eigen::Quaternion<T>> ra(qa), rb(qb);
    const Eigen::Quaternion<T> ra_inv = ra.conjugate();
    const Eigen::Quaternion<T> rel = ra_inv * rb;
    const Eigen::Matrix<T, 3, 1> trel = ra_inv * (tb - ta);
    const Eigen::Quaternion<T> dq = c_.r.template cast<T>() * rel.conjugate();
    Eigen::Map<Eigen::Matrix<T, 6, 1>> out(res);
    out.template head<3>() = trel - c_.t.template cast<T>();
    out.template tail<3>() = T(2.0) * dq.vec();
    out.applyOnTheLeft(c_.sqrt_info.template cast<T>());
    return true;
  }
  const Constraint c_;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s graph.g2o [max_iterations]\n", argv[0]); return 2; }
  const int max_it = argc > 2 ? std::atoi(argv[2]) : 1000;
  std::map<int, Vertex> verts0;
  std::vector<Constraint> cons;
  std::ifstream in(argv[1]);
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "VERTEX_SE3:QUAT") {
      int id; Vertex v; double x, y, z, w;
      ss >> id >> v.p[0] >> v.p[1] >> v.p[2] >> x >> y >> z >> w;
      v.q = Eigen::Quaterniond(w, x, y, z);
      verts0[id] = v;
    } else if (tag == "EDGE_SE3:QUAT") {
      Constraint c; double x, y, z, w;
      ss >> c.a >> c.b >> c.t[0] >> c.t[1] >> c.t[2] >> x >> y >> z >> w;
      c.r = Eigen::Quaterniond(w, x, y, z);
      Eigen::Matrix<double, 6, 6> info = Eigen::Matrix<double, 6, 6>::Zero();
      for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { ss >> info(i, j); info(j, i) = info(i, j); }
      c.sqrt_info = info.llt().matrixL();
      cons.push_back(c);
    }
  }
  for (int threads : {1, (int)std::thread::hardware_concurrency()}) {
    std::map<int, Vertex> verts = verts0;
    ceres::Problem problem;
    ceres::LossFunction* loss = new ceres::HuberLoss(1.0);
    ceres::LocalParameterization* quat = new ceres::EigenQuaternionParameterization;
    for (const Constraint& c : cons) {
      Vertex& va = verts[c.a];
      Vertex& vb = verts[c.b];
      problem.AddResidualBlock(new ceres::AutoDiffCostFunction<BetweenError, 6, 3, 4, 3, 4>(new BetweenError(c)), loss,
                               va.p.data(), va.q.coeffs().data(), vb.p.data(), vb.q.coeffs().data());
      problem.SetParameterization(va.q.coeffs().data(), quat);
      problem.SetParameterization(vb.q.coeffs().data(), quat);
    }
    problem.SetParameterBlockConstant(verts.begin()->second.p.data());
    problem.SetParameterBlockConstant(verts.begin()->second.q.coeffs().data());
    ceres::Solver::Options options;
    options.max_num_iterations = max_it;
    options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
    options.num_threads = threads;
    ceres::Solver::Summary summary;
    const auto t0 = std::chrono::steady_clock::now();
    ceres::Solve(options, &problem, &summary);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"ceres_version\": \"%s\", \"num_threads\": %d, \"seconds\": %.6f, \"iterations\": %d, \"initial_cost\": %.12e, \"final_cost\": %.12e, "
                "\"lm_iters_per_sec\": %.3f, \"edges\": %zu}\n", CERES_VERSION_STRING, threads, dt, (int)summary.iterations.size() - 1,
                summary.initial_cost, summary.final_cost, ((int)summary.iterations.size() - 1) / dt, cons.size());
  }
  return 0;
}
