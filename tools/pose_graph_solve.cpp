// tools/pose_graph_solve.cpp — the reference's BuildOptimizationProblem / SolveOptimizationProblem /
// OutputPoses flow (src/POSE_GRAPH_CERES_PLUS/test/pose_graph_ceres_plus_finial.cpp:491-567) written
// against the `namespace ceres` facade in include/ceres/, i.e. the same calls the reference makes, with
// the image front-end replaced by a g2o text file.  It demonstrates the drop-in: the only change a
// maintainer makes to the reference is -I<repo>/include and -lpgo_hip.
//
//   pose_graph_solve <in.g2o> <out_poses.txt> [max_iterations] [cgnr]
//
// The residual functor below restates PLUS/include/PoseGraph3dError.h:21-54 without Eigen (Eigen is not
// installed in the build image); where Eigen is available the reference's own header works unchanged.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <ceres/ceres.h>

namespace {

struct Pose3d { double p[3]; double q[4]; };                      // types.h:15-20 (q = x,y,z,w)
typedef std::map<int, Pose3d> MapOfPoses;                          // types.h:22-24
struct Edge3d { int id_begin, id_end; Pose3d t_be; double information[36]; };   // types.h:28-42
typedef std::vector<Edge3d> VectorOfEdges;

template <typename T> void QuatProduct(const T* a, const T* b, T* r) {
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

class PoseGraph3dErrorTerm {
 public:
  PoseGraph3dErrorTerm(const Pose3d& t_ab_measured, const double* sqrt_information) : t_ab_measured_(t_ab_measured) {
    for (int i = 0; i < 36; ++i) sqrt_information_[i] = sqrt_information[i];
  }
  template <typename T>
  bool operator()(const T* const p_a, const T* const q_a, const T* const p_b, const T* const q_b, T* residuals) const {
    const T qa_inv[4] = {-q_a[0], -q_a[1], -q_a[2], q_a[3]};
    T q_ab[4];
    QuatProduct(qa_inv, q_b, q_ab);
    const T d[3] = {p_b[0] - p_a[0], p_b[1] - p_a[1], p_b[2] - p_a[2]};
    // v + 2w(u x v) + 2 u x (u x v)
    const T u[3] = {qa_inv[0], qa_inv[1], qa_inv[2]};
    T uv[3] = {u[1] * d[2] - u[2] * d[1], u[2] * d[0] - u[0] * d[2], u[0] * d[1] - u[1] * d[0]};
    for (int i = 0; i < 3; ++i) uv[i] = uv[i] + uv[i];
    const T c[3] = {u[1] * uv[2] - u[2] * uv[1], u[2] * uv[0] - u[0] * uv[2], u[0] * uv[1] - u[1] * uv[0]};
    T e[6];
    for (int i = 0; i < 3; ++i) e[i] = d[i] + qa_inv[3] * uv[i] + c[i] - T(t_ab_measured_.p[i]);
    const T qm[4] = {T(t_ab_measured_.q[0]), T(t_ab_measured_.q[1]), T(t_ab_measured_.q[2]), T(t_ab_measured_.q[3])};
    const T q_ab_conj[4] = {-q_ab[0], -q_ab[1], -q_ab[2], q_ab[3]};
    T dq[4];
    QuatProduct(qm, q_ab_conj, dq);
    for (int i = 0; i < 3; ++i) e[3 + i] = T(2.0) * dq[i];
    for (int i = 0; i < 6; ++i) {
      T s = T(0.0);
      for (int j = 0; j < 6; ++j) s = s + T(sqrt_information_[6 * i + j]) * e[j];
      residuals[i] = s;
    }
    return true;
  }
  static ceres::CostFunction* Create(const Pose3d& t_ab_measured, const double* sqrt_information) {
    return new ceres::AutoDiffCostFunction<PoseGraph3dErrorTerm, 6, 3, 4, 3, 4>(
        new PoseGraph3dErrorTerm(t_ab_measured, sqrt_information));
  }
 private:
  const Pose3d t_ab_measured_;
  double sqrt_information_[36];
};

// information.llt().matrixL() (finial.cpp:508)
bool CholeskyLower6(const double* A, double* L) {
  for (int i = 0; i < 36; ++i) L[i] = 0;
  for (int j = 0; j < 6; ++j) {
    double d = A[7 * j];
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
    if (!(d > 0)) return false;
    L[7 * j] = std::sqrt(d);
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      L[6 * i + j] = s / L[7 * j];
    }
  }
  return true;
}

// finial.cpp:491-528
void BuildOptimizationProblem(const VectorOfEdges& Edges, MapOfPoses* poses, ceres::Problem* problem) {
  ceres::LossFunction* loss_function = new ceres::HuberLoss(1.0);
  ceres::LocalParameterization* quaternion_local_parameterization = new ceres::EigenQuaternionParameterization;
  for (VectorOfEdges::const_iterator it = Edges.begin(); it != Edges.end(); ++it) {
    const Edge3d& edge = *it;
    MapOfPoses::iterator pose_begin_iter = poses->find(edge.id_begin);
    MapOfPoses::iterator pose_end_iter = poses->find(edge.id_end);
    double sqrt_information[36];
    if (!CholeskyLower6(edge.information, sqrt_information)) { std::cerr << "information not SPD\n"; std::exit(2); }
    ceres::CostFunction* cost_function = PoseGraph3dErrorTerm::Create(edge.t_be, sqrt_information);
    problem->AddResidualBlock(cost_function, loss_function, pose_begin_iter->second.p, pose_begin_iter->second.q,
                              pose_end_iter->second.p, pose_end_iter->second.q);
    problem->SetParameterization(pose_begin_iter->second.q, quaternion_local_parameterization);
    problem->SetParameterization(pose_end_iter->second.q, quaternion_local_parameterization);
  }
  MapOfPoses::iterator pose_start_iter = poses->begin();
  problem->SetParameterBlockConstant(pose_start_iter->second.p);
  problem->SetParameterBlockConstant(pose_start_iter->second.q);
}

// finial.cpp:531-544
bool SolveOptimizationProblem(ceres::Problem* problem, int max_iterations, bool cgnr) {
  ceres::Solver::Options options;
  options.max_num_iterations = max_iterations;
  options.linear_solver_type = cgnr ? ceres::CGNR : ceres::SPARSE_NORMAL_CHOLESKY;
  ceres::Solver::Summary summary;
  ceres::Solve(options, problem, &summary);
  std::cout << summary.FullReport() << '\n';
  return summary.IsSolutionUsable();
}

// finial.cpp:547-567 (Eigen prints p.transpose() with its coefficients right-aligned to a common width)
bool OutputPoses(const std::string& filename, const MapOfPoses& poses) {
  std::fstream outfile;
  outfile.open(filename.c_str(), std::istream::out);
  if (!outfile) { std::cout << "Error opening the file: " << filename; return false; }
  for (MapOfPoses::const_iterator it = poses.begin(); it != poses.end(); ++it) {
    std::string s[3];
    size_t w = 0;
    for (int i = 0; i < 3; ++i) { std::ostringstream o; o << it->second.p[i]; s[i] = o.str(); if (s[i].size() > w) w = s[i].size(); }
    outfile << it->first << " ";
    for (int i = 0; i < 3; ++i) outfile << std::string(w - s[i].size(), ' ') << s[i] << (i < 2 ? " " : "");
    outfile << " " << it->second.q[0] << " " << it->second.q[1] << " " << it->second.q[2] << " " << it->second.q[3] << '\n';
  }
  return true;
}

bool ReadG2o(const std::string& path, MapOfPoses* poses, VectorOfEdges* edges) {
  std::ifstream in(path.c_str());
  if (!in) return false;
  std::string line, tag;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    if (!(ss >> tag)) continue;
    if (tag == "VERTEX_SE3:QUAT") {
      int id; Pose3d p;
      ss >> id >> p.p[0] >> p.p[1] >> p.p[2] >> p.q[0] >> p.q[1] >> p.q[2] >> p.q[3];
      (*poses)[id] = p;
    } else if (tag == "EDGE_SE3:QUAT") {
      Edge3d e;
      ss >> e.id_begin >> e.id_end >> e.t_be.p[0] >> e.t_be.p[1] >> e.t_be.p[2] >> e.t_be.q[0] >> e.t_be.q[1] >> e.t_be.q[2] >> e.t_be.q[3];
      for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { ss >> e.information[6 * i + j]; e.information[6 * j + i] = e.information[6 * i + j]; }
      edges->push_back(e);
    }
  }
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::cerr << "usage: pose_graph_solve <in.g2o> <out_poses.txt> [max_iterations] [cgnr]\n"; return 2; }
  MapOfPoses poses;
  VectorOfEdges Edges;
  if (!ReadG2o(argv[1], &poses, &Edges)) { std::cerr << "cannot read " << argv[1] << "\n"; return 2; }
  const int max_it = argc > 3 ? std::atoi(argv[3]) : 1000;   // finial.cpp:535
  const bool cgnr = argc > 4 && std::string(argv[4]) == "cgnr";
  std::cout << "Number of poses: " << poses.size() << "\nNumber of edges: " << Edges.size() << '\n';
  ceres::Problem problem;
  BuildOptimizationProblem(Edges, &poses, &problem);
  const bool ok = SolveOptimizationProblem(&problem, max_it, cgnr);
  if (!ok) std::cout << "The solve was not successful, exiting.\n";
  OutputPoses(argv[2], poses);
  return ok ? 0 : 1;
}
