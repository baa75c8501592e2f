// =============================================================================================
// oracle/pgo_oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (plain C++17, no third-party code) of the pose-graph NLLS hot path of
// TurtleZhong/PoseGraph-Ceres.  It is the *checker* for the HIP path: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
// (posegraph-ceres_amd/csrc) never includes, links or calls anything in this directory.
//
// PARITY UNPINNED: the reference holds no golden vector / known-answer test for this path and its
// own implementation (Ceres 1.13.0 + Eigen + CHOLMOD, un-vendored) cannot be built in this image
// (SURVEY.md §8c).  This oracle is therefore pinned only by (a) an independent numpy/scipy
// restatement with finite differences (tests/test_oracle_numpy.py) and (b) the committed fixtures
// under tests/golden/ that this file generated.
//
// What each part follows (REF = /root/reference/src/POSE_GRAPH_CERES_PLUS):
//   * residual functor ............ REF/include/PoseGraph3dError.h:21-54 (templated operator())
//   * AutoDiff<6,3,4,3,4> ......... REF/include/PoseGraph3dError.h:56-61 ; Jet<double,14>  [Ceres 1.13]
//   * quaternion Plus + 4x3 Jac ... src/other_projects/bundle_adjustment/ceres_extensions.h:25-50
//   * quaternion product .......... ceres_extensions.h:144-150 (== Eigen quat_product)
//   * problem construction ........ REF/test/pose_graph_ceres_plus_finial.cpp:491-528
//                                   (HuberLoss(1.0), L = information.llt().matrixL(), pose 0 const)
//   * solver options .............. REF/test/pose_graph_ceres_plus_finial.cpp:531-544 + the Ceres
//                                   1.13 defaults recovered from the binary
//   * LM loop / Huber corrector / Jacobi scaling / termination: SURVEY.md Appendix A.4-A.6
//     [Ceres 1.13 TrustRegionMinimizer + LevenbergMarquardtStrategy + ConjugateGradientsSolver]
// =============================================================================================
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <queue>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

namespace {

// ---- threads (the CPU baseline "single-thread and all-cores") ----------------------------------------------------
// A fixed pool; run(n, fn) calls fn(i) for i in [0, n) on the pool's threads (dynamic pick) and returns when all are done.  Every
// threaded routine below produces the bits of its sequential form: work is split by OWNER (a block is only ever added to by one
// thread, in edge order) or by independent subtrees of the elimination tree, never by a reduction whose order depends on timing.
class Pool {
 public:
  explicit Pool(int nthreads) : n_(std::max(1, nthreads)) {
    for (int t = 1; t < n_; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int width() const { return n_; }
  void run(int n, const std::function<void(int)>& fn) {
    if (n_ == 1 || n <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    { std::lock_guard<std::mutex> lk(mu_); fn_ = &fn; total_ = n; next_.store(0); pending_ = (int)workers_.size(); ++gen_; }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  void work() { for (int i; (i = next_.fetch_add(1)) < total_;) (*fn_)(i); }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
      work();
      { std::lock_guard<std::mutex> lk(mu_); if (--pending_ == 0) done_cv_.notify_one(); }
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int total_ = 0, pending_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// ---------------------------------------------------------------------------------------------
// Forward-mode dual number, the role Jet<double,14> plays in Ceres' AutoDiffCostFunction.
// ---------------------------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1.0; }
};
template <int N> Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> Jet<N> operator*(double s, const Jet<N>& g) { Jet<N> h; h.a = s * g.a; for (int i = 0; i < N; ++i) h.v[i] = s * g.v[i]; return h; }
template <int N> Jet<N> operator*(const Jet<N>& g, double s) { return s * g; }
template <int N> Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }

// ---------------------------------------------------------------------------------------------
// Eigen-equivalent quaternion helpers, coefficient order x,y,z,w (Eigen coeffs()).
// ---------------------------------------------------------------------------------------------
template <class T> struct Q4 { T x, y, z, w; };
template <class T> struct V3 { T x, y, z; };

template <class T> Q4<T> qconj(const Q4<T>& q) { return Q4<T>{-q.x, -q.y, -q.z, q.w}; }
// Hamilton product a*b (Eigen quat_product / ceres_extensions.h:144-150).
template <class T> Q4<T> qmul(const Q4<T>& a, const Q4<T>& b) {
  Q4<T> r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
template <class T> V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return V3<T>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen QuaternionBase::_transformVector: v + 2w(u x v) + 2 u x (u x v)   (no normalisation)
template <class T> V3<T> qrot(const Q4<T>& q, const V3<T>& v) {
  V3<T> u{q.x, q.y, q.z};
  V3<T> uv = cross(u, v);
  uv = V3<T>{uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
  V3<T> c = cross(u, uv);
  return V3<T>{v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}

// ---------------------------------------------------------------------------------------------
// The residual functor.  REF/include/PoseGraph3dError.h:21-54, statement by statement.
//   L is the 6x6 "sqrt_information" (row-major), applied on the left: r = L * e   (:51)
// ---------------------------------------------------------------------------------------------
template <class T>
void functor(const double* mp, const double* mq, const double* L, const T* p_a, const T* q_a,
             const T* p_b, const T* q_b, T* res) {
  Q4<T> qa{q_a[0], q_a[1], q_a[2], q_a[3]};
  Q4<T> qb{q_b[0], q_b[1], q_b[2], q_b[3]};
  Q4<T> qa_inv = qconj(qa);                                       // :32
  Q4<T> q_ab = qmul(qa_inv, qb);                                  // :33
  V3<T> d{p_b[0] - p_a[0], p_b[1] - p_a[1], p_b[2] - p_a[2]};
  V3<T> p_ab = qrot(qa_inv, d);                                   // :36
  Q4<T> qm{T(mq[0]), T(mq[1]), T(mq[2]), T(mq[3])};
  Q4<T> dq = qmul(qm, qconj(q_ab));                               // :39-40
  T e[6] = {p_ab.x - T(mp[0]), p_ab.y - T(mp[1]), p_ab.z - T(mp[2]),   // :45-46
            T(2.0) * dq.x, T(2.0) * dq.y, T(2.0) * dq.z};              // :47-48
  for (int i = 0; i < 6; ++i) {                                   // :51 applyOnTheLeft(L)
    T s = T(0.0);
    for (int j = 0; j < 6; ++j) s = s + T(L[6 * i + j]) * e[j];
    res[i] = s;
  }
}

// EigenQuaternionParameterization::ComputeJacobian, ceres_extensions.h:44-50 (4x3 row-major).
void quat_plus_jacobian(const double* x, double* J) {
  J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
  J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
  J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
  J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
}
// EigenQuaternionParameterization::Plus, ceres_extensions.h:25-42.
void quat_plus(const double* x, const double* delta, double* out) {
  const double n = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (n > 0.0) {
    const double s = std::sin(n) / n;
    Q4<double> t{s * delta[0], s * delta[1], s * delta[2], std::cos(n)};
    Q4<double> q{x[0], x[1], x[2], x[3]};
    Q4<double> r = qmul(t, q);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
  } else {
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3];
  }
}

// ---------------------------------------------------------------------------------------------
// MotionEstimate reprojection problem: REF/include/MotionEstimate.h:34-91.
// predictions = K * (q * P + t) with Eigen's quaternion-vector product, residual = prediction - observation;
// AutoDiffCostFunction<ReprojectionError3Dto2D, 2, 4, 3, 3> -> Jet over (q[4], t[3]) (the 3-D point is constant:
// MotionEstimate.cc:111-114), then the 4x3 Jacobian of EigenQuaternionParameterization::Plus.
// ---------------------------------------------------------------------------------------------
template <int N> Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> h;
  const double inv = 1.0 / g.a;
  h.a = f.a * inv;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - h.a * g.v[i]) * inv;
  return h;
}
inline double operator_div(double a, double b) { return a / b; }

template <class T>
void reproj_functor(const double* intr, const double* obs, const T* q, const T* t, const double* p, T* res) {
  Q4<T> Q{q[0], q[1], q[2], q[3]};
  V3<T> P{T(p[0]), T(p[1]), T(p[2])};
  V3<T> r = qrot(Q, P);                                             // MotionEstimate.h:44  p_p = q*p + t
  T x = r.x + t[0], y = r.y + t[1], z = r.z + t[2];
  res[0] = (T(intr[0]) * x) / z + T(intr[2]) - T(obs[0]);           // :57, :78   (fx * x) / z + cx - u
  res[1] = (T(intr[1]) * y) / z + T(intr[3]) - T(obs[1]);           // :58, :79
}

// residual (2) and local Jacobian (2x6: columns dtheta(3) of q's Plus, dt(3)) of one observation
void reproj_eval_point(const double* intr, const double* obs, const double* q, const double* t, const double* p,
                       double* r, double* J) {
  typedef Jet<7> J7;
  J7 jq[4], jt[3], jr[2];
  for (int i = 0; i < 4; ++i) jq[i] = J7(q[i], i);
  for (int i = 0; i < 3; ++i) jt[i] = J7(t[i], 4 + i);
  reproj_functor<J7>(intr, obs, jq, jt, p, jr);
  double PJ[12];
  quat_plus_jacobian(q, PJ);
  for (int a = 0; a < 2; ++a) {
    r[a] = jr[a].a;
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += jr[a].v[k] * PJ[3 * k + c];
      J[6 * a + c] = s;
      J[6 * a + 3 + c] = jr[a].v[4 + c];
    }
  }
}

const double kIdentity6[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
                               0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};

// AutoDiffCostFunction<..,6,3,4,3,4>::Evaluate followed by the local-parameterization chain rule
// Ceres applies in ResidualBlock::Evaluate: J_local = J_global(6x4) * PlusJacobian(4x3).
// Outputs r[6], Ja[36], Jb[36] (row-major 6x6, columns = [dp(3) | dtheta(3)]), no loss applied.
void edge_eval_autodiff(const double* pa, const double* qa, const double* pb, const double* qb,
                        const double* mp, const double* mq, const double* L, double* r, double* Ja,
                        double* Jb) {
  typedef Jet<14> J14;
  J14 jpa[3], jqa[4], jpb[3], jqb[4], res[6];
  for (int i = 0; i < 3; ++i) jpa[i] = J14(pa[i], i);
  for (int i = 0; i < 4; ++i) jqa[i] = J14(qa[i], 3 + i);
  for (int i = 0; i < 3; ++i) jpb[i] = J14(pb[i], 7 + i);
  for (int i = 0; i < 4; ++i) jqb[i] = J14(qb[i], 10 + i);
  functor<J14>(mp, mq, L, jpa, jqa, jpb, jqb, res);
  double PJa[12], PJb[12];
  quat_plus_jacobian(qa, PJa);
  quat_plus_jacobian(qb, PJb);
  for (int i = 0; i < 6; ++i) {
    r[i] = res[i].a;
    for (int j = 0; j < 3; ++j) {
      Ja[6 * i + j] = res[i].v[j];
      Jb[6 * i + j] = res[i].v[7 + j];
      double sa = 0, sb = 0;
      for (int k = 0; k < 4; ++k) {
        sa += res[i].v[3 + k] * PJa[3 * k + j];
        sb += res[i].v[10 + k] * PJb[3 * k + j];
      }
      Ja[6 * i + 3 + j] = sa;
      Jb[6 * i + 3 + j] = sb;
    }
  }
}

// Closed-form local Jacobians (generalised so that it equals the autodiff
// chain for quaternions that are not exactly unit: the translation rows differentiate Eigen's
// v + 2w(u x v) + 2u x (u x v) formula itself instead of assuming R(q) is orthonormal).
void edge_eval_analytic(const double* pa, const double* qa, const double* pb, const double* qb,
                        const double* mp, const double* mq, const double* L, double* r, double* Ja,
                        double* Jb) {
  // c = conj(q_a): vector part u, scalar w
  const V3<double> u{-qa[0], -qa[1], -qa[2]};
  const double w = qa[3];
  const V3<double> d{pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  // Rt = I + 2w[u]x + 2[u]x[u]x  (matrix of v -> qrot(conj(q_a), v))
  double Rt[9];
  {
    const double ux[9] = {0, -u.z, u.y, u.z, 0, -u.x, -u.y, u.x, 0};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double uu = 0;
        for (int k = 0; k < 3; ++k) uu += ux[3 * i + k] * ux[3 * k + j];
        Rt[3 * i + j] = (i == j ? 1.0 : 0.0) + 2.0 * w * ux[3 * i + j] + 2.0 * uu;
      }
  }
  const V3<double> ud = cross(u, d);
  double e[6];
  {
    const V3<double> uud = cross(u, ud);
    e[0] = d.x + 2.0 * w * ud.x + 2.0 * uud.x - mp[0];
    e[1] = d.y + 2.0 * w * ud.y + 2.0 * uud.y - mp[1];
    e[2] = d.z + 2.0 * w * ud.z + 2.0 * uud.z - mp[2];
  }
  // G = d e_p / d theta_a.  Under q_a <- [dth;1] (x) q_a, conj(q_a) changes by
  //   du = -(w_a dth + dth x u_a) ,  dw = -dth . u_a      (u_a = vec(q_a) = -u)
  double G[9];
  for (int k = 0; k < 3; ++k) {
    double eh[3] = {0, 0, 0};
    eh[k] = 1.0;
    const V3<double> ek{eh[0], eh[1], eh[2]};
    const V3<double> ua{qa[0], qa[1], qa[2]};
    const V3<double> exu = cross(ek, ua);
    const V3<double> du{-(w * ek.x + exu.x), -(w * ek.y + exu.y), -(w * ek.z + exu.z)};
    const double dw = -(ek.x * ua.x + ek.y * ua.y + ek.z * ua.z);
    const V3<double> dud = cross(du, d);
    const V3<double> t1 = cross(du, ud);
    const V3<double> t2 = cross(u, dud);
    G[0 + k] = 2.0 * dw * ud.x + 2.0 * w * dud.x + 2.0 * t1.x + 2.0 * t2.x;
    G[3 + k] = 2.0 * dw * ud.y + 2.0 * w * dud.y + 2.0 * t1.y + 2.0 * t2.y;
    G[6 + k] = 2.0 * dw * ud.z + 2.0 * w * dud.z + 2.0 * t1.z + 2.0 * t2.z;
  }
  // rotation part: e_q = 2 vec(qm (x) conj(q_b) (x) q_a); A = qm (x) conj(q_b)
  const Q4<double> qm{mq[0], mq[1], mq[2], mq[3]};
  const Q4<double> qbq{qb[0], qb[1], qb[2], qb[3]};
  const Q4<double> qaq{qa[0], qa[1], qa[2], qa[3]};
  const Q4<double> A = qmul(qm, qconj(qbq));
  const Q4<double> dq = qmul(A, qaq);
  e[3] = 2.0 * dq.x; e[4] = 2.0 * dq.y; e[5] = 2.0 * dq.z;
  // M = d vec(A (x) [dth;0] (x) q_a) / d dth  (3x3)
  double M[9];
  for (int k = 0; k < 3; ++k) {
    Q4<double> dth{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0, 0.0};
    Q4<double> t = qmul(qmul(A, dth), qaq);
    M[0 + k] = t.x; M[3 + k] = t.y; M[6 + k] = t.z;
  }
  double Ea[36], Eb[36];
  for (int i = 0; i < 36; ++i) Ea[i] = Eb[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Ea[6 * i + j] = -Rt[3 * i + j];
      Ea[6 * i + 3 + j] = G[3 * i + j];
      Ea[6 * (3 + i) + 3 + j] = 2.0 * M[3 * i + j];
      Eb[6 * i + j] = Rt[3 * i + j];
      Eb[6 * (3 + i) + 3 + j] = -2.0 * M[3 * i + j];
    }
  for (int i = 0; i < 6; ++i) {
    double s = 0;
    for (int j = 0; j < 6; ++j) s += L[6 * i + j] * e[j];
    r[i] = s;
    for (int c = 0; c < 6; ++c) {
      double sa = 0, sb = 0;
      for (int j = 0; j < 6; ++j) { sa += L[6 * i + j] * Ea[6 * j + c]; sb += L[6 * i + j] * Eb[6 * j + c]; }
      Ja[6 * i + c] = sa;
      Jb[6 * i + c] = sb;
    }
  }
}

// ceres::LossFunction::Evaluate restated for HuberLoss (kind 1; the reference's, SURVEY.md App. E @0xbccc0),
// SoftLOneLoss (2), CauchyLoss (3), ArctanLoss (4) [Ceres 1.13 loss_function.cc]; kind 0 / a<=0 = trivial.
void loss_eval(int kind, double a, double s, double rho[3]) {
  const double tiny = std::numeric_limits<double>::min();
  if (kind == 1 && a > 0) {
    const double b = a * a;
    if (s > b) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a * r - b;
      rho[1] = std::max(tiny, a / r);
      rho[2] = -rho[1] / (2.0 * s);
      return;
    }
  } else if (kind == 2 && a > 0) {
    const double b = a * a, c = 1.0 / b, sum = 1.0 + s * c, tmp = std::sqrt(sum);
    rho[0] = 2.0 * b * (tmp - 1.0);
    rho[1] = std::max(tiny, 1.0 / tmp);
    rho[2] = -(c * rho[1]) / (2.0 * sum);
    return;
  } else if (kind == 3 && a > 0) {
    const double b = a * a, c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
    rho[0] = b * std::log(sum);
    rho[1] = std::max(tiny, inv);
    rho[2] = -c * (inv * inv);
    return;
  } else if (kind == 4 && a > 0) {
    const double b = 1.0 / (a * a), sum = 1.0 + s * s * b, inv = 1.0 / sum;
    rho[0] = a * std::atan2(s, a);
    rho[1] = std::max(tiny, inv);
    rho[2] = -2.0 * s * b * (inv * inv);
    return;
  }
  else if (kind == 5 && a > 0) {   // switchable constraint, switch eliminated: rho = Phi s / (Phi + s), Phi = a (include/pgo.h)
    const double q = a / (a + s);
    rho[0] = s * q;
    rho[1] = std::max(tiny, q * q);
    rho[2] = -2.0 * q * q / (a + s);
    return;
  }
  rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// Problem view (arrays owned by the caller).
//   poses : N x 7 (px py pz qx qy qz qw)       cmask : N bytes, bit0 = p constant, bit1 = q constant
//   ia/ib : id_begin / id_end pose indices      meas  : E x 7      sqrt_info : E x 36 row-major or NULL
// ---------------------------------------------------------------------------------------------
struct Prob {
  int N, E;
  const uint8_t* cmask;
  const int *ia, *ib;
  const double* meas;
  const double* sqrt_info;
  int loss_kind;
  double loss_a;
};

// One residual block with Ceres' Corrector applied: returns 0.5*rho(s).
double edge_linearize(const Prob& P, const double* poses, int e, double* r, double* Ja, double* Jb,
                      bool want_jac) {
  const int a = P.ia[e], b = P.ib[e];
  const double* L = P.sqrt_info ? P.sqrt_info + 36 * (size_t)e : kIdentity6;
  const double* pa = poses + 7 * (size_t)a;
  const double* pb = poses + 7 * (size_t)b;
  double JaL[36], JbL[36];
  if (want_jac) {
    edge_eval_analytic(pa, pa + 3, pb, pb + 3, P.meas + 7 * (size_t)e, P.meas + 7 * (size_t)e + 3, L, r, JaL, JbL);
  } else {
    functor<double>(P.meas + 7 * (size_t)e, P.meas + 7 * (size_t)e + 3, L, pa, pa + 3, pb, pb + 3, r);
  }
  double s = 0;
  for (int i = 0; i < 6; ++i) s += r[i] * r[i];
  double rho[3];
  loss_eval(P.loss_kind, P.loss_a, s, rho);
  const double sc = std::sqrt(rho[1]);  // Corrector with alpha = 0 (rho'' <= 0)
  if (want_jac) {
    for (int i = 0; i < 6; ++i) r[i] *= sc;
    for (int i = 0; i < 36; ++i) { Ja[i] = JaL[i] * sc; Jb[i] = JbL[i] * sc; }
    // constant parameter blocks: their Jacobian columns are removed from the program
    const uint8_t ma = P.cmask[a], mb = P.cmask[b];
    for (int i = 0; i < 6; ++i) {
      if (ma & 1) Ja[6 * i] = Ja[6 * i + 1] = Ja[6 * i + 2] = 0.0;
      if (ma & 2) Ja[6 * i + 3] = Ja[6 * i + 4] = Ja[6 * i + 5] = 0.0;
      if (mb & 1) Jb[6 * i] = Jb[6 * i + 1] = Jb[6 * i + 2] = 0.0;
      if (mb & 2) Jb[6 * i + 3] = Jb[6 * i + 4] = Jb[6 * i + 5] = 0.0;
    }
  }
  return 0.5 * rho[0];
}

double total_cost(const Prob& P, const double* poses) {
  double c = 0, r[6];
  for (int e = 0; e < P.E; ++e) c += edge_linearize(P, poses, e, r, nullptr, nullptr, false);
  return c;
}
// threaded: the per-edge terms in parallel, their sum in edge order (the bits of total_cost)
double total_cost_mt(const Prob& P, const double* poses, Pool& pool, std::vector<double>& term) {
  term.resize((size_t)P.E);
  const int chunks = pool.width() * 4;
  pool.run(chunks, [&](int c) {
    const int lo = (int)((long long)P.E * c / chunks), hi = (int)((long long)P.E * (c + 1) / chunks);
    double r[6];
    for (int e = lo; e < hi; ++e) term[(size_t)e] = edge_linearize(P, poses, e, r, nullptr, nullptr, false);
  });
  double c = 0;
  for (int e = 0; e < P.E; ++e) c += term[(size_t)e];
  return c;
}

// ---------------------------------------------------------------------------------------------
// Block-sparse normal equations, 6x6 pose blocks.  Lower triangle by block column:
// col j holds the diagonal block first, then off-diagonal blocks (i > j) sorted by row.
// ---------------------------------------------------------------------------------------------
typedef std::array<double, 36> Blk;

struct BlockSym {
  int n = 0;
  std::vector<int> colptr, rowidx;  // block CSC of the lower triangle (diag included, first)
  std::vector<Blk> val;
  std::vector<int> edge_slot;       // for each edge: slot of its off-diagonal block
  std::vector<uint8_t> edge_transposed;  // 1 if the stored block is (b,a) i.e. J_b^T J_a
};

void build_structure(const Prob& P, BlockSym& H) {
  const int n = P.N;
  H.n = n;
  std::vector<std::vector<int>> cols(n);
  for (int e = 0; e < P.E; ++e) {
    int a = P.ia[e], b = P.ib[e];
    if (a == b) continue;
    int j = std::min(a, b), i = std::max(a, b);
    cols[j].push_back(i);
  }
  H.colptr.assign(n + 1, 0);
  for (int j = 0; j < n; ++j) {
    std::sort(cols[j].begin(), cols[j].end());
    cols[j].erase(std::unique(cols[j].begin(), cols[j].end()), cols[j].end());
    H.colptr[j + 1] = H.colptr[j] + 1 + (int)cols[j].size();
  }
  H.rowidx.resize(H.colptr[n]);
  H.val.resize(H.colptr[n]);
  for (int j = 0; j < n; ++j) {
    int p = H.colptr[j];
    H.rowidx[p++] = j;
    for (int i : cols[j]) H.rowidx[p++] = i;
  }
  H.edge_slot.resize(P.E);
  H.edge_transposed.resize(P.E);
  for (int e = 0; e < P.E; ++e) {
    int a = P.ia[e], b = P.ib[e];
    int j = std::min(a, b), i = std::max(a, b);
    const int* lo = &H.rowidx[H.colptr[j] + 1];
    const int* hi = &H.rowidx[0] + H.colptr[j + 1];
    H.edge_slot[e] = (int)(std::lower_bound(lo, hi, i) - &H.rowidx[0]);
    // block (i,j) = J_i^T J_j.  If a is the larger index, block = J_a^T J_b, else J_b^T J_a.
    H.edge_transposed[e] = (a > b) ? 0 : 1;
  }
}

// C(6x6) += A^T B
inline void atb_add(const double* A, const double* B, double* C) {
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += A[6 * k + i] * B[6 * k + j];
      C[6 * i + j] += s;
    }
}

// H = J^T J (blocks), g = J^T r.  Constant dims: zero row/col + unit diagonal, g = 0.
double linearize(const Prob& P, const double* poses, BlockSym& H, std::vector<double>& g) {
  for (auto& b : H.val) b.fill(0.0);
  g.assign((size_t)6 * P.N, 0.0);
  double cost = 0;
  double r[6], Ja[36], Jb[36];
  for (int e = 0; e < P.E; ++e) {
    cost += edge_linearize(P, poses, e, r, Ja, Jb, true);
    const int a = P.ia[e], b = P.ib[e];
    atb_add(Ja, Ja, H.val[H.colptr[a]].data());
    atb_add(Jb, Jb, H.val[H.colptr[b]].data());
    if (a != b) {
      if (H.edge_transposed[e]) atb_add(Jb, Ja, H.val[H.edge_slot[e]].data());
      else atb_add(Ja, Jb, H.val[H.edge_slot[e]].data());
    }
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 6; ++k) {
        g[6 * (size_t)a + i] += Ja[6 * k + i] * r[k];
        g[6 * (size_t)b + i] += Jb[6 * k + i] * r[k];
      }
  }
  for (int v = 0; v < P.N; ++v) {
    double* D = H.val[H.colptr[v]].data();
    for (int i = 0; i < 6; ++i) {
      const bool c = (i < 3) ? (P.cmask[v] & 1) : (P.cmask[v] & 2);
      if (c) D[7 * i] = 1.0;
    }
  }
  return cost;
}

// threaded linearize: residuals and Jacobians of all edges in parallel, then every thread walks ALL edges in order and adds only
// into the blocks / gradient rows of the poses it owns (a contiguous range): each block receives its terms in edge order, as in
// linearize(), so the result is the same to the bit.
struct EdgeLin { double r[6], Ja[36], Jb[36], cost; };
double linearize_mt(const Prob& P, const double* poses, BlockSym& H, std::vector<double>& g, Pool& pool, std::vector<EdgeLin>& el) {
  el.resize((size_t)P.E);
  g.assign((size_t)6 * P.N, 0.0);
  const int T = pool.width(), chunks = T * 4;
  pool.run(chunks, [&](int c) {
    const int lo = (int)((long long)P.E * c / chunks), hi = (int)((long long)P.E * (c + 1) / chunks);
    for (int e = lo; e < hi; ++e) { EdgeLin& L = el[(size_t)e]; L.cost = edge_linearize(P, poses, e, L.r, L.Ja, L.Jb, true); }
  });
  pool.run(T, [&](int t) {
    const int v0 = (int)((long long)P.N * t / T), v1 = (int)((long long)P.N * (t + 1) / T);
    auto mine = [&](int v) { return v >= v0 && v < v1; };
    for (int v = v0; v < v1; ++v)
      for (int p = H.colptr[v]; p < H.colptr[v + 1]; ++p) H.val[p].fill(0.0);     // column v: its diagonal block and the blocks below it
    for (int e = 0; e < P.E; ++e) {
      const int a = P.ia[e], b = P.ib[e];
      const bool ma = mine(a), mb = mine(b);
      if (!ma && !mb) continue;
      const EdgeLin& L = el[(size_t)e];
      if (ma) atb_add(L.Ja, L.Ja, H.val[H.colptr[a]].data());
      if (mb) atb_add(L.Jb, L.Jb, H.val[H.colptr[b]].data());
      if (a != b && mine(std::min(a, b))) {       // the off-diagonal block lives in column min(a, b)
        if (H.edge_transposed[e]) atb_add(L.Jb, L.Ja, H.val[H.edge_slot[e]].data());
        else atb_add(L.Ja, L.Jb, H.val[H.edge_slot[e]].data());
      }
      for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 6; ++k) {
          if (ma) g[6 * (size_t)a + i] += L.Ja[6 * k + i] * L.r[k];
          if (mb) g[6 * (size_t)b + i] += L.Jb[6 * k + i] * L.r[k];
        }
    }
    for (int v = v0; v < v1; ++v) {
      double* D = H.val[H.colptr[v]].data();
      for (int i = 0; i < 6; ++i) {
        const bool c = (i < 3) ? (P.cmask[v] & 1) : (P.cmask[v] & 2);
        if (c) D[7 * i] = 1.0;
      }
    }
  });
  double cost = 0;
  for (int e = 0; e < P.E; ++e) cost += el[(size_t)e].cost;
  return cost;
}

// y = (H + diag(d2)) x  using the symmetric lower storage
void sym_matvec(const BlockSym& H, const double* d2, const double* x, double* y) {
  const int n = H.n;
  for (size_t i = 0; i < (size_t)6 * n; ++i) y[i] = d2 ? d2[i] * x[i] : 0.0;
  for (int j = 0; j < n; ++j) {
    for (int p = H.colptr[j]; p < H.colptr[j + 1]; ++p) {
      const int i = H.rowidx[p];
      const double* B = H.val[p].data();
      const double* xj = x + 6 * (size_t)j;
      double* yi = y + 6 * (size_t)i;
      for (int r = 0; r < 6; ++r) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += B[6 * r + c] * xj[c];
        yi[r] += s;
      }
      if (i != j) {
        const double* xi = x + 6 * (size_t)i;
        double* yj = y + 6 * (size_t)j;
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int r = 0; r < 6; ++r) s += B[6 * r + c] * xi[r];
          yj[c] += s;
        }
      }
    }
  }
}

// dense 6x6 Cholesky (lower), in place; returns false if not positive definite
bool chol6(double* A) {
  for (int j = 0; j < 6; ++j) {
    double d = A[7 * j];
    for (int k = 0; k < j; ++k) d -= A[6 * j + k] * A[6 * j + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[7 * j] = d;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k];
      A[6 * i + j] = s / d;
    }
    for (int i = 0; i < j; ++i) A[6 * i + j] = 0.0;
  }
  return true;
}
// solve L L^T x = b for 6-vector, L lower from chol6
void chol6_solve(const double* L, const double* b, double* x) {
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
    y[i] = s / L[7 * i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
    x[i] = s / L[7 * i];
  }
}

// ---------------------------------------------------------------------------------------------
// Exact linear solver: block-sparse Cholesky  P (H + D^2) P^T = L L^T  with a minimum-degree
// ordering on the pose graph, block up-looking numeric phase.  Plays the role of
// SparseNormalCholeskySolver + CHOLMOD in the reference.
// ---------------------------------------------------------------------------------------------
struct SparseChol {
  int n = 0;
  std::vector<int> perm, iperm, parent;
  // permuted upper pattern per column k: rows i<k with A(i,k) != 0, with source slot + transpose flag
  struct Src { int row; int slot; uint8_t transpose; };
  std::vector<std::vector<Src>> upper;
  std::vector<int> diag_slot;
  // factor: per column, rows ascending (diag first) and blocks
  std::vector<std::vector<int>> Lrow;
  std::vector<std::vector<Blk>> Lval;
  long long nnz_blocks = 0;
  double flops = 0;

  static void min_degree(int n, const std::vector<std::vector<int>>& adj0, std::vector<int>& perm) {
    std::vector<std::vector<int>> adj = adj0;
    std::vector<char> dead(n, 0);
    typedef std::pair<int, int> DI;
    std::priority_queue<DI, std::vector<DI>, std::greater<DI>> pq;
    std::vector<int> deg(n);
    for (int v = 0; v < n; ++v) { deg[v] = (int)adj[v].size(); pq.push(DI(deg[v], v)); }
    perm.clear();
    perm.reserve(n);
    std::vector<int> merged;
    while (!pq.empty()) {
      DI top = pq.top(); pq.pop();
      const int v = top.second;
      if (dead[v] || top.first != deg[v]) continue;
      dead[v] = 1;
      perm.push_back(v);
      std::vector<int> nb;
      for (int u : adj[v]) if (!dead[u]) nb.push_back(u);
      for (int u : nb) {
        merged.clear();
        std::set_union(adj[u].begin(), adj[u].end(), nb.begin(), nb.end(), std::back_inserter(merged));
        std::vector<int> out;
        out.reserve(merged.size());
        for (int w : merged) if (w != u && !dead[w]) out.push_back(w);
        adj[u].swap(out);
        deg[u] = (int)adj[u].size();
        pq.push(DI(deg[u], u));
      }
      std::vector<int>().swap(adj[v]);
    }
  }

  void analyze(const BlockSym& H) {
    n = H.n;
    std::vector<std::vector<int>> adj(n);
    for (int j = 0; j < n; ++j)
      for (int p = H.colptr[j] + 1; p < H.colptr[j + 1]; ++p) {
        adj[j].push_back(H.rowidx[p]);
        adj[H.rowidx[p]].push_back(j);
      }
    for (auto& a : adj) std::sort(a.begin(), a.end());
    min_degree(n, adj, perm);
    iperm.assign(n, 0);
    for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
    upper.assign(n, {});
    diag_slot.assign(n, 0);
    for (int j = 0; j < n; ++j) {
      diag_slot[iperm[j]] = H.colptr[j];
      for (int p = H.colptr[j] + 1; p < H.colptr[j + 1]; ++p) {
        const int i = H.rowidx[p];  // stored block = (i,j), i>j in original numbering
        const int pi = iperm[i], pj = iperm[j];
        // we need the block at (row=min, col=max) of the permuted upper triangle
        if (pi < pj) upper[pj].push_back(Src{pi, p, 0});  // (pi,pj) = stored (i,j) as is
        else upper[pi].push_back(Src{pj, p, 1});          // (pj,pi) = stored^T
      }
    }
    // elimination tree (Liu), with path compression
    parent.assign(n, -1);
    std::vector<int> anc(n, -1);
    for (int k = 0; k < n; ++k)
      for (const Src& s : upper[k]) {
        int i = s.row;
        while (i != -1 && i < k) {
          int nx = anc[i];
          anc[i] = k;
          if (nx == -1) parent[i] = k;
          i = nx;
        }
      }
    Lrow.assign(n, {});
    Lval.assign(n, {});
  }

  // numeric factorisation of H + diag(d2); returns false on a non-positive pivot.  Up-looking, one block row at a time; a row only
  // touches the columns of its descendants in the elimination tree, so disjoint subtrees can be factorised by different threads
  // (pool != null): the subtrees below a size cut in parallel, the top of the tree — the separators, most of the flops on a mesh
  // — by the calling thread.  Every row does the same operations in the same order either way: same bits.
  struct Work {
    std::vector<Blk> x;
    std::vector<int> mark, stack, reach;
    double flops = 0;
    explicit Work(int n) : x(n), mark(n, -1), stack(n), reach(n) {}
  };
  bool factor_row(const BlockSym& H, const double* d2, int k, Work& W) {
    std::vector<Blk>& x = W.x;
    std::vector<int>&mark = W.mark, &stack = W.stack, &reach = W.reach;
    {
      // scatter column k of the permuted upper triangle into x, compute reach
      int top = n;
      mark[k] = k;
      for (const Src& s : upper[k]) {
        const double* B = H.val[s.slot].data();
        Blk& xb = x[s.row];
        if (s.transpose) { for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) xb[6 * r + c] = B[6 * c + r]; }
        else { for (int q = 0; q < 36; ++q) xb[q] = B[q]; }
        int len = 0;
        for (int i = s.row; mark[i] != k; i = parent[i]) { stack[len++] = i; mark[i] = k; }
        while (len > 0) reach[--top] = stack[--len];
      }
      // NOTE: entries reached through the etree but not present in A start at zero
      Blk d;
      {
        const int orig = perm[k];
        const double* B = H.val[diag_slot[k]].data();
        for (int q = 0; q < 36; ++q) d[q] = B[q];
        if (d2) for (int i = 0; i < 6; ++i) d[7 * i] += d2[6 * (size_t)orig + i];
      }
      for (int t = top; t < n; ++t) {
        const int j = reach[t];
        // L_kj = x_j^T L_jj^{-T}   <=>  solve L_jj * Lkj^T = x_j   (column by column)
        const double* Ljj = Lval[j][0].data();
        Blk lkjT;  // holds Lkj^T (6x6): column c solves L_jj y = x_j[:,c]
        for (int c = 0; c < 6; ++c) {
          double y[6];
          for (int i = 0; i < 6; ++i) {
            double s = x[j][6 * i + c];
            for (int q = 0; q < i; ++q) s -= Ljj[6 * i + q] * y[q];
            y[i] = s / Ljj[7 * i];
          }
          for (int i = 0; i < 6; ++i) lkjT[6 * i + c] = y[i];
        }
        Blk lkj;
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) lkj[6 * r + c] = lkjT[6 * c + r];
        x[j].fill(0.0);
        // x_i -= L_ij * L_kj^T  for stored rows i of column j (j < i < k)
        const size_t cnt = Lrow[j].size();
        for (size_t p = 1; p < cnt; ++p) {
          const int i = Lrow[j][p];
          const double* Lij = Lval[j][p].data();
          Blk& xi = x[i];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
              double s = 0;
              for (int q = 0; q < 6; ++q) s += Lij[6 * r + q] * lkj[6 * c + q];
              xi[6 * r + c] -= s;
            }
        }
        W.flops += 432.0 * (double)(cnt - 1) + 432.0 + 216.0;
        // d -= L_kj L_kj^T
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            double s = 0;
            for (int q = 0; q < 6; ++q) s += lkj[6 * r + q] * lkj[6 * c + q];
            d[6 * r + c] -= s;
          }
        Lrow[j].push_back(k);
        Lval[j].push_back(lkj);
      }
      // the x entries for rows in reach were initialised lazily: rows never scattered must be zero.
      if (!chol6(d.data())) return false;
      Lrow[k].push_back(k);
      Lval[k].push_back(d);
    }
    return true;
  }
  // subtrees of the elimination tree handed to the pool (rows of a subtree in ascending order), and the rows left to the caller
  std::vector<std::vector<int>> par_tasks;
  std::vector<int> par_top;
  std::vector<Work> par_work;
  void plan_parallel(int nthreads) {
    par_tasks.clear(); par_top.clear();
    std::vector<int> size(n, 1);
    for (int k = 0; k < n; ++k) if (parent[k] >= 0) size[parent[k]] += size[k];       // (children have smaller numbers than their parents)
    const int cap = std::max(64, n / (8 * nthreads));
    std::vector<int> task_of(n, -1);
    for (int k = n - 1; k >= 0; --k) {
      const int pa = parent[k];
      if (pa >= 0 && task_of[pa] >= 0) { task_of[k] = task_of[pa]; continue; }
      if (size[k] <= cap) { task_of[k] = (int)par_tasks.size(); par_tasks.emplace_back(); }
    }
    for (int k = 0; k < n; ++k) { if (task_of[k] >= 0) par_tasks[task_of[k]].push_back(k); else par_top.push_back(k); }
    std::sort(par_tasks.begin(), par_tasks.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
    par_work.clear();
    for (int t = 0; t < nthreads; ++t) par_work.emplace_back(n);
  }
  bool factor(const BlockSym& H, const double* d2, Pool* pool = nullptr) {
    for (int j = 0; j < n; ++j) { Lrow[j].clear(); Lval[j].clear(); }
    nnz_blocks = 0;
    flops = 0;
    bool ok = true;
    if (!pool || pool->width() <= 1) {
      Work W(n);
      for (int k = 0; k < n && ok; ++k) ok = factor_row(H, d2, k, W);
      flops = W.flops;
    } else {
      if ((int)par_work.size() != pool->width()) plan_parallel(pool->width());
      for (Work& W : par_work) { W.flops = 0; std::fill(W.mark.begin(), W.mark.end(), -1); }   // (marks are row numbers: stale ones from the last factorisation would cut the reach short)
      std::atomic<int> slot{0}, bad{0};
      std::vector<int> who(par_tasks.size(), -1);
      // a thread keeps one workspace for all the subtrees it picks up
      std::mutex mu;
      std::vector<std::thread::id> ids;
      pool->run((int)par_tasks.size(), [&](int t) {
        int w;
        {
          std::lock_guard<std::mutex> lk(mu);
          const auto id = std::this_thread::get_id();
          w = (int)(std::find(ids.begin(), ids.end(), id) - ids.begin());
          if (w == (int)ids.size()) ids.push_back(id);
        }
        for (int k : par_tasks[t]) if (!factor_row(H, d2, k, par_work[w])) { bad.store(1); break; }
      });
      ok = bad.load() == 0;
      for (size_t i = 0; i < par_top.size() && ok; ++i) ok = factor_row(H, d2, par_top[i], par_work[0]);
      for (const Work& W : par_work) flops += W.flops;
      (void)slot; (void)who;
    }
    if (!ok) return false;
    for (int j = 0; j < n; ++j) nnz_blocks += (long long)Lrow[j].size();
    return true;
  }

  void solve(const double* b, double* out) const {
    std::vector<double> y((size_t)6 * n);
    for (int k = 0; k < n; ++k) for (int i = 0; i < 6; ++i) y[6 * (size_t)k + i] = b[6 * (size_t)perm[k] + i];
    // forward: L y = b
    for (int j = 0; j < n; ++j) {
      double* yj = &y[6 * (size_t)j];
      const double* Ljj = Lval[j][0].data();
      for (int i = 0; i < 6; ++i) {
        double s = yj[i];
        for (int q = 0; q < i; ++q) s -= Ljj[6 * i + q] * yj[q];
        yj[i] = s / Ljj[7 * i];
      }
      for (size_t p = 1; p < Lrow[j].size(); ++p) {
        double* yi = &y[6 * (size_t)Lrow[j][p]];
        const double* B = Lval[j][p].data();
        for (int r = 0; r < 6; ++r) {
          double s = 0;
          for (int c = 0; c < 6; ++c) s += B[6 * r + c] * yj[c];
          yi[r] -= s;
        }
      }
    }
    // backward: L^T x = y
    for (int j = n - 1; j >= 0; --j) {
      double* yj = &y[6 * (size_t)j];
      for (size_t p = 1; p < Lrow[j].size(); ++p) {
        const double* yi = &y[6 * (size_t)Lrow[j][p]];
        const double* B = Lval[j][p].data();
        for (int c = 0; c < 6; ++c) {
          double s = 0;
          for (int r = 0; r < 6; ++r) s += B[6 * r + c] * yi[r];
          yj[c] -= s;
        }
      }
      const double* Ljj = Lval[j][0].data();
      for (int i = 5; i >= 0; --i) {
        double s = yj[i];
        for (int q = i + 1; q < 6; ++q) s -= Ljj[6 * q + i] * yj[q];
        yj[i] = s / Ljj[7 * i];
      }
    }
    for (int k = 0; k < n; ++k) for (int i = 0; i < 6; ++i) out[6 * (size_t)perm[k] + i] = y[6 * (size_t)k + i];
  }
};

// ---------------------------------------------------------------------------------------------
// Ceres 1.13 ConjugateGradientsSolver restated (Q-tolerance termination, residual reset every 10
// iterations) with a block-Jacobi preconditioner on the 6x6 pose blocks of H + D^2.
// Solves (H + D^2) x = b, x0 = 0.  Returns iterations used; *ok=false on breakdown.
// ---------------------------------------------------------------------------------------------
// Dense SPD solve helpers for the cluster preconditioner (n <= 6 * cluster)
bool chol_dense(std::vector<double>& A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double t = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = t / d;
    }
  }
  return true;
}
void chol_dense_solve(const std::vector<double>& L, int n, const double* b, double* x) {
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) { double t = b[i]; for (int k = 0; k < i; ++k) t -= L[(size_t)i * n + k] * y[k]; y[i] = t / L[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < n; ++k) t -= L[(size_t)k * n + i] * x[k]; x[i] = t / L[(size_t)i * n + i]; }
}

// Threaded form of sym_matvec: per row the contributions in the order the sequential loop over (column, entry) delivers them, so
// a row summed by any thread has the bits of the one-thread product.
struct RowEvents {
  struct Ev { int p, col; bool transposed; };
  std::vector<std::vector<Ev>> rows;
  explicit RowEvents(const BlockSym& H) : rows(H.n) {
    for (int j = 0; j < H.n; ++j)
      for (int p = H.colptr[j]; p < H.colptr[j + 1]; ++p) {
        const int i = H.rowidx[p];
        rows[i].push_back(Ev{p, j, false});
        if (i != j) rows[j].push_back(Ev{p, i, true});
      }
  }
  void matvec(const BlockSym& H, const double* d2, const double* x, double* y, Pool& pool) const {
    const int n = H.n, chunks = pool.width() * 4;
    pool.run(chunks, [&](int c) {
      const int k0 = (int)((long long)n * c / chunks), k1 = (int)((long long)n * (c + 1) / chunks);
      for (int k = k0; k < k1; ++k) {
        double* yk = y + 6 * (size_t)k;
        for (int r = 0; r < 6; ++r) yk[r] = d2 ? d2[6 * (size_t)k + r] * x[6 * (size_t)k + r] : 0.0;
        for (const Ev& e : rows[k]) {
          const double* B = H.val[e.p].data();
          const double* xc = x + 6 * (size_t)e.col;
          if (!e.transposed) {
            for (int r = 0; r < 6; ++r) { double s2 = 0; for (int cc = 0; cc < 6; ++cc) s2 += B[6 * r + cc] * xc[cc]; yk[r] += s2; }
          } else {
            for (int cc = 0; cc < 6; ++cc) { double s2 = 0; for (int r = 0; r < 6; ++r) s2 += B[6 * r + cc] * xc[r]; yk[cc] += s2; }
          }
        }
      }
    });
  }
};

// form 0: Ceres' ConjugateGradientsSolver statement by statement.  form 1: the same Krylov iterates through the pipelined
// recurrences of Ghysels & Vanroose (2014) — u = M^-1 r, w = A u, m = M^-1 w, n = A m; z = n + beta z, qq = m + beta qq,
// s = w + beta s, p = u + beta p; x += alpha p, r -= alpha s, u -= alpha qq, w -= alpha z — with ONE set of inner products
// (gamma = (r,u), delta = (w,u), Q = -x'(b + r)) per iteration, the form the product's one-launch CG iteration computes
// (posegraph-ceres_amd/csrc/pgo_uni_fused.h); same stop rules on the same quantities, no residual refresh.
// r06 EXPERIMENT (measured here before any kernel is written, as the chain preconditioner was in r05): a TWO-LEVEL additive
// preconditioner  M^-1 = M_J^-1 + P (P' A P)^-1 P'  — M_J the 2-pose cluster Jacobi, P an aggregation coarse space: aggregates of `agg`
// consecutive poses of the trajectory, six rigid-body modes each (a translation t and a rotation w about the aggregate's centre:
// dp_i = t + 2 w x (p_i - c), dtheta_i = w in the tangent coordinates of Plus, which multiplies exp(delta) from the LEFT), expressed in
// the Jacobi-scaled unknowns the CG works in (P~ = S^-1 P, rows of constant blocks zero).  Galerkin coarse matrix P~' (H~ + D^2) P~,
// dense 6 A x 6 A, factored once per linear solve.  Selected by pcg_cluster = -agg (agg >= 8; -1 stays the chain).
struct CoarseSpace { int agg; const double* poses; const double* scale; const uint8_t* cmask; };
// Row shards (the product's sharded solve: aggregates never straddle ranks, csrc/pgo_coarse.h): the segments' first poses + the end, set by
// pgo_oracle_set_coarse_cuts; empty = one segment.  Aggregate of pose v = (aggregates of the segments in front) + (v - segment start) / agg.
static std::vector<long long> g_coarse_cuts;

int pcg_solve(const BlockSym& H, const double* d2, const double* b, double* x, double q_tol,
              int max_it, int min_it, int residual_reset_period, bool* ok, double* final_rnorm, int cluster = 1, Pool* pool = nullptr,
              int form = 0, const CoarseSpace* cz = nullptr) {
  const int n = H.n;
  const size_t m = (size_t)6 * n;
  if (pool && pool->width() <= 1) pool = nullptr;
  std::unique_ptr<RowEvents> rows;
  if (pool) rows.reset(new RowEvents(H));
  auto matvec = [&](const double* in, double* outv) { if (pool) rows->matvec(H, d2, in, outv, *pool); else sym_matvec(H, d2, in, outv); };
  // block-Jacobi preconditioner; cluster > 1 groups `cluster` consecutive poses (a piece of the odometry
  // chain) into one dense diagonal block of H + D^2 (the product's cluster-Jacobi option)
  // cluster == -1: the odometry chain solved exactly (M = the block-tridiagonal part of H + D^2: every diagonal block,
  // and the blocks (i+1, i) where an edge couples consecutive poses; SPD, since every dropped off-diagonal block leaves
  // its edge's diagonal contributions behind).  Block LDL^T down the chain: S_i = A_ii - W_i S_{i-1} W_i^T with
  // W_i = A_{i,i-1} S_{i-1}^-1 — the recurrence the product's chain preconditioner factorises by segments.
  const bool chain = cluster == -1;
  // ---- two-level: the coarse space and its Galerkin matrix ----
  const bool two_level = cluster <= -8 && cz != nullptr;
  int nagg = 0, cdim = 0;
  std::vector<int> aggof, agg_begin;   // aggregate of every pose; first pose of every aggregate (+ n)
  std::vector<double> Pt;              // P~ per pose: 6 x 6 row-major (row = fine component, column = mode of the pose's aggregate)
  std::vector<double> Ac;              // Cholesky factor of the coarse matrix
  if (two_level) {
    const int agg = -cluster;
    std::vector<long long> cuts = g_coarse_cuts;
    if (cuts.size() < 2 || cuts.front() != 0 || cuts.back() != n) cuts = {0, (long long)n};
    agg_begin.clear(); aggof.assign((size_t)n, 0);
    for (size_t sgm = 0; sgm + 1 < cuts.size(); ++sgm)
      for (long long v0 = cuts[sgm]; v0 < cuts[sgm + 1]; v0 += agg) {
        const long long v1 = std::min(cuts[sgm + 1], v0 + agg);
        for (long long v = v0; v < v1; ++v) aggof[(size_t)v] = (int)agg_begin.size();
        agg_begin.push_back((int)v0);
      }
    agg_begin.push_back(n);
    nagg = (int)agg_begin.size() - 1; cdim = 6 * nagg;
    Pt.assign((size_t)36 * n, 0.0);
    for (int a = 0; a < nagg; ++a) {
      const int v0 = agg_begin[(size_t)a], v1 = std::min((int)n, std::min(agg_begin[(size_t)a + 1], v0 + agg));
      double c[3] = {0, 0, 0};
      for (int v = v0; v < v1; ++v) for (int k = 0; k < 3; ++k) c[k] += cz->poses[7 * (size_t)v + k] / (v1 - v0);
      for (int v = v0; v < v1; ++v) {
        const double d[3] = {cz->poses[7 * (size_t)v] - c[0], cz->poses[7 * (size_t)v + 1] - c[1], cz->poses[7 * (size_t)v + 2] - c[2]};
        double* Pv = &Pt[(size_t)36 * v];
        for (int k = 0; k < 3; ++k) { Pv[6 * k + k] = 1.0; Pv[6 * (3 + k) + 3 + k] = 1.0; }
        // dp = 2 w x d: column 3 + j (w = e_j) -> 2 e_j x d
        Pv[6 * 1 + 3] = -2 * d[2]; Pv[6 * 2 + 3] = 2 * d[1];      // e_x x d = (0, -d_z, d_y)
        Pv[6 * 0 + 4] = 2 * d[2];  Pv[6 * 2 + 4] = -2 * d[0];     // e_y x d = (d_z, 0, -d_x)
        Pv[6 * 0 + 5] = -2 * d[1]; Pv[6 * 1 + 5] = 2 * d[0];      // e_z x d = (-d_y, d_x, 0)
        for (int r = 0; r < 6; ++r) {
          const bool cst = (r < 3) ? (cz->cmask[v] & 1) : (cz->cmask[v] & 2);
          for (int q = 0; q < 6; ++q) Pv[6 * r + q] = cst ? 0.0 : Pv[6 * r + q] / cz->scale[6 * (size_t)v + r];
        }
      }
    }
    Ac.assign((size_t)cdim * cdim, 0.0);
    auto add = [&](int i, int j, const double* B, bool with_d2) {      // Ac[a(i), a(j)] += P_i' B P_j  (B = block (i, j) of H~ + D^2)
      const int ai = aggof[(size_t)i], aj = aggof[(size_t)j];
      const double *Pi = &Pt[(size_t)36 * i], *Pj = &Pt[(size_t)36 * j];
      double BP[36];
      for (int r = 0; r < 6; ++r)
        for (int q = 0; q < 6; ++q) {
          double acc = 0;
          for (int k = 0; k < 6; ++k) acc += (B[6 * r + k] + ((with_d2 && r == k) ? d2[6 * (size_t)i + r] : 0.0)) * Pj[6 * k + q];
          BP[6 * r + q] = acc;
        }
      for (int p2 = 0; p2 < 6; ++p2)
        for (int q = 0; q < 6; ++q) {
          double acc = 0;
          for (int r = 0; r < 6; ++r) acc += Pi[6 * r + p2] * BP[6 * r + q];
          Ac[(size_t)(6 * ai + p2) * cdim + 6 * aj + q] += acc;
          if (i != j) Ac[(size_t)(6 * aj + q) * cdim + 6 * ai + p2] += acc;
        }
    };
    for (int j = 0; j < n; ++j)
      for (int p2 = H.colptr[j]; p2 < H.colptr[j + 1]; ++p2) add(H.rowidx[p2], j, H.val[p2].data(), H.rowidx[p2] == j);
    for (int k = 0; k < cdim; ++k) if (!(Ac[(size_t)k * cdim + k] > 0.0)) Ac[(size_t)k * cdim + k] = 1.0;      // an aggregate of constant blocks only
    if (!chol_dense(Ac, cdim)) { *ok = false; return 0; }
    cluster = 2;
  }
  std::vector<Blk> chS, chW;        // Cholesky factor of S_i (lower, row-major 6x6), W_i (row-major)
  std::vector<uint8_t> chHas;
  if (chain) {
    chS.resize(n); chW.resize(n); chHas.assign(n, 0);
    for (int i = 0; i < n; ++i) {
      double S[36];
      for (int k = 0; k < 36; ++k) S[k] = H.val[H.colptr[i]][k];
      for (int k = 0; k < 6; ++k) S[7 * k] += d2[6 * (size_t)i + k];
      // symmetrise from the stored lower triangle convention (full 6x6 diagonal blocks are stored)
      if (i > 0) {
        const int j = i - 1;
        int slot = -1;
        if (H.colptr[j] + 1 < H.colptr[j + 1] && H.rowidx[H.colptr[j] + 1] == i) slot = H.colptr[j] + 1;
        if (slot >= 0) {
          chHas[i] = 1;
          const double* A = H.val[slot].data();     // block (i, i-1), row-major
          // W = A S_{i-1}^-1: solve S_{i-1} W^T = A^T column by column
          double* W = chW[i].data();
          for (int r = 0; r < 6; ++r) {
            double rhs[6], sol[6];
            for (int c = 0; c < 6; ++c) rhs[c] = A[6 * r + c];
            chol6_solve(chS[j].data(), rhs, sol);
            for (int c = 0; c < 6; ++c) W[6 * r + c] = sol[c];
          }
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
              double acc = 0;
              for (int k = 0; k < 6; ++k) acc += W[6 * r + k] * A[6 * c + k];
              S[6 * r + c] -= acc;
            }
        }
      }
      for (int k = 0; k < 36; ++k) chS[i][k] = S[k];
      if (!chol6(chS[i].data())) { *ok = false; return 0; }
    }
    cluster = 1;
  }
  if (cluster < 1) cluster = 1;
  const int ncl = chain ? 0 : (n + cluster - 1) / cluster;
  std::vector<std::vector<double>> Mfac(ncl);
  for (int c = 0; c < ncl; ++c) {
    const int v0 = c * cluster, v1 = std::min(n, v0 + cluster), dim = 6 * (v1 - v0);
    std::vector<double>& A = Mfac[c];
    A.assign((size_t)dim * dim, 0.0);
    for (int j = v0; j < v1; ++j)
      for (int p = H.colptr[j]; p < H.colptr[j + 1]; ++p) {
        const int i = H.rowidx[p];
        if (i < v0 || i >= v1) continue;
        for (int r = 0; r < 6; ++r)
          for (int cc = 0; cc < 6; ++cc) {
            const double val = H.val[p][6 * r + cc];
            A[(size_t)(6 * (i - v0) + r) * dim + 6 * (j - v0) + cc] = val;
            A[(size_t)(6 * (j - v0) + cc) * dim + 6 * (i - v0) + r] = val;
          }
      }
    for (int k = 0; k < dim; ++k) A[(size_t)k * dim + k] += d2[6 * (size_t)v0 + k];
    if (!chol_dense(A, dim)) { *ok = false; return 0; }
  }
  auto apply_M = [&](const double* rin, double* zout) {
    if (chain) {
      std::vector<double> y(m);
      for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 6; ++k) y[6 * (size_t)i + k] = rin[6 * (size_t)i + k];
        if (chHas[i])
          for (int r = 0; r < 6; ++r) {
            double acc = 0;
            for (int k = 0; k < 6; ++k) acc += chW[i][6 * r + k] * y[6 * (size_t)(i - 1) + k];
            y[6 * (size_t)i + r] -= acc;
          }
      }
      for (int i = n - 1; i >= 0; --i) {
        chol6_solve(chS[i].data(), &y[6 * (size_t)i], zout + 6 * (size_t)i);
        if (i + 1 < n && chHas[i + 1])
          for (int c = 0; c < 6; ++c) {
            double acc = 0;
            for (int k = 0; k < 6; ++k) acc += chW[i + 1][6 * k + c] * zout[6 * (size_t)(i + 1) + k];
            zout[6 * (size_t)i + c] -= acc;
          }
      }
      return;
    }
    auto one = [&](int c) {
      const int v0 = c * cluster, v1 = std::min(n, v0 + cluster);
      chol_dense_solve(Mfac[c], 6 * (v1 - v0), rin + 6 * (size_t)v0, zout + 6 * (size_t)v0);
    };
    if (!pool) { for (int c = 0; c < ncl; ++c) one(c); }
    else {
      const int chunks = pool->width() * 4;
      pool->run(chunks, [&](int ch) { for (int c = (int)((long long)ncl * ch / chunks); c < (int)((long long)ncl * (ch + 1) / chunks); ++c) one(c); });
    }
    if (two_level) {                   // + P (P' A P)^-1 P' r
      std::vector<double> rc(cdim, 0.0), xc(cdim, 0.0);
      for (int v = 0; v < n; ++v) {
        const int a = aggof[(size_t)v];
        const double* Pv = &Pt[(size_t)36 * v];
        for (int q = 0; q < 6; ++q) { double acc = 0; for (int r = 0; r < 6; ++r) acc += Pv[6 * r + q] * rin[6 * (size_t)v + r]; rc[6 * a + q] += acc; }
      }
      chol_dense_solve(Ac, cdim, rc.data(), xc.data());
      for (int v = 0; v < n; ++v) {
        const int a = aggof[(size_t)v];
        const double* Pv = &Pt[(size_t)36 * v];
        for (int r = 0; r < 6; ++r) { double acc = 0; for (int q = 0; q < 6; ++q) acc += Pv[6 * r + q] * xc[6 * a + q]; zout[6 * (size_t)v + r] += acc; }
      }
    }
  };
  std::vector<double> r(b, b + m), z(m), p(m, 0.0), q(m), tmp(m);
  std::fill(x, x + m, 0.0);
  double rho = 1.0;
  auto dot = [&](const double* a_, const double* b_) { double s = 0; for (size_t i = 0; i < m; ++i) s += a_[i] * b_[i]; return s; };
  if (form == 1) {
    std::vector<double> u(m), w(m), mv(m), nv(m), zz(m, 0.0), qq(m, 0.0), sv(m, 0.0);
    apply_M(r.data(), u.data());
    matvec(u.data(), w.data());
    *ok = true;
    double gamma_prev = 0.0, alpha_prev = 0.0, q_prev = 0.0;
    int cnt = 0;
    for (;;) {
      apply_M(w.data(), mv.data());
      const double gamma = dot(r.data(), u.data()), delta = dot(w.data(), u.data());
      double sQ = 0;
      for (size_t i = 0; i < m; ++i) sQ += x[i] * (b[i] + r[i]);
      const double Q1 = -1.0 * sQ;
      if (cnt > 0) {
        const double zeta = cnt * (Q1 - q_prev) / Q1;
        if (zeta < q_tol && cnt >= min_it) break;
        if (cnt >= max_it) break;
      }
      if (gamma == 0.0 || !std::isfinite(gamma)) { *ok = (gamma == 0.0); break; }
      double beta = 0.0;
      if (cnt > 0) {
        beta = gamma / gamma_prev;
        if (beta == 0.0 || !std::isfinite(beta)) { *ok = false; break; }
      }
      const double den = cnt > 0 ? delta - beta * gamma / alpha_prev : delta;
      if (!(den > 0.0) || !std::isfinite(den)) break;     // "matrix is indefinite": x of the previous iteration stands
      const double alpha = gamma / den;
      matvec(mv.data(), nv.data());
      for (size_t i = 0; i < m; ++i) {
        const double zn = nv[i] + beta * zz[i], qn = mv[i] + beta * qq[i], sn = w[i] + beta * sv[i], pn = u[i] + beta * p[i];
        zz[i] = zn; qq[i] = qn; sv[i] = sn; p[i] = pn;
        x[i] += alpha * pn; r[i] -= alpha * sn; u[i] -= alpha * qn; w[i] -= alpha * zn;
      }
      gamma_prev = gamma; alpha_prev = alpha; q_prev = Q1;
      ++cnt;
    }
    if (final_rnorm) *final_rnorm = std::sqrt(dot(r.data(), r.data()));
    return cnt;
  }
  double Q0;
  { double s = 0; for (size_t i = 0; i < m; ++i) s += x[i] * (b[i] + r[i]); Q0 = -1.0 * s; }
  *ok = true;
  int it = 1;
  for (;; ++it) {
    apply_M(r.data(), z.data());
    const double last_rho = rho;
    rho = dot(r.data(), z.data());
    if (rho == 0.0 || !std::isfinite(rho)) { *ok = (rho == 0.0); break; }
    if (it == 1) p = z;
    else {
      const double beta = rho / last_rho;
      if (beta == 0.0 || !std::isfinite(beta)) { *ok = false; break; }
      for (size_t i = 0; i < m; ++i) p[i] = z[i] + beta * p[i];
    }
    matvec(p.data(), q.data());
    const double pq = dot(p.data(), q.data());
    if (pq <= 0.0 || !std::isfinite(pq)) break;  // NO_CONVERGENCE: keep current x
    const double alpha = rho / pq;
    if (!std::isfinite(alpha)) { *ok = false; break; }
    for (size_t i = 0; i < m; ++i) x[i] += alpha * p[i];
    if (residual_reset_period > 0 && it % residual_reset_period == 0) {
      matvec(x, tmp.data());
      for (size_t i = 0; i < m; ++i) r[i] = b[i] - tmp[i];
    } else {
      for (size_t i = 0; i < m; ++i) r[i] -= alpha * q[i];
    }
    double s = 0;
    for (size_t i = 0; i < m; ++i) s += x[i] * (b[i] + r[i]);
    const double Q1 = -1.0 * s;
    const double zeta = it * (Q1 - Q0) / Q1;
    if (zeta < q_tol && it >= min_it) break;
    Q0 = Q1;
    if (it >= max_it) break;
  }
  if (final_rnorm) *final_rnorm = std::sqrt(dot(r.data(), r.data()));
  return it;
}

}  // namespace

// =============================================================================================
// C interface (ctypes)
// =============================================================================================
extern "C" {

struct oracle_options {
  int max_num_iterations;        // 50 (Ceres default); the reference sets 1000 (finial.cpp:535)
  int linear_solver;             // 0 = exact sparse Cholesky (SPARSE_NORMAL_CHOLESKY), 1 = block-Jacobi PCG (CGNR/JACOBI)
  int jacobi_scaling;            // 1
  int max_linear_solver_iterations;  // 500
  int min_linear_solver_iterations;  // 0
  int residual_reset_period;     // 10
  int max_num_consecutive_invalid_steps;  // 5
  int loss_kind;                 // 0 trivial, 1 Huber
  double loss_a;                 // 1.0
  double function_tolerance;     // 1e-6
  double gradient_tolerance;     // 1e-10
  double parameter_tolerance;    // 1e-8
  double initial_trust_region_radius;  // 1e4
  double max_trust_region_radius;      // 1e16
  double min_trust_region_radius;      // 1e-32
  double min_relative_decrease;        // 1e-3
  double min_lm_diagonal;              // 1e-6
  double max_lm_diagonal;              // 1e32
  double eta;                          // 0.1
  int pcg_cluster;                     // poses per Jacobi block of the PCG preconditioner (1 = 6x6 blocks, Ceres JACOBI-like); -1: the odometry chain (block-tridiagonal part of H + D^2) solved exactly
  int num_threads;   /* 0 / 1: sequential (the reference sets num_threads = 1); > 1: Jacobian evaluation, cost evaluation and the
                        numeric Cholesky on that many threads — same results to the bit */
  int pcg_form;      /* 0: Ceres' CG statement by statement; 1: the pipelined recurrences (pcg_solve above) */
  int reserved;
};

struct oracle_summary {
  int termination_type;  // 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE
  int num_successful_steps;
  int num_unsuccessful_steps;
  int num_iterations;            // entries in the trace (iteration 0 included)
  int num_linear_iterations;     // CG iterations summed over the solve
  int reason;                    // 1 function tol, 2 parameter tol, 3 gradient tol, 4 min radius, 5 max iterations, 6 invalid steps, 7 linear solver failure
  double initial_cost;
  double final_cost;
  double total_seconds;
  double linear_solver_seconds;
  double jacobian_seconds;
  double cost_eval_seconds;
  long long factor_nnz_blocks;
  double factor_flops;
};

// trace row: [iteration, cost, cost_change, gradient_max_norm, step_norm, relative_decrease,
//             trust_region_radius, linear_iterations, step_is_successful]
enum { ORACLE_TRACE_COLS = 9 };

// Row shards for the two-level PCG (pcg_cluster <= -8): cuts[0 .. world], cuts[0] = 0, cuts[world] = number of poses — aggregates are then
// formed inside every segment [cuts[r], cuts[r + 1]) as the product's sharded solve forms them (csrc/pgo_coarse.h: they never straddle
// ranks).  world <= 0 (or a mismatch with the problem's size): one segment.  Process-wide, like the checker it serves.
void oracle_set_coarse_cuts(const long long* cuts, int world) {
  g_coarse_cuts.clear();
  if (cuts && world > 0) g_coarse_cuts.assign(cuts, cuts + world + 1);
}
void oracle_default_options(oracle_options* o) {
  o->max_num_iterations = 50;
  o->linear_solver = 0;
  o->jacobi_scaling = 1;
  o->max_linear_solver_iterations = 500;
  o->min_linear_solver_iterations = 0;
  o->residual_reset_period = 10;
  o->max_num_consecutive_invalid_steps = 5;
  o->loss_kind = 1;
  o->loss_a = 1.0;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->eta = 0.1;
  o->pcg_cluster = 1;
  o->num_threads = 0;
  o->pcg_form = 0;
  o->reserved = 0;
}

void oracle_edge_eval_autodiff(const double* pa, const double* qa, const double* pb, const double* qb,
                               const double* mp, const double* mq, const double* L, double* r,
                               double* Ja, double* Jb) {
  edge_eval_autodiff(pa, qa, pb, qb, mp, mq, L ? L : kIdentity6, r, Ja, Jb);
}
void oracle_edge_eval_analytic(const double* pa, const double* qa, const double* pb, const double* qb,
                               const double* mp, const double* mq, const double* L, double* r,
                               double* Ja, double* Jb) {
  edge_eval_analytic(pa, qa, pb, qb, mp, mq, L ? L : kIdentity6, r, Ja, Jb);
}
void oracle_quat_plus(const double* q, const double* delta, double* out) { quat_plus(q, delta, out); }
void oracle_loss(int kind, double a, double s, double* rho3) { loss_eval(kind, a, s, rho3); }

// lower Cholesky factor of a 6x6 information matrix (finial.cpp:508 information.llt().matrixL())
int oracle_chol6(const double* info, double* L) {
  for (int i = 0; i < 36; ++i) L[i] = info[i];
  return chol6(L) ? 0 : -1;
}

// Per-edge evaluation with loss correction and constant-block masking, as the evaluator sees it.
// residuals E x 6, Ja/Jb E x 36 (may be NULL), rho E x 3 (may be NULL). Returns total cost.
double oracle_evaluate(int N, int E, const double* poses, const uint8_t* cmask, const int* ia,
                       const int* ib, const double* meas, const double* sqrt_info, int loss_kind,
                       double loss_a, double* residuals, double* Ja, double* Jb) {
  Prob P{N, E, cmask, ia, ib, meas, sqrt_info, loss_kind, loss_a};
  double cost = 0, r[6], A[36], B[36];
  for (int e = 0; e < E; ++e) {
    const bool wj = (Ja != nullptr) || (residuals != nullptr);
    cost += edge_linearize(P, poses, e, r, A, B, wj);
    if (residuals) std::memcpy(residuals + 6 * (size_t)e, r, sizeof r);
    if (Ja) std::memcpy(Ja + 36 * (size_t)e, A, sizeof A);
    if (Jb) std::memcpy(Jb + 36 * (size_t)e, B, sizeof B);
  }
  return cost;
}

double oracle_cost(int N, int E, const double* poses, const uint8_t* cmask, const int* ia, const int* ib,
                   const double* meas, const double* sqrt_info, int loss_kind, double loss_a) {
  Prob P{N, E, cmask, ia, ib, meas, sqrt_info, loss_kind, loss_a};
  return total_cost(P, poses);
}

// Dense normal equations for small problems: Hdense (6N x 6N, row-major, full symmetric), g (6N).
double oracle_normal_equations_dense(int N, int E, const double* poses, const uint8_t* cmask,
                                     const int* ia, const int* ib, const double* meas,
                                     const double* sqrt_info, int loss_kind, double loss_a,
                                     double* Hdense, double* g_out) {
  Prob P{N, E, cmask, ia, ib, meas, sqrt_info, loss_kind, loss_a};
  BlockSym H;
  build_structure(P, H);
  std::vector<double> g;
  const double cost = linearize(P, poses, H, g);
  const size_t m = (size_t)6 * N;
  std::fill(Hdense, Hdense + m * m, 0.0);
  for (int j = 0; j < N; ++j)
    for (int p = H.colptr[j]; p < H.colptr[j + 1]; ++p) {
      const int i = H.rowidx[p];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
          Hdense[(6 * (size_t)i + r) * m + 6 * (size_t)j + c] = H.val[p][6 * r + c];
          Hdense[(6 * (size_t)j + c) * m + 6 * (size_t)i + r] = H.val[p][6 * r + c];
        }
    }
  std::memcpy(g_out, g.data(), m * sizeof(double));
  return cost;
}

// Solve (H + diag(d2)) x = b on the current linearisation with the exact factorisation or PCG
// (for solver-level tests).  Returns CG iterations (0 for the direct solver), <0 on failure.
int oracle_linear_solve(int N, int E, const double* poses, const uint8_t* cmask, const int* ia,
                        const int* ib, const double* meas, const double* sqrt_info, int loss_kind,
                        double loss_a, const double* d2, const double* b, int linear_solver,
                        double q_tol, int max_it, double* x) {
  Prob P{N, E, cmask, ia, ib, meas, sqrt_info, loss_kind, loss_a};
  BlockSym H;
  build_structure(P, H);
  std::vector<double> g;
  linearize(P, poses, H, g);
  if (linear_solver == 0) {
    SparseChol C;
    C.analyze(H);
    if (!C.factor(H, d2)) return -1;
    C.solve(b, x);
    return 0;
  }
  bool ok;
  // linear_solver: 1 = 6x6 Jacobi blocks, 100 + c = clusters of c poses, 99 = the odometry chain solved exactly (pcg_cluster -1);
  // + 1000 = the pipelined recurrences
  const int form = linear_solver >= 1000 ? 1 : 0;
  if (form) linear_solver -= 1000;
  int it = pcg_solve(H, d2, b, x, q_tol, max_it, 0, 10, &ok, nullptr, linear_solver == 99 ? -1 : linear_solver >= 100 ? linear_solver - 100 : 1, nullptr, form);
  return ok ? it : -1;
}

// ---------------------------------------------------------------------------------------------
// ceres::Solve restated: TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy [Ceres 1.13],
// SURVEY.md Appendix A.6 step by step.  poses (N x 7) are updated in place, like the parameter
// blocks the reference hands to Problem (finial.cpp:513-517).
// ---------------------------------------------------------------------------------------------
int oracle_solve(int N, int E, double* poses, const uint8_t* cmask, const int* ia, const int* ib,
                 const double* meas, const double* sqrt_info, const oracle_options* opt,
                 oracle_summary* sum, double* trace, int trace_capacity) {
  typedef std::chrono::steady_clock Clock;
  auto secs = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto t_start = Clock::now();
  Prob P{N, E, cmask, ia, ib, meas, sqrt_info, opt->loss_kind, opt->loss_a};
  const size_t m = (size_t)6 * N;
  std::memset(sum, 0, sizeof *sum);

  BlockSym H;
  build_structure(P, H);
  SparseChol chol;
  if (opt->linear_solver == 0) chol.analyze(H);
  std::unique_ptr<Pool> pool;
  if (opt->num_threads > 1) pool.reset(new Pool(opt->num_threads));
  std::vector<EdgeLin> mt_edges;
  std::vector<double> mt_terms;
  auto do_linearize = [&](const double* xx, std::vector<double>& gg) { return pool ? linearize_mt(P, xx, H, gg, *pool, mt_edges) : linearize(P, xx, H, gg); };
  auto do_cost = [&](const double* xx) { return pool ? total_cost_mt(P, xx, *pool, mt_terms) : total_cost(P, xx); };

  std::vector<double> x(poses, poses + 7 * (size_t)N), cand(7 * (size_t)N);
  std::vector<double> g, scale(m, 1.0), diag(m), d2(m), gs(m), step(m), delta(m), tmp(m);
  double x_cost = 0, radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int n_trace = 0;

  auto is_const = [&](int v, int i) { return (i < 3) ? (cmask[v] & 1) != 0 : (cmask[v] & 2) != 0; };
  auto x_norm_of = [&](const std::vector<double>& xx) {
    double s = 0;
    for (int v = 0; v < N; ++v) {
      if (!(cmask[v] & 1)) for (int i = 0; i < 3; ++i) s += xx[7 * (size_t)v + i] * xx[7 * (size_t)v + i];
      if (!(cmask[v] & 2)) for (int i = 3; i < 7; ++i) s += xx[7 * (size_t)v + i] * xx[7 * (size_t)v + i];
    }
    return std::sqrt(s);
  };
  auto plus = [&](const std::vector<double>& xx, const double* dl, std::vector<double>& out) {
    for (int v = 0; v < N; ++v) {
      const double* xv = &xx[7 * (size_t)v];
      double* ov = &out[7 * (size_t)v];
      if (cmask[v] & 1) { ov[0] = xv[0]; ov[1] = xv[1]; ov[2] = xv[2]; }
      else for (int i = 0; i < 3; ++i) ov[i] = xv[i] + dl[6 * (size_t)v + i];
      if (cmask[v] & 2) { ov[3] = xv[3]; ov[4] = xv[4]; ov[5] = xv[5]; ov[6] = xv[6]; }
      else quat_plus(xv + 3, dl + 6 * (size_t)v + 3, ov + 3);
    }
  };
  double gradient_max_norm = 0;
  bool scaled_once = false;
  // EvaluateGradientAndJacobian: H~ = S H S, g (unscaled) kept for the gradient test, gs = S g
  auto evaluate_gradient_and_jacobian = [&]() {
    const auto t0 = Clock::now();
    x_cost = do_linearize(x.data(), g);
    if (opt->jacobi_scaling) {
      if (!scaled_once) {
        for (int v = 0; v < N; ++v)
          for (int i = 0; i < 6; ++i) {
            const double cn = is_const(v, i) ? 0.0 : H.val[H.colptr[v]][7 * i];
            scale[6 * (size_t)v + i] = 1.0 / (1.0 + std::sqrt(cn));
          }
        scaled_once = true;
      }
      for (int j = 0; j < N; ++j)
        for (int p = H.colptr[j]; p < H.colptr[j + 1]; ++p) {
          const int i = H.rowidx[p];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c)
              H.val[p][6 * r + c] *= scale[6 * (size_t)i + r] * scale[6 * (size_t)j + c];
        }
      // keep the unit diagonal of constant dims exactly 1
      for (int v = 0; v < N; ++v)
        for (int i = 0; i < 6; ++i) if (is_const(v, i)) H.val[H.colptr[v]][7 * i] = 1.0;
    }
    for (size_t i = 0; i < m; ++i) gs[i] = g[i] * scale[i];
    // gradient_max_norm = || x - Plus(x, -g) ||_inf over the non-constant blocks
    for (size_t i = 0; i < m; ++i) tmp[i] = -g[i];
    plus(x, tmp.data(), cand);
    double gm = 0;
    for (int v = 0; v < N; ++v) {
      if (!(cmask[v] & 1)) for (int i = 0; i < 3; ++i) gm = std::max(gm, std::fabs(x[7 * (size_t)v + i] - cand[7 * (size_t)v + i]));
      if (!(cmask[v] & 2)) for (int i = 3; i < 7; ++i) gm = std::max(gm, std::fabs(x[7 * (size_t)v + i] - cand[7 * (size_t)v + i]));
    }
    gradient_max_norm = gm;
    sum->jacobian_seconds += secs(t0, Clock::now());
  };

  struct Iter { int iteration; double cost, cost_change, gmax, step_norm, rel_dec, radius; int lin_it; bool ok, valid; } it{};
  auto push_trace = [&]() {
    if (trace && n_trace < trace_capacity) {
      double* t = trace + (size_t)ORACLE_TRACE_COLS * n_trace;
      t[0] = it.iteration; t[1] = it.cost; t[2] = it.cost_change; t[3] = it.gmax; t[4] = it.step_norm;
      t[5] = it.rel_dec; t[6] = it.radius; t[7] = it.lin_it; t[8] = it.ok ? 1.0 : 0.0;
    }
    ++n_trace;
  };

  // ---- Init + IterationZero ----
  double x_norm = x_norm_of(x);
  evaluate_gradient_and_jacobian();
  sum->initial_cost = x_cost;
  it = Iter{0, x_cost, 0, gradient_max_norm, 0, 0, radius, 0, true, true};
  int num_consecutive_invalid = 0;
  int term = 1, reason = 5;

  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue ----
    if (it.ok) ++sum->num_successful_steps; else ++sum->num_unsuccessful_steps;
    it.radius = radius;
    push_trace();
    if (it.iteration >= opt->max_num_iterations) { term = 1; reason = 5; break; }
    if (it.ok && it.gmax <= opt->gradient_tolerance) { term = 0; reason = 3; break; }
    if (it.radius <= opt->min_trust_region_radius) { term = 0; reason = 4; break; }

    Iter nx{};
    nx.iteration = it.iteration + 1;
    nx.gmax = it.gmax;

    // ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep) ----
    const auto t_lin = Clock::now();
    if (!reuse_diagonal) {
      for (int v = 0; v < N; ++v)
        for (int i = 0; i < 6; ++i) {
          double dv = H.val[H.colptr[v]][7 * i];
          diag[6 * (size_t)v + i] = std::min(std::max(dv, opt->min_lm_diagonal), opt->max_lm_diagonal);
        }
    }
    for (size_t i = 0; i < m; ++i) d2[i] = diag[i] / radius;  // lm_diagonal^2
    bool lin_ok = true;
    int lin_it = 0;
    if (opt->linear_solver == 0) {
      lin_ok = chol.factor(H, d2.data(), pool.get());
      if (lin_ok) chol.solve(gs.data(), step.data());
      sum->factor_nnz_blocks = chol.nnz_blocks;
      sum->factor_flops = chol.flops;
    } else {
      const CoarseSpace cz{-opt->pcg_cluster, x.data(), scale.data(), cmask};
      lin_it = pcg_solve(H, d2.data(), gs.data(), step.data(), opt->eta, opt->max_linear_solver_iterations,
                         opt->min_linear_solver_iterations, opt->residual_reset_period, &lin_ok, nullptr, opt->pcg_cluster, pool.get(), opt->pcg_form,
                         opt->pcg_cluster <= -8 ? &cz : nullptr);
      sum->num_linear_iterations += lin_it;
    }
    if (lin_ok) for (size_t i = 0; i < m; ++i) { if (!std::isfinite(step[i])) { lin_ok = false; break; } }
    for (size_t i = 0; i < m; ++i) step[i] = -step[i];
    reuse_diagonal = true;
    sum->linear_solver_seconds += secs(t_lin, Clock::now());
    nx.lin_it = lin_it;

    double model_cost_change = 0;
    bool step_valid = false;
    if (lin_ok) {
      // model_cost_change = -(J~ step)'(r + J~ step / 2) = -step' g~ - step' H~ step / 2
      sym_matvec(H, nullptr, step.data(), tmp.data());
      double a = 0, b = 0;
      for (int v = 0; v < N; ++v)
        for (int i = 0; i < 6; ++i) {
          const size_t k = 6 * (size_t)v + i;
          if (is_const(v, i)) continue;
          a += step[k] * gs[k];
          b += step[k] * tmp[k];
        }
      model_cost_change = -a - 0.5 * b;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      // ---- HandleInvalidStep ----
      ++num_consecutive_invalid;
      if (num_consecutive_invalid >= opt->max_num_consecutive_invalid_steps) { term = 2; reason = 6; it = nx; break; }
      radius *= 0.5;  // LevenbergMarquardtStrategy::StepIsInvalid
      reuse_diagonal = true;
      nx.cost = x_cost; nx.cost_change = 0; nx.step_norm = 0; nx.rel_dec = 0; nx.ok = false; nx.valid = false;
      it = nx;
      continue;
    }
    num_consecutive_invalid = 0;
    for (size_t i = 0; i < m; ++i) delta[i] = step[i] * scale[i];

    // ---- ComputeCandidatePointAndEvaluateCost ----
    const auto t_c = Clock::now();
    plus(x, delta.data(), cand);
    const double cand_cost = do_cost(cand.data());
    sum->cost_eval_seconds += secs(t_c, Clock::now());

    // ---- ParameterToleranceReached ----
    {
      double s = 0;
      for (int v = 0; v < N; ++v) {
        if (!(cmask[v] & 1)) for (int i = 0; i < 3; ++i) { const double d = x[7 * (size_t)v + i] - cand[7 * (size_t)v + i]; s += d * d; }
        if (!(cmask[v] & 2)) for (int i = 3; i < 7; ++i) { const double d = x[7 * (size_t)v + i] - cand[7 * (size_t)v + i]; s += d * d; }
      }
      nx.step_norm = std::sqrt(s);
    }
    if (nx.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { term = 0; reason = 2; it = nx; it.cost = x_cost; break; }
    // ---- FunctionToleranceReached ----
    nx.cost_change = x_cost - cand_cost;
    if (std::fabs(nx.cost_change) <= opt->function_tolerance * x_cost) { term = 0; reason = 1; it = nx; it.cost = x_cost; break; }
    // ---- IsStepSuccessful ----
    nx.rel_dec = nx.cost_change / model_cost_change;
    if (nx.rel_dec > opt->min_relative_decrease) {
      // ---- HandleSuccessfulStep ----
      x.swap(cand);
      x_norm = x_norm_of(x);
      evaluate_gradient_and_jacobian();
      nx.ok = true; nx.valid = true; nx.cost = x_cost; nx.gmax = gradient_max_norm;
      // LevenbergMarquardtStrategy::StepAccepted
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * nx.rel_dec - 1.0, 3));
      radius = std::min(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      // ---- HandleUnsuccessfulStep / StepRejected ----
      nx.ok = false; nx.valid = true; nx.cost = cand_cost;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
    it = nx;
  }

  std::memcpy(poses, x.data(), sizeof(double) * 7 * (size_t)N);
  sum->termination_type = term;
  sum->reason = reason;
  sum->final_cost = x_cost;
  sum->num_iterations = n_trace;
  sum->total_seconds = secs(t_start, Clock::now());
  return term == 2 ? -1 : 0;
}

// Timed pieces for bench.py's cpu_baseline: one Jacobian evaluation sweep (residual + analytic
// Jacobians + corrector for every edge), returns seconds for `repeats` sweeps.
double oracle_time_jacobian_eval(int N, int E, const double* poses, const uint8_t* cmask, const int* ia,
                                 const int* ib, const double* meas, const double* sqrt_info, int loss_kind,
                                 double loss_a, int repeats, double* checksum) {
  Prob P{N, E, cmask, ia, ib, meas, sqrt_info, loss_kind, loss_a};
  double r[6], A[36], B[36], acc = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < repeats; ++k)
    for (int e = 0; e < E; ++e) { acc += edge_linearize(P, poses, e, r, A, B, true); acc += A[7] + B[14]; }
  const auto t1 = std::chrono::steady_clock::now();
  if (checksum) *checksum = acc;
  return std::chrono::duration<double>(t1 - t0).count();
}


// ---------------------------------------------------------------------------------------------
// MotionEstimate: the small dense Ceres problem of REF/src/MotionEstimate.cc:71-129 —
// one quaternion block (EigenQuaternionParameterization; the reference sets it CONSTANT, :108) + one translation block,
// n reprojection residual blocks with HuberLoss(1.0) sharing them, max_num_iterations = 1000, SPARSE_NORMAL_CHOLESKY
// (= an exact solve of the 6x6 / 3x3 normal equations).  Same trust-region rules as oracle_solve above.
// cmask bit0: t constant, bit1: q constant.  trace rows as oracle_solve (ORACLE_TRACE_COLS).
// ---------------------------------------------------------------------------------------------
void oracle_reproj_eval(int n, const double* points, const double* obs, const double* intr, const double* q, const double* t,
                        double* residuals, double* jacobians) {
  for (int i = 0; i < n; ++i) reproj_eval_point(intr, obs + 2 * i, q, t, points + 3 * i, residuals + 2 * i, jacobians + 12 * i);
}

int oracle_reproj_solve(int n, const double* points, const double* obs, const double* intr, double* q, double* t, int cmask,
                        const oracle_options* opt, oracle_summary* sum, double* trace, int trace_capacity) {
  std::memset(sum, 0, sizeof *sum);
  const bool t_const = (cmask & 1) != 0, q_const = (cmask & 2) != 0;
  auto is_const = [&](int i) { return i < 3 ? q_const : t_const; };   // local dims: [dtheta(3) | dt(3)]
  double x[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]}, cand[7];
  double H[36], g[6], scale[6] = {1, 1, 1, 1, 1, 1}, diag[6], gs[6], step[6], delta[6];
  double x_cost = 0, radius = opt->initial_trust_region_radius, decrease_factor = 2.0, gradient_max_norm = 0;
  bool reuse_diagonal = false, scaled_once = false;
  int n_trace = 0;

  auto cost_at = [&](const double* xx) {
    double c = 0;
    for (int i = 0; i < n; ++i) {
      double r[2], J[12], rho[3];
      reproj_eval_point(intr, obs + 2 * i, xx, xx + 4, points + 3 * i, r, J);
      loss_eval(opt->loss_kind, opt->loss_a, r[0] * r[0] + r[1] * r[1], rho);
      c += 0.5 * rho[0];
    }
    return c;
  };
  auto plus = [&](const double* xx, const double* dl, double* out) {
    if (q_const) { for (int i = 0; i < 4; ++i) out[i] = xx[i]; } else quat_plus(xx, dl, out);
    for (int i = 0; i < 3; ++i) out[4 + i] = t_const ? xx[4 + i] : xx[4 + i] + dl[3 + i];
  };
  auto x_norm_of = [&](const double* xx) {
    double s = 0;
    if (!q_const) for (int i = 0; i < 4; ++i) s += xx[i] * xx[i];
    if (!t_const) for (int i = 4; i < 7; ++i) s += xx[i] * xx[i];
    return std::sqrt(s);
  };
  auto evaluate_gradient_and_jacobian = [&]() {
    for (int k = 0; k < 36; ++k) H[k] = 0;
    for (int k = 0; k < 6; ++k) g[k] = 0;
    double c = 0;
    for (int i = 0; i < n; ++i) {
      double r[2], J[12], rho[3];
      reproj_eval_point(intr, obs + 2 * i, x, x + 4, points + 3 * i, r, J);
      loss_eval(opt->loss_kind, opt->loss_a, r[0] * r[0] + r[1] * r[1], rho);
      c += 0.5 * rho[0];
      const double w = rho[1];          // corrector with rho'' <= 0: residual and Jacobian scaled by sqrt(rho')
      for (int a = 0; a < 2; ++a)
        for (int u = 0; u < 6; ++u) {
          if (is_const(u)) continue;
          g[u] += w * J[6 * a + u] * r[a];
          for (int v = 0; v < 6; ++v) if (!is_const(v)) H[6 * u + v] += w * J[6 * a + u] * J[6 * a + v];
        }
    }
    for (int u = 0; u < 6; ++u) if (is_const(u)) H[7 * u] = 1.0;
    x_cost = c;
    if (opt->jacobi_scaling) {
      if (!scaled_once) {
        for (int u = 0; u < 6; ++u) scale[u] = 1.0 / (1.0 + std::sqrt(is_const(u) ? 0.0 : H[7 * u]));
        scaled_once = true;
      }
      for (int u = 0; u < 6; ++u) for (int v = 0; v < 6; ++v) H[6 * u + v] *= scale[u] * scale[v];
      for (int u = 0; u < 6; ++u) if (is_const(u)) H[7 * u] = 1.0;
    }
    for (int u = 0; u < 6; ++u) gs[u] = g[u] * scale[u];
    double neg[6];
    for (int u = 0; u < 6; ++u) neg[u] = -g[u];
    plus(x, neg, cand);
    double gm = 0;
    if (!q_const) for (int i = 0; i < 4; ++i) gm = std::max(gm, std::fabs(x[i] - cand[i]));
    if (!t_const) for (int i = 4; i < 7; ++i) gm = std::max(gm, std::fabs(x[i] - cand[i]));
    gradient_max_norm = gm;
  };
  struct Iter { int iteration; double cost, cost_change, gmax, step_norm, rel_dec, radius; bool ok; } it{};
  auto push_trace = [&]() {
    if (trace && n_trace < trace_capacity) {
      double* tr = trace + (size_t)ORACLE_TRACE_COLS * n_trace;
      tr[0] = it.iteration; tr[1] = it.cost; tr[2] = it.cost_change; tr[3] = it.gmax; tr[4] = it.step_norm;
      tr[5] = it.rel_dec; tr[6] = it.radius; tr[7] = 0; tr[8] = it.ok ? 1.0 : 0.0;
    }
    ++n_trace;
  };
  double x_norm = x_norm_of(x);
  evaluate_gradient_and_jacobian();
  sum->initial_cost = x_cost;
  it = Iter{0, x_cost, 0, gradient_max_norm, 0, 0, radius, true};
  int num_consecutive_invalid = 0, term = 1, reason = 5;
  for (;;) {
    if (it.ok) ++sum->num_successful_steps; else ++sum->num_unsuccessful_steps;
    it.radius = radius;
    push_trace();
    if (it.iteration >= opt->max_num_iterations) { term = 1; reason = 5; break; }
    if (it.ok && it.gmax <= opt->gradient_tolerance) { term = 0; reason = 3; break; }
    if (it.radius <= opt->min_trust_region_radius) { term = 0; reason = 4; break; }
    Iter nx{};
    nx.iteration = it.iteration + 1;
    nx.gmax = it.gmax;
    if (!reuse_diagonal) for (int u = 0; u < 6; ++u) diag[u] = std::min(std::max(H[7 * u], opt->min_lm_diagonal), opt->max_lm_diagonal);
    std::vector<double> A(H, H + 36);
    for (int u = 0; u < 6; ++u) A[7 * u] += diag[u] / radius;
    bool lin_ok = chol_dense(A, 6);
    if (lin_ok) chol_dense_solve(A, 6, gs, step);
    if (lin_ok) for (int u = 0; u < 6; ++u) if (!std::isfinite(step[u])) lin_ok = false;
    for (int u = 0; u < 6; ++u) step[u] = -step[u];
    reuse_diagonal = true;
    double model_cost_change = 0;
    bool step_valid = false;
    if (lin_ok) {
      double a = 0, b = 0;
      for (int u = 0; u < 6; ++u) {
        if (is_const(u)) continue;
        double hs = 0;
        for (int v = 0; v < 6; ++v) hs += H[6 * u + v] * step[v];
        a += step[u] * gs[u];
        b += step[u] * hs;
      }
      model_cost_change = -a - 0.5 * b;
      step_valid = model_cost_change > 0.0;
    }
    if (!step_valid) {
      ++num_consecutive_invalid;
      if (num_consecutive_invalid >= opt->max_num_consecutive_invalid_steps) { term = 2; reason = 6; it = nx; break; }
      radius *= 0.5;
      nx.cost = x_cost; nx.ok = false;
      it = nx;
      continue;
    }
    num_consecutive_invalid = 0;
    for (int u = 0; u < 6; ++u) delta[u] = is_const(u) ? 0.0 : step[u] * scale[u];
    plus(x, delta, cand);
    const double cand_cost = cost_at(cand);
    {
      double sq = 0;
      if (!q_const) for (int i = 0; i < 4; ++i) sq += (x[i] - cand[i]) * (x[i] - cand[i]);
      if (!t_const) for (int i = 4; i < 7; ++i) sq += (x[i] - cand[i]) * (x[i] - cand[i]);
      nx.step_norm = std::sqrt(sq);
    }
    if (nx.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { term = 0; reason = 2; it = nx; it.cost = x_cost; break; }
    nx.cost_change = x_cost - cand_cost;
    if (std::fabs(nx.cost_change) <= opt->function_tolerance * x_cost) { term = 0; reason = 1; it = nx; it.cost = x_cost; break; }
    nx.rel_dec = nx.cost_change / model_cost_change;
    if (nx.rel_dec > opt->min_relative_decrease) {
      for (int i = 0; i < 7; ++i) x[i] = cand[i];
      x_norm = x_norm_of(x);
      evaluate_gradient_and_jacobian();
      nx.ok = true; nx.cost = x_cost; nx.gmax = gradient_max_norm;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * nx.rel_dec - 1.0, 3));
      radius = std::min(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      nx.ok = false; nx.cost = cand_cost;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
    it = nx;
  }
  for (int i = 0; i < 4; ++i) q[i] = x[i];
  for (int i = 0; i < 3; ++i) t[i] = x[4 + i];
  sum->termination_type = term;
  sum->reason = reason;
  sum->final_cost = x_cost;
  sum->num_iterations = n_trace;
  return term == 2 ? -1 : 0;
}

}  // extern "C"
