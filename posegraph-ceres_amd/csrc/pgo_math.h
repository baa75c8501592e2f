// pgo_math.h — device/host FP64 helpers for the SE(3) between-factor of TurtleZhong/PoseGraph-Ceres.
//
// Written from the math in SURVEY.md Appendix A (not from Ceres/Eigen source):
//   residual   : PLUS/include/PoseGraph3dError.h:21-54
//   retraction : EigenQuaternionParameterization (finial.cpp:496-497; in-tree statement
//                src/other_projects/bundle_adjustment/ceres_extensions.h:25-42)
// Quaternions are Hamilton, stored x,y,z,w.
#pragma once
#include <hip/hip_runtime.h>

#define PGO_HD __host__ __device__ __forceinline__

namespace pgo {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };
struct M3 { double m[9]; };  // row-major

PGO_HD V3 cross(const V3& a, const V3& b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PGO_HD double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PGO_HD Q4 qmul(const Q4& a, const Q4& b) {
  Q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
PGO_HD Q4 qconj(const Q4& q) { return Q4{-q.x, -q.y, -q.z, q.w}; }

// C = A * B, C = A^T * B, 3x3
PGO_HD M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
PGO_HD M3 mulT(const M3& A, const M3& B) {  // A^T B
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
  return C;
}
PGO_HD V3 mulTv(const M3& A, const V3& v) {  // A^T v
  return V3{A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z,
            A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z};
}
PGO_HD V3 mulv(const M3& A, const V3& v) {
  return V3{A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
            A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
PGO_HD M3 transpose(const M3& A) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * j + i];
  return C;
}
PGO_HD M3 axpby(double a, const M3& X, double b, const M3& Y) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 9; ++i) C.m[i] = a * X.m[i] + b * Y.m[i];
  return C;
}
PGO_HD M3 scaled(double a, const M3& X) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 9; ++i) C.m[i] = a * X.m[i];
  return C;
}

// The geometric part of one edge (a = id_begin, b = id_end), before whitening / loss:
//   e  = [ T(conj q_a)(p_b - p_a) - p_hat ; 2 vec(q_hat (x) conj(q_b) (x) q_a) ]
//   A_a = [ -Rt  G ; 0  2M ]      A_b = [ Rt  0 ; 0  -2M ]        (d e / d local tangent)
// T(c)v = v + 2 c_w (c_v x v) + 2 c_v x (c_v x v) is Eigen's unit-quaternion rotation formula; Rt is
// its matrix and G its derivative w.r.t. the left (world-side) half-angle perturbation of q_a, so
// the result equals AutoDiff(functor) * PlusJacobian also when |q_a| != 1 to rounding.
struct EdgeGeom {
  double e[6];
  M3 Rt, G, M;
};

PGO_HD EdgeGeom edge_geometry(const V3& pa, const Q4& qa, const V3& pb, const Q4& qb, const V3& mp,
                              const Q4& mq) {
  EdgeGeom g;
  const V3 u{-qa.x, -qa.y, -qa.z};
  const double w = qa.w;
  const V3 d{pb.x - pa.x, pb.y - pa.y, pb.z - pa.z};
  const double uu = dot(u, u);
  // Rt = (1 - 2 u.u) I + 2 u u^T + 2 w [u]x
  g.Rt.m[0] = 1.0 - 2.0 * uu + 2.0 * u.x * u.x;
  g.Rt.m[1] = 2.0 * u.x * u.y - 2.0 * w * u.z;
  g.Rt.m[2] = 2.0 * u.x * u.z + 2.0 * w * u.y;
  g.Rt.m[3] = 2.0 * u.y * u.x + 2.0 * w * u.z;
  g.Rt.m[4] = 1.0 - 2.0 * uu + 2.0 * u.y * u.y;
  g.Rt.m[5] = 2.0 * u.y * u.z - 2.0 * w * u.x;
  g.Rt.m[6] = 2.0 * u.z * u.x - 2.0 * w * u.y;
  g.Rt.m[7] = 2.0 * u.z * u.y + 2.0 * w * u.x;
  g.Rt.m[8] = 1.0 - 2.0 * uu + 2.0 * u.z * u.z;
  const V3 ud = cross(u, d);
  const V3 uud = cross(u, ud);
  g.e[0] = d.x + 2.0 * w * ud.x + 2.0 * uud.x - mp.x;
  g.e[1] = d.y + 2.0 * w * ud.y + 2.0 * uud.y - mp.y;
  g.e[2] = d.z + 2.0 * w * ud.z + 2.0 * uud.z - mp.z;
  // G[:,k]: q_a <- [e_k;1](x)q_a  =>  du = -w e_k + e_k x u ,  dw = e_k . u   (u = -vec q_a)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const V3 ek{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
    const V3 exu = cross(ek, u);
    const V3 du{-w * ek.x + exu.x, -w * ek.y + exu.y, -w * ek.z + exu.z};
    const double dw = dot(ek, u);
    const V3 dud = cross(du, d);
    const V3 t1 = cross(du, ud);
    const V3 t2 = cross(u, dud);
    g.G.m[0 + k] = 2.0 * (dw * ud.x + w * dud.x + t1.x + t2.x);
    g.G.m[3 + k] = 2.0 * (dw * ud.y + w * dud.y + t1.y + t2.y);
    g.G.m[6 + k] = 2.0 * (dw * ud.z + w * dud.z + t1.z + t2.z);
  }
  const Q4 A = qmul(mq, qconj(qb));
  const Q4 dq = qmul(A, qa);
  g.e[3] = 2.0 * dq.x;
  g.e[4] = 2.0 * dq.y;
  g.e[5] = 2.0 * dq.z;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const Q4 ek{k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0, 0.0};
    const Q4 t = qmul(A, qmul(ek, qa));
    g.M.m[0 + k] = t.x;
    g.M.m[3 + k] = t.y;
    g.M.m[6 + k] = t.z;
  }
  return g;
}

// residual only (cost evaluation)
PGO_HD void edge_error(const V3& pa, const Q4& qa, const V3& pb, const Q4& qb, const V3& mp, const Q4& mq,
                       double* e) {
  const V3 u{-qa.x, -qa.y, -qa.z};
  const double w = qa.w;
  const V3 d{pb.x - pa.x, pb.y - pa.y, pb.z - pa.z};
  V3 ud = cross(u, d);
  ud = V3{ud.x + ud.x, ud.y + ud.y, ud.z + ud.z};
  const V3 c = cross(u, ud);
  e[0] = d.x + w * ud.x + c.x - mp.x;
  e[1] = d.y + w * ud.y + c.y - mp.y;
  e[2] = d.z + w * ud.z + c.z - mp.z;
  const Q4 dq = qmul(mq, qmul(qconj(qb), qa));
  e[3] = 2.0 * dq.x;
  e[4] = 2.0 * dq.y;
  e[5] = 2.0 * dq.z;
}

// ceres::LossFunction::Evaluate for the kinds of include/pgo.h [Ceres 1.13 definitions; SURVEY.md A.4]: rho, rho'.
// Every kind here has rho'' <= 0, so Ceres' corrector reduces to scaling r and J by sqrt(rho') (alpha = 0 branch).
PGO_HD void loss_eval(int kind, double a, double s, double* rho0, double* rho1) {
  const double tiny = 2.2250738585072014e-308;
  if (kind == 1) {                       // HuberLoss(a)
    const double b = a * a;
    if (s > b) {
      const double r = sqrt(s);
      *rho0 = 2.0 * a * r - b;
      const double q = a / r;
      *rho1 = q > tiny ? q : tiny;
      return;
    }
  } else if (kind == 2) {                // SoftLOneLoss(a)
    const double b = a * a, sum = 1.0 + s / b, tmp = sqrt(sum);
    *rho0 = 2.0 * b * (tmp - 1.0);
    const double q = 1.0 / tmp;
    *rho1 = q > tiny ? q : tiny;
    return;
  } else if (kind == 3) {                // CauchyLoss(a)
    const double b = a * a, sum = 1.0 + s / b, inv = 1.0 / sum;
    *rho0 = b * log(sum);
    *rho1 = inv > tiny ? inv : tiny;
    return;
  } else if (kind == 4) {                // ArctanLoss(a)
    const double sum = 1.0 + s * s / (a * a), inv = 1.0 / sum;
    *rho0 = a * atan2(s, a);
    *rho1 = inv > tiny ? inv : tiny;
    return;
  }
  else if (kind == 5) {                  // switchable constraint with the switch eliminated (a = Phi), see include/pgo.h
    const double q = a / (a + s);
    *rho0 = s * q;
    const double w = q * q;
    *rho1 = w > tiny ? w : tiny;
    return;
  }
  *rho0 = s;
  *rho1 = 1.0;
}

// EigenQuaternionParameterization::Plus: q+ = [sin|d| d/|d| ; cos|d|] (x) q, identity if |d| == 0.
PGO_HD Q4 quat_plus(const Q4& q, const V3& dl) {
  const double n = sqrt(dl.x * dl.x + dl.y * dl.y + dl.z * dl.z);
  if (n > 0.0) {
    const double s = sin(n) / n;
    const Q4 t{s * dl.x, s * dl.y, s * dl.z, cos(n)};
    return qmul(t, q);
  }
  return q;
}

// In-register Cholesky based inverse of a symmetric positive definite 6x6 (row-major, full storage).
// Returns false when a pivot is not positive.
PGO_HD bool spd6_inverse(const double* A, double* Ainv) {
  double L[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) L[i] = 0.0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[7 * j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
    if (!(d > 0.0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    L[7 * j] = d;
    const double inv = 1.0 / d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
      L[6 * i + j] = s * inv;
    }
  }
  // Linv (lower) by forward substitution, then Ainv = Linv^T Linv
  double Li[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) Li[i] = 0.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
#pragma unroll
    for (int i = c; i < 6; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = c; k < i; ++k) s -= L[6 * i + k] * Li[6 * k + c];
      Li[6 * i + c] = s / L[7 * i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = i; k < 6; ++k) s += Li[6 * k + i] * Li[6 * k + j];
      Ainv[6 * i + j] = s;
      Ainv[6 * j + i] = s;
    }
  return ok;
}

}  // namespace pgo
