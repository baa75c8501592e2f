// pgo_lin_lean.h — one incidence of the linearisation in its LEAN form, for information matrices without a position / rotation
// coupling (INFO 0 identity, 2 block-diagonal, 3 diagonal: every dataset and benchmark configuration of SURVEY.md section 8).
//
// Same quantities as linearize_body() of pgo_kernels.hip (residual, closed-form Jacobians, loss corrector, J'WJ blocks, J'Wr), with the
// algebra done by hand instead of through general 3x3 helpers:
//   * G = d(Rt d)/d(theta_a) = 2 (Rt + (|q_a|^2 - 1) I) [d]x  (25 operations; the general derivative of the rotation polynomial costs about 150),
//   * M[:,k] = vec(A (x) e_k (x) q_a) with e_k (x) q_a written out as the signed permutation of q_a it is (36 operations),
//   * C1 = Rt' Wpp Rt, MQ = M' Wrr M, GU = G' Wpp G are symmetric: six entries each; W is applied as what it is (a diagonal in INFO 3),
//   * the incidence seen from the END pose is the transpose of the one seen from the BEGIN pose: one set of products, selects at the end,
//   * nothing multiplies by a constant zero (IEEE forbids folding 0 * x, and the general helpers carry W_pr = 0 through every product).
// About 0.4 of the FP64 instructions of the general body and half its live values.  Every a*b + c is written as an explicit fma() and
// contraction is switched off for the rest, so the numbers do not depend on the code around the call (cf. pgo_lin.h); the function
// also compiles for the host, where tools/lean_check_cli holds it against the general formulas in the CPU suite.  Results agree with
// linearize_body() to rounding (1e-13 relative to the block's largest entry; tests/test_lean_host.py, tests/test_gpu_sym.py).
// Reference behaviour: PLUS/include/PoseGraph3dError.h:21-54 (residual), SURVEY.md Appendix A.3-A.5 (closed-form blocks).
#pragma once
#include "pgo_math.h"

namespace pgo {

// upper-triangle position of (i, j), i <= j, in the 21 diagonal-block values of a row (the order linearize_body() uses)
PGO_HD constexpr int lean_upper(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// wp / wr: INFO 3 the diagonals of W_pp / W_rr (3 each); INFO 2 their upper triangles (xx xy xz yy yz zz); INFO 0 unused.
// so / st: Jacobi scale of the incidence's own pose (row) and of the other one (column), 0 in constant dimensions; mo: 1 / 0 likewise.
// Sink: blk(k, x) receives the k-th of the 27 packed values of the off-diagonal block (pgo_kernels.h: 0..8 top-left, 9..17
// bottom-right, 18..26 the non-zero mixed quadrant), dia(k, x) the k-th of the row's 27 own values (21 diagonal-block entries, gradient);
// blk_ready(lo, hi) says that values [lo, hi) will not change any more (the kernel stores them then).
template <int INFO, class Sink>
PGO_HD void lean_incidence(bool begin, const V3& pa, const Q4& qa, const V3& pb, const Q4& qb, const V3& mp, const Q4& mq, const double* wp,
                           const double* wr, const double (&so)[6], const double (&st)[6], const double (&mo)[6], int loss_kind, double loss_a,
                           Sink& out) {
#pragma clang fp contract(off)
  // ---- rotation of a (transposed), translation residual ----
  const double tx = 2.0 * qa.x, ty = 2.0 * qa.y, tz = 2.0 * qa.z;
  const double xx = tx * qa.x, yy = ty * qa.y, zz = tz * qa.z, xy = tx * qa.y, xz = tx * qa.z, yz = ty * qa.z;
  const double wx = tx * qa.w, wy = ty * qa.w, wz = tz * qa.w;
  double Rt[9];
  Rt[0] = 1.0 - (yy + zz); Rt[1] = xy + wz;         Rt[2] = xz - wy;
  Rt[3] = xy - wz;         Rt[4] = 1.0 - (xx + zz); Rt[5] = yz + wx;
  Rt[6] = xz + wy;         Rt[7] = yz - wx;         Rt[8] = 1.0 - (xx + yy);
  const double dx = pb.x - pa.x, dy = pb.y - pa.y, dz = pb.z - pa.z;
  double ep[3], er[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ep[i] = fma(Rt[3 * i + 2], dz, fma(Rt[3 * i + 1], dy, Rt[3 * i] * dx)) - (i == 0 ? mp.x : i == 1 ? mp.y : mp.z);
  // ---- rotation residual and its derivative: A = q_hat (x) conj(q_b); e_r = 2 vec(A (x) q_a); M[:,k] = vec(A (x) e_k (x) q_a) ----
  const double Aw = fma(mq.z, qb.z, fma(mq.y, qb.y, fma(mq.x, qb.x, mq.w * qb.w)));
  const double Ax = fma(mq.z, qb.y, fma(-mq.y, qb.z, fma(mq.x, qb.w, -(mq.w * qb.x))));
  const double Ay = fma(mq.x, qb.z, fma(-mq.z, qb.x, fma(mq.y, qb.w, -(mq.w * qb.y))));
  const double Az = fma(mq.y, qb.x, fma(-mq.x, qb.y, fma(mq.z, qb.w, -(mq.w * qb.z))));
  // vec(A (x) P) = A.w P.v + P.w A.v + A.v x P.v
#define PGO_LEAN_VX(Px, Py, Pz, Pw) fma(-Az, (Py), fma(Ay, (Pz), fma(Ax, (Pw), Aw * (Px))))
#define PGO_LEAN_VY(Px, Py, Pz, Pw) fma(-Ax, (Pz), fma(Az, (Px), fma(Ay, (Pw), Aw * (Py))))
#define PGO_LEAN_VZ(Px, Py, Pz, Pw) fma(-Ay, (Px), fma(Ax, (Py), fma(Az, (Pw), Aw * (Pz))))
  er[0] = 2.0 * PGO_LEAN_VX(qa.x, qa.y, qa.z, qa.w);
  er[1] = 2.0 * PGO_LEAN_VY(qa.x, qa.y, qa.z, qa.w);
  er[2] = 2.0 * PGO_LEAN_VZ(qa.x, qa.y, qa.z, qa.w);
  double M[9];
  // e_0 (x) q = ( w, -z,  y | -x)     e_1 (x) q = ( z,  w, -x | -y)     e_2 (x) q = (-y,  x,  w | -z)
  M[0] = PGO_LEAN_VX(qa.w, -qa.z, qa.y, -qa.x); M[3] = PGO_LEAN_VY(qa.w, -qa.z, qa.y, -qa.x); M[6] = PGO_LEAN_VZ(qa.w, -qa.z, qa.y, -qa.x);
  M[1] = PGO_LEAN_VX(qa.z, qa.w, -qa.x, -qa.y); M[4] = PGO_LEAN_VY(qa.z, qa.w, -qa.x, -qa.y); M[7] = PGO_LEAN_VZ(qa.z, qa.w, -qa.x, -qa.y);
  M[2] = PGO_LEAN_VX(-qa.y, qa.x, qa.w, -qa.z); M[5] = PGO_LEAN_VY(-qa.y, qa.x, qa.w, -qa.z); M[8] = PGO_LEAN_VZ(-qa.y, qa.x, qa.w, -qa.z);
#undef PGO_LEAN_VX
#undef PGO_LEAN_VY
#undef PGO_LEAN_VZ
  // ---- W e, the squared norm, the loss ----
  // apply a 3x3 information block to the columns of a 3 x n matrix held row-major with row stride n (n = 1: a vector)
  auto apply_w = [&](const double* w, const double* X, int n, double* Y) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j >= n) break;
      if (INFO == 0) { Y[j] = X[j]; Y[n + j] = X[n + j]; Y[2 * n + j] = X[2 * n + j]; }
      else if (INFO == 3) { Y[j] = w[0] * X[j]; Y[n + j] = w[1] * X[n + j]; Y[2 * n + j] = w[2] * X[2 * n + j]; }
      else {
        Y[j] = fma(w[2], X[2 * n + j], fma(w[1], X[n + j], w[0] * X[j]));
        Y[n + j] = fma(w[4], X[2 * n + j], fma(w[3], X[n + j], w[1] * X[j]));
        Y[2 * n + j] = fma(w[5], X[2 * n + j], fma(w[4], X[n + j], w[2] * X[j]));
      }
    }
  };
  double wep[3], wer[3];
  apply_w(wp, ep, 1, wep);
  apply_w(wr, er, 1, wer);
  const double s = fma(er[2], wer[2], fma(er[1], wer[1], fma(er[0], wer[0], fma(ep[2], wep[2], fma(ep[1], wep[1], ep[0] * wep[0])))));
  double rho0, rho1;
  loss_eval(loss_kind, loss_a, s, &rho0, &rho1);
  (void)rho0;
  double rs[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) rs[i] = rho1 * so[i];
  // A' B for 3x3 row-major A, B: entry (i, j)
#define PGO_LEAN_TB(A_, B_, i, j) fma((A_)[6 + (i)], (B_)[6 + (j)], fma((A_)[3 + (i)], (B_)[3 + (j)], (A_)[(i)] * (B_)[(j)]))
  // ---- rotation / rotation: 4 M' Wrr M ----
  double F4[6];        // upper triangle, (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
  double mtw[3];
  {
    double Qm[9];
    apply_w(wr, M, 3, Qm);
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) F4[k++] = 4.0 * PGO_LEAN_TB(M, Qm, i, j);
#pragma unroll
    for (int i = 0; i < 3; ++i) mtw[i] = fma(M[6 + i], wer[2], fma(M[3 + i], wer[1], M[i] * wer[0]));
  }
  auto sym6 = [](int i, int j) { return i <= j ? i * 3 - (i * (i - 1)) / 2 + (j - i) : j * 3 - (j * (j - 1)) / 2 + (i - j); };
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out.blk(9 + 3 * i + j, (rs[3 + i] * st[3 + j]) * -F4[sym6(i, j)]);
  out.blk_ready(9, 18);
  // ---- position / position: Rt' Wpp Rt ----
  {
    double X[9];
    apply_w(wp, Rt, 3, X);
    double C1[6];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) C1[k++] = PGO_LEAN_TB(Rt, X, i, j);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) out.blk(3 * i + j, (rs[i] * st[j]) * -C1[sym6(i, j)]);
    out.blk_ready(0, 9);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j) out.dia(lean_upper(i, j), (rs[i] * so[j]) * C1[sym6(i, j)]);
  }
  // gradient, position part: -/+ Rt' W e_p
  const double sg = begin ? -1.0 : 1.0, fb = begin ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double rtw = fma(Rt[6 + i], wep[2], fma(Rt[3 + i], wep[1], Rt[i] * wep[0]));
    out.dia(21 + i, (rho1 * mo[i]) * (sg * rtw));
  }
  // ---- G = 2 Rt [d]x and what carries it: Rt' Wpp G (mixed quadrant; the BEGIN pose's diagonal), G' Wpp G, G' W e_p ----
  // (exactly: the derivative of the rotation POLYNOMIAL v + 2 w (u x v) + 2 u x (u x v) along the perturbation of q_a is
  // 2 (Rt + (|q_a|^2 - 1) I) [d]x — what AutoDiff of the reference's functor yields for a quaternion that is a little off the unit
  // sphere, as 1e5 composed poses of a generated trajectory are by 1e-10)
  const double Dx = 2.0 * dx, Dy = 2.0 * dy, Dz = 2.0 * dz;
  const double nm1 = fma(qa.w, qa.w, fma(qa.z, qa.z, fma(qa.y, qa.y, qa.x * qa.x))) - 1.0;
  const double g00 = Rt[0] + nm1, g11 = Rt[4] + nm1, g22 = Rt[8] + nm1;
  double G[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double ri0 = i == 0 ? g00 : Rt[3 * i], ri1 = i == 1 ? g11 : Rt[3 * i + 1], ri2 = i == 2 ? g22 : Rt[3 * i + 2];
    G[3 * i] = fma(ri1, Dz, -(ri2 * Dy));
    G[3 * i + 1] = fma(ri2, Dx, -(ri0 * Dz));
    G[3 * i + 2] = fma(ri0, Dy, -(ri1 * Dx));
  }
  const double c2 = begin ? 2.0 : -2.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double gtw = fma(G[6 + i], wep[2], fma(G[3 + i], wep[1], G[i] * wep[0]));
    out.dia(24 + i, (rho1 * mo[3 + i]) * fma(fb, gtw, c2 * mtw[i]));
  }
  double U[9];
  apply_w(wp, G, 3, U);
  {
    double RU[9];        // Rt' Wpp G
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) RU[3 * i + j] = PGO_LEAN_TB(Rt, U, i, j);
    // BEGIN: bottom-left of H_ab, (i, j) = S_a[3+i] RU[j][i] S_b[j].   END: top-right of H_ba, (i, j) = S_b[i] RU[i][j] S_a[3+j].
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double r = begin ? rs[3 + i] : rs[i], c = begin ? st[j] : st[3 + j];
        const double x = i == j ? RU[4 * i] : (begin ? RU[3 * j + i] : RU[3 * i + j]);
        out.blk(18 + 3 * i + j, (r * c) * x);
      }
    out.blk_ready(18, 27);
    // the BEGIN pose's diagonal block carries -RU in its top-right quadrant; the END pose's a zero
    const double nb = begin ? -1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) out.dia(lean_upper(i, 3 + j), (rs[i] * so[3 + j]) * (nb * RU[3 * i + j]));
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {
      const double gu = PGO_LEAN_TB(G, U, i, j);
      out.dia(lean_upper(3 + i, 3 + j), (rs[3 + i] * so[3 + j]) * fma(fb, gu, F4[sym6(i, j)]));
    }
#undef PGO_LEAN_TB
}

}  // namespace pgo
