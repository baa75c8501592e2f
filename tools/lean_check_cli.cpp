// tools/lean_check_cli.cpp — host-only check of the lean incidence (csrc/pgo_lin_lean.h) against the general closed-form blocks
// written with the 3x3 helpers of csrc/pgo_math.h (edge_geometry: the derivative of the rotation polynomial, quaternion products),
// the way linearize_body() of pgo_kernels.hip states them: random poses, measurements, information (identity / block-diagonal /
// diagonal), Jacobi scales, constant blocks and loss kinds, both sides of the edge.  Prints the largest difference relative to the
// largest entry of the compared group.  Needs no GPU: tests/test_lean_host.py runs it in the CPU suite.
// usage: lean_check_cli [cases] [seed]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../posegraph-ceres_amd/csrc/pgo_lin_lean.h"

using namespace pgo;

struct Collect {
  double b[28], d[27];
  PGO_HD void blk(int k, double x) { b[k] = x; }
  PGO_HD void dia(int k, double x) { d[k] = x; }
  PGO_HD void blk_ready(int, int) {}
};

// the general statement (W with W_pr = 0): H_ab = [ -C1 , 0 ; RU' , -4MQ ], H_aa = [ C1 , -RU ; . , GU + 4MQ ], H_bb = [ C1 , 0 ; 0 , 4MQ ]
static void general(bool begin, const V3& pa, const Q4& qa, const V3& pb, const Q4& qb, const V3& mp, const Q4& mq, const M3& Wpp, const M3& Wrr,
                    const double* so, const double* st, const double* mo, int kind, double la, double* b, double* d) {
  const EdgeGeom eg = edge_geometry(pa, qa, pb, qb, mp, mq);
  const V3 ep{eg.e[0], eg.e[1], eg.e[2]}, er{eg.e[3], eg.e[4], eg.e[5]};
  const V3 wep = mulv(Wpp, ep), wer = mulv(Wrr, er);
  const M3 X = mul(Wpp, eg.Rt), Qm = mul(Wrr, eg.M), U = mul(Wpp, eg.G);
  const M3 C1 = mulT(eg.Rt, X), RU = mulT(eg.Rt, U), MQ = mulT(eg.M, Qm), GU = mulT(eg.G, U);
  double rho0, rho1;
  loss_eval(kind, la, dot(ep, wep) + dot(er, wer), &rho0, &rho1);
  const V3 rtw = mulTv(eg.Rt, wep), gtw = mulTv(eg.G, wep), mtw = mulTv(eg.M, wer);
  double off[36] = {0}, dg[36] = {0}, gv[6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      if (begin) {
        off[6 * i + j] = -C1.m[3 * i + j]; off[6 * (3 + i) + j] = RU.m[3 * j + i]; off[6 * (3 + i) + 3 + j] = -4.0 * MQ.m[3 * i + j];
        dg[6 * i + j] = C1.m[3 * i + j]; dg[6 * i + 3 + j] = -RU.m[3 * i + j]; dg[6 * (3 + i) + 3 + j] = GU.m[3 * i + j] + 4.0 * MQ.m[3 * i + j];
      } else {
        off[6 * i + j] = -C1.m[3 * j + i]; off[6 * i + 3 + j] = RU.m[3 * i + j]; off[6 * (3 + i) + 3 + j] = -4.0 * MQ.m[3 * j + i];
        dg[6 * i + j] = C1.m[3 * i + j]; dg[6 * (3 + i) + 3 + j] = 4.0 * MQ.m[3 * i + j];
      }
    }
  if (begin) { gv[0] = -rtw.x; gv[1] = -rtw.y; gv[2] = -rtw.z; gv[3] = gtw.x + 2 * mtw.x; gv[4] = gtw.y + 2 * mtw.y; gv[5] = gtw.z + 2 * mtw.z; }
  else { gv[0] = rtw.x; gv[1] = rtw.y; gv[2] = rtw.z; gv[3] = -2 * mtw.x; gv[4] = -2 * mtw.y; gv[5] = -2 * mtw.z; }
  for (int q = 0; q < 9; ++q) {
    const int i = q / 3, j = q % 3;
    b[q] = rho1 * so[i] * st[j] * off[6 * i + j];
    b[9 + q] = rho1 * so[3 + i] * st[3 + j] * off[6 * (3 + i) + 3 + j];
    b[18 + q] = begin ? rho1 * so[3 + i] * st[j] * off[6 * (3 + i) + j] : rho1 * so[i] * st[3 + j] * off[6 * i + 3 + j];
  }
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) d[k++] = rho1 * so[i] * so[j] * dg[6 * i + j];
  for (int i = 0; i < 6; ++i) d[21 + i] = rho1 * mo[i] * gv[i];
}

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 20000;
  std::mt19937_64 rng(argc > 2 ? atoi(argv[2]) : 1);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  auto rq = [&]() { Q4 q{nd(rng), nd(rng), nd(rng), nd(rng)}; const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return Q4{q.x / n, q.y / n, q.z / n, q.w / n}; };
  double worst_b = 0, worst_d = 0, worst_g = 0;
  for (int c = 0; c < cases; ++c) {
    const double span = std::pow(10.0, 3.0 * ud(rng) - 1.0);      // pose distances 0.1 .. 100 m
    const V3 pa{span * nd(rng), span * nd(rng), span * nd(rng)}, pb{pa.x + span * nd(rng), pa.y + span * nd(rng), pa.z + span * nd(rng)};
    Q4 qa = rq();
    const Q4 qb = rq(), mq = rq();
    if (c % 4 == 1) { const double f = 1.0 + 1e-6 * nd(rng); qa = Q4{qa.x * f, qa.y * f, qa.z * f, qa.w * f}; }      // off the unit sphere: the polynomial's derivative, not the rotation's
    const V3 mp{span * nd(rng), span * nd(rng), span * nd(rng)};
    const int info = (c % 3 == 0) ? 0 : (c % 3 == 1) ? 2 : 3;
    double wp[6] = {1, 0, 0, 1, 0, 1}, wr[6] = {1, 0, 0, 1, 0, 1};      // upper triangles xx xy xz yy yz zz
    if (info == 3) { wp[0] = 1 + 99 * ud(rng); wp[3] = 1 + 99 * ud(rng); wp[5] = 1 + 99 * ud(rng); wr[0] = 1 + 999 * ud(rng); wr[3] = 1 + 999 * ud(rng); wr[5] = 1 + 999 * ud(rng); }
    if (info == 2)
      for (double* w : {wp, wr}) {      // L L' of a random lower triangle
        double L[9] = {1 + ud(rng), 0, 0, nd(rng), 1 + ud(rng), 0, nd(rng), nd(rng), 1 + ud(rng)};
        int k = 0;
        for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) { double s = 0; for (int t = 0; t < 3; ++t) s += L[3 * i + t] * L[3 * j + t]; w[k++] = s; }
      }
    const M3 Wpp{{wp[0], wp[1], wp[2], wp[1], wp[3], wp[4], wp[2], wp[4], wp[5]}}, Wrr{{wr[0], wr[1], wr[2], wr[1], wr[3], wr[4], wr[2], wr[4], wr[5]}};
    double so[6], st[6], mo[6];
    const int cm_own = (c % 7 == 0) ? 1 + (c / 7) % 3 : 0, cm_oth = (c % 11 == 0) ? 1 + (c / 11) % 3 : 0;
    for (int i = 0; i < 6; ++i) {
      const bool co = i < 3 ? (cm_own & 1) : (cm_own & 2), ct = i < 3 ? (cm_oth & 1) : (cm_oth & 2);
      mo[i] = co ? 0.0 : 1.0; so[i] = co ? 0.0 : 0.05 + ud(rng); st[i] = ct ? 0.0 : 0.05 + ud(rng);
    }
    const int kind = c % 5 < 4 ? c % 5 : 1;        // trivial, Huber, SoftLOne, Cauchy
    const double la = 0.5 + 2 * ud(rng);
    for (int side = 0; side < 2; ++side) {
      const bool begin = side == 0;
      double b0[28], d0[27];
      general(begin, pa, qa, pb, qb, mp, mq, Wpp, Wrr, so, st, mo, kind, la, b0, d0);
      Collect out;
      double wd[3] = {wp[0], wp[3], wp[5]}, rd[3] = {wr[0], wr[3], wr[5]};
      if (info == 0) lean_incidence<0>(begin, pa, qa, pb, qb, mp, mq, nullptr, nullptr, so, st, mo, kind, la, out);
      else if (info == 2) lean_incidence<2>(begin, pa, qa, pb, qb, mp, mq, wp, wr, so, st, mo, kind, la, out);
      else lean_incidence<3>(begin, pa, qa, pb, qb, mp, mq, wd, rd, so, st, mo, kind, la, out);
      double mb = 0, md = 0, mg = 0, eb = 0, ed = 0, eg = 0;
      for (int k = 0; k < 27; ++k) { mb = std::max(mb, std::fabs(b0[k])); eb = std::max(eb, std::fabs(b0[k] - out.b[k])); }
      for (int k = 0; k < 21; ++k) { md = std::max(md, std::fabs(d0[k])); ed = std::max(ed, std::fabs(d0[k] - out.d[k])); }
      for (int k = 21; k < 27; ++k) { mg = std::max(mg, std::fabs(d0[k])); eg = std::max(eg, std::fabs(d0[k] - out.d[k])); }
      if (mb > 0) worst_b = std::max(worst_b, eb / mb);
      if (md > 0) worst_d = std::max(worst_d, ed / md);
      if (mg > 0) worst_g = std::max(worst_g, eg / mg);
    }
  }
  std::printf("lean_check: %d cases x 2 sides, worst relative difference: off-diagonal block %.3e, diagonal block %.3e, gradient %.3e\n", cases, worst_b, worst_d, worst_g);
  return (worst_b < 1e-12 && worst_d < 1e-12 && worst_g < 1e-11) ? 0 : 1;
}
