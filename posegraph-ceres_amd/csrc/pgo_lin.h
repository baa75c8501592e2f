// pgo_lin.h — one incidence of the linearisation (residual, closed-form Jacobians, Huber corrector, J'J / J'r pieces) for the
// symmetric-tile kernel of pgo_sym_kernels.hip.  The expressions are those of linearize_body() in pgo_kernels.hip, statement by
// statement; that kernel keeps its own inline copy on purpose: hipcc contracts a*b + c*d into FMAs differently depending on the
// surrounding code, a shared function moved the last bit of a few products, and the row kernel's results are pinned bit for bit
// (CG iteration counts equal to the oracle's over whole solves, tests/test_gpu_landmarks.py, test_gpu_parity.py).  The symmetric
// form is held to the row kernels to rounding (tests/test_gpu_sym.py).
// Reference behaviour: PLUS/include/PoseGraph3dError.h:21-54 (residual), SURVEY.md Appendix A.3-A.5 (closed-form blocks).
#pragma once
#include "pgo_kernels.h"
#include "pgo_math.h"

namespace pgo {
namespace {

struct PoseRec { V3 p; Q4 q; };
__device__ __forceinline__ PoseRec load_pose(const double* poses, int v) {
  const double2* s = reinterpret_cast<const double2*>(poses + (size_t)POSE_STRIDE * v);
  const double2 a = s[0], b = s[1], c = s[2], d = s[3];
  return PoseRec{V3{a.x, a.y, b.x}, Q4{b.y, c.x, c.y, d.x}};
}

__device__ __forceinline__ int upper_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

struct WBlocks { M3 pp, pr, rr; };
__device__ __forceinline__ WBlocks load_W(const double* W, size_t stride, size_t idx) {
  double u[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) u[k] = W[(size_t)k * stride + idx];
  WBlocks w;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      w.pp.m[3 * i + j] = (i <= j) ? u[upper_index(i, j)] : u[upper_index(j, i)];
      w.pr.m[3 * i + j] = u[upper_index(i, 3 + j)];
      w.rr.m[3 * i + j] = (i <= j) ? u[upper_index(3 + i, 3 + j)] : u[upper_index(3 + j, 3 + i)];
    }
  return w;
}

__device__ __forceinline__ WBlocks load_W_blockdiag(const double* W, size_t stride, size_t idx) {
  WBlocks w;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {
      const double a = W[(size_t)upper_index(i, j) * stride + idx], b = W[(size_t)upper_index(3 + i, 3 + j) * stride + idx];
      w.pp.m[3 * i + j] = a; w.pp.m[3 * j + i] = a;
      w.rr.m[3 * i + j] = b; w.rr.m[3 * j + i] = b;
    }
#pragma unroll
  for (int k = 0; k < 9; ++k) w.pr.m[k] = 0.0;
  return w;
}

// diagonal information (W = diag(w), C2 / C4's diag(1/sigma^2)): six of the 21 planes are read; the entries set to 0.0 here are the
// exact zeros load_W_blockdiag would have fetched, so the arithmetic behind it is the same to the bit
__device__ __forceinline__ WBlocks load_W_diag(const double* W, size_t stride, size_t idx) {
  WBlocks w;
#pragma unroll
  for (int k = 0; k < 9; ++k) { w.pp.m[k] = 0.0; w.pr.m[k] = 0.0; w.rr.m[k] = 0.0; }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    w.pp.m[4 * i] = W[(size_t)upper_index(i, i) * stride + idx];
    w.rr.m[4 * i] = W[(size_t)upper_index(3 + i, 3 + i) * stride + idx];
  }
  return w;
}


// The incidence (row, col) of an edge a -> b seen from `side` (BEGIN: row = a, END: row = b).  A / Bp: the poses of a / b.
// Out: wv = the off-diagonal block rho' S_row (..) S_col in STORAGE order (INFO != 1: the 27 packed entries of pgo_kernels.h, wv[27] = 0;
// INFO == 1: the 36 row-major entries), v = this incidence's part of the row's diagonal block (21 upper-triangle entries) and gradient (6).
template <int INFO>
__device__ __forceinline__ void lin_slot(const DeviceGraph& g, int side, int row, int col, const PoseRec& A, const PoseRec& Bp, const V3& mp,
                                         const Q4& mq, const double* Wp, size_t wstride, size_t widx, double (&wv)[36], double (&v)[27]) {
      const EdgeGeom eg = edge_geometry(A.p, A.q, Bp.p, Bp.q, mp, mq);
      const V3 ep{eg.e[0], eg.e[1], eg.e[2]}, er{eg.e[3], eg.e[4], eg.e[5]};

      V3 wep, wer;
      M3 C1, C2, RU, GP, MQ, GU;
      if (INFO >= 2) {
        // block-diagonal information (W_pr = 0): only W_pp and W_rr are read (12 of 21 entries; 6 when W is diagonal, INFO 3);
        // every term that carries W_pr in the general branch below is exactly zero there, so both branches give the same numbers
        const WBlocks W = INFO == 3 ? load_W_diag(Wp, wstride, widx) : load_W_blockdiag(Wp, wstride, widx);
        wep = mulv(W.pp, ep);
        wer = mulv(W.rr, er);
        const M3 X = mul(W.pp, eg.Rt), Qm = mul(W.rr, eg.M), U = mul(W.pp, eg.G);
        C1 = mulT(eg.Rt, X); RU = mulT(eg.Rt, U); MQ = mulT(eg.M, Qm); GU = mulT(eg.G, U);
#pragma unroll
        for (int k = 0; k < 9; ++k) { C2.m[k] = 0.0; GP.m[k] = 0.0; }
      } else if (INFO) {
        const WBlocks W = load_W(Wp, wstride, widx);
        const V3 a1 = mulv(W.pp, ep), a2 = mulv(W.pr, er), b1 = mulTv(W.pr, ep), b2 = mulv(W.rr, er);
        wep = V3{a1.x + a2.x, a1.y + a2.y, a1.z + a2.z};
        wer = V3{b1.x + b2.x, b1.y + b2.y, b1.z + b2.z};
        const M3 X = mul(W.pp, eg.Rt), P = mul(W.pr, eg.M), Qm = mul(W.rr, eg.M), U = mul(W.pp, eg.G);
        C1 = mulT(eg.Rt, X); C2 = mulT(eg.Rt, P); RU = mulT(eg.Rt, U);
        GP = mulT(eg.G, P); MQ = mulT(eg.M, Qm); GU = mulT(eg.G, U);
      } else {
        wep = ep; wer = er;
        C1 = mulT(eg.Rt, eg.Rt); RU = mulT(eg.Rt, eg.G); MQ = mulT(eg.M, eg.M); GU = mulT(eg.G, eg.G);
#pragma unroll
        for (int k = 0; k < 9; ++k) { C2.m[k] = 0.0; GP.m[k] = 0.0; }
      }
      const double s = dot(ep, wep) + dot(er, wer);
      double rho0, rho1;
      loss_eval(g.loss_kind, g.loss_a, s, &rho0, &rho1);

      // 6x6 results for this row: off-diagonal block, own diagonal contribution, own gradient
      double off[36], dg[36], gv[6];
      const M3 RU2C2 = axpby(1.0, RU, 2.0, C2);          // Rt^T (U + 2P)
      const M3 GP4MQ = axpby(2.0, GP, 4.0, MQ);          // 2 G^T P + 4 M^T Qm
      const V3 rtw = mulTv(eg.Rt, wep), gtw = mulTv(eg.G, wep), mtw = mulTv(eg.M, wer);
      if (side == SIDE_BEGIN) {
        // H_ab = [ -C1 , 2C2 ; (RU+2C2)^T , -(2GP+4MQ) ]    H_aa = [ C1 , -(RU+2C2) ; sym , GU + 2(GP+GP^T) + 4MQ ]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            off[6 * i + j] = -C1.m[3 * i + j];
            off[6 * i + 3 + j] = 2.0 * C2.m[3 * i + j];
            off[6 * (3 + i) + j] = RU2C2.m[3 * j + i];
            off[6 * (3 + i) + 3 + j] = -GP4MQ.m[3 * i + j];
            dg[6 * i + j] = C1.m[3 * i + j];
            dg[6 * i + 3 + j] = -RU2C2.m[3 * i + j];
            dg[6 * (3 + i) + j] = -RU2C2.m[3 * j + i];
            dg[6 * (3 + i) + 3 + j] = GU.m[3 * i + j] + 2.0 * (GP.m[3 * i + j] + GP.m[3 * j + i]) + 4.0 * MQ.m[3 * i + j];
          }
        gv[0] = -rtw.x; gv[1] = -rtw.y; gv[2] = -rtw.z;
        gv[3] = gtw.x + 2.0 * mtw.x; gv[4] = gtw.y + 2.0 * mtw.y; gv[5] = gtw.z + 2.0 * mtw.z;
      } else {
        // H_ba = H_ab^T                                      H_bb = [ C1 , -2C2 ; -2C2^T , 4MQ ]
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            off[6 * i + j] = -C1.m[3 * j + i];
            off[6 * i + 3 + j] = RU2C2.m[3 * i + j];
            off[6 * (3 + i) + j] = 2.0 * C2.m[3 * j + i];
            off[6 * (3 + i) + 3 + j] = -GP4MQ.m[3 * j + i];
            dg[6 * i + j] = C1.m[3 * i + j];
            dg[6 * i + 3 + j] = -2.0 * C2.m[3 * i + j];
            dg[6 * (3 + i) + j] = -2.0 * C2.m[3 * j + i];
            dg[6 * (3 + i) + 3 + j] = 4.0 * MQ.m[3 * i + j];
          }
        gv[0] = rtw.x; gv[1] = rtw.y; gv[2] = rtw.z;
        gv[3] = -2.0 * mtw.x; gv[4] = -2.0 * mtw.y; gv[5] = -2.0 * mtw.z;
      }
      // constant parameter blocks drop out of the program; Jacobi column scaling S
      const uint8_t m_own = g.cmask[row], m_oth = g.cmask[col];
      double so[6], st[6], mo[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bool co = (i < 3) ? (m_own & 1) : (m_own & 2);
        const bool ct = (i < 3) ? (m_oth & 1) : (m_oth & 2);
        mo[i] = co ? 0.0 : 1.0;
        so[i] = co ? 0.0 : g.scale[6 * (size_t)row + i];
        st[i] = ct ? 0.0 : g.scale[6 * (size_t)col + i];
      }
      if (INFO != 1) {
        // packed slot: positions 0..8 top-left, 9..17 bottom-right, 18..26 the stored off-diagonal quadrant (bottom-left of
        // H_ab for the BEGIN slot, top-right of H_ba for the END slot), 27 unused.  Same products as the full layout.
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int i = q / 3, j = q % 3;
          const int ktl = 6 * i + j, kbr = 6 * (3 + i) + 3 + j, kbl = 6 * (3 + i) + j, ktr = 6 * i + 3 + j;
          wv[q] = rho1 * so[ktl / 6] * st[ktl % 6] * off[ktl];
          wv[9 + q] = rho1 * so[kbr / 6] * st[kbr % 6] * off[kbr];
          const double vbl = rho1 * so[kbl / 6] * st[kbl % 6] * off[kbl];
          const double vtr = rho1 * so[ktr / 6] * st[ktr % 6] * off[ktr];
          wv[18 + q] = side == SIDE_BEGIN ? vbl : vtr;
        }
        wv[27] = 0.0;
      } else {
#pragma unroll
        for (int k0 = 0; k0 < 36; ++k0) wv[k0] = rho1 * so[k0 / 6] * st[k0 % 6] * off[k0];
      }
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) v[k++] = rho1 * so[i] * so[j] * dg[6 * i + j];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[21 + i] = rho1 * mo[i] * gv[i];
}

}  // namespace
}  // namespace pgo
