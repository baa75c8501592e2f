// pgo_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the pose-graph NLLS hot path.
//
// Reference behaviour implemented (never its code): PLUS/include/PoseGraph3dError.h:21-54 residual,
// finial.cpp:491-544 problem/solver set-up, Ceres 1.13 semantics restated in SURVEY.md Appendix A.
//
// Design: the normal equations are kept as a block-sparse row matrix whose block
// slots ARE the pose-graph incidences.  Row v owns one diagonal slot plus one slot per incident edge;
// rows are packed into workgroups of `block` slots.  One lane owns one slot:
//   * k_linearize   : lane recomputes the edge geometry from the two 64-byte pose records, forms its
//                     off-diagonal 6x6 block (written as 18 x 16 B, 1 KiB per wave instruction) and
//                     its 21+6 contribution to the row's diagonal block / gradient, which are summed
//                     per row through LDS in a fixed order (deterministic, no FP64 atomics).
//   * k_pcg_spmv    : lane multiplies its block with the gathered 6-vector, LDS segmented row sums.
// Everything is FP64 and HBM/L2 bound; MFMA is deliberately not used for 6x6 blocks.
#include <algorithm>
#include <cstdlib>

#include "pgo_kernels.h"
#include "pgo_lm_rules.h"
#include "pgo_math.h"
#include "pgo_wave.h"
#include "pgo_lin_lean.h"

namespace pgo {

namespace {

constexpr int VEC_BLOCK = 384;  // 64 poses x 6 tangent dims (a pose or 2-/4-pose cluster never straddles a workgroup); measured 192/384/768: 384 best on C2
constexpr int POSE_BLOCK = 256;
constexpr int EDGE_BLOCK = 256;
constexpr int NV_LIN = 27;      // 21 (symmetric diagonal block) + 6 (gradient)
constexpr int SPMV_LDS_STRIDE = 7;

// Index of component k of pose `row` in the exchange buffer g.cg_q.  Each rank owns `rows_per` consecutive poses and
// a segment of `seg` doubles: [rows_per*6 entries of q = A p][pq_cap partial sums of p'q].  With one rank this is 6*row+k.
__device__ __forceinline__ size_t q_index(const DeviceGraph& g, int row, int k) {
  if (g.world == 1) return (size_t)row * 6 + k;   // one rank: no integer division on the hot path
  const int rk = row / g.rows_per;
  return (size_t)rk * g.seg + (size_t)(row - rk * g.rows_per) * 6 + k;
}
__device__ __forceinline__ size_t q_index_flat(const DeviceGraph& g, int idx) {   // idx = 6*row + k
  if (g.world == 1) return (size_t)idx;
  const int row = idx / 6;
  return q_index(g, row, idx - 6 * row);
}

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

// ------------------------------------------------------------------------------------------------
// K_linearize: residual + closed-form Jacobians + Huber corrector + J^T J / J^T r, fused.
// INFO: 0 identity information, 1 general W, 2 block-diagonal W.  0 and 2 write the packed 27-entry slots (pgo_kernels.h).
// ------------------------------------------------------------------------------------------------
// gate: a speculative launch behind the step tail of a CG batch (the linearisation of the candidate point into the spare buffers)
// runs only once the CG has stopped, like the tail itself.
// (the body is shared with the universal slot kernel k_uni_s further down)
// SYMOUT: the off-diagonal blocks go to the symmetric tile form (pgo_sym.h) instead — slot t's block to stored slot g.sym_dst[t], the
// mirrored incidence of an interior edge (sym_dst < 0) writes none; diagonal blocks and gradient as always (compile-time: the
// other instantiations are the code they were)
__device__ __forceinline__ void store_pair_nt(double2* p, double a, double b) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d x;
  x.x = a; x.y = b;
  __builtin_nontemporal_store(x, reinterpret_cast<v2d*>(p));
}
template <int INFO, int PASSES = 1, bool SYMOUT = false>
__device__ __forceinline__ void linearize_body(const DeviceGraph& g, double* lds) {
  constexpr int NVP = (NV_LIN + PASSES - 1) / PASSES;   // values per round
  constexpr int NVS = NVP | 1;                          // LDS stride per lane (odd)
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  const int s_begin = g.wg_slot_begin[wg], s_end = g.wg_slot_begin[wg + 1];
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  const bool single = (s_end - s_begin) == B;
  double acc[PASSES];
  // the row bookkeeping of this lane's first sum of every round, requested before anything else (it used to be a dependent global
  // load behind the barrier)
  int pre_rb[PASSES], pre_rc[PASSES];
  uint8_t pre_cm[PASSES];
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    acc[ps] = 0.0; pre_rb[ps] = 0; pre_rc[ps] = 0; pre_cm[ps] = 0;
    if (tid < nrows * NVP) {
      const int row = r0 + tid / NVP;
      pre_rb[ps] = g.row_slot_begin[row]; pre_rc[ps] = g.row_slot_cnt[row]; pre_cm[ps] = g.cmask[row];
    }
  }

  for (int cb = s_begin; cb < s_end; cb += B) {
    const int t = cb + tid;
    const uint8_t side = g.slot_side[t];
    double v[NV_LIN];
#pragma unroll
    for (int k = 0; k < NV_LIN; ++k) v[k] = 0.0;

    if (side <= SIDE_END) {
      const int row = g.slot_row[t], col = g.slot_col[t];
      const int a = (side == SIDE_BEGIN) ? row : col;
      const int b = (side == SIDE_BEGIN) ? col : row;
      const PoseRec A = load_pose(g.pose_x, a), Bp = load_pose(g.pose_x, b);
      const size_t ns = (size_t)g.n_slots;
      const V3 mp{g.smeas[t], g.smeas[ns + t], g.smeas[2 * ns + t]};
      const Q4 mq{g.smeas[3 * ns + t], g.smeas[4 * ns + t], g.smeas[5 * ns + t], g.smeas[6 * ns + t]};
      const EdgeGeom eg = edge_geometry(A.p, A.q, Bp.p, Bp.q, mp, mq);
      const V3 ep{eg.e[0], eg.e[1], eg.e[2]}, er{eg.e[3], eg.e[4], eg.e[5]};

      V3 wep, wer;
      M3 C1, C2, RU, GP, MQ, GU;
      if (INFO >= 2) {
        // block-diagonal information (W_pr = 0): only W_pp and W_rr are read (12 of 21 entries; 6 when W is diagonal, INFO 3);
        // every term that carries W_pr in the general branch below is exactly zero there, so both branches give the same numbers
        const WBlocks W = INFO == 3 ? load_W_diag(g.sW, ns, (size_t)t) : load_W_blockdiag(g.sW, ns, (size_t)t);
        wep = mulv(W.pp, ep);
        wer = mulv(W.rr, er);
        const M3 X = mul(W.pp, eg.Rt), Qm = mul(W.rr, eg.M), U = mul(W.pp, eg.G);
        C1 = mulT(eg.Rt, X); RU = mulT(eg.Rt, U); MQ = mulT(eg.M, Qm); GU = mulT(eg.G, U);
#pragma unroll
        for (int k = 0; k < 9; ++k) { C2.m[k] = 0.0; GP.m[k] = 0.0; }
      } else if (INFO) {
        const WBlocks W = load_W(g.sW, ns, (size_t)t);
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
      double2* out = reinterpret_cast<double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
      bool st_blk = true;
      if (SYMOUT) {
        const int dst = g.sym_dst[t];
        st_blk = dst >= 0;
        const int d0 = st_blk ? dst : 0;
        out = reinterpret_cast<double2*>(g.sym_val + (size_t)(d0 >> 6) * TILE_DOUBLES + (size_t)(d0 & 63) * 2);
      }
      if (INFO != 1) {
        // packed slot: positions 0..8 top-left, 9..17 bottom-right, 18..26 the stored off-diagonal quadrant (bottom-left of
        // H_ab for the BEGIN slot, top-right of H_ba for the END slot), 27 unused.  Same products as the full layout.
        double wv[28];
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
#pragma unroll
        for (int kk = 0; kk < BLK_PAIRS_PACKED; ++kk) {
          if (SYMOUT) {      // (a symmetric-form session is large: the blocks are written once and next read from HBM — nontemporal)
            if (st_blk) store_pair_nt(out + (size_t)kk * 64, wv[2 * kk], wv[2 * kk + 1]);
          } else {
            out[(size_t)kk * 64] = double2{wv[2 * kk], wv[2 * kk + 1]};
          }
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) {
          const int k0 = 2 * kk, k1 = 2 * kk + 1;
          double2 w;
          w.x = rho1 * so[k0 / 6] * st[k0 % 6] * off[k0];
          w.y = rho1 * so[k1 / 6] * st[k1 % 6] * off[k1];
          if (SYMOUT) { if (st_blk) store_pair_nt(out + (size_t)kk * 64, w.x, w.y); }
          else out[(size_t)kk * 64] = w;
        }
      }
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) v[k++] = rho1 * so[i] * so[j] * dg[6 * i + j];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[21 + i] = rho1 * mo[i] * gv[i];
    }

    // per-row sums of the 27 values through LDS, in PASSES rounds of NVP values (stride NVS doubles per lane: odd, so the 8-byte
    // writes of a wave are conflict-free).  One round needs 27 * 8 B of LDS per lane — 55 KB per 256 lanes, two work-groups per CU;
    // two rounds of 14 / 13 values need 15 * 8 B.  Measured in r03 (100 k poses / 1 M edges): the kernel holds 219 VGPRs (2 waves per
    // SIMD), so fewer LDS bytes alone buy no occupancy, and capping the registers to 168 / 128 (amdgpu_waves_per_eu 3 / 4) makes the
    // compiler spill 176 / 372 bytes per lane: 244 us -> 423 / 409 (one / two rounds, 3 waves) -> 602 / 627 us (two / three rounds,
    // 4 waves).  PASSES stays 1.
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int k0 = ps * NVP;
#pragma unroll
      for (int k = 0; k < NVP; ++k) if (k0 + k < NV_LIN) lds[tid * NVS + k] = v[k0 + k];
      __syncthreads();
      for (int idx = tid; idx < nrows * NVP; idx += B) {
        const int rl = idx / NVP, kk = idx - rl * NVP, k = k0 + kk;
        if (k >= NV_LIN) continue;
        const int row = r0 + rl;
        const bool first = idx == tid && cb == s_begin;
        const int rb = first ? pre_rb[ps] : g.row_slot_begin[row];
        const int rc = first ? pre_rc[ps] : g.row_slot_cnt[row];
        const int sb = max(rb, cb) - cb, se = min(rb + rc, cb + B) - cb;
        double s0 = 0.0, s1 = 0.0;
        int j = sb;
        for (; j + 1 < se; j += 2) { s0 += lds[j * NVS + kk]; s1 += lds[(j + 1) * NVS + kk]; }
        if (j < se) s0 += lds[j * NVS + kk];
        double s = s0 + s1;
        if (single) {
          if (k < 21) {
            // k -> (i,j) of the upper triangle
            int i = 0, base = 0;
            while (k >= base + (6 - i)) { base += 6 - i; ++i; }
            const int j2 = i + (k - base);
            if (i == j2) {
              const uint8_t cmk = first ? pre_cm[ps] : g.cmask[row];
              const bool c = (i < 3) ? (cmk & 1) : (cmk & 2);
              if (c) s = 1.0;  // unit diagonal keeps the constant dims decoupled and the block SPD
            }
            g.Hdiag[36 * (size_t)row + 6 * i + j2] = s;
            g.Hdiag[36 * (size_t)row + 6 * j2 + i] = s;
          } else {
            g.grad[6 * (size_t)row + (k - 21)] = s;
          }
        } else {
          acc[ps] += s;
        }
      }
      __syncthreads();
    }
  }
  if (!single) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int k = ps * NVP + tid;
      if (tid < NVP && k < NV_LIN) {
        const int row = r0;
        double s = acc[ps];
        if (k < 21) {
          int i = 0, base = 0;
          while (k >= base + (6 - i)) { base += 6 - i; ++i; }
          const int j = i + (k - base);
          if (i == j) {
            const bool c = (i < 3) ? (g.cmask[row] & 1) : (g.cmask[row] & 2);
            if (c) s = 1.0;
          }
          g.Hdiag[36 * (size_t)row + 6 * i + j] = s;
          g.Hdiag[36 * (size_t)row + 6 * j + i] = s;
        } else {
          g.grad[6 * (size_t)row + (k - 21)] = s;
        }
      }
    }
  }
}
template <int INFO>
__global__ __launch_bounds__(256) void k_linearize(DeviceGraph g, int gate) {
  extern __shared__ double lds[];  // NV_LIN * block
  if (gate == 1 && !g.cg->done) return;
  if (gate == 2 && (g.lm->halt || !g.lm->accepted)) return;   // device-resident LM: behind an accepted step only (g.pose_x is the candidate)
  linearize_body<INFO>(g, lds);
}

template <int INFO>
__global__ __launch_bounds__(256) void k_linearize_symout(DeviceGraph g, int gate) {
  extern __shared__ double lds[];
  if (gate == 1 && !g.cg->done) return;
  if (gate == 2 && (g.lm->halt || !g.lm->accepted)) return;
  linearize_body<INFO, 1, true>(g, lds);
}

// The lean per-incidence algebra (pgo_lin_lean.h: 0.68 of the FP64 instructions, half the LDS, 168 registers) is THE linearisation wherever
// the information has no position / rotation coupling (INFO 0, 2, 3: packed 27-entry slots); the general body above stays for INFO 1.
#include "pgo_lean_body.h"

template <int INFO, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_linearize_lean_bsr(DeviceGraph g, int gate) {
  extern __shared__ double lds[];      // LEAN_NV doubles per lane PAIR
  if (gate == 1 && !g.cg->done) return;
  if (gate == 2 && (g.lm->halt || !g.lm->accepted)) return;
  lean_linearize_body<INFO, true>(g, lds);
}
// the body a universal-stream kernel runs for its LIN operation (compile-time: one body per instantiation)
template <int INFO>
__device__ __forceinline__ void linearize_any(const DeviceGraph& g, double* lds) {
  if constexpr (INFO == 1) linearize_body<1>(g, lds);
  else lean_linearize_body<INFO, true>(g, lds);
}

// Jacobi scaling, computed once at iteration 0 from the unscaled diag(J^T J):  S = 1 / (1 + sqrt(d)).
__global__ void k_scale_from_diag(DeviceGraph g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 6 * g.N) return;
  const int v = idx / 6, i = idx - 6 * v;
  const bool c = (i < 3) ? (g.cmask[v] & 1) : (g.cmask[v] & 2);
  const double d = c ? 0.0 : g.Hdiag[36 * (size_t)v + 7 * i];
  g.scale[idx] = 1.0 / (1.0 + sqrt(d));
}

// LM damping (LevenbergMarquardtStrategy::ComputeStep) per pose:
//   D^2 = clamp(diag(H~)) / radius, A_vv = H~_vv + D^2 -> diagonal BSR slot, M_v = A_vv^-1.
// mode 0: clamp fresh, 1: reuse the clamped diagonal (rejected step), 2: take g.d2 as given (tests)
__device__ __forceinline__ void damping_pose(const DeviceGraph& g, int v, double radius, double min_diag, double max_diag, int mode) {
  double A[36];
  const double2* src = reinterpret_cast<const double2*>(g.Hdiag + 36 * (size_t)v);
#pragma unroll
  for (int k = 0; k < 18; ++k) { const double2 t = src[k]; A[2 * k] = t.x; A[2 * k + 1] = t.y; }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double d2;
    if (mode == 2) {
      d2 = g.d2[6 * (size_t)v + i];
    } else {
      double dc;
      if (mode == 1) dc = g.diag_clamped[6 * (size_t)v + i];
      else { dc = fmin(fmax(A[7 * i], min_diag), max_diag); g.diag_clamped[6 * (size_t)v + i] = dc; }
      d2 = dc / radius;
      g.d2[6 * (size_t)v + i] = d2;
    }
    A[7 * i] += d2;
  }
  if (g.row_slot_cnt[v] > 0) {   // rows owned by this rank (all rows with one rank)
    const int slot = g.row_slot_begin[v];
#pragma unroll
    for (int k = 0; k < 36; ++k) g.bsr_val[bsr_index(slot, bsr_pos(g.blk_packed, SIDE_DIAG, k))] = A[k];   // packed: the mirrored quadrant writes the same values twice
  }
  if (g.cluster > 1) return;  // the cluster preconditioner kernel builds M^-1
  double Ai[36];
  if (!spd6_inverse(A, Ai)) atomicOr(&g.flags[1], 1);
  double2* dst = reinterpret_cast<double2*>(g.Minv + 36 * (size_t)v);
#pragma unroll
  for (int k = 0; k < 18; ++k) dst[k] = double2{Ai[2 * k], Ai[2 * k + 1]};
}
__global__ void k_damping(DeviceGraph g, double radius, double min_diag, double max_diag, int mode) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (g.lm) {   // device-resident LM: radius and "reuse the clamped diagonal" come from the device state
    if (g.lm->halt || g.lm->phase == LM_PHASE_CONT) return;
    radius = g.lm->core.radius; mode = g.lm->core.reuse_diagonal ? 1 : 0;
  }
  if (v >= g.N) return;
  damping_pose(g, v, radius, min_diag, max_diag, mode);
}

// Cluster-Jacobi preconditioner: CL consecutive poses (a piece of the odometry chain, plus whatever loop
// edges fall inside it) form one dense (6 CL)^2 diagonal block of H~ + D^2, inverted in LDS by in-place
// Gauss-Jordan (SPD, no pivoting).  One wave per cluster.  Row r of the inverse is what lane r of the
// vector kernels reads (contiguous 6 CL doubles).
// One wave: the clusters [cl_first, cl_first + 64 / (6 CL)) of this rank, as far as they lie below cl_end.
// CGSTART (the HEAD operation of the one-launch streams): the wave also starts the CG for its rows — b = S g, x0 = 0, r0 = b and
// u0 = M^-1 b straight from the registers the inverse sits in (lane j holds row j of it), written to cg_b / cg_x / cg_r / cg_u and to
// `u_out` (the exchange buffer the next launch gathers from): no store -> barrier -> reload of the Jacobi block in between.
template <int CL, bool CGSTART = false>
__device__ __forceinline__ void cluster_precond_wave(const DeviceGraph& g, double radius, double min_diag, double max_diag, int mode,
                                                     int cl_first, int cl_end, int lane, double* u_out = nullptr) {
  // mode >= 0: this kernel also does k_damping's job for its poses (D^2, clamped diagonal, damped diagonal BSR slot) —
  // one launch fewer per LM iteration; mode < 0: k_damping ran before (several ranks: D^2 is needed for ALL rows).
  //
  // Register-resident Gauss-Jordan: lane j of a group of DIM lanes holds COLUMN j of the cluster matrix (DIM doubles),
  // 64/DIM clusters per wave.  Pivot k: column k is broadcast from lane k of the group (ds_bpermute), row k is the k-th
  // register of every lane; no LDS array, no barriers (the LDS version spent ~1.2 us per pivot on dependent LDS
  // round trips).
  constexpr int DIM = 6 * CL, CPW = 64 / DIM;
  const int grp = lane / DIM, j = lane - grp * DIM;
  const int cl_local = cl_first + grp;                      // cluster index among this rank's clusters
  const bool live = grp < CPW && cl_local < cl_end;
  const int base = grp * DIM;
  const int c = g.row_lo / CL + cl_local;
  const int v0 = c * CL;
  const int lp = j / 6, jc = j - 6 * lp;                    // this lane's column: pose lp of the cluster, component jc
  const int v = v0 + lp;
  double a[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) a[i] = (i == j) ? 1.0 : 0.0;   // poses past the end: identity
  double bj = 0.0;
  if constexpr (CGSTART) {
    if (live && v < g.N) bj = g.scale[6 * (size_t)v + jc] * g.grad[6 * (size_t)v + jc];     // (requested in front of everything the inverse waits for)
  }
  if (live && v < g.N) {
    // own diagonal block: rows of pose lp, column jc
#pragma unroll
    for (int ic = 0; ic < 6; ++ic) {
      const double hv = g.Hdiag[36 * (size_t)v + 6 * ic + jc];
#pragma unroll
      for (int q = 0; q < CL; ++q) if (q == lp) a[6 * q + ic] = hv;
    }
    double d2 = 0.0, diag = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) if (i == j) diag = a[i];
    if (mode < 0 || mode == 2) d2 = g.d2[6 * (size_t)v + jc];
    else if (mode == 1) d2 = g.diag_clamped[6 * (size_t)v + jc] / radius;
    else {
      const double dc = fmin(fmax(diag, min_diag), max_diag);
      g.diag_clamped[6 * (size_t)v + jc] = dc;
      d2 = dc / radius;
    }
    if (mode == 0 || mode == 1) g.d2[6 * (size_t)v + jc] = d2;
#pragma unroll
    for (int i = 0; i < DIM; ++i) if (i == j) a[i] += d2;
    if (mode >= 0) {   // damped diagonal BSR slot (k_damping's job)
      const int slot = g.row_slot_begin[v];
#pragma unroll
      for (int ic = 0; ic < 6; ++ic) {
        double val = 0.0;
#pragma unroll
        for (int q = 0; q < CL; ++q) if (q == lp) val = a[6 * q + ic];
        g.bsr_val[bsr_index(slot, bsr_pos(g.blk_packed, SIDE_DIAG, 6 * ic + jc))] = val;
      }
    }
  }
  if (live) {
    // in-cluster off-diagonal blocks (BEGIN slots only; the matrix is symmetric, this lane needs column j)
    for (int sidx = g.cl_ptr[cl_local]; sidx < g.cl_ptr[cl_local + 1]; ++sidx) {
      const int slot = g.cl_slot[sidx];
      const int rc = g.cl_rc[sidx];
      const int r = rc >> 4, cc = rc & 15;    // block H_{r,cc}: rows of pose r, columns of pose cc
      if (cc == lp) {           // A[6r + a][j] += B[a][jc]
#pragma unroll
        for (int ai = 0; ai < 6; ++ai) {
          const double bv = bsr_elem(g, slot, SIDE_BEGIN, 6 * ai + jc);
#pragma unroll
          for (int q = 0; q < CL; ++q) if (q == r) a[6 * q + ai] += bv;
        }
      } else if (r == lp) {     // A[6cc + b][j] += B[jc][b]   (mirror)
#pragma unroll
        for (int bi = 0; bi < 6; ++bi) {
          const double bv = bsr_elem(g, slot, SIDE_BEGIN, 6 * jc + bi);
#pragma unroll
          for (int q = 0; q < CL; ++q) if (q == cc) a[6 * q + bi] += bv;
        }
      }
    }
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < DIM; ++k) {
    double colk[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) colk[i] = __shfl(a[i], base + k);   // A[i][k], held by lane k of the group
    const double pk = colk[k];
    if (!(pk > 0.0)) ok = false;
    const double ip = 1.0 / pk;
    const double rowk = a[k];                                          // A[k][j]
    if (j != k) {
#pragma unroll
      for (int i = 0; i < DIM; ++i) if (i != k) a[i] -= colk[i] * rowk * ip;
      a[k] = rowk * ip;
    } else {
#pragma unroll
      for (int i = 0; i < DIM; ++i) a[i] = (i == k) ? ip : colk[i] * -ip;
    }
  }
  // Lane j holds COLUMN j of the inverse; it is stored as ROW j (16-byte stores, a lane's row contiguous): the inverse of a symmetric
  // matrix is symmetric up to rounding, and every reader — the vector kernels' row lanes, the CG start below — applies the same stored rows.
  if (live) {
    if (!ok && j == 0) atomicOr(&g.flags[1], 1);
    double2* out = reinterpret_cast<double2*>(g.Minv + (size_t)c * DIM * DIM + (size_t)j * DIM);
#pragma unroll
    for (int i = 0; i < DIM / 2; ++i) out[i] = double2{a[2 * i], a[2 * i + 1]};
  }
  if constexpr (CGSTART) {
    double u = 0.0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) u += a[i] * __shfl(bj, base + i);
    if (live && v < g.N) {
      const size_t ridx = 6 * (size_t)v + jc;
      g.cg_b[ridx] = bj;
      g.cg_x[ridx] = 0.0;
      g.cg_r[ridx] = bj;
      g.cg_u[ridx] = u;
      u_out[ridx] = u;
    }
  }
}
template <int CL>
__global__ __launch_bounds__(64) void k_cluster_precond(DeviceGraph g, double radius, double min_diag, double max_diag, int mode) {
  constexpr int CPW = 64 / (6 * CL);
  if (g.lm) {   // device-resident LM (k_damping)
    if (g.lm->halt || g.lm->phase == LM_PHASE_CONT) return;
    radius = g.lm->core.radius;
    if (mode >= 0) mode = g.lm->core.reuse_diagonal ? 1 : 0;
  }
  const int n_cl = (g.row_hi - g.row_lo + CL - 1) / CL;
  cluster_precond_wave<CL>(g, radius, min_diag, max_diag, mode, (int)blockIdx.x * CPW, n_cl, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Cost only: 0.5 * sum rho(|L e|^2) over edges at `poses` (ComputeCandidatePointAndEvaluateCost).
// ------------------------------------------------------------------------------------------------
// 0.5 rho(|L e|^2) of edge e at `poses`
template <int INFO>
__device__ __forceinline__ double edge_cost(const DeviceGraph& g, const double* poses, int e) {
  const PoseRec A = load_pose(poses, g.edge_a[e]), B = load_pose(poses, g.edge_b[e]);
  const size_t E = (size_t)g.E;
  const V3 mp{g.emeas[e], g.emeas[E + e], g.emeas[2 * E + e]};
  const Q4 mq{g.emeas[3 * E + e], g.emeas[4 * E + e], g.emeas[5 * E + e], g.emeas[6 * E + e]};
  double er[6];
  edge_error(A.p, A.q, B.p, B.q, mp, mq, er);
  double s;
  if (INFO) {
    const WBlocks W = g.info_mode == 3 ? load_W_diag(g.eW, E, (size_t)e) : load_W(g.eW, E, (size_t)e);   // (diagonal W: 6 of the 21 planes)
    const V3 ep{er[0], er[1], er[2]}, eq{er[3], er[4], er[5]};
    const V3 a1 = mulv(W.pp, ep), a2 = mulv(W.pr, eq), b1 = mulTv(W.pr, ep), b2 = mulv(W.rr, eq);
    s = dot(ep, V3{a1.x + a2.x, a1.y + a2.y, a1.z + a2.z}) + dot(eq, V3{b1.x + b2.x, b1.y + b2.y, b1.z + b2.z});
  } else {
    s = er[0] * er[0] + er[1] * er[1] + er[2] * er[2] + er[3] * er[3] + er[4] * er[4] + er[5] * er[5];
  }
  double rho0, rho1;
  loss_eval(g.loss_kind, g.loss_a, s, &rho0, &rho1);
  return 0.5 * rho0;
}

// n_edge_wg work-groups stride over the edges (one edge per lane up to MAX_EDGE_WG * EDGE_BLOCK edges: beyond that the
// partial row — and, in the step tail, the ticket counter every work-group bumps — would grow with E; 4 300 tickets on one
// address cost 0.17 ms at 1 M edges, r03 profile).
template <int INFO>
__global__ void k_cost(DeviceGraph g, const double* poses, double* part, int gate) {
  __shared__ double scratch[8];
  if (gate && !g.cg->done) return;
  double c[1] = {0.0};
  {       // (four edges of the lane's stride in flight, added in the one-by-one order: step_tail_body)
    const int stride = gridDim.x * blockDim.x;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < g.E; e += 4 * stride) {
      double c4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = edge_cost<INFO>(g, poses, min(e + k * stride, g.E - 1));
#pragma unroll
      for (int k = 0; k < 4; ++k) c[0] += (e + k * stride < g.E) ? c4[k] : 0.0;
    }
  }
  block_sum<1>(c, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = c[0];
}

// ------------------------------------------------------------------------------------------------
// Materialising evaluation (the Problem::Evaluate analogue): per edge r (6), J_begin, J_end (6x6
// row-major, local tangent columns [dp|dtheta]) with the loss corrector and constant masks applied.
// ------------------------------------------------------------------------------------------------
// Writes a [count][W] row-major matrix whose rows were produced one per lane: staged through LDS (row stride 37
// doubles: conflict-free 8-byte writes) so that the global stores are 16 B per lane, contiguous across the wave.
template <int W>
__device__ __forceinline__ void store_rows_coalesced(double* lds_w, double* out, int e0, int count, int lane) {
  __syncthreads();
  double2* dst = reinterpret_cast<double2*>(out + (size_t)e0 * W);
  const int n2 = count * W / 2;
  for (int i = lane; i < n2; i += 64) {
    const int d0 = 2 * i, el = d0 / W, kk = d0 - el * W;
    // nontemporal: 600+ bytes per edge written once, read next by the host or a later launch from HBM anyway (C4: 188 -> ... us)
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d x;
    x.x = lds_w[el * 37 + kk]; x.y = lds_w[el * 37 + kk + 1];
    __builtin_nontemporal_store(x, reinterpret_cast<v2d*>(dst + i));
  }
  __syncthreads();
}

__global__ __launch_bounds__(128) void k_evaluate_edges(DeviceGraph g, const double* poses, double* res, double* ja, double* jb) {
  __shared__ double stage[2][64 * 37];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* lds_w = stage[wave];
  const int e0 = blockIdx.x * blockDim.x + wave * 64;
  const int count = max(0, min(64, g.E - e0));
  const bool live = e < g.E;
  const size_t E = (size_t)g.E;
  double r[6] = {0, 0, 0, 0, 0, 0}, sc = 1.0;
  double L[36], Aa[36], Ab[36];
  uint8_t ma = 0, mb = 0;
#pragma unroll
  for (int k = 0; k < 36; ++k) { Aa[k] = 0.0; Ab[k] = 0.0; L[k] = (k % 7 == 0) ? 1.0 : 0.0; }
  if (live) {
    const int a = g.edge_a[e], b = g.edge_b[e];
    const PoseRec A = load_pose(poses, a), B = load_pose(poses, b);
    const V3 mp{g.emeas[e], g.emeas[E + e], g.emeas[2 * E + e]};
    const Q4 mq{g.emeas[3 * E + e], g.emeas[4 * E + e], g.emeas[5 * E + e], g.emeas[6 * E + e]};
    const EdgeGeom eg = edge_geometry(A.p, A.q, B.p, B.q, mp, mq);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Aa[6 * i + j] = -eg.Rt.m[3 * i + j];
        Aa[6 * i + 3 + j] = eg.G.m[3 * i + j];
        Aa[6 * (3 + i) + 3 + j] = 2.0 * eg.M.m[3 * i + j];
        Ab[6 * i + j] = eg.Rt.m[3 * i + j];
        Ab[6 * (3 + i) + 3 + j] = -2.0 * eg.M.m[3 * i + j];
      }
    if (g.eL) {
#pragma unroll
      for (int k = 0; k < 36; ++k) L[k] = g.eL[(size_t)k * E + e];
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) t += L[6 * i + j] * eg.e[j];
      r[i] = t;
      s += t * t;
    }
    double rho0, rho1;
    loss_eval(g.loss_kind, g.loss_a, s, &rho0, &rho1);
    sc = sqrt(rho1);
    ma = g.cmask[a];
    mb = g.cmask[b];
  }
  if (res) {
#pragma unroll
    for (int i = 0; i < 6; ++i) lds_w[lane * 37 + i] = sc * r[i];
    store_rows_coalesced<6>(lds_w, res, e0, count, lane);
  }
  if (ja) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double ta = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) ta += L[6 * i + j] * Aa[6 * j + c];
        const bool ca = (c < 3) ? (ma & 1) : (ma & 2);
        lds_w[lane * 37 + 6 * i + c] = ca ? 0.0 : sc * ta;
      }
    store_rows_coalesced<36>(lds_w, ja, e0, count, lane);
  }
  if (jb) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double tb = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) tb += L[6 * i + j] * Ab[6 * j + c];
        const bool cb = (c < 3) ? (mb & 1) : (mb & 2);
        lds_w[lane * 37 + 6 * i + c] = cb ? 0.0 : sc * tb;
      }
    store_rows_coalesced<36>(lds_w, jb, e0, count, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// Block-Jacobi preconditioned CG on (H~ + D^2) x = S g   [Ceres 1.13 ConjugateGradientsSolver with
// the Nash-Sofer Q-tolerance stop, SURVEY §7.2 #1].  Two kernels per iteration; all scalars are
// re-derived by every workgroup from per-workgroup partial sums (fixed order), so there is no host
// round trip, no atomics and no grid barrier inside the iteration.
// ------------------------------------------------------------------------------------------------
template <int CL>
__global__ void k_pcg_init(DeviceGraph g) {
  __shared__ double rl[VEC_BLOCK];
  __shared__ double scratch[2 * (VEC_BLOCK / 64)];
  const int tid = threadIdx.x;
  if (g.lm && (g.lm->halt || g.lm->phase == LM_PHASE_CONT)) return;   // device-resident LM: halted, or the previous sequence's CG goes on
  double acc[2] = {0.0, 0.0};
  for (int base = blockIdx.x * VEC_BLOCK; base < 6 * g.N; base += gridDim.x * VEC_BLOCK) {
    const int idx = base + tid;
    const bool live = idx < 6 * g.N;
    double b = 0.0;
    if (live) {
      b = g.scale[idx] * g.grad[idx];
      g.cg_b[idx] = b;
      g.cg_x[idx] = 0.0;
      g.cg_r[idx] = b;
      g.cg_p0[idx] = 0.0;
      g.cg_p1[idx] = 0.0;
    }
    rl[tid] = b;
    __syncthreads();
    if (live) {
      constexpr int DIM = 6 * CL;
      const double* Mi = g.Minv + (size_t)idx * DIM;      // row (idx mod DIM) of cluster idx / DIM
      const double* rv = rl + DIM * (tid / DIM);
      double z = 0.0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) z += Mi[k] * rv[k];
      g.cg_z[idx] = z;
      acc[0] += b * z;
      acc[1] += b * b;
    }
    __syncthreads();
  }
  block_sum<2>(acc, scratch);
  if (tid == 0) {
    g.part_rz[blockIdx.x] = acc[0];
    g.part_rz[g.n_part + blockIdx.x] = 0.0;
    g.part_bb[blockIdx.x] = acc[1];
    g.part_rr[blockIdx.x] = acc[1];
    g.part_rr[g.n_part + blockIdx.x] = acc[1];
    g.part_q[blockIdx.x] = 0.0;
    g.part_q[g.n_part + blockIdx.x] = 0.0;
    if (blockIdx.x == 0) {
      g.cg->done = 0; g.cg->iters = 0; g.cg->status = 0; g.cg->cnt_a = 0; g.cg->cnt_b = 0;
    }
  }
}

// Row-partitioned block SpMV: dst_row = sum_slots B_slot * src[col].  MODE 0: PCG step
// (src = z + beta p_old, also writes p_new, partial p'q).  MODE 1: plain q = A x for the model change.
//
// Latency structure (the kernel is latency- not bandwidth-bound at KITTI/Manhattan sizes): every load
// that does not depend on another load is issued at the top — CG state, BOTH parities of the partial-sum
// rows, the slot's column index and its whole 6x6 block — so that the prologue reduction overlaps the
// block fetch and only the z/p gathers (which need the column index) form a second round trip.
// `odd` is the parity of the CG iteration number, fixed at launch (batches have even length and
// start at iteration 1), so the ping-pong buffers are known without waiting for device state.
template <int MODE, bool PACKED>
__global__ __launch_bounds__(256) void k_spmv(DeviceGraph g, CgParams prm, int odd) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;   // 16-byte loads per slot (pgo_kernels.h: packed 27-entry slots)
  extern __shared__ double lds[];  // SPMV_LDS_STRIDE * block (slot results) + 6 * block (own-row p_new)
  __shared__ double scratch[32];   // >= 7 sums x 4 waves
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  double* lds_p = lds + (size_t)SPMV_LDS_STRIDE * B;

  if (MODE == 1 && (odd & 64) && !g.cg->done) return;  // step tail behind a batch of the owner-only CG (k_pipe_cg applied the stop test): only once the CG has stopped
  if (MODE == 1 && (odd & 8) && g.cg->done) return;   // A x of a residual refresh: nothing to do once the CG has stopped
  if (MODE == 1 && !(odd & 8) && lm_halted(g)) return;  // step tail of a sequence enqueued ahead of a halt
  if (MODE == 1 && (odd & 2)) {
    // Step tail behind a CG batch: this kernel first does what k_pcg_finish does (every workgroup evaluates the stop test
    // of the last completed iteration from the same partial rows, workgroup 0 publishes the state), and computes
    // q = A x only once the CG has stopped.
    int done = g.cg->done;
    const int iters0 = g.cg->iters, status = g.cg->status;
    const int it = done ? iters0 : g.cg->cnt_b;
    double fs[4];
    fs[0] = partial_sum(g.part_q + (size_t)(it & 1) * g.n_part, g.n_vec_wg);
    fs[1] = partial_sum(g.part_q + (size_t)((it + 1) & 1) * g.n_part, g.n_vec_wg);
    fs[2] = partial_sum(g.part_rr + (size_t)(it & 1) * g.n_part, g.n_vec_wg);
    fs[3] = partial_sum(g.part_bb, g.n_vec_wg);
    block_sum<4>(fs, scratch);
    const int was_done = done;
    if (!done && it >= 1) {
      const double Q1 = -fs[0], Q0 = -fs[1];
      const double zeta = it * (Q1 - Q0) / Q1;
      if ((zeta < prm.q_tolerance && it >= prm.min_iterations) || it >= prm.max_iterations) done = 1;
      if (prm.r_tolerance >= 0.0 && sqrt(fs[2]) <= prm.r_tolerance * sqrt(fs[3]) && it >= prm.min_iterations) done = 1;
    }
    if (wg == 0 && tid == 0) {
      if (done && !was_done) { g.cg->iters = it; __threadfence(); g.cg->done = 1; }
      g.scal->cg_iterations = it;
      g.scal->cg_status = done ? status : -1;
      g.scal->cg_residual_sq = fs[2];
    }
    if (!done) return;
  }
  const int s_begin = g.wg_slot_begin[wg], s_end = g.wg_slot_begin[wg + 1];
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  const bool single = (s_end - s_begin) == B;

  // ---- independent loads, first chunk ----
  const double* p_old = odd ? g.cg_p0 : g.cg_p1;
  double* p_new = odd ? g.cg_p1 : g.cg_p0;
  const double* src = g.cg_x;
  int t = s_begin + tid;
  int col = g.slot_col[t];
  int row = g.slot_row[t];
  uint8_t side = g.slot_side[t];
  double2 blk[NPAIR];
  {
    const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) blk[k] = bp[(size_t)k * 64];
  }
  // row bookkeeping of the segmented sum (first item of this thread)
  int seg_rb = 0, seg_cnt = 0;
  if (tid < nrows * 6) { seg_rb = g.row_slot_begin[r0 + tid / 6]; seg_cnt = g.row_slot_cnt[r0 + tid / 6]; }
  int done = 0, cnt_b = 0;
  double hist_rho = 0.0, hist_q = 0.0;
  double sums[6] = {0, 0, 0, 0, 0, 0};
  if (MODE == 0 && !PGO_ABLATION(g, 1)) {
    done = g.cg->done;
    cnt_b = g.cg->cnt_b;
    const double* rz_cur = g.part_rz + (size_t)(odd ? 0 : g.n_part);   // (it-1)&1
    const double* rz_prev = g.part_rz + (size_t)(odd ? g.n_part : 0);
    const double* q_cur = g.part_q + (size_t)(odd ? 0 : g.n_part);
    const double* q_prev = g.part_q + (size_t)(odd ? g.n_part : 0);
    const double* rr_cur = g.part_rr + (size_t)(odd ? 0 : g.n_part);
    const bool need_rr = prm.r_tolerance >= 0.0;   // |r| <= tol |b| is tested only by the exact-request (PCG to 1e-13) path
    (void)rz_prev; (void)q_prev;                   // rho and Q of the previous iteration come from CgState (see below)
    hist_rho = g.cg->rho_hist[odd ? 0 : 1];        // slot (it-1)&1... it is odd <=> `odd`: previous iteration is even -> slot 0
    hist_q = g.cg->q_hist[odd ? 0 : 1];
    if (need_rr) {
      for (int i = tid; i < g.n_vec_wg; i += B) {
        sums[0] += rz_cur[i]; sums[2] += q_cur[i]; sums[4] += rr_cur[i]; sums[5] += g.part_bb[i];
      }
    } else {
      // LM mode: every WAVE folds the two partial rows on its own (lane-strided, then DPP) — no LDS hop, no barriers
      for (int i = tid & 63; i < g.n_vec_wg; i += 64) { sums[0] += rz_cur[i]; sums[2] += q_cur[i]; }
    }
  }
  // residual refresh on one rank (flags 16 | parity 32): x of this iteration is not stored yet — gather x_old and p and
  // form x = x_old + alpha p on the fly, alpha = rho / p'q as the update kernel computes it
  const bool fly = MODE == 1 && (odd & 16);
  const double* p_cur = (odd & 32) ? g.cg_p1 : g.cg_p0;
  double alpha_fly = 0.0;
  if (fly) {
    double pqs[1] = {0.0};
    const double* pqp = g.cg_q + (size_t)g.rows_per * 6;
    for (int i = tid; i < g.pq_cap; i += B) pqs[0] += pqp[i];
    block_sum<1>(pqs, scratch);
    if (!(pqs[0] > 0.0) || !isfinite(pqs[0])) return;     // indefinite: the update kernel stops the CG
    alpha_fly = g.cg->rho / pqs[0];
  }
  // gathers of the first chunk: need only the column index
  double2 gz[3] = {{0, 0}, {0, 0}, {0, 0}}, gp[3] = {{0, 0}, {0, 0}, {0, 0}};
  if (col >= 0) {
    if (fly) {
      const double2* xs = reinterpret_cast<const double2*>(src + 6 * (size_t)col);
      const double2* ps = reinterpret_cast<const double2*>(p_cur + 6 * (size_t)col);
#pragma unroll
      for (int k = 0; k < 3; ++k) { gz[k] = xs[k]; gp[k] = ps[k]; }
    } else if (MODE == 0) {
      const double2* zs = reinterpret_cast<const double2*>(g.cg_z + 6 * (size_t)col);
      const double2* ps = reinterpret_cast<const double2*>(p_old + 6 * (size_t)col);
#pragma unroll
      for (int k = 0; k < 3; ++k) { gz[k] = zs[k]; gp[k] = ps[k]; }
    } else {
      const double2* xs = reinterpret_cast<const double2*>(src + 6 * (size_t)col);
#pragma unroll
      for (int k = 0; k < 3; ++k) gz[k] = xs[k];
    }
  }

  double beta = 0.0, rho_pub = 0.0, q_pub = 0.0;
  int it = 1;
  if (MODE == 0 && !PGO_ABLATION(g, 1)) {
    if (done) return;
    it = cnt_b + 1;
    double rr = 0.0, bb = 0.0;
    if (prm.r_tolerance >= 0.0) {
      block_sum<6>(sums, scratch);
      rr = sums[4]; bb = sums[5];
    } else {
      sums[0] = wave_sum(sums[0]);
      sums[2] = wave_sum(sums[2]);
    }
    const double rho = sums[0], rho_prev = hist_rho, Q1 = -sums[2], Q0 = hist_q;
    rho_pub = rho;
    q_pub = Q1;
    int stop = 0, status = 0;
    if (it > 1) {
      const int done_it = it - 1;
      const double zeta = done_it * (Q1 - Q0) / Q1;
      if (zeta < prm.q_tolerance && done_it >= prm.min_iterations) stop = 1;
      if (prm.r_tolerance >= 0.0 && sqrt(rr) <= prm.r_tolerance * sqrt(bb) && done_it >= prm.min_iterations) stop = 1;
      if (done_it >= prm.max_iterations) stop = 1;
    }
    if (!stop && (rho == 0.0 || !isfinite(rho))) { stop = 1; status = (rho == 0.0) ? 0 : 2; }
    if (!stop && it > 1) {
      beta = rho / rho_prev;
      if (beta == 0.0 || !isfinite(beta)) { stop = 1; status = 2; }
    }
    if (stop) {
      if (wg == 0 && tid == 0) { g.cg->iters = it - 1; g.cg->status = status; g.cg->done = 1; }
      return;
    }
  }

  double acc = 0.0;
  double pq[1] = {0.0};
  for (int cb = s_begin; cb < s_end; cb += B) {
    if (cb != s_begin) {  // further chunks of a fat row
      t = cb + tid;
      col = g.slot_col[t];
      row = g.slot_row[t];
      side = g.slot_side[t];
      const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) blk[k] = bp[(size_t)k * 64];
    }
    if (cb != s_begin && col >= 0) {
      if (fly) {
        const double2* xs = reinterpret_cast<const double2*>(src + 6 * (size_t)col);
        const double2* ps = reinterpret_cast<const double2*>(p_cur + 6 * (size_t)col);
#pragma unroll
        for (int k = 0; k < 3; ++k) { gz[k] = xs[k]; gp[k] = ps[k]; }
      } else if (MODE == 0) {
        const double2* zs = reinterpret_cast<const double2*>(g.cg_z + 6 * (size_t)col);
        const double2* ps = reinterpret_cast<const double2*>(p_old + 6 * (size_t)col);
#pragma unroll
        for (int k = 0; k < 3; ++k) { gz[k] = zs[k]; gp[k] = ps[k]; }
      } else {
        const double2* xs = reinterpret_cast<const double2*>(src + 6 * (size_t)col);
#pragma unroll
        for (int k = 0; k < 3; ++k) gz[k] = xs[k];
      }
    }
    double y[6] = {0, 0, 0, 0, 0, 0};
    if (col >= 0) {
      double x[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double f = (MODE == 0) ? beta : alpha_fly;      // MODE 1 without `fly`: gp = 0
        x[2 * k] = gz[k].x + f * gp[k].x;
        x[2 * k + 1] = gz[k].y + f * gp[k].y;
      }
      if (MODE == 1 && (odd & 4) && side == SIDE_DIAG && cb == s_begin) {
        // step tail: delta = -S x and the candidate Plus(x, delta) of this row (k_retract's job, one launch fewer)
        const PoseRec P = load_pose(g.pose_x, row);
        const uint8_t m = g.cmask[row];
        double d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const bool c = (i < 3) ? (m & 1) : (m & 2);
          d[i] = c ? 0.0 : -g.scale[6 * (size_t)row + i] * x[i];
          g.delta[6 * (size_t)row + i] = d[i];
        }
        V3 pc = P.p;
        Q4 qc = P.q;
        if (!(m & 1)) pc = V3{P.p.x + d[0], P.p.y + d[1], P.p.z + d[2]};
        if (!(m & 2)) qc = quat_plus(P.q, V3{d[3], d[4], d[5]});
        double2* o = reinterpret_cast<double2*>(g.pose_c + (size_t)POSE_STRIDE * row);
        o[0] = double2{pc.x, pc.y};
        o[1] = double2{pc.z, qc.x};
        o[2] = double2{qc.y, qc.z};
        o[3] = double2{qc.w, 0.0};
      }
      if (MODE == 0 && side == SIDE_DIAG) {
        double2* pn = reinterpret_cast<double2*>(p_new + 6 * (size_t)col);
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[k] = double2{x[2 * k], x[2 * k + 1]};
        if (cb == s_begin) {
          const int rl = row - r0;
#pragma unroll
          for (int k = 0; k < 6; ++k) lds_p[6 * rl + k] = x[k];
        }
      }
      if (!PGO_ABLATION(g, 2)) {
        if (PACKED) {
          // packed slot: TL = positions 0..8, BR = 9..17, Q = 18..26 (bottom-left for BEGIN / DIAG, top-right for END); the
          // row sums keep the expression of the full layout (entries that are structurally zero there are zero here), so
          // both layouts round identically
          double el[28];
#pragma unroll
          for (int k = 0; k < BLK_PAIRS_PACKED; ++k) { el[2 * k] = blk[k].x; el[2 * k + 1] = blk[k].y; }
          const bool is_end = side == SIDE_END, is_diag = side == SIDE_DIAG;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double a3 = is_end ? el[18 + 3 * i] : is_diag ? el[18 + i] : 0.0;
            const double a4 = is_end ? el[18 + 3 * i + 1] : is_diag ? el[21 + i] : 0.0;
            const double a5 = is_end ? el[18 + 3 * i + 2] : is_diag ? el[24 + i] : 0.0;
            y[i] = el[3 * i] * x[0] + el[3 * i + 1] * x[1] + el[3 * i + 2] * x[2] + a3 * x[3] + a4 * x[4] + a5 * x[5];
            const double b0 = is_end ? 0.0 : el[18 + 3 * i], b1 = is_end ? 0.0 : el[18 + 3 * i + 1], b2 = is_end ? 0.0 : el[18 + 3 * i + 2];
            y[3 + i] = b0 * x[0] + b1 * x[1] + b2 * x[2] + el[9 + 3 * i] * x[3] + el[9 + 3 * i + 1] * x[4] + el[9 + 3 * i + 2] * x[5];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i)
            y[i] = blk[3 * i].x * x[0] + blk[3 * i].y * x[1] + blk[3 * i + 1].x * x[2] + blk[3 * i + 1].y * x[3] +
                   blk[3 * i + 2].x * x[4] + blk[3 * i + 2].y * x[5];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) y[i] = x[i];
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
    __syncthreads();
    for (int idx = tid; idx < nrows * 6 && !PGO_ABLATION(g, 4); idx += B) {
      const int rl = idx / 6, k = idx - rl * 6;
      const int rw = r0 + rl;
      const int rb = (idx == tid) ? seg_rb : g.row_slot_begin[rw];
      const int rc = (idx == tid) ? seg_cnt : g.row_slot_cnt[rw];
      const int sb = max(rb, cb) - cb, se = min(rb + rc, cb + B) - cb;
      double s0 = 0.0, s1 = 0.0;
      int j = sb;
      for (; j + 1 < se; j += 2) { s0 += lds[j * SPMV_LDS_STRIDE + k]; s1 += lds[(j + 1) * SPMV_LDS_STRIDE + k]; }
      if (j < se) s0 += lds[j * SPMV_LDS_STRIDE + k];
      const double s = s0 + s1;
      if (single) {
        g.cg_q[q_index(g, rw, k)] = s;
        if (MODE == 0) pq[0] += s * lds_p[idx];
      } else {
        acc += s;
      }
    }
    __syncthreads();
  }
  if (!single && tid < 6) {
    g.cg_q[q_index(g, r0, tid)] = acc;
    if (MODE == 0) pq[0] += acc * lds_p[tid];
  }
  if (MODE == 0) {
    block_sum<1>(pq, scratch);
    if (tid == 0) {
      g.cg_q[(size_t)g.rank * g.seg + (size_t)g.rows_per * 6 + wg] = pq[0];   // p'q partial rides in the exchange segment
      if (wg == 0) {
        g.cg->cnt_a = it; g.cg->beta = beta; g.cg->rho = rho_pub;
        g.cg->rho_hist[odd ? 1 : 0] = rho_pub;   // slot it&1: read by the SpMV launch of iteration it+1
        g.cg->q_hist[odd ? 1 : 0] = q_pub;
      }
    }
  }
}

// x += alpha p ; r -= alpha q ; z = M^-1 r ; partial r'z, Q = x'(b + r), r'r.  All loads up front.
// mode 0: the normal iteration.  Every residual_reset_period-th iteration Ceres recomputes r = b - A x instead
// (conjugate_gradients_solver.cc).  Several ranks: mode 1 = only x += alpha p, then k_spmv<1> puts A x into cg_q, then
// mode 2 = r = b - cg_q, z, partial sums and the iteration counter.  One rank: k_spmv<1> forms x = x_old + alpha p on the
// fly and mode 3 = x += alpha p AND r = b - cg_q in one launch.
template <int CL>
__global__ __launch_bounds__(VEC_BLOCK) void k_pcg_update(DeviceGraph g, int odd, int mode) {
  constexpr int DIM = 6 * CL;
  __shared__ double rl[VEC_BLOCK];
  __shared__ double scratch[3 * (VEC_BLOCK / 64)];
  const int tid = threadIdx.x;
  const int m = 6 * g.N;
  // ---- independent loads ----
  int idx = blockIdx.x * VEC_BLOCK + tid;
  bool live = idx < m;
  // p of this iteration: the SpMV kernel wrote it for the rows this rank owns; with several ranks the vector update
  // is replicated over ALL rows, so p = z + beta p_old is rebuilt here from the (replicated) previous vectors.
  const bool rebuild_p = g.world > 1;
  double* p = odd ? g.cg_p1 : g.cg_p0;
  const double* p_prev = odd ? g.cg_p0 : g.cg_p1;
  double x0 = 0, pa = 0, zo = 0, r0 = 0, q0 = 0, b0 = 0;
  double2 mi[DIM / 2];
#pragma unroll
  for (int k = 0; k < DIM / 2; ++k) mi[k] = double2{0, 0};
  if (live) {
    x0 = g.cg_x[idx]; r0 = g.cg_r[idx]; q0 = g.cg_q[q_index_flat(g, idx)]; b0 = g.cg_b[idx];
    if (rebuild_p) { pa = p_prev[idx]; zo = g.cg_z[idx]; } else { pa = p[idx]; }
    const double2* Mi = reinterpret_cast<const double2*>(g.Minv + (size_t)idx * DIM);   // row (idx mod DIM) of cluster idx / DIM
#pragma unroll
    for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
  }
  const int done = g.cg->done;
  const int it = g.cg->cnt_a;
  const double beta = g.cg->beta;   // NOT re-derived here: this kernel overwrites the partial row rho_{it-1} lives in
  const double rho = g.cg->rho;     // r'z of this iteration as the SpMV kernel summed it (one reduction fewer here)
  double sums[1] = {0};
  if (g.world == 1) {
    const double* pqp = g.cg_q + (size_t)g.rows_per * 6;
    for (int i = tid; i < g.pq_cap; i += VEC_BLOCK) sums[0] += pqp[i];
  } else {
    for (int i = tid; i < g.world * g.pq_cap; i += VEC_BLOCK) {   // unused partial slots stay zero
      const int rk = i / g.pq_cap;
      sums[0] += g.cg_q[(size_t)rk * g.seg + (size_t)g.rows_per * 6 + (i - rk * g.pq_cap)];
    }
  }
  if (done) return;
  double alpha = 0.0;
  if (mode != 2) {
    block_sum<1>(sums, scratch);
    const double pq = sums[0];
    if (!(pq > 0.0) || !isfinite(pq)) {
      // "Matrix is indefinite, no more progress can be made": keep x of the previous iteration
      if (blockIdx.x == 0 && tid == 0) { g.cg->iters = it - 1; g.cg->status = 1; g.cg->done = 1; }
      return;
    }
    alpha = rho / pq;
  }
  double acc[3] = {0.0, 0.0, 0.0};
  for (int base = blockIdx.x * VEC_BLOCK; base < m; base += gridDim.x * VEC_BLOCK) {
    if (base != (int)blockIdx.x * VEC_BLOCK) {
      idx = base + tid;
      live = idx < m;
      if (live) {
        x0 = g.cg_x[idx]; r0 = g.cg_r[idx]; q0 = g.cg_q[q_index_flat(g, idx)]; b0 = g.cg_b[idx];
        if (rebuild_p) { pa = p_prev[idx]; zo = g.cg_z[idx]; } else { pa = p[idx]; }
        const double2* Mi = reinterpret_cast<const double2*>(g.Minv + (size_t)idx * DIM);
#pragma unroll
        for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
      }
    }
    double x = 0.0, r = 0.0;
    if (live) {
      if (mode == 2) {              // refresh: cg_q holds A x
        x = x0;
        r = b0 - q0;
        g.cg_r[idx] = r;
      } else if (mode == 3) {       // refresh, one rank: cg_q holds A (x + alpha p)
        x = x0 + alpha * pa;
        r = b0 - q0;
        g.cg_x[idx] = x;
        g.cg_r[idx] = r;
      } else {
        if (rebuild_p) { pa = zo + beta * pa; p[idx] = pa; }
        x = x0 + alpha * pa;
        g.cg_x[idx] = x;
        if (mode == 0) { r = r0 - alpha * q0; g.cg_r[idx] = r; }
      }
    }
    if (mode == 1) continue;        // x only: r, z and the partial sums follow in the mode-2 launch
    rl[tid] = r;
    __syncthreads();
    if (live) {
      const double* rv = rl + DIM * (tid / DIM);
      double z = 0.0;
#pragma unroll
      for (int k = 0; k < DIM / 2; ++k) z += mi[k].x * rv[2 * k] + mi[k].y * rv[2 * k + 1];
      g.cg_z[idx] = z;
      acc[0] += r * z;
      acc[1] += x * (b0 + r);
      acc[2] += r * r;
    }
    __syncthreads();
  }
  if (mode == 1) return;
  block_sum<3>(acc, scratch);
  if (tid == 0) {
    const size_t o = (size_t)(odd ? g.n_part : 0) + blockIdx.x;
    g.part_rz[o] = acc[0];
    g.part_q[o] = acc[1];
    g.part_rr[o] = acc[2];
    if (blockIdx.x == 0) g.cg->cnt_b = it;
  }
}

// Host hand-off without a stream synchronise: the host clears LmScalars::seq before it enqueues a sequence, the last
// kernel of the sequence sets it AFTER its results (system-scope fence), and the host spins on that word.
__device__ __forceinline__ void publish_sequence(const DeviceGraph& g) {
  __threadfence_system();
  __hip_atomic_store(&g.scal->seq, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Runs after a batch of iterations: applies the termination test to the last completed iteration
// (the SpMV kernel of the next iteration would do it) and publishes the CG state for the host.
__global__ void k_pcg_finish(DeviceGraph g, CgParams prm, int publish) {
  __shared__ double scratch[16];
  const int tid = threadIdx.x;
  int done = g.cg->done, iters = g.cg->iters, status = g.cg->status;
  const int it = done ? iters : g.cg->cnt_b;  // completed iterations
  double sums[4];
  sums[0] = partial_sum(g.part_q + (size_t)(it & 1) * g.n_part, g.n_vec_wg);
  sums[1] = partial_sum(g.part_q + (size_t)((it + 1) & 1) * g.n_part, g.n_vec_wg);
  sums[2] = partial_sum(g.part_rr + (size_t)(it & 1) * g.n_part, g.n_vec_wg);
  sums[3] = partial_sum(g.part_bb, g.n_vec_wg);
  block_sum<4>(sums, scratch);
  if (!done) {
    iters = it;
    if (it >= 1) {
      const double Q1 = -sums[0], Q0 = -sums[1];
      const double zeta = it * (Q1 - Q0) / Q1;
      if ((zeta < prm.q_tolerance && it >= prm.min_iterations) || it >= prm.max_iterations) done = 1;
      if (prm.r_tolerance >= 0.0 && sqrt(sums[2]) <= prm.r_tolerance * sqrt(sums[3]) && it >= prm.min_iterations) done = 1;
    }
  }
  if (tid == 0) {
    if (done && !g.cg->done) { g.cg->done = 1; g.cg->iters = iters; }
    g.scal->cg_iterations = iters;
    g.scal->cg_status = done ? status : -1;  // -1: not finished, host launches another batch
    g.scal->cg_residual_sq = sums[2];
    if (publish) publish_sequence(g);
  }
}

// model_cost_change = -(J~ step)'(r + J~ step/2) with step = -x:  x'b - x'(q - D^2 x)/2, q = A x;
// delta = S * step.
// `gate`: the step tail rides behind every CG batch; it runs only once the CG has stopped (device flag).
__global__ void k_model_delta(DeviceGraph g, int gate) {
  __shared__ double scratch[VEC_BLOCK / 64];
  if (gate && !g.cg->done) return;
  double acc[1] = {0.0};
  for (int idx = blockIdx.x * VEC_BLOCK + threadIdx.x; idx < 6 * g.N; idx += gridDim.x * VEC_BLOCK) {
    const double x = g.cg_x[idx];
    const double hx = g.cg_q[q_index_flat(g, idx)] - g.d2[idx] * x;
    const int v = idx / 6, i = idx - 6 * v;
    const bool c = (i < 3) ? (g.cmask[v] & 1) : (g.cmask[v] & 2);
    acc[0] += c ? 0.0 : (x * g.cg_b[idx] - 0.5 * x * hx);
    g.delta[idx] = c ? 0.0 : -g.scale[idx] * x;
  }
  block_sum<1>(acc, scratch);
  if (threadIdx.x == 0) g.part_misc[1 * (size_t)g.n_part + blockIdx.x] = acc[0];
}

// x_cand = Plus(x, delta) per pose (p += dp ; q <- exp(dtheta) (x) q); ambient step / state norms.
__global__ void k_retract(DeviceGraph g, int gate) {
  __shared__ double scratch[16];
  if (gate && !g.cg->done) return;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  double acc[2] = {0.0, 0.0};
  if (v < g.N) {
    const PoseRec P = load_pose(g.pose_x, v);
    const uint8_t m = g.cmask[v];
    const double* d = g.delta + 6 * (size_t)v;
    V3 p = P.p;
    Q4 q = P.q;
    if (!(m & 1)) {
      p = V3{P.p.x + d[0], P.p.y + d[1], P.p.z + d[2]};
      const double dx = P.p.x - p.x, dy = P.p.y - p.y, dz = P.p.z - p.z;
      acc[0] += dx * dx + dy * dy + dz * dz;
      acc[1] += P.p.x * P.p.x + P.p.y * P.p.y + P.p.z * P.p.z;
    }
    if (!(m & 2)) {
      q = quat_plus(P.q, V3{d[3], d[4], d[5]});
      const double dx = P.q.x - q.x, dy = P.q.y - q.y, dz = P.q.z - q.z, dw = P.q.w - q.w;
      acc[0] += dx * dx + dy * dy + dz * dz + dw * dw;
      acc[1] += P.q.x * P.q.x + P.q.y * P.q.y + P.q.z * P.q.z + P.q.w * P.q.w;
    }
    double2* o = reinterpret_cast<double2*>(g.pose_c + (size_t)POSE_STRIDE * v);
    o[0] = double2{p.x, p.y};
    o[1] = double2{p.z, q.x};
    o[2] = double2{q.y, q.z};
    o[3] = double2{q.w, 0.0};
  }
  block_sum<2>(acc, scratch);
  if (threadIdx.x == 0) {
    g.part_misc[2 * (size_t)g.n_part + blockIdx.x] = acc[0];
    g.part_misc[3 * (size_t)g.n_part + blockIdx.x] = acc[1];
  }
}

// gradient_max_norm = | x - Plus(x, -g) |_inf over the non-constant blocks.
__global__ void k_gradient_norm(DeviceGraph g) {
  __shared__ double scratch[8];
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  double m = 0.0;
  if (v < g.N) {
    const PoseRec P = load_pose(g.pose_x, v);
    const uint8_t cm = g.cmask[v];
    const double* gr = g.grad + 6 * (size_t)v;
    if (!(cm & 1)) m = fmax(m, fmax(fabs(gr[0]), fmax(fabs(gr[1]), fabs(gr[2]))));
    if (!(cm & 2)) {
      const Q4 q = quat_plus(P.q, V3{-gr[3], -gr[4], -gr[5]});
      m = fmax(m, fmax(fmax(fabs(P.q.x - q.x), fabs(P.q.y - q.y)), fmax(fabs(P.q.z - q.z), fabs(P.q.w - q.w))));
    }
  }
  m = wave_max(m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmax(t, scratch[w]);
    g.part_misc[4 * (size_t)g.n_part + blockIdx.x] = t;
  }
}

// One workgroup folds the partial rows into the pinned scalar block the host reads.
__global__ void k_finalize_scalars(DeviceGraph g, int n_cost_part, int gate) {
  __shared__ double scratch[32];
  if (gate && !g.cg->done) {   // CG still running: only hand the (unfinished) status over to the host
    if (threadIdx.x == 0) publish_sequence(g);
    return;
  }
  double s[4];
  s[0] = partial_sum(g.part_misc, n_cost_part);
  s[1] = partial_sum(g.part_misc + 1 * (size_t)g.n_part, g.n_vec_wg);
  s[2] = partial_sum(g.part_misc + 2 * (size_t)g.n_part, g.n_pose_wg);
  s[3] = partial_sum(g.part_misc + 3 * (size_t)g.n_part, g.n_pose_wg);
  block_sum<4>(s, scratch);
  double m = 0.0;
  for (int i = threadIdx.x; i < g.n_pose_wg; i += blockDim.x) m = fmax(m, g.part_misc[4 * (size_t)g.n_part + i]);
  m = wave_max(m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch[wave] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmax(t, scratch[w]);
    g.scal->cand_cost = s[0];
    g.scal->model_change = s[1];
    g.scal->step_norm_sq = s[2];
    g.scal->x_norm_sq = s[3];
    g.scal->gradient_max = t;
    g.scal->linearize_bad = g.flags[1] | (g.flags[2] << 1);  // bit0: block inverse failed, bit1: Cholesky pivot
    g.flags[1] = 0;
    g.flags[2] = 0;
    publish_sequence(g);
  }
}

// ---- device-resident LM: the decision (pgo_kernels.h LmDev, pgo_lm_rules.h) ---------------------------------------------
// Everything below runs on ONE lane (the last work-group of the step tail / of the accept-finish kernel).
__device__ __forceinline__ void lm_mirror(const DeviceGraph& g) {   // device state -> pinned mirror, then the words the host polls
  LmDev& D = *g.lm;
  g.scal->lm = D;
  g.scal->last_cg = D.last_cg;
  __threadfence_system();
  __hip_atomic_store(&g.scal->lm_done, D.lm_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&g.scal->halt, D.halt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void lm_terminate(LmDev& D, int termination, int reason, double value) {
  D.halt = LM_HALT_TERMINATED; D.termination = termination; D.reason = reason; D.term_value = value; D.accepted = 0;
}
// FinalizeIterationAndCheckIfMinimizerCanContinue of the NEXT pass (SURVEY A.6 step 7 order: maximum iterations, gradient
// tolerance — behind a successful step only —, minimum radius), evaluated as soon as its inputs exist.
__device__ __forceinline__ void lm_pre_step_checks(LmDev& D, bool successful) {
  if (D.core.iteration >= D.tol.max_num_iterations) lm_terminate(D, 1 /*NO_CONVERGENCE*/, 5, (double)D.core.iteration);
  else if (successful && D.core.gmax <= D.tol.gradient_tolerance) lm_terminate(D, 0 /*CONVERGENCE*/, 3, D.core.gmax);
  else if (D.core.radius <= D.tol.min_radius) lm_terminate(D, 0, 4, D.core.radius);
}
// The CG of this sequence has not stopped: it goes on in the next sequence if the refresh launches line up (a multiple of the
// refresh period has been completed), otherwise the host has to enqueue the continuation.
__device__ __forceinline__ void lm_cg_unfinished(const DeviceGraph& g) {
  LmDev& D = *g.lm;
  const int completed = g.cg->cnt_b;
  // (and an EVEN number of iterations: a continuation sequence is enqueued as iterations 1, 2, ... and the ping-pong buffer of p
  // is chosen by that launch-time parity, so the absolute iteration completed + 1 has to be odd as well)
  if ((completed & 1) == 0 && (D.cg_period <= 0 || completed % D.cg_period == 0)) {
    D.phase = LM_PHASE_CONT;
  } else {
    D.halt = LM_HALT_CG_STALL;
    g.cg->done = 2;            // the CG kernels of the sequences already enqueued exit
  }
  D.accepted = 0;
  lm_mirror(g);
}
// cg_iters / cg_status >= 0: the caller knows how the CG ended (the resident stream decides in the launch that ran the CG: the CG state
// words may sit, stale, in the deciding compute unit's L1); otherwise they are read from the CG state
__device__ __forceinline__ void lm_device_decide(const DeviceGraph& g, double cand_cost, double model_change, double step_norm_sq,
                                                 double x_norm_sq, int bad, int direct, int cg_iters = -1, int cg_status = -1) {
  LmDev& D = *g.lm;
  const long long now = (long long)__builtin_amdgcn_s_memrealtime();
  D.ticks_linear += now - D.t_mark;
  D.t_mark = now;
  D.accepted = 0;
  D.phase = LM_PHASE_NEW;
  if (bad & 4) {               // an in-kernel wait of a single-launch factorisation ran out: not a numerical event, the host repeats
    D.halt = LM_HALT_REFACTOR; //   this iteration with the factorisation in its launch-per-step form
    lm_mirror(g);
    return;
  }
  const LmStepIn in{cand_cost, model_change, step_norm_sq, x_norm_sq, (direct & 1) ? 0 : cg_iters >= 0 ? cg_iters : g.cg->iters,
                    (direct & 1) ? 0 : cg_iters >= 0 ? cg_status : g.cg->status, bad, 0};
  LmRecord nx;
  double tv = 0.0;
  const LmOutcome out = lm_decide(D.core, D.tol, in, nx, tv);
  D.lm_done += 1;
  D.num_linear_iterations += in.cg_iterations;
  D.last_cg = in.cg_iterations;
  bool record = true;
  switch (out) {
    case LM_OUT_INVALID_FAIL: lm_terminate(D, 2 /*FAILURE*/, 6, 0.0); record = false; break;
    case LM_OUT_PARAM_TOL: lm_terminate(D, 0, 2, tv); record = false; break;
    case LM_OUT_FUNC_TOL: lm_terminate(D, 0, 1, tv); record = false; break;
    case LM_OUT_ACCEPT: D.num_successful += 1; D.accepted = 1; break;
    default: D.num_unsuccessful += 1; break;
  }
  if (record) {
    g.scal->ring[nx.iteration % LM_RING] = nx;
    D.num_records = nx.iteration + 1;
    if (out != LM_OUT_ACCEPT) lm_pre_step_checks(D, false);   // behind an accepted step the accept-finish kernel runs them (it has the gradient)
  }
  if (direct & 2) {            // universal stream: what the next slot-shaped and vector-shaped launches do
    if (D.halt) { g.cg->op_s = UNI_EXIT; g.cg->op_v = UNI_EXIT; }
    else {
      g.cg->op_s = D.accepted ? UNI_S_LINEARIZE : UNI_NOP;
      g.cg->op_v = UNI_V_HEAD;
      if (D.lm_done >= D.decision_limit) D.pause = 1;   // pgo_solver_step(n): the head launch only finishes the accepted step, then the stream pauses
    }
  }
  if (!(direct & 8)) lm_mirror(g);      // (8: the caller publishes the state later — the fused stream, from its next launch)
}

// Fused step tail (one launch instead of k_model_delta + k_cost + k_finalize_scalars; every kernel boundary costs ~4 us
// at pose-graph sizes).  The candidate poses were written by the preceding k_spmv<1> launch (diagonal lanes).  Pose
// part = model cost change and step / state norms; edge part = candidate cost; the LAST workgroup to finish (ticket
// counter) folds the partial rows and hands off to the host.
// bid / nblocks: this work-group's index among the n_edge_wg + n_pose_wg that take part; EDGE_BLOCK live threads (the universal
// vector kernel runs it on the first four waves of its work-groups)
template <int INFO>
__device__ __forceinline__ void step_tail_body(const DeviceGraph& g, int gate, int bid, int nblocks, double* scratch, int* is_last_p) {
  const int tid = threadIdx.x;
  int& is_last = *is_last_p;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // candidate cost, model change, |step|^2, |x|^2
  // workgroups [0, n_edge_wg) take the edges, [n_edge_wg, n_edge_wg + n_pose_wg) the poses: both parts run side by side
  const int pose_wg = bid - g.n_edge_wg;
  for (int v = pose_wg * EDGE_BLOCK + tid; pose_wg >= 0 && v < g.N; v += g.n_pose_wg * EDGE_BLOCK) {
    const PoseRec P = load_pose(g.pose_x, v), C = load_pose(g.pose_c, v);
    const uint8_t m = g.cmask[v];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const size_t idx = 6 * (size_t)v + i;
      const double x = g.cg_x[idx];
      const double hx = g.cg_q[q_index(g, v, i)] - g.d2[idx] * x;
      const bool c = (i < 3) ? (m & 1) : (m & 2);
      acc[1] += c ? 0.0 : (x * g.cg_b[idx] - 0.5 * x * hx);
    }
    if (!(m & 1)) {
      const double dx = P.p.x - C.p.x, dy = P.p.y - C.p.y, dz = P.p.z - C.p.z;
      acc[2] += dx * dx + dy * dy + dz * dz;
      acc[3] += P.p.x * P.p.x + P.p.y * P.p.y + P.p.z * P.p.z;
    }
    if (!(m & 2)) {
      const double dx = P.q.x - C.q.x, dy = P.q.y - C.q.y, dz = P.q.z - C.q.z, dw = P.q.w - C.q.w;
      acc[2] += dx * dx + dy * dy + dz * dz + dw * dw;
      acc[3] += P.q.x * P.q.x + P.q.y * P.q.y + P.q.z * P.q.z + P.q.w * P.q.w;
    }
  }
  // (r06: four edges of the lane's stride in flight at once — clamped indices, so every load is issued unconditionally in front of the
  // first wait — and added in the order the one-by-one loop added them: the same bits, a quarter of the dependent round trips; C4: 94 -> us)
  if (pose_wg < 0) {
    const int stride = g.n_edge_wg * EDGE_BLOCK;
    for (int e = bid * EDGE_BLOCK + tid; e < g.E; e += 4 * stride) {
      double c4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = edge_cost<INFO>(g, g.pose_c, min(e + k * stride, g.E - 1));
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[0] += (e + k * stride < g.E) ? c4[k] : 0.0;
    }
  }
  block_sum_w<4>(acc, scratch, EDGE_BLOCK / 64);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) g.part_misc[(size_t)k * g.n_part + bid] = acc[k];
    __threadfence();
    is_last = (atomicAdd(&g.flags[3], 1) == nblocks - 1);
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < nblocks; i += EDGE_BLOCK) {   // other workgroups' partials: read at device scope (not from this CU's L1)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += __hip_atomic_load(&g.part_misc[(size_t)k * g.n_part + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  block_sum_w<4>(s, scratch, EDGE_BLOCK / 64);
  double m = 0.0;
  for (int i = tid; i < g.n_pose_wg; i += EDGE_BLOCK) m = fmax(m, g.part_misc[4 * (size_t)g.n_part + i]);
  m = wave_max(m);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) scratch[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < EDGE_BLOCK / 64; ++w) t = fmax(t, scratch[w]);
    g.scal->cand_cost = s[0];
    g.scal->model_change = s[1];
    g.scal->step_norm_sq = s[2];
    g.scal->x_norm_sq = s[3];
    g.scal->gradient_max = t;
    const int bad = g.flags[1] | (g.flags[2] << 1);
    g.scal->linearize_bad = bad;
    g.flags[1] = 0;
    g.flags[2] = 0;
    g.flags[3] = 0;
    if (g.lm) lm_device_decide(g, s[0], s[1], s[2], s[3], bad, ((gate & 2) ? 1 : 0) | ((gate & 4) ? 2 : 0));   // accept / reject / stop, on the spot
    else publish_sequence(g);
  }
}

template <int INFO>
__global__ __launch_bounds__(EDGE_BLOCK) void k_step_tail(DeviceGraph g, int gate) {
  __shared__ double scratch[4 * (EDGE_BLOCK / 64)];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  if (lm_halted(g)) return;    // device-resident LM: a sequence enqueued ahead of a halt
  if ((gate & 1) && !g.cg->done) {   // CG still running: only hand the (unfinished) status over to the host
    if (blockIdx.x == 0 && tid == 0) { if (g.lm) lm_cg_unfinished(g); else publish_sequence(g); }
    return;
  }
  step_tail_body<INFO>(g, gate, (int)blockIdx.x, (int)gridDim.x, scratch, &is_last);
}

// Device-resident LM: last kernel of a sequence.  Behind an accepted step: candidate -> current point (the linearisation that ran
// before it read the candidate buffer), gradient_max_norm = |x - Plus(x, -g)|_inf of the new point (k_gradient_norm's job), and the
// last work-group to finish applies the tests that open the next pass (maximum iterations, gradient tolerance, minimum radius).
// Whatever happened, ONE lane tells the host that sequence `seq_id` is through.
__global__ __launch_bounds__(256) void k_accept_finish(DeviceGraph g, int seq_id) {
  __shared__ double scratch[8];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  LmDev& D = *g.lm;
  if (D.halt || !D.accepted) {
    if (blockIdx.x == 0 && tid == 0) {
      __threadfence_system();
      __hip_atomic_store(&g.scal->seq_done, seq_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const int v = blockIdx.x * blockDim.x + tid;
  double m = 0.0;
  if (v < g.N) {
    const PoseRec P = load_pose(g.pose_c, v);
    const double2* src = reinterpret_cast<const double2*>(g.pose_c + (size_t)POSE_STRIDE * v);
    double2* dst = reinterpret_cast<double2*>(g.pose_x + (size_t)POSE_STRIDE * v);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    const uint8_t cm = g.cmask[v];
    const double* gr = g.grad + 6 * (size_t)v;
    if (!(cm & 1)) m = fmax(m, fmax(fabs(gr[0]), fmax(fabs(gr[1]), fabs(gr[2]))));
    if (!(cm & 2)) {
      const Q4 q = quat_plus(P.q, V3{-gr[3], -gr[4], -gr[5]});
      m = fmax(m, fmax(fmax(fabs(P.q.x - q.x), fabs(P.q.y - q.y)), fmax(fabs(P.q.z - q.z), fabs(P.q.w - q.w))));
    }
  }
  m = wave_max(m);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) scratch[wave] = m;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmax(t, scratch[w]);
    g.part_misc[4 * (size_t)g.n_part + blockIdx.x] = t;
    __threadfence();
    is_last = (atomicAdd(&g.flags[3], 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  double mm = 0.0;
  for (int i = tid; i < (int)gridDim.x; i += blockDim.x)
    mm = fmax(mm, __hip_atomic_load(&g.part_misc[4 * (size_t)g.n_part + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  mm = wave_max(mm);
  if (lane == 0) scratch[wave] = mm;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmax(t, scratch[w]);
    g.flags[3] = 0;
    D.core.gmax = t;
    g.scal->ring[D.core.iteration % LM_RING].gradient_max_norm = t;
    g.scal->gradient_max = t;
    lm_pre_step_checks(D, true);
    const long long now = (long long)__builtin_amdgcn_s_memrealtime();
    D.ticks_jacobian += now - D.t_mark;
    D.t_mark = now;
    lm_mirror(g);
    __hip_atomic_store(&g.scal->seq_done, seq_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Device-resident LM: the host steps in behind a halt that is not a termination (LM_HALT_CG_STALL: it enqueues the rest of the
// CG itself; LM_HALT_REFACTOR: it repeats the factorisation in another form) and lets the sequences run again.
__global__ void k_lm_resume(DeviceGraph g, int cg_goes_on) {
  LmDev& D = *g.lm;
  D.t_mark = (long long)__builtin_amdgcn_s_memrealtime();   // the phase clocks do not count the time the stream sat idle
  if (cg_goes_on < 0) return;                               // (start of a run: only the time mark)
  D.halt = LM_RUN;
  D.phase = cg_goes_on ? LM_PHASE_CONT : LM_PHASE_NEW;
  if (cg_goes_on) g.cg->done = 0;
  g.scal->halt = 0;
}


// ------------------------------------------------------------------------------------------------
// The universal stream (pgo_kernels.h UniOp): two kernels, enqueued alternately V S V S ... by a host that does not know what
// each launch will do.  Every branch below is the arithmetic of the kernel it replaces (k_spmv, k_linearize, k_pcg_update,
// k_cluster_precond / k_damping + k_pcg_init, k_step_tail, k_accept_finish), in the same work decomposition and the same
// summation orders, so a solve through the stream is bit-identical to the same solve through those kernels
// (tests/test_gpu_pipeline.py).  What differs is WHEN things are known: the parity of the CG iteration and the operation itself
// come from device memory, so everything the likely branch needs and that does not depend on them is requested before the state
// arrives (both parities of the partial rows and of p), and the state costs no round trip of its own.
// ------------------------------------------------------------------------------------------------
constexpr int UNI_V_BLOCK = 512;   // vector-shaped launches: 6 waves of vector update (VEC_BLOCK rows), 4 waves of step tail / accept-finish
                                   // (EDGE_BLOCK / POSE_BLOCK), up to 8 waves of cluster inverses for the 64 poses of a row chunk

// profiling aid, see DeviceGraph::oplog (one entry per launch, written by work-group 0)
__device__ __forceinline__ void uni_oplog(const DeviceGraph& g, int what) {
  if (g.oplog && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long i = g.oplog[0];
    if (i + 1 < g.oplog_cap) { g.oplog[1 + i] = ((long long)__builtin_amdgcn_s_memrealtime() << 3) | (long long)what; g.oplog[0] = i + 1; }
  }
}

template <bool PACKED, int INFO>
__global__ __launch_bounds__(256) void k_uni_s(DeviceGraph g, CgParams prm, int period) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  extern __shared__ double lds[];  // NV_LIN * block (linearise); the SpMV uses (SPMV_LDS_STRIDE + 6) * block of it
  __shared__ double scratch[32];
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  double* lds_p = lds + (size_t)SPMV_LDS_STRIDE * B;
  // ---- requested before the state is known: what the CG / tail / refresh SpMV needs of its first chunk ----
  const int s_begin = g.wg_slot_begin[wg], s_end = g.wg_slot_begin[wg + 1];
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  const bool single = (s_end - s_begin) == B;
  int t = s_begin + tid;
  int col = g.slot_col[t];
  int row = g.slot_row[t];
  uint8_t side = g.slot_side[t];
  double2 blk[NPAIR];
  {
    const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) blk[k] = bp[(size_t)k * 64];
  }
  int seg_rb = 0, seg_cnt = 0;
  if (tid < nrows * 6) { seg_rb = g.row_slot_begin[r0 + tid / 6]; seg_cnt = g.row_slot_cnt[r0 + tid / 6]; }
  const int op = g.cg->op_s;
  const int done0 = g.cg->done, cnt_b = g.cg->cnt_b;
  const double hr0 = g.cg->rho_hist[0], hr1 = g.cg->rho_hist[1], hq0 = g.cg->q_hist[0], hq1 = g.cg->q_hist[1];
  const bool need_rr = prm.r_tolerance >= 0.0;
  double se[6] = {0, 0, 0, 0, 0, 0}, so[6] = {0, 0, 0, 0, 0, 0};   // partial rows as an even / an odd iteration would read them
  if (need_rr) {
    for (int i = tid; i < g.n_vec_wg; i += B) {
      so[0] += g.part_rz[i]; so[2] += g.part_q[i]; so[4] += g.part_rr[i];
      se[0] += g.part_rz[g.n_part + i]; se[2] += g.part_q[g.n_part + i]; se[4] += g.part_rr[g.n_part + i];
      const double bb = g.part_bb[i];
      so[5] += bb; se[5] += bb;
    }
  } else {
    for (int i = tid & 63; i < g.n_vec_wg; i += 64) {
      so[0] += g.part_rz[i]; so[2] += g.part_q[i];
      se[0] += g.part_rz[g.n_part + i]; se[2] += g.part_q[g.n_part + i];
    }
  }
  if (op <= UNI_NOP) { uni_oplog(g, 0); return; }
  if (op == UNI_S_LINEARIZE) {
    uni_oplog(g, UNI_S_LINEARIZE);
    DeviceGraph gl = g;
    gl.pose_x = g.pose_c;          // the accepted candidate; the accept-finish part of the next head launch copies it over
    linearize_any<INFO>(gl, lds);
    return;
  }
  // ---- CG iteration `it`: the stop test of iteration it - 1 first (k_spmv<0>'s prologue), by every work-group alike ----
  const int it = cnt_b + 1;
  const int odd = it & 1;
  bool tail = false;           // this launch multiplies A x for the step tail instead of A p
  double beta = 0.0, rho_pub = 0.0, q_pub = 0.0;
  if (op == UNI_S_CG) {
    double sums[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sums[k] = odd ? so[k] : se[k];
    const double hist_rho = odd ? hr0 : hr1, hist_q = odd ? hq0 : hq1;
    int stop = 0, status = 0;
    if (done0) {
      tail = true;
    } else {
      double rr = 0.0, bb = 0.0;
      if (need_rr) {
        block_sum<6>(sums, scratch);
        rr = sums[4]; bb = sums[5];
      } else {
        sums[0] = wave_sum(sums[0]);
        sums[2] = wave_sum(sums[2]);
      }
      const double rho = sums[0], rho_prev = hist_rho, Q1 = -sums[2], Q0 = hist_q;
      rho_pub = rho;
      q_pub = Q1;
      if (it > 1) {
        const int done_it = it - 1;
        const double zeta = done_it * (Q1 - Q0) / Q1;
        if (zeta < prm.q_tolerance && done_it >= prm.min_iterations) stop = 1;
        if (prm.r_tolerance >= 0.0 && sqrt(rr) <= prm.r_tolerance * sqrt(bb) && done_it >= prm.min_iterations) stop = 1;
        if (done_it >= prm.max_iterations) stop = 1;
      }
      if (!stop && (rho == 0.0 || !isfinite(rho))) { stop = 1; status = (rho == 0.0) ? 0 : 2; }
      if (!stop && it > 1) {
        beta = rho / rho_prev;
        if (beta == 0.0 || !isfinite(beta)) { stop = 1; status = 2; }
      }
      if (stop) {
        tail = true;
        if (wg == 0 && tid == 0) { g.cg->iters = it - 1; g.cg->status = status; g.cg->done = 1; }
      }
    }
  }
  const bool cg_step = op == UNI_S_CG && !tail;
  uni_oplog(g, cg_step ? UNI_S_CG : tail ? 4 : op);     // 1 CG product, 2 refresh product, 4 tail product (A x of the step tail)
  const double* p_old = odd ? g.cg_p0 : g.cg_p1;
  double* p_new = odd ? g.cg_p1 : g.cg_p0;
  // gathers of the first chunk
  double2 gz[3] = {{0, 0}, {0, 0}, {0, 0}}, gp[3] = {{0, 0}, {0, 0}, {0, 0}};
  if (col >= 0) {
    if (cg_step) {
      const double2* zs = reinterpret_cast<const double2*>(g.cg_z + 6 * (size_t)col);
      const double2* ps = reinterpret_cast<const double2*>(p_old + 6 * (size_t)col);
#pragma unroll
      for (int k = 0; k < 3; ++k) { gz[k] = zs[k]; gp[k] = ps[k]; }
    } else {
      const double2* xs = reinterpret_cast<const double2*>(g.cg_x + 6 * (size_t)col);
#pragma unroll
      for (int k = 0; k < 3; ++k) gz[k] = xs[k];
    }
  }
  double acc = 0.0;
  double pq[1] = {0.0};
  for (int cb = s_begin; cb < s_end; cb += B) {
    if (cb != s_begin) {  // further chunks of a fat row
      t = cb + tid;
      col = g.slot_col[t];
      row = g.slot_row[t];
      side = g.slot_side[t];
      const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) blk[k] = bp[(size_t)k * 64];
      if (col >= 0) {
        if (cg_step) {
          const double2* zs = reinterpret_cast<const double2*>(g.cg_z + 6 * (size_t)col);
          const double2* ps = reinterpret_cast<const double2*>(p_old + 6 * (size_t)col);
#pragma unroll
          for (int k = 0; k < 3; ++k) { gz[k] = zs[k]; gp[k] = ps[k]; }
        } else {
          const double2* xs = reinterpret_cast<const double2*>(g.cg_x + 6 * (size_t)col);
#pragma unroll
          for (int k = 0; k < 3; ++k) { gz[k] = xs[k]; gp[k] = double2{0, 0}; }
        }
      }
    }
    double y[6] = {0, 0, 0, 0, 0, 0};
    if (col >= 0) {
      double x[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double f = cg_step ? beta : 0.0;
        x[2 * k] = gz[k].x + f * gp[k].x;
        x[2 * k + 1] = gz[k].y + f * gp[k].y;
      }
      if (tail && side == SIDE_DIAG && cb == s_begin) {
        // step tail: delta = -S x and the candidate Plus(x, delta) of this row
        const PoseRec P = load_pose(g.pose_x, row);
        const uint8_t m = g.cmask[row];
        double d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const bool c = (i < 3) ? (m & 1) : (m & 2);
          d[i] = c ? 0.0 : -g.scale[6 * (size_t)row + i] * x[i];
          g.delta[6 * (size_t)row + i] = d[i];
        }
        V3 pc = P.p;
        Q4 qc = P.q;
        if (!(m & 1)) pc = V3{P.p.x + d[0], P.p.y + d[1], P.p.z + d[2]};
        if (!(m & 2)) qc = quat_plus(P.q, V3{d[3], d[4], d[5]});
        double2* o = reinterpret_cast<double2*>(g.pose_c + (size_t)POSE_STRIDE * row);
        o[0] = double2{pc.x, pc.y};
        o[1] = double2{pc.z, qc.x};
        o[2] = double2{qc.y, qc.z};
        o[3] = double2{qc.w, 0.0};
      }
      if (cg_step && side == SIDE_DIAG) {
        double2* pn = reinterpret_cast<double2*>(p_new + 6 * (size_t)col);
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[k] = double2{x[2 * k], x[2 * k + 1]};
        if (cb == s_begin) {
          const int rl = row - r0;
#pragma unroll
          for (int k = 0; k < 6; ++k) lds_p[6 * rl + k] = x[k];
        }
      }
      if (PACKED) {
        double el[28];
#pragma unroll
        for (int k = 0; k < BLK_PAIRS_PACKED; ++k) { el[2 * k] = blk[k].x; el[2 * k + 1] = blk[k].y; }
        const bool is_end = side == SIDE_END, is_diag = side == SIDE_DIAG;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double a3 = is_end ? el[18 + 3 * i] : is_diag ? el[18 + i] : 0.0;
          const double a4 = is_end ? el[18 + 3 * i + 1] : is_diag ? el[21 + i] : 0.0;
          const double a5 = is_end ? el[18 + 3 * i + 2] : is_diag ? el[24 + i] : 0.0;
          y[i] = el[3 * i] * x[0] + el[3 * i + 1] * x[1] + el[3 * i + 2] * x[2] + a3 * x[3] + a4 * x[4] + a5 * x[5];
          const double b0 = is_end ? 0.0 : el[18 + 3 * i], b1 = is_end ? 0.0 : el[18 + 3 * i + 1], b2 = is_end ? 0.0 : el[18 + 3 * i + 2];
          y[3 + i] = b0 * x[0] + b1 * x[1] + b2 * x[2] + el[9 + 3 * i] * x[3] + el[9 + 3 * i + 1] * x[4] + el[9 + 3 * i + 2] * x[5];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i)
          y[i] = blk[3 * i].x * x[0] + blk[3 * i].y * x[1] + blk[3 * i + 1].x * x[2] + blk[3 * i + 1].y * x[3] +
                 blk[3 * i + 2].x * x[4] + blk[3 * i + 2].y * x[5];
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
    __syncthreads();
    for (int idx = tid; idx < nrows * 6; idx += B) {
      const int rl = idx / 6, k = idx - rl * 6;
      const int rw = r0 + rl;
      const int rb = (idx == tid) ? seg_rb : g.row_slot_begin[rw];
      const int rc = (idx == tid) ? seg_cnt : g.row_slot_cnt[rw];
      const int sb = max(rb, cb) - cb, sE = min(rb + rc, cb + B) - cb;
      double s0 = 0.0, s1 = 0.0;
      int j = sb;
      for (; j + 1 < sE; j += 2) { s0 += lds[j * SPMV_LDS_STRIDE + k]; s1 += lds[(j + 1) * SPMV_LDS_STRIDE + k]; }
      if (j < sE) s0 += lds[j * SPMV_LDS_STRIDE + k];
      const double sm = s0 + s1;
      if (single) {
        g.cg_q[(size_t)rw * 6 + k] = sm;
        if (cg_step) pq[0] += sm * lds_p[idx];
      } else {
        acc += sm;
      }
    }
    __syncthreads();
  }
  if (!single && tid < 6) {
    g.cg_q[(size_t)r0 * 6 + tid] = acc;
    if (cg_step) pq[0] += acc * lds_p[tid];
  }
  if (cg_step) {
    block_sum<1>(pq, scratch);
    if (tid == 0) {
      g.cg_q[(size_t)g.rows_per * 6 + wg] = pq[0];   // p'q partial rides behind q (one rank: rank 0's segment)
      if (wg == 0) {
        g.cg->cnt_a = it; g.cg->beta = beta; g.cg->rho = rho_pub;
        g.cg->rho_hist[odd ? 1 : 0] = rho_pub;
        g.cg->q_hist[odd ? 1 : 0] = q_pub;
        g.cg->op_v = (period > 0 && it % period == 0) ? UNI_V_UPDATE_X : UNI_V_UPDATE;   // residual refresh: x first, then A x, then r = b - A x
      }
    }
  } else if (wg == 0 && tid == 0) {
    g.cg->op_v = tail ? UNI_V_STEP_TAIL : UNI_V_UPDATE_R;
  }
}

// CL: poses per Jacobi block of the preconditioner (1: k_damping's 6x6 inverses)
template <int CL, int INFO>
__global__ __launch_bounds__(UNI_V_BLOCK) void k_uni_v(DeviceGraph g, CgParams prm, double min_diag, double max_diag) {
  constexpr int DIM = 6 * CL;
  __shared__ double rl[VEC_BLOCK];
  __shared__ double scratch[4 * (UNI_V_BLOCK / 64)];
  __shared__ int is_last;
  const int tid = threadIdx.x, bid = blockIdx.x;
  const int m = 6 * g.N;
  // ---- requested before the state is known: the first row chunk of a CG vector update ----
  int idx = bid * VEC_BLOCK + tid;
  bool live = tid < VEC_BLOCK && bid < g.n_vec_wg && idx < m;
  double x0 = 0, p0v = 0, p1v = 0, r0 = 0, q0 = 0, b0 = 0;
  double2 mi[DIM / 2];
#pragma unroll
  for (int k = 0; k < DIM / 2; ++k) mi[k] = double2{0, 0};
  if (live) {
    x0 = g.cg_x[idx]; r0 = g.cg_r[idx]; q0 = g.cg_q[idx]; b0 = g.cg_b[idx];
    p0v = g.cg_p0[idx]; p1v = g.cg_p1[idx];
    // (four-pose clusters: 24 doubles per row requested here do not stay in registers across the operation switch — the compiler
    // parked each of them in scratch behind a wait for its load, 8 us at the top of every launch; they are requested in the update)
    if (CL != 4) {
      const double2* Mi = reinterpret_cast<const double2*>(g.Minv + (size_t)idx * DIM);
#pragma unroll
      for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
    }
  }
  const int op = g.cg->op_v;
  const int it = g.cg->cnt_a;
  const double rho = g.cg->rho;
  double sums[1] = {0};
  if (tid < VEC_BLOCK && bid < g.n_vec_wg) {
    const double* pqp = g.cg_q + (size_t)g.rows_per * 6;
    for (int i = tid; i < g.pq_cap; i += VEC_BLOCK) sums[0] += pqp[i];
  }
  if (bid == 0 && tid == 0) {    // the host keeps a bounded number of launches enqueued ahead of this counter
    const int n = g.cg->slots + 1;
    g.cg->slots = n;
    __hip_atomic_store(&g.scal->slots_done, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (op <= UNI_NOP) return;

  if (op == UNI_V_UPDATE || op == UNI_V_UPDATE_X || op == UNI_V_UPDATE_R) {
    // ---- k_pcg_update, mode 0 / 1 (x only) / 2 (r = b - A x) ----
    if (tid >= VEC_BLOCK || bid >= g.n_vec_wg) return;
    const int mode = op == UNI_V_UPDATE ? 0 : op == UNI_V_UPDATE_X ? 1 : 2;
    const int odd = it & 1;
    double pa = odd ? p1v : p0v;
    const double* p = odd ? g.cg_p1 : g.cg_p0;
    double alpha = 0.0;
    if (mode != 2) {
      block_sum_w<1>(sums, scratch, VEC_BLOCK / 64);
      const double pq = sums[0];
      if (!(pq > 0.0) || !isfinite(pq)) {
        // "Matrix is indefinite, no more progress can be made": keep x of the previous iteration; the next slot launch finds the CG stopped
        if (bid == 0 && tid == 0) { g.cg->iters = it - 1; g.cg->status = 1; g.cg->done = 1; g.cg->op_s = UNI_S_CG; }
        return;
      }
      alpha = rho / pq;
    }
    double acc[3] = {0.0, 0.0, 0.0};
    for (int base = bid * VEC_BLOCK; base < m; base += g.n_vec_wg * VEC_BLOCK) {
      if (base != bid * VEC_BLOCK) {
        idx = base + tid;
        live = idx < m;
        if (live) {
          x0 = g.cg_x[idx]; r0 = g.cg_r[idx]; q0 = g.cg_q[idx]; b0 = g.cg_b[idx];
          pa = p[idx];
          const double2* Mi = reinterpret_cast<const double2*>(g.Minv + (size_t)idx * DIM);
#pragma unroll
          for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
        }
      } else if (CL == 4 && live) {
        const double2* Mi = reinterpret_cast<const double2*>(g.Minv + (size_t)idx * DIM);
#pragma unroll
        for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
      }
      double x = 0.0, r = 0.0;
      if (live) {
        if (mode == 2) {
          x = x0;
          r = b0 - q0;
          g.cg_r[idx] = r;
        } else {
          x = x0 + alpha * pa;
          g.cg_x[idx] = x;
          if (mode == 0) { r = r0 - alpha * q0; g.cg_r[idx] = r; }
        }
      }
      if (mode == 1) continue;
      rl[tid] = r;
      __syncthreads();
      if (live) {
        const double* rv = rl + DIM * (tid / DIM);
        double z = 0.0;
#pragma unroll
        for (int k = 0; k < DIM / 2; ++k) z += mi[k].x * rv[2 * k] + mi[k].y * rv[2 * k + 1];
        g.cg_z[idx] = z;
        acc[0] += r * z;
        acc[1] += x * (b0 + r);
        acc[2] += r * r;
      }
      __syncthreads();
    }
    if (mode == 1) {
      if (bid == 0 && tid == 0) g.cg->op_s = UNI_S_REFRESH;
      return;
    }
    block_sum_w<3>(acc, scratch, VEC_BLOCK / 64);
    if (tid == 0) {
      const size_t o = (size_t)(odd ? g.n_part : 0) + bid;
      g.part_rz[o] = acc[0];
      g.part_q[o] = acc[1];
      g.part_rr[o] = acc[2];
      if (bid == 0) { g.cg->cnt_b = it; g.cg->op_s = UNI_S_CG; }
    }
    return;
  }

  if (op == UNI_V_STEP_TAIL) {
    const int nblocks = g.n_edge_wg + g.n_pose_wg;
    if (tid >= EDGE_BLOCK || bid >= nblocks) return;
    step_tail_body<INFO>(g, 4, bid, nblocks, scratch, &is_last);   // gate 4: the decision also sets the stream's next operations
    return;
  }

  // ---- UNI_V_HEAD: accept-finish of the step just accepted, then damping / preconditioner and the CG start of the next pass ----
  LmDev& D = *g.lm;
  const int accepted = D.accepted, pause = D.pause;
  double gm = 0.0;
  if (accepted && tid < POSE_BLOCK && bid < g.n_pose_wg) {
    const int v = bid * POSE_BLOCK + tid;
    if (v < g.N) {
      const PoseRec P = load_pose(g.pose_c, v);
      const double2* src = reinterpret_cast<const double2*>(g.pose_c + (size_t)POSE_STRIDE * v);
      double2* dst = reinterpret_cast<double2*>(g.pose_x + (size_t)POSE_STRIDE * v);
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      const uint8_t cm = g.cmask[v];
      const double* gr = g.grad + 6 * (size_t)v;
      if (!(cm & 1)) gm = fmax(gm, fmax(fabs(gr[0]), fmax(fabs(gr[1]), fabs(gr[2]))));
      if (!(cm & 2)) {
        const Q4 q = quat_plus(P.q, V3{-gr[3], -gr[4], -gr[5]});
        gm = fmax(gm, fmax(fmax(fabs(P.q.x - q.x), fabs(P.q.y - q.y)), fmax(fabs(P.q.z - q.z), fabs(P.q.w - q.w))));
      }
    }
  }
  gm = wave_max(gm);
  if ((tid & 63) == 0) scratch[tid >> 6] = gm;
  __syncthreads();
  if (tid == 0 && accepted && bid < g.n_pose_wg) {
    double t = 0.0;
    for (int w = 0; w < POSE_BLOCK / 64; ++w) t = fmax(t, scratch[w]);
    g.part_misc[4 * (size_t)g.n_part + bid] = t;
  }
  __syncthreads();
  if (!pause && bid < g.n_vec_wg) {
    const double radius = D.core.radius;
    const int mode = D.core.reuse_diagonal ? 1 : 0;
    double acc[2] = {0.0, 0.0};
    for (int chunk = bid; chunk * (VEC_BLOCK / 6) < g.N; chunk += g.n_vec_wg) {
      const int pose0 = chunk * (VEC_BLOCK / 6);     // 64 poses = VEC_BLOCK rows per chunk
      // LM damping and the Jacobi blocks of these poses (k_damping / k_cluster_precond)
      if (CL == 1) {
        if (tid < 64 && pose0 + tid < g.N) damping_pose(g, pose0 + tid, radius, min_diag, max_diag, mode);
      } else {
        constexpr int CPW = 64 / DIM;
        const int wave = tid >> 6, lane = tid & 63;
        const int n_cl = (g.N + CL - 1) / CL;
        for (int cw = wave; cw * CPW < 64 / CL; cw += UNI_V_BLOCK / 64)
          cluster_precond_wave<CL>(g, radius, min_diag, max_diag, mode, pose0 / CL + cw * CPW, min(n_cl, pose0 / CL + 64 / CL), lane);
      }
      __syncthreads();      // the inverses were written by other waves of this work-group
      // CG start (k_pcg_init)
      const int ridx = chunk * VEC_BLOCK + tid;
      const bool rlive = tid < VEC_BLOCK && ridx < m;
      double b = 0.0;
      if (rlive) {
        b = g.scale[ridx] * g.grad[ridx];
        g.cg_b[ridx] = b;
        g.cg_x[ridx] = 0.0;
        g.cg_r[ridx] = b;
        g.cg_p0[ridx] = 0.0;
        g.cg_p1[ridx] = 0.0;
      }
      if (tid < VEC_BLOCK) rl[tid] = b;
      __syncthreads();
      if (rlive) {
        const double* Mi = g.Minv + (size_t)ridx * DIM;
        const double* rv = rl + DIM * (tid / DIM);
        double z = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) z += Mi[k] * rv[k];
        g.cg_z[ridx] = z;
        acc[0] += b * z;
        acc[1] += b * b;
      }
      __syncthreads();
    }
    // (waves 6 and 7 carry zeros: the sums of the six row waves are what k_pcg_init's work-group forms)
    block_sum<2>(acc, scratch);
    if (tid == 0) {
      g.part_rz[bid] = acc[0];
      g.part_rz[g.n_part + bid] = 0.0;
      g.part_bb[bid] = acc[1];
      g.part_rr[bid] = acc[1];
      g.part_rr[g.n_part + bid] = acc[1];
      g.part_q[bid] = 0.0;
      g.part_q[g.n_part + bid] = 0.0;
      if (bid == 0) { g.cg->done = 0; g.cg->iters = 0; g.cg->status = 0; g.cg->cnt_a = 0; g.cg->cnt_b = 0; }
    }
  }
  // the last work-group to finish: gradient norm of the accepted point, the opening tests of the next pass, the next operations
  if (tid == 0) {
    __threadfence();
    is_last = (atomicAdd(&g.flags[3], 1) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  double mm = 0.0;
  if (accepted)
    for (int i = tid; i < g.n_pose_wg; i += UNI_V_BLOCK)
      mm = fmax(mm, __hip_atomic_load(&g.part_misc[4 * (size_t)g.n_part + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  mm = wave_max(mm);
  if ((tid & 63) == 0) scratch[tid >> 6] = mm;
  __syncthreads();
  if (tid == 0) {
    g.flags[3] = 0;
    const long long now = (long long)__builtin_amdgcn_s_memrealtime();
    if (accepted) {
      double t = 0.0;
      for (int w = 0; w < UNI_V_BLOCK / 64; ++w) t = fmax(t, scratch[w]);
      D.core.gmax = t;
      g.scal->ring[D.core.iteration % LM_RING].gradient_max_norm = t;
      g.scal->gradient_max = t;
      D.accepted = 0;
      lm_pre_step_checks(D, true);
      D.ticks_jacobian += now - D.t_mark;    // (the damping / CG start share of this launch is booked here too: one clock per launch)
      D.t_mark = now;
    }
    if (D.halt) { g.cg->op_s = UNI_EXIT; g.cg->op_v = UNI_EXIT; }
    else if (pause) { D.halt = LM_HALT_BUDGET; g.cg->op_s = UNI_EXIT; g.cg->op_v = UNI_EXIT; }
    else { g.cg->op_s = UNI_S_CG; g.cg->op_v = UNI_NOP; }
    lm_mirror(g);
  }
}

// ------------------------------------------------------------------------------------------------
// Several ranks, truncated CG: owner-only pipelined preconditioned CG (pgo_kernels.h DeviceGraph::pipe_buf; Ghysels & Vanroose
// 2014).  Standard CG has two global reductions per iteration (p'q before alpha, r'z before beta); with ONE collective per
// iteration that forces every rank to update all 6 N rows (k_pcg_update).  The pipelined recurrences
//     u = M^-1 r, w = A u, m = M^-1 w;   gamma = (r,u), delta = (w,u);   beta = gamma / gamma_prev, alpha = gamma / (delta - beta gamma / alpha_prev)
//     n = A m;   z = n + beta z, qq = m + beta qq, s = w + beta s, p = u + beta p;   x += alpha p, r -= alpha s, u -= alpha qq, w -= alpha z
// have one: a rank multiplies its rows (gathering m of every row from the exchange buffer), updates the eight vectors of ITS rows,
// applies its own Jacobi blocks (m = M^-1 w: no exchange of the inverses either) and leaves its m segment and its three sums —
// (r,u), (w,u), x'(b + r) over its rows — in the buffer the all-gather then completes.  Same iterates as Ceres'
// ConjugateGradientsSolver in exact arithmetic, same stop rules on the same quantities; no periodic residual refresh (four
// products): truncated at eta = 0.1 the recurrences do not run long enough to drift, and a request that runs the CG to 1e-13
// (exact request on several ranks) keeps the standard form.
// Launch `seq` reads pipe_buf[seq & 1] and CgState::pipe[seq & 1] and writes the other ones: seq 0 multiplies u0 (w0 = A u0),
// seq i + 1 is iteration i.  mode 1 / 2: only the stop test the launch `seq` would apply, the CG state for the host, and (1) the
// hand-over.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t pipe_index(const DeviceGraph& g, int row, int k) {
  const int rk = row / g.rows_per;
  return (size_t)rk * g.pipe_seg + (size_t)(row - rk * g.rows_per) * 6 + k;
}

// b = S g of every row (the step tail's model change reads it everywhere); r0 = b, x0 = 0, u0 = M^-1 r0 of the owned rows.
template <int CL>
__global__ __launch_bounds__(VEC_BLOCK) void k_pipe_init(DeviceGraph g) {
  constexpr int DIM = 6 * CL;
  __shared__ double rl[VEC_BLOCK];
  const int tid = threadIdx.x;
  for (int base = blockIdx.x * VEC_BLOCK; base < 6 * g.N; base += gridDim.x * VEC_BLOCK) {   // (a chunk = 64 poses: whole Jacobi blocks, one owner)
    const int idx = base + tid;
    const bool live = idx < 6 * g.N;
    const int row = idx / 6;
    const bool own = live && row >= g.row_lo && row < g.row_hi;
    double b = 0.0;
    if (live) {
      b = g.scale[idx] * g.grad[idx];
      g.cg_b[idx] = b;
      g.cg_x[idx] = 0.0;
      if (own) g.cg_r[idx] = b;
    }
    rl[tid] = b;
    __syncthreads();
    if (own) {
      const double* Mi = g.Minv + (size_t)idx * DIM;
      const double* rv = rl + DIM * (tid / DIM);
      double u = 0.0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) u += Mi[k] * rv[k];
      g.cg_u[idx] = u;
      const size_t pi = pipe_index(g, row, idx - 6 * row);
      if (g.peer_tab) { for (int rk = 0; rk < g.world; ++rk) static_cast<double*>(g.peer_tab[3 * rk])[pi] = u; }
      else g.pipe_buf[0][pi] = u;
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) {
    g.cg->done = 0; g.cg->iters = 0; g.cg->status = 0; g.cg->cnt_a = 0; g.cg->cnt_b = 0;
    CgState::Pipe n;
    n.cnt = 0; n.pad = 0; n.gamma_prev = 0.0; n.alpha_prev = 0.0; n.q_prev = 0.0;
    g.cg->pipe[0] = n;
    g.cg->pipe[1] = n;
  }
}

template <bool PACKED, int CL>
__global__ __launch_bounds__(256) void k_pipe_cg(DeviceGraph g, CgParams prm, int seq, int mode) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  constexpr int DIM = 6 * CL;
  extern __shared__ double lds[];  // SPMV_LDS_STRIDE * block (slot results) + 6 * block + 6 (w of the owned rows)
  __shared__ double scratch[32];
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  double* lds_w = lds + (size_t)SPMV_LDS_STRIDE * B;
  const int rs = seq & 1, ws = rs ^ 1;
  const double* rd = g.pipe_buf[rs];
  double* wr = g.pipe_buf[ws];
  // ---- independent loads: the slot's words and block, the row bookkeeping, the state, every rank's partial sums ----
  const int s_begin = g.wg_slot_begin[wg];
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  const int t = s_begin + tid;
  const int col = g.slot_col[t];
  double2 blk[NPAIR];
  if (mode == 0) {
    const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) blk[k] = bp[(size_t)k * 64];
  }
  const uint8_t side = g.slot_side[t];
  const int nown = nrows * 6;
  int seg_rb = 0, seg_cnt = 0;
  if (tid < nown) { seg_rb = g.row_slot_begin[r0 + tid / 6]; seg_cnt = g.row_slot_cnt[r0 + tid / 6]; }
  const int done0 = g.cg->done;
  const CgState::Pipe st = g.cg->pipe[rs];
  double f_gamma = 0, f_delta = 0, f_q = 0;     // every rank's three sums, added in rank order by every lane alike: same bits everywhere
  for (int rk = 0; rk < g.world; ++rk) {
    const double* pp = g.bx[0] ? g.bx[rs] + (size_t)(rk + 1) * g.bx_cseg - 4 : rd + (size_t)rk * g.pipe_seg + (size_t)g.rows_per * 6;
    f_gamma += pp[0]; f_delta += pp[1]; f_q += pp[2];
  }
  const int moff = g.bx[0] ? g.bx_slot_off[t] : 0;        // (boundary exchange: where this slot's column lives)
  // the first pass of the owned rows' operands, requested with the blocks
  const bool own0 = mode == 0 && tid < nown;
  const size_t gi = 6 * (size_t)r0 + tid;
  double pr = 0, pu = 0, pw = 0, pz = 0, pq = 0, ps = 0, ppv = 0, px = 0, pb = 0, pm = 0;
  double2 mi[DIM / 2];
#pragma unroll
  for (int k = 0; k < DIM / 2; ++k) mi[k] = double2{0, 0};
  if (own0) {
    pr = g.cg_r[gi]; pu = g.cg_u[gi];
    if (seq != 0) {
      pw = g.cg_w[gi]; pz = g.cg_z[gi]; pq = g.cg_qq[gi]; ps = g.cg_s[gi]; ppv = g.cg_p0[gi]; px = g.cg_x[gi]; pb = g.cg_b[gi];
      pm = rd[pipe_index(g, r0 + tid / 6, tid % 6)];
    }
    const double2* Mi = reinterpret_cast<const double2*>(g.Minv + gi * DIM);
#pragma unroll
    for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
  }
  const bool w0 = seq == 0;
  int stop = 0, status = 0;
  double alpha = 0.0, beta = 0.0, gamma = 0.0, Q1 = 0.0;
  const int cnt = st.cnt;
  if (!done0 && !w0) {
    gamma = f_gamma;
    const double delta = f_delta;
    Q1 = -f_q;
    if (cnt > 0) {
      const double zeta = cnt * (Q1 - st.q_prev) / Q1;
      if (zeta < prm.q_tolerance && cnt >= prm.min_iterations) stop = 1;
      if (cnt >= prm.max_iterations) stop = 1;
    }
    if (!stop && (gamma == 0.0 || !isfinite(gamma))) { stop = 1; status = (gamma == 0.0) ? 0 : 2; }
    if (!stop && cnt > 0) {
      beta = gamma / st.gamma_prev;
      if (beta == 0.0 || !isfinite(beta)) { stop = 1; status = 2; }
    }
    if (!stop) {
      const double den = cnt > 0 ? delta - beta * gamma / st.alpha_prev : delta;
      if (!(den > 0.0) || !isfinite(den)) { stop = 1; status = 1; }     // "matrix is indefinite": x of the previous iteration stands
      else alpha = gamma / den;
    }
  }
  if (mode != 0) {      // batch end: the state for the host
    if (wg == 0 && tid == 0) {
      int done = done0, iters = done0 ? g.cg->iters : cnt, stat = done0 ? g.cg->status : status;
      if (!done0 && stop) { g.cg->iters = cnt; g.cg->status = status; __threadfence(); g.cg->done = 1; done = 1; }
      g.scal->cg_iterations = iters;
      g.scal->cg_status = done ? stat : -1;
      g.scal->cg_residual_sq = 0.0;
      if (mode == 1) publish_sequence(g);
    }
    return;
  }
  if (done0) return;
  if (stop) {
    if (wg == 0 && tid == 0) { g.cg->iters = cnt; g.cg->status = status; __threadfence(); g.cg->done = 1; }
    return;
  }
  if (wg == 0 && tid == 0) {
    CgState::Pipe n;
    n.pad = 0;
    if (w0) { n.cnt = 0; n.gamma_prev = 0.0; n.alpha_prev = 0.0; n.q_prev = 0.0; }
    else { n.cnt = cnt + 1; n.gamma_prev = gamma; n.alpha_prev = alpha; n.q_prev = Q1; }
    g.cg->pipe[ws] = n;
  }
  // ---- n = A m over this work-group's slots ----
  double y[6] = {0, 0, 0, 0, 0, 0};
  if (col >= 0) {
    const double2* ms = reinterpret_cast<const double2*>(!g.bx[0] ? rd + pipe_index(g, col, 0) : moff >= 0 ? rd + moff : g.bx[rs] + (-1 - moff));
    const double2 g0 = ms[0], g1 = ms[1], g2 = ms[2];
    const double x[6] = {g0.x, g0.y, g1.x, g1.y, g2.x, g2.y};
    if (PACKED) {
      double el[28];
#pragma unroll
      for (int k = 0; k < BLK_PAIRS_PACKED; ++k) { el[2 * k] = blk[k].x; el[2 * k + 1] = blk[k].y; }
      const bool is_end = side == SIDE_END, is_diag = side == SIDE_DIAG;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double a3 = is_end ? el[18 + 3 * i] : is_diag ? el[18 + i] : 0.0;
        const double a4 = is_end ? el[18 + 3 * i + 1] : is_diag ? el[21 + i] : 0.0;
        const double a5 = is_end ? el[18 + 3 * i + 2] : is_diag ? el[24 + i] : 0.0;
        y[i] = el[3 * i] * x[0] + el[3 * i + 1] * x[1] + el[3 * i + 2] * x[2] + a3 * x[3] + a4 * x[4] + a5 * x[5];
        const double b0 = is_end ? 0.0 : el[18 + 3 * i], b1 = is_end ? 0.0 : el[18 + 3 * i + 1], b2 = is_end ? 0.0 : el[18 + 3 * i + 2];
        y[3 + i] = b0 * x[0] + b1 * x[1] + b2 * x[2] + el[9 + 3 * i] * x[3] + el[9 + 3 * i + 1] * x[4] + el[9 + 3 * i + 2] * x[5];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        y[i] = blk[3 * i].x * x[0] + blk[3 * i].y * x[1] + blk[3 * i + 1].x * x[2] + blk[3 * i + 1].y * x[3] +
               blk[3 * i + 2].x * x[4] + blk[3 * i + 2].y * x[5];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
  __syncthreads();
  // ---- the owned rows: lane idx owns component idx % 6 of row r0 + idx / 6 ----
  double acc[3] = {0.0, 0.0, 0.0};
  for (int idx = tid; idx < nown; idx += B) {
    const size_t gj = 6 * (size_t)r0 + idx;
    if (idx != tid) { seg_rb = g.row_slot_begin[r0 + idx / 6]; seg_cnt = g.row_slot_cnt[r0 + idx / 6]; }
    const int k = idx % 6;
    const int sb = seg_rb - s_begin, sE = sb + seg_cnt;
    double s0 = 0.0, s1 = 0.0;
    int j = sb;
    for (; j + 1 < sE; j += 2) { s0 += lds[j * SPMV_LDS_STRIDE + k]; s1 += lds[(j + 1) * SPMV_LDS_STRIDE + k]; }
    if (j < sE) s0 += lds[j * SPMV_LDS_STRIDE + k];
    const double sm = s0 + s1;
    if (idx != tid) {
      pr = g.cg_r[gj]; pu = g.cg_u[gj];
      if (!w0) {
        pw = g.cg_w[gj]; pz = g.cg_z[gj]; pq = g.cg_qq[gj]; ps = g.cg_s[gj]; ppv = g.cg_p0[gj]; px = g.cg_x[gj]; pb = g.cg_b[gj];
        pm = rd[pipe_index(g, r0 + idx / 6, k)];
      }
    }
    double vr = pr, un = pu, wn;
    if (w0) {
      wn = sm;
      g.cg_w[gj] = wn;
      g.cg_z[gj] = 0.0; g.cg_qq[gj] = 0.0; g.cg_s[gj] = 0.0; g.cg_p0[gj] = 0.0;
    } else {
      const double vw = pw, vm = pm;
      const double zn = sm + beta * pz, qn = vm + beta * pq, sn = vw + beta * ps, pn = un + beta * ppv;
      const double xn = px + alpha * pn, rn = vr - alpha * sn;
      un = un - alpha * qn;
      wn = vw - alpha * zn;
      g.cg_z[gj] = zn; g.cg_qq[gj] = qn; g.cg_s[gj] = sn; g.cg_p0[gj] = pn;
      g.cg_x[gj] = xn; g.cg_r[gj] = rn; g.cg_u[gj] = un; g.cg_w[gj] = wn;
      acc[2] += xn * (pb + rn);
      vr = rn;
    }
    acc[0] += vr * un;
    acc[1] += wn * un;
    lds_w[idx] = wn;
  }
  if (tid < 6) lds_w[nown + tid] = 0.0;       // the missing half of a last odd pair
  __syncthreads();
  for (int idx = tid; idx < nown; idx += B) {
    const size_t gj = 6 * (size_t)r0 + idx;
    if (idx != tid) {
      const double2* Mi = reinterpret_cast<const double2*>(g.Minv + gj * DIM);
#pragma unroll
      for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
    }
    const double* wv = lds_w + DIM * (idx / DIM);
    double mn = 0.0;
#pragma unroll
    for (int k = 0; k < DIM / 2; ++k) mn += mi[k].x * wv[2 * k] + mi[k].y * wv[2 * k + 1];
    const size_t pi = pipe_index(g, r0 + idx / 6, idx % 6);
    if (g.peer_tab) { for (int rk = 0; rk < g.world; ++rk) static_cast<double*>(g.peer_tab[3 * rk + ws])[pi] = mn; }
    else wr[pi] = mn;
  }
  // this rank's three sums: per work-group partials here, folded by k_pipe_fold (one work-group, behind this launch) — the exchange
  // carries three numbers per rank.  (A ticket for the last work-group to fold them cost more than the launch: > 1000
  // work-groups per rank bump one counter at 100 k poses / 8 ranks.)
  block_sum_w<3>(acc, scratch, B / 64);
  if (tid == 0) { g.part_rz[wg] = acc[0]; g.part_q[wg] = acc[1]; g.part_rr[wg] = acc[2]; }
}
// device-initiated exchange: "this rank's launch gseq has stored everything" to every rank, then wait for everybody's (one lane)
__device__ __forceinline__ void peer_signal_and_wait(const DeviceGraph& g, unsigned long long gseq) {
  __threadfence_system();
  for (int rk = 0; rk < g.world; ++rk)
    __hip_atomic_store(static_cast<unsigned long long*>(g.peer_tab[3 * rk + 2]) + g.rank, gseq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // (a peer that never signals — e.g. virtual ranks whose streams share one hardware queue, so that this spinning launch blocks the
  // very kernel it waits for — must not hang the device: after 2 s of the 100 MHz clock the CG is stopped with the "broke down"
  // status, which the LM loop reports as a linear-solver failure)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int rk = 0; rk < g.world; ++rk)
    while (__hip_atomic_load(g.peer_flags + rk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < gseq) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { g.cg->status = 2; g.cg->done = 1; return; }   // (status / done report it; flags[3] is the ticket counter of the step tail, not a status word)
    }
}
__global__ __launch_bounds__(256) void k_pipe_fold(DeviceGraph g, int seq, unsigned long long gseq) {
  __shared__ double scratch[16];
  if (g.cg->done) return;       // (the same decision on every rank: nobody waits for this launch's number)
  double t3[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < g.n_wg; i += 256) { t3[0] += g.part_rz[i]; t3[1] += g.part_q[i]; t3[2] += g.part_rr[i]; }
  block_sum<3>(t3, scratch);
  if (threadIdx.x == 0) {
    const size_t off = (size_t)g.rank * g.pipe_seg + (size_t)g.rows_per * 6;
    if (g.peer_tab) {
      for (int rk = 0; rk < g.world; ++rk) {
        double* pp = static_cast<double*>(g.peer_tab[3 * rk + ((seq & 1) ^ 1)]) + off;
        pp[0] = t3[0]; pp[1] = t3[1]; pp[2] = t3[2];
      }
      peer_signal_and_wait(g, gseq);
    } else {
      double* pp = g.pipe_buf[(seq & 1) ^ 1] + off;
      pp[0] = t3[0]; pp[1] = t3[1]; pp[2] = t3[2];
    }
  }
}
__global__ void k_peer_signal(DeviceGraph g, unsigned long long gseq) { peer_signal_and_wait(g, gseq); }
// boundary exchange: work-group 0 (fold_seq >= 0) does k_pipe_fold's job — the producing launch's partial triples -> this rank's three sums,
// here at the end of its segment of bx — the others copy the rank's boundary rows out of the full-layout buffer into the segment
__global__ __launch_bounds__(256) void k_pipe_pack(DeviceGraph g, int buf, int fold_seq, int n_entries) {
  __shared__ double scratch[16];
  if (fold_seq >= 0 && g.cg->done) return;
  double* seg = g.bx[buf] + (size_t)g.rank * g.bx_cseg;
  if (blockIdx.x == 0) {
    if (fold_seq < 0) return;
    double t3[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n_entries; i += 256) { t3[0] += g.part_rz[i]; t3[1] += g.part_q[i]; t3[2] += g.part_rr[i]; }
    block_sum<3>(t3, scratch);
    if (threadIdx.x == 0) { double* pp = seg + g.bx_cseg - 4; pp[0] = t3[0]; pp[1] = t3[1]; pp[2] = t3[2]; }
    return;
  }
  const int i = ((int)blockIdx.x - 1) * 256 + (int)threadIdx.x;
  if (i >= 3 * g.bx_nb) return;
  const int e = i / 3, k = i - 3 * e;
  const double2* src = reinterpret_cast<const double2*>(g.pipe_buf[buf] + pipe_index(g, g.bx_brow[e], 0));
  reinterpret_cast<double2*>(seg + 6 * (size_t)e)[k] = src[k];
}

// Several ranks, owner-only CG: of the other ranks' diagonal blocks only the six diagonal entries are needed anywhere (column
// scaling, LM damping, the model change of the step tail), so those travel — 6 doubles per pose instead of 36.  phase 0: the
// owned rows' diagonals into the exchange buffer (laid out like cg_x); phase 1: the other rows' diagonals out of it.
__global__ void k_hdiag6(DeviceGraph g, double* buf, int phase) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 6 * g.N) return;
  const int v = idx / 6, i = idx - 6 * v;
  const bool own = v >= g.row_lo && v < g.row_hi;
  if (phase == 0) { if (own) buf[idx] = g.Hdiag[36 * (size_t)v + 7 * i]; }
  else if (!own) {      // (the rest of the row is cleared: an earlier solve in the standard form left the owner's entries of that time)
    const double d = buf[idx];
#pragma unroll
    for (int j = 0; j < 6; ++j) g.Hdiag[36 * (size_t)v + 6 * i + j] = j == i ? d : 0.0;
  }
}

// (Re)opens the universal stream for `decisions` more LM iterations; behind LM_HALT_BUDGET the head launch that paused it is due again.
// next_launch: index of the first launch behind this kernel (fused form: its parity names the state slot that launch reads)
__global__ void k_lm_budget(DeviceGraph g, int decisions, int next_launch) {
  LmDev& D = *g.lm;
  D.t_mark = (long long)__builtin_amdgcn_s_memrealtime();
  D.decision_limit = decisions < 0 ? 0x7fffffff : D.lm_done + decisions;
  D.pause = 0;
  if (D.halt == LM_HALT_BUDGET || D.halt == LM_RUN) {
    D.halt = LM_RUN;
    g.cg->op_s = UNI_NOP;
    g.cg->op_v = UNI_V_HEAD;
    CgState::Fused n{};
    n.op = F_HEAD;
    g.cg->f[next_launch & 1] = n;
    g.scal->halt = 0;
  }
}
__global__ void k_lm_publish(DeviceGraph g) { publish_sequence(g); }

// ------------------------------------------------------------------------------------------------
// Batched solve of independent graphs (BatchPlan): the per-component pieces of the LM iteration.
// ------------------------------------------------------------------------------------------------
// LevenbergMarquardtStrategy's diagonal with the radius of the pose's own component (k_damping mode 0 does the same with
// one radius; the quotient is formed by the same operation, so a component gets the bits it gets when solved alone).
__global__ void k_batch_d2(DeviceGraph g, BatchPlan b, double min_diag, double max_diag) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 6 * g.N) return;
  const int v = idx / 6, i = idx - 6 * v;
  const double radius = b.radius[b.pose_comp[v]];
  const double dc = fmin(fmax(g.Hdiag[36 * (size_t)v + 7 * i], min_diag), max_diag);
  g.diag_clamped[idx] = dc;
  g.d2[idx] = dc / radius;
}

// b.split workgroups per component: candidate cost over its edges, model cost change / step norm / state norm / gradient norm
// over its poses (the arithmetic of k_step_tail and k_gradient_norm, reduced per component) -> partials, folded in a fixed
// order by k_batch_fold into pinned memory.
template <int INFO>
__global__ __launch_bounds__(256) void k_batch_scalars(DeviceGraph g, BatchPlan b) {
  __shared__ double scratch[4 * 4];
  __shared__ double smax[4];
  const int c = blockIdx.x / b.split, part = blockIdx.x - c * b.split, tid = threadIdx.x;
  const int stride = 256 * b.split;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // candidate cost, model change, |step|^2, |x|^2
  double gm = 0.0;
  for (int v = b.pose_begin[c] + part * 256 + tid; v < b.pose_begin[c + 1]; v += stride) {
    const PoseRec P = load_pose(g.pose_x, v), C = load_pose(g.pose_c, v);
    const uint8_t m = g.cmask[v];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const size_t idx = 6 * (size_t)v + i;
      const double x = g.cg_x[idx];
      const double hx = g.cg_q[q_index(g, v, i)] - g.d2[idx] * x;
      const bool cst = (i < 3) ? (m & 1) : (m & 2);
      acc[1] += cst ? 0.0 : (x * g.cg_b[idx] - 0.5 * x * hx);
    }
    const double* gr = g.grad + 6 * (size_t)v;
    if (!(m & 1)) {
      const double dx = P.p.x - C.p.x, dy = P.p.y - C.p.y, dz = P.p.z - C.p.z;
      acc[2] += dx * dx + dy * dy + dz * dz;
      acc[3] += P.p.x * P.p.x + P.p.y * P.p.y + P.p.z * P.p.z;
      gm = fmax(gm, fmax(fabs(gr[0]), fmax(fabs(gr[1]), fabs(gr[2]))));
    }
    if (!(m & 2)) {
      const double dx = P.q.x - C.q.x, dy = P.q.y - C.q.y, dz = P.q.z - C.q.z, dw = P.q.w - C.q.w;
      acc[2] += dx * dx + dy * dy + dz * dz + dw * dw;
      acc[3] += P.q.x * P.q.x + P.q.y * P.q.y + P.q.z * P.q.z + P.q.w * P.q.w;
      const Q4 q = quat_plus(P.q, V3{-gr[3], -gr[4], -gr[5]});
      gm = fmax(gm, fmax(fmax(fabs(P.q.x - q.x), fabs(P.q.y - q.y)), fmax(fabs(P.q.z - q.z), fabs(P.q.w - q.w))));
    }
  }
  for (int e = b.edge_begin[c] + part * 256 + tid; e < b.edge_begin[c + 1]; e += stride) {
    const PoseRec A = load_pose(g.pose_c, g.edge_a[e]), B = load_pose(g.pose_c, g.edge_b[e]);
    const size_t E = (size_t)g.E;
    const V3 mp{g.emeas[e], g.emeas[E + e], g.emeas[2 * E + e]};
    const Q4 mq{g.emeas[3 * E + e], g.emeas[4 * E + e], g.emeas[5 * E + e], g.emeas[6 * E + e]};
    double er[6];
    edge_error(A.p, A.q, B.p, B.q, mp, mq, er);
    double sq;
    if (INFO) {
      const WBlocks W = g.info_mode == 3 ? load_W_diag(g.eW, E, (size_t)e) : load_W(g.eW, E, (size_t)e);   // (diagonal W: 6 of the 21 planes)
      const V3 ep{er[0], er[1], er[2]}, eq{er[3], er[4], er[5]};
      const V3 a1 = mulv(W.pp, ep), a2 = mulv(W.pr, eq), b1 = mulTv(W.pr, ep), b2 = mulv(W.rr, eq);
      sq = dot(ep, V3{a1.x + a2.x, a1.y + a2.y, a1.z + a2.z}) + dot(eq, V3{b1.x + b2.x, b1.y + b2.y, b1.z + b2.z});
    } else {
      sq = er[0] * er[0] + er[1] * er[1] + er[2] * er[2] + er[3] * er[3] + er[4] * er[4] + er[5] * er[5];
    }
    double rho0, rho1;
    loss_eval(g.loss_kind, g.loss_a, sq, &rho0, &rho1);
    acc[0] += 0.5 * rho0;
  }
  block_sum<4>(acc, scratch);
  gm = wave_max(gm);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) smax[wave] = gm;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 4; ++w) t = fmax(t, smax[w]);
    double* o = b.partial + 5 * (size_t)blockIdx.x;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3]; o[4] = t;
  }
}

__global__ void k_batch_fold(BatchPlan b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.n_comp) return;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, t = 0.0;
  for (int j = 0; j < b.split; ++j) {
    const double* q = b.partial + 5 * ((size_t)c * b.split + j);
    s[0] += q[0]; s[1] += q[1]; s[2] += q[2]; s[3] += q[3];
    t = fmax(t, q[4]);
  }
  BatchScalars o;
  o.cand_cost = s[0]; o.model_change = s[1]; o.step_norm_sq = s[2]; o.x_norm_sq = s[3]; o.gradient_max = t;
  o.pad[0] = o.pad[1] = o.pad[2] = 0.0;
  b.out[c] = o;
}

__global__ void k_batch_accept(DeviceGraph g, BatchPlan b) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one lane per 16 bytes of a pose record
  if (idx >= 4 * g.N) return;
  const int v = idx >> 2;
  if (!b.accept[b.pose_comp[v]]) return;
  reinterpret_cast<double2*>(g.pose_x)[idx] = reinterpret_cast<const double2*>(g.pose_c)[idx];
}

__global__ void k_empty(DeviceGraph g) {}
__global__ void k_touch(DeviceGraph g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < 6 * g.N) g.cg_r[idx] = g.cg_z[idx] + 1.0;
}
__global__ void k_copy_delta(DeviceGraph g, const double* step) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < 6 * g.N) g.delta[idx] = step[idx];
}

#include "pgo_uni_fused.h"
#include "pgo_uni_resident.h"

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// Two translation units are built from this text (csrc/Makefile): pgo_kernels.o — everything but the resident stream, compiled without
// MachineLICM — and pgo_res_kernels.o (PGO_TU_RESIDENT) — the resident stream's two kernels and their launcher, with it: k_res_cg spins
// through its CG turns inside one loop, where the hoisted literals are worth 0.17 us per turn (0.1884 -> 0.1865 ms per LM iteration at
// BASELINE configs[1]); everywhere else they only cost registers.
#ifndef PGO_TU_RESIDENT
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

bool lean_bsr_fits(const DeviceGraph& g) {
  // packed slots (no position / rotation coupling in the information) and 32-bit byte offsets: 21 information planes, the blocks and
  // the pose array each below 4 GiB
  return g.info_mode != 1 && g.blk_packed && (long long)g.n_slots < 14000000LL && (long long)g.N < 60000000LL;
}
template <int WAVES>
static void launch_lean_bsr_w(const DeviceGraph& g, hipStream_t s, int gate) {
  const size_t lds = (size_t)LEAN_NV * (g.block / 2) * sizeof(double);
  const dim3 grid(g.n_wg), block(g.block);
  if (g.info_mode == 3) hipLaunchKernelGGL((k_linearize_lean_bsr<3, WAVES>), grid, block, lds, s, g, gate);
  else if (g.info_mode == 2) hipLaunchKernelGGL((k_linearize_lean_bsr<2, WAVES>), grid, block, lds, s, g, gate);
  else hipLaunchKernelGGL((k_linearize_lean_bsr<0, WAVES>), grid, block, lds, s, g, gate);
}
void launch_linearize(const DeviceGraph& g, hipStream_t s, int gate) {
  if (lean_bsr_fits(g)) {
    // three waves per SIMD: 168 registers without a spill (INFO 0, 3); block-diagonal information needs 12 bytes of scratch there: two
    if (g.info_mode == 2) launch_lean_bsr_w<2>(g, s, gate); else launch_lean_bsr_w<3>(g, s, gate);
    return;
  }
  const size_t lds = (size_t)NV_LIN * g.block * sizeof(double);
  if (g.info_mode == 3) hipLaunchKernelGGL(k_linearize<3>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
  else if (g.info_mode == 2) hipLaunchKernelGGL(k_linearize<2>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
  else if (g.info_mode) hipLaunchKernelGGL(k_linearize<1>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
  else hipLaunchKernelGGL(k_linearize<0>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
}
void launch_linearize_symout(const DeviceGraph& g, hipStream_t s, int gate) {
  const size_t lds = (size_t)NV_LIN * g.block * sizeof(double);
  if (g.info_mode == 3) hipLaunchKernelGGL(k_linearize_symout<3>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
  else if (g.info_mode == 2) hipLaunchKernelGGL(k_linearize_symout<2>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
  else if (g.info_mode) hipLaunchKernelGGL(k_linearize_symout<1>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
  else hipLaunchKernelGGL(k_linearize_symout<0>, dim3(g.n_wg), dim3(g.block), lds, s, g, gate);
}
void launch_scale_from_diag(const DeviceGraph& g, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_from_diag, dim3(cdiv(6 * g.N, 256)), dim3(256), 0, s, g);
}
void launch_damping(const DeviceGraph& g, double radius, double min_diag, double max_diag, int mode, hipStream_t s) {
  const int owned = g.row_hi - g.row_lo;
  // one rank + cluster preconditioner: the cluster kernel does the damping of its poses itself (every pose is in a cluster)
  const bool fused = g.cluster > 1 && g.world == 1 && owned > 0;
  if (!fused) hipLaunchKernelGGL(k_damping, dim3(cdiv(g.N, 64)), dim3(64), 0, s, g, radius, min_diag, max_diag, mode);
  const int cm = fused ? mode : -1;
  // 64 / (6 CL) clusters per wave: 5 (CL = 2) or 2 (CL = 4)
  if (g.cluster == 2 && owned > 0) hipLaunchKernelGGL(k_cluster_precond<2>, dim3(cdiv(cdiv(owned, 2), 5)), dim3(64), 0, s, g, radius, min_diag, max_diag, cm);
  else if (g.cluster == 4 && owned > 0) hipLaunchKernelGGL(k_cluster_precond<4>, dim3(cdiv(cdiv(owned, 4), 2)), dim3(64), 0, s, g, radius, min_diag, max_diag, cm);
}
void launch_cost(const DeviceGraph& g, const double* poses, int part_row, hipStream_t s, int gate) {
  double* part = g.part_misc + (size_t)part_row * g.n_part;
  if (g.info_mode) hipLaunchKernelGGL(k_cost<1>, dim3(g.n_edge_wg), dim3(EDGE_BLOCK), 0, s, g, poses, part, gate);
  else hipLaunchKernelGGL(k_cost<0>, dim3(g.n_edge_wg), dim3(EDGE_BLOCK), 0, s, g, poses, part, gate);
}
void launch_evaluate_edges(const DeviceGraph& g, const double* poses, double* res, double* ja, double* jb, hipStream_t s) {
  hipLaunchKernelGGL(k_evaluate_edges, dim3(cdiv(g.E, 128)), dim3(128), 0, s, g, poses, res, ja, jb);
}
void launch_pcg_init(const DeviceGraph& g, hipStream_t s) {
  if (g.cluster == 2) hipLaunchKernelGGL(k_pcg_init<2>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g);
  else if (g.cluster == 4) hipLaunchKernelGGL(k_pcg_init<4>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g);
  else hipLaunchKernelGGL(k_pcg_init<1>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g);
}
static void launch_update(const DeviceGraph& g, int odd, hipStream_t s, int mode = 0) {
  if (g.cluster == 2) hipLaunchKernelGGL(k_pcg_update<2>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g, odd, mode);
  else if (g.cluster == 4) hipLaunchKernelGGL(k_pcg_update<4>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g, odd, mode);
  else hipLaunchKernelGGL(k_pcg_update<1>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g, odd, mode);
}
void launch_pcg_iteration(const DeviceGraph& g, const CgParams& p, int odd, hipStream_t s) {
  const size_t lds = (size_t)(SPMV_LDS_STRIDE + 6) * g.block * sizeof(double);
  if (g.blk_packed) hipLaunchKernelGGL((k_spmv<0, true>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, odd);
  else hipLaunchKernelGGL((k_spmv<0, false>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, odd);
  launch_update(g, odd, s);
}
void launch_pcg_finish(const DeviceGraph& g, const CgParams& p, hipStream_t s, int publish) {
  hipLaunchKernelGGL(k_pcg_finish, dim3(1), dim3(256), 0, s, g, p, publish);
}
void launch_spmv_plain(const DeviceGraph& g, hipStream_t s) {
  const size_t lds = (size_t)(SPMV_LDS_STRIDE + 6) * g.block * sizeof(double);
  CgParams dummy{0.0, -1.0, 0, 0};
  if (g.blk_packed) hipLaunchKernelGGL((k_spmv<1, true>), dim3(g.n_wg), dim3(g.block), lds, s, g, dummy, 1);
  else hipLaunchKernelGGL((k_spmv<1, false>), dim3(g.n_wg), dim3(g.block), lds, s, g, dummy, 1);
}
// finish: 1 = apply the stop test of the last CG iteration first and run only once the CG has stopped; 2 = only the latter (the
// owner-only CG applied its own test)
void launch_spmv_tail(const DeviceGraph& g, const CgParams& p, hipStream_t s, int finish, int candidates) {
  const size_t lds = (size_t)(SPMV_LDS_STRIDE + 6) * g.block * sizeof(double);
  const int flags = 1 | (finish == 1 ? 2 : 0) | (finish == 2 ? 64 : 0) | (candidates ? 4 : 0);
  if (g.blk_packed) hipLaunchKernelGGL((k_spmv<1, true>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, flags);
  else hipLaunchKernelGGL((k_spmv<1, false>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, flags);
}
void launch_step_tail(const DeviceGraph& g, hipStream_t s, int gate) {
  const int grid = g.n_edge_wg + g.n_pose_wg;   // <= 2 * n_part
  if (g.info_mode) hipLaunchKernelGGL(k_step_tail<1>, dim3(grid), dim3(EDGE_BLOCK), 0, s, g, gate);
  else hipLaunchKernelGGL(k_step_tail<0>, dim3(grid), dim3(EDGE_BLOCK), 0, s, g, gate);
}
void launch_pcg_spmv_only(const DeviceGraph& g, const CgParams& p, int odd, hipStream_t s) {
  const size_t lds = (size_t)(SPMV_LDS_STRIDE + 6) * g.block * sizeof(double);
  if (g.blk_packed) hipLaunchKernelGGL((k_spmv<0, true>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, odd);
  else hipLaunchKernelGGL((k_spmv<0, false>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, odd);
}
void launch_pcg_update_only(const DeviceGraph& g, int odd, hipStream_t s, int mode) {
  launch_update(g, odd, s, mode);
}
// A x -> cg_q for the residual refresh; skipped once the CG has stopped.  on_the_fly (one rank): x = x_old + alpha p is
// formed while gathering (it_odd = parity of the iteration, selects the p buffer)
void launch_spmv_refresh(const DeviceGraph& g, hipStream_t s, int on_the_fly, int it_odd) {
  const size_t lds = (size_t)(SPMV_LDS_STRIDE + 6) * g.block * sizeof(double);
  CgParams dummy{0.0, -1.0, 0, 0};
  if (g.blk_packed) hipLaunchKernelGGL((k_spmv<1, true>), dim3(g.n_wg), dim3(g.block), lds, s, g, dummy, 1 | 8 | (on_the_fly ? 16 : 0) | (it_odd ? 32 : 0));
  else hipLaunchKernelGGL((k_spmv<1, false>), dim3(g.n_wg), dim3(g.block), lds, s, g, dummy, 1 | 8 | (on_the_fly ? 16 : 0) | (it_odd ? 32 : 0));
}
void launch_model_delta_and_retract(const DeviceGraph& g, hipStream_t s, int gate) {
  hipLaunchKernelGGL(k_model_delta, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g, gate);
  hipLaunchKernelGGL(k_retract, dim3(g.n_pose_wg), dim3(POSE_BLOCK), 0, s, g, gate);
}
void launch_gradient_norm(const DeviceGraph& g, hipStream_t s) {
  hipLaunchKernelGGL(k_gradient_norm, dim3(g.n_pose_wg), dim3(POSE_BLOCK), 0, s, g);
}
void launch_accept_finish(const DeviceGraph& g, int seq_id, hipStream_t s) {
  hipLaunchKernelGGL(k_accept_finish, dim3(g.n_pose_wg), dim3(POSE_BLOCK), 0, s, g, seq_id);
}
void launch_lm_resume(const DeviceGraph& g, int cg_goes_on, hipStream_t s) {
  hipLaunchKernelGGL(k_lm_resume, dim3(1), dim3(1), 0, s, g, cg_goes_on);
}
static inline int uni_v_grid(const DeviceGraph& g) { return std::max(g.n_vec_wg, g.n_edge_wg + g.n_pose_wg); }
// (the universal streams' LIN operation is the lean body wherever the slots are packed: 32-bit byte offsets, see lean_bsr_fits)
bool uni_supported(const DeviceGraph& g) { return g.world == 1 && g.block <= 256 && (g.info_mode == 1 || lean_bsr_fits(g)); }
void launch_uni_s(const DeviceGraph& g, const CgParams& p, int period, hipStream_t s) {
  const size_t lds = (size_t)NV_LIN * g.block * sizeof(double);
#define PGO_UNI_S(PK, INF) hipLaunchKernelGGL((k_uni_s<PK, INF>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, period)
  if (g.blk_packed) {         // info modes 0, 2, 3 (prepare(): general information, or PGO_BLK_FULL=1, keeps 36-entry slots and mode 1)
    if (g.info_mode == 3) PGO_UNI_S(true, 3);
    else if (g.info_mode == 2) PGO_UNI_S(true, 2);
    else PGO_UNI_S(true, 0);
  } else if (g.info_mode == 0) PGO_UNI_S(false, 0);
  else PGO_UNI_S(false, 1);
#undef PGO_UNI_S
}
void launch_uni_v(const DeviceGraph& g, const CgParams& p, double min_diag, double max_diag, hipStream_t s) {
  const int grid = uni_v_grid(g);
#define PGO_UNI_V(CLV) do { if (g.info_mode) hipLaunchKernelGGL((k_uni_v<CLV, 1>), dim3(grid), dim3(UNI_V_BLOCK), 0, s, g, p, min_diag, max_diag); \
                            else hipLaunchKernelGGL((k_uni_v<CLV, 0>), dim3(grid), dim3(UNI_V_BLOCK), 0, s, g, p, min_diag, max_diag); } while (0)
  if (g.cluster == 2) PGO_UNI_V(2);
  else if (g.cluster == 4) PGO_UNI_V(4);
  else PGO_UNI_V(1);
#undef PGO_UNI_V
}
void launch_hdiag6(const DeviceGraph& g, double* buf, int phase, hipStream_t s) {
  hipLaunchKernelGGL(k_hdiag6, dim3(cdiv(6 * g.N, 256)), dim3(256), 0, s, g, buf, phase);
}
bool pipe_supported(const DeviceGraph& g, const CgParams& p, int cluster) {
  return g.world > 1 && g.pairs_whole && g.pipe_buf[0] && (cluster == 1 || cluster == 2) && p.q_tolerance >= 0.0 && p.r_tolerance < 0.0;
}
void launch_pipe_init(const DeviceGraph& g, hipStream_t s) {
  if (g.cluster == 2) hipLaunchKernelGGL(k_pipe_init<2>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g);
  else hipLaunchKernelGGL(k_pipe_init<1>, dim3(g.n_vec_wg), dim3(VEC_BLOCK), 0, s, g);
}
void launch_peer_signal(const DeviceGraph& g, unsigned long long gseq, hipStream_t s) { hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(1), 0, s, g, gseq); }
void launch_pipe_cg(const DeviceGraph& g, const CgParams& p, int seq, int mode, hipStream_t s, unsigned long long gseq, bool fold) {
  const size_t lds = ((size_t)(SPMV_LDS_STRIDE + 6) * g.block + 8) * sizeof(double);
  const dim3 grid(mode ? 1 : g.n_wg);
#define PGO_PIPE(PK) do { if (g.cluster == 2) hipLaunchKernelGGL((k_pipe_cg<PK, 2>), grid, dim3(g.block), lds, s, g, p, seq, mode); \
                          else hipLaunchKernelGGL((k_pipe_cg<PK, 1>), grid, dim3(g.block), lds, s, g, p, seq, mode); } while (0)
  if (g.blk_packed) PGO_PIPE(true); else PGO_PIPE(false);
#undef PGO_PIPE
  if (mode == 0 && fold) { if (g.bx[0]) launch_pipe_pack(g, (seq & 1) ^ 1, seq, g.n_wg, s); else launch_pipe_fold(g, seq, gseq, s); }
}
void launch_pipe_pack(const DeviceGraph& g, int buf, int fold_seq, int n_entries, hipStream_t s) {
  hipLaunchKernelGGL(k_pipe_pack, dim3(1 + (3 * g.bx_nb + 255) / 256), dim3(256), 0, s, g, buf, fold_seq, n_entries);
}
void launch_pipe_fold(const DeviceGraph& g, int seq, unsigned long long gseq, hipStream_t s) {
  hipLaunchKernelGGL(k_pipe_fold, dim3(1), dim3(256), 0, s, g, seq, gseq);
}
void launch_lm_budget(const DeviceGraph& g, int decisions, hipStream_t s, int next_launch) {
  hipLaunchKernelGGL(k_lm_budget, dim3(1), dim3(1), 0, s, g, decisions, next_launch);
}
// The fused stream serves the truncated CG (Q-tolerance stop, no residual test) with 6x6 or 12x12 Jacobi blocks on a row
// partition whose work-groups hold whole pose pairs and no row fatter than a work-group (prepare(): pairs_whole).
bool uni_f_supported(const DeviceGraph& g, const CgParams& p, int cluster) {
  return g.world == 1 && g.block <= 256 && g.pairs_whole && g.n_wg <= UNI_F_FOLD * g.block && g.pipe_buf[0] && g.cg_u && g.part_f &&
         (cluster == 1 || cluster == 2) &&
         p.q_tolerance >= 0.0 && p.r_tolerance < 0.0;
}
void launch_uni_f(const DeviceGraph& g, const CgParams& p, int launch, double min_diag, double max_diag, hipStream_t s) {
  const size_t lds = (size_t)NV_LIN * g.block * sizeof(double);
#define PGO_UNI_F2(PK, INF, CLV) hipLaunchKernelGGL((k_uni_f<PK, INF, CLV>), dim3(g.n_wg), dim3(g.block), lds, s, g, p, launch, min_diag, max_diag)
#define PGO_UNI_F(PK, INF) do { if (g.cluster == 2) PGO_UNI_F2(PK, INF, 2); else PGO_UNI_F2(PK, INF, 1); } while (0)
  if (g.blk_packed) {
    if (g.info_mode == 3) PGO_UNI_F(true, 3);
    else if (g.info_mode == 2) PGO_UNI_F(true, 2);
    else PGO_UNI_F(true, 0);
  } else if (g.info_mode == 0) PGO_UNI_F(false, 0);
  else PGO_UNI_F(false, 1);
#undef PGO_UNI_F
#undef PGO_UNI_F2
}
void launch_lm_publish(const DeviceGraph& g, hipStream_t s) { hipLaunchKernelGGL(k_lm_publish, dim3(1), dim3(1), 0, s, g); }
void launch_finalize_scalars(const DeviceGraph& g, int n_cost_part, hipStream_t s, int gate) {
  hipLaunchKernelGGL(k_finalize_scalars, dim3(1), dim3(256), 0, s, g, n_cost_part, gate);
}
void launch_apply_step(const DeviceGraph& g, const double* step, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_delta, dim3(cdiv(6 * g.N, 256)), dim3(256), 0, s, g, step);
  hipLaunchKernelGGL(k_retract, dim3(g.n_pose_wg), dim3(POSE_BLOCK), 0, s, g, 0);
}

void launch_batch_d2(const DeviceGraph& g, const BatchPlan& b, double min_diag, double max_diag, hipStream_t s) {
  hipLaunchKernelGGL(k_batch_d2, dim3(cdiv(6 * g.N, 256)), dim3(256), 0, s, g, b, min_diag, max_diag);
}
void launch_batch_scalars(const DeviceGraph& g, const BatchPlan& b, hipStream_t s) {
  if (g.info_mode) hipLaunchKernelGGL(k_batch_scalars<1>, dim3(b.n_comp * b.split), dim3(256), 0, s, g, b);
  else hipLaunchKernelGGL(k_batch_scalars<0>, dim3(b.n_comp * b.split), dim3(256), 0, s, g, b);
  hipLaunchKernelGGL(k_batch_fold, dim3(cdiv(b.n_comp, 64)), dim3(64), 0, s, b);
}
void launch_batch_accept(const DeviceGraph& g, const BatchPlan& b, hipStream_t s) {
  hipLaunchKernelGGL(k_batch_accept, dim3(cdiv(4 * g.N, 256)), dim3(256), 0, s, g, b);
}
void launch_debug(const DeviceGraph& g, int which, hipStream_t s) {
  if (which == 0) hipLaunchKernelGGL(k_empty, dim3(g.n_wg), dim3(g.block), 0, s, g);
  else hipLaunchKernelGGL(k_touch, dim3(cdiv(6 * g.N, 256)), dim3(256), 0, s, g);
}
int vec_block() { return VEC_BLOCK; }
int pose_block() { return POSE_BLOCK; }
int edge_block() { return EDGE_BLOCK; }
int max_edge_wg() { return 1024; }
#else   // PGO_TU_RESIDENT
// The resident stream needs what the fused one needs, every row lane in one pass (rows_fit) and a grid that is resident at once at two
// waves per SIMD (the grid barrier of k_res_cg): 8 waves per CU, 256 CUs.
bool uni_r_supported(const DeviceGraph& g, const CgParams& p, int cluster) {
  // (k_res_cg runs two waves per SIMD — __launch_bounds__(256, 2), 214 registers — i.e. eight waves per compute unit, 27 KB of LDS per work-group)
  return uni_f_supported(g, p, cluster) && g.rows_fit && g.block >= 64 && g.n_cu > 0 && (long long)g.n_wg * (g.block / 64) <= 8LL * g.n_cu;
}
int uni_r_abort_word() { return RES_ABORT; }
void launch_uni_r(const DeviceGraph& g, const CgParams& p, int launch, double min_diag, double max_diag, hipStream_t s) {
  // r06: a cycle of TWO — [LIN (behind an accepted step) + HEAD, every work-group on its own rows] | [the whole CG + the step tail + the
  // decision] — where r05 ran four launches per LM iteration (HEAD | CG | TAIL | LIN)
  const int role = launch & 1;
  const dim3 grid(g.n_wg), blk(g.block);
  if (role == 0) {
    const bool lean = g.info_mode != 1;      // (information without position / rotation coupling: packed slots, the lean algebra)
    const size_t lds = std::max((size_t)g.block, lean ? (size_t)LEAN_NV * (g.block / 2) : (size_t)NV_LIN * g.block) * sizeof(double);
#define PGO_RES_LH(INF, LN) do { if (g.cluster == 2) hipLaunchKernelGGL((k_res_lh<INF, 2, LN>), grid, blk, lds, s, g, launch, min_diag, max_diag); \
                                 else hipLaunchKernelGGL((k_res_lh<INF, 1, LN>), grid, blk, lds, s, g, launch, min_diag, max_diag); } while (0)
    if (g.info_mode == 3) PGO_RES_LH(3, true); else if (g.info_mode == 2) PGO_RES_LH(2, true); else if (g.info_mode == 1) PGO_RES_LH(1, false); else PGO_RES_LH(0, true);
#undef PGO_RES_LH
  } else {
    const size_t lds = ((size_t)(SPMV_LDS_STRIDE + 6) * g.block + 8) * sizeof(double);
#define PGO_RES_CG(PK, INF) do { if (g.cluster == 2) hipLaunchKernelGGL((k_res_cg<PK, 2, INF>), grid, blk, lds, s, g, p, launch); \
                                 else hipLaunchKernelGGL((k_res_cg<PK, 1, INF>), grid, blk, lds, s, g, p, launch); } while (0)
    // (packed 27-entry slots <=> information without position / rotation coupling: INFO 0, 2, 3)
    if (g.info_mode == 3) PGO_RES_CG(true, 3); else if (g.info_mode == 2) PGO_RES_CG(true, 2); else if (g.info_mode == 1) PGO_RES_CG(false, 1); else PGO_RES_CG(true, 0);
#undef PGO_RES_CG
  }
}
#endif  // PGO_TU_RESIDENT

}  // namespace pgo
