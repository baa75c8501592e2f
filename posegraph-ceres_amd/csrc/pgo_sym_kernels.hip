// pgo_sym_kernels.hip — block SpMV from the symmetric tile form (pgo_sym.h): every interior off-diagonal block is read once and
// used for both of its rows.  gfx950, FP64, HBM-bound: the only streams are the blocks (14 or 18 x 16 B per lane, 1 KiB per wave
// instruction, prefetched one chunk ahead) and 4 + 2 bytes of indices per slot; the vectors live in LDS.
#include "pgo_math.h"
#include "pgo_sym.h"
#include "pgo_wave.h"

namespace pgo {

namespace {

constexpr int VSTRIDE = 7;     // doubles per 6-vector in the LDS exchange buffers (odd: conflict-free 8-byte accesses)

// y = H x for a slot of side BEGIN / END / DIAG (packed: TL = el[0..8], BR = el[9..17], Q = el[18..26] — bottom-left for BEGIN and
// DIAG, top-right for END; the expressions are k_spmv's, so a cut slot rounds exactly as there)
template <bool PACKED, int NPAIR>
__device__ __forceinline__ void blk_mul(const double2 (&blk)[NPAIR], int side, const double (&x)[6], double (&y)[6]) {
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
// v = H^T x for an interior slot (always stored in the BEGIN orientation: packed H = [[TL, 0], [Q, BR]])
template <bool PACKED, int NPAIR>
__device__ __forceinline__ void blk_mul_t(const double2 (&blk)[NPAIR], const double (&x)[6], double (&v)[6]) {
  if (PACKED) {
    double el[28];
#pragma unroll
    for (int k = 0; k < BLK_PAIRS_PACKED; ++k) { el[2 * k] = blk[k].x; el[2 * k + 1] = blk[k].y; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      v[j] = el[j] * x[0] + el[3 + j] * x[1] + el[6 + j] * x[2] + el[18 + j] * x[3] + el[21 + j] * x[4] + el[24 + j] * x[5];
      v[3 + j] = el[9 + j] * x[3] + el[12 + j] * x[4] + el[15 + j] * x[5];
    }
  } else {
    double el[36];
#pragma unroll
    for (int k = 0; k < 18; ++k) { el[2 * k] = blk[k].x; el[2 * k + 1] = blk[k].y; }
#pragma unroll
    for (int j = 0; j < 6; ++j)
      v[j] = el[j] * x[0] + el[6 + j] * x[1] + el[12 + j] * x[2] + el[18 + j] * x[3] + el[24 + j] * x[4] + el[30 + j] * x[5];
  }
}

// Segmented inclusive scan of six doubles over the wave; segments = runs of equal `row` (contiguous by construction).  Inside the
// 16-lane DPP rows: row_shr 1 / 2 / 4 / 8 (pure VALU, no LDS); then the three carries across the 16-lane rows, one after the other,
// through v_readlane (a lane takes the carry iff its run reaches back to its row-of-16's first lane AND that lane continues the
// previous row-of-16's last run).  A fixed tree: the same bits on every run.
template <int D>
__device__ __forceinline__ double dpp_row_shr_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + D, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + D, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int D, int NV>
__device__ __forceinline__ void seg_scan_step(double (&u)[NV], int row, int lane) {
  const int rd = __builtin_amdgcn_update_dpp(0, row, 0x110 + D, 0xf, 0xf, true);
  const bool same = (lane & 15) >= D && rd == row;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double t = dpp_row_shr_f64<D>(u[k]);
    u[k] += same ? t : 0.0;
  }
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <int NV>
__device__ __forceinline__ void seg_scan(double (&u)[NV], int row, int lane) {
  seg_scan_step<1, NV>(u, row, lane);
  seg_scan_step<2, NV>(u, row, lane);
  seg_scan_step<4, NV>(u, row, lane);
  seg_scan_step<8, NV>(u, row, lane);
#pragma unroll
  for (int b = 1; b < 4; ++b) {
    const int r_prev = __builtin_amdgcn_readlane(row, 16 * b - 1), r_first = __builtin_amdgcn_readlane(row, 16 * b);
    const bool take = (lane >> 4) == b && r_prev == r_first && row == r_first;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const double carry = readlane_f64(u[k], 16 * b - 1);
      u[k] += take ? carry : 0.0;
    }
  }
}

// One work-group per tile.  MODE 0: the SpMV of a CG iteration with k_spmv<0>'s contract (stop test of the previous iteration and
// beta from the partial rows, x = z + beta p_old, p_new written for the tile's rows, q = A x into cg_q, partial p'q, CG state
// published by work-group 0).  MODE 1: q = A cg_x.
template <int MODE, bool PACKED>
__global__ __launch_bounds__(SYM_LANES) void k_spmv_sym(DeviceGraph g, SymGraph sg, CgParams prm, int odd) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  extern __shared__ double lds[];          // xs[6 * x_cap] | ubuf[SYM_LANES * VSTRIDE] | vbuf[SYM_LANES * VSTRIDE]
  __shared__ double scratch[32];
  const int tid = threadIdx.x, tile = blockIdx.x, lane = threadIdx.x & 63;
  double* xs = lds;
  double* ubuf = lds + (size_t)sg.x_cap * 6;
  double* vbuf = ubuf + SYM_LANES * VSTRIDE;

  // MODE 1, `odd` carries k_spmv<1>'s flags: 2 = step tail behind a CG batch (CG bookkeeping first, A x only once the CG has
  // stopped), 4 = the row lanes also write delta = -S x and the candidate Plus(x, delta), 8 = A x of a residual refresh (nothing
  // once the CG has stopped), 16 = ... of x_old + alpha p formed on the fly (32: parity of p)
  if (MODE == 1) {
    if ((odd & 64) && !g.cg->done) return;     // step tail behind a batch of the owner-only CG (k_pipe_cg_sym applied the stop test): only once the CG has stopped
    if ((odd & 8) && g.cg->done) return;
    if (!(odd & 8) && lm_halted(g)) return;
    if (odd & 2) {
      int done = g.cg->done;
      const int iters0 = g.cg->iters, status = g.cg->status;
      const int itd = done ? iters0 : g.cg->cnt_b;
      double fs[4];
      fs[0] = partial_sum(g.part_q + (size_t)(itd & 1) * g.n_part, g.n_vec_wg);
      fs[1] = partial_sum(g.part_q + (size_t)((itd + 1) & 1) * g.n_part, g.n_vec_wg);
      fs[2] = partial_sum(g.part_rr + (size_t)(itd & 1) * g.n_part, g.n_vec_wg);
      fs[3] = partial_sum(g.part_bb, g.n_vec_wg);
      block_sum<4>(fs, scratch);
      const int was_done = done;
      if (!done && itd >= 1) {
        const double Q1 = -fs[0], Q0 = -fs[1];
        const double zeta = itd * (Q1 - Q0) / Q1;
        if ((zeta < prm.q_tolerance && itd >= prm.min_iterations) || itd >= prm.max_iterations) done = 1;
        if (prm.r_tolerance >= 0.0 && sqrt(fs[2]) <= prm.r_tolerance * sqrt(fs[3]) && itd >= prm.min_iterations) done = 1;
      }
      if (tile == 0 && tid == 0) {
        if (done && !was_done) { g.cg->iters = itd; __threadfence(); g.cg->done = 1; }
        g.scal->cg_iterations = itd;
        g.scal->cg_status = done ? status : -1;
        g.scal->cg_residual_sq = fs[2];
      }
      if (!done) return;
    }
  }
  double alpha_fly = 0.0;
  if (MODE == 1 && (odd & 16)) {
    double pqs[1] = {0.0};
    const double* pqp = g.cg_q + (size_t)g.rows_per * 6;
    for (int i = tid; i < g.pq_cap; i += SYM_LANES) pqs[0] += pqp[i];
    block_sum<1>(pqs, scratch);
    if (!(pqs[0] > 0.0) || !isfinite(pqs[0])) return;     // indefinite: the update kernel stops the CG
    alpha_fly = g.cg->rho / pqs[0];
  }

  // ---- loads that depend on nothing: the tile and its first two chunks (the chunk loop keeps two chunks of blocks in flight) ----
  const SymTile T = sg.tile[tile];
  const int nch = T.nchunks;
  struct Chunk { double2 b[NPAIR]; uint32_t meta, rin; int n; };
  auto load_blocks = [&](Chunk& C, int ci, int base, int n) {
    C.n = n;
    C.rin = sg.rinfo[(size_t)ci * SYM_LANES + tid];
    C.meta = 0;
    if (tid < n) {
      const int t = base + tid;
      C.meta = sg.meta[t];
      const double2* bp = reinterpret_cast<const double2*>(sg.val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) C.b[k] = bp[(size_t)k * 64];
    }
  };
  // (a tile's chunks are consecutive: base and size follow from the tile record.  A look-up in the chunk table is a uniform load,
  // which the compiler issues as a vector load and waits for with vmcnt(0) — a wait for every block in flight: 81.8 -> 78.4 us)
  auto load_chunk = [&](Chunk& C, int c) { load_blocks(C, T.chunk0 + c, T.base0 + SYM_LANES * c, min((int)SYM_LANES, T.total - SYM_LANES * c)); };
  Chunk CA, CB, CC;
  load_blocks(CA, T.chunk0, T.base0, T.n0);
  if (nch > 1) load_blocks(CB, T.chunk0 + 1, T.base1, T.n1);
  // x staging, first half (independent of beta): z (MODE 1: x itself) of every staged column into xs, p_old into the exchange
  // buffers' space as far as it reaches (597 columns; they are not needed before the chunk loop) — so that these gathers are in
  // flight together with the first blocks and the partial rows of the prologue
  double* ps = ubuf;
  constexpr int PS_CAP = 2 * SYM_LANES * VSTRIDE / 6;
  const double* p_old = odd ? g.cg_p0 : g.cg_p1;
  for (int i = tid; i < T.nx * 3; i += SYM_LANES) {
    const int e = i / 3, k = i - 3 * e;
    const int pose = sg.xlist[T.x0 + e];
    double2 z = reinterpret_cast<const double2*>((MODE == 0 ? g.cg_z : g.cg_x) + 6 * (size_t)pose)[k];
    if (MODE == 1 && (odd & 16)) {
      const double2 pc = reinterpret_cast<const double2*>(((odd & 32) ? g.cg_p1 : g.cg_p0) + 6 * (size_t)pose)[k];
      z = double2{z.x + alpha_fly * pc.x, z.y + alpha_fly * pc.y};
    }
    xs[6 * e + 2 * k] = z.x;
    xs[6 * e + 2 * k + 1] = z.y;
    if (MODE == 0 && e < PS_CAP) {
      const double2 p = reinterpret_cast<const double2*>(p_old + 6 * (size_t)pose)[k];
      ps[6 * e + 2 * k] = p.x;
      ps[6 * e + 2 * k + 1] = p.y;
    }
  }

  // ---- CG prologue (k_spmv<0>): every work-group re-derives rho, beta and the stop test from the same partial rows ----
  double beta = 0.0, rho_pub = 0.0, q_pub = 0.0;
  int it = 1;
  if (MODE == 0) {
    const int done = g.cg->done, cnt_b = g.cg->cnt_b;
    const double* rz_cur = g.part_rz + (size_t)(odd ? 0 : g.n_part);
    const double* q_cur = g.part_q + (size_t)(odd ? 0 : g.n_part);
    const double* rr_cur = g.part_rr + (size_t)(odd ? 0 : g.n_part);
    const double hist_rho = g.cg->rho_hist[odd ? 0 : 1], hist_q = g.cg->q_hist[odd ? 0 : 1];
    const bool need_rr = prm.r_tolerance >= 0.0;
    double sums[6] = {0, 0, 0, 0, 0, 0};
    if (need_rr) {
      for (int i = tid; i < g.n_vec_wg; i += SYM_LANES) { sums[0] += rz_cur[i]; sums[2] += q_cur[i]; sums[4] += rr_cur[i]; sums[5] += g.part_bb[i]; }
    } else {
      for (int i = tid & 63; i < g.n_vec_wg; i += 64) { sums[0] += rz_cur[i]; sums[2] += q_cur[i]; }
    }
    if (done) return;
    it = cnt_b + 1;
    double rr = 0.0, bb = 0.0;
    if (need_rr) { block_sum<6>(sums, scratch); rr = sums[4]; bb = sums[5]; }
    else { sums[0] = wave_sum(sums[0]); sums[2] = wave_sum(sums[2]); }
    const double rho = sums[0], Q1 = -sums[2];
    rho_pub = rho;
    q_pub = Q1;
    int stop = 0, status = 0;
    if (it > 1) {
      const int done_it = it - 1;
      const double zeta = done_it * (Q1 - hist_q) / Q1;
      if (zeta < prm.q_tolerance && done_it >= prm.min_iterations) stop = 1;
      if (need_rr && sqrt(rr) <= prm.r_tolerance * sqrt(bb) && done_it >= prm.min_iterations) stop = 1;
      if (done_it >= prm.max_iterations) stop = 1;
    }
    if (!stop && (rho == 0.0 || !isfinite(rho))) { stop = 1; status = (rho == 0.0) ? 0 : 2; }
    if (!stop && it > 1) {
      beta = rho / hist_rho;
      if (beta == 0.0 || !isfinite(beta)) { stop = 1; status = 2; }
    }
    if (stop) {
      if (tile == 0 && tid == 0) { g.cg->iters = it - 1; g.cg->status = status; g.cg->done = 1; }
      return;
    }
  }

  // ---- x staging, second half: x = z + beta p_old (every lane combines the entries it staged itself: no barrier in between),
  // which is also p_new of the tile's rows ----
  if (MODE == 0) {
    double* p_new = odd ? g.cg_p1 : g.cg_p0;
    for (int i = tid; i < T.nx * 3; i += SYM_LANES) {
      const int e = i / 3, k = i - 3 * e;
      double2 p;
      if (e < PS_CAP) p = double2{ps[6 * e + 2 * k], ps[6 * e + 2 * k + 1]};
      else p = reinterpret_cast<const double2*>(p_old + 6 * (size_t)sg.xlist[T.x0 + e])[k];
      const double2 v{xs[6 * e + 2 * k] + beta * p.x, xs[6 * e + 2 * k + 1] + beta * p.y};
      xs[6 * e + 2 * k] = v.x;
      xs[6 * e + 2 * k + 1] = v.y;
      if (e < T.nrows) reinterpret_cast<double2*>(p_new + 6 * (size_t)sg.xlist[T.x0 + e])[k] = v;
    }
  }
  __syncthreads();

  double y[6] = {0, 0, 0, 0, 0, 0};       // row `tid` of the tile
  auto process = [&](const Chunk& C) {
    // u = H x of this lane's slot (0 for the idle lanes of a tile's last chunk), v = H^T x_row of an interior slot
    double u[6] = {0, 0, 0, 0, 0, 0};
    int row = -1 - lane;                     // idle lanes: a row id nobody shares
    if (tid < C.n) {
      const int xcol = (int)(C.meta & 0xFFFu), side = (int)((C.meta >> 12) & 3u);
      row = (int)((C.meta >> 23) & 0xFFu);
      double x[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = xs[6 * xcol + k];
      blk_mul<PACKED, NPAIR>(C.b, side, x, u);
      if (C.meta & (1u << 14)) {
        const int vpos = (int)((C.meta >> 15) & 0xFFu);
        double xr[6], v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xr[k] = xs[6 * row + k];
        blk_mul_t<PACKED, NPAIR>(C.b, xr, v);
#pragma unroll
        for (int k = 0; k < 6; ++k) vbuf[vpos * VSTRIDE + k] = v[k];
      }
    }
    // The slots of a row sit in consecutive lanes: segmented inclusive scan inside the wave (fixed tree, deterministic), so that
    // the LAST lane of every (row, wave) run holds the run's sum and only those lanes go through LDS — the row's lane then adds
    // one entry per wave its slots span instead of one per slot.
    if (!PGO_ABLATION(g, 512)) seg_scan<6>(u, row, lane);
    {
      const int rn = __shfl_down(row, 1, 64);
      if (tid < C.n && (lane == 63 || rn != row)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ubuf[tid * VSTRIDE + k] = u[k];
      }
    }
    __syncthreads();
    {
      const int ub = (int)(C.rin & 0xFFu), uc = (int)((C.rin >> 8) & 0x1FFu), vb = (int)((C.rin >> 17) & 0xFFu), vc = (int)(C.rin >> 25);
      if (uc > 0) {
        const int last = ub + uc - 1;
        for (int w = ub >> 6; w <= (last >> 6); ++w) {
          const int tail = min(w * 64 + 63, last);
#pragma unroll
          for (int k = 0; k < 6; ++k) y[k] += ubuf[tail * VSTRIDE + k];
        }
      }
      for (int j = 0; j < vc; ++j) {
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] += vbuf[(vb + j) * VSTRIDE + k];
      }
    }
    __syncthreads();
  };
  const int nrun = PGO_ABLATION(g, 256) ? 0 : nch;     // (development ablation)
  for (int c = 0; c < nrun; c += 3) {
    if (c + 2 < nch) load_chunk(CC, c + 2);
    process(CA);
    if (c + 1 < nch) {
      if (c + 3 < nch) load_chunk(CA, c + 3);
      process(CB);
    }
    if (c + 2 < nch) {
      if (c + 4 < nch) load_chunk(CB, c + 4);
      process(CC);
    }
  }

  double pq[1] = {0.0};
  if (tid < T.nrows) {
    const int pose = sg.xlist[T.x0 + tid];
    if (g.world == 1) {
      double2* q = reinterpret_cast<double2*>(g.cg_q + 6 * (size_t)pose);      // one rank: q_index(row, k) = 6 row + k
      q[0] = double2{y[0], y[1]};
      q[1] = double2{y[2], y[3]};
      q[2] = double2{y[4], y[5]};
    } else {                // several ranks: the owner's segment of the exchange buffer (pgo_kernels.hip q_index; a segment need not start on 16 bytes)
      const int rk = pose / g.rows_per;
      double* q = g.cg_q + (size_t)rk * g.seg + (size_t)(pose - rk * g.rows_per) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) q[k] = y[k];
    }
    if (MODE == 1 && (odd & 4)) {
      // step tail: delta = -S x and the candidate Plus(x, delta) of this row (k_retract's job)
      const double2* ps2 = reinterpret_cast<const double2*>(g.pose_x + (size_t)POSE_STRIDE * pose);
      const double2 a = ps2[0], b = ps2[1], c2 = ps2[2], d2 = ps2[3];
      const V3 Pp{a.x, a.y, b.x};
      const Q4 Pq{b.y, c2.x, c2.y, d2.x};
      const uint8_t m = g.cmask[pose];
      double d[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bool cst = (i < 3) ? (m & 1) : (m & 2);
        d[i] = cst ? 0.0 : -g.scale[6 * (size_t)pose + i] * xs[6 * tid + i];
        g.delta[6 * (size_t)pose + i] = d[i];
      }
      V3 pc = Pp;
      Q4 qc = Pq;
      if (!(m & 1)) pc = V3{Pp.x + d[0], Pp.y + d[1], Pp.z + d[2]};
      if (!(m & 2)) qc = quat_plus(Pq, V3{d[3], d[4], d[5]});
      double2* o = reinterpret_cast<double2*>(g.pose_c + (size_t)POSE_STRIDE * pose);
      o[0] = double2{pc.x, pc.y};
      o[1] = double2{pc.z, qc.x};
      o[2] = double2{qc.y, qc.z};
      o[3] = double2{qc.w, 0.0};
    }
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pq[0] += y[k] * xs[6 * tid + k];
    }
  }
  if (MODE == 0) {
    block_sum<1>(pq, scratch);
    if (tid == 0) {
      double* pqp = g.cg_q + (size_t)g.rows_per * 6;
      pqp[tile] = pq[0];
      for (int i = tile + sg.n_tiles; i < g.pq_cap; i += sg.n_tiles) pqp[i] = 0.0;   // the update kernel folds pq_cap partials
      if (tile == 0) {
        g.cg->cnt_a = it; g.cg->beta = beta; g.cg->rho = rho_pub;
        g.cg->rho_hist[odd ? 1 : 0] = rho_pub;
        g.cg->q_hist[odd ? 1 : 0] = q_pub;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// r06 — the owner-only pipelined CG (pgo_kernels.hip k_pipe_cg: Ghysels & Vanroose, one global reduction per iteration) with its
// product taken from the symmetric tile form: ONE launch per CG iteration does
//     fold of every rank's three sums -> stop test, alpha, beta          (every work-group alike: same bits everywhere)
//     m of the tile's rows and ghost columns from the exchange buffer -> LDS
//     n = A m over the tile's stored slots (every interior block read once, used for both of its rows: k_spmv_sym's chunk loop)
//     the eight vector recurrences of the tile's rows (lane r = row r, six components each)
//     m_new = M^-1 w with the row's 6x6 / 12x12 Jacobi block (the cluster's w through LDS: tiles are made of whole clusters)
//     m_new into the exchange buffer(s), three partial sums per tile (folded by k_pipe_fold behind this launch)
// It serves the sharded ranks (each rank holds the form of ITS rows: cut edges to other ranks are ghost columns) and, on one rank,
// the graphs above the universal stream's size limit (BASELINE configs[3] on one GPU), where it replaces k_spmv_sym<0> +
// k_pcg_update + k_cluster_precond's reloads: one launch and one pass over the vectors per iteration instead of two.
// Launch `seq` reads pipe_buf[seq & 1] / CgState::pipe[seq & 1] and writes the other ones; seq 0: w0 = A u0.  The single-work-group
// "stop test only" launches at the end of a batch are k_pipe_cg's (mode 1 / 2): same state words, same sums.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t sym_pipe_index(const DeviceGraph& g, int row) {
  if (g.world == 1) return (size_t)row * 6;
  const int rk = row / g.rows_per;
  return (size_t)rk * g.pipe_seg + (size_t)(row - rk * g.rows_per) * 6;
}

// "am I the last tile of this launch to get here?" — the fused stream's two-level ticket (pgo_uni_fused.h uni_f_last_arrival) on the same words
// of g.flags: tile t bumps the counter of its class t % 8, the last of a class the top counter, the last of those resets the nine words.  One
// lane calls it, behind its write-through stores.
__device__ __forceinline__ bool sym_last_arrival(const DeviceGraph& g, int t, int n) {
  const int cls = t & 7;
  const int in_class = (n - cls + 7) >> 3;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (atomicAdd(&g.flags[4 + cls], 1) != in_class - 1) return false;
  const int classes = min(n, 8);
  if (atomicAdd(&g.flags[12], 1) != classes - 1) return false;
#pragma unroll
  for (int k = 0; k < 9; ++k) g.flags[4 + k] = 0;
  return true;
}

template <bool PACKED, int CL>
__global__ __launch_bounds__(SYM_LANES) void k_pipe_cg_sym(DeviceGraph g, SymGraph sg, CgParams prm, int seq) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  constexpr int DIM = 6 * CL;
  extern __shared__ double lds[];          // xs[6 * x_cap] | ubuf[SYM_LANES * VSTRIDE] | vbuf[SYM_LANES * VSTRIDE]
  __shared__ double scratch[32];
  const int tid = threadIdx.x, tile = blockIdx.x, lane = threadIdx.x & 63;
  double* xs = lds;
  double* ubuf = lds + (size_t)sg.x_cap * 6;
  double* vbuf = ubuf + SYM_LANES * VSTRIDE;
  const int rs = seq & 1, ws = rs ^ 1;
  const double* rd = g.pipe_buf[rs];
  double* wr = g.pipe_buf[ws];

  // ---- loads that depend on nothing.  What the stop test needs — the CG state and the sums — is requested FIRST, so that a launch
  // enqueued past the CG's end leaves without waiting for blocks; then the tile's first two chunks ----
  const SymTile T = sg.tile[tile];
  const int done0 = g.cg->done;
  const CgState::Pipe st = g.cg->pipe[rs];
  const bool w0 = seq == 0;
  double f3[3] = {0.0, 0.0, 0.0};
  if (g.world == 1) {
    // one rank: every work-group folds the per-tile partial triples the launch before left (row `rs` of the partial-sum arrays; this
    // launch writes row `ws`) — k_pipe_fold's loop and block sum, the same bits — instead of a one-work-group launch in between
    if (!w0) {
      const double* a0 = g.part_rz + (size_t)rs * g.n_part, *a1 = g.part_q + (size_t)rs * g.n_part, *a2 = g.part_rr + (size_t)rs * g.n_part;
      for (int i = tid; i < sg.n_tiles; i += SYM_LANES) { f3[0] += a0[i]; f3[1] += a1[i]; f3[2] += a2[i]; }
    }
  } else {
    // every rank's three sums, added in rank order by every lane alike: same bits everywhere
    for (int rk = 0; rk < g.world; ++rk) {
      const double* pp = g.bx[0] ? g.bx[rs] + (size_t)(rk + 1) * g.bx_cseg - 4 : rd + (size_t)rk * g.pipe_seg + (size_t)g.rows_per * 6;
      f3[0] += pp[0]; f3[1] += pp[1]; f3[2] += pp[2];
    }
  }
  const int nch = T.nchunks;
  struct Chunk { double2 b[NPAIR]; uint32_t meta, rin; int n; };
  auto load_blocks = [&](Chunk& C, int ci, int base, int n) {
    C.n = n;
    C.rin = sg.rinfo[(size_t)ci * SYM_LANES + tid];
    C.meta = 0;
    if (tid < n) {
      const int t = base + tid;
      C.meta = sg.meta[t];
      const double2* bp = reinterpret_cast<const double2*>(sg.val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) C.b[k] = bp[(size_t)k * 64];
    }
  };
  auto load_chunk = [&](Chunk& C, int c) { load_blocks(C, T.chunk0 + c, T.base0 + SYM_LANES * c, min((int)SYM_LANES, T.total - SYM_LANES * c)); };
  Chunk CA, CB, CC;
  load_blocks(CA, T.chunk0, T.base0, T.n0);
  if (nch > 1) load_blocks(CB, T.chunk0 + 1, T.base1, T.n1);
  if (g.world == 1 && !w0) block_sum<3>(f3, scratch);
  const double f_gamma = f3[0], f_delta = f3[1], f_q = f3[2];
  int stop = 0, status = 0;
  double alpha = 0.0, beta = 0.0, gamma = 0.0, Q1 = 0.0;
  const int cnt = st.cnt;
  if (!done0 && !w0) {          // (k_pipe_cg's statements: the two kernels take the same decisions from the same numbers)
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
  if (done0) return;
  if (stop) {
    if (tile == 0 && tid == 0) { g.cg->iters = cnt; g.cg->status = status; __threadfence(); g.cg->done = 1; }
    return;
  }
  if (tile == 0 && tid == 0) {
    CgState::Pipe n;
    n.pad = 0;
    if (w0) { n.cnt = 0; n.gamma_prev = 0.0; n.alpha_prev = 0.0; n.q_prev = 0.0; }
    else { n.cnt = cnt + 1; n.gamma_prev = gamma; n.alpha_prev = alpha; n.q_prev = Q1; }
    g.cg->pipe[ws] = n;
  }
  // m of every staged column (the tile's rows first, then its ghosts — other tiles' rows, other ranks' rows)
  for (int i = tid; i < T.nx * 3; i += SYM_LANES) {
    const int e = i / 3, k = i - 3 * e;
    const double* src;
    if (g.bx[0]) { const int off = sg.xoff[T.x0 + e]; src = off >= 0 ? rd + off : g.bx[rs] + (-1 - off); }
    else src = rd + sym_pipe_index(g, sg.xlist[T.x0 + e]);
    const double2 z = reinterpret_cast<const double2*>(src)[k];
    xs[6 * e + 2 * k] = z.x;
    xs[6 * e + 2 * k + 1] = z.y;
  }
  __syncthreads();

  // ---- n = A m: k_spmv_sym's chunk loop (lane r keeps the sum of the tile's r-th row) ----
  double y[6] = {0, 0, 0, 0, 0, 0};
  auto process = [&](const Chunk& C) {
    double u[6] = {0, 0, 0, 0, 0, 0};
    int row = -1 - lane;                     // idle lanes: a row id nobody shares
    if (tid < C.n) {
      const int xcol = (int)(C.meta & 0xFFFu), side = (int)((C.meta >> 12) & 3u);
      row = (int)((C.meta >> 23) & 0xFFu);
      double x[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = xs[6 * xcol + k];
      blk_mul<PACKED, NPAIR>(C.b, side, x, u);
      if (C.meta & (1u << 14)) {
        const int vpos = (int)((C.meta >> 15) & 0xFFu);
        double xr[6], v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) xr[k] = xs[6 * row + k];
        blk_mul_t<PACKED, NPAIR>(C.b, xr, v);
#pragma unroll
        for (int k = 0; k < 6; ++k) vbuf[vpos * VSTRIDE + k] = v[k];
      }
    }
    seg_scan<6>(u, row, lane);
    {
      const int rn = __shfl_down(row, 1, 64);
      if (tid < C.n && (lane == 63 || rn != row)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ubuf[tid * VSTRIDE + k] = u[k];
      }
    }
    __syncthreads();
    {
      const int ub = (int)(C.rin & 0xFFu), uc = (int)((C.rin >> 8) & 0x1FFu), vb = (int)((C.rin >> 17) & 0xFFu), vc = (int)(C.rin >> 25);
      if (uc > 0) {
        const int last = ub + uc - 1;
        for (int w = ub >> 6; w <= (last >> 6); ++w) {
          const int tail = min(w * 64 + 63, last);
#pragma unroll
          for (int k = 0; k < 6; ++k) y[k] += ubuf[tail * VSTRIDE + k];
        }
      }
      for (int j = 0; j < vc; ++j) {
#pragma unroll
        for (int k = 0; k < 6; ++k) y[k] += vbuf[(vb + j) * VSTRIDE + k];
      }
    }
    __syncthreads();
  };
  // the row lane's vectors are requested while the last chunks are multiplied (they depend on nothing the loop produces)
  const bool own = tid < T.nrows;
  const int pose = own ? sg.xlist[T.x0 + tid] : 0;
  const int bxi = (own && g.world > 1 && g.bx[0]) ? sg.xbidx[T.x0 + tid] : -1;      // (boundary exchange: this row's place in the rank's segment)
  const size_t gj = 6 * (size_t)pose;
  double2 vr[3], vu[3], vw[3], vz[3], vq[3], vs[3], vp[3], vx[3], vb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { vr[k] = vu[k] = vw[k] = vz[k] = vq[k] = vs[k] = vp[k] = vx[k] = vb[k] = double2{0.0, 0.0}; }
  auto load_vectors = [&]() {
    if (!own) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      vr[k] = reinterpret_cast<const double2*>(g.cg_r + gj)[k];
      vu[k] = reinterpret_cast<const double2*>(g.cg_u + gj)[k];
      if (!w0) {
        vw[k] = reinterpret_cast<const double2*>(g.cg_w + gj)[k];
        vz[k] = reinterpret_cast<const double2*>(g.cg_z + gj)[k];
        vq[k] = reinterpret_cast<const double2*>(g.cg_qq + gj)[k];
        vs[k] = reinterpret_cast<const double2*>(g.cg_s + gj)[k];
        vp[k] = reinterpret_cast<const double2*>(g.cg_p0 + gj)[k];
        vx[k] = reinterpret_cast<const double2*>(g.cg_x + gj)[k];
        vb[k] = reinterpret_cast<const double2*>(g.cg_b + gj)[k];
      }
    }
  };
  double pm[6] = {0, 0, 0, 0, 0, 0};        // the row's own entries of m (the recurrence of qq needs them), read before xs is reused
  if (own) {
#pragma unroll
    for (int k = 0; k < 6; ++k) pm[k] = xs[6 * tid + k];
  }
  for (int c = 0; c < nch; c += 3) {
    if (c + 2 < nch) load_chunk(CC, c + 2);
    process(CA);
    if (c + 1 < nch) {
      if (c + 3 < nch) load_chunk(CA, c + 3);
      process(CB);
    }
    if (c + 2 < nch) {
      if (c + 4 < nch) load_chunk(CB, c + 4);
      process(CC);
    }
  }
  load_vectors();

  // ---- the recurrences of the tile's rows (k_pipe_cg's statements per component) ----
  double acc[3] = {0.0, 0.0, 0.0};
  double wn[6] = {0, 0, 0, 0, 0, 0};
  if (own) {
    double r6[6], u6[6], w6[6], z6[6], q6[6], s6[6], p6[6], x6[6], b6[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      r6[2 * k] = vr[k].x; r6[2 * k + 1] = vr[k].y; u6[2 * k] = vu[k].x; u6[2 * k + 1] = vu[k].y;
      w6[2 * k] = vw[k].x; w6[2 * k + 1] = vw[k].y; z6[2 * k] = vz[k].x; z6[2 * k + 1] = vz[k].y;
      q6[2 * k] = vq[k].x; q6[2 * k + 1] = vq[k].y; s6[2 * k] = vs[k].x; s6[2 * k + 1] = vs[k].y;
      p6[2 * k] = vp[k].x; p6[2 * k + 1] = vp[k].y; x6[2 * k] = vx[k].x; x6[2 * k + 1] = vx[k].y;
      b6[2 * k] = vb[k].x; b6[2 * k + 1] = vb[k].y;
    }
    double zn[6], qn[6], sn[6], pn[6], xn[6], rn[6], un[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double sm = y[k];
      rn[k] = r6[k]; un[k] = u6[k];
      if (w0) {
        wn[k] = sm;
        zn[k] = 0.0; qn[k] = 0.0; sn[k] = 0.0; pn[k] = 0.0; xn[k] = 0.0;
      } else {
        zn[k] = sm + beta * z6[k]; qn[k] = pm[k] + beta * q6[k]; sn[k] = w6[k] + beta * s6[k]; pn[k] = u6[k] + beta * p6[k];
        xn[k] = x6[k] + alpha * pn[k];
        rn[k] = r6[k] - alpha * sn[k];
        un[k] = u6[k] - alpha * qn[k];
        wn[k] = w6[k] - alpha * zn[k];
        acc[2] += xn[k] * (b6[k] + rn[k]);
      }
      acc[0] += rn[k] * un[k];
      acc[1] += wn[k] * un[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      reinterpret_cast<double2*>(g.cg_w + gj)[k] = double2{wn[2 * k], wn[2 * k + 1]};
      reinterpret_cast<double2*>(g.cg_z + gj)[k] = double2{zn[2 * k], zn[2 * k + 1]};
      reinterpret_cast<double2*>(g.cg_qq + gj)[k] = double2{qn[2 * k], qn[2 * k + 1]};
      reinterpret_cast<double2*>(g.cg_s + gj)[k] = double2{sn[2 * k], sn[2 * k + 1]};
      reinterpret_cast<double2*>(g.cg_p0 + gj)[k] = double2{pn[2 * k], pn[2 * k + 1]};
      if (!w0) {
        reinterpret_cast<double2*>(g.cg_x + gj)[k] = double2{xn[2 * k], xn[2 * k + 1]};
        reinterpret_cast<double2*>(g.cg_r + gj)[k] = double2{rn[2 * k], rn[2 * k + 1]};
        reinterpret_cast<double2*>(g.cg_u + gj)[k] = double2{un[2 * k], un[2 * k + 1]};
      }
    }
  }
  // ---- m_new = M^-1 w: the cluster's w through LDS (a tile is made of whole clusters in consecutive lanes: pgo_sym_host.h `unit`) ----
  double* wl = ubuf;                         // (free: the chunk loop ended with a barrier)
  if (tid < T.nrows + CL) {
#pragma unroll
    for (int k = 0; k < 6; ++k) wl[6 * tid + k] = own ? wn[k] : 0.0;      // (the lanes behind the last row: the missing poses of a last, partial cluster)
  }
  __syncthreads();
  if (own) {
    const double* wv = wl + DIM * (tid / CL);
    double mn[6];
#pragma unroll
    for (int c6 = 0; c6 < 6; ++c6) {
      const double2* Mi = reinterpret_cast<const double2*>(g.Minv + (gj + c6) * DIM);
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < DIM / 2; ++k) { const double2 mk = Mi[k]; a += mk.x * wv[2 * k] + mk.y * wv[2 * k + 1]; }
      mn[c6] = a;
    }
    const size_t pi = sym_pipe_index(g, pose);
    if (g.peer_tab) {
      for (int rk = 0; rk < g.world; ++rk) {
        double2* o = reinterpret_cast<double2*>(static_cast<double*>(g.peer_tab[3 * rk + ws]) + pi);
        o[0] = double2{mn[0], mn[1]}; o[1] = double2{mn[2], mn[3]}; o[2] = double2{mn[4], mn[5]};
      }
    } else {
      double2* o = reinterpret_cast<double2*>(wr + pi);
      o[0] = double2{mn[0], mn[1]}; o[1] = double2{mn[2], mn[3]}; o[2] = double2{mn[4], mn[5]};
      if (bxi >= 0) {        // boundary exchange: a boundary row goes straight into this rank's segment as well (no copying launch behind this one)
        double2* ob = reinterpret_cast<double2*>(g.bx[ws] + (size_t)g.rank * g.bx_cseg + 6 * (size_t)bxi);
        ob[0] = double2{mn[0], mn[1]}; ob[1] = double2{mn[2], mn[3]}; ob[2] = double2{mn[4], mn[5]};
      }
    }
  }
  block_sum<3>(acc, scratch);
  if (g.world > 1 && g.bx[0]) {
    // boundary exchange: the LAST tile to get here folds every tile's three sums (tile order: the same bits whoever is last) into the
    // end of this rank's segment — k_pipe_fold's job without its launch.  The partials travel as write-through stores in front of the
    // ticket and are read with device-scope loads (pgo_uni_fused.h uni_f_last_arrival: the fused stream's protocol, its words of g.flags —
    // a sharded session never runs that stream).
    __shared__ int last_tile;
    if (tid == 0) {
      __hip_atomic_store(&g.part_rz[tile], acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g.part_q[tile], acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g.part_rr[tile], acc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_tile = sym_last_arrival(g, tile, sg.n_tiles) ? 1 : 0;
    }
    __syncthreads();
    if (!last_tile) return;
    double t3[3] = {0.0, 0.0, 0.0};
    for (int i = tid; i < sg.n_tiles; i += SYM_LANES) {
      t3[0] += __hip_atomic_load(&g.part_rz[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t3[1] += __hip_atomic_load(&g.part_q[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t3[2] += __hip_atomic_load(&g.part_rr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_sum<3>(t3, scratch);
    if (tid == 0) { double* pp = g.bx[ws] + (size_t)(g.rank + 1) * g.bx_cseg - 4; pp[0] = t3[0]; pp[1] = t3[1]; pp[2] = t3[2]; }
    return;
  }
  if (tid == 0) {       // one rank: row `ws` (the next launch folds it itself); several: row 0, folded by k_pipe_fold behind this launch
    const size_t at = (g.world == 1 ? (size_t)ws * g.n_part : 0) + tile;
    g.part_rz[at] = acc[0]; g.part_q[at] = acc[1]; g.part_rr[at] = acc[2];
  }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_sym_repack(DeviceGraph g, SymGraph sg) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= sg.n_slots) return;
  const int src = sg.src_slot[t];
  if (src < 0) return;
  const double2* sp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(src >> 6) * TILE_DOUBLES + (size_t)(src & 63) * 2);
  double2* dp = reinterpret_cast<double2*>(sg.val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
  double2 b[NPAIR];
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) b[k] = sp[(size_t)k * 64];
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) dp[(size_t)k * 64] = b[k];
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_sym_repack_diag(DeviceGraph g, SymGraph sg) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= g.N) return;
  const int src = g.row_slot_begin[v], t = sg.diag_slot[v];
  const double2* sp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(src >> 6) * TILE_DOUBLES + (size_t)(src & 63) * 2);
  double2* dp = reinterpret_cast<double2*>(sg.val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) dp[(size_t)k * 64] = sp[(size_t)k * 64];
}

}  // namespace

size_t sym_lds_bytes(const SymGraph& sg) {
  return ((size_t)sg.x_cap * 6 + 2 * SYM_LANES * VSTRIDE) * sizeof(double);
}

void launch_spmv_sym(const DeviceGraph& g, const SymGraph& sg, const CgParams& p, int odd, int mode, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {     // more than the default 64 KiB of dynamic LDS for tiles with many ghost columns (gfx950: 160 KiB per CU)
    const int cap = 160 * 1024 - 512;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sym<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sym<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sym<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sym<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_set = true;
  }
  const size_t lds = sym_lds_bytes(sg);
  const dim3 grid(sg.n_tiles), block(SYM_LANES);
  if (mode == 0) {
    if (g.blk_packed) hipLaunchKernelGGL((k_spmv_sym<0, true>), grid, block, lds, s, g, sg, p, odd);
    else hipLaunchKernelGGL((k_spmv_sym<0, false>), grid, block, lds, s, g, sg, p, odd);
  } else {
    if (g.blk_packed) hipLaunchKernelGGL((k_spmv_sym<1, true>), grid, block, lds, s, g, sg, p, odd);
    else hipLaunchKernelGGL((k_spmv_sym<1, false>), grid, block, lds, s, g, sg, p, odd);
  }
}

void launch_pipe_cg_sym(const DeviceGraph& g, const SymGraph& sg, const CgParams& p, int seq, hipStream_t s, unsigned long long gseq, bool fold) {
  static bool attr_set = false;
  if (!attr_set) {
    const int cap = 160 * 1024 - 512;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_cg_sym<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_cg_sym<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_cg_sym<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_cg_sym<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_set = true;
  }
  const size_t lds = sym_lds_bytes(sg);
  const dim3 grid(sg.n_tiles), block(SYM_LANES);      // (n_tiles >= 1: sym_wanted keeps ranks without rows off the form — an empty grid is a launch error, not a no-op)
#define PGO_PIPE_SYM(PK) do { if (g.cluster == 2) hipLaunchKernelGGL((k_pipe_cg_sym<PK, 2>), grid, block, lds, s, g, sg, p, seq); \
                              else hipLaunchKernelGGL((k_pipe_cg_sym<PK, 1>), grid, block, lds, s, g, sg, p, seq); } while (0)
  if (g.blk_packed) PGO_PIPE_SYM(true); else PGO_PIPE_SYM(false);
#undef PGO_PIPE_SYM
  // several ranks: the rank's three sums go into the exchange buffer(s) behind every launch.  One rank: the next launch folds the
  // partial triples itself; only the "stop test only" launch at the end of a batch (k_pipe_cg's, one work-group) wants them folded
  if (g.world == 1 && !fold) return;
  if (g.world > 1 && g.bx[0]) return;      // (boundary exchange: the tiles stored their boundary rows into the segment themselves, the last one folded the sums)
  DeviceGraph gf = g;
  gf.n_wg = sg.n_tiles;         // the fold adds one entry per work-group of the producing launch
  if (g.world == 1) { const size_t off = (size_t)((seq & 1) ^ 1) * g.n_part; gf.part_rz += off; gf.part_q += off; gf.part_rr += off; }
  launch_pipe_fold(gf, seq, gseq, s);
}

void launch_sym_repack(const DeviceGraph& g, const SymGraph& sg, hipStream_t s, int diag_only) {
  if (diag_only) {
    const dim3 grid((g.N + 255) / 256), block(256);
    if (g.blk_packed) hipLaunchKernelGGL(k_sym_repack_diag<true>, grid, block, 0, s, g, sg);
    else hipLaunchKernelGGL(k_sym_repack_diag<false>, grid, block, 0, s, g, sg);
    return;
  }
  const dim3 grid((sg.n_slots + 255) / 256), block(256);
  if (g.blk_packed) hipLaunchKernelGGL(k_sym_repack<true>, grid, block, 0, s, g, sg);
  else hipLaunchKernelGGL(k_sym_repack<false>, grid, block, 0, s, g, sg);
}

}  // namespace pgo
