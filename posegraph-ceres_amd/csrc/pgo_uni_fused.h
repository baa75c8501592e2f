// pgo_uni_fused.h — the universal stream in its FUSED form (r05): ONE kernel symbol, one launch per operation, ONE launch per CG
// iteration.  Included by pgo_kernels.hip inside `namespace pgo { namespace {` (it uses that file's device functions: linearize_body,
// cluster_precond_wave, damping_pose, edge_cost, lm_device_decide, ...).
//
// Why: the r03 stream alternates a slot-shaped and a vector-shaped kernel, two dependent launches per CG iteration (13.2 us at
// BASELINE configs[1], 83 % of an LM iteration).  Standard CG needs both: p'q must be known everywhere before x, r move, r'z before
// p moves.  The pipelined recurrences (Ghysels & Vanroose 2014; the form the sharded path already runs, k_pipe_cg) have ONE global
// reduction per iteration and everything else of an iteration is local to a row once n = A m is known — so a work-group multiplies
// its rows, updates the eight vectors of its rows, applies its own Jacobi blocks and leaves three partial sums: one launch.  The
// next launch folds the partials itself (every work-group alike, fixed order: same bits everywhere), decides stop / alpha / beta
// and goes on.  An LM iteration with n CG iterations is
//     HEAD (accept-finish, damping, Jacobi blocks, r0 = b, u0 = M^-1 r0) | W0 (w0 = A u0, m0 = M^-1 w0) | CG x n |
//     the launch that finds the CG stopped multiplies A x for the step tail at once | TAIL (candidate cost, model change, DECISION) |
//     LIN (behind an accepted step)
// = n + 5 launches instead of 2 n + 4, none wasted, whatever n turns out to be: the host enqueues the one symbol and nothing else.
//
// State: CgState::Fused f[2], double-buffered by LAUNCH parity (a kernel argument, so every address that depends on it is known
// before the first load returns): launch L reads f[L & 1], the partial-sum rows L & 1 and the exchange buffer pipe_buf[L & 1] and
// writes the other ones.  No word is read and written by the same launch.  Who writes f[(L & 1) ^ 1]: HEAD / TAIL — the last
// work-group through the ticket (it has the decision); everything else — lane 0 of work-group 0 (every work-group derives the
// same stop decision from the same partial sums).
//
// Same iterates as Ceres' ConjugateGradientsSolver in exact arithmetic, same stop rules on the same quantities (Q-tolerance eta,
// iteration limits, breakdown tests); no periodic residual refresh.  Requests this form does not serve (a CG
// run to a relative residual, four-pose Jacobi blocks, rows fatter than a work-group) keep the two-kernel stream.

// y = block * x for one incidence slot (packed 27-entry or full 36-entry layout, pgo_kernels.h)
template <bool PACKED, int NPAIR>
__device__ __forceinline__ void slot_block_times(const double2 (&blk)[NPAIR], uint8_t side, const double (&x)[6], double (&y)[6]) {
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

constexpr int UNI_F_FOLD = 8;        // partial-sum entries a lane folds at the start of a CG launch: n_wg <= UNI_F_FOLD * block
// "Am I the last work-group of this launch to get here?" — two levels (a ticket costs ~12 ns on its one address and a launch has
// hundreds of work-groups): work-group w bumps the counter of its class w % 8 (g.flags[4 + class]); the last of a class bumps the
// top counter (g.flags[12]); the last of those is the last of all and resets the nine words.  One lane calls it, after it has
// written its work-group's results with device-scope (write-through) stores: no cache write-back, no fence — a __threadfence() per
// work-group (L2 write-back + L1 invalidate, ~3.5 us each and slower the more work-groups per CU issue one) made HEAD and TAIL
// 10 / 20 us SLOWER when all 740 work-groups of BASELINE configs[1] took part instead of 256.  The reader (the last work-group)
// loads those results with device-scope loads.
__device__ __forceinline__ bool uni_f_last_arrival(const DeviceGraph& g, int wg, int n_wg) {
  const int cls = wg & 7;
  const int in_class = (n_wg - cls + 7) >> 3;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the result stores have reached the L2 / fabric before the ticket is taken
  if (atomicAdd(&g.flags[4 + cls], 1) != in_class - 1) return false;
  const int classes = min(n_wg, 8);
  if (atomicAdd(&g.flags[12], 1) != classes - 1) return false;
#pragma unroll
  for (int k = 0; k < 9; ++k) g.flags[4 + k] = 0;
  return true;
}

// trace of the fused stream (DeviceGraph::oplog, pgo_solver_trace_*): UNI_F_TRACE_WORDS words per launch — [0] (tick at the top
// of work-group 0 << 3 | operation), [2 + s] the latest end tick among the work-groups with index % 64 == s (one atomic per
// work-group, 64 addresses: a dozen per address at BASELINE configs[1])
// (UNI_F_TRACE_WORDS: pgo_kernels.h)
__device__ __forceinline__ bool uni_f_traced(const DeviceGraph& g, int launch) {
  return g.oplog && g.oplog_indexed && 1 + UNI_F_TRACE_WORDS * ((long long)launch + 1) <= g.oplog_cap;
}
__device__ __forceinline__ void uni_f_trace_begin(const DeviceGraph& g, int launch, int what, long long t_top) {
  if (!g.oplog || blockIdx.x != 0 || threadIdx.x != 0) return;
  if (!g.oplog_indexed) {      // PGO_UNI_OPLOG: one appended entry per launch (tools/rocprof_summary.py buckets the dispatches of this one symbol with it)
    const long long i = g.oplog[0];
    if (i + 1 < g.oplog_cap) { g.oplog[1 + i] = (t_top << 3) | (long long)(what & 7); g.oplog[0] = i + 1; }
  } else if (uni_f_traced(g, launch)) {
    g.oplog[1 + UNI_F_TRACE_WORDS * (size_t)launch] = (t_top << 3) | (long long)(what & 7);
  }
}
__device__ __forceinline__ void uni_f_trace_end(const DeviceGraph& g, int launch) {
  if (threadIdx.x == 0 && uni_f_traced(g, launch))
    atomicMax(reinterpret_cast<unsigned long long*>(g.oplog + 1 + UNI_F_TRACE_WORDS * (size_t)launch + 2 + (blockIdx.x & 63)),
              (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

template <bool PACKED, int INFO, int CL>
__device__ __forceinline__ void uni_f_body(const DeviceGraph& g, const CgParams& prm, int launch, double min_diag, double max_diag,
                                           double* lds, double* scratch, int* is_last_p, int* publish) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  constexpr int DIM = 6 * CL;
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  const int rp = launch & 1, wp = rp ^ 1;
  const double* rd = g.pipe_buf[rp];
  double* wr = g.pipe_buf[wp];
  const int m = 6 * g.N;
  const long long t_top = g.oplog ? (long long)__builtin_amdgcn_s_memrealtime() : 0;     // (both log forms order their entries by it)
  // ---- requested before the state is known: what a CG launch needs (work-groups of this form hold exactly B slots: no look-up) ----
  const CgState::Fused st = g.cg->f[rp];
  const int s_begin = wg * B;
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  const int t = s_begin + tid;
  const int col = g.slot_col[t];
  const int row = g.slot_row[t];
  const uint8_t side = g.slot_side[t];
  double2 blk[NPAIR];
  {
    const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(t >> 6) * TILE_DOUBLES + (size_t)(t & 63) * 2);
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) blk[k] = bp[(size_t)k * 64];
  }
  // column gather of what a CG launch multiplies: m (once the CG has stopped it is x, fetched then: once per LM iteration)
  double2 gm[3] = {{0, 0}, {0, 0}, {0, 0}};
  if (col >= 0) {
    const double2* ms = reinterpret_cast<const double2*>(rd + 6 * (size_t)col);
#pragma unroll
    for (int k = 0; k < 3; ++k) gm[k] = ms[k];
  }
  // the previous launch's partial sums, one entry {(r,u), (w,u), x'(b + r), -} per work-group: every lane takes the entries
  // tid, tid + B, ... — all requested at once (clamped index, zero weight: no load sits behind a branch)
  // (only requested here: they are added up inside the CG branch, behind the product — a launch that turns out to do something else
  // must not wait for them, nor for the blocks in front of them)
  double2 e0[UNI_F_FOLD], e1[UNI_F_FOLD];
  {
    const double2* pf = reinterpret_cast<const double2*>(g.part_f + (size_t)rp * 4 * g.n_part);
#pragma unroll
    for (int k = 0; k < UNI_F_FOLD; ++k) {
      const int i = min(tid + k * B, g.n_wg - 1);
      e0[k] = pf[2 * (size_t)i]; e1[k] = pf[2 * (size_t)i + 1];
    }
  }
  const int nown = nrows * 6;
  int seg_rb = 0, seg_cnt = 0;
  if (tid < nown) { seg_rb = g.row_slot_begin[r0 + tid / 6]; seg_cnt = g.row_slot_cnt[r0 + tid / 6]; }
  // the first pass of the owned rows' operands
  const bool own0 = tid < nown;
  const size_t gi = 6 * (size_t)r0 + tid;
  double pr = 0, pu = 0, pw = 0, pz = 0, pq = 0, ps = 0, ppv = 0, px = 0, pb = 0, pm = 0;
  double2 mi[DIM / 2];
#pragma unroll
  for (int k = 0; k < DIM / 2; ++k) mi[k] = double2{0, 0};
  if (own0) {
    pr = g.cg_r[gi]; pu = g.cg_u[gi];
    pw = g.cg_w[gi]; pz = g.cg_z[gi]; pq = g.cg_qq[gi]; ps = g.cg_s[gi]; ppv = g.cg_p0[gi]; px = g.cg_x[gi]; pb = g.cg_b[gi];
    pm = rd[gi];
    const double2* Mi = reinterpret_cast<const double2*>(g.Minv + gi * DIM);
#pragma unroll
    for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
  }
  if (wg == 0 && tid == 0) __hip_atomic_store(&g.scal->slots_done, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int op = st.op;
  uni_f_trace_begin(g, launch, op < 0 ? 0 : op, t_top);
  // the decisions of earlier launches go to the host from work-group 0 of a launch that does not touch the LM state, once its own
  // work is done (k_uni_f below: the deciding work-group would have held up the end of its launch for the 4 us the pinned-memory
  // stores and their fence take; up here the fence would sit in front of every launch's work)
  *publish = st.mirror && op != F_HEAD && op != F_TAIL;
  if (op <= F_IDLE) {          // stopped (terminated, paused) or not opened yet: the state stands
    if (wg == 0 && tid == 0) { CgState::Fused n = st; n.mirror = 0; g.cg->f[wp] = n; }
    return;
  }

  if (op == F_LIN) {
    DeviceGraph gl = g;
    gl.pose_x = g.pose_c;          // the accepted candidate; the accept-finish part of the next HEAD copies it over
    linearize_any<INFO>(gl, lds);
    if (wg == 0 && tid == 0) { CgState::Fused n{}; n.op = F_HEAD; g.cg->f[wp] = n; }
    return;
  }

  const int nT = g.n_wg;       // every work-group takes part in the vector-shaped operations
  int& is_last = *is_last_p;
  if (op == F_HEAD) {
#define PGO_UNI_HEAD_BLOCK
#define PGO_UNI_HEAD_NEXT F_W0
#include "pgo_uni_head_tail.inc"
#undef PGO_UNI_HEAD_NEXT
#undef PGO_UNI_HEAD_BLOCK
    return;
  }
  if (op == F_TAIL) {
#define PGO_UNI_TAIL_BLOCK
#include "pgo_uni_head_tail.inc"
#undef PGO_UNI_TAIL_BLOCK
    return;
  }

  // ---- F_W0 (w0 = A u0) / F_CG (one iteration; or, once the CG has stopped, q = A x for the step tail) ----
  const bool w0 = op == F_W0;
  // the product over this work-group's slots first (n = A m): the partial sums arrive meanwhile
  double y[6] = {0, 0, 0, 0, 0, 0};
  if (col >= 0) {
    double x[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { x[2 * k] = gm[k].x; x[2 * k + 1] = gm[k].y; }
    slot_block_times<PACKED, NPAIR>(blk, side, x, y);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
  double f3[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < UNI_F_FOLD; ++k) {
    const double wgt = tid + k * B < g.n_wg ? 1.0 : 0.0;
    f3[0] += wgt * e0[k].x; f3[1] += wgt * e0[k].y; f3[2] += wgt * e1[k].x;
  }
  const long long t_mul = uni_f_traced(g, launch) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  block_sum<3>(f3, scratch);     // every work-group folds the same entries in the same order: same bits everywhere (its barriers
                                 // also publish the slot results)
  int stop = 0, status = 0;
  double alpha = 0.0, beta = 0.0, gamma = 0.0, Q1 = 0.0;
  const int cnt = st.cnt;
  if (!w0) {
    gamma = f3[0];
    const double delta = f3[1];
    Q1 = -f3[2];
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
  if (wg == 0 && tid == 0) {
    CgState::Fused n{};
    if (stop) { n.op = F_TAIL; g.cg->iters = cnt; g.cg->status = status; g.cg->done = 1; }
    else if (w0) { n.op = F_CG; n.cnt = 0; }
    else { n.op = F_CG; n.cnt = cnt + 1; n.gamma_prev = gamma; n.alpha_prev = alpha; n.q_prev = Q1; }
    g.cg->f[wp] = n;
  }
  if (stop) {
    // the CG has stopped (once per LM iteration): this launch multiplies q = A x for the step tail instead, and the diagonal lanes
    // write delta = -S x and the candidate Plus(x, delta) of their rows
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = 0.0;
    if (col >= 0) {
      const double2* xs = reinterpret_cast<const double2*>(g.cg_x + 6 * (size_t)col);
      double x[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double2 v = xs[k]; x[2 * k] = v.x; x[2 * k + 1] = v.y; }
      if (side == SIDE_DIAG) {
        const PoseRec P = load_pose(g.pose_x, row);
        const uint8_t cm = g.cmask[row];
        double d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const bool c = (i < 3) ? (cm & 1) : (cm & 2);
          d[i] = c ? 0.0 : -g.scale[6 * (size_t)row + i] * x[i];
          g.delta[6 * (size_t)row + i] = d[i];
        }
        V3 pc = P.p;
        Q4 qc = P.q;
        if (!(cm & 1)) pc = V3{P.p.x + d[0], P.p.y + d[1], P.p.z + d[2]};
        if (!(cm & 2)) qc = quat_plus(P.q, V3{d[3], d[4], d[5]});
        double2* o = reinterpret_cast<double2*>(g.pose_c + (size_t)POSE_STRIDE * row);
        o[0] = double2{pc.x, pc.y};
        o[1] = double2{pc.z, qc.x};
        o[2] = double2{qc.y, qc.z};
        o[3] = double2{qc.w, 0.0};
      }
      slot_block_times<PACKED, NPAIR>(blk, side, x, y);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
    __syncthreads();
  }
  const long long t_fold = uni_f_traced(g, launch) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  double* lds_w = lds + (size_t)SPMV_LDS_STRIDE * B;
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
    if (stop) { g.cg_q[gj] = sm; continue; }
    if (idx != tid) {
      pr = g.cg_r[gj]; pu = g.cg_u[gj];
      if (!w0) {
        pw = g.cg_w[gj]; pz = g.cg_z[gj]; pq = g.cg_qq[gj]; ps = g.cg_s[gj]; ppv = g.cg_p0[gj]; px = g.cg_x[gj]; pb = g.cg_b[gj];
        pm = rd[gj];
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
  if (stop) return;
  const long long t_rows = uni_f_traced(g, launch) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
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
    wr[gj] = mn;
  }
  block_sum<3>(acc, scratch);
  if (tid == 0) {
    double2* pf = reinterpret_cast<double2*>(g.part_f + ((size_t)wp * g.n_part + wg) * 4);
    pf[0] = double2{acc[0], acc[1]};
    pf[1] = double2{acc[2], 0.0};
    if (wg == 0 && uni_f_traced(g, launch)) {   // phase stamps of work-group 0 (ticks from its top)
      const long long t_end = (long long)__builtin_amdgcn_s_memrealtime();
      g.oplog[1 + UNI_F_TRACE_WORDS * (size_t)launch + 1] = ((t_mul - t_top) & 0xffff) | (((t_fold - t_top) & 0xffff) << 16) |
                                                            (((t_rows - t_top) & 0xffff) << 32) | (((t_end - t_top) & 0xffff) << 48);
    }
  }
}

template <bool PACKED, int INFO, int CL>
__global__ __launch_bounds__(256, 2) void k_uni_f(DeviceGraph g, CgParams prm, int launch, double min_diag, double max_diag) {
  extern __shared__ double lds[];  // NV_LIN * block (the linearisation); the CG uses (SPMV_LDS_STRIDE + 6) * block + 6 of it
  __shared__ double scratch[32];
  __shared__ int is_last;
  int publish = 0;
  uni_f_body<PACKED, INFO, CL>(g, prm, launch, min_diag, max_diag, lds, scratch, &is_last, &publish);
  if (publish && blockIdx.x == 0 && threadIdx.x == 0) lm_mirror(g);
  uni_f_trace_end(g, launch);
}
