// pgo_uni_resident.h — the universal stream in its RESIDENT form (r05): the whole truncated CG of an LM iteration is ONE launch whose
// work-groups keep their 6x6 blocks, their Jacobi blocks and the eight vectors of their rows in REGISTERS from the first product to the
// last, and meet at a grid barrier once per iteration.  Included by pgo_kernels.hip behind pgo_uni_fused.h (same namespace, same helpers).
//
// Why: in the fused stream (pgo_uni_fused.h) a CG iteration is one launch of ~10 us + 2.3 us of kernel boundary that re-requests 44 MB
// (blocks, vectors, Jacobi blocks) the previous launch already had — at BASELINE configs[1] the whole working set is 26 MB and a launch is
// a chain of dependent round trips, not a stream.  What has to cross work-groups per iteration is 480 KB: the vector m = M^-1 w every
// product gathers by column, and three partial sums per work-group.  So: load the blocks once, then per iteration
//     gather m (6 doubles per slot)  ->  n = A m (LDS row sums)  ->  the pipelined recurrences on the row lanes' registers  ->
//     m_new = M^-1 w from LDS  ->  publish m_new + {(r,u), (w,u), x'(b + r)}  ->  GRID BARRIER  ->  fold the partial sums, stop / alpha / beta.
// Same recurrences, same fold order, same stop rules as k_uni_f: same decisions and CG counts, costs to ~1e-9 of the fused stream's (the
// compiler contracts the two bodies' multiply-adds differently; tests/test_gpu_resident.py), and both are held to the oracle's pipelined CG.
//
// Cross-work-group data (m, the partial sums, x at the end) moves through write-through stores (8-byte device-scope atomics) and
// L1-bypassing loads (sc1: 8-byte atomics, or 16-byte raw buffer loads where a lane wants neighbours — the texture addresser works per
// request, 3 + 2 x 2 requests per lane and turn instead of 6 + 2 x 3 took 0.5 us off a turn), so no cache write-back or invalidate is
// needed around the barrier (a release + acquire fence pair is 3.4 us; MI355X_MICROARCH.md, hand-off forms).  The barrier is one level of
// arrival counters (RES_NCLS classes) whose last arrival publishes the class's generation word; lanes 0..RES_NCLS-1 of every work-group
// poll one class word each with relaxed loads and leave together (2.6 us per turn as work-group 0 sees it, which includes waiting for the
// slowest work-group).  Measured and dropped: a second counter level + one generation word per class written by the last arrival of all
// (3.1 us); no counters at all — every entry of the partial sums tagged with the turn's number and the fold polling the tags (392
// work-groups polling 392 entries each: 5.7 us per turn against 2.6 + 1.9).  What a turn costs is round trips to L2, so they are merged:
// the operand of the next product is requested together with the fold's partial sums, and the acknowledgements of the m store and of
// the partial-sum store are waited for once.  A work-group that waits longer than ~2 s (a grid that is not fully resident — two
// such launches sharing the device — would wait forever) sets the abort word: everybody leaves, the LM iteration goes back to its HEAD,
// the host is told (LmScalars::resident_abort) and carries on with the fused stream's launches instead of hanging the device.  The host admits ONE resident session per device at a time
// (pgo_lm.cpp) and only grids that fit the chip at two waves per SIMD.
//
// The stream (r06): TWO kernel symbols in a fixed cycle, launch L plays role L % 2 —
//     k_res_lh (LIN behind an accepted step + HEAD, every work-group on its own rows) | k_res_cg (the whole CG + the step tail + the decision)
// — each acting only if the state word says its operation is next, state double-buffered by launch parity exactly as in the fused
// stream; the host enqueues whole cycles ahead of the device's launch counter.  (r05 ran four launches per LM iteration: HEAD | CG |
// TAIL | LIN.  Measured on one box, C2, 20 / 25 steps: 0.1872 / 0.1676 ms per LM iteration with four, 0.1852 / 0.1663 with the tail
// inside the CG launch, 0.1874 / 0.1666 with LIN + HEAD in one launch as well — the boundaries saved are paid back by Jacobi blocks
// inverted on all 392 work-groups, two per compute unit on half the chip, instead of on 250: EXPERIMENTS.md r06.)

// barrier words in g.flags (zeroed whenever the device state is uploaded), each on a 128-byte line of its own (32 ints): arrivals of
// class c (work-group index % RES_NCLS) at RES_CLS + 32 c, generation = barriers class c has completed at RES_GEN + 32 c, abort at
// RES_ABORT.  Lines of their own: with a polled word next to the counters, 392 polling lanes kept the arrival atomics of the slower
// work-groups queued behind their loads — 15 us per barrier.  The counters only ever grow (barrier number b is complete for a class
// when it has seen in_class * b arrivals): nothing is reset between two barriers.
constexpr int RES_NCLS = 32;                          // arrival classes (work-group index % RES_NCLS): ~12 arrivals queue on a counter at C2, not 49
constexpr int RES_CLS = 64, RES_GEN = RES_CLS + 32 * RES_NCLS, RES_ABORT = RES_GEN + 32 * RES_NCLS, RES_FLAG_WORDS = RES_ABORT + 32;

// Lanes 0..RES_NCLS-1 of a work-group call it together (after __syncthreads()); returns false if the barrier was aborted.  `gen` = the
// generation this work-group has seen complete; every arrival targets gen + 1.  Lane 0 arrives at its class; the last arrival of a
// class publishes the class's generation word; lane c polls the word of class c and they leave together when all classes have moved.
__device__ __forceinline__ bool res_grid_barrier(const DeviceGraph& g, int wg, int n_wg, int& gen, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this work-group's write-through stores have left
  const int target = gen + 1;
  if (lane == 0) {
    const int cls = wg % RES_NCLS;
    const int in_class = (n_wg - cls + RES_NCLS - 1) / RES_NCLS;
    if (atomicAdd(&g.flags[RES_CLS + 32 * cls], 1) + 1 == in_class * target)
      __hip_atomic_store(&g.flags[RES_GEN + 32 * cls], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int* word = &g.flags[RES_GEN + 32 * min(lane, min(n_wg, RES_NCLS) - 1)];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  while (__ballot(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0)) {
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 0x3ff) == 0) {
      if (__hip_atomic_load(&g.flags[RES_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(&g.flags[RES_ABORT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
  }
  gen = target;
  return true;
}

// 16 bytes per request, past the CU's L1 like res_ld (raw buffer load with sc1): what was written before a grid barrier and is read
// behind it needs no single-copy atomicity, and the texture addresser works per request — the gather of a slot's six entries of m is
// three requests instead of six, a partial-sum entry two instead of three.
typedef unsigned int res_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t res_buf(const double* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ double2 res_ld2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ double res_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void res_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The abort word, read at the top of every launch of the cycle: once a grid barrier has given up, the kernels of the cycle only hand the
// state on — the work-group that gave up has put the LM iteration back to its HEAD and told the host (LmScalars::resident_abort), which
// carries on with the fused stream's launches (k_uni_f does whatever operation the state names, whatever the launch number).
__device__ __forceinline__ bool res_aborted(const DeviceGraph& g) {
  return __hip_atomic_load(&g.flags[RES_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// what every kernel of the cycle does when its operation is not the next one (or the stream is stopped): hand the state on
// (The launches keep only the two words of the state they branch on, `st_op` and `st_mirror`, in registers, and state records are
// written word by word: a record held as a local made the compiler park it in LDS, whose addressing reads the work-group size from
// the dispatch packet — host memory, 10-25 us per launch.)
__device__ __forceinline__ void res_put_state(const DeviceGraph& g, int wp, int op, int cnt, double gamma_prev, double alpha_prev, double q_prev) {
  CgState::Fused* d = &g.cg->f[wp];
  d->op = op; d->cnt = cnt; d->mirror = 0; d->pad = 0;
  d->gamma_prev = gamma_prev; d->alpha_prev = alpha_prev; d->q_prev = q_prev;
}
__device__ __forceinline__ void res_pass_on(const DeviceGraph& g, int rp, int wp) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const CgState::Fused* s = &g.cg->f[rp];
    if (s->mirror) lm_mirror(g);          // (nothing touches the LM state in an idle launch)
    res_put_state(g, wp, s->op, s->cnt, s->gamma_prev, s->alpha_prev, s->q_prev);
  }
}
// A launch of the cycle that finds the abort word set (or the CG launch whose barrier just gave up): the state is handed on — an LM
// iteration that was in its CG goes back to its HEAD (idempotent: same damping, same CG start; the fused stream's CG wants its own
// first product) — and the host is told.  One lane calls it.
__device__ __forceinline__ void res_when_aborted(const DeviceGraph& g, int rp, int wp) {
  const CgState::Fused* s = &g.cg->f[rp];
  if (s->mirror) lm_mirror(g);
  if (s->op == F_CG) res_put_state(g, wp, F_HEAD, 0, 0.0, 0.0, 0.0);
  else res_put_state(g, wp, s->op, s->cnt, s->gamma_prev, s->alpha_prev, s->q_prev);
  __hip_atomic_store(&g.scal->resident_abort, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- r06, role 0 of the TWO-launch cycle: LIN (behind an accepted step) and HEAD in one launch, every work-group on ITS OWN rows ----
// HEAD as a launch of its own (r05: the fused stream's text, pgo_uni_head_tail.inc) handed Jacobi blocks out in chunks of 40 poses to the first 250
// work-groups, whatever rows those own.  Everything HEAD reads of a row, though — its diagonal block and gradient, the blocks between
// the two poses of its cluster, its scales — is written by the work-group that OWNS the row when it linearises, and a work-group holds
// whole clusters (prepare(): pairs_whole).  So the work-group that has just linearised its rows goes straight on: candidate -> current
// point and gradient norm of its rows, damping, its clusters' 12 x 12 inverses in registers (cluster_precond_wave), r0 = b, u0 = M^-1 b
// into the exchange buffer — no kernel boundary, no idle LIN slot behind a rejected step (the launch then runs HEAD alone), 392 waves'
// worth of Gauss-Jordan spread over all work-groups instead of 250.  Only the last work-group to arrive does the global part (gradient
// max-norm, the opening tests of the pass, the next operation), as in HEAD.
template <int INFO, int CL, bool LEAN>
__global__ __launch_bounds__(256, 2) void k_res_lh(DeviceGraph g, int launch, double min_diag, double max_diag) {
  extern __shared__ double lds[];  // the linearisation's row sums (LEAN_NV * block / 2 or NV_LIN * block doubles) / HEAD: block doubles
  __shared__ double scratch[32];
  __shared__ int is_last_s;
  constexpr int DIM = 6 * CL;
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  const int nT = g.n_wg;
  const int rp = launch & 1, wp = rp ^ 1;
  double* wr = g.pipe_buf[wp];
  const long long t_top = g.oplog ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  const int st_op = g.cg->f[rp].op;
  if (wg == 0 && tid == 0) __hip_atomic_store(&g.scal->slots_done, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool aborted = res_aborted(g);
  const bool mine = (st_op == F_LIN || st_op == F_HEAD) && !aborted;
  uni_f_trace_begin(g, launch, mine ? st_op : 0, t_top);
  if (!mine) {
    if (!aborted) res_pass_on(g, rp, wp);
    else if (wg == 0 && tid == 0) res_when_aborted(g, rp, wp);
    uni_f_trace_end(g, launch);
    return;
  }
  if (st_op == F_LIN) {
    DeviceGraph gl = g;
    gl.pose_x = g.pose_c;          // the accepted candidate; the accept-finish part below copies it over
    if constexpr (LEAN) lean_linearize_body<INFO, true>(gl, lds);
    else linearize_body<INFO>(gl, lds);
    __syncthreads();               // this work-group's diagonal blocks, gradient and blocks are in place for its other waves
  }
  const long long t_lin = uni_f_traced(g, launch) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  LmDev& D = *g.lm;
  const int accepted = D.accepted, pause = D.pause;
  const double radius = D.core.radius;
  const int mode = D.core.reuse_diagonal ? 1 : 0;
  const int r0 = g.wg_row_begin[wg], nrows = g.wg_row_begin[wg + 1] - r0;
  // ---- accept-finish of the work-group's rows ----
  double gmx = 0.0;
  if (accepted && tid < nrows) {
    const int v = r0 + tid;
    const PoseRec P = load_pose(g.pose_c, v);
    const double2* src = reinterpret_cast<const double2*>(g.pose_c + (size_t)POSE_STRIDE * v);
    double2* dst = reinterpret_cast<double2*>(g.pose_x + (size_t)POSE_STRIDE * v);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    const uint8_t cm = g.cmask[v];
    const double* gr = g.grad + 6 * (size_t)v;
    if (!(cm & 1)) gmx = fmax(gmx, fmax(fabs(gr[0]), fmax(fabs(gr[1]), fabs(gr[2]))));
    if (!(cm & 2)) {
      const Q4 q = quat_plus(P.q, V3{-gr[3], -gr[4], -gr[5]});
      gmx = fmax(gmx, fmax(fmax(fabs(P.q.x - q.x), fabs(P.q.y - q.y)), fmax(fabs(P.q.z - q.z), fabs(P.q.w - q.w))));
    }
  }
  gmx = wave_max(gmx);
  if ((tid & 63) == 0) scratch[tid >> 6] = gmx;
  __syncthreads();
  if (tid == 0 && accepted) {
    double tm = 0.0;
    for (int w = 0; w < (B + 63) / 64; ++w) tm = fmax(tm, scratch[w]);
    __hip_atomic_store(&g.part_misc[4 * (size_t)g.n_part + wg], tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // ---- damping, Jacobi blocks, CG start of the work-group's rows ----
  if (!pause) {
    if constexpr (CL == 1) {
      if (tid < nrows) damping_pose(g, r0 + tid, radius, min_diag, max_diag, mode);
      __syncthreads();      // the inverses were written by other lanes of this work-group
      const int ridx = 6 * r0 + tid;
      const bool rlive = tid < 6 * nrows;
      double b = 0.0;
      if (rlive) {
        b = g.scale[ridx] * g.grad[ridx];
        g.cg_b[ridx] = b;
        g.cg_x[ridx] = 0.0;
        g.cg_r[ridx] = b;
      }
      lds[tid] = b;
      __syncthreads();
      if (rlive) {
        const double* Mi = g.Minv + (size_t)ridx * DIM;
        const double* rv = lds + DIM * (tid / DIM);
        double u = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) u += Mi[k] * rv[k];
        g.cg_u[ridx] = u;
        wr[ridx] = u;
      }
    } else {                // a wave inverts Jacobi blocks in registers and starts the CG of their rows from there (no barrier)
      constexpr int CPW = 64 / DIM;
      const int wave = tid >> 6, lane = tid & 63, nw = B >> 6;
      const int c_first = r0 / CL, c_end = (r0 + nrows + CL - 1) / CL;       // (r0 is a multiple of CL: a work-group holds whole clusters)
      for (int c0 = c_first + wave * CPW; c0 < c_end; c0 += nw * CPW)
        cluster_precond_wave<CL, true>(g, radius, min_diag, max_diag, mode, c0, min(c_end, c0 + CPW), lane, wr);
    }
  }
  // ---- the last work-group to finish: gradient norm of the accepted point, the opening tests of the next pass, the next operation ----
  if (tid == 0) is_last_s = uni_f_last_arrival(g, wg, nT);
  __syncthreads();
  if (is_last_s) {
    double mm = 0.0;
    if (accepted)
      for (int i = tid; i < nT; i += B)
        mm = fmax(mm, __hip_atomic_load(&g.part_misc[4 * (size_t)g.n_part + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    mm = wave_max(mm);
    if ((tid & 63) == 0) scratch[tid >> 6] = mm;
    __syncthreads();
    if (tid == 0) {
      const long long now = (long long)__builtin_amdgcn_s_memrealtime();
      if (accepted) {
        double tm = 0.0;
        for (int w = 0; w < (B + 63) / 64; ++w) tm = fmax(tm, scratch[w]);
        D.core.gmax = tm;
        g.scal->ring[D.core.iteration % LM_RING].gradient_max_norm = tm;
        g.scal->gradient_max = tm;
        D.accepted = 0;
        lm_pre_step_checks(D, true);
        D.ticks_jacobian += now - D.t_mark;
        D.t_mark = now;
      }
      int next_op = F_CG;
      if (D.halt) next_op = F_EXIT;
      else if (pause) { D.halt = LM_HALT_BUDGET; next_op = F_EXIT; }
      CgState::Fused* d = &g.cg->f[wp];
      d->op = next_op; d->cnt = 0; d->pad = 0;
      d->mirror = 1;         // the next launch publishes the state to the host (lane 0 of its work-group 0, beside its work)
      d->gamma_prev = 0.0; d->alpha_prev = 0.0; d->q_prev = 0.0;
      g.cg->done = 0; g.cg->iters = 0; g.cg->status = 0;
      if (uni_f_traced(g, launch)) {   // phase stamps of the LAST work-group to arrive: its linearisation done / end
        const long long t_end = (long long)__builtin_amdgcn_s_memrealtime();
        g.oplog[1 + UNI_F_TRACE_WORDS * (size_t)launch + 1] = ((t_lin - t_top) & 0xffff) | (((t_end - t_top) & 0xffff) << 48);
      }
    }
  }
  uni_f_trace_end(g, launch);
}

// ---- role 1: the whole CG ----
template <bool PACKED, int CL, int INFO>
__global__ __launch_bounds__(256, 2) void k_res_cg(DeviceGraph g, CgParams prm, int launch) {
  constexpr int NPAIR = PACKED ? BLK_PAIRS_PACKED : BLK_PAIRS_FULL;
  constexpr int DIM = 6 * CL;
  extern __shared__ double lds[];  // SPMV_LDS_STRIDE * block (slot results) + 6 * block + 6 (w of the owned rows)
  __shared__ double scratch[32];
  __shared__ int sh_ok;
  const int B = blockDim.x, tid = threadIdx.x, wg = blockIdx.x;
  const int rp = launch & 1, wp = rp ^ 1;
  const long long t_top = g.oplog ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  const int st_op = g.cg->f[rp].op, st_mirror = g.cg->f[rp].mirror;
  if (wg == 0 && tid == 0) __hip_atomic_store(&g.scal->slots_done, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // ---- what stays in registers for the whole CG: the slot's block, the row lane's vectors and its Jacobi-block row.  The slot's words
  // and its block are asked for BEFORE the state word is looked at (an idle launch of this role — a stopped stream — reads them in vain) ----
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
  const bool aborted = res_aborted(g);
  const bool mine = st_op == F_CG && !aborted;
  uni_f_trace_begin(g, launch, mine ? F_CG : 0, t_top);
  if (!mine) {
    if (!aborted) res_pass_on(g, rp, wp);
    else if (wg == 0 && tid == 0) res_when_aborted(g, rp, wp);
    uni_f_trace_end(g, launch);
    return;
  }
  const int nown = nrows * 6;                        // <= B (res_supported)
  const bool own = tid < nown;
  const size_t gi = 6 * (size_t)r0 + tid;
  int sb = 0, sE = 0;
  double vr = 0, vu = 0, vw = 0, vz = 0, vq = 0, vs = 0, vp = 0, vx = 0, vb = 0;
  double2 mi[DIM / 2];
#pragma unroll
  for (int k = 0; k < DIM / 2; ++k) mi[k] = double2{0, 0};
  if (own) {
    sb = g.row_slot_begin[r0 + tid / 6] - s_begin;
    sE = sb + g.row_slot_cnt[r0 + tid / 6];
    vb = g.cg_b[gi]; vr = vb; vu = g.cg_u[gi];
    const double2* Mi = reinterpret_cast<const double2*>(g.Minv + gi * DIM);
#pragma unroll
    for (int k = 0; k < DIM / 2; ++k) mi[k] = Mi[k];
  }
  const int kc = tid % 6;
  double* lds_w = lds + (size_t)SPMV_LDS_STRIDE * B;
  const double* const pbuf0 = g.pipe_buf[0];        // (both in registers: an index that changes per turn would re-load the pointer from the kernel arguments — a round trip in front of every gather)
  const double* const pbuf1 = g.pipe_buf[1];
  int cur = rp;                                      // exchange buffer / partial-sum row this iteration READS (HEAD wrote u0 into pipe_buf[rp])
  int gen = 0;
  if (tid < RES_NCLS) gen = __hip_atomic_load(&g.flags[RES_GEN + 32 * (wg % RES_NCLS)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (nobody moves it before everybody has arrived once)
  int cnt = 0, stop = 0, status = 0;
  double alpha = 0.0, beta = 0.0, gamma_prev = 0.0, alpha_prev = 0.0, q_prev = 0.0;
  bool ok = true;
  const bool traced = wg == 0 && tid == 0 && uni_f_traced(g, launch);
  long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;      // (trace: ticks work-group 0 spent up to the publish / in the barrier / in the fold, summed over the iterations)
  // The operand of a turn's product (the slot's six entries of m, and the row lane's own entry) is asked for as soon as the
  // barrier of the turn before has been passed, TOGETHER with the partial sums of the fold: one round trip to L2, not two.
  double x[6] = {0, 0, 0, 0, 0, 0};
  double mine_m = 0.0;
  const unsigned col_off = 48u * (unsigned)max(col, 0);
  if (col >= 0) {
    const __amdgpu_buffer_rsrc_t mb = res_buf(cur ? pbuf1 : pbuf0);
    const double2 a = res_ld2(mb, col_off), b = res_ld2(mb, col_off + 16), c = res_ld2(mb, col_off + 32);
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; x[4] = c.x; x[5] = c.y;
  }
  // A turn of the loop: [fold of the turn before, stop test / alpha / beta] + product -> LDS row sums -> recurrences -> Jacobi block ->
  // publish -> grid barrier -> requests.  The two block-wide sums of a turn (the fold's and the row lanes' partial sums) share the
  // barriers the product and the Jacobi block need anyway (wave sums into `scratch`, summed in wave order behind the barrier:
  // block_sum's arithmetic exactly, four __syncthreads per turn instead of eight).
  double f3[3] = {0.0, 0.0, 0.0};
  const int lane = tid & 63, wave = tid >> 6, nw = (B + 63) >> 6;
  for (int it = 0;; ++it) {
    const long long tp0 = traced ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
    if (traced && it > 0) ph2 += tp0 - ph3;
    if (it > 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double sw = wave_sum(f3[k]);
        if (lane == 0) scratch[wave * 3 + k] = sw;
      }
    }
    // ---- n = A m over this work-group's slots (it == 0: w0 = A u0) ----
    double y[6] = {0, 0, 0, 0, 0, 0};
    if (col >= 0) slot_block_times<PACKED, NPAIR>(blk, side, x, y);
#pragma unroll
    for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
    __syncthreads();
    if (it > 0) {
      // ---- the fold of the turn before (every work-group alike, the fused stream's order), its stop test / alpha / beta ----
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double sw = 0.0;
        for (int w = 0; w < nw; ++w) sw += scratch[w * 3 + k];
        f3[k] = sw;
      }
      const double gamma = f3[0], delta = f3[1], Q1 = -f3[2];
      if (cnt > 0) {
        const double zeta = cnt * (Q1 - q_prev) / Q1;
        if (zeta < prm.q_tolerance && cnt >= prm.min_iterations) stop = 1;
        if (cnt >= prm.max_iterations) stop = 1;
      }
      if (!stop && (gamma == 0.0 || !isfinite(gamma))) { stop = 1; status = (gamma == 0.0) ? 0 : 2; }
      if (!stop && cnt > 0) {
        beta = gamma / gamma_prev;
        if (beta == 0.0 || !isfinite(beta)) { stop = 1; status = 2; }
      }
      if (!stop) {
        const double den = cnt > 0 ? delta - beta * gamma / alpha_prev : delta;
        if (!(den > 0.0) || !isfinite(den)) { stop = 1; status = 1; }     // "matrix is indefinite": x of the previous iteration stands
        else alpha = gamma / den;
      }
      if (stop) break;           // (the product above was in vain; x as published in front of the last barrier is final)
      gamma_prev = gamma; alpha_prev = alpha; q_prev = Q1;
      cnt = it;                  // the update this turn applies is iteration `cnt`
    }
    double acc[3] = {0.0, 0.0, 0.0};
    if (own) {
      double s0 = 0.0, s1 = 0.0;
      int j = sb;
      for (; j + 1 < sE; j += 2) { s0 += lds[j * SPMV_LDS_STRIDE + kc]; s1 += lds[(j + 1) * SPMV_LDS_STRIDE + kc]; }
      if (j < sE) s0 += lds[j * SPMV_LDS_STRIDE + kc];
      const double sm = s0 + s1;
      if (it == 0) {
        vw = sm;
      } else {
        const double zn = sm + beta * vz, qn = mine_m + beta * vq, sn = vw + beta * vs, pn = vu + beta * vp;
        vx = vx + alpha * pn;
        vr = vr - alpha * sn;
        vu = vu - alpha * qn;
        vw = vw - alpha * zn;
        vz = zn; vq = qn; vs = sn; vp = pn;
        acc[2] = vx * (vb + vr);
      }
      acc[0] = vr * vu;
      acc[1] = vw * vu;
      lds_w[tid] = vw;
    }
    if (tid < 6) lds_w[nown + tid] = 0.0;       // the missing half of a last odd pair
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double sw = wave_sum(acc[k]);
      if (lane == 0) scratch[12 + wave * 3 + k] = sw;
    }
    __syncthreads();
    if (own) {
      const double* wv = lds_w + DIM * (tid / DIM);
      double mn = 0.0;
#pragma unroll
      for (int k = 0; k < DIM / 2; ++k) mn += mi[k].x * wv[2 * k] + mi[k].y * wv[2 * k + 1];
      res_st(const_cast<double*>(cur ? pbuf0 : pbuf1) + gi, mn);
      res_st(g.cg_x + gi, vx);        // x of this turn too: if the fold behind the barrier says "stop", it is already everybody's to read
    }
    if (tid == 0) {
      double* pf = g.part_f + ((size_t)(cur ^ 1) * g.n_part + wg) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double sw = 0.0;
        for (int w = 0; w < nw; ++w) sw += scratch[12 + w * 3 + k];
        res_st(pf + k, sw);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY lane: its store of m (and lane 0's partial sums) has left — ONE wait for both, the acknowledgements overlap —
    __syncthreads();                                       // ... before the arrival below says so
    if (tid < RES_NCLS) {
      const long long tp1 = traced ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
      const bool passed = res_grid_barrier(g, wg, g.n_wg, gen, tid);
      if (tid == 0) sh_ok = passed ? 1 : 0;
      if (traced) { const long long tp2 = (long long)__builtin_amdgcn_s_memrealtime(); ph0 += tp1 - tp0; ph1 += tp2 - tp1; ph3 = tp2; }
    }
    __syncthreads();
    if (!sh_ok) { ok = false; break; }
    // ---- behind the barrier: everything the next turn reads from other work-groups, in one round trip ----
    cur ^= 1;
    if (own) mine_m = res_ld((cur ? pbuf1 : pbuf0) + gi);
    if (col >= 0) {
      const __amdgpu_buffer_rsrc_t mb = res_buf(cur ? pbuf1 : pbuf0);
      const double2 a = res_ld2(mb, col_off), b = res_ld2(mb, col_off + 16), c = res_ld2(mb, col_off + 32);
      x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; x[4] = c.x; x[5] = c.y;
    }
    f3[0] = f3[1] = f3[2] = 0.0;
    {
      const __amdgpu_buffer_rsrc_t pf = res_buf(g.part_f + (size_t)cur * 4 * g.n_part);
      for (int k = 0; k * B < g.n_wg; ++k) {          // (the fused stream's order: entry tid + k B in turn k; it adds 0 x entry for the lanes past the end)
        const int i = min(tid + k * B, g.n_wg - 1);
        const double wgt = tid + k * B < g.n_wg ? 1.0 : 0.0;
        const double2 a01 = res_ld2(pf, 32u * (unsigned)i), a2 = res_ld2(pf, 32u * (unsigned)i + 16);
        f3[0] += wgt * a01.x; f3[1] += wgt * a01.y; f3[2] += wgt * a2.x;
      }
    }
  }
  if (!ok) {          // a barrier gave up (every work-group leaves the same way, sooner or later: the abort word)
    if (wg == 0 && tid == 0) res_when_aborted(g, rp, wp);
    uni_f_trace_end(g, launch);
    return;
  }
  // ---- the CG has stopped after `cnt` iterations (every work-group alike, behind the same barrier: the x every row lane stored in front
  // of it is the final one).  r06: the STEP TAIL runs here, in the same launch and without another grid barrier — q = A x from the blocks
  // still in registers; the model change from the row lanes' own x, b, q; the candidate of a pose recomputed by every lane that needs it
  // (Plus(x_v, -S_v x_v) from the gathered x of the slot's column and the work-group's own rows: the formula of the DIAG lane that writes
  // pose_c); the candidate cost of an edge by its BEGIN-side slot lane from the slot-order measurement / information arrays (edge_cost's
  // arithmetic per edge, summed per work-group); four partial sums per work-group, and the last work-group to arrive folds and DECIDES
  // (the TAIL operation of pgo_uni_head_tail.inc from its ticket on).  What the TAIL launch of r05 cost — a kernel boundary, its state and
  // argument round trips, two strided gather loops — is one round trip of loads here. ----
  if (wg == 0 && tid == 0) {
    if (st_mirror) lm_mirror(g);
    if (traced) g.oplog[1 + UNI_F_TRACE_WORDS * (size_t)launch + 1] = (ph0 & 0xffff) | ((ph1 & 0xffff) << 16) | ((ph2 & 0xffff) << 32) | ((long long)(cnt & 0xffff) << 48);
  }
  const size_t ns = (size_t)g.n_slots;
  double y[6] = {0, 0, 0, 0, 0, 0};
  double acc4[4] = {0.0, 0.0, 0.0, 0.0};   // candidate cost, model change, |step|^2, |x|^2
  // everything the tail reads, requested together: x of the column (written by other work-groups: past the L1), the static records of
  // the slot's two poses, the slot's measurement and information, the row lane's damping entry
  const bool is_begin = col >= 0 && side == SIDE_BEGIN, is_diag = col >= 0 && side == SIDE_DIAG;
  PoseRec Pc{}, Pr{};
  double sc_c[6] = {0, 0, 0, 0, 0, 0}, sc_r[6] = {0, 0, 0, 0, 0, 0};
  uint8_t cm_c = 0, cm_r = 0;
  double ms[7] = {0, 0, 0, 0, 0, 0, 1};
  WBlocks Wt{};
  if (col >= 0) {
    const __amdgpu_buffer_rsrc_t xb = res_buf(g.cg_x);
    const double2 a = res_ld2(xb, col_off), b = res_ld2(xb, col_off + 16), c = res_ld2(xb, col_off + 32);
    x[0] = a.x; x[1] = a.y; x[2] = b.x; x[3] = b.y; x[4] = c.x; x[5] = c.y;
  }
  if (is_begin || is_diag) {
    Pc = load_pose(g.pose_x, col);
    cm_c = g.cmask[col];
#pragma unroll
    for (int i = 0; i < 6; ++i) sc_c[i] = g.scale[6 * (size_t)col + i];
  }
  if (is_begin) {
    Pr = load_pose(g.pose_x, row);
    cm_r = g.cmask[row];
#pragma unroll
    for (int i = 0; i < 6; ++i) sc_r[i] = g.scale[6 * (size_t)row + i];
#pragma unroll
    for (int i = 0; i < 7; ++i) ms[i] = g.smeas[(size_t)i * ns + t];
    if constexpr (INFO == 3) Wt = load_W_diag(g.sW, ns, (size_t)t);
    else if constexpr (INFO == 2) Wt = load_W_blockdiag(g.sW, ns, (size_t)t);
    else if constexpr (INFO == 1) Wt = load_W(g.sW, ns, (size_t)t);
  }
  double d2v = 0.0;
  uint8_t cm_own = 0;
  if (own) { d2v = g.d2[gi]; cm_own = g.cmask[r0 + tid / 6]; lds_w[tid] = vx; }      // (x of the work-group's rows for its BEGIN lanes)
  __syncthreads();
  // Plus(pose, -S x) of a pose: the candidate (the DIAG lane's statements)
  auto candidate = [](const PoseRec& P, uint8_t cm, const double (&sc)[6], const double (&xv)[6], double (&d)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bool c = (i < 3) ? (cm & 1) : (cm & 2);
      d[i] = c ? 0.0 : -sc[i] * xv[i];
    }
    PoseRec C = P;
    if (!(cm & 1)) C.p = V3{P.p.x + d[0], P.p.y + d[1], P.p.z + d[2]};
    if (!(cm & 2)) C.q = quat_plus(P.q, V3{d[3], d[4], d[5]});
    return C;
  };
  if (col >= 0) {
    if (is_diag) {
      double d[6];
      const PoseRec C = candidate(Pc, cm_c, sc_c, x, d);
#pragma unroll
      for (int i = 0; i < 6; ++i) g.delta[6 * (size_t)row + i] = d[i];
      double2* o = reinterpret_cast<double2*>(g.pose_c + (size_t)POSE_STRIDE * row);
      o[0] = double2{C.p.x, C.p.y};
      o[1] = double2{C.p.z, C.q.x};
      o[2] = double2{C.q.y, C.q.z};
      o[3] = double2{C.q.w, 0.0};
      if (!(cm_c & 1)) {
        const double dx = Pc.p.x - C.p.x, dy = Pc.p.y - C.p.y, dz = Pc.p.z - C.p.z;
        acc4[2] += dx * dx + dy * dy + dz * dz;
        acc4[3] += Pc.p.x * Pc.p.x + Pc.p.y * Pc.p.y + Pc.p.z * Pc.p.z;
      }
      if (!(cm_c & 2)) {
        const double dx = Pc.q.x - C.q.x, dy = Pc.q.y - C.q.y, dz = Pc.q.z - C.q.z, dw = Pc.q.w - C.q.w;
        acc4[2] += dx * dx + dy * dy + dz * dz + dw * dw;
        acc4[3] += Pc.q.x * Pc.q.x + Pc.q.y * Pc.q.y + Pc.q.z * Pc.q.z + Pc.q.w * Pc.q.w;
      }
    } else if (is_begin) {
      double xr[6], dr[6], dc[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) xr[i] = lds_w[6 * (row - r0) + i];
      const PoseRec CA = candidate(Pr, cm_r, sc_r, xr, dr), CB = candidate(Pc, cm_c, sc_c, x, dc);
      double er[6];
      edge_error(CA.p, CA.q, CB.p, CB.q, V3{ms[0], ms[1], ms[2]}, Q4{ms[3], ms[4], ms[5], ms[6]}, er);
      double sq;
      if constexpr (INFO != 0) {       // (edge_cost's statements)
        const V3 ep{er[0], er[1], er[2]}, eq{er[3], er[4], er[5]};
        const V3 a1 = mulv(Wt.pp, ep), a2 = mulv(Wt.pr, eq), b1 = mulTv(Wt.pr, ep), b2 = mulv(Wt.rr, eq);
        sq = dot(ep, V3{a1.x + a2.x, a1.y + a2.y, a1.z + a2.z}) + dot(eq, V3{b1.x + b2.x, b1.y + b2.y, b1.z + b2.z});
      } else {
        sq = er[0] * er[0] + er[1] * er[1] + er[2] * er[2] + er[3] * er[3] + er[4] * er[4] + er[5] * er[5];
      }
      double rho0, rho1;
      loss_eval(g.loss_kind, g.loss_a, sq, &rho0, &rho1);
      acc4[0] = 0.5 * rho0;
    }
    slot_block_times<PACKED, NPAIR>(blk, side, x, y);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) lds[tid * SPMV_LDS_STRIDE + k] = y[k];
  __syncthreads();
  if (own) {
    double s0 = 0.0, s1 = 0.0;
    int j = sb;
    for (; j + 1 < sE; j += 2) { s0 += lds[j * SPMV_LDS_STRIDE + kc]; s1 += lds[(j + 1) * SPMV_LDS_STRIDE + kc]; }
    if (j < sE) s0 += lds[j * SPMV_LDS_STRIDE + kc];
    const double qv = s0 + s1;
    g.cg_q[gi] = qv;
    const double hx = qv - d2v * vx;
    const bool c = (kc < 3) ? (cm_own & 1) : (cm_own & 2);
    acc4[1] = c ? 0.0 : (vx * vb - 0.5 * vx * hx);
  }
  block_sum<4>(acc4, scratch);
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store(&g.part_misc[(size_t)k * g.n_part + wg], acc4[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_ok = uni_f_last_arrival(g, wg, g.n_wg) ? 1 : 0;
  }
  __syncthreads();
  if (sh_ok) {
    // the last work-group to arrive: fold (the other work-groups' partials were stored write-through before their tickets and are read
    // with device-scope loads) and decide — the TAIL operation's statements
    const int nT = g.n_wg;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk * B < nT; ++kk) {
      const int i = min(tid + kk * B, nT - 1);
      const double wgt = tid + kk * B < nT ? 1.0 : 0.0;
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __hip_atomic_load(&g.part_misc[(size_t)k * g.n_part + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int k = 0; k < 4; ++k) s4[k] += wgt * v[k];
    }
    block_sum<4>(s4, scratch);
    if (tid == 0) {
      g.scal->cand_cost = s4[0];
      g.scal->model_change = s4[1];
      g.scal->step_norm_sq = s4[2];
      g.scal->x_norm_sq = s4[3];
      const int bad = g.flags[1] | (g.flags[2] << 1);
      g.scal->linearize_bad = bad;
      g.flags[1] = 0;
      g.flags[2] = 0;
      g.cg->iters = cnt; g.cg->status = status; g.cg->done = 1;
      lm_device_decide(g, s4[0], s4[1], s4[2], s4[3], bad, 2 | 8, cnt, status);    // accept / reject / stop, on the spot (pgo_lm_rules.h); 8: no mirror here
      LmDev& D = *g.lm;
      CgState::Fused* d = &g.cg->f[wp];
      d->op = D.halt ? F_EXIT : D.accepted ? F_LIN : F_HEAD;
      d->cnt = 0; d->pad = 0;
      d->mirror = 1;         // the next launch publishes the decision to the host
      d->gamma_prev = 0.0; d->alpha_prev = 0.0; d->q_prev = 0.0;
    }
  }
  uni_f_trace_end(g, launch);
}
