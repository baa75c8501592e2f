// pgo_linear.cpp — host side, part 2: what solves (H~ + D^2) x = S g inside an LM iteration.  CG batches with their device-side stop
// and the pinned hand-off, the cluster-Jacobi preconditioner topology, and the three GPU factorisations behind
// SPARSE_NORMAL_CHOLESKY (finial.cpp:536): host analysis, plan uploads, launch sequences.
#include "pgo_internal.h"

// ---- CG driver: batches of iterations, one host check per batch ----
// Host <-> device hand-off (publish_sequence in pgo_kernels.hip): clear the flag, enqueue a sequence whose last kernel
// sets it, spin on the pinned word.  A stream synchronise costs ~20-30 us of wake-up latency per LM iteration phase;
// the spin sees the result ~2 us after the kernel stored it.  Falls back to the blocking call after 50 ms.
int wait_handoff(pgo_problem* P) {
  {
    const auto t0 = Clock::now();
    for (unsigned spins = 1;; ++spins) {
      if (__atomic_load_n(&P->scal->seq, __ATOMIC_ACQUIRE) != 0) return PGO_OK;
      __builtin_ia32_pause();
      if ((spins & 0x3ff) == 0 && seconds_since(t0) > 0.05) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(P->stream));
  if (__atomic_load_n(&P->scal->seq, __ATOMIC_ACQUIRE) == 0) return set_error(PGO_ERR_HIP, "device hand-off flag was not set by the enqueued sequence");
  return PGO_OK;
}

// The step tail (model cost change, delta, candidate, candidate cost, scalar fold).  `gate`: the kernels run only once
// the device-side CG state says "stopped", so the tail can ride behind every CG batch (no host round trip between the
// last CG iteration and the tail); the scalar fold always hands off to the host.
// Several ranks, truncated CG: the owner-only pipelined form (pgo_kernels.h DeviceGraph::pipe_buf) unless PGO_SHARD_PIPE=0
bool pipe_mode(const pgo_problem* P, const pgo::CgParams& prm) {
  const char* e = getenv("PGO_SHARD_PIPE");       // (read per call: the tests run both forms in one process; every rank sees the same environment)
  const bool off = e && e[0] == '0';
  if (off || P->use_graph || P->force_standard_cg) return false;
  if (P->g.world > 1) return pgo::pipe_supported(P->g, prm, P->g.cluster);
  // a session with a coarse level (pgo_coarse.h): incidence-slot storage, k_pipe_cg with the correction between its launches
  if (P->coarse_on) return (P->g.cluster == 1 || P->g.cluster == 2) && prm.q_tolerance >= 0.0 && prm.r_tolerance < 0.0 && P->g.pipe_buf[0] != nullptr && P->g.pairs_whole;
  // One rank (r06): a session that keeps the normal equations in the symmetric tile form — the graphs above the universal stream's size
  // limit — runs the same one-launch CG iteration on it (k_pipe_cg_sym) where the library would take the pipelined recurrences anyway
  // (truncated CG with a forcing term >= 0.01, or pcg_form 2); pcg_form 1 and tighter forcing terms keep Ceres' refreshed CG (k_spmv_sym<0> + k_pcg_update)
  // ... and a session with a coarse level (pgo_coarse.h): incidence-slot storage, k_pipe_cg with the correction between its launches
  const bool asked = P->opt.pcg_form == 2 || (P->opt.pcg_form == 0 && P->opt.eta >= 1e-2);
  return P->sym_storage && asked && !P->universal && !P->pipelined && (P->g.cluster == 1 || P->g.cluster == 2) && prm.q_tolerance >= 0.0 && prm.r_tolerance < 0.0 &&
         P->g.pipe_buf[0] != nullptr;
}
// coarse level (pgo_coarse.h): out += P (P'AP)^-1 P' vec on this rank's rows.  Several ranks: every rank restricts over ITS aggregates, the
// restricted vector (6 doubles per aggregate) is all-gathered, every rank applies the (replicated, bit-identical) inverse to its own rows.
int coarse_apply(pgo_problem* P, const double* vec, double* out, double* out2, int fold_seq) {
  const pgo::CoarsePlan& c = P->coarse;
  pgo::launch_coarse_restrict(P->g, c, vec, P->stream, fold_seq);
  if (P->g.world > 1) { int rc = exchange(P, c.rc, (size_t)6 * c.per_rank); if (rc) return rc; }
  pgo::launch_coarse_correct(P->g, c, out, P->g.pipe_seg, out2, P->stream);
  return PGO_OK;
}
// ... and its set-up for the damped system of this LM iteration: P~, the Galerkin row panels of this rank's aggregates, (all-gather,) inverse
int coarse_setup(pgo_problem* P) {
  const pgo::CoarsePlan& c = P->coarse;
  pgo::launch_coarse_galerkin(P->g, c, P->stream);
  if (P->g.world > 1) { int rc = exchange(P, c.Ac, (size_t)6 * c.per_rank * c.npad); if (rc) return rc; }
  pgo::launch_coarse_invert(c, P->stream);
  return PGO_OK;
}
// several ranks, host-enqueued exchange: only the ranks' boundary rows (+ three sums each) travel per CG iteration
// (pgo_kernels.h DeviceGraph::bx; 5 % of the rows of BASELINE configs[3] on 8 ranks), the full-layout buffer stays the rank's own
static bool boundary_exchange(const pgo_problem* P) { return P->g.world > 1 && P->g.bx[0] != nullptr; }
static int pipe_exchange(pgo_problem* P, int buf) {
  if (boundary_exchange(P)) return exchange(P, P->g.bx[buf], (size_t)P->g.bx_cseg);
  return exchange(P, P->g.pipe_buf[buf], (size_t)P->g.pipe_seg);
}
// one product launch of the owner-only CG (+ its fold): from the symmetric tile form where the session keeps its blocks there
static int pipe_cg_launch(pgo_problem* P, const pgo::CgParams& prm, int seq, unsigned long long gseq = 0, bool last_of_batch = false) {
  if (P->sym_storage) pgo::launch_pipe_cg_sym(sym_view(P), P->sym, prm, seq, P->stream, gseq, last_of_batch);
  else pgo::launch_pipe_cg(P->g, prm, seq, 0, P->stream, gseq, !P->coarse_on);
  // coarse level: the launch left m_J = M_J^-1 w of its rows in the buffer the next launch reads; add P (P'AP)^-1 P' w (the restriction's
  // launch folds the CG launch's partial sums too)
  if (P->coarse_on) return coarse_apply(P, P->g.cg_w, P->g.pipe_buf[(seq & 1) ^ 1], nullptr, seq);
  return PGO_OK;
}

int enqueue_tail(pgo_problem* P, const pgo::CgParams* finish_prm) {
  hipStream_t s = P->stream;
  const pgo::CgParams none{0.0, -1.0, 0, 0};
  const int gate = finish_prm ? 1 : 0;
  if (finish_prm && pipe_mode(P, *finish_prm)) {
    // owner-only CG: the batch's last launch applied the stop test; x of every row is gathered first (cg_x is laid out like an
    // exchange buffer: rank r owns [r * rows_per * 6, ...)), then the tail of the sharded path as below, gated on the CG having stopped
    if (P->g.world == 1) {        // (one rank: the two-launch tail of the one-rank path, gated on the CG state the pipelined kernels keep)
      if (P->sym_storage) pgo::launch_spmv_sym(P->g, P->sym, none, 1 | 64 | 4, 1, s);
      else pgo::launch_spmv_tail(P->g, none, s, 2, 1);
      pgo::launch_step_tail(P->g, s, 1);
      return PGO_OK;
    }
    int rc = exchange(P, P->g.cg_x, (size_t)6 * P->g.rows_per);
    if (rc) return rc;
    if (P->sym_storage) pgo::launch_spmv_sym(P->g, P->sym, none, 1 | 64, 1, s);
    else pgo::launch_spmv_tail(P->g, none, s, 2, 0);
    rc = exchange(P, P->g.cg_q, (size_t)P->g.seg);
    if (rc) return rc;
    pgo::launch_model_delta_and_retract(P->g, s, 1);
    pgo::launch_cost(P->g, P->g.pose_c, 0, s, 1);
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s, 1);
    return PGO_OK;
  }
  if (P->g.world == 1) {
    // two launches: q = A x + candidate poses (diagonal lanes), then model change / norms / candidate cost / fold
    if (P->sym_storage) pgo::launch_spmv_sym(P->g, P->sym, finish_prm ? *finish_prm : none, 1 | (gate ? 2 : 0) | 4, 1, s);
    else pgo::launch_spmv_tail(P->g, finish_prm ? *finish_prm : none, s, gate, 1);
    pgo::launch_step_tail(P->g, s, gate);
    return PGO_OK;
  }
  // several ranks: the vector kernels are replicated over all rows, q crosses the wire in between
  pgo::launch_spmv_tail(P->g, finish_prm ? *finish_prm : none, s, gate, 0);
  int rc = exchange(P, P->g.cg_q, (size_t)P->g.seg);
  if (rc) return rc;
  pgo::launch_model_delta_and_retract(P->g, s, gate);
  pgo::launch_cost(P->g, P->g.pose_c, 0, s, gate);
  pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s, gate);
  return PGO_OK;
}

// A batch is a captured hipGraph of `batch` x (SpMV kernel, update kernel) + the finish kernel.  The
// kernels stop by themselves (device-side `done` flag), so an over-long batch only costs early-exit
// launches; the batch length follows the previous solve's iteration count.
// with_tail: the gated step tail follows in the same graph and its scalar fold hands off; otherwise the finish kernel does.
// start_it: absolute index (1-based) of the batch's first CG iteration — decides which iterations refresh the residual.
int launch_cg_batch(pgo_problem* P, const pgo::CgParams& prm, int batch, bool with_tail, int start_it) {
  hipStream_t s = P->stream;
  // The refresh r = b - A x belongs to Ceres' truncated CG (Q-tolerance stop).  An exact request served by PCG runs to a
  // 1e-13 relative residual, below what a recomputed residual can show in FP64 on an ill-conditioned chain: with the
  // refresh the test would never fire (measured: 27x the iterations on sphere x10), so that mode keeps the recurrence.
  const int period = prm.q_tolerance < 0.0 ? 0 : P->opt.cg_residual_reset_period;
  auto refresh_at = [&](int i) { return period > 0 && ((start_it + i) % period) == 0; };
  if (pipe_mode(P, prm)) {
    // owner-only CG: one launch and one all-gather per iteration (launch seq = iteration index + 1, pipe_begin() ran seq 0)
    for (int i = 0; i < batch; ++i) {
      const int seq = start_it + i;
      if (P->g.peer_tab) { int rcp = pipe_cg_launch(P, prm, seq, ++P->peer_gseq); if (rcp) return rcp; continue; }
      int rc = pipe_cg_launch(P, prm, seq, 0, i == batch - 1);
      if (rc) return rc;
      rc = pipe_exchange(P, (seq & 1) ^ 1);
      if (rc) return rc;
    }
    pgo::launch_pipe_cg(P->g, prm, start_it + batch, with_tail ? 2 : 1, s);    // the stop test of the last iteration; without a tail also the hand-over
    return with_tail ? enqueue_tail(P, &prm) : PGO_OK;
  }
  if (P->use_graph) {
    if (memcmp(&P->cg_graph_params, &prm, sizeof prm) != 0) { P->drop_graph(); P->cg_graph_params = prm; }
    // the captured kernels hold the DeviceGraph by value; the tail touches the pose ping-pong, so the key carries its parity
    // ... and the positions of the residual refreshes depend on the start index modulo the period
    const int key = (4 * batch + (with_tail ? 2 : 0) + ((with_tail && P->g.pose_x != P->d_pose_x.p) ? 1 : 0)) * 64 +
                    (period > 0 ? start_it % period : 0);
    auto it = P->cg_graphs.find(key);
    if (it == P->cg_graphs.end()) {
      pgo_problem::CapturedBatch cb;
      hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        int rc_it = PGO_OK;
        for (int i = 0; i < batch && rc_it == PGO_OK; ++i) rc_it = cg_iteration(P, prm, (i & 1) ^ 1, refresh_at(i));
        if (!with_tail) pgo::launch_pcg_finish(P->g, prm, s, 1);
        else if (rc_it == PGO_OK) rc_it = enqueue_tail(P, &prm);
        e = hipStreamEndCapture(s, &cb.graph);
        if (e == hipSuccess && rc_it != PGO_OK) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipGraphInstantiate(&cb.exec, cb.graph, nullptr, nullptr, 0);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (cb.exec) (void)hipGraphExecDestroy(cb.exec);
        if (cb.graph) (void)hipGraphDestroy(cb.graph);
        P->drop_graph();
        P->use_graph = false;  // fall back to plain stream launches (same kernels)
      } else {
        it = P->cg_graphs.emplace(key, cb).first;
      }
    }
    if (P->use_graph) {
      HIP_TRY(hipGraphLaunch(it->second.exec, s));
      return PGO_OK;
    }
  }
  for (int i = 0; i < batch; ++i) { int rc = cg_iteration(P, prm, (i & 1) ^ 1, refresh_at(i)); if (rc) return rc; }
  if (with_tail) return enqueue_tail(P, &prm);
  pgo::launch_pcg_finish(P->g, prm, s, 1);
  return PGO_OK;
}

// Batch schedule.  The kernels stop on their own, but every iteration enqueued past the stopping point still costs two
// early-exit launches (~5 us, ~10 us under the profiler) and every extra batch a host hand-off plus two gated tail
// launches (~12-17 us).  With n iterations expected, the cheapest fixed batch is ~sqrt(3.4 n); n is not known, so the
// first batch follows the previous solve's count (capped at 8: early LM iterations are poor predictors, late ones need
// 3-6 iterations) and later batches grow like sqrt(3.4 * iterations already enqueued) — not by doubling, which wastes up
// to half of the last batch.  Sizes are quantised so that only a handful of graphs is ever captured.
int pick_batch(const pgo::CgParams& prm, int user_batch, int round, int enqueued, int last_iterations) {
  // (the r01 doubling schedule and the constants 3.4 / 8 were switches PGO_CG_DOUBLING / _SQRTC / _CAP0 / _BATCH0 while they were
  // being measured — 0.397 -> 0.354 ms per C2 LM iteration in r01; closed in r03)
  static const int sizes[] = {2, 4, 6, 8, 12, 16, 24, 32, 48, 64};
  int batch;
  if (user_batch > 0) {
    batch = user_batch;
  } else {
    double target;
    if (round == 0) target = last_iterations > 0 ? std::min(last_iterations, 8) : 6;
    else target = std::sqrt(3.4 * std::max(1, enqueued));
    batch = 64;
    for (int sz : sizes) if (sz >= target) { batch = sz; break; }
  }
  batch = std::max(1, std::min(batch, prm.max_iterations));
  return (batch + 1) & ~1;  // even: every batch starts at an odd iteration (kernels take the parity at launch)
}

// CG start: r0 = b, z0 / u0 = M^-1 r0 (and, owner-only form, the first product w0 = A u0 with its two all-gathers)
int pcg_begin(pgo_problem* P, const pgo::CgParams& prm) {
  hipStream_t s = P->stream;
  // the linearisation decided how much of the other ranks' diagonal blocks to fetch from a predicate of its own (the cluster
  // size is chosen after the first linearisation): the replicated standard CG on diagonal-only blocks would build different
  // preconditioners on different ranks and diverge silently — refuse instead
  if (P->g.world > 1 && P->lin_diag_only == 1 && !pipe_mode(P, prm))
    return set_error(PGO_ERR_INVALID_ARGUMENT, "internal: the linearisation exchanged only the diagonals of the other ranks' blocks but the replicated CG is about to run");
  if (!pipe_mode(P, prm)) {
    if (P->sym_active && !P->sym_storage) {     // the blocks of this linearisation + damping into the symmetric tile form (a rejected step: the damped diagonal slots only)
      pgo::launch_sym_repack(P->g, P->sym, s, P->sym_stale ? 0 : 1);
      P->sym_stale = false;
    }
    pgo::launch_pcg_init(P->g, s);
    return PGO_OK;
  }
  pgo::launch_pipe_init(P->g, s);
  // coarse level: u0 = M^-1 r0 gets its coarse part too (r0 = b; into the exchange buffer the first product reads and into u)
  if (P->coarse_on) { int rcc = coarse_apply(P, P->g.cg_r, P->g.pipe_buf[0], P->g.cg_u, -1); if (rcc) return rcc; }
  if (P->g.peer_tab) {       // device-initiated exchange: the kernels store into every rank's buffer and signal each other
    pgo::launch_peer_signal(P->g, ++P->peer_gseq, s);
    return pipe_cg_launch(P, prm, 0, ++P->peer_gseq);
  }
  if (boundary_exchange(P)) pgo::launch_pipe_pack(P->g, 0, -1, 0, s);      // u0 of the rank's boundary rows into its segment
  int rc = pipe_exchange(P, 0);
  if (rc) return rc;
  rc = pipe_cg_launch(P, prm, 0);
  if (rc) return rc;
  return pipe_exchange(P, 1);
}

int run_pcg(pgo_problem* P, const pgo::CgParams& prm, int batch, int* iterations, int* status) {
  hipStream_t s = P->stream;
  { int rc0 = pcg_begin(P, prm); if (rc0) return rc0; }
  for (int round = 0, enqueued = 0;; ++round) {
    arm_handoff(P);
    const int nb = pick_batch(prm, batch, round, enqueued, 0);
    int rc = launch_cg_batch(P, prm, nb, false, enqueued + 1);
    enqueued += nb;
    if (rc) return rc;
    rc = wait_handoff(P);
    if (rc) return rc;
    if (P->scal->cg_status != -1) break;
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  *iterations = P->scal->cg_iterations;
  *status = P->scal->cg_status;
  P->last_cg_iterations = *iterations;
  return PGO_OK;
}

// ---- cluster-Jacobi preconditioner: which BSR slots fall inside a cluster of CL consecutive poses ----
int prepare_clusters(pgo_problem* P, int CL) {
  if (CL != 2 && CL != 4) CL = 1;
  if (P->cluster_built == CL) { P->g.cluster = CL; return PGO_OK; }
  P->drop_graph();  // captured CG batches hold the DeviceGraph by value
  if (CL > 1) {
    // clusters of the rows this rank owns (row_lo is a multiple of 4); indices local to the rank
    const int c0 = P->g.row_lo / CL;
    const int ncl = std::max(1, (P->g.row_hi - P->g.row_lo + CL - 1) / CL);
    // the BEGIN slot of every edge whose two poses share a cluster (the END twin holds the transposed block: the kernel mirrors)
    std::vector<int> ptr(ncl + 1, 0), slots;
    std::vector<uint8_t> rcs;
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (int t = 0; t < P->g.n_slots; ++t) {
        if (P->h_slot_side[t] != pgo::SIDE_BEGIN) continue;
        const int r = P->h_slot_row[t], c = P->h_slot_col[t];
        if (r / CL != c / CL) continue;
        if (pass == 0) ++ptr[r / CL - c0 + 1];
        else { const int q = fill[r / CL - c0]++; slots[q] = t; rcs[q] = (uint8_t)(((r % CL) << 4) | (c % CL)); }
      }
      if (pass == 0) { for (int k = 0; k < ncl; ++k) ptr[k + 1] += ptr[k]; slots.resize(ptr[ncl]); rcs.resize(ptr[ncl]); }
    }
    HIP_TRY(P->d_cl_ptr.store(ptr, P->stream));
    HIP_TRY(P->d_cl_slot.store(slots, P->stream));
    P->h_cl_slot = slots;
    HIP_TRY(P->d_cl_rc.store(rcs, P->stream));
    const size_t need = (size_t)P->g.world * P->g.rows_per * 36 * CL;   // every rank's clusters, padded
    if (P->d_Minv.n < need) { HIP_TRY(P->d_Minv.alloc(need)); HIP_TRY(P->d_Minv.zero(P->stream)); }
    P->g.Minv = P->d_Minv.p;
    P->g.cl_ptr = P->d_cl_ptr.p;
    P->g.cl_slot = P->d_cl_slot.p;
    P->g.cl_rc = P->d_cl_rc.p;
  }
  P->g.cluster = CL;
  P->cluster_built = CL;
  return PGO_OK;
}

// ---- exact solver: GPU block-sparse Cholesky (pgo_direct.*) ----
// Multifrontal solver: host analysis (front_analyzed_ok) and, once chosen, plan upload (front_usable).
long long front_memory_budget() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)64 << 30; }
  return (long long)(0.6 * (double)free_b);
}
// host only (runs on the analysis thread)
void analyze_front(pgo_problem* P, int N, int n_slots, long long budget, bool* ok, int small_max) {
  pgo::FrontSymbolic& S = P->fsym;
  const auto t_an = Clock::now();
  *ok = pgo::front_analyze(N, P->t_ia(), P->t_ib(), n_slots, P->h_slot_row, P->h_slot_col, P->h_slot_side, budget, &S, small_max, &P->t_is_point());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] front: symbolic analysis %.2f ms (%s)\n", 1e3 * seconds_since(t_an), *ok ? "usable" : "declined");
}

int upload_front(pgo_problem* P) {
  pgo::FrontSymbolic& S = P->fsym;
  const auto t_up = Clock::now();
  hipStream_t s = P->stream;
  UploadScope upload_scope(s);
  HIP_TRY(P->df_perm.upload(S.perm, s));
  HIP_TRY(P->df_idx.upload(S.idx, s));
  HIP_TRY(P->df_child.upload(S.child, s));
  HIP_TRY(P->df_rel.upload(S.rel, s));
  HIP_TRY(P->df_cstart.upload(S.cstart, s));
  HIP_TRY(P->df_wg_job.upload(S.wg_job, s));
  HIP_TRY(P->df_wg_tile.upload(S.wg_tile, s));
  HIP_TRY(P->df_bwd_front.upload(S.bwd_front, s));
  HIP_TRY(P->df_bwd_chunk.upload(S.bwd_chunk, s));
  HIP_TRY(P->df_bwdb_front.upload(S.bwdb_front, s));
  HIP_TRY(P->df_bwdb_chunk.upload(S.bwdb_chunk, s));
  HIP_TRY(P->df_asm_tile.upload(S.asm_tile, s));
  HIP_TRY(P->df_asm_contrib.upload(S.asm_contrib, s));
  HIP_TRY(P->df_col_front.upload(S.col_front, s));
  HIP_TRY(P->df_ablk_ptr.upload(S.ablk_ptr, s));
  HIP_TRY(P->df_ablk_slot.upload(S.ablk_slot, s));
  HIP_TRY(P->df_ablk_front.upload(S.ablk_front, s));
  HIP_TRY(P->df_ablk_pos.upload(S.ablk_pos, s));
  HIP_TRY(P->df_fronts.upload(S.fronts, s));
  HIP_TRY(P->df_jobs.upload(S.jobs, s));
  HIP_TRY(P->df_Fval.alloc((size_t)S.fval_size));
  HIP_TRY(P->df_Winv.alloc((size_t)S.winv_size));
  HIP_TRY(P->df_x.alloc((size_t)6 * S.n));
  HIP_TRY(P->df_x.zero(s));
  pgo::FrontPlan& f = P->fplan;
  f.n = S.n; f.nf = S.nf;
  f.perm = P->df_perm.p; f.fronts = P->df_fronts.p; f.idx = P->df_idx.p; f.child = P->df_child.p; f.rel = P->df_rel.p; f.cstart = P->df_cstart.p;
  f.wg_job = P->df_wg_job.p; f.wg_tile = P->df_wg_tile.p; f.bwd_front = P->df_bwd_front.p; f.bwd_chunk = P->df_bwd_chunk.p;
  f.bwdb_front = P->df_bwdb_front.p; f.bwdb_chunk = P->df_bwdb_chunk.p;
  f.asm_tile = P->df_asm_tile.p; f.asm_contrib = P->df_asm_contrib.p;
  f.col_front = P->df_col_front.p; f.ablk_ptr = P->df_ablk_ptr.p; f.ablk_slot = P->df_ablk_slot.p;
  f.ablk_front = P->df_ablk_front.p; f.ablk_pos = P->df_ablk_pos.p; f.n_ablk = (int)S.ablk_front.size();
  f.jobs = P->df_jobs.p; f.Fval = P->df_Fval.p; f.Winv = P->df_Winv.p; f.x = P->df_x.p;
  // the stages of the single-launch form (pgo_front.h FrontStages); knob factor_fused = 0: one launch per phase of a round
  f.st_table = nullptr; f.st_pred_ptr = nullptr; f.st_pred = nullptr; f.st_need = nullptr; f.st_count = nullptr;
  P->front_epoch = 0; P->front_tickets = 0;
  {
    // Measured (factorisation, single launch vs one launch per phase of a round): Manhattan 2 k / 8 k 1.09 vs 1.23 ms, KITTI-00
    // dense (0.9 GFLOP) 28.1 vs 32.0 ms per 14-iteration solve, Manhattan 10 k (4.5 GFLOP) 3.54 vs 3.57 ms, sphere x10 (383
    // GFLOP) 46 vs 28 ms: every work-group pays a cache write-back and an invalidation of its XCD's L2 where a kernel boundary
    // pays them once, which the GEMM-heavy factorisations cannot afford.  Default: single launch up to 3 GFLOP
    // (knob factor_fused, pgo_tuning.h: 1 always, 0 never — one switch for the three factorisations).
    const double fu = pgo::tuning("factor_fused", -1.0);
    const double limit = 3.0;
    const bool on = fu >= 0.0 ? fu != 0.0 : S.flops <= limit * 1e9;
    P->front_launches = !on || S.st_table.empty();
  }
  if (!S.st_table.empty()) {
    HIP_TRY(P->df_st_table.upload(S.st_table, s));
    HIP_TRY(P->df_st_pred_ptr.upload(S.st_pred_ptr, s));
    HIP_TRY(P->df_st_pred.upload(S.st_pred, s));
    if (S.st_pred.empty()) HIP_TRY(P->df_st_pred.alloc(1));
    HIP_TRY(P->df_st_need.upload(S.st_need, s));
    HIP_TRY(P->df_st_count.alloc(S.st_need.size() + 1));
    HIP_TRY(P->df_st_count.zero(s));
    f.st_table = P->df_st_table.p; f.st_pred_ptr = P->df_st_pred_ptr.p; f.st_pred = P->df_st_pred.p; f.st_need = P->df_st_need.p;
    f.st_count = P->df_st_count.p;
  }
  if (S.mixed) {
    // the small fronts at the bottom of the tree take the small-front kernels (pgo_front.h): their compact arrays
    HIP_TRY(P->ds_sf.upload(S.sfronts, s));
    HIP_TRY(P->ds_urel.upload(S.urel, s));
    HIP_TRY(P->ds_osrc.upload(S.osrc, s));
    HIP_TRY(P->ds_list.upload(S.slevel_front, s));
    HIP_TRY(P->ds_L.alloc((size_t)S.sl_size));
    HIP_TRY(P->ds_U.alloc((size_t)S.su_size));
    HIP_TRY(P->ds_W.alloc((size_t)S.sw_size));
    HIP_TRY(P->ds_upos.alloc((size_t)S.su_size));
    P->splan = pgo::SFrontPlan{P->ds_sf.p, P->ds_list.p, P->ds_urel.p, P->ds_upos.p, P->ds_osrc.p, P->ds_L.p, P->ds_U.p, P->ds_W.p, nullptr};
    pgo::launch_sfront_prepare(P->fplan, P->splan, S, s);
  }
  P->front_usable = true;
  HIP_TRY(upload_scope.finish());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] front: plan upload %.2f ms\n", 1e3 * seconds_since(t_up));
  return PGO_OK;
}

int upload_sfront(pgo_problem* P) {
  pgo::FrontSymbolic& S = P->fsym;
  const auto t_up = Clock::now();
  hipStream_t s = P->stream;
  UploadScope upload_scope(s);
  HIP_TRY(P->df_perm.upload(S.perm, s));
  HIP_TRY(P->df_idx.upload(S.idx, s));
  HIP_TRY(P->df_child.upload(S.child, s));
  HIP_TRY(P->df_rel.upload(S.rel, s));
  HIP_TRY(P->df_ablk_ptr.upload(S.ablk_ptr, s));
  HIP_TRY(P->df_ablk_slot.upload(S.ablk_slot, s));
  HIP_TRY(P->df_ablk_pos.upload(S.ablk_pos, s));
  HIP_TRY(P->df_fronts.upload(S.fronts, s));
  HIP_TRY(P->ds_sf.upload(S.sfronts, s));
  HIP_TRY(P->ds_urel.upload(S.urel, s));
  HIP_TRY(P->ds_osrc.upload(S.osrc, s));
  HIP_TRY(P->ds_L.alloc((size_t)S.sl_size));
  HIP_TRY(P->ds_U.alloc((size_t)S.su_size));
  HIP_TRY(P->ds_W.alloc((size_t)S.sw_size));
  HIP_TRY(P->ds_upos.alloc((size_t)S.su_size));
  HIP_TRY(P->df_x.alloc((size_t)6 * S.n));
  HIP_TRY(P->df_x.zero(s));
  pgo::FrontPlan& f = P->fplan;
  f = pgo::FrontPlan{};
  f.n = S.n; f.nf = S.nf;
  f.perm = P->df_perm.p; f.fronts = P->df_fronts.p; f.idx = P->df_idx.p; f.child = P->df_child.p; f.rel = P->df_rel.p;
  f.ablk_ptr = P->df_ablk_ptr.p; f.ablk_slot = P->df_ablk_slot.p; f.ablk_pos = P->df_ablk_pos.p; f.n_ablk = (int)S.ablk_front.size();
  f.x = P->df_x.p;
  HIP_TRY(P->ds_done.alloc(2 * ((size_t)S.nf + 1)));      // flags of the single-launch forms (factorisation, backward substitution) + their ticket counters
  HIP_TRY(P->ds_done.zero(s));
  P->sfront_epoch = 0;
  P->sfront_tickets = 0;
  {
    P->sfront_levels = pgo::tuning("factor_fused", -1.0) == 0.0;
  }
  P->splan = pgo::SFrontPlan{P->ds_sf.p, nullptr, P->ds_urel.p, P->ds_upos.p, P->ds_osrc.p, P->ds_L.p, P->ds_U.p, P->ds_W.p, P->ds_done.p};
  pgo::launch_sfront_prepare(P->fplan, P->splan, S, s);
  P->sfront_usable = true;
  HIP_TRY(upload_scope.finish());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] front: small-front plan upload %.2f ms\n", 1e3 * seconds_since(t_up));
  return PGO_OK;
}

// Which factorisation serves an exact request on this topology: the host analyses and the choice between them.  No HIP calls
// (it runs on the analysis thread); returns 0 none (the iterative path serves the request), 1 enumerated 6x6 pairs (P->dsym),
// 2 MFMA fronts, 3 small fronts (P->fsym).
int decide_direct_host(pgo_problem* P, int N, int E, int n_slots, long long front_budget) {
  const char* off = getenv("PGO_NO_DIRECT");
  if (off && off[0] == '1') return 0;
  if (P->comm && P->comm->world > 1) return 0;   // the factorisation needs every row: sharded runs use PCG to 1e-13
  // Two GPU factorisations serve an exact request (measured, tools/front_vs_direct.py: KITTI-00 replay 0.34 ms with the
  // enumerated 6x6 pairs vs 0.44 ms multifrontal; KITTI-00 dense 2.1 vs 2.8 ms; Manhattan 2 k 8.6 vs 1.6 ms; Manhattan 10 k
  // 6.3 vs 5.6 ms; sphere x10: declined vs 39 ms).  The multifrontal analysis is the cheap one and runs first; chain-like
  // graphs (largest front below PGO_FRONT_MIN scalars, default 192) then go to the enumerated schedule, everything else stays
  // multifrontal.  PGO_FRONT=1: always multifrontal; 0: never.
  const char* fr = getenv("PGO_FRONT");
  const int front_mode = !fr ? -1 : (fr[0] == '1' ? 1 : 0);
  const int front_min = 192;
  pgo::DirectSymbolic& S = P->dsym;
  bool front_ok = false;
  // a trajectory with a few chords (KITTI-00 replay: 1.14 edges per pose) is the enumerated schedule's case: its analysis runs
  // first there and the multifrontal one is skipped (one-shot solves pay every millisecond of host analysis)
  bool pair_first_done = false, usable = false, front_done = false;
  // Chain-like graphs (E < 1.5 N) first try the small-front plan: when every front fits the LDS of one workgroup (<= 96 scalars;
  // KITTI-00 replay: 84) the factorisation is one launch per tree level (pgo_front.h) — KITTI-00 0.41 vs 0.46 ms per LM
  // iteration, 7.1 vs 7.6 ms per solve against the enumerated schedule.  Not for the union of a batched solve (one 58 KB
  // workgroup per front: 37 vs 25 ms for 16 graphs).  PGO_SFRONT=0 never, =1 for any graph whose fronts are small enough.
  const char* sfe = getenv("PGO_SFRONT");
  const int sf_mode = !sfe ? ((!P->no_sfront && (double)E < 1.5 * (double)N) ? 1 : 0) : (sfe[0] == '1' ? 1 : 0);
  if (front_mode != 0 && sf_mode == 1) {
    analyze_front(P, N, n_slots, front_budget, &front_ok, pgo::SFRONT_MAX);
    front_done = true;
    if (front_ok && P->fsym.small) { S = pgo::DirectSymbolic(); return 3; }
  }
  auto pairs = [&]() {
    const auto t_an = Clock::now();
    usable = pgo::direct_analyze(N, P->t_ia(), P->t_ib(), n_slots, P->h_slot_row, P->h_slot_col, P->h_slot_side, P->h_row_slot_begin, &S, &P->t_is_point());
    if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: symbolic analysis %.2f ms\n", 1e3 * seconds_since(t_an));
  };
  if (front_mode < 0 && (double)E < 1.5 * (double)N) { pairs(); pair_first_done = true; }
  if (front_mode != 0 && !(pair_first_done && usable && !S.hybrid)) {
    if (!front_done) analyze_front(P, N, n_slots, front_budget, &front_ok);
    if (front_ok && (front_mode == 1 || P->fsym.max_front > front_min)) { S = pgo::DirectSymbolic(); return 2; }
  }
  if (front_mode != 1 && !pair_first_done) pairs();
  if (front_ok && (!usable || S.hybrid)) { S.hybrid = false; return 2; }
  return usable ? 1 : 0;   // 0: too much fill / too deep for the enumerated schedule: the iterative path serves the request
}

int prepare_direct(pgo_problem* P) {
  if (P->direct_analyzed) return PGO_OK;
  int kind;
  if (P->analysis_thread.joinable()) {
    P->analysis_thread.join();
    kind = P->analysis_kind;
  } else {
    kind = decide_direct_host(P, P->g.N, P->g.E, P->g.n_slots, front_memory_budget());
  }
  P->direct_analyzed = true;
  P->direct_usable = false;
  P->front_usable = false;
  P->sfront_usable = false;
  pgo::DirectSymbolic& S = P->dsym;
  if (kind == 0) return PGO_OK;
  P->direct_usable = true;      // (cleared again if an upload fails: the error is returned)
  if (kind == 3) { const int rc = upload_sfront(P); if (rc) P->direct_usable = false; return rc; }
  if (kind == 2) { const int rc = upload_front(P); if (rc) P->direct_usable = false; return rc; }
  P->direct_usable = false;
  const auto t_up = Clock::now();
  hipStream_t s = P->stream;
  UploadScope upload_scope(s);
  HIP_TRY(P->dd_perm.upload(S.perm, s));
  HIP_TRY(P->dd_col_ptr.upload(S.col_ptr, s));
  HIP_TRY(P->dd_blk_row.upload(S.blk_row, s));
  HIP_TRY(P->dd_asrc_ptr.upload(S.asrc_ptr, s));
  HIP_TRY(P->dd_asrc_slot.upload(S.asrc_slot, s));
  HIP_TRY(P->dd_upd_ptr.upload(S.upd_ptr, s));
  HIP_TRY(P->dd_upd_a.upload(S.upd_a, s));
  HIP_TRY(P->dd_upd_b.upload(S.upd_b, s));
  HIP_TRY(P->dd_level_ptr.upload(S.level_ptr, s));
  HIP_TRY(P->dd_level_cols.upload(S.level_cols, s));
  HIP_TRY(P->dd_rowl_ptr.upload(S.rowl_ptr, s));
  HIP_TRY(P->dd_rowl_blk.upload(S.rowl_blk, s));
  HIP_TRY(P->dd_rowl_col.upload(S.rowl_col, s));
  HIP_TRY(P->dd_split_blk.upload(S.split_blk, s));
  HIP_TRY(P->dd_split_diag.upload(S.split_diag, s));
  HIP_TRY(P->dd_split_sub.upload(S.split_sub, s));
  HIP_TRY(P->dd_split_sub_diag.upload(S.split_sub_diag, s));
  HIP_TRY(P->dd_upd_split.upload(S.upd_split, s));
  HIP_TRY(P->dd_panel_cols.upload(S.panel_cols, s));
  HIP_TRY(P->dd_blk_lpos.upload(S.blk_lpos, s));
  if (S.panel_cols.empty()) HIP_TRY(P->dd_panel_cols.alloc(1));
  HIP_TRY(P->dd_split_dblk.upload(S.split_dblk, s));
  if (S.split_blk.empty()) { HIP_TRY(P->dd_split_blk.alloc(1)); HIP_TRY(P->dd_split_diag.alloc(1)); HIP_TRY(P->dd_split_dblk.alloc(1)); }
  HIP_TRY(P->dd_col_flag.alloc((size_t)S.nb));
  HIP_TRY(P->dd_col_flag.zero(s));
  P->direct_epoch = 0;
  if (S.split_sub.empty()) { HIP_TRY(P->dd_split_sub.alloc(1)); HIP_TRY(P->dd_split_sub_diag.alloc(1)); }
  HIP_TRY(P->dd_Lval.alloc((size_t)36 * S.nb));
  HIP_TRY(P->dd_y.alloc((size_t)6 * S.n));
  HIP_TRY(P->dd_y.zero(s));
  pgo::DirectPlan& d = P->dplan;
  d.n = S.n; d.nb = S.nb; d.n_levels = S.n_levels;
  d.perm = P->dd_perm.p; d.col_ptr = P->dd_col_ptr.p; d.blk_row = P->dd_blk_row.p;
  d.asrc_ptr = P->dd_asrc_ptr.p; d.asrc_slot = P->dd_asrc_slot.p; d.upd_ptr = P->dd_upd_ptr.p;
  d.upd_a = P->dd_upd_a.p; d.upd_b = P->dd_upd_b.p; d.level_ptr = P->dd_level_ptr.p; d.level_cols = P->dd_level_cols.p;
  d.rowl_ptr = P->dd_rowl_ptr.p; d.rowl_blk = P->dd_rowl_blk.p; d.rowl_col = P->dd_rowl_col.p;
  d.Lval = P->dd_Lval.p; d.y = P->dd_y.p; d.split_blk = P->dd_split_blk.p;
  d.split_diag = P->dd_split_diag.p; d.split_sub = P->dd_split_sub.p; d.split_sub_diag = P->dd_split_sub_diag.p;
  d.upd_split = P->dd_upd_split.p; d.panel_cols = P->dd_panel_cols.p; d.blk_lpos = P->dd_blk_lpos.p;
  d.split_dblk = P->dd_split_dblk.p; d.col_flag = P->dd_col_flag.p;
  P->drop_direct_graph();
  P->direct_usable = true;
  HIP_TRY(upload_scope.finish());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: plan upload %.2f ms\n", 1e3 * seconds_since(t_up));
  return PGO_OK;
}

// the multifrontal factorisation: all stages in one launch (pgo_front.h FrontStages), or one launch per phase of a round
void enqueue_front_factor(pgo_problem* P, const pgo::DeviceGraph& G) {
  hipStream_t s = P->stream;
  if (!P->front_launches) {
    const char* sp_env = getenv("PGO_WAIT_SPINS");
    const int n_tickets = (int)(P->fsym.st_table.size() / 2);
    static const bool want_stamps = false;
    if (want_stamps && P->ds_stamps.n == 0 && P->ds_stamps.alloc(3 * (size_t)n_tickets) != hipSuccess) return;
    const pgo::FrontStages fs{++P->front_epoch, P->front_tickets, n_tickets, (int)P->fsym.st_need.size(), sp_env ? atoi(sp_env) : (1 << 20),
                              want_stamps ? P->ds_stamps.p : nullptr};
    P->front_tickets += (unsigned long long)n_tickets;
    pgo::launch_front_factor(G, P->fplan, P->fsym, s, nullptr, &fs);
    if (want_stamps && P->front_epoch == 3) {      // development aid: the chain of stages that ends last, from the last stage back
      (void)hipStreamSynchronize(s);
      const pgo::FrontSymbolic& S = P->fsym;
      std::vector<long long> st(3 * (size_t)n_tickets);
      (void)hipMemcpy(st.data(), P->ds_stamps.p, st.size() * sizeof(long long), hipMemcpyDeviceToHost);
      const int ns = (int)S.st_need.size();
      std::vector<long long> first(ns, (long long)1 << 62), ready(ns, 0), done(ns, 0);
      std::vector<int> kind(ns, 0), wg1(ns, 0);
      long long t0 = (long long)1 << 62;
      for (int t = 0; t < n_tickets; ++t) {
        const int sg = S.st_table[2 * (size_t)t + 1];
        first[sg] = std::min(first[sg], st[3 * (size_t)t]); ready[sg] = std::max(ready[sg], st[3 * (size_t)t + 1]); done[sg] = std::max(done[sg], st[3 * (size_t)t + 2]);
        kind[sg] = S.st_table[2 * (size_t)t] & 3; wg1[sg] = S.st_table[2 * (size_t)t] >> 2;
        t0 = std::min(t0, st[3 * (size_t)t]);
      }
      auto us = [&](long long t) { return (double)(t - t0) / 100.0; };
      int sg = 0;
      for (int q = 0; q < ns; ++q) if (done[q] > done[sg]) sg = q;
      std::fprintf(stderr, "[pgo] front stamps: %d tickets, %d stages; chain from the last stage back: stage kind(0 asm 1 panel 2 gemm64 3 gemm32) front wgs | first start, last ready, last done (us)\n", n_tickets, ns);
      for (int hops = 0; hops < 400 && sg >= 0; ++hops) {
        const int front = kind[sg] == 0 ? S.asm_tile[8 * (size_t)wg1[sg]] : S.job_front[S.wg_job[wg1[sg]]];
        std::fprintf(stderr, "[pgo]   %5d %d front %4d (c %3d r %3d) wgs %4d | %8.2f %8.2f %8.2f\n", sg, kind[sg], front, S.fronts[front].c, S.fronts[front].r, S.st_need[sg], us(first[sg]), us(ready[sg]), us(done[sg]));
        int best = -1;
        for (int q = S.st_pred_ptr[sg]; q < S.st_pred_ptr[sg + 1]; ++q) if (best < 0 || done[S.st_pred[q]] > done[best]) best = S.st_pred[q];
        sg = best;
      }
    }
  } else {
    pgo::launch_front_factor(G, P->fplan, P->fsym, s, P->fsym.mixed ? &P->splan : nullptr);
  }
}

// factorise (H~ + D^2) and solve for cg_x = (H~ + D^2)^-1 S g; the launch sequence is static -> one hipGraph
// G: P->g, or its copy that carries the device-resident LM state (the kernels then gate themselves on its halt word)
int run_direct(pgo_problem* P, const pgo::DeviceGraph& G) {
  hipStream_t s = P->stream;
  if (P->sfront_usable) {
    if (!P->sfront_levels) {
      // all levels in one launch (SFrontSync): a parent waits for its children's flags instead of for the end of their launch
      const char* sp_env = getenv("PGO_WAIT_SPINS");
      const int max_spins = sp_env ? atoi(sp_env) : (1 << 20);     // ~1 s of polling before the fallback
      if (++P->sfront_epoch == 0x7fffffff) {     // (the ticket counters keep counting: they wrap with the host's copy)
        P->sfront_epoch = 1;
        HIP_TRY(hipMemsetAsync(P->ds_done.p, 0, (size_t)P->fsym.nf * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(P->ds_done.p + P->fsym.nf + 1, 0, (size_t)P->fsym.nf * sizeof(int), s));
      }
      static const bool want_stamps = false;
      if (want_stamps && P->ds_stamps.n == 0) HIP_TRY(P->ds_stamps.alloc(6 * (size_t)P->fsym.nf));
      const pgo::SFrontSync sy{P->ds_done.p, P->sfront_tickets, P->sfront_epoch, max_spins, want_stamps ? P->ds_stamps.p : nullptr};
      const pgo::SFrontSync sy_bwd{P->ds_done.p + P->fsym.nf + 1, P->sfront_tickets, P->sfront_epoch, max_spins, nullptr};
      P->sfront_tickets += (unsigned)P->fsym.nf;
      pgo::launch_sfront_factor(G, P->fplan, P->splan, P->fsym, s, &sy);
      if (want_stamps && P->sfront_epoch == 3) {      // development aid: the critical path of the third factorisation, from the root down
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<long long> st(6 * (size_t)P->fsym.nf);
        HIP_TRY(hipMemcpy(st.data(), P->ds_stamps.p, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        const pgo::FrontSymbolic& S = P->fsym;
        long long t0 = st[0];
        for (int q = 0; q < S.nf; ++q) t0 = std::min(t0, st[6 * (size_t)q]);
        auto us = [&](long long t) { return (double)(t - t0) / 100.0; };     // s_memrealtime: 100 MHz
        int q = S.nf - 1;
        std::fprintf(stderr, "[pgo] sfront stamps (us since the first front started): front c r kids | start wait_done extend_done factor_done published end\n");
        while (q >= 0) {
          const pgo::FrontDesc& D = S.fronts[q];
          std::fprintf(stderr, "[pgo]   front %4d c %2d r %2d kids %2d | %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f\n", q, D.c, D.r, D.child_end - D.child_begin,
                       us(st[6 * (size_t)q]), us(st[6 * (size_t)q + 1]), us(st[6 * (size_t)q + 2]), us(st[6 * (size_t)q + 3]), us(st[6 * (size_t)q + 4]), us(st[6 * (size_t)q + 5]));
          int last = -1;
          for (int ci = D.child_begin; ci < D.child_end; ++ci) { const int ch = S.child[ci]; if (last < 0 || st[6 * (size_t)ch + 4] > st[6 * (size_t)last + 4]) last = ch; }
          q = last;
        }
      }
      // The backward substitution stays one launch per level: in its single-launch form (measured in r02 behind a switch, closed in
      // r03) every front of the tree polls its parent's flag at once and the ten hand-overs take 88 us against 50 us for the ten
      // launches (KITTI-00).
      (void)sy_bwd;
      pgo::launch_sfront_solve(G, P->fplan, P->splan, P->fsym, s, nullptr);
    } else {
      pgo::launch_sfront_factor(G, P->fplan, P->splan, P->fsym, s);
      pgo::launch_sfront_solve(G, P->fplan, P->splan, P->fsym, s);
    }
    return PGO_OK;
  }
  if (P->front_usable) {
    enqueue_front_factor(P, G);
    pgo::launch_front_solve(G, P->fplan, P->fsym, s, P->fsym.mixed ? &P->splan : nullptr);
    return PGO_OK;
  }
  const pgo::DirectSymbolic& S = P->dsym;
  if (P->use_graph && !P->direct_exec) {
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      pgo::launch_direct_factor(G, P->dplan, S, s);
      pgo::launch_direct_solve(G, P->dplan, S.level_ptr.data(), S.fused_from_level, s);
      e = hipStreamEndCapture(s, &P->direct_graph);
      if (e == hipSuccess) e = hipGraphInstantiate(&P->direct_exec, P->direct_graph, nullptr, nullptr, 0);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); P->drop_direct_graph(); P->use_graph = false; }
  }
  if (P->direct_exec) {
    HIP_TRY(hipGraphLaunch(P->direct_exec, s));
  } else {
    if (++P->direct_epoch == 0x7fffffff) { P->direct_epoch = 1; HIP_TRY(P->dd_col_flag.zero(s)); }
    pgo::launch_direct_factor(G, P->dplan, S, s, P->split_two_launch ? 0 : P->direct_epoch);
    pgo::launch_direct_solve(G, P->dplan, S.level_ptr.data(), S.fused_from_level, s);
  }
  return PGO_OK;
}
int run_direct(pgo_problem* P) { return run_direct(P, P->g); }

pgo::CgParams cg_params_for(const pgo_solver_options& o) {
  pgo::CgParams prm;
  if (o.linear_solver_type == PGO_BLOCK_JACOBI_PCG) {
    prm.q_tolerance = o.eta;
    prm.r_tolerance = -1.0;  // LevenbergMarquardtStrategy disables the residual test
    prm.max_iterations = o.max_linear_solver_iterations;
    prm.min_iterations = o.min_linear_solver_iterations;
  } else {
    // SPARSE_NORMAL_CHOLESKY is an exact solve.  Until the direct factorisation path is wired in it is
    // served by the same PCG run to a tight relative residual.
    prm.q_tolerance = -1.0;
    prm.r_tolerance = o.exact_r_tolerance;
    prm.max_iterations = 200000;
    prm.min_iterations = 0;
  }
  return prm;
}

// the in-kernel counters of the single-launch factorisations no longer match the host's after launches that exited at a halt
int resync_direct_counters(pgo_problem* P) {
  hipStream_t s = P->stream;
  if (P->front_usable && P->df_st_count.n) { HIP_TRY(P->df_st_count.zero(s)); P->front_epoch = 0; P->front_tickets = 0; }
  if (P->sfront_usable && P->ds_done.n) { HIP_TRY(P->ds_done.zero(s)); P->sfront_epoch = 0; P->sfront_tickets = 0; }
  if (P->dd_col_flag.n) { HIP_TRY(P->dd_col_flag.zero(s)); P->direct_epoch = 0; }
  return PGO_OK;
}
