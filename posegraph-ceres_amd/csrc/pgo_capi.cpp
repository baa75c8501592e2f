// pgo_capi.cpp — the rest of the C ABI: Problem::Evaluate analogue, normal equations and linear solves for tests, kernel timing hooks,
// row-shard arithmetic and the communicator entry points.
#include "pgo_internal.h"

namespace pgo { int comm_stress(Comm* c, int iters, size_t seg, hipStream_t s, int* mismatches); }

// =================================================================================================
// C ABI (include/pgo.h)
// =================================================================================================
extern "C" {

int pgo_evaluate(pgo_problem* P, double* cost, double* residuals, double* jac_begin, double* jac_end, double* gradient) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_evaluate during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.loss_kind = P->loss_kind; P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  const int E = P->g.E;
  if (residuals || jac_begin || jac_end) {
    if (residuals) HIP_TRY(P->d_tmp_a.alloc((size_t)6 * E));
    if (jac_begin) HIP_TRY(P->d_tmp_b.alloc((size_t)36 * E));
    if (jac_end) HIP_TRY(P->d_tmp_c.alloc((size_t)36 * E));
    if (E > 0) pgo::launch_evaluate_edges(P->g, P->g.pose_x, residuals ? P->d_tmp_a.p : nullptr, jac_begin ? P->d_tmp_b.p : nullptr,
                               jac_end ? P->d_tmp_c.p : nullptr, s);
    if (residuals && E) HIP_TRY(staged_d2h(residuals, P->d_tmp_a.p, sizeof(double) * 6 * E, s));
    if (jac_begin && E) HIP_TRY(staged_d2h(jac_begin, P->d_tmp_b.p, sizeof(double) * 36 * E, s));
    if (jac_end && E) HIP_TRY(staged_d2h(jac_end, P->d_tmp_c.p, sizeof(double) * 36 * E, s));
  }
  if (cost) {
    pgo::launch_cost(P->g, P->g.pose_x, 0, s);
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s);
  }
  if (gradient) {
    rc = fill_scale_one(P);
    if (rc) return rc;
    rc = linearize_all(P);
    if (rc) return rc;
    rc = pose_rows_to_host(P, gradient, P->g.grad, 6);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (cost) *cost = P->scal->cand_cost;
  return PGO_OK;
}

int pgo_normal_equations(pgo_problem* P, double* diag, double* offdiag, double* gradient) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_normal_equations during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.loss_kind = P->loss_kind; P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  rc = fill_scale_one(P);
  if (rc) return rc;
  rc = linearize_all(P);
  if (rc) return rc;
  const int E = P->g.E;
  if (diag) { rc = pose_rows_to_host(P, diag, P->g.Hdiag, 36); if (rc) return rc; }
  if (gradient) { rc = pose_rows_to_host(P, gradient, P->g.grad, 6); if (rc) return rc; }
  std::vector<double> bsr;
  if (offdiag) {
    bsr.resize((size_t)P->g.n_slots * 36);
    HIP_TRY(staged_d2h(bsr.data(), P->g.bsr_val, bsr.size() * sizeof(double), s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (offdiag)
    for (int e = 0; e < E; ++e) {
      const int t = P->edge_begin_slot[e];
      for (int k = 0; k < 36; ++k) {
        const int pos = pgo::bsr_pos(P->g.blk_packed, pgo::SIDE_BEGIN, k);
        offdiag[(size_t)36 * e + k] = pos < 0 ? 0.0 : bsr[pgo::bsr_index(t, pos)];
      }
    }
  return PGO_OK;
}

int pgo_linear_solve(pgo_problem* P, const pgo_solver_options* options, const double* d2, const double* b, double* x, int* iterations) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_linear_solve during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P || !options || !d2 || !b || !x) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_linear_solve");
  int rc = prepare(P);
  if (rc) return rc;
  struct StandardCg { pgo_problem* p; explicit StandardCg(pgo_problem* q) : p(q) { p->force_standard_cg = true; } ~StandardCg() { p->force_standard_cg = false; } } standard_cg(P);
  hipStream_t s = P->stream;
  P->g.loss_kind = P->loss_kind; P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  rc = fill_scale_one(P);
  if (rc) return rc;
  rc = prepare_clusters(P, options->pcg_cluster_poses);
  if (rc) return rc;
  rc = linearize_all(P);
  if (rc) return rc;
  rc = pose_rows_to_device(P, P->g.d2, d2, 6, 1.0);      // (padding poses of a sharded problem: an identity row, right-hand side 0)
  if (rc) return rc;
  rc = pose_rows_to_device(P, P->g.grad, b, 6);  // rhs = scale(=1) * grad
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  rc = damping_all(P, 1.0, 0.0, 0.0, 2);
  if (rc) return rc;
  int it = 0, status = 0;
  if (options->linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY) {
    rc = prepare_direct(P);
    if (rc) return rc;
  }
  if (options->linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable) {
    rc = run_direct(P);
    if (rc) return rc;
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s);
    HIP_TRY(hipStreamSynchronize(s));
    if ((P->scal->linearize_bad & 4) && (P->front_usable ? !P->front_launches : P->sfront_usable ? !P->sfront_levels : !P->split_two_launch)) {   // an in-kernel wait ran out (lm_advance)
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      rc = run_direct(P);
      if (rc) return rc;
      pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s);
      HIP_TRY(hipStreamSynchronize(s));
    }
    if (P->scal->linearize_bad) status = 2;
  } else {
    P->opt.cg_residual_reset_period = options->cg_residual_reset_period;   // launch_cg_batch reads the refresh period from P->opt
    P->sym_active = false; P->sym_storage = false;
    if (sym_wanted(P)) {               // large graph on one rank (or PGO_SYM=1): the CG products read the symmetric tile form
      rc = sym_prepare(P);
      if (rc) return rc;
      P->sym_active = P->sym_ready;
    }
    rc = run_pcg(P, cg_params_for(*options), options->cg_batch, &it, &status);
    P->sym_active = false;
  }
  if (rc) return rc;
  rc = pose_rows_to_host(P, x, P->g.cg_x, 6);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  if (iterations) *iterations = it;
  if (status == 2) return set_error(PGO_ERR_NUMERICAL, "PCG broke down with non-finite values");
  return PGO_OK;
}

int pgo_plus(pgo_problem* P, const double* delta) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_plus during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P || !delta) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_plus");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  const size_t m = (size_t)6 * P->g.N;
  HIP_TRY(P->d_tmp_a.alloc(m));
  rc = pose_rows_to_device(P, P->d_tmp_a.p, delta, 6);
  if (rc) return rc;
  pgo::launch_apply_step(P->g, P->d_tmp_a.p, s);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return download_poses(P, P->g.pose_c);
}

int pgo_solver_cg_form(pgo_problem* P) {
  if (!P || !P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_cg_form needs a stepping session (call pgo_solver_begin first)");
  return P->g.world > 1 ? (pipe_mode(P, cg_params_for(P->opt)) ? 2 : 1) : (P->uni_resident ? 4 : P->uni_fused ? 3 : pipe_mode(P, cg_params_for(P->opt)) ? 2 : 0);
}

int pgo_solver_exchange_doubles(pgo_problem* P) {
  if (!P || !P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_exchange_doubles needs a stepping session (call pgo_solver_begin first)");
  if (P->g.world <= 1) return 0;
  if (!pipe_mode(P, cg_params_for(P->opt))) return P->g.seg;
  return P->g.bx[0] ? P->g.bx_cseg : P->g.pipe_seg;
}

static const int TRACE_WORDS = pgo::UNI_F_TRACE_WORDS;
int pgo_solver_trace_start(pgo_problem* P, int max_launches) {
  if (!P || max_launches < 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_solver_trace_start");
  if (!P->lm.active || !P->stream_ready) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_trace_start needs a stepping session (call pgo_solver_begin first)");
  if (!P->uni_fused) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_trace_start: this session does not run the fused universal stream");
  HIP_TRY(hipStreamSynchronize(P->stream));
  if (max_launches == 0) { P->g.oplog = nullptr; P->g.oplog_cap = 0; P->g.oplog_indexed = 0; return PGO_OK; }
  if (max_launches > (0x7fffffff - 1) / TRACE_WORDS) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_trace_start: at most %d launches can be traced", (0x7fffffff - 1) / TRACE_WORDS);
  // the launch index the kernels record under restarts whenever the device state is uploaded afresh; make that happen now
  P->pipe_dirty = true;
  const size_t cap = 1 + (size_t)TRACE_WORDS * max_launches;
  HIP_TRY(P->d_oplog.alloc(cap));
  HIP_TRY(hipMemsetAsync(P->d_oplog.p, 0, cap * sizeof(long long), P->stream));
  HIP_TRY(hipStreamSynchronize(P->stream));
  P->g.oplog = P->d_oplog.p; P->g.oplog_cap = (int)cap; P->g.oplog_indexed = 1;
  P->uni_host_launches = 0; P->uni_host_enqueue_s = 0.0;
  return PGO_OK;
}

int pgo_solver_trace_read(pgo_problem* P, long long* records, int capacity, double host[2]) {
  if (!P || !records || capacity < 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_solver_trace_read");
  if (!P->g.oplog || !P->g.oplog_indexed || !P->uni_fused) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_trace_read: no trace is being recorded (pgo_solver_trace_start)");
  HIP_TRY(hipStreamSynchronize(P->stream));
  const int n = std::min(std::min(P->uni_enq, (P->g.oplog_cap - 1) / TRACE_WORDS), capacity);
  std::vector<long long> h((size_t)TRACE_WORDS * std::max(n, 1));
  if (n) HIP_TRY(hipMemcpy(h.data(), P->g.oplog + 1, (size_t)TRACE_WORDS * n * sizeof(long long), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    const long long* w = h.data() + (size_t)TRACE_WORDS * i;
    long long end = 0;
    for (int x = 0; x < TRACE_WORDS - 2; ++x) end = std::max(end, w[2 + x]);
    records[4 * (size_t)i] = w[0] & 7;
    records[4 * (size_t)i + 1] = w[0] >> 3;
    records[4 * (size_t)i + 2] = end;
    records[4 * (size_t)i + 3] = w[1];
  }
  if (host) { host[0] = (double)P->uni_host_launches; host[1] = P->uni_host_enqueue_s; }
  return n;
}

int pgo_time_kernel(pgo_problem* P, const char* kernel, int repeats, double* avg_ms) {
  if (!P || !kernel || repeats <= 0 || !avg_ms) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_time_kernel");
  if (P->topo_dirty || !P->stream_ready) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel needs a prepared problem (call pgo_solver_begin first)");
  hipStream_t s = P->stream;
  const std::string k = kernel;
  pgo::CgParams prm = cg_params_for(P->opt);
#ifdef PGO_ABLATE
  P->g.debug = (int)pgo::tuning("debug", 0.0);
#endif
  if (k == "evaluate") {
    HIP_TRY(P->d_tmp_a.alloc((size_t)6 * P->g.E));
    HIP_TRY(P->d_tmp_b.alloc((size_t)36 * P->g.E));
    HIP_TRY(P->d_tmp_c.alloc((size_t)36 * P->g.E));
  }
  // "pcg_spmv" repeats the SpMV kernel of CG iteration 1 (the update kernel never runs, so the
  // iteration counter stays put); "pcg_update" likewise repeats the update of iteration 1.
  if (k == "sym_spmv" || k == "sym_repack" || k == "sym_plain" || k == "sym_linearize_rows" || k == "sym_linearize_lean" || k == "sym_lean_check") {
    int rcs = sym_prepare(P);
    if (rcs) return rcs;
    if (!P->sym_ready) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('%s'): the graph does not fit the symmetric tile form", kernel);
  }
  // (a session that keeps the symmetric form as its ONLY storage has no current incidence-slot blocks to repack from: damping goes
  // through the form's view, a repack would overwrite the live form with the blocks of iteration zero)
  if (k == "sym_repack" && P->sym_storage)
    return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('sym_repack'): this session stores the normal equations in the symmetric form only; there is nothing to repack");
  if (k == "pcg_spmv" || k == "pcg_update" || k == "pcg_iteration" || k == "sym_spmv") {
    if (P->sym_storage) pgo::launch_damping(sym_view(P), P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    else pgo::launch_damping(P->g, P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    pgo::launch_pcg_init(P->g, s);
  }
  if (k == "pcg_update") pgo::launch_pcg_spmv_only(P->g, prm, 1, s);
  if ((k == "sym_spmv" || k == "sym_plain") && !P->sym_storage) pgo::launch_sym_repack(P->g, P->sym, s);
  const bool wants_factor = k == "direct" || k == "front_factor" || k == "front_solve";
  if (wants_factor) {
    if (!P->direct_usable) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('%s'): no GPU factorisation prepared for this problem", kernel);
    if ((k != "direct") && !P->front_usable) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('%s'): the multifrontal solver is not in use", kernel);
    pgo::launch_damping(P->g, P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    if (k == "front_solve") enqueue_front_factor(P, P->g);
  }
  auto once = [&]() -> int {
    if (k == "exchange") {       // the collective of the sharded path as enqueued between two CG launches: the boundary segments where the session
      if (!P->comm) return -2;   // exchanges those (in place, between launches: every rank re-sends what its segment holds), else the q segment
      const char* what = "";
      if (P->g.bx[0] && pipe_mode(P, cg_params_for(P->opt))) return P->comm->all_gather(P->g.bx[0], (size_t)P->g.bx_cseg, s, &what) != 0 ? -3 : 0;
      return P->comm->all_gather(P->g.cg_q, (size_t)P->g.seg, s, &what) != 0 ? -3 : 0;
    }
    if (k == "direct") return run_direct(P);
    if (k == "front_factor") { enqueue_front_factor(P, P->g); return 0; }
    if (k == "front_solve") { pgo::launch_front_solve(P->g, P->fplan, P->fsym, s, P->fsym.mixed ? &P->splan : nullptr); return 0; }
    if (k == "linearize") pgo::launch_linearize(P->g, s);
    else if (k == "cost") pgo::launch_cost(P->g, P->g.pose_x, 5, s);
    else if (k == "evaluate") pgo::launch_evaluate_edges(P->g, P->g.pose_x, P->d_tmp_a.p, P->d_tmp_b.p, P->d_tmp_c.p, s);
    else if (k == "spmv") pgo::launch_spmv_plain(P->g, s);
    else if (k == "pcg_spmv") pgo::launch_pcg_spmv_only(P->g, prm, 1, s);
    else if (k == "pcg_update") pgo::launch_pcg_update_only(P->g, 1, s);
    else if (k == "sym_spmv") pgo::launch_spmv_sym(P->g, P->sym, prm, 1, 0, s);
    else if (k == "sym_plain") pgo::launch_spmv_sym(P->g, P->sym, prm, 1, 1, s);
    else if (k == "sym_repack") pgo::launch_sym_repack(P->g, P->sym, s);
    else if (k == "sym_linearize_rows") { pgo::DeviceGraph gs = P->g; gs.sym_dst = P->sy_dst.p; gs.sym_val = P->sym.val; pgo::launch_linearize_symout(gs, s); }
    else if (k == "sym_linearize_lean") { pgo::DeviceGraph gs = P->g; gs.sym_dst = P->sy_dst.p; gs.sym_val = P->sym.val; pgo::launch_linearize_lean(gs, s); }
    else if (k == "pcg_iteration") pgo::launch_pcg_iteration(P->g, prm, 1, s);
    else if (k == "empty") pgo::launch_debug(P->g, 0, s);
    else if (k == "touch") pgo::launch_debug(P->g, 1, s);
    else return -1;
    return 0;
  };
  if (k == "uni_cg") {
    // one CG iteration of the universal stream in situ: a head launch starts a CG whose stopping tests are disabled, then 200
    // (vector, slot) pairs run between two events on the solver stream — what the timed region of bench.py consists of
    if (!P->universal) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('uni_cg'): this session does not use the universal stream");
    if (P->pipe_dirty) { int rc = lm_upload_state(P); if (rc) return rc; P->pipe_dirty = false; }
    pgo::CgParams np{-1.0, -1.0, 1 << 30, 0};
    pgo::DeviceGraph gp = P->g;
    gp.lm = P->d_lm.p;
    const int pairs = 200;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    double total = 0;
    for (int r = 0; r < repeats + 1; ++r) {
      if (P->uni_resident) {
        // resident form: head, then ONE launch that runs exactly `pairs` CG iterations (stopping tests off, iteration limit = pairs)
        pgo::CgParams nr{-1.0, -1.0, pairs, 0};
        pgo::launch_lm_budget(gp, -1, s, 0);
        pgo::launch_uni_r(gp, nr, 0, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s);
        HIP_TRY(hipEventRecord(a, s));
        pgo::launch_uni_r(gp, nr, 1, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s);
      } else if (P->uni_fused) {
        // fused form: head, first product, then 200 CG launches (one launch is one CG iteration)
        int L = 0;
        pgo::launch_lm_budget(gp, -1, s, L);
        pgo::launch_uni_f(gp, np, L++, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s);
        pgo::launch_uni_f(gp, np, L++, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s);
        HIP_TRY(hipEventRecord(a, s));
        for (int i = 0; i < pairs; ++i) pgo::launch_uni_f(gp, np, L++, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s);
      } else {
        pgo::launch_lm_budget(gp, -1, s);
        pgo::launch_uni_v(gp, np, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s);    // head: damping, preconditioner, CG start
        pgo::launch_uni_s(gp, np, 0, s);
        HIP_TRY(hipEventRecord(a, s));
        for (int i = 0; i < pairs; ++i) { pgo::launch_uni_v(gp, np, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, s); pgo::launch_uni_s(gp, np, 0, s); }
      }
      HIP_TRY(hipEventRecord(b, s));
      HIP_TRY(hipEventSynchronize(b));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, a, b));
      if (r > 0) total += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    P->uni_enq = 0;
    P->pipe_dirty = true;        // the device state (CG counters, operation words) is re-uploaded before the next LM step
    int its = pairs;
    if (P->uni_resident) {       // the one launch ran until its own breakdown test or the iteration limit: divide by what it did
      pgo::CgState cs;
      HIP_TRY(hipMemcpy(&cs, P->d_cg.p, sizeof cs, hipMemcpyDeviceToHost));
      its = std::max(1, cs.iters);
    }
    *avg_ms = total / repeats / its;
    return PGO_OK;
  }
  if (k == "sym_pipe_cg") {
    // one iteration of the pipelined CG on the symmetric tile form (k_pipe_cg_sym + its fold) in situ: groups of 16 consecutive
    // iterations of a freshly started CG with the stopping tests off, HIP events on the solver stream around each group
    if (!P->sym_storage || !pipe_mode(P, prm)) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('sym_pipe_cg'): this session does not run the pipelined CG on the symmetric form");
    if (P->g.world > 1) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('sym_pipe_cg'): one rank only (no exchange is enqueued here)");
    pgo::CgParams np{-1.0, -1.0, 1 << 30, 0};
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    pgo::launch_damping(sym_view(P), P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    const int group = 16, groups = std::max(1, repeats / group);
    double total = 0;
    for (int r = 0; r < groups + 1; ++r) {
      pgo::launch_pipe_init(P->g, s);
      pgo::launch_pipe_cg_sym(sym_view(P), P->sym, np, 0, s);
      HIP_TRY(hipEventRecord(a, s));
      for (int i = 1; i <= group; ++i) pgo::launch_pipe_cg_sym(sym_view(P), P->sym, np, i, s);
      HIP_TRY(hipEventRecord(b, s));
      HIP_TRY(hipEventSynchronize(b));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, a, b));
      if (r > 0) total += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *avg_ms = total / groups / group;
    return PGO_OK;
  }
  if (k == "pcg_graph") {
    // average time of one CG iteration inside a captured batch with every stopping test disabled
    pgo::CgParams np{-1.0, -1.0, 1 << 30, 0};
    const int batch = 200;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    pgo::launch_damping(P->g, P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    double total = 0;
    for (int r = 0; r < repeats + 1; ++r) {
      pgo::launch_pcg_init(P->g, s);
      HIP_TRY(hipEventRecord(a, s));
      int rc = launch_cg_batch(P, np, batch);
      if (rc) return rc;
      HIP_TRY(hipEventRecord(b, s));
      HIP_TRY(hipEventSynchronize(b));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, a, b));
      if (r > 0) total += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    P->drop_graph();
    *avg_ms = total / repeats / batch;
    return PGO_OK;
  }
  if (k == "sym_lean_check") {
    // not a timing: k_linearize_symout and k_linearize_lean (pgo_lean_kernels.hip) write the symmetric form, the diagonal blocks and
    // the gradient of the current point one after the other (the form pre-set to a sentinel before the second, so a block it does
    // not write shows); *avg_ms receives the largest difference relative to the largest entry of its group (blocks / diagonal / gradient)
    pgo::DeviceGraph gs = P->g;
    gs.sym_dst = P->sy_dst.p; gs.sym_val = P->sym.val;
    const size_t nv = P->sy_val.n, nh = (size_t)36 * P->g.N, ng = (size_t)6 * P->g.N;
    std::vector<double> v0(nv), v1(nv), h0(nh), h1(nh), g0(ng), g1(ng);
    pgo::launch_linearize_symout(gs, s);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(v0.data(), P->sym.val, nv * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h0.data(), P->g.Hdiag, nh * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(g0.data(), P->g.grad, ng * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> sentinel(nv, 1.0e300);
    HIP_TRY(hipMemcpy(P->sym.val, sentinel.data(), nv * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(P->g.Hdiag, 0x7f, nh * sizeof(double), s));      // (on the solver's stream: it does not wait for the null stream)
    HIP_TRY(hipMemsetAsync(P->g.grad, 0x7f, ng * sizeof(double), s));
    pgo::launch_linearize_lean(gs, s);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(v1.data(), P->sym.val, nv * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h1.data(), P->g.Hdiag, nh * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(g1.data(), P->g.grad, ng * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(P->sym.val, v0.data(), nv * sizeof(double), hipMemcpyHostToDevice));       // the session goes on with the first kernel's result
    HIP_TRY(hipMemcpy(P->g.Hdiag, h0.data(), nh * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(P->g.grad, g0.data(), ng * sizeof(double), hipMemcpyHostToDevice));
    auto rel = [](const std::vector<double>& a, const std::vector<double>& b, bool written_only) {
      double m = 0, e = 0;
      for (size_t i = 0; i < a.size(); ++i) {
        if (written_only && b[i] == 1.0e300) continue;      // (places of the form nobody linearises into: diagonal slots, padding)
        m = std::max(m, std::fabs(a[i]));
        const double d = std::fabs(a[i] - b[i]);
        e = std::max(e, d == d ? d : 1.0e300);
      }
      return m > 0 ? e / m : e;
    };
    // a place the first kernel wrote and the second did not shows as 1e300 against a block entry: count those separately
    size_t unwritten = 0;
    { std::vector<double> vs(nv);
      HIP_TRY(hipMemcpy(P->sym.val, sentinel.data(), nv * sizeof(double), hipMemcpyHostToDevice));
      pgo::launch_linearize_symout(gs, s);
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipMemcpy(vs.data(), P->sym.val, nv * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < nv; ++i) unwritten += (vs[i] == 1.0e300) != (v1[i] == 1.0e300);
      HIP_TRY(hipMemcpy(P->sym.val, v0.data(), nv * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(P->g.Hdiag, h0.data(), nh * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(P->g.grad, g0.data(), ng * sizeof(double), hipMemcpyHostToDevice)); }
    const double worst = std::max(rel(v0, v1, true), std::max(rel(h0, h1, false), rel(g0, g1, false)));
    *avg_ms = unwritten ? 1.0e300 : worst;
    return PGO_OK;
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  if (const int orc = once()) {
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return set_error(PGO_ERR_INVALID_ARGUMENT, orc == -2 ? "pgo_time_kernel('%s'): no communicator attached (pgo_comm_init)" : orc == -3 ? "pgo_time_kernel('%s'): the all-gather failed" : "unknown kernel '%s'", kernel);
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipEventRecord(e0, s));
  for (int i = 0; i < repeats; ++i) once();
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / repeats;
  if (wants_factor) {
    // the factorisations report through flags[2] (bit 0: pivot, bit 1: an in-kernel wait of a single-launch form ran out); nobody
    // folds it here, so read and clear it: a timed-out wait means the figure is worthless and the next LM iteration must not
    // inherit the bit
    int f2 = 0;
    HIP_TRY(hipMemcpyAsync(&f2, P->d_flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(P->d_flags.p + 2, 0, sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    if (f2 & 2) {
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      int rc = resync_direct_counters(P);
      if (rc) return rc;
      return set_error(PGO_ERR_NUMERICAL, "pgo_time_kernel('%s'): an in-kernel wait of the single-launch factorisation timed out; the "
                       "problem now uses one launch per step, time it again", kernel);
    }
  }
  return PGO_OK;
}

// ---- sharding helpers ------------------------------------------------------------------------------
// Development / test hook: ONE application of the trust-region rules (pgo_lm_rules.h lm_decide — the function both the device and
// the host driver apply) on the host, no GPU involved.  state = {radius, decrease_factor, x_cost, x_norm} in and out;
// step = {cand_cost, model_change, step_norm_sq, x_norm_sq}; out = {outcome, step_is_successful, relative_decrease, cost_change,
// radius after, value quoted by a termination message}.
int pgo_debug_lm_decide(const pgo_solver_options* options, double state[4], const double step[4], int cg_status, double out[6]) {
  if (!options || !state || !step || !out) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_debug_lm_decide");
  pgo::LmCore C{state[0], state[1], state[2], state[3], 0.0, 0, 0, 0, 0};
  const pgo::LmStepIn in{step[0], step[1], step[2], step[3], 0, cg_status, 0, 0};
  pgo::LmRecord nx{};
  double value = 0.0;
  const pgo::LmOutcome o = pgo::lm_decide(C, lm_tolerances(*options), in, nx, value);
  state[0] = C.radius; state[1] = C.decrease_factor; state[2] = C.x_cost; state[3] = C.x_norm;
  out[0] = (double)o; out[1] = (double)nx.step_is_successful; out[2] = nx.relative_decrease; out[3] = nx.cost_change; out[4] = C.radius; out[5] = value;
  return PGO_OK;
}

int pgo_shard_range(long long n, int rank, int world, long long* begin, long long* end) {
  if (n < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_shard_range");
  const long long base = n / world, rem = n % world;
  *begin = base * rank + std::min<long long>(rank, rem);
  *end = *begin + base + (rank < rem ? 1 : 0);
  return PGO_OK;
}

// Row ownership of the sharded solve: equal segments (the all-gather exchanges equal-sized pieces) of rows_per poses, rows_per a
// multiple of 4 so that the 2- and 4-pose preconditioner clusters never straddle two ranks; the last ranks may own fewer
// rows or none.  prepare() calls this very function.
int pgo_row_shard_range(long long n_poses, int rank, int world, long long* begin, long long* end, int* rows_per_out) {
  if (n_poses < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_row_shard_range");
  long long rows_per = (n_poses + world - 1) / world;
  rows_per = std::max<long long>(4, (rows_per + 3) / 4 * 4);
  *begin = std::min(n_poses, (long long)rank * rows_per);
  *end = std::min(n_poses, (long long)(rank + 1) * rows_per);
  if (rows_per_out) *rows_per_out = (int)rows_per;
  return PGO_OK;
}

// THE row ownership of the sharded solve since r06 (prepare() calls this very function): contiguous shares of the poses cut where the
// INCIDENCE SLOTS balance — pose v weighs 1 + degree(v), the slots of its block row — at multiples of 4 (2- and 4-pose preconditioner
// clusters never straddle two ranks).  cut[r] .. cut[r + 1] is rank r's share; rows_per = the longest share rounded up to a multiple of
// 4 = the segment every rank's rows occupy in the exchanged arrays (the device numbers the poses of rank r from r * rows_per).
// development / test knobs (pgo_tuning.h): what sixteen environment variables were until r06
int pgo_tuning_set(const char* name, double value) {
  if (!pgo::tuning_set(name, value)) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_tuning_set: no knob named '%s' (pgo_tuning_describe lists them)", name ? name : "(null)");
  return PGO_OK;
}
int pgo_tuning_get(const char* name, double* value, int* is_set) {
  if (!pgo::tuning_known(name)) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_tuning_get: no knob named '%s'", name ? name : "(null)");
  if (is_set) *is_set = pgo::tuning_is_set(name) ? 1 : 0;
  if (value) *value = pgo::tuning(name, std::nan(""));
  return PGO_OK;
}
int pgo_tuning_describe(int index, const char** name, const char** what) {
  int n = 0;
  const pgo::TuningKnob* k = pgo::tuning_knobs(&n);
  if (index < 0 || index >= n) return n;          // (the count, for the caller's loop; no error text: asking is how one learns it)
  if (name) *name = k[index].name;
  if (what) *what = k[index].what;
  return n;
}
int pgo_row_shard_cuts(long long n_poses, long long n_edges, const int* id_begin, const int* id_end, int world, long long* cut, int* rows_per_out) {
  if (n_poses < 0 || n_edges < 0 || world <= 0 || !cut || (n_edges > 0 && (!id_begin || !id_end)))
    return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_row_shard_cuts");
  std::vector<long long> pre((size_t)n_poses + 1, 0);
  for (long long e = 0; e < n_edges; ++e) {
    if (id_begin[e] < 0 || id_begin[e] >= n_poses || id_end[e] < 0 || id_end[e] >= n_poses) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_row_shard_cuts: edge %lld names a pose that does not exist", e);
    ++pre[(size_t)id_begin[e] + 1]; ++pre[(size_t)id_end[e] + 1];
  }
  for (long long v = 0; v < n_poses; ++v) pre[(size_t)v + 1] += pre[(size_t)v] + 1;      // pre[v] = slots of the poses before v
  const long long total = pre[(size_t)n_poses];
  cut[0] = 0;
  for (int r = 1; r < world; ++r) {
    const long long want = total * r / world;
    long long c = std::lower_bound(pre.begin(), pre.end(), want) - pre.begin();       // first pose count whose slots reach the target
    c = (c + 2) / 4 * 4;                                                               // the nearest multiple of 4
    cut[r] = std::min(n_poses, std::max(cut[r - 1], c));
  }
  cut[world] = n_poses;
  long long longest = 0;
  for (int r = 0; r < world; ++r) longest = std::max(longest, cut[r + 1] - cut[r]);
  if (rows_per_out) *rows_per_out = (int)std::max<long long>(4, (longest + 3) / 4 * 4);
  return PGO_OK;
}

int pgo_comm_get_unique_id(unsigned char id[128]) {
  if (!id) return set_error(PGO_ERR_INVALID_ARGUMENT, "null id");
  const char* what = "";
  if (pgo::rccl_unique_id(id, &what) != 0) return set_error(PGO_ERR_HIP, "ncclGetUniqueId failed: %s", what);
  return PGO_OK;
}

static int attach_comm(pgo_problem* P, pgo::Comm* c) {
  delete P->comm;
  P->comm = c;
  P->topo_dirty = true;   // ownership changes the slot topology
  return PGO_OK;
}

int pgo_comm_init(pgo_problem* P, const unsigned char id[128], int rank, int world) {
  if (!P || !id || world < 1 || rank < 0 || rank >= world) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_comm_init");
  int rc = ensure_device(P);
  if (rc) return rc;
  const char* what = "";
  pgo::Comm* c = pgo::make_rccl_comm(id, rank, world, &what);
  if (!c) return set_error(PGO_ERR_HIP, "ncclCommInitRank failed: %s", what);
  return attach_comm(P, c);
}

int pgo_comm_init_ipc(pgo_problem* P, const char* name, int rank, int world) {
  if (!P || !name || name[0] != '/' || world < 1 || rank < 0 || rank >= world) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_comm_init_ipc (name: a POSIX shared-memory name, \"/...\")");
  int rc = ensure_device(P);
  if (rc) return rc;
  const char* what = "";
  pgo::Comm* c = pgo::make_ipc_comm(name, rank, world, &what);
  if (!c) return set_error(PGO_ERR_HIP, "the IPC group could not be formed: %s", what);
  return attach_comm(P, c);
}

// development hook (not part of include/pgo.h): stress the attached transport, returns mismatching words
int pgo_debug_comm_stress(pgo_problem* P, int iters, int seg_doubles) {
  if (!P || !P->comm) return -1;
  if (ensure_device(P)) return -1;
  int bad = -1;
  if (pgo::comm_stress(P->comm, iters, (size_t)seg_doubles, P->stream, &bad) != 0) return -2;
  return bad;
}

void* pgo_loopback_create(int world) { return world >= 1 ? new (std::nothrow) pgo::LoopbackGroup(world) : nullptr; }
void pgo_loopback_destroy(void* group) { delete static_cast<pgo::LoopbackGroup*>(group); }
int pgo_comm_init_loopback(pgo_problem* P, void* group, int rank) {
  pgo::LoopbackGroup* g = static_cast<pgo::LoopbackGroup*>(group);
  if (!P || !g || rank < 0 || rank >= g->world) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_comm_init_loopback");
  return attach_comm(P, pgo::make_loopback_comm(g, rank));
}

}  // extern "C"
