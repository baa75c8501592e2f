// pgo_batch.cpp — host side, part 4: pgo_solve_batch, n independent graphs as the components of one block-diagonal problem.
#include "pgo_internal.h"

// ---- batched solve of independent graphs -----------------------------------------------------------------------------------
// KITTI-scale graphs do not fill the machine: one LM iteration is a chain of ~45 small dependent launches.  n independent
// problems are therefore solved as ONE block-diagonal problem — the same launch sequence, n times the work per launch — with
// everything Levenberg-Marquardt decides kept per component: trust-region radius, accept / reject, every termination test,
// iteration records and summaries.  Device: damping with the radius of the pose's component, the factorisation of the union
// (a forest: nothing crosses components), per-component step scalars, acceptance by component.  Host: lm_pre_step /
// lm_post_step per component — the very functions the single-problem driver runs, so a component follows the trace it
// follows when solved alone (to the rounding of the differently grouped sums).  Exact steps only (SPARSE_NORMAL_CHOLESKY,
// the reference's setting): a per-component CG would need per-component iteration control.
int solve_batch(pgo_problem* const* probs, int n, const pgo_solver_options* options, pgo_solver_summary* summaries,
                pgo_iteration_record* records, int capacity) {
  const auto t_begin = Clock::now();
  const pgo_solver_options& o = *options;
  if (o.linear_solver_type != PGO_SPARSE_NORMAL_CHOLESKY)
    return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch serves exact requests (PGO_SPARSE_NORMAL_CHOLESKY) only");
  if (!probs || n <= 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: no problems");
  for (int c = 0; c < n; ++c) {
    if (!probs[c] || probs[c]->pp.empty()) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: problem %d is null or empty", c);
    if (probs[c]->device != probs[0]->device)
      return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: problem %d lives on device %d, problem 0 on device %d (one batch = one GPU)", c,
                       probs[c]->device, probs[0]->device);
  }
  // ---- the union ----
  pgo_problem M;
  M.device = probs[0]->device;
  M.loss_kind = probs[0]->loss_kind;
  M.loss_a = probs[0]->loss_a;
  std::vector<int> pose_begin(n + 1, 0), edge_begin(n + 1, 0);
  bool any_info = false;
  for (int c = 0; c < n; ++c) {
    const pgo_problem* Q = probs[c];
    if (!Q || Q->pp.empty()) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: problem %d is null or empty", c);
    if (Q->comm) return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: problem %d is attached to a communicator", c);
    for (uint8_t f : Q->is_point) if (f) return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: problem %d has point blocks (pose graphs only)", c);
    if (Q->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: problem %d is inside a solver session", c);
    if (Q->loss_kind != M.loss_kind || Q->loss_a != M.loss_a)
      return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: all problems must use the same loss function (problem %d differs)", c);
    pose_begin[c + 1] = pose_begin[c] + (int)Q->pp.size();
    edge_begin[c + 1] = edge_begin[c] + (int)Q->ia.size();
    any_info = any_info || Q->has_info;
  }
  const int N = pose_begin[n], E = edge_begin[n];
  static const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto mark = [&](const char* what) {
    if (verbose) std::fprintf(stderr, "[pgo] batch: %-34s at %.2f ms\n", what, 1e3 * seconds_since(t_begin));
  };
  M.pp.reserve(N); M.qq.reserve(N); M.cmask.reserve(N);
  M.ia.reserve(E); M.ib.reserve(E); M.meas.reserve((size_t)7 * E);
  if (any_info) M.sqrt_info.reserve((size_t)36 * E);
  M.has_info = any_info;
  for (int c = 0; c < n; ++c) {
    const pgo_problem* Q = probs[c];
    M.pp.insert(M.pp.end(), Q->pp.begin(), Q->pp.end());
    M.qq.insert(M.qq.end(), Q->qq.begin(), Q->qq.end());
    M.cmask.insert(M.cmask.end(), Q->cmask.begin(), Q->cmask.end());
    for (int v : Q->ia) M.ia.push_back(v + pose_begin[c]);
    for (int v : Q->ib) M.ib.push_back(v + pose_begin[c]);
    M.meas.insert(M.meas.end(), Q->meas.begin(), Q->meas.end());
    if (any_info) {
      if (Q->has_info) M.sqrt_info.insert(M.sqrt_info.end(), Q->sqrt_info.begin(), Q->sqrt_info.end());
      else
        for (size_t e = 0; e < Q->ia.size(); ++e)
          for (int k = 0; k < 36; ++k) M.sqrt_info.push_back(k % 7 == 0 ? 1.0 : 0.0);
    }
  }
  pgo_problem* P = &M;
  mark("union built");
  P->no_sfront = true;
  P->want_direct = true;       // the union's host analysis runs beside the array fills and uploads of prepare()
  int rc = prepare(P);
  P->want_direct = false;
  if (rc) return rc;
  mark("prepare");
  P->opt = o;
  P->g.loss_kind = P->loss_kind;
  P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p;
  P->g.pose_c = P->d_pose_c.p;
  rc = prepare_direct(P);
  if (rc) return rc;
  mark("prepare_direct");
  if (!P->direct_usable || P->dsym.hybrid)
    return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: the union of the problems is beyond the factorisation's budget; solve them one by one");
  P->split_two_launch = true;   // the device-wide failure flag is not consulted per component: no in-kernel waits in a batch
  P->sfront_levels = true;
  P->front_launches = true;
  rc = prepare_clusters(P, 1);
  if (rc) return rc;
  mark("prepare_clusters");
  hipStream_t s = P->stream;
  // component tables (device) and the per-component hand-over block (pinned, device visible)
  std::vector<int> pose_comp(N);
  for (int c = 0; c < n; ++c) std::fill(pose_comp.begin() + pose_begin[c], pose_comp.begin() + pose_begin[c + 1], c);
  DevBuf<int> d_pose_begin, d_edge_begin, d_pose_comp;
  HIP_TRY(d_pose_begin.upload(pose_begin, s));
  HIP_TRY(d_edge_begin.upload(edge_begin, s));
  HIP_TRY(d_pose_comp.upload(pose_comp, s));
  struct Pinned {
    void* p = nullptr;
    size_t cap = 0;
    hipStream_t s = nullptr;
    ~Pinned() {
      if (!p) return;
      (void)hipStreamSynchronize(s);          // an error return may leave kernels that write the block in flight
      host_side_pool().put_pinned(p, cap);
    }
  } pin;
  pin.s = s;
  const size_t pin_bytes = (size_t)n * (sizeof(pgo::BatchScalars) + sizeof(double) + sizeof(int)) + 64;
  HIP_TRY(host_side_pool().get_pinned(pin_bytes, &pin.p, &pin.cap));
  memset(pin.p, 0, pin_bytes);
  pgo::BatchScalars* out = static_cast<pgo::BatchScalars*>(pin.p);
  double* radius = reinterpret_cast<double*>(out + n);
  int* accept = reinterpret_cast<int*>(radius + n);
  // workgroups per component of the scalar reduction: enough of them to cover the machine, no more than the largest component needs
  int max_items = 1;
  for (int c = 0; c < n; ++c) max_items = std::max(max_items, std::max(pose_begin[c + 1] - pose_begin[c], edge_begin[c + 1] - edge_begin[c]));
  const int split = std::max(1, std::min((max_items + 255) / 256, std::max(1, 1024 / n)));
  DevBuf<double> d_partial;
  HIP_TRY(d_partial.alloc((size_t)5 * n * split));
  const pgo::BatchPlan plan{n, d_pose_begin.p, d_edge_begin.p, d_pose_comp.p, radius, accept, out, d_partial.p, split};

  mark("component tables");
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(P->g.pose_c, P->g.pose_x, P->d_pose_x.n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(P->d_flags.zero(s));
  HIP_TRY(P->d_cg_x.zero(s));
  HIP_TRY(P->d_cg_q.zero(s));
  HIP_TRY(P->d_cg_b.zero(s));
  HIP_TRY(P->d_d2.zero(s));
  mark("poses uploaded");
  const double t_setup = seconds_since(t_begin);
  // Init + IterationZero: cost, state norm and gradient norm of every component at its start (candidate == current point)
  rc = evaluate_gradient_and_jacobian(P, true);
  if (rc) return rc;
  pgo::launch_batch_scalars(P->g, plan, s);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  mark("iteration zero");
  std::vector<LmState> Ls(n);
  for (int c = 0; c < n; ++c) {
    LmState& L = Ls[c];
    L.x_cost = out[c].cand_cost;
    L.initial_cost = L.x_cost;
    L.x_norm = std::sqrt(out[c].x_norm_sq);
    L.gmax = out[c].gradient_max;
    L.radius = o.initial_trust_region_radius;
    L.cur = pgo_iteration_record{};
    L.cur.step_is_successful = 1;
    L.cur.cost = L.x_cost;
    L.cur.gradient_max_norm = L.gmax;
    L.pending_record = true;
    L.active = true;
    L.t_setup = t_setup;
    if (!std::isfinite(L.x_cost)) terminate(L, PGO_FAILURE, 7, "Initial cost is not finite.");
    radius[c] = L.radius;
  }
  int n_rounds = 0;
  for (;;) {
    int alive = 0;
    for (int c = 0; c < n; ++c) {
      LmState& L = Ls[c];
      if (L.terminated) continue;
      if (lm_pre_step(L, o)) { ++alive; radius[c] = L.radius; }
    }
    if (!alive) break;
    // the trial step of every component at once
    pgo::launch_batch_d2(P->g, plan, o.min_lm_diagonal, o.max_lm_diagonal, s);
    rc = damping_all(P, 1.0, o.min_lm_diagonal, o.max_lm_diagonal, 2);
    if (rc) return rc;
    rc = run_direct(P);
    if (rc) return rc;
    const pgo::CgParams none{0.0, -1.0, 0, 0};
    pgo::launch_spmv_tail(P->g, none, s, 0, 1);
    pgo::launch_batch_scalars(P->g, plan, s);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    ++n_rounds;
    bool any_accept = false;
    for (int c = 0; c < n; ++c) {
      LmState& L = Ls[c];
      accept[c] = 0;
      if (L.terminated) continue;
      // a pivot that fails inside one component leaves NaNs in that component's step only (nothing crosses components):
      // its model change is not finite and the step is handled as invalid; the device-wide flag is not consulted
      const StepScalars sc{out[c].cand_cost, out[c].model_change, out[c].step_norm_sq, out[c].x_norm_sq, out[c].gradient_max, 0, 0, 0};
      ++L.n_factorizations;
      if (lm_post_step(L, o, sc, 0) == STEP_ACCEPT) { accept[c] = 1; any_accept = true; }
    }
    if (any_accept) {
      pgo::launch_batch_accept(P->g, plan, s);
      rc = evaluate_gradient_and_jacobian(P, false);   // every component: the unchanged ones reproduce their values bit for bit
      if (rc) return rc;
    }
  }
  // gradient norm of the points accepted last (deferred like in the single-problem driver)
  bool need_g = false;
  for (int c = 0; c < n; ++c) need_g = need_g || Ls[c].gmax_deferred;
  if (need_g) {
    HIP_TRY(hipMemcpyAsync(P->g.pose_c, P->g.pose_x, P->d_pose_x.n * sizeof(double), hipMemcpyDeviceToDevice, s));
    pgo::launch_batch_scalars(P->g, plan, s);
    HIP_TRY(hipStreamSynchronize(s));
  }
  mark("rounds done");
  rc = download_poses(P, P->g.pose_x);
  if (rc) return rc;
  mark("poses downloaded");
  HIP_TRY(hipMemsetAsync(P->d_flags.p, 0, P->d_flags.n * sizeof(int), s));
  const double t_total = seconds_since(t_begin);
  for (int c = 0; c < n; ++c) {
    LmState& L = Ls[c];
    if (L.gmax_deferred) {
      L.gmax = out[c].gradient_max;
      L.cur.gradient_max_norm = L.gmax;
      if (!L.pending_record && !L.records.empty()) L.records.back().gradient_max_norm = L.gmax;
      L.gmax_deferred = false;
    }
    if (L.pending_record) {
      if (L.cur.step_is_successful) ++L.num_successful; else ++L.num_unsuccessful;
      L.cur.trust_region_radius = L.radius;
      L.records.push_back(L.cur);
      L.pending_record = false;
    }
    if (summaries) {
      pgo_solver_summary* sm = summaries + c;
      memset(sm, 0, sizeof *sm);
      sm->termination_type = L.termination;
      sm->reason = L.reason;
      sm->num_successful_steps = L.num_successful;
      sm->num_unsuccessful_steps = L.num_unsuccessful;
      sm->num_iterations = (int)L.records.size();
      sm->num_poses = pose_begin[c + 1] - pose_begin[c];
      sm->num_edges = edge_begin[c + 1] - edge_begin[c];
      sm->linear_solver_used = 0;
      // the factorisation is the union's: fill, levels and flops are those of all components together
      const bool fronts = P->front_usable || P->sfront_usable;
      sm->factor_nnz_blocks = fronts ? (int)std::min<long long>(P->fsym.factor_blocks, 0x7fffffff) : P->dsym.nb;
      sm->factor_levels = fronts ? P->fsym.n_levels : P->dsym.n_levels;
      int const_p = 0, const_q = 0;
      for (int v = pose_begin[c]; v < pose_begin[c + 1]; ++v) { const_p += P->cmask[v] & 1; const_q += (P->cmask[v] >> 1) & 1; }
      sm->num_parameter_blocks_reduced = 2 * sm->num_poses - const_p - const_q;
      sm->num_parameters_reduced = 7 * sm->num_poses - 3 * const_p - 4 * const_q;
      sm->num_effective_parameters_reduced = 6 * sm->num_poses - 3 * const_p - 3 * const_q;
      sm->factor_kind = P->sfront_usable ? 3 : P->front_usable ? 2 : 1;
      sm->factor_max_front = fronts ? P->fsym.max_front : 0;
      sm->factor_flops = fronts ? P->fsym.flops : P->dsym.flops;
      sm->num_factorizations = L.n_factorizations;
      sm->initial_cost = L.initial_cost;
      sm->final_cost = L.x_cost;
      sm->total_time_in_seconds = t_total;        // of the whole batch
      sm->setup_time_in_seconds = t_setup;
      sm->final_gradient_max_norm = L.gmax;
      sm->final_trust_region_radius = L.radius;
      snprintf(sm->message, sizeof sm->message, "%s", L.message.c_str());
    }
    if (records) {
      const int k = std::min(capacity, (int)L.records.size());
      for (int i = 0; i < k; ++i) records[(size_t)c * capacity + i] = L.records[i];
    }
  }
  if (getenv("PGO_VERBOSE"))
    std::fprintf(stderr, "[pgo] batch: %d problems, %d poses, %d edges, %d rounds, setup %.2f ms, total %.2f ms\n", n, N, E, n_rounds, 1e3 * t_setup, 1e3 * t_total);
  HIP_TRY(hipStreamSynchronize(s));   // the component tables below go back to the pool
  return PGO_OK;
}

// =================================================================================================
// C ABI (include/pgo.h)
// =================================================================================================
extern "C" {

int pgo_solve_batch(pgo_problem* const* problems, int n_problems, const pgo_solver_options* options, pgo_solver_summary* summaries,
                    pgo_iteration_record* records, int capacity) {
  if (!problems || n_problems <= 0 || !options || (records && capacity < 0)) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_solve_batch");
  const auto t0 = Clock::now();
  const int rc = solve_batch(problems, n_problems, options, summaries, records, capacity);
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] batch: call returned after %.2f ms (the union's device memory released)\n", 1e3 * seconds_since(t0));
  return rc;
}

}  // extern "C"
