// pgo_lm.cpp — host side, part 3: ceres::Solve (finial.cpp:531-544) — the Levenberg-Marquardt trust-region loop of SURVEY.md Appendix A.6
// in its three drivers: host in the loop (several ranks, hybrid exact requests), kernel sequences enqueued ahead of device-side
// decisions (exact steps), and the universal stream (PCG on one rank).  All arithmetic runs in the kernels; the rules are pgo_lm_rules.h.
#include "pgo_internal.h"

// ---- LM driver ------------------------------------------------------------------------------

int evaluate_gradient_and_jacobian(pgo_problem* P, bool first) {
  const auto t0 = Clock::now();
  hipStream_t s = P->stream;
  // several ranks, owner-only CG ahead (whatever block size the preconditioner ends up with: the standard form exchanges the
  // inverses its owners build): only the diagonals of the diagonal blocks travel
  const char* pe = getenv("PGO_SHARD_PIPE");
  const bool diag_only = P->g.world > 1 && !(pe && pe[0] == '0') && !P->use_graph &&
                         pgo::pipe_supported(P->g, cg_params_for(P->opt), P->opt.pcg_cluster_poses == 2 ? 2 : 1) && P->opt.pcg_cluster_poses != 4;
  if (first) {
    int rc = fill_scale_one(P);
    if (rc) return rc;
    rc = linearize_all(P, diag_only);
    if (rc) return rc;
    if (P->opt.jacobi_scaling) {
      pgo::launch_scale_from_diag(P->g, s);
      rc = linearize_all(P, diag_only);
      if (rc) return rc;
    }
  } else {
    int rc = linearize_all(P, diag_only);
    if (rc) return rc;
  }
  pgo::launch_gradient_norm(P->g, s);
  P->lm.t_jacobian += seconds_since(t0);
  return PGO_OK;
}

static bool pipeline_wanted(const pgo_problem* P);
static bool universal_wanted(const pgo_problem* P);

// The resident CG (pgo_uni_resident.h) meets at a grid barrier: every work-group of its launch has to be on the chip at once, which
// two such launches sharing a device cannot both count on.  ONE session per device holds the right to run it; the others take the fused
// stream (whose launches need no co-residency and simply share the chip with it).
static std::atomic<int> g_resident_busy[64];
static bool resident_slot_acquire(pgo_problem* P) {
  if (P->resident_slot) return true;
  int expected = 0;
  if (P->device < 0 || P->device >= 64 || !g_resident_busy[P->device].compare_exchange_strong(expected, 1)) return false;
  P->resident_slot = true;
  return true;
}
void resident_slot_release(pgo_problem* P) {
  if (!P->resident_slot) return;
  g_resident_busy[P->device].store(0);
  P->resident_slot = false;
}
int lm_begin(pgo_problem* P, const pgo_solver_options* options) {
  P->want_direct = options->linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY;
  {   // a session with a coarse level runs the one-launch CG iteration on the incidence slots, which wants whole pose pairs in a work-group:
      // work-groups of 256 slots (a topology built otherwise is rebuilt)
    const bool want_coarse = options->linear_solver_type == PGO_BLOCK_JACOBI_PCG && options->pcg_coarse_aggregate >= 8;
    const int fb = want_coarse && !getenv("PGO_BLOCK") ? 256 : 0;
    if (fb != P->force_block) { P->force_block = fb; P->topo_dirty = true; }
  }
  int rc = prepare(P);
  P->want_direct = false;
  if (rc) return rc;
  rc = peer_direct_setup(P);
  if (rc) return rc;
  if (getenv("PGO_VERBOSE") && P->g.world > 1)
    std::fprintf(stderr, "[pgo] rank %d of %d: pairs_whole %d, exchange %s\n", P->g.rank, P->g.world, P->g.pairs_whole, P->g.peer_tab ? "by the kernels (peer table)" : "all-gather");
  P->opt = *options;
  LmState& L = P->lm;
  const double t_setup = L.t_setup;
  L = LmState();
  L.t_setup = t_setup;
  const auto t0 = Clock::now();
  P->g.loss_kind = P->loss_kind;
  P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p;
  P->g.pose_c = P->d_pose_c.p;
  static const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(P->d_pose_0.p, P->g.pose_x, P->d_pose_0.n * sizeof(double), hipMemcpyDeviceToDevice, P->stream));
  HIP_TRY(P->d_flags.zero(P->stream));
  // Init + IterationZero, enqueued before the factorisation's plan is waited for (its host analysis runs on a helper thread
  // since prepare(), its uploads queue up behind these kernels)
  P->g.cluster = 1;
  rc = evaluate_gradient_and_jacobian(P, true);
  if (rc) return rc;
  pgo::launch_cost(P->g, P->g.pose_x, 0, P->stream);
  // x_norm: run the retraction with a zero step (delta is zero after prepare/reset)
  HIP_TRY(P->d_delta.zero(P->stream));
  pgo::launch_apply_step(P->g, P->g.delta, P->stream);
  pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, P->stream);
  int cluster = P->opt.pcg_cluster_poses;
  // coarse level of the PCG (pgo_coarse.h): one rank, truncated PCG; on top of the 2-pose cluster Jacobi
  P->coarse_on = P->opt.linear_solver_type == PGO_BLOCK_JACOBI_PCG && P->opt.pcg_coarse_aggregate >= 8;
  if (P->opt.pcg_coarse_aggregate != 0 && !P->coarse_on)
    return set_error(PGO_ERR_UNSUPPORTED, "pcg_coarse_aggregate = %d: the coarse level needs BLOCK_JACOBI_PCG and aggregates of at least 8 poses", P->opt.pcg_coarse_aggregate);
  if (P->coarse_on) {
    cluster = 2;
    pgo::CoarsePlan& c = P->coarse;
    c.agg = P->opt.pcg_coarse_aggregate;
    // several ranks: aggregates never straddle ranks — every rank's segment of rows_per poses gets the same number of aggregate slots
    c.per_rank = ((P->g.world > 1 ? P->g.rows_per : P->g.N) + c.agg - 1) / c.agg;
    c.n_agg = P->g.world * c.per_rank;
    c.a_lo = P->g.rank * c.per_rank; c.a_hi = c.a_lo + c.per_rank;
    if (P->g.peer_tab) { P->g.peer_tab = nullptr; P->g.peer_flags = nullptr; P->peer_dirty = true; }     // (the correction sits between a CG launch and its exchange: host-enqueued exchange)
    c.cdim = 6 * c.n_agg;
    { const int pb = pgo::coarse_pivot_block(); c.npad = (c.cdim + pb - 1) / pb * pb; }
    if ((size_t)(6 * c.npad + 64 * 37) * sizeof(double) > 160 * 1024 - 2048)
      return set_error(PGO_ERR_UNSUPPORTED, "pcg_coarse_aggregate = %d gives %d aggregates: more than the coarse level's row panel holds (use larger aggregates)", c.agg, c.n_agg);
    HIP_TRY(P->dc_Pt.alloc((size_t)36 * P->g.N));
    HIP_TRY(P->dc_Ac.alloc((size_t)c.npad * c.npad));
    HIP_TRY(P->dc_Ac2.alloc((size_t)c.npad * c.npad));
    HIP_TRY(P->dc_rc.alloc((size_t)c.npad));
    HIP_TRY(P->dc_rc.zero(P->stream));
    c.Pt = P->dc_Pt.p; c.Ac = P->dc_Ac.p; c.Ac2 = P->dc_Ac2.p; c.rc = P->dc_rc.p;
    c.Ainv = ((c.npad / pgo::coarse_pivot_block()) & 1) ? c.Ac2 : c.Ac;
    c.rank_end = nullptr;
    if (P->g.world > 1) {
      std::vector<int> re((size_t)P->g.world);
      for (int k = 0; k < P->g.world; ++k) re[(size_t)k] = k * P->g.rows_per + (int)(P->shard_cut[(size_t)k + 1] - P->shard_cut[(size_t)k]);
      HIP_TRY(P->dc_rank_end.upload(re, P->stream));
      c.rank_end = P->dc_rank_end.p;
    }
    if (!P->g.pairs_whole) {        // (never silently the block Jacobi alone: the correction is applied between the launches of k_pipe_cg, which needs whole pairs)
      P->coarse_on = false;
      return set_error(PGO_ERR_UNSUPPORTED, "pcg_coarse_aggregate: a pose pair of this graph has more than 256 incidence slots — the one-launch CG iteration the coarse level rides on cannot hold it in one work-group");
    }
  }
  // the boundary exchange serves the host-enqueued transports; where the kernels store into every rank's buffers themselves (peer table)
  // and where a coarse correction is applied to the full-layout buffer between launch and exchange the whole segments travel as before
  {
    const bool bx = P->g.world > 1 && P->bx_ready && !P->g.peer_tab && !P->coarse_on && pgo::tuning("shard_boundary", 1.0) != 0.0;
    P->g.bx[0] = bx ? P->d_bx0.p : nullptr; P->g.bx[1] = bx ? P->d_bx1.p : nullptr;
    P->g.bx_brow = P->d_bx_brow.p; P->g.bx_slot_off = P->d_bx_slot_off.p; P->g.bx_nb = P->bx_nb; P->g.bx_cseg = P->bx_cseg;
  }
  if (P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY) {
    const auto t_sym = Clock::now();
    rc = prepare_direct(P);
    if (rc) return rc;
    L.t_setup += seconds_since(t_sym);
    // exact request served by PCG to exact_r_tolerance: the preconditioner is ours to choose — 2-pose chain clusters need
    // ~2.5x fewer iterations than 6x6 blocks at almost the same cost per iteration
    if ((!P->direct_usable || P->dsym.hybrid) && cluster < 2) cluster = 2;
  }
  rc = prepare_clusters(P, cluster);
  if (rc) return rc;
  if (verbose) std::fprintf(stderr, "[pgo] lm_begin: plan + clusters prepared    %.2f ms\n", 1e3 * seconds_since(t0));
  HIP_TRY(hipStreamSynchronize(P->stream));
  HIP_TRY(hipGetLastError());
  if (verbose) std::fprintf(stderr, "[pgo] lm_begin: iteration zero evaluated    %.2f ms\n", 1e3 * seconds_since(t0));
  L.x_cost = P->scal->cand_cost;
  L.initial_cost = L.x_cost;
  L.x_norm = std::sqrt(P->scal->x_norm_sq);
  L.initial_x_norm = L.x_norm;
  L.gmax = P->scal->gradient_max;
  L.radius = P->opt.initial_trust_region_radius;
  L.decrease_factor = 2.0;
  L.reuse_diagonal = false;
  L.cur = pgo_iteration_record{};
  L.cur.iteration = 0;
  L.cur.step_is_successful = 1;
  L.cur.cost = L.x_cost;
  L.cur.gradient_max_norm = L.gmax;
  L.pending_record = true;
  P->pipelined = pipeline_wanted(P);
  P->universal = universal_wanted(P);
  // ... in its fused form (one launch per CG iteration, pipelined recurrences) unless the caller asks for the standard CG or the
  // request is one the fused kernel does not serve (pgo_kernels.hip uni_f_supported)
  // (its recurrences carry no residual refresh: left to itself the library takes them for forcing terms eta >= 0.01 — CG runs of
  // tens of iterations — and keeps Ceres' refreshed CG for tighter ones, whose runs of hundreds of iterations are where a
  // pipelined CG loses attainable accuracy; pcg_form 2 asks for them regardless)
  const bool fused_asked = P->opt.pcg_form == 2 || (P->opt.pcg_form == 0 && P->opt.eta >= 1e-2);
  P->uni_fused = P->universal && fused_asked && pgo::uni_f_supported(P->g, cg_params_for(P->opt), P->g.cluster);
  // ... or in its resident form (the whole CG one launch, blocks and vectors in registers, a grid barrier per iteration): the library's
  // choice where the grid fits the chip and this session gets the device's one resident slot; pcg_form 3 asks for it (and gets the fused
  // form where it cannot be had), pcg_form 2 keeps the fused form
  const bool res_asked = P->opt.pcg_form == 3 || (P->opt.pcg_form == 0 && P->opt.eta >= 1e-2);
  P->uni_resident = P->universal && res_asked && pgo::uni_r_supported(P->g, cg_params_for(P->opt), P->g.cluster) && resident_slot_acquire(P);
  if (!P->uni_resident) resident_slot_release(P);
  if (P->opt.pcg_form == 3 && !P->uni_resident) P->uni_fused = P->universal && pgo::uni_f_supported(P->g, cg_params_for(P->opt), P->g.cluster);
  if (P->uni_resident) P->uni_fused = true;
  P->uni_host_launches = 0; P->uni_host_enqueue_s = 0.0;
  P->resident_aborts = 0;
  P->pipe_dirty = true;
  // symmetric tile form for the CG products: host-driven PCG of a large graph on one rank (pgo_sym.h)
  P->sym_active = false; P->sym_storage = false;
  // ... when the solve can run long enough to repay the form's construction (44 ms of host time at 100 k / 1 M against 0.7 ms saved
  // per LM iteration of ~13 CG iterations: 64 iterations, or a tight forcing term whose CG runs are long); PGO_SYM=1 forces it
  // ... on one rank when the solve can run long enough to repay the form's construction: 44 ms of host time at 100 k / 1 M against
  // 0.9 ms saved per LM iteration (2.7 -> 1.8), i.e. from ~50 iterations on — max_num_iterations >= 64, a tight forcing term (long CG
  // runs), or a form that exists already.  (r06 tried "always": BASELINE configs[3] with 25 iterations went from 242 to 282 ms of wall,
  // profiles/r06_a_config_table_sym_always.md.)  Several ranks: always — a rank builds the form of ITS rows (5 ms at world 8).  PGO_SYM=1 forces it.
  const char* sym_env = getenv("PGO_SYM");
  const bool sym_pays = P->g.world > 1 || (sym_env && sym_env[0] == '1') || P->sym_ready || P->opt.max_num_iterations >= 64 || P->opt.eta <= 0.02;
  if (P->opt.linear_solver_type == PGO_BLOCK_JACOBI_PCG && !P->universal && !P->pipelined && sym_pays && sym_wanted(P)) {
    rc = sym_prepare(P);
    if (rc) return rc;
    P->sym_active = P->sym_ready;
    const bool rp = pgo::tuning("sym_repack", 0.0) != 0.0;        // (A/B knob, pgo_tuning.h: keep the incidence-slot linearisation and copy its blocks per LM iteration)
    if (P->sym_active && !rp) { rc = sym_enter_storage(P); if (rc) return rc; }
  }
  L.active = true;            // (only a session that got this far is one: an error above leaves the problem as it was)
  L.t_total += seconds_since(t0);
  if (!std::isfinite(L.x_cost)) {
    L.terminated = true; L.termination = PGO_FAILURE; L.reason = 7;
    L.message = "Initial cost is not finite.";
  }
  return PGO_OK;
}

void terminate(LmState& L, int termination, int reason, const char* fmt, ...) {
  char buf[240];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  L.terminated = true;
  L.termination = termination;
  L.reason = reason;
  L.message = buf;
}

// ---- speculative linearisation -----------------------------------------------------------------------------------------
// Between the step tail and the re-linearisation of an accepted point the GPU used to wait for the host (hand-off, decision,
// launch: 11-16 us of the 130-300 us LM iteration of Manhattan 10 k).  The candidate is therefore linearised at once, behind the
// tail, into a spare set of buffers; the host decides meanwhile.  Accepted (the usual case): the sets are swapped, nothing
// is recomputed.  Rejected: the current set was never touched.  Same kernel, same inputs: results are bit-identical.
// One rank (the exchange of the diagonal blocks would have to ride along), eager enqueue (captured batches hold pointers).
int ensure_spec_buffers(pgo_problem* P) {
  if (P->spec_ready) return PGO_OK;
  hipStream_t s = P->stream;
  HIP_TRY(P->d_bsr2.alloc(P->d_bsr.n));
  HIP_TRY(P->d_bsr2.zero(s));
  HIP_TRY(P->d_Hdiag2.alloc(P->d_Hdiag.n));
  HIP_TRY(P->d_Hdiag2.zero(s));
  HIP_TRY(P->d_grad2.alloc(P->d_grad.n));
  HIP_TRY(P->d_grad2.zero(s));
  P->spec_ready = true;
  return PGO_OK;
}
bool speculation_on(const pgo_problem* P) {
  return P->g.world == 1 && !P->use_graph && !(P->comm && P->comm->world > 1) && !P->sym_storage;   // (the symmetric form has one set of blocks)
}
struct SpareSet { double *bsr, *Hdiag, *grad; };
inline SpareSet spare_set(pgo_problem* P) {
  const bool primary_in_use = P->g.bsr_val == P->d_bsr.p;
  return primary_in_use ? SpareSet{P->d_bsr2.p, P->d_Hdiag2.p, P->d_grad2.p} : SpareSet{P->d_bsr.p, P->d_Hdiag.p, P->d_grad.p};
}
void launch_speculative_linearize(pgo_problem* P, int gate) {
  pgo::DeviceGraph gs = P->g;
  const SpareSet sp = spare_set(P);
  gs.pose_x = P->g.pose_c;
  gs.bsr_val = sp.bsr; gs.Hdiag = sp.Hdiag; gs.grad = sp.grad;
  pgo::launch_linearize(gs, P->stream, gate);
}

// ---- the host half of one TrustRegionMinimizer pass (SURVEY.md A.6 step 7 order), shared by the single-problem driver and
// the batched one (one LmState per component there): pure bookkeeping on LmState, no device work ----
// FinalizeIterationAndCheckIfMinimizerCanContinue.  Returns false when the minimizer stops here.
bool lm_pre_step(LmState& L, const pgo_solver_options& o) {
  if (L.pending_record) {
    if (L.cur.step_is_successful) ++L.num_successful; else ++L.num_unsuccessful;
    L.cur.trust_region_radius = L.radius;
    L.records.push_back(L.cur);
    L.pending_record = false;
  }
  if (L.cur.iteration >= o.max_num_iterations) {
    terminate(L, PGO_NO_CONVERGENCE, 5, "Maximum number of iterations reached. Number of iterations: %d.", L.cur.iteration);
    return false;
  }
  if (!L.gmax_deferred && L.cur.step_is_successful && L.cur.gradient_max_norm <= o.gradient_tolerance) {
    terminate(L, PGO_CONVERGENCE, 3, "Gradient tolerance reached. Gradient max norm: %e <= %e", L.cur.gradient_max_norm, o.gradient_tolerance);
    return false;
  }
  if (L.radius <= o.min_trust_region_radius) {
    terminate(L, PGO_CONVERGENCE, 4, "Minimum trust region radius reached. Trust region radius: %e <= %e", L.radius, o.min_trust_region_radius);
    return false;
  }
  return true;
}

// Everything after the trial step came back: deferred gradient test, then the rules of pgo_lm_rules.h (step validity,
// parameter / function tolerance, IsStepSuccessful, radius update) — the very function the device applies when it decides
// itself.  STEP_ACCEPT: the caller makes the candidate the current point and re-linearises.
pgo::LmTolerances lm_tolerances(const pgo_solver_options& o) {
  return pgo::LmTolerances{o.min_relative_decrease, o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance,
                           o.max_trust_region_radius, o.min_trust_region_radius, o.max_num_iterations, o.max_num_consecutive_invalid_steps};
}
inline pgo_iteration_record to_record(const pgo::LmRecord& r) { pgo_iteration_record o; memcpy(&o, &r, sizeof o); return o; }
void terminate_by_reason(LmState& L, const pgo_solver_options& o, int termination, int reason, double value) {
  switch (reason) {
    case 1: terminate(L, termination, 1, "Function tolerance reached. |cost_change|/cost: %e <= %e", value, o.function_tolerance); break;
    case 2: terminate(L, termination, 2, "Parameter tolerance reached. Relative step_norm: %e <= %e.", value, o.parameter_tolerance); break;
    case 3: terminate(L, termination, 3, "Gradient tolerance reached. Gradient max norm: %e <= %e", value, o.gradient_tolerance); break;
    case 4: terminate(L, termination, 4, "Minimum trust region radius reached. Trust region radius: %e <= %e", value, o.min_trust_region_radius); break;
    case 5: terminate(L, termination, 5, "Maximum number of iterations reached. Number of iterations: %d.", (int)value); break;
    case 6: terminate(L, termination, 6, "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d",
                      o.max_num_consecutive_invalid_steps); break;
    default: terminate(L, termination, reason, "Terminated (reason %d).", reason); break;
  }
}
StepAction lm_post_step(LmState& L, const pgo_solver_options& o, const StepScalars& sc, int extra_linear_iterations) {
  ++L.num_trial_steps;
  const int cg_it = sc.cg_iterations;
  L.num_linear_iterations += cg_it + extra_linear_iterations;   // the iterations of an over-budget PCG try are work done, counted in the summary
  if (L.gmax_deferred) {
    // the gradient test of FinalizeIterationAndCheckIfMinimizerCanContinue for the point accepted last
    // iteration: if it fires, the step just computed is discarded (x was not touched)
    L.gmax_deferred = false;
    L.gmax = sc.gradient_max;
    L.cur.gradient_max_norm = L.gmax;
    if (!L.records.empty()) L.records.back().gradient_max_norm = L.gmax;
    if (L.gmax <= o.gradient_tolerance) {
      L.num_linear_iterations -= cg_it;
      L.reuse_diagonal = true;
      terminate_by_reason(L, o, PGO_CONVERGENCE, 3, L.gmax);
      return STEP_NONE;
    }
  }
  pgo::LmCore C{L.radius, L.decrease_factor, L.x_cost, L.x_norm, L.cur.gradient_max_norm, L.cur.iteration, L.reuse_diagonal ? 1 : 0,
                L.num_consecutive_invalid, 0};
  const pgo::LmStepIn in{sc.cand_cost, sc.model_change, sc.step_norm_sq, sc.x_norm_sq, cg_it, sc.cg_status, sc.linearize_bad, 0};
  pgo::LmRecord nx{};
  double value = 0.0;
  const pgo::LmOutcome out = pgo::lm_decide(C, lm_tolerances(o), in, nx, value);
  L.radius = C.radius; L.decrease_factor = C.decrease_factor; L.x_cost = C.x_cost; L.x_norm = C.x_norm;
  L.reuse_diagonal = C.reuse_diagonal != 0; L.num_consecutive_invalid = C.num_consecutive_invalid;
  switch (out) {
    case pgo::LM_OUT_INVALID_FAIL:
      terminate_by_reason(L, o, PGO_FAILURE, 6, 0.0);
      L.cur = to_record(nx);
      return STEP_NONE;
    case pgo::LM_OUT_INVALID:
      L.cur = to_record(nx);
      L.pending_record = true;
      return STEP_NONE;
    case pgo::LM_OUT_PARAM_TOL: terminate_by_reason(L, o, PGO_CONVERGENCE, 2, value); return STEP_NONE;
    case pgo::LM_OUT_FUNC_TOL: terminate_by_reason(L, o, PGO_CONVERGENCE, 1, value); return STEP_NONE;
    case pgo::LM_OUT_ACCEPT:
      L.gmax_deferred = true;  // filled in at the next host sync (or at the end of the solve)
      L.cur = to_record(nx);
      L.pending_record = true;
      return STEP_ACCEPT;
    default:
      L.cur = to_record(nx);
      L.pending_record = true;
      return STEP_REJECT;
  }
}

// One pass of the TrustRegionMinimizer loop body (SURVEY.md A.6 step 7 order).
int lm_advance(pgo_problem* P) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  hipStream_t s = P->stream;
  if (L.terminated) return PGO_OK;
  const auto t_it = Clock::now();

  if (!lm_pre_step(L, o)) return PGO_OK;

  // ComputeTrustRegionStep + ComputeCandidatePointAndEvaluateCost, enqueued back to back: damping, one
  // batch of CG iterations, model cost change / delta / candidate, candidate cost, scalar fold.  ONE host
  // sync per LM iteration in the common case; if the CG batch was too short, further batches follow and
  // the (cheap) tail is re-enqueued.
  const auto t_lin = Clock::now();
  const pgo::CgParams prm = cg_params_for(o);
  int rc = damping_all(P, L.radius, o.min_lm_diagonal, o.max_lm_diagonal, L.reuse_diagonal ? 1 : 0);
  if (rc) return rc;
  // coarse level: P~ at the current point, the Galerkin matrix of the damped system, its inverse (pgo_coarse.h)
  if (P->coarse_on) { rc = coarse_setup(P); if (rc) return rc; }
  const bool direct = o.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  // Both available (hybrid): PCG gets the budget of ~1.5 factorisations, in CG iterations priced by the same deterministic
  // cost model that admitted the factorisation (0.7 us per schedule step; 10 us + 42 ps per slot per CG iteration), so the
  // choice depends on iteration counts only, never on a clock.  Over budget = redo the iteration with the factorisation.
  const bool hybrid = direct && P->dsym.hybrid;
  const double budget_knob = pgo::tuning("hybrid_budget", -1.0);     // experiments / tests: CG iterations a PCG try may take
  const int cg_budget = !hybrid ? 0 : budget_knob >= 0.0 ? std::max(1, (int)budget_knob)
                                : std::max(50, (int)(1.5 * 0.7 * P->dsym.est_steps / (10.0 + 4.2e-5 * (double)P->g.n_slots)));
  bool use_direct = direct;
  if (hybrid && (L.hybrid_pcg || L.hybrid_direct_run >= L.hybrid_probe_after ||
                 (L.hybrid_direct_run >= 1 && L.hybrid_fail_radius > 0.0 && L.radius < 0.1 * L.hybrid_fail_radius)))
    use_direct = false;
  int wasted_cg = 0;
  const bool spec = speculation_on(P);
  bool spec_in_flight = false;   // the candidate's linearisation was enqueued behind the tail that produced the scalars read below
  if (spec) { rc = ensure_spec_buffers(P); if (rc) return rc; }
  arm_handoff(P);
  if (!use_direct) {
    // every batch carries the gated tail: the host hears back once per batch and finds the step scalars ready
    // as soon as the CG has stopped
    pgo::CgParams run_prm = prm;
    if (hybrid) run_prm.max_iterations = cg_budget;
    rc = pcg_begin(P, run_prm);
    if (rc) return rc;
    for (int round = 0, enqueued = 0;; ++round) {
      const int nb = pick_batch(run_prm, o.cg_batch, round, enqueued, P->last_cg_iterations);
      rc = launch_cg_batch(P, run_prm, nb, true, enqueued + 1);
      enqueued += nb;
      if (rc) return rc;
      // the speculative launch costs an early-exit launch (~3.5 us) in a batch the CG does not finish in and saves the
      // host gap (~12 us) in the one it does: skipped in a first batch shorter than the previous solve's iteration count
      spec_in_flight = spec && (round > 0 || P->last_cg_iterations <= nb);
      if (spec_in_flight) launch_speculative_linearize(P, 1);
      rc = wait_handoff(P);
      if (rc) return rc;
      if (P->scal->cg_status != -1) break;
      spec_in_flight = false;      // gated out: the CG was still running
      arm_handoff(P);
    }
    if (hybrid) {
      if (P->scal->cg_iterations >= cg_budget && P->scal->cg_status == 0) {   // not converged within the budget
        wasted_cg = P->scal->cg_iterations;
        L.hybrid_probe_after = L.hybrid_pcg ? 2 : std::min(16, 2 * L.hybrid_probe_after);
        L.hybrid_pcg = false;
        L.hybrid_direct_run = 0;
        L.hybrid_fail_radius = L.radius;
        ++L.hybrid_pcg_over;
        use_direct = true;
        arm_handoff(P);
      } else {
        L.hybrid_pcg = true;
        L.hybrid_probe_after = 1;
        ++L.hybrid_pcg_ok;
      }
    }
  }
  if (use_direct) {
    P->scal->cg_status = 0;       // host-visible block: the CG kernels that normally fill these do not run
    P->scal->cg_iterations = 0;
    rc = run_direct(P);
    if (rc) return rc;
    rc = enqueue_tail(P, nullptr);
    if (rc) return rc;
    spec_in_flight = spec;
    if (spec) launch_speculative_linearize(P, 0);
    rc = wait_handoff(P);
    if (rc) return rc;
    if ((P->scal->linearize_bad & 4) && (P->front_usable ? !P->front_launches : P->sfront_usable ? !P->sfront_levels : !P->split_two_launch)) {
      // a single-launch SPLIT step waited in vain for a column's diagonal block (its workgroups were not all resident), or a
      // front of the single-launch small-front factorisation for a child: not a numerical failure — repeat this factorisation
      // in the form without in-kernel waits and keep to it
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: an in-kernel wait of the single-launch factorisation timed out; one launch per step from now on\n");
      arm_handoff(P);
      rc = run_direct(P);
      if (rc) return rc;
      rc = enqueue_tail(P, nullptr);
      if (rc) return rc;
      if (spec) launch_speculative_linearize(P, 0);
      rc = wait_handoff(P);
      if (rc) return rc;
    }
    ++L.n_factorizations;
    if (hybrid) { ++L.hybrid_direct_run; ++L.hybrid_direct; }
  }
  HIP_TRY(hipGetLastError());
  const pgo::LmScalars dsc = *P->scal;
  if (getenv("PGO_VERBOSE") && dsc.cg_status != 0) std::fprintf(stderr, "[pgo] rank %d iteration %d: CG ended with status %d after %d iterations\n", P->g.rank, L.cur.iteration + 1, dsc.cg_status, dsc.cg_iterations);
  P->last_cg_iterations = dsc.cg_iterations;
  L.t_linear += seconds_since(t_lin);
  const StepScalars sc{dsc.cand_cost, dsc.model_change, dsc.step_norm_sq, dsc.x_norm_sq, dsc.gradient_max,
                       dsc.cg_iterations, dsc.cg_status, dsc.linearize_bad};
  const StepAction action = lm_post_step(L, o, sc, wasted_cg);
  if (action == STEP_ACCEPT) {
    // HandleSuccessfulStep (device half): x <- candidate, re-linearise
    std::swap(P->g.pose_x, P->g.pose_c);
    P->sym_stale = true;
    if (spec_in_flight) {   // the candidate was linearised behind the tail: its set becomes the current one
      const SpareSet sp = spare_set(P);
      P->g.bsr_val = sp.bsr; P->g.Hdiag = sp.Hdiag; P->g.grad = sp.grad;
      pgo::launch_gradient_norm(P->g, s);
    } else {
      rc = evaluate_gradient_and_jacobian(P, false);
      if (rc) return rc;
    }
  }
  L.t_total += seconds_since(t_it);
  return PGO_OK;
}

// ---- device-resident LM: the host enqueues sequences ahead of the decisions (pgo_kernels.h LmDev) -----------------------------
// r02 ended every LM iteration in a hand-off: the device folded the step scalars, the host decided accept / reject, updated the
// radius and enqueued the next iteration, the GPU idle meanwhile (13 us on the development box, ~60 us on the driver's: 16 % of
// the Manhattan 10 k step; half of a KITTI-00 iteration was not GPU work).  Now the last work-group of the step tail applies the
// rules of pgo_lm_rules.h itself, the kernels read radius / reuse-diagonal / "was the step accepted" from device memory, and the
// host's only job is to keep the stream fed: it enqueues the sequence of the NEXT iteration while the current one runs, and reads
// the iteration records afterwards.  What it cannot know when it enqueues — whether the CG will be through within the batch it
// allots, whether the step will be accepted, whether the solve ends — the kernels find out for themselves (LmDev::phase /
// accepted / halt): a wrong guess costs early-exit launches (~2.6 us each), never a wrong result.
static bool pipeline_wanted(const pgo_problem* P) {
  const bool off = getenv("PGO_NO_PIPELINE") && getenv("PGO_NO_PIPELINE")[0] == '1';   // (read per solve: the tests compare both drivers in one process)
  if (off || P->g.world != 1 || (P->comm && P->comm->world > 1) || P->use_graph || P->coarse_on) return false;
  const bool direct = P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  if (direct && P->dsym.hybrid && !P->front_usable && !P->sfront_usable) return false;   // factorisation or PCG chosen per iteration by the host
  // Exact steps: the launch sequence of an iteration is the same every time, so enqueueing ahead costs nothing.  PCG: the host has
  // to allot CG iterations to a sequence before it knows how many the CG will take (Manhattan 10 k: 32, 8, 20, 85, 15, 125, 14 ...
  // then 3-6), every unused one is two early-exit launches and every CG that outlives its sequence a gated tail: measured 0.38 ms
  // per LM iteration against 0.30 with the host in the loop on a host that answers within 13 us.  the knob pipeline_pcg = 1 (pgo_tuning.h) selects the
  // sequences for PCG as well (tests do).
  if (!direct) return pgo::tuning("pipeline_pcg", 0.0) != 0.0;
  return true;
}

// host LmState -> device (begin / reset; the stream is idle or the copy is ordered behind what is in flight)
int lm_upload_state(pgo_problem* P) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  if (P->d_lm.n == 0) HIP_TRY(P->d_lm.alloc(1));
  pgo::LmDev D{};
  D.halt = pgo::LM_RUN; D.phase = pgo::LM_PHASE_NEW; D.accepted = 0;
  D.lm_done = 0;
  D.num_successful = L.num_successful; D.num_unsuccessful = L.num_unsuccessful; D.num_linear_iterations = L.num_linear_iterations;
  D.num_records = L.cur.iteration + 1;       // record index = iteration number (the ring slot of record r is r % LM_RING)
  P->pipe_pulled = L.cur.iteration + 1;
  const pgo::CgParams prm = cg_params_for(o);
  D.cg_period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;   // (launch_cg_batch: an exact request served by PCG never refreshes)
  D.last_cg = P->last_cg_iterations;
  D.core = pgo::LmCore{L.radius, L.decrease_factor, L.x_cost, L.x_norm, L.gmax, L.cur.iteration, L.reuse_diagonal ? 1 : 0, L.num_consecutive_invalid, 0};
  D.tol = lm_tolerances(o);
  D.min_diag = o.min_lm_diagonal; D.max_diag = o.max_lm_diagonal;
  HIP_TRY(hipStreamSynchronize(P->stream));      // nobody writes the pinned block while the host fills it
  if (P->universal) {
    HIP_TRY(P->d_cg.zero(P->stream));            // the stream's operation words and its launch counter (the budget kernel opens it)
    HIP_TRY(P->d_flags.zero(P->stream));
    P->uni_enq = 0;
    P->scal->slots_done = 0;
    P->scal->resident_abort = 0;
    // (diagnostic, tests/test_gpu_resident.py: the abort word set by hand — the cycle's kernels find the barrier "given up" before the
    // first CG and the session has to carry on with the fused stream, as it would behind a real time-out)
    if (P->uni_resident && pgo::tuning("resident_abort_test", 0.0) != 0.0)
      HIP_TRY(hipMemsetAsync(P->d_flags.p + pgo::uni_r_abort_word(), 1, sizeof(int), P->stream));
  }
  P->scal->lm = D;
  P->scal->lm_done = 0; P->scal->halt = 0; P->scal->last_cg = D.last_cg;
  P->scal->seq_done = P->pipe_seq;
  HIP_TRY(hipMemcpyAsync(P->d_lm.p, &P->scal->lm, sizeof(pgo::LmDev), hipMemcpyHostToDevice, P->stream));
  HIP_TRY(hipStreamSynchronize(P->stream));      // (the pinned source doubles as the mirror the device writes)
  P->pipe_t_linear0 = L.t_linear; P->pipe_t_jacobian0 = L.t_jacobian;
  return PGO_OK;
}

// records the device has finished with -> LmState (a record is final once a later one exists, or once everything has drained)
void lm_pull_records(pgo_problem* P, bool drained) {
  LmState& L = P->lm;
  const int have = __atomic_load_n(&P->scal->lm.num_records, __ATOMIC_ACQUIRE);
  const int upto = drained ? have : have - 1;
  for (; P->pipe_pulled < upto; ++P->pipe_pulled) L.records.push_back(to_record(P->scal->ring[P->pipe_pulled % pgo::LM_RING]));
}

// device -> host LmState, everything drained
void lm_pull_state(pgo_problem* P) {
  LmState& L = P->lm;
  lm_pull_records(P, true);
  const pgo::LmDev& M = P->scal->lm;
  L.radius = M.core.radius; L.decrease_factor = M.core.decrease_factor; L.x_cost = M.core.x_cost; L.x_norm = M.core.x_norm;
  L.gmax = M.core.gmax; L.reuse_diagonal = M.core.reuse_diagonal != 0; L.num_consecutive_invalid = M.core.num_consecutive_invalid;
  L.num_successful = M.num_successful; L.num_unsuccessful = M.num_unsuccessful; L.num_linear_iterations = M.num_linear_iterations;
  if (!L.records.empty()) L.cur = L.records.back();
  L.cur.iteration = M.core.iteration;
  L.pending_record = false;
  L.gmax_deferred = false;
  L.t_linear = P->pipe_t_linear0 + 1e-8 * (double)M.ticks_linear;       // s_memrealtime: 100 MHz
  L.t_jacobian = P->pipe_t_jacobian0 + 1e-8 * (double)M.ticks_jacobian;
  P->last_cg_iterations = M.last_cg;
}

// CG iterations allotted to a sequence.  The CG stops by itself; an iteration enqueued past its end costs two early-exit
// launches (~5 us), a CG that outlives its sequence goes on in the next one at the price of that sequence's skipped head and
// gated tail (~18 us) — provided a multiple of the refresh period has been completed (the refresh launches sit at fixed
// positions), else the host steps in (~60 us).  Short CGs (the steady state of an LM run: 3-6 iterations) get the last count
// + 2; long ones a multiple of the period.  cont_streak: sequences that just ended without a decision.
int pipe_pick_batch(const pgo_problem* P, const pgo::CgParams& prm, int period, int pred, int cont_streak) {
  int nb;
  if (P->opt.cg_batch > 0) nb = P->opt.cg_batch;
  else {
    if (pred <= 0) pred = 6;
    if (pred <= 7 && cont_streak == 0) nb = pred + 2;
    else {
      const int want = std::max(pred + pred / 4 + 1, 8) << std::min(cont_streak, 3);
      const int per2 = (period & 1) ? 2 * period : period;     // sequences end on EVEN multiples of the period (lm_cg_unfinished)
      nb = period > 0 ? (want + per2 - 1) / per2 * per2 : want;
      nb = std::min(nb, period > 0 ? std::max(per2, 60 / per2 * per2) : 64);
    }
  }
  nb = std::max(1, std::min(nb, prm.max_iterations));
  return (nb + 1) & ~1;
}

// one sequence: the kernels of one prospective LM iteration (or the continuation of the previous one's CG)
int enqueue_sequence(pgo_problem* P, const pgo::DeviceGraph& gp, const pgo::DeviceGraph& gl, const pgo::CgParams& prm, bool direct, int nb,
                     int start_it, bool head) {
  hipStream_t s = P->stream;
  const pgo_solver_options& o = P->opt;
  if (head) pgo::launch_damping(gp, 1.0, o.min_lm_diagonal, o.max_lm_diagonal, 0, s);   // radius and mode: LmDev
  if (direct) {
    int rc = run_direct(P, gp);
    if (rc) return rc;
    const pgo::CgParams none{0.0, -1.0, 0, 0};
    pgo::launch_spmv_tail(gp, none, s, 0, 1);
    pgo::launch_step_tail(gp, s, 2);
  } else {
    if (head) pgo::launch_pcg_init(gp, s);
    const int period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;
    for (int i = 0; i < nb; ++i) {
      const bool refresh = period > 0 && ((start_it + i) % period) == 0;
      int rc = cg_iteration(P, gp, prm, ((start_it + i) & 1), refresh);
      if (rc) return rc;
    }
    pgo::launch_spmv_tail(gp, prm, s, 1, 1);
    pgo::launch_step_tail(gp, s, 1);
  }
  pgo::launch_linearize(gl, s, 2);
  pgo::launch_accept_finish(gp, ++P->pipe_seq, s);
  P->pipe_last_nb = nb;
  return PGO_OK;
}

// waits until every enqueued sequence is through (pinned counter; the stream synchronise is the fallback for long waits)
int pipe_drain(pgo_problem* P) {
  const auto t0 = Clock::now();
  for (unsigned spins = 1; __atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE) != P->pipe_seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0x3ff) == 0 && seconds_since(t0) > 0.002) {
      HIP_TRY(hipStreamSynchronize(P->stream));
      if (__atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE) != P->pipe_seq)
        return set_error(PGO_ERR_HIP, "the enqueued LM sequences did not report completion (%d of %d)", P->scal->seq_done, P->pipe_seq);
    }
  }
  HIP_TRY(hipGetLastError());
  return PGO_OK;
}

// ---- the universal stream (pgo_kernels.h UniOp): PCG on one rank ---------------------------------------------------------------
// The host enqueues  V S V S ...  and nothing else; what each launch does is the device's business.  It keeps between `lo` and
// `hi` pairs ahead of the device's launch counter — enough that the GPU never waits for a launch (a pair is ~13 us of work), few
// enough that the launches left over when the stream stops (terminated, or the step budget of pgo_solver_step used up) drain in
// well under 0.1 ms.
static bool universal_wanted(const pgo_problem* P) {
  const bool off = getenv("PGO_NO_PIPELINE") && getenv("PGO_NO_PIPELINE")[0] == '1';
  const char* u = getenv("PGO_UNI");
  if (off || (u && u[0] == '0') || P->g.world != 1 || (P->comm && P->comm->world > 1) || P->use_graph || P->coarse_on) return false;
  const bool direct = P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  if (direct || !pgo::uni_supported(P->g)) return false;
  // large graphs (kernels of 100+ us) gain nothing from it and the slot kernel's LDS footprint (the linearisation's) would cost the
  // SpMV occupancy there: they keep the host-driven loop
  const long long limit = 600000;
  return (u && u[0] == '1') || P->g.n_slots <= limit;
}

int lm_run_universal(pgo_problem* P, int budget, int* ran) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  hipStream_t s = P->stream;
  if (ran) *ran = 0;
  if (L.terminated || budget == 0) return PGO_OK;
  const auto t_run = Clock::now();
  if (!lm_pre_step(L, o)) { L.t_total += seconds_since(t_run); return PGO_OK; }
  if (P->pipe_dirty) {
    int rc0 = lm_upload_state(P);
    if (rc0) return rc0;
    P->pipe_dirty = false;
  }
  // (an LM iteration is at least two pairs, records are pulled once per turn of the loop below: at most hi / 2 decisions between
  // two pulls, which has to stay below the LM_RING - 2 record slots in flight)
  // (fused form: the unit is one launch, an LM iteration is at least four)
  const bool fused = P->uni_fused;
  const int hi = fused ? 24 : 12;
  const int lo = std::max(1, hi / 3);
  const pgo::CgParams prm = cg_params_for(o);
  const int period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;
  pgo::DeviceGraph gp = P->g;
  gp.lm = P->d_lm.p;
  const int d0 = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
  P->scal->halt = 0;
  pgo::launch_lm_budget(gp, budget, s, P->uni_enq);
  unsigned idle_spins = 0;
  auto t_idle = Clock::now();
  for (;;) {
    if (__atomic_load_n(&P->scal->halt, __ATOMIC_ACQUIRE)) break;
    if (P->uni_resident && __atomic_load_n(&P->scal->resident_abort, __ATOMIC_ACQUIRE)) {
      // a grid barrier of the resident CG gave up (its grid was not all on the chip: somebody else's kernels hold CUs).  The device put
      // the LM iteration back to its HEAD and the cycle's kernels only pass the state on from here: the rest of the session runs the
      // fused stream, whose one symbol does whatever the state names at any launch number.
      P->uni_resident = false;
      resident_slot_release(P);
      ++P->resident_aborts;
      if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] resident CG gave up at a grid barrier (launch %d): continuing with the fused stream\n", P->uni_enq);
    }
    const int pending = P->uni_enq - __atomic_load_n(&P->scal->slots_done, __ATOMIC_ACQUIRE);
    if (pending <= hi - lo) {
      const auto t_enq = Clock::now();
      for (int i = 0; i < lo; ++i) {
        if (P->uni_resident) pgo::launch_uni_r(gp, prm, P->uni_enq, o.min_lm_diagonal, o.max_lm_diagonal, s);
        else if (fused) pgo::launch_uni_f(gp, prm, P->uni_enq, o.min_lm_diagonal, o.max_lm_diagonal, s);
        else {
          pgo::launch_uni_v(gp, prm, o.min_lm_diagonal, o.max_lm_diagonal, s);
          pgo::launch_uni_s(gp, prm, period, s);
        }
        ++P->uni_enq;
      }
      P->uni_host_enqueue_s += seconds_since(t_enq);
      P->uni_host_launches += fused ? lo : 2 * lo;
      lm_pull_records(P, false);
      idle_spins = 0; t_idle = Clock::now();
      continue;
    }
    __builtin_ia32_pause();
    if ((++idle_spins & 0xfff) == 0 && seconds_since(t_idle) > 0.5) {
      HIP_TRY(hipStreamSynchronize(s));
      if (P->uni_enq != __atomic_load_n(&P->scal->slots_done, __ATOMIC_ACQUIRE))
        return set_error(PGO_ERR_HIP, "the universal LM stream stopped reporting progress (%d of %d launches)", P->scal->slots_done, P->uni_enq);
      t_idle = Clock::now();
    }
  }
  // whatever was enqueued behind the halt exits at once; one more launch tells the host when the stream has drained
  arm_handoff(P);
  pgo::launch_lm_publish(gp, s);
  int rc = wait_handoff(P);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  const int d1 = P->scal->lm_done;
  lm_pull_state(P);
  L.num_trial_steps += d1 - d0;
  if (ran) *ran = d1 - d0;
  const pgo::LmDev& M = P->scal->lm;
  if (M.halt == pgo::LM_HALT_TERMINATED) terminate_by_reason(L, o, M.termination, M.reason, M.term_value);
  L.t_total += seconds_since(t_run);
  return PGO_OK;
}

// Runs up to `budget` LM iterations (decisions; < 0: until the solve terminates).  *ran = iterations executed.
int lm_run_pipelined(pgo_problem* P, int budget, int* ran) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  if (ran) *ran = 0;
  if (L.terminated || budget == 0) return PGO_OK;
  const auto t_run = Clock::now();
  // the opening tests of the first pass (they push the iteration-0 record; the device applies them from then on)
  if (!lm_pre_step(L, o)) { L.t_total += seconds_since(t_run); return PGO_OK; }
  if (P->pipe_dirty) {
    int rc0 = lm_upload_state(P);
    if (rc0) return rc0;
    P->pipe_dirty = false;
  }
  // (one decision per sequence at most; the iteration records of the sequences in flight share the LM_RING pinned slots)
  static const int lookahead = 1;
  const bool direct = o.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  const pgo::CgParams prm = cg_params_for(o);
  const int period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;
  pgo::DeviceGraph gp = P->g;
  gp.lm = P->d_lm.p;
  pgo::DeviceGraph gl = gp;
  gl.pose_x = gp.pose_c;          // the linearisation of an accepted step reads the candidate buffer
  pgo::launch_lm_resume(gp, -1, P->stream);    // time mark of the device's phase clocks
  const int d0 = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
  const long long target = budget < 0 ? (1LL << 40) : (long long)d0 + budget;
  int cont_streak = 0, seen_seq = P->pipe_seq, seen_done = d0;
  int rc = PGO_OK;
  unsigned idle_spins = 0;
  auto t_idle = Clock::now();
  for (;;) {
    // seq_done first: decisions of sequences counted as in flight may already be in lm_done, never the other way round
    const int sdone = __atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE);
    const int d = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
    const int halt = __atomic_load_n(&P->scal->halt, __ATOMIC_ACQUIRE);
    const int in_flight = P->pipe_seq - sdone;
    if (sdone != seen_seq) {        // sequences that ended without a decision: their CG goes on
      cont_streak = (d == seen_done) ? cont_streak + (sdone - seen_seq) : 0;
      seen_seq = sdone; seen_done = d;
      lm_pull_records(P, false);
    }
    if (halt) {
      rc = pipe_drain(P);
      if (rc) break;
      const int h = P->scal->lm.halt;
      if (h == pgo::LM_HALT_TERMINATED) break;
      rc = resync_direct_counters(P);
      if (rc) break;
      P->scal->halt = 0;
      seen_seq = P->pipe_seq; seen_done = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
      if (h == pgo::LM_HALT_CG_STALL) {
        // the rest of this iteration's CG, from where it stands, up to the next multiple of the refresh period (from there the
        // sequences line up again), then the tail
        const int completed = P->scal->cg_iterations;
        int nb = pipe_pick_batch(P, prm, period, std::max(P->pipe_last_nb, 8), 1);
        // (an even multiple: the sequences behind this one are enqueued as iterations 1, 2, ... and rely on that parity)
        if (period > 0) { const int per2 = (period & 1) ? 2 * period : period; nb = ((completed + nb + per2 - 1) / per2) * per2 - completed; }
        else if ((completed + nb) & 1) ++nb;
        pgo::launch_lm_resume(gp, 1, P->stream);
        rc = enqueue_sequence(P, gp, gl, prm, false, nb, completed + 1, false);
        if (rc) break;
        cont_streak = 1;
        continue;
      }
      // LM_HALT_REFACTOR: a wait inside a single-launch factorisation ran out (its work-groups were not all resident): the
      // iteration is repeated with one launch per step, and the problem keeps to that form
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: an in-kernel wait of the single-launch factorisation timed out; one launch per step from now on\n");
      pgo::launch_lm_resume(gp, 0, P->stream);
      continue;
    }
    if (d >= target && in_flight == 0) break;
    if ((long long)d + in_flight < target && in_flight <= lookahead) {
      const int nb = direct ? 0 : pipe_pick_batch(P, prm, period, __atomic_load_n(&P->scal->last_cg, __ATOMIC_RELAXED), cont_streak);
      rc = enqueue_sequence(P, gp, gl, prm, direct, nb, 1, true);
      if (rc) break;
      idle_spins = 0; t_idle = Clock::now();
      continue;
    }
    __builtin_ia32_pause();
    if ((++idle_spins & 0xfff) == 0 && seconds_since(t_idle) > 0.002) {
      std::this_thread::sleep_for(std::chrono::microseconds(100));   // long iterations (sphere x10: 34 ms): no need to burn the core
      if (seconds_since(t_idle) > 20.0) {                            // watchdog: nothing enqueued for 20 s and still sequences in flight
        HIP_TRY(hipStreamSynchronize(P->stream));
        if (__atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE) == sdone && __atomic_load_n(&P->scal->halt, __ATOMIC_ACQUIRE) == 0 && in_flight > 0) {
          rc = set_error(PGO_ERR_HIP, "the enqueued LM sequences made no progress (%d of %d through)", sdone, P->pipe_seq);
          break;
        }
        t_idle = Clock::now();
      }
    }
  }
  if (rc == PGO_OK) rc = pipe_drain(P);
  if (rc) return rc;
  const int d1 = P->scal->lm_done;
  lm_pull_state(P);
  L.num_trial_steps += d1 - d0;
  if (direct) L.n_factorizations += d1 - d0;
  if (ran) *ran = d1 - d0;
  const pgo::LmDev& M = P->scal->lm;
  if (M.halt == pgo::LM_HALT_TERMINATED) {
    terminate_by_reason(L, o, M.termination, M.reason, M.term_value);
    rc = resync_direct_counters(P);     // sequences enqueued ahead of the halt left their tickets untouched
    if (rc) return rc;
  }
  L.t_total += seconds_since(t_run);
  return PGO_OK;
}

int lm_end(pgo_problem* P, pgo_solver_summary* summary, pgo_iteration_record* records, int capacity) {
  LmState& L = P->lm;
  if (!L.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_end without pgo_solver_begin");
  struct SymOff { pgo_problem* p; ~SymOff() { p->sym_active = false; p->sym_storage = false; } } sym_off{P};   // the session's storage choice ends with it
  if (L.gmax_deferred) {
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, P->stream);
    HIP_TRY(hipStreamSynchronize(P->stream));
    L.gmax = P->scal->gradient_max;
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
  if (!L.terminated) terminate(L, PGO_NO_CONVERGENCE, 5, "Stepping stopped by the caller after %d iterations.", L.cur.iteration);
  if (P->dsym.hybrid && P->direct_usable && getenv("PGO_VERBOSE"))
    std::fprintf(stderr, "[pgo] exact request, per-iteration choice: %d factorisations, %d PCG solves within budget, %d over budget (redone)\n",
                 L.hybrid_direct, L.hybrid_pcg_ok, L.hybrid_pcg_over);
  resident_slot_release(P);
  int rc = download_poses(P, P->g.pose_x);
  if (rc) return rc;
  if (P->g.oplog && !P->g.oplog_indexed) {      // profiling aid (PGO_UNI_OPLOG): "<s_memrealtime tick> <operation>" per k_uni_s launch of this session
    long long n = 0;
    HIP_TRY(hipMemcpy(&n, P->g.oplog, sizeof n, hipMemcpyDeviceToHost));
    std::vector<long long> h((size_t)n);
    if (n) HIP_TRY(hipMemcpy(h.data(), P->g.oplog + 1, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(P->g.oplog, 0, sizeof(long long)));
    const char* path = getenv("PGO_UNI_OPLOG");       // (may have been unset since prepare() read it)
    if (FILE* f = path ? std::fopen(path, "a") : nullptr) {
      for (long long v : h) std::fprintf(f, "%lld %d\n", v >> 3, (int)(v & 7) + (P->uni_resident ? 32 : P->uni_fused ? 16 : 0));    // (16 +: operations of k_uni_f; 32 +: of the resident stream's kernels, symbols of their own)
      std::fclose(f);
    }
  }
  if (summary) {
    memset(summary, 0, sizeof *summary);
    summary->cg_exchange = P->g.world > 1 ? (P->g.peer_tab && pipe_mode(P, cg_params_for(P->opt)) ? 2 : (P->g.bx[0] && pipe_mode(P, cg_params_for(P->opt))) ? 3 : 1) : 0;
    summary->cg_form = P->g.world > 1 ? (pipe_mode(P, cg_params_for(P->opt)) ? 2 : 1) : (P->uni_resident ? 4 : P->uni_fused ? 3 : pipe_mode(P, cg_params_for(P->opt)) ? 2 : 0);
    summary->sym_form = P->sym_storage ? 1 : 0;
    summary->coarse_level = P->coarse_on ? P->coarse.n_agg : 0;
    summary->termination_type = L.termination;
    summary->reason = L.reason;
    summary->num_successful_steps = L.num_successful;
    summary->num_unsuccessful_steps = L.num_unsuccessful;
    summary->num_iterations = (int)L.records.size();
    summary->num_linear_solver_iterations = L.num_linear_iterations;
    summary->num_poses = (int)P->pp.size();
    summary->num_edges = P->g.E;
    const bool want_exact = P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY;
    summary->linear_solver_used = want_exact ? (P->direct_usable ? (P->dsym.hybrid ? 3 : 0) : 2) : 1;
    const bool fronts = P->front_usable || P->sfront_usable;
    summary->factor_nnz_blocks = (want_exact && P->direct_usable) ? (fronts ? (int)std::min<long long>(P->fsym.factor_blocks, 0x7fffffff) : P->dsym.nb) : 0;
    summary->factor_levels = (want_exact && P->direct_usable) ? (fronts ? P->fsym.n_levels : P->dsym.n_levels) : 0;
    {
      int const_p = 0, const_q = 0;
      for (uint8_t m : P->cmask) { const_p += m & 1; const_q += (m >> 1) & 1; }
      const int n_caller = (int)P->pp.size();
      summary->num_parameter_blocks_reduced = 2 * n_caller - const_p - const_q;
      summary->num_parameters_reduced = 7 * n_caller - 3 * const_p - 4 * const_q;
      summary->num_effective_parameters_reduced = 6 * n_caller - 3 * const_p - 3 * const_q;
    }
    summary->factor_kind = (want_exact && P->direct_usable) ? (P->sfront_usable ? 3 : P->front_usable ? 2 : 1) : 0;
    summary->factor_max_front = (want_exact && fronts) ? P->fsym.max_front : 0;
    summary->factor_flops = (want_exact && P->direct_usable) ? (fronts ? P->fsym.flops : P->dsym.flops) : 0.0;
    summary->num_factorizations = L.n_factorizations;
    summary->initial_cost = L.initial_cost;
    summary->final_cost = L.x_cost;
    summary->total_time_in_seconds = L.t_total;
    summary->setup_time_in_seconds = L.t_setup;
    summary->linear_solver_time_in_seconds = L.t_linear;
    summary->jacobian_evaluation_time_in_seconds = L.t_jacobian;
    summary->residual_evaluation_time_in_seconds = L.t_residual;
    summary->final_gradient_max_norm = L.gmax;
    summary->final_trust_region_radius = L.radius;
    snprintf(summary->message, sizeof summary->message, "%s", L.message.c_str());
  }
  if (records) {
    const int n = std::min(capacity, (int)L.records.size());
    for (int i = 0; i < n; ++i) records[i] = L.records[i];
  }
  L.active = false;
  return PGO_OK;
}

// =================================================================================================
// C ABI (include/pgo.h)
// =================================================================================================
extern "C" {

int pgo_solver_begin(pgo_problem* P, const pgo_solver_options* options) {
  if (!P || !options) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_solver_begin");
  return lm_begin(P, options);
}

int pgo_solver_step(pgo_problem* P, int n, int* executed, int* done) {
  if (!P || !P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_step without pgo_solver_begin");
  int ran = 0;
  if (P->pipelined || P->universal) {
    int rc = P->universal ? lm_run_universal(P, n, &ran) : lm_run_pipelined(P, n, &ran);
    if (rc) { P->lm.active = false; return rc; }
    if (executed) *executed = ran;
    if (done) *done = P->lm.terminated ? 1 : 0;
    return PGO_OK;
  }
  for (int i = 0; i < n && !P->lm.terminated; ++i) {
    const int solves_before = P->lm.num_trial_steps;
    int rc = lm_advance(P);
    if (rc) { P->lm.active = false; return rc; }   // a failed session is closed: the device state is not trustworthy any more
    // an iteration counts when a trial step was computed (a pure termination check does not), also when that step ended the
    // run on the parameter or function tolerance
    if (P->lm.num_trial_steps != solves_before) ++ran;
  }
  if (executed) *executed = ran;
  if (done) *done = P->lm.terminated ? 1 : 0;
  return PGO_OK;
}

int pgo_solver_reset(pgo_problem* P) {
  if (!P || !P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_reset without pgo_solver_begin");
  LmState& L = P->lm;
  HIP_TRY(hipMemcpyAsync(P->g.pose_x, P->d_pose_0.p, P->d_pose_0.n * sizeof(double), hipMemcpyDeviceToDevice, P->stream));
  int rc = evaluate_gradient_and_jacobian(P, true);
  if (rc) return rc;
  pgo::launch_cost(P->g, P->g.pose_x, 0, P->stream);
  pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, P->stream);
  HIP_TRY(hipStreamSynchronize(P->stream));
  // the session restarts as pgo_solver_begin left it: state, counters and the iteration records (the steps taken so far leave no
  // trace — what pgo_solver_end reports is the solve from here on)
  L.x_cost = P->scal->cand_cost;
  L.x_norm = L.initial_x_norm;
  L.gmax = P->scal->gradient_max;
  L.radius = P->opt.initial_trust_region_radius;
  L.decrease_factor = 2.0;
  L.reuse_diagonal = false;
  L.terminated = false;
  L.gmax_deferred = false;
  L.num_consecutive_invalid = 0;
  L.num_successful = L.num_unsuccessful = L.num_linear_iterations = 0;
  L.num_trial_steps = 0;
  L.n_factorizations = 0;
  L.records.clear();
  pgo_iteration_record r{};
  r.iteration = 0;  // the iteration budget restarts with the state
  r.step_is_successful = 1;
  r.cost = L.x_cost;
  r.gradient_max_norm = L.gmax;
  L.cur = r;
  L.pending_record = true;
  P->pipe_dirty = true;
  return PGO_OK;
}

int pgo_solver_end(pgo_problem* P, pgo_solver_summary* summary, pgo_iteration_record* records, int capacity) {
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  return lm_end(P, summary, records, capacity);
}

int pgo_solve(pgo_problem* P, const pgo_solver_options* options, pgo_solver_summary* summary,
              pgo_iteration_record* records, int capacity) {
  if (!P || !options) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_solve");
  int rc = lm_begin(P, options);
  if (rc) { if (P->comm) P->comm->give_up(); return rc; }       // (peers of an in-process group must not wait for this rank in a collective)
  while (!P->lm.terminated) {
    rc = P->universal ? lm_run_universal(P, -1, nullptr) : P->pipelined ? lm_run_pipelined(P, -1, nullptr) : lm_advance(P);
    if (rc) { P->lm.active = false; if (P->comm) P->comm->give_up(); return rc; }   // caller memory keeps the poses it came with; the session is closed
  }
  return lm_end(P, summary, records, capacity);
}

}  // extern "C"
