// pgo_problem.cpp — host side of libpgo_hip.so, part 1: ceres::Problem-style bookkeeping (REF/test/pose_graph_ceres_plus_finial.cpp:491-528),
// the error channel, device set-up, the incidence-slot topology build and its uploads, pose transfers.
#include "pgo_internal.h"

thread_local std::string g_error;

int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}
const char* last_error_string() { return g_error.c_str(); }

// error channel for the other translation units of the library
int pgo_candidates_set_error(int code, const char* msg) { return set_error(code, "%s", msg); }


int ensure_device(pgo_problem* P) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return set_error(PGO_ERR_NO_DEVICE,
                     "no HIP device available: the pose-graph path runs on gfx950 only and has no CPU fallback");
  }
  HIP_TRY(hipSetDevice(P->device));
  if (!P->stream_ready) {
    HIP_TRY(host_side_pool().get_stream(P->device, &P->stream));
    P->stream_ready = true;
    // Launch sequences are enqueued eagerly by default: on this stack (ROCm 7.2, MI355X) the host runs ahead of the GPU and a
    // captured hipGraph of the same kernels is no faster (C2: 0.317 vs 0.317 ms per LM iteration without residual refreshes,
    // 0.322 eager vs 0.353 graph with them; KITTI-00 exact 0.79 vs 0.81 ms).  The knob graph = 1 (pgo_tuning.h) replays captured batches instead.
    P->use_graph = pgo::tuning("graph", 0.0) != 0.0;
  }
  if (!P->scal) {
    void* blk = nullptr;
    HIP_TRY(host_side_pool().get_pinned(sizeof(pgo::LmScalars), &blk, &P->scal_cap));
    P->scal = static_cast<pgo::LmScalars*>(blk);
    memset(P->scal, 0, sizeof(pgo::LmScalars));
  }
  return PGO_OK;
}

// In-place all-gather of equal segments (rank r owns buf[r*seg, (r+1)*seg)); no-op with a single rank.
int exchange(pgo_problem* P, double* buf, size_t seg_doubles) {
  static const bool force = getenv("PGO_FORCE_EXCHANGE") && getenv("PGO_FORCE_EXCHANGE")[0] == '1';   // exercise the transport at world 1
  if (!P->comm || (P->comm->world <= 1 && !force)) return PGO_OK;
  const char* what = "";
  if (P->comm->all_gather(buf, seg_doubles, P->stream, &what) != 0) return set_error(PGO_ERR_HIP, "all-gather failed: %s", what);
  return PGO_OK;
}

// linearise the owned rows, then make J'J diagonal blocks and J'r of ALL rows available on every rank
// diag_only (several ranks, the solve will run the owner-only CG): nobody reads another rank's off-diagonal entries of the
// diagonal blocks then — 6 doubles per pose travel instead of 36 (28.8 -> 4.8 MB per accepted step at 100 k poses)
int linearize_all(pgo_problem* P, bool diag_only) {
  if (P->sym_storage) {
    // two kernels write the form: the row kernel with the lean per-incidence algebra (k_linearize_lean, the default where it fits:
    // information without position / rotation coupling) and the row kernel with the general body and redirected block stores
    // (k_linearize_symout: information with the coupling; the knob sym_lin_rows = 1 runs it everywhere — tests/test_gpu_sym.py holds one to the other)
    const bool rows_writer = pgo::tuning("sym_lin_rows", 0.0) != 0.0;
    pgo::DeviceGraph gs = P->g;
    gs.sym_dst = P->sy_dst.p; gs.sym_val = P->sym.val;
    if (rows_writer) pgo::launch_linearize_symout(gs, P->stream);
    else pgo::launch_linearize_lean(gs, P->stream);
    if (P->g.world == 1) return PGO_OK;        // (one rank: nothing to exchange; several: the diagonal blocks' diagonals and the gradient, as below)
  } else {
    pgo::launch_linearize(P->g, P->stream);
    P->sym_stale = true;
  }
  int rc;
  // what THIS linearisation exchanges is what the CG start checks against its form (pcg_begin): recorded here, from the branch
  // actually taken, so that callers which linearise by themselves (pgo_linear_solve: a full exchange) are covered too
  P->lin_diag_only = (diag_only && P->comm && P->comm->world > 1 && P->d_pipe_x.p) ? 1 : 0;
  if (P->lin_diag_only) {
    pgo::launch_hdiag6(P->g, P->d_pipe_x.p, 0, P->stream);
    rc = exchange(P, P->d_pipe_x.p, (size_t)6 * P->g.rows_per);
    if (rc) return rc;
    pgo::launch_hdiag6(P->g, P->d_pipe_x.p, 1, P->stream);
  } else {
    rc = exchange(P, P->g.Hdiag, (size_t)36 * P->g.rows_per);
    if (rc) return rc;
  }
  return exchange(P, P->g.grad, (size_t)6 * P->g.rows_per);
}

// LM damping + preconditioner.  6x6 blocks are rebuilt on every rank from the gathered diagonal; cluster blocks need
// the in-cluster off-diagonal blocks, which only the owner holds, so their inverses are exchanged.
int damping_all(pgo_problem* P, double radius, double min_diag, double max_diag, int mode) {
  if (P->sym_storage) { pgo::launch_damping(sym_view(P), radius, min_diag, max_diag, mode, P->stream); return PGO_OK; }
  pgo::launch_damping(P->g, radius, min_diag, max_diag, mode, P->stream);
  if (P->g.cluster > 1 && !pipe_mode(P, cg_params_for(P->opt))) return exchange(P, P->g.Minv, (size_t)36 * P->g.cluster * P->g.rows_per);   // (the owner-only CG applies its own blocks only)
  return PGO_OK;
}

// one CG iteration: SpMV on the owned rows, exchange of q (+ p'q partials), replicated vector update
// `refresh`: this is a residual_reset_period-th iteration — r is recomputed as b - A x (Ceres conjugate_gradients_solver.cc)
// instead of updated: x-only update, A x into the exchange buffer, then r / z / partial sums.
int cg_iteration(pgo_problem* P, const pgo::DeviceGraph& g, const pgo::CgParams& prm, int odd, bool refresh) {
  if (P->sym_active) pgo::launch_spmv_sym(g, P->sym, prm, odd, 0, P->stream);    // one rank, large graph: every interior block read once
  else pgo::launch_pcg_spmv_only(g, prm, odd, P->stream);
  int rc = exchange(P, g.cg_q, (size_t)g.seg);
  if (rc) return rc;
  if (!refresh) {
    pgo::launch_pcg_update_only(g, odd, P->stream);
    return PGO_OK;
  }
  if (g.world == 1) {   // x = x_old + alpha p formed on the fly by the SpMV, one combined vector launch
    if (P->sym_storage) pgo::launch_spmv_sym(g, P->sym, prm, 1 | 8 | 16 | (odd ? 32 : 0), 1, P->stream);
    else pgo::launch_spmv_refresh(g, P->stream, 1, odd);
    pgo::launch_pcg_update_only(g, odd, P->stream, 3);
    return PGO_OK;
  }
  pgo::launch_pcg_update_only(g, odd, P->stream, 1);
  pgo::launch_spmv_refresh(g, P->stream);
  rc = exchange(P, g.cg_q, (size_t)g.seg);
  if (rc) return rc;
  pgo::launch_pcg_update_only(g, odd, P->stream, 2);
  return PGO_OK;
}
int cg_iteration(pgo_problem* P, const pgo::CgParams& prm, int odd, bool refresh) { return cg_iteration(P, P->g, prm, odd, refresh); }

int choose_block(long long total_slots) {
  const int env_block = getenv("PGO_BLOCK") ? atoi(getenv("PGO_BLOCK")) : 0;   // tuning experiments / tests: 64, 128 or 256 (read per call)
  if (env_block == 64 || env_block == 128 || env_block == 256) return env_block;
  // (r05, re-measured on this round's boxes at BASELINE configs[1], 90 k slots: 256-slot work-groups 0.226 / 0.220 ms per LM
  // iteration (two-kernel / fused stream) against 0.238 / 0.227 with 128 — half the work-groups to dispatch, half the partial sums)
  if (total_slots >= 256LL * 320) return 256;
  if (total_slots >= 128LL * 384) return 128;
  return 64;
}


// Builds the incidence-slot topology and uploads every static array.
int prepare(pgo_problem* P) {
  int rc = ensure_device(P);
  if (rc) return rc;
  if (!P->topo_dirty) return PGO_OK;
  if (P->analysis_thread.joinable()) P->analysis_thread.join();     // (of a topology that is being replaced)
  const auto t0 = Clock::now();
  const int N_caller = (int)P->pp.size(), E = (int)P->ia.size();
  if (N_caller == 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "problem has no poses");
  hipStream_t s = P->stream;
  P->drop_graph();

  const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto lap = [&, tl = Clock::now()](const char* what) mutable {
    if (verbose) std::fprintf(stderr, "[pgo] prepare: %-28s %.2f ms\n", what, 1e3 * seconds_since(tl));
    tl = Clock::now();
  };
  // ---- row ownership: rank r owns a contiguous share of the poses, cut where the incidence slots (1 + degree per pose)
  // balance (r06; by row count until r05: 1.18x the mean on the heaviest of 8 ranks at BASELINE configs[3]); cut edges are evaluated by
  // the owners of both endpoints.  The cuts are multiples of 4 so that preconditioner clusters never straddle ranks; the exchanges want
  // equal segments, so the poses are renumbered for the device (pgo_internal.h pose_int): rank r's share starts at r * rows_per.
  const int world = P->comm ? P->comm->world : 1, rank = P->comm ? P->comm->rank : 0;
  int rows_per = 0;
  P->pose_int.clear(); P->ia_int.clear(); P->ib_int.clear(); P->cmask_int.clear(); P->is_point_int.clear(); P->shard_cut.clear();
  if (world > 1) {
    std::vector<long long> cut((size_t)world + 1, 0);
    if (pgo_row_shard_cuts(N_caller, E, P->ia.data(), P->ib.data(), world, cut.data(), &rows_per) != PGO_OK) return PGO_ERR_INVALID_ARGUMENT;   // THE ownership rule
    P->shard_cut.assign(cut.begin(), cut.end());
    P->n_int = world * rows_per;
    P->pose_int.resize(N_caller);
    for (int r = 0; r < world; ++r)
      for (long long v = cut[r]; v < cut[r + 1]; ++v) P->pose_int[(size_t)v] = r * rows_per + (int)(v - cut[r]);
    P->ia_int.resize(E); P->ib_int.resize(E);
    for (int e = 0; e < E; ++e) { P->ia_int[e] = P->pose_int[P->ia[e]]; P->ib_int[e] = P->pose_int[P->ib[e]]; }
    P->cmask_int.assign(P->n_int, 3);                 // padding poses: both blocks constant, no edges
    for (int v = 0; v < N_caller; ++v) P->cmask_int[P->pose_int[v]] = P->cmask[v];
    if (!P->is_point.empty()) { P->is_point_int.assign(P->n_int, 0); for (int v = 0; v < N_caller; ++v) P->is_point_int[P->pose_int[v]] = P->is_point[v]; }
  } else {
    long long rl = 0, rh = 0;
    if (pgo_row_shard_range(N_caller, 0, 1, &rl, &rh, &rows_per) != PGO_OK) return PGO_ERR_INVALID_ARGUMENT;
    P->n_int = N_caller;
  }
  const int N = P->n_int;                             // poses as the device counts them
  const std::vector<int>&t_ia = P->t_ia(), &t_ib = P->t_ib();
  const int row_lo = world > 1 ? rank * rows_per : 0;
  const int row_hi = world > 1 ? row_lo + (int)(P->shard_cut[rank + 1] - P->shard_cut[rank]) : N;
  std::vector<int> deg(N, 0);
  for (int e = 0; e < E; ++e) { ++deg[t_ia[e]]; ++deg[t_ib[e]]; }
  long long total = 0;
  for (int v = 0; v < N; ++v) total += 1 + deg[v];
  const int NP = world * rows_per;   // padded pose count of every replicated / exchanged array
  const int B = P->force_block ? P->force_block : choose_block(total / world);

  // rows -> workgroups (greedy packing of `block` slots; a row with more incidences gets its own multi-chunk group)
  std::vector<int> wg_row_begin, wg_slot_begin, row_slot_begin(N, 0), row_slot_cnt(N, 0);
  long long slot = 0;
  // A work-group boundary never separates the poses 2i and 2i + 1: the one-launch CG iteration (several ranks: k_pipe_cg; one rank:
  // the fused universal stream k_uni_f) applies the 12 x 12 Jacobi blocks inside the work-group that owns their rows.
  const bool keep_pairs = true;
  bool pairs_whole = keep_pairs;
  // rows per work-group: at most block / 6 where that costs little (the resident CG, pgo_uni_resident.h, keeps a row lane per vector
  // component in registers: one pass), unlimited on graphs of low degree (a chain would pay ~2x the work-groups for it)
  int row_cap = 0;
  auto pack = [&](int lo, int hi, bool record) -> int {
    long long sl = 0;
    int cur = 0, n = 0, rows = 0;
    if (record) { wg_row_begin.assign(1, lo); wg_slot_begin.assign(1, 0); }
    auto close_wg = [&](int next_row) {
      sl = (sl + B - 1) / B * B;
      ++n;
      if (record) { wg_row_begin.push_back(next_row); wg_slot_begin.push_back((int)sl); }
      cur = 0;
      rows = 0;
    };
    for (int v = lo; v < hi; ++v) {
      const int c = 1 + deg[v];
      if (c > B) {
        if (cur > 0) close_wg(v);
        if (record) { row_slot_begin[v] = (int)sl; row_slot_cnt[v] = c; }
        pairs_whole = false;        // (whatever rank's rows: every rank must take the same decision about the owner-only CG)
        sl += c;
        close_wg(v + 1);
        continue;
      }
      int need = c;
      if (keep_pairs && ((v - lo) & 1) == 0 && v + 1 < hi) {
        const int c1 = 1 + deg[v + 1];
        if (c + c1 <= B) need = c + c1;
        else pairs_whole = false;
      }
      if (cur + need > B || (row_cap > 0 && ((v - lo) & 1) == 0 && rows + 2 > row_cap && cur > 0)) close_wg(v);
      if (record) { row_slot_begin[v] = (int)sl; row_slot_cnt[v] = c; }
      sl += c;
      cur += c;
      ++rows;
    }
    if (cur > 0) close_wg(hi);
    if (n == 0) { sl += B; close_wg(hi); }   // a rank without rows still launches one (empty) workgroup
    if (record) slot = sl;
    return n;
  };
  if (world == 1) {
    const bool pw = pairs_whole;
    const int n_free = pack(row_lo, row_hi, false);
    row_cap = (B / 6) & ~1;
    if (pack(row_lo, row_hi, false) > n_free + n_free / 10) row_cap = 0;
    pairs_whole = pw;
  }
  int pq_cap = 1;
  for (int r = 0; r < world; ++r) {
    const int lo_r = world > 1 ? r * rows_per : 0, hi_r = world > 1 ? lo_r + (int)(P->shard_cut[r + 1] - P->shard_cut[r]) : N;
    pq_cap = std::max(pq_cap, pack(lo_r, hi_r, false));
  }
  // (one rank: room for one p'q partial per tile of the symmetric form as well — tiles of >= 32 rows, filled to ~0.8 by the row and
  // weight caps of pgo_sym.cpp: N / 26 of them; several ranks: the same room in the partial-sum rows, below)
  if (world == 1) pq_cap = std::max(pq_cap, N / 16 + 8);
  const int n_wg = pack(row_lo, row_hi, true);
  if (slot > 0x7fffffffLL - 1024) return set_error(PGO_ERR_UNSUPPORTED, "graph too large for 32-bit slot indices");
  const int n_slots = (int)slot;
  const int seg = rows_per * 6 + pq_cap;

  // slots: diagonal first, then the row's incidences in edge order (owned rows only)
  std::vector<int> slot_col(n_slots, -1), slot_row(n_slots, 0), slot_edge(n_slots, -1), fill(N, 0);
  std::vector<uint8_t> slot_side(n_slots, pgo::SIDE_PAD);
  for (int v = row_lo; v < row_hi; ++v) {
    const int sb = row_slot_begin[v];
    slot_col[sb] = v; slot_row[sb] = v; slot_side[sb] = pgo::SIDE_DIAG;
    fill[v] = sb + 1;
  }
  P->edge_begin_slot.assign(E, -1);
  for (int e = 0; e < E; ++e) {
    const int a = t_ia[e], b = t_ib[e];
    if (a >= row_lo && a < row_hi) {
      const int t = fill[a]++;
      slot_col[t] = b; slot_row[t] = a; slot_side[t] = pgo::SIDE_BEGIN; slot_edge[t] = e;
      P->edge_begin_slot[e] = t;
    }
    if (b >= row_lo && b < row_hi) {
      const int t = fill[b]++;
      slot_col[t] = a; slot_row[t] = b; slot_side[t] = pgo::SIDE_END; slot_edge[t] = e;
    }
  }
  // pad slots keep a valid row index so that loads stay in range
  for (int w = 0; w < n_wg; ++w) {
    const int r = std::max(0, std::min(wg_row_begin[w], N - 1));
    for (int t = wg_slot_begin[w]; t < wg_slot_begin[w + 1]; ++t) if (slot_side[t] == pgo::SIDE_PAD) slot_row[t] = r;
  }

  lap("rows -> workgroups, slots");
  P->h_slot_row = slot_row; P->h_slot_col = slot_col; P->h_slot_side = slot_side; P->h_row_slot_begin = row_slot_begin;
  P->direct_analyzed = false; P->direct_usable = false; P->front_usable = false; P->sfront_usable = false; P->cluster_built = 0; P->g.cluster = 1;
  P->sym_built = false; P->sym_ready = false; P->sym_active = false; P->sym_storage = false;
  if (P->want_direct) {
    // an exact request: its host analysis (ordering, symbolic factorisation, schedule) needs nothing but the slot topology
    const long long budget = front_memory_budget();
    const int ns = (int)n_slots;
    P->analysis_kind = 0;
    P->analysis_thread = std::thread([P, N, E, ns, budget]() { P->analysis_kind = decide_direct_host(P, N, E, ns, budget); });
  }
  // measurements: edge order and slot order, component major.  The gathers into slot order are cache-hostile (21-36
  // strided streams indexed by a random edge): split over host threads, each owning a contiguous range.
  HostArray emeas, smeas, eW, sW, eL;
  emeas.resize((size_t)7 * E);
  smeas.resize((size_t)7 * n_slots);
  parallel_for(E, [&](int lo, int hi) {
    for (int e = lo; e < hi; ++e) for (int c = 0; c < 7; ++c) emeas[(size_t)c * E + e] = P->meas[(size_t)7 * e + c];
  });
  parallel_for(n_slots, [&](int lo, int hi) {
    for (int t = lo; t < hi; ++t) {
      const int e = slot_edge[t];
      for (int c = 0; c < 7; ++c) smeas[(size_t)c * n_slots + t] = e < 0 ? (c == 6 ? 1.0 : 0.0) : P->meas[(size_t)7 * e + c];
    }
  });
  std::atomic<int> w_has_pr(0), w_has_offdiag(0);
  if (P->has_info) {
    eW.resize((size_t)21 * E); eL.resize((size_t)36 * E); sW.resize((size_t)21 * n_slots);
    parallel_for(E, [&](int lo, int hi) {
      for (int e = lo; e < hi; ++e) {
        const double* L = &P->sqrt_info[(size_t)36 * e];
        int k = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = i; j < 6; ++j) {
            double w = 0;
            for (int r = 0; r < 6; ++r) w += L[6 * r + i] * L[6 * r + j];  // W = L^T L
            if (i < 3 && j >= 3 && w != 0.0) w_has_pr.store(1, std::memory_order_relaxed);
            if (i != j && w != 0.0) w_has_offdiag.store(1, std::memory_order_relaxed);
            eW[(size_t)k * E + e] = w;
            ++k;
          }
        for (int q = 0; q < 36; ++q) eL[(size_t)q * E + e] = L[q];
      }
    });
    parallel_for(n_slots, [&](int lo, int hi) {
      for (int t = lo; t < hi; ++t) {
        const int e = slot_edge[t];
        if (e < 0) { for (int k = 0; k < 21; ++k) sW[(size_t)k * n_slots + t] = 0.0; continue; }
        // W of the slot's edge, recomputed from L (contiguous 288 B) instead of 21 strided reads of eW
        const double* L = &P->sqrt_info[(size_t)36 * e];
        int k = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = i; j < 6; ++j) {
            double w = 0;
            for (int r = 0; r < 6; ++r) w += L[6 * r + i] * L[6 * r + j];
            sW[(size_t)k * n_slots + t] = w;
            ++k;
          }
      }
    });
  }
  const bool w_blockdiag = P->has_info && w_has_pr.load() == 0;
  const bool w_diag = P->has_info && w_has_offdiag.load() == 0;      // W = diag(w) (the generators' diag(1/sigma^2)): six planes of sW are read
  lap("measurement / W arrays");
  UploadScope upload_scope(s);     // the copies below are enqueued side by side; this function's final synchronisation is their wait
  HIP_TRY(P->d_slot_col.upload(slot_col, s));
  HIP_TRY(P->d_slot_row.upload(slot_row, s));
  HIP_TRY(P->d_slot_side.upload(slot_side, s));
  HIP_TRY(P->d_wg_slot_begin.upload(wg_slot_begin, s));
  HIP_TRY(P->d_wg_row_begin.upload(wg_row_begin, s));
  HIP_TRY(P->d_row_slot_begin.upload(row_slot_begin, s));
  HIP_TRY(P->d_row_slot_cnt.upload(row_slot_cnt, s));
  HIP_TRY(P->d_cmask.upload(P->t_cmask(), s));
  HIP_TRY(P->d_edge_a.upload(t_ia, s));
  HIP_TRY(P->d_edge_b.upload(t_ib, s));
  HIP_TRY(P->d_smeas.upload(smeas.data(), smeas.n, s));
  HIP_TRY(P->d_emeas.upload(emeas.data(), emeas.n, s));
  HIP_TRY(P->d_sW.upload(sW.data(), sW.n, s));
  HIP_TRY(P->d_eW.upload(eW.data(), eW.n, s));
  HIP_TRY(P->d_eL.upload(eL.data(), eL.n, s));
  // cluster-preconditioner lists (prepare_clusters fills them): allocated here, at their upper bounds, because on this
  // stack an upload into a buffer allocated AFTER the large allocations below takes 6-25 ms to complete
  HIP_TRY(P->d_cl_ptr.alloc((size_t)N + 2));
  HIP_TRY(P->d_cl_slot.alloc((size_t)E + 1));
  HIP_TRY(P->d_cl_rc.alloc((size_t)E + 1));

  lap("uploads");
  const size_t m = (size_t)6 * NP;
  HIP_TRY(P->d_pose_x.alloc((size_t)pgo::POSE_STRIDE * N));
  HIP_TRY(P->d_pose_c.alloc((size_t)pgo::POSE_STRIDE * N));
  HIP_TRY(P->d_pose_0.alloc((size_t)pgo::POSE_STRIDE * N));
  HIP_TRY(P->d_bsr.alloc((size_t)n_slots * 36));
  HIP_TRY(P->d_bsr.zero(s));
  P->spec_ready = false;     // the spare linearisation set follows the new sizes when it is next needed
  HIP_TRY(P->d_Hdiag.alloc((size_t)36 * NP));
  HIP_TRY(P->d_Hdiag.zero(s));
  HIP_TRY(P->d_Minv.alloc((size_t)36 * NP * 4 + (size_t)world * 144 * 4));   // room for 4-pose clusters of every rank, padded
  HIP_TRY(P->d_Minv.zero(s));
  DevBuf<double>* vecs[] = {&P->d_grad, &P->d_scale, &P->d_d2, &P->d_diagc, &P->d_cg_b, &P->d_cg_x, &P->d_cg_r,
                            &P->d_cg_z, &P->d_cg_p0, &P->d_cg_p1, &P->d_delta};
  DevBuf<double>* pipe_vecs[] = {&P->d_cg_u, &P->d_cg_w, &P->d_cg_s, &P->d_cg_qq};
  for (DevBuf<double>* b : vecs) { HIP_TRY(b->alloc(m)); HIP_TRY(b->zero(s)); }
  HIP_TRY(P->d_cg_q.alloc((size_t)world * seg));   // exchange buffer: q segments + p'q partials (unused partial slots stay 0)
  HIP_TRY(P->d_cg_q.zero(s));
  const int pipe_seg = rows_per * 6 + 4;      // m of the owned rows, then this rank's (r,u), (w,u), x'(b + r)
  {                    // one-launch CG iterations (pgo_kernels.h DeviceGraph::pipe_buf): owner-only CG of several ranks, fused stream of one
    for (DevBuf<double>* b : pipe_vecs) { HIP_TRY(b->alloc(m)); HIP_TRY(b->zero(s)); }
    // (r06: what kernels of other devices store into — the IPC transport's device-initiated exchange — is fine-grained memory)
    const bool remote_peers = world > 1 && P->comm && P->comm->peers_may_be_remote();
    if (remote_peers) { HIP_TRY(P->d_pipe_a.alloc_fine((size_t)world * pipe_seg)); HIP_TRY(P->d_pipe_b.alloc_fine((size_t)world * pipe_seg)); }
    else { HIP_TRY(P->d_pipe_a.alloc((size_t)world * pipe_seg)); HIP_TRY(P->d_pipe_b.alloc((size_t)world * pipe_seg)); }
    HIP_TRY(P->d_pipe_a.zero(s));
    HIP_TRY(P->d_pipe_b.zero(s));
    HIP_TRY(P->d_pipe_x.alloc((size_t)world * rows_per * 6)); HIP_TRY(P->d_pipe_x.zero(s));    // exchange buffer of the diagonal blocks' diagonals (linearize_all)
  }
  const int n_vec_wg = std::max(1, std::min((int)(((size_t)6 * N + pgo::vec_block() - 1) / pgo::vec_block()), 256));
  const int n_edge_wg = std::min(std::max(1, (E + pgo::edge_block() - 1) / pgo::edge_block()), pgo::max_edge_wg());   // k_cost and the step tail stride beyond that
  const int n_pose_wg = (N + pgo::pose_block() - 1) / pgo::pose_block();
  const int n_part = std::max(std::max(std::max(n_wg, n_vec_wg), n_edge_wg + n_pose_wg), std::max(pq_cap, rows_per / 16 + 8));   // the fused step tail runs n_edge_wg + n_pose_wg workgroups; one entry per tile of the symmetric form
  // (blocks may come from the pool with a previous problem's contents: everything that is not fully written before it is read
  // is cleared here)
  HIP_TRY(P->d_part_rz.alloc((size_t)2 * n_part));
  HIP_TRY(P->d_part_rz.zero(s));
  HIP_TRY(P->d_part_q.alloc((size_t)2 * n_part));
  HIP_TRY(P->d_part_q.zero(s));
  HIP_TRY(P->d_part_rr.alloc((size_t)2 * n_part));
  HIP_TRY(P->d_part_rr.zero(s));
  HIP_TRY(P->d_part_bb.alloc((size_t)n_part));
  HIP_TRY(P->d_part_bb.zero(s));
  HIP_TRY(P->d_part_misc.alloc((size_t)8 * n_part));
  HIP_TRY(P->d_part_misc.zero(s));
  HIP_TRY(P->d_part_f.alloc((size_t)2 * 4 * n_part));     // fused stream: [launch parity][work-group][gamma, delta, Q, -]
  HIP_TRY(P->d_part_f.zero(s));
  HIP_TRY(P->d_cg.alloc(1));
  HIP_TRY(P->d_cg.zero(s));
  HIP_TRY(P->d_flags.alloc(2240));     // [0..3] as pgo_kernels.h says, [4..12] the two-level ticket of the fused stream (k_uni_f), [64..2239] the grid barrier of the resident CG (k_res_cg: a 128-byte line per word)
  HIP_TRY(P->d_flags.zero(s));

  pgo::DeviceGraph& g = P->g;
  g.N = N; g.E = E; g.n_wg = n_wg; g.n_slots = n_slots; g.block = B;
  g.world = world; g.rank = rank; g.rows_per = rows_per; g.row_lo = row_lo; g.row_hi = row_hi; g.pq_cap = pq_cap; g.seg = seg;
  // Several ranks: the collectives are enqueued eagerly (capturing them into the CG batch graph was only ever validated at world
  // size 1); the per-iteration cost is then dominated by the collective's latency, not by launch overhead.
  if (P->comm && (!P->comm->capturable() || world > 1)) P->use_graph = false;
  // 0 identity, 1 general, 2 block-diagonal W (every W_pr entry exactly zero: diag(1/sigma^2) and the like); 0 and 2 use the packed
  // 27-entry slots.
  {
    const bool force_full = false;
    const bool no_diag = pgo::tuning("no_diag_info", 0.0) != 0.0;    // (A/B knob: the 12-entry reads of mode 2)
    g.info_mode = !P->has_info ? 0 : (w_diag && !force_full && !no_diag) ? 3 : (w_blockdiag && !force_full) ? 2 : 1;
    g.blk_packed = (g.info_mode != 1 && !force_full) ? 1 : 0;
  }
  g.loss_kind = P->loss_kind; g.loss_a = P->loss_a;
  g.slot_col = P->d_slot_col.p; g.slot_row = P->d_slot_row.p; g.slot_side = P->d_slot_side.p;
  g.wg_slot_begin = P->d_wg_slot_begin.p; g.wg_row_begin = P->d_wg_row_begin.p;
  g.row_slot_begin = P->d_row_slot_begin.p; g.row_slot_cnt = P->d_row_slot_cnt.p; g.cmask = P->d_cmask.p;
  g.smeas = P->d_smeas.p; g.sW = P->d_sW.p; g.edge_a = P->d_edge_a.p; g.edge_b = P->d_edge_b.p;
  g.emeas = P->d_emeas.p; g.eW = P->d_eW.p; g.eL = P->d_eL.p;
  g.pose_x = P->d_pose_x.p; g.pose_c = P->d_pose_c.p; g.bsr_val = P->d_bsr.p; g.Hdiag = P->d_Hdiag.p;
  g.Minv = P->d_Minv.p; g.grad = P->d_grad.p; g.scale = P->d_scale.p; g.d2 = P->d_d2.p;
  g.diag_clamped = P->d_diagc.p; g.cg_b = P->d_cg_b.p; g.cg_x = P->d_cg_x.p; g.cg_r = P->d_cg_r.p;
  g.cg_z = P->d_cg_z.p; g.cg_q = P->d_cg_q.p; g.cg_p0 = P->d_cg_p0.p; g.cg_p1 = P->d_cg_p1.p;
  g.delta = P->d_delta.p; g.part_rz = P->d_part_rz.p; g.part_q = P->d_part_q.p;
  g.part_rr = P->d_part_rr.p; g.part_bb = P->d_part_bb.p; g.part_misc = P->d_part_misc.p; g.part_f = P->d_part_f.p;
  g.cg_u = P->d_cg_u.p; g.cg_w = P->d_cg_w.p; g.cg_s = P->d_cg_s.p; g.cg_qq = P->d_cg_qq.p;
  g.pipe_buf[0] = P->d_pipe_a.p; g.pipe_buf[1] = P->d_pipe_b.p; g.pipe_seg = pipe_seg;
  g.bx[0] = g.bx[1] = nullptr; g.bx_brow = nullptr; g.bx_slot_off = nullptr; g.bx_nb = 0; g.bx_cseg = 0;     // (a session engages the boundary exchange: lm_begin)
  P->bx_ready = false;
  if (world > 1) {
    // boundary exchange of the owner-only CG (pgo_kernels.h DeviceGraph::bx): a row travels per CG iteration only if an edge leaves its
    // rank.  Every rank derives every rank's list from the (replicated) edge list: the segments' layout is the same everywhere without a
    // word exchanged.
    std::vector<uint8_t> bnd((size_t)N, 0);
    for (int e = 0; e < E; ++e)
      if (t_ia[e] / rows_per != t_ib[e] / rows_per) { bnd[(size_t)t_ia[e]] = 1; bnd[(size_t)t_ib[e]] = 1; }
    std::vector<int> count((size_t)world, 0), brow;
    P->h_bpos.assign((size_t)N, -1);
    for (int v = 0; v < N; ++v)
      if (bnd[(size_t)v]) {
        const int k = v / rows_per;
        P->h_bpos[(size_t)v] = count[(size_t)k]++;
        if (k == rank) brow.push_back(v);
      }
    int bmax = 2;
    for (int k = 0; k < world; ++k) bmax = std::max(bmax, count[(size_t)k]);
    bmax = (bmax + 1) & ~1;
    const int cseg = 6 * bmax + 4;
    bool fits = (long long)world * pipe_seg < (1LL << 31) && (long long)world * cseg < (1LL << 31);
    std::vector<int> soff((size_t)n_slots, 0);
    for (int t = 0; t < n_slots && fits; ++t) {
      const int c = slot_col[(size_t)t];
      if (c < 0) continue;
      const int k = c / rows_per;
      if (k == rank) soff[(size_t)t] = k * pipe_seg + (c - k * rows_per) * 6;
      else if (P->h_bpos[(size_t)c] >= 0) soff[(size_t)t] = -1 - (k * cseg + 6 * P->h_bpos[(size_t)c]);
      else fits = false;          // (cannot happen: the far end of a cut edge is a boundary row of its rank)
    }
    if (fits) {
      P->bx_nb = (int)brow.size();
      if (brow.empty()) brow.push_back(row_lo);       // (an upload of nothing is not one)
      HIP_TRY(P->d_bx_brow.upload(brow, s));
      HIP_TRY(P->d_bx_slot_off.upload(soff, s));
      HIP_TRY(P->d_bx0.alloc((size_t)world * cseg)); HIP_TRY(P->d_bx0.zero(s));
      HIP_TRY(P->d_bx1.alloc((size_t)world * cseg)); HIP_TRY(P->d_bx1.zero(s));
      P->bx_cseg = cseg;
      P->bx_ready = true;
      if (verbose) std::fprintf(stderr, "[pgo] rank %d: %d of %d rows are boundary rows; a CG iteration exchanges %d doubles per rank instead of %d\n", rank, P->bx_nb, row_hi - row_lo, cseg, pipe_seg);
    }
  }
  g.peer_tab = nullptr; g.peer_flags = nullptr;
  P->peer_dirty = true;       // (the table is exchanged by peer_direct_setup(), outside this function's upload scope: it is a collective call)
  g.pairs_whole = pairs_whole ? 1 : 0;
  {
    int most = 0;
    for (int w = 0; w < n_wg; ++w) most = std::max(most, wg_row_begin[w + 1] - wg_row_begin[w]);
    g.rows_fit = 6 * most <= B ? 1 : 0;
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, P->device) != hipSuccess) { (void)hipGetLastError(); n_cu = 0; }
    g.n_cu = n_cu;
  }
  g.n_part = n_part; g.n_vec_wg = n_vec_wg; g.n_edge_wg = n_edge_wg; g.n_pose_wg = n_pose_wg;
  g.cg = P->d_cg.p; g.flags = P->d_flags.p;
  g.oplog = nullptr; g.oplog_cap = 0; g.oplog_indexed = 0;
  g.sym_dst = nullptr; g.sym_val = nullptr;
  if (getenv("PGO_UNI_OPLOG")) {
    const size_t cap = (size_t)1 << 21;
    HIP_TRY(P->d_oplog.alloc(cap));
    HIP_TRY(hipMemsetAsync(P->d_oplog.p, 0, sizeof(long long), s));
    g.oplog = P->d_oplog.p; g.oplog_cap = (int)cap;
  }
  void* dscal = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dscal, P->scal, 0));
  g.scal = reinterpret_cast<pgo::LmScalars*>(dscal);
  HIP_TRY(hipStreamSynchronize(s));
  lap("device buffers");
  P->topo_dirty = false;
  P->lm.t_setup = seconds_since(t0);
  return PGO_OK;
}

// Device-initiated exchange where the transport can hand kernels the peers' buffers (the loopback transport; over RCCL the buffers
// would have to be fine-grained IPC allocations: not built, the all-gather stays).  A COLLECTIVE call: every rank makes it after
// its prepare() (never inside it: prepare() holds the process-wide staging buffers while it uploads, and a rank waiting for its
// peers there would keep them from ever arriving).
int peer_direct_setup(pgo_problem* P) {
  if (!P->peer_dirty) return PGO_OK;
  P->peer_dirty = false;
  pgo::DeviceGraph& g = P->g;
  const int world = g.world;
  hipStream_t s = P->stream;
  g.peer_tab = nullptr; g.peer_flags = nullptr;
  if (world > 1 && P->comm && P->d_pipe_a.p) {
    // Opt-in (PGO_PEER_DIRECT=1): virtual ranks are streams of ONE device, and streams that share a hardware queue (ROCm hands out
    // GPU_MAX_HW_QUEUES = 4 per process by default) can put a waiting launch in front of the kernel it waits for — that only ends
    // by the wait's time-out.  On real ranks (one device each) the question does not arise.  Every rank reads the same environment.
    // A transport made of processes (IpcComm) has it on by default; PGO_PEER_DIRECT=0 / 1 overrides either way.
    const char* pd = getenv("PGO_PEER_DIRECT");
    if (pd ? pd[0] == '1' : P->comm->peer_direct_default()) {
      if (P->comm->peers_may_be_remote()) HIP_TRY(P->d_peer_flags.alloc_fine((size_t)world));
      else HIP_TRY(P->d_peer_flags.alloc((size_t)world));
      HIP_TRY(P->d_peer_flags.zero(s));
      HIP_TRY(hipStreamSynchronize(s));
      void* mine[3] = {P->d_pipe_a.p, P->d_pipe_b.p, P->d_peer_flags.p};
      std::vector<void*> tab((size_t)3 * world, nullptr);
      const char* what = "";
      if (P->comm->peer_table(mine, tab.data(), &what) == 0) {
        HIP_TRY(P->d_peer_tab.upload(tab, s));
        HIP_TRY(hipStreamSynchronize(s));
        g.peer_tab = P->d_peer_tab.p; g.peer_flags = P->d_peer_flags.p;
        P->peer_gseq = 0;
      }
    }
  }
  return PGO_OK;
}

int upload_poses(pgo_problem* P, double* dst) {
  const int N = (int)P->pp.size();
  const bool ren = P->renumbered();
  std::vector<double> h((size_t)pgo::POSE_STRIDE * (ren ? P->n_int : N), 0.0);
  if (ren) for (int v = 0; v < P->n_int; ++v) h[(size_t)pgo::POSE_STRIDE * v + 6] = 1.0;      // padding poses: the identity
  for (int v = 0; v < N; ++v) {
    double* o = &h[(size_t)pgo::POSE_STRIDE * (ren ? P->pose_int[v] : v)];
    o[0] = P->pp[v][0]; o[1] = P->pp[v][1]; o[2] = P->pp[v][2];
    o[3] = P->qq[v][0]; o[4] = P->qq[v][1]; o[5] = P->qq[v][2]; o[6] = P->qq[v][3];
  }
  HIP_TRY(staged_h2d(dst, h.data(), h.size() * sizeof(double), P->stream));
  return PGO_OK;
}

int download_poses(pgo_problem* P, const double* src) {
  const int N = (int)P->pp.size();
  const bool ren = P->renumbered();
  std::vector<double> h((size_t)pgo::POSE_STRIDE * (ren ? P->n_int : N));
  HIP_TRY(staged_d2h(h.data(), src, h.size() * sizeof(double), P->stream));
  for (int v = 0; v < N; ++v) {
    const double* o = &h[(size_t)pgo::POSE_STRIDE * (ren ? P->pose_int[v] : v)];
    // constant blocks are never written (row 0 of the reference's before/after files is identical)
    if (!(P->cmask[v] & 1)) { P->pp[v][0] = o[0]; P->pp[v][1] = o[1]; P->pp[v][2] = o[2]; }
    if (!(P->cmask[v] & 2)) { P->qq[v][0] = o[3]; P->qq[v][1] = o[4]; P->qq[v][2] = o[5]; P->qq[v][3] = o[6]; }
  }
  return PGO_OK;
}

int pose_rows_to_host(pgo_problem* P, double* host, const double* dev, int width) {
  const size_t N = P->pp.size();
  if (!P->renumbered()) { HIP_TRY(staged_d2h(host, dev, sizeof(double) * width * N, P->stream)); return PGO_OK; }
  std::vector<double> h((size_t)width * P->n_int);
  HIP_TRY(staged_d2h(h.data(), dev, h.size() * sizeof(double), P->stream));
  for (size_t v = 0; v < N; ++v) memcpy(host + v * width, &h[(size_t)P->pose_int[v] * width], sizeof(double) * width);
  return PGO_OK;
}
int pose_rows_to_device(pgo_problem* P, double* dev, const double* host, int width, double pad_value) {
  const size_t N = P->pp.size();
  if (!P->renumbered()) { HIP_TRY(staged_h2d(dev, host, sizeof(double) * width * N, P->stream)); return PGO_OK; }
  std::vector<double> h((size_t)width * P->n_int, pad_value);
  for (size_t v = 0; v < N; ++v) memcpy(&h[(size_t)P->pose_int[v] * width], host + v * width, sizeof(double) * width);
  HIP_TRY(staged_h2d(dev, h.data(), h.size() * sizeof(double), P->stream));
  return PGO_OK;
}

int fill_scale_one(pgo_problem* P) {
  std::vector<double> one((size_t)6 * P->g.N, 1.0);
  HIP_TRY(staged_h2d(P->g.scale, one.data(), one.size() * sizeof(double), P->stream));
  return PGO_OK;
}

// =================================================================================================
// C ABI (include/pgo.h)
// =================================================================================================
extern "C" {

int pgo_version(void) { return PGO_VERSION; }
const char* pgo_last_error(void) { return last_error_string(); }

int pgo_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return count;
}

static int g_default_device = 0;
int pgo_set_device(int device) {
  if (device < 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "negative device index");
  g_default_device = device;
  return PGO_OK;
}

pgo_problem* pgo_problem_create(void) {
  pgo_problem* p = new (std::nothrow) pgo_problem();
  if (p) p->device = g_default_device;
  return p;
}
void pgo_problem_destroy(pgo_problem* problem) { if (problem) resident_slot_release(problem); delete problem; }

int pgo_problem_add_pose(pgo_problem* P, double* p, double* q) {
  if (!P || !p || !q) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_problem_add_pose");
  auto ip = P->block_of_ptr.find(p), iq = P->block_of_ptr.find(q);
  if (ip != P->block_of_ptr.end() || iq != P->block_of_ptr.end()) {
    if (ip != P->block_of_ptr.end() && iq != P->block_of_ptr.end() && (ip->second >> 1) == (iq->second >> 1) &&
        (ip->second & 1) == 0 && (iq->second & 1) == 1)
      return ip->second >> 1;
    return set_error(PGO_ERR_UNSUPPORTED, "a parameter block is already paired with a different translation/rotation block");
  }
  const int idx = (int)P->pp.size();
  P->pp.push_back(p);
  P->qq.push_back(q);
  P->cmask.push_back(0);
  P->is_point.push_back(0);
  P->block_of_ptr[p] = 2 * idx;
  P->block_of_ptr[q] = 2 * idx + 1;
  P->topo_dirty = true;
  return idx;
}

// ---- pose / landmark problems (g2o analogue Thirdparty/g2o/g2o/core/block_solver.hpp:47-87) ----
int pgo_problem_add_point(pgo_problem* P, double* xyz) {
  if (!P || !xyz) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_problem_add_point");
  auto it = P->block_of_ptr.find(xyz);
  if (it != P->block_of_ptr.end()) {
    if ((it->second & 1) == 0 && P->is_point[it->second >> 1]) return it->second >> 1;
    return set_error(PGO_ERR_UNSUPPORTED, "the block is already a pose block");
  }
  const int idx = (int)P->pp.size();
  P->point_q.push_back(std::array<double, 4>{0.0, 0.0, 0.0, 1.0});     // (deque: the address stays put)
  P->pp.push_back(xyz);
  P->qq.push_back(P->point_q.back().data());
  P->cmask.push_back(2);             // the rotation block is constant: the node has three free dimensions
  P->is_point.push_back(1);
  P->block_of_ptr[xyz] = 2 * idx;
  P->topo_dirty = true;
  return idx;
}

int pgo_problem_add_points(pgo_problem* P, int n, double* base, int stride) {
  if (!P || !base || n < 0 || stride < 3) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_add_points");
  int first = (int)P->pp.size();
  for (int i = 0; i < n; ++i) {
    const int r = pgo_problem_add_point(P, base + (size_t)i * stride);
    if (r < 0) return r;
    if (i == 0) first = r;         // (blocks added before keep their index)
  }
  return first;
}

int pgo_problem_add_point_observation_batch(pgo_problem* P, int n, const int* pose, const int* point, const double* z, const double* sqrt_information3) {
  if (!P || n < 0 || (n > 0 && (!pose || !point || !z))) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_add_point_observation_batch");
  const int N = (int)P->pp.size();
  for (int i = 0; i < n; ++i) {
    if (pose[i] < 0 || pose[i] >= N || P->is_point[pose[i]]) return set_error(PGO_ERR_INVALID_ARGUMENT, "observation %d: %d is not a pose", i, pose[i]);
    if (point[i] < 0 || point[i] >= N || !P->is_point[point[i]]) return set_error(PGO_ERR_INVALID_ARGUMENT, "observation %d: %d is not a point", i, point[i]);
  }
  // residual L3 (R(q_pose)^T (l - p_pose) - z): the translation rows of the between-factor from the pose to the point's node, no rotation rows
  std::vector<double> t((size_t)7 * n, 0.0), L((size_t)36 * n, 0.0);
  for (int i = 0; i < n; ++i) {
    t[(size_t)7 * i] = z[3 * i]; t[(size_t)7 * i + 1] = z[3 * i + 1]; t[(size_t)7 * i + 2] = z[3 * i + 2]; t[(size_t)7 * i + 6] = 1.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) L[(size_t)36 * i + 6 * r + c] = sqrt_information3 ? sqrt_information3[(size_t)9 * i + 3 * r + c] : (r == c ? 1.0 : 0.0);
  }
  return pgo_problem_add_se3_between_batch(P, n, pose, point, t.data(), L.data());
}

int pgo_problem_add_poses(pgo_problem* P, int n, double* base, int stride) {
  if (!P || !base || n < 0 || stride < 7) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_add_poses");
  const int first = (int)P->pp.size();
  P->pp.reserve(first + n); P->qq.reserve(first + n); P->cmask.reserve(first + n);
  P->block_of_ptr.reserve((size_t)2 * (first + n));
  for (int i = 0; i < n; ++i) {
    const int r = pgo_problem_add_pose(P, base + (size_t)i * stride, base + (size_t)i * stride + 3);
    if (r < 0) return r;
  }
  return first;
}

int pgo_problem_add_se3_between_batch(pgo_problem* P, int n, const int* begin, const int* end, const double* t_be,
                                      const double* sqrt_information) {
  if (!P || n < 0 || (n > 0 && (!begin || !end || !t_be))) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_add_se3_between_batch");
  const int N = (int)P->pp.size();
  for (int i = 0; i < n; ++i) {
    if (begin[i] < 0 || begin[i] >= N || end[i] < 0 || end[i] >= N)
      return set_error(PGO_ERR_INVALID_ARGUMENT, "edge %d references a pose that was never added", i);
    if (begin[i] == end[i]) return set_error(PGO_ERR_INVALID_ARGUMENT, "edge %d connects a pose to itself", i);
  }
  const int first = (int)P->ia.size();
  if (sqrt_information && !P->has_info) {
    // earlier edges used the identity
    P->sqrt_info.assign((size_t)36 * first, 0.0);
    for (int e = 0; e < first; ++e) for (int d = 0; d < 6; ++d) P->sqrt_info[(size_t)36 * e + 7 * d] = 1.0;
    P->has_info = true;
  }
  P->ia.insert(P->ia.end(), begin, begin + n);
  P->ib.insert(P->ib.end(), end, end + n);
  P->meas.insert(P->meas.end(), t_be, t_be + (size_t)7 * n);
  if (P->has_info) {
    if (sqrt_information) {
      P->sqrt_info.insert(P->sqrt_info.end(), sqrt_information, sqrt_information + (size_t)36 * n);
    } else {
      const size_t old = P->sqrt_info.size();
      P->sqrt_info.resize(old + (size_t)36 * n, 0.0);
      for (int e = 0; e < n; ++e) for (int d = 0; d < 6; ++d) P->sqrt_info[old + (size_t)36 * e + 7 * d] = 1.0;
    }
  }
  P->topo_dirty = true;
  return first;
}

int pgo_problem_add_se3_between(pgo_problem* P, int pose_begin, int pose_end, const double* t_be_p, const double* t_be_q,
                                const double* sqrt_information) {
  if (!P || !t_be_p || !t_be_q) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_problem_add_se3_between");
  double t[7] = {t_be_p[0], t_be_p[1], t_be_p[2], t_be_q[0], t_be_q[1], t_be_q[2], t_be_q[3]};
  return pgo_problem_add_se3_between_batch(P, 1, &pose_begin, &pose_end, t, sqrt_information);
}

int pgo_problem_set_loss(pgo_problem* P, int kind, double a) {
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  if (kind < PGO_LOSS_TRIVIAL || kind > PGO_LOSS_SWITCHABLE) return set_error(PGO_ERR_UNSUPPORTED, "unknown loss kind %d", kind);
  if (kind != PGO_LOSS_TRIVIAL && !(a > 0.0)) return set_error(PGO_ERR_INVALID_ARGUMENT, "loss scale must be positive");
  P->loss_kind = kind;
  P->loss_a = a;
  return PGO_OK;
}

int pgo_problem_set_pose_constant(pgo_problem* P, int pose, int which) {
  if (!P || pose < 0 || pose >= (int)P->pp.size() || (which & ~3) || which == 0)
    return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_set_pose_constant");
  P->cmask[pose] |= (uint8_t)which;
  P->topo_dirty = true;
  return PGO_OK;
}

int pgo_problem_set_parameter_block_constant(pgo_problem* P, const double* block) {
  if (!P || !block) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument");
  auto it = P->block_of_ptr.find(block);
  if (it == P->block_of_ptr.end()) return set_error(PGO_ERR_INVALID_ARGUMENT, "parameter block not found in the problem");
  return pgo_problem_set_pose_constant(P, it->second >> 1, (it->second & 1) ? 2 : 1);
}

int pgo_problem_num_poses(const pgo_problem* P) { return P ? (int)P->pp.size() : 0; }
int pgo_problem_num_edges(const pgo_problem* P) { return P ? (int)P->ia.size() : 0; }

void pgo_solver_options_init(pgo_solver_options* o) {
  memset(o, 0, sizeof *o);
  o->max_num_iterations = 50;
  o->linear_solver_type = PGO_SPARSE_NORMAL_CHOLESKY;
  o->jacobi_scaling = 1;
  o->max_linear_solver_iterations = 500;
  o->min_linear_solver_iterations = 0;
  o->max_num_consecutive_invalid_steps = 5;
  o->cg_batch = 0;
  o->pcg_cluster_poses = 1;
  o->cg_residual_reset_period = 10;   // LinearSolver::Options::residual_reset_period of Ceres 1.13
  o->pcg_form = 0;
  o->pcg_coarse_aggregate = 0;
  o->reserved_options = 0;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->eta = 1e-1;
  o->exact_r_tolerance = 1e-13;
}

int pgo_release_device_memory(void) {
  device_pool().trim();
  host_side_pool().trim();
  return PGO_OK;
}

}  // extern "C"
