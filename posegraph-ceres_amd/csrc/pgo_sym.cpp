// pgo_sym.cpp — host side of the symmetric tile form (pgo_sym.h): row tiles with graph locality, the row-after-row layout of the
// stored slots in chunks of 256, and every index the kernels need.  One-off per topology (one rank; built when a PCG solve of a
// large graph starts and can repay it: 44 ms at 100 k poses / 1 M edges).
#include <algorithm>
#include <numeric>

#include "pgo_internal.h"
#include "pgo_sym_host.h"

bool sym_wanted(const pgo_problem* P) {
  const char* e = getenv("PGO_SYM");
  if (e && e[0] == '0') return false;
  if (P->use_graph || P->coarse_on) return false;
  // several ranks (r06): every rank keeps the form of ITS rows; the only CG that multiplies with it there is the owner-only pipelined
  // one (k_pipe_cg_sym), so the request has to be one that form serves
  if (P->g.world > 1 && !pgo::pipe_supported(P->g, cg_params_for(P->opt), P->g.cluster)) return false;
  if (P->g.world > 1) { const char* pe = getenv("PGO_SHARD_PIPE"); if ((pe && pe[0] == '0') || P->force_standard_cg) return false; }
  // (every rank needs rows to make tiles of — a share without any would launch an empty grid; the shares are known to every rank alike)
  if (P->g.world > 1)
    for (size_t k = 0; k + 1 < P->shard_cut.size(); ++k) if (P->shard_cut[k + 1] - P->shard_cut[k] < 4) return false;
  if (e && e[0] == '1') return true;
  // the universal stream serves the graphs below its slot limit (latency-bound kernels of a few microseconds: nothing to gain
  // from fewer bytes there); above it the host-driven CG runs and the SpMV is bandwidth-bound.  The limit is on the WHOLE graph
  // (N + 2 E incidence slots), whatever share of it this rank holds.
  const long long limit = 600000;
  return (long long)P->pp.size() + 2LL * (long long)P->ia.size() > limit;
}

// Builds P->sym (device arrays + pgo::SymGraph).  Returns PGO_OK with P->sym_ready false when the graph does not fit the form
// (a tile with more than SYM_X_MAX staged columns): the caller keeps the incidence-slot kernels.
int sym_prepare(pgo_problem* P) {
  if (P->sym_built) return PGO_OK;
  P->sym_built = true;
  P->sym_ready = false;
  const auto t0 = Clock::now();
  const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto lap = [&, tl = Clock::now()](const char* what) mutable {
    if (verbose) std::fprintf(stderr, "[pgo] sym_prepare: %-24s %.2f ms\n", what, 1e3 * seconds_since(tl));
    tl = Clock::now();
  };
  const int N = P->n_int, E = (int)P->ia.size();      // (the device's numbering: pgo_internal.h pose_int)
  const std::vector<int>&t_ia = P->t_ia(), &t_ib = P->t_ib();
  hipStream_t s = P->stream;
  // tile caps: up to 256 rows; enough tiles to fill the chip on small graphs; the weight cap keeps the tiles' stored slots alike
  pgo::SymHostParams hp;
  const int re = (int)pgo::tuning("sym_rows", 0.0);
  const int N_own = std::max(1, P->g.row_hi - P->g.row_lo);
  // (r06, tools/sym_rows_probe.py on a shard-sized problem alone on the chip — 12.5 k / 25 k rows: k_pipe_cg_sym 27.0 / 44.5 us per CG
  // iteration with 32-row tiles, 22.0 / 37.5 with 48, 23.9 / 33.4 with 64, 27.6 / 31.2 with 96, 49.5 / 51.4 with 256: the optimum sits at
  // ~320 tiles, i.e. rows / 260 per tile; larger tiles store fewer blocks — 1.14x the algorithmic bytes per product with 256 rows against
  // 1.44x with 32 on an eighth of BASELINE configs[3] — but a shard of this size is latency-bound, not bandwidth-bound)
  hp.row_cap = re > 0 ? re : std::min(256, std::max(32, (N_own / 260) & ~1));
  hp.row_cap = std::max(8, std::min(hp.row_cap, (int)pgo::SYM_LANES));
  // (several ranks: this rank's rows and their incidences)
  long long own_slots = P->g.row_hi - P->g.row_lo;
  if (P->g.world > 1) { for (int e = 0; e < E; ++e) own_slots += (t_ia[e] >= P->g.row_lo && t_ia[e] < P->g.row_hi) + (t_ib[e] >= P->g.row_lo && t_ib[e] < P->g.row_hi); }
  const double avg_w = P->g.world > 1 ? std::max(1.0, (double)own_slots / N_own) : (double)(N + 2LL * E) / std::max(1, N);
  const double w_mult = 0.95;
  hp.w_cap = std::max<long long>(64, (long long)(w_mult * hp.row_cap * avg_w));
  hp.sort_tiles = true;
  // whole 2-pose clusters per tile, always: what the one-launch CG iteration on the form needs for 12 x 12 Jacobi blocks is also fine
  // for 6 x 6 ones, and the form is built once per topology whatever preconditioner later sessions ask for
  hp.row_lo = P->g.row_lo; hp.row_hi = P->g.row_hi; hp.unit = 2;
  pgo::SymHostLayout H;
  pgo::sym_build_host(N, E, t_ia.data(), t_ib.data(), P->h_row_slot_begin.data(), hp, &H);
  if (verbose) std::fprintf(stderr, "[pgo] sym_prepare: partition %.2f ms, tile layout %.2f ms\n", H.ms_partition, H.ms_layout);
  if (H.unfit) {
    if (verbose) std::fprintf(stderr, "[pgo] sym: tile %d %s: the incidence-slot kernels stay\n", H.unfit_tile, H.unfit);
    return PGO_OK;
  }
  const int T = (int)H.tiles.size(), row_cap = hp.row_cap, x_cap = H.x_cap;
  const long long interior_edges = H.interior_edges, stored = H.stored;
  std::vector<pgo::SymTile>& tiles = H.tiles;
  std::vector<int>&xlist = H.xlist, &chunk_base = H.chunk_base, &chunk_n = H.chunk_n, &src_slot = H.src_slot, &diag_slot = H.diag_slot;
  std::vector<uint32_t>&meta = H.meta, &rinfo = H.rinfo;
  lap("tile layout");
  const int n_slots = (int)meta.size();
  // one rank: the p'q partials of the tiles (k_spmv_sym<0>, Ceres' CG) ride in the slots of the row partition's work-groups; several
  // ranks run the pipelined CG only (k_pipe_cg_sym), whose three partial sums per tile go to the partial-sum rows (n_part entries)
  const int part_cap = P->g.world > 1 ? P->g.n_part : std::min(P->g.pq_cap, P->g.n_part);
  if (T > part_cap) {
    if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] sym: %d tiles > %d partial-sum slots: the incidence-slot kernels stay\n", T, part_cap);
    return PGO_OK;
  }
  if (pgo::sym_lds_bytes(pgo::SymGraph{T, (int)chunk_base.size(), n_slots, x_cap}) > 160 * 1024 - 1024) return PGO_OK;

  HIP_TRY(P->sy_tile.upload(tiles, s));
  HIP_TRY(P->sy_xlist.upload(xlist, s));
  HIP_TRY(P->sy_chunk_base.upload(chunk_base, s));
  HIP_TRY(P->sy_chunk_n.upload(chunk_n, s));
  HIP_TRY(P->sy_meta.upload(meta, s));
  HIP_TRY(P->sy_rinfo.upload(rinfo, s));
  HIP_TRY(P->sy_src.upload(src_slot, s));
  HIP_TRY(P->sy_diag.upload(diag_slot, s));
  HIP_TRY(P->sy_val.alloc((size_t)n_slots * 36));
  HIP_TRY(P->sy_val.zero(s));
  pgo::SymGraph& sg = P->sym;
  sg.n_tiles = T; sg.n_chunks = (int)chunk_base.size(); sg.n_slots = n_slots; sg.x_cap = x_cap;
  sg.tile = P->sy_tile.p; sg.xlist = P->sy_xlist.p; sg.chunk_base = P->sy_chunk_base.p; sg.chunk_n = P->sy_chunk_n.p;
  sg.meta = P->sy_meta.p; sg.rinfo = P->sy_rinfo.p; sg.src_slot = P->sy_src.p; sg.diag_slot = P->sy_diag.p;
  sg.val = P->sy_val.p;
  sg.xoff = nullptr; sg.xbidx = nullptr;
  if (P->g.world > 1 && P->bx_ready) {      // boundary exchange (prepare() made the lists): where every staged column of every tile lives
    const int rows_per = P->g.rows_per, rank = P->g.rank;
    std::vector<int> xoff(xlist.size());
    bool fits = true;
    for (size_t i = 0; i < xlist.size() && fits; ++i) {
      const int v = xlist[i], k = v / rows_per;
      if (k == rank) xoff[i] = k * P->g.pipe_seg + (v - k * rows_per) * 6;
      else if (P->h_bpos[(size_t)v] >= 0) xoff[i] = -1 - (k * P->bx_cseg + 6 * P->h_bpos[(size_t)v]);
      else fits = false;          // (cannot happen: the far end of a cut edge is a boundary row of its rank)
    }
    if (fits) {
      std::vector<int> xbidx(xlist.size(), -1);
      for (size_t i = 0; i < xlist.size(); ++i) if (xlist[i] / rows_per == rank) xbidx[i] = P->h_bpos[(size_t)xlist[i]];
      HIP_TRY(P->sy_xoff.upload(xoff, s)); sg.xoff = P->sy_xoff.p;
      HIP_TRY(P->sy_xbidx.upload(xbidx, s)); sg.xbidx = P->sy_xbidx.p;
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  lap("index uploads + sync");
  P->h_sym_of_old.assign(P->g.n_slots, -1);
  for (int t = 0; t < n_slots; ++t) if (src_slot[t] >= 0) P->h_sym_of_old[src_slot[t]] = t;
  {   // for the row kernel that linearises into this form (k_linearize_symout): diagonal slots are written by the damping, not by it
    std::vector<int> dst(P->h_sym_of_old);
    for (int v = 0; v < N; ++v) dst[P->h_row_slot_begin[v]] = -1;
    HIP_TRY(P->sy_dst.upload(dst, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  P->sym_ready = true;
  P->sym_stale = true;
  P->sym_interior_fraction = E ? (double)interior_edges / E : 0.0;
  P->sym_stored_slots = stored;
  if (getenv("PGO_VERBOSE"))
    std::fprintf(stderr, "[pgo] sym: %d tiles (<= %d rows), %.1f %% interior edges, %lld stored blocks (%.2f of N + 2E), %d chunks, x_cap %d, %.1f ms\n",
                 T, row_cap, 100.0 * P->sym_interior_fraction, stored, (double)stored / (N + 2.0 * E), sg.n_chunks, x_cap, 1e3 * seconds_since(t0));
  return PGO_OK;
}

pgo::DeviceGraph sym_view(const pgo_problem* P) {
  pgo::DeviceGraph gs = P->g;
  gs.bsr_val = P->sym.val;
  gs.row_slot_begin = P->sy_diag.p;
  if (P->g.cluster > 1) gs.cl_slot = P->sy_cl_slot.p;
  return gs;
}

// The symmetric form becomes the only storage of this LM session: the blocks of the linearisation that just ran (incidence-slot
// kernels, iteration zero) are copied once, the cluster lists are re-indexed; from here on the linearisation writes the form itself.
int sym_enter_storage(pgo_problem* P) {
  if (P->g.cluster > 1) {
    std::vector<int> cl(P->h_cl_slot.size());
    for (size_t i = 0; i < cl.size(); ++i) {
      cl[i] = P->h_sym_of_old[P->h_cl_slot[i]];
      if (cl[i] < 0) return set_error(PGO_ERR_UNSUPPORTED, "internal: a cluster block has no slot in the symmetric tile form");
    }
    HIP_TRY(P->sy_cl_slot.upload(cl, P->stream));
  }
  pgo::launch_sym_repack(P->g, P->sym, P->stream, 0);
  P->sym_stale = false;
  P->sym_storage = true;
  if (P->g.bx[0] && !P->sym.xoff) P->g.bx[0] = P->g.bx[1] = nullptr;      // (the form has no offsets for the boundary exchange: whole segments)
  return PGO_OK;
}
