// pgo_sym.cpp — host side of the symmetric tile form (pgo_sym.h): row tiles with graph locality, the row-after-row layout of the
// stored slots in chunks of 256, and every index the kernels need.  One-off per topology (one rank; built when a PCG solve of a
// large graph starts and can repay it: 44 ms at 100 k poses / 1 M edges).
#include <algorithm>
#include <numeric>

#include "pgo_internal.h"

namespace {

// Tiles: the natural order cut into runs of <= 0.85 * (row cap, weight cap), then greedy refinement — a pose moves to the tile
// that holds most of its neighbours while the caps allow (pose-graph ids follow the trajectory, so the runs are already local;
// the refinement pulls the loop-closure partners together: BASELINE config 4 goes from 56 % to 72-78 % interior edges).
void partition_rows(int N, const std::vector<int>& adj_ptr, const std::vector<int>& adj, int row_cap, long long w_cap, std::vector<int>& part, int& T) {
  part.assign(N, 0);
  const int r0 = std::max(1, (int)(0.85 * row_cap));
  const long long w0 = std::max<long long>(1, (long long)(0.85 * w_cap));
  int t = 0, r = 0;
  long long w = 0;
  for (int v = 0; v < N; ++v) {
    const int wv = 1 + adj_ptr[v + 1] - adj_ptr[v];
    if (r > 0 && (r >= r0 || w + wv > w0)) { ++t; r = 0; w = 0; }
    part[v] = t; ++r; w += wv;
  }
  T = t + 1;
  std::vector<int> rows(T, 0);
  std::vector<long long> wt(T, 0);
  for (int v = 0; v < N; ++v) { ++rows[part[v]]; wt[part[v]] += 1 + adj_ptr[v + 1] - adj_ptr[v]; }
  std::vector<int> cnt(T, 0), touched;
  for (int pass = 0; pass < 6; ++pass) {
    int moved = 0;
    for (int v = 0; v < N; ++v) {
      const int cur = part[v];
      touched.clear();
      for (int j = adj_ptr[v]; j < adj_ptr[v + 1]; ++j) {
        const int tv = part[adj[j]];
        if (cnt[tv]++ == 0) touched.push_back(tv);
      }
      int best = cur, best_c = cnt[cur];
      for (int tv : touched) if (cnt[tv] > best_c || (cnt[tv] == best_c && tv < best && best != cur)) { best = tv; best_c = cnt[tv]; }
      for (int tv : touched) cnt[tv] = 0;
      const int wv = 1 + adj_ptr[v + 1] - adj_ptr[v];
      if (best != cur && rows[best] < row_cap && wt[best] + wv <= w_cap && rows[cur] > 1) {
        --rows[cur]; ++rows[best]; wt[cur] -= wv; wt[best] += wv; part[v] = best; ++moved;
      }
    }
    if (moved < N / 500) break;
  }
}

}  // namespace

bool sym_wanted(const pgo_problem* P) {
  const char* e = getenv("PGO_SYM");
  if (e && e[0] == '0') return false;
  if (P->g.world != 1 || (P->comm && P->comm->world > 1) || P->use_graph) return false;
  if (e && e[0] == '1') return true;
  // the universal stream serves the graphs below its slot limit (latency-bound kernels of a few microseconds: nothing to gain
  // from fewer bytes there); above it the host-driven CG runs and the SpMV is bandwidth-bound
  const long long limit = getenv("PGO_UNI_MAX_SLOTS") ? atoll(getenv("PGO_UNI_MAX_SLOTS")) : 600000;
  return (long long)P->g.n_slots > limit;
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
  const int N = (int)P->pp.size(), E = (int)P->ia.size();
  hipStream_t s = P->stream;
  // adjacency (both directions)
  std::vector<int> adj_ptr(N + 1, 0);
  for (int e = 0; e < E; ++e) { ++adj_ptr[P->ia[e] + 1]; ++adj_ptr[P->ib[e] + 1]; }
  for (int v = 0; v < N; ++v) adj_ptr[v + 1] += adj_ptr[v];
  std::vector<int> adj(adj_ptr[N]), fillp(adj_ptr.begin(), adj_ptr.end() - 1);
  for (int e = 0; e < E; ++e) { adj[fillp[P->ia[e]]++] = P->ib[e]; adj[fillp[P->ib[e]]++] = P->ia[e]; }
  // tile caps: up to 256 rows; enough tiles to fill the chip twice on small graphs; weight cap 1.35 x the average rows' weight
  const char* re = getenv("PGO_SYM_ROWS");
  int row_cap = re ? atoi(re) : std::min(256, std::max(32, N / 384));
  row_cap = std::max(8, std::min(row_cap, (int)pgo::SYM_LANES));
  const double avg_w = (double)(N + 2LL * E) / std::max(1, N);
  const double w_mult = getenv("PGO_SYM_WCAP") ? atof(getenv("PGO_SYM_WCAP")) : 0.95;
  long long w_cap = std::max<long long>(64, (long long)(w_mult * row_cap * avg_w));
  if (getenv("PGO_SYM_TILES")) w_cap = std::max<long long>(64, (long long)((N + 2.0 * E) / (0.85 * atof(getenv("PGO_SYM_TILES")))));
  std::vector<int> part;
  int T0 = 0;
  lap("adjacency");
  partition_rows(N, adj_ptr, adj, row_cap, w_cap, part, T0);
  lap("partition");
  // compact tile ids, rows per tile ascending
  std::vector<int> tile_of(T0, -1);
  int T = 0;
  for (int v = 0; v < N; ++v) if (tile_of[part[v]] < 0) tile_of[part[v]] = T++;
  for (int v = 0; v < N; ++v) part[v] = tile_of[part[v]];
  if (!(getenv("PGO_SYM_NOSORT"))) {
    // largest tiles first: work-groups are handed out in index order, so the small tiles fill the tail of the launch
    std::vector<long long> wt(T, 0);
    for (int v = 0; v < N; ++v) wt[part[v]] += 1 + adj_ptr[v + 1] - adj_ptr[v];
    std::vector<int> order(T), rank_of(T);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wt[a] > wt[b]; });
    for (int k = 0; k < T; ++k) rank_of[order[k]] = k;
    for (int v = 0; v < N; ++v) part[v] = rank_of[part[v]];
  }
  std::vector<std::vector<int>> trow(T);
  for (int v = 0; v < N; ++v) trow[part[v]].push_back(v);

  // old slots of every edge (prepare(): the row's diagonal first, then its incidences in edge order)
  std::vector<int> fill(N), beg_slot(E), end_slot(E);
  for (int v = 0; v < N; ++v) fill[v] = P->h_row_slot_begin[v] + 1;
  for (int e = 0; e < E; ++e) { beg_slot[e] = fill[P->ia[e]]++; end_slot[e] = fill[P->ib[e]]++; }
  // incidences per row: (edge, side)
  std::vector<int> inc_ptr(N + 1, 0);
  for (int e = 0; e < E; ++e) { ++inc_ptr[P->ia[e] + 1]; ++inc_ptr[P->ib[e] + 1]; }
  for (int v = 0; v < N; ++v) inc_ptr[v + 1] += inc_ptr[v];
  std::vector<int> inc(inc_ptr[N]);
  { std::vector<int> f(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int e = 0; e < E; ++e) { inc[f[P->ia[e]]++] = 2 * e; inc[f[P->ib[e]]++] = 2 * e + 1; } }

  // ---- per-tile layout, tiles in parallel (every index below is relative to the tile; offsets are added afterwards) ----
  struct TileOut {
    std::vector<int> xlist, chunk_n, src;        // src: per stored slot incl. chunk padding
    std::vector<uint32_t> meta, meta2, rinfo, rinfo2;
    std::vector<int> diag_local;                 // per row: stored slot of its diagonal block
    int nr = 0, nx = 0, total = 0, L = 0, e_cap = 1;
    long long interior = 0;
    const char* unfit = nullptr;
  };
  std::vector<TileOut> outs(T);
  struct Slot { int src; uint32_t m; int row, dst_row; };   // m without vpos; dst_row: local row an interior slot's v goes to (-1 none)
  const int nthreads = std::max(1, std::min(pgo::HostPool::get().width(), std::min(16, T / 8 + 1)));
  pgo::HostPool::get().run(nthreads, [&](int th) {
    std::vector<int> local(N, -1);             // pose -> LDS index inside the tile being built
    std::vector<Slot> slots;
    std::vector<int> ghosts;
    std::vector<std::pair<int, int>> vs;       // (destination row, producing lane) of a chunk's v entries
    for (int t = th; t < T; t += nthreads) {
      TileOut& O = outs[t];
      const std::vector<int>& rows = trow[t];
      const int nr = (int)rows.size();
      O.nr = nr;
      for (int i = 0; i < nr; ++i) local[rows[i]] = i;
      // ghosts: far ends of cut edges, ascending pose id
      ghosts.clear();
      for (int v : rows)
        for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j) {
          const int e = inc[j] >> 1, o = (inc[j] & 1) ? P->ia[e] : P->ib[e];
          if (part[o] != t) ghosts.push_back(o);
        }
      std::sort(ghosts.begin(), ghosts.end());
      ghosts.erase(std::unique(ghosts.begin(), ghosts.end()), ghosts.end());
      const int nx = nr + (int)ghosts.size();
      O.nx = nx;
      auto reset_local = [&] { for (int v : rows) local[v] = -1; for (int gp : ghosts) local[gp] = -1; };
      if (nx > pgo::SYM_X_MAX) { O.unfit = "stages too many columns"; reset_local(); continue; }
      for (size_t gi = 0; gi < ghosts.size(); ++gi) local[ghosts[gi]] = nr + (int)gi;
      O.xlist.assign(rows.begin(), rows.end());
      O.xlist.insert(O.xlist.end(), ghosts.begin(), ghosts.end());
      // stored slots, row after row: the diagonal, then the row's incidences in edge order (interior edges once, by the begin side)
      slots.clear();
      for (int i = 0; i < nr; ++i) {
        const int v = rows[i];
        slots.push_back(Slot{P->h_row_slot_begin[v], (uint32_t)i | ((uint32_t)pgo::SIDE_DIAG << 12) | ((uint32_t)i << 23), i, -1});
        for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j) {
          const int e = inc[j] >> 1, end_side = inc[j] & 1;
          const int o = end_side ? P->ia[e] : P->ib[e];
          const bool interior = part[o] == t;
          if (interior && end_side) continue;
          if (interior) ++O.interior;
          slots.push_back(Slot{end_side ? end_slot[e] : beg_slot[e],
                               (uint32_t)local[o] | ((uint32_t)(end_side ? pgo::SIDE_END : pgo::SIDE_BEGIN) << 12) | (interior ? (1u << 14) : 0u) | ((uint32_t)i << 23),
                               i, interior ? local[o] : -1});
        }
      }
      const int total = (int)slots.size();
      O.total = total;
      const int L = (total + pgo::SYM_LANES - 1) / pgo::SYM_LANES;
      O.L = L;
      O.diag_local.assign(nr, 0);
      O.chunk_n.resize(L);
      O.rinfo.assign((size_t)L * pgo::SYM_LANES, 0u);
      O.rinfo2.assign((size_t)L * pgo::SYM_LANES, 0u);
      const int padded_total = (L - 1) * pgo::SYM_LANES + (total - (L - 1) * pgo::SYM_LANES + 63) / 64 * 64;
      O.meta.assign(padded_total, 0u);
      O.meta2.assign(padded_total, 0xFFFFFFFFu);
      O.src.assign(padded_total, -1);
      for (int c = 0; c < L && !O.unfit; ++c) {
        const int lo = c * pgo::SYM_LANES, n = std::min((int)pgo::SYM_LANES, total - lo);
        const int base = lo;                           // relative to the tile (full chunks are 256 = 4 x 64 slots)
        O.chunk_n[c] = n;
        uint32_t* ri = &O.rinfo[(size_t)c * pgo::SYM_LANES];
        vs.clear();
        for (int l = 0; l < n; ++l) {
          const Slot& sl = slots[lo + l];
          O.src[base + l] = sl.src;
          O.meta[base + l] = sl.m;
          if (((sl.m >> 12) & 3u) == (uint32_t)pgo::SIDE_DIAG) O.diag_local[sl.row] = base + l;
          uint32_t& w = ri[sl.row];                    // u range of the slot's row: [ub, ub + uc)
          if (((w >> 8) & 0x1FFu) == 0) w = (w & ~0xFFu) | (uint32_t)l;
          w += 1u << 8;
          if (sl.dst_row >= 0) vs.push_back({sl.dst_row, l});
        }
        std::sort(vs.begin(), vs.end());
        for (size_t k = 0; k < vs.size(); ++k) {
          O.meta[base + vs[k].second] |= (uint32_t)k << 15;
          uint32_t& w = ri[vs[k].first];
          if ((w >> 25) == 0) w = (w & ~(0xFFu << 17)) | ((uint32_t)k << 17);
          if ((w >> 25) == 127) { O.unfit = "has a row that receives more than 127 mirrored products in one chunk"; break; }
          w += 1u << 25;
        }
        if (O.unfit) break;
        // exchange entries of the linearisation (k_linearize_sym): per destination row, ascending: the tails of its (row, wave)
        // runs, then the mirrored contributions it receives — contiguous, so the row's lanes add one range
        uint32_t* r2 = &O.rinfo2[(size_t)c * pgo::SYM_LANES];
        int pos = 0;
        size_t kv = 0;
        for (int r = 0; r < nr; ++r) {
          const uint32_t w = ri[r];
          const int ub = (int)(w & 0xFFu), uc = (int)((w >> 8) & 0x1FFu);
          const int e0 = pos;
          if (uc > 0) {
            const int last = ub + uc - 1;
            for (int wv = ub >> 6; wv <= (last >> 6); ++wv) {
              const int tail = std::min(wv * 64 + 63, last);
              O.meta2[base + tail] = (O.meta2[base + tail] & 0xFFFF0000u) | (uint32_t)pos++;
            }
          }
          while (kv < vs.size() && vs[kv].first == r) {
            const int l = vs[kv].second;
            O.meta2[base + l] = (O.meta2[base + l] & 0x0000FFFFu) | ((uint32_t)pos++ << 16);
            ++kv;
          }
          r2[r] = (uint32_t)e0 | ((uint32_t)(pos - e0) << 16);
        }
        O.e_cap = std::max(O.e_cap, pos);
      }
      reset_local();
    }
  });
  // ---- offsets and the global arrays ----
  std::vector<pgo::SymTile> tiles(T);
  std::vector<int> xlist, chunk_base, chunk_n, src_slot, diag_slot(N, 0);
  std::vector<uint32_t> meta, rinfo, meta2, rinfo2;
  int e_cap = 1, x_cap = 0;
  long long interior_edges = 0, stored = 0;
  {
    size_t nxs = 0, nsl = 0, nch = 0;
    for (int t = 0; t < T; ++t) {
      const TileOut& O = outs[t];
      if (O.unfit) {
        if (verbose) std::fprintf(stderr, "[pgo] sym: tile %d %s: the incidence-slot kernels stay\n", t, O.unfit);
        return PGO_OK;
      }
      pgo::SymTile& TT = tiles[t];
      TT.chunk0 = (int)nch; TT.nchunks = O.L; TT.x0 = (int)nxs; TT.nx = O.nx; TT.nrows = O.nr; TT.total = O.total;
      TT.base0 = (int)nsl; TT.n0 = O.L > 0 ? O.chunk_n[0] : 0;
      TT.base1 = (int)nsl + pgo::SYM_LANES; TT.n1 = O.L > 1 ? O.chunk_n[1] : 0;
      TT.pad[0] = TT.pad[1] = 0;
      nxs += O.xlist.size(); nsl += O.meta.size(); nch += O.L;
      x_cap = std::max(x_cap, O.nx); e_cap = std::max(e_cap, O.e_cap);
      interior_edges += O.interior; stored += O.total;
    }
    xlist.resize(nxs); meta.resize(nsl); meta2.resize(nsl); src_slot.resize(nsl);
    chunk_base.resize(nch); chunk_n.resize(nch); rinfo.resize(nch * pgo::SYM_LANES); rinfo2.resize(nch * pgo::SYM_LANES);
    pgo::HostPool::get().run(nthreads, [&](int th) {
      for (int t = th; t < T; t += nthreads) {
        const TileOut& O = outs[t];
        const pgo::SymTile& TT = tiles[t];
        std::copy(O.xlist.begin(), O.xlist.end(), xlist.begin() + TT.x0);
        std::copy(O.meta.begin(), O.meta.end(), meta.begin() + TT.base0);
        std::copy(O.meta2.begin(), O.meta2.end(), meta2.begin() + TT.base0);
        std::copy(O.src.begin(), O.src.end(), src_slot.begin() + TT.base0);
        std::copy(O.rinfo.begin(), O.rinfo.end(), rinfo.begin() + (size_t)TT.chunk0 * pgo::SYM_LANES);
        std::copy(O.rinfo2.begin(), O.rinfo2.end(), rinfo2.begin() + (size_t)TT.chunk0 * pgo::SYM_LANES);
        for (int c = 0; c < O.L; ++c) { chunk_base[TT.chunk0 + c] = TT.base0 + c * pgo::SYM_LANES; chunk_n[TT.chunk0 + c] = O.chunk_n[c]; }
        for (int i = 0; i < O.nr; ++i) diag_slot[trow[t][i]] = TT.base0 + O.diag_local[i];
      }
    });
  }
  lap("tile layout");
  const int n_slots = (int)meta.size();
  if (T > P->g.pq_cap) {       // the p'q partials of the tiles ride in the slots of the row partition's work-groups
    if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] sym: %d tiles > %d partial-sum slots: the incidence-slot kernels stay\n", T, P->g.pq_cap);
    return PGO_OK;
  }
  if (pgo::sym_lds_bytes(pgo::SymGraph{T, (int)chunk_base.size(), n_slots, x_cap}) > 160 * 1024 - 1024) return PGO_OK;
  { pgo::SymGraph probe{}; probe.x_cap = x_cap; probe.e_cap = e_cap; P->sym_lin_fits = pgo::sym_lin_lds_bytes(probe) <= 160 * 1024 - 1024; }

  HIP_TRY(P->sy_tile.upload(tiles, s));
  HIP_TRY(P->sy_xlist.upload(xlist, s));
  HIP_TRY(P->sy_chunk_base.upload(chunk_base, s));
  HIP_TRY(P->sy_chunk_n.upload(chunk_n, s));
  HIP_TRY(P->sy_meta.upload(meta, s));
  HIP_TRY(P->sy_rinfo.upload(rinfo, s));
  HIP_TRY(P->sy_src.upload(src_slot, s));
  HIP_TRY(P->sy_diag.upload(diag_slot, s));
  HIP_TRY(P->sy_meta2.upload(meta2, s));
  HIP_TRY(P->sy_rinfo2.upload(rinfo2, s));
  HIP_TRY(P->sy_val.alloc((size_t)n_slots * 36));
  HIP_TRY(P->sy_val.zero(s));
  pgo::SymGraph& sg = P->sym;
  sg.n_tiles = T; sg.n_chunks = (int)chunk_base.size(); sg.n_slots = n_slots; sg.x_cap = x_cap;
  sg.tile = P->sy_tile.p; sg.xlist = P->sy_xlist.p; sg.chunk_base = P->sy_chunk_base.p; sg.chunk_n = P->sy_chunk_n.p;
  sg.meta = P->sy_meta.p; sg.rinfo = P->sy_rinfo.p; sg.src_slot = P->sy_src.p; sg.diag_slot = P->sy_diag.p;
  sg.meta2 = P->sy_meta2.p; sg.rinfo2 = P->sy_rinfo2.p; sg.e_cap = e_cap; sg.val = P->sy_val.p;
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
    std::fprintf(stderr, "[pgo] sym: %d tiles (<= %d rows), %.1f %% interior edges, %lld stored blocks (%.2f of N + 2E), %d chunks, x_cap %d, e_cap %d, %.1f ms\n",
                 T, row_cap, 100.0 * P->sym_interior_fraction, stored, (double)stored / (N + 2.0 * E), sg.n_chunks, x_cap, e_cap, 1e3 * seconds_since(t0));
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
// kernels, iteration zero) are copied once, the cluster lists are re-indexed; from here on k_linearize_sym writes the form itself.
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
  return PGO_OK;
}
