// pgo_sym_host.cpp — row partition and slot layout of the symmetric tile form, host only (see pgo_sym_host.h).
#include "pgo_sym_host.h"

#include <algorithm>
#include <chrono>
#include <numeric>

#include "pgo_pool.h"

namespace pgo {

namespace {
// Tiles: the natural order cut into runs of <= 0.85 * (row cap, weight cap), then greedy refinement — a pose moves to the tile
// that holds most of its neighbours while the caps allow (pose-graph ids follow the trajectory, so the runs are already local;
// the refinement pulls the loop-closure partners together: BASELINE config 4 goes from 56 % to 72-78 % interior edges).
void partition_rows(int N, int lo, int hi, int unit, const std::vector<int>& adj_ptr, const std::vector<int>& adj, int row_cap, long long w_cap, std::vector<int>& part, int& T) {
  // part[v] = -1 outside [lo, hi).  The items that move are UNITS: `unit` consecutive poses starting at lo + unit * k.
  part.assign(N, -1);
  const int nu = (hi - lo + unit - 1) / unit;
  auto u_lo = [&](int k) { return lo + unit * k; };
  auto u_hi = [&](int k) { return std::min(hi, lo + unit * (k + 1)); };
  auto u_w = [&](int k) { long long w = 0; for (int v = u_lo(k); v < u_hi(k); ++v) w += 1 + adj_ptr[v + 1] - adj_ptr[v]; return w; };
  const int r0 = std::max(unit, (int)(0.85 * row_cap));
  const long long w0 = std::max<long long>(1, (long long)(0.85 * w_cap));
  int t = 0, r = 0;
  long long w = 0;
  for (int k = 0; k < nu; ++k) {
    const long long wv = u_w(k);
    const int nr = u_hi(k) - u_lo(k);
    if (r > 0 && (r + nr > r0 || w + wv > w0)) { ++t; r = 0; w = 0; }
    for (int v = u_lo(k); v < u_hi(k); ++v) part[v] = t;
    r += nr; w += wv;
  }
  T = nu > 0 ? t + 1 : 0;
  std::vector<int> rows(T, 0);
  std::vector<long long> wt(T, 0);
  for (int v = lo; v < hi; ++v) { ++rows[part[v]]; wt[part[v]] += 1 + adj_ptr[v + 1] - adj_ptr[v]; }
  std::vector<int> cnt(T, 0), touched;
  for (int pass = 0; pass < 6; ++pass) {
    int moved = 0;
    for (int k = 0; k < nu; ++k) {
      const int cur = part[u_lo(k)];
      touched.clear();
      for (int v = u_lo(k); v < u_hi(k); ++v)
        for (int j = adj_ptr[v]; j < adj_ptr[v + 1]; ++j) {
          const int tv = part[adj[j]];
          if (tv < 0) continue;                      // another rank's row
          if (cnt[tv]++ == 0) touched.push_back(tv);
        }
      int best = cur, best_c = cnt[cur];
      for (int tv : touched) if (cnt[tv] > best_c || (cnt[tv] == best_c && tv < best && best != cur)) { best = tv; best_c = cnt[tv]; }
      for (int tv : touched) cnt[tv] = 0;
      const long long wv = u_w(k);
      const int nr = u_hi(k) - u_lo(k);
      if (best != cur && rows[best] + nr <= row_cap && wt[best] + wv <= w_cap && rows[cur] > nr) {
        rows[cur] -= nr; rows[best] += nr; wt[cur] -= wv; wt[best] += wv;
        for (int v = u_lo(k); v < u_hi(k); ++v) part[v] = best;
        ++moved;
      }
    }
    if (moved < nu / 500) break;
  }
}

}  // namespace

void sym_build_host(int N, int E, const int* ia, const int* ib, const int* row_slot_begin, const SymHostParams& prm, SymHostLayout* out) {
  const int lo = std::max(0, prm.row_lo), hi = prm.row_hi < 0 ? N : std::min(N, prm.row_hi);
  const int unit = std::max(1, prm.unit);
  auto owned = [&](int v) { return v >= lo && v < hi; };
  // adjacency of the owned rows (both directions; the neighbours may be anybody's)
  std::vector<int> adj_ptr(N + 1, 0);
  for (int e = 0; e < E; ++e) { if (owned(ia[e])) ++adj_ptr[ia[e] + 1]; if (owned(ib[e])) ++adj_ptr[ib[e] + 1]; }
  for (int v = 0; v < N; ++v) adj_ptr[v + 1] += adj_ptr[v];
  std::vector<int> adj(adj_ptr[N]), fillp(adj_ptr.begin(), adj_ptr.end() - 1);
  for (int e = 0; e < E; ++e) { if (owned(ia[e])) adj[fillp[ia[e]]++] = ib[e]; if (owned(ib[e])) adj[fillp[ib[e]]++] = ia[e]; }
  const int row_cap = prm.row_cap;
  const long long w_cap = prm.w_cap;
  std::vector<int> part;
  int T0 = 0;
  const auto t_part = std::chrono::steady_clock::now();
  partition_rows(N, lo, hi, unit, adj_ptr, adj, row_cap, w_cap, part, T0);
  out->ms_partition = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_part).count();
  const auto t_lay = std::chrono::steady_clock::now();
  // compact tile ids, rows per tile ascending
  std::vector<int> tile_of(T0, -1);
  int T = 0;
  for (int v = lo; v < hi; ++v) if (tile_of[part[v]] < 0) tile_of[part[v]] = T++;
  for (int v = lo; v < hi; ++v) part[v] = tile_of[part[v]];
  if (prm.sort_tiles) {
    // largest tiles first: work-groups are handed out in index order, so the small tiles fill the tail of the launch
    std::vector<long long> wt(T, 0);
    for (int v = lo; v < hi; ++v) wt[part[v]] += 1 + adj_ptr[v + 1] - adj_ptr[v];
    std::vector<int> order(T), rank_of(T);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return wt[a] > wt[b]; });
    for (int k = 0; k < T; ++k) rank_of[order[k]] = k;
    for (int v = lo; v < hi; ++v) part[v] = rank_of[part[v]];
  }
  std::vector<std::vector<int>> trow(T);
  for (int v = lo; v < hi; ++v) trow[part[v]].push_back(v);      // (ascending pose ids: the poses of a unit sit in consecutive lanes, the unit's first one in a lane that is a multiple of `unit` when every unit of the tile is whole)

  // old slots of every edge (prepare(): the row's diagonal first, then its incidences in edge order)
  std::vector<int> fill(N, 0), beg_slot(E, -1), end_slot(E, -1);
  for (int v = lo; v < hi; ++v) fill[v] = row_slot_begin[v] + 1;
  for (int e = 0; e < E; ++e) { if (owned(ia[e])) beg_slot[e] = fill[ia[e]]++; if (owned(ib[e])) end_slot[e] = fill[ib[e]]++; }
  // incidences per owned row: (edge, side)
  std::vector<int> inc_ptr(N + 1, 0);
  for (int e = 0; e < E; ++e) { if (owned(ia[e])) ++inc_ptr[ia[e] + 1]; if (owned(ib[e])) ++inc_ptr[ib[e] + 1]; }
  for (int v = 0; v < N; ++v) inc_ptr[v + 1] += inc_ptr[v];
  std::vector<int> inc(inc_ptr[N]);
  { std::vector<int> f(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int e = 0; e < E; ++e) { if (owned(ia[e])) inc[f[ia[e]]++] = 2 * e; if (owned(ib[e])) inc[f[ib[e]]++] = 2 * e + 1; } }

  // ---- per-tile layout, tiles in parallel (every index below is relative to the tile; offsets are added afterwards) ----
  struct TileOut {
    std::vector<int> xlist, chunk_n, src;        // src: per stored slot incl. chunk padding
    std::vector<uint32_t> meta, rinfo;
    std::vector<int> diag_local;                 // per row: stored slot of its diagonal block
    int nr = 0, nx = 0, total = 0, L = 0;
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
          const int e = inc[j] >> 1, o = (inc[j] & 1) ? ia[e] : ib[e];
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
        slots.push_back(Slot{row_slot_begin[v], (uint32_t)i | ((uint32_t)pgo::SIDE_DIAG << 12) | ((uint32_t)i << 23), i, -1});
        for (int j = inc_ptr[v]; j < inc_ptr[v + 1]; ++j) {
          const int e = inc[j] >> 1, end_side = inc[j] & 1;
          const int o = end_side ? ia[e] : ib[e];
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
      const int padded_total = (L - 1) * pgo::SYM_LANES + (total - (L - 1) * pgo::SYM_LANES + 63) / 64 * 64;
      O.meta.assign(padded_total, 0u);
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
      }
      reset_local();
    }
  });
  // ---- offsets and the global arrays ----
  std::vector<pgo::SymTile>& tiles = out->tiles;
  tiles.assign(T, pgo::SymTile{});
  std::vector<int>&xlist = out->xlist, &chunk_base = out->chunk_base, &chunk_n = out->chunk_n, &src_slot = out->src_slot, &diag_slot = out->diag_slot;
  diag_slot.assign(N, 0);
  std::vector<uint32_t>&meta = out->meta, &rinfo = out->rinfo;
  int& x_cap = out->x_cap;
  long long& interior_edges = out->interior_edges;
  long long& stored = out->stored;
  x_cap = 0; interior_edges = 0; stored = 0;
  {
    size_t nxs = 0, nsl = 0, nch = 0;
    for (int t = 0; t < T; ++t) {
      const TileOut& O = outs[t];
      if (O.unfit) { out->unfit = O.unfit; out->unfit_tile = t; return; }
      pgo::SymTile& TT = tiles[t];
      TT.chunk0 = (int)nch; TT.nchunks = O.L; TT.x0 = (int)nxs; TT.nx = O.nx; TT.nrows = O.nr; TT.total = O.total;
      TT.base0 = (int)nsl; TT.n0 = O.L > 0 ? O.chunk_n[0] : 0;
      TT.base1 = (int)nsl + pgo::SYM_LANES; TT.n1 = O.L > 1 ? O.chunk_n[1] : 0;
      TT.pad[0] = TT.pad[1] = 0;
      nxs += O.xlist.size(); nsl += O.meta.size(); nch += O.L;
      x_cap = std::max(x_cap, O.nx);
      interior_edges += O.interior; stored += O.total;
    }
    xlist.resize(nxs); meta.resize(nsl); src_slot.resize(nsl);
    chunk_base.resize(nch); chunk_n.resize(nch); rinfo.resize(nch * pgo::SYM_LANES);
    pgo::HostPool::get().run(nthreads, [&](int th) {
      for (int t = th; t < T; t += nthreads) {
        const TileOut& O = outs[t];
        const pgo::SymTile& TT = tiles[t];
        std::copy(O.xlist.begin(), O.xlist.end(), xlist.begin() + TT.x0);
        std::copy(O.meta.begin(), O.meta.end(), meta.begin() + TT.base0);
        std::copy(O.src.begin(), O.src.end(), src_slot.begin() + TT.base0);
        std::copy(O.rinfo.begin(), O.rinfo.end(), rinfo.begin() + (size_t)TT.chunk0 * pgo::SYM_LANES);
        for (int c = 0; c < O.L; ++c) { chunk_base[TT.chunk0 + c] = TT.base0 + c * pgo::SYM_LANES; chunk_n[TT.chunk0 + c] = O.chunk_n[c]; }
        for (int i = 0; i < O.nr; ++i) diag_slot[trow[t][i]] = TT.base0 + O.diag_local[i];
      }
    });
  }
  out->n_slots = (int)meta.size();
  out->ms_layout = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_lay).count();
}

}  // namespace pgo
