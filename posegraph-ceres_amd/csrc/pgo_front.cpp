// pgo_front.cpp — host symbolic phase of the multifrontal GPU Cholesky (see pgo_front.h).
//
//  1. nested-dissection ordering (pgo_direct.cpp), block symbolic factorisation through the elimination tree;
//  2. fundamental supernodes, then relaxed amalgamation: a child is merged into its parent when the merged supernode is
//     small or the explicit zeros stay below a fraction of its entries (CHOLMOD's idea with much looser thresholds — an
//     MI355X retires the extra flops in microseconds, a tree level costs tens of microseconds of dependent launches);
//  3. postorder renumbering so every supernode owns consecutive columns; row structures recomputed in that numbering;
//  4. fronts numbered level by level (leaves first), child -> parent index maps, BSR source lists per front block;
//  5. the launch schedule: per level and 48-column panel step ONE launch over all fronts of the level, plus one GEMM launch
//     behind every finished 192-column outer panel / for the Schur updates.
#include "pgo_front.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "pgo_direct.h"
#include "pgo_tuning.h"

namespace pgo {


bool front_analyze(int N, const std::vector<int>& ia, const std::vector<int>& ib, int n_slots,
                   const std::vector<int>& slot_row, const std::vector<int>& slot_col,
                   const std::vector<uint8_t>& slot_side, long long max_bytes, FrontSymbolic* out, int small_max,
                   const std::vector<uint8_t>* is_point) {
  FrontSymbolic& S = *out;
  S = FrontSymbolic();
  S.n = N;
  if (N <= 0) return false;
  const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto t_phase = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[pgo] front analysis: %-30s %.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_phase).count());
    t_phase = now;
  };
  // ---- 1. ordering + adjacency in that ordering ----
  std::vector<int> perm0;
  if (!nested_dissection_order(N, ia, ib, &perm0, is_point)) return false;
  std::vector<int> iperm0(N);
  for (int k = 0; k < N; ++k) iperm0[perm0[k]] = k;
  std::vector<int> aptr(N + 1, 0), aidx;   // full neighbour lists in the perm0 numbering (duplicates allowed)
  for (size_t e = 0; e < ia.size(); ++e) {
    if (ia[e] == ib[e]) continue;
    ++aptr[iperm0[ia[e]] + 1];
    ++aptr[iperm0[ib[e]] + 1];
  }
  for (int v = 0; v < N; ++v) aptr[v + 1] += aptr[v];
  aidx.resize(aptr[N]);
  {
    std::vector<int> fill(aptr.begin(), aptr.end() - 1);
    for (size_t e = 0; e < ia.size(); ++e) {
      if (ia[e] == ib[e]) continue;
      const int a = iperm0[ia[e]], b = iperm0[ib[e]];
      aidx[fill[a]++] = b;
      aidx[fill[b]++] = a;
    }
  }
  phase("ordering + adjacency");
  // ---- 2. column structures through the elimination tree (only the sizes and the parents are kept) ----
  std::vector<int> parent(N, -1), st_len(N, 0), nchild(N, 0);
  {
    // flat storage (column j's structure at st_idx[st_ptr[j] .. st_ptr[j + 1])), children as sibling lists: no per-column heap blocks
    std::vector<int> st_ptr(N + 1, 0), st_idx, first_child(N, -1), next_sibling(N, -1), mark(N, -1), tmp;
    st_idx.reserve((size_t)4 * N);
    for (int j = 0; j < N; ++j) {
      tmp.clear();
      mark[j] = j;
      for (int p = aptr[j]; p < aptr[j + 1]; ++p) {
        const int i = aidx[p];
        if (i > j && mark[i] != j) { mark[i] = j; tmp.push_back(i); }
      }
      for (int c = first_child[j]; c >= 0; c = next_sibling[c])
        for (int q = st_ptr[c]; q < st_ptr[c + 1]; ++q) {
          const int i = st_idx[q];
          if (i > j && mark[i] != j) { mark[i] = j; tmp.push_back(i); }
        }
      st_len[j] = (int)tmp.size();
      if (!tmp.empty()) {
        const int par = *std::min_element(tmp.begin(), tmp.end());
        parent[j] = par;
        next_sibling[j] = first_child[par];
        first_child[par] = j;
        ++nchild[par];
      }
      st_idx.insert(st_idx.end(), tmp.begin(), tmp.end());
      st_ptr[j + 1] = (int)st_idx.size();
    }
  }
  phase("column structures");
  // ---- 3. fundamental supernodes ----
  std::vector<int> sn_start;
  sn_start.push_back(0);
  for (int j = 1; j < N; ++j) {
    if (parent[j - 1] == j && st_len[j - 1] == st_len[j] + 1 && nchild[j] == 1) continue;
    sn_start.push_back(j);
  }
  const int ns0 = (int)sn_start.size();
  sn_start.push_back(N);
  std::vector<int> sn_of(N);
  for (int s = 0; s < ns0; ++s) for (int j = sn_start[s]; j < sn_start[s + 1]; ++j) sn_of[j] = s;
  std::vector<long long> cols(ns0), rows(ns0), zeros(ns0, 0);
  std::vector<int> sparent(ns0, -1);
  std::vector<std::vector<int>> kids(ns0), members(ns0);
  for (int s = 0; s < ns0; ++s) {
    const int last = sn_start[s + 1] - 1;
    cols[s] = sn_start[s + 1] - sn_start[s];
    rows[s] = st_len[last];
    sparent[s] = parent[last] >= 0 ? sn_of[parent[last]] : -1;
    members[s].push_back(s);
  }
  for (int s = 0; s < ns0; ++s) if (sparent[s] >= 0) kids[sparent[s]].push_back(s);
  phase("supernodes");
  // ---- 4. relaxed amalgamation (parents have larger indices than their children) ----
  const double zfrac = tuning("front_zfrac", 0.25);
  const long long small = (long long)tuning("front_small", 8);
  const long long max_cols = (long long)tuning("front_maxcols", 1 << 30);
  std::vector<char> alive(ns0, 1);
  for (int p = 0; p < ns0; ++p) {
    std::vector<int> ch = kids[p], keep;
    std::sort(ch.begin(), ch.end(), [&](int a, int b) { return rows[a] != rows[b] ? rows[a] > rows[b] : a < b; });
    for (int c : ch) {
      const long long add = cols[c] * (cols[p] + rows[p] - rows[c]);
      const long long C = cols[c] + cols[p], R = rows[p];
      const long long tot = C * (C + 1) / 2 + C * R;
      const long long z = zeros[c] + zeros[p] + add;
      if (C <= max_cols && (C <= small || (double)z <= zfrac * (double)tot)) {
        cols[p] = C;
        zeros[p] = z;
        alive[c] = 0;
        members[p].insert(members[p].end(), members[c].begin(), members[c].end());
        for (int gc : kids[c]) { keep.push_back(gc); sparent[gc] = p; }
      } else {
        keep.push_back(c);
      }
    }
    kids[p].swap(keep);
  }
  phase("amalgamation");
  // ---- 5. postorder renumbering of the amalgamated tree ----
  std::vector<int> post;   // supernodes (old ids) in postorder
  post.reserve(ns0);
  {
    std::vector<std::pair<int, size_t>> stack;
    for (int r = 0; r < ns0; ++r) {
      if (!alive[r] || sparent[r] >= 0) continue;
      stack.emplace_back(r, 0);
      while (!stack.empty()) {
        const int s = stack.back().first;
        if (stack.back().second == 0) std::sort(kids[s].begin(), kids[s].end());
        if (stack.back().second < kids[s].size()) {
          const int c = kids[s][stack.back().second++];
          stack.emplace_back(c, 0);
        } else {
          post.push_back(s);
          stack.pop_back();
        }
      }
    }
  }
  const int nf = (int)post.size();
  S.nf = nf;
  std::vector<int> sn_new(ns0, -1);   // old supernode id -> postorder id
  for (int k = 0; k < nf; ++k) sn_new[post[k]] = k;
  std::vector<int> first(nf), ccount(nf), fparent(nf, -1);
  std::vector<int> newidx(N, -1);     // perm0 index -> new index
  {
    int next = 0;
    for (int k = 0; k < nf; ++k) {
      const int s = post[k];
      std::sort(members[s].begin(), members[s].end());
      first[k] = next;
      for (int m : members[s]) for (int j = sn_start[m]; j < sn_start[m + 1]; ++j) newidx[j] = next++;
      ccount[k] = next - first[k];
      fparent[k] = sparent[s] >= 0 ? sn_new[sparent[s]] : -1;
    }
    if (next != N) return false;
  }
  S.perm.resize(N);
  S.iperm.resize(N);
  for (int j = 0; j < N; ++j) { S.perm[newidx[j]] = perm0[j]; }
  for (int k = 0; k < N; ++k) S.iperm[S.perm[k]] = k;
  std::vector<int> colf(N);           // postorder front id of each new column
  for (int k = 0; k < nf; ++k) for (int j = first[k]; j < first[k] + ccount[k]; ++j) colf[j] = k;
  // adjacency in the new numbering (old vertex -> neighbours), reuse aptr/aidx through newidx
  phase("postorder");
  // ---- 6. row structures in the new numbering ----
  std::vector<std::vector<int>> R(nf);
  std::vector<std::vector<int>> fkids(nf);
  for (int k = 0; k < nf; ++k) if (fparent[k] >= 0) fkids[fparent[k]].push_back(k);
  {
    std::vector<int> mark(N, -1), tmp;
    for (int k = 0; k < nf; ++k) {
      const int last = first[k] + ccount[k] - 1;
      tmp.clear();
      for (int jn = first[k]; jn <= last; ++jn) {
        const int j0 = iperm0[S.perm[jn]];
        for (int p = aptr[j0]; p < aptr[j0 + 1]; ++p) {
          const int i = newidx[aidx[p]];
          if (i > last && mark[i] != k) { mark[i] = k; tmp.push_back(i); }
        }
      }
      for (int c : fkids[k]) for (int i : R[c]) if (i > last && mark[i] != k) { mark[i] = k; tmp.push_back(i); }
      std::sort(tmp.begin(), tmp.end());
      R[k] = tmp;
      if (!tmp.empty() && colf[tmp[0]] != fparent[k]) return false;   // tree consistency
    }
  }
  phase("row structures");
  // ---- 7. levels (leaves first); fronts renumbered level by level, fronts with children last inside a level ----
  std::vector<int> level(nf, 0);
  int n_levels = 1;
  for (int k = 0; k < nf; ++k) {
    for (int c : fkids[k]) level[k] = std::max(level[k], level[c] + 1);
    n_levels = std::max(n_levels, level[k] + 1);
  }
  S.n_levels = n_levels;
  std::vector<int> order(nf);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (level[a] != level[b]) return level[a] < level[b];
    const bool ca = !fkids[a].empty(), cb = !fkids[b].empty();
    if (ca != cb) return cb;
    return false;
  });
  std::vector<int> fid(nf);   // postorder id -> final id
  for (int k = 0; k < nf; ++k) fid[order[k]] = k;
  S.fronts.resize(nf);
  S.col_front.resize(N);
  long long fbase = 0, wbase = 0, blocks = 0;
  double flops = 0;
  for (int f = 0; f < nf; ++f) {
    const int k = order[f];
    FrontDesc& D = S.fronts[f];
    D.first = first[k];
    D.c = ccount[k];
    D.r = (int)R[k].size();
    const int n = 6 * (D.c + D.r);
    D.ld = n + 2;
    D.fbase = fbase;
    fbase += ((long long)(n + 1) * D.ld + 15) / 16 * 16;
    D.idx_begin = (int)S.idx.size();
    S.idx.insert(S.idx.end(), R[k].begin(), R[k].end());
    D.child_begin = (int)S.child.size();
    for (int c : fkids[k]) S.child.push_back(fid[c]);
    std::sort(S.child.begin() + D.child_begin, S.child.end());
    D.child_end = (int)S.child.size();
    D.parent = fparent[k] >= 0 ? fid[fparent[k]] : -1;
    const int npanels = (6 * D.c + FRONT_NB - 1) / FRONT_NB;
    D.wbase = (int)wbase;
    wbase += (long long)npanels * FRONT_NB * FRONT_NB;
    D.ntp = (D.c + D.r + FRONT_ASM_TP - 1) / FRONT_ASM_TP;
    D.asm_wg_begin = 0;
    D.rel_begin = 0;
    for (int j = D.first; j < D.first + D.c; ++j) S.col_front[j] = f;
    S.max_front = std::max(S.max_front, n);
    const double c6 = 6.0 * D.c, r6 = 6.0 * D.r;
    flops += c6 * c6 * c6 / 3.0 + c6 * c6 * r6 + c6 * r6 * r6;
    blocks += (long long)D.c * (D.c + 1) / 2 + (long long)D.c * D.r;
  }
  S.fval_size = fbase;
  S.winv_size = std::max(1LL, wbase);
  S.flops = flops;
  S.factor_blocks = blocks;
  if (wbase > 0x7fffffffLL) return false;
  if ((S.fval_size + S.winv_size) * 8 > max_bytes) {
    if (getenv("PGO_VERBOSE"))
      std::fprintf(stderr, "[pgo] front: %.2f GB of fronts exceed the budget of %.2f GB\n", 8e-9 * (double)S.fval_size, 1e-9 * (double)max_bytes);
    return false;
  }
  phase("levels + fronts");
  // ---- 8. child -> parent maps ----
  for (int f = 0; f < nf; ++f) {
    FrontDesc& D = S.fronts[f];
    D.rel_begin = (int)S.rel.size();
    if (D.parent < 0) continue;
    const FrontDesc& Pd = S.fronts[D.parent];
    const int* pr = S.idx.data() + Pd.idx_begin;
    int q = 0;
    for (int t = 0; t < D.r; ++t) {
      const int i = S.idx[D.idx_begin + t];
      if (i < Pd.first + Pd.c) {
        if (i < Pd.first) return false;
        S.rel.push_back(i - Pd.first);
      } else {
        while (q < Pd.r && pr[q] < i) ++q;
        if (q >= Pd.r || pr[q] != i) return false;
        S.rel.push_back(Pd.c + q);
      }
    }
  }
  // tile starts of every child inside its parent (extend-add: no searching on the device)
  for (int f = 0; f < nf; ++f) {
    FrontDesc& D = S.fronts[f];
    D.cs_begin = (int)S.cstart.size();
    D.pad = 0;
    if (D.parent < 0) continue;
    const int ntp = S.fronts[D.parent].ntp;
    const int* rel = S.rel.data() + D.rel_begin;
    int k = 0;
    for (int t = 0; t <= ntp; ++t) {
      while (k < D.r && rel[k] < FRONT_ASM_TP * t) ++k;
      S.cstart.push_back(k);
    }
  }
  if (S.cstart.empty()) S.cstart.push_back(0);
  if (S.rel.empty()) S.rel.push_back(0);
  if (S.idx.empty()) S.idx.push_back(0);
  if (S.child.empty()) S.child.push_back(0);
  phase("child maps");
  // ---- 9. BSR sources per front block ----
  {
    struct Ent { long long key; int slot; };
    std::vector<Ent> ents;
    ents.reserve(n_slots);
    for (int t = 0; t < n_slots; ++t) {
      const uint8_t side = slot_side[t];
      if (side == SIDE_PAD) continue;
      const int i = S.iperm[slot_row[t]];
      const int j = side == SIDE_DIAG ? i : S.iperm[slot_col[t]];
      if (i < j) continue;   // the twin slot carries the transposed block
      if (i == j && side != SIDE_DIAG) continue;
      const int f = S.col_front[j];
      const FrontDesc& D = S.fronts[f];
      const int bj = j - D.first;
      int bi;
      if (i < D.first + D.c) bi = i - D.first;
      else {
        const int* lo = S.idx.data() + D.idx_begin;
        const int* hi = lo + D.r;
        const int* it = std::lower_bound(lo, hi, i);
        if (it == hi || *it != i) return false;
        bi = D.c + (int)(it - lo);
      }
      if (bi >= 32768 || bj >= 65536) return false;
      ents.push_back(Ent{((long long)f << 32) | ((long long)bi << 16) | bj, t});
    }
    {   // order by (front, block row, block column, slot): bucket by front (the fronts' entries are few), then sort each bucket
      std::vector<int> fptr(nf + 1, 0);
      for (const Ent& en : ents) ++fptr[(int)(en.key >> 32) + 1];
      for (int f = 0; f < nf; ++f) fptr[f + 1] += fptr[f];
      std::vector<Ent> sorted(ents.size());
      std::vector<int> fill(fptr.begin(), fptr.end() - 1);
      for (const Ent& en : ents) sorted[fill[(int)(en.key >> 32)]++] = en;
      for (int f = 0; f < nf; ++f)
        std::sort(sorted.begin() + fptr[f], sorted.begin() + fptr[f + 1], [](const Ent& a, const Ent& b) { return a.key != b.key ? a.key < b.key : a.slot < b.slot; });
      ents.swap(sorted);
    }
    S.ablk_ptr.push_back(0);
    for (size_t e = 0; e < ents.size(); ++e) {
      if (e == 0 || ents[e].key != ents[e - 1].key) {
        if (e) S.ablk_ptr.push_back((int)e);
        S.ablk_front.push_back((int)(ents[e].key >> 32));
        S.ablk_pos.push_back((int)(ents[e].key & 0xffffffffLL));
      }
      S.ablk_slot.push_back(ents[e].slot);
    }
    S.ablk_ptr.push_back((int)ents.size());
  }
  S.levels.resize(n_levels);
  {
    int f = 0;
    for (int l = 0; l < n_levels; ++l) {
      S.levels[l].front_begin = f;
      while (f < nf && level[order[f]] == l) ++f;
      S.levels[l].front_end = f;
    }
  }
  phase("slot sources");
  // ---- 9b. small fronts: compact storage, one launch per level, no schedule ----
  // pure plan: every front is small.  Mixed plan (knob front_mixed = <scalars>, pgo_tuning.h; off by default): the fronts whose whole subtree is
  // small (chains and leaves at the bottom of a mesh: KITTI-00 dense 447 of 562 fronts, Manhattan 10 k 730 of 1243) take the
  // small-front kernels level by level before the round schedule of the rest starts; the roots of those subtrees hand their
  // update matrices over in the regular layout.  Measured: no gain on those two graphs (Manhattan 10 k 3.48 vs 3.41 ms per
  // factorisation, KITTI-00 dense 12.1 vs 11.7 ms per three iterations) — the rounds already run the small fronts beside
  // the long panel chains of the big ones, which are the critical path; taking them out in front only adds their launches.
  const bool pure_small = small_max > 0 && S.max_front <= small_max;
  const int mixed_max = pure_small ? small_max : std::min((int)tuning("front_mixed", 0), (int)SFRONT_MAX);
  std::vector<char> is_small(nf, 0);
  {
    int cnt = 0;
    if (mixed_max > 0)
      for (int f = 0; f < nf; ++f) {     // children have smaller numbers
        const FrontDesc& D = S.fronts[f];
        bool ok = 6 * (D.c + D.r) <= mixed_max;
        for (int ci = D.child_begin; ci < D.child_end && ok; ++ci) ok = is_small[S.child[ci]] != 0;
        is_small[f] = ok ? 1 : 0;
        cnt += ok ? 1 : 0;
      }
    S.n_small = cnt;
    S.mixed = !pure_small && cnt >= 32;
    if (!pure_small && !S.mixed) std::fill(is_small.begin(), is_small.end(), 0);
  }
  if (pure_small || S.mixed) {
    S.small = pure_small;
    S.sfronts.assign(nf, SFront{0, 0, 0, 0, 0, 0, 0, 0});
    long long lb = 0, ub = 0, wb = 0;
    S.urel.clear();
    for (int f = 0; f < nf; ++f) {
      if (!is_small[f]) continue;
      FrontDesc& D = S.fronts[f];
      D.pad = S.mixed ? 1 : 0;
      const int c6 = 6 * D.c, r6 = 6 * D.r;
      const bool packed = D.parent >= 0 && is_small[D.parent];
      const long long ucnt = packed ? (long long)r6 * (r6 + 1) / 2 + r6 : 0;
      S.sfronts[f] = SFront{(int)lb, (int)ub, (int)ucnt, (int)S.urel.size(), (D.parent >= 0 && !packed) ? 1 : 0, 0, 0, (int)wb};
      lb += ((long long)(c6 + r6 + 1) * c6 + 1) / 2 * 2;
      ub += (ucnt + 1) / 2 * 2;
      wb += (long long)c6 * c6;
      if (!packed) continue;
      const int* rel = S.rel.data() + D.rel_begin;
      for (int t = 0; t < D.r; ++t) for (int a = 0; a < 6; ++a) S.urel.push_back(6 * rel[t] + a);   // parent column of column 6 t + a
    }
    if (S.urel.empty()) S.urel.push_back(0);
    if (lb > 0x7fffffffLL || ub > 0x7fffffffLL || wb > 0x7fffffffLL) return false;
    S.sl_size = std::max(2LL, lb);
    S.su_size = std::max(2LL, ub);
    S.sw_size = std::max(2LL, wb);
    S.osrc.assign(S.ablk_front.size() + 1, -1);
    for (size_t a = 0; a < S.ablk_front.size(); ++a)
      if (S.ablk_ptr[a + 1] - S.ablk_ptr[a] == 1) {
        const int slot = S.ablk_slot[S.ablk_ptr[a]];
        if (slot < (1 << 28)) S.osrc[a] = slot | ((int)slot_side[slot] << 28);
      }
    const int na = (int)S.ablk_front.size();
    for (int a = 0; a < na; ++a) {       // ablk entries are sorted by front
      SFront& F = S.sfronts[S.ablk_front[a]];
      if (F.ablk_end == F.ablk_begin) F.ablk_begin = a;
      F.ablk_end = a + 1;
    }
    // the small fronts of every level (fronts are numbered level by level)
    S.slevel_ptr.assign(1, 0);
    S.slevel_front.clear();
    for (int l = 0; l < n_levels; ++l) {
      for (int f = S.levels[l].front_begin; f < S.levels[l].front_end; ++f) if (is_small[f]) S.slevel_front.push_back(f);
      S.slevel_ptr.push_back((int)S.slevel_front.size());
    }
    if (S.slevel_front.empty()) S.slevel_front.push_back(0);
    phase("small-front plan");
  }
  if (pure_small) {
    S.n_launches = 2 * n_levels + 2;
    S.est_us = 8.0 * S.n_launches;
    if (getenv("PGO_VERBOSE")) {
      std::fprintf(stderr, "[pgo] front: n=%d supernodes %d -> %d small fronts (largest %d scalars), %d levels = %d launches, %.3g flops\n", N, ns0, nf,
                   S.max_front, n_levels, S.n_launches, flops);
      for (int l = 0; l < n_levels; ++l) {     // what bounds a level's launch: its largest front, its longest pivot chain, its widest fan-in
        int mc = 0, mn = 0, mk = 0;
        for (int f = S.levels[l].front_begin; f < S.levels[l].front_end; ++f) {
          const FrontDesc& D = S.fronts[f];
          mc = std::max(mc, D.c); mn = std::max(mn, D.c + D.r); mk = std::max(mk, D.child_end - D.child_begin);
        }
        std::fprintf(stderr, "[pgo] front:   level %2d: %4d fronts, at most %2d pivot poses, %2d poses per front, %2d children\n", l,
                     S.levels[l].front_end - S.levels[l].front_begin, mc, mn, mk);
      }
    }
    return true;
  }
  // ---- 10. schedule: rounds over the whole tree, not level by level ----
  // A front runs the chain  [extend-add] -> panel 0 -> [GEMM] -> panel 1 -> ... ; it may start as soon as ITS children are
  // done.  Every round issues up to three launches — extend-add, panel, GEMM — each carrying the next phase of every front
  // that is ready for it, whatever its tree level.  The number of rounds is the longest chain of panel steps from a leaf to
  // the root (Manhattan 10 k: 69 instead of the 133 a level-by-level schedule needs; sphere x10: 248 instead of 508), and
  // the chain, not the flops, is what bounds these factorisations.  The backward substitution is scheduled the same way from
  // the root down.
  const long long tile32_below = (long long)tuning("front_tile32_below", 192);
  // width of the outer panels of the factorisation (left-looking inside, one right-looking GEMM behind each): a multiple of 48
  const int nbo = FRONT_NBO;     // (288 / 384 / 576 measured slower in r02 behind PGO_FRONT_NBO: the longer left-looking sums cost more than the longer-K updates gain)
  {
    std::vector<int> next_step(nf, 0), nsteps(nf), kids_left(nf), pending(nf, -1);
    std::vector<char> asm_done(nf, 0), finished(nf, 0);
    for (int q = 0; q < nf; ++q) {
      nsteps[q] = (6 * S.fronts[q].c + FRONT_NB - 1) / FRONT_NB;
      kids_left[q] = S.fronts[q].child_end - S.fronts[q].child_begin;
      asm_done[q] = kids_left[q] == 0;           // a leaf has nothing to assemble
    }
    int n_finished = 0;
    for (int q = 0; q < nf; ++q)                 // mixed plan: the small fronts are done when the rounds start
      if (is_small[q]) {
        finished[q] = 1; asm_done[q] = 1; next_step[q] = nsteps[q];
        ++n_finished;
        if (S.fronts[q].parent >= 0) --kids_left[S.fronts[q].parent];
      }
    std::vector<int> done_now, gjobs;
    struct Rec { long long key; int child, kk, mm; };
    std::vector<Rec> recs;
    std::vector<int> occ;
    while (n_finished < nf) {
      done_now.clear();
      // (1) extend-add of every front whose children are all done: one workgroup per parent tile (8 x 8 poses, or the
      // right-hand-side row x 8 poses) that receives anything, with the list of contributing children (list order)
      {
        FrontLaunch la{FrontLaunch::ASM, 0, 0, (int)(S.asm_tile.size() / 8)};
        for (int q = 0; q < nf; ++q) {
          if (asm_done[q] || kids_left[q] > 0) continue;
          asm_done[q] = 1;
          const FrontDesc& P = S.fronts[q];
          recs.clear();
          for (int ci = P.child_begin; ci < P.child_end; ++ci) {
            const FrontDesc& C = S.fronts[S.child[ci]];
            const int* cs = S.cstart.data() + C.cs_begin;
            occ.clear();
            for (int t = 0; t < P.ntp; ++t) if (cs[t + 1] > cs[t]) occ.push_back(t);
            for (size_t a = 0; a <= occ.size(); ++a) {            // a == occ.size(): the right-hand-side row
              const bool rhs = a == occ.size();
              const int ti = rhs ? P.ntp : occ[a];
              const int kk = rhs ? (C.r << 16) | (C.r + 1) : (cs[ti] << 16) | cs[ti + 1];
              for (size_t b2 = 0; b2 < occ.size() && (rhs || b2 <= a); ++b2) {
                const int tj = occ[b2];
                recs.push_back(Rec{((long long)ti << 40) | ((long long)tj << 20) | (long long)(ci - P.child_begin), S.child[ci], kk, (cs[tj] << 16) | cs[tj + 1]});
              }
            }
          }
          std::sort(recs.begin(), recs.end(), [](const Rec& x, const Rec& y) { return x.key < y.key; });
          for (size_t e = 0; e < recs.size();) {
            size_t f2 = e;
            while (f2 < recs.size() && (recs[f2].key >> 20) == (recs[e].key >> 20)) ++f2;
            // records carry everything the kernel needs of the parent / child fronts (one dependent load each, not three)
            S.asm_tile.push_back(q);
            S.asm_tile.push_back((int)(((recs[e].key >> 40) << 16) | ((recs[e].key >> 20) & 0xfffff)));
            S.asm_tile.push_back((int)(S.asm_contrib.size() / 8));
            for (size_t u = e; u < f2; ++u) {
              const FrontDesc& C = S.fronts[recs[u].child];
              S.asm_contrib.push_back((int)(C.fbase & 0xffffffffLL));
              S.asm_contrib.push_back((int)(C.fbase >> 32));
              S.asm_contrib.push_back(C.ld);
              S.asm_contrib.push_back(C.c);
              S.asm_contrib.push_back(recs[u].kk);
              S.asm_contrib.push_back(recs[u].mm);
              S.asm_contrib.push_back(C.rel_begin);
              S.asm_contrib.push_back(C.r);
            }
            S.asm_tile.push_back((int)(S.asm_contrib.size() / 8));
            S.asm_tile.push_back((int)(P.fbase & 0xffffffffLL));
            S.asm_tile.push_back((int)(P.fbase >> 32));
            S.asm_tile.push_back(P.ld);
            S.asm_tile.push_back(6 * (P.c + P.r) | (P.ntp << 20));
            e = f2;
          }
        }
        la.n_wg = (int)(S.asm_tile.size() / 8) - la.wg_begin;
        if (la.n_wg > 0) S.launches.push_back(la);
      }
      // (2) the next 48-column panel of every front that is assembled and not waiting for a GEMM
      {
        FrontLaunch lp{FrontLaunch::PANEL, 0, FRONT_TILE, (int)S.wg_job.size()};
        for (int q = 0; q < nf; ++q) {
          if (finished[q] || !asm_done[q] || kids_left[q] > 0 || pending[q] >= 0 || next_step[q] >= nsteps[q]) continue;
          const FrontDesc& D = S.fronts[q];
          const int step = next_step[q]++;
          const int c6 = 6 * D.c, n = 6 * (D.c + D.r), k0 = step * FRONT_NB;
          const int nb = std::min<int>(FRONT_NB, c6 - k0), kend = k0 + nb;
          const int ostart = (k0 / nbo) * nbo, oend = std::min(c6, ostart + nbo);
          const int job = (int)S.jobs.size();
          S.jobs.push_back(FrontJob{D.fbase, D.ld, kend, n + 1, ostart, 0, k0, nb, D.wbase + step * FRONT_NB * FRONT_NB});
          S.job_front.push_back(q);
          const int ntr = (n + 1 - kend + FRONT_TILE - 1) / FRONT_TILE;
          for (int t = 0; t < ntr; ++t) { S.wg_job.push_back(job); S.wg_tile.push_back(t << 16); }
          // the GEMM this panel is followed by: right-looking update behind a finished outer panel, or the Schur update
          if (kend >= oend) {
            int r0, r1, cc0, cc1, kk0, klen;
            if (oend < c6) { r0 = oend; r1 = n + 1; cc0 = oend; cc1 = c6; kk0 = ostart; klen = oend - ostart; }
            else { r0 = c6; r1 = n + 1; cc0 = c6; cc1 = n; kk0 = 0; klen = c6; }
            if (cc1 > cc0) {
              pending[q] = (int)S.jobs.size();
              S.jobs.push_back(FrontJob{D.fbase, D.ld, r0, r1, cc0, cc1, kk0, klen, 0});
              S.job_front.push_back(q);
            }
          }
          if (next_step[q] >= nsteps[q] && pending[q] < 0) { finished[q] = 1; done_now.push_back(q); }
        }
        lp.n_wg = (int)S.wg_job.size() - lp.wg_begin;
        if (lp.n_wg > 0) S.launches.push_back(lp);
      }
      // (3) the pending GEMMs; launches with few 64 x 64 tiles are cut into 32 x 32 tiles instead
      {
        gjobs.clear();
        long long tiles64 = 0;
        for (int q = 0; q < nf; ++q) {
          if (pending[q] < 0) continue;
          const FrontJob& J = S.jobs[pending[q]];
          gjobs.push_back(pending[q]);
          tiles64 += (long long)((J.r1 - J.r0 + 63) / 64) * ((J.c1 - J.c0 + 63) / 64);
          pending[q] = -1;
          if (next_step[q] >= nsteps[q]) { finished[q] = 1; done_now.push_back(q); }
        }
        const int T = tiles64 <= tile32_below ? 32 : 64;
        FrontLaunch lg{FrontLaunch::GEMM, 0, T, (int)S.wg_job.size()};
        for (int job : gjobs) {
          const FrontJob& J = S.jobs[job];
          const int ntr = (J.r1 - J.r0 + T - 1) / T, ntc = (J.c1 - J.c0 + T - 1) / T;
          for (int ti = 0; ti < ntr; ++ti)
            for (int tj = 0; tj < ntc; ++tj) {
              const int row_last = std::min(J.r0 + T * (ti + 1), J.r1) - 1;
              if (row_last < J.c0 + T * tj) continue;   // entirely above the diagonal
              S.wg_job.push_back(job);
              S.wg_tile.push_back((ti << 16) | tj);
            }
        }
        lg.n_wg = (int)S.wg_job.size() - lg.wg_begin;
        if (lg.n_wg > 0) S.launches.push_back(lg);
      }
      if (done_now.empty() && S.launches.size() > 100000000u) return false;   // (cannot happen: every round advances some front)
      for (int q : done_now) {
        ++n_finished;
        if (S.fronts[q].parent >= 0) --kids_left[S.fronts[q].parent];
      }
    }
  }
  // ---- 10b. the same work as stages for the single-launch form (FrontStages): a stage = one job or one front's extend-add ----
  S.st_table.clear(); S.st_pred_ptr.assign(1, 0); S.st_pred.clear(); S.st_need.clear();
  if (!S.mixed) {
    std::vector<int> last_stage(nf, -1), stage_of_job(S.jobs.size(), -1), asm_stage(nf, -1), touched;
    for (const FrontLaunch& La : S.launches) {
      touched.clear();
      for (int w = La.wg_begin; w < La.wg_begin + La.n_wg; ++w) {
        int stage, q;
        if (La.type == FrontLaunch::ASM) {
          q = S.asm_tile[8 * (size_t)w];
          if (asm_stage[q] < 0) {
            asm_stage[q] = (int)S.st_need.size();
            S.st_need.push_back(0);
            for (int ci = S.fronts[q].child_begin; ci < S.fronts[q].child_end; ++ci) S.st_pred.push_back(last_stage[S.child[ci]]);
            S.st_pred_ptr.push_back((int)S.st_pred.size());
            touched.push_back(q);
          }
          stage = asm_stage[q];
        } else {
          const int job = S.wg_job[w];
          q = S.job_front[job];
          if (stage_of_job[job] < 0) {
            stage_of_job[job] = (int)S.st_need.size();
            S.st_need.push_back(0);
            if (last_stage[q] >= 0) S.st_pred.push_back(last_stage[q]);
            S.st_pred_ptr.push_back((int)S.st_pred.size());
            touched.push_back(q);
          }
          stage = stage_of_job[job];
        }
        ++S.st_need[stage];
        const int kind = La.type == FrontLaunch::ASM ? 0 : La.type == FrontLaunch::PANEL ? 1 : La.tile == 64 ? 2 : 3;
        S.st_table.push_back(kind | (w << 2));
        S.st_table.push_back(stage);
      }
      // a front has one stage per launch: its chain moves on once the launch has been walked
      for (int q : touched) {
        if (La.type == FrontLaunch::ASM) last_stage[q] = asm_stage[q];
      }
      if (La.type != FrontLaunch::ASM)
        for (int w = La.wg_begin; w < La.wg_begin + La.n_wg; ++w) last_stage[S.job_front[S.wg_job[w]]] = stage_of_job[S.wg_job[w]];
    }
    for (int v : S.st_pred) if (v < 0) { S.st_table.clear(); break; }     // (cannot happen: a child is finished before its parent's extend-add is scheduled)
  }
  // backward substitution, root first: a front may start once its parent is done.  Per round: phase A (t = y - L21^T x_r, 64
  // columns per workgroup) of the fronts that became ready, then one 192-column block step of every front under way (the
  // block nblk - s updates the chunks below it, the workgroup of chunk nblk - 1 - s then solves that chunk's diagonal block;
  // step 0 only solves the last chunk).
  {
    std::vector<int> st(nf, -1), nblk(nf);       // st: -1 not started, else next block step
    std::vector<char> fin(nf, 0), parent_done(nf, 0);
    for (int q = 0; q < nf; ++q) { nblk[q] = (6 * S.fronts[q].c + FRONT_NBO - 1) / FRONT_NBO; parent_done[q] = S.fronts[q].parent < 0; }
    int n_fin = 0;
    for (int q = 0; q < nf; ++q) if (is_small[q]) { fin[q] = 1; st[q] = 0; ++n_fin; }     // (their backward levels follow the rounds)
    std::vector<int> done_now;
    while (n_fin < nf) {
      done_now.clear();
      FrontBwdLaunch la{0, (int)S.bwd_front.size(), 0, 0};
      size_t lds_a = 0;
      for (int q = 0; q < nf; ++q) {
        if (st[q] >= 0 || !parent_done[q]) continue;
        st[q] = 0;
        for (int ch = 0; ch < (6 * S.fronts[q].c + 63) / 64; ++ch) { S.bwd_front.push_back(q); S.bwd_chunk.push_back(ch); }
        lds_a = std::max(lds_a, (size_t)(6 * S.fronts[q].r + 512) * sizeof(double));
      }
      la.n_wg = (int)S.bwd_front.size() - la.wg_begin;
      la.lds_bytes = (int)lds_a;
      if (la.n_wg > 0) S.bwd_launches.push_back(la);
      FrontBwdLaunch lb{1, (int)S.bwdb_front.size(), 0, 0};
      for (int q = 0; q < nf; ++q) {
        if (st[q] < 0 || fin[q]) continue;
        const int s2 = st[q]++;
        const int src = nblk[q] - s2, solver = nblk[q] - 1 - s2;
        for (int ch = 0; ch <= solver; ++ch) {
          if (s2 == 0 && ch != solver) continue;
          S.bwdb_front.push_back(q);
          S.bwdb_chunk.push_back(((s2 == 0 ? solver : src) << 16) | ch);
        }
        if (st[q] >= nblk[q]) { fin[q] = 1; done_now.push_back(q); }
      }
      lb.n_wg = (int)S.bwdb_front.size() - lb.wg_begin;
      if (lb.n_wg > 0) S.bwd_launches.push_back(lb);
      for (int q : done_now) {
        ++n_fin;
        for (int ci = S.fronts[q].child_begin; ci < S.fronts[q].child_end; ++ci) parent_done[S.child[ci]] = 1;
      }
      if (done_now.empty() && la.n_wg == 0 && lb.n_wg == 0) return false;
    }
  }
  if (S.wg_job.empty()) { S.wg_job.push_back(0); S.wg_tile.push_back(0); }
  if (S.asm_tile.empty()) S.asm_tile.assign(8, 0);
  if (S.asm_contrib.empty()) S.asm_contrib.assign(8, 0);
  if (S.bwd_front.empty()) { S.bwd_front.push_back(0); S.bwd_chunk.push_back(0); }
  if (S.bwdb_front.empty()) { S.bwdb_front.push_back(0); S.bwdb_chunk.push_back(0); }
  S.n_launches = (int)S.launches.size() + (int)S.bwd_launches.size() + 2;
  S.est_us = 4.0 * S.n_launches + flops / 2.0e7;   // ~4 us per dependent launch, ~20 TFLOP/s sustained
  if (getenv("PGO_VERBOSE"))
    std::fprintf(stderr, "[pgo] front: n=%d supernodes %d -> %d fronts, %d levels, %d launches, largest front %d, %.3g flops, %.1f MB, est %.0f us\n",
                 N, ns0, nf, n_levels, S.n_launches, S.max_front, flops, 8e-6 * (double)(S.fval_size + S.winv_size), S.est_us);
  return true;
}

}  // namespace pgo
