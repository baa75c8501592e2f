// pgo_direct.cpp — host symbolic phase of the GPU block-sparse Cholesky (see pgo_direct.h).
//
//  1. nested-dissection ordering of the pose graph: recursive bisection with breadth-first level structures
//     (pseudo-peripheral start, middle level = vertex separator, trimmed to the vertices that touch the far side).
//     Pose graphs of the reference's kind are a trajectory chain plus loop-closure chords, so BFS
//     levels cut across the "street" and separators stay a handful of poses; the elimination tree gets
//     logarithmic depth instead of the O(N) chain a minimum-degree ordering produces.
//  2. symbolic factorisation (column structures through the elimination tree), 6x6 block granularity.
//  3. for every block of L: the BSR slots that initialise it and the list of (L_ik, L_jk) update pairs.
//  4. elimination-tree levels = the launch schedule; row lists for the forward solve.
#include "pgo_direct.h"
#include "pgo_pool.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <memory>
#include <numeric>
#include <thread>

namespace pgo {

namespace {

struct Csr {
  std::vector<int> ptr, idx;
  int deg(int v) const { return ptr[v + 1] - ptr[v]; }
};

// fn(lo, hi) over contiguous pieces of [0, n) on the host worker pool (one piece per `grain` items at least)
template <class F>
void parallel_ranges(int n, int grain, F&& fn) {
  const int nt = n < 2 * grain ? 1 : std::min(HostPool::get().width(), n / grain);
  if (nt <= 1) { fn(0, n); return; }
  HostPool::get().run(nt, [&](int i) { fn((int)((long long)n * i / nt), (int)((long long)n * (i + 1) / nt)); });
}

Csr build_adjacency(int N, const std::vector<int>& ia, const std::vector<int>& ib) {
  // neighbour lists, each sorted and without duplicates (bucket by vertex, then sort the short lists)
  Csr g;
  std::vector<int> ptr(N + 1, 0);
  for (size_t k = 0; k < ia.size(); ++k) {
    if (ia[k] == ib[k]) continue;
    ++ptr[ia[k] + 1];
    ++ptr[ib[k] + 1];
  }
  for (int v = 0; v < N; ++v) ptr[v + 1] += ptr[v];
  std::vector<int> idx(ptr[N]), fill(ptr.begin(), ptr.end() - 1);
  for (size_t k = 0; k < ia.size(); ++k) {
    if (ia[k] == ib[k]) continue;
    idx[fill[ia[k]]++] = ib[k];
    idx[fill[ib[k]]++] = ia[k];
  }
  // sort + unique in place (fill[v] <- the number of distinct neighbours), prefix sum, compaction: vertex ranges side by side
  parallel_ranges(N, 8192, [&](int lo, int hi) {
    for (int v = lo; v < hi; ++v) {
      int* b = idx.data() + ptr[v];
      int* e = idx.data() + ptr[v + 1];
      std::sort(b, e);
      fill[v] = (int)(std::unique(b, e) - b);
    }
  });
  g.ptr.assign(N + 1, 0);
  for (int v = 0; v < N; ++v) g.ptr[v + 1] = g.ptr[v] + fill[v];
  g.idx.resize(g.ptr[N]);
  parallel_ranges(N, 8192, [&](int lo, int hi) {
    for (int v = lo; v < hi; ++v) std::copy(idx.data() + ptr[v], idx.data() + ptr[v] + fill[v], g.idx.data() + g.ptr[v]);
  });
  return g;
}

// Nested dissection.  `label` marks the subset being ordered (label[v] == id).
// The subsets are ranges [begin, end) of ONE work array, partitioned in place as [A | B | separator]; the final content of the
// array IS the elimination order (A's order, then B's, the separator last), so disjoint ranges are independent tasks: large
// subsets are split by whichever host thread picks them up, the two halves go back to a shared task list, and subsets below
// kSequential vertices are finished by one thread.  The order does not depend on the number of threads or on timing: a
// subset's result is a function of its vertex sequence alone (labels are unique ids, nothing else is shared).
struct DissectShared {
  const Csr& g;
  std::vector<int> label, dist;      // per vertex; a vertex belongs to one live subset at a time
  std::vector<int>& work;
  std::atomic<int> next_worker{0};   // label ids are (worker << 22) | counter: unique without a shared counter in the hot loop
  DissectShared(const Csr& graph, std::vector<int>& w) : g(graph), label(graph.ptr.size() - 1, 0), dist(graph.ptr.size() - 1, -1), work(w) {}
  int label_of(int v) const { return __atomic_load_n(&label[v], __ATOMIC_RELAXED); }     // neighbours may belong to a subset another thread is relabelling
  void set_label(int v, int id) { __atomic_store_n(&label[v], id, __ATOMIC_RELAXED); }
};

struct Dissector {
  DissectShared& sh;
  std::vector<int> queue, tmp, lvl_cnt, sep;
  int label_base, label_next = 1;
  explicit Dissector(DissectShared& shared) : sh(shared), label_base(shared.next_worker.fetch_add(1) << 22) {}

  // BFS inside the subset `id` from `start`; fills queue (visit order) and dist; returns the last vertex visited
  int bfs(int start, int id) {
    const Csr& g = sh.g;
    queue.clear();
    queue.push_back(start);
    sh.dist[start] = 0;
    for (size_t h = 0; h < queue.size(); ++h) {
      const int v = queue[h];
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; ++p) {
        const int u = g.idx[p];
        if (sh.label_of(u) == id && sh.dist[u] < 0) { sh.dist[u] = sh.dist[v] + 1; queue.push_back(u); }
      }
    }
    return queue.back();
  }
  void clear_dist() { for (int v : queue) sh.dist[v] = -1; }

  // One step on the subset [sb, se): either it is final (returns false), or it has been rearranged into two sub-ranges
  // [a0, a1) and [b0, b1) that remain to be ordered (returns true).
  bool split(int sb, int se, int* a0, int* a1, int* b0, int* b1) {
    const Csr& g = sh.g;
    const int total = se - sb;
    if (total <= 2) return false;                      // kept in subset order
    int* S = sh.work.data() + sb;
    if ((int)tmp.size() < total) tmp.resize(total);
    if (label_next >= (1 << 22)) label_next = 1;      // (a worker reuses an id only after 4 M subsets: none of the earlier ones is live)
    const int id = label_base | label_next++;
    for (int i = 0; i < total; ++i) sh.set_label(S[i], id);
    // one connected component at a time
    bfs(S[0], id);
    if ((int)queue.size() < total) {
      // [rest (subset order) | component (BFS order)]
      int nr = 0;
      for (int i = 0; i < total; ++i) if (sh.dist[S[i]] < 0) tmp[nr++] = S[i];
      const int ncomp = (int)queue.size();
      for (int i = 0; i < ncomp; ++i) tmp[nr + i] = queue[i];
      clear_dist();
      for (int i = 0; i < total; ++i) sh.set_label(S[i], 0);
      std::copy(tmp.begin(), tmp.begin() + total, S);
      *a0 = sb; *a1 = sb + nr; *b0 = sb + nr; *b1 = se;
      return true;
    }
    // pseudo-peripheral vertex: two more sweeps
    int far = queue.back();
    clear_dist();
    far = bfs(far, id);
    clear_dist();
    bfs(far, id);
    const int depth = sh.dist[queue.back()];
    if (depth < 2) {
      // (nearly) a clique: no useful separator, eliminate in subset order
      clear_dist();
      for (int i = 0; i < total; ++i) sh.set_label(S[i], 0);
      return false;
    }
    // level sizes; separator = smallest level among those whose prefix holds 35..65 % of the vertices
    lvl_cnt.assign(depth + 1, 0);
    for (int v : queue) ++lvl_cnt[sh.dist[v]];
    int best = -1, acc = 0;
    for (int l = 0; l <= depth; ++l) {
      const int before = acc;
      acc += lvl_cnt[l];
      if (l == 0 || l == depth) continue;
      if (before >= total * 0.35 && before <= total * 0.65) {
        if (best < 0 || lvl_cnt[l] < lvl_cnt[best]) best = l;
      }
    }
    if (best < 0) {  // no level in the window: take the one closest to the middle
      acc = 0;
      int bd = total;
      for (int l = 0; l <= depth; ++l) {
        if (l > 0 && l < depth) { const int d = std::abs(2 * acc - total); if (d < bd) { bd = d; best = l; } }
        acc += lvl_cnt[l];
      }
    }
    // A from the front of tmp, B from its back (reversed below), the separator behind both; all in BFS order
    int na = 0, nb = 0;
    sep.clear();
    for (int v : queue) {
      if (sh.dist[v] < best) tmp[na++] = v;
      else if (sh.dist[v] > best) tmp[total - 1 - nb++] = v;
      else {
        bool touches = false;
        for (int p = g.ptr[v]; p < g.ptr[v + 1] && !touches; ++p) {
          const int u = g.idx[p];
          if (sh.label_of(u) == id && sh.dist[u] == best + 1) touches = true;
        }
        if (touches) sep.push_back(v); else tmp[na++] = v;
      }
    }
    clear_dist();
    for (int i = 0; i < total; ++i) sh.set_label(S[i], 0);
    std::copy(tmp.begin(), tmp.begin() + na, S);
    for (int i = 0; i < nb; ++i) S[na + i] = tmp[total - 1 - i];
    std::copy(sep.begin(), sep.end(), S + na + nb);
    *a0 = sb; *a1 = sb + na; *b0 = sb + na; *b1 = sb + na + nb;
    return true;
  }

  // the whole subtree below [sb, se) on this thread
  void finish(int sb, int se) {
    std::vector<std::pair<int, int>> stack;
    stack.emplace_back(sb, se);
    while (!stack.empty()) {
      const std::pair<int, int> r = stack.back();
      stack.pop_back();
      int a0, a1, b0, b1;
      if (!split(r.first, r.second, &a0, &a1, &b0, &b1)) continue;
      if (a1 > a0) stack.emplace_back(a0, a1);
      if (b1 > b0) stack.emplace_back(b0, b1);
    }
  }
};

// Orders every range of `roots` (disjoint ranges of sh.work).  Ranges above kSequential vertices are split by the thread that
// takes them and their halves handed back; up to `max_threads` host threads, none when everything is small.
void dissect_ranges(DissectShared& sh, std::vector<std::pair<int, int>> roots, int max_threads) {
  static const int kSequential = 600;
  long long total = 0;
  for (const auto& r : roots) total += r.second - r.first;
  // one pool worker per ~500 vertices, subsets above 600 vertices split as tasks (PGO_ND_PER_THREAD / PGO_ND_SEQ).  With a fresh
  // std::thread per worker these were 2000 / 3000 (a start cost tens of microseconds); handing a task to a pool worker costs a
  // few: KITTI-00's 4541 poses are ordered in 0.43 instead of 0.68 ms, the setup of a solve 1.72 -> 1.45 ms.  The order is the same
  // whatever the numbers (a subset's result is a function of its vertex sequence alone).
  static const int per_thread = 500;
  const int nt = (int)std::max<long long>(1, std::min<long long>(std::min(max_threads, (int)std::thread::hardware_concurrency()), total / per_thread));
  if (nt <= 1) {
    Dissector d(sh);
    for (const auto& r : roots) d.finish(r.first, r.second);
    return;
  }
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::pair<int, int>> tasks = std::move(roots);
  int in_flight = 0;                                  // tasks taken and not yet retired
  auto worker = [&]() {
    Dissector d(sh);
    for (;;) {
      std::pair<int, int> r;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !tasks.empty() || in_flight == 0; });
        if (tasks.empty()) return;                    // nothing queued and nobody working: done
        // largest first: the big splits are the critical path
        size_t bi = 0;
        for (size_t i = 1; i < tasks.size(); ++i) if (tasks[i].second - tasks[i].first > tasks[bi].second - tasks[bi].first) bi = i;
        r = tasks[bi];
        tasks[bi] = tasks.back();
        tasks.pop_back();
        ++in_flight;
      }
      if (r.second - r.first <= kSequential) {
        d.finish(r.first, r.second);
        std::lock_guard<std::mutex> lk(mu);
        --in_flight;
      } else {
        int a0, a1, b0, b1;
        const bool more = d.split(r.first, r.second, &a0, &a1, &b0, &b1);
        std::lock_guard<std::mutex> lk(mu);
        if (more) {
          if (a1 > a0) tasks.emplace_back(a0, a1);
          if (b1 > b0) tasks.emplace_back(b0, b1);
        }
        --in_flight;
      }
      cv.notify_all();
    }
  };
  HostPool::get().run(nt, [&](int) { worker(); });    // (a worker only waits while another one is at work: slots need not overlap)
}

}  // namespace

// Connected components (vertices of a component in ascending order, components by their smallest vertex).
struct Components {
  std::vector<int> ptr, verts, of;
  int count() const { return (int)ptr.size() - 1; }
};
static Components find_components(const Csr& g) {
  const int N = (int)g.ptr.size() - 1;
  Components C;
  C.of.assign(N, -1);
  std::vector<int> queue;
  int nc = 0;
  for (int v0 = 0; v0 < N; ++v0) {
    if (C.of[v0] >= 0) continue;
    queue.assign(1, v0);
    C.of[v0] = nc;
    for (size_t h = 0; h < queue.size(); ++h) {
      const int v = queue[h];
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; ++p) {
        const int u = g.idx[p];
        if (C.of[u] < 0) { C.of[u] = nc; queue.push_back(u); }
      }
    }
    ++nc;
  }
  C.ptr.assign(nc + 1, 0);
  for (int v = 0; v < N; ++v) ++C.ptr[C.of[v] + 1];
  for (int c = 0; c < nc; ++c) C.ptr[c + 1] += C.ptr[c];
  C.verts.resize(N);
  std::vector<int> fill(C.ptr.begin(), C.ptr.end() - 1);
  for (int v = 0; v < N; ++v) C.verts[fill[C.of[v]]++] = v;
  return C;
}

static int component_workers(const Components& C) {
  static const int cap = 16;
  return std::max(1, std::min(std::min(C.count(), cap), (int)std::thread::hardware_concurrency()));
}

// body(component, worker index) for every component, on up to 16 host threads when there are several components (the
// graphs of a batched solve, pgo_solve_batch); one component = the calling thread, no thread is started.
template <class Body>
static void for_components(const Components& C, Body&& body) {
  const int nc = C.count();
  const int nt = component_workers(C);
  if (nt <= 1) { for (int c = 0; c < nc; ++c) body(c, 0); return; }
  std::atomic<int> next(0);
  HostPool::get().run(nt, [&](int t) {
    for (;;) {
      const int c = next.fetch_add(1);
      if (c >= nc) return;
      body(c, t);
    }
  });
}

// Nested dissection of every component on its own (a component gets the order it gets as a graph of its own), components
// one after the other in the permutation.
static bool order_components(const Csr& g, const Components& C, std::vector<int>* perm) {
  *perm = C.verts;                                    // component c occupies [C.ptr[c], C.ptr[c + 1]), vertices ascending
  DissectShared sh(g, *perm);
  std::vector<std::pair<int, int>> roots;
  roots.reserve(C.count());
  for (int c = 0; c < C.count(); ++c) roots.emplace_back(C.ptr[c], C.ptr[c + 1]);
  static const int cap = 16;
  dissect_ranges(sh, std::move(roots), cap);
  return true;
}

// Pose / landmark problems (g2o's BlockSolver, Thirdparty/g2o/g2o/core/block_solver.hpp:47-87, splits the
// Hessian into Hpp, Hll, Hpl and eliminates the landmark blocks by the Schur complement Hpp - Hpl Hll^-1 Hpl'): in a sparse Cholesky
// that IS an elimination order — every 3-D point first (points are only linked to poses, so they are leaves of the elimination tree
// and eliminating them creates no fill among themselves), then the poses, ordered by nested dissection of the REDUCED graph in which
// two poses that observe a common point are neighbours (the pattern of the Schur complement).  Component by component, a component's
// columns contiguous as the analyses expect.  A point seen by more than 12 poses links its observers as a chain instead of a clique
// (the links only steer the dissection; the symbolic factorisation works on the true graph).
static bool order_components_points_first(const Csr& g, const Components& C, const std::vector<uint8_t>& is_point, std::vector<int>* perm) {
  const int N = (int)g.ptr.size() - 1;
  std::vector<int> pose_id(N, -1), poses;
  for (int v = 0; v < N; ++v) if (!is_point[v]) { pose_id[v] = (int)poses.size(); poses.push_back(v); }
  const int Np = (int)poses.size();
  std::vector<int> ra, rb, obs;
  for (int v = 0; v < N; ++v) {
    if (!is_point[v]) {
      for (int p = g.ptr[v]; p < g.ptr[v + 1]; ++p) { const int u = g.idx[p]; if (!is_point[u] && u > v) { ra.push_back(pose_id[v]); rb.push_back(pose_id[u]); } }
      continue;
    }
    obs.clear();
    for (int p = g.ptr[v]; p < g.ptr[v + 1]; ++p) if (!is_point[g.idx[p]]) obs.push_back(pose_id[g.idx[p]]);
    if (obs.size() <= 12) {
      for (size_t i = 0; i < obs.size(); ++i) for (size_t j = i + 1; j < obs.size(); ++j) { ra.push_back(obs[i]); rb.push_back(obs[j]); }
    } else {
      for (size_t i = 0; i + 1 < obs.size(); ++i) { ra.push_back(obs[i]); rb.push_back(obs[i + 1]); }
    }
  }
  std::vector<int> rperm;
  if (Np > 0) {
    const Csr gr = build_adjacency(Np, ra, rb);
    const Components Cr = find_components(gr);
    if (!order_components(gr, Cr, &rperm) || (int)rperm.size() != Np) return false;
  }
  // position of every pose in the reduced order; a component of the full graph takes its points (ascending), then its poses by that position
  std::vector<int> rpos(Np, 0);
  for (int k = 0; k < Np; ++k) rpos[rperm[k]] = k;
  perm->clear();
  perm->reserve(N);
  std::vector<int> cp;
  for (int c = 0; c < C.count(); ++c) {
    cp.clear();
    for (int q = C.ptr[c]; q < C.ptr[c + 1]; ++q) { const int v = C.verts[q]; if (is_point[v]) perm->push_back(v); else cp.push_back(v); }
    std::sort(cp.begin(), cp.end(), [&](int a, int b) { return rpos[pose_id[a]] < rpos[pose_id[b]]; });
    perm->insert(perm->end(), cp.begin(), cp.end());
  }
  return true;
}
static bool has_points(const std::vector<uint8_t>* is_point) {
  if (!is_point) return false;
  for (uint8_t f : *is_point) if (f) return true;
  return false;
}

bool nested_dissection_order(int N, const std::vector<int>& ia, const std::vector<int>& ib, std::vector<int>* perm, const std::vector<uint8_t>* is_point) {
  const Csr g = build_adjacency(N, ia, ib);
  const Components C = find_components(g);
  if (has_points(is_point)) return order_components_points_first(g, C, *is_point, perm) && (int)perm->size() == N;
  return order_components(g, C, perm) && (int)perm->size() == N;
}

bool direct_analyze(int N, const std::vector<int>& ia, const std::vector<int>& ib, int n_slots,
                    const std::vector<int>& slot_row, const std::vector<int>& slot_col,
                    const std::vector<uint8_t>& slot_side, const std::vector<int>& row_slot_begin,
                    DirectSymbolic* out, const std::vector<uint8_t>* is_point) {
  DirectSymbolic& S = *out;
  S = DirectSymbolic();
  S.n = N;
  const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto t_phase = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[pgo] direct analysis: %-28s %.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_phase).count());
    t_phase = now;
  };
  const Csr g = build_adjacency(N, ia, ib);
  phase("adjacency");

  // ---- 1. ordering: nested dissection per connected component ----
  const Components comps = find_components(g);
  if (has_points(is_point) ? !order_components_points_first(g, comps, *is_point, &S.perm) : !order_components(g, comps, &S.perm)) return false;
  S.iperm.assign(N, -1);
  for (int k = 0; k < N; ++k) S.iperm[S.perm[k]] = k;
  phase("nested dissection");

  static const long long max_pairs = 16000000LL;
  // ---- 2. symbolic factorisation ----
  // struct(j) = rows > j of column j, sorted: the neighbours of j plus the structures of its etree children.  Flat storage
  // (st_ptr / st_idx), children as sibling lists; a component's columns are a contiguous range of the new numbering and
  // touch nothing outside it, so the components of a batched solve are processed side by side.
  std::vector<int> parent(N, -1), first_child(N, -1), next_sibling(N, -1), mark(N, -1);
  std::vector<int> st_ptr(N + 1, 0), st_idx;
  long long nb = 0, pairs = 0;
  {
    const int ncomp = comps.count();
    std::vector<std::vector<int>> comp_idx(ncomp);          // struct entries of each component, columns in order
    std::vector<long long> comp_nb(ncomp, 0), comp_pairs(ncomp, 0);
    std::atomic<bool> too_big(false);
    for_components(comps, [&](int c, int) {
      const int j0 = comps.ptr[c], j1 = comps.ptr[c + 1];
      std::vector<int>& idx = comp_idx[c];
      std::vector<int> tmp, lptr(j1 - j0 + 1, 0);            // lptr: column starts inside idx
      long long lnb = 0, lpairs = 0;
      for (int j = j0; j < j1 && !too_big; ++j) {
        tmp.clear();
        mark[j] = j;
        const int old = S.perm[j];
        for (int p = g.ptr[old]; p < g.ptr[old + 1]; ++p) {
          const int i = S.iperm[g.idx[p]];
          if (i > j && mark[i] != j) { mark[i] = j; tmp.push_back(i); }
        }
        for (int ch = first_child[j]; ch >= 0; ch = next_sibling[ch])
          for (int q = lptr[ch - j0]; q < lptr[ch - j0 + 1]; ++q) {
            const int i = idx[q];
            if (i > j && mark[i] != j) { mark[i] = j; tmp.push_back(i); }
          }
        std::sort(tmp.begin(), tmp.end());
        idx.insert(idx.end(), tmp.begin(), tmp.end());
        lptr[j - j0 + 1] = (int)idx.size();
        st_ptr[j + 1] = (int)tmp.size();                      // sizes first, prefix sum below
        if (!tmp.empty()) { parent[j] = tmp[0]; next_sibling[j] = first_child[tmp[0]]; first_child[tmp[0]] = j; }
        lnb += 1 + (long long)tmp.size();
        lpairs += (long long)tmp.size() * ((long long)tmp.size() + 1) / 2;
        // too much fill for the enumerated schedule.  The gate below accepts ~7000 critical-path steps (Manhattan 10 k: 6.4 M
        // pairs = 9.5 k steps, rejected; KITTI-00 dense: 1.2 M pairs, accepted): a graph past this budget is going to be
        // rejected anyway, so stop before the pair lists (seconds of host time and GBs) are built.
        if (lnb > 3000000LL || lpairs > max_pairs) too_big = true;
      }
      comp_nb[c] = lnb;
      comp_pairs[c] = lpairs;
    });
    if (too_big) return false;
    for (int c = 0; c < ncomp; ++c) { nb += comp_nb[c]; pairs += comp_pairs[c]; }
    if (nb > 3000000LL || pairs > max_pairs) return false;
    for (int j = 0; j < N; ++j) st_ptr[j + 1] += st_ptr[j];
    st_idx.reserve((size_t)st_ptr[N]);
    for (int c = 0; c < ncomp; ++c) st_idx.insert(st_idx.end(), comp_idx[c].begin(), comp_idx[c].end());
  }
  auto st_size = [&](int j) { return st_ptr[j + 1] - st_ptr[j]; };
  auto st_at = [&](int j) { return st_idx.data() + st_ptr[j]; };
  S.nb = (int)nb;
  S.n_pairs = pairs;
  S.flops = 432.0 * (double)pairs + 650.0 * (double)nb;
  S.col_ptr.assign(N + 1, 0);
  for (int j = 0; j < N; ++j) S.col_ptr[j + 1] = S.col_ptr[j] + 1 + st_size(j);
  S.blk_row.resize(S.nb);
  for (int j = 0; j < N; ++j) {
    int p = S.col_ptr[j];
    S.blk_row[p++] = j;
    for (int a = 0; a < st_size(j); ++a) S.blk_row[p++] = st_at(j)[a];
  }
  auto block_of = [&](int i, int j) -> int {   // i >= j
    if (i == j) return S.col_ptr[j];
    const int* lo = &S.blk_row[S.col_ptr[j] + 1];
    const int* hi = &S.blk_row[0] + S.col_ptr[j + 1];
    const int* it = std::lower_bound(lo, hi, i);
    return (it != hi && *it == i) ? (int)(it - &S.blk_row[0]) : -1;
  };

  phase("symbolic factorisation");
  // ---- 3a. row lists (forward solve; also the source columns of every target column below) ----
  S.rowl_ptr.assign(N + 1, 0);
  for (int q = 0; q < st_ptr[N]; ++q) ++S.rowl_ptr[st_idx[q] + 1];
  for (int j = 0; j < N; ++j) S.rowl_ptr[j + 1] += S.rowl_ptr[j];
  S.rowl_blk.resize(S.rowl_ptr[N]);
  S.rowl_col.resize(S.rowl_ptr[N]);
  {
    // (a row and the columns that reach it belong to one component: the components of a batched solve side by side)
    std::vector<int> fill(S.rowl_ptr.begin(), S.rowl_ptr.end() - 1);
    for_components(comps, [&](int c, int) {
      for (int k = comps.ptr[c]; k < comps.ptr[c + 1]; ++k)
        for (int a = 0; a < st_size(k); ++a) {
          const int q = fill[st_at(k)[a]]++;
          S.rowl_blk[q] = S.col_ptr[k] + 1 + a;
          S.rowl_col[q] = k;
        }
    });
  }
  phase("row lists");
  // ---- 3b. update pairs per target block ----
  // Target-centric: the pairs of the blocks (i, j) of column j come from the columns k of row j (ascending k = the fixed
  // summation order): with a = position of j in struct(k), the targets (sk[b], j), b >= a, are found by one forward
  // walk through column j's row list.  All writes of column j fall in its own range of the pair arrays, so columns are
  // processed by several host threads; two passes (count, fill) around one prefix sum.
  S.upd_ptr.assign(S.nb + 1, 0);
  std::atomic<bool> broken(false);
  auto walk_column = [&](int j, auto&& emit) {
    const int* lo = &S.blk_row[0] + S.col_ptr[j] + 1;
    const int* hi = &S.blk_row[0] + S.col_ptr[j + 1];
    for (int q = S.rowl_ptr[j]; q < S.rowl_ptr[j + 1]; ++q) {
      const int k = S.rowl_col[q];
      const int* sk = st_at(k);
      const size_t sk_size = (size_t)st_size(k);
      const int base = S.col_ptr[k] + 1;
      const size_t a = (size_t)(S.rowl_blk[q] - base);
      emit(S.col_ptr[j], base + (int)a, base + (int)a);
      const int* it = lo;
      for (size_t b = a + 1; b < sk_size; ++b) {
        const int i = sk[b];
        for (int hop = 0; hop < 4 && it != hi && *it < i; ++hop) ++it;
        if (it != hi && *it < i) it = std::lower_bound(it, hi, i);
        if (it == hi || *it != i) { broken = true; return; }   // cannot happen: struct(k) \ {j} is contained in struct(j)
        emit((int)(it - &S.blk_row[0]), base + (int)b, base + (int)a);   // L(i,k), i = sk[b];  L(j,k)
      }
    }
  };
  auto for_columns = [&](auto&& body) {
    if (comps.count() > 1) {   // the graphs of a batched solve: one component at a time per thread
      for_components(comps, [&](int c, int) { for (int j = comps.ptr[c + 1] - 1; j >= comps.ptr[c]; --j) body(j); });
      return;
    }
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = (pairs < 200000 || hw < 2) ? 1 : std::min(hw, 16);
    if (nt <= 1) { for (int j = 0; j < N; ++j) body(j); return; }
    std::atomic<int> next(0);
    HostPool::get().run(nt, [&](int) {
      for (;;) {   // the work sits in the separator columns at the end: small chunks, handed out from the end
        const int c = next.fetch_add(16);
        if (c >= N) return;
        for (int j = N - 1 - c; j >= std::max(0, N - 16 - c); --j) body(j);
      }
    });
  };
  for_columns([&](int j) { walk_column(j, [&](int t, int, int) { ++S.upd_ptr[t + 1]; }); });
  if (broken) return false;
  for (int t = 0; t < S.nb; ++t) S.upd_ptr[t + 1] += S.upd_ptr[t];
  if ((long long)S.upd_ptr[S.nb] != pairs) return false;
  S.upd_a.resize(S.upd_ptr[S.nb]);
  S.upd_b.resize(S.upd_ptr[S.nb]);
  {
    std::vector<int> fill(S.upd_ptr.begin(), S.upd_ptr.end() - 1);
    for_columns([&](int j) {
      walk_column(j, [&](int t, int la, int lb) { const int q = fill[t]++; S.upd_a[q] = la; S.upd_b[q] = lb; });
    });
  }

  phase("update pairs");
  // ---- 3c. BSR sources per block ----
  S.asrc_ptr.assign(S.nb + 1, 0);
  std::vector<int> src_block(n_slots, -1);
  parallel_ranges(n_slots, 16384, [&](int lo, int hi) {
    for (int t = lo; t < hi; ++t) {
      const uint8_t side = slot_side[t];
      if (side == SIDE_PAD) continue;
      const int i = S.iperm[slot_row[t]];
      if (side == SIDE_DIAG) { src_block[t] = S.col_ptr[i]; continue; }
      const int j = S.iperm[slot_col[t]];
      if (i > j) src_block[t] = block_of(i, j);   // the (j,i) twin slot carries the transposed block: skipped
    }
  });
  for (int t = 0; t < n_slots; ++t) if (src_block[t] >= 0) ++S.asrc_ptr[src_block[t] + 1];
  for (int t = 0; t < S.nb; ++t) S.asrc_ptr[t + 1] += S.asrc_ptr[t];
  S.asrc_slot.resize(S.asrc_ptr[S.nb]);
  {
    std::vector<int> fill(S.asrc_ptr.begin(), S.asrc_ptr.end() - 1);
    for (int t = 0; t < n_slots; ++t) if (src_block[t] >= 0) S.asrc_slot[fill[src_block[t]]++] = t;
  }
  (void)row_slot_begin;

  phase("slot sources");
  // ---- 4. levels ----
  std::vector<int> level(N, 0);
  int max_level = 0;
  for (int j = 0; j < N; ++j) {   // children have smaller indices: level[j] is final when j is reached
    max_level = std::max(max_level, level[j]);
    if (parent[j] >= 0) level[parent[j]] = std::max(level[parent[j]], level[j] + 1);
  }
  S.n_levels = max_level + 1;
  S.level_ptr.assign(S.n_levels + 1, 0);
  for (int j = 0; j < N; ++j) ++S.level_ptr[level[j] + 1];
  for (int l = 0; l < S.n_levels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
  S.level_cols.resize(N);
  {
    std::vector<int> fill(S.level_ptr.begin(), S.level_ptr.end() - 1);
    for (int j = 0; j < N; ++j) S.level_cols[fill[level[j]]++] = j;
  }
  S.blk_lpos.resize(S.nb);
  {
    std::vector<int> lpos(N);
    for (int q = 0; q < N; ++q) lpos[S.level_cols[q]] = q;
    parallel_ranges(S.nb, 32768, [&](int lo, int hi) { for (int b = lo; b < hi; ++b) S.blk_lpos[b] = lpos[S.blk_row[b]]; });
  }
  // fused tail: the longest suffix of levels that each hold at most 8 columns
  S.fused_from_level = S.n_levels;
  while (S.fused_from_level > 0 && S.level_ptr[S.fused_from_level] - S.level_ptr[S.fused_from_level - 1] <= 8) --S.fused_from_level;

  phase("levels");
  // ---- 5. schedule + cost model: serial "pair steps" on the critical path ----
  // COLUMN/FUSED: a column is processed ten blocks at a time by one wave; a block's update list is walked serially by its
  // 6-lane group(s).  SPLIT: every block of the level has its own wave (ten groups share its list), then a cheap
  // per-column pass.  A level is split when its one-wave-per-column cost exceeds HEAVY steps.
  static const double HEAVY = 16.0;
  std::vector<double> level_cost(S.n_levels, 0.0), split_cost(S.n_levels, 0.0);
  {
    // per column: the cost of walking it with one wave, and of its heaviest block; maxima per level (positions of level_cols
    // side by side, every piece with maxima of its own, folded afterwards)
    std::mutex fold;
    std::vector<double> worst_blk(S.n_levels, 0.0);
    parallel_ranges(N, 8192, [&](int qlo, int qhi) {
      std::vector<double> w(S.n_levels, 0.0), wb(S.n_levels, 0.0);
      int l = (int)(std::upper_bound(S.level_ptr.begin(), S.level_ptr.end(), qlo) - S.level_ptr.begin()) - 1;
      for (int q = qlo; q < qhi; ++q) {
        while (q >= S.level_ptr[l + 1]) ++l;
        const int j = S.level_cols[q];
        const int b0 = S.col_ptr[j], nblk = S.col_ptr[j + 1] - b0;
        double col = 1.0 + (S.upd_ptr[b0 + 1] - S.upd_ptr[b0] + 9) / 10;
        for (int t = 1; t < nblk; t += 10) {
          const int bc = std::min(10, nblk - t), gpb = 10 / bc;
          int mx = 0;
          for (int u = t; u < t + bc; ++u) mx = std::max(mx, (S.upd_ptr[b0 + u + 1] - S.upd_ptr[b0 + u] + gpb - 1) / gpb);
          col += 1.0 + mx;
        }
        w[l] = std::max(w[l], col);
        for (int u = 0; u < nblk; ++u) wb[l] = std::max(wb[l], 1.0 + (S.upd_ptr[b0 + u + 1] - S.upd_ptr[b0 + u] + 9) / 10);
      }
      std::lock_guard<std::mutex> lk(fold);
      for (int k = 0; k < S.n_levels; ++k) { level_cost[k] = std::max(level_cost[k], w[k]); worst_blk[k] = std::max(worst_blk[k], wb[k]); }
    });
    for (int l = 0; l < S.n_levels; ++l) split_cost[l] = worst_blk[l] + 2.0 + 4.0;   // assemble critical path + one scaling step + one more launch
  }
  S.steps.clear();
  S.split_blk.clear(); S.split_diag.clear(); S.split_sub.clear(); S.split_sub_diag.clear(); S.panel_cols.clear(); S.split_dblk.clear();
  S.upd_split.assign(S.nb, 0);
  parallel_ranges(S.nb, 32768, [&](int lo, int hi) { for (int t = lo; t < hi; ++t) S.upd_split[t] = S.upd_ptr[t + 1]; });
  std::vector<int> blk_col(S.nb);                 // column of every block of L (source column of an update pair)
  parallel_ranges(N, 8192, [&](int lo, int hi) {
    for (int j = lo; j < hi; ++j) for (int bi = S.col_ptr[j]; bi < S.col_ptr[j + 1]; ++bi) blk_col[bi] = j;
  });
  static const int PANEL_MAX = 8;
  auto is_heavy = [&](int l) { return level_cost[l] > HEAVY && split_cost[l] < level_cost[l]; };
  std::vector<int> tmp_a, tmp_b, tmp_pa, tmp_pb, chain_pos(N, -1);
  double steps = 0.0;
  // Light levels of the fused tail (chain-like tops: one wave per column, ~10 us per level of dependent round trips) also
  // go through PANEL when at least LIGHT_PANEL_MIN consecutive levels form parent chains: two launches, then ~3 us per
  // column with eight waves on each column.
  const int LIGHT_PANEL_MIN = 3;
  auto panel_ok = [&](int l) { return is_heavy(l) || l >= S.fused_from_level; };
  // chains of columns starting at level l: the next level has the same number of columns (<= 8), each the etree parent
  // of one column here; returns the width
  auto grow = [&](int l, std::vector<std::vector<int>>& chain) {
    const int nc = S.level_ptr[l + 1] - S.level_ptr[l];
    chain.assign(nc, std::vector<int>());
    for (int c = 0; c < nc; ++c) chain[c].push_back(S.level_cols[S.level_ptr[l] + c]);
    int w = 1;
    while (nc <= 8 && w < PANEL_MAX && l + w < S.n_levels && panel_ok(l + w) && is_heavy(l + w) == is_heavy(l) &&
           S.level_ptr[l + w + 1] - S.level_ptr[l + w] == nc) {
      std::vector<int> next(nc, -1);
      bool ok = true;
      for (int c = 0; c < nc && ok; ++c) {
        const int par = parent[chain[c].back()];
        bool found = false;
        for (int q = S.level_ptr[l + w]; q < S.level_ptr[l + w + 1]; ++q) if (S.level_cols[q] == par) found = true;
        for (int c2 = 0; c2 < c; ++c2) if (next[c2] == par) found = false;   // two chains merging: stop here
        if (!found) ok = false; else next[c] = par;
      }
      if (!ok) break;
      for (int c = 0; c < nc; ++c) chain[c].push_back(next[c]);
      ++w;
    }
    return w;
  };
  std::vector<std::vector<int>> chain, probe;
  auto light_panel_at = [&](int l) { return !is_heavy(l) && l >= S.fused_from_level && grow(l, probe) >= LIGHT_PANEL_MIN; };
  for (int l = 0; l < S.n_levels;) {
    if (panel_ok(l) && (is_heavy(l) || light_panel_at(l))) {
      const int nc = S.level_ptr[l + 1] - S.level_ptr[l];
      const int w = grow(l, chain);
      if (w >= 2) {
        DirectStep st{DirectStep::PANEL, l, l + w, (int)S.split_blk.size(), 0, (int)S.panel_cols.size(), nc};
        double p1 = 0.0, p2 = 0.0;
        for (int c = 0; c < nc; ++c) {
          const int first = chain[c][0];     // columns of one chain are consecutive in the elimination order? not assumed
          for (int i = 0; i < w; ++i) chain_pos[chain[c][i]] = i;   // position within this chain, -1 elsewhere
          double chain_p2 = 0.0;
          for (int i = 0; i < w; ++i) {
            const int j = chain[c][i];
            S.panel_cols.push_back(j);
            const int nblk = S.col_ptr[j + 1] - S.col_ptr[j];
            for (int bi = S.col_ptr[j]; bi < S.col_ptr[j + 1]; ++bi) {
              S.split_blk.push_back(bi);
              S.split_diag.push_back(0);
              S.split_dblk.push_back(S.col_ptr[j]);
              // in-panel pairs = those whose source column is one of this chain's earlier panel columns; they are moved
              // (stably) behind the pairs of the columns before the panel, whose L blocks are final when phase 1 runs
              int q = S.upd_ptr[bi + 1];
              if (i > 0) {
                tmp_a.clear(); tmp_b.clear(); tmp_pa.clear(); tmp_pb.clear();
                for (int u = S.upd_ptr[bi]; u < S.upd_ptr[bi + 1]; ++u) {
                  const int k = blk_col[S.upd_a[u]];
                  const bool in_panel = chain_pos[k] >= 0 && chain_pos[k] < i;
                  (in_panel ? tmp_pa : tmp_a).push_back(S.upd_a[u]);
                  (in_panel ? tmp_pb : tmp_b).push_back(S.upd_b[u]);
                }
                int u = S.upd_ptr[bi];
                for (size_t z = 0; z < tmp_a.size(); ++z, ++u) { S.upd_a[u] = tmp_a[z]; S.upd_b[u] = tmp_b[z]; }
                q = u;
                for (size_t z = 0; z < tmp_pa.size(); ++z, ++u) { S.upd_a[u] = tmp_pa[z]; S.upd_b[u] = tmp_pb[z]; }
              }
              S.upd_split[bi] = q;
              p1 = std::max(p1, 1.0 + (q - S.upd_ptr[bi] + 9) / 10);
            }
            chain_p2 += 3.0 + ((nblk - 1 + 79) / 80) * (1.0 + i);
          }
          (void)first;
          for (int i = 0; i < w; ++i) chain_pos[chain[c][i]] = -1;
          p2 = std::max(p2, chain_p2);
        }
        st.blk_end = (int)S.split_blk.size();
        S.steps.push_back(st);
        steps += p1 + p2 + 8.0;
        l += w;
        continue;
      }
      DirectStep st{DirectStep::SPLIT, l, l + 1, (int)S.split_blk.size(), 0, (int)S.split_sub.size(), 0};
      for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
        const int j = S.level_cols[q];
        for (int bi = S.col_ptr[j]; bi < S.col_ptr[j + 1]; ++bi) {
          S.split_blk.push_back(bi);
          S.split_diag.push_back(bi == S.col_ptr[j] ? 1 : 0);
          S.split_dblk.push_back(S.col_ptr[j]);
          if (bi != S.col_ptr[j]) { S.split_sub.push_back(bi); S.split_sub_diag.push_back(S.col_ptr[j]); }
        }
      }
      st.blk_end = (int)S.split_blk.size();
      st.sub_end = (int)S.split_sub.size();
      S.steps.push_back(st);
      steps += split_cost[l];
      ++l;
    } else if (l >= S.fused_from_level) {
      int e = l;
      while (e < S.n_levels && !is_heavy(e) && !(e > l && light_panel_at(e))) { steps += level_cost[e]; ++e; }
      S.steps.push_back(DirectStep{DirectStep::FUSED, l, e, 0, 0, 0, 0});
      l = e;
    } else {
      S.steps.push_back(DirectStep{DirectStep::COLUMN, l, l + 1, 0, 0, 0, 0});
      steps += level_cost[l];
      ++l;
    }
  }
  // triangular solves: two launches per level before the fused tail, ~6 us of in-workgroup hand-over per tail level plus
  // the row / column lists shared by 80 six-lane groups
  for (int l = 0; l < S.n_levels; ++l) {
    if (l < S.fused_from_level) { steps += 8.0; continue; }
    double worst = 0.0;
    for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
      const int j = S.level_cols[q];
      worst = std::max(worst, (double)(S.rowl_ptr[j + 1] - S.rowl_ptr[j] + S.col_ptr[j + 1] - S.col_ptr[j]) / 80.0);
    }
    steps += 8.0 + worst;
  }
  S.est_steps = steps;
  phase("schedule");
  {
    int n_split = 0, n_panel = 0;
    for (const DirectStep& st : S.steps) { n_split += st.type == DirectStep::SPLIT; n_panel += st.type == DirectStep::PANEL; }
    if (getenv("PGO_VERBOSE"))
      std::fprintf(stderr, "[pgo] direct: n=%d blocks=%d pairs=%lld levels=%d (fused from %d, %d split, %d panels, %zu launches) est_steps=%.0f\n",
                   S.n, S.nb, S.n_pairs, S.n_levels, S.fused_from_level, n_split, n_panel, S.steps.size() + n_split + n_panel, steps);
  }
  // The alternative for an exact request is PCG run to 1e-13, whose cost depends on the conditioning, not on the fill:
  // ~8 ms per solve on Manhattan 10 k (well conditioned, est 8.7 k steps = 12 ms direct), ~40 ms on KITTI-00 dense (est
  // 4.2 ms direct), seconds on the open chain of the KITTI-00 replay.  One step is ~1 us; the budget below
  // (PGO_DIRECT_MAX_STEPS to override) sends the well-conditioned mesh-like graphs to PCG and keeps the rest direct.
  // Between that budget and PGO_DIRECT_HYBRID_STEPS the factorisation is kept as one of two ways to serve an iteration
  // (`hybrid`): early LM iterations of a mesh are ill-conditioned (Manhattan 10 k: ~3000 CG iterations = 42 ms vs 6.9 ms
  // direct), late ones are not (~90 CG iterations = 1.3 ms).
  const double max_steps = getenv("PGO_DIRECT_MAX_STEPS") ? atof(getenv("PGO_DIRECT_MAX_STEPS")) : 7000.0;
  const double hybrid_steps = 30000.0;
  if (S.steps.size() > 4000 || steps > std::max(max_steps, hybrid_steps)) return false;
  S.hybrid = steps > max_steps;
  return true;
}

}  // namespace pgo
