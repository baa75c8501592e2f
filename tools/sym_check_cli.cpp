// tools/sym_check_cli.cpp — host-only check of the symmetric tile form's layout (csrc/pgo_sym_host.cpp): random graphs, the layout built
// exactly as sym_prepare() builds it, and the kernel that consumes it EMULATED in scalar code on the host:
//   * the product (k_spmv_sym): chunk by chunk, u = H x_col at the lane's position, v = H^T x_row at vpos, row r adds its ranges
//     [ub, ub + uc) and [vb, vb + vc) — against a plain sum over all incidences;
// plus the invariants the kernels rely on (one diagonal slot per row, an interior edge stored once in the begin orientation, a cut edge
// twice, x indices in range, chunk bases multiples of 64).  Needs no GPU: tests/test_sym_host.py runs it in the CPU suite.
// r06: with a fourth argument `1` every case builds the form of ONE RANK's rows — a random owned range [lo, hi), lo even, tiles made of
// whole 2-pose units (SymHostParams::row_lo / row_hi / unit) — and checks, besides the product of the owned rows, that no other row sits
// in a tile, that an edge is stored once per owned end (once in all when interior), and that the poses of a unit occupy consecutive
// lanes starting at an even one (what k_pipe_cg_sym's Jacobi-block step relies on).
// usage: sym_check_cli [cases] [first_seed] [damage 1..4: self-test, one damaged layout entry per case must be noticed] [ranks 0 / 1]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "../posegraph-ceres_amd/csrc/pgo_sym_host.h"

using pgo::SymTile;

static long long g_unfit = 0, g_tiles = 0, g_interior = 0, g_stored = 0;
static int g_mutate = 0;      // self-test: damage one entry of the layout and expect the checks to notice
static int g_ranks = 0;       // 1: the form of one rank's rows (owned range + 2-pose units)

static int check_case(unsigned seed) {
  std::mt19937_64 rng(seed);
  auto U = [&](int n) { return (int)(rng() % (unsigned long long)std::max(1, n)); };
  const int kind = U(4);
  const int N = 2 + U(kind == 0 ? 4000 : 1500);
  std::vector<int> ia, ib;
  for (int v = 1; v < N; ++v) { ia.push_back(v); ib.push_back(v - 1); }
  const int extra = U(5 * N);
  const int hub0 = U(N), hub1 = U(N);
  for (int k = 0; k < extra; ++k) {
    int a = U(N), b = kind == 1 ? (a + 1 + U(40)) % N : U(N);       // kind 1: local chords (high interior fraction)
    if (kind == 2 && (k & 1)) b = (k & 2) ? hub0 : hub1;             // kind 2: hubs with hundreds of incidences
    if (kind == 3 && k < extra / 3 && !ia.empty()) { a = ia[N - 1 < (int)ia.size() ? N - 1 : 0]; b = ib[N - 1 < (int)ib.size() ? N - 1 : 0]; }   // duplicates
    if (a != b) { ia.push_back(a); ib.push_back(b); }
  }
  const int E = (int)ia.size();
  int lo = 0, hi = N;
  const int unit = g_ranks ? 2 : 1;
  if (g_ranks) { const int world = 2 + U(7), rk = U(world), per = ((N + world - 1) / world + 3) / 4 * 4; lo = std::min(N, rk * per); hi = std::min(N, lo + per); if (lo >= hi) { lo = 0; hi = std::min(N, per); } }
  auto owned = [&](int v) { return v >= lo && v < hi; };
  // incidence-slot numbering as prepare() lays it out: the (owned) row's diagonal, then its incidences in edge order
  std::vector<int> deg(N, 0), rsb(N, 0);
  for (int e = 0; e < E; ++e) { ++deg[ia[e]]; ++deg[ib[e]]; }
  int n_old = 0;
  for (int v = lo; v < hi; ++v) { rsb[v] = n_old; n_old += 1 + deg[v]; }
  std::vector<int> fill(N, 0), beg(E, -1), end(E, -1);
  for (int v = lo; v < hi; ++v) fill[v] = rsb[v] + 1;
  for (int e = 0; e < E; ++e) { if (owned(ia[e])) beg[e] = fill[ia[e]]++; if (owned(ib[e])) end[e] = fill[ib[e]]++; }
  // a scalar per slot stands for its 6x6 block: H_ab = h[e], H_ba = its "transpose" (same scalar), diagonal d[v]
  std::vector<double> h_old(n_old, 0.0), x(N), y_ref(N, 0.0);
  std::uniform_real_distribution<double> ud(-1.0, 1.0);
  for (int v = 0; v < N; ++v) { x[v] = ud(rng); const double d = 2.0 + ud(rng); if (owned(v)) h_old[rsb[v]] = d; y_ref[v] = d * x[v]; }
  for (int e = 0; e < E; ++e) { const double w = ud(rng); if (beg[e] >= 0) h_old[beg[e]] = w; if (end[e] >= 0) h_old[end[e]] = w; y_ref[ia[e]] += w * x[ib[e]]; y_ref[ib[e]] += w * x[ia[e]]; }

  pgo::SymHostParams hp;
  const int caps[] = {8, 16, 33, 64, 256};
  hp.row_cap = caps[U(5)];
  hp.w_cap = std::max<long long>(64, (long long)((0.6 + 0.1 * U(8)) * hp.row_cap * (double)(N + 2LL * E) / N));
  hp.sort_tiles = U(2) != 0;
  if (g_ranks) { hp.row_lo = lo; hp.row_hi = hi; hp.unit = unit; }
  pgo::SymHostLayout L;
  pgo::sym_build_host(N, E, ia.data(), ib.data(), rsb.data(), hp, &L);
  if (L.unfit) { ++g_unfit; return g_mutate ? 1 : 0; }       // a declared "does not fit" is a valid outcome (the incidence-slot kernels stay)
  g_tiles += (long long)L.tiles.size(); g_interior += L.interior_edges; g_stored += L.stored;
  if (g_mutate) {
    const size_t at = (size_t)(rng() % L.meta.size());
    const int pick = L.src_slot[at] >= 0 ? g_mutate : 0;
    if (pick == 1) L.meta[at] ^= 1u << 23;                               // another row
    else if (pick == 2) L.meta[at] ^= 1u;                                // another column
    else if (pick == 3) { size_t r = (size_t)(rng() % L.rinfo.size()); while (!L.rinfo[r]) r = (r + 1) % L.rinfo.size(); L.rinfo[r] += 1u << 8; }   // a longer u range
    else if (pick == 4) { size_t r = at, tries = 0; while (!((L.meta[r] >> 14) & 1u) && tries++ < L.meta.size()) r = (r + 1) % L.meta.size();
                          if (!((L.meta[r] >> 14) & 1u)) return 1; L.meta[r] ^= 1u << 15; }   // the position of a mirrored product moved
    else return 1;
  }
  int bad = 0;
  auto fail = [&](const char* what, long long a = 0, long long b = 0) { if (bad++ < 5) std::fprintf(stderr, "seed %u (N %d E %d rows %d): %s (%lld, %lld)\n", seed, N, E, hp.row_cap, what, a, b); };
  // ---- invariants ----
  std::vector<int> seen_old(n_old, 0), row_tile(N, -1);
  long long stored = 0, interior = 0;
  for (size_t t = 0; t < L.tiles.size(); ++t) {
    const SymTile& T = L.tiles[t];
    if (T.nrows > hp.row_cap || T.nrows > (int)pgo::SYM_LANES) fail("tile has too many rows", T.nrows);
    if (T.nx > L.x_cap || T.nx > (int)pgo::SYM_X_MAX) fail("x_cap", T.nx, L.x_cap);
    for (int i = 0; i < T.nrows; ++i) { const int v = L.xlist[T.x0 + i]; if (row_tile[v] != -1) fail("row in two tiles", v); row_tile[v] = (int)t; }
    int total = 0;
    for (int c = 0; c < T.nchunks; ++c) {
      const int ci = T.chunk0 + c, base = L.chunk_base[ci], n = L.chunk_n[ci];
      if (base % 64 || base != T.base0 + 256 * c || n != std::min(256, T.total - 256 * c) || n <= 0) fail("chunk table", base, n);
      total += n;
    }
    if (total != T.total) fail("chunk sizes do not add up", total, T.total);
    stored += T.total;
  }
  for (int v = 0; v < N; ++v) { if (owned(v) && row_tile[v] < 0) fail("row in no tile", v); if (!owned(v) && row_tile[v] >= 0) fail("another rank's row in a tile", v); }
  if (unit > 1)
    for (size_t t = 0; t < L.tiles.size(); ++t) {
      const SymTile& T = L.tiles[t];
      for (int i = 0; i < T.nrows; ++i) {
        const int v = L.xlist[T.x0 + i];
        if ((v - lo) % unit == 0) { if (i % unit) fail("a unit starts at an odd lane", v, i); if (v + 1 < hi && (i + 1 >= T.nrows || L.xlist[T.x0 + i + 1] != v + 1)) fail("a unit is split", v); }
      }
    }
  // ---- the product, emulated ----
  std::vector<double> y(N, 0.0);
  for (size_t t = 0; t < L.tiles.size(); ++t) {
    const SymTile& T = L.tiles[t];
    std::vector<double> acc(T.nrows, 0.0);
    for (int c = 0; c < T.nchunks; ++c) {
      const int ci = T.chunk0 + c, base = L.chunk_base[ci], n = L.chunk_n[ci];
      std::vector<double> ubuf(256, 0.0), vbuf(256, 0.0);
      std::vector<int> vused(256, 0);
      for (int l = 0; l < n; ++l) {
        const uint32_t m = L.meta[base + l];
        const int xcol = m & 0xFFF, side = (m >> 12) & 3, inter = (m >> 14) & 1, vpos = (m >> 15) & 0xFF, xrow = (m >> 23) & 0xFF;
        const int src = L.src_slot[base + l];
        if (src < 0 || src >= n_old) { fail("src_slot out of range", src); continue; }
        if (seen_old[src]++) fail("incidence slot stored twice", src);
        if (xcol >= T.nx || xrow >= T.nrows) { fail("x index out of range", xcol, xrow); continue; }
        const int col = L.xlist[T.x0 + xcol], row = L.xlist[T.x0 + xrow];
        if (side == pgo::SIDE_DIAG) { if (src != rsb[row] || xcol != xrow || L.diag_slot[row] != base + l) fail("diagonal slot", row); }
        if (inter) { ++interior; if (side != pgo::SIDE_BEGIN || xcol >= T.nrows) fail("interior slot not in begin orientation / column not a tile row", row, col); }
        else if (side != pgo::SIDE_DIAG && xcol < T.nrows) fail("cut slot whose column is a tile row", row, col);
        ubuf[l] = h_old[src] * x[col];
        if (inter) { if (vused[vpos]++) fail("two v entries at one position", vpos); vbuf[vpos] = h_old[src] * x[row]; }
      }
      for (int r = 0; r < T.nrows; ++r) {
        const uint32_t w = L.rinfo[(size_t)ci * 256 + r];
        const int ub = w & 0xFF, uc = (w >> 8) & 0x1FF, vb = (w >> 17) & 0xFF, vc = w >> 25;
        if (ub + uc > n || vb + vc > 256) { fail("row range out of the chunk", ub + uc, vb + vc); continue; }
        for (int j = 0; j < uc; ++j) {
          if (((L.meta[base + ub + j] >> 23) & 0xFF) != (uint32_t)r) fail("u range holds another row's slot", r);
          acc[r] += ubuf[ub + j];
        }
        for (int j = 0; j < vc; ++j) { if (!vused[vb + j]) fail("v range reads an unwritten position", r, vb + j); acc[r] += vbuf[vb + j]; }
      }
      for (int r = T.nrows; r < 256; ++r) if (L.rinfo[(size_t)ci * 256 + r]) fail("range for a lane that is no row", r);
    }
    for (int r = 0; r < T.nrows; ++r) y[L.xlist[T.x0 + r]] = acc[r];
  }
  double worst = 0;
  for (int v = lo; v < hi; ++v) worst = std::max(worst, std::fabs(y[v] - y_ref[v]) / (1.0 + std::fabs(y_ref[v])));
  if (worst > 1e-12) fail("emulated product differs from the plain sum", (long long)(worst * 1e15));
  // an interior edge once, a cut edge twice, every diagonal once
  if (stored != L.stored || interior != L.interior_edges || stored != (long long)n_old - interior) fail("stored-slot count", stored, (long long)n_old - interior);
  for (int e = 0; e < E; ++e) {
    const bool oa = owned(ia[e]), ob = owned(ib[e]);
    const int sb = oa ? seen_old[beg[e]] : -1, se = ob ? seen_old[end[e]] : -1;
    if (oa && ob) { if (!(sb == 1 && (se == 0 || se == 1)) || (se == 0) != (row_tile[ia[e]] == row_tile[ib[e]])) fail("edge storage", e, sb * 10 + se); }
    else if ((oa && sb != 1) || (ob && se != 1)) fail("edge with one owned end", e, sb * 10 + se);
  }
  return bad;
}

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 200;
  const unsigned first = argc > 2 ? (unsigned)atoi(argv[2]) : 0u;
  g_mutate = argc > 3 ? atoi(argv[3]) : 0;
  g_ranks = argc > 4 ? atoi(argv[4]) : 0;
  int bad = 0;
  if (g_mutate) std::fclose(stderr);
  for (int k = 0; k < cases; ++k) bad += check_case(first + k) != 0;
  std::printf("sym_check: %d cases, %d bad, %lld unfit, %lld tiles, %lld stored slots, %lld interior edges%s\n", cases, bad, g_unfit, g_tiles, g_stored,
              g_interior, g_mutate ? " (damaged layouts: every case must be bad)" : "");
  if (g_mutate) return bad == cases ? 0 : 1;
  return bad ? 1 : 0;
}
