// tools/direct_analyze_cli.cpp — runs the HOST symbolic analysis of the GPU Cholesky (pgo_direct.cpp) on an edge list, no
// GPU needed: timing / statistics of the ordering, fill and launch schedule.  Build: see tools/Makefile (hipcc, host only).
// usage: direct_analyze_cli <edges.txt> [-v]     (first line: N E, then E lines "id_begin id_end")
#include <chrono>
#include <cstdio>
#include <vector>

#include "../posegraph-ceres_amd/csrc/pgo_direct.h"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s edges.txt\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 2;
  int N = 0, E = 0;
  if (std::fscanf(f, "%d %d", &N, &E) != 2) return 2;
  std::vector<int> ia(E), ib(E);
  for (int e = 0; e < E; ++e) if (std::fscanf(f, "%d %d", &ia[e], &ib[e]) != 2) return 2;
  std::fclose(f);
  // incidence slots as prepare() builds them for one rank (diagonal first, then incidences in edge order)
  std::vector<int> deg(N, 0), row_slot_begin(N, 0);
  for (int e = 0; e < E; ++e) { ++deg[ia[e]]; ++deg[ib[e]]; }
  int n_slots = 0;
  for (int v = 0; v < N; ++v) { row_slot_begin[v] = n_slots; n_slots += 1 + deg[v]; }
  std::vector<int> slot_row(n_slots), slot_col(n_slots), fill(N);
  std::vector<uint8_t> slot_side(n_slots);
  for (int v = 0; v < N; ++v) { const int t = row_slot_begin[v]; slot_row[t] = v; slot_col[t] = v; slot_side[t] = pgo::SIDE_DIAG; fill[v] = t + 1; }
  for (int e = 0; e < E; ++e) {
    int t = fill[ia[e]]++; slot_row[t] = ia[e]; slot_col[t] = ib[e]; slot_side[t] = pgo::SIDE_BEGIN;
    t = fill[ib[e]]++; slot_row[t] = ib[e]; slot_col[t] = ia[e]; slot_side[t] = pgo::SIDE_END;
  }
  pgo::DirectSymbolic S;
  const auto t0 = std::chrono::steady_clock::now();
  const bool ok = pgo::direct_analyze(N, ia, ib, n_slots, slot_row, slot_col, slot_side, row_slot_begin, &S);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("N %d E %d usable %d seconds %.3f blocks %d pairs %lld levels %d steps %zu est_steps %.0f\n", N, E, ok ? 1 : 0, dt, S.nb,
              S.n_pairs, S.n_levels, S.steps.size(), S.est_steps);
  if (argc > 2) {   // per-step detail: levels, columns, blocks, update pairs, longest per-column pair list
    static const char* names[] = {"COLUMN", "FUSED", "SPLIT", "PANEL"};
    for (const pgo::DirectStep& st : S.steps) {
      long long cols = 0, blks = 0, pairs = 0, worst = 0, rowl = 0, crit = 0;
      for (int l = st.level_begin; l < st.level_end; ++l) {
        long long level_worst = 0;
        for (int c = S.level_ptr[l]; c < S.level_ptr[l + 1]; ++c) {
          const int j = S.level_cols[c];
          const long long pj = S.upd_ptr[S.col_ptr[j + 1]] - S.upd_ptr[S.col_ptr[j]];
          ++cols; blks += S.col_ptr[j + 1] - S.col_ptr[j]; pairs += pj; rowl += S.rowl_ptr[j + 1] - S.rowl_ptr[j];
          if (pj > level_worst) level_worst = pj;
        }
        if (level_worst > worst) worst = level_worst;
        crit += level_worst;
      }
      std::printf("%-6s levels [%d,%d) cols %lld blocks %lld pairs %lld worst-column pairs %lld sum-of-level-worst %lld row-list %lld\n",
                  names[st.type], st.level_begin, st.level_end, cols, blks, pairs, worst, crit, rowl);
    }
  }
  return 0;
}
