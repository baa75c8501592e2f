// tools/front_check_cli.cpp — development check of the multifrontal solver's HOST analysis (pgo_front.cpp) without a GPU:
// runs front_analyze on an edge list, then executes the launch schedule with plain scalar loops that state what every
// kernel of pgo_front_kernels.hip has to do (scatter, extend-add, POTRF + inverse, TRSM, GEMM, backward substitution) on
// a random symmetric positive definite block matrix of that sparsity, and reports |A x - b| / |b|.
// Test infrastructure only; the product never contains these loops.  Build: tools/Makefile (hipcc, host code only).
// usage: front_check_cli <edges.txt>     (first line: N E, then E lines "id_begin id_end")
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../posegraph-ceres_amd/csrc/pgo_front.h"

using namespace pgo;

int main(int argc, char** argv) {
  // (this tool emulates the regular round schedule only: the knob front_mixed stays at its default, 0 — the small-front kernels are checked on the GPU)
  if (argc < 2) { std::fprintf(stderr, "usage: %s edges.txt\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 2;
  int N = 0, E = 0;
  if (std::fscanf(f, "%d %d", &N, &E) != 2) return 2;
  std::vector<int> ia(E), ib(E);
  for (int e = 0; e < E; ++e) if (std::fscanf(f, "%d %d", &ia[e], &ib[e]) != 2) return 2;
  std::fclose(f);
  std::vector<int> deg(N, 0), row_slot_begin(N, 0);
  for (int e = 0; e < E; ++e) { ++deg[ia[e]]; ++deg[ib[e]]; }
  int n_slots = 0;
  for (int v = 0; v < N; ++v) { row_slot_begin[v] = n_slots; n_slots += 1 + deg[v]; }
  std::vector<int> slot_row(n_slots), slot_col(n_slots), fill(N), slot_edge(n_slots, -1);
  std::vector<uint8_t> slot_side(n_slots);
  for (int v = 0; v < N; ++v) { const int t = row_slot_begin[v]; slot_row[t] = v; slot_col[t] = v; slot_side[t] = SIDE_DIAG; fill[v] = t + 1; }
  for (int e = 0; e < E; ++e) {
    int t = fill[ia[e]]++; slot_row[t] = ia[e]; slot_col[t] = ib[e]; slot_side[t] = SIDE_BEGIN; slot_edge[t] = e;
    t = fill[ib[e]]++; slot_row[t] = ib[e]; slot_col[t] = ia[e]; slot_side[t] = SIDE_END; slot_edge[t] = e;
  }
  FrontSymbolic S;
  const auto t0 = std::chrono::steady_clock::now();
  const bool ok = front_analyze(N, ia, ib, n_slots, slot_row, slot_col, slot_side, 200LL << 30, &S, getenv("SMALL_MAX") ? atoi(getenv("SMALL_MAX")) : 0);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("N %d E %d ok %d seconds %.3f fronts %d levels %d launches %d jobs %zu largest %d flops %.3e MB %.1f blocks %lld\n", N, E, ok ? 1 : 0, dt,
              S.nf, S.n_levels, S.n_launches, S.jobs.size(), S.max_front, S.flops, 8e-6 * (double)S.fval_size, S.factor_blocks);
  if (!ok) return 1;
  // children before parents in the front numbering: the ticket order of the single-launch small-front factorisation (a front
  // waits only for fronts with smaller numbers) and of its backward substitution (the root first) rests on it
  for (int f = 0; f < S.nf; ++f) {
    for (int ci = S.fronts[f].child_begin; ci < S.fronts[f].child_end; ++ci)
      if (S.child[ci] >= f || S.fronts[S.child[ci]].parent != f) { std::printf("front %d: child %d out of order\n", f, S.child[ci]); return 1; }
    if (S.fronts[f].parent >= 0 && S.fronts[f].parent <= f) { std::printf("front %d: parent %d out of order\n", f, S.fronts[f].parent); return 1; }
  }
  std::printf("front numbering: children before parents ok\n");
  if (S.small) return 0;      // (the small-front kernels are checked on the GPU: this tool emulates the round schedule)
  // the stages of the single-launch form (FrontStages): the ticket order must be a topological order — every work-group of every
  // stage a stage waits for holds an EARLIER ticket (that is what makes the in-kernel waits deadlock-free whatever is resident),
  // every work-group of the launch schedule appears exactly once, and the counts per stage add up
  if (!S.st_table.empty()) {
    const int ns = (int)S.st_need.size(), nt = (int)(S.st_table.size() / 2);
    std::vector<int> first(ns, 1 << 30), last(ns, -1), seen(ns, 0);
    for (int t = 0; t < nt; ++t) {
      const int sg = S.st_table[2 * (size_t)t + 1];
      if (sg < 0 || sg >= ns) { std::printf("stage plan: ticket %d has stage %d of %d\n", t, sg, ns); return 1; }
      first[sg] = std::min(first[sg], t); last[sg] = std::max(last[sg], t); ++seen[sg];
    }
    long long total = 0;
    for (const FrontLaunch& La : S.launches) total += La.n_wg;
    bool good = total == nt;
    for (int sg = 0; sg < ns && good; ++sg) {
      good = seen[sg] == S.st_need[sg] && seen[sg] > 0;
      for (int q = S.st_pred_ptr[sg]; q < S.st_pred_ptr[sg + 1] && good; ++q) good = last[S.st_pred[q]] < first[sg];
    }
    std::printf("stage plan: %d tickets, %d stages, %s\n", nt, ns, good ? "topological order ok" : "BROKEN");
    if (!good) return 1;
  }
  if (argc > 2) {
    int np = 0, ng = 0, na = 0;
    for (const FrontLaunch& La : S.launches) { np += La.type == FrontLaunch::PANEL; ng += La.type == FrontLaunch::GEMM; na += La.type == FrontLaunch::ASM; }
    std::printf("factor launches: %d panel, %d gemm, %d extend-add; backward launches: %zu\n", np, ng, na, S.bwd_launches.size());
    if (argv[2][0] == 'd') {   // per-front dump: id c r parent
      for (int f = 0; f < S.nf; ++f) std::printf("F %d %d %d %d\n", f, S.fronts[f].c, S.fronts[f].r, S.fronts[f].parent);
      return 0;
    }
    if (argv[2][0] == 'g') {   // per GEMM launch: work-groups, tile, flops on the tiles, longest K, jobs
      for (const FrontLaunch& La : S.launches) {
        if (La.type != FrontLaunch::GEMM) continue;
        double fl = 0; int kmax = 0, kmin = 1 << 30, nj = 0, prev = -1;
        for (int w = La.wg_begin; w < La.wg_begin + La.n_wg; ++w) {
          const FrontJob& J = S.jobs[S.wg_job[w]];
          fl += 2.0 * La.tile * La.tile * J.klen;
          kmax = std::max(kmax, J.klen); kmin = std::min(kmin, J.klen);
          if (S.wg_job[w] != prev) { prev = S.wg_job[w]; ++nj; }
        }
        std::printf("G %d %d %.4g %d %d %d\n", La.n_wg, La.tile, fl, kmin, kmax, nj);
      }
      return 0;
    }
    if (argv[2][0] == 's') return 0;
  }
  // random SPD matrix in slot form: off-diagonal blocks random, diagonal = strictly dominant
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> eb((size_t)36 * E), slot_val((size_t)36 * n_slots, 0.0), b(6 * (size_t)N);
  for (double& v : eb) v = U(rng);
  for (double& v : b) v = U(rng);
  std::vector<double> rowsum(6 * (size_t)N, 0.0);
  for (int t = 0; t < n_slots; ++t) {
    if (slot_side[t] == SIDE_DIAG) continue;
    const double* B = &eb[(size_t)36 * slot_edge[t]];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) {
        const double v = slot_side[t] == SIDE_BEGIN ? B[6 * r + c] : B[6 * c + r];
        slot_val[(size_t)36 * t + 6 * r + c] = v;
        rowsum[6 * (size_t)slot_row[t] + r] += std::fabs(v);
      }
  }
  for (int v = 0; v < N; ++v) {
    double* D = &slot_val[(size_t)36 * row_slot_begin[v]];
    for (int r = 0; r < 6; ++r) for (int c = 0; c <= r; ++c) { const double x = 0.1 * U(rng); D[6 * r + c] += x; if (c != r) D[6 * c + r] += x; }
    for (int r = 0; r < 6; ++r) D[7 * r] += 1.0 + rowsum[6 * (size_t)v + r];
  }
  // ---- the device algorithm, in scalar loops ----
  std::vector<double> F((size_t)S.fval_size, 0.0), W((size_t)S.winv_size, 0.0), x(6 * (size_t)N, 0.0);
  // scatter: BSR blocks and the right-hand side row
  for (size_t a = 0; a + 1 < S.ablk_ptr.size(); ++a) {
    const FrontDesc& D = S.fronts[S.ablk_front[a]];
    const int bi = S.ablk_pos[a] >> 16, bj = S.ablk_pos[a] & 0xffff;
    for (int q = S.ablk_ptr[a]; q < S.ablk_ptr[a + 1]; ++q)
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c)
        F[D.fbase + (size_t)(6 * bi + r) * D.ld + 6 * bj + c] += slot_val[(size_t)36 * S.ablk_slot[q] + 6 * r + c];
  }
  for (int j = 0; j < N; ++j) {
    const FrontDesc& D = S.fronts[S.col_front[j]];
    const int n = 6 * (D.c + D.r);
    for (int k = 0; k < 6; ++k) F[D.fbase + (size_t)n * D.ld + 6 * (j - D.first) + k] = b[6 * (size_t)S.perm[j] + k];
  }
  bool bad_pivot = false;
  long long n_updated = 0;
  {
    for (size_t li = 0; li < S.launches.size(); ++li) {
      const FrontLaunch& La = S.launches[li];
      if (La.type == FrontLaunch::ASM) {
        // extend-add (children in list order) of every parent front named by the launch's tile records
        int prevq = -1;
        for (int w = La.wg_begin; w < La.wg_begin + La.n_wg; ++w) {
          const int q = S.asm_tile[8 * (size_t)w];
          if (q == prevq) continue;
          prevq = q;
          const FrontDesc& P = S.fronts[q];
          const int np = 6 * (P.c + P.r);
          for (int ci = P.child_begin; ci < P.child_end; ++ci) {
            const FrontDesc& C = S.fronts[S.child[ci]];
            const int nc = 6 * (C.c + C.r);
            const int* rel = &S.rel[C.rel_begin];
            for (int k = 0; k <= C.r; ++k) {          // k == C.r: the right-hand side row
              const int prow = k < C.r ? 6 * rel[k] : np;
              const int crow = k < C.r ? 6 * (C.c + k) : nc;
              const int nr = k < C.r ? 6 : 1;
              for (int m = 0; m <= std::min(k, C.r - 1); ++m)
                for (int a = 0; a < nr; ++a) for (int bb = 0; bb < 6; ++bb)
                  F[P.fbase + (size_t)(prow + a) * P.ld + 6 * rel[m] + bb] += F[C.fbase + (size_t)(crow + a) * C.ld + 6 * (C.c + m) + bb];
            }
          }
        }
        continue;
      }

      // the jobs of a launch = the distinct entries of its workgroup -> job map (consecutive); a GEMM launch is executed
      // work-group by work-group, tile by tile (its tile list is part of what is checked)
      int prev = -1;
      for (int w = La.wg_begin; w < La.wg_begin + La.n_wg; ++w) {
        const int ji = S.wg_job[w];
        const FrontJob& J = S.jobs[ji];
        double* A = &F[J.fbase];
        if (J.c1 > 0) {
          const int T = La.tile, ti = S.wg_tile[w] >> 16, tj = S.wg_tile[w] & 0xffff;
          for (int i = J.r0 + T * ti; i < std::min(J.r0 + T * (ti + 1), J.r1); ++i)
            for (int c = J.c0 + T * tj; c < std::min(std::min(J.c0 + T * (tj + 1), J.c1), i + 1); ++c) {
              double s = 0.0;
              for (int m = 0; m < J.klen; ++m) s += A[(size_t)i * J.ld + J.k0 + m] * A[(size_t)c * J.ld + J.k0 + m];
              A[(size_t)i * J.ld + c] -= s;
              ++n_updated;
            }
          continue;
        }
        if (ji == prev) continue;
        prev = ji;
        if (La.type == FrontLaunch::PANEL) {
          const int nb = J.klen, k0 = J.k0;
          double Dk[FRONT_NB][FRONT_NB], Lk[FRONT_NB][FRONT_NB];
          double* Wp = &W[J.wbase];
          for (int i = 0; i < nb; ++i)
            for (int j = 0; j <= i; ++j) {
              double s = A[(size_t)(k0 + i) * J.ld + k0 + j];
              for (int m = J.c0; m < k0; ++m) s -= A[(size_t)(k0 + i) * J.ld + m] * A[(size_t)(k0 + j) * J.ld + m];
              Dk[i][j] = s;
            }
          for (int k = 0; k < nb; ++k) {
            double d = Dk[k][k];
            for (int m = 0; m < k; ++m) d -= Lk[k][m] * Lk[k][m];
            if (!(d > 0.0)) bad_pivot = true;
            d = std::sqrt(d);
            Lk[k][k] = d;
            for (int i = k + 1; i < nb; ++i) {
              double s = Dk[i][k];
              for (int m = 0; m < k; ++m) s -= Lk[i][m] * Lk[k][m];
              Lk[i][k] = s / d;
            }
          }
          for (int j = 0; j < FRONT_NB * FRONT_NB; ++j) Wp[j] = 0.0;
          for (int j = 0; j < nb; ++j)
            for (int i = j; i < nb; ++i) {
              double s = i == j ? 1.0 : 0.0;
              for (int m = j; m < i; ++m) s -= Lk[i][m] * Wp[m * FRONT_NB + j];
              Wp[i * FRONT_NB + j] = s / Lk[i][i];
            }
          double pr[FRONT_NB];
          for (int i = J.r0; i < J.r1; ++i) {
            for (int c = 0; c < nb; ++c) {
              double s = A[(size_t)i * J.ld + k0 + c];
              for (int m = J.c0; m < k0; ++m) s -= A[(size_t)i * J.ld + m] * A[(size_t)(k0 + c) * J.ld + m];
              pr[c] = s;
            }
            for (int c = 0; c < nb; ++c) {
              double s = 0.0;
              for (int m = 0; m <= c; ++m) s += pr[m] * Wp[c * FRONT_NB + m];
              A[(size_t)i * J.ld + k0 + c] = s;
            }
          }
        }
      }
    }
  }
  // backward substitution, top level first
  // (order of the device schedule: bwd_launches; numerically any parent-before-child order gives the same x)
  {
    std::vector<int> order_b;
    std::vector<char> seen(S.nf, 0);
    for (const FrontBwdLaunch& La : S.bwd_launches)
      if (La.kind == 0) for (int w = La.wg_begin; w < La.wg_begin + La.n_wg; ++w) if (!seen[S.bwd_front[w]]) { seen[S.bwd_front[w]] = 1; order_b.push_back(S.bwd_front[w]); }
    if ((int)order_b.size() != S.nf) { std::printf("backward schedule covers %zu of %d fronts\n", order_b.size(), S.nf); return 1; }
    for (int q : order_b) {
      if (S.fronts[q].parent >= 0 && !seen[S.fronts[q].parent]) { std::printf("child before parent\n"); return 1; }
      const FrontDesc& D = S.fronts[q];
      const int c6 = 6 * D.c, n = 6 * (D.c + D.r);
      const double* A = &F[D.fbase];
      std::vector<double> t(c6);
      for (int j = 0; j < c6; ++j) {
        double s = A[(size_t)n * D.ld + j];
        for (int i = c6; i < n; ++i) s -= A[(size_t)i * D.ld + j] * x[6 * (size_t)S.idx[D.idx_begin + (i - c6) / 6] + (i - c6) % 6];
        t[j] = s;
      }
      const int npanels = (c6 + FRONT_NB - 1) / FRONT_NB;
      for (int p = npanels - 1; p >= 0; --p) {
        const int k0 = p * FRONT_NB, nb = std::min<int>(FRONT_NB, c6 - k0);
        const double* Wp = &W[D.wbase + (size_t)p * FRONT_NB * FRONT_NB];
        double xs[FRONT_NB];
        for (int a = 0; a < nb; ++a) {
          double s = 0.0;
          for (int bq = a; bq < nb; ++bq) s += Wp[bq * FRONT_NB + a] * t[k0 + bq];
          xs[a] = s;
        }
        for (int a = 0; a < nb; ++a) x[6 * (size_t)D.first + k0 + a] = xs[a];
        for (int j = 0; j < k0; ++j) {
          double s = 0.0;
          for (int a = 0; a < nb; ++a) s += A[(size_t)(k0 + a) * D.ld + j] * xs[a];
          t[j] -= s;
        }
      }
    }
  }
  // residual in the original numbering
  std::vector<double> xo(6 * (size_t)N), res(b);
  for (int j = 0; j < N; ++j) for (int k = 0; k < 6; ++k) xo[6 * (size_t)S.perm[j] + k] = x[6 * (size_t)j + k];
  for (int t = 0; t < n_slots; ++t)
    for (int r = 0; r < 6; ++r) {
      double s = 0.0;
      for (int c = 0; c < 6; ++c) s += slot_val[(size_t)36 * t + 6 * r + c] * xo[6 * (size_t)slot_col[t] + c];
      res[6 * (size_t)slot_row[t] + r] -= s;
    }
  double rn = 0.0, bn = 0.0;
  for (size_t i = 0; i < res.size(); ++i) { rn += res[i] * res[i]; bn += b[i] * b[i]; }
  std::printf("updated entries %lld\n", n_updated);
  std::printf("bad_pivot %d relative residual %.3e\n", bad_pivot ? 1 : 0, std::sqrt(rn / bn));
  return std::sqrt(rn / bn) < 1e-10 && !bad_pivot ? 0 : 1;
}
