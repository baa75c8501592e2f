// pgo_sym_host.h — the host-only part of the symmetric tile form (pgo_sym.h): row partition and slot layout from plain arrays.  No device,
// no pgo_problem: pgo_sym.cpp uploads what it returns, tools/sym_check_cli.cpp emulates the kernels on it (tests/test_sym_host.py, CPU).
#pragma once
#include <vector>

#include "pgo_sym.h"

namespace pgo {

struct SymHostParams {
  int row_cap = 256;          // poses per tile (<= SYM_LANES)
  long long w_cap = 1 << 30;  // incidences (1 + degree, summed over the rows) per tile
  bool sort_tiles = true;     // largest tiles first
  // r06 — the form of ONE RANK's rows (several ranks: every rank builds the form of the rows it owns; the far end of an edge that leaves
  // the range is a ghost column like the far end of any cut edge, an edge with neither end in the range is not the rank's business) and
  // tiles made of whole preconditioner clusters (the one-launch CG iteration on the form, k_pipe_cg_sym, applies a row's Jacobi block
  // inside the tile that owns the row: poses unit*c .. unit*c + unit - 1 move together and sit in consecutive lanes)
  int row_lo = 0, row_hi = -1;   // owned rows [row_lo, row_hi); row_hi < 0: all of them
  int unit = 1;                  // poses per indivisible group (1, 2 or 4; row_lo is a multiple of it)
};
struct SymHostLayout {
  std::vector<SymTile> tiles;
  std::vector<int> xlist, chunk_base, chunk_n, src_slot, diag_slot;
  std::vector<uint32_t> meta, rinfo;
  int n_slots = 0, x_cap = 0;
  long long interior_edges = 0, stored = 0;
  const char* unfit = nullptr;    // why the graph does not fit the form (the caller keeps the incidence-slot kernels)
  int unfit_tile = -1;
  double ms_partition = 0, ms_layout = 0;
};
// row_slot_begin[v]: the incidence-slot BSR's diagonal slot of pose v (its incidences follow in edge order) — where src_slot points
void sym_build_host(int N, int E, const int* ia, const int* ib, const int* row_slot_begin, const SymHostParams& prm, SymHostLayout* out);

}  // namespace pgo
