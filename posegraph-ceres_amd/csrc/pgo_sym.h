// pgo_sym.h — "store every off-diagonal block once": the symmetric tile form of the block-sparse normal equations
// (SURVEY.md §8d accounts K3 as (N + E) * 288 B per product: every off-diagonal 6x6 block counted ONCE; the incidence-slot BSR
// of pgo_kernels.h stores and reads H_ab and H_ba = H_ab^T, 1.8x that traffic at 100 k poses / 1 M edges).
//
// Rows are partitioned into TILES of up to 256 poses with graph locality (natural order + greedy refinement, pgo_sym.cpp), one
// work-group of 256 lanes per tile:
//   * an edge whose two poses lie in the same tile (INTERIOR, 77 % of the edges of BASELINE config 4) keeps ONE block, H_ab in
//     the begin-side orientation; the lane that loads it multiplies  u = H_ab x_b  for row a and  v = H_ab^T x_a  for row b;
//   * a CUT edge keeps its two blocks, one in each tile (exactly the incidence-slot form);
//   * x of the tile's rows and of its GHOST columns (the far ends of its cut edges) is staged in LDS once per product.
// The stored slots of a tile are laid out row after row and processed in CHUNKS of 256 consecutive slots, one slot per lane (every
// lane of a chunk but the last loads a block: the stream is as dense as the incidence-slot kernel's).  Products go to LDS — u at
// the lane's own position, v at a position the producer knows, so that the entries of one destination row are contiguous — and
// lane r, which keeps the sum of the tile's r-th row in registers across the chunks, adds its two ranges in a fixed order: bitwise
// reproducible, no FP64 atomics, no index lookups inside the loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgo_kernels.h"

namespace pgo {

enum { SYM_LANES = 256, SYM_X_MAX = 4095 };

struct SymTile {
  int chunk0;     // first entry of this tile in the chunk tables
  int nchunks;
  int x0;         // first entry of this tile in xlist
  int nx;         // rows + ghosts staged in LDS
  int nrows;      // the first nrows entries of the tile's xlist are its own rows (row r <-> lane r)
  int base0, n0;  // chunk_base / chunk_n of the tile's first two chunks: their block loads start as soon as the descriptor is in
  int base1, n1;
  int total;      // stored slots of the tile: chunk c holds min(256, total - 256 c) of them, starting at base0 + 256 c
  int pad[2];
};

// per stored slot (index = chunk_base[chunk] + lane):
//   meta = xcol | side << 12 | interior << 14 | vpos << 15 | xrow << 23
//   xcol: LDS index of the column's x (rows first, then ghosts); xrow: LDS index (= local row) of the slot's own row;
//   vpos: where an interior slot's v goes in the chunk's v buffer
// per (chunk, lane r):  rinfo = ub | uc << 8 | vb << 17 | vc << 25 — the u entries [ub, ub + uc) and the v entries [vb, vb + vc) of
//   the chunk that belong to the tile's r-th row (uc <= 256, vc <= 127: a graph that needs more keeps the incidence-slot kernels)
struct SymGraph {
  int n_tiles, n_chunks, n_slots;     // n_slots: stored slots incl. alignment padding (multiple of 64)
  int x_cap;                          // max over tiles of rows + ghosts (LDS sizing)
  const SymTile* tile;                // [n_tiles]
  const int* xlist;                   // pose id per staged x entry
  const int* chunk_base;              // [n_chunks] first stored slot of the chunk (multiple of 64)
  const int* chunk_n;                 // [n_chunks] active lanes
  const uint32_t* meta;               // [n_slots]
  const uint32_t* rinfo;              // [n_chunks * SYM_LANES]
  const int* src_slot;                // [n_slots] slot of the incidence-slot BSR that holds the same block (-1: padding)
  const int* diag_slot;               // [N] stored slot of every row's diagonal block
  double* val;                        // [n_slots * 36] blocks, bsr_index() layout
  // several ranks, boundary exchange (DeviceGraph::bx): xoff[i], parallel to xlist — where k_pipe_cg_sym finds m of that staged column:
  // >= 0 a double offset into pipe_buf (a row of this rank), < 0: -1 - (double offset into bx) (another rank's boundary row)
  const int* xoff;
  const int* xbidx;      // parallel to xlist, for a tile's own rows: the row's index inside this rank's boundary segment (-1: not a boundary row)
};

// q = A p of a CG iteration (MODE 0: prologue, x = z + beta p, p_new, p'q partials — exactly k_spmv<0>'s contract) or plain
// q = A cg_x (MODE 1) from the symmetric tile form
void launch_spmv_sym(const DeviceGraph& g, const SymGraph& sg, const CgParams& p, int odd, int mode, hipStream_t s);
// copies the blocks of the incidence-slot BSR (g.bsr_val) into the symmetric tile form (after a linearisation / damping)
// diag_only: a rejected LM step changed nothing but the damping, i.e. the diagonal slots
void launch_sym_repack(const DeviceGraph& g, const SymGraph& sg, hipStream_t s, int diag_only = 0);
size_t sym_lds_bytes(const SymGraph& sg);
// r06: one iteration of the owner-only pipelined CG (k_pipe_cg's contract: launch `seq` reads pipe_buf[seq & 1], seq 0 = w0 = A u0) with
// the product from the symmetric tile form, followed by k_pipe_fold; the tiles must be made of whole preconditioner clusters
// fold (one rank only; several ranks always fold): also run k_pipe_fold behind the launch — wanted in front of a "stop test only" launch
void launch_pipe_cg_sym(const DeviceGraph& g, const SymGraph& sg, const CgParams& p, int seq, hipStream_t s, unsigned long long gseq = 0, bool fold = false);
// pgo_lean_kernels.hip: the row kernel writing the form (g.sym_dst / g.sym_val set) with the lean per-incidence algebra of pgo_lin_lean.h
void launch_linearize_lean(const DeviceGraph& g, hipStream_t s, int gate = 0);      // (falls back to launch_linearize_symout when it does not fit)
bool linearize_lean_fits(const DeviceGraph& g);

}  // namespace pgo
