// pgo_front.h — supernodal MULTIFRONTAL Cholesky of (H~ + D^2) on the GPU with FP64 MFMA dense fronts.
//
// Role in the reference: the numeric factorisation CHOLMOD (supernodal) performs behind
// `options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY` (finial.cpp:536).  pgo_direct.* serves the chain-like
// graphs (KITTI-00: a trajectory plus a few hundred chords) with enumerated 6x6 block pairs; mesh-like graphs
// (Manhattan 10 k, sphere x10) have separators of hundreds of poses, and this solver treats them as dense algebra:
//
//   host, once per topology  : nested-dissection ordering (shared with pgo_direct), block symbolic factorisation,
//                              fundamental supernodes, relaxed amalgamation (flops are cheap here, launches and
//                              dependent steps are not), postorder renumbering, one dense FRONT per supernode
//                              [F11 . ; F21 F22] over its c own poses + r update poses, tree levels, launch schedule;
//   device, every LM iteration: scatter the BSR blocks + the right-hand side into the fronts, then per tree level
//                              (leaves first): extend-add of the children's update matrices (gather per parent tile,
//                              fixed child order: deterministic, no atomics), blocked factorisation of the c own
//                              columns with ONE launch per 48-wide panel (left-looking inside 192-wide outer panels:
//                              every row-tile workgroup forms and factorises the 48 x 48 diagonal block itself — one
//                              wave, registers — and does its own TRSM; all sums and the TRSM are
//                              v_mfma_f64_16x16x4_f64), one right-looking GEMM launch per outer panel (K = 192) and the
//                              Schur update U = F22 - L21 L21^T with K = 6c.
//                              The right-hand side rides along as ONE EXTRA ROW of every front (Cholesky of the
//                              bordered matrix [[A b],[b' .]] leaves y = L^-1 b in that row), so the forward
//                              substitution costs nothing of its own; the backward substitution walks the tree
//                              top-down, one workgroup per front.
//
// All fronts are row-major, leading dimension = front size + 2 (the RHS row is row n of an (n+1) x n matrix; ld is even so
// every row starts 16-byte aligned).  Only the lower triangle is meaningful.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "pgo_kernels.h"

namespace pgo {

enum { FRONT_NB = 48, FRONT_NBO = 192, FRONT_TILE = 64, FRONT_ASM_TP = 8 };

// One unit of dense work on a front.
//   PANEL: columns [k0, k0 + klen) (klen <= 48), left-looking inside the 192-wide outer panel that starts at column c0:
//          every workgroup (one per 64-row tile of the rows [r0, r1) below the diagonal block) forms
//          D = F[kb, kb] - sum_{m in [c0, k0)} L[kb, m] L[kb, m]^T, factorises it (one wave, registers) and inverts the
//          factor (W = L_kk^-1, also stored at Winv + wbase for the backward substitution), then computes ITS rows
//          L[i, kb] = (F[i, kb] - sum_m L[i, m] L[kb, m]^T) W^T on the matrix cores.  The diagonal block itself is
//          never written back (nothing reads L_kk: the triangular solves use W).
//   GEMM : C[r0:r1, c0:c1] -= F[r0:r1, k0:k0+klen) * F[c0:c1, k0:k0+klen)^T (right-looking update of everything
//          behind an outer panel, K = 192; Schur update U = F22 - L21 L21^T, K = 6c), lower tiles only.
struct FrontJob {
  long long fbase;   // offset of the front in Fval (doubles)
  int ld;
  int r0, r1, c0, c1;
  int k0, klen;
  int wbase;         // PANEL: offset of the panel's W in Winv (doubles)
};

struct FrontLaunch {
  enum Type { PANEL = 0, ASM = 1, GEMM = 2 };
  int type, n_wg;
  int tile;          // GEMM: rows = columns of a workgroup's tile, 64 or 32 (launches with few tiles)
  int wg_begin;      // PANEL / GEMM: workgroup w runs job wg_job[wg_begin + w] on tile wg_tile[wg_begin + w] = (ti << 16) | tj;
                     // ASM: workgroup w handles the extend-add tile record wg_begin + w of asm_tile
};

// backward substitution: phase A (kind 0: workgroup w handles columns 64 * bwd_chunk[wg_begin + w] of front bwd_front[.]) or
// one block step (kind 1: bwdb_front / bwdb_chunk)
struct FrontBwdLaunch {
  int kind, wg_begin, n_wg, lds_bytes;
};

struct FrontLevel {
  int front_begin, front_end;     // fronts are numbered level by level (statistics; the schedule is not per level)
};

// Per-front descriptor on the device.
struct FrontDesc {
  long long fbase;
  int first;        // first own column (new numbering); own columns are [first, first + c)
  int c, r;         // own poses, update poses
  int ld;
  int idx_begin;    // update rows: idx[idx_begin .. idx_begin + r)   (new numbering, ascending)
  int child_begin, child_end;   // children: child[child_begin .. child_end)
  int rel_begin;    // as a child: rel[rel_begin .. rel_begin + r) = position (pose units) of each update row in the parent's front
  int cs_begin;     // as a child: cstart[cs_begin + t] = first update row whose parent position is >= 8 t, t = 0 .. parent ntp
  int wbase;        // W of panel p at Winv + wbase + p * FRONT_NB * FRONT_NB
  int asm_wg_begin; // (unused)
  int ntp;          // pose tiles of FRONT_ASM_TP per side
  int parent;
  int pad;
};

// ---- small fronts (chain-like graphs: every front of KITTI-00 has at most 84 scalars) -------------------------------------
// When no front exceeds SFRONT_MAX scalars the whole front lives in the LDS of one workgroup, and a tree level is ONE launch:
// zero + original entries + right-hand side, extend-add of the children's update matrices, blocked (6 columns = one pose)
// right-looking Cholesky with the 6 x 6 pivot block factorised redundantly in registers, then the L panel and the update
// matrix go to compact global arrays.  The backward substitution is one launch per level as well.  No 48-column panels, no
// GEMM launches, no round schedule.
enum { SFRONT_MAX = 96 };
struct SFront {
  int lbase;        // L panel: (n + 1) x c6 row-major (row n = y) at Lval + lbase
  int ubase, ucnt;  // update matrix, PACKED: the lower triangle of the r6 x r6 block row by row (row i at i (i + 1) / 2), then the
                    // right-hand side row (r6 entries) at r6 (r6 + 1) / 2
  int urel;         // urel[urel + j], j < r6: the parent's column that receives column j of the update matrix
  int to_fval;      // mixed plan: the parent is a regular front — the update matrix goes to Fval in the regular front layout
  int ablk_begin, ablk_end;   // the front's range of ablk_* entries
  int wbase;        // W = L11^-1, c6 x c6 row-major lower triangular at Wval + wbase (backward substitution = two gemv)
};
struct SFrontPlan {
  const SFront* sf;   // [nf]
  const int* list;    // mixed plan: the small fronts level by level (a launch covers list[front_begin .. )); null: all fronts are small
  const int* urel;    // see SFront::urel
  int* upos;          // [su_size] LDS offset in the PARENT's front of every packed update entry (filled once on the device)
  const int* osrc;    // [n_ablk] the one BSR slot of an original block | side << 28, or -1 (several slots: ablk_ptr / ablk_slot)
  double* Lval;
  double* Uval;
  double* Wval;
  int* done;          // [2 (nf + 1)] single-launch form: done[f] = epoch once front f has published its update matrix, done[nf]: the ticket
                      // counter; the second half: the same for the backward substitution (x of front f published)
};
// The single-launch form of the small-front factorisation (all levels in one launch): workgroups take fronts in ticket order
// (children have smaller numbers than their parents, so whatever a workgroup waits for is held by a workgroup that started
// before it: no residency assumption), a parent polls its children's flags.  epoch: this factorisation's flag value;
// ticket_base: tickets handed out by the launches before this one; max_spins: a wait that runs out raises flags[2] |= 2 and the
// host repeats the factorisation level by level (and keeps to that).  done == nullptr: one launch per level, no waiting.
struct SFrontSync {
  int* done;
  unsigned ticket_base;
  int epoch;
  int max_spins;
  long long* stamps;   // development aid (PGO_SF_STAMPS=1): six s_memrealtime stamps per front, null otherwise
};

struct FrontPlan {
  int n, nf;
  const int* perm;          // [n] new -> old
  const FrontDesc* fronts;  // [nf]
  const int* idx;
  const int* child;
  const int* rel;
  const int* cstart;
  const int* col_front;     // [n] front owning each column (new numbering)
  const int* ablk_ptr;      // [n_ablk+1] BSR slots summed into one 6x6 block of a front
  const int* ablk_slot;
  const int* ablk_front;    // [n_ablk]
  const int* ablk_pos;      // [n_ablk] (bi << 16) | bj, pose units inside the front
  int n_ablk;
  const FrontJob* jobs;
  const int* wg_job;
  const int* wg_tile;
  const int* asm_tile;      // extend-add workgroups, 8 ints each: parent front, (ti << 16) | tj, first / end contributor, parent fbase lo / hi,
                            // parent ld, parent size (scalars) | ntp << 20
  const int* asm_contrib;   // contributors, 8 ints each: child fbase lo / hi, ld, c, (ks << 16) | ke, (ms << 16) | me (child update rows),
                            // rel_begin, r
  const int* bwd_front;
  const int* bwd_chunk;
  const int* bwdb_front;    // phase B workgroups: front ...
  const int* bwdb_chunk;    // ... and (source block << 16) | target chunk, FRONT_NBO columns per chunk; source == target: solve only
  double* Fval;
  double* Winv;
  double* x;                // [6n] solution, new numbering (the backward substitution keeps t = y - L21^T x_r here in between)
  // single-launch form of the factorisation (FrontStages below): null when not in use
  const int* st_table;      // [n_tickets][2]: (kind | index << 2), stage — kind 0 extend-add record, 1 panel, 2 GEMM 64, 3 GEMM 32 work-group
  const int* st_pred_ptr;   // [n_stages + 1] the stages a stage waits for ...
  const int* st_pred;       // ... (one: the front's previous stage; an extend-add: the last stage of every child)
  const int* st_need;       // [n_stages] work-groups of the stage
  unsigned long long* st_count;   // [n_stages + 1] finished work-groups, never reset: stage s of factorisation e is complete at need[s] * e; [n_stages]: tickets
  const int* halt;          // device-resident LM (pgo_kernels.h LmDev::halt) or null: kernels that see only the plan exit while *halt != 0
};
// The single-launch form of the regular multifrontal factorisation: every work-group of every round of the launch schedule in ONE
// grid, taken in ticket order (= the order of the launches, a topological order of the stages), a work-group waits until the
// stages its own stage depends on are complete (counters, agent scope; one wave acquires / releases, as in the small-front plan).
// A front then advances at the pace of its own chain extend-add -> panel -> panel ... -> GEMM, not at the pace of the rounds,
// in which every launch waits for the slowest front of the one before and a round costs the SUM of its three launches.
struct FrontStages {
  unsigned long long epoch;       // this factorisation's number (1, 2, ...)
  unsigned long long ticket_base; // tickets handed out before this launch
  int n_tickets;
  int n_stages;
  int max_spins;
  long long* stamps;    // development aid (PGO_FRONT_STAMPS=1): three s_memrealtime stamps per ticket, null otherwise
};

struct FrontSymbolic {
  int n = 0, nf = 0, n_levels = 0;
  std::vector<int> perm, iperm;
  std::vector<FrontDesc> fronts;
  std::vector<int> idx, child, rel, cstart, col_front, wg_job, wg_tile, asm_tile, asm_contrib, bwd_front, bwd_chunk, bwdb_front, bwdb_chunk;
  std::vector<int> ablk_ptr, ablk_slot, ablk_front, ablk_pos;
  std::vector<FrontJob> jobs;
  std::vector<int> job_front;       // front of every job
  std::vector<int> st_table, st_pred_ptr, st_pred, st_need;   // FrontPlan::st_* (single-launch form)
  std::vector<FrontLaunch> launches;
  std::vector<FrontBwdLaunch> bwd_launches;
  std::vector<FrontLevel> levels;
  long long fval_size = 0;   // doubles
  long long winv_size = 0;
  long long factor_blocks = 0;   // 6x6 blocks of L held by the fronts (incl. the amalgamation's explicit zeros)
  double flops = 0;          // factorisation flops of the dense fronts
  double est_us = 0;         // rough time estimate of factor + solve (launch floor + flops), microseconds
  int max_front = 0;         // largest front dimension (scalars)
  int n_launches = 0;
  // small-front path (every front <= SFRONT_MAX scalars): per-front storage offsets, no launch schedule
  bool small = false;
  // mixed plan: the fronts whose whole subtree is small take the small-front kernels (FrontDesc::pad = 1), level by level,
  // before the round schedule of the others starts
  bool mixed = false;
  int n_small = 0;
  std::vector<int> slevel_ptr, slevel_front;
  std::vector<SFront> sfronts;
  std::vector<int> urel, osrc;
  long long sl_size = 0, su_size = 0, sw_size = 0;   // doubles of Lval / Uval / Wval
};

// Host analysis.  Returns false when the fronts would not fit the memory budget (bytes).
// small_max > 0: when no front exceeds that many scalars the analysis stops at the small-front plan (out->small).
bool front_analyze(int N, const std::vector<int>& ia, const std::vector<int>& ib, int n_slots,
                   const std::vector<int>& slot_row, const std::vector<int>& slot_col,
                   const std::vector<uint8_t>& slot_side, long long max_bytes, FrontSymbolic* out, int small_max = 0,
                   const std::vector<uint8_t>* is_point = nullptr);

// Device launches: factorisation (includes the forward substitution) and backward substitution into g.cg_x.
// flags[2] is set when a pivot is not positive.
// sp: the small-front arrays of a mixed plan (sym.mixed), else null
void launch_front_factor(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s, const SFrontPlan* sp = nullptr,
                         const FrontStages* stages = nullptr);
void launch_front_solve(const DeviceGraph& g, const FrontPlan& p, const FrontSymbolic& sym, hipStream_t s, const SFrontPlan* sp = nullptr);
// the small-front path: one launch per tree level each
// once per topology: fills SFrontPlan::upos
void launch_sfront_prepare(const FrontPlan& p, const SFrontPlan& sp, const FrontSymbolic& sym, hipStream_t s);
void launch_sfront_factor(const DeviceGraph& g, const FrontPlan& p, const SFrontPlan& sp, const FrontSymbolic& sym, hipStream_t s,
                          const SFrontSync* fused = nullptr);
void launch_sfront_solve(const DeviceGraph& g, const FrontPlan& p, const SFrontPlan& sp, const FrontSymbolic& sym, hipStream_t s,
                         const SFrontSync* fused = nullptr);

}  // namespace pgo
