// pgo_kernels.h — launch interface between the host LM driver (pgo_lm.cpp, pgo_linear.cpp) and the gfx950
// kernels (pgo_kernels.hip).  Plain structs of device pointers; no torch types.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgo {

// Pose storage: 8 doubles per pose (px py pz qx qy qz qw pad) = one 64-byte line per gather.
enum { POSE_STRIDE = 8 };
// bsr_val tile layout: 64 slots per tile, 36 doubles per slot, element k of slot t at
//   (t>>6)*2304 + (k>>1)*128 + (t&63)*2 + (k&1)          (16 B per lane per access, 1 KiB per wave)
enum { TILE_DOUBLES = 36 * 64 };
__host__ __device__ inline size_t bsr_index(int slot, int k) {
  return (size_t)(slot >> 6) * TILE_DOUBLES + (size_t)(k >> 1) * 128 + (size_t)(slot & 63) * 2 + (k & 1);
}

enum SlotSide : uint8_t { SIDE_BEGIN = 0, SIDE_END = 1, SIDE_DIAG = 2, SIDE_PAD = 3 };

// ---- packed 27-entry slots (DeviceGraph::blk_packed) -------------------------------------------------------------------
// With identity or block-diagonal information (W_pr = 0: the reference's I_6, diag(1/sigma^2)) the off-diagonal block
// H_ab = rho' S_a A^T W B S_b has an exactly zero translation-rotation quadrant: [[-C1, 0], [(RU)^T, -(..)]] — its top-right
// 3x3 for the BEGIN slot, the bottom-left one for the mirrored END slot — and the diagonal block is symmetric.  All three
// kinds of slot then need 27 entries: positions 0..8 = top-left 3x3 (row-major), 9..17 = bottom-right 3x3, 18..26 = the one
// stored off-diagonal quadrant Q (BEGIN and DIAG: bottom-left; END: top-right).  14 pairs per slot are written and read
// instead of 18 (same tile layout and stride: the unused rows of a tile are simply never touched), whatever the side — so
// the SpMV issues its block loads without knowing the side first.  General information keeps the full 36-entry layout
// (position = element).  bsr_pos: position of element k of a slot, or -1 for a structural zero.
enum { BLK_PAIRS_FULL = 18, BLK_PAIRS_PACKED = 14 };
__host__ __device__ inline int bsr_pos(int packed, int side, int k) {
  if (!packed) return k;
  const int r = k / 6, c = k - 6 * r;
  if (r < 3 && c < 3) return 3 * r + c;
  if (r >= 3 && c >= 3) return 9 + 3 * (r - 3) + (c - 3);
  if (r >= 3) return side == SIDE_END ? -1 : 18 + 3 * (r - 3) + c;            // bottom-left
  return side == SIDE_END ? 18 + 3 * r + (c - 3) : side == SIDE_DIAG ? 18 + 3 * (c - 3) + r : -1;   // top-right (DIAG: mirror)
}

// ---- device-resident Levenberg-Marquardt state (r03): the trust-region decisions of SURVEY.md A.6 steps 4-7 are taken ON THE
// DEVICE, by the last work-group of the step tail, so the host can enqueue the kernel sequences of several LM iterations ahead
// and never sits between two of them (r02: every iteration ended in a hand-off to the host, which decided accept / reject and
// enqueued the next one: 13 us on the builder's box, ~60 us on the driver's, of a 300 us iteration).
// A SEQUENCE is what the host enqueues for one prospective LM iteration:
//   damping(+preconditioner) | CG start | nb x (SpMV, update) | tail SpMV | step tail + DECISION | linearise | accept-finish
// Kernels read what the host used to pass by value (trust-region radius, "reuse the clamped diagonal") from LmDev and gate
// themselves on it:  halt != 0 -> every kernel exits at once (terminated, or the host has to step in);  phase CONT -> the CG of
// the previous sequence has not stopped yet: damping / CG start exit, the CG kernels simply go on;  accepted -> the linearisation
// of the candidate (in place: nothing needs the old one any more) and the accept-finish kernel (candidate -> current point,
// gradient norm, gradient-tolerance and minimum-radius tests) run only behind an accepted step.
enum LmHalt { LM_RUN = 0, LM_HALT_TERMINATED = 1, LM_HALT_CG_STALL = 2, LM_HALT_REFACTOR = 3, LM_HALT_BUDGET = 4 };
// ---- the universal stream (PCG on one rank, r03) ---------------------------------------------------------------------------
// How many CG iterations an LM iteration takes is only known when the CG stops, so a host that enqueues kernels ahead of the
// device either allots too many (every unused pair of launches costs ~5 us) or too few.  The universal stream removes the
// guess: the host enqueues ONE alternating pattern  V S V S V S ...  of two kernels, a slot-shaped one (k_uni_s: one lane per BSR
// slot — CG SpMV, tail SpMV, refresh SpMV, linearisation) and a vector-shaped one (k_uni_v: head = accept-finish + damping /
// preconditioner + CG start, CG vector update, step tail + decision), and every launch does whatever comes next, which it reads
// from CgState::op_s / op_v.  An LM iteration with n CG iterations is
//     V head | (S spmv, V update) x n | S tail (the launch that finds the CG stopped multiplies A x at once) | V step tail + decision
//     | S linearise (accepted) or nothing (rejected)
// = 2 n + 4 launches, none wasted but the one slot behind a rejected step, whatever n turns out to be and however slow the host is.
enum UniOp { UNI_EXIT = -1, UNI_NOP = 0, UNI_S_CG = 1, UNI_S_REFRESH = 2, UNI_S_LINEARIZE = 3,
             UNI_V_HEAD = 1, UNI_V_UPDATE = 2, UNI_V_UPDATE_X = 3, UNI_V_UPDATE_R = 4, UNI_V_STEP_TAIL = 5 };
enum LmPhase { LM_PHASE_NEW = 0, LM_PHASE_CONT = 1 };
struct LmRecord {            // layout of pgo_iteration_record (include/pgo.h; static_assert in pgo_internal.h)
  int iteration, step_is_successful, linear_solver_iterations, reserved;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
};
// What the trust-region rules (pgo_lm_rules.h) read and update, and the tolerances they apply.
struct LmCore {
  double radius, decrease_factor, x_cost, x_norm, gmax;
  int iteration;                 // TrustRegionMinimizer's iteration counter = record index of the current point
  int reuse_diagonal;
  int num_consecutive_invalid;
  int pad;
};
struct LmTolerances {
  double min_relative_decrease, function_tolerance, parameter_tolerance, gradient_tolerance, max_radius, min_radius;
  int max_num_iterations, max_consecutive_invalid;
};
struct LmDev {
  int halt;                  // LmHalt — FIRST member: kernels that only see a plan struct gate on `const int* halt`
  int phase;                 // LmPhase
  int accepted;              // outcome of the last decision (1: the linearise / accept-finish kernels of the sequence run)
  int termination, reason;   // pgo_termination / reason code once halt == LM_HALT_TERMINATED
  int lm_done;               // decisions taken since pgo_solver_begin / pgo_solver_reset
  int num_successful, num_unsuccessful, num_linear_iterations, num_records;
  int cg_period;             // residual refresh period of the CG (a CG that outlives its sequence goes on in the next one only
                             // from a multiple of it: the refresh launches sit at fixed positions of every sequence)
  int last_cg;
  int decision_limit;        // universal stream: the stream pauses (LM_HALT_BUDGET) once lm_done reaches it (pgo_solver_step(n))
  int pause;                 //   ... set by the decision that reaches the limit: the head launch behind it only finishes the accepted step
  LmCore core;
  LmTolerances tol;
  double min_diag, max_diag; // LevenbergMarquardtStrategy's clamp of the diagonal
  double term_value;         // the number the termination message quotes (relative step norm, |cost change| / cost, gradient norm, radius)
  long long t_mark, ticks_linear, ticks_jacobian;   // s_memrealtime (100 MHz): end of the previous phase, time spent per phase
};
enum { LM_RING = 64 };       // iteration records in flight between two host polls (the host keeps at most a few sequences ahead)

// Scalars the host reads once per LM iteration (pinned, device-visible).
struct LmScalars {
  double cand_cost;        // 0.5 * sum rho(s) at the candidate point
  double model_change;     // -(J~ step)'(r + J~ step / 2)
  double step_norm_sq;     // |x - x_cand|^2 in the ambient space, non-constant blocks
  double x_norm_sq;        // |x|^2 ambient, non-constant blocks
  double gradient_max;     // |x - Plus(x, -g)|_inf
  double cg_residual_sq;   // |b - A x|^2 from the CG recurrence
  int cg_iterations;
  int cg_status;           // 0 ok, 1 p'q <= 0 (no further progress), 2 non-finite
  int linearize_bad;       // non-finite values seen while linearising
  int seq;                 // hand-off flag: cleared by the host, set last by the publishing kernel (host spins on it)
  // ---- device-resident LM (LmDev above): what the host polls while sequences are in flight ----
  int seq_done;            // id of the last sequence whose final kernel has run
  int lm_done;             // decisions taken
  int halt;                // mirror of LmDev::halt
  int last_cg;             // CG iterations of the last decided iteration (batch-length prediction)
  int slots_done;          // universal stream: vector-shaped launches completed
  int resident_abort;      // resident stream: its grid barrier gave up (pgo_uni_resident.h); the host carries on with the fused stream
  LmDev lm;                // mirror of the device state, complete whenever seq_done == the last enqueued sequence
  LmRecord ring[LM_RING];  // iteration records, slot = record index % LM_RING
};

struct CgState {
  int done;        // set by the first kernel that decides to stop; later kernels exit at once
  int iters;       // CG iterations completed when `done` was set
  int status;      // as LmScalars::cg_status
  int cnt_a;       // iteration index published by the SpMV kernel for the update kernel
  int cnt_b;       // iterations completed, published by the update kernel for the next SpMV
  int op_s;        // universal stream (k_uni_s / k_uni_v below): what the next slot-shaped launch does (UniOp), written by vector-shaped launches
  int op_v;        //   ... and what the next vector-shaped launch does, written by slot-shaped launches (or by the last work-group of a vector-shaped one)
  int slots;       // vector-shaped launches so far (mirrored to LmScalars::slots_done: the host keeps a bounded number enqueued ahead)
  double beta;     // beta of the current iteration, published by the SpMV kernel (the update kernel rebuilds p with it)
  double rho;      // r'z of the current iteration, likewise (the update kernel needs it for alpha = rho / p'q)
  double rho_hist[2];  // r'z and Q of the iterations, by iteration parity: the next SpMV launch reads the other slot
  double q_hist[2];    //   (written during the previous launch), so it re-reduces two partial rows instead of four
  // owner-only CG of the sharded path (k_pipe_cg below): launch `seq` reads pipe[seq & 1] and writes the other slot
  struct Pipe { int cnt, pad; double gamma_prev, alpha_prev, q_prev; } pipe[2];
  // universal stream, fused form (k_uni_f, r05): ONE kernel per launch, the whole stream state double-buffered by launch parity —
  // launch L reads f[L & 1] (written during launch L - 1 by one lane) and one lane writes f[(L & 1) ^ 1]: no word is read and
  // written by the same launch.  op: FusedOp; cnt: CG iterations completed; the scalars of the pipelined recurrences as in Pipe.
  struct Fused { int op, cnt, mirror, pad; double gamma_prev, alpha_prev, q_prev; } f[2];   // mirror: this launch publishes LmDev to the host
};
// launch trace of the one-launch universal streams (DeviceGraph::oplog, pgo_solver_trace_*): words per launch — the kernels
// (pgo_uni_fused.h) and the reader (pgo_capi.cpp) stride by this ONE constant
constexpr int UNI_F_TRACE_WORDS = 66;
// what a launch of the fused universal stream does (CgState::Fused::op)
enum FusedOp { F_EXIT = -1, F_IDLE = 0, F_HEAD = 1, F_W0 = 2, F_CG = 3, F_TAIL = 4, F_LIN = 5 };

struct DeviceGraph {
  int N, E;
  int n_wg;          // workgroups of the row partition
  int n_slots;       // padded (multiple of block)
  int block;         // threads per workgroup of the row-partitioned kernels = slots per chunk
  int info_mode;     // 0 identity information, 1 general W = L^T L, 2 block-diagonal W (W_pr = 0: only W_pp, W_rr are read), 3 diagonal W
  int blk_packed;    // 1: 27-entry slots (info_mode 0 or 2), see bsr_pos
  int loss_kind;
  double loss_a;

  // static topology
  const int* slot_col;        // [n_slots] pose index of the block column (self for the diagonal slot), -1 pad
  const int* slot_row;        // [n_slots] pose index of the row
  const uint8_t* slot_side;   // [n_slots] SlotSide
  const int* wg_slot_begin;   // [n_wg+1]
  const int* wg_row_begin;    // [n_wg+1]
  const int* row_slot_begin;  // [N]
  const int* row_slot_cnt;    // [N]
  const uint8_t* cmask;       // [N] bit0: p constant, bit1: q constant
  // measurements, slot order (component major: [c][n_slots]) and edge order ([c][E])
  const double* smeas;        // 7 x n_slots
  const double* sW;           // 21 x n_slots (upper triangle of L^T L, row-major order) or null
  const int* edge_a;          // [E]
  const int* edge_b;          // [E]
  const double* emeas;        // 7 x E
  const double* eW;           // 21 x E or null
  const double* eL;           // 36 x E (sqrt information, row-major element major) or null

  // state
  double* pose_x;     // [N][8] current iterate
  double* pose_c;     // [N][8] candidate
  double* bsr_val;    // tile layout, n_slots slots
  double* Hdiag;      // [N][36] scaled Gauss-Newton diagonal blocks (no damping)
  double* Minv;       // [N][36] (Hdiag + D^2)^-1
  double* grad;       // [6N] unscaled gradient J'r
  double* scale;      // [6N] Jacobi column scaling
  double* d2;         // [6N] LM diagonal squared
  double* diag_clamped;  // [6N] clamp(diag(H~), min, max) kept while the diagonal is reused
  double* cg_b;       // [6N] rhs = S g
  double* cg_x;       // [6N]
  double* cg_r;       // [6N]
  double* cg_z;       // [6N]
  double* cg_q;       // exchange buffer [world][seg]: q = A p of the owned rows + p'q partials (q_index())
  double* cg_p0;      // [6N] p ping
  double* cg_p1;      // [6N] p pong
  double* delta;      // [6N] tangent step actually applied (S * step)
  // Several ranks, truncated CG (pgo_linear.cpp pipe_*): pipelined preconditioned CG (Ghysels & Vanroose 2014) — u = M^-1 r,
  // w = A u, m = M^-1 w, n = A m and recurrences for z (cg_z), qq, s, p (cg_p0), x, r, u, w — has ONE global reduction per
  // iteration, so every rank updates the vectors of ITS rows only and the single all-gather per iteration carries the m
  // segments plus three sums per rank.  pipe_buf[0 / 1]: exchange buffers [world][pipe_seg] (rows_per * 6
  // doubles of m, then the rank's three sums), read / written by launch parity (x is gathered in place at the end: cg_x is
  // laid out like an exchange buffer of rows_per * 6 doubles per rank).
  double* cg_u; double* cg_w; double* cg_s; double* cg_qq;
  double* pipe_buf[2];
  int pipe_seg;
  // Device-initiated exchange of the owner-only CG (null: the host enqueues an all-gather between the launches).  peer_tab[3 r + k]:
  // rank r's pipe_buf[0], pipe_buf[1], peer_flags.  A producing launch stores its segment into EVERY rank's buffer; its one-work-group
  // tail (k_pipe_fold / k_peer_signal) adds the rank's sums, then — release at system scope — writes the launch's global sequence
  // number into peer_flags[this rank] of every rank and waits until its own flag array shows that number from everybody: the next
  // launch finds every segment in place and nobody still reading the buffer it will overwrite.  No host, no collective launch.
  void* const* peer_tab;
  unsigned long long* peer_flags;     // [world] this rank's flag array (device)
  int n_cu;           // compute units of the device (hipDeviceAttributeMultiprocessorCount: 256 on an MI355X in SPX mode, fewer in a partition): what a grid that has to sit on the chip at once is sized against
  int rows_fit;       // every work-group owns at most block / 6 rows: a row lane per vector component (the resident CG keeps them in registers)
  int pairs_whole;    // the row partition keeps poses 2i, 2i + 1 in one single-chunk work-group (several ranks: prepare() sees to it)
  // partial sums
  double* part_rz;    // [2][n_part]
  double* part_q;     // [2][n_part]
  double* part_rr;    // [2][n_part]  |r|^2
  double* part_bb;    // [n_part]     |b|^2 (written by pcg_init)
  double* part_misc;  // [8][n_part] scratch partials for LM scalars
  double* part_f;     // [2][n_part][4] fused universal stream: per work-group (r,u), (w,u), x'(b + r) by launch parity (k_uni_f)
  int n_part;         // capacity of one partial row (>= max grid of any reducing kernel)
  int n_vec_wg;       // grid of the element-wise kernels (6N lanes)
  int n_edge_wg;      // grid of edge-parallel kernels
  int n_pose_wg;      // grid of pose-parallel kernels
  CgState* cg;        // device
  LmDev* lm;          // device-resident LM state; null: the host decides and passes radius / mode by value (several ranks, batched solve, tests)
  LmScalars* scal;    // device-visible pinned host memory
  int* flags;         // [16] device flags: [0] linearize saw non-finite; [4..12] two-level ticket of the fused universal stream
  int debug;          // timing-ablation bits: read by the kernels only in a -DPGO_ABLATE build (PGO_ABLATION below); 0 otherwise
  // profiling aid (PGO_UNI_OPLOG=<file>, null otherwise): every k_uni_s launch appends (operation it performed, s_memrealtime) so
  // that tools/rocprof_summary.py can bucket the dispatches of that one kernel symbol by what they did.  [0] = entries so far.
  // Fused stream (pgo_solver_trace_*): launch L owns the 66 words at 1 + 66 L: [0] (start tick << 3 | operation) by work-group 0,
  // [2 + s] the latest end tick among the work-groups with index % 64 == s (atomic max).
  long long* oplog;
  int oplog_cap;      // words (pgo_solver_trace_start refuses more launches than fit an int)
  int oplog_indexed;  // 1: the fused stream's per-launch records (pgo_solver_trace_*); 0: appended (tick, operation) entries (PGO_UNI_OPLOG)
  // linearisation into the symmetric tile form by the row kernel (k_linearize_symout): stored slot of every incidence slot (-1: the
  // mirrored incidence of an interior edge, not stored) and the form's block array
  const int* sym_dst;
  double* sym_val;
  // one process per GPU: contiguous row ownership (all rows when world == 1)
  int world, rank, rows_per, row_lo, row_hi, pq_cap, seg;
  int cluster;        // poses per Jacobi block of the preconditioner: 1 (6x6), 2 (12x12) or 4 (24x24)
  const int* cl_ptr;  // [n_clusters+1] BSR slots whose row AND column lie inside the cluster (off-diagonal ones)
  const int* cl_slot;
  const uint8_t* cl_rc;  // per entry: (row - cluster base) << 4 | (col - cluster base)
  // several ranks, host-enqueued exchange of the owner-only CG (r06): only the rank's BOUNDARY rows — rows with an edge to another rank:
  // 5 % of the rows of BASELINE configs[3] on 8 ranks — travel per CG iteration.  bx[0 / 1]: exchange buffers [world][bx_cseg] (a rank's
  // boundary rows in ascending order, 6 doubles each, then its three sums); bx_brow: this rank's boundary rows; bx_slot_off[t]: where
  // k_pipe_cg finds m of slot t's column — >= 0: a double offset into pipe_buf (a row of this rank: the full-layout buffer stays the rank's
  // own), < 0: -1 - (double offset into bx) (another rank's boundary row).  bx[0] == nullptr: whole segments travel in pipe_buf.
  double* bx[2];
  const int* bx_brow;
  const int* bx_slot_off;
  int bx_nb, bx_cseg;
};

// Timing ablations (a kernel with a phase switched off: its results are wrong) exist in builds made with -DPGO_ABLATE only; a
// production build compiles every such test to `false` and carries none of the branches.
#ifdef PGO_ABLATE
#define PGO_ABLATION(g, bits) (((g).debug & (bits)) != 0)
#else
#define PGO_ABLATION(g, bits) false
#endif

// element k (row-major 6x6) of slot `slot`, whatever the layout
__device__ __forceinline__ double bsr_elem(const DeviceGraph& g, int slot, int side, int k) {
  const int p = bsr_pos(g.blk_packed, side, k);
  return p < 0 ? 0.0 : g.bsr_val[bsr_index(slot, p)];
}

// the whole 6 x 6 block of a slot, row-major, with ALL loads issued before the side is looked at (bsr_elem's loads sit behind its
// branches: 36 dependent round trips where this is one — k_coarse_galerkin spent most of its time there)
__device__ __forceinline__ void bsr_block_full(const DeviceGraph& g, int slot, int side, double (&B)[36]) {
  const double2* bp = reinterpret_cast<const double2*>(g.bsr_val + (size_t)(slot >> 6) * TILE_DOUBLES + (size_t)(slot & 63) * 2);
  if (!g.blk_packed) {
#pragma unroll
    for (int k = 0; k < BLK_PAIRS_FULL; ++k) { const double2 v = bp[(size_t)k * 64]; B[2 * k] = v.x; B[2 * k + 1] = v.y; }
    return;
  }
  double el[2 * BLK_PAIRS_PACKED];
#pragma unroll
  for (int k = 0; k < BLK_PAIRS_PACKED; ++k) { const double2 v = bp[(size_t)k * 64]; el[2 * k] = v.x; el[2 * k + 1] = v.y; }
  const bool is_end = side == SIDE_END, is_diag = side == SIDE_DIAG;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      B[6 * r + cc] = el[3 * r + cc];                                            // top-left
      B[6 * (r + 3) + cc + 3] = el[9 + 3 * r + cc];                              // bottom-right
      B[6 * (r + 3) + cc] = is_end ? 0.0 : el[18 + 3 * r + cc];                  // bottom-left (BEGIN, DIAG)
      B[6 * r + cc + 3] = is_end ? el[18 + 3 * r + cc] : is_diag ? el[18 + 3 * cc + r] : 0.0;      // top-right (END; DIAG: the mirror)
    }
}

// ---- batched solve of independent graphs (pgo_solve_batch): the graphs are the components of one block-diagonal problem;
// poses and edges of component c are the contiguous ranges [pose_begin[c], pose_begin[c+1]) / [edge_begin[c], edge_begin[c+1]).
// Every LM scalar exists once per component. ----
struct BatchScalars {
  double cand_cost, model_change, step_norm_sq, x_norm_sq, gradient_max;
  double pad[3];
};
struct BatchPlan {
  int n_comp;
  const int* pose_begin;     // [n_comp + 1] device
  const int* edge_begin;     // [n_comp + 1] device
  const int* pose_comp;      // [N] device
  const double* radius;      // [n_comp] trust-region radius per component (pinned host memory, device visible)
  const int* accept;         // [n_comp] 1: the candidate of this component becomes its current point (pinned)
  BatchScalars* out;         // [n_comp] (pinned)
  double* partial;           // [n_comp][split][5] device scratch of launch_batch_scalars
  int split;                 // workgroups per component
};

struct CgParams {
  double q_tolerance;   // eta
  double r_tolerance;   // |r| <= r_tolerance * |b| stop; negative disables (Ceres LM passes -1)
  int max_iterations;
  int min_iterations;
};

__device__ __forceinline__ bool lm_halted(const DeviceGraph& g) { return g.lm && g.lm->halt != 0; }

// launches (all asynchronous on `s`)
void launch_linearize(const DeviceGraph& g, hipStream_t s, int gate = 0);   // gate 1: run only once the CG has stopped (speculative launch behind a step tail); 2: device-resident LM, behind an accepted step
void launch_linearize_symout(const DeviceGraph& g, hipStream_t s, int gate = 0);   // ... with the off-diagonal blocks written into the symmetric tile form (g.sym_dst / g.sym_val)
void launch_scale_from_diag(const DeviceGraph& g, hipStream_t s);
void launch_damping(const DeviceGraph& g, double radius, double min_diag, double max_diag, int mode, hipStream_t s);
void launch_cost(const DeviceGraph& g, const double* poses, int part_row, hipStream_t s, int gate = 0);
void launch_evaluate_edges(const DeviceGraph& g, const double* poses, double* res, double* ja, double* jb, hipStream_t s);
void launch_pcg_init(const DeviceGraph& g, hipStream_t s);
void launch_pcg_iteration(const DeviceGraph& g, const CgParams& p, int odd, hipStream_t s);   // SpMV + update kernels of an odd/even iteration
void launch_pcg_finish(const DeviceGraph& g, const CgParams& p, hipStream_t s, int publish = 1);   // termination bookkeeping; publish: hand off to the host
void launch_gradient_norm(const DeviceGraph& g, hipStream_t s);
void launch_finalize_scalars(const DeviceGraph& g, int n_cost_part, hipStream_t s, int gate = 0);   // always hands off to the host
void launch_apply_step(const DeviceGraph& g, const double* step, hipStream_t s);     // for tests: delta -> candidate
void launch_spmv_plain(const DeviceGraph& g, hipStream_t s);
// q = A x of the step tail.  finish: CG termination bookkeeping first, run only once the CG has stopped; candidates: the
// diagonal lanes also write delta and the candidate poses (rows of this rank)
void launch_spmv_tail(const DeviceGraph& g, const CgParams& p, hipStream_t s, int finish, int candidates);
void launch_step_tail(const DeviceGraph& g, hipStream_t s, int gate);                  // fused model change / candidate / cost / scalar fold + hand-off
                                                                                       // gate bit 0: only once the CG has stopped; bit 1 (g.lm): exact step, no CG ran
// device-resident LM (g.lm != null): last kernel of a sequence (candidate -> current, gradient norm, opening tests of the next
// pass, "sequence seq_id is through" for the host), and the host's way back in after a non-terminal halt
void launch_accept_finish(const DeviceGraph& g, int seq_id, hipStream_t s);
void launch_lm_resume(const DeviceGraph& g, int cg_goes_on, hipStream_t s);
// the universal stream (UniOp above): one slot-shaped and one vector-shaped launch; the budget kernel (re)opens the stream for
// `decisions` more LM iterations; the publish kernel sets LmScalars::seq behind everything enqueued so far
void launch_uni_s(const DeviceGraph& g, const CgParams& p, int period, hipStream_t s);
void launch_uni_v(const DeviceGraph& g, const CgParams& p, double min_diag, double max_diag, hipStream_t s);
// fused form: one kernel symbol; `launch` = index of this launch in the stream (its parity selects the state
// slot, the partial-sum rows and the exchange buffer it reads; the device reports launch + 1 as LmScalars::slots_done)
void launch_uni_f(const DeviceGraph& g, const CgParams& p, int launch, double min_diag, double max_diag, hipStream_t s);
bool uni_f_supported(const DeviceGraph& g, const CgParams& p, int cluster);
// resident form (pgo_uni_resident.h): four kernel symbols in a fixed cycle, launch L plays role L % 4 (HEAD, whole CG, TAIL, LIN)
void launch_uni_r(const DeviceGraph& g, const CgParams& p, int launch, double min_diag, double max_diag, hipStream_t s);
bool uni_r_supported(const DeviceGraph& g, const CgParams& p, int cluster);
int uni_r_abort_word();       // index of the resident stream's abort word in DeviceGraph::flags
void launch_lm_budget(const DeviceGraph& g, int decisions, hipStream_t s, int next_launch = 0);
void launch_lm_publish(const DeviceGraph& g, hipStream_t s);
bool uni_supported(const DeviceGraph& g);
// owner-only pipelined CG of the sharded path: r0 / u0 of the owned rows; one launch per product (seq 0: w0 = A u0, seq i + 1:
// iteration i; mode 1 / 2: only the stop test the launch `seq` would apply and the CG state for the host, 1: with the hand-over).
// x needs no buffer of its own: cg_x is laid out like an exchange buffer of rows_per * 6 doubles per rank.
bool pipe_supported(const DeviceGraph& g, const CgParams& p, int cluster);
void launch_hdiag6(const DeviceGraph& g, double* buf, int phase, hipStream_t s);   // diagonals of the diagonal blocks: owned rows into buf (0) / other rows out of it (1)
void launch_pipe_init(const DeviceGraph& g, hipStream_t s);
void launch_pipe_cg(const DeviceGraph& g, const CgParams& p, int seq, int mode, hipStream_t s, unsigned long long gseq = 0, bool fold = true);   // gseq: device-initiated exchange, global number of this producing launch; fold = false: the caller's next launch folds the partial sums (the coarse level's restriction)
void launch_peer_signal(const DeviceGraph& g, unsigned long long gseq, hipStream_t s);   // ... behind k_pipe_init
// boundary exchange (DeviceGraph::bx): this rank's boundary rows of pipe_buf[buf] -> its segment of bx[buf]; fold_seq >= 0: and k_pipe_fold's
// job behind the CG launch fold_seq (n_entries partial triples, row 0 of the partial-sum arrays) with the sums at the end of the segment
void launch_pipe_pack(const DeviceGraph& g, int buf, int fold_seq, int n_entries, hipStream_t s);
void launch_pipe_fold(const DeviceGraph& g, int seq, unsigned long long gseq, hipStream_t s);   // one work-group: the g.n_wg partial triples of the producing launch -> this rank's three sums in the exchange buffer(s) (+ signal / wait)
void launch_pcg_spmv_only(const DeviceGraph& g, const CgParams& p, int odd, hipStream_t s);
void launch_pcg_update_only(const DeviceGraph& g, int odd, hipStream_t s, int mode = 0);   // mode: see k_pcg_update
void launch_spmv_refresh(const DeviceGraph& g, hipStream_t s, int on_the_fly = 0, int it_odd = 0);
void launch_model_delta_and_retract(const DeviceGraph& g, hipStream_t s, int gate = 0);
void launch_debug(const DeviceGraph& g, int which, hipStream_t s);
// batched solve: D^2 = clamp(diag(H~)) / radius[component] into g.d2 (then launch_damping mode 2), per-component step scalars
// (after launch_spmv_tail wrote q = A x, delta and the candidates), candidate -> current for the accepted components
void launch_batch_d2(const DeviceGraph& g, const BatchPlan& b, double min_diag, double max_diag, hipStream_t s);
void launch_batch_scalars(const DeviceGraph& g, const BatchPlan& b, hipStream_t s);
void launch_batch_accept(const DeviceGraph& g, const BatchPlan& b, hipStream_t s);
int vec_block();
int pose_block();
int edge_block();
int max_edge_wg();

}  // namespace pgo
