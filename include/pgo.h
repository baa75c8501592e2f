/* pgo.h — C ABI of the MI355X-native pose-graph nonlinear-least-squares core (libpgo_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of TurtleZhong/PoseGraph-Ceres: the solve that
 * src/POSE_GRAPH_CERES_PLUS/test/pose_graph_ceres_plus_finial.cpp hands to Ceres.  The reference
 * reaches that path through the Ceres C++ API (9 imported symbols); the header-only
 * facade in include/ceres/ maps those C++ calls 1:1 onto the entry points below, and any other host
 * language binds them directly (plain pointers and sizes, no C++ or torch types).  Each entry point
 * cites the reference interface it replaces (REF = src/POSE_GRAPH_CERES_PLUS).
 *
 * Conventions
 *   - every function returns PGO_OK (0) or a negative pgo_status; pgo_last_error() has the text.
 *     Nothing aborts or throws across this boundary (Ceres itself CHECK-aborts on API misuse).
 *   - quaternions are Hamilton, stored x,y,z,w (Eigen coeffs() order, REF/include/types.h:15-20).
 *   - the compute path is HIP on gfx950 only.  There is no CPU fallback: without a usable GPU
 *     pgo_solve / pgo_evaluate return PGO_ERR_NO_DEVICE.
 */
#ifndef PGO_H_
#define PGO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct below changes size or layout (101: pgo_solver_summary grew by the factor_* /
 * *_reduced fields in r02).  pgo_solve / pgo_solver_end / pgo_solve_batch write sizeof(pgo_solver_summary) bytes: a caller must
 * check pgo_version() == PGO_VERSION of the header it was compiled against before handing structs over (the facade's
 * ceres::Solve does, include/ceres/solver.h). */
#define PGO_VERSION 105

/* The library is built with -fvisibility=hidden: these entry points are all it exports. */
#if defined(__GNUC__)
#define PGO_API __attribute__((visibility("default")))
#else
#define PGO_API
#endif

typedef struct pgo_problem pgo_problem;

typedef enum pgo_status {
  PGO_OK = 0,
  PGO_ERR_INVALID_ARGUMENT = -1,
  PGO_ERR_NO_DEVICE = -2,
  PGO_ERR_HIP = -3,
  PGO_ERR_UNSUPPORTED = -4,
  PGO_ERR_NUMERICAL = -5
} pgo_status;

/* ceres::LossFunction kinds (REF/test/pose_graph_ceres_plus_finial.cpp:495 uses HuberLoss(1.0)). */
typedef enum pgo_loss_kind {
  PGO_LOSS_TRIVIAL = 0, PGO_LOSS_HUBER = 1,          /* the reference's */
  PGO_LOSS_SOFT_L_ONE = 2, PGO_LOSS_CAUCHY = 3, PGO_LOSS_ARCTAN = 4,  /* other Ceres 1.13 losses with rho'' <= 0 */
  /* Switchable constraint (Suenderhauf & Protzel) in closed form: the edge cost min_w [ w^2 s + Phi (1 - w)^2 ] over its
   * switch variable w has w* = Phi / (Phi + s), i.e. the robust kernel rho(s) = Phi s / (Phi + s) (rho' = (Phi / (Phi + s))^2
   * = w*^2, rho'' < 0), so loop-closure rejection needs no extra parameter block and no 7th row in the 6x6 BSR; loss_a = Phi
   * (SURVEY.md section 8f rank 3).  Same corrector branch as the Ceres kinds above (alpha = 0). */
  PGO_LOSS_SWITCHABLE = 5
} pgo_loss_kind;

/* ceres::LinearSolverType values the path understands (finial.cpp:536 sets SPARSE_NORMAL_CHOLESKY). */
typedef enum pgo_linear_solver {
  PGO_SPARSE_NORMAL_CHOLESKY = 0, /* exact solve of the LM system */
  PGO_BLOCK_JACOBI_PCG = 1        /* Ceres CGNR + JACOBI analogue: truncated PCG, Q-tolerance eta */
} pgo_linear_solver;

/* ceres::TerminationType */
typedef enum pgo_termination { PGO_CONVERGENCE = 0, PGO_NO_CONVERGENCE = 1, PGO_FAILURE = 2 } pgo_termination;

/* ceres::Solver::Options — the fields the path reads; defaults are the Ceres 1.13.0 defaults
 * recovered from the reference binary. */
typedef struct pgo_solver_options {
  int max_num_iterations;                 /* 50   (finial.cpp:535 sets 1000) */
  int linear_solver_type;                 /* pgo_linear_solver, default PGO_SPARSE_NORMAL_CHOLESKY */
  int jacobi_scaling;                     /* 1 */
  int max_linear_solver_iterations;       /* 500 */
  int min_linear_solver_iterations;       /* 0 */
  int max_num_consecutive_invalid_steps;  /* 5 */
  int cg_batch;                           /* CG iterations enqueued per host check (0 = auto) */
  int pcg_cluster_poses;                  /* poses per Jacobi block of the PCG preconditioner: 1 = 6x6 pose blocks (default;
                                             Ceres JACOBI is per parameter block), 2 = 12x12, 4 = 24x24 pieces of the odometry
                                             chain (Ceres CLUSTER_JACOBI analogue) */
  int cg_residual_reset_period;           /* 10: every so many CG iterations r is recomputed as b - A x instead of updated
                                             (Ceres conjugate_gradients_solver.cc, LinearSolver::Options::residual_reset_period);
                                             0 = never */
  int pcg_form;                           /* recurrences of the truncated PCG on one GPU: 0 = the library chooses (the one-launch
                                             pipelined form where it applies and eta >= 0.01), 1 = standard CG (Ceres' ConjugateGradientsSolver
                                             statement by statement: two dependent launches per iteration, residual refresh),
                                             2 = pipelined CG (Ghysels-Vanroose: same iterates in exact arithmetic, one launch per
                                             iteration), 3 = the same recurrences with the whole CG of an LM iteration in ONE launch (blocks and
                                             vectors resident in registers, a grid barrier per iteration; one such session per device at a
                                             time, the fused form otherwise); Summary::cg_form says which ran */
  int pcg_coarse_aggregate;               /* 0 (default): the block / cluster Jacobi alone.  >= 8: a COARSE LEVEL on top of the 2-pose cluster
                                             Jacobi (r06) — aggregates of this many consecutive poses of the trajectory with six rigid-body modes
                                             each, Galerkin coarse matrix inverted once per LM iteration, M^-1 = M_J^-1 + P (P'AP)^-1 P'.  It
                                             carries the long-wavelength correction a block Jacobi misses from dead reckoning: BASELINE
                                             configs[1] with eta = 0.1 ends 5.6 % BELOW the exact path's cost instead of 11 % above it, at the
                                             same number of CG iterations (32 .. 64 is the useful range).  The session runs the host-driven loop
                                             with the one-launch pipelined CG iteration (Summary::cg_form 2, ::coarse_level = aggregates).  Row
                                             shards: aggregates never straddle ranks; the Galerkin row panels (per LM iteration) and the
                                             restricted vector (per CG iteration, 6 doubles per aggregate) are all-gathered */
  int reserved_options;
  double function_tolerance;              /* 1e-6 */
  double gradient_tolerance;              /* 1e-10 */
  double parameter_tolerance;             /* 1e-8 */
  double initial_trust_region_radius;     /* 1e4 */
  double max_trust_region_radius;         /* 1e16 */
  double min_trust_region_radius;         /* 1e-32 */
  double min_relative_decrease;           /* 1e-3 */
  double min_lm_diagonal;                 /* 1e-6 */
  double max_lm_diagonal;                 /* 1e32 */
  double eta;                             /* 0.1 */
  double exact_r_tolerance;               /* relative residual of the exact path when it is iterative, 1e-13 */
} pgo_solver_options;

/* ceres::Solver::Summary — the fields the path fills (finial.cpp:538-543). */
typedef struct pgo_solver_summary {
  int termination_type;         /* pgo_termination */
  int num_successful_steps;
  int num_unsuccessful_steps;
  int num_iterations;           /* iteration records, iteration 0 included */
  int num_linear_solver_iterations;
  int num_poses;
  int num_edges;
  int reason;                   /* 1 function tol, 2 parameter tol, 3 gradient tol, 4 min radius, 5 max iterations,
                                   6 invalid steps, 7 linear solver failure */
  int linear_solver_used;       /* 0 GPU block-sparse Cholesky, 1 block-Jacobi PCG (eta), 2 PCG run to exact_r_tolerance
                                   (exact solve requested but the factorisation schedule was impractical), 3 exact solve
                                   requested, factorisation or PCG to exact_r_tolerance chosen per LM iteration */
  int factor_nnz_blocks;        /* 6x6 blocks of the Cholesky factor (0 when not used) */
  int factor_levels;            /* elimination-tree levels = dependent launches per factorisation */
  int num_factorizations;       /* LM iterations served by the GPU factorisation (the others of an exact request: PCG) */
  double initial_cost;
  double final_cost;
  double total_time_in_seconds;
  double setup_time_in_seconds;          /* topology build + upload */
  double linear_solver_time_in_seconds;
  double jacobian_evaluation_time_in_seconds;
  double residual_evaluation_time_in_seconds;
  double final_gradient_max_norm;
  double final_trust_region_radius;
  char message[256];
  int factor_kind;              /* 0 no factorisation, 1 enumerated 6x6 block pairs (pgo_direct), 2 supernodal multifrontal
                                   with FP64 MFMA fronts (pgo_front), 3 supernodal multifrontal with every front in the LDS of
                                   one workgroup (chain-like graphs) */
  int factor_max_front;         /* multifrontal: largest dense front (scalars) */
  double factor_flops;          /* flops of one numeric factorisation */
  int num_parameter_blocks_reduced;     /* FullReport's Reduced column: constant blocks removed */
  int num_parameters_reduced;
  int num_effective_parameters_reduced;
  int cg_form;                  /* CG of the PCG solves: 0 one rank, standard CG (two-kernel universal stream / batches), 1 several ranks,
                                   replicated standard CG (every rank updates every row, q all-gathered per iteration), 2 owner-only
                                   pipelined CG, one launch per iteration (several ranks: every rank updates its own rows, one all-gather per
                                   iteration; one rank: the same kernel on the symmetric tile form, graphs above the universal stream's size limit), 3 one rank,
                                   pipelined CG in the fused universal stream (one launch per CG iteration), 4 one rank, the same CG resident in one
                                   launch per LM iteration (grid barrier per CG iteration) */
  int cg_exchange;              /* how the ranks' CG exchanged their segments: 0 nothing to exchange (one rank), 1 a host-enqueued collective
                                   per CG iteration (RCCL all-gather, loopback copies), 2 by the kernels themselves (peer table: stores
                                   into every rank's buffer + flags; the IPC transport's normal mode), 3 a host-enqueued collective of the ranks'
                                   BOUNDARY rows only (rows with an edge to another rank + three sums per rank) */
  int sym_form;                 /* 1: the session kept the normal equations in the symmetric tile form (every interior off-diagonal block stored
                                   and read once; graphs above 600 k incidence slots, on one rank or row-sharded over several), 0: incidence-slot blocks */
  int coarse_level;             /* aggregates of the PCG's coarse level (0: none ran) */
} pgo_solver_summary;

/* One row per iteration, the numbers Summary::FullReport() tabulates with
 * minimizer_progress_to_stdout. */
typedef struct pgo_iteration_record {
  int iteration;
  int step_is_successful;
  int linear_solver_iterations;
  int reserved;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} pgo_iteration_record;

/* ---- library ---- */
PGO_API int pgo_version(void);
PGO_API const char* pgo_last_error(void);
/* number of HIP devices visible (0 on a CPU-only host; never an error) */
PGO_API int pgo_device_count(void);
/* binds the calling process to a device (one process per GPU); default device 0 */
PGO_API int pgo_set_device(int device);

/* ---- problem construction: ceres::Problem (finial.cpp:58, 491-528) ---- */
PGO_API pgo_problem* pgo_problem_create(void);                  /* ceres::Problem::Problem()  */
PGO_API void pgo_problem_destroy(pgo_problem* problem);         /* ceres::Problem::~Problem() */

/* Declares a pose whose translation lives at p[3] and quaternion at q[4] in CALLER memory; both are
 * updated in place by pgo_solve, exactly like the parameter blocks handed to
 * Problem::AddResidualBlock (finial.cpp:513-517).  Identity is the pointer value: re-adding the same
 * (p,q) pair returns the existing index.  Returns the pose index (>= 0) or a negative status. */
PGO_API int pgo_problem_add_pose(pgo_problem* problem, double* p, double* q);
/* n poses laid out at base + i*stride_doubles: p = 3 doubles, q = the 4 doubles after them.
 * Returns the index of the first pose. */
PGO_API int pgo_problem_add_poses(pgo_problem* problem, int n, double* base, int stride_doubles);

/* PoseGraph3dErrorTerm::Create(t_be, sqrt_information) + Problem::AddResidualBlock(cost, loss,
 * p_begin, q_begin, p_end, q_end) + SetParameterization(q, EigenQuaternionParameterization)
 * (REF/include/PoseGraph3dError.h:56-61, finial.cpp:508-522).  sqrt_information is the 6x6 row-major
 * matrix applied on the LEFT of the residual (the lower Cholesky factor of the information matrix);
 * NULL means identity. Returns the edge index (>= 0) or a negative status. */
PGO_API int pgo_problem_add_se3_between(pgo_problem* problem, int pose_begin, int pose_end, const double* t_be_p,
                                const double* t_be_q, const double* sqrt_information);
/* batch form: begin/end [n], t_be [n][7] (p then q), sqrt_information [n][36] or NULL */
PGO_API int pgo_problem_add_se3_between_batch(pgo_problem* problem, int n, const int* pose_begin, const int* pose_end,
                                      const double* t_be, const double* sqrt_information);

/* the single LossFunction instance shared by every residual block (finial.cpp:495,513) */
/* ---- pose / landmark problems (role of g2o's BlockSolver, Thirdparty/g2o/g2o/core/block_solver.hpp:47-87) ----
 * A 3-D point is a parameter block of size 3 (Euclidean) in CALLER memory, updated in place by pgo_solve; it shares the node
 * index space of the poses.  An observation is the point expressed in the observing pose's frame,
 *     r = L3 ( R(q_pose)^T (l - p_pose) - z ),      L3 = 3x3 row-major square-root information (NULL: identity),
 * with the problem's loss function.  Exact solves eliminate the point blocks first (the Schur complement onto the poses, as
 * BlockSolver does with Hll^-1), then the poses by nested dissection of the reduced graph.  Returns the index / the first index. */
PGO_API int pgo_problem_add_point(pgo_problem* problem, double* xyz);
PGO_API int pgo_problem_add_points(pgo_problem* problem, int n, double* base, int stride_doubles);
PGO_API int pgo_problem_add_point_observation_batch(pgo_problem* problem, int n, const int* pose, const int* point, const double* z,
                                                    const double* sqrt_information3);

PGO_API int pgo_problem_set_loss(pgo_problem* problem, int loss_kind, double loss_parameter);

/* Problem::SetParameterBlockConstant (finial.cpp:525-527). which: 1 = translation, 2 = rotation, 3 = both */
PGO_API int pgo_problem_set_pose_constant(pgo_problem* problem, int pose, int which);
/* same, addressed by the parameter-block pointer as Ceres does */
PGO_API int pgo_problem_set_parameter_block_constant(pgo_problem* problem, const double* block);

PGO_API int pgo_problem_num_poses(const pgo_problem* problem);
PGO_API int pgo_problem_num_edges(const pgo_problem* problem);

/* ---- solve: ceres::Solve(options, problem, &summary) (finial.cpp:534-539) ---- */
PGO_API void pgo_solver_options_init(pgo_solver_options* options);
/* Runs the Levenberg-Marquardt trust-region loop on the GPU and writes the result into the caller's
 * p/q memory (also on NO_CONVERGENCE).  records (may be NULL) receives up to records_capacity rows. */
PGO_API int pgo_solve(pgo_problem* problem, const pgo_solver_options* options, pgo_solver_summary* summary,
              pgo_iteration_record* records, int records_capacity);
/* Summary::IsSolutionUsable (finial.cpp:543) */
/* Device memory of destroyed problems is kept in a process-wide pool and reused by later problems (hipFree synchronises the
 * device: tearing a problem down the blocking way costs as much as a KITTI-scale solve).  This call returns the pooled blocks
 * to the driver; the pool holds at most 16 GB. */
PGO_API int pgo_release_device_memory(void);

/* Several INDEPENDENT problems solved together on one GPU (KITTI-scale graphs leave the machine idle: an LM iteration is a
 * chain of small dependent launches).  The problems become the components of one block-diagonal problem — one launch
 * sequence, n times the work per launch — while everything ceres::Solve decides stays per problem: trust-region radius,
 * accept / reject, every termination test (finial.cpp:534-543 per problem), iteration records, summary.  Each problem follows
 * the trace it follows through pgo_solve (to rounding).  Requirements: linear_solver_type PGO_SPARSE_NORMAL_CHOLESKY (the
 * reference's), one loss function shared by all problems, no communicator attached.  summaries: [n_problems];
 * records: [n_problems][capacity] or NULL.  total/setup times in the summaries are those of the whole batch.  Poses are
 * updated in caller memory exactly as by pgo_solve.  GPU only. */
PGO_API int pgo_solve_batch(pgo_problem* const* problems, int n_problems, const pgo_solver_options* options, pgo_solver_summary* summaries,
                    pgo_iteration_record* records, int capacity);

PGO_API int pgo_summary_is_solution_usable(const pgo_solver_summary* summary);
/* Summary::FullReport (finial.cpp:541): writes a NUL-terminated report, returns the length needed */
PGO_API size_t pgo_summary_full_report(const pgo_solver_summary* summary, const pgo_iteration_record* records,
                               int num_records, char* buffer, size_t capacity);

/* ---- evaluation at the current caller-side parameter values (ceres::Problem::Evaluate analogue;
 * what ResidualBlock::Evaluate produces per block: loss-corrected residuals and local-tangent
 * Jacobians, columns [dp(3) | dtheta(3)], constant blocks zeroed).  Any output may be NULL.
 *   residuals [E][6], jacobian_begin/end [E][36] row-major, gradient [N][6] ---- */
PGO_API int pgo_evaluate(pgo_problem* problem, double* cost, double* residuals, double* jacobian_begin,
                 double* jacobian_end, double* gradient);
/* Gauss-Newton blocks of J'J at the current values, without Jacobi scaling or damping:
 *   diag [N][36], offdiag [E][36] = J_begin' J_end per edge.  Any output may be NULL. */
PGO_API int pgo_normal_equations(pgo_problem* problem, double* diag, double* offdiag, double* gradient);
/* Solves (J'J + diag(d2)) x = b on the GPU with the selected linear solver at the current values
 * (solver-level parity tests).  d2, b, x are [N][6]; returns CG iterations through *iterations. */
PGO_API int pgo_linear_solve(pgo_problem* problem, const pgo_solver_options* options, const double* d2,
                     const double* b, double* x, int* iterations);
/* Plus: x_plus = x [+] delta for every pose (EigenQuaternionParameterization::Plus on q, p += dp),
 * written back into the caller's p/q memory.  delta [N][6]. */
PGO_API int pgo_plus(pgo_problem* problem, const double* delta);

/* ---- device-resident stepping for benchmarks: poses stay in HBM between calls ---- */
PGO_API int pgo_solver_begin(pgo_problem* problem, const pgo_solver_options* options);
/* runs up to n LM iterations (successful or not); *executed = iterations actually run, *done becomes 1
 * when a termination test fired (either may be NULL) */
PGO_API int pgo_solver_step(pgo_problem* problem, int n, int* executed, int* done);
/* restores the device state to the poses given at pgo_solver_begin (no host traffic) */
PGO_API int pgo_solver_reset(pgo_problem* problem);
PGO_API int pgo_solver_end(pgo_problem* problem, pgo_solver_summary* summary, pgo_iteration_record* records,
                   int records_capacity);
/* ---- launch trace of a stepping session (profiling aid; PCG on one GPU in the fused or the resident universal stream,
 * Summary::cg_form 3 / 4) ----
 * pgo_solver_trace_start (between pgo_solver_begin / pgo_solver_reset and the pgo_solver_step calls to be traced): from now on
 * every launch of the stream records what it did and when (device clock, 100 MHz ticks), up to max_launches launches; 0 stops
 * recording.  pgo_solver_trace_read (the stream is idle between pgo_solver_step calls): records[i] = {operation, start tick of
 * work-group 0, end tick of the last work-group to finish, phase stamps} of launch i since the trace started (4 words per
 * launch; phase stamps: four 16-bit tick counts from the top of ONE work-group — a CG launch: work-group 0's product done /
 * sums folded / rows updated / end, in the resident stream (one launch = the whole CG): ticks summed over the turns of its loop
 * spent up to the arrival at the grid barrier / inside the barrier / behind it until the next turn starts, and the CG iteration
 * count of the launch; a step tail: the deciding work-group's loops done / last arrival known / partials folded /
 * decided); operation: 0 nothing (stream
 * stopped or paused), 1 head (accept-finish, damping, Jacobi blocks, CG start), 2 first product, 3 CG iteration (or, once the CG
 * has stopped, the step tail's A x), 4 step tail + decision, 5 linearisation.  host[0] = launches the host enqueued, host[1] =
 * seconds it spent inside the launch calls since the trace started.  Returns the number of records or a negative status. */
/* which CG form the stepping session runs (the value Summary::cg_form will carry), or a negative status without a session */
PGO_API int pgo_solver_cg_form(pgo_problem* problem);
/* doubles one rank contributes to the collective a CG iteration of the stepping session enqueues (0 with one rank; the exchange is `world`
 * times that: the ranks' boundary rows + three sums where Summary::cg_exchange is 3, whole row segments otherwise), or a negative status */
PGO_API int pgo_solver_exchange_doubles(pgo_problem* problem);
PGO_API int pgo_solver_trace_start(pgo_problem* problem, int max_launches);
PGO_API int pgo_solver_trace_read(pgo_problem* problem, long long* records, int capacity, double host[2]);
/* repeats one kernel of the path `repeats` times on the solver stream between two HIP events and
 * returns the average milliseconds per launch.  kernel: "linearize", "spmv", "cost", "evaluate",
 * "pcg_update". */
PGO_API int pgo_time_kernel(pgo_problem* problem, const char* kernel, int repeats, double* avg_ms);

/* ---- loop-closure candidate search ----
 * Replaces getCandidatesIndex / isInSearchRange of PLUS/test/generate_edges_from_trajectory_origion.cpp:58-111 (the
 * generator of Edge_Candidates_index.txt): for frame k >= 1 the list is k-1 followed by every i < k - gap (reference:
 * gap = 100) with float32 squared distance <= search_radius^2 (reference: Config search_radius = 6), ascending.
 * xyz: n x 3 float32 camera centres.  Output in CSR form: row_ptr[n+1] (row 0 is empty), indices[row_ptr[n]].
 * Call with indices = NULL to obtain row_ptr (sizes) only; capacity = entries available in `indices`.
 * kernel_ms (optional): device time of the search kernels (HIP events).  GPU only: PGO_ERR_NO_DEVICE without one. */
PGO_API int pgo_generate_candidates(const float* xyz, int n, float search_radius, int gap, long long* row_ptr, int* indices,
                            long long capacity, double* kernel_ms);

/* ---- graph construction around the solve ----
 * pgo_read_trajectory: GroundTruth::loadPoses1 / loadPoses2 (REF/src/GroundTruth.cc:22-73).  format 1: rows
 * "x y z qx qy qz qw" — INCLUDING the reference's quaternion scramble (GroundTruth.cc:59-62: the file's qw lands in Eigen's
 * x, qx in y, qy in z, qz in w); format 2: KITTI rows of a 3x4 matrix.  Output: camera-to-world
 * transforms Twc, row-major 4x4, 16 doubles per pose holding the float32 values the reference keeps (CV_32F).  count receives
 * the number of complete rows in the file (also when it exceeds capacity; only `capacity` poses are written).  Host only.
 * (The reference's `while (inFile.good())` loop appends one more, unread, pose at the end of the file; not reproduced.) */
PGO_API int pgo_read_trajectory(const char* path, int format, double* Twc, int capacity, int* count);

/* Odometry measurements of ALL consecutive frames at once on the GPU (finial.cpp:206-224): edge e has id_begin = e + 1,
 * id_end = e, t_be[e] = Converter::toPose3d(float32(Tcw(e + 1) * Twc(e))) (converter.cc:150-155, 221-234), 7 doubles
 * p[3] q[4] (xyzw).  kernel_ms optional.  GPU only. */
PGO_API int pgo_build_odometry_edges(int n_frames, const double* Twc, double* t_be, double* kernel_ms);

/* checkFrame's acceptance rules (finial.cpp:162-293, 486-489) over RECORDED front-end results — ORB matching and PnP are
 * outside the path, their outputs per candidate pair come in `obs` (parallel to cand_idx; may be NULL: odometry only).
 * Frames are visited in id order, the candidates of frame f are cand_idx[cand_ptr[f] .. cand_ptr[f+1]) in file order
 * (Edge_Candidates_index.txt): a candidate one frame back yields the odometry edge; any other needs nmatches >
 * match_threshold, inliers > inlier_threshold, normofTransform(rvec, tvec) < norm_threshold, at most one loop edge per
 * current frame, and is skipped when it obtained a loop edge as a current frame itself.  Output edges in the reference's
 * order (id_begin = current, id_end = earlier frame, t_be = toPose3d of the float32 transform); loop_list = the
 * "cur prev" rows of edges_for_loop.txt (pairs more than loop_list_gap frames apart), 2 ints per row.  n_edges / n_loop_list
 * always receive the required counts.  The odometry measurements are computed on the GPU (pgo_build_odometry_edges). */
typedef struct pgo_pair_observation {
  int nmatches;        /* ORBmatcher::MatcheTwoFrames */
  int inliers;         /* PnP-RANSAC inliers */
  double rvec[3];      /* rotation vector of T_cur<-prev (cv::Rodrigues convention) */
  double tvec[3];
} pgo_pair_observation;
typedef struct pgo_edge_rules {
  int match_threshold;     /* 280, finial.cpp:226 */
  int inlier_threshold;    /* 100, finial.cpp:234 */
  double norm_threshold;   /* 0.7, finial.cpp:234 */
  int loop_list_gap;       /* 100, finial.cpp:285 */
  int reserved;
} pgo_edge_rules;
PGO_API void pgo_edge_rules_init(pgo_edge_rules* rules);
PGO_API int pgo_build_edges(int n_frames, const double* Twc, const long long* cand_ptr, const int* cand_idx, const pgo_pair_observation* obs,
                    const pgo_edge_rules* rules, int* id_begin, int* id_end, double* t_be, long long capacity, long long* n_edges,
                    int* loop_list, long long loop_capacity, long long* n_loop_list);

/* ---- batched MotionEstimate solves ----
 * Replaces MotionEstimate::BuildOptimizationProblem / SolveOptimizationProblem (REF/src/MotionEstimate.cc:71-129) and the
 * ReprojectionError3Dto2D functor (REF/include/MotionEstimate.h:34-91) for MANY candidate frame pairs at once: problem k
 * has the points [point_ptr[k], point_ptr[k+1]) of `points` ([total][3], 3-D points in the last frame) and
 * `observations` ([total][2], pixels in the current frame), the shared intrinsics fx fy cx cy, and the parameter blocks
 * q[k] (xyzw, EigenQuaternionParameterization) and t[k], updated in place.  The reference keeps q constant (:108) and
 * estimates t with HuberLoss(1.0), max_num_iterations = 1000, exact steps: those are the defaults.
 * summaries (optional) [n_problems]; kernel_ms (optional) = device time of the one launch.  GPU only. */
typedef struct pgo_reproj_options {
  int max_num_iterations;
  int q_constant, t_constant;
  int loss_kind;                 /* pgo_loss_kind */
  int jacobi_scaling;
  int max_num_consecutive_invalid_steps;
  double loss_a;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
} pgo_reproj_options;
typedef struct pgo_reproj_summary {
  int termination_type;          /* pgo_termination_type */
  int reason;                    /* 1 function tol, 2 parameter tol, 3 gradient tol, 4 min radius, 5 max iterations, 6 invalid steps */
  int num_iterations;            /* iteration 0 included, as Ceres counts */
  int num_successful_steps, num_unsuccessful_steps;
  int num_points;
  double initial_cost, final_cost;
} pgo_reproj_summary;
PGO_API void pgo_reproj_options_init(pgo_reproj_options* options);
PGO_API int pgo_reproj_solve_batch(int n_problems, const long long* point_ptr, const double* points, const double* observations,
                           const double intrinsics[4], double* q, double* t, const pgo_reproj_options* options,
                           pgo_reproj_summary* summaries, double* kernel_ms);

/* ---- one process per GPU: edge/row sharding over RCCL ---- */
/* contiguous share [begin,end) of n units for `rank` of `world` (host-only helper, no GPU needed) */
PGO_API int pgo_shard_range(long long n, int rank, int world, long long* begin, long long* end);
/* Equal shares of the rows (the ownership rule until r05; kept as a helper): rank r owns the poses [r * rows_per, (r + 1) * rows_per)
 * cut at n_poses, rows_per = ceil(n_poses / world) rounded up to a multiple of 4.  rows_per optional. */
PGO_API int pgo_row_shard_range(long long n_poses, int rank, int world, long long* begin, long long* end, int* rows_per);
/* Row ownership of the sharded solve — the rule pgo_comm_init itself applies (r06): contiguous shares of the poses cut where the
 * incidence slots balance (pose v weighs 1 + its degree), at multiples of 4 (preconditioner clusters never straddle ranks).
 * cut[0 .. world]: rank r owns the poses [cut[r], cut[r + 1]); rows_per (optional) = the longest share rounded up to a multiple of
 * 4 = the equal segment every rank's rows occupy in the exchanged arrays.  Host only, no GPU needed.
 * (BASELINE configs[3] over 8 ranks: heaviest rank 1.18x the mean by row count, <= 1.01x by this rule.) */
PGO_API int pgo_row_shard_cuts(long long n_poses, long long n_edges, const int* id_begin, const int* id_end, int world, long long* cut, int* rows_per);
/* Development / test knobs (r06): process-wide values the library's A/B paths and plan heuristics read — what sixteen of its environment
 * variables were (the fifteen variables that remain are listed in EXPERIMENTS.md).  value = NAN puts a knob back to its default.  Unknown
 * name: PGO_ERR_INVALID_ARGUMENT.  pgo_tuning_describe(i, &name, &what) returns the number of knobs and, for 0 <= i < that, the i-th
 * knob's name and one line on what it does.  Not for production callers: every knob's default is the measured choice. */
PGO_API int pgo_tuning_set(const char* name, double value);
PGO_API int pgo_tuning_get(const char* name, double* value, int* is_set);
PGO_API int pgo_tuning_describe(int index, const char** name, const char** what);
/* 128-byte RCCL unique id created on rank 0 and handed to every rank by the launcher */
PGO_API int pgo_comm_get_unique_id(unsigned char id[128]);
/* Attaches rank `rank` of `world` to the problem BEFORE the first solve/evaluate.  Every rank must hold the same problem
 * (same poses, same edges in the same order); rank r then owns a contiguous range of pose rows, evaluates every edge
 * incident to them, and the ranks exchange one all-gather per CG iteration (q = A p and the p'q partials), one per
 * accepted LM step (J'J diagonal blocks, J'r) and one per LM iteration for cluster preconditioners.  All ranks take
 * identical decisions from identical scalars, so no other coordination is needed. */
PGO_API int pgo_comm_init(pgo_problem* problem, const unsigned char id[128], int rank, int world);
/* The same ownership over a transport of PROCESSES that map each other's exchange buffers through hipIpc memory handles (one
 * process per rank; the ranks' devices must be IPC-capable peers — the GPUs of one node, or one GPU shared by several processes):
 * the kernels of the owner-only CG then store into every rank's buffer and flag array themselves (no host-enqueued collective per
 * CG iteration); the few exchanges per LM iteration outside the CG go through an IPC-mapped staging buffer.  `name`: a POSIX
 * shared-memory name ("/...") unique to the group; rank 0 creates it, every rank calls this collectively. */
PGO_API int pgo_comm_init_ipc(pgo_problem* problem, const char* name, int rank, int world);
/* Test transport: `world` virtual ranks = problems driven by host threads of ONE process on one GPU, segments exchanged
 * by device-to-device copies.  Lets the sharded path be validated on a single-GPU machine. */
/* development / test hook (tests/test_lm_rules_cpu.py): one application of the trust-region rules the device and the host driver share
 * (csrc/pgo_lm_rules.h), on the host, no GPU.  state[4] = radius, decrease factor, current cost, |x| (in / out); step[4] = candidate
 * cost, model cost change, |step|^2, |x|^2; out[6] = outcome (0 invalid, 1 invalid and failed, 2 parameter tolerance, 3 function
 * tolerance, 4 accepted, 5 rejected), step_is_successful, relative decrease, cost change, radius after, the message's number. */
PGO_API int pgo_debug_lm_decide(const pgo_solver_options* options, double state[4], const double step[4], int cg_status, double out[6]);
/* development aid (tools/comm_stress.py): `iters` exchanges of a test pattern through the problem's communicator, mismatches counted */
PGO_API int pgo_debug_comm_stress(pgo_problem* problem, int iters, int seg_doubles);
PGO_API void* pgo_loopback_create(int world);
PGO_API void pgo_loopback_destroy(void* group);
PGO_API int pgo_comm_init_loopback(pgo_problem* problem, void* group, int rank);

#ifdef __cplusplus
}
#endif
#endif /* PGO_H_ */
