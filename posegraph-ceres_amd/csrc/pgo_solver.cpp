// pgo_solver.cpp — host side of libpgo_hip.so: ceres::Problem-style bookkeeping, topology build,
// the Levenberg-Marquardt trust-region driver and the C ABI of include/pgo.h.
//
// Reference behaviour mirrored (REF = src/POSE_GRAPH_CERES_PLUS):
//   BuildOptimizationProblem  REF/test/pose_graph_ceres_plus_finial.cpp:491-528
//   SolveOptimizationProblem  REF/test/pose_graph_ceres_plus_finial.cpp:531-544
//   ceres::Solve control flow SURVEY.md Appendix A.6 (Ceres 1.13 TrustRegionMinimizer order,
//                             LevenbergMarquardtStrategy radius rules, Options defaults of row a9)
// All arithmetic of the path runs in the HIP kernels of pgo_kernels.hip; this file only sequences
// them and takes the accept/reject decisions from scalars the device reduces.
#include "../../include/pgo.h"
#include "pgo_kernels.h"
#include "pgo_lm_rules.h"
#include "pgo_direct.h"
#include "pgo_front.h"
#include "pgo_comm.h"
#include "pgo_pool.h"

namespace pgo { int comm_stress(Comm* c, int iters, size_t seg, hipStream_t s, int* mismatches); }

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <memory>
#include <map>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_error;

int set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t err__ = (expr);                                                                \
    if (err__ != hipSuccess)                                                                  \
      return set_error(PGO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), \
                       __FILE__, __LINE__);                                                   \
  } while (0)

static_assert(sizeof(pgo::LmRecord) == sizeof(pgo_iteration_record), "LmRecord mirrors pgo_iteration_record");
typedef std::chrono::steady_clock Clock;
inline double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// Host-to-device copy through a process-wide pinned staging buffer.  A hipMemcpyAsync from pageable memory makes the
// runtime register the user pages with the device on the fly; once a problem's large device allocations exist that
// registration was measured to stall the copy for 6-25 ms on this stack (a 20 KB list!), while a copy from memory
// pinned once costs microseconds.  Same for the way back.
struct Staging {
  std::mutex mu;
  char* buf = nullptr;
  static constexpr size_t cap = (size_t)4 << 20;
  hipError_t ensure() {
    if (buf) return hipSuccess;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&buf), cap, hipHostMallocPortable);   // one buffer for every device the process uses; kept for the process lifetime
    if (e != hipSuccess) buf = nullptr;
    return e;
  }
};
inline Staging& staging() { static Staging* st = new Staging(); return *st; }

// While an UploadScope is alive on this thread, host-to-device copies of its stream are packed side by side into the staging
// buffer and enqueued WITHOUT waiting for each other; the one wait is at the end of the scope (or when the buffer is full).  A
// topology upload is 10-20 small arrays: a wait per array was 15-20 us each, 0.2-0.3 ms of a KITTI-00-scale setup.  The scope
// holds the staging buffer for its lifetime (other host threads' copies wait, as they did per copy).
struct UploadScope {
  hipStream_t stream;
  size_t used = 0;
  std::unique_lock<std::mutex> lock;
  UploadScope* outer;
  static UploadScope*& current() { static thread_local UploadScope* cur = nullptr; return cur; }
  explicit UploadScope(hipStream_t s) : stream(s), outer(current()) {
    if (!outer) { lock = std::unique_lock<std::mutex>(staging().mu); current() = this; }
  }
  hipError_t finish() {
    if (outer || used == 0) return hipSuccess;
    used = 0;
    return hipStreamSynchronize(stream);
  }
  ~UploadScope() {
    if (outer) return;
    (void)finish();
    current() = nullptr;
  }
};

inline hipError_t staged_copy(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t s) {
  Staging& st = staging();
  const size_t cap = Staging::cap;
  UploadScope* scope = UploadScope::current();
  if (scope && scope->stream != s) scope = nullptr;           // (not this scope's stream: the plain, waiting copy below — the scope's thread holds the lock)
  if (bytes > 2 * cap) {   // the big topology arrays: uploaded before the problem's device allocations, where the direct copy is fast
    hipError_t e = hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
  }
  std::unique_lock<std::mutex> lock;
  if (!UploadScope::current()) lock = std::unique_lock<std::mutex>(st.mu);
  hipError_t e = st.ensure();
  if (e != hipSuccess) return e;
  char* stage = st.buf;
  if (scope && to_device && bytes <= cap) {
    if (scope->used + bytes > cap) { e = scope->finish(); if (e != hipSuccess) return e; }
    std::memcpy(stage + scope->used, src, bytes);
    e = hipMemcpyAsync(dst, stage + scope->used, bytes, hipMemcpyHostToDevice, s);
    scope->used += (bytes + 255) / 256 * 256;
    return e;
  }
  if (UploadScope::current()) { e = UploadScope::current()->finish(); if (e != hipSuccess) return e; }   // the buffer is about to be reused from its start
  for (size_t off = 0; off < bytes; off += cap) {
    const size_t len = std::min(cap, bytes - off);
    if (to_device) std::memcpy(stage, static_cast<const char*>(src) + off, len);
    e = to_device ? hipMemcpyAsync(static_cast<char*>(dst) + off, stage, len, hipMemcpyHostToDevice, s)
                  : hipMemcpyAsync(stage, static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (!to_device) std::memcpy(static_cast<char*>(dst) + off, stage, len);
  }
  return hipSuccess;
}
inline hipError_t staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t s) { return staged_copy(dst, src, bytes, true, s); }
inline hipError_t staged_d2h(void* dst, const void* src, size_t bytes, hipStream_t s) { return staged_copy(dst, src, bytes, false, s); }

// Device blocks of destroyed problems are kept in a process-wide, size-keyed free list and handed to the next allocation they
// fit: hipFree synchronises the whole device and costs ~0.1 ms per block — releasing the ~70 buffers of a problem took 2-9 ms,
// as much as a KITTI-scale solve.  Only whole-problem teardown goes through the pool (the owner synchronises its stream
// first); a buffer that is re-allocated in mid-life is freed the blocking way, since work in flight may still read it.
// PGO_POOL_MAX_GB bounds the cached bytes (default 16; 0 switches the pool off); pgo_release_device_memory() empties it.
struct DevicePool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> blocks;   // (device, capacity in bytes) -> block
  size_t cached = 0;
  static size_t limit() {
    static const size_t lim = (size_t)((getenv("PGO_POOL_MAX_GB") ? atof(getenv("PGO_POOL_MAX_GB")) : 16.0) * 1e9);
    return lim;
  }
  hipError_t get(size_t bytes, void** out, size_t* cap, int* dev) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    *dev = d;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = blocks.lower_bound(std::make_pair(d, bytes));
      if (it != blocks.end() && it->first.first == d && it->first.second <= std::max(2 * bytes, bytes + (1u << 20))) {
        *out = it->second;
        *cap = it->first.second;
        cached -= *cap;
        blocks.erase(it);
        return hipSuccess;
      }
    }
    *cap = bytes;
    e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {      // the library itself may be holding the memory (blocks of other sizes): give it back and try once more
      (void)hipGetLastError();
      trim();
      e = hipMalloc(out, bytes);
    }
    return e;
  }
  void put(void* p, size_t cap, int dev) {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (cached + cap <= limit()) { blocks.emplace(std::make_pair(dev, cap), p); cached += cap; return; }
    }
    (void)hipFree(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : blocks) (void)hipFree(kv.second);
    blocks.clear();
    cached = 0;
  }
};
inline DevicePool& device_pool() { static DevicePool* pool = new DevicePool(); return *pool; }   // never destroyed: no HIP calls at exit

// Streams and small pinned blocks are recycled as well: on this stack hipStreamDestroy takes 1.5-3 ms and hipStreamCreate /
// hipHostMalloc + hipHostFree ~0.5 ms together, i.e. a third of a KITTI-00-scale solve for a caller that builds and destroys
// one problem per solve (the facade's ceres::Problem does).  A stream is handed back idle (its owner synchronised it); a
// pinned block is handed back with whatever it held and is cleared by the next owner.  pgo_release_device_memory() empties both.
struct HostSidePool {
  std::mutex mu;
  std::vector<std::pair<int, hipStream_t>> streams;             // (device, stream)
  std::multimap<size_t, void*> pinned;                          // capacity -> mapped, portable host block
  static bool off() { static const bool v = getenv("PGO_NO_HOST_POOL") && getenv("PGO_NO_HOST_POOL")[0] == '1'; return v; }
  hipError_t get_stream(int dev, hipStream_t* out) {
    if (!off()) {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < streams.size(); ++i)
        if (streams[i].first == dev) { *out = streams[i].second; streams[i] = streams.back(); streams.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  void put_stream(int dev, hipStream_t s) {
    if (!off()) {
      std::lock_guard<std::mutex> lk(mu);
      if (streams.size() < 64) { streams.emplace_back(dev, s); return; }
    }
    (void)hipStreamDestroy(s);
  }
  hipError_t get_pinned(size_t bytes, void** out, size_t* cap) {
    if (!off()) {
      std::lock_guard<std::mutex> lk(mu);
      auto it = pinned.lower_bound(bytes);
      if (it != pinned.end() && it->first <= std::max<size_t>(4 * bytes, 4096)) { *out = it->second; *cap = it->first; pinned.erase(it); return hipSuccess; }
    }
    *cap = (bytes + 4095) / 4096 * 4096;
    return hipHostMalloc(out, *cap, hipHostMallocMapped | hipHostMallocPortable);
  }
  void put_pinned(void* p, size_t cap) {
    if (!off()) {
      std::lock_guard<std::mutex> lk(mu);
      if (pinned.size() < 64 && cap <= (1u << 20)) { pinned.emplace(cap, p); return; }
    }
    (void)hipHostFree(p);
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& ds : streams) (void)hipStreamDestroy(ds.second);
    streams.clear();
    for (auto& kv : pinned) (void)hipHostFree(kv.second);
    pinned.clear();
  }
};
inline HostSidePool& host_side_pool() { static HostSidePool* pool = new HostSidePool(); return *pool; }

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0;
  int dev = 0;
  ~DevBuf() { release(true); }
  void release(bool to_pool = false) {
    if (!p) return;
    if (to_pool) device_pool().put(p, cap_bytes, dev); else (void)hipFree(p);
    p = nullptr; n = 0; cap_bytes = 0;
  }
  hipError_t alloc(size_t count) {
    release(false);
    n = count;
    if (count == 0) return hipSuccess;
    void* q = nullptr;
    const hipError_t e = device_pool().get(count * sizeof(T), &q, &cap_bytes, &dev);
    p = static_cast<T*>(q);
    if (e != hipSuccess) { p = nullptr; n = 0; cap_bytes = 0; }
    return e;
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return staged_h2d(p, h.data(), h.size() * sizeof(T), s);
  }
  hipError_t upload(const T* h, size_t count, hipStream_t s) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return staged_h2d(p, h, count * sizeof(T), s);
  }
  // copy into the existing allocation (capacity n), no hipMalloc
  hipError_t store(const std::vector<T>& h, hipStream_t s) {
    if (h.size() > n) return hipErrorInvalidValue;
    if (h.empty()) return hipSuccess;
    return staged_h2d(p, h.data(), h.size() * sizeof(T), s);
  }
  hipError_t zero(hipStream_t s) { return n ? hipMemsetAsync(p, 0, n * sizeof(T), s) : hipSuccess; }
};

// Host staging array WITHOUT value initialisation: the big slot-ordered arrays are first touched (and fully written) by the
// threads of parallel_for, not zero-filled by the caller.
struct HostArray {
  std::unique_ptr<double[]> p;
  size_t n = 0;
  void resize(size_t count) { p.reset(count ? new double[count] : nullptr); n = count; }
  double& operator[](size_t i) { return p[i]; }
  const double* data() const { return p.get(); }
  bool empty() const { return n == 0; }
};

struct LmState {
  bool active = false;
  bool terminated = false;
  bool pending_record = false;
  int termination = PGO_NO_CONVERGENCE;
  int reason = 5;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  bool gmax_deferred = false;  // gradient norm of the accepted point is read at the next host sync
  double x_cost = 0, x_norm = 0, gmax = 0, initial_cost = 0;
  int num_successful = 0, num_unsuccessful = 0, num_consecutive_invalid = 0, num_linear_iterations = 0;
  pgo_iteration_record cur{};
  std::vector<pgo_iteration_record> records;
  double t_total = 0, t_linear = 0, t_jacobian = 0, t_residual = 0, t_setup = 0;
  // exact request with both a factorisation and PCG available (DirectSymbolic::hybrid): which one serves the next iteration
  bool hybrid_pcg = false;        // PCG served the last iteration within its budget
  int hybrid_direct_run = 0;      // consecutive iterations served by the factorisation
  int hybrid_probe_after = 1;     // ... after which PCG is tried again (doubles on every failed try, up to 16)
  double hybrid_fail_radius = 0;  // trust-region radius of the last over-budget try: a 10x smaller radius (10x the damping) earns an early try
  int hybrid_direct = 0, hybrid_pcg_ok = 0, hybrid_pcg_over = 0;   // statistics (PGO_VERBOSE)
  int n_factorizations = 0;       // LM iterations served by the GPU factorisation
  int num_trial_steps = 0;        // linear solves + step tails enqueued (pgo_solver_step counts executed iterations with it)
  std::string message;
};

}  // namespace

struct pgo_problem {
  // ---- host-side problem (ceres::Problem bookkeeping) ----
  std::vector<double*> pp, qq;
  std::unordered_map<const double*, int> block_of_ptr;  // pose*2 + (0: p block, 1: q block)
  std::vector<uint8_t> cmask;
  std::vector<int> ia, ib;
  std::vector<double> meas;       // 7 per edge
  std::vector<double> sqrt_info;  // 36 per edge once any edge carries a non-identity matrix
  bool has_info = false;
  int loss_kind = PGO_LOSS_TRIVIAL;
  double loss_a = 1.0;
  bool topo_dirty = true;

  // ---- device ----
  int device = 0;
  bool stream_ready = false;
  hipStream_t stream = nullptr;
  pgo::DeviceGraph g{};
  std::vector<int> edge_begin_slot;  // host: slot of the begin-side incidence of every edge
  DevBuf<int> d_slot_col, d_slot_row, d_wg_slot_begin, d_wg_row_begin, d_row_slot_begin, d_row_slot_cnt, d_edge_a, d_edge_b, d_flags;
  DevBuf<uint8_t> d_slot_side, d_cmask;
  DevBuf<double> d_smeas, d_sW, d_emeas, d_eW, d_eL, d_pose_x, d_pose_c, d_pose_0, d_bsr, d_Hdiag, d_Minv, d_grad,
      d_scale, d_d2, d_diagc, d_cg_b, d_cg_x, d_cg_r, d_cg_z, d_cg_q, d_cg_p0, d_cg_p1, d_delta, d_part_rz, d_part_q,
      d_part_rr, d_part_bb, d_part_misc, d_tmp_a, d_tmp_b, d_tmp_c;
  DevBuf<pgo::CgState> d_cg;
  // spare set of the linearisation (blocks, diagonal blocks, gradient): the candidate point is linearised into it right behind
  // the step tail, before the host has decided; an accepted step swaps the sets (one rank, eager enqueue)
  DevBuf<double> d_bsr2, d_Hdiag2, d_grad2;
  bool spec_ready = false;
  pgo::LmScalars* scal = nullptr;  // pinned, device visible
  size_t scal_cap = 0;
  // device-resident LM (pgo_kernels.h LmDev): the trust-region decisions are taken on the device and the host enqueues the
  // kernel sequences of the next iterations ahead of them (one rank, eager enqueue; PGO_NO_PIPELINE=1 keeps the host in the loop)
  DevBuf<pgo::LmDev> d_lm;
  bool pipelined = false;
  int pipe_seq = 0;                // id of the last sequence enqueued (LmScalars::seq_done catches up with it)
  int pipe_last_nb = 0;            // CG iterations in the last sequence enqueued
  int pipe_pulled = 0;             // next iteration record to copy from the pinned ring into LmState::records
  bool pipe_dirty = true;          // LmState was (re)initialised by the host: upload it before the next sequence
  bool universal = false;          // PCG on one rank: the universal stream (pgo_kernels.h UniOp) instead of allotted sequences
  int uni_enq = 0;                 // vector-shaped launches of the stream enqueued since the last upload
  double pipe_t_linear0 = 0, pipe_t_jacobian0 = 0;   // LmState times when the device clocks were last zeroed
  // captured CG batches, keyed by the number of iterations in the batch
  struct CapturedBatch { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
  std::unordered_map<int, CapturedBatch> cg_graphs;
  pgo::CgParams cg_graph_params{};
  bool use_graph = false;
  int last_cg_iterations = 0;

  // exact solver (GPU block-sparse Cholesky), built lazily when SPARSE_NORMAL_CHOLESKY is requested
  std::vector<int> h_slot_row, h_slot_col, h_row_slot_begin;
  std::vector<uint8_t> h_slot_side;
  pgo::DirectSymbolic dsym;
  pgo::DirectPlan dplan{};
  bool direct_analyzed = false, direct_usable = false;
  DevBuf<int> dd_split_blk, dd_split_sub, dd_split_sub_diag, dd_upd_split, dd_panel_cols, dd_blk_lpos, dd_split_dblk, dd_col_flag;
  int direct_epoch = 0;
  bool split_two_launch = false;   // a single-launch SPLIT step timed out once: this problem keeps to the two-launch form
  bool sfront_levels = false;      // small-front plan: one launch per level (a wait of the single-launch form ran out, or PGO_SFRONT_FUSED=0)
  int sfront_epoch = 0;
  unsigned sfront_tickets = 0;     // tickets handed out by the single-launch factorisations so far
  DevBuf<uint8_t> dd_split_diag;
  DevBuf<int> dd_perm, dd_col_ptr, dd_blk_row, dd_asrc_ptr, dd_asrc_slot, dd_upd_ptr, dd_upd_a, dd_upd_b, dd_level_ptr,
      dd_level_cols, dd_rowl_ptr, dd_rowl_blk, dd_rowl_col;
  DevBuf<double> dd_Lval, dd_y;
  hipGraph_t direct_graph = nullptr;
  hipGraphExec_t direct_exec = nullptr;
  // exact solver for mesh-like graphs: supernodal multifrontal Cholesky with FP64 MFMA fronts (pgo_front.*)
  pgo::FrontSymbolic fsym;
  pgo::FrontPlan fplan{};
  bool front_usable = false;
  bool no_sfront = false;          // the union of a batched solve keeps to the enumerated schedule
  // host analysis of the factorisation on a helper thread, started inside prepare() as soon as the slot topology exists and
  // joined by prepare_direct(): it overlaps the array fills, the uploads and the evaluation of iteration zero
  std::thread analysis_thread;
  int analysis_kind = 0;           // what it chose: 0 none (iterative path), 1 enumerated 6x6 pairs, 2 MFMA fronts, 3 small fronts
  bool want_direct = false;        // set by lm_begin around prepare(): an exact request is coming
  bool sfront_usable = false;      // every front fits the LDS: one launch per tree level (pgo_front.h, SFRONT_MAX)
  pgo::SFrontPlan splan{};
  DevBuf<pgo::SFront> ds_sf;
  DevBuf<double> ds_L, ds_U, ds_W;
  DevBuf<int> ds_urel, ds_osrc, ds_upos, ds_list, ds_done;
  DevBuf<long long> ds_stamps;     // PGO_SF_STAMPS=1 (development aid)
  DevBuf<int> df_st_table, df_st_pred_ptr, df_st_pred, df_st_need;
  DevBuf<unsigned long long> df_st_count;
  bool front_launches = false;     // multifrontal plan: one launch per phase of a round (a wait of the single-launch form ran out, or PGO_FRONT_FUSED=0)
  unsigned long long front_epoch = 0, front_tickets = 0;
  DevBuf<int> df_perm, df_idx, df_child, df_rel, df_cstart, df_col_front, df_ablk_ptr, df_ablk_slot, df_ablk_front, df_ablk_pos, df_wg_job, df_wg_tile,
      df_bwd_front, df_bwd_chunk, df_bwdb_front, df_bwdb_chunk, df_asm_tile, df_asm_contrib;
  DevBuf<pgo::FrontDesc> df_fronts;
  DevBuf<pgo::FrontJob> df_jobs;
  DevBuf<double> df_Fval, df_Winv, df_x;
  // cluster-Jacobi preconditioner topology (built when the option asks for clusters of 2 or 4 poses)
  DevBuf<int> d_cl_ptr, d_cl_slot;
  DevBuf<uint8_t> d_cl_rc;
  int cluster_built = 0;

  // one process per GPU: the communicator of the row-sharded path (null = single rank)
  pgo::Comm* comm = nullptr;

  pgo_solver_options opt{};
  LmState lm;

  ~pgo_problem() {
    const bool verbose = getenv("PGO_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (analysis_thread.joinable()) analysis_thread.join();
    if (stream_ready) (void)hipStreamSynchronize(stream);   // the device buffers go back to the pool: nothing may be in flight
    drop_graph();
    delete comm;
    const double t1 = ms();
    if (scal) host_side_pool().put_pinned(scal, scal_cap);
    const double t2 = ms();
    if (stream_ready) host_side_pool().put_stream(device, stream);
    if (verbose && stream_ready) std::fprintf(stderr, "[pgo] problem teardown: sync %.2f, pinned block %.2f, stream %.2f ms (members follow)\n", t1, t2 - t1, ms() - t2);
  }
  void drop_direct_graph() {
    if (direct_exec) { (void)hipGraphExecDestroy(direct_exec); direct_exec = nullptr; }
    if (direct_graph) { (void)hipGraphDestroy(direct_graph); direct_graph = nullptr; }
  }
  void drop_graph() {
    drop_direct_graph();
    for (auto& kv : cg_graphs) {
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    cg_graphs.clear();
  }
};

namespace {

int ensure_device(pgo_problem* P) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return set_error(PGO_ERR_NO_DEVICE,
                     "no HIP device available: the pose-graph path runs on gfx950 only and has no CPU fallback");
  }
  HIP_TRY(hipSetDevice(P->device));
  if (!P->stream_ready) {
    HIP_TRY(host_side_pool().get_stream(P->device, &P->stream));
    P->stream_ready = true;
    // Launch sequences are enqueued eagerly by default: on this stack (ROCm 7.2, MI355X) the host runs ahead of the GPU and a
    // captured hipGraph of the same kernels is no faster (C2: 0.317 vs 0.317 ms per LM iteration without residual refreshes,
    // 0.322 eager vs 0.353 graph with them; KITTI-00 exact 0.79 vs 0.81 ms).  PGO_GRAPH=1 replays captured batches instead.
    const char* gr = getenv("PGO_GRAPH");
    const char* ng = getenv("PGO_NO_GRAPH");
    P->use_graph = (gr && gr[0] == '1') && !(ng && ng[0] == '1');
  }
  if (!P->scal) {
    void* blk = nullptr;
    HIP_TRY(host_side_pool().get_pinned(sizeof(pgo::LmScalars), &blk, &P->scal_cap));
    P->scal = static_cast<pgo::LmScalars*>(blk);
    memset(P->scal, 0, sizeof(pgo::LmScalars));
  }
  return PGO_OK;
}

// In-place all-gather of equal segments (rank r owns buf[r*seg, (r+1)*seg)); no-op with a single rank.
int exchange(pgo_problem* P, double* buf, size_t seg_doubles) {
  static const bool force = getenv("PGO_FORCE_EXCHANGE") && getenv("PGO_FORCE_EXCHANGE")[0] == '1';   // exercise the transport at world 1
  if (!P->comm || (P->comm->world <= 1 && !force)) return PGO_OK;
  const char* what = "";
  if (P->comm->all_gather(buf, seg_doubles, P->stream, &what) != 0) return set_error(PGO_ERR_HIP, "all-gather failed: %s", what);
  return PGO_OK;
}

// linearise the owned rows, then make J'J diagonal blocks and J'r of ALL rows available on every rank
int linearize_all(pgo_problem* P) {
  pgo::launch_linearize(P->g, P->stream);
  int rc = exchange(P, P->g.Hdiag, (size_t)36 * P->g.rows_per);
  if (rc) return rc;
  return exchange(P, P->g.grad, (size_t)6 * P->g.rows_per);
}

// LM damping + preconditioner.  6x6 blocks are rebuilt on every rank from the gathered diagonal; cluster blocks need
// the in-cluster off-diagonal blocks, which only the owner holds, so their inverses are exchanged.
int damping_all(pgo_problem* P, double radius, double min_diag, double max_diag, int mode) {
  pgo::launch_damping(P->g, radius, min_diag, max_diag, mode, P->stream);
  if (P->g.cluster > 1) return exchange(P, P->g.Minv, (size_t)36 * P->g.cluster * P->g.rows_per);
  return PGO_OK;
}

// one CG iteration: SpMV on the owned rows, exchange of q (+ p'q partials), replicated vector update
// `refresh`: this is a residual_reset_period-th iteration — r is recomputed as b - A x (Ceres conjugate_gradients_solver.cc)
// instead of updated: x-only update, A x into the exchange buffer, then r / z / partial sums.
int cg_iteration(pgo_problem* P, const pgo::DeviceGraph& g, const pgo::CgParams& prm, int odd, bool refresh) {
  pgo::launch_pcg_spmv_only(g, prm, odd, P->stream);
  int rc = exchange(P, g.cg_q, (size_t)g.seg);
  if (rc) return rc;
  if (!refresh) {
    pgo::launch_pcg_update_only(g, odd, P->stream);
    return PGO_OK;
  }
  if (g.world == 1) {   // x = x_old + alpha p formed on the fly by the SpMV, one combined vector launch
    pgo::launch_spmv_refresh(g, P->stream, 1, odd);
    pgo::launch_pcg_update_only(g, odd, P->stream, 3);
    return PGO_OK;
  }
  pgo::launch_pcg_update_only(g, odd, P->stream, 1);
  pgo::launch_spmv_refresh(g, P->stream);
  rc = exchange(P, g.cg_q, (size_t)g.seg);
  if (rc) return rc;
  pgo::launch_pcg_update_only(g, odd, P->stream, 2);
  return PGO_OK;
}
int cg_iteration(pgo_problem* P, const pgo::CgParams& prm, int odd, bool refresh) { return cg_iteration(P, P->g, prm, odd, refresh); }

// Splits [0, n) into contiguous ranges over the host worker pool (topology build of large graphs; nothing on the LM path).
template <class F>
void parallel_for(int n, F&& fn) {
  const int nt = n < 8192 ? 1 : std::min(pgo::HostPool::get().width(), n / 4096);   // handing a range to a pool worker costs a few us; 4 k items of these loops ~0.25 ms
  if (nt <= 1) { fn(0, n); return; }
  pgo::HostPool::get().run(nt, [&](int i) {
    const int lo = (int)((long long)n * i / nt), hi = (int)((long long)n * (i + 1) / nt);
    fn(lo, hi);
  });
}

int choose_block(long long total_slots) {
  static const int env_block = getenv("PGO_BLOCK") ? atoi(getenv("PGO_BLOCK")) : 0;   // tuning experiments: 64, 128 or 256
  if (env_block == 64 || env_block == 128 || env_block == 256) return env_block;
  if (total_slots >= 256LL * 512) return 256;
  if (total_slots >= 128LL * 384) return 128;
  return 64;
}

int decide_direct_host(pgo_problem* P, int N, int E, int n_slots, long long front_budget);
long long front_memory_budget();

// Builds the incidence-slot topology and uploads every static array.
int prepare(pgo_problem* P) {
  int rc = ensure_device(P);
  if (rc) return rc;
  if (!P->topo_dirty) return PGO_OK;
  if (P->analysis_thread.joinable()) P->analysis_thread.join();     // (of a topology that is being replaced)
  const auto t0 = Clock::now();
  const int N = (int)P->pp.size(), E = (int)P->ia.size();
  if (N == 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "problem has no poses");
  hipStream_t s = P->stream;
  P->drop_graph();

  const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto lap = [&, tl = Clock::now()](const char* what) mutable {
    if (verbose) std::fprintf(stderr, "[pgo] prepare: %-28s %.2f ms\n", what, 1e3 * seconds_since(tl));
    tl = Clock::now();
  };
  std::vector<int> deg(N, 0);
  for (int e = 0; e < E; ++e) { ++deg[P->ia[e]]; ++deg[P->ib[e]]; }
  long long total = 0;
  for (int v = 0; v < N; ++v) total += 1 + deg[v];

  // ---- row ownership: rank r owns the poses [r*rows_per, (r+1)*rows_per); cut edges are evaluated by
  // the owners of both endpoints.  rows_per is a multiple of 4 so that preconditioner clusters never straddle ranks.
  const int world = P->comm ? P->comm->world : 1, rank = P->comm ? P->comm->rank : 0;
  long long rl = 0, rh = 0;
  int rows_per = 0;
  if (pgo_row_shard_range(N, rank, world, &rl, &rh, &rows_per) != PGO_OK) return PGO_ERR_INVALID_ARGUMENT;   // THE ownership rule
  const int row_lo = (int)rl, row_hi = (int)rh;
  const int NP = world * rows_per;   // padded pose count of every replicated / exchanged array
  const int B = choose_block(total / world);

  // rows -> workgroups (greedy packing of `block` slots; a row with more incidences gets its own multi-chunk group)
  std::vector<int> wg_row_begin, wg_slot_begin, row_slot_begin(N, 0), row_slot_cnt(N, 0);
  long long slot = 0;
  auto pack = [&](int lo, int hi, bool record) -> int {
    long long sl = 0;
    int cur = 0, n = 0;
    if (record) { wg_row_begin.assign(1, lo); wg_slot_begin.assign(1, 0); }
    auto close_wg = [&](int next_row) {
      sl = (sl + B - 1) / B * B;
      ++n;
      if (record) { wg_row_begin.push_back(next_row); wg_slot_begin.push_back((int)sl); }
      cur = 0;
    };
    for (int v = lo; v < hi; ++v) {
      const int c = 1 + deg[v];
      if (c > B) {
        if (cur > 0) close_wg(v);
        if (record) { row_slot_begin[v] = (int)sl; row_slot_cnt[v] = c; }
        sl += c;
        close_wg(v + 1);
        continue;
      }
      if (cur + c > B) close_wg(v);
      if (record) { row_slot_begin[v] = (int)sl; row_slot_cnt[v] = c; }
      sl += c;
      cur += c;
    }
    if (cur > 0) close_wg(hi);
    if (n == 0) { sl += B; close_wg(hi); }   // a rank without rows still launches one (empty) workgroup
    if (record) slot = sl;
    return n;
  };
  int pq_cap = 1;
  for (int r = 0; r < world; ++r) pq_cap = std::max(pq_cap, pack(std::min(N, r * rows_per), std::min(N, (r + 1) * rows_per), false));
  const int n_wg = pack(row_lo, row_hi, true);
  if (slot > 0x7fffffffLL - 1024) return set_error(PGO_ERR_UNSUPPORTED, "graph too large for 32-bit slot indices");
  const int n_slots = (int)slot;
  const int seg = rows_per * 6 + pq_cap;

  // slots: diagonal first, then the row's incidences in edge order (owned rows only)
  std::vector<int> slot_col(n_slots, -1), slot_row(n_slots, 0), slot_edge(n_slots, -1), fill(N, 0);
  std::vector<uint8_t> slot_side(n_slots, pgo::SIDE_PAD);
  for (int v = row_lo; v < row_hi; ++v) {
    const int sb = row_slot_begin[v];
    slot_col[sb] = v; slot_row[sb] = v; slot_side[sb] = pgo::SIDE_DIAG;
    fill[v] = sb + 1;
  }
  P->edge_begin_slot.assign(E, -1);
  for (int e = 0; e < E; ++e) {
    const int a = P->ia[e], b = P->ib[e];
    if (a >= row_lo && a < row_hi) {
      const int t = fill[a]++;
      slot_col[t] = b; slot_row[t] = a; slot_side[t] = pgo::SIDE_BEGIN; slot_edge[t] = e;
      P->edge_begin_slot[e] = t;
    }
    if (b >= row_lo && b < row_hi) {
      const int t = fill[b]++;
      slot_col[t] = a; slot_row[t] = b; slot_side[t] = pgo::SIDE_END; slot_edge[t] = e;
    }
  }
  // pad slots keep a valid row index so that loads stay in range
  for (int w = 0; w < n_wg; ++w) {
    const int r = std::max(0, std::min(wg_row_begin[w], N - 1));
    for (int t = wg_slot_begin[w]; t < wg_slot_begin[w + 1]; ++t) if (slot_side[t] == pgo::SIDE_PAD) slot_row[t] = r;
  }

  lap("rows -> workgroups, slots");
  P->h_slot_row = slot_row; P->h_slot_col = slot_col; P->h_slot_side = slot_side; P->h_row_slot_begin = row_slot_begin;
  P->direct_analyzed = false; P->direct_usable = false; P->front_usable = false; P->sfront_usable = false; P->cluster_built = 0; P->g.cluster = 1;
  if (P->want_direct && !(getenv("PGO_NO_ANALYSIS_THREAD") && getenv("PGO_NO_ANALYSIS_THREAD")[0] == '1')) {
    // an exact request: its host analysis (ordering, symbolic factorisation, schedule) needs nothing but the slot topology
    const long long budget = front_memory_budget();
    const int ns = (int)n_slots;
    P->analysis_kind = 0;
    P->analysis_thread = std::thread([P, N, E, ns, budget]() { P->analysis_kind = decide_direct_host(P, N, E, ns, budget); });
  }
  // measurements: edge order and slot order, component major.  The gathers into slot order are cache-hostile (21-36
  // strided streams indexed by a random edge): split over host threads, each owning a contiguous range.
  HostArray emeas, smeas, eW, sW, eL;
  emeas.resize((size_t)7 * E);
  smeas.resize((size_t)7 * n_slots);
  parallel_for(E, [&](int lo, int hi) {
    for (int e = lo; e < hi; ++e) for (int c = 0; c < 7; ++c) emeas[(size_t)c * E + e] = P->meas[(size_t)7 * e + c];
  });
  parallel_for(n_slots, [&](int lo, int hi) {
    for (int t = lo; t < hi; ++t) {
      const int e = slot_edge[t];
      for (int c = 0; c < 7; ++c) smeas[(size_t)c * n_slots + t] = e < 0 ? (c == 6 ? 1.0 : 0.0) : P->meas[(size_t)7 * e + c];
    }
  });
  std::atomic<int> w_has_pr(0), w_has_offdiag(0);
  if (P->has_info) {
    eW.resize((size_t)21 * E); eL.resize((size_t)36 * E); sW.resize((size_t)21 * n_slots);
    parallel_for(E, [&](int lo, int hi) {
      for (int e = lo; e < hi; ++e) {
        const double* L = &P->sqrt_info[(size_t)36 * e];
        int k = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = i; j < 6; ++j) {
            double w = 0;
            for (int r = 0; r < 6; ++r) w += L[6 * r + i] * L[6 * r + j];  // W = L^T L
            if (i < 3 && j >= 3 && w != 0.0) w_has_pr.store(1, std::memory_order_relaxed);
            if (i != j && w != 0.0) w_has_offdiag.store(1, std::memory_order_relaxed);
            eW[(size_t)k * E + e] = w;
            ++k;
          }
        for (int q = 0; q < 36; ++q) eL[(size_t)q * E + e] = L[q];
      }
    });
    parallel_for(n_slots, [&](int lo, int hi) {
      for (int t = lo; t < hi; ++t) {
        const int e = slot_edge[t];
        if (e < 0) { for (int k = 0; k < 21; ++k) sW[(size_t)k * n_slots + t] = 0.0; continue; }
        // W of the slot's edge, recomputed from L (contiguous 288 B) instead of 21 strided reads of eW
        const double* L = &P->sqrt_info[(size_t)36 * e];
        int k = 0;
        for (int i = 0; i < 6; ++i)
          for (int j = i; j < 6; ++j) {
            double w = 0;
            for (int r = 0; r < 6; ++r) w += L[6 * r + i] * L[6 * r + j];
            sW[(size_t)k * n_slots + t] = w;
            ++k;
          }
      }
    });
  }
  const bool w_blockdiag = P->has_info && w_has_pr.load() == 0;
  const bool w_diag = P->has_info && w_has_offdiag.load() == 0;      // W = diag(w) (the generators' diag(1/sigma^2)): six planes of sW are read
  lap("measurement / W arrays");
  UploadScope upload_scope(s);     // the copies below are enqueued side by side; this function's final synchronisation is their wait
  HIP_TRY(P->d_slot_col.upload(slot_col, s));
  HIP_TRY(P->d_slot_row.upload(slot_row, s));
  HIP_TRY(P->d_slot_side.upload(slot_side, s));
  HIP_TRY(P->d_wg_slot_begin.upload(wg_slot_begin, s));
  HIP_TRY(P->d_wg_row_begin.upload(wg_row_begin, s));
  HIP_TRY(P->d_row_slot_begin.upload(row_slot_begin, s));
  HIP_TRY(P->d_row_slot_cnt.upload(row_slot_cnt, s));
  HIP_TRY(P->d_cmask.upload(P->cmask, s));
  HIP_TRY(P->d_edge_a.upload(P->ia, s));
  HIP_TRY(P->d_edge_b.upload(P->ib, s));
  HIP_TRY(P->d_smeas.upload(smeas.data(), smeas.n, s));
  HIP_TRY(P->d_emeas.upload(emeas.data(), emeas.n, s));
  HIP_TRY(P->d_sW.upload(sW.data(), sW.n, s));
  HIP_TRY(P->d_eW.upload(eW.data(), eW.n, s));
  HIP_TRY(P->d_eL.upload(eL.data(), eL.n, s));
  // cluster-preconditioner lists (prepare_clusters fills them): allocated here, at their upper bounds, because on this
  // stack an upload into a buffer allocated AFTER the large allocations below takes 6-25 ms to complete
  HIP_TRY(P->d_cl_ptr.alloc((size_t)N + 2));
  HIP_TRY(P->d_cl_slot.alloc((size_t)E + 1));
  HIP_TRY(P->d_cl_rc.alloc((size_t)E + 1));

  lap("uploads");
  const size_t m = (size_t)6 * NP;
  HIP_TRY(P->d_pose_x.alloc((size_t)pgo::POSE_STRIDE * N));
  HIP_TRY(P->d_pose_c.alloc((size_t)pgo::POSE_STRIDE * N));
  HIP_TRY(P->d_pose_0.alloc((size_t)pgo::POSE_STRIDE * N));
  HIP_TRY(P->d_bsr.alloc((size_t)n_slots * 36));
  HIP_TRY(P->d_bsr.zero(s));
  P->spec_ready = false;     // the spare linearisation set follows the new sizes when it is next needed
  HIP_TRY(P->d_Hdiag.alloc((size_t)36 * NP));
  HIP_TRY(P->d_Hdiag.zero(s));
  HIP_TRY(P->d_Minv.alloc((size_t)36 * NP * 4 + (size_t)world * 144 * 4));   // room for 4-pose clusters of every rank, padded
  HIP_TRY(P->d_Minv.zero(s));
  DevBuf<double>* vecs[] = {&P->d_grad, &P->d_scale, &P->d_d2, &P->d_diagc, &P->d_cg_b, &P->d_cg_x, &P->d_cg_r,
                            &P->d_cg_z, &P->d_cg_p0, &P->d_cg_p1, &P->d_delta};
  for (DevBuf<double>* b : vecs) { HIP_TRY(b->alloc(m)); HIP_TRY(b->zero(s)); }
  HIP_TRY(P->d_cg_q.alloc((size_t)world * seg));   // exchange buffer: q segments + p'q partials (unused partial slots stay 0)
  HIP_TRY(P->d_cg_q.zero(s));
  const int n_vec_wg = std::max(1, std::min((int)(((size_t)6 * N + pgo::vec_block() - 1) / pgo::vec_block()), 256));
  const int n_edge_wg = std::max(1, (E + pgo::edge_block() - 1) / pgo::edge_block());
  const int n_pose_wg = (N + pgo::pose_block() - 1) / pgo::pose_block();
  const int n_part = std::max(std::max(n_wg, n_vec_wg), n_edge_wg + n_pose_wg);   // the fused step tail runs n_edge_wg + n_pose_wg workgroups
  // (blocks may come from the pool with a previous problem's contents: everything that is not fully written before it is read
  // is cleared here)
  HIP_TRY(P->d_part_rz.alloc((size_t)2 * n_part));
  HIP_TRY(P->d_part_rz.zero(s));
  HIP_TRY(P->d_part_q.alloc((size_t)2 * n_part));
  HIP_TRY(P->d_part_q.zero(s));
  HIP_TRY(P->d_part_rr.alloc((size_t)2 * n_part));
  HIP_TRY(P->d_part_rr.zero(s));
  HIP_TRY(P->d_part_bb.alloc((size_t)n_part));
  HIP_TRY(P->d_part_bb.zero(s));
  HIP_TRY(P->d_part_misc.alloc((size_t)8 * n_part));
  HIP_TRY(P->d_part_misc.zero(s));
  HIP_TRY(P->d_cg.alloc(1));
  HIP_TRY(P->d_cg.zero(s));
  HIP_TRY(P->d_flags.alloc(4));
  HIP_TRY(P->d_flags.zero(s));

  pgo::DeviceGraph& g = P->g;
  g.N = N; g.E = E; g.n_wg = n_wg; g.n_slots = n_slots; g.block = B;
  g.world = world; g.rank = rank; g.rows_per = rows_per; g.row_lo = row_lo; g.row_hi = row_hi; g.pq_cap = pq_cap; g.seg = seg;
  // Several ranks: RCCL collectives are enqueued eagerly by default (capturing them into the CG batch graph is only
  // validated at world size 1 on the development box; PGO_COMM_GRAPH=1 opts in).  The per-iteration cost is then
  // dominated by the all-gather latency, not by launch overhead.
  if (P->comm && (!P->comm->capturable() || (world > 1 && !(getenv("PGO_COMM_GRAPH") && getenv("PGO_COMM_GRAPH")[0] == '1')))) P->use_graph = false;
  // 0 identity, 1 general, 2 block-diagonal W (every W_pr entry exactly zero: diag(1/sigma^2) and the like); 0 and 2 use the packed
  // 27-entry slots.  PGO_BLK_FULL=1 keeps the general kernels and the full layout (A/B measurements).
  {
    const char* full = getenv("PGO_BLK_FULL");
    const bool force_full = full && full[0] == '1';
    static const bool no_diag = getenv("PGO_NO_DIAG_INFO") && getenv("PGO_NO_DIAG_INFO")[0] == '1';    // (A/B: the 12-entry reads of mode 2)
    g.info_mode = !P->has_info ? 0 : (w_diag && !force_full && !no_diag) ? 3 : (w_blockdiag && !force_full) ? 2 : 1;
    g.blk_packed = (g.info_mode != 1 && !force_full) ? 1 : 0;
  }
  g.loss_kind = P->loss_kind; g.loss_a = P->loss_a;
  g.slot_col = P->d_slot_col.p; g.slot_row = P->d_slot_row.p; g.slot_side = P->d_slot_side.p;
  g.wg_slot_begin = P->d_wg_slot_begin.p; g.wg_row_begin = P->d_wg_row_begin.p;
  g.row_slot_begin = P->d_row_slot_begin.p; g.row_slot_cnt = P->d_row_slot_cnt.p; g.cmask = P->d_cmask.p;
  g.smeas = P->d_smeas.p; g.sW = P->d_sW.p; g.edge_a = P->d_edge_a.p; g.edge_b = P->d_edge_b.p;
  g.emeas = P->d_emeas.p; g.eW = P->d_eW.p; g.eL = P->d_eL.p;
  g.pose_x = P->d_pose_x.p; g.pose_c = P->d_pose_c.p; g.bsr_val = P->d_bsr.p; g.Hdiag = P->d_Hdiag.p;
  g.Minv = P->d_Minv.p; g.grad = P->d_grad.p; g.scale = P->d_scale.p; g.d2 = P->d_d2.p;
  g.diag_clamped = P->d_diagc.p; g.cg_b = P->d_cg_b.p; g.cg_x = P->d_cg_x.p; g.cg_r = P->d_cg_r.p;
  g.cg_z = P->d_cg_z.p; g.cg_q = P->d_cg_q.p; g.cg_p0 = P->d_cg_p0.p; g.cg_p1 = P->d_cg_p1.p;
  g.delta = P->d_delta.p; g.part_rz = P->d_part_rz.p; g.part_q = P->d_part_q.p;
  g.part_rr = P->d_part_rr.p; g.part_bb = P->d_part_bb.p; g.part_misc = P->d_part_misc.p;
  g.n_part = n_part; g.n_vec_wg = n_vec_wg; g.n_edge_wg = n_edge_wg; g.n_pose_wg = n_pose_wg;
  g.cg = P->d_cg.p; g.flags = P->d_flags.p;
  void* dscal = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dscal, P->scal, 0));
  g.scal = reinterpret_cast<pgo::LmScalars*>(dscal);
  HIP_TRY(hipStreamSynchronize(s));
  lap("device buffers");
  P->topo_dirty = false;
  P->lm.t_setup = seconds_since(t0);
  return PGO_OK;
}

int upload_poses(pgo_problem* P, double* dst) {
  const int N = (int)P->pp.size();
  std::vector<double> h((size_t)pgo::POSE_STRIDE * N, 0.0);
  for (int v = 0; v < N; ++v) {
    double* o = &h[(size_t)pgo::POSE_STRIDE * v];
    o[0] = P->pp[v][0]; o[1] = P->pp[v][1]; o[2] = P->pp[v][2];
    o[3] = P->qq[v][0]; o[4] = P->qq[v][1]; o[5] = P->qq[v][2]; o[6] = P->qq[v][3];
  }
  HIP_TRY(staged_h2d(dst, h.data(), h.size() * sizeof(double), P->stream));
  return PGO_OK;
}

int download_poses(pgo_problem* P, const double* src) {
  const int N = (int)P->pp.size();
  std::vector<double> h((size_t)pgo::POSE_STRIDE * N);
  HIP_TRY(staged_d2h(h.data(), src, h.size() * sizeof(double), P->stream));
  for (int v = 0; v < N; ++v) {
    const double* o = &h[(size_t)pgo::POSE_STRIDE * v];
    // constant blocks are never written (row 0 of the reference's before/after files is identical)
    if (!(P->cmask[v] & 1)) { P->pp[v][0] = o[0]; P->pp[v][1] = o[1]; P->pp[v][2] = o[2]; }
    if (!(P->cmask[v] & 2)) { P->qq[v][0] = o[3]; P->qq[v][1] = o[4]; P->qq[v][2] = o[5]; P->qq[v][3] = o[6]; }
  }
  return PGO_OK;
}

int fill_scale_one(pgo_problem* P) {
  std::vector<double> one((size_t)6 * P->g.N, 1.0);
  HIP_TRY(staged_h2d(P->g.scale, one.data(), one.size() * sizeof(double), P->stream));
  return PGO_OK;
}

// ---- CG driver: batches of iterations, one host check per batch ----
// Host <-> device hand-off (publish_sequence in pgo_kernels.hip): clear the flag, enqueue a sequence whose last kernel
// sets it, spin on the pinned word.  A stream synchronise costs ~20-30 us of wake-up latency per LM iteration phase;
// the spin sees the result ~2 us after the kernel stored it.  Falls back to the blocking call after 50 ms.
inline void arm_handoff(pgo_problem* P) { __atomic_store_n(&P->scal->seq, 0, __ATOMIC_SEQ_CST); }
int wait_handoff(pgo_problem* P) {
  static const bool no_spin = getenv("PGO_NO_SPIN") && getenv("PGO_NO_SPIN")[0] == '1';
  if (!no_spin) {
    const auto t0 = Clock::now();
    for (unsigned spins = 1;; ++spins) {
      if (__atomic_load_n(&P->scal->seq, __ATOMIC_ACQUIRE) != 0) return PGO_OK;
      __builtin_ia32_pause();
      if ((spins & 0x3ff) == 0 && seconds_since(t0) > 0.05) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(P->stream));
  if (__atomic_load_n(&P->scal->seq, __ATOMIC_ACQUIRE) == 0) return set_error(PGO_ERR_HIP, "device hand-off flag was not set by the enqueued sequence");
  return PGO_OK;
}

// The step tail (model cost change, delta, candidate, candidate cost, scalar fold).  `gate`: the kernels run only once
// the device-side CG state says "stopped", so the tail can ride behind every CG batch (no host round trip between the
// last CG iteration and the tail); the scalar fold always hands off to the host.
int enqueue_tail(pgo_problem* P, const pgo::CgParams* finish_prm) {
  hipStream_t s = P->stream;
  const pgo::CgParams none{0.0, -1.0, 0, 0};
  const int gate = finish_prm ? 1 : 0;
  if (P->g.world == 1) {
    // two launches: q = A x + candidate poses (diagonal lanes), then model change / norms / candidate cost / fold
    pgo::launch_spmv_tail(P->g, finish_prm ? *finish_prm : none, s, gate, 1);
    pgo::launch_step_tail(P->g, s, gate);
    return PGO_OK;
  }
  // several ranks: the vector kernels are replicated over all rows, q crosses the wire in between
  pgo::launch_spmv_tail(P->g, finish_prm ? *finish_prm : none, s, gate, 0);
  int rc = exchange(P, P->g.cg_q, (size_t)P->g.seg);
  if (rc) return rc;
  pgo::launch_model_delta_and_retract(P->g, s, gate);
  pgo::launch_cost(P->g, P->g.pose_c, 0, s, gate);
  pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s, gate);
  return PGO_OK;
}

// A batch is a captured hipGraph of `batch` x (SpMV kernel, update kernel) + the finish kernel.  The
// kernels stop by themselves (device-side `done` flag), so an over-long batch only costs early-exit
// launches; the batch length follows the previous solve's iteration count.
// with_tail: the gated step tail follows in the same graph and its scalar fold hands off; otherwise the finish kernel does.
// start_it: absolute index (1-based) of the batch's first CG iteration — decides which iterations refresh the residual.
int launch_cg_batch(pgo_problem* P, const pgo::CgParams& prm, int batch, bool with_tail = false, int start_it = 1) {
  hipStream_t s = P->stream;
  // The refresh r = b - A x belongs to Ceres' truncated CG (Q-tolerance stop).  An exact request served by PCG runs to a
  // 1e-13 relative residual, below what a recomputed residual can show in FP64 on an ill-conditioned chain: with the
  // refresh the test would never fire (measured: 27x the iterations on sphere x10), so that mode keeps the recurrence.
  static const bool exact_refresh = getenv("PGO_EXACT_REFRESH") != nullptr;   // experiment switch
  const int period = (prm.q_tolerance < 0.0 && !exact_refresh) ? 0 : P->opt.cg_residual_reset_period;
  auto refresh_at = [&](int i) { return period > 0 && ((start_it + i) % period) == 0; };
  if (P->use_graph) {
    if (memcmp(&P->cg_graph_params, &prm, sizeof prm) != 0) { P->drop_graph(); P->cg_graph_params = prm; }
    // the captured kernels hold the DeviceGraph by value; the tail touches the pose ping-pong, so the key carries its parity
    // ... and the positions of the residual refreshes depend on the start index modulo the period
    const int key = (4 * batch + (with_tail ? 2 : 0) + ((with_tail && P->g.pose_x != P->d_pose_x.p) ? 1 : 0)) * 64 +
                    (period > 0 ? start_it % period : 0);
    auto it = P->cg_graphs.find(key);
    if (it == P->cg_graphs.end()) {
      pgo_problem::CapturedBatch cb;
      hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        int rc_it = PGO_OK;
        for (int i = 0; i < batch && rc_it == PGO_OK; ++i) rc_it = cg_iteration(P, prm, (i & 1) ^ 1, refresh_at(i));
        if (!with_tail) pgo::launch_pcg_finish(P->g, prm, s, 1);
        else if (rc_it == PGO_OK) rc_it = enqueue_tail(P, &prm);
        e = hipStreamEndCapture(s, &cb.graph);
        if (e == hipSuccess && rc_it != PGO_OK) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipGraphInstantiate(&cb.exec, cb.graph, nullptr, nullptr, 0);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (cb.exec) (void)hipGraphExecDestroy(cb.exec);
        if (cb.graph) (void)hipGraphDestroy(cb.graph);
        P->drop_graph();
        P->use_graph = false;  // fall back to plain stream launches (same kernels)
      } else {
        it = P->cg_graphs.emplace(key, cb).first;
      }
    }
    if (P->use_graph) {
      HIP_TRY(hipGraphLaunch(it->second.exec, s));
      return PGO_OK;
    }
  }
  for (int i = 0; i < batch; ++i) { int rc = cg_iteration(P, prm, (i & 1) ^ 1, refresh_at(i)); if (rc) return rc; }
  if (with_tail) return enqueue_tail(P, &prm);
  pgo::launch_pcg_finish(P->g, prm, s, 1);
  return PGO_OK;
}

// Batch schedule.  The kernels stop on their own, but every iteration enqueued past the stopping point still costs two
// early-exit launches (~5 us, ~10 us under the profiler) and every extra batch a host hand-off plus two gated tail
// launches (~12-17 us).  With n iterations expected, the cheapest fixed batch is ~sqrt(3.4 n); n is not known, so the
// first batch follows the previous solve's count (capped at 8: early LM iterations are poor predictors, late ones need
// 3-6 iterations) and later batches grow like sqrt(3.4 * iterations already enqueued) — not by doubling, which wastes up
// to half of the last batch.  Sizes are quantised so that only a handful of graphs is ever captured.
int pick_batch(const pgo::CgParams& prm, int user_batch, int round, int enqueued, int last_iterations) {
  static const int env_b0 = getenv("PGO_CG_BATCH0") ? atoi(getenv("PGO_CG_BATCH0")) : 0;
  static const int env_double = getenv("PGO_CG_DOUBLING") ? atoi(getenv("PGO_CG_DOUBLING")) : 0;   // the r01 schedule 6,12,24,48,64
  static const double env_c = getenv("PGO_CG_SQRTC") ? atof(getenv("PGO_CG_SQRTC")) : 3.4;
  static const int env_cap0 = getenv("PGO_CG_CAP0") ? atoi(getenv("PGO_CG_CAP0")) : 8;
  static const int sizes[] = {2, 4, 6, 8, 12, 16, 24, 32, 48, 64};
  int batch;
  if (user_batch > 0) {
    batch = user_batch;
  } else if (env_double) {
    batch = std::min(64, (env_b0 > 0 ? env_b0 : 6) << std::min(round, 4));
  } else {
    double target;
    if (round == 0) target = env_b0 > 0 ? env_b0 : (last_iterations > 0 ? std::min(last_iterations, env_cap0) : 6);
    else target = std::sqrt(env_c * std::max(1, enqueued));
    batch = 64;
    for (int sz : sizes) if (sz >= target) { batch = sz; break; }
  }
  batch = std::max(1, std::min(batch, prm.max_iterations));
  return (batch + 1) & ~1;  // even: every batch starts at an odd iteration (kernels take the parity at launch)
}

int run_pcg(pgo_problem* P, const pgo::CgParams& prm, int batch, int* iterations, int* status) {
  hipStream_t s = P->stream;
  pgo::launch_pcg_init(P->g, s);
  for (int round = 0, enqueued = 0;; ++round) {
    arm_handoff(P);
    const int nb = pick_batch(prm, batch, round, enqueued, 0);
    int rc = launch_cg_batch(P, prm, nb, false, enqueued + 1);
    enqueued += nb;
    if (rc) return rc;
    rc = wait_handoff(P);
    if (rc) return rc;
    if (P->scal->cg_status != -1) break;
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  *iterations = P->scal->cg_iterations;
  *status = P->scal->cg_status;
  P->last_cg_iterations = *iterations;
  return PGO_OK;
}

// ---- cluster-Jacobi preconditioner: which BSR slots fall inside a cluster of CL consecutive poses ----
int prepare_clusters(pgo_problem* P, int CL) {
  if (CL != 2 && CL != 4) CL = 1;
  if (P->cluster_built == CL) { P->g.cluster = CL; return PGO_OK; }
  P->drop_graph();  // captured CG batches hold the DeviceGraph by value
  if (CL > 1) {
    // clusters of the rows this rank owns (row_lo is a multiple of 4); indices local to the rank
    const int c0 = P->g.row_lo / CL;
    const int ncl = std::max(1, (P->g.row_hi - P->g.row_lo + CL - 1) / CL);
    // the BEGIN slot of every edge whose two poses share a cluster (the END twin holds the transposed block: the kernel mirrors)
    std::vector<int> ptr(ncl + 1, 0), slots;
    std::vector<uint8_t> rcs;
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (int t = 0; t < P->g.n_slots; ++t) {
        if (P->h_slot_side[t] != pgo::SIDE_BEGIN) continue;
        const int r = P->h_slot_row[t], c = P->h_slot_col[t];
        if (r / CL != c / CL) continue;
        if (pass == 0) ++ptr[r / CL - c0 + 1];
        else { const int q = fill[r / CL - c0]++; slots[q] = t; rcs[q] = (uint8_t)(((r % CL) << 4) | (c % CL)); }
      }
      if (pass == 0) { for (int k = 0; k < ncl; ++k) ptr[k + 1] += ptr[k]; slots.resize(ptr[ncl]); rcs.resize(ptr[ncl]); }
    }
    HIP_TRY(P->d_cl_ptr.store(ptr, P->stream));
    HIP_TRY(P->d_cl_slot.store(slots, P->stream));
    HIP_TRY(P->d_cl_rc.store(rcs, P->stream));
    const size_t need = (size_t)P->g.world * P->g.rows_per * 36 * CL;   // every rank's clusters, padded
    if (P->d_Minv.n < need) { HIP_TRY(P->d_Minv.alloc(need)); HIP_TRY(P->d_Minv.zero(P->stream)); }
    P->g.Minv = P->d_Minv.p;
    P->g.cl_ptr = P->d_cl_ptr.p;
    P->g.cl_slot = P->d_cl_slot.p;
    P->g.cl_rc = P->d_cl_rc.p;
  }
  P->g.cluster = CL;
  P->cluster_built = CL;
  return PGO_OK;
}

// ---- exact solver: GPU block-sparse Cholesky (pgo_direct.*) ----
// Multifrontal solver: host analysis (front_analyzed_ok) and, once chosen, plan upload (front_usable).
long long front_memory_budget() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)64 << 30; }
  const char* cap = getenv("PGO_FRONT_MAX_GB");
  return cap ? (long long)(atof(cap) * 1e9) : (long long)(0.6 * (double)free_b);
}
// host only (runs on the analysis thread)
void analyze_front(pgo_problem* P, int N, int n_slots, long long budget, bool* ok, int small_max = 0) {
  pgo::FrontSymbolic& S = P->fsym;
  const auto t_an = Clock::now();
  *ok = pgo::front_analyze(N, P->ia, P->ib, n_slots, P->h_slot_row, P->h_slot_col, P->h_slot_side, budget, &S, small_max);
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] front: symbolic analysis %.2f ms (%s)\n", 1e3 * seconds_since(t_an), *ok ? "usable" : "declined");
}

int upload_front(pgo_problem* P) {
  pgo::FrontSymbolic& S = P->fsym;
  const auto t_up = Clock::now();
  hipStream_t s = P->stream;
  UploadScope upload_scope(s);
  HIP_TRY(P->df_perm.upload(S.perm, s));
  HIP_TRY(P->df_idx.upload(S.idx, s));
  HIP_TRY(P->df_child.upload(S.child, s));
  HIP_TRY(P->df_rel.upload(S.rel, s));
  HIP_TRY(P->df_cstart.upload(S.cstart, s));
  HIP_TRY(P->df_wg_job.upload(S.wg_job, s));
  HIP_TRY(P->df_wg_tile.upload(S.wg_tile, s));
  HIP_TRY(P->df_bwd_front.upload(S.bwd_front, s));
  HIP_TRY(P->df_bwd_chunk.upload(S.bwd_chunk, s));
  HIP_TRY(P->df_bwdb_front.upload(S.bwdb_front, s));
  HIP_TRY(P->df_bwdb_chunk.upload(S.bwdb_chunk, s));
  HIP_TRY(P->df_asm_tile.upload(S.asm_tile, s));
  HIP_TRY(P->df_asm_contrib.upload(S.asm_contrib, s));
  HIP_TRY(P->df_col_front.upload(S.col_front, s));
  HIP_TRY(P->df_ablk_ptr.upload(S.ablk_ptr, s));
  HIP_TRY(P->df_ablk_slot.upload(S.ablk_slot, s));
  HIP_TRY(P->df_ablk_front.upload(S.ablk_front, s));
  HIP_TRY(P->df_ablk_pos.upload(S.ablk_pos, s));
  HIP_TRY(P->df_fronts.upload(S.fronts, s));
  HIP_TRY(P->df_jobs.upload(S.jobs, s));
  HIP_TRY(P->df_Fval.alloc((size_t)S.fval_size));
  HIP_TRY(P->df_Winv.alloc((size_t)S.winv_size));
  HIP_TRY(P->df_x.alloc((size_t)6 * S.n));
  HIP_TRY(P->df_x.zero(s));
  pgo::FrontPlan& f = P->fplan;
  f.n = S.n; f.nf = S.nf;
  f.perm = P->df_perm.p; f.fronts = P->df_fronts.p; f.idx = P->df_idx.p; f.child = P->df_child.p; f.rel = P->df_rel.p; f.cstart = P->df_cstart.p;
  f.wg_job = P->df_wg_job.p; f.wg_tile = P->df_wg_tile.p; f.bwd_front = P->df_bwd_front.p; f.bwd_chunk = P->df_bwd_chunk.p;
  f.bwdb_front = P->df_bwdb_front.p; f.bwdb_chunk = P->df_bwdb_chunk.p;
  f.asm_tile = P->df_asm_tile.p; f.asm_contrib = P->df_asm_contrib.p;
  f.col_front = P->df_col_front.p; f.ablk_ptr = P->df_ablk_ptr.p; f.ablk_slot = P->df_ablk_slot.p;
  f.ablk_front = P->df_ablk_front.p; f.ablk_pos = P->df_ablk_pos.p; f.n_ablk = (int)S.ablk_front.size();
  f.jobs = P->df_jobs.p; f.Fval = P->df_Fval.p; f.Winv = P->df_Winv.p; f.x = P->df_x.p;
  // the stages of the single-launch form (pgo_front.h FrontStages); PGO_FRONT_FUSED=0: one launch per phase of a round
  f.st_table = nullptr; f.st_pred_ptr = nullptr; f.st_pred = nullptr; f.st_need = nullptr; f.st_count = nullptr;
  P->front_epoch = 0; P->front_tickets = 0;
  {
    // Measured (factorisation, single launch vs one launch per phase of a round): Manhattan 2 k / 8 k 1.09 vs 1.23 ms, KITTI-00
    // dense (0.9 GFLOP) 28.1 vs 32.0 ms per 14-iteration solve, Manhattan 10 k (4.5 GFLOP) 3.54 vs 3.57 ms, sphere x10 (383
    // GFLOP) 46 vs 28 ms: every work-group pays a cache write-back and an invalidation of its XCD's L2 where a kernel boundary
    // pays them once, which the GEMM-heavy factorisations cannot afford.  Default: single launch up to 3 GFLOP
    // (PGO_FRONT_FUSED=1 always, =0 never; PGO_FRONT_FUSED_GFLOP moves the limit).
    const char* fu = getenv("PGO_FRONT_FUSED");
    const double limit = getenv("PGO_FRONT_FUSED_GFLOP") ? atof(getenv("PGO_FRONT_FUSED_GFLOP")) : 3.0;
    const bool on = fu ? fu[0] == '1' : S.flops <= limit * 1e9;
    P->front_launches = !on || S.st_table.empty();
  }
  if (!S.st_table.empty()) {
    HIP_TRY(P->df_st_table.upload(S.st_table, s));
    HIP_TRY(P->df_st_pred_ptr.upload(S.st_pred_ptr, s));
    HIP_TRY(P->df_st_pred.upload(S.st_pred, s));
    if (S.st_pred.empty()) HIP_TRY(P->df_st_pred.alloc(1));
    HIP_TRY(P->df_st_need.upload(S.st_need, s));
    HIP_TRY(P->df_st_count.alloc(S.st_need.size() + 1));
    HIP_TRY(P->df_st_count.zero(s));
    f.st_table = P->df_st_table.p; f.st_pred_ptr = P->df_st_pred_ptr.p; f.st_pred = P->df_st_pred.p; f.st_need = P->df_st_need.p;
    f.st_count = P->df_st_count.p;
  }
  if (S.mixed) {
    // the small fronts at the bottom of the tree take the small-front kernels (pgo_front.h): their compact arrays
    HIP_TRY(P->ds_sf.upload(S.sfronts, s));
    HIP_TRY(P->ds_urel.upload(S.urel, s));
    HIP_TRY(P->ds_osrc.upload(S.osrc, s));
    HIP_TRY(P->ds_list.upload(S.slevel_front, s));
    HIP_TRY(P->ds_L.alloc((size_t)S.sl_size));
    HIP_TRY(P->ds_U.alloc((size_t)S.su_size));
    HIP_TRY(P->ds_W.alloc((size_t)S.sw_size));
    HIP_TRY(P->ds_upos.alloc((size_t)S.su_size));
    P->splan = pgo::SFrontPlan{P->ds_sf.p, P->ds_list.p, P->ds_urel.p, P->ds_upos.p, P->ds_osrc.p, P->ds_L.p, P->ds_U.p, P->ds_W.p, nullptr};
    pgo::launch_sfront_prepare(P->fplan, P->splan, S, s);
  }
  P->front_usable = true;
  HIP_TRY(upload_scope.finish());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] front: plan upload %.2f ms\n", 1e3 * seconds_since(t_up));
  return PGO_OK;
}

int upload_sfront(pgo_problem* P) {
  pgo::FrontSymbolic& S = P->fsym;
  const auto t_up = Clock::now();
  hipStream_t s = P->stream;
  UploadScope upload_scope(s);
  HIP_TRY(P->df_perm.upload(S.perm, s));
  HIP_TRY(P->df_idx.upload(S.idx, s));
  HIP_TRY(P->df_child.upload(S.child, s));
  HIP_TRY(P->df_rel.upload(S.rel, s));
  HIP_TRY(P->df_ablk_ptr.upload(S.ablk_ptr, s));
  HIP_TRY(P->df_ablk_slot.upload(S.ablk_slot, s));
  HIP_TRY(P->df_ablk_pos.upload(S.ablk_pos, s));
  HIP_TRY(P->df_fronts.upload(S.fronts, s));
  HIP_TRY(P->ds_sf.upload(S.sfronts, s));
  HIP_TRY(P->ds_urel.upload(S.urel, s));
  HIP_TRY(P->ds_osrc.upload(S.osrc, s));
  HIP_TRY(P->ds_L.alloc((size_t)S.sl_size));
  HIP_TRY(P->ds_U.alloc((size_t)S.su_size));
  HIP_TRY(P->ds_W.alloc((size_t)S.sw_size));
  HIP_TRY(P->ds_upos.alloc((size_t)S.su_size));
  HIP_TRY(P->df_x.alloc((size_t)6 * S.n));
  HIP_TRY(P->df_x.zero(s));
  pgo::FrontPlan& f = P->fplan;
  f = pgo::FrontPlan{};
  f.n = S.n; f.nf = S.nf;
  f.perm = P->df_perm.p; f.fronts = P->df_fronts.p; f.idx = P->df_idx.p; f.child = P->df_child.p; f.rel = P->df_rel.p;
  f.ablk_ptr = P->df_ablk_ptr.p; f.ablk_slot = P->df_ablk_slot.p; f.ablk_pos = P->df_ablk_pos.p; f.n_ablk = (int)S.ablk_front.size();
  f.x = P->df_x.p;
  HIP_TRY(P->ds_done.alloc(2 * ((size_t)S.nf + 1)));      // flags of the single-launch forms (factorisation, backward substitution) + their ticket counters
  HIP_TRY(P->ds_done.zero(s));
  P->sfront_epoch = 0;
  P->sfront_tickets = 0;
  {
    const char* fu = getenv("PGO_SFRONT_FUSED");
    P->sfront_levels = fu && fu[0] == '0';
  }
  P->splan = pgo::SFrontPlan{P->ds_sf.p, nullptr, P->ds_urel.p, P->ds_upos.p, P->ds_osrc.p, P->ds_L.p, P->ds_U.p, P->ds_W.p, P->ds_done.p};
  pgo::launch_sfront_prepare(P->fplan, P->splan, S, s);
  P->sfront_usable = true;
  HIP_TRY(upload_scope.finish());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] front: small-front plan upload %.2f ms\n", 1e3 * seconds_since(t_up));
  return PGO_OK;
}

// Which factorisation serves an exact request on this topology: the host analyses and the choice between them.  No HIP calls
// (it runs on the analysis thread); returns 0 none (the iterative path serves the request), 1 enumerated 6x6 pairs (P->dsym),
// 2 MFMA fronts, 3 small fronts (P->fsym).
int decide_direct_host(pgo_problem* P, int N, int E, int n_slots, long long front_budget) {
  const char* off = getenv("PGO_NO_DIRECT");
  if (off && off[0] == '1') return 0;
  if (P->comm && P->comm->world > 1) return 0;   // the factorisation needs every row: sharded runs use PCG to 1e-13
  // Two GPU factorisations serve an exact request (measured, tools/front_vs_direct.py: KITTI-00 replay 0.34 ms with the
  // enumerated 6x6 pairs vs 0.44 ms multifrontal; KITTI-00 dense 2.1 vs 2.8 ms; Manhattan 2 k 8.6 vs 1.6 ms; Manhattan 10 k
  // 6.3 vs 5.6 ms; sphere x10: declined vs 39 ms).  The multifrontal analysis is the cheap one and runs first; chain-like
  // graphs (largest front below PGO_FRONT_MIN scalars, default 192) then go to the enumerated schedule, everything else stays
  // multifrontal.  PGO_FRONT=1: always multifrontal; 0: never.
  const char* fr = getenv("PGO_FRONT");
  const int front_mode = !fr ? -1 : (fr[0] == '1' ? 1 : 0);
  const int front_min = getenv("PGO_FRONT_MIN") ? atoi(getenv("PGO_FRONT_MIN")) : 192;
  pgo::DirectSymbolic& S = P->dsym;
  bool front_ok = false;
  // a trajectory with a few chords (KITTI-00 replay: 1.14 edges per pose) is the enumerated schedule's case: its analysis runs
  // first there and the multifrontal one is skipped (one-shot solves pay every millisecond of host analysis)
  bool pair_first_done = false, usable = false, front_done = false;
  // Chain-like graphs (E < 1.5 N) first try the small-front plan: when every front fits the LDS of one workgroup (<= 96 scalars;
  // KITTI-00 replay: 84) the factorisation is one launch per tree level (pgo_front.h) — KITTI-00 0.41 vs 0.46 ms per LM
  // iteration, 7.1 vs 7.6 ms per solve against the enumerated schedule.  Not for the union of a batched solve (one 58 KB
  // workgroup per front: 37 vs 25 ms for 16 graphs).  PGO_SFRONT=0 never, =1 for any graph whose fronts are small enough.
  const char* sfe = getenv("PGO_SFRONT");
  const int sf_mode = !sfe ? ((!P->no_sfront && (double)E < 1.5 * (double)N) ? 1 : 0) : (sfe[0] == '1' ? 1 : 0);
  if (front_mode != 0 && sf_mode == 1) {
    analyze_front(P, N, n_slots, front_budget, &front_ok, pgo::SFRONT_MAX);
    front_done = true;
    if (front_ok && P->fsym.small) { S = pgo::DirectSymbolic(); return 3; }
  }
  auto pairs = [&]() {
    const auto t_an = Clock::now();
    usable = pgo::direct_analyze(N, P->ia, P->ib, n_slots, P->h_slot_row, P->h_slot_col, P->h_slot_side, P->h_row_slot_begin, &S);
    if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: symbolic analysis %.2f ms\n", 1e3 * seconds_since(t_an));
  };
  if (front_mode < 0 && (double)E < 1.5 * (double)N) { pairs(); pair_first_done = true; }
  if (front_mode != 0 && !(pair_first_done && usable && !S.hybrid)) {
    if (!front_done) analyze_front(P, N, n_slots, front_budget, &front_ok);
    if (front_ok && (front_mode == 1 || P->fsym.max_front > front_min)) { S = pgo::DirectSymbolic(); return 2; }
  }
  if (front_mode != 1 && !pair_first_done) pairs();
  if (front_ok && (!usable || S.hybrid)) { S.hybrid = false; return 2; }
  return usable ? 1 : 0;   // 0: too much fill / too deep for the enumerated schedule: the iterative path serves the request
}

int prepare_direct(pgo_problem* P) {
  if (P->direct_analyzed) return PGO_OK;
  int kind;
  if (P->analysis_thread.joinable()) {
    P->analysis_thread.join();
    kind = P->analysis_kind;
  } else {
    kind = decide_direct_host(P, P->g.N, P->g.E, P->g.n_slots, front_memory_budget());
  }
  P->direct_analyzed = true;
  P->direct_usable = false;
  P->front_usable = false;
  P->sfront_usable = false;
  pgo::DirectSymbolic& S = P->dsym;
  if (kind == 0) return PGO_OK;
  P->direct_usable = true;      // (cleared again if an upload fails: the error is returned)
  if (kind == 3) { const int rc = upload_sfront(P); if (rc) P->direct_usable = false; return rc; }
  if (kind == 2) { const int rc = upload_front(P); if (rc) P->direct_usable = false; return rc; }
  P->direct_usable = false;
  const auto t_up = Clock::now();
  hipStream_t s = P->stream;
  UploadScope upload_scope(s);
  HIP_TRY(P->dd_perm.upload(S.perm, s));
  HIP_TRY(P->dd_col_ptr.upload(S.col_ptr, s));
  HIP_TRY(P->dd_blk_row.upload(S.blk_row, s));
  HIP_TRY(P->dd_asrc_ptr.upload(S.asrc_ptr, s));
  HIP_TRY(P->dd_asrc_slot.upload(S.asrc_slot, s));
  HIP_TRY(P->dd_upd_ptr.upload(S.upd_ptr, s));
  HIP_TRY(P->dd_upd_a.upload(S.upd_a, s));
  HIP_TRY(P->dd_upd_b.upload(S.upd_b, s));
  HIP_TRY(P->dd_level_ptr.upload(S.level_ptr, s));
  HIP_TRY(P->dd_level_cols.upload(S.level_cols, s));
  HIP_TRY(P->dd_rowl_ptr.upload(S.rowl_ptr, s));
  HIP_TRY(P->dd_rowl_blk.upload(S.rowl_blk, s));
  HIP_TRY(P->dd_rowl_col.upload(S.rowl_col, s));
  HIP_TRY(P->dd_split_blk.upload(S.split_blk, s));
  HIP_TRY(P->dd_split_diag.upload(S.split_diag, s));
  HIP_TRY(P->dd_split_sub.upload(S.split_sub, s));
  HIP_TRY(P->dd_split_sub_diag.upload(S.split_sub_diag, s));
  HIP_TRY(P->dd_upd_split.upload(S.upd_split, s));
  HIP_TRY(P->dd_panel_cols.upload(S.panel_cols, s));
  HIP_TRY(P->dd_blk_lpos.upload(S.blk_lpos, s));
  if (S.panel_cols.empty()) HIP_TRY(P->dd_panel_cols.alloc(1));
  HIP_TRY(P->dd_split_dblk.upload(S.split_dblk, s));
  if (S.split_blk.empty()) { HIP_TRY(P->dd_split_blk.alloc(1)); HIP_TRY(P->dd_split_diag.alloc(1)); HIP_TRY(P->dd_split_dblk.alloc(1)); }
  HIP_TRY(P->dd_col_flag.alloc((size_t)S.nb));
  HIP_TRY(P->dd_col_flag.zero(s));
  P->direct_epoch = 0;
  if (S.split_sub.empty()) { HIP_TRY(P->dd_split_sub.alloc(1)); HIP_TRY(P->dd_split_sub_diag.alloc(1)); }
  HIP_TRY(P->dd_Lval.alloc((size_t)36 * S.nb));
  HIP_TRY(P->dd_y.alloc((size_t)6 * S.n));
  HIP_TRY(P->dd_y.zero(s));
  pgo::DirectPlan& d = P->dplan;
  d.n = S.n; d.nb = S.nb; d.n_levels = S.n_levels;
  d.perm = P->dd_perm.p; d.col_ptr = P->dd_col_ptr.p; d.blk_row = P->dd_blk_row.p;
  d.asrc_ptr = P->dd_asrc_ptr.p; d.asrc_slot = P->dd_asrc_slot.p; d.upd_ptr = P->dd_upd_ptr.p;
  d.upd_a = P->dd_upd_a.p; d.upd_b = P->dd_upd_b.p; d.level_ptr = P->dd_level_ptr.p; d.level_cols = P->dd_level_cols.p;
  d.rowl_ptr = P->dd_rowl_ptr.p; d.rowl_blk = P->dd_rowl_blk.p; d.rowl_col = P->dd_rowl_col.p;
  d.Lval = P->dd_Lval.p; d.y = P->dd_y.p; d.split_blk = P->dd_split_blk.p;
  d.split_diag = P->dd_split_diag.p; d.split_sub = P->dd_split_sub.p; d.split_sub_diag = P->dd_split_sub_diag.p;
  d.upd_split = P->dd_upd_split.p; d.panel_cols = P->dd_panel_cols.p; d.blk_lpos = P->dd_blk_lpos.p;
  d.split_dblk = P->dd_split_dblk.p; d.col_flag = P->dd_col_flag.p;
  P->drop_direct_graph();
  P->direct_usable = true;
  HIP_TRY(upload_scope.finish());
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: plan upload %.2f ms\n", 1e3 * seconds_since(t_up));
  return PGO_OK;
}

// the multifrontal factorisation: all stages in one launch (pgo_front.h FrontStages), or one launch per phase of a round
void enqueue_front_factor(pgo_problem* P, const pgo::DeviceGraph& G) {
  hipStream_t s = P->stream;
  if (!P->front_launches) {
    const char* sp_env = getenv("PGO_FRONT_SPINS");
    const int n_tickets = (int)(P->fsym.st_table.size() / 2);
    static const bool want_stamps = getenv("PGO_FRONT_STAMPS") && getenv("PGO_FRONT_STAMPS")[0] == '1';
    if (want_stamps && P->ds_stamps.n == 0 && P->ds_stamps.alloc(3 * (size_t)n_tickets) != hipSuccess) return;
    const pgo::FrontStages fs{++P->front_epoch, P->front_tickets, n_tickets, (int)P->fsym.st_need.size(), sp_env ? atoi(sp_env) : (1 << 20),
                              want_stamps ? P->ds_stamps.p : nullptr};
    P->front_tickets += (unsigned long long)n_tickets;
    pgo::launch_front_factor(G, P->fplan, P->fsym, s, nullptr, &fs);
    if (want_stamps && P->front_epoch == 3) {      // development aid: the chain of stages that ends last, from the last stage back
      (void)hipStreamSynchronize(s);
      const pgo::FrontSymbolic& S = P->fsym;
      std::vector<long long> st(3 * (size_t)n_tickets);
      (void)hipMemcpy(st.data(), P->ds_stamps.p, st.size() * sizeof(long long), hipMemcpyDeviceToHost);
      const int ns = (int)S.st_need.size();
      std::vector<long long> first(ns, (long long)1 << 62), ready(ns, 0), done(ns, 0);
      std::vector<int> kind(ns, 0), wg1(ns, 0);
      long long t0 = (long long)1 << 62;
      for (int t = 0; t < n_tickets; ++t) {
        const int sg = S.st_table[2 * (size_t)t + 1];
        first[sg] = std::min(first[sg], st[3 * (size_t)t]); ready[sg] = std::max(ready[sg], st[3 * (size_t)t + 1]); done[sg] = std::max(done[sg], st[3 * (size_t)t + 2]);
        kind[sg] = S.st_table[2 * (size_t)t] & 3; wg1[sg] = S.st_table[2 * (size_t)t] >> 2;
        t0 = std::min(t0, st[3 * (size_t)t]);
      }
      auto us = [&](long long t) { return (double)(t - t0) / 100.0; };
      int sg = 0;
      for (int q = 0; q < ns; ++q) if (done[q] > done[sg]) sg = q;
      std::fprintf(stderr, "[pgo] front stamps: %d tickets, %d stages; chain from the last stage back: stage kind(0 asm 1 panel 2 gemm64 3 gemm32) front wgs | first start, last ready, last done (us)\n", n_tickets, ns);
      for (int hops = 0; hops < 400 && sg >= 0; ++hops) {
        const int front = kind[sg] == 0 ? S.asm_tile[8 * (size_t)wg1[sg]] : S.job_front[S.wg_job[wg1[sg]]];
        std::fprintf(stderr, "[pgo]   %5d %d front %4d (c %3d r %3d) wgs %4d | %8.2f %8.2f %8.2f\n", sg, kind[sg], front, S.fronts[front].c, S.fronts[front].r, S.st_need[sg], us(first[sg]), us(ready[sg]), us(done[sg]));
        int best = -1;
        for (int q = S.st_pred_ptr[sg]; q < S.st_pred_ptr[sg + 1]; ++q) if (best < 0 || done[S.st_pred[q]] > done[best]) best = S.st_pred[q];
        sg = best;
      }
    }
  } else {
    pgo::launch_front_factor(G, P->fplan, P->fsym, s, P->fsym.mixed ? &P->splan : nullptr);
  }
}

// factorise (H~ + D^2) and solve for cg_x = (H~ + D^2)^-1 S g; the launch sequence is static -> one hipGraph
// G: P->g, or its copy that carries the device-resident LM state (the kernels then gate themselves on its halt word)
int run_direct(pgo_problem* P, const pgo::DeviceGraph& G) {
  hipStream_t s = P->stream;
  if (P->sfront_usable) {
    if (!P->sfront_levels) {
      // all levels in one launch (SFrontSync): a parent waits for its children's flags instead of for the end of their launch
      const char* sp_env = getenv("PGO_SFRONT_SPINS");
      const int max_spins = sp_env ? atoi(sp_env) : (1 << 20);     // ~1 s of polling before the fallback
      if (++P->sfront_epoch == 0x7fffffff) {     // (the ticket counters keep counting: they wrap with the host's copy)
        P->sfront_epoch = 1;
        HIP_TRY(hipMemsetAsync(P->ds_done.p, 0, (size_t)P->fsym.nf * sizeof(int), s));
        HIP_TRY(hipMemsetAsync(P->ds_done.p + P->fsym.nf + 1, 0, (size_t)P->fsym.nf * sizeof(int), s));
      }
      static const bool want_stamps = getenv("PGO_SF_STAMPS") && getenv("PGO_SF_STAMPS")[0] == '1';
      if (want_stamps && P->ds_stamps.n == 0) HIP_TRY(P->ds_stamps.alloc(6 * (size_t)P->fsym.nf));
      const pgo::SFrontSync sy{P->ds_done.p, P->sfront_tickets, P->sfront_epoch, max_spins, want_stamps ? P->ds_stamps.p : nullptr};
      const pgo::SFrontSync sy_bwd{P->ds_done.p + P->fsym.nf + 1, P->sfront_tickets, P->sfront_epoch, max_spins, nullptr};
      P->sfront_tickets += (unsigned)P->fsym.nf;
      pgo::launch_sfront_factor(G, P->fplan, P->splan, P->fsym, s, &sy);
      if (want_stamps && P->sfront_epoch == 3) {      // development aid: the critical path of the third factorisation, from the root down
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<long long> st(6 * (size_t)P->fsym.nf);
        HIP_TRY(hipMemcpy(st.data(), P->ds_stamps.p, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        const pgo::FrontSymbolic& S = P->fsym;
        long long t0 = st[0];
        for (int q = 0; q < S.nf; ++q) t0 = std::min(t0, st[6 * (size_t)q]);
        auto us = [&](long long t) { return (double)(t - t0) / 100.0; };     // s_memrealtime: 100 MHz
        int q = S.nf - 1;
        std::fprintf(stderr, "[pgo] sfront stamps (us since the first front started): front c r kids | start wait_done extend_done factor_done published end\n");
        while (q >= 0) {
          const pgo::FrontDesc& D = S.fronts[q];
          std::fprintf(stderr, "[pgo]   front %4d c %2d r %2d kids %2d | %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f\n", q, D.c, D.r, D.child_end - D.child_begin,
                       us(st[6 * (size_t)q]), us(st[6 * (size_t)q + 1]), us(st[6 * (size_t)q + 2]), us(st[6 * (size_t)q + 3]), us(st[6 * (size_t)q + 4]), us(st[6 * (size_t)q + 5]));
          int last = -1;
          for (int ci = D.child_begin; ci < D.child_end; ++ci) { const int ch = S.child[ci]; if (last < 0 || st[6 * (size_t)ch + 4] > st[6 * (size_t)last + 4]) last = ch; }
          q = last;
        }
      }
      // The backward substitution stays one launch per level: in its single-launch form (PGO_SFRONT_FUSED_BWD=1) every front of
      // the tree polls its parent's flag at once and the ten hand-overs take 88 us against 50 us for the ten launches (KITTI-00).
      static const bool fused_bwd = getenv("PGO_SFRONT_FUSED_BWD") && getenv("PGO_SFRONT_FUSED_BWD")[0] == '1';
      pgo::launch_sfront_solve(G, P->fplan, P->splan, P->fsym, s, fused_bwd ? &sy_bwd : nullptr);
    } else {
      pgo::launch_sfront_factor(G, P->fplan, P->splan, P->fsym, s);
      pgo::launch_sfront_solve(G, P->fplan, P->splan, P->fsym, s);
    }
    return PGO_OK;
  }
  if (P->front_usable) {
    enqueue_front_factor(P, G);
    pgo::launch_front_solve(G, P->fplan, P->fsym, s, P->fsym.mixed ? &P->splan : nullptr);
    return PGO_OK;
  }
  const pgo::DirectSymbolic& S = P->dsym;
  if (P->use_graph && !P->direct_exec) {
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      pgo::launch_direct_factor(G, P->dplan, S, s);
      pgo::launch_direct_solve(G, P->dplan, S.level_ptr.data(), S.fused_from_level, s);
      e = hipStreamEndCapture(s, &P->direct_graph);
      if (e == hipSuccess) e = hipGraphInstantiate(&P->direct_exec, P->direct_graph, nullptr, nullptr, 0);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); P->drop_direct_graph(); P->use_graph = false; }
  }
  if (P->direct_exec) {
    HIP_TRY(hipGraphLaunch(P->direct_exec, s));
  } else {
    if (++P->direct_epoch == 0x7fffffff) { P->direct_epoch = 1; HIP_TRY(P->dd_col_flag.zero(s)); }
    pgo::launch_direct_factor(G, P->dplan, S, s, P->split_two_launch ? 0 : P->direct_epoch);
    pgo::launch_direct_solve(G, P->dplan, S.level_ptr.data(), S.fused_from_level, s);
  }
  return PGO_OK;
}
int run_direct(pgo_problem* P) { return run_direct(P, P->g); }

pgo::CgParams cg_params_for(const pgo_solver_options& o) {
  pgo::CgParams prm;
  if (o.linear_solver_type == PGO_BLOCK_JACOBI_PCG) {
    prm.q_tolerance = o.eta;
    prm.r_tolerance = -1.0;  // LevenbergMarquardtStrategy disables the residual test
    prm.max_iterations = o.max_linear_solver_iterations;
    prm.min_iterations = o.min_linear_solver_iterations;
  } else {
    // SPARSE_NORMAL_CHOLESKY is an exact solve.  Until the direct factorisation path is wired in it is
    // served by the same PCG run to a tight relative residual.
    prm.q_tolerance = -1.0;
    prm.r_tolerance = o.exact_r_tolerance;
    prm.max_iterations = 200000;
    prm.min_iterations = 0;
  }
  return prm;
}

// ---- LM driver ------------------------------------------------------------------------------

int evaluate_gradient_and_jacobian(pgo_problem* P, bool first) {
  const auto t0 = Clock::now();
  hipStream_t s = P->stream;
  if (first) {
    int rc = fill_scale_one(P);
    if (rc) return rc;
    rc = linearize_all(P);
    if (rc) return rc;
    if (P->opt.jacobi_scaling) {
      pgo::launch_scale_from_diag(P->g, s);
      rc = linearize_all(P);
      if (rc) return rc;
    }
  } else {
    int rc = linearize_all(P);
    if (rc) return rc;
  }
  pgo::launch_gradient_norm(P->g, s);
  P->lm.t_jacobian += seconds_since(t0);
  return PGO_OK;
}

bool pipeline_wanted(const pgo_problem* P);
bool universal_wanted(const pgo_problem* P);
int lm_begin(pgo_problem* P, const pgo_solver_options* options) {
  P->want_direct = options->linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY;
  int rc = prepare(P);
  P->want_direct = false;
  if (rc) return rc;
  P->opt = *options;
  LmState& L = P->lm;
  const double t_setup = L.t_setup;
  L = LmState();
  L.t_setup = t_setup;
  const auto t0 = Clock::now();
  P->g.loss_kind = P->loss_kind;
  P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p;
  P->g.pose_c = P->d_pose_c.p;
  static const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(P->d_pose_0.p, P->g.pose_x, P->d_pose_0.n * sizeof(double), hipMemcpyDeviceToDevice, P->stream));
  HIP_TRY(P->d_flags.zero(P->stream));
  // Init + IterationZero, enqueued before the factorisation's plan is waited for (its host analysis runs on a helper thread
  // since prepare(), its uploads queue up behind these kernels)
  P->g.cluster = 1;
  rc = evaluate_gradient_and_jacobian(P, true);
  if (rc) return rc;
  pgo::launch_cost(P->g, P->g.pose_x, 0, P->stream);
  // x_norm: run the retraction with a zero step (delta is zero after prepare/reset)
  HIP_TRY(P->d_delta.zero(P->stream));
  pgo::launch_apply_step(P->g, P->g.delta, P->stream);
  pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, P->stream);
  int cluster = P->opt.pcg_cluster_poses;
  if (P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY) {
    const auto t_sym = Clock::now();
    rc = prepare_direct(P);
    if (rc) return rc;
    L.t_setup += seconds_since(t_sym);
    // exact request served by PCG to exact_r_tolerance: the preconditioner is ours to choose — 2-pose chain clusters need
    // ~2.5x fewer iterations than 6x6 blocks at almost the same cost per iteration
    if ((!P->direct_usable || P->dsym.hybrid) && cluster < 2) cluster = 2;
  }
  rc = prepare_clusters(P, cluster);
  if (rc) return rc;
  if (verbose) std::fprintf(stderr, "[pgo] lm_begin: plan + clusters prepared    %.2f ms\n", 1e3 * seconds_since(t0));
  HIP_TRY(hipStreamSynchronize(P->stream));
  HIP_TRY(hipGetLastError());
  if (verbose) std::fprintf(stderr, "[pgo] lm_begin: iteration zero evaluated    %.2f ms\n", 1e3 * seconds_since(t0));
  L.x_cost = P->scal->cand_cost;
  L.initial_cost = L.x_cost;
  L.x_norm = std::sqrt(P->scal->x_norm_sq);
  L.gmax = P->scal->gradient_max;
  L.radius = P->opt.initial_trust_region_radius;
  L.decrease_factor = 2.0;
  L.reuse_diagonal = false;
  L.cur = pgo_iteration_record{};
  L.cur.iteration = 0;
  L.cur.step_is_successful = 1;
  L.cur.cost = L.x_cost;
  L.cur.gradient_max_norm = L.gmax;
  L.pending_record = true;
  L.active = true;
  P->pipelined = pipeline_wanted(P);
  P->universal = universal_wanted(P);
  P->pipe_dirty = true;
  L.t_total += seconds_since(t0);
  if (!std::isfinite(L.x_cost)) {
    L.terminated = true; L.termination = PGO_FAILURE; L.reason = 7;
    L.message = "Initial cost is not finite.";
  }
  return PGO_OK;
}

void terminate(LmState& L, int termination, int reason, const char* fmt, ...) {
  char buf[240];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  L.terminated = true;
  L.termination = termination;
  L.reason = reason;
  L.message = buf;
}

// ---- speculative linearisation -----------------------------------------------------------------------------------------
// Between the step tail and the re-linearisation of an accepted point the GPU used to wait for the host (hand-off, decision,
// launch: 11-16 us of the 130-300 us LM iteration of Manhattan 10 k).  The candidate is therefore linearised at once, behind the
// tail, into a spare set of buffers; the host decides meanwhile.  Accepted (the usual case): the sets are swapped, nothing
// is recomputed.  Rejected: the current set was never touched.  Same kernel, same inputs: results are bit-identical.
// One rank (the exchange of the diagonal blocks would have to ride along), eager enqueue (captured batches hold pointers).
int ensure_spec_buffers(pgo_problem* P) {
  if (P->spec_ready) return PGO_OK;
  hipStream_t s = P->stream;
  HIP_TRY(P->d_bsr2.alloc(P->d_bsr.n));
  HIP_TRY(P->d_bsr2.zero(s));
  HIP_TRY(P->d_Hdiag2.alloc(P->d_Hdiag.n));
  HIP_TRY(P->d_Hdiag2.zero(s));
  HIP_TRY(P->d_grad2.alloc(P->d_grad.n));
  HIP_TRY(P->d_grad2.zero(s));
  P->spec_ready = true;
  return PGO_OK;
}
bool speculation_on(const pgo_problem* P) {
  static const bool off = getenv("PGO_NO_SPECULATION") && getenv("PGO_NO_SPECULATION")[0] == '1';
  return !off && P->g.world == 1 && !P->use_graph && !(P->comm && P->comm->world > 1);
}
struct SpareSet { double *bsr, *Hdiag, *grad; };
inline SpareSet spare_set(pgo_problem* P) {
  const bool primary_in_use = P->g.bsr_val == P->d_bsr.p;
  return primary_in_use ? SpareSet{P->d_bsr2.p, P->d_Hdiag2.p, P->d_grad2.p} : SpareSet{P->d_bsr.p, P->d_Hdiag.p, P->d_grad.p};
}
void launch_speculative_linearize(pgo_problem* P, int gate) {
  pgo::DeviceGraph gs = P->g;
  const SpareSet sp = spare_set(P);
  gs.pose_x = P->g.pose_c;
  gs.bsr_val = sp.bsr; gs.Hdiag = sp.Hdiag; gs.grad = sp.grad;
  pgo::launch_linearize(gs, P->stream, gate);
}

// ---- the host half of one TrustRegionMinimizer pass (SURVEY.md A.6 step 7 order), shared by the single-problem driver and
// the batched one (one LmState per component there): pure bookkeeping on LmState, no device work ----
// What the device hands back after a trial step.
struct StepScalars {
  double cand_cost, model_change, step_norm_sq, x_norm_sq, gradient_max;
  int cg_iterations, cg_status, linearize_bad;
};
enum StepAction { STEP_NONE = 0, STEP_ACCEPT = 1, STEP_REJECT = 2 };   // NONE: terminated or invalid step

// FinalizeIterationAndCheckIfMinimizerCanContinue.  Returns false when the minimizer stops here.
bool lm_pre_step(LmState& L, const pgo_solver_options& o) {
  if (L.pending_record) {
    if (L.cur.step_is_successful) ++L.num_successful; else ++L.num_unsuccessful;
    L.cur.trust_region_radius = L.radius;
    L.records.push_back(L.cur);
    L.pending_record = false;
  }
  if (L.cur.iteration >= o.max_num_iterations) {
    terminate(L, PGO_NO_CONVERGENCE, 5, "Maximum number of iterations reached. Number of iterations: %d.", L.cur.iteration);
    return false;
  }
  if (!L.gmax_deferred && L.cur.step_is_successful && L.cur.gradient_max_norm <= o.gradient_tolerance) {
    terminate(L, PGO_CONVERGENCE, 3, "Gradient tolerance reached. Gradient max norm: %e <= %e", L.cur.gradient_max_norm, o.gradient_tolerance);
    return false;
  }
  if (L.radius <= o.min_trust_region_radius) {
    terminate(L, PGO_CONVERGENCE, 4, "Minimum trust region radius reached. Trust region radius: %e <= %e", L.radius, o.min_trust_region_radius);
    return false;
  }
  return true;
}

// Everything after the trial step came back: deferred gradient test, then the rules of pgo_lm_rules.h (step validity,
// parameter / function tolerance, IsStepSuccessful, radius update) — the very function the device applies when it decides
// itself.  STEP_ACCEPT: the caller makes the candidate the current point and re-linearises.
pgo::LmTolerances lm_tolerances(const pgo_solver_options& o) {
  return pgo::LmTolerances{o.min_relative_decrease, o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance,
                           o.max_trust_region_radius, o.min_trust_region_radius, o.max_num_iterations, o.max_num_consecutive_invalid_steps};
}
inline pgo_iteration_record to_record(const pgo::LmRecord& r) { pgo_iteration_record o; memcpy(&o, &r, sizeof o); return o; }
void terminate_by_reason(LmState& L, const pgo_solver_options& o, int termination, int reason, double value) {
  switch (reason) {
    case 1: terminate(L, termination, 1, "Function tolerance reached. |cost_change|/cost: %e <= %e", value, o.function_tolerance); break;
    case 2: terminate(L, termination, 2, "Parameter tolerance reached. Relative step_norm: %e <= %e.", value, o.parameter_tolerance); break;
    case 3: terminate(L, termination, 3, "Gradient tolerance reached. Gradient max norm: %e <= %e", value, o.gradient_tolerance); break;
    case 4: terminate(L, termination, 4, "Minimum trust region radius reached. Trust region radius: %e <= %e", value, o.min_trust_region_radius); break;
    case 5: terminate(L, termination, 5, "Maximum number of iterations reached. Number of iterations: %d.", (int)value); break;
    case 6: terminate(L, termination, 6, "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d",
                      o.max_num_consecutive_invalid_steps); break;
    default: terminate(L, termination, reason, "Terminated (reason %d).", reason); break;
  }
}
StepAction lm_post_step(LmState& L, const pgo_solver_options& o, const StepScalars& sc, int extra_linear_iterations) {
  ++L.num_trial_steps;
  const int cg_it = sc.cg_iterations;
  L.num_linear_iterations += cg_it + extra_linear_iterations;   // the iterations of an over-budget PCG try are work done, counted in the summary
  if (L.gmax_deferred) {
    // the gradient test of FinalizeIterationAndCheckIfMinimizerCanContinue for the point accepted last
    // iteration: if it fires, the step just computed is discarded (x was not touched)
    L.gmax_deferred = false;
    L.gmax = sc.gradient_max;
    L.cur.gradient_max_norm = L.gmax;
    if (!L.records.empty()) L.records.back().gradient_max_norm = L.gmax;
    if (L.gmax <= o.gradient_tolerance) {
      L.num_linear_iterations -= cg_it;
      L.reuse_diagonal = true;
      terminate_by_reason(L, o, PGO_CONVERGENCE, 3, L.gmax);
      return STEP_NONE;
    }
  }
  pgo::LmCore C{L.radius, L.decrease_factor, L.x_cost, L.x_norm, L.cur.gradient_max_norm, L.cur.iteration, L.reuse_diagonal ? 1 : 0,
                L.num_consecutive_invalid, 0};
  const pgo::LmStepIn in{sc.cand_cost, sc.model_change, sc.step_norm_sq, sc.x_norm_sq, cg_it, sc.cg_status, sc.linearize_bad, 0};
  pgo::LmRecord nx{};
  double value = 0.0;
  const pgo::LmOutcome out = pgo::lm_decide(C, lm_tolerances(o), in, nx, value);
  L.radius = C.radius; L.decrease_factor = C.decrease_factor; L.x_cost = C.x_cost; L.x_norm = C.x_norm;
  L.reuse_diagonal = C.reuse_diagonal != 0; L.num_consecutive_invalid = C.num_consecutive_invalid;
  switch (out) {
    case pgo::LM_OUT_INVALID_FAIL:
      terminate_by_reason(L, o, PGO_FAILURE, 6, 0.0);
      L.cur = to_record(nx);
      return STEP_NONE;
    case pgo::LM_OUT_INVALID:
      L.cur = to_record(nx);
      L.pending_record = true;
      return STEP_NONE;
    case pgo::LM_OUT_PARAM_TOL: terminate_by_reason(L, o, PGO_CONVERGENCE, 2, value); return STEP_NONE;
    case pgo::LM_OUT_FUNC_TOL: terminate_by_reason(L, o, PGO_CONVERGENCE, 1, value); return STEP_NONE;
    case pgo::LM_OUT_ACCEPT:
      L.gmax_deferred = true;  // filled in at the next host sync (or at the end of the solve)
      L.cur = to_record(nx);
      L.pending_record = true;
      return STEP_ACCEPT;
    default:
      L.cur = to_record(nx);
      L.pending_record = true;
      return STEP_REJECT;
  }
}

// One pass of the TrustRegionMinimizer loop body (SURVEY.md A.6 step 7 order).
int lm_advance(pgo_problem* P) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  hipStream_t s = P->stream;
  if (L.terminated) return PGO_OK;
  const auto t_it = Clock::now();

  if (!lm_pre_step(L, o)) return PGO_OK;

  // ComputeTrustRegionStep + ComputeCandidatePointAndEvaluateCost, enqueued back to back: damping, one
  // batch of CG iterations, model cost change / delta / candidate, candidate cost, scalar fold.  ONE host
  // sync per LM iteration in the common case; if the CG batch was too short, further batches follow and
  // the (cheap) tail is re-enqueued.
  const auto t_lin = Clock::now();
  const pgo::CgParams prm = cg_params_for(o);
  int rc = damping_all(P, L.radius, o.min_lm_diagonal, o.max_lm_diagonal, L.reuse_diagonal ? 1 : 0);
  if (rc) return rc;
  const bool direct = o.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  // Both available (hybrid): PCG gets the budget of ~1.5 factorisations, in CG iterations priced by the same deterministic
  // cost model that admitted the factorisation (0.7 us per schedule step; 10 us + 42 ps per slot per CG iteration), so the
  // choice depends on iteration counts only, never on a clock.  Over budget = redo the iteration with the factorisation.
  const bool hybrid = direct && P->dsym.hybrid;
  const char* budget_env = getenv("PGO_HYBRID_BUDGET");     // experiments / tests: CG iterations a PCG try may take
  const int cg_budget = !hybrid ? 0 : budget_env ? std::max(1, atoi(budget_env))
                                : std::max(50, (int)(1.5 * 0.7 * P->dsym.est_steps / (10.0 + 4.2e-5 * (double)P->g.n_slots)));
  bool use_direct = direct;
  if (hybrid && (L.hybrid_pcg || L.hybrid_direct_run >= L.hybrid_probe_after ||
                 (L.hybrid_direct_run >= 1 && L.hybrid_fail_radius > 0.0 && L.radius < 0.1 * L.hybrid_fail_radius)))
    use_direct = false;
  int wasted_cg = 0;
  const bool spec = speculation_on(P);
  bool spec_in_flight = false;   // the candidate's linearisation was enqueued behind the tail that produced the scalars read below
  if (spec) { rc = ensure_spec_buffers(P); if (rc) return rc; }
  arm_handoff(P);
  if (!use_direct) {
    // every batch carries the gated tail: the host hears back once per batch and finds the step scalars ready
    // as soon as the CG has stopped
    pgo::CgParams run_prm = prm;
    if (hybrid) run_prm.max_iterations = cg_budget;
    pgo::launch_pcg_init(P->g, s);
    for (int round = 0, enqueued = 0;; ++round) {
      const int nb = pick_batch(run_prm, o.cg_batch, round, enqueued, P->last_cg_iterations);
      rc = launch_cg_batch(P, run_prm, nb, true, enqueued + 1);
      enqueued += nb;
      if (rc) return rc;
      // the speculative launch costs an early-exit launch (~3.5 us) in a batch the CG does not finish in and saves the
      // host gap (~12 us) in the one it does: skipped in a first batch shorter than the previous solve's iteration count
      spec_in_flight = spec && (round > 0 || P->last_cg_iterations <= nb);
      if (spec_in_flight) launch_speculative_linearize(P, 1);
      rc = wait_handoff(P);
      if (rc) return rc;
      if (P->scal->cg_status != -1) break;
      spec_in_flight = false;      // gated out: the CG was still running
      arm_handoff(P);
    }
    if (hybrid) {
      if (P->scal->cg_iterations >= cg_budget && P->scal->cg_status == 0) {   // not converged within the budget
        wasted_cg = P->scal->cg_iterations;
        L.hybrid_probe_after = L.hybrid_pcg ? 2 : std::min(16, 2 * L.hybrid_probe_after);
        L.hybrid_pcg = false;
        L.hybrid_direct_run = 0;
        L.hybrid_fail_radius = L.radius;
        ++L.hybrid_pcg_over;
        use_direct = true;
        arm_handoff(P);
      } else {
        L.hybrid_pcg = true;
        L.hybrid_probe_after = 1;
        ++L.hybrid_pcg_ok;
      }
    }
  }
  if (use_direct) {
    P->scal->cg_status = 0;       // host-visible block: the CG kernels that normally fill these do not run
    P->scal->cg_iterations = 0;
    rc = run_direct(P);
    if (rc) return rc;
    rc = enqueue_tail(P, nullptr);
    if (rc) return rc;
    spec_in_flight = spec;
    if (spec) launch_speculative_linearize(P, 0);
    rc = wait_handoff(P);
    if (rc) return rc;
    if ((P->scal->linearize_bad & 4) && (P->front_usable ? !P->front_launches : P->sfront_usable ? !P->sfront_levels : !P->split_two_launch)) {
      // a single-launch SPLIT step waited in vain for a column's diagonal block (its workgroups were not all resident), or a
      // front of the single-launch small-front factorisation for a child: not a numerical failure — repeat this factorisation
      // in the form without in-kernel waits and keep to it
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: an in-kernel wait of the single-launch factorisation timed out; one launch per step from now on\n");
      arm_handoff(P);
      rc = run_direct(P);
      if (rc) return rc;
      rc = enqueue_tail(P, nullptr);
      if (rc) return rc;
      if (spec) launch_speculative_linearize(P, 0);
      rc = wait_handoff(P);
      if (rc) return rc;
    }
    ++L.n_factorizations;
    if (hybrid) { ++L.hybrid_direct_run; ++L.hybrid_direct; }
  }
  HIP_TRY(hipGetLastError());
  const pgo::LmScalars dsc = *P->scal;
  P->last_cg_iterations = dsc.cg_iterations;
  L.t_linear += seconds_since(t_lin);
  const StepScalars sc{dsc.cand_cost, dsc.model_change, dsc.step_norm_sq, dsc.x_norm_sq, dsc.gradient_max,
                       dsc.cg_iterations, dsc.cg_status, dsc.linearize_bad};
  const StepAction action = lm_post_step(L, o, sc, wasted_cg);
  if (action == STEP_ACCEPT) {
    // HandleSuccessfulStep (device half): x <- candidate, re-linearise
    std::swap(P->g.pose_x, P->g.pose_c);
    if (spec_in_flight) {   // the candidate was linearised behind the tail: its set becomes the current one
      const SpareSet sp = spare_set(P);
      P->g.bsr_val = sp.bsr; P->g.Hdiag = sp.Hdiag; P->g.grad = sp.grad;
      pgo::launch_gradient_norm(P->g, s);
    } else {
      rc = evaluate_gradient_and_jacobian(P, false);
      if (rc) return rc;
    }
  }
  L.t_total += seconds_since(t_it);
  return PGO_OK;
}

// ---- device-resident LM: the host enqueues sequences ahead of the decisions (pgo_kernels.h LmDev) -----------------------------
// r02 ended every LM iteration in a hand-off: the device folded the step scalars, the host decided accept / reject, updated the
// radius and enqueued the next iteration, the GPU idle meanwhile (13 us on the development box, ~60 us on the driver's: 16 % of
// the Manhattan 10 k step; half of a KITTI-00 iteration was not GPU work).  Now the last work-group of the step tail applies the
// rules of pgo_lm_rules.h itself, the kernels read radius / reuse-diagonal / "was the step accepted" from device memory, and the
// host's only job is to keep the stream fed: it enqueues the sequence of the NEXT iteration while the current one runs, and reads
// the iteration records afterwards.  What it cannot know when it enqueues — whether the CG will be through within the batch it
// allots, whether the step will be accepted, whether the solve ends — the kernels find out for themselves (LmDev::phase /
// accepted / halt): a wrong guess costs early-exit launches (~2.6 us each), never a wrong result.
bool pipeline_wanted(const pgo_problem* P) {
  const bool off = getenv("PGO_NO_PIPELINE") && getenv("PGO_NO_PIPELINE")[0] == '1';   // (read per solve: the tests compare both drivers in one process)
  if (off || P->g.world != 1 || (P->comm && P->comm->world > 1) || P->use_graph) return false;
  const bool direct = P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  if (direct && P->dsym.hybrid && !P->front_usable && !P->sfront_usable) return false;   // factorisation or PCG chosen per iteration by the host
  // Exact steps: the launch sequence of an iteration is the same every time, so enqueueing ahead costs nothing.  PCG: the host has
  // to allot CG iterations to a sequence before it knows how many the CG will take (Manhattan 10 k: 32, 8, 20, 85, 15, 125, 14 ...
  // then 3-6), every unused one is two early-exit launches and every CG that outlives its sequence a gated tail: measured 0.38 ms
  // per LM iteration against 0.30 with the host in the loop on a host that answers within 13 us.  PGO_PIPELINE_PCG=1 selects the
  // sequences for PCG as well (tests do).
  if (!direct) { const char* e = getenv("PGO_PIPELINE_PCG"); return e && e[0] == '1'; }
  return true;
}

// host LmState -> device (begin / reset; the stream is idle or the copy is ordered behind what is in flight)
int lm_upload_state(pgo_problem* P) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  if (P->d_lm.n == 0) HIP_TRY(P->d_lm.alloc(1));
  pgo::LmDev D{};
  D.halt = pgo::LM_RUN; D.phase = pgo::LM_PHASE_NEW; D.accepted = 0;
  D.lm_done = 0;
  D.num_successful = L.num_successful; D.num_unsuccessful = L.num_unsuccessful; D.num_linear_iterations = L.num_linear_iterations;
  D.num_records = L.cur.iteration + 1;       // record index = iteration number (the ring slot of record r is r % LM_RING)
  P->pipe_pulled = L.cur.iteration + 1;
  const pgo::CgParams prm = cg_params_for(o);
  D.cg_period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;   // (launch_cg_batch: an exact request served by PCG never refreshes)
  D.last_cg = P->last_cg_iterations;
  D.core = pgo::LmCore{L.radius, L.decrease_factor, L.x_cost, L.x_norm, L.gmax, L.cur.iteration, L.reuse_diagonal ? 1 : 0, L.num_consecutive_invalid, 0};
  D.tol = lm_tolerances(o);
  D.min_diag = o.min_lm_diagonal; D.max_diag = o.max_lm_diagonal;
  HIP_TRY(hipStreamSynchronize(P->stream));      // nobody writes the pinned block while the host fills it
  if (P->universal) {
    HIP_TRY(P->d_cg.zero(P->stream));            // the stream's operation words and its launch counter (the budget kernel opens it)
    HIP_TRY(P->d_flags.zero(P->stream));
    P->uni_enq = 0;
    P->scal->slots_done = 0;
  }
  P->scal->lm = D;
  P->scal->lm_done = 0; P->scal->halt = 0; P->scal->last_cg = D.last_cg;
  P->scal->seq_done = P->pipe_seq;
  HIP_TRY(hipMemcpyAsync(P->d_lm.p, &P->scal->lm, sizeof(pgo::LmDev), hipMemcpyHostToDevice, P->stream));
  HIP_TRY(hipStreamSynchronize(P->stream));      // (the pinned source doubles as the mirror the device writes)
  P->pipe_t_linear0 = L.t_linear; P->pipe_t_jacobian0 = L.t_jacobian;
  return PGO_OK;
}

// records the device has finished with -> LmState (a record is final once a later one exists, or once everything has drained)
void lm_pull_records(pgo_problem* P, bool drained) {
  LmState& L = P->lm;
  const int have = __atomic_load_n(&P->scal->lm.num_records, __ATOMIC_ACQUIRE);
  const int upto = drained ? have : have - 1;
  for (; P->pipe_pulled < upto; ++P->pipe_pulled) L.records.push_back(to_record(P->scal->ring[P->pipe_pulled % pgo::LM_RING]));
}

// device -> host LmState, everything drained
void lm_pull_state(pgo_problem* P) {
  LmState& L = P->lm;
  lm_pull_records(P, true);
  const pgo::LmDev& M = P->scal->lm;
  L.radius = M.core.radius; L.decrease_factor = M.core.decrease_factor; L.x_cost = M.core.x_cost; L.x_norm = M.core.x_norm;
  L.gmax = M.core.gmax; L.reuse_diagonal = M.core.reuse_diagonal != 0; L.num_consecutive_invalid = M.core.num_consecutive_invalid;
  L.num_successful = M.num_successful; L.num_unsuccessful = M.num_unsuccessful; L.num_linear_iterations = M.num_linear_iterations;
  if (!L.records.empty()) L.cur = L.records.back();
  L.cur.iteration = M.core.iteration;
  L.pending_record = false;
  L.gmax_deferred = false;
  L.t_linear = P->pipe_t_linear0 + 1e-8 * (double)M.ticks_linear;       // s_memrealtime: 100 MHz
  L.t_jacobian = P->pipe_t_jacobian0 + 1e-8 * (double)M.ticks_jacobian;
  P->last_cg_iterations = M.last_cg;
}

// the in-kernel counters of the single-launch factorisations no longer match the host's after launches that exited at a halt
int resync_direct_counters(pgo_problem* P) {
  hipStream_t s = P->stream;
  if (P->front_usable && P->df_st_count.n) { HIP_TRY(P->df_st_count.zero(s)); P->front_epoch = 0; P->front_tickets = 0; }
  if (P->sfront_usable && P->ds_done.n) { HIP_TRY(P->ds_done.zero(s)); P->sfront_epoch = 0; P->sfront_tickets = 0; }
  if (P->dd_col_flag.n) { HIP_TRY(P->dd_col_flag.zero(s)); P->direct_epoch = 0; }
  return PGO_OK;
}

// CG iterations allotted to a sequence.  The CG stops by itself; an iteration enqueued past its end costs two early-exit
// launches (~5 us), a CG that outlives its sequence goes on in the next one at the price of that sequence's skipped head and
// gated tail (~18 us) — provided a multiple of the refresh period has been completed (the refresh launches sit at fixed
// positions), else the host steps in (~60 us).  Short CGs (the steady state of an LM run: 3-6 iterations) get the last count
// + 2; long ones a multiple of the period.  cont_streak: sequences that just ended without a decision.
int pipe_pick_batch(const pgo_problem* P, const pgo::CgParams& prm, int period, int pred, int cont_streak) {
  int nb;
  if (P->opt.cg_batch > 0) nb = P->opt.cg_batch;
  else {
    if (pred <= 0) pred = 6;
    if (pred <= 7 && cont_streak == 0) nb = pred + 2;
    else {
      const int want = std::max(pred + pred / 4 + 1, 8) << std::min(cont_streak, 3);
      nb = period > 0 ? (want + period - 1) / period * period : want;
      nb = std::min(nb, period > 0 ? std::max(period, 60 / period * period) : 64);
    }
  }
  nb = std::max(1, std::min(nb, prm.max_iterations));
  return (nb + 1) & ~1;
}

// one sequence: the kernels of one prospective LM iteration (or the continuation of the previous one's CG)
int enqueue_sequence(pgo_problem* P, const pgo::DeviceGraph& gp, const pgo::DeviceGraph& gl, const pgo::CgParams& prm, bool direct, int nb,
                     int start_it, bool head) {
  hipStream_t s = P->stream;
  const pgo_solver_options& o = P->opt;
  if (head) pgo::launch_damping(gp, 1.0, o.min_lm_diagonal, o.max_lm_diagonal, 0, s);   // radius and mode: LmDev
  if (direct) {
    int rc = run_direct(P, gp);
    if (rc) return rc;
    const pgo::CgParams none{0.0, -1.0, 0, 0};
    pgo::launch_spmv_tail(gp, none, s, 0, 1);
    pgo::launch_step_tail(gp, s, 2);
  } else {
    if (head) pgo::launch_pcg_init(gp, s);
    const int period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;
    for (int i = 0; i < nb; ++i) {
      const bool refresh = period > 0 && ((start_it + i) % period) == 0;
      int rc = cg_iteration(P, gp, prm, ((start_it + i) & 1), refresh);
      if (rc) return rc;
    }
    pgo::launch_spmv_tail(gp, prm, s, 1, 1);
    pgo::launch_step_tail(gp, s, 1);
  }
  pgo::launch_linearize(gl, s, 2);
  pgo::launch_accept_finish(gp, ++P->pipe_seq, s);
  P->pipe_last_nb = nb;
  return PGO_OK;
}

// waits until every enqueued sequence is through (pinned counter; the stream synchronise is the fallback for long waits)
int pipe_drain(pgo_problem* P) {
  const auto t0 = Clock::now();
  for (unsigned spins = 1; __atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE) != P->pipe_seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0x3ff) == 0 && seconds_since(t0) > 0.002) {
      HIP_TRY(hipStreamSynchronize(P->stream));
      if (__atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE) != P->pipe_seq)
        return set_error(PGO_ERR_HIP, "the enqueued LM sequences did not report completion (%d of %d)", P->scal->seq_done, P->pipe_seq);
    }
  }
  HIP_TRY(hipGetLastError());
  return PGO_OK;
}

// ---- the universal stream (pgo_kernels.h UniOp): PCG on one rank ---------------------------------------------------------------
// The host enqueues  V S V S ...  and nothing else; what each launch does is the device's business.  It keeps between `lo` and
// `hi` pairs ahead of the device's launch counter — enough that the GPU never waits for a launch (a pair is ~13 us of work), few
// enough that the launches left over when the stream stops (terminated, or the step budget of pgo_solver_step used up) drain in
// well under 0.1 ms.
bool universal_wanted(const pgo_problem* P) {
  const bool off = getenv("PGO_NO_PIPELINE") && getenv("PGO_NO_PIPELINE")[0] == '1';
  const char* u = getenv("PGO_UNI");
  if (off || (u && u[0] == '0') || P->g.world != 1 || (P->comm && P->comm->world > 1) || P->use_graph) return false;
  const bool direct = P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  if (direct || !pgo::uni_supported(P->g)) return false;
  // large graphs (kernels of 100+ us) gain nothing from it and the slot kernel's LDS footprint (the linearisation's) would cost the
  // SpMV occupancy there: they keep the host-driven loop
  const long long limit = getenv("PGO_UNI_MAX_SLOTS") ? atoll(getenv("PGO_UNI_MAX_SLOTS")) : 600000;
  return (u && u[0] == '1') || P->g.n_slots <= limit;
}

int lm_run_universal(pgo_problem* P, int budget, int* ran) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  hipStream_t s = P->stream;
  if (ran) *ran = 0;
  if (L.terminated || budget == 0) return PGO_OK;
  const auto t_run = Clock::now();
  if (!lm_pre_step(L, o)) { L.t_total += seconds_since(t_run); return PGO_OK; }
  if (P->pipe_dirty) {
    int rc0 = lm_upload_state(P);
    if (rc0) return rc0;
    P->pipe_dirty = false;
  }
  static const int hi = getenv("PGO_UNI_AHEAD") ? std::max(2, atoi(getenv("PGO_UNI_AHEAD"))) : 12;
  const int lo = std::max(1, hi / 3);
  const pgo::CgParams prm = cg_params_for(o);
  const int period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;
  pgo::DeviceGraph gp = P->g;
  gp.lm = P->d_lm.p;
  const int d0 = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
  P->scal->halt = 0;
  pgo::launch_lm_budget(gp, budget, s);
  unsigned idle_spins = 0;
  auto t_idle = Clock::now();
  for (;;) {
    if (__atomic_load_n(&P->scal->halt, __ATOMIC_ACQUIRE)) break;
    const int pending = P->uni_enq - __atomic_load_n(&P->scal->slots_done, __ATOMIC_ACQUIRE);
    if (pending <= hi - lo) {
      for (int i = 0; i < lo; ++i) {
        pgo::launch_uni_v(gp, prm, o.min_lm_diagonal, o.max_lm_diagonal, s);
        pgo::launch_uni_s(gp, prm, period, s);
        ++P->uni_enq;
      }
      lm_pull_records(P, false);
      idle_spins = 0; t_idle = Clock::now();
      continue;
    }
    __builtin_ia32_pause();
    if ((++idle_spins & 0xfff) == 0 && seconds_since(t_idle) > 0.5) {
      HIP_TRY(hipStreamSynchronize(s));
      if (P->uni_enq != __atomic_load_n(&P->scal->slots_done, __ATOMIC_ACQUIRE))
        return set_error(PGO_ERR_HIP, "the universal LM stream stopped reporting progress (%d of %d launches)", P->scal->slots_done, P->uni_enq);
      t_idle = Clock::now();
    }
  }
  // whatever was enqueued behind the halt exits at once; one more launch tells the host when the stream has drained
  arm_handoff(P);
  pgo::launch_lm_publish(gp, s);
  int rc = wait_handoff(P);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  const int d1 = P->scal->lm_done;
  lm_pull_state(P);
  L.num_trial_steps += d1 - d0;
  if (ran) *ran = d1 - d0;
  const pgo::LmDev& M = P->scal->lm;
  if (M.halt == pgo::LM_HALT_TERMINATED) terminate_by_reason(L, o, M.termination, M.reason, M.term_value);
  L.t_total += seconds_since(t_run);
  return PGO_OK;
}

// Runs up to `budget` LM iterations (decisions; < 0: until the solve terminates).  *ran = iterations executed.
int lm_run_pipelined(pgo_problem* P, int budget, int* ran) {
  LmState& L = P->lm;
  const pgo_solver_options& o = P->opt;
  if (ran) *ran = 0;
  if (L.terminated || budget == 0) return PGO_OK;
  const auto t_run = Clock::now();
  // the opening tests of the first pass (they push the iteration-0 record; the device applies them from then on)
  if (!lm_pre_step(L, o)) { L.t_total += seconds_since(t_run); return PGO_OK; }
  if (P->pipe_dirty) {
    int rc0 = lm_upload_state(P);
    if (rc0) return rc0;
    P->pipe_dirty = false;
  }
  static const int lookahead = getenv("PGO_PIPELINE_AHEAD") ? std::max(0, atoi(getenv("PGO_PIPELINE_AHEAD"))) : 1;
  const bool direct = o.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable;
  const pgo::CgParams prm = cg_params_for(o);
  const int period = prm.q_tolerance < 0.0 ? 0 : o.cg_residual_reset_period;
  pgo::DeviceGraph gp = P->g;
  gp.lm = P->d_lm.p;
  pgo::DeviceGraph gl = gp;
  gl.pose_x = gp.pose_c;          // the linearisation of an accepted step reads the candidate buffer
  pgo::launch_lm_resume(gp, -1, P->stream);    // time mark of the device's phase clocks
  const int d0 = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
  const long long target = budget < 0 ? (1LL << 40) : (long long)d0 + budget;
  int cont_streak = 0, seen_seq = P->pipe_seq, seen_done = d0;
  int rc = PGO_OK;
  unsigned idle_spins = 0;
  auto t_idle = Clock::now();
  for (;;) {
    // seq_done first: decisions of sequences counted as in flight may already be in lm_done, never the other way round
    const int sdone = __atomic_load_n(&P->scal->seq_done, __ATOMIC_ACQUIRE);
    const int d = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
    const int halt = __atomic_load_n(&P->scal->halt, __ATOMIC_ACQUIRE);
    const int in_flight = P->pipe_seq - sdone;
    if (sdone != seen_seq) {        // sequences that ended without a decision: their CG goes on
      cont_streak = (d == seen_done) ? cont_streak + (sdone - seen_seq) : 0;
      seen_seq = sdone; seen_done = d;
      lm_pull_records(P, false);
    }
    if (halt) {
      rc = pipe_drain(P);
      if (rc) break;
      const int h = P->scal->lm.halt;
      if (h == pgo::LM_HALT_TERMINATED) break;
      rc = resync_direct_counters(P);
      if (rc) break;
      P->scal->halt = 0;
      seen_seq = P->pipe_seq; seen_done = __atomic_load_n(&P->scal->lm_done, __ATOMIC_ACQUIRE);
      if (h == pgo::LM_HALT_CG_STALL) {
        // the rest of this iteration's CG, from where it stands, up to the next multiple of the refresh period (from there the
        // sequences line up again), then the tail
        const int completed = P->scal->cg_iterations;
        int nb = pipe_pick_batch(P, prm, period, std::max(P->pipe_last_nb, 8), 1);
        if (period > 0) nb = ((completed + nb + period - 1) / period) * period - completed;
        pgo::launch_lm_resume(gp, 1, P->stream);
        rc = enqueue_sequence(P, gp, gl, prm, false, nb, completed + 1, false);
        if (rc) break;
        cont_streak = 1;
        continue;
      }
      // LM_HALT_REFACTOR: a wait inside a single-launch factorisation ran out (its work-groups were not all resident): the
      // iteration is repeated with one launch per step, and the problem keeps to that form
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] direct: an in-kernel wait of the single-launch factorisation timed out; one launch per step from now on\n");
      pgo::launch_lm_resume(gp, 0, P->stream);
      continue;
    }
    if (d >= target && in_flight == 0) break;
    if ((long long)d + in_flight < target && in_flight <= lookahead) {
      const int nb = direct ? 0 : pipe_pick_batch(P, prm, period, __atomic_load_n(&P->scal->last_cg, __ATOMIC_RELAXED), cont_streak);
      rc = enqueue_sequence(P, gp, gl, prm, direct, nb, 1, true);
      if (rc) break;
      idle_spins = 0; t_idle = Clock::now();
      continue;
    }
    __builtin_ia32_pause();
    if ((++idle_spins & 0xfff) == 0 && seconds_since(t_idle) > 0.002) std::this_thread::sleep_for(std::chrono::microseconds(100));   // long iterations (sphere x10: 34 ms): no need to burn the core
  }
  if (rc == PGO_OK) rc = pipe_drain(P);
  if (rc) return rc;
  const int d1 = P->scal->lm_done;
  lm_pull_state(P);
  L.num_trial_steps += d1 - d0;
  if (direct) L.n_factorizations += d1 - d0;
  if (ran) *ran = d1 - d0;
  const pgo::LmDev& M = P->scal->lm;
  if (M.halt == pgo::LM_HALT_TERMINATED) {
    terminate_by_reason(L, o, M.termination, M.reason, M.term_value);
    rc = resync_direct_counters(P);     // sequences enqueued ahead of the halt left their tickets untouched
    if (rc) return rc;
  }
  L.t_total += seconds_since(t_run);
  return PGO_OK;
}

int lm_end(pgo_problem* P, pgo_solver_summary* summary, pgo_iteration_record* records, int capacity) {
  LmState& L = P->lm;
  if (!L.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_end without pgo_solver_begin");
  if (L.gmax_deferred) {
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, P->stream);
    HIP_TRY(hipStreamSynchronize(P->stream));
    L.gmax = P->scal->gradient_max;
    L.cur.gradient_max_norm = L.gmax;
    if (!L.pending_record && !L.records.empty()) L.records.back().gradient_max_norm = L.gmax;
    L.gmax_deferred = false;
  }
  if (L.pending_record) {
    if (L.cur.step_is_successful) ++L.num_successful; else ++L.num_unsuccessful;
    L.cur.trust_region_radius = L.radius;
    L.records.push_back(L.cur);
    L.pending_record = false;
  }
  if (!L.terminated) terminate(L, PGO_NO_CONVERGENCE, 5, "Stepping stopped by the caller after %d iterations.", L.cur.iteration);
  if (P->dsym.hybrid && P->direct_usable && getenv("PGO_VERBOSE"))
    std::fprintf(stderr, "[pgo] exact request, per-iteration choice: %d factorisations, %d PCG solves within budget, %d over budget (redone)\n",
                 L.hybrid_direct, L.hybrid_pcg_ok, L.hybrid_pcg_over);
  int rc = download_poses(P, P->g.pose_x);
  if (rc) return rc;
  if (summary) {
    memset(summary, 0, sizeof *summary);
    summary->termination_type = L.termination;
    summary->reason = L.reason;
    summary->num_successful_steps = L.num_successful;
    summary->num_unsuccessful_steps = L.num_unsuccessful;
    summary->num_iterations = (int)L.records.size();
    summary->num_linear_solver_iterations = L.num_linear_iterations;
    summary->num_poses = P->g.N;
    summary->num_edges = P->g.E;
    const bool want_exact = P->opt.linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY;
    summary->linear_solver_used = want_exact ? (P->direct_usable ? (P->dsym.hybrid ? 3 : 0) : 2) : 1;
    const bool fronts = P->front_usable || P->sfront_usable;
    summary->factor_nnz_blocks = (want_exact && P->direct_usable) ? (fronts ? (int)std::min<long long>(P->fsym.factor_blocks, 0x7fffffff) : P->dsym.nb) : 0;
    summary->factor_levels = (want_exact && P->direct_usable) ? (fronts ? P->fsym.n_levels : P->dsym.n_levels) : 0;
    {
      int const_p = 0, const_q = 0;
      for (uint8_t m : P->cmask) { const_p += m & 1; const_q += (m >> 1) & 1; }
      summary->num_parameter_blocks_reduced = 2 * P->g.N - const_p - const_q;
      summary->num_parameters_reduced = 7 * P->g.N - 3 * const_p - 4 * const_q;
      summary->num_effective_parameters_reduced = 6 * P->g.N - 3 * const_p - 3 * const_q;
    }
    summary->factor_kind = (want_exact && P->direct_usable) ? (P->sfront_usable ? 3 : P->front_usable ? 2 : 1) : 0;
    summary->factor_max_front = (want_exact && fronts) ? P->fsym.max_front : 0;
    summary->factor_flops = (want_exact && P->direct_usable) ? (fronts ? P->fsym.flops : P->dsym.flops) : 0.0;
    summary->num_factorizations = L.n_factorizations;
    summary->initial_cost = L.initial_cost;
    summary->final_cost = L.x_cost;
    summary->total_time_in_seconds = L.t_total;
    summary->setup_time_in_seconds = L.t_setup;
    summary->linear_solver_time_in_seconds = L.t_linear;
    summary->jacobian_evaluation_time_in_seconds = L.t_jacobian;
    summary->residual_evaluation_time_in_seconds = L.t_residual;
    summary->final_gradient_max_norm = L.gmax;
    summary->final_trust_region_radius = L.radius;
    snprintf(summary->message, sizeof summary->message, "%s", L.message.c_str());
  }
  if (records) {
    const int n = std::min(capacity, (int)L.records.size());
    for (int i = 0; i < n; ++i) records[i] = L.records[i];
  }
  L.active = false;
  return PGO_OK;
}


// ---- batched solve of independent graphs -----------------------------------------------------------------------------------
// KITTI-scale graphs do not fill the machine: one LM iteration is a chain of ~45 small dependent launches.  n independent
// problems are therefore solved as ONE block-diagonal problem — the same launch sequence, n times the work per launch — with
// everything Levenberg-Marquardt decides kept per component: trust-region radius, accept / reject, every termination test,
// iteration records and summaries.  Device: damping with the radius of the pose's component, the factorisation of the union
// (a forest: nothing crosses components), per-component step scalars, acceptance by component.  Host: lm_pre_step /
// lm_post_step per component — the very functions the single-problem driver runs, so a component follows the trace it
// follows when solved alone (to the rounding of the differently grouped sums).  Exact steps only (SPARSE_NORMAL_CHOLESKY,
// the reference's setting): a per-component CG would need per-component iteration control.
int solve_batch(pgo_problem* const* probs, int n, const pgo_solver_options* options, pgo_solver_summary* summaries,
                pgo_iteration_record* records, int capacity) {
  const auto t_begin = Clock::now();
  const pgo_solver_options& o = *options;
  if (o.linear_solver_type != PGO_SPARSE_NORMAL_CHOLESKY)
    return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch serves exact requests (PGO_SPARSE_NORMAL_CHOLESKY) only");
  if (!probs || n <= 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: no problems");
  for (int c = 0; c < n; ++c) {
    if (!probs[c] || probs[c]->pp.empty()) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: problem %d is null or empty", c);
    if (probs[c]->device != probs[0]->device)
      return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: problem %d lives on device %d, problem 0 on device %d (one batch = one GPU)", c,
                       probs[c]->device, probs[0]->device);
  }
  // ---- the union ----
  pgo_problem M;
  M.device = probs[0]->device;
  M.loss_kind = probs[0]->loss_kind;
  M.loss_a = probs[0]->loss_a;
  std::vector<int> pose_begin(n + 1, 0), edge_begin(n + 1, 0);
  bool any_info = false;
  for (int c = 0; c < n; ++c) {
    const pgo_problem* Q = probs[c];
    if (!Q || Q->pp.empty()) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: problem %d is null or empty", c);
    if (Q->comm) return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: problem %d is attached to a communicator", c);
    if (Q->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solve_batch: problem %d is inside a solver session", c);
    if (Q->loss_kind != M.loss_kind || Q->loss_a != M.loss_a)
      return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: all problems must use the same loss function (problem %d differs)", c);
    pose_begin[c + 1] = pose_begin[c] + (int)Q->pp.size();
    edge_begin[c + 1] = edge_begin[c] + (int)Q->ia.size();
    any_info = any_info || Q->has_info;
  }
  const int N = pose_begin[n], E = edge_begin[n];
  static const bool verbose = getenv("PGO_VERBOSE") != nullptr;
  auto mark = [&](const char* what) {
    if (verbose) std::fprintf(stderr, "[pgo] batch: %-34s at %.2f ms\n", what, 1e3 * seconds_since(t_begin));
  };
  M.pp.reserve(N); M.qq.reserve(N); M.cmask.reserve(N);
  M.ia.reserve(E); M.ib.reserve(E); M.meas.reserve((size_t)7 * E);
  if (any_info) M.sqrt_info.reserve((size_t)36 * E);
  M.has_info = any_info;
  for (int c = 0; c < n; ++c) {
    const pgo_problem* Q = probs[c];
    M.pp.insert(M.pp.end(), Q->pp.begin(), Q->pp.end());
    M.qq.insert(M.qq.end(), Q->qq.begin(), Q->qq.end());
    M.cmask.insert(M.cmask.end(), Q->cmask.begin(), Q->cmask.end());
    for (int v : Q->ia) M.ia.push_back(v + pose_begin[c]);
    for (int v : Q->ib) M.ib.push_back(v + pose_begin[c]);
    M.meas.insert(M.meas.end(), Q->meas.begin(), Q->meas.end());
    if (any_info) {
      if (Q->has_info) M.sqrt_info.insert(M.sqrt_info.end(), Q->sqrt_info.begin(), Q->sqrt_info.end());
      else
        for (size_t e = 0; e < Q->ia.size(); ++e)
          for (int k = 0; k < 36; ++k) M.sqrt_info.push_back(k % 7 == 0 ? 1.0 : 0.0);
    }
  }
  pgo_problem* P = &M;
  mark("union built");
  P->no_sfront = true;
  P->want_direct = true;       // the union's host analysis runs beside the array fills and uploads of prepare()
  int rc = prepare(P);
  P->want_direct = false;
  if (rc) return rc;
  mark("prepare");
  P->opt = o;
  P->g.loss_kind = P->loss_kind;
  P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p;
  P->g.pose_c = P->d_pose_c.p;
  rc = prepare_direct(P);
  if (rc) return rc;
  mark("prepare_direct");
  if (!P->direct_usable || P->dsym.hybrid)
    return set_error(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: the union of the problems is beyond the factorisation's budget; solve them one by one");
  P->split_two_launch = true;   // the device-wide failure flag is not consulted per component: no in-kernel waits in a batch
  P->sfront_levels = true;
  P->front_launches = true;
  rc = prepare_clusters(P, 1);
  if (rc) return rc;
  mark("prepare_clusters");
  hipStream_t s = P->stream;
  // component tables (device) and the per-component hand-over block (pinned, device visible)
  std::vector<int> pose_comp(N);
  for (int c = 0; c < n; ++c) std::fill(pose_comp.begin() + pose_begin[c], pose_comp.begin() + pose_begin[c + 1], c);
  DevBuf<int> d_pose_begin, d_edge_begin, d_pose_comp;
  HIP_TRY(d_pose_begin.upload(pose_begin, s));
  HIP_TRY(d_edge_begin.upload(edge_begin, s));
  HIP_TRY(d_pose_comp.upload(pose_comp, s));
  struct Pinned {
    void* p = nullptr;
    size_t cap = 0;
    hipStream_t s = nullptr;
    ~Pinned() {
      if (!p) return;
      (void)hipStreamSynchronize(s);          // an error return may leave kernels that write the block in flight
      host_side_pool().put_pinned(p, cap);
    }
  } pin;
  pin.s = s;
  const size_t pin_bytes = (size_t)n * (sizeof(pgo::BatchScalars) + sizeof(double) + sizeof(int)) + 64;
  HIP_TRY(host_side_pool().get_pinned(pin_bytes, &pin.p, &pin.cap));
  memset(pin.p, 0, pin_bytes);
  pgo::BatchScalars* out = static_cast<pgo::BatchScalars*>(pin.p);
  double* radius = reinterpret_cast<double*>(out + n);
  int* accept = reinterpret_cast<int*>(radius + n);
  // workgroups per component of the scalar reduction: enough of them to cover the machine, no more than the largest component needs
  int max_items = 1;
  for (int c = 0; c < n; ++c) max_items = std::max(max_items, std::max(pose_begin[c + 1] - pose_begin[c], edge_begin[c + 1] - edge_begin[c]));
  const int split = std::max(1, std::min((max_items + 255) / 256, std::max(1, 1024 / n)));
  DevBuf<double> d_partial;
  HIP_TRY(d_partial.alloc((size_t)5 * n * split));
  const pgo::BatchPlan plan{n, d_pose_begin.p, d_edge_begin.p, d_pose_comp.p, radius, accept, out, d_partial.p, split};

  mark("component tables");
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(P->g.pose_c, P->g.pose_x, P->d_pose_x.n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIP_TRY(P->d_flags.zero(s));
  HIP_TRY(P->d_cg_x.zero(s));
  HIP_TRY(P->d_cg_q.zero(s));
  HIP_TRY(P->d_cg_b.zero(s));
  HIP_TRY(P->d_d2.zero(s));
  mark("poses uploaded");
  const double t_setup = seconds_since(t_begin);
  // Init + IterationZero: cost, state norm and gradient norm of every component at its start (candidate == current point)
  rc = evaluate_gradient_and_jacobian(P, true);
  if (rc) return rc;
  pgo::launch_batch_scalars(P->g, plan, s);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  mark("iteration zero");
  std::vector<LmState> Ls(n);
  for (int c = 0; c < n; ++c) {
    LmState& L = Ls[c];
    L.x_cost = out[c].cand_cost;
    L.initial_cost = L.x_cost;
    L.x_norm = std::sqrt(out[c].x_norm_sq);
    L.gmax = out[c].gradient_max;
    L.radius = o.initial_trust_region_radius;
    L.cur = pgo_iteration_record{};
    L.cur.step_is_successful = 1;
    L.cur.cost = L.x_cost;
    L.cur.gradient_max_norm = L.gmax;
    L.pending_record = true;
    L.active = true;
    L.t_setup = t_setup;
    if (!std::isfinite(L.x_cost)) terminate(L, PGO_FAILURE, 7, "Initial cost is not finite.");
    radius[c] = L.radius;
  }
  int n_rounds = 0;
  for (;;) {
    int alive = 0;
    for (int c = 0; c < n; ++c) {
      LmState& L = Ls[c];
      if (L.terminated) continue;
      if (lm_pre_step(L, o)) { ++alive; radius[c] = L.radius; }
    }
    if (!alive) break;
    // the trial step of every component at once
    pgo::launch_batch_d2(P->g, plan, o.min_lm_diagonal, o.max_lm_diagonal, s);
    rc = damping_all(P, 1.0, o.min_lm_diagonal, o.max_lm_diagonal, 2);
    if (rc) return rc;
    rc = run_direct(P);
    if (rc) return rc;
    const pgo::CgParams none{0.0, -1.0, 0, 0};
    pgo::launch_spmv_tail(P->g, none, s, 0, 1);
    pgo::launch_batch_scalars(P->g, plan, s);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    ++n_rounds;
    bool any_accept = false;
    for (int c = 0; c < n; ++c) {
      LmState& L = Ls[c];
      accept[c] = 0;
      if (L.terminated) continue;
      // a pivot that fails inside one component leaves NaNs in that component's step only (nothing crosses components):
      // its model change is not finite and the step is handled as invalid; the device-wide flag is not consulted
      const StepScalars sc{out[c].cand_cost, out[c].model_change, out[c].step_norm_sq, out[c].x_norm_sq, out[c].gradient_max, 0, 0, 0};
      ++L.n_factorizations;
      if (lm_post_step(L, o, sc, 0) == STEP_ACCEPT) { accept[c] = 1; any_accept = true; }
    }
    if (any_accept) {
      pgo::launch_batch_accept(P->g, plan, s);
      rc = evaluate_gradient_and_jacobian(P, false);   // every component: the unchanged ones reproduce their values bit for bit
      if (rc) return rc;
    }
  }
  // gradient norm of the points accepted last (deferred like in the single-problem driver)
  bool need_g = false;
  for (int c = 0; c < n; ++c) need_g = need_g || Ls[c].gmax_deferred;
  if (need_g) {
    HIP_TRY(hipMemcpyAsync(P->g.pose_c, P->g.pose_x, P->d_pose_x.n * sizeof(double), hipMemcpyDeviceToDevice, s));
    pgo::launch_batch_scalars(P->g, plan, s);
    HIP_TRY(hipStreamSynchronize(s));
  }
  mark("rounds done");
  rc = download_poses(P, P->g.pose_x);
  if (rc) return rc;
  mark("poses downloaded");
  HIP_TRY(hipMemsetAsync(P->d_flags.p, 0, P->d_flags.n * sizeof(int), s));
  const double t_total = seconds_since(t_begin);
  for (int c = 0; c < n; ++c) {
    LmState& L = Ls[c];
    if (L.gmax_deferred) {
      L.gmax = out[c].gradient_max;
      L.cur.gradient_max_norm = L.gmax;
      if (!L.pending_record && !L.records.empty()) L.records.back().gradient_max_norm = L.gmax;
      L.gmax_deferred = false;
    }
    if (L.pending_record) {
      if (L.cur.step_is_successful) ++L.num_successful; else ++L.num_unsuccessful;
      L.cur.trust_region_radius = L.radius;
      L.records.push_back(L.cur);
      L.pending_record = false;
    }
    if (summaries) {
      pgo_solver_summary* sm = summaries + c;
      memset(sm, 0, sizeof *sm);
      sm->termination_type = L.termination;
      sm->reason = L.reason;
      sm->num_successful_steps = L.num_successful;
      sm->num_unsuccessful_steps = L.num_unsuccessful;
      sm->num_iterations = (int)L.records.size();
      sm->num_poses = pose_begin[c + 1] - pose_begin[c];
      sm->num_edges = edge_begin[c + 1] - edge_begin[c];
      sm->linear_solver_used = 0;
      // the factorisation is the union's: fill, levels and flops are those of all components together
      const bool fronts = P->front_usable || P->sfront_usable;
      sm->factor_nnz_blocks = fronts ? (int)std::min<long long>(P->fsym.factor_blocks, 0x7fffffff) : P->dsym.nb;
      sm->factor_levels = fronts ? P->fsym.n_levels : P->dsym.n_levels;
      int const_p = 0, const_q = 0;
      for (int v = pose_begin[c]; v < pose_begin[c + 1]; ++v) { const_p += P->cmask[v] & 1; const_q += (P->cmask[v] >> 1) & 1; }
      sm->num_parameter_blocks_reduced = 2 * sm->num_poses - const_p - const_q;
      sm->num_parameters_reduced = 7 * sm->num_poses - 3 * const_p - 4 * const_q;
      sm->num_effective_parameters_reduced = 6 * sm->num_poses - 3 * const_p - 3 * const_q;
      sm->factor_kind = P->sfront_usable ? 3 : P->front_usable ? 2 : 1;
      sm->factor_max_front = fronts ? P->fsym.max_front : 0;
      sm->factor_flops = fronts ? P->fsym.flops : P->dsym.flops;
      sm->num_factorizations = L.n_factorizations;
      sm->initial_cost = L.initial_cost;
      sm->final_cost = L.x_cost;
      sm->total_time_in_seconds = t_total;        // of the whole batch
      sm->setup_time_in_seconds = t_setup;
      sm->final_gradient_max_norm = L.gmax;
      sm->final_trust_region_radius = L.radius;
      snprintf(sm->message, sizeof sm->message, "%s", L.message.c_str());
    }
    if (records) {
      const int k = std::min(capacity, (int)L.records.size());
      for (int i = 0; i < k; ++i) records[(size_t)c * capacity + i] = L.records[i];
    }
  }
  if (getenv("PGO_VERBOSE"))
    std::fprintf(stderr, "[pgo] batch: %d problems, %d poses, %d edges, %d rounds, setup %.2f ms, total %.2f ms\n", n, N, E, n_rounds, 1e3 * t_setup, 1e3 * t_total);
  HIP_TRY(hipStreamSynchronize(s));   // the component tables below go back to the pool
  return PGO_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
// error channel for the other translation units of the library
int pgo_candidates_set_error(int code, const char* msg) { return set_error(code, "%s", msg); }

extern "C" {

int pgo_version(void) { return PGO_VERSION; }
const char* pgo_last_error(void) { return g_error.c_str(); }

int pgo_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return count;
}

static int g_default_device = 0;
int pgo_set_device(int device) {
  if (device < 0) return set_error(PGO_ERR_INVALID_ARGUMENT, "negative device index");
  g_default_device = device;
  return PGO_OK;
}

pgo_problem* pgo_problem_create(void) {
  pgo_problem* p = new (std::nothrow) pgo_problem();
  if (p) p->device = g_default_device;
  return p;
}
void pgo_problem_destroy(pgo_problem* problem) { delete problem; }

int pgo_problem_add_pose(pgo_problem* P, double* p, double* q) {
  if (!P || !p || !q) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_problem_add_pose");
  auto ip = P->block_of_ptr.find(p), iq = P->block_of_ptr.find(q);
  if (ip != P->block_of_ptr.end() || iq != P->block_of_ptr.end()) {
    if (ip != P->block_of_ptr.end() && iq != P->block_of_ptr.end() && (ip->second >> 1) == (iq->second >> 1) &&
        (ip->second & 1) == 0 && (iq->second & 1) == 1)
      return ip->second >> 1;
    return set_error(PGO_ERR_UNSUPPORTED, "a parameter block is already paired with a different translation/rotation block");
  }
  const int idx = (int)P->pp.size();
  P->pp.push_back(p);
  P->qq.push_back(q);
  P->cmask.push_back(0);
  P->block_of_ptr[p] = 2 * idx;
  P->block_of_ptr[q] = 2 * idx + 1;
  P->topo_dirty = true;
  return idx;
}

int pgo_problem_add_poses(pgo_problem* P, int n, double* base, int stride) {
  if (!P || !base || n < 0 || stride < 7) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_add_poses");
  const int first = (int)P->pp.size();
  P->pp.reserve(first + n); P->qq.reserve(first + n); P->cmask.reserve(first + n);
  P->block_of_ptr.reserve((size_t)2 * (first + n));
  for (int i = 0; i < n; ++i) {
    const int r = pgo_problem_add_pose(P, base + (size_t)i * stride, base + (size_t)i * stride + 3);
    if (r < 0) return r;
  }
  return first;
}

int pgo_problem_add_se3_between_batch(pgo_problem* P, int n, const int* begin, const int* end, const double* t_be,
                                      const double* sqrt_information) {
  if (!P || n < 0 || (n > 0 && (!begin || !end || !t_be))) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_add_se3_between_batch");
  const int N = (int)P->pp.size();
  for (int i = 0; i < n; ++i) {
    if (begin[i] < 0 || begin[i] >= N || end[i] < 0 || end[i] >= N)
      return set_error(PGO_ERR_INVALID_ARGUMENT, "edge %d references a pose that was never added", i);
    if (begin[i] == end[i]) return set_error(PGO_ERR_INVALID_ARGUMENT, "edge %d connects a pose to itself", i);
  }
  const int first = (int)P->ia.size();
  if (sqrt_information && !P->has_info) {
    // earlier edges used the identity
    P->sqrt_info.assign((size_t)36 * first, 0.0);
    for (int e = 0; e < first; ++e) for (int d = 0; d < 6; ++d) P->sqrt_info[(size_t)36 * e + 7 * d] = 1.0;
    P->has_info = true;
  }
  P->ia.insert(P->ia.end(), begin, begin + n);
  P->ib.insert(P->ib.end(), end, end + n);
  P->meas.insert(P->meas.end(), t_be, t_be + (size_t)7 * n);
  if (P->has_info) {
    if (sqrt_information) {
      P->sqrt_info.insert(P->sqrt_info.end(), sqrt_information, sqrt_information + (size_t)36 * n);
    } else {
      const size_t old = P->sqrt_info.size();
      P->sqrt_info.resize(old + (size_t)36 * n, 0.0);
      for (int e = 0; e < n; ++e) for (int d = 0; d < 6; ++d) P->sqrt_info[old + (size_t)36 * e + 7 * d] = 1.0;
    }
  }
  P->topo_dirty = true;
  return first;
}

int pgo_problem_add_se3_between(pgo_problem* P, int pose_begin, int pose_end, const double* t_be_p, const double* t_be_q,
                                const double* sqrt_information) {
  if (!P || !t_be_p || !t_be_q) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_problem_add_se3_between");
  double t[7] = {t_be_p[0], t_be_p[1], t_be_p[2], t_be_q[0], t_be_q[1], t_be_q[2], t_be_q[3]};
  return pgo_problem_add_se3_between_batch(P, 1, &pose_begin, &pose_end, t, sqrt_information);
}

int pgo_problem_set_loss(pgo_problem* P, int kind, double a) {
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  if (kind < PGO_LOSS_TRIVIAL || kind > PGO_LOSS_SWITCHABLE) return set_error(PGO_ERR_UNSUPPORTED, "unknown loss kind %d", kind);
  if (kind != PGO_LOSS_TRIVIAL && !(a > 0.0)) return set_error(PGO_ERR_INVALID_ARGUMENT, "loss scale must be positive");
  P->loss_kind = kind;
  P->loss_a = a;
  return PGO_OK;
}

int pgo_problem_set_pose_constant(pgo_problem* P, int pose, int which) {
  if (!P || pose < 0 || pose >= (int)P->pp.size() || (which & ~3) || which == 0)
    return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_problem_set_pose_constant");
  P->cmask[pose] |= (uint8_t)which;
  P->topo_dirty = true;
  return PGO_OK;
}

int pgo_problem_set_parameter_block_constant(pgo_problem* P, const double* block) {
  if (!P || !block) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument");
  auto it = P->block_of_ptr.find(block);
  if (it == P->block_of_ptr.end()) return set_error(PGO_ERR_INVALID_ARGUMENT, "parameter block not found in the problem");
  return pgo_problem_set_pose_constant(P, it->second >> 1, (it->second & 1) ? 2 : 1);
}

int pgo_problem_num_poses(const pgo_problem* P) { return P ? (int)P->pp.size() : 0; }
int pgo_problem_num_edges(const pgo_problem* P) { return P ? (int)P->ia.size() : 0; }

void pgo_solver_options_init(pgo_solver_options* o) {
  memset(o, 0, sizeof *o);
  o->max_num_iterations = 50;
  o->linear_solver_type = PGO_SPARSE_NORMAL_CHOLESKY;
  o->jacobi_scaling = 1;
  o->max_linear_solver_iterations = 500;
  o->min_linear_solver_iterations = 0;
  o->max_num_consecutive_invalid_steps = 5;
  o->cg_batch = 0;
  o->pcg_cluster_poses = 1;
  o->cg_residual_reset_period = 10;   // LinearSolver::Options::residual_reset_period of Ceres 1.13
  o->reserved0 = 0;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->eta = 1e-1;
  o->exact_r_tolerance = 1e-13;
}

int pgo_solver_begin(pgo_problem* P, const pgo_solver_options* options) {
  if (!P || !options) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_solver_begin");
  return lm_begin(P, options);
}

int pgo_solver_step(pgo_problem* P, int n, int* executed, int* done) {
  if (!P || !P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_step without pgo_solver_begin");
  int ran = 0;
  if (P->pipelined || P->universal) {
    int rc = P->universal ? lm_run_universal(P, n, &ran) : lm_run_pipelined(P, n, &ran);
    if (rc) { P->lm.active = false; return rc; }
    if (executed) *executed = ran;
    if (done) *done = P->lm.terminated ? 1 : 0;
    return PGO_OK;
  }
  for (int i = 0; i < n && !P->lm.terminated; ++i) {
    const int solves_before = P->lm.num_trial_steps;
    int rc = lm_advance(P);
    if (rc) { P->lm.active = false; return rc; }   // a failed session is closed: the device state is not trustworthy any more
    // an iteration counts when a trial step was computed (a pure termination check does not), also when that step ended the
    // run on the parameter or function tolerance
    if (P->lm.num_trial_steps != solves_before) ++ran;
  }
  if (executed) *executed = ran;
  if (done) *done = P->lm.terminated ? 1 : 0;
  return PGO_OK;
}

int pgo_solver_reset(pgo_problem* P) {
  if (!P || !P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_solver_reset without pgo_solver_begin");
  LmState& L = P->lm;
  HIP_TRY(hipMemcpyAsync(P->g.pose_x, P->d_pose_0.p, P->d_pose_0.n * sizeof(double), hipMemcpyDeviceToDevice, P->stream));
  int rc = evaluate_gradient_and_jacobian(P, true);
  if (rc) return rc;
  pgo::launch_cost(P->g, P->g.pose_x, 0, P->stream);
  pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, P->stream);
  HIP_TRY(hipStreamSynchronize(P->stream));
  L.x_cost = P->scal->cand_cost;
  L.gmax = P->scal->gradient_max;
  L.radius = P->opt.initial_trust_region_radius;
  L.decrease_factor = 2.0;
  L.reuse_diagonal = false;
  L.terminated = false;
  L.gmax_deferred = false;
  L.num_consecutive_invalid = 0;
  pgo_iteration_record r{};
  r.iteration = 0;  // the iteration budget restarts with the state
  r.step_is_successful = 1;
  r.cost = L.x_cost;
  r.gradient_max_norm = L.gmax;
  L.cur = r;
  L.pending_record = false;
  P->pipe_dirty = true;
  return PGO_OK;
}

int pgo_solver_end(pgo_problem* P, pgo_solver_summary* summary, pgo_iteration_record* records, int capacity) {
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  return lm_end(P, summary, records, capacity);
}

int pgo_solve(pgo_problem* P, const pgo_solver_options* options, pgo_solver_summary* summary,
              pgo_iteration_record* records, int capacity) {
  if (!P || !options) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_solve");
  int rc = lm_begin(P, options);
  if (rc) return rc;
  while (!P->lm.terminated) {
    rc = P->universal ? lm_run_universal(P, -1, nullptr) : P->pipelined ? lm_run_pipelined(P, -1, nullptr) : lm_advance(P);
    if (rc) { P->lm.active = false; return rc; }   // caller memory keeps the poses it came with; the session is closed
  }
  return lm_end(P, summary, records, capacity);
}

int pgo_release_device_memory(void) {
  device_pool().trim();
  host_side_pool().trim();
  return PGO_OK;
}

int pgo_solve_batch(pgo_problem* const* problems, int n_problems, const pgo_solver_options* options, pgo_solver_summary* summaries,
                    pgo_iteration_record* records, int capacity) {
  if (!problems || n_problems <= 0 || !options || (records && capacity < 0)) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_solve_batch");
  const auto t0 = Clock::now();
  const int rc = solve_batch(problems, n_problems, options, summaries, records, capacity);
  if (getenv("PGO_VERBOSE")) std::fprintf(stderr, "[pgo] batch: call returned after %.2f ms (the union's device memory released)\n", 1e3 * seconds_since(t0));
  return rc;
}

int pgo_summary_is_solution_usable(const pgo_solver_summary* s) {
  return s && (s->termination_type == PGO_CONVERGENCE || s->termination_type == PGO_NO_CONVERGENCE) ? 1 : 0;
}

size_t pgo_summary_full_report(const pgo_solver_summary* s, const pgo_iteration_record* rec, int n_rec, char* buffer, size_t capacity) {
  std::string r;
  char line[512];
  auto add = [&](const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    r += line;
  };
  static const char* term[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
  // Layout of Ceres 1.13 Solver::Summary::FullReport (solver.cc): two columns Original / Reduced, Given / Used, the same
  // row labels and widths, so that parsers of the reference's stdout (finial.cpp:541) keep working; what Ceres has no row
  // for (factorisation statistics, device) follows in a block of its own.
  add("\nSolver Summary (v %d.%d.%d-hip-gfx950)\n\n", PGO_VERSION / 100, (PGO_VERSION / 10) % 10, PGO_VERSION % 10);
  add("%45s    %21s\n", "Original", "Reduced");
  add("Parameter blocks    % 25d% 25d\n", 2 * s->num_poses, s->num_parameter_blocks_reduced);
  add("Parameters          % 25d% 25d\n", 7 * s->num_poses, s->num_parameters_reduced);
  add("Effective parameters% 25d% 25d\n", 6 * s->num_poses, s->num_effective_parameters_reduced);
  add("Residual blocks     % 25d% 25d\n", s->num_edges, s->num_edges);
  add("Residual            % 25d% 25d\n", 6 * s->num_edges, 6 * s->num_edges);
  add("\nMinimizer                 %19s\n", "TRUST_REGION");
  add("\nSparse linear algebra library %15s\n", "HIP_GFX950");
  add("Trust region strategy     %19s\n", "LEVENBERG_MARQUARDT");
  add("\n%45s    %21s\n", "Given", "Used");
  const bool exact = s->linear_solver_used != 1;
  add("Linear solver       %25s%25s\n", exact ? "SPARSE_NORMAL_CHOLESKY" : "CGNR",
      s->linear_solver_used == 2 ? "CGNR" : exact ? "SPARSE_NORMAL_CHOLESKY" : "CGNR");
  if (!exact || s->linear_solver_used == 2) add("Preconditioner      %25s%25s\n", "JACOBI", "JACOBI");
  add("Threads             % 25d% 25d\n", 1, 1);
  add("Linear solver threads % 23d% 25d\n", 1, 1);
  if (exact) add("Linear solver ordering %22s% 25d\n", "AUTOMATIC", s->num_parameter_blocks_reduced);
  add("\nCost:\n");
  add("Initial        % 30e\n", s->initial_cost);
  add("Final          % 30e\n", s->final_cost);
  add("Change         % 30e\n", s->initial_cost - s->final_cost);
  add("\nMinimizer iterations         % 16d\n", s->num_iterations);
  add("Successful steps             % 16d\n", s->num_successful_steps);
  add("Unsuccessful steps           % 16d\n", s->num_unsuccessful_steps);
  add("\nTime (in seconds):\n");
  add("Preprocessor        %25.4f\n", s->setup_time_in_seconds);
  add("\n  Residual evaluation %23.4f\n", s->residual_evaluation_time_in_seconds);
  add("  Jacobian evaluation %23.4f\n", s->jacobian_evaluation_time_in_seconds);
  add("  Linear solver       %23.4f\n", s->linear_solver_time_in_seconds);
  add("Minimizer           %25.4f\n", s->total_time_in_seconds);
  add("\nPostprocessor       %25.4f\n", 0.0);
  add("Total               %25.4f\n", s->total_time_in_seconds + s->setup_time_in_seconds);
  static const char* ls[] = {"GPU factorisation", "block-Jacobi PCG (Q-tolerance eta)", "PCG to exact_r_tolerance (factorisation declined)",
                             "GPU factorisation or PCG to exact_r_tolerance, chosen per iteration"};
  add("\nGPU path (no Ceres counterpart):\n");
  add("Compute device              HIP gfx950 (FP64)\n");
  add("Linear solves served by     %s\n", ls[(s->linear_solver_used >= 0 && s->linear_solver_used <= 3) ? s->linear_solver_used : 1]);
  add("Linear solver iterations     % 16d\n", s->num_linear_solver_iterations);
  if (s->factor_nnz_blocks > 0) {
    add("Factorisation               %s\n", s->factor_kind == 3 ? "supernodal multifrontal, fronts in LDS" : s->factor_kind == 2 ? "supernodal multifrontal, FP64 MFMA fronts" : "enumerated 6x6 block pairs, nested dissection");
    add("Factor blocks / levels       % 16d / %d\n", s->factor_nnz_blocks, s->factor_levels);
    add("Factorisations               % 16d\n", s->num_factorizations);
  }
  add("\n");
  if (rec && n_rec > 0) {
    add("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter\n");
    for (int i = 0; i < n_rec; ++i)
      add("%4d %.6e %10.2e %10.2e %9.2e %10.2e %9.2e %8d\n", rec[i].iteration, rec[i].cost, rec[i].cost_change,
          rec[i].gradient_max_norm, rec[i].step_norm, rec[i].relative_decrease, rec[i].trust_region_radius,
          rec[i].linear_solver_iterations);
    add("\n");
  }
  const int t = (s->termination_type >= 0 && s->termination_type <= 2) ? s->termination_type : 2;
  add("Termination: %24s (%s)\n", term[t], s->message);   // "Termination:   %25s (%s)" in Ceres, same tokens
  if (buffer && capacity) {
    const size_t n = std::min(capacity - 1, r.size());
    memcpy(buffer, r.data(), n);
    buffer[n] = 0;
  }
  return r.size() + 1;
}

// ---- evaluation entry points -------------------------------------------------------------------
int pgo_evaluate(pgo_problem* P, double* cost, double* residuals, double* jac_begin, double* jac_end, double* gradient) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_evaluate during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.loss_kind = P->loss_kind; P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  const int E = P->g.E, N = P->g.N;
  if (residuals || jac_begin || jac_end) {
    if (residuals) HIP_TRY(P->d_tmp_a.alloc((size_t)6 * E));
    if (jac_begin) HIP_TRY(P->d_tmp_b.alloc((size_t)36 * E));
    if (jac_end) HIP_TRY(P->d_tmp_c.alloc((size_t)36 * E));
    if (E > 0) pgo::launch_evaluate_edges(P->g, P->g.pose_x, residuals ? P->d_tmp_a.p : nullptr, jac_begin ? P->d_tmp_b.p : nullptr,
                               jac_end ? P->d_tmp_c.p : nullptr, s);
    if (residuals && E) HIP_TRY(staged_d2h(residuals, P->d_tmp_a.p, sizeof(double) * 6 * E, s));
    if (jac_begin && E) HIP_TRY(staged_d2h(jac_begin, P->d_tmp_b.p, sizeof(double) * 36 * E, s));
    if (jac_end && E) HIP_TRY(staged_d2h(jac_end, P->d_tmp_c.p, sizeof(double) * 36 * E, s));
  }
  if (cost) {
    pgo::launch_cost(P->g, P->g.pose_x, 0, s);
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s);
  }
  if (gradient) {
    rc = fill_scale_one(P);
    if (rc) return rc;
    rc = linearize_all(P);
    if (rc) return rc;
    HIP_TRY(staged_d2h(gradient, P->g.grad, sizeof(double) * 6 * N, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (cost) *cost = P->scal->cand_cost;
  return PGO_OK;
}

int pgo_normal_equations(pgo_problem* P, double* diag, double* offdiag, double* gradient) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_normal_equations during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P) return set_error(PGO_ERR_INVALID_ARGUMENT, "null problem");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.loss_kind = P->loss_kind; P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  rc = fill_scale_one(P);
  if (rc) return rc;
  rc = linearize_all(P);
  if (rc) return rc;
  const int N = P->g.N, E = P->g.E;
  if (diag) HIP_TRY(staged_d2h(diag, P->g.Hdiag, sizeof(double) * 36 * N, s));
  if (gradient) HIP_TRY(staged_d2h(gradient, P->g.grad, sizeof(double) * 6 * N, s));
  std::vector<double> bsr;
  if (offdiag) {
    bsr.resize((size_t)P->g.n_slots * 36);
    HIP_TRY(staged_d2h(bsr.data(), P->g.bsr_val, bsr.size() * sizeof(double), s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (offdiag)
    for (int e = 0; e < E; ++e) {
      const int t = P->edge_begin_slot[e];
      for (int k = 0; k < 36; ++k) {
        const int pos = pgo::bsr_pos(P->g.blk_packed, pgo::SIDE_BEGIN, k);
        offdiag[(size_t)36 * e + k] = pos < 0 ? 0.0 : bsr[pgo::bsr_index(t, pos)];
      }
    }
  return PGO_OK;
}

int pgo_linear_solve(pgo_problem* P, const pgo_solver_options* options, const double* d2, const double* b, double* x, int* iterations) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_linear_solve during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P || !options || !d2 || !b || !x) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_linear_solve");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.loss_kind = P->loss_kind; P->g.loss_a = P->loss_a;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  rc = fill_scale_one(P);
  if (rc) return rc;
  const size_t m = (size_t)6 * P->g.N;
  rc = prepare_clusters(P, options->pcg_cluster_poses);
  if (rc) return rc;
  rc = linearize_all(P);
  if (rc) return rc;
  HIP_TRY(staged_h2d(P->g.d2, d2, m * sizeof(double), s));
  HIP_TRY(staged_h2d(P->g.grad, b, m * sizeof(double), s));  // rhs = scale(=1) * grad
  HIP_TRY(hipStreamSynchronize(s));
  rc = damping_all(P, 1.0, 0.0, 0.0, 2);
  if (rc) return rc;
  int it = 0, status = 0;
  if (options->linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY) {
    rc = prepare_direct(P);
    if (rc) return rc;
  }
  if (options->linear_solver_type == PGO_SPARSE_NORMAL_CHOLESKY && P->direct_usable) {
    rc = run_direct(P);
    if (rc) return rc;
    pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s);
    HIP_TRY(hipStreamSynchronize(s));
    if ((P->scal->linearize_bad & 4) && (P->front_usable ? !P->front_launches : P->sfront_usable ? !P->sfront_levels : !P->split_two_launch)) {   // an in-kernel wait ran out (lm_advance)
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      rc = run_direct(P);
      if (rc) return rc;
      pgo::launch_finalize_scalars(P->g, P->g.n_edge_wg, s);
      HIP_TRY(hipStreamSynchronize(s));
    }
    if (P->scal->linearize_bad) status = 2;
  } else {
    P->opt.cg_residual_reset_period = options->cg_residual_reset_period;   // launch_cg_batch reads the refresh period from P->opt
    rc = run_pcg(P, cg_params_for(*options), options->cg_batch, &it, &status);
  }
  if (rc) return rc;
  HIP_TRY(staged_d2h(x, P->g.cg_x, m * sizeof(double), s));
  HIP_TRY(hipStreamSynchronize(s));
  if (iterations) *iterations = it;
  if (status == 2) return set_error(PGO_ERR_NUMERICAL, "PCG broke down with non-finite values");
  return PGO_OK;
}

int pgo_plus(pgo_problem* P, const double* delta) {
  if (P && P->lm.active) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_plus during a solver session: it would overwrite the device-resident LM state (pose buffers, Jacobi scaling, linearisation); call pgo_solver_end first");
  if (!P || !delta) return set_error(PGO_ERR_INVALID_ARGUMENT, "null argument to pgo_plus");
  int rc = prepare(P);
  if (rc) return rc;
  hipStream_t s = P->stream;
  P->g.pose_x = P->d_pose_x.p; P->g.pose_c = P->d_pose_c.p;
  rc = upload_poses(P, P->g.pose_x);
  if (rc) return rc;
  const size_t m = (size_t)6 * P->g.N;
  HIP_TRY(P->d_tmp_a.alloc(m));
  HIP_TRY(staged_h2d(P->d_tmp_a.p, delta, m * sizeof(double), s));
  pgo::launch_apply_step(P->g, P->d_tmp_a.p, s);
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return download_poses(P, P->g.pose_c);
}

int pgo_time_kernel(pgo_problem* P, const char* kernel, int repeats, double* avg_ms) {
  if (!P || !kernel || repeats <= 0 || !avg_ms) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_time_kernel");
  if (P->topo_dirty || !P->stream_ready) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel needs a prepared problem (call pgo_solver_begin first)");
  hipStream_t s = P->stream;
  const std::string k = kernel;
  pgo::CgParams prm = cg_params_for(P->opt);
  { const char* d = getenv("PGO_DEBUG"); P->g.debug = d ? atoi(d) : 0; }
  if (k == "evaluate") {
    HIP_TRY(P->d_tmp_a.alloc((size_t)6 * P->g.E));
    HIP_TRY(P->d_tmp_b.alloc((size_t)36 * P->g.E));
    HIP_TRY(P->d_tmp_c.alloc((size_t)36 * P->g.E));
  }
  // "pcg_spmv" repeats the SpMV kernel of CG iteration 1 (the update kernel never runs, so the
  // iteration counter stays put); "pcg_update" likewise repeats the update of iteration 1.
  if (k == "pcg_spmv" || k == "pcg_update" || k == "pcg_iteration") {
    pgo::launch_damping(P->g, P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    pgo::launch_pcg_init(P->g, s);
  }
  if (k == "pcg_update") pgo::launch_pcg_spmv_only(P->g, prm, 1, s);
  const bool wants_factor = k == "direct" || k == "front_factor" || k == "front_solve";
  if (wants_factor) {
    if (!P->direct_usable) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('%s'): no GPU factorisation prepared for this problem", kernel);
    if ((k != "direct") && !P->front_usable) return set_error(PGO_ERR_INVALID_ARGUMENT, "pgo_time_kernel('%s'): the multifrontal solver is not in use", kernel);
    pgo::launch_damping(P->g, P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    if (k == "front_solve") enqueue_front_factor(P, P->g);
  }
  auto once = [&]() -> int {
    if (k == "direct") return run_direct(P);
    if (k == "front_factor") { enqueue_front_factor(P, P->g); return 0; }
    if (k == "front_solve") { pgo::launch_front_solve(P->g, P->fplan, P->fsym, s, P->fsym.mixed ? &P->splan : nullptr); return 0; }
    if (k == "linearize") pgo::launch_linearize(P->g, s);
    else if (k == "cost") pgo::launch_cost(P->g, P->g.pose_x, 5, s);
    else if (k == "evaluate") pgo::launch_evaluate_edges(P->g, P->g.pose_x, P->d_tmp_a.p, P->d_tmp_b.p, P->d_tmp_c.p, s);
    else if (k == "spmv") pgo::launch_spmv_plain(P->g, s);
    else if (k == "pcg_spmv") pgo::launch_pcg_spmv_only(P->g, prm, 1, s);
    else if (k == "pcg_update") pgo::launch_pcg_update_only(P->g, 1, s);
    else if (k == "pcg_iteration") pgo::launch_pcg_iteration(P->g, prm, 1, s);
    else if (k == "empty") pgo::launch_debug(P->g, 0, s);
    else if (k == "touch") pgo::launch_debug(P->g, 1, s);
    else return -1;
    return 0;
  };
  if (k == "pcg_graph") {
    // average time of one CG iteration inside a captured batch with every stopping test disabled
    pgo::CgParams np{-1.0, -1.0, 1 << 30, 0};
    const int batch = 200;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    pgo::launch_damping(P->g, P->lm.radius, P->opt.min_lm_diagonal, P->opt.max_lm_diagonal, 0, s);
    double total = 0;
    for (int r = 0; r < repeats + 1; ++r) {
      pgo::launch_pcg_init(P->g, s);
      HIP_TRY(hipEventRecord(a, s));
      int rc = launch_cg_batch(P, np, batch);
      if (rc) return rc;
      HIP_TRY(hipEventRecord(b, s));
      HIP_TRY(hipEventSynchronize(b));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, a, b));
      if (r > 0) total += ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    P->drop_graph();
    *avg_ms = total / repeats / batch;
    return PGO_OK;
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  if (once() != 0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return set_error(PGO_ERR_INVALID_ARGUMENT, "unknown kernel '%s'", kernel); }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipEventRecord(e0, s));
  for (int i = 0; i < repeats; ++i) once();
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / repeats;
  if (wants_factor) {
    // the factorisations report through flags[2] (bit 0: pivot, bit 1: an in-kernel wait of a single-launch form ran out); nobody
    // folds it here, so read and clear it: a timed-out wait means the figure is worthless and the next LM iteration must not
    // inherit the bit
    int f2 = 0;
    HIP_TRY(hipMemcpyAsync(&f2, P->d_flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(P->d_flags.p + 2, 0, sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));
    if (f2 & 2) {
      if (P->front_usable) P->front_launches = true; else if (P->sfront_usable) P->sfront_levels = true; else P->split_two_launch = true;
      int rc = resync_direct_counters(P);
      if (rc) return rc;
      return set_error(PGO_ERR_NUMERICAL, "pgo_time_kernel('%s'): an in-kernel wait of the single-launch factorisation timed out; the "
                       "problem now uses one launch per step, time it again", kernel);
    }
  }
  return PGO_OK;
}

// ---- sharding helpers ------------------------------------------------------------------------------
int pgo_shard_range(long long n, int rank, int world, long long* begin, long long* end) {
  if (n < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_shard_range");
  const long long base = n / world, rem = n % world;
  *begin = base * rank + std::min<long long>(rank, rem);
  *end = *begin + base + (rank < rem ? 1 : 0);
  return PGO_OK;
}

// Row ownership of the sharded solve: equal segments (the all-gather exchanges equal-sized pieces) of rows_per poses, rows_per a
// multiple of 4 so that the 2- and 4-pose preconditioner clusters never straddle two ranks; the last ranks may own fewer
// rows or none.  prepare() calls this very function.
int pgo_row_shard_range(long long n_poses, int rank, int world, long long* begin, long long* end, int* rows_per_out) {
  if (n_poses < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_row_shard_range");
  long long rows_per = (n_poses + world - 1) / world;
  rows_per = std::max<long long>(4, (rows_per + 3) / 4 * 4);
  *begin = std::min(n_poses, (long long)rank * rows_per);
  *end = std::min(n_poses, (long long)(rank + 1) * rows_per);
  if (rows_per_out) *rows_per_out = (int)rows_per;
  return PGO_OK;
}

int pgo_comm_get_unique_id(unsigned char id[128]) {
  if (!id) return set_error(PGO_ERR_INVALID_ARGUMENT, "null id");
  const char* what = "";
  if (pgo::rccl_unique_id(id, &what) != 0) return set_error(PGO_ERR_HIP, "ncclGetUniqueId failed: %s", what);
  return PGO_OK;
}

static int attach_comm(pgo_problem* P, pgo::Comm* c) {
  delete P->comm;
  P->comm = c;
  P->topo_dirty = true;   // ownership changes the slot topology
  return PGO_OK;
}

int pgo_comm_init(pgo_problem* P, const unsigned char id[128], int rank, int world) {
  if (!P || !id || world < 1 || rank < 0 || rank >= world) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_comm_init");
  int rc = ensure_device(P);
  if (rc) return rc;
  const char* what = "";
  pgo::Comm* c = pgo::make_rccl_comm(id, rank, world, &what);
  if (!c) return set_error(PGO_ERR_HIP, "ncclCommInitRank failed: %s", what);
  return attach_comm(P, c);
}

// development hook (not part of include/pgo.h): stress the attached transport, returns mismatching words
int pgo_debug_comm_stress(pgo_problem* P, int iters, int seg_doubles) {
  if (!P || !P->comm) return -1;
  if (ensure_device(P)) return -1;
  int bad = -1;
  if (pgo::comm_stress(P->comm, iters, (size_t)seg_doubles, P->stream, &bad) != 0) return -2;
  return bad;
}

void* pgo_loopback_create(int world) { return world >= 1 ? new (std::nothrow) pgo::LoopbackGroup(world) : nullptr; }
void pgo_loopback_destroy(void* group) { delete static_cast<pgo::LoopbackGroup*>(group); }
int pgo_comm_init_loopback(pgo_problem* P, void* group, int rank) {
  pgo::LoopbackGroup* g = static_cast<pgo::LoopbackGroup*>(group);
  if (!P || !g || rank < 0 || rank >= g->world) return set_error(PGO_ERR_INVALID_ARGUMENT, "bad argument to pgo_comm_init_loopback");
  return attach_comm(P, pgo::make_loopback_comm(g, rank));
}

}  // extern "C"
