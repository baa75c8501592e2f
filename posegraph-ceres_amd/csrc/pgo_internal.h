// pgo_internal.h — what the host-side translation units of libpgo_hip.so share (r03: pgo_solver.cpp, one 3 100-line file until r02,
// is now pgo_problem.cpp / pgo_linear.cpp / pgo_lm.cpp / pgo_batch.cpp / pgo_report.cpp / pgo_capi.cpp): the error channel,
// the staging / pooling helpers, DevBuf, the LM driver's state, struct pgo_problem and the prototypes of the functions that
// cross files.  Internal: nothing here is part of the C ABI (include/pgo.h); the library is built with -fvisibility=hidden, so none
// of these names leave it.
#pragma once
#include "pgo_tuning.h"
#include "../../include/pgo.h"
#include "pgo_kernels.h"
#include "pgo_sym.h"
#include "pgo_lm_rules.h"
#include "pgo_direct.h"
#include "pgo_front.h"
#include "pgo_comm.h"
#include "pgo_pool.h"
#include "pgo_coarse.h"

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
#include <deque>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>

int set_error(int code, const char* fmt, ...);

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
// At most 16 GB are cached; pgo_release_device_memory() empties the pool.
struct DevicePool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> blocks;   // (device, capacity in bytes) -> block
  size_t cached = 0;
  static size_t limit() {
    return (size_t)16e9;
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
  static bool off() { static const bool v = false; return v; }
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
  bool fine = false;     // fine-grained device memory (alloc_fine): what kernels of OTHER devices store into and poll; never pooled
  ~DevBuf() { release(true); }
  void release(bool to_pool = false) {
    if (!p) return;
    if (to_pool && !fine) device_pool().put(p, cap_bytes, dev); else (void)hipFree(p);
    p = nullptr; n = 0; cap_bytes = 0; fine = false;
  }
  // Exchange buffers and flag words that peer devices write through IPC / peer mappings (DeviceGraph::peer_tab): ordinary hipMalloc
  // memory is coarse-grained — a remote store is only guaranteed visible at kernel boundaries — so these are allocated fine-grained
  // (coherent across agents while kernels run).  Falls back to the ordinary allocation where the runtime refuses the flag.
  hipError_t alloc_fine(size_t count) {
    release(false);
    n = count;
    if (count == 0) return hipSuccess;
    void* q = nullptr;
    hipError_t e = hipExtMallocWithFlags(&q, count * sizeof(T), hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; return alloc(count); }
    (void)hipGetDevice(&dev);
    p = static_cast<T*>(q); cap_bytes = count * sizeof(T); fine = true;
    return hipSuccess;
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
  double x_cost = 0, x_norm = 0, gmax = 0, initial_cost = 0, initial_x_norm = 0;
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


struct pgo_problem {
  // ---- host-side problem (ceres::Problem bookkeeping) ----
  std::vector<double*> pp, qq;
  std::unordered_map<const double*, int> block_of_ptr;  // pose*2 + (0: p block, 1: q block)
  std::vector<uint8_t> cmask;
  // pose / landmark problems: a 3-D point is a node whose translation block is the caller's 3 doubles and whose
  // rotation block is a constant identity quaternion owned by the problem; an observation is a between-factor whose information
  // has no rotation part.  is_point steers the elimination order (points first = the Schur complement onto the poses, pgo_direct.cpp)
  std::vector<uint8_t> is_point;
  std::deque<std::array<double, 4>> point_q;
  std::vector<int> ia, ib;
  std::vector<double> meas;       // 7 per edge
  std::vector<double> sqrt_info;  // 36 per edge once any edge carries a non-identity matrix
  bool has_info = false;
  int loss_kind = PGO_LOSS_TRIVIAL;
  double loss_a = 1.0;
  bool topo_dirty = true;
  // Several ranks (r06): the rows are cut where the INCIDENCE SLOTS balance, not the row counts, and the exchanges still want equal
  // segments — so prepare() renumbers the poses for the device: rank r's share [cut[r], cut[r + 1]) of the caller's poses becomes the
  // internal poses [r * rows_per, r * rows_per + cut[r + 1] - cut[r]), the rest of the segment is padding (constant poses without
  // edges).  pose_int[v] = internal index of the caller's pose v (empty: one rank, the identity); ia_int / ib_int / cmask_int /
  // is_point_int are the topology in internal numbering; everything on the device, and every host array derived from the slot
  // topology, is internal.  Only the C-ABI entry points that hand per-pose arrays over translate (pose_rows_to_host / _to_device).
  std::vector<int> pose_int, ia_int, ib_int, shard_cut;
  std::vector<uint8_t> cmask_int, is_point_int;
  int n_int = 0;                                      // internal pose count (world * rows_per when renumbered)
  bool renumbered() const { return !pose_int.empty(); }
  const std::vector<int>& t_ia() const { return renumbered() ? ia_int : ia; }
  const std::vector<int>& t_ib() const { return renumbered() ? ib_int : ib; }
  const std::vector<uint8_t>& t_cmask() const { return renumbered() ? cmask_int : cmask; }
  const std::vector<uint8_t>& t_is_point() const { return renumbered() ? is_point_int : is_point; }

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
      d_part_rr, d_part_bb, d_part_misc, d_tmp_a, d_tmp_b, d_tmp_c, d_cg_u, d_cg_w, d_cg_s, d_cg_qq, d_pipe_a, d_pipe_b, d_pipe_x, d_part_f;
  DevBuf<pgo::CgState> d_cg;
  DevBuf<long long> d_oplog;        // PGO_UNI_OPLOG=<file>: per-launch operation log of k_uni_s (DeviceGraph::oplog), appended to the file at pgo_solver_end
  // device-initiated exchange of the owner-only CG (DeviceGraph::peer_tab): the table of every rank's buffers, this rank's flags, the
  // global number of the last producing launch (the same on every rank: the launch sequences are replicated)
  DevBuf<void*> d_peer_tab;
  DevBuf<unsigned long long> d_peer_flags;
  unsigned long long peer_gseq = 0;
  bool peer_dirty = false;
  bool force_standard_cg = false;   // pgo_linear_solve (one linear system, every rank reads the whole x): the replicated standard CG even on several ranks
  // spare set of the linearisation (blocks, diagonal blocks, gradient): the candidate point is linearised into it right behind
  // the step tail, before the host has decided; an accepted step swaps the sets (one rank, eager enqueue)
  DevBuf<double> d_bsr2, d_Hdiag2, d_grad2;
  bool spec_ready = false;
  int lin_diag_only = -1;           // several ranks: what the last linearisation exchanged of the other ranks' diagonal blocks (1: only their diagonals — valid for the owner-only CG alone); checked against the CG form at every CG start
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
  int uni_enq = 0;                 // vector-shaped launches (fused form: launches) of the stream enqueued since the last upload
  int resident_aborts = 0;         // times a resident session gave up at a grid barrier and went on with the fused stream (0 or 1 per session)
  bool uni_resident = false;       // ... in its resident form (pgo_uni_resident.h: four kernels in a fixed cycle, the whole CG of an LM iteration one launch with a
                                   //     grid barrier per iteration); uni_fused is set as well (same state words, same launch trace)
  bool resident_slot = false;      // this problem holds its device's one resident-session slot (pgo_lm.cpp)
  bool uni_fused = false;          // ... in its fused form (k_uni_f: one kernel symbol, one launch per CG iteration, pipelined recurrences)
  // what the host spent enqueueing the stream (pgo_solver_trace): launches and seconds inside the launch calls, since pgo_solver_begin
  long long uni_host_launches = 0;
  double uni_host_enqueue_s = 0.0;
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
  bool sfront_levels = false;      // small-front plan: one launch per level (a wait of the single-launch form ran out, or the knob factor_fused = 0)
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
  bool front_launches = false;     // multifrontal plan: one launch per phase of a round (a wait of the single-launch form ran out, or the knob factor_fused = 0)
  unsigned long long front_epoch = 0, front_tickets = 0;
  DevBuf<int> df_perm, df_idx, df_child, df_rel, df_cstart, df_col_front, df_ablk_ptr, df_ablk_slot, df_ablk_front, df_ablk_pos, df_wg_job, df_wg_tile,
      df_bwd_front, df_bwd_chunk, df_bwdb_front, df_bwdb_chunk, df_asm_tile, df_asm_contrib;
  DevBuf<pgo::FrontDesc> df_fronts;
  DevBuf<pgo::FrontJob> df_jobs;
  DevBuf<double> df_Fval, df_Winv, df_x;
  // symmetric tile form of the normal equations (pgo_sym.h): every interior off-diagonal block stored and read once by the CG
  // products of large graphs on one rank (built at the start of such a solve; the incidence-slot arrays stay the system of record)
  pgo::SymGraph sym{};
  bool sym_built = false, sym_ready = false, sym_active = false;
  double sym_interior_fraction = 0.0;
  long long sym_stored_slots = 0;
  DevBuf<pgo::SymTile> sy_tile;
  DevBuf<int> sy_xlist, sy_chunk_base, sy_chunk_n, sy_src, sy_diag, sy_xoff, sy_xbidx;
  // boundary exchange of the sharded owner-only CG (pgo_kernels.h DeviceGraph::bx): lists and buffers of this topology (prepare())
  bool bx_ready = false;
  int bx_cseg = 0, bx_nb = 0;
  std::vector<int> h_bpos;             // [n_int] index of every boundary row inside its rank's segment (-1: not a boundary row)
  DevBuf<int> d_bx_brow, d_bx_slot_off;
  DevBuf<double> d_bx0, d_bx1;
  bool sym_stale = true;            // the off-diagonal blocks were rewritten since the last repack (else only the damped diagonal slots are copied)
  DevBuf<uint32_t> sy_meta, sy_rinfo;
  bool sym_storage = false;         // this LM session keeps the normal equations in the symmetric tile form ONLY: linearisation, damping, the cluster
                                    // preconditioner and every product work on it (the incidence-slot blocks are not maintained)
  std::vector<int> h_cl_slot, h_sym_of_old;
  DevBuf<int> sy_cl_slot, sy_dst;
  DevBuf<double> sy_val;
  // cluster-Jacobi preconditioner topology (built when the option asks for clusters of 2 or 4 poses)
  DevBuf<int> d_cl_ptr, d_cl_slot;
  DevBuf<uint8_t> d_cl_rc;
  int cluster_built = 0;

  // coarse level of the PCG (pgo_coarse.h; options.pcg_coarse_aggregate >= 8, one rank): plan + buffers of this session
  pgo::CoarsePlan coarse{};
  bool coarse_on = false;
  int force_block = 0;             // slots per work-group the topology has to be built with (0: choose_block's rule); a session with a coarse level asks for 256: whole pose pairs per work-group
  DevBuf<double> dc_Pt, dc_Ac, dc_Ac2, dc_rc;
  DevBuf<int> dc_rank_end;

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


// ---- pgo_problem.cpp: device set-up, topology build, pose transfers, the exchange primitive ----
const char* last_error_string();
int ensure_device(pgo_problem* P);
int exchange(pgo_problem* P, double* buf, size_t seg_doubles);
int linearize_all(pgo_problem* P, bool diag_only = false);
int damping_all(pgo_problem* P, double radius, double min_diag, double max_diag, int mode);
int cg_iteration(pgo_problem* P, const pgo::DeviceGraph& g, const pgo::CgParams& prm, int odd, bool refresh);
int cg_iteration(pgo_problem* P, const pgo::CgParams& prm, int odd, bool refresh);
int prepare(pgo_problem* P);
int peer_direct_setup(pgo_problem* P);
bool sym_wanted(const pgo_problem* P);     // pgo_sym.cpp
pgo::DeviceGraph sym_view(const pgo_problem* P);   // P->g with the block storage, the diagonal slots and the cluster lists of the symmetric form
int sym_enter_storage(pgo_problem* P);
int sym_prepare(pgo_problem* P);
int upload_poses(pgo_problem* P, double* dst);
int download_poses(pgo_problem* P, const double* src);
// per-pose arrays of `width` doubles per pose between the caller's numbering (host) and the device's (pgo_problem::pose_int)
int pose_rows_to_host(pgo_problem* P, double* host, const double* dev, int width);
int pose_rows_to_device(pgo_problem* P, double* dev, const double* host, int width, double pad_value = 0.0);   // pad_value: what the padding poses of a renumbered problem get
int fill_scale_one(pgo_problem* P);

// ---- pgo_linear.cpp: the linear solvers behind an LM iteration — CG batches and their hand-off, preconditioner clusters, the
// three GPU factorisations (analysis on the host, uploads, launch sequences) ----
inline void arm_handoff(pgo_problem* P) { __atomic_store_n(&P->scal->seq, 0, __ATOMIC_SEQ_CST); }
int wait_handoff(pgo_problem* P);
int enqueue_tail(pgo_problem* P, const pgo::CgParams* finish_prm);
int launch_cg_batch(pgo_problem* P, const pgo::CgParams& prm, int batch, bool with_tail = false, int start_it = 1);
int pick_batch(const pgo::CgParams& prm, int user_batch, int round, int enqueued, int last_iterations);
bool pipe_mode(const pgo_problem* P, const pgo::CgParams& prm);
int pcg_begin(pgo_problem* P, const pgo::CgParams& prm);
int run_pcg(pgo_problem* P, const pgo::CgParams& prm, int batch, int* iterations, int* status);
int prepare_clusters(pgo_problem* P, int CL);
int coarse_setup(pgo_problem* P);                 // coarse level of the PCG (pgo_coarse.h): per LM iteration, behind the damping
int coarse_apply(pgo_problem* P, const double* vec, double* out, double* out2, int fold_seq);
long long front_memory_budget();
void analyze_front(pgo_problem* P, int N, int n_slots, long long budget, bool* ok, int small_max = 0);
int upload_front(pgo_problem* P);
int upload_sfront(pgo_problem* P);
int decide_direct_host(pgo_problem* P, int N, int E, int n_slots, long long front_budget);
int prepare_direct(pgo_problem* P);
void enqueue_front_factor(pgo_problem* P, const pgo::DeviceGraph& G);
int run_direct(pgo_problem* P, const pgo::DeviceGraph& G);
int run_direct(pgo_problem* P);
pgo::CgParams cg_params_for(const pgo_solver_options& o);
int resync_direct_counters(pgo_problem* P);

// ---- pgo_lm.cpp: the Levenberg-Marquardt drivers (host in the loop / sequences enqueued ahead / universal stream) ----
struct StepScalars {        // what the device hands back after a trial step
  double cand_cost, model_change, step_norm_sq, x_norm_sq, gradient_max;
  int cg_iterations, cg_status, linearize_bad;
};
enum StepAction { STEP_NONE = 0, STEP_ACCEPT = 1, STEP_REJECT = 2 };   // NONE: terminated or invalid step
int evaluate_gradient_and_jacobian(pgo_problem* P, bool first);
int lm_begin(pgo_problem* P, const pgo_solver_options* options);
void terminate(LmState& L, int termination, int reason, const char* fmt, ...);
void terminate_by_reason(LmState& L, const pgo_solver_options& o, int termination, int reason, double value);
bool lm_pre_step(LmState& L, const pgo_solver_options& o);
StepAction lm_post_step(LmState& L, const pgo_solver_options& o, const StepScalars& sc, int extra_linear_iterations);
pgo::LmTolerances lm_tolerances(const pgo_solver_options& o);
int lm_advance(pgo_problem* P);
int lm_upload_state(pgo_problem* P);
int lm_run_pipelined(pgo_problem* P, int budget, int* ran);
int lm_run_universal(pgo_problem* P, int budget, int* ran);
void resident_slot_release(pgo_problem* P);     // the device's one resident-CG session slot (pgo_lm.cpp)
int lm_end(pgo_problem* P, pgo_solver_summary* summary, pgo_iteration_record* records, int capacity);

// ---- pgo_batch.cpp ----
int solve_batch(pgo_problem* const* probs, int n, const pgo_solver_options* options, pgo_solver_summary* summaries,
                pgo_iteration_record* records, int capacity);

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

