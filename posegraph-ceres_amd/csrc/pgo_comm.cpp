// pgo_comm.cpp — transports of the sharded path's all-gather (see pgo_comm.h).
#include "pgo_comm.h"

#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace pgo {

bool LoopbackGroup::barrier() {
  std::unique_lock<std::mutex> lk(mu);
  if (aborted) return false;
  const long long gen = generation;
  if (++arrived == world) {
    arrived = 0;
    ++generation;
    cv.notify_all();
  } else {
    cv.wait(lk, [&] { return generation != gen || aborted; });
  }
  return !aborted;
}
void LoopbackGroup::abort() {
  std::lock_guard<std::mutex> lk(mu);
  aborted = true;
  cv.notify_all();
}

namespace {

struct LoopbackComm : Comm {
  LoopbackGroup* g;
  explicit LoopbackComm(LoopbackGroup* group, int r) : g(group) { world = group->world; rank = r; }
  bool capturable() const override { return false; }
  void give_up() override { g->abort(); }
  int fail(const char** what, const char* where, hipError_t e) {
    static thread_local char msg[256];
    snprintf(msg, sizeof msg, "%s: %s", where, hipGetErrorString(e));
    *what = msg;
    g->abort();   // release the peers waiting in a barrier
    return -1;
  }
  // One event pair per rank, created once and re-recorded on every exchange: a stream wait binds to the record that was
  // current when the wait was enqueued, and the third barrier below guarantees every peer has enqueued its waits before
  // the owner records again (exact requests run tens of thousands of exchanges per LM step: no per-call allocation).
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  ~LoopbackComm() override {
    if (ev_ready) (void)hipEventDestroy(ev_ready);
    if (ev_done) (void)hipEventDestroy(ev_done);
  }
  // all virtual ranks live in one process on one GPU: a peer's device pointer is simply valid here
  int peer_table(void* const mine[3], void** out, const char** what) override {
    { std::lock_guard<std::mutex> lk(g->mu);
      if (g->peer_ptrs.size() != (size_t)3 * world) g->peer_ptrs.assign((size_t)3 * world, nullptr);
      for (int k = 0; k < 3; ++k) g->peer_ptrs[3 * rank + k] = mine[k]; }
    if (!g->barrier()) { *what = "a peer rank failed"; return -1; }
    { std::lock_guard<std::mutex> lk(g->mu);
      for (int i = 0; i < 3 * world; ++i) out[i] = g->peer_ptrs[i]; }
    if (!g->barrier()) { *what = "a peer rank failed"; return -1; }     // everybody has read the table before anybody publishes again
    return 0;
  }
  int all_gather(double* buf, size_t seg, hipStream_t s, const char** what) override {
    hipError_t e;
    if (!ev_ready && (e = hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming)) != hipSuccess) return fail(what, "hipEventCreate", e);
    if (!ev_done && (e = hipEventCreateWithFlags(&ev_done, hipEventDisableTiming)) != hipSuccess) return fail(what, "hipEventCreate", e);
    if ((e = hipEventRecord(ev_ready, s)) != hipSuccess) return fail(what, "hipEventRecord", e);
    { std::lock_guard<std::mutex> lk(g->mu); g->bufs[rank] = buf; g->ready[rank] = ev_ready; g->done[rank] = ev_done; }
    if (!g->barrier()) { *what = "a peer rank failed"; return -1; }   // every rank has published buffer + "segment ready" event
    for (int p = 0; p < world; ++p) {
      if (p == rank) continue;
      if ((e = hipStreamWaitEvent(s, g->ready[p], 0)) != hipSuccess) return fail(what, "hipStreamWaitEvent", e);
      if ((e = hipMemcpyAsync(buf + (size_t)p * seg, g->bufs[p] + (size_t)p * seg, seg * sizeof(double), hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return fail(what, "loopback hipMemcpyAsync", e);
    }
    if ((e = hipEventRecord(ev_done, s)) != hipSuccess) return fail(what, "hipEventRecord", e);
    if (!g->barrier()) { *what = "a peer rank failed"; return -1; }   // every rank has enqueued its copies
    for (int p = 0; p < world; ++p)                                   // nobody overwrites a segment a peer is still reading
      if (p != rank && (e = hipStreamWaitEvent(s, g->done[p], 0)) != hipSuccess) return fail(what, "hipStreamWaitEvent", e);
    if (!g->barrier()) { *what = "a peer rank failed"; return -1; }   // every peer has enqueued its waits: the events may be re-recorded
    return 0;
  }
};

struct RcclComm : Comm {
  ncclComm_t comm = nullptr;
  ~RcclComm() override { if (comm) (void)ncclCommDestroy(comm); }
  bool capturable() const override { return true; }
  int all_gather(double* buf, size_t seg, hipStream_t s, const char** what) override {
    const ncclResult_t r = ncclAllGather(buf + (size_t)rank * seg, buf, seg, ncclDouble, comm, s);
    if (r != ncclSuccess) { *what = ncclGetErrorString(r); return -1; }
    return 0;
  }
};

// ---- one process per rank, peer memory through hipIpc handles ------------------------------------------------------------
constexpr int IPC_MAX_WORLD = 16;
constexpr int IPC_MAGIC = 0x50474f35;
struct IpcShared {                      // lives in POSIX shared memory; rank 0 initialises it and sets `magic` last
  std::atomic<int> magic;
  std::atomic<int> arrived;
  std::atomic<long long> generation;
  std::atomic<int> aborted;
  unsigned long long nonce;             // (pid of rank 0, time of creation): tells two groups of the same name apart in a debugger / log
  // proof that rank 0 is alive on THIS block: rank r > 0 writes a random challenge, the live rank 0 echoes it (a block left behind by
  // a crashed run has nobody to answer)
  std::atomic<unsigned long long> chal[IPC_MAX_WORLD], echo[IPC_MAX_WORLD];
  int world;
  unsigned long long stage_bytes;
  hipIpcMemHandle_t stage[IPC_MAX_WORLD];
  hipIpcMemHandle_t peer[IPC_MAX_WORLD][3];
};

struct IpcComm : Comm {
  std::string name;
  IpcShared* sh = nullptr;
  double* stage = nullptr;                        // this rank's staging buffer (device), mapped by every peer
  double* peer_stage[IPC_MAX_WORLD] = {};
  void* opened[IPC_MAX_WORLD][3] = {};
  hipIpcMemHandle_t opened_h[IPC_MAX_WORLD][3] = {};
  static constexpr size_t STAGE_BYTES = (size_t)64 << 20;
  bool capturable() const override { return false; }
  bool peer_direct_default() const override { return true; }
  bool peers_may_be_remote() const override { return true; }
  ~IpcComm() override {
    for (int p = 0; p < world; ++p) {
      if (peer_stage[p] && p != rank) (void)hipIpcCloseMemHandle(peer_stage[p]);
      for (int k = 0; k < 3; ++k) if (opened[p][k]) (void)hipIpcCloseMemHandle(opened[p][k]);
    }
    if (stage) (void)hipFree(stage);
    if (sh && rank == 0) sh->magic.store(0);        // nothing of a finished group can be taken for a live one
    if (sh) munmap(sh, sizeof(IpcShared));
    if (rank == 0 && !name.empty() && my_inode && name_inode(name.c_str()) == my_inode) shm_unlink(name.c_str());
  }
  int fail(const char** what, const char* where, hipError_t e) {
    static thread_local char msg[256];
    snprintf(msg, sizeof msg, "%s: %s", where, e == hipSuccess ? "failed" : hipGetErrorString(e));
    *what = msg;
    if (sh) sh->aborted.store(1);
    return -1;
  }
  // sense-reversing barrier over the shared block; gives up (and aborts the group) after 60 s
  bool barrier() {
    if (sh->aborted.load()) return false;
    const long long gen = sh->generation.load();
    if (sh->arrived.fetch_add(1) + 1 == world) {
      sh->arrived.store(0);
      sh->generation.fetch_add(1);
      return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1; sh->generation.load() == gen; ++spins) {
      if (sh->aborted.load()) return false;
      if ((spins & 0xff) == 0) {
        std::this_thread::yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) { sh->aborted.store(1); return false; }
      }
    }
    return true;
  }
  // inode the name currently points at (0: no such object)
  static unsigned long long name_inode(const char* nm) {
    const int f = shm_open(nm, O_RDWR, 0600);
    if (f < 0) return 0;
    struct stat st;
    const unsigned long long ino = fstat(f, &st) == 0 ? (unsigned long long)st.st_ino : 0;
    close(f);
    return ino;
  }
  // Group identity (r06): a block left behind by a crashed run carries a valid magic, and a rank that starts early can map it before
  // rank 0 has replaced it.  So: rank 0 unlinks whatever holds the name and creates the block O_EXCL (a fresh inode), fills it and sets
  // `magic` last; a rank > 0 maps what the name points at, waits for the magic, writes a random challenge and waits for rank 0 to echo
  // it — only a LIVE rank 0 answers — while it keeps checking that the name still points at the inode it mapped.  If it does not, the
  // block it holds is a stale one: it lets go and attaches again.  The destructor clears the magic (a clean exit leaves nothing to be
  // mistaken for a live group) and rank 0 unlinks the name if it is still its own.
  unsigned long long my_inode = 0;
  int attach(const char** what) {
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0; };
    for (;;) {
      int f;
      while ((f = shm_open(name.c_str(), O_RDWR, 0600)) < 0) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        if (timed_out()) { *what = "the group's shared-memory block did not appear"; return -1; }
      }
      struct stat st;
      while (fstat(f, &st) == 0 && (size_t)st.st_size < sizeof(IpcShared) && !timed_out()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
      if ((size_t)st.st_size < sizeof(IpcShared)) { close(f); *what = "the group's shared-memory block was never sized"; return -1; }
      my_inode = (unsigned long long)st.st_ino;
      sh = static_cast<IpcShared*>(mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, f, 0));
      close(f);
      if (sh == MAP_FAILED) { sh = nullptr; *what = "mmap of the shared block failed"; return -1; }
      bool stale = false;
      while (sh->magic.load() != IPC_MAGIC && !stale) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (name_inode(name.c_str()) != my_inode) stale = true;
        if (timed_out()) { *what = "rank 0 never initialised the shared block"; return -1; }
      }
      if (!stale && name_inode(name.c_str()) == my_inode) return 0;
      munmap(sh, sizeof(IpcShared));        // a block of an earlier group: rank 0 of this one has replaced (or is replacing) it
      sh = nullptr;
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  int init(const char** what) {
    if (rank == 0) {
      (void)shm_unlink(name.c_str());        // whatever an earlier (crashed) group left under this name
      const int f = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (f < 0 || ftruncate(f, sizeof(IpcShared)) != 0) { if (f >= 0) close(f); *what = "shm_open (O_EXCL) / ftruncate failed"; return -1; }
      struct stat st;
      if (fstat(f, &st) == 0) my_inode = (unsigned long long)st.st_ino;
      sh = static_cast<IpcShared*>(mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, f, 0));
      close(f);
      if (sh == MAP_FAILED) { sh = nullptr; *what = "mmap of the shared block failed"; return -1; }
      sh->arrived.store(0); sh->generation.store(0); sh->aborted.store(0);
      for (int r = 0; r < IPC_MAX_WORLD; ++r) { sh->chal[r].store(0); sh->echo[r].store(0); }
      sh->world = world; sh->stage_bytes = STAGE_BYTES;
      sh->nonce = ((unsigned long long)getpid() << 32) ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
      sh->magic.store(IPC_MAGIC);
      // the group has formed once every other rank's challenge has been answered on THIS block
      const auto t0 = std::chrono::steady_clock::now();
      for (int answered = 0; answered < world - 1;) {
        answered = 0;
        for (int r = 1; r < world; ++r) {
          const unsigned long long c = sh->chal[r].load();
          if (c != 0) { sh->echo[r].store(c); ++answered; }
        }
        if (answered < world - 1) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) { *what = "the other ranks never joined the group"; sh->aborted.store(1); return -1; }
      }
    } else {
      for (;;) {
        if (attach(what) != 0) return -1;
        if (sh->world != world) { *what = "the group was created for another world size"; return -1; }
        unsigned long long c = ((unsigned long long)getpid() << 32) ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)rank << 56);
        if (c == 0) c = 1;
        sh->echo[rank].store(0);
        sh->chal[rank].store(c);
        const auto t0 = std::chrono::steady_clock::now();
        bool stale = false;
        while (sh->echo[rank].load() != c && !stale) {
          std::this_thread::sleep_for(std::chrono::microseconds(500));
          if (name_inode(name.c_str()) != my_inode) stale = true;      // challenged a block that has since been replaced: attach again
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) { *what = "rank 0 never answered on the group's shared block"; return -1; }
        }
        if (!stale) break;
        munmap(sh, sizeof(IpcShared));
        sh = nullptr;
      }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&stage), STAGE_BYTES);
    if (e != hipSuccess) return fail(what, "hipMalloc (staging buffer)", e);
    if ((e = hipIpcGetMemHandle(&sh->stage[rank], stage)) != hipSuccess) return fail(what, "hipIpcGetMemHandle (staging buffer)", e);
    if (!barrier()) { *what = "a peer rank failed while the group formed"; return -1; }
    for (int p = 0; p < world; ++p) {
      if (p == rank) { peer_stage[p] = stage; continue; }
      void* q = nullptr;
      if ((e = hipIpcOpenMemHandle(&q, sh->stage[p], hipIpcMemLazyEnablePeerAccess)) != hipSuccess) return fail(what, "hipIpcOpenMemHandle (staging buffer)", e);
      peer_stage[p] = static_cast<double*>(q);
    }
    if (!barrier()) { *what = "a peer rank failed while the group formed"; return -1; }
    return 0;
  }
  int all_gather(double* buf, size_t seg, hipStream_t s, const char** what) override {
    if (seg * sizeof(double) > STAGE_BYTES) { *what = "segment larger than the IPC staging buffer (64 MiB)"; sh->aborted.store(1); return -1; }
    hipError_t e;
    if ((e = hipMemcpyAsync(stage, buf + (size_t)rank * seg, seg * sizeof(double), hipMemcpyDeviceToDevice, s)) != hipSuccess) return fail(what, "ipc stage copy", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(what, "hipStreamSynchronize", e);
    if (!barrier()) { *what = "a peer rank failed"; return -1; }              // every segment is in its owner's staging buffer
    for (int p = 0; p < world; ++p)
      if (p != rank && (e = hipMemcpyAsync(buf + (size_t)p * seg, peer_stage[p], seg * sizeof(double), hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return fail(what, "ipc peer copy", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(what, "hipStreamSynchronize", e);
    if (!barrier()) { *what = "a peer rank failed"; return -1; }              // nobody refills a staging buffer a peer is still reading
    return 0;
  }
  int peer_table(void* const mine[3], void** out, const char** what) override {
    hipError_t e;
    for (int k = 0; k < 3; ++k)
      if ((e = hipIpcGetMemHandle(&sh->peer[rank][k], mine[k])) != hipSuccess) return fail(what, "hipIpcGetMemHandle (exchange buffer)", e);
    if (!barrier()) { *what = "a peer rank failed"; return -1; }
    for (int p = 0; p < world; ++p)
      for (int k = 0; k < 3; ++k) {
        if (p == rank) { out[3 * p + k] = mine[k]; continue; }
        if (opened[p][k] && memcmp(&opened_h[p][k], &sh->peer[p][k], sizeof(hipIpcMemHandle_t)) == 0) { out[3 * p + k] = opened[p][k]; continue; }
        if (opened[p][k]) { (void)hipIpcCloseMemHandle(opened[p][k]); opened[p][k] = nullptr; }
        void* q = nullptr;
        if ((e = hipIpcOpenMemHandle(&q, sh->peer[p][k], hipIpcMemLazyEnablePeerAccess)) != hipSuccess) return fail(what, "hipIpcOpenMemHandle (exchange buffer)", e);
        opened[p][k] = q; opened_h[p][k] = sh->peer[p][k];
        out[3 * p + k] = q;
      }
    if (!barrier()) { *what = "a peer rank failed"; return -1; }              // everybody has mapped the table before anybody publishes again
    return 0;
  }
};

}  // namespace

Comm* make_ipc_comm(const char* name, int rank, int world, const char** what) {
  if (world < 1 || world > IPC_MAX_WORLD) { *what = "world size out of range (1..16)"; return nullptr; }
  IpcComm* c = new IpcComm();
  c->name = name; c->rank = rank; c->world = world;
  if (c->init(what) != 0) { delete c; return nullptr; }
  return c;
}

Comm* make_loopback_comm(LoopbackGroup* group, int rank) { return new LoopbackComm(group, rank); }

int rccl_unique_id(unsigned char id[128], const char** what) {
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the ABI slot");
  ncclUniqueId u;
  const ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) { *what = ncclGetErrorString(r); return -1; }
  std::memset(id, 0, 128);
  std::memcpy(id, &u, sizeof u);
  return 0;
}

Comm* make_rccl_comm(const unsigned char id[128], int rank, int world, const char** what) {
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  RcclComm* c = new RcclComm();
  c->world = world;
  c->rank = rank;
  const ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) { *what = ncclGetErrorString(r); c->comm = nullptr; delete c; return nullptr; }
  return c;
}

}  // namespace pgo
