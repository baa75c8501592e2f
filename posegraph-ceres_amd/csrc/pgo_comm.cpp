// pgo_comm.cpp — transports of the sharded path's all-gather (see pgo_comm.h).
#include "pgo_comm.h"

#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

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

}  // namespace

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
