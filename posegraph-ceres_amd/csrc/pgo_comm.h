// pgo_comm.h — the one collective of the sharded path: an in-place all-gather of equal segments of doubles
// (rank r contributes buf[r*seg .. (r+1)*seg)).  Two transports:
//   * RcclComm     : ncclAllGather over RCCL/xGMI, one process per GPU (production)
//   * LoopbackComm : several "virtual ranks" = pgo_problem instances driven by host threads of ONE process on one
//                    GPU; segments are exchanged with device-to-device copies ordered by events.  It exists so that
//                    the sharding logic can be tested on a single-GPU box; it is not capturable into a hipGraph.
//   * IpcComm (r05): one PROCESS per rank, the ranks' exchange buffers mapped into each other through hipIpc memory handles —
//                    what lets the kernels of the owner-only CG store into every rank's buffer and flag array themselves
//                    (peer_table, DeviceGraph::peer_tab) across process boundaries: several GPUs of one node, or (tests) several
//                    processes on ONE GPU.  Control plane: a POSIX shared-memory block (handles, a barrier).  Its all-gather —
//                    used by the few exchanges per LM iteration outside the CG — goes through an IPC-mapped staging buffer
//                    with two host barriers: correct, not fast, not capturable.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <condition_variable>
#include <mutex>
#include <vector>

namespace pgo {

struct Comm {
  int world = 1, rank = 0;
  virtual ~Comm() {}
  // returns hipSuccess-like 0 or a negative value; `what` receives a static description on failure
  virtual int all_gather(double* buf, size_t seg_doubles, hipStream_t s, const char** what) = 0;
  virtual bool capturable() const = 0;
  // Device-initiated exchange (pgo_kernels.h DeviceGraph::peer_tab): every rank publishes three device pointers (its two exchange
  // buffers and its flag array) and receives everybody's, `out[3 * rank + k]`; a collective call.  Returns 0, or -1 when the
  // transport cannot give kernels access to peer memory (then the all-gather above stays in use).
  virtual int peer_table(void* const mine[3], void** out, const char** what) { (void)mine; (void)out; *what = "not supported by this transport"; return -1; }
  // whether the device-initiated exchange is this transport's normal mode (PGO_PEER_DIRECT=0 / 1 overrides either way)
  virtual bool peer_direct_default() const { return false; }
  // whether the peers of peer_table() may be OTHER devices (processes, one GPU each): their exchange buffers and flag words must then be
  // fine-grained allocations (pgo_internal.h DevBuf::alloc_fine); virtual ranks share one device and its L2
  virtual bool peers_may_be_remote() const { return false; }
  // this rank gives up (its solve failed outside a collective): peers waiting for it in one must not wait for ever.  The in-process
  // transport releases its barriers; RCCL / IPC ranks are processes — their launcher's watchdog is what ends them (bench.py).
  virtual void give_up() {}
};

struct LoopbackGroup {
  explicit LoopbackGroup(int n) : world(n), bufs(n, nullptr), ready(n, nullptr), done(n, nullptr) {}
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long long generation = 0;
  std::vector<double*> bufs;
  std::vector<void*> peer_ptrs;      // 3 per rank (peer_table)
  std::vector<hipEvent_t> ready, done, garbage;
  ~LoopbackGroup() { for (hipEvent_t e : garbage) (void)hipEventDestroy(e); }
  bool aborted = false;
  bool barrier();   // false once any rank has failed
  void abort();
};

Comm* make_loopback_comm(LoopbackGroup* group, int rank);
// RCCL: id is the 128-byte ncclUniqueId created by rccl_unique_id() on rank 0
// IPC: `name` = POSIX shared-memory object every rank of the group opens (rank 0 creates it), e.g. "/pgo_ipc_<pid>"
Comm* make_ipc_comm(const char* name, int rank, int world, const char** what);
int rccl_unique_id(unsigned char id[128], const char** what);
Comm* make_rccl_comm(const unsigned char id[128], int rank, int world, const char** what);

}  // namespace pgo
