// pgo_pool.h — process-wide pool of host worker threads for the topology build and the symbolic analysis (nothing on the LM
// path).  Starting a std::thread costs 20-50 us; the analysis of a batched solve (16-64 graphs as the components of one
// problem) has a dozen short parallel phases of 0.3-2 ms each, and with sixteen fresh threads per phase the starts were a third
// of its 8 ms.  Workers are created once, sleep on a condition variable between phases and are never joined (they make no HIP
// calls; the pool object is leaked on purpose so that nothing runs at process exit).
//
//   HostPool::get().run(slots, fn)   calls fn(0) .. fn(slots - 1), the caller taking part, and returns when all have returned.
//                                    Slots are handed out dynamically: they need not run at the same time (fewer workers than
//                                    slots, workers busy with another caller's phase), so fn must not wait for a sibling slot
//                                    that has not started.  Several host threads may call run() at once.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pgo {

class HostPool {
 public:
  static HostPool& get() {
    static HostPool* pool = new HostPool();
    return *pool;
  }
  // threads that can work on one phase (the caller included)
  int width() {
    start();
    return (int)workers_.size() + 1;
  }
  template <class F>
  void run(int slots, F&& fn) {
    if (slots <= 0) return;
    if (slots > 1) start();
    if (slots == 1 || workers_.empty()) {
      for (int i = 0; i < slots; ++i) fn(i);
      return;
    }
    const std::function<void(int)> call(std::ref(fn));
    Job job;
    job.fn = &call;
    job.slots = slots;
    {
      std::lock_guard<std::mutex> lk(mu_);
      jobs_.push_back(&job);
    }
    wake_.notify_all();
    work_on(job);
    std::unique_lock<std::mutex> lk(mu_);
    idle_.wait(lk, [&] { return job.done.load() == slots && job.active == 0; });
    jobs_.erase(std::remove(jobs_.begin(), jobs_.end(), &job), jobs_.end());
  }

 private:
  struct Job {
    const std::function<void(int)>* fn = nullptr;
    int slots = 0;
    std::atomic<int> next{0}, done{0};
    int active = 0;               // workers inside work_on (guarded by mu_)
  };
  HostPool() {}
  void start() {
    std::call_once(once_, [this] {
      const char* env = getenv("PGO_HOST_THREADS");
      const int hw = (int)std::thread::hardware_concurrency();
      const int want = env ? atoi(env) : std::min(hw, 32);
      for (int i = 1; i < want; ++i) {
        workers_.emplace_back([this] { loop(); });
        workers_.back().detach();
      }
    });
  }
  static void work_on(Job& job) {
    for (;;) {
      const int i = job.next.fetch_add(1);
      if (i >= job.slots) return;
      (*job.fn)(i);
      job.done.fetch_add(1);
    }
  }
  Job* open_job() {             // (mu_ held)
    for (Job* j : jobs_) if (j->next.load() < j->slots) return j;
    return nullptr;
  }
  void loop() {
    for (;;) {
      Job* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        // (the slot counter moves without the lock: the job seen by the predicate is the one taken, open or not by now)
        wake_.wait(lk, [&] { return (job = open_job()) != nullptr; });
        ++job->active;
      }
      work_on(*job);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --job->active;
      }
      idle_.notify_all();
    }
  }
  std::once_flag once_;
  std::mutex mu_;
  std::condition_variable wake_, idle_;
  std::vector<Job*> jobs_;
  std::vector<std::thread> workers_;
};

}  // namespace pgo
