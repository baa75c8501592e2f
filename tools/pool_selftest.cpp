// tools/pool_selftest.cpp — host-only stress of the worker pool (posegraph-ceres_amd/csrc/pgo_pool.h): several caller threads
// run phases of different widths at once, phases whose slots hand work to each other (the shape of the nested-dissection task
// loop), and a fresh caller thread per phase (the shape of the analysis thread).  Prints "ok <checksum>"; built with
// -fsanitize=thread by tests/test_host_pool.py when the compiler has it.
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "../posegraph-ceres_amd/csrc/pgo_pool.h"

int main() {
  std::atomic<long long> total(0);
  // (1) concurrent callers, plain slots
  auto caller = [&](int reps, int slots) {
    for (int rep = 0; rep < reps; ++rep) {
      std::atomic<int> sum(0);
      pgo::HostPool::get().run(slots, [&](int i) { for (int k = 0; k < 200; ++k) sum += (i + k) & 1; });
      total += sum;
    }
  };
  {
    std::thread a(caller, 3000, 16), b(caller, 3000, 5), c(caller, 3000, 2), d(caller, 3000, 1);
    a.join(); b.join(); c.join(); d.join();
  }
  // (2) a task loop inside the slots: a slot waits only while another one is at work
  for (int rep = 0; rep < 3000; ++rep) {
    std::thread t([&] {
      std::atomic<int> sum(0);
      std::mutex mu;
      std::condition_variable cv;
      int in_flight = 0;
      std::vector<int> tasks = {37, 11};
      pgo::HostPool::get().run(3, [&](int) {
        for (;;) {
          int r;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !tasks.empty() || in_flight == 0; });
            if (tasks.empty()) return;
            r = tasks.back();
            tasks.pop_back();
            ++in_flight;
          }
          sum += r;
          {
            std::lock_guard<std::mutex> lk(mu);
            if (r > 1) { tasks.push_back(r / 2); tasks.push_back(r - r / 2); }
            --in_flight;
          }
          cv.notify_all();
        }
      });
      total += sum;
    });
    t.join();
  }
  std::printf("ok %lld\n", (long long)total);
  return 0;
}
