// CPU stress test of kc::WorkerPool (csrc/kc_pool.h): several caller threads
// hammer parallel_for() on the process-wide pool at once (two controller
// contexts on two threads do exactly this in the host-trig fallback), a third
// resizes the pool now and then.  Every job must cover its whole range exactly
// once.  Exit code 0 = ok.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "kc_pool.h"

int main(int argc, char **argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
  const int callers = argc > 2 ? std::atoi(argv[2]) : 3;
  kc::WorkerPool &pool = kc::WorkerPool::instance();
  pool.resize(4);
  std::atomic<int> bad{0};
  std::atomic<bool> stop{false};
  auto caller = [&](int id) {
    const size_t n = 97 + 13 * id;
    std::vector<int> hits(n);
    for (int it = 0; it < iters; ++it) {
      std::fill(hits.begin(), hits.end(), 0);
      auto fn = [&hits](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) hits[i]++;
      };
      pool.parallel_for(n, (it + id) % 3 == 0 ? 2 : 5, fn);
      for (size_t i = 0; i < n; ++i)
        if (hits[i] != 1) {
          bad++;
          break;
        }
    }
  };
  std::vector<std::thread> th;
  for (int c = 0; c < callers; ++c) th.emplace_back(caller, c);
  std::thread resizer([&] {
    int k = 0;
    while (!stop.load()) {
      pool.resize(1 + (k++ % 6));  // 0..5 workers
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  });
  for (auto &t : th) t.join();
  stop.store(true);
  resizer.join();
  std::printf("pool_stress: %d callers x %d jobs, %d bad\n", callers, iters, bad.load());
  return bad.load() ? 1 : 0;
}
