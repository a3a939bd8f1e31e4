// Small low-latency worker pool for the per-cycle host prep (the libm trig
// table).  Workers spin briefly on a generation counter before sleeping, so a
// controller running at a steady rate wakes them in well under a microsecond;
// an idle controller costs nothing.
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace kc {

class WorkerPool {
 public:
  static WorkerPool &instance() {
    static WorkerPool pool;
    return pool;
  }

  int workers() const { return static_cast<int>(threads_.size()); }

  // fn(begin, end) over [0, n) split into contiguous chunks; the caller takes
  // part.  Serial when n is small or the pool has no workers.
  template <typename F>
  void parallel_for(size_t n, size_t min_chunk, F &&fn) {
    const size_t parts =
        std::min<size_t>(threads_.size() + 1, min_chunk ? n / min_chunk : n);
    if (parts <= 1) {
      fn(size_t(0), n);
      return;
    }
    std::lock_guard<std::mutex> serial(run_mu_);  // one job at a time
    auto job = std::make_shared<Job>();
    job->parts = parts;
    job->fn = [&fn, n, parts](size_t part) {
      const size_t b = n * part / parts, e = n * (part + 1) / parts;
      if (b < e) fn(b, e);
    };
    job->next.store(1, std::memory_order_relaxed);  // part 0 is the caller's
    job->pending.store(parts - 1, std::memory_order_relaxed);
    std::atomic_store(&cur_, job);
    {
      std::lock_guard<std::mutex> lk(mu_);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    job->fn(0);
    run_parts(*job);  // help with whatever is left, then wait for stragglers
    while (job->pending.load(std::memory_order_acquire) != 0) cpu_relax();
  }

 private:
  WorkerPool() {
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw >= 32 ? 7 : hw >= 8 ? 3 : hw >= 4 ? 1 : 0;
    for (int i = 0; i < n; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  static void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  void loop() {
    uint64_t seen = gen_.load(std::memory_order_acquire);
    for (;;) {
      // spin for a while (~50-100 us), then sleep
      int spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        if (++spins < 40000) {
          cpu_relax();
        } else {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] {
            return gen_.load(std::memory_order_acquire) != seen || stop_;
          });
        }
        if (stop_) return;
      }
      if (stop_) return;
      seen = gen_.load(std::memory_order_acquire);
      // a late worker may pick up an already finished job: its part counter
      // is exhausted, so it never calls into a dead caller frame
      std::shared_ptr<Job> job = std::atomic_load(&cur_);
      if (job) run_parts(*job);
    }
  }

  struct Job {
    std::function<void(size_t)> fn;
    size_t parts = 0;
    std::atomic<size_t> next{0}, pending{0};
  };
  static void run_parts(Job &j) {
    for (;;) {
      const size_t p = j.next.fetch_add(1, std::memory_order_acq_rel);
      if (p >= j.parts) break;
      j.fn(p);
      j.pending.fetch_sub(1, std::memory_order_acq_rel);
    }
  }

  std::vector<std::thread> threads_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> gen_{0};
  std::shared_ptr<Job> cur_;
  std::atomic<bool> stop_{false};
};

}  // namespace kc
