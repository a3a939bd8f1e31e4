// Small low-latency worker pool for the per-cycle host prep (the libm trig
// table).  Static partition: worker w always takes part w+1 of the range and
// owns a cache line for its "done" word, so a job costs no contended atomic
// (on a two-socket host a shared work counter bounces between sockets and
// costs more than the work itself).  Workers spin briefly on a generation
// word before sleeping, so a controller running at a steady rate finds them
// hot; an idle controller costs nothing.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
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

  int workers() {
    std::lock_guard<std::mutex> serial(run_mu_);
    return static_cast<int>(threads_.size());
  }

  // fn(begin, end) over [0, n) split into one contiguous chunk per thread; the
  // caller takes the first chunk.  Serial when n is small or no workers exist.
  // Asynchronous form: the workers take the whole range while the caller does
  // something else (e.g. a kernel launch); wait() returns when they are done.
  // Without workers the range is processed on the spot.  fn is copied.
  // A ticket ties wait() to the begin() of the SAME caller: the pool is a
  // process-wide singleton and several contexts may be driven from several
  // threads, so the asynchronous state lives with the caller, not in the pool.
  // begin() returns with run_mu_ held (one job at a time); wait(ticket) of the
  // same thread joins the job and releases it.  A spent or empty ticket makes
  // wait() a no-op, so a scope guard may call it again on any exit path.
  struct Ticket {
    uint64_t gen = 0;    // 0: nothing to wait for
    bool self = false;   // the caller carries part 0 inside wait()
  };
  template <typename F>
  Ticket begin(size_t n, size_t min_chunk, F fn) {
    Ticket t;
    if (n == 0) return t;
    run_mu_.lock();  // released by wait(t) on this thread (the worker set is stable while held)
    if (threads_.empty()) {
      run_mu_.unlock();
      fn(size_t(0), n);
      return t;
    }
    // with a small pool the caller takes a share too (run inside wait(), after
    // whatever it does in between); with a large one its share would only delay
    // the result
    const size_t self = threads_.size() <= 3 ? 1 : 0;
    const size_t parts = std::min<size_t>(threads_.size() + self,
                                          min_chunk ? std::max<size_t>(n / min_chunk, 1) : n);
    job_fn_ = [fn, n, parts, self](size_t part) {
      // worker w carries part w + 1; without a caller share the parts shift down
      const size_t q = part - (self ? 0 : 1);
      const size_t b = n * q / parts, e = n * (q + 1) / parts;
      if (b < e) fn(b, e);
    };
    job_parts_ = parts + (self ? 0 : 1);
    const uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      gen_.store(g, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    t.gen = g;
    t.self = self != 0;
    return t;
  }
  // The same with the decomposition handed to the job: fn(q, parts) for q = 0 .. parts - 1 (a job that wants
  // to publish partial results in a global order deals its work over the parts by itself).
  template <typename F>
  Ticket begin_parts(size_t n, size_t min_chunk, F fn) {
    Ticket t;
    if (n == 0) return t;
    run_mu_.lock();
    if (threads_.empty()) {
      run_mu_.unlock();
      fn(size_t(0), size_t(1));
      return t;
    }
    const size_t self = threads_.size() <= 3 ? 1 : 0;
    const size_t parts = std::min<size_t>(threads_.size() + self,
                                          min_chunk ? std::max<size_t>(n / min_chunk, 1) : n);
    job_fn_ = [fn, parts, self](size_t part) {
      const size_t q = part - (self ? 0 : 1);
      if (q < parts) fn(q, parts);
    };
    job_parts_ = parts + (self ? 0 : 1);
    const uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      gen_.store(g, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    t.gen = g;
    t.self = self != 0;
    return t;
  }
  void wait(Ticket &t) {
    if (!t.gen) return;
    if (t.self) job_fn_(0);
    for (size_t w = 0; w < threads_.size(); ++w)
      while (slots_[w].done.load(std::memory_order_acquire) != t.gen) cpu_relax();
    t.gen = 0;
    run_mu_.unlock();
  }

  // Worker count of the process-wide pool (kc_set_host_threads): joins the
  // current workers and starts `total - 1` new ones (the caller is the first
  // thread of a job).  Waits for a running job.
  void resize(int total) {
    std::lock_guard<std::mutex> serial(run_mu_);
    stop_workers();
    start_workers(std::max(0, std::min(total, 64) - 1));
  }

  template <typename F>
  void parallel_for(size_t n, size_t min_chunk, F &&fn) {
    std::unique_lock<std::mutex> serial(run_mu_);  // one job at a time; the worker set is stable while held
    const size_t parts =
        std::min<size_t>(threads_.size() + 1, min_chunk ? n / min_chunk : n);
    if (parts <= 1) {
      serial.unlock();
      fn(size_t(0), n);
      return;
    }
    job_fn_ = [&fn, n, parts](size_t part) {
      const size_t b = n * part / parts, e = n * (part + 1) / parts;
      if (b < e) fn(b, e);
    };
    job_parts_ = parts;
    const uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      gen_.store(g, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    job_fn_(0);
    // every worker acknowledges the generation (those without a part at once)
    for (size_t w = 0; w < threads_.size(); ++w)
      while (slots_[w].done.load(std::memory_order_acquire) != g) cpu_relax();
  }

 private:
  struct alignas(64) Slot {
    std::atomic<uint64_t> done{0};
  };

  // CPUs this process may actually use: hardware threads, capped by the cgroup
  // CPU quota, shared between the ranks of a node (one process per GPU)
  static unsigned usable_cpus() {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
      char q[32] = {0};
      long period = 0;
      if (std::fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && q[0] != 'm') {
        const long quota = std::atol(q);
        if (quota > 0) hw = std::min<unsigned>(hw, static_cast<unsigned>(std::max<long>(1, quota / period)));
      }
      std::fclose(f);
    }
    else {  // cgroup v1
      long quota = -1, period = 0;
      if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(fq, "%ld", &quota) != 1) quota = -1;
        std::fclose(fq);
      }
      if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(fp, "%ld", &period) != 1) period = 0;
        std::fclose(fp);
      }
      if (quota > 0 && period > 0)
        hw = std::min<unsigned>(hw, static_cast<unsigned>(std::max<long>(1, quota / period)));
    }
    unsigned ranks = 1;
    for (const char *name : {"LOCAL_WORLD_SIZE", "WORLD_SIZE"})
      if (const char *e = std::getenv(name)) {
        const int r = std::atoi(e);
        if (r > 1) {
          ranks = static_cast<unsigned>(r);
          break;
        }
      }
    return std::max(1u, hw / ranks);
  }

  WorkerPool() {
    const unsigned hw = usable_cpus();
    int n = hw >= 16 ? 11 : hw >= 12 ? 7 : hw >= 6 ? 3 : hw >= 3 ? 1 : 0;
    if (const char *e = std::getenv("KC_HOST_THREADS")) {  // process default; kc_set_host_threads overrides
      const int want = std::atoi(e);
      if (want >= 1 && want <= 64) n = want - 1;
    }
    start_workers(n);
  }
  ~WorkerPool() { stop_workers(); }
  void start_workers(int n) {
    stop_.store(false);
    slots_ = std::unique_ptr<Slot[]>(new Slot[n > 0 ? n : 1]);
    // a new worker starts from the generation current at its creation: it must
    // not take an old job description for a new one
    const uint64_t g0 = gen_.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i) slots_[i].done.store(g0, std::memory_order_relaxed);
    for (int i = 0; i < n; ++i) threads_.emplace_back([this, i, g0] { loop(i, g0); });
  }
  void stop_workers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_.store(true);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
    threads_.clear();
    job_parts_ = 0;
  }
  static void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  void loop(int w, uint64_t seen) {
    // `seen` = the generation at creation; a job published before this thread
    // got to run must still be seen (and acknowledged)
    for (;;) {
      // spin for up to ~0.5 ms of wall time (a controller at a steady rate of a
      // few kHz finds the workers hot), then sleep on the condition variable: an
      // idle or slow caller must not burn the CPU quota of its container
      int spins = 0;
      auto t_spin = std::chrono::steady_clock::now();
      while (gen_.load(std::memory_order_acquire) == seen) {
        cpu_relax();
        if ((++spins & 255) == 0 &&
            std::chrono::steady_clock::now() - t_spin > std::chrono::microseconds(500)) {
          std::unique_lock<std::mutex> lk(mu_);
          sleepers_.fetch_add(1, std::memory_order_acq_rel);
          cv_.wait(lk, [&] {
            return gen_.load(std::memory_order_acquire) != seen || stop_.load();
          });
          sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        }
        if (stop_.load(std::memory_order_relaxed)) return;
      }
      if (stop_.load()) return;
      seen = gen_.load(std::memory_order_acquire);
      // the job description is published before gen_ (release/acquire)
      const size_t part = static_cast<size_t>(w) + 1;
      if (part < job_parts_) job_fn_(part);
      slots_[w].done.store(seen, std::memory_order_release);
    }
  }

  std::vector<std::thread> threads_;
  std::unique_ptr<Slot[]> slots_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_;
  alignas(64) std::atomic<uint64_t> gen_{0};
  alignas(64) std::atomic<int> sleepers_{0};
  std::function<void(size_t)> job_fn_;
  size_t job_parts_ = 0;
  std::atomic<bool> stop_{false};
};

}  // namespace kc
