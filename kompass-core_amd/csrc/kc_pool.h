// Worker pool of the host-trig FALLBACK (kc_dwa.hip, rollout_impl: `device_trig` off, another libm, |yaw| beyond
// the restated range): the rows of the cos / sin table are shared between the caller and a few workers, the call
// returns when all of them are done.  Rounds 1-3 ran this on the critical path of every cycle (spinning workers,
// asynchronous tickets, staged hand-off to a kernel that was already waiting); since device trig (round 3) it is
// a fallback and since round 4 it is only this: a blocking parallel-for.  Threads are created on first use.
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
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
  ~WorkerPool() { stop_workers(); }

  int workers() {
    std::lock_guard<std::mutex> serial(run_mu_);
    return static_cast<int>(threads_.size());
  }
  // kc_set_host_threads: `total` threads including the caller
  void resize(int total) {
    std::lock_guard<std::mutex> serial(run_mu_);
    stop_workers();
    start_workers(std::max(0, std::min(total, 64) - 1));
  }

  // fn(begin, end) over [0, n) split into one contiguous chunk per thread; the caller takes the first chunk.
  // Serial when n is small or no workers exist.  One job at a time (several contexts may call from several threads).
  template <typename F>
  void parallel_for(size_t n, size_t min_chunk, F &&fn) {
    std::unique_lock<std::mutex> serial(run_mu_);
    const size_t parts = std::min<size_t>(threads_.size() + 1, min_chunk ? n / min_chunk : n);
    if (parts <= 1) {
      serial.unlock();
      fn(size_t(0), n);
      return;
    }
    auto part = [&fn, n, parts](size_t p) {
      const size_t b = n * p / parts, e = n * (p + 1) / parts;
      if (b < e) fn(b, e);
    };
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = part;
      job_parts_ = parts;
      pending_ = threads_.size();
      ++gen_;
    }
    cv_.notify_all();
    part(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  WorkerPool() {
    const unsigned hw = usable_cpus();
    start_workers(hw >= 16 ? 11 : hw >= 12 ? 7 : hw >= 6 ? 3 : hw >= 3 ? 1 : 0);
  }

  // CPUs this process may actually use: hardware threads, capped by the cgroup CPU quota, shared between the
  // ranks of a node (one process per GPU)
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

  void start_workers(int n) {
    unsigned long long g0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = false;
      g0 = gen_;  // a new worker starts from the generation current at its CREATION (the caller holds run_mu_: no
                  // job can be posted before this returns; read inside the thread it could already be the next one)
    }
    for (int w = 0; w < n; ++w) threads_.emplace_back([this, w, g0] { loop(static_cast<size_t>(w), g0); });
  }
  void stop_workers() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
    threads_.clear();
  }
  void loop(size_t w, unsigned long long seen) {
    for (;;) {
      std::function<void(size_t)> job;
      size_t parts = 0;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        job = job_;
        parts = job_parts_;
      }
      if (job && w + 1 < parts) job(w + 1);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (pending_ > 0 && --pending_ == 0) done_cv_.notify_all();
      }
    }
  }

  std::mutex run_mu_;  // one job at a time; the worker set is stable while held
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  std::function<void(size_t)> job_;
  size_t job_parts_ = 0, pending_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

}  // namespace kc
