// Launch path of one kernel, two ways ("the hipStreamWaitValue64-gated pre-queued kernel or a
// measured reason it cannot work"):
//   plain : hipLaunchKernelGGL at t0, the kernel's first wave posts a word into pinned memory -> host sees it
//   gated : hipStreamWaitValue64 (>= seq) + the kernel are queued AHEAD of time; at t0 the host stores seq into the
//           gate word -> the command processor's poll sees it -> dispatch -> first wave posts -> host sees it
// for a grid of 256 workgroups x 1024 threads (the cycle kernel's shape) and of one small workgroup, with the gate
// word in pinned host memory, in signal memory (hipMallocSignalMemory) and in device memory written over the BAR.
// hipcc --offload-arch=gfx950 -O2 tools/gate_probe.hip -o tools/gate_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void post_kernel(long long *host, long long v) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    __hip_atomic_store(host, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Stat {
  std::vector<double> v;
  void add(double x) { v.push_back(x); }
  double q(double p) { std::sort(v.begin(), v.end()); return v.empty() ? -1 : v[static_cast<size_t>(p * (v.size() - 1))]; }
};

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  long long *host;
  CK(hipHostMalloc(&host, 64));
  volatile long long *hv = host;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int iters = 2000;
  for (int shape = 0; shape < 2; ++shape) {
    const dim3 grid(shape == 0 ? 256 : 1), block(shape == 0 ? 1024 : 64);
    // ---- plain launches
    Stat call, total;
    for (int i = 0; i < iters + 100; ++i) {
      const long long seq = 1 + i;
      const double t0 = now_us();
      hipLaunchKernelGGL(post_kernel, grid, block, 0, s, host, seq);
      const double t1 = now_us();
      while (hv[0] != seq) {}
      const double t2 = now_us();
      if (i >= 100) { call.add(t1 - t0); total.add(t2 - t0); }
    }
    printf("grid %3u x %4u  plain launch : call %.2f us (p50), call -> first wave's word seen %.2f us p50 / %.2f p90\n", grid.x, block.x,
           call.q(0.5), total.q(0.5), total.q(0.9));
    // ---- gated, three kinds of gate memory
    for (int kind = 0; kind < 3; ++kind) {
      long long *gate = nullptr;
      const char *name = kind == 0 ? "pinned host" : kind == 1 ? "signal memory" : "device memory over the BAR";
      hipError_t e = hipSuccess;
      if (kind == 0) e = hipHostMalloc(&gate, 64);
      else if (kind == 1) e = hipExtMallocWithFlags(reinterpret_cast<void **>(&gate), 64, hipMallocSignalMemory);
      else e = hipMalloc(&gate, 64);
      if (e != hipSuccess) { printf("  gate in %s: allocation failed (%s)\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
      if (kind == 2) {
        // host-addressable only with a large BAR: probe by attribute
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, gate) != hipSuccess) { (void)hipGetLastError(); }
      }
      volatile long long *gv = gate;
      bool ok = true;
      if (kind == 2) {
        // a store through the BAR faults where device memory is not host-visible: try it guarded by hipMemset first
        CK(hipMemset(gate, 0, 64));
        CK(hipDeviceSynchronize());
      } else {
        gv[0] = 0;
      }
      Stat open_to_seen, queue_cost;
      const long long base = 1000000ll * (kind + 1 + 3 * shape);
      for (int i = 0; i < iters + 100 && ok; ++i) {
        const long long seq = base + i;
        const double q0 = now_us();
        e = hipStreamWaitValue64(s, gate, static_cast<uint64_t>(seq), hipStreamWaitValueGte, ~0ull);
        if (e != hipSuccess) { printf("  gate in %s: hipStreamWaitValue64 failed (%s)\n", name, hipGetErrorString(e)); (void)hipGetLastError(); ok = false; break; }
        hipLaunchKernelGGL(post_kernel, grid, block, 0, s, host, seq);
        const double q1 = now_us();
        // let the command processor reach the wait (a controller would queue at the end of the previous cycle)
        const double w0 = now_us();
        while (now_us() - w0 < 30.0) {}
        const double t0 = now_us();
        gv[0] = seq;
#if defined(__x86_64__)
        _mm_sfence();
#endif
        while (hv[0] != seq) {
          if (now_us() - t0 > 2.0e6) { printf("  gate in %s: kernel never ran\n", name); ok = false; break; }
        }
        const double t1 = now_us();
        if (i >= 100) { open_to_seen.add(t1 - t0); queue_cost.add(q1 - q0); }
      }
      if (ok)
        printf("  gate in %-28s: queueing wait + kernel %.2f us; gate store -> first wave's word seen %.2f us p50 / %.2f p90\n", name,
               queue_cost.q(0.5), open_to_seen.q(0.5), open_to_seen.q(0.9));
      CK(hipStreamSynchronize(s));
      if (kind == 0) (void)hipHostFree(gate); else (void)hipFree(gate);
    }
  }
  return 0;
}
