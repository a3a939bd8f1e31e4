// Diagnostic: can the CPU write hipMalloc'ed memory directly (large BAR), how fast, and does a
// kernel launched afterwards see the data?
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void sum_kernel(const double *p, int n, double *out) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  atomicAdd(out, s);
}
int main() {
  const int n = 12800;  // 100 KB of doubles
  double *d = nullptr, *out = nullptr;
  if (hipMalloc(&d, n * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, n * 8);
  hipDeviceSynchronize();
  hipPointerAttribute_t at{};
  hipPointerGetAttributes(&at, d);
  printf("devicePointer %p hostPointer %p type %d\n", at.devicePointer, at.hostPointer, (int)at.type);
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  std::vector<double> src(n);
  for (int rep = 0; rep < 5; ++rep) {
    for (int i = 0; i < n; ++i) src[i] = rep + i * 1e-3;
    if (sigsetjmp(jb, 1)) { printf("CPU write to device memory faulted\n"); return 0; }
    auto t0 = std::chrono::steady_clock::now();
    std::memcpy(d, src.data(), n * 8);
    __builtin_ia32_sfence();
    auto t1 = std::chrono::steady_clock::now();
    hipMemset(out, 0, 8);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, d, n, out);
    double h = 0;
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    double want = 0;
    for (int i = 0; i < n; ++i) want += src[i];
    printf("rep %d: cpu->vram 100KB in %.2f us; kernel sum %.6f expected %.6f %s\n", rep,
           std::chrono::duration<double, std::micro>(t1 - t0).count(), h, want,
           (std::abs(h - want) < 1e-6 * want + 1e-9) ? "OK" : "MISMATCH");
  }
  return 0;
}
