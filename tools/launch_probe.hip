// Host cost of one kernel launch through the entry points HIP offers (a 1.1 KB argument block like the cycle
// kernel's): hipLaunchKernelGGL, hipLaunchKernel with a prepared pointer array, hipModuleLaunchKernel with the
// arguments as ONE buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER).  Times: the call itself, and call -> the kernel's
// word visible in pinned memory.     hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_probe tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
struct Big { unsigned long long *out; long long seq; char pad[1088]; };
__global__ void k_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) *b.out = b.seq; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  unsigned long long *host;
  CK(hipHostMalloc(&host, 64, hipHostMallocMapped));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipFunction_t f;
  CK(hipGetFuncBySymbol(&f, reinterpret_cast<const void *>(k_big)));
  Big b{};
  b.out = host;
  const int N = 3000;
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<double> call, seen;
    for (int i = 0; i < N + 200; ++i) {
      b.seq = i + 1 + mode * 100000;
      *host = 0;
      const double t0 = now();
      if (mode == 0) {
        hipLaunchKernelGGL(k_big, dim3(256), dim3(1024), 0, s, b);
      } else if (mode == 1) {
        void *args[] = {&b};
        (void)hipLaunchKernel(reinterpret_cast<const void *>(k_big), dim3(256), dim3(1024), args, 0, s);
      } else {
        size_t sz = sizeof(b);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &b, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        (void)hipModuleLaunchKernel(f, 256, 1, 1, 1024, 1, 1, 0, s, nullptr, extra);
      }
      const double t1 = now();
      while (*reinterpret_cast<volatile unsigned long long *>(host) != static_cast<unsigned long long>(b.seq)) {}
      const double t2 = now();
      if (i >= 200) { call.push_back(t1 - t0); seen.push_back(t2 - t0); }
    }
    std::sort(call.begin(), call.end());
    std::sort(seen.begin(), seen.end());
    const char *nm[] = {"hipLaunchKernelGGL", "hipLaunchKernel(args[])", "hipModuleLaunchKernel(buffer)"};
    printf("%-32s call p50 %.2f us  p10 %.2f | call -> word seen p50 %.2f us p10 %.2f\n", nm[mode], call[N / 2], call[N / 10],
           seen[N / 2], seen[N / 10]);
  }
  CK(hipStreamSynchronize(s));
  return 0;
}
