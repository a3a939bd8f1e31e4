// Launch overhead: three dependent short kernels as three launches vs one
// hipGraphLaunch of a captured graph (args through a device block).
// hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o tools/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k1(volatile long long *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[8] + 1; }
__global__ void k2(volatile long long *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] = p[0] + 1; }
__global__ void k3(volatile long long *p, long long *host) { if (threadIdx.x == 0) { __threadfence_system(); host[0] = p[1] + 1; } }
int main() {
  long long *d, *h;
  CK(hipMalloc(&d, 128));
  CK(hipHostMalloc(&h, 64));
  CK(hipMemset(d, 0, 128));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  volatile long long *hv = h;
  auto run = [&](bool graph, hipGraphExec_t ge, int iters) {
    double tot = 0;
    for (int i = 0; i < iters; ++i) {
      const long long seq = 1000 + i * 3;
      // "argument block": a host store into device memory is not portable here; use a tiny async copy substitute:
      // the probe keeps p[8] fixed and polls for a change of host[0]
      hv[0] = -1;
      auto t0 = std::chrono::steady_clock::now();
      if (graph) {
        if (hipGraphLaunch(ge, s) != hipSuccess) return -1.0;
      } else {
        hipLaunchKernelGGL(k1, dim3(256), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k2, dim3(256), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, s, d, h);
      }
      while (hv[0] == -1) {}
      tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      (void)seq;
    }
    return tot / iters;
  };
  printf("3 launches warm-up: %.2f us\n", run(false, nullptr, 200));
  printf("3 launches        : %.2f us per chain\n", run(false, nullptr, 2000));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k1, dim3(256), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k2, dim3(256), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, s, d, h);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("graph warm-up     : %.2f us\n", run(true, ge, 200));
  printf("graph launch      : %.2f us per chain\n", run(true, ge, 2000));
  printf("3 launches again  : %.2f us per chain\n", run(false, nullptr, 2000));
  return 0;
}
