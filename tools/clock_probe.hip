// Diagnostic: in-kernel shader clock (s_memtime / s_memrealtime) of short
// kernels launched with host gaps vs back-to-back.  Not part of the product.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void probe(unsigned long long *out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  double x = threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
    out[2] = (unsigned long long)(x * 1e6);
  }
}

int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int gap_us : {0, 50, 100, 200, 1000, 10000}) {
    std::vector<double> mhz;
    for (int rep = 0; rep < 200; ++rep) {
      hipLaunchKernelGGL(probe, dim3(128), dim3(64), 0, s, d, 2000);
      hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, s);
      hipStreamSynchronize(s);
      if (rep >= 100) mhz.push_back(100.0 * (double)h[0] / (double)h[1]);
      if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
    }
    double m = 0;
    for (double v : mhz) m += v;
    printf("gap %6d us: shader clock %.0f MHz (kernel %.2f us realtime)\n", gap_us, m / mhz.size(), h[1] / 100.0);
  }
  return 0;
}
