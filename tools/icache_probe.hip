// Diagnostic: cost of executing straight-line code for the first time in a kernel (cold I$)
// versus a second pass over the same code, and whether the I$ stays warm across launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N> struct Unroll {
  __device__ static inline void run(float &a, float &b) { a = a * 1.0001f + b; b = b * 0.9999f + a; Unroll<N - 1>::run(a, b); }
};
template <> struct Unroll<0> { __device__ static inline void run(float &, float &) {} };
__global__ void probe(float *out, unsigned long long *clk, int reps) {
  float a = threadIdx.x, b = blockIdx.x;
  unsigned long long t[5];
  t[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
    Unroll<1024>::run(a, b);   // 2048 dependent FMAs, ~16 KB of code
    t[r + 1] = __builtin_amdgcn_s_memtime();
    asm volatile("" : "+v"(a), "+v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
  if (threadIdx.x == 0) for (int r = 0; r < reps; ++r) clk[blockIdx.x * 4 + r] = t[r + 1] - t[r];
}
int main() {
  const int blocks = 256, threads = 64, reps = 3;
  float *out; unsigned long long *clk;
  hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, blocks * 4 * 8);
  std::vector<unsigned long long> h(blocks * 4);
  for (int launch = 0; launch < 4; ++launch) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, out, clk, reps);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), clk, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s[3] = {0, 0, 0};
    for (int b = 0; b < blocks; ++b) for (int r = 0; r < reps; ++r) s[r] += h[b * 4 + r];
    printf("launch %d: pass ticks avg  %.0f  %.0f  %.0f  (2048 FMAs each)\n", launch, s[0] / blocks, s[1] / blocks, s[2] / blocks);
  }
  return 0;
}
