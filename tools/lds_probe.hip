// Diagnostic: cost of a short LDS search loop under different lane->address maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void probe(const float *g, int S, float *out, unsigned long long *clk, int fill_words) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < fill_words; i += 512) lds[i] = g[i % (3 * S)];
  __syncthreads();
  const float *sx = lds, *sy = lds + S, *sz = lds + 2 * S;
  const int sub = threadIdx.x & 7;
  const float x = (threadIdx.x >> 3) * 0.01f, y = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float best = 3.4e38f; int arg = 0;
  if (MODE == 0) {  // 8 lanes per point, broadcast between groups
#pragma unroll 4
    for (int j = sub; j < S; j += 8) {
      const float dx = sx[j] - x, dy = sy[j] - y;
      const float d = dx * dx + (dy * dy + sz[j]);
      if (d < best) { best = d; arg = j; }
    }
  } else if (MODE == 1) {  // 64 lanes per point
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < S; j += 64) {
      const float dx = sx[j] - x, dy = sy[j] - y;
      const float d = dx * dx + (dy * dy + sz[j]);
      if (d < best) { best = d; arg = j; }
    }
  } else {  // same as 0 but from global memory
    const float *gx = g, *gy = g + S, *gz = g + 2 * S;
#pragma unroll 4
    for (int j = sub; j < S; j += 8) {
      const float dx = gx[j] - x, dy = gy[j] - y;
      const float d = dx * dx + (dy * dy + gz[j]);
      if (d < best) { best = d; arg = j; }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = best + arg;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  const int S = 101, blocks = 387;
  std::vector<float> h(3 * S);
  for (int i = 0; i < 3 * S; ++i) h[i] = (i * 37 % 101) * 0.01f;
  float *g, *out; unsigned long long *clk;
  CK(hipMalloc(&g, 3 * S * 4)); CK(hipMalloc(&out, blocks * 512 * 4)); CK(hipMalloc(&clk, blocks * 8));
  CK(hipMemcpy(g, h.data(), 3 * S * 4, hipMemcpyHostToDevice));
  std::vector<unsigned long long> hc(blocks);
  for (int lds_kb : {2, 63}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        const size_t lds = lds_kb * 1024; const int words = lds / 4;
        if (mode == 0) { hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(512), lds, 0, g, S, out, clk, words); }
        if (mode == 1) { hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(512), lds, 0, g, S, out, clk, words); }
        if (mode == 2) { hipFuncSetAttribute((const void *)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(512), lds, 0, g, S, out, clk, words); }
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(hc.data(), clk, blocks * 8, hipMemcpyDeviceToHost));
      double s = 0, mx = 0; for (auto v : hc) { s += v; if (v > mx) mx = v; }
      printf("lds %2d KB mode %d: loop ticks avg %.0f max %.0f (s_memtime, 100 MHz => x10 ns)\n", lds_kb, mode, s / blocks, mx);
    }
  }
  return 0;
}
