// Shader clock under load: s_memtime (shader clocks) against wall_clock64 (constant 100 MHz) around a loop of fp32 MFMAs or of VALU FMAs
// on every CU.  hipcc -O3 --offload-arch=gfx950 clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int iters) {
  f32x16 acc[4];
  for (int s = 0; s < 4; ++s)
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = __builtin_fmaf(acc[s][r], b, a);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float t = 0.f;
  for (int s = 0; s < 4; ++s)
    for (int r = 0; r < 16; ++r) t += acc[s][r];
  if (t == 123.456f) sink[0] = t;
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = c1 - c0;
    out[blockIdx.x * 2 + 1] = w1 - w0;
  }
}

int main() {
  unsigned long long* out;
  float* sink;
  const int blocks = 256 * 3;      // 12 waves per CU, as the chain kernels
  hipMalloc(&out, blocks * 16);
  hipMalloc(&sink, 4);
  for (int mode = 0; mode < 2; ++mode)
    for (int iters : {2000, 20000, 200000}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
      else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(blocks * 2);
      hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
      std::vector<double> mhz;
      for (int b = 0; b < blocks; ++b) mhz.push_back(100.0 * (double)h[2 * b] / (double)h[2 * b + 1]);
      std::sort(mhz.begin(), mhz.end());
      // MFMA mode: 3 waves per SIMD x iters x 4 MFMAs x 64 cycles = the cycles the matrix pipe needs
      const double need = mode == 0 ? 3.0 * iters * 4 * 64 : 3.0 * iters * 64 * 4;
      printf("%s iters %6d: kernel %.3f ms | shader clock (s_memtime / wall_clock64) min %.0f med %.0f max %.0f MHz | block med %.0f clocks, pipe needs %.0f (%.2f)\n",
             mode == 0 ? "mfma f32 32x32x2" : "valu fma        ", iters, ms, mhz.front(), mhz[mhz.size() / 2], mhz.back(), (double)h[2 * (blocks / 2)], need,
             need / (double)h[2 * (blocks / 2)]);
    }
  return 0;
}
