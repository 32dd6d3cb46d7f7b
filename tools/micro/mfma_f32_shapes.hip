// Bare fp32 MFMA loops on random register operands, one or two waves per SIMD: FLOP/s of v_mfma_f32_32x32x2_f32 against
// v_mfma_f32_16x16x4_f32 (same FLOPs per cycle on paper; the chip's clock under load may differ by shape).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* __restrict__ clk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + i + 77) & 0xffff]; }
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 7], b[(u * 3 + j) & 7], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    dst[t] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
  } else {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 7], b[(u * 3 + j) & 7], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    dst[t] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
  }
}

int main() {
  std::vector<float> h(65536);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *src, *dst;
  hipMalloc(&src, h.size() * 4); hipMalloc(&dst, 4 * 512 * 2048);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  unsigned long long* clk; hipMalloc(&clk, 16 * 1024);
  for (int cfg = 0; cfg < 3; ++cfg) {
    // cfg 0: 256 blocks x 4 waves (1 wave/SIMD); cfg 1: 512 blocks x 4 waves (2 blocks per CU); cfg 2: 256 blocks x 8 waves
    const int blocks = cfg == 1 ? 512 : 256, threads = cfg == 2 ? 512 : 256, wps = cfg == 0 ? 1 : 2;
    for (int shape : {32, 16}) {
      const int iters = 20000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(threads), 0, 0, src, dst, iters, clk);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(threads), 0, 0, src, dst, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration per wave: 32 MFMAs x 4096 FLOP (32x32x2) or 64 MFMAs x 2048 FLOP (16x16x4) = 131072 FLOP
        const double flops = (double)blocks * (threads / 64) * iters * 131072.0;
        unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        if (rep == 2) printf("cfg %d (%d blocks x %d waves, %d waves/SIMD)  shape %dx%d: %.1f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz\n", cfg, blocks, threads / 64, wps,
                             shape, shape, ms, flops / ms * 1e-9, (double)hc[0] / (double)hc[1] * 0.1);
      }
    }
  }
  return 0;
}
