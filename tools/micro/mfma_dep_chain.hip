// Does a DEPENDENT chain of v_mfma_f32_32x32x2_f32 (every MFMA accumulates into the same 32x32 tile, as a wave with a single 32x32
// output tile issues them) run at the rate of independent accumulators?  NACC = 1, 2, 4 accumulators per wave, 1 or 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_dep_chain.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + i + 77) & 0xffff]; }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 7], b[(u * 3 + j) & 7], acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  dst[t] = s;
}

template <int NACC>
void run(const float* src, float* dst, int blocks, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, src, dst, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = (double)blocks * (threads / 64) * iters * 32 * 4096.0;
  printf("%d accumulator(s), %d blocks x %d waves: %.1f ms  %.1f TFLOP/s\n", NACC, blocks, threads / 64, ms, flops / ms * 1e-9);
}

int main() {
  std::vector<float> h(65536);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *src, *dst;
  hipMalloc(&src, h.size() * 4); hipMalloc(&dst, 4 * 512 * 2048);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    run<1>(src, dst, 256, threads);
    run<2>(src, dst, 256, threads);
    run<4>(src, dst, 256, threads);
  }
  // r5: more waves per SIMD (blocks per CU), the occupancy of the sparse block-per-group kernel: 4 waves per SIMD, two accumulators each
  run<2>(src, dst, 512, 512);
  run<2>(src, dst, 1024, 256);
  run<2>(src, dst, 1024, 512);
  run<2>(src, dst, 768, 256);      // 3 waves per SIMD
  run<4>(src, dst, 768, 256);
  run<1>(src, dst, 1024, 256);
  run<4>(src, dst, 1024, 256);
  return 0;
}
