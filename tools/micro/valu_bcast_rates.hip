// What does the PFN's layer-1 inner loop cost per term?  One wave owns a pillar: y[lane] += w[lane][k] * h[k] with h[k] living in lane k of a
// register -- the broadcast is the question.  Variants, each 64 terms per iteration, two waves per SIMD (the kernel's occupancy):
//   0  v_pk_fma_f32, VGPR operands, ONE dependent chain                  (the arithmetic alone, dependent)
//   1  v_pk_fma_f32, VGPR operands, four independent chains              (the arithmetic alone)
//   2  v_readlane_b32 -> SGPR -> v_pk_fma_f32 with the SGPR operand, one chain  (what pfn.hip does)
//   3  same, four chains
//   4  32 v_readlane_b32 alone
//   5  h through LDS: one ds_write_b32 + 8 ds_read_b128 (same address in every lane) per 32 values, then v_pk_fma_f32 on VGPRs, one chain
//   6  v_fma_f32 (not packed) with v_readlane operands, two chains (the r4 form)
//   7  scalar fmaf source the compiler pairs into v_pk_fma_f32 + one v_mov_b32 per packed fma: the moves cost nothing beside the packed fmas
//   8  v_pk_fma_f32 on distinct register pairs, four chains
// MI355X, r5: 0: 7.6   1: 6.6   2: 10.2   3: 9.7   5: 9.9   6: 9.9   7: 6.2   8: 6.2 cycles of the SIMD per term (at 2.4 GHz): a packed fp32 fma
// does not issue every 4 cycles from two waves per SIMD, the readlane in front of it costs 3.6, LDS instead of readlane buys nothing.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_bcast_rates.hip -o /tmp/vbr && /tmp/vbr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float lane_bcast(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); }

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
  __shared__ __attribute__((aligned(16))) float s_h[4][64];
  const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x2 w[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) w[i] = f32x2{src[(t * 64 + 2 * i) & 0xffff], src[(t * 64 + 2 * i + 1) & 0xffff]};
  float h = src[(t + 99) & 0xffff];
  f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  float sacc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      if constexpr (V == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[0] = __builtin_elementwise_fma(w[i], f32x2{h, h}, acc[0]);
      } else if constexpr (V == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_elementwise_fma(w[i], f32x2{h, h}, acc[i & 3]);
      } else if constexpr (V == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { const float hb = lane_bcast(h, i); acc[0] = __builtin_elementwise_fma(w[i], f32x2{hb, hb}, acc[0]); }
      } else if constexpr (V == 3) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { const float hb = lane_bcast(h, i); acc[i & 3] = __builtin_elementwise_fma(w[i], f32x2{hb, hb}, acc[i & 3]); }
      } else if constexpr (V == 4) {
#pragma unroll
        for (int i = 0; i < 32; ++i) sacc += lane_bcast(h, i);     // scalar adds: the readlanes are what is counted
      } else if constexpr (V == 5) {
        s_h[wv][lane] = h;
        f32x4 hv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = *reinterpret_cast<const f32x4*>(&s_h[wv][4 * i]);
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[0] = __builtin_elementwise_fma(w[i], f32x2{hv[i >> 2][i & 3], hv[i >> 2][i & 3]}, acc[0]);
      } else if constexpr (V == 7) {      // clock calibration: plain v_fma_f32, eight independent chains (4 cycles per instruction by the book)
        float c[8] = {acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], acc[3][0], acc[3][1]};
#pragma unroll
        for (int i = 0; i < 32; ++i) { c[i & 7] = fmaf(w[i][0], h, c[i & 7]); }
#pragma unroll
        for (int i = 0; i < 32; ++i) { c[i & 7] = fmaf(w[i][1], h, c[i & 7]); }
        acc[0] = f32x2{c[0], c[1]}; acc[1] = f32x2{c[2], c[3]}; acc[2] = f32x2{c[4], c[5]}; acc[3] = f32x2{c[6], c[7]};
      } else if constexpr (V == 8) {      // packed fma, both sources register PAIRS that differ (no op_sel broadcast), four chains
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_elementwise_fma(w[i], w[(i + 5) & 31], acc[i & 3]);
      } else {
        float a0 = acc[0][0], a1 = acc[0][1];
#pragma unroll
        for (int i = 0; i < 32; ++i) { const float hb = lane_bcast(h, i); a0 = fmaf(w[i][0], hb, a0); a1 = fmaf(w[i][1], hb, a1); }
        acc[0] = f32x2{a0, a1};
      }
      h = h * 0.999f + acc[0][0] * 1e-9f;      // the next pillar's h depends on nothing expensive; keeps the loop from being hoisted
    }
  }
  dst[t] = acc[0][0] + acc[0][1] + acc[1][0] + acc[2][1] + acc[3][0] + sacc + h;
}

template <int V>
void run(const float* src, float* dst, const char* what) {
  const int iters = 4000, blocks = 512, threads = 256;      // two blocks per CU: two waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, src, dst, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  // per SIMD: 2 waves x iters x 64 terms
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters * 64);
  printf("%d  %-70s %.2f ms   %.1f cycles per term per wave-slot (2.4 GHz assumed)\n", V, what, ms, cyc);
}

int main() {
  std::vector<float> h(65536);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *src, *dst;
  hipMalloc(&src, h.size() * 4); hipMalloc(&dst, 4 * 512 * 256);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>(src, dst, "pk_fma VGPR operands, one chain");
  run<1>(src, dst, "pk_fma VGPR operands, four chains");
  run<2>(src, dst, "readlane -> pk_fma, one chain (pfn.hip)");
  run<3>(src, dst, "readlane -> pk_fma, four chains");
  run<4>(src, dst, "32 readlanes alone");
  run<5>(src, dst, "h through LDS (1 write + 8 broadcast b128 reads), pk_fma, one chain");
  run<6>(src, dst, "readlane -> two v_fma_f32, two chains (r4 form)");
  run<7>(src, dst, "plain v_fma_f32, eight chains: 64 instructions per 64 'terms' (clock calibration)");
  run<8>(src, dst, "pk_fma, distinct register pairs, four chains");
  return 0;
}
