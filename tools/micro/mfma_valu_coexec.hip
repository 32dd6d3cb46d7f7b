// Do the fp32-input MFMA (v_mfma_f32_32x32x2_f32, which runs at the fp32 VECTOR rate) and VALU work of ANOTHER wave on the same SIMD
// execute concurrently on gfx950, or do they share the SIMD's fp32 lanes?  512-thread blocks, one per CU: waves 0-3 (one per SIMD) run
// an MFMA loop, waves 4-7 (their SIMD partners) a VALU loop of one kind; timed alone and together.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// kind: 0 v_fma_f32, 1 v_add_u32 (integer), 2 v_exp_f32, 3 ds_read_b128 (LDS), 4 v_pk_fma_f32
template <int MF32>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ dst, int mfma_iters, int valu_iters, int kind) {
  __shared__ float lds[4096];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = src[t & 0xffff];
  lds[threadIdx.x + 512] = src[(t + 7) & 0xffff];
  __syncthreads();
  if (wave < 4) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + i + 77) & 0xffff]; }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if (MF32) {
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 7], b[(u * 3 + j) & 7], acc[j], 0, 0, 0);
      }
    } else {
      bf16x8 xa, xb;
      for (int i = 0; i < 8; ++i) { xa[i] = (__bf16)a[i]; xb[i] = (__bf16)b[i]; }
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[j], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    dst[t] = s;
  } else {
    float x0 = src[t & 0xffff], x1 = src[(t + 1) & 0xffff], x2 = src[(t + 2) & 0xffff], x3 = src[(t + 3) & 0xffff];
    float y0 = x1, y1 = x2, y2 = x3, y3 = x0;
    unsigned u0 = t, u1 = t * 3, u2 = t * 5, u3 = t * 7;
    const float c = src[(t + 9) & 0xffff];
    if (kind == 0) {
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { x0 = fmaf(x0, c, y0); x1 = fmaf(x1, c, y1); x2 = fmaf(x2, c, y2); x3 = fmaf(x3, c, y3); }
      }
    } else if (kind == 1) {
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { u0 = u0 * 3u + u1; u1 = u1 + u2; u2 = u2 ^ u3; u3 = u3 + u0; }
      }
    } else if (kind == 2) {
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3); }
      }
    } else if (kind == 3) {
      const float4* p = reinterpret_cast<const float4*>(lds) + (threadIdx.x & 63);
      for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float4 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];
          x0 += v0.x; x1 += v1.y; x2 += v2.z; x3 += v3.w;
          asm volatile("" ::: "memory");
        }
      }
    }
    dst[t] = x0 + x1 + x2 + x3 + (float)(u0 + u1 + u2 + u3);
  }
}

template <int MF32>
float run(const float* src, float* dst, int mi, int vi, int kind) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MF32>, dim3(256), dim3(512), 0, 0, src, dst, mi, vi, kind);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  std::vector<float> h(65536);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *src, *dst;
  hipMalloc(&src, h.size() * 4); hipMalloc(&dst, 4 * 512 * 256);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const char* names[] = {"v_fma_f32", "integer VALU", "v_exp_f32", "ds_read_b128 + v_add"};
  const int MI = 4000;     // 4000 x 32 MFMAs x 64 cycles = 8.2 M cycles ~ 3.4 ms
  for (int mf32 = 1; mf32 >= 0; --mf32) {
    const int mi = mf32 ? MI : MI;    // bf16: 4000 x 64 x 32 cycles: the same
    const float tm = mf32 ? run<1>(src, dst, mi, 0, 0) : run<0>(src, dst, mi, 0, 0);
    printf("%s MFMA alone: %.3f ms\n", mf32 ? "f32 32x32x2" : "bf16 32x32x16", tm);
    for (int kind = 0; kind < 4; ++kind) {
      // size the VALU loop to about the MFMA loop's time
      int vi = 20000;
      float tv = mf32 ? run<1>(src, dst, 0, vi, kind) : run<0>(src, dst, 0, vi, kind);
      vi = (int)(vi * tm / tv);
      tv = mf32 ? run<1>(src, dst, 0, vi, kind) : run<0>(src, dst, 0, vi, kind);
      const float tb = mf32 ? run<1>(src, dst, mi, vi, kind) : run<0>(src, dst, mi, vi, kind);
      printf("  partner wave %-22s alone %.3f ms, together %.3f ms  (sum %.3f, max %.3f) -> %s\n", names[kind], tv, tb, tm + tv, tm > tv ? tm : tv,
             tb > 0.85f * (tm + tv) ? "EXCLUSIVE (times add)" : tb < 1.15f * (tm > tv ? tm : tv) ? "concurrent" : "partly overlapped");
    }
  }
  return 0;
}
