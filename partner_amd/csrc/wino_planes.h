// The F(4, 3) planes of conv_wchain.hip, shared with the kernels that WRITE them from something other than a convolution's accumulators
// (norm.hip: the head's RSNorm + ReLU written straight as planes).
//   V[p 6][cg C/8][h 2][b][y H+2][xq W/4][j 4]   channel 8 cg + 4 h + j; rows y = 0 and y = H + 1 of every image are zero padding, written by
//   whoever writes the first / last image row.  (H, W) = the FRAME: the stored map or, transposed, its transpose (Winograd axis = the map's H).
#pragma once
#include <hip/hip_runtime.h>

namespace pn {

using wp_f32x4 = __attribute__((ext_vector_type(4))) float;

// B^T d for four channels at once: d[0..5] = the input pixels x = 4 t - 1 .. 4 t + 4 of a row.  One expression tree for every writer: the
// planes of a map are bit-identical whoever forms them.
__device__ __forceinline__ void wino4_input_transform4(const wp_f32x4 (&d)[6], wp_f32x4 (&v)[6]) {
  const wp_f32x4 c4 = {4.f, 4.f, 4.f, 4.f}, m4 = {-4.f, -4.f, -4.f, -4.f}, m5 = {-5.f, -5.f, -5.f, -5.f}, c2 = {2.f, 2.f, 2.f, 2.f}, m2 = {-2.f, -2.f, -2.f, -2.f};
  const wp_f32x4 e = __builtin_elementwise_fma(m4, d[2], d[4]), o = __builtin_elementwise_fma(m4, d[1], d[3]);
  const wp_f32x4 f = d[4] - d[2], t = d[3] - d[1];
  v[0] = __builtin_elementwise_fma(c4, d[0], __builtin_elementwise_fma(m5, d[2], d[4]));
  v[1] = e + o;
  v[2] = e - o;
  v[3] = __builtin_elementwise_fma(c2, t, f);
  v[4] = __builtin_elementwise_fma(m2, t, f);
  v[5] = __builtin_elementwise_fma(c4, d[1], __builtin_elementwise_fma(m5, d[3], d[5]));
}

// store the six fragments of (image img, frame row r, quad xq, channels 4 c4 .. 4 c4 + 3) and, next to the first / last row, the zero padding rows
__device__ __forceinline__ void wino4_store_planes(float* planes, const wp_f32x4 (&v)[6], int c4, int c4n, size_t plane_floats, int img, int r, int xq, int H, int Wq) {
  float* o = planes + (size_t)c4 * plane_floats + ((size_t)(img * (H + 2) + r + 1) * Wq + xq) * 4;      // plane (cg = c4 >> 1, h = c4 & 1)
  const size_t pstride = (size_t)c4n * plane_floats;
  const wp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 6; ++q) *reinterpret_cast<wp_f32x4*>(o + q * pstride) = v[q];
  if (r == 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<wp_f32x4*>(o + q * pstride - (size_t)Wq * 4) = z;
  }
  if (r == H - 1) {
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<wp_f32x4*>(o + q * pstride + (size_t)Wq * 4) = z;
  }
}

}  // namespace pn
