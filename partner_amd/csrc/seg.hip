// Segmentation head SingleConvHead (det3d/models/seg_heads/seg_head.py:53-83, 176-195; secondary to the detection hot path).
//   seg_preds = Conv1x1( cat[ canvas (B,C1,H,W), bilinear_up(x2 (B,C2,h,w) -> H x W) ] )
// A 1x1 convolution commutes with bilinear interpolation (both are linear, the convolution mixes channels per pixel), so
//   seg_preds = Conv1x1_a(canvas) + bias + bilinear_up( Conv1x1_b(x2) ):
// the 512-channel concatenation at 512 x 512 (537 MB) never exists; the two convolutions run on the MFMA kernel, this file
// adds the low-resolution term and turns logits into per-point labels.
#include "pn_common.h"
#include <algorithm>

namespace {

// out (B,H,W,C) += bilinear( low (B,h,w,C) ), torch.nn.functional.interpolate(mode='bilinear', align_corners=False):
// source coordinate = (dst + 0.5) * (in / out) - 0.5, clamped below at 0; the upper neighbour is clamped to the last row / column
__global__ void bilinear_up_add_kernel(const float* __restrict__ low, int B, int h, int w, int C, int H, int W, float* __restrict__ out) {
  const int c4n = C / 4;
  const size_t total = (size_t)B * H * W * c4n;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    size_t p = i / c4n;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    float fy = ((float)Y + 0.5f) * sy - 0.5f, fx = ((float)X + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float4* src = reinterpret_cast<const float4*>(low) + (size_t)b * h * w * c4n;
    const float4 a = src[((size_t)y0 * w + x0) * c4n + c4], bq = src[((size_t)y0 * w + x1) * c4n + c4];
    const float4 cq = src[((size_t)y1 * w + x0) * c4n + c4], d = src[((size_t)y1 * w + x1) * c4n + c4];
    float4* o = reinterpret_cast<float4*>(out) + i;
    float4 v = *o;
    v.x += hy * (hx * a.x + lx * bq.x) + ly * (hx * cq.x + lx * d.x);
    v.y += hy * (hx * a.y + lx * bq.y) + ly * (hx * cq.y + lx * d.y);
    v.z += hy * (hx * a.z + lx * bq.z) + ly * (hx * cq.z + lx * d.z);
    v.w += hy * (hx * a.w + lx * bq.w) + ly * (hx * cq.w + lx * d.w);
    *o = v;
  }
}

// label of point i = 1 + argmax_c seg[b, y, x, c] at the point's BEV cell (first maximum, as torch.argmax);
// grid_ind rows are [z, y(theta), x(r)] as the reference's valid_grid_ind (seg_head.py:186-192, 2-D prediction map)
__global__ void seg_point_labels_kernel(const float* __restrict__ seg, int H, int W, int C, const int64_t* __restrict__ grid_ind, int n,
                                        int64_t* __restrict__ labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t y = grid_ind[(size_t)i * 3 + 1], x = grid_ind[(size_t)i * 3 + 2];
  if (y < 0 || y >= H || x < 0 || x >= W) { labels[i] = 0; return; }
  const float* p = seg + ((size_t)y * W + x) * C;
  float best = p[0];
  int arg = 0;
  for (int c = 1; c < C; ++c)
    if (p[c] > best) { best = p[c]; arg = c; }
  labels[i] = arg + 1;
}

}  // namespace

extern "C" {

int pn_bilinear_upsample_add_f32(const float* low, int batch, int h, int w, int c, int out_h, int out_w, float* out, pn_stream_t stream) {
  PN_REQUIRE(low && out && batch >= 1 && h >= 1 && w >= 1 && out_h >= 1 && out_w >= 1, "bilinear_upsample_add: bad arguments");
  PN_REQUIRE(c >= 4 && c % 4 == 0 && ((uintptr_t)low & 15) == 0 && ((uintptr_t)out & 15) == 0,
             "bilinear_upsample_add: channel count must be a multiple of 4 and the maps 16-byte aligned");
  const size_t total = (size_t)batch * out_h * out_w * (c / 4);
  hipLaunchKernelGGL(bilinear_up_add_kernel, dim3((unsigned)std::min<size_t>(8192, (total + 255) / 256)), dim3(256), 0, pn::S(stream), low, batch,
                     h, w, c, out_h, out_w, out);
  return pn::check_launch("bilinear_up_add_kernel");
}

int pn_seg_point_labels(const float* seg_sample, int h, int w, int classes, const int64_t* grid_ind, int n, int64_t* labels, pn_stream_t stream) {
  PN_REQUIRE(seg_sample && labels && h >= 1 && w >= 1 && classes >= 1 && n >= 0, "seg_point_labels: bad arguments");
  if (n == 0) return PN_OK;
  PN_REQUIRE(grid_ind != nullptr, "seg_point_labels: null grid_ind");
  hipLaunchKernelGGL(seg_point_labels_kernel, dim3(pn::cdiv(n, 256)), dim3(256), 0, pn::S(stream), seg_sample, h, w, classes, grid_ind, n, labels);
  return pn::check_launch("seg_point_labels_kernel");
}

}  // extern "C"
