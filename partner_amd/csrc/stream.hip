// Sector streaming: a sweep is cut into azimuth sectors that are processed one after the other.
//   pn_split_polar_sectors_f32   Voxelization.voxelize_streaming_polar, det3d/datasets/pipelines/voxelization.py:305-393:
//                                stable partition of the polar points by (sector, sample), azimuth shifted into the first sector's
//                                range, x / y recomputed, grid indices against the sector grid
//   pn_assemble_rows_f32         the row concatenations of the context-padding convolutions ConvContext / ConvBDCP,
//                                det3d/models/necks/rpn_context.py:10-44, 98-158 (torch.cat / F.pad along the azimuth axis):
//                                every output sample = up to three row pieces taken from other maps (or zeros)
#include "pn_common.h"
#include <algorithm>

namespace {

constexpr int kT = 256, kItems = 8, kMaxParts = 64;   // parts = sectors * samples

struct SplitArgs {
  const float* pts; int n_cap; int f; const int32_t* offs; int batch; int nsectors;
  float min_az, interval;             // fp32, as numpy forms them from the fp32 range array
  float lo[3], vs[3]; int g[3];       // base range / voxel size, SECTOR grid (theta cells / nsectors)
  float* out; int64_t* grid_ind; uint32_t* keys; int32_t* out_offs;
  uint32_t* tile; int ntiles;
};

// sector of an azimuth: i == 0: phi < hi_0; last: phi >= lo_last; else lo_i <= phi < hi_i  (voxelization.py:346-351)
__device__ __forceinline__ int sector_of(const SplitArgs& a, float phi) {
  int s = 0;
  for (int i = 1; i < a.nsectors; ++i)
    if (phi >= __fadd_rn(a.min_az, __fmul_rn((float)i, a.interval))) s = i;
  return s;
}

__device__ __forceinline__ int part_of(const SplitArgs& a, int i) {   // -1: past the last sample
  if (i >= min(a.offs[a.batch], a.n_cap)) return -1;
  int b = 0;
  while (b + 1 < a.batch && i >= a.offs[b + 1]) ++b;
  return sector_of(a, a.pts[(size_t)i * a.f + 1]) * a.batch + b;
}

__global__ __launch_bounds__(kT) void split_count_kernel(SplitArgs a) {
  const int base = (blockIdx.x * kT + threadIdx.x) * kItems;
  int part[kItems];
  for (int k = 0; k < kItems; ++k) part[k] = base + k < a.n_cap ? part_of(a, base + k) : -1;
  const int nparts = a.nsectors * a.batch;
  for (int p = 0; p < nparts; ++p) {
    uint32_t c = 0;
    for (int k = 0; k < kItems; ++k) c += part[k] == p;
    uint32_t tot;
    pn::block_exclusive_scan<kT>(c, &tot);
    if (threadIdx.x == 0) a.tile[(size_t)p * a.ntiles + blockIdx.x] = tot;
  }
}

__global__ __launch_bounds__(kT) void split_offsets_kernel(SplitArgs a) {
  uint32_t carry = 0;
  const int nparts = a.nsectors * a.batch;
  for (int p = 0; p < nparts; ++p) {
    if (threadIdx.x == 0) a.out_offs[p] = (int32_t)carry;
    for (int b0 = 0; b0 < a.ntiles; b0 += kT) {
      const int i = b0 + threadIdx.x;
      const uint32_t v = i < a.ntiles ? a.tile[(size_t)p * a.ntiles + i] : 0;
      uint32_t tot;
      const uint32_t ex = pn::block_exclusive_scan<kT>(v, &tot);
      if (i < a.ntiles) a.tile[(size_t)p * a.ntiles + i] = carry + ex;
      carry += tot;
    }
  }
  if (threadIdx.x == 0) a.out_offs[nparts] = (int32_t)carry;
}

__device__ __forceinline__ int cell_idx(float p, float lo, float vs, int g) {
  float q = (float)((double)__fsub_rn(p, lo) / (double)vs);   // IEEE fp32 quotient (as pn_polar_grid_index_f32)
  const float hi = (float)(g - 1);
  q = q < 0.f ? 0.f : q;
  q = q > hi ? hi : q;
  return (int)floorf(q);
}

__global__ __launch_bounds__(kT) void split_scatter_kernel(SplitArgs a) {
  const int base = (blockIdx.x * kT + threadIdx.x) * kItems;
  int part[kItems];
  for (int k = 0; k < kItems; ++k) part[k] = base + k < a.n_cap ? part_of(a, base + k) : -1;
  const int nparts = a.nsectors * a.batch;
  for (int p = 0; p < nparts; ++p) {
    uint32_t c = 0;
    for (int k = 0; k < kItems; ++k) c += part[k] == p;
    uint32_t tot;
    uint32_t pos = a.tile[(size_t)p * a.ntiles + blockIdx.x] + pn::block_exclusive_scan<kT>(c, &tot);
    const int s = p / a.batch, b = p - s * a.batch;
    const float shift = __fsub_rn(__fadd_rn(a.min_az, __fmul_rn((float)s, a.interval)), a.min_az);   // cur_pc_range[1] - pc_range[1]
    for (int k = 0; k < kItems; ++k) {
      if (part[k] != p) continue;
      const float* src = a.pts + (size_t)(base + k) * a.f;
      float* o = a.out + (size_t)pos * a.f;
      const float rho = src[0], phi = __fsub_rn(src[1], shift);
      o[0] = rho; o[1] = phi; o[2] = src[2];
      o[3] = __fmul_rn(rho, cosf(phi));
      o[4] = __fmul_rn(rho, sinf(phi));
      for (int c2 = 5; c2 < a.f; ++c2) o[c2] = src[c2];
      const int r = cell_idx(rho, a.lo[0], a.vs[0], a.g[0]), t = cell_idx(phi, a.lo[1], a.vs[1], a.g[1]), z = cell_idx(src[2], a.lo[2], a.vs[2], a.g[2]);
      if (a.grid_ind) {
        int64_t* gi = a.grid_ind + (size_t)pos * 4;
        gi[0] = b; gi[1] = z; gi[2] = t; gi[3] = r;
      }
      if (a.keys) a.keys[pos] = (uint32_t)((((size_t)b * a.g[2] + z) * a.g[1] + t) * a.g[0] + r);
      ++pos;
    }
  }
}

// ------------------------------------------------------------------------------------------------ row assembly
struct Piece {
  const float* src;   // first element of the first source row (already offset to sample / row / channel); nullptr: zeros
  int ps;             // source pixel stride (floats)
  int rows;
};
struct AsmArgs {
  float* out; int n_out, out_rows, w, c, out_ps, out_co;
  Piece piece[kMaxParts][3];
};

// one thread per 16 bytes of an output row
__global__ __launch_bounds__(kT) void assemble_rows_kernel(AsmArgs a) {
  const int c4 = a.c / 4;
  const size_t per = (size_t)a.out_rows * a.w * c4;
  const size_t total = per * a.n_out;
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < total; i += (size_t)gridDim.x * kT) {
    const int k = (int)(i / per);
    size_t r = i - (size_t)k * per;
    const int cv = (int)(r % c4);
    r /= c4;
    const int x = (int)(r % a.w);
    int y = (int)(r / a.w);
    float4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const Piece& pc = a.piece[k][p];
      if (y >= 0 && y < pc.rows && pc.src) v = *reinterpret_cast<const float4*>(pc.src + ((size_t)y * a.w + x) * pc.ps + cv * 4);
      y -= pc.rows;
    }
    const int yo = (int)((i - (size_t)k * per) / ((size_t)a.w * c4));
    *reinterpret_cast<float4*>(a.out + (((size_t)k * a.out_rows + yo) * a.w + x) * a.out_ps + a.out_co + cv * 4) = v;
  }
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---- ego-motion warp of the previous sweep's per-layer feature maps (PolarStreamBDCP.forward_one_sweep, polarstream.py:318-372):
// every cell (m = azimuth row, n = range column) of the polar map is taken to Cartesian coordinates at its LOWER edge
// (az = lo_a + m * span_a / H, r = lo_r + n * span_r / W: get_grids :223-238), rotated by the sample's 2x2 matrix, turned back into
// (rho, azimuth), normalised to [-1, 1] with the map's centre / half-span (get_center :239-247) and sampled bilinearly with zero padding
// (torch.nn.functional.grid_sample defaults: bilinear, zeros, align_corners = False).  NHWC maps (B, H, W, C); one thread per
// (cell, 4 channels).
struct WarpArgs {
  const float* in; const float* rot; float* out;
  int B, H, W, C;
  float lo_r, hi_r, lo_a, hi_a;
  int nsec, hs;   // input stacked sector-major: sample (sector * B + b), hs = H / nsec rows each
};
__global__ void polar_warp_kernel(WarpArgs a) {
  const int c4 = a.C >> 2;
  const long long total = (long long)a.B * a.H * a.W * c4;
  const float span_r = a.hi_r - a.lo_r, span_a = a.hi_a - a.lo_a;
  const float cen_r = (a.hi_r + a.lo_r) * 0.5f, cen_a = (a.hi_a + a.lo_a) * 0.5f, half_r = span_r * 0.5f, half_a = span_a * 0.5f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4);
    long long r = i / c4;
    const int n = (int)(r % a.W); r /= a.W;
    const int m = (int)(r % a.H);
    const int b = (int)(r / a.H);
    const float az = span_a / (float)a.H * (float)m + a.lo_a;
    const float rr = span_r / (float)a.W * (float)n + a.lo_r;
    const float x = rr * cosf(az), y = rr * sinf(az);
    const float* t = a.rot + (size_t)b * 4;
    const float xs = t[0] * x + t[1] * y, ys = t[2] * x + t[3] * y;
    const float rho = sqrtf(xs * xs + ys * ys), phi = atan2f(ys, xs);
    const float gx = (rho - cen_r) / half_r, gy = (phi - cen_a) / half_a;
    const float ix = ((gx + 1.f) * (float)a.W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)a.H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* base = a.in + q * 4;
    auto tap = [&](int yy, int xx, float wgt) {
      if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
        const int sec = yy / a.hs, ys_ = yy - sec * a.hs;
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((((size_t)sec * a.B + b) * a.hs + ys_) * a.W + xx) * a.C);
        acc += v * wgt;
      }
    };
    tap(y0, x0, wy0 * wx0);
    tap(y0, x0 + 1, wy0 * wx1);
    tap(y0 + 1, x0, wy1 * wx0);
    tap(y0 + 1, x0 + 1, wy1 * wx1);
    *reinterpret_cast<f32x4*>(a.out + i * 4) = acc;
  }
}

}  // namespace

extern "C" {

size_t pn_split_polar_sectors_workspace_bytes(int n_capacity, int nsectors, int batch) {
  return (size_t)nsectors * batch * pn::cdiv(std::max(n_capacity, 1), kT * kItems) * sizeof(uint32_t) + 256;
}

int pn_split_polar_sectors_f32(const float* points, int n_capacity, int f, const int32_t* sample_offsets, int batch, int nsectors,
                               const float* pc_range, const float* voxel_size, const int32_t* grid, float* out_points, int64_t* grid_ind,
                               uint32_t* keys, int32_t* out_offsets, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(points && sample_offsets && pc_range && voxel_size && grid && out_points && out_offsets && workspace, "split_polar_sectors: null pointer");
  PN_REQUIRE(n_capacity >= 0 && f >= 5 && batch >= 1 && nsectors >= 1 && nsectors * batch <= kMaxParts, "split_polar_sectors: bad sizes (sectors * batch <= 64)");
  PN_REQUIRE(grid[1] % nsectors == 0, "split_polar_sectors: the azimuth axis must divide into the sectors");
  if (workspace_bytes < pn_split_polar_sectors_workspace_bytes(n_capacity, nsectors, batch)) return pn::fail(PN_ERR_WORKSPACE, "split_polar_sectors: workspace too small");
  SplitArgs a;
  a.pts = points; a.n_cap = n_capacity; a.f = f; a.offs = sample_offsets; a.batch = batch; a.nsectors = nsectors;
  a.min_az = pc_range[1];
  a.interval = (pc_range[4] - pc_range[1]) / (float)nsectors;     // fp32 scalars, as numpy: (max_az - min_az) / nsectors
  for (int k = 0; k < 3; ++k) {
    a.lo[k] = pc_range[k];
    a.vs[k] = voxel_size[k];
    a.g[k] = grid[k];
  }
  a.g[1] = grid[1] / nsectors;
  PN_REQUIRE((uint64_t)batch * a.g[0] * a.g[1] * a.g[2] < (1ull << 32), "split_polar_sectors: more than 2^32 cells");
  a.out = out_points; a.grid_ind = grid_ind; a.keys = keys; a.out_offs = out_offsets;
  a.tile = static_cast<uint32_t*>(workspace);
  a.ntiles = pn::cdiv(std::max(n_capacity, 1), kT * kItems);
  hipStream_t st = pn::S(stream);
  hipLaunchKernelGGL(split_count_kernel, dim3(a.ntiles), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(split_offsets_kernel, dim3(1), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(split_scatter_kernel, dim3(a.ntiles), dim3(kT), 0, st, a);
  return pn::check_launch("split_polar_sectors");
}

int pn_assemble_rows_f32(const pn_row_piece* pieces, int n_out, int out_rows, int w, int c, float* out, int out_pixel_stride,
                         int out_channel_offset, pn_stream_t stream) {
  PN_REQUIRE(pieces && out && n_out >= 1 && n_out <= kMaxParts && out_rows >= 1 && w >= 1 && c >= 4 && c % 4 == 0, "assemble_rows: bad arguments");
  PN_REQUIRE(out_pixel_stride % 4 == 0 && out_channel_offset % 4 == 0 && ((uintptr_t)out & 15) == 0, "assemble_rows: output must be 16-byte aligned");
  AsmArgs a;
  a.out = out; a.n_out = n_out; a.out_rows = out_rows; a.w = w; a.c = c; a.out_ps = out_pixel_stride; a.out_co = out_channel_offset;
  for (int k = 0; k < n_out; ++k) {
    int rows = 0;
    for (int p = 0; p < 3; ++p) {
      const pn_row_piece& s = pieces[k * 3 + p];
      PN_REQUIRE(s.rows >= 0 && (s.src == nullptr || (s.pixel_stride % 4 == 0 && ((uintptr_t)s.src & 15) == 0)), "assemble_rows: bad piece");
      a.piece[k][p].src = s.src; a.piece[k][p].ps = s.pixel_stride; a.piece[k][p].rows = s.rows;
      rows += s.rows;
    }
    PN_REQUIRE(rows == out_rows, "assemble_rows: the pieces of a sample must add up to out_rows");
  }
  const size_t total = (size_t)n_out * out_rows * w * (c / 4);
  hipLaunchKernelGGL(assemble_rows_kernel, dim3((unsigned)std::min<size_t>(4096, (total + kT - 1) / kT)), dim3(kT), 0, pn::S(stream), a);
  return pn::check_launch("assemble_rows_kernel");
}

int pn_polar_warp_f32(const float* in, const float* rot2x2, int batch, int nsectors, int h, int w, int c, float range_lo, float range_hi,
                      float azimuth_lo, float azimuth_hi, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && rot2x2 && out && in != out && batch >= 1 && h >= 1 && w >= 1 && c >= 4 && c % 4 == 0, "polar_warp: bad arguments (channels a multiple of 4)");
  PN_REQUIRE(nsectors >= 1 && h % nsectors == 0, "polar_warp: the azimuth rows must divide into the sectors");
  PN_REQUIRE(range_hi > range_lo && azimuth_hi > azimuth_lo, "polar_warp: empty range");
  WarpArgs a{in, rot2x2, out, batch, h, w, c, range_lo, range_hi, azimuth_lo, azimuth_hi, nsectors, h / nsectors};
  const long long total = (long long)batch * h * w * (c / 4);
  hipLaunchKernelGGL(polar_warp_kernel, dim3((unsigned)std::min<long long>(65535, (total + 255) / 256)), dim3(256), 0, pn::S(stream), a);
  return pn::check_launch("polar_warp_kernel");
}

}  // extern "C"
