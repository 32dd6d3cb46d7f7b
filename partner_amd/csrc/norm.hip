// GroupNorm family on NHWC BEV maps (HBM-bound, two passes over a few MB).
//
// A statistics group is {a contiguous channel block} x {all theta rows} x {one range stratum}.
// That one shape covers nn.GroupNorm (strata = 1), RSNorm (norm.py:58-75: the reference slices
// the range axis, stacks the slices on channels, applies GroupNorm and un-stacks -- i.e. a
// GroupNorm whose groups are range strata) and the GroupNorm inside RangeStratified.
// Pass 1 writes fp64 partial (sum, sum of squares) per (batch, stratum, channel block, row
// split); pass 2 adds the splits in a fixed order (deterministic), normalises, applies the
// affine + activation and optionally the position-conditioned calibration x*W(pos)+b(pos)
// (center_head_parallel.py:268) as a second output.
//
// Thread mapping: 16-byte (4-channel) vectors; a block of 256 threads covers 256*4/C pixels per
// pass, consecutive lanes read consecutive 16 B => every wave instruction moves 1 KiB.
#include "pn_common.h"

namespace {

constexpr int kThreads = 256;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GnArgs {
  const float* x;
  int B, H, W, C, ps, co;
  int cgroups, strata, splits, rows_per_split;
  const float* gamma;
  const float* beta;
  float eps;
  int act;
  float* out;
  int ops, oco;
  const float* mul;
  const float* add;
  float* out2;
  double* part;  // [B][strata][cgroups][splits][2]
  float* stat;   // [B][strata][cgroups][2] = (mean, rstd), written by gn_finalize_kernel
};

// grid: (splits, strata, B)
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(GnArgs a) {
  __shared__ double red[2][kThreads * 4];
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int wps = a.W / a.strata;
  const int vpc = a.C / 4;                 // vectors per pixel
  const int cv = threadIdx.x % vpc;        // this thread's channel vector
  const int pl = threadIdx.x / vpc, ppb = kThreads / vpc;
  const int y0 = split * a.rows_per_split, y1 = min(a.H, y0 + a.rows_per_split);
  double sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
  const int npix = (y1 - y0) * wps;
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wps, x = s * wps + p % wps;
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(b * a.H + y) * a.W + x) * a.ps + a.co + cv * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sum[k] += v[k];
      sq[k] += (double)v[k] * v[k];
    }
  }
  // per-channel partials -> LDS [channel][pixel lane]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][(cv * 4 + k) * ppb + pl] = sum[k];
    red[1][(cv * 4 + k) * ppb + pl] = sq[k];
  }
  __syncthreads();
  // Fixed-order tree: every group owns a contiguous run of `per_group` entries; fold runs by 4
  // until at most 8 entries per group remain, then one thread per (group, moment) finishes.
  int n = kThreads * 4;
  int per_group = n / a.cgroups;
  while (per_group > 8) {  // uniform over the block
    const int q = n >> 2;  // 2*q <= 512 = two entries per thread at most
    const int i0 = threadIdx.x, i1 = threadIdx.x + kThreads;
    double v0 = 0.0, v1 = 0.0;
    if (i0 < 2 * q) {
      const int m = i0 / q, i = i0 - m * q;
      v0 = (red[m][4 * i] + red[m][4 * i + 1]) + (red[m][4 * i + 2] + red[m][4 * i + 3]);
    }
    if (i1 < 2 * q) {
      const int m = i1 / q, i = i1 - m * q;
      v1 = (red[m][4 * i] + red[m][4 * i + 1]) + (red[m][4 * i + 2] + red[m][4 * i + 3]);
    }
    __syncthreads();
    if (i0 < 2 * q) red[i0 / q][i0 % q] = v0;
    if (i1 < 2 * q) red[i1 / q][i1 % q] = v1;
    __syncthreads();
    n = q;
    per_group >>= 2;
  }
  if (threadIdx.x < a.cgroups * 2) {
    const int g = threadIdx.x >> 1, m = threadIdx.x & 1;
    double t = 0.0;
    for (int i = 0; i < per_group; ++i) t += red[m][g * per_group + i];
    a.part[((((size_t)b * a.strata + s) * a.cgroups + g) * a.splits + split) * 2 + m] = t;
  }
}

// one wave per statistics group: fixed-order sum of the row-split partials -> (mean, rstd)
__global__ __launch_bounds__(64) void gn_finalize_kernel(GnArgs a) {
  const int gidx = blockIdx.x;  // (b * strata + s) * cgroups + g
  const double* o = a.part + (size_t)gidx * a.splits * 2;
  double t0 = 0.0, t1 = 0.0;
  for (int k = threadIdx.x; k < a.splits; k += 64) {
    t0 += o[2 * k];
    t1 += o[2 * k + 1];
  }
  t0 = pn::wave_sum(t0);  // xor butterfly: the same association order on every run
  t1 = pn::wave_sum(t1);
  if (threadIdx.x == 0) {
    const double n = (double)(a.C / a.cgroups) * a.H * (a.W / a.strata);
    const double mean = t0 / n;
    double var = t1 / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    a.stat[2 * gidx] = (float)mean;
    a.stat[2 * gidx + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
}

__global__ __launch_bounds__(kThreads) void gn_apply_kernel(GnArgs a) {
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int wps = a.W / a.strata;
  const int cpg = a.C / a.cgroups;
  const float* smean = a.stat + ((size_t)b * a.strata + s) * a.cgroups * 2;
  const int vpc = a.C / 4;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc, ppb = kThreads / vpc;
  float mean[4], rstd[4], ga[4], be[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cv * 4 + k;
    mean[k] = smean[2 * (c / cpg)];
    rstd[k] = smean[2 * (c / cpg) + 1];
    ga[k] = a.gamma ? a.gamma[s * a.C + c] : 1.f;
    be[k] = a.beta ? a.beta[s * a.C + c] : 0.f;
  }
  const int y0 = split * a.rows_per_split, y1 = min(a.H, y0 + a.rows_per_split);
  const int npix = (y1 - y0) * wps;
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wps, x = s * wps + p % wps;
    const size_t pix = (size_t)(b * a.H + y) * a.W + x;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.x + pix * a.ps + a.co + cv * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = pn::apply_act((v[k] - mean[k]) * rstd[k] * ga[k] + be[k], a.act);
    *reinterpret_cast<f32x4*>(a.out + pix * a.ops + a.oco + cv * 4) = v;
    if (a.out2) {
      const size_t q = ((size_t)y * a.W + x) * a.C + cv * 4;
      const f32x4 m = *reinterpret_cast<const f32x4*>(a.mul + q);
      const f32x4 d = *reinterpret_cast<const f32x4*>(a.add + q);
      f32x4 w;
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = v[k] * m[k] + d[k];
      *reinterpret_cast<f32x4*>(a.out2 + pix * a.C + cv * 4) = w;
    }
  }
}

int pick_splits(int B, int H, int strata) {
  int splits = 1;
  while (splits < H && (long long)B * strata * splits < 512) splits *= 2;
  return splits;
}

}  // namespace

extern "C" {

size_t pn_groupnorm_workspace_bytes(int batch, int channel_groups, int range_strata) {
  return (size_t)batch * range_strata * channel_groups * (256 * 2 * sizeof(double) + 2 * sizeof(double));
}

int pn_groupnorm_strat_fwd(const float* x, int batch, int h, int w, int c, int pixel_stride, int channel_offset,
                           int channel_groups, int range_strata, const float* gamma, const float* beta, float eps, int act,
                           float* out, int out_pixel_stride, int out_channel_offset, const float* mul, const float* add,
                           float* out2, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && out && workspace, "groupnorm: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && c >= 1, "groupnorm: bad sizes");
  PN_REQUIRE(c % 4 == 0 && c <= 4 * kThreads && (4 * kThreads) % c == 0, "groupnorm: channel count must be a multiple of 4 dividing 1024");
  PN_REQUIRE(pixel_stride % 4 == 0 && channel_offset % 4 == 0 && out_pixel_stride % 4 == 0 && out_channel_offset % 4 == 0,
             "groupnorm: strides / offsets must be multiples of 4 floats");
  PN_REQUIRE(channel_groups >= 1 && c % channel_groups == 0 && channel_groups <= 128 && (channel_groups & (channel_groups - 1)) == 0,
             "groupnorm: channel_groups must be a power of two <= 128 dividing the channel count");
  PN_REQUIRE(range_strata >= 1 && w % range_strata == 0, "groupnorm: range axis not divisible by range_strata");
  PN_REQUIRE((out2 == nullptr) || (mul && add), "groupnorm: out2 needs mul and add");
  if (workspace_bytes < pn_groupnorm_workspace_bytes(batch, channel_groups, range_strata))
    return pn::fail(PN_ERR_WORKSPACE, "groupnorm: workspace too small");
  GnArgs a;
  a.x = x; a.B = batch; a.H = h; a.W = w; a.C = c; a.ps = pixel_stride; a.co = channel_offset;
  a.cgroups = channel_groups; a.strata = range_strata;
  a.splits = pick_splits(batch, h, range_strata);
  if (a.splits > 256) a.splits = 256;
  a.rows_per_split = pn::cdiv(h, a.splits);
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.act = act;
  a.out = out; a.ops = out_pixel_stride; a.oco = out_channel_offset;
  a.mul = mul; a.add = add; a.out2 = out2;
  a.part = static_cast<double*>(workspace);
  const size_t ngroups = (size_t)batch * range_strata * channel_groups;
  a.stat = reinterpret_cast<float*>(a.part + ngroups * 256 * 2);
  dim3 grid(a.splits, range_strata, batch);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)ngroups), dim3(64), 0, pn::S(stream), a);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  return pn::check_launch("groupnorm_strat");
}

}  // extern "C"
