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
#include "wino_planes.h"

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
  int o2ps, o2co;
  double* part;  // [B][strata][cgroups][splits][2]
  float* stat;   // [B][strata][cgroups][2] = (mean, rstd), written by gn_finalize_kernel
  int xt;        // 1: x is stored transposed, [b][x][y][channels] (pn_groupnorm_strat_planes_f32 mode 2); 0 everywhere else
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
  const int rows = y1 - y0;
  for (int p = pl; p < npix; p += ppb) {
    // (a transposed source is walked along its own contiguous axis: y fastest)
    const int y = a.xt ? y0 + p % rows : y0 + p / wps, x = s * wps + (a.xt ? p / rows : p % wps);
    const size_t pix = a.xt ? ((size_t)b * a.W + x) * a.H + y : ((size_t)b * a.H + y) * a.W + x;
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + pix * a.ps + a.co + cv * 4);
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
__device__ __forceinline__ void gn_finalize_group(const GnArgs& a, int gidx, int lane, float* stat) {      // gidx = (b * strata + s) * cgroups + g
  const double* o = a.part + (size_t)gidx * a.splits * 2;
  double t0 = 0.0, t1 = 0.0;
  for (int k = lane; k < a.splits; k += 64) {
    t0 += o[2 * k];
    t1 += o[2 * k + 1];
  }
  t0 = pn::wave_sum(t0);  // xor butterfly: the same association order on every run
  t1 = pn::wave_sum(t1);
  if (lane == 0) {
    const double n = (double)(a.C / a.cgroups) * a.H * (a.W / a.strata);
    const double mean = t0 / n;
    double var = t1 / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stat[2 * gidx] = (float)mean;
    stat[2 * gidx + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(GnArgs a) { gn_finalize_group(a, blockIdx.x, threadIdx.x, a.stat); }

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
      *reinterpret_cast<f32x4*>(a.out2 + pix * a.o2ps + a.o2co + cv * 4) = w;
    }
  }
}

// The normalisation pass writing the F(4, 3) PLANES of its result (conv_wchain.hip) instead of the map: RSNorm + ReLU of the head's shared
// convolution feed only 3x3 convolutions, which then run in the Winograd domain without a separate transform pass.  One thread = one quad
// of the frame (four pixels along the Winograd axis) x four channels: it normalises the quad's six input pixels itself (the two
// neighbours' values are recomputed, nothing is exchanged) and stores six fragments; out2 = out * mul + add likewise (the
// position-calibrated copy for the heat-map branch).  transposed: the Winograd axis is the map's H (rows of the frame = range positions).
constexpr int kFoldGroups = 64;      // up to this many statistics groups gn_apply_planes_kernel finalizes itself (no finalize launch)
struct GnPlanesArgs {
  GnArgs g;
  float* planes;
  float* planes2;
  int fold;            // every block folds the row-split partials of all groups first (the arithmetic of gn_finalize_kernel, wave per group)
  int transposed;
  int FH, FW, Wq;      // the frame
  unsigned plane_floats;
};

__global__ __launch_bounds__(256) void gn_apply_planes_kernel(GnPlanesArgs p) {
  const GnArgs& a = p.g;
  __shared__ float folded[2 * kFoldGroups];
  const float* stat = a.stat;
  if (p.fold) {
    const int ngroups = a.B * a.strata * a.cgroups;
    for (int g = threadIdx.x >> 6; g < ngroups; g += 4) gn_finalize_group(a, g, threadIdx.x & 63, folded);
    __syncthreads();
    stat = folded;
  }
  const int c4n = a.C >> 2;
  const int wps = a.W / a.strata, cpg = a.C / a.cgroups;
  const long long total = (long long)a.B * p.FH * p.Wq * c4n;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    // lanes along the quads of a frame row (the planes' contiguous axis), then rows, channel vectors, images
    const int xq = (int)(it % p.Wq);
    long long rest = it / p.Wq;
    const int r = (int)(rest % p.FH); rest /= p.FH;
    const int c4 = (int)(rest % c4n);
    const int b = (int)(rest / c4n);
    f32x4 d[6], d2[6];
    // every load of the quad is requested before the first use (the kernel is a few waves per CU: its time is the memory latency times the
    // number of dependent rounds); out-of-map neighbours read a clamped pixel and are zeroed afterwards
    f32x4 xv[6], mv[6], av[6];
    size_t q6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int fx = min(max(4 * xq - 1 + k, 0), p.FW - 1);      // position along the Winograd axis
      const int my = p.transposed ? fx : r, mx = p.transposed ? r : fx;      // map coordinates
      // (xt: the source, mul and add are stored as the frame, [b][mx][my][channels])
      const size_t pix = a.xt ? ((size_t)b * a.W + mx) * a.H + my : ((size_t)b * a.H + my) * a.W + mx;
      xv[k] = *reinterpret_cast<const f32x4*>(a.x + pix * a.ps + a.co + c4 * 4);
      q6[k] = (a.xt ? (size_t)mx * a.H + my : (size_t)my * a.W + mx) * a.C + c4 * 4;
    }
    if (p.planes2) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        mv[k] = *reinterpret_cast<const f32x4*>(a.mul + q6[k]);
        av[k] = *reinterpret_cast<const f32x4*>(a.add + q6[k]);
      }
    }
    // the stratum's (mean, rstd) and the channels' (gamma, beta) are reloaded only when the stratum changes along the quad (never, on a
    // transposed frame: its rows are the range positions)
    int s_cur = -1;
    float mean[4], rstd[4], ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int fx = 4 * xq - 1 + k;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      d[k] = z;
      d2[k] = z;
      if (fx < 0 || fx >= p.FW) continue;
      const int mx = p.transposed ? r : fx;
      const int s = mx / wps;
      if (s != s_cur) {
        s_cur = s;
        const float* smean = stat + ((size_t)b * a.strata + s) * a.cgroups * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c4 * 4 + e;
          mean[e] = smean[2 * (c / cpg)];
          rstd[e] = smean[2 * (c / cpg) + 1];
          ga[e] = a.gamma ? a.gamma[s * a.C + c] : 1.f;
          be[e] = a.beta ? a.beta[s * a.C + c] : 0.f;
        }
      }
      f32x4 v = xv[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = pn::apply_act((v[e] - mean[e]) * rstd[e] * ga[e] + be[e], a.act);      // (the expression of gn_apply_kernel: bit-identical values)
      d[k] = v;
      if (p.planes2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) d2[k][e] = v[e] * mv[k][e] + av[k][e];
      }
    }
    f32x4 v6[6];
    pn::wino4_input_transform4(d, v6);
    pn::wino4_store_planes(p.planes, v6, c4, c4n, p.plane_floats, b, r, xq, p.FH, p.Wq);
    if (p.planes2) {
      pn::wino4_input_transform4(d2, v6);
      pn::wino4_store_planes(p.planes2, v6, c4, c4n, p.plane_floats, b, r, xq, p.FH, p.Wq);
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
  return pn_groupnorm_strat_fwd_stat(x, batch, h, w, c, pixel_stride, channel_offset, channel_groups, range_strata, gamma, beta, eps, act, out,
                                     out_pixel_stride, out_channel_offset, mul, add, out2, nullptr, workspace, workspace_bytes, stream);
}

int pn_groupnorm_strat_fwd_stat(const float* x, int batch, int h, int w, int c, int pixel_stride, int channel_offset,
                                int channel_groups, int range_strata, const float* gamma, const float* beta, float eps, int act,
                                float* out, int out_pixel_stride, int out_channel_offset, const float* mul, const float* add,
                                float* out2, float* mean_rstd_out, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
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
  GnArgs a{};
  a.x = x; a.B = batch; a.H = h; a.W = w; a.C = c; a.ps = pixel_stride; a.co = channel_offset;
  a.cgroups = channel_groups; a.strata = range_strata;
  a.splits = pick_splits(batch, h, range_strata);
  if (a.splits > 256) a.splits = 256;
  a.rows_per_split = pn::cdiv(h, a.splits);
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.act = act;
  a.out = out; a.ops = out_pixel_stride; a.oco = out_channel_offset;
  a.mul = mul; a.add = add; a.out2 = out2; a.o2ps = c; a.o2co = 0;
  a.part = static_cast<double*>(workspace);
  const size_t ngroups = (size_t)batch * range_strata * channel_groups;
  a.stat = mean_rstd_out ? mean_rstd_out : reinterpret_cast<float*>(a.part + ngroups * 256 * 2);   // kept by the caller for the backward
  dim3 grid(a.splits, range_strata, batch);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)ngroups), dim3(64), 0, pn::S(stream), a);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  return pn::check_launch("groupnorm_strat");
}

/* pn_groupnorm_strat_fwd whose result leaves as F(4, 3) planes (pn_wino4_planes_floats(batch, h, w, c) floats each; transpose_hw 1: of the
 * transposed map, then sized (batch, w, h, c); transpose_hw 2: the same planes from a source that is STORED transposed -- x [b][w][h][c]
 * as pn_conv2d_wino4_nhwc_f32 writes it under pn_conv_desc.transpose_hw, mul / add [w][h][c] -- so that both the reads and the plane
 * stores run along contiguous memory); planes2 (nullable) = the planes of out * mul + add */
int pn_groupnorm_strat_planes_f32(const float* x, int batch, int h, int w, int c, int pixel_stride, int channel_offset, int channel_groups,
                                  int range_strata, const float* gamma, const float* beta, float eps, int act, const float* mul, const float* add,
                                  int transpose_hw, float* planes, float* planes2, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && planes && workspace, "groupnorm_planes: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && c >= 8 && c % 8 == 0 && c <= 4 * kThreads && (4 * kThreads) % c == 0, "groupnorm_planes: bad sizes");
  PN_REQUIRE(pixel_stride % 4 == 0 && channel_offset % 4 == 0, "groupnorm_planes: strides / offsets must be multiples of 4 floats");
  PN_REQUIRE(channel_groups >= 1 && c % channel_groups == 0 && channel_groups <= 128 && (channel_groups & (channel_groups - 1)) == 0,
             "groupnorm_planes: channel_groups must be a power of two <= 128 dividing the channel count");
  PN_REQUIRE(range_strata >= 1 && w % range_strata == 0, "groupnorm_planes: range axis not divisible by range_strata");
  PN_REQUIRE((planes2 == nullptr) || (mul && add), "groupnorm_planes: planes2 needs mul and add");
  PN_REQUIRE(transpose_hw >= 0 && transpose_hw <= 2, "groupnorm_planes: transpose_hw 0, 1 or 2");
  const int fh = transpose_hw ? w : h, fw = transpose_hw ? h : w;
  PN_REQUIRE(fw % 4 == 0, "groupnorm_planes: the Winograd axis must be a multiple of 4 pixels");
  if (workspace_bytes < pn_groupnorm_workspace_bytes(batch, channel_groups, range_strata))
    return pn::fail(PN_ERR_WORKSPACE, "groupnorm_planes: workspace too small");
  GnPlanesArgs p{};
  GnArgs& a = p.g;
  a.x = x; a.B = batch; a.H = h; a.W = w; a.C = c; a.ps = pixel_stride; a.co = channel_offset;
  a.cgroups = channel_groups; a.strata = range_strata;
  a.splits = pick_splits(batch, h, range_strata);
  if (a.splits > 256) a.splits = 256;
  a.rows_per_split = pn::cdiv(h, a.splits);
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.act = act;
  a.out = nullptr; a.ops = 0; a.oco = 0; a.mul = mul; a.add = add; a.out2 = nullptr; a.o2ps = 0; a.o2co = 0;
  a.part = static_cast<double*>(workspace);
  const size_t ngroups = (size_t)batch * range_strata * channel_groups;
  a.stat = reinterpret_cast<float*>(a.part + ngroups * 256 * 2);
  a.xt = transpose_hw == 2;
  p.planes = planes; p.planes2 = planes2; p.transposed = transpose_hw != 0; p.FH = fh; p.FW = fw; p.Wq = fw / 4;
  p.plane_floats = (unsigned)((size_t)batch * (fh + 2) * p.Wq * 4);
  p.fold = ngroups <= (size_t)kFoldGroups;
  dim3 grid(a.splits, range_strata, batch);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  if (!p.fold) hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)ngroups), dim3(64), 0, pn::S(stream), a);
  const long long total = (long long)batch * fh * p.Wq * (c / 4);
  hipLaunchKernelGGL(gn_apply_planes_kernel, dim3((unsigned)std::min<long long>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), p);
  return pn::check_launch("groupnorm_planes");
}

/* normalisation pass alone, with the group statistics already on the device (written by the producing convolution's
 * epilogue, pn_conv2d_multi_f32 stat_mean_rstd): one launch instead of statistics + finalize + apply */
int pn_groupnorm_apply_f32(const float* x, int batch, int h, int w, int c, int pixel_stride, int channel_offset, int channel_groups,
                           int range_strata, const float* mean_rstd, const float* gamma, const float* beta, int act, float* out,
                           int out_pixel_stride, int out_channel_offset, const float* mul, const float* add, float* out2,
                           int out2_pixel_stride, int out2_channel_offset, pn_stream_t stream) {
  PN_REQUIRE(x && out && mean_rstd, "groupnorm_apply: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && c >= 1, "groupnorm_apply: bad sizes");
  PN_REQUIRE(c % 4 == 0 && c <= 4 * kThreads && (4 * kThreads) % c == 0, "groupnorm_apply: channel count must be a multiple of 4 dividing 1024");
  PN_REQUIRE(pixel_stride % 4 == 0 && channel_offset % 4 == 0 && out_pixel_stride % 4 == 0 && out_channel_offset % 4 == 0,
             "groupnorm_apply: strides / offsets must be multiples of 4 floats");
  PN_REQUIRE(channel_groups >= 1 && c % channel_groups == 0, "groupnorm_apply: channel_groups must divide the channel count");
  PN_REQUIRE(range_strata >= 1 && w % range_strata == 0, "groupnorm_apply: range axis not divisible by range_strata");
  PN_REQUIRE((out2 == nullptr) || (mul && add && out2_pixel_stride % 4 == 0 && out2_channel_offset % 4 == 0), "groupnorm_apply: out2 needs mul, add and aligned strides");
  GnArgs a{};
  a.x = x; a.B = batch; a.H = h; a.W = w; a.C = c; a.ps = pixel_stride; a.co = channel_offset;
  a.cgroups = channel_groups; a.strata = range_strata;
  a.splits = pick_splits(batch, h, range_strata);
  if (a.splits > 256) a.splits = 256;
  a.rows_per_split = pn::cdiv(h, a.splits);
  a.gamma = gamma; a.beta = beta; a.eps = 0.f; a.act = act;
  a.out = out; a.ops = out_pixel_stride; a.oco = out_channel_offset;
  a.mul = mul; a.add = add; a.out2 = out2; a.o2ps = out2_pixel_stride; a.o2co = out2_channel_offset;
  a.part = nullptr;
  a.stat = const_cast<float*>(mean_rstd);
  dim3 grid(a.splits, range_strata, batch);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  return pn::check_launch("groupnorm_apply");
}

}  // extern "C"

// =================================================================================================
// BatchNorm2d in training mode (rpn.py:128-140 under model.train(): batch statistics over B,H,W,
// running-stat update with momentum 0.01, eps 1e-3) and its backward, fused with the ReLU that
// follows it everywhere in the reference.  Same two-pass, fixed-order scheme as above; the pixel
// range is cut into slices, fp64 partials, no atomics.
// =================================================================================================
namespace {

struct BnArgs {
  const float* x;      // conv output (pre-norm), NHWC slice
  long long pixels;
  int C, ps, co;
  int slices;
  long long per_slice;
  const float* gamma;
  const float* beta;
  float eps, momentum;
  int act;
  float* out;
  int ops, oco;
  float* running_mean;  // nullable
  float* running_var;
  float* stat;          // [C][2] (mean, rstd) saved for backward
  double* part;         // [slices][C][2]
  // backward
  const float* dout;
  int dps, dco;
  float* dx;
  int xps, xco;
  float* dgamma;
  float* dbeta;
  int accumulate;
  float* coef;          // [C][2]: mean(g), mean(g * xhat)
};

// slice partials of (sum, sum of squares) per channel.  mode 0: of x; mode 1 (backward): of
// g = dout * [act'(.)] and g * xhat
template <int MODE>
__global__ __launch_bounds__(kThreads) void bn_partial_kernel(BnArgs a) {
  __shared__ double red[2][kThreads * 4];
  const int vpc = a.C / 4;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc, ppb = kThreads / vpc;
  const long long p0 = blockIdx.x * a.per_slice, p1 = min(a.pixels, p0 + a.per_slice);
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  float mean[4], rstd[4], ga[4], be[4];
  if (MODE == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cv * 4 + k;
      mean[k] = a.stat[2 * c]; rstd[k] = a.stat[2 * c + 1];
      ga[k] = a.gamma ? a.gamma[c] : 1.f; be[k] = a.beta ? a.beta[c] : 0.f;
    }
  }
  if (threadIdx.x < vpc * ppb) {
    // four pixels per trip, their loads issued before the first is used: beside the F(4,3) weight gradient of the side stream (178 VGPRs,
    // eight waves per CU) a CU has room for one or two of these waves per SIMD, and with one 16-byte load in flight per lane the pass
    // ran at a third of its speed.  The sums keep their order (p, p + ppb, ...): same bits as the one-pixel loop.
    auto accum = [&](const f32x4& v, const f32x4& d) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += v[k]; s1[k] += (double)v[k] * v[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (v[k] - mean[k]) * rstd[k];
          const float y = xh * ga[k] + be[k];
          const float g = (a.act == PN_ACT_RELU && !(y > 0.f)) ? 0.f : d[k];
          s0[k] += g; s1[k] += (double)g * xh;
        }
      }
    };
    const float* xb = a.x + a.co + cv * 4;
    const float* db = MODE == 1 ? a.dout + a.dco + cv * 4 : nullptr;
    long long p = p0 + pl;
    for (; p + 3 * ppb < p1; p += 4 * ppb) {
      f32x4 v[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(xb + (p + u * ppb) * a.ps);
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = *reinterpret_cast<const f32x4*>(db + (p + u * ppb) * a.dps);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(v[u], d[u]);
    }
    for (; p < p1; p += ppb) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xb + p * a.ps);
      f32x4 d = v;
      if (MODE == 1) d = *reinterpret_cast<const f32x4*>(db + p * a.dps);
      accum(v, d);
    }
  }
  if (threadIdx.x < vpc * ppb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[0][(cv * 4 + k) * ppb + pl] = s0[k];
      red[1][(cv * 4 + k) * ppb + pl] = s1[k];
    }
  }
  __syncthreads();
  // one thread per (channel, moment): fixed-order sum over the ppb pixel lanes
  for (int i = threadIdx.x; i < 2 * a.C; i += kThreads) {
    const int m = i / a.C, c = i - m * a.C;
    double t = 0.0;
    for (int j = 0; j < ppb; ++j) t += red[m][c * ppb + j];
    a.part[((size_t)blockIdx.x * a.C + c) * 2 + m] = t;
  }
}

// one wave per channel: lane-strided partial sums, then the xor butterfly (same association on every run)
__global__ __launch_bounds__(64) void bn_finalize_kernel(BnArgs a) {
  const int c = blockIdx.x;
  double t0 = 0.0, t1 = 0.0;
  for (int k = threadIdx.x; k < a.slices; k += 64) { t0 += a.part[((size_t)k * a.C + c) * 2]; t1 += a.part[((size_t)k * a.C + c) * 2 + 1]; }
  t0 = pn::wave_sum(t0);
  t1 = pn::wave_sum(t1);
  if (threadIdx.x != 0) return;
  const double n = (double)a.pixels;
  const double mean = t0 / n;
  double var = t1 / n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  a.stat[2 * c] = (float)mean;
  a.stat[2 * c + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
  if (a.running_mean) {  // torch: running = (1 - m) * running + m * batch; the variance one is unbiased
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mean);
    a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unb);
  }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(BnArgs a) {
  const int c = blockIdx.x;
  double t0 = 0.0, t1 = 0.0;
  for (int k = threadIdx.x; k < a.slices; k += 64) { t0 += a.part[((size_t)k * a.C + c) * 2]; t1 += a.part[((size_t)k * a.C + c) * 2 + 1]; }
  t0 = pn::wave_sum(t0);
  t1 = pn::wave_sum(t1);
  if (threadIdx.x != 0) return;
  if (a.dbeta) a.dbeta[c] = (a.accumulate ? a.dbeta[c] : 0.f) + (float)t0;
  if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.f) + (float)t1;
  a.coef[2 * c] = (float)(t0 / (double)a.pixels);
  a.coef[2 * c + 1] = (float)(t1 / (double)a.pixels);
}

// MODE 0: out = act((x - mean) * rstd * gamma + beta);  MODE 1: dx = gamma * rstd * (g - mean(g) - xhat * mean(g xhat))
// A thread keeps ONE group of four channels for the whole launch (the grid stride is a multiple of the groups per pixel whenever that
// count divides 256: every channel count of the model), so its 16 / 24 per-channel constants are loaded once.  r2 re-read them from
// global memory for every element -- six dword loads per channel next to the two 16-byte loads that carry the data: the backward
// apply of a 256 x 256 x 128 x 4 layer took 197 us for 400 MB (2.0 TB/s); the pixel index came from a 64-bit division per element.
template <int MODE>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(BnArgs a) {
  const int vpc = a.C / 4;
  const long long total = a.pixels * vpc;
  const long long stride = (long long)gridDim.x * kThreads;
  const bool fixed = (kThreads % vpc) == 0;      // block-uniform
  long long i = blockIdx.x * (long long)kThreads + threadIdx.x;
  long long p = i / vpc;
  int cv = (int)(i - p * vpc);
  const long long pstep = stride / vpc;          // exact when `fixed`
  float mean[4], rstd[4], ga[4], be[4], c0[4], c1[4];
  auto load_consts = [&](int cvv) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cvv * 4 + k;
      mean[k] = a.stat[2 * c]; rstd[k] = a.stat[2 * c + 1];
      ga[k] = a.gamma ? a.gamma[c] : 1.f; be[k] = a.beta ? a.beta[c] : 0.f;
      if (MODE == 1) { c0[k] = a.coef[2 * c]; c1[k] = a.coef[2 * c + 1]; }
    }
  };
  if (i < total) load_consts(cv);
  auto one = [&](const f32x4& v, const f32x4& d, long long pp) {
    f32x4 o;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = pn::apply_act((v[k] - mean[k]) * rstd[k] * ga[k] + be[k], a.act);
      *reinterpret_cast<f32x4*>(a.out + pp * a.ops + a.oco + cv * 4) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (v[k] - mean[k]) * rstd[k];
        const float y = xh * ga[k] + be[k];
        const float g = (a.act == PN_ACT_RELU && !(y > 0.f)) ? 0.f : d[k];
        o[k] = ga[k] * rstd[k] * (g - c0[k] - xh * c1[k]);
      }
      *reinterpret_cast<f32x4*>(a.dx + pp * a.xps + a.xco + cv * 4) = o;
    }
  };
  if (fixed) {
    // four elements per trip with their loads in flight together (see bn_partial_kernel)
    for (; i + 3 * stride < total; i += 4 * stride) {
      f32x4 v[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(a.x + (p + u * pstep) * a.ps + a.co + cv * 4);
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = *reinterpret_cast<const f32x4*>(a.dout + (p + u * pstep) * a.dps + a.dco + cv * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(v[u], d[u], p + u * pstep);
      p += 4 * pstep;
    }
  }
  for (; i < total; i += stride) {
    if (!fixed) {
      p = i / vpc;
      cv = (int)(i - p * vpc);
      load_consts(cv);
    }
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + p * a.ps + a.co + cv * 4);
    f32x4 d = v;
    if (MODE == 1) d = *reinterpret_cast<const f32x4*>(a.dout + p * a.dps + a.dco + cv * 4);
    one(v, d, p);
    p += pstep;
  }
}

constexpr int kBnSlices = 1024;
#ifndef PN_BN_EXP
#define PN_BN_EXP 0      // diagnostic builds (tools/bnq.sh): 1 no forward statistics pass, 2 no backward statistics pass -- wrong results; the iteration
#endif                   // time then bounds what fusing that pass into the producing convolution's epilogue could save

int bn_common(BnArgs& a, long long pixels, int c, void* workspace, size_t workspace_bytes) {
  PN_REQUIRE(pixels >= 1 && c >= 4 && c % 4 == 0 && c <= 4 * kThreads, "batchnorm: channel count must be a multiple of 4, at most 1024");
  PN_REQUIRE(workspace != nullptr, "batchnorm: null workspace");
  if (workspace_bytes < pn_batchnorm_workspace_bytes(c)) return pn::fail(PN_ERR_WORKSPACE, "batchnorm: workspace too small");
  a.pixels = pixels; a.C = c;
  a.slices = (int)std::min<long long>(kBnSlices, pn::cdiv(pixels, 64));
  a.per_slice = (pixels + a.slices - 1) / a.slices;
  a.slices = (int)((pixels + a.per_slice - 1) / a.per_slice);
  a.part = static_cast<double*>(workspace);
  a.coef = reinterpret_cast<float*>(a.part + (size_t)kBnSlices * c * 2);
  return PN_OK;
}

}  // namespace

extern "C" {

size_t pn_batchnorm_workspace_bytes(int c) { return (size_t)kBnSlices * c * 2 * sizeof(double) + (size_t)c * 2 * sizeof(float); }

int pn_batchnorm_train_fwd(const float* x, long long pixels, int c, int pixel_stride, int channel_offset, const float* gamma,
                           const float* beta, float eps, float momentum, int act, float* running_mean, float* running_var,
                           float* out, int out_pixel_stride, int out_channel_offset, float* saved_stat, void* workspace,
                           size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && out && saved_stat, "batchnorm_train_fwd: null pointer");
  PN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "batchnorm_train_fwd: running_mean and running_var go together");
  PN_REQUIRE(pixel_stride % 4 == 0 && channel_offset % 4 == 0 && out_pixel_stride % 4 == 0 && out_channel_offset % 4 == 0,
             "batchnorm: strides / offsets must be multiples of 4 floats");
  BnArgs a{};
  if (int rc = bn_common(a, pixels, c, workspace, workspace_bytes)) return rc;
  a.x = x; a.ps = pixel_stride; a.co = channel_offset; a.gamma = gamma; a.beta = beta; a.eps = eps; a.momentum = momentum; a.act = act;
  a.out = out; a.ops = out_pixel_stride; a.oco = out_channel_offset; a.running_mean = running_mean; a.running_var = running_var;
  a.stat = saved_stat;
  hipStream_t st = pn::S(stream);
  if (!(PN_BN_EXP & 1)) hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(a.slices), dim3(kThreads), 0, st, a);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(64), 0, st, a);
  const long long total = pixels * (c / 4);
  hipLaunchKernelGGL(bn_apply_kernel<0>, dim3((unsigned)std::min<long long>(4096, pn::cdiv(total, kThreads))), dim3(kThreads), 0, st, a);
  return pn::check_launch("batchnorm_train_fwd");
}

int pn_batchnorm_bwd(const float* x, const float* dout, long long pixels, int c, int pixel_stride, int channel_offset,
                     int dout_pixel_stride, int dout_channel_offset, const float* gamma, const float* beta, int act,
                     const float* saved_stat, float* dx, int dx_pixel_stride, int dx_channel_offset, float* dgamma, float* dbeta,
                     int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && dout && saved_stat && dx, "batchnorm_bwd: null pointer");
  PN_REQUIRE(pixel_stride % 4 == 0 && channel_offset % 4 == 0 && dout_pixel_stride % 4 == 0 && dout_channel_offset % 4 == 0 &&
                 dx_pixel_stride % 4 == 0 && dx_channel_offset % 4 == 0, "batchnorm: strides / offsets must be multiples of 4 floats");
  BnArgs a{};
  if (int rc = bn_common(a, pixels, c, workspace, workspace_bytes)) return rc;
  a.x = x; a.ps = pixel_stride; a.co = channel_offset; a.gamma = gamma; a.beta = beta; a.act = act;
  a.stat = const_cast<float*>(saved_stat);
  a.dout = dout; a.dps = dout_pixel_stride; a.dco = dout_channel_offset;
  a.dx = dx; a.xps = dx_pixel_stride; a.xco = dx_channel_offset; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
  hipStream_t st = pn::S(stream);
  if (!(PN_BN_EXP & 2)) hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(a.slices), dim3(kThreads), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(64), 0, st, a);
  const long long total = pixels * (c / 4);
  hipLaunchKernelGGL(bn_apply_kernel<1>, dim3((unsigned)std::min<long long>(4096, pn::cdiv(total, kThreads))), dim3(kThreads), 0, st, a);
  return pn::check_launch("batchnorm_bwd");
}

}  // extern "C"

// =================================================================================================
// Backward of the GroupNorm family (+ ReLU, + the x*W(pos)+b(pos) calibration output).
//   do    = dout + dout2 * mul                         (gradient reaching out = act(y))
//   g     = do * act'(y),  y = xhat * gamma + beta
//   dgamma[s][c] = sum_{b, pixels of stratum s} g * xhat,   dbeta[s][c] = sum g
//   dxhat = g * gamma;  per statistics group: m1 = mean(dxhat), m2 = mean(dxhat * xhat)
//   dx    = rstd * (dxhat - m1 - xhat * m2)
//   dmul[pixel][c] = sum_b dout2 * out,  dadd[pixel][c] = sum_b dout2
// The statistics are recomputed from x (two cheap passes over a few MB) instead of being carried
// from the forward call.
// =================================================================================================
namespace {

struct GnBwdArgs {
  GnArgs f;            // forward description (x, sizes, gamma/beta, eps, act, mul; part/stat)
  const float* dout;
  int dps, dco;
  const float* dout2;  // nullable, pixel stride C
  float* dx;
  int xps, xco;
  float* dgamma;
  float* dbeta;
  float* dmul;
  float* dadd;
  int accumulate;
  double* part2;       // [B][strata][splits][C][2]
  float* coef;         // [B][strata][cgroups][2] = (m1, m2)
  double* chan;        // [B][strata][C][2] per-sample channel sums (dbeta, dgamma contributions)
};

__device__ __forceinline__ void gn_point(const GnBwdArgs& a, const float* smean, int s, int cv, size_t pix, int y, int x,
                                         float (&g)[4], float (&xh)[4], float (&ga)[4]) {
  const GnArgs& f = a.f;
  const int cpg = f.C / f.cgroups;
  const f32x4 v = *reinterpret_cast<const f32x4*>(f.x + pix * f.ps + f.co + cv * 4);
  f32x4 d = *reinterpret_cast<const f32x4*>(a.dout + pix * a.dps + a.dco + cv * 4);
  if (a.dout2) {
    const f32x4 d2 = *reinterpret_cast<const f32x4*>(a.dout2 + pix * f.C + cv * 4);
    const f32x4 m = *reinterpret_cast<const f32x4*>(f.mul + ((size_t)y * f.W + x) * f.C + cv * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] += d2[k] * m[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cv * 4 + k;
    ga[k] = f.gamma ? f.gamma[s * f.C + c] : 1.f;
    const float be = f.beta ? f.beta[s * f.C + c] : 0.f;
    xh[k] = (v[k] - smean[2 * (c / cpg)]) * smean[2 * (c / cpg) + 1];
    const float yv = xh[k] * ga[k] + be;
    g[k] = (f.act == PN_ACT_RELU && !(yv > 0.f)) ? 0.f : d[k];
  }
}

// grid (splits, strata, B): per-channel (sum g, sum g*xhat) of this block's rows
__global__ __launch_bounds__(kThreads) void gn_bwd_partial_kernel(GnBwdArgs a) {
  __shared__ double red[2][kThreads * 4];
  const GnArgs& f = a.f;
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int wps = f.W / f.strata;
  const float* smean = f.stat + ((size_t)b * f.strata + s) * f.cgroups * 2;
  const int vpc = f.C / 4;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc, ppb = kThreads / vpc;
  const int y0 = split * f.rows_per_split, y1 = min(f.H, y0 + f.rows_per_split);
  const int npix = (y1 - y0) * wps;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wps, x = s * wps + p % wps;
    float g[4], xh[4], ga[4];
    gn_point(a, smean, s, cv, (size_t)(b * f.H + y) * f.W + x, y, x, g, xh, ga);
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0[k] += g[k]; s1[k] += (double)g[k] * xh[k]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][(cv * 4 + k) * ppb + pl] = s0[k];
    red[1][(cv * 4 + k) * ppb + pl] = s1[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * f.C; i += kThreads) {
    const int m = i / f.C, c = i - m * f.C;
    double t = 0.0;
    for (int j = 0; j < ppb; ++j) t += red[m][c * ppb + j];
    a.part2[((((size_t)b * f.strata + s) * f.splits + split) * f.C + c) * 2 + m] = t;
  }
}

// block per (stratum s, sample b): per-channel sums over the row splits (256 / C threads per channel, fixed order), the
// per-(b, group) coefficients m1, m2, and the channel sums of this sample for gn_bwd_param_kernel
__global__ __launch_bounds__(kThreads) void gn_bwd_finalize_kernel(GnBwdArgs a) {
  extern __shared__ double sh[];  // [256][2] partial sums, then [C][2] gamma-weighted per-channel sums
  const GnArgs& f = a.f;
  const int s = blockIdx.x, b = blockIdx.y;
  const int cpg = f.C / f.cgroups;
  const double n = (double)cpg * f.H * (f.W / f.strata);
  const int nparts = kThreads / f.C, c = threadIdx.x % f.C, part = threadIdx.x / f.C;
  double t0 = 0.0, t1 = 0.0;
  if (part < nparts) {
    const double* p = a.part2 + (((size_t)b * f.strata + s) * f.splits) * f.C * 2 + (size_t)c * 2;
    for (int k = part; k < f.splits; k += nparts) { t0 += p[(size_t)k * f.C * 2]; t1 += p[(size_t)k * f.C * 2 + 1]; }
  }
  sh[2 * threadIdx.x] = t0;
  sh[2 * threadIdx.x + 1] = t1;
  __syncthreads();
  if (part == 0) {
    for (int q = 1; q < nparts; ++q) { t0 += sh[2 * (q * f.C + c)]; t1 += sh[2 * (q * f.C + c) + 1]; }
    double* cs = a.chan + (((size_t)b * f.strata + s) * f.C + c) * 2;
    cs[0] = t0;
    cs[1] = t1;
  }
  __syncthreads();
  if (part == 0) {
    const double ga = f.gamma ? (double)f.gamma[s * f.C + c] : 1.0;
    sh[2 * c] = ga * t0;
    sh[2 * c + 1] = ga * t1;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < f.cgroups; g += blockDim.x) {
    double m1 = 0.0, m2 = 0.0;
    for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) { m1 += sh[2 * cc]; m2 += sh[2 * cc + 1]; }
    float* co = a.coef + (((size_t)b * f.strata + s) * f.cgroups + g) * 2;
    co[0] = (float)(m1 / n);
    co[1] = (float)(m2 / n);
  }
}

// dgamma / dbeta: thread per (stratum, channel), samples added in order
__global__ void gn_bwd_param_kernel(GnBwdArgs a) {
  const GnArgs& f = a.f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.strata * f.C) return;
  const int s = i / f.C, c = i - s * f.C;
  double tb = 0.0, tg = 0.0;
  for (int b = 0; b < f.B; ++b) {
    const double* cs = a.chan + (((size_t)b * f.strata + s) * f.C + c) * 2;
    tb += cs[0];
    tg += cs[1];
  }
  if (a.dbeta) a.dbeta[i] = (a.accumulate ? a.dbeta[i] : 0.f) + (float)tb;
  if (a.dgamma) a.dgamma[i] = (a.accumulate ? a.dgamma[i] : 0.f) + (float)tg;
}

__global__ __launch_bounds__(kThreads) void gn_bwd_apply_kernel(GnBwdArgs a) {
  const GnArgs& f = a.f;
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int wps = f.W / f.strata;
  const int cpg = f.C / f.cgroups;
  const float* smean = f.stat + ((size_t)b * f.strata + s) * f.cgroups * 2;
  const float* coef = a.coef + ((size_t)b * f.strata + s) * f.cgroups * 2;
  const int vpc = f.C / 4;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc, ppb = kThreads / vpc;
  const int y0 = split * f.rows_per_split, y1 = min(f.H, y0 + f.rows_per_split);
  const int npix = (y1 - y0) * wps;
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wps, x = s * wps + p % wps;
    const size_t pix = (size_t)(b * f.H + y) * f.W + x;
    float g[4], xh[4], ga[4];
    gn_point(a, smean, s, cv, pix, y, x, g, xh, ga);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int grp = (cv * 4 + k) / cpg;
      o[k] = smean[2 * grp + 1] * (g[k] * ga[k] - coef[2 * grp] - xh[k] * coef[2 * grp + 1]);
    }
    *reinterpret_cast<f32x4*>(a.dx + pix * a.xps + a.xco + cv * 4) = o;
  }
}

// dmul / dadd: thread per (pixel of one sample, 4 channels), loop over the batch
__global__ void gn_bwd_calib_kernel(GnBwdArgs a) {
  const GnArgs& f = a.f;
  const int vpc = f.C / 4;
  const size_t total = (size_t)f.H * f.W * vpc;
  const int cpg = f.C / f.cgroups, wps = f.W / f.strata;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pq = i / vpc;
    const int cv = (int)(i - pq * vpc);
    const int y = (int)(pq / f.W), x = (int)(pq - (size_t)y * f.W), s = x / wps;
    f32x4 sm = {0.f, 0.f, 0.f, 0.f}, sa = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < f.B; ++b) {
      const size_t pix = (size_t)(b * f.H + y) * f.W + x;
      const float* smean = f.stat + ((size_t)b * f.strata + s) * f.cgroups * 2;
      const f32x4 v = *reinterpret_cast<const f32x4*>(f.x + pix * f.ps + f.co + cv * 4);
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(a.dout2 + pix * f.C + cv * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = cv * 4 + k;
        const float xh = (v[k] - smean[2 * (c / cpg)]) * smean[2 * (c / cpg) + 1];
        const float o = pn::apply_act(xh * (f.gamma ? f.gamma[s * f.C + c] : 1.f) + (f.beta ? f.beta[s * f.C + c] : 0.f), f.act);
        sm[k] += d2[k] * o;
        sa[k] += d2[k];
      }
    }
    float* pm = a.dmul + pq * f.C + cv * 4;
    float* pa = a.dadd + pq * f.C + cv * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pm[k] = (a.accumulate ? pm[k] : 0.f) + sm[k];
      pa[k] = (a.accumulate ? pa[k] : 0.f) + sa[k];
    }
  }
}

}  // namespace

extern "C" {

size_t pn_groupnorm_bwd_workspace_bytes(int batch, int c, int channel_groups, int range_strata) {
  return pn_groupnorm_workspace_bytes(batch, channel_groups, range_strata) +
         (size_t)batch * range_strata * 256 * c * 2 * sizeof(double) + (size_t)batch * range_strata * c * 2 * sizeof(double) +
         (size_t)batch * range_strata * channel_groups * 2 * sizeof(float);
}

int pn_groupnorm_strat_bwd(const float* x, const float* dout, const float* dout2, const float* mul, int batch, int h, int w, int c,
                           int pixel_stride, int channel_offset, int dout_pixel_stride, int dout_channel_offset,
                           int channel_groups, int range_strata, const float* gamma, const float* beta, float eps, int act,
                           float* dx, int dx_pixel_stride, int dx_channel_offset, float* dgamma, float* dbeta, float* dmul,
                           float* dadd, int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  return pn_groupnorm_strat_bwd_stat(x, dout, dout2, mul, batch, h, w, c, pixel_stride, channel_offset, dout_pixel_stride, dout_channel_offset,
                                     channel_groups, range_strata, gamma, beta, eps, act, dx, dx_pixel_stride, dx_channel_offset, dgamma, dbeta,
                                     dmul, dadd, accumulate, nullptr, workspace, workspace_bytes, stream);
}

int pn_groupnorm_strat_bwd_stat(const float* x, const float* dout, const float* dout2, const float* mul, int batch, int h, int w, int c,
                                int pixel_stride, int channel_offset, int dout_pixel_stride, int dout_channel_offset,
                                int channel_groups, int range_strata, const float* gamma, const float* beta, float eps, int act,
                                float* dx, int dx_pixel_stride, int dx_channel_offset, float* dgamma, float* dbeta, float* dmul,
                                float* dadd, int accumulate, const float* mean_rstd, void* workspace, size_t workspace_bytes,
                                pn_stream_t stream) {
  PN_REQUIRE(x && dout && dx && workspace, "groupnorm_bwd: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && c >= 1, "groupnorm_bwd: bad sizes");
  PN_REQUIRE(c % 4 == 0 && c <= kThreads && (4 * kThreads) % c == 0, "groupnorm_bwd: channel count must be a multiple of 4 dividing 1024, at most 256");
  PN_REQUIRE(pixel_stride % 4 == 0 && channel_offset % 4 == 0 && dout_pixel_stride % 4 == 0 && dout_channel_offset % 4 == 0 &&
                 dx_pixel_stride % 4 == 0 && dx_channel_offset % 4 == 0, "groupnorm_bwd: strides / offsets must be multiples of 4 floats");
  PN_REQUIRE(channel_groups >= 1 && c % channel_groups == 0 && channel_groups <= 128 && (channel_groups & (channel_groups - 1)) == 0,
             "groupnorm_bwd: channel_groups must be a power of two <= 128 dividing the channel count");
  PN_REQUIRE(range_strata >= 1 && w % range_strata == 0, "groupnorm_bwd: range axis not divisible by range_strata");
  PN_REQUIRE(act == PN_ACT_NONE || act == PN_ACT_RELU, "groupnorm_bwd: activation must be none or ReLU");
  PN_REQUIRE((dout2 == nullptr) || (mul && dmul && dadd), "groupnorm_bwd: dout2 needs mul, dmul and dadd");
  if (workspace_bytes < pn_groupnorm_bwd_workspace_bytes(batch, c, channel_groups, range_strata))
    return pn::fail(PN_ERR_WORKSPACE, "groupnorm_bwd: workspace too small");
  GnBwdArgs a{};
  GnArgs& f = a.f;
  f.x = x; f.B = batch; f.H = h; f.W = w; f.C = c; f.ps = pixel_stride; f.co = channel_offset;
  f.cgroups = channel_groups; f.strata = range_strata;
  f.splits = std::min(256, pick_splits(batch, h, range_strata));
  f.rows_per_split = pn::cdiv(h, f.splits);
  f.gamma = gamma; f.beta = beta; f.eps = eps; f.act = act; f.mul = mul;
  f.part = static_cast<double*>(workspace);
  const size_t ngroups = (size_t)batch * range_strata * channel_groups;
  f.stat = mean_rstd ? const_cast<float*>(mean_rstd) : reinterpret_cast<float*>(f.part + ngroups * 256 * 2);
  char* p = static_cast<char*>(workspace) + pn_groupnorm_workspace_bytes(batch, channel_groups, range_strata);
  a.part2 = reinterpret_cast<double*>(p);
  a.chan = reinterpret_cast<double*>(p + (size_t)batch * range_strata * 256 * c * 2 * sizeof(double));
  a.coef = reinterpret_cast<float*>(a.chan + (size_t)batch * range_strata * c * 2);
  a.dout = dout; a.dps = dout_pixel_stride; a.dco = dout_channel_offset; a.dout2 = dout2;
  a.dx = dx; a.xps = dx_pixel_stride; a.xco = dx_channel_offset;
  a.dgamma = dgamma; a.dbeta = dbeta; a.dmul = dmul; a.dadd = dadd; a.accumulate = accumulate;
  hipStream_t st = pn::S(stream);
  dim3 grid(f.splits, range_strata, batch);
  if (!mean_rstd) {   // the forward's statistics were not kept: two more passes
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(kThreads), 0, st, f);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)ngroups), dim3(64), 0, st, f);
  }
  hipLaunchKernelGGL(gn_bwd_partial_kernel, grid, dim3(kThreads), 0, st, a);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(range_strata, batch), dim3(kThreads), (size_t)kThreads * 2 * sizeof(double), st, a);
  if (dgamma || dbeta)
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3(pn::cdiv(range_strata * c, 256)), dim3(256), 0, st, a);
  if (dout2) {
    const size_t total = (size_t)h * w * (c / 4);
    hipLaunchKernelGGL(gn_bwd_calib_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, dim3(kThreads), 0, st, a);  // last: dx may alias dout
  return pn::check_launch("groupnorm_strat_bwd");
}

}  // extern "C"
