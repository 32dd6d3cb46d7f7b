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
#include "pn_common.h"

namespace {

constexpr int kThreads = 256;

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
};

// grid: (splits, strata, B)
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(GnArgs a) {
  __shared__ double ssum[kThreads], ssq[kThreads];
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int wps = a.W / a.strata;
  const int c = threadIdx.x % a.C, pl = threadIdx.x / a.C, ppb = kThreads / a.C;
  const int y0 = split * a.rows_per_split, y1 = min(a.H, y0 + a.rows_per_split);
  double sum = 0.0, sq = 0.0;
  const int npix = (y1 - y0) * wps;
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wps, x = s * wps + p % wps;
    const float v = a.x[((size_t)(b * a.H + y) * a.W + x) * a.ps + a.co + c];
    sum += v;
    sq += (double)v * v;
  }
  ssum[threadIdx.x] = sum;
  ssq[threadIdx.x] = sq;
  __syncthreads();
  const int cpg = a.C / a.cgroups;
  if (threadIdx.x < a.cgroups) {
    const int g = threadIdx.x;
    double t0 = 0.0, t1 = 0.0;
    for (int t = 0; t < kThreads; ++t)
      if ((t % a.C) / cpg == g) {
        t0 += ssum[t];
        t1 += ssq[t];
      }
    double* o = a.part + ((((size_t)b * a.strata + s) * a.cgroups + g) * a.splits + split) * 2;
    o[0] = t0;
    o[1] = t1;
  }
}

__global__ __launch_bounds__(kThreads) void gn_apply_kernel(GnArgs a) {
  __shared__ float smean[kThreads], srstd[kThreads];
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int wps = a.W / a.strata;
  const int cpg = a.C / a.cgroups;
  if (threadIdx.x < a.cgroups) {
    const double* o = a.part + (((size_t)b * a.strata + s) * a.cgroups + threadIdx.x) * a.splits * 2;
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < a.splits; ++k) {
      t0 += o[2 * k];
      t1 += o[2 * k + 1];
    }
    const double n = (double)cpg * a.H * wps;
    const double mean = t0 / n;
    double var = t1 / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    smean[threadIdx.x] = (float)mean;
    srstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  const int c = threadIdx.x % a.C, pl = threadIdx.x / a.C, ppb = kThreads / a.C;
  const float mean = smean[c / cpg], rstd = srstd[c / cpg];
  const float ga = a.gamma ? a.gamma[s * a.C + c] : 1.f, be = a.beta ? a.beta[s * a.C + c] : 0.f;
  const int y0 = split * a.rows_per_split, y1 = min(a.H, y0 + a.rows_per_split);
  const int npix = (y1 - y0) * wps;
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wps, x = s * wps + p % wps;
    const size_t pix = (size_t)(b * a.H + y) * a.W + x;
    float v = a.x[pix * a.ps + a.co + c];
    v = pn::apply_act((v - mean) * rstd * ga + be, a.act);
    a.out[pix * a.ops + a.oco + c] = v;
    if (a.out2) {
      const size_t q = ((size_t)y * a.W + x) * a.C + c;
      a.out2[pix * a.C + c] = v * a.mul[q] + a.add[q];
    }
  }
}

int pick_splits(int B, int H, int strata) {
  int splits = 1;
  while (splits < H && (long long)B * strata * splits < 512 && H / (splits * 2) >= 1) splits *= 2;
  return splits;
}

}  // namespace

extern "C" {

size_t pn_groupnorm_workspace_bytes(int batch, int channel_groups, int range_strata) {
  return (size_t)batch * range_strata * channel_groups * 256 * 2 * sizeof(double);
}

int pn_groupnorm_strat_fwd(const float* x, int batch, int h, int w, int c, int pixel_stride, int channel_offset,
                           int channel_groups, int range_strata, const float* gamma, const float* beta, float eps, int act,
                           float* out, int out_pixel_stride, int out_channel_offset, const float* mul, const float* add,
                           float* out2, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && out && workspace, "groupnorm: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && c >= 1, "groupnorm: bad sizes");
  PN_REQUIRE(c <= kThreads && kThreads % c == 0, "groupnorm: channel count must divide 256");
  PN_REQUIRE(channel_groups >= 1 && c % channel_groups == 0, "groupnorm: channels not divisible by channel_groups");
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
  dim3 grid(a.splits, range_strata, batch);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(kThreads), 0, pn::S(stream), a);
  return pn::check_launch("groupnorm_strat");
}

}  // extern "C"
