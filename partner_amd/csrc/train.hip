// Small kernels of the training step that are neither convolutions nor norms:
// gradient norm + clipping coefficient, the fused decoupled-weight-decay Adam update over the flat
// parameter buffer, tanh backward, the dense expansion that turns the RangeStratified gradient into
// an ordinary convolution gradient, and a plain element-wise add.
// Reference: clip_grad_norm_            det3d/torchie/trainer/hooks/optimizer.py:10-13
//            OptimWrapper.step          det3d/solver/fastai_optim.py:155-171
//            torch.optim.Adam           det3d/torchie/apis/train.py:198-215
#include "pn_common.h"
#include <algorithm>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kNormParts = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ part) {
  __shared__ double red[256];
  // contiguous chunk per block: the association order depends only on n
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t i0 = blockIdx.x * per, i1 = std::min(n, i0 + per);
  double acc = 0.0;
  for (size_t i = i0 + threadIdx.x; i < i1; i += 256) acc += (double)g[i] * g[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void norm_final_kernel(const double* __restrict__ part, int nparts, float* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sqrt(red[0]);
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  size_t n;
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;
  const float* total_norm;  // device scalar, nullable
  float max_norm;
};

__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a) {
  float coef = 1.f;
  if (a.total_norm) {
    const float c = a.max_norm / (a.total_norm[0] + 1e-6f);
    coef = c < 1.f ? c : 1.f;
  }
  const float decay = 1.f - a.wd * a.lr;
  const float step = a.lr / a.bc1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = a.g[i] * coef;
    float p = a.p[i] * decay;
    // exp_avg.lerp_(grad, 1 - beta1): m + (g - m) * (1 - beta1)
    const float m = a.m[i] + (g - a.m[i]) * (1.f - a.beta1);
    const float v = a.v[i] * a.beta2 + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= step * (m / denom);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}

__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

// dy (B,H,W,C) -> out (B,H,W,strata*C): block s(x) of pixel (.., x) holds dy, the rest zeros
__global__ void strat_expand_kernel(const float* __restrict__ dy, int W, int C, int strata, float* __restrict__ out, size_t total4) {
  const int wps = W / strata, vpc = C / 4, ovpc = strata * vpc;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / ovpc;
    const int cv = (int)(i - pix * ovpc);
    const int s = (int)(pix % W) / wps;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (cv / vpc == s) v = *reinterpret_cast<const f32x4*>(dy + pix * C + (cv - s * vpc) * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// data gradient of the range-stratified 3x3 convolution, second half.  z (B,H,W,3*C) holds, per width tap kx, the column convolution of
// dy with the weights of the dy pixel's OWN stratum (a stratified 3x1 convolution, conv_mfma.hip MODE_STRAT):
// dx[y, x] = z[y, x+1, 0:C] + z[y, x, C:2C] + z[y, x-1, 2C:3C], columns outside the map contributing zero
__global__ void strat_dgrad_combine_kernel(const float* __restrict__ z, int W, int C, float* __restrict__ dx, int dx_ps, int dx_co, int accumulate,
                                           size_t total4) {
  const int vpc = C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / vpc;
    const int c = (int)(i - pix * vpc) * 4;
    const int x = (int)(pix % W);
    const float* zp = z + pix * (size_t)(3 * C) + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(zp + C);
    if (x + 1 < W) v += *reinterpret_cast<const f32x4*>(zp + 3 * C);
    if (x > 0) v += *reinterpret_cast<const f32x4*>(zp - 3 * C + 2 * C);
    float* o = dx + pix * (size_t)dx_ps + dx_co + c;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

inline unsigned grid_for(size_t n) { return (unsigned)std::min<size_t>(4096, (n + 255) / 256); }

}  // namespace

extern "C" {

size_t pn_grad_norm_workspace_bytes(void) { return kNormParts * sizeof(double); }

int pn_grad_norm_f32(const float* grads, size_t n, float* total_norm, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(grads && total_norm && workspace && n > 0, "grad_norm: bad arguments");
  PN_REQUIRE(workspace_bytes >= pn_grad_norm_workspace_bytes(), "grad_norm: workspace too small");
  const int parts = (int)std::min<size_t>(kNormParts, (n + 255) / 256);
  double* part = static_cast<double*>(workspace);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(parts), dim3(256), 0, pn::S(stream), grads, n, part);
  hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, pn::S(stream), part, parts, total_norm);
  return pn::check_launch("grad_norm");
}

int pn_adam_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, int step, float lr, float beta1,
                     float beta2, float eps, float weight_decay, const float* total_norm, float max_norm, pn_stream_t stream) {
  PN_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
  AdamArgs a{params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
             (float)(1.0 - pow((double)beta1, step)), (float)sqrt(1.0 - pow((double)beta2, step)), total_norm, max_norm};
  hipLaunchKernelGGL(adam_step_kernel, dim3(grid_for(n)), dim3(256), 0, pn::S(stream), a);
  return pn::check_launch("adam_step_kernel");
}

int pn_tanh_bwd_f32(const float* y, const float* dy, float* dx, size_t n, pn_stream_t stream) {
  PN_REQUIRE(y && dy && dx, "tanh_bwd: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, pn::S(stream), y, dy, dx, n);
  return pn::check_launch("tanh_bwd_kernel");
}

int pn_relu_bwd_f32(const float* y, const float* dy, float* dx, size_t n, pn_stream_t stream) {
  PN_REQUIRE(y && dy && dx, "relu_bwd: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, pn::S(stream), y, dy, dx, n);
  return pn::check_launch("relu_bwd_kernel");
}

int pn_add_f32(const float* a, const float* b, float* out, size_t n, pn_stream_t stream) {
  PN_REQUIRE(a && b && out, "add: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, pn::S(stream), a, b, out, n);
  return pn::check_launch("add_kernel");
}

int pn_strat_expand_f32(const float* dy, int batch, int h, int w, int c, int strata, float* out, pn_stream_t stream) {
  PN_REQUIRE(dy && out && batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && strata >= 1 && w % strata == 0, "strat_expand: bad arguments");
  const size_t total4 = (size_t)batch * h * w * strata * (c / 4);
  hipLaunchKernelGGL(strat_expand_kernel, dim3(grid_for(total4)), dim3(256), 0, pn::S(stream), dy, w, c, strata, out, total4);
  return pn::check_launch("strat_expand_kernel");
}

int pn_strat_dgrad_combine_f32(const float* z, int batch, int h, int w, int c, float* dx, int dx_pixel_stride, int dx_channel_offset, int accumulate,
                               pn_stream_t stream) {
  PN_REQUIRE(z && dx && batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && dx_pixel_stride % 4 == 0 && dx_channel_offset % 4 == 0 &&
             dx_pixel_stride >= dx_channel_offset + c, "strat_dgrad_combine: bad arguments");
  const size_t total4 = (size_t)batch * h * w * (c / 4);
  hipLaunchKernelGGL(strat_dgrad_combine_kernel, dim3(grid_for(total4)), dim3(256), 0, pn::S(stream), z, w, c, dx, dx_pixel_stride, dx_channel_offset,
                     accumulate, total4);
  return pn::check_launch("strat_dgrad_combine_kernel");
}

}  // extern "C"
