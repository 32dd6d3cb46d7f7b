// Backward side of the sparse 3-D convolutions of SpMiddleResNetFHD (det3d/models/backbones/scn.py:97-192; in the reference the
// arithmetic and its autograd are spconv's SubMConv3d / SparseConv3d -- third party, absent from the tree: PARITY UNPINNED, the
// checker is fp64 autograd over the dense-with-masks restatement oracle/polar_oracle.py::sp_middle_resnet_fhd).
//
// With out[i] = sum_t W_t in[nbr[i][t]]  (nbr = the forward's neighbour table, -1 = inactive tap):
//   data gradient   din[j]  = sum_{(i,t): nbr[i][t] = j} W_t^T dout[i].  For a fixed (j, t) at most one output site i reads input j
//                   through tap t, so the transposed table inv[j][t] = i is a plain scatter of nbr (pn_sparse_neighbors_transpose) and
//                   the data gradient is the SAME gathered MFMA GEMM as the forward (pn_sparse_conv_f32) over inv with the
//                   (Cin, Cout)-transposed weights -- no atomics, deterministic.
//   weight gradient dW_t[co][ci] = sum_i dout[i][co] * in[nbr[i][t]][ci]: pn_sparse_conv_wgrad_f32 materialises the gathered im2col
//                   matrix G (rows x taps*Cin) and runs ONE dense GEMM dW = dout^T G on the MFMA wgrad kernel of conv_bwd.hip (split over
//                   rows, partials reduced in split order: fixed association, no atomics).  A first VALU version ran at ~5 TFLOP/s and
//                   was a quarter of the PARTNER training iteration; G costs rows*taps*Cin*8 bytes of HBM traffic per convolution.
//   densify         pn_sparse_from_dense_nhwc gathers the gradient of SparseConvTensor.dense() back to the active rows.
#include "pn_common.h"
#include <algorithm>

namespace {

__global__ void transpose_nbr_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ n_out, int out_cap, int taps, int in_rows,
                                     int32_t* __restrict__ inv) {
  const long long total = (long long)min(*n_out, out_cap) * taps;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = nbr[i];
    if (j >= 0 && j < in_rows) inv[(size_t)j * taps + (int)(i % taps)] = (int)(i / taps);
  }
}

// G[r][t][ci] = in[nbr[row0 + r][t]][ci] (zeros for an inactive tap or a row past the live count): the im2col matrix of the gathered
// convolution, so that the weight gradient is ONE dense GEMM dW (Cout x taps*Cin) = dout^T G on the MFMA wgrad kernel
__global__ void gather_taps_kernel(const float* __restrict__ in, int cin, const int32_t* __restrict__ nbr, const int32_t* __restrict__ n_out, int out_cap,
                                   int row0, int rows, int taps, float* __restrict__ g) {
  const int n = min(*n_out, out_cap);
  const int c4 = cin >> 2;
  const long long total = (long long)rows * taps * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4);
    const long long rt = i / c4;
    const int row = row0 + (int)(rt / taps);
    const int j = row < n ? nbr[(size_t)row0 * taps + rt] : -1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j >= 0) v = reinterpret_cast<const float4*>(in + (size_t)j * cin)[q];
    reinterpret_cast<float4*>(g)[i] = v;
  }
}

// dw[co][t][ci < cin_real] (+)= tmp[co][t][ci < cin]
__global__ void compact_cin_kernel(const float* __restrict__ tmp, int cout, int taps, int cin, int cin_real, int accumulate, float* __restrict__ dw) {
  const size_t total = (size_t)cout * taps * cin_real;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_real);
    const size_t ct = i / cin_real;
    const float v = tmp[ct * cin + ci];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

struct WgradSplit { int rows_chunk; size_t g_bytes, tmp_bytes, inner_bytes; pn_conv_desc desc; };
WgradSplit wgrad_split(int out_capacity, int taps, int cout, int cin) {
  WgradSplit w{};
  const long long row_bytes = (long long)taps * cin * 4;
  const long long max_rows = ((1ll << 31) - (1ll << 20)) / std::max<long long>(row_bytes, (long long)cout * 4);   // buffer descriptors address < 2 GiB
  w.rows_chunk = (int)std::min<long long>(out_capacity, std::max<long long>(1, max_rows));
  w.g_bytes = ((size_t)w.rows_chunk * row_bytes + 255) / 256 * 256;
  w.tmp_bytes = ((size_t)cout * taps * cin * 4 + 255) / 256 * 256;
  pn_conv_desc& d = w.desc;
  d.batch = 1; d.in_h = w.rows_chunk; d.in_w = 1; d.cin = taps * cin; d.cout = cout; d.groups = 1; d.kh = 1; d.kw = 1; d.stride = 1;
  d.pad_h = 0; d.pad_w = 0; d.in_pixel_stride = taps * cin; d.in_channel_offset = 0; d.out_pixel_stride = cout; d.out_channel_offset = 0;
  w.inner_bytes = pn_conv2d_wgrad_workspace_bytes(&d);
  return w;
}

struct Dims4 { int B, D, H, W; };
__global__ void from_dense_kernel(const float* __restrict__ dense, const uint32_t* __restrict__ keys, int cap, const int32_t* __restrict__ n_dev, Dims4 d,
                                  int c, float* __restrict__ feats) {
  const long long total = (long long)min(*n_dev, cap) * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / c), ch = (int)(i % c);
    uint32_t key = keys[row];
    const int x = key % d.W; key /= d.W;
    const int y = key % d.H; key /= d.H;
    const int z = key % d.D;
    const int b = key / d.D;
    feats[i] = dense[(((size_t)b * d.H + y) * d.W + x) * ((size_t)c * d.D) + (size_t)ch * d.D + z];
  }
}

__global__ void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaxf(a[i] + b[i], 0.f);
}

}  // namespace

extern "C" {

int pn_sparse_neighbors_transpose(const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps, int in_rows, int32_t* inv,
                                  pn_stream_t stream) {
  PN_REQUIRE(nbr && n_out && inv && out_capacity >= 1 && taps >= 1 && in_rows >= 1, "sparse_neighbors_transpose: bad arguments");
  hipStream_t st = pn::S(stream);
  if (hipMemsetAsync(inv, 0xFF, (size_t)in_rows * taps * sizeof(int32_t), st) != hipSuccess) return pn::fail(PN_ERR_LAUNCH, "sparse_neighbors_transpose: memset");
  const long long total = (long long)out_capacity * taps;
  hipLaunchKernelGGL(transpose_nbr_kernel, dim3((unsigned)std::min<long long>(65535, (total + 255) / 256)), dim3(256), 0, st, nbr, n_out, out_capacity,
                     taps, in_rows, inv);
  return pn::check_launch("transpose_nbr_kernel");
}

size_t pn_sparse_conv_wgrad_workspace_bytes(int out_capacity, int taps, int cout, int cin) {
  if (out_capacity < 1 || taps < 1 || cout < 1 || cin < 1) return 0;
  const WgradSplit w = wgrad_split(out_capacity, taps, cout, cin);
  // the dense wgrad's own workspace is sized for the largest chunk (its split count grows with the row count)
  return w.g_bytes + w.tmp_bytes + w.inner_bytes;
}

int pn_sparse_conv_wgrad_f32(const float* in, int cin, int cin_real, const float* dout, int cout, const int32_t* nbr, const int32_t* n_out,
                             int out_capacity, int taps, float* dw, int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(in && dout && nbr && n_out && dw && workspace, "sparse_conv_wgrad: null pointer");
  PN_REQUIRE(cin >= 4 && cin % 4 == 0 && cout >= 1 && cin_real >= 1 && cin_real <= cin && taps >= 1 && out_capacity >= 1,
             "sparse_conv_wgrad: bad sizes (row width a multiple of 4)");
  if (workspace_bytes < pn_sparse_conv_wgrad_workspace_bytes(out_capacity, taps, cout, cin)) return pn::fail(PN_ERR_WORKSPACE, "sparse_conv_wgrad: workspace too small");
  WgradSplit w = wgrad_split(out_capacity, taps, cout, cin);
  char* base = static_cast<char*>(workspace);
  float* g = reinterpret_cast<float*>(base);
  float* tmp = reinterpret_cast<float*>(base + w.g_bytes);
  void* inner = base + w.g_bytes + w.tmp_bytes;
  hipStream_t st = pn::S(stream);
  const bool direct = cin_real == cin;
  float* target = direct ? dw : tmp;
  for (int row0 = 0, chunk = 0; row0 < out_capacity; row0 += w.rows_chunk, ++chunk) {
    const int rows = std::min(w.rows_chunk, out_capacity - row0);
    const long long total = (long long)rows * taps * (cin / 4);
    hipLaunchKernelGGL(gather_taps_kernel, dim3((unsigned)std::min<long long>(1 << 20, (total + 255) / 256)), dim3(256), 0, st, in, cin, nbr, n_out, out_capacity,
                       row0, rows, taps, g);
    pn_conv_desc d = w.desc;
    d.in_h = rows;
    const int acc = direct ? (accumulate || chunk > 0) : (chunk > 0);
    if (int rc = pn_conv2d_wgrad_f32(&d, g, dout + (size_t)row0 * cout, target, acc, inner, w.inner_bytes, stream)) return rc;
  }
  if (!direct) {
    const size_t total = (size_t)cout * taps * cin_real;
    hipLaunchKernelGGL(compact_cin_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, st, tmp, cout, taps, cin, cin_real,
                       accumulate, dw);
  }
  return pn::check_launch("sparse_conv_wgrad");
}

int pn_sparse_from_dense_nhwc(const float* dense, const uint32_t* keys, int capacity, const int32_t* n_dev, const int32_t* dims, int c, float* feats,
                              pn_stream_t stream) {
  PN_REQUIRE(dense && keys && n_dev && dims && feats && capacity >= 1 && c >= 1, "sparse_from_dense: bad arguments");
  Dims4 d{dims[0], dims[1], dims[2], dims[3]};
  const long long total = (long long)capacity * c;
  hipLaunchKernelGGL(from_dense_kernel, dim3((unsigned)std::min<long long>(65535, (total + 255) / 256)), dim3(256), 0, pn::S(stream), dense, keys, capacity,
                     n_dev, d, c, feats);
  return pn::check_launch("from_dense_kernel");
}

int pn_add_relu_f32(const float* a, const float* b, float* out, size_t n, pn_stream_t stream) {
  PN_REQUIRE(a && b && out, "add_relu: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(add_relu_kernel, dim3((unsigned)std::min<size_t>(65535, (n + 255) / 256)), dim3(256), 0, pn::S(stream), a, b, out, n);
  return pn::check_launch("add_relu_kernel");
}

}  // extern "C"
