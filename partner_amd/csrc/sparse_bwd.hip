// Backward side of the sparse 3-D convolutions of SpMiddleResNetFHD (det3d/models/backbones/scn.py:97-192; in the reference the
// arithmetic and its autograd are spconv's SubMConv3d / SparseConv3d -- third party, absent from the tree: PARITY UNPINNED, the
// checker is fp64 autograd over the dense-with-masks restatement oracle/polar_oracle.py::sp_middle_resnet_fhd).
//
// With out[i] = sum_t W_t in[nbr[i][t]]  (nbr = the forward's neighbour table, -1 = inactive tap):
//   data gradient   din[j]  = sum_{(i,t): nbr[i][t] = j} W_t^T dout[i].  For a fixed (j, t) at most one output site i reads input j
//                   through tap t, so the transposed table inv[j][t] = i is a plain scatter of nbr (pn_sparse_neighbors_transpose) and
//                   the data gradient is the SAME gathered MFMA GEMM as the forward (pn_sparse_conv_f32) over inv with the
//                   (Cin, Cout)-transposed weights -- no atomics, deterministic.
//   weight gradient dW_t[co][ci] = sum_i dout[i][co] * in[nbr[i][t]][ci]: pn_sparse_conv_wgrad_f32, row chunks -> partials in a
//                   workspace -> folded in chunk order (fixed association, no atomics).
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

constexpr int kRowsPerChunk = 2048, kRB = 16;

// one block = (row chunk, tap); thread (ty, tx) owns outputs co = ty + 16 a, ci = tx + 16 b
template <int TM, int TN>
__global__ __launch_bounds__(256) void sparse_wgrad_kernel(const float* __restrict__ in, int cin, const float* __restrict__ dout, int cout,
                                                           const int32_t* __restrict__ nbr, const int32_t* __restrict__ n_out, int out_cap, int taps,
                                                           float* __restrict__ part) {
  __shared__ float s_d[kRB][16 * TM];
  __shared__ float s_x[kRB][16 * TN];
  __shared__ int s_j[kRB];
  const int t = blockIdx.y, chunk = blockIdx.x;
  const int n = min(*n_out, out_cap);
  const int r0 = chunk * kRowsPerChunk, r1 = min(n, r0 + kRowsPerChunk);
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  float acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = 0.f;
  for (int r = r0; r < r1; r += kRB) {
    __syncthreads();
    if (threadIdx.x < kRB) {
      const int row = r + threadIdx.x;
      s_j[threadIdx.x] = row < r1 ? nbr[(size_t)row * taps + t] : -1;
    }
    __syncthreads();
    bool any = false;
    for (int k = 0; k < kRB; ++k) any |= s_j[k] >= 0;
    if (!any) continue;   // uniform across the block
    for (int e = threadIdx.x; e < kRB * 16 * TM; e += 256) {
      const int k = e / (16 * TM), c = e % (16 * TM);
      s_d[k][c] = (s_j[k] >= 0 && c < cout) ? dout[(size_t)(r + k) * cout + c] : 0.f;
    }
    for (int e = threadIdx.x; e < kRB * 16 * TN; e += 256) {
      const int k = e / (16 * TN), c = e % (16 * TN);
      s_x[k][c] = (s_j[k] >= 0 && c < cin) ? in[(size_t)s_j[k] * cin + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kRB; ++k) {
      float dv[TM], xv[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) dv[a] = s_d[k][ty + 16 * a];
#pragma unroll
      for (int b = 0; b < TN; ++b) xv[b] = s_x[k][tx + 16 * b];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = fmaf(dv[a], xv[b], acc[a][b]);
    }
  }
  float* p = part + ((size_t)chunk * taps + t) * (size_t)cout * cin;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int co = ty + 16 * a, ci = tx + 16 * b;
      if (co < cout && ci < cin) p[(size_t)co * cin + ci] = acc[a][b];
    }
}

// dw[co][t][ci] (+)= sum over chunks, in chunk order; cin_real <= cin drops the zero-padded input channels
__global__ void sparse_wgrad_fold_kernel(const float* __restrict__ part, int chunks, int taps, int cout, int cin, int cin_real, int accumulate,
                                         float* __restrict__ dw) {
  const size_t total = (size_t)cout * taps * cin_real;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin_real);
    const int t = (int)((i / cin_real) % taps);
    const int co = (int)(i / ((size_t)cin_real * taps));
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += part[(((size_t)c * taps + t) * cout + co) * cin + ci];
    dw[i] = accumulate ? dw[i] + (float)s : (float)s;
  }
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

template <int TM, int TN>
void launch_wgrad(int chunks, int taps, hipStream_t st, const float* in, int cin, const float* dout, int cout, const int32_t* nbr, const int32_t* n_out,
                  int out_cap, float* part) {
  hipLaunchKernelGGL((sparse_wgrad_kernel<TM, TN>), dim3(chunks, taps), dim3(256), 0, st, in, cin, dout, cout, nbr, n_out, out_cap, taps, part);
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
  return (size_t)pn::cdiv(out_capacity, kRowsPerChunk) * taps * (size_t)cout * cin * sizeof(float);
}

int pn_sparse_conv_wgrad_f32(const float* in, int cin, int cin_real, const float* dout, int cout, const int32_t* nbr, const int32_t* n_out,
                             int out_capacity, int taps, float* dw, int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(in && dout && nbr && n_out && dw && workspace, "sparse_conv_wgrad: null pointer");
  PN_REQUIRE(cin >= 1 && cin <= 128 && cout >= 1 && cout <= 128 && cin_real >= 1 && cin_real <= cin && taps >= 1 && out_capacity >= 1,
             "sparse_conv_wgrad: bad sizes (channels <= 128)");
  if (workspace_bytes < pn_sparse_conv_wgrad_workspace_bytes(out_capacity, taps, cout, cin)) return pn::fail(PN_ERR_WORKSPACE, "sparse_conv_wgrad: workspace too small");
  const int chunks = pn::cdiv(out_capacity, kRowsPerChunk);
  const int tm = pn::cdiv(cout, 16), tn = pn::cdiv(cin, 16);
  float* part = static_cast<float*>(workspace);
  hipStream_t st = pn::S(stream);
#define PN_WG(TM, TN) launch_wgrad<TM, TN>(chunks, taps, st, in, cin, dout, cout, nbr, n_out, out_capacity, part)
  const int a = tm <= 1 ? 1 : tm <= 2 ? 2 : tm <= 4 ? 4 : 8, b = tn <= 1 ? 1 : tn <= 2 ? 2 : tn <= 4 ? 4 : 8;
  switch (a * 10 + b) {
    case 11: PN_WG(1, 1); break;
    case 12: PN_WG(1, 2); break;
    case 21: PN_WG(2, 1); break;
    case 22: PN_WG(2, 2); break;
    case 24: PN_WG(2, 4); break;
    case 42: PN_WG(4, 2); break;
    case 44: PN_WG(4, 4); break;
    case 48: PN_WG(4, 8); break;
    case 84: PN_WG(8, 4); break;
    case 88: PN_WG(8, 8); break;
    default: return pn::fail(PN_ERR_INVALID, "sparse_conv_wgrad: unsupported channel combination %d -> %d", cin, cout);
  }
#undef PN_WG
  const size_t total = (size_t)cout * taps * cin_real;
  hipLaunchKernelGGL(sparse_wgrad_fold_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, st, part, chunks, taps, cout, cin,
                     cin_real, accumulate, dw);
  return pn::check_launch("sparse_wgrad");
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
