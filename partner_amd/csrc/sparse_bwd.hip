// Backward side of the sparse 3-D convolutions of SpMiddleResNetFHD (det3d/models/backbones/scn.py:97-192; in the reference the
// arithmetic and its autograd are spconv's SubMConv3d / SparseConv3d -- third party, absent from the tree: PARITY UNPINNED, the
// checker is fp64 autograd over the dense-with-masks restatement oracle/polar_oracle.py::sp_middle_resnet_fhd).
//
// With out[i] = sum_t W_t in[nbr[i][t]]  (nbr = the forward's neighbour table, -1 = inactive tap):
//   data gradient   din[j]  = sum_{(i,t): nbr[i][t] = j} W_t^T dout[i].  For a fixed (j, t) at most one output site i reads input j
//                   through tap t, so the transposed table inv[j][t] = i is a plain scatter of nbr (pn_sparse_neighbors_transpose) and
//                   the data gradient is the SAME gathered MFMA GEMM as the forward (pn_sparse_conv_f32) over inv with the
//                   (Cin, Cout)-transposed weights -- no atomics, deterministic.
//   weight gradient dW_t[co][ci] = sum_i dout[i][co] * in[nbr[i][t]][ci]: pn_sparse_conv_wgrad_f32 lives in conv_bwd.hip -- the dense
//                   MFMA weight-gradient kernel with the neighbour table in its X loader (row of pixel m under tap t = nbr[m][t]), row
//                   slices reduced in slice order (fixed association, no atomics).  Earlier versions: a VALU kernel (~5 TFLOP/s, a quarter
//                   of the PARTNER training iteration) and a materialised im2col matrix + dense GEMM (HBM-bound on the 0.5-0.9 GB matrix).
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

__global__ void fill_minus_one_kernel(int32_t* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = -1;
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
  // (a fill kernel, not hipMemsetAsync: memset nodes of a captured hipGraph are not re-executed reliably, pn_common.h)
  hipLaunchKernelGGL(fill_minus_one_kernel, dim3((unsigned)std::min<size_t>(2048, ((size_t)in_rows * taps + 255) / 256)), dim3(256), 0, st, inv, (size_t)in_rows * taps);
  const long long total = (long long)out_capacity * taps;
  hipLaunchKernelGGL(transpose_nbr_kernel, dim3((unsigned)std::min<long long>(65535, (total + 255) / 256)), dim3(256), 0, st, nbr, n_out, out_capacity,
                     taps, in_rows, inv);
  return pn::check_launch("transpose_nbr_kernel");
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
