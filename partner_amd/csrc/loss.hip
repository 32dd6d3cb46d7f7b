// CenterPoint loss, forward value.
// Reference: CenterHead.loss / _sigmoid  det3d/models/bbox_heads/center_head.py:244-288,
//            FastFocalLoss / RegLoss     det3d/models/losses/centernet_loss.py:26-54, 6-24,
//            _transpose_and_gather_feat  det3d/core/utils/center_utils.py:66-80
// Two launches: (1) the dense negative focal term, fp64 block partials; (2) one block folds the
// partials in a fixed order and adds the <= B*max_objs positive / box terms.  Deterministic.
#include "pn_common.h"
#include <algorithm>

namespace {

struct BoxSrc {
  const float* p[5];
  int ps[5];   // pixel stride (floats)
  int nch[5];  // channels taken from this source
  int n;
  int sel[16]; // anno_box column of every box dimension (host array copied by value)
};

struct BoxDst {
  float* p[5];
  int ps[5];
};

__global__ __launch_bounds__(256) void focal_neg_kernel(const float* __restrict__ logit, int ps, const float* __restrict__ tgt,
                                                        int B, int C, int H, int W, double* __restrict__ partial) {
  __shared__ double red[256];
  const size_t total = (size_t)B * C * H * W;
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i enumerates the NCHW target; the logits are NHWC
    size_t r = i;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const size_t b = r / C;
    float p = 1.f / (1.f + expf(-logit[((b * H + y) * W + x) * ps + c]));
    p = fminf(fmaxf(p, 1e-4f), 1.f - 1e-4f);
    const float g = 1.f - tgt[i];
    const float g2 = g * g;
    acc += (double)(logf(1.f - p) * (p * p) * (g2 * g2));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out: [det_loss, hm_loss, loc_loss, num_pos, elem[0..ndim)]
__global__ __launch_bounds__(256) void loss_finish_kernel(const double* __restrict__ partial, int nparts,
                                                          const float* __restrict__ logit, int ps, int C, int H, int W, BoxSrc bs,
                                                          const int64_t* __restrict__ ind, const uint8_t* __restrict__ mask,
                                                          const int64_t* __restrict__ cat, const float* __restrict__ anno, int anno_dim, int B, int M, int ndim,
                                                          const float* __restrict__ code_w, float weight, float* __restrict__ out) {
  __shared__ double red[256];
  __shared__ double res[20];
  auto block_sum = [&](double v) -> double {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double neg = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) neg += partial[i];
  neg = block_sum(neg);
  double pos = 0.0, npos = 0.0;
  for (int i = threadIdx.x; i < B * M; i += 256) {
    const float m = mask[i] ? 1.f : 0.f;
    npos += m;
    const int b = i / M;
    const int64_t pix = ind[i];
    float p = 1.f / (1.f + expf(-logit[((size_t)b * H * W + pix) * ps + cat[i]]));
    p = fminf(fmaxf(p, 1e-4f), 1.f - 1e-4f);
    pos += (double)(logf(p) * ((1.f - p) * (1.f - p)) * m);
  }
  pos = block_sum(pos);
  npos = block_sum(npos);
  double loc = 0.0;
  for (int d = 0; d < ndim; ++d) {
    // which source / channel holds box dimension d
    int src = 0, ch = d;
    while (src < bs.n && ch >= bs.nch[src]) ch -= bs.nch[src++];
    double e = 0.0;
    for (int i = threadIdx.x; i < B * M; i += 256) {
      const float m = mask[i] ? 1.f : 0.f;
      const int b = i / M;
      const float pr = bs.p[src][((size_t)b * H * W + ind[i]) * bs.ps[src] + ch];
      const float tg = anno[(size_t)i * anno_dim + bs.sel[d]];
      e += (double)fabsf(pr * m - tg * m);
    }
    e = block_sum(e) / (npos + 1e-4);
    if (threadIdx.x == 0) {
      out[4 + d] = (float)e;
      res[d] = e * (double)code_w[d];
    }
    __syncthreads();
    loc += res[d];
  }
  if (threadIdx.x == 0) {
    const double hm = npos == 0.0 ? -neg : -(pos + neg) / npos;
    out[0] = (float)(hm + (double)weight * loc);
    out[1] = (float)hm;
    out[2] = (float)loc;
    out[3] = (float)npos;
  }
}


// ---- backward -----------------------------------------------------------------------------------
// d det_loss / d hm logits, dense part: the negative focal term of every cell.
//   p = clamp(sigmoid(l)); torch's clamp passes the gradient only inside [1e-4, 1-1e-4]
//   d(-N/npos)/dl = -(1-t)^4 * (2 p log(1-p) - p^2/(1-p)) * p(1-p) / npos
// also zero-fills the pad channels of the NHWC gradient map.
__global__ __launch_bounds__(256) void focal_neg_bwd_kernel(const float* __restrict__ logit, int ps, const float* __restrict__ tgt,
                                                            int B, int C, int H, int W, const float* __restrict__ fwd_out,
                                                            float gscale, float* __restrict__ dhm, int dps) {
  const size_t total = (size_t)B * H * W * dps;
  const float npos = fwd_out[3];
  const float s = gscale / (npos == 0.f ? 1.f : npos);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dps);
    const size_t pix = i / dps;  // (b*H + y)*W + x
    float g = 0.f;
    if (c < C) {
      const size_t b = pix / ((size_t)H * W);
      const size_t yx = pix - b * (size_t)H * W;
      const float sg = 1.f / (1.f + expf(-logit[pix * ps + c]));
      if (sg >= 1e-4f && sg <= 1.f - 1e-4f) {
        const float t = 1.f - tgt[(b * C + c) * (size_t)H * W + yx];
        const float t2 = t * t;
        const float dn = t2 * t2 * (2.f * sg * logf(1.f - sg) - sg * sg / (1.f - sg));
        g = -s * dn * sg * (1.f - sg);
      }
    }
    dhm[i] = g;
  }
}

// positives of the focal term and the L1 box terms: one block per sample, thread per object.
// Objects sharing a cell (same ind, and same cat for the focal term) are folded into the first of
// them in object order, so the result does not depend on scheduling (the reference's gather
// backward is an atomic index_add).
__global__ __launch_bounds__(256) void loss_sparse_bwd_kernel(const float* __restrict__ logit, int ps, int C, int H, int W, BoxSrc bs,
                                                              BoxDst bd, const int64_t* __restrict__ ind, const uint8_t* __restrict__ mask,
                                                              const int64_t* __restrict__ cat, const float* __restrict__ anno, int anno_dim,
                                                              int M, int ndim, const float* __restrict__ code_w, float weight,
                                                              const float* __restrict__ fwd_out, float gscale, float* __restrict__ dhm, int dps) {
  // masked objects of this sample, in object order (LDS list: the duplicate search is O(n_masked^2), not O(M^2))
  __shared__ int live[1024];
  __shared__ int n_live;
  const int b = blockIdx.x;
  const float npos = fwd_out[3];
  const float s_hm = npos == 0.f ? 0.f : gscale / npos;  // npos == 0: the positive term is dropped (centernet_loss.py:50-52)
  const float s_box = gscale * weight / (npos + 1e-4f);
  const int64_t* bind = ind + (size_t)b * M;
  const int64_t* bcat = cat + (size_t)b * M;
  const uint8_t* bmask = mask + (size_t)b * M;
  if (threadIdx.x < 64) {  // first wave compacts with ballots: order preserving
    int base = 0;
    for (int i0 = 0; i0 < M; i0 += 64) {
      const int i = i0 + threadIdx.x;
      const bool m = i < M && bmask[i];
      const unsigned long long bal = __ballot(m);
      if (m) live[base + __popcll(bal & ((1ull << threadIdx.x) - 1ull))] = i;
      base += __popcll(bal);
    }
    if (threadIdx.x == 0) n_live = base;
  }
  __syncthreads();
  const int L = n_live;
  for (int li = threadIdx.x; li < L; li += blockDim.x) {
    const int i = live[li];
    const int64_t pix = bind[i];
    const size_t gp = (size_t)b * H * W + pix;
    // ---- focal positive: owner = first masked object with the same (ind, cat)
    bool owner = true;
    for (int lj = 0; lj < li; ++lj) { const int j = live[lj]; owner = owner && !(bind[j] == pix && bcat[j] == bcat[i]); }
    if (owner) {
      int mult = 1;
      for (int lj = li + 1; lj < L; ++lj) { const int j = live[lj]; mult += (bind[j] == pix && bcat[j] == bcat[i]) ? 1 : 0; }
      const float sg = 1.f / (1.f + expf(-logit[gp * ps + bcat[i]]));
      if (sg >= 1e-4f && sg <= 1.f - 1e-4f) {
        const float dp = (1.f - sg) * (1.f - sg) / sg - 2.f * (1.f - sg) * logf(sg);
        dhm[gp * dps + bcat[i]] += -s_hm * (float)mult * dp * sg * (1.f - sg);
      }
    }
    // ---- boxes: owner = first masked object with the same ind; it adds every duplicate in object order
    bool bowner = true;
    for (int lj = 0; lj < li; ++lj) bowner = bowner && !(bind[live[lj]] == pix);
    if (!bowner) continue;
    for (int d = 0; d < ndim; ++d) {
      int src = 0, ch = d;
      while (src < bs.n && ch >= bs.nch[src]) ch -= bs.nch[src++];
      const float pr = bs.p[src][gp * bs.ps[src] + ch];
      float g = 0.f;
      for (int lj = li; lj < L; ++lj) {
        const int j = live[lj];
        if (bind[j] != pix) continue;
        const float df = pr - anno[((size_t)b * M + j) * anno_dim + bs.sel[d]];
        g += (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
      }
      bd.p[src][gp * bd.ps[src] + ch] = s_box * code_w[d] * g;
    }
  }
}

}  // namespace

extern "C" {

size_t pn_center_loss_workspace_bytes(void) { return 1024 * sizeof(double); }

int pn_center_loss_fwd(const float* hm_logits, int hm_pixel_stride, const float* hm_target, int batch, int classes, int h, int w,
                       const float* const* box_ptrs, const int* box_pixel_strides, const int* box_channels, int n_box_src,
                       const int64_t* ind, const uint8_t* mask, const int64_t* cat, const float* anno_box, int anno_dim,
                       const int* anno_sel, int max_objs, int box_dims, const float* code_weights, float weight, float* out,
                       void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(hm_logits && hm_target && box_ptrs && box_pixel_strides && box_channels && ind && mask && cat && anno_box && anno_sel &&
                 code_weights && out && workspace,
             "center_loss: null pointer");
  PN_REQUIRE(n_box_src >= 1 && n_box_src <= 5 && box_dims >= 1 && box_dims <= 16, "center_loss: bad box description");
  PN_REQUIRE(workspace_bytes >= pn_center_loss_workspace_bytes(), "center_loss: workspace too small");
  BoxSrc bs;
  bs.n = n_box_src;
  int tot = 0;
  for (int i = 0; i < 5; ++i) {
    bs.p[i] = i < n_box_src ? box_ptrs[i] : nullptr;
    bs.ps[i] = i < n_box_src ? box_pixel_strides[i] : 0;
    bs.nch[i] = i < n_box_src ? box_channels[i] : 0;
    tot += bs.nch[i];
  }
  PN_REQUIRE(tot == box_dims, "center_loss: box sources do not add up to box_dims");
  for (int d = 0; d < 16; ++d) bs.sel[d] = d < box_dims ? anno_sel[d] : 0;
  const size_t total = (size_t)batch * classes * h * w;
  const int nparts = (int)std::min<size_t>(1024, (total + 255) / 256);
  double* partial = static_cast<double*>(workspace);
  hipLaunchKernelGGL(focal_neg_kernel, dim3(nparts), dim3(256), 0, pn::S(stream), hm_logits, hm_pixel_stride, hm_target, batch,
                     classes, h, w, partial);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, pn::S(stream), partial, nparts, hm_logits, hm_pixel_stride, classes,
                     h, w, bs, ind, mask, cat, anno_box, anno_dim, batch, max_objs, box_dims, code_weights, weight, out);
  return pn::check_launch("center_loss");
}

int pn_center_loss_bwd(const float* hm_logits, int hm_pixel_stride, const float* hm_target, int batch, int classes, int h, int w,
                       const float* const* box_ptrs, const int* box_pixel_strides, const int* box_channels, int n_box_src,
                       const int64_t* ind, const uint8_t* mask, const int64_t* cat, const float* anno_box, int anno_dim,
                       const int* anno_sel, int max_objs, int box_dims, const float* code_weights, float weight,
                       const float* fwd_out, float grad_scale, float* d_hm, int d_hm_pixel_stride, float* const* d_box_ptrs,
                       const int* d_box_pixel_strides, pn_stream_t stream) {
  PN_REQUIRE(hm_logits && hm_target && box_ptrs && box_pixel_strides && box_channels && ind && mask && cat && anno_box && anno_sel &&
                 code_weights && fwd_out && d_hm && d_box_ptrs && d_box_pixel_strides,
             "center_loss_bwd: null pointer");
  PN_REQUIRE(n_box_src >= 1 && n_box_src <= 5 && box_dims >= 1 && box_dims <= 16, "center_loss_bwd: bad box description");
  PN_REQUIRE(d_hm_pixel_stride >= classes, "center_loss_bwd: gradient pixel stride smaller than the class count");
  PN_REQUIRE(max_objs <= 1024, "center_loss_bwd: at most 1024 objects per sample");
  BoxSrc bs;
  BoxDst bd;
  bs.n = n_box_src;
  int tot = 0;
  for (int i = 0; i < 5; ++i) {
    bs.p[i] = i < n_box_src ? box_ptrs[i] : nullptr;
    bs.ps[i] = i < n_box_src ? box_pixel_strides[i] : 0;
    bs.nch[i] = i < n_box_src ? box_channels[i] : 0;
    bd.p[i] = i < n_box_src ? d_box_ptrs[i] : nullptr;
    bd.ps[i] = i < n_box_src ? d_box_pixel_strides[i] : 0;
    tot += bs.nch[i];
    if (i < n_box_src) {
      PN_REQUIRE(bd.p[i] && bd.ps[i] >= bs.nch[i], "center_loss_bwd: bad gradient buffer for a box source");
      if (int rc = pn::zero_async(bd.p[i], (size_t)batch * h * w * bd.ps[i] * sizeof(float), pn::S(stream))) return rc;
    }
  }
  PN_REQUIRE(tot == box_dims, "center_loss_bwd: box sources do not add up to box_dims");
  for (int d = 0; d < 16; ++d) bs.sel[d] = d < box_dims ? anno_sel[d] : 0;
  const size_t total = (size_t)batch * h * w * d_hm_pixel_stride;
  hipLaunchKernelGGL(focal_neg_bwd_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream),
                     hm_logits, hm_pixel_stride, hm_target, batch, classes, h, w, fwd_out, grad_scale, d_hm, d_hm_pixel_stride);
  hipLaunchKernelGGL(loss_sparse_bwd_kernel, dim3(batch), dim3(256), 0, pn::S(stream), hm_logits, hm_pixel_stride, classes, h, w, bs, bd,
                     ind, mask, cat, anno_box, anno_dim, max_objs, box_dims, code_weights, weight, fwd_out, grad_scale, d_hm,
                     d_hm_pixel_stride);
  return pn::check_launch("center_loss_bwd");
}

}  // extern "C"
