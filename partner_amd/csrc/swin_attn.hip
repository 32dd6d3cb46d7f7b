// Window attention of the geometry-aware head.
// Reference (not executable there; this follows the repaired restatement in
// oracle/polar_oracle.py::swv_window_attention):
//   WindowAttention.forward        det3d/models/bbox_heads/swin_utils/sw2votev4_util.py:65-103
//   SwinTransformerBlock.forward   sw2votev4_util.py:127-188  (zero padding to window multiples, cyclic
//                                  shift, window partition / reverse, shift mask of BasicLayer :259-276)
// One wave per (window, head): the 49 tokens' q, k, v head slices (+ the vote embedding, a 3->16->C
// MLP of (pred_centers, vote_cls) added to all three) are staged in LDS; lane i owns query i:
//   s[j] = <q_i, k_j> / max(|q_i||k_j|, 1e-6) / max(tau, 0.01) + rpe(pos_i - pos_j) + mask(i, j)
// softmax over j in registers, out_i = sum_j p_j v_j, written back through LDS as contiguous rows.
// Padding, cyclic shift, partition and their inverses are index arithmetic on the token map: nothing is
// copied.  Padded tokens take part as keys exactly as in Swin (LayerNorm output zero => q = k = v = bias
// + vote_mlp(0)).
#include "pn_common.h"

namespace {

struct SwvParams {
  const float* qkv_bias;  // (3C) or null
  const float* vm_w1;     // (16, 3)
  const float* vm_b1;     // (16)
  const float* vm_w2;     // (C, 16)
  const float* vm_b2;     // (C)
  const float* rp_w1;     // (16, 2)
  const float* rp_b1;     // (16)
  const float* rp_w2;     // (heads, 16)
  const float* rp_b2;     // (heads)
  const float* tau;       // (heads)
};

constexpr int WS = 7, NT = WS * WS, HD = 64, LD = HD + 1;

__global__ __launch_bounds__(64) void swv_window_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ vote, int vote_ps,
                                                             const float* __restrict__ pos, SwvParams P, int H, int W, int C, int shift,
                                                             float* __restrict__ out) {
  __shared__ float Q[NT][LD], K[NT][LD], V[NT][LD];
  __shared__ float nk[NT], px[NT], py[NT];
  __shared__ int region[NT], tok[NT];
  __shared__ float vhid[NT][16];
  const int lane = threadIdx.x;
  const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
  const int nww = Wp / WS;
  const int wy = blockIdx.x / nww, wx = blockIdx.x - wy * nww;
  const int head = blockIdx.y, b = blockIdx.z;
  // ---- token bookkeeping: lane t < 49
  if (lane < NT) {
    const int r = lane / WS, c = lane - r * WS;
    const int hs = wy * WS + r, wsx = wx * WS + c;                 // coordinates in the (shifted) window frame
    const int hp = (hs + shift) % Hp, wp = (wsx + shift) % Wp;     // padded-map coordinates of this token
    const bool valid = hp < H && wp < W;
    tok[lane] = valid ? (b * H + hp) * W + wp : -1;
    px[lane] = valid ? pos[(hp * W + wp) * 2] : 0.f;
    py[lane] = valid ? pos[(hp * W + wp) * 2 + 1] : 0.f;
    const int ih = hs < Hp - WS ? 0 : (hs < Hp - shift ? 1 : 2), iw = wsx < Wp - WS ? 0 : (wsx < Wp - shift ? 1 : 2);
    region[lane] = shift > 0 ? ih * 3 + iw : 0;
    // hidden layer of the vote MLP for this token (zeros for padded tokens)
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (valid) {
      const float* vp = vote + (size_t)tok[lane] * vote_ps;
      v0 = vp[0]; v1 = vp[1]; v2 = vp[2];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float hsum = P.vm_b1[m] + P.vm_w1[m * 3] * v0 + P.vm_w1[m * 3 + 1] * v1 + P.vm_w1[m * 3 + 2] * v2;
      vhid[lane][m] = hsum > 0.f ? hsum : 0.f;
    }
  }
  __syncthreads();
  // ---- stage q, k, v (+ vote embedding): lane = channel of the head
  {
    const int ch = head * HD + lane;
    float w2[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) w2[m] = P.vm_w2[ch * 16 + m];
    const float b2 = P.vm_b2[ch];
    const float bq = P.qkv_bias ? P.qkv_bias[ch] : 0.f, bk = P.qkv_bias ? P.qkv_bias[C + ch] : 0.f, bv = P.qkv_bias ? P.qkv_bias[2 * C + ch] : 0.f;
    for (int t = 0; t < NT; ++t) {
      float ve = b2;
#pragma unroll
      for (int m = 0; m < 16; ++m) ve = fmaf(w2[m], vhid[t][m], ve);
      const int tk = tok[t];
      float q = bq, k = bk, v = bv;
      if (tk >= 0) {
        const float* row = qkv + (size_t)tk * 3 * C;
        q = row[ch]; k = row[C + ch]; v = row[2 * C + ch];
      }
      Q[t][lane] = q + ve; K[t][lane] = k + ve; V[t][lane] = v + ve;
    }
  }
  __syncthreads();
  float qi[HD], nq = 0.f;
  if (lane < NT) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { qi[d] = Q[lane][d]; nq = fmaf(qi[d], qi[d], nq); const float kv = K[lane][d]; s = fmaf(kv, kv, s); }
    nk[lane] = sqrtf(s);
    nq = sqrtf(nq);
  }
  __syncthreads();
  float rw1x[16], rw1y[16], rb1[16], rw2[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { rw1x[m] = P.rp_w1[m * 2]; rw1y[m] = P.rp_w1[m * 2 + 1]; rb1[m] = P.rp_b1[m]; rw2[m] = P.rp_w2[head * 16 + m]; }
  const float rb2 = P.rp_b2[head];
  const float inv_tau = 1.f / fmaxf(P.tau[head], 0.01f);
  float s[NT];
  float smax = -3.0e38f;
  if (lane < NT) {
    const float xi = px[lane], yi = py[lane];
    const int ri = region[lane];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float dot = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) dot = fmaf(qi[d], K[j][d], dot);
      float a = dot / fmaxf(nq * nk[j], 1e-6f) * inv_tau;
      const float dx = xi - px[j], dy = yi - py[j];
      float rp = rb2;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float hdn = fmaf(rw1x[m], dx, fmaf(rw1y[m], dy, rb1[m]));
        rp = fmaf(rw2[m], hdn > 0.f ? hdn : 0.f, rp);
      }
      a += rp;
      if (region[j] != ri) a += -100.f;
      s[j] = a;
      smax = fmaxf(smax, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) { s[j] = expf(s[j] - smax); sum += s[j]; }
    const float inv = 1.f / sum;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float p = s[j] * inv;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = fmaf(p, V[j][d], o[d]);
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) Q[lane][d] = o[d];  // row `lane` of Q is only read by this lane: safe to overwrite
  }
  __syncthreads();
  for (int t = 0; t < NT; ++t) {
    const int tk = tok[t];
    if (tk >= 0) out[(size_t)tk * C + head * HD + lane] = Q[t][lane];
  }
}

}  // namespace

extern "C" {

int pn_swv_window_attn(const float* qkv, const float* vote, int vote_pixel_stride, const float* pos, const float* qkv_bias,
                       const float* vote_w1, const float* vote_b1, const float* vote_w2, const float* vote_b2, const float* rpe_w1,
                       const float* rpe_b1, const float* rpe_w2, const float* rpe_b2, const float* tau, int batch, int h, int w, int c,
                       int heads, int window, int shift, float* out, pn_stream_t stream) {
  PN_REQUIRE(qkv && vote && pos && vote_w1 && vote_b1 && vote_w2 && vote_b2 && rpe_w1 && rpe_b1 && rpe_w2 && rpe_b2 && tau && out,
             "swv_window_attn: null pointer");
  PN_REQUIRE(window == WS && heads >= 1 && c == heads * HD, "swv_window_attn: built for window 7 and head_dim 64");
  PN_REQUIRE(shift >= 0 && shift < WS && batch >= 1 && h >= 1 && w >= 1 && vote_pixel_stride >= 3, "swv_window_attn: bad sizes");
  SwvParams p{qkv_bias, vote_w1, vote_b1, vote_w2, vote_b2, rpe_w1, rpe_b1, rpe_w2, rpe_b2, tau};
  const int nwh = (h + WS - 1) / WS, nww = (w + WS - 1) / WS;
  hipLaunchKernelGGL(swv_window_attn_kernel, dim3(nwh * nww, heads, batch), dim3(64), 0, pn::S(stream), qkv, vote, vote_pixel_stride, pos,
                     p, h, w, c, shift, out);
  return pn::check_launch("swv_window_attn_kernel");
}

}  // extern "C"
