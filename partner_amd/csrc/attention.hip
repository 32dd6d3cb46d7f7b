// Global representation re-alignment (PARTNER SetBlock / SetAttention) -- the non-GEMM parts.
// Reference: det3d/models/utils/set_transformer.py:56-166 (SetAttention), :262-354 (SectorAttention),
// :169-259 (RangeAttention), :357-440 (SectorAttentionV2).
//
// Token layout: (B, H, W, C) fp32, H = range rows, W = azimuth columns, channel contiguous, always
// in PHYSICAL (un-rolled) column order.  The odd blocks' azimuth roll (shift = win_w/2) is an index
// mapping: "rolled" column wr lives at physical column (wr + shift) mod W; nothing is moved.
// All linear layers run on the MFMA GEMM (pn_gemm_bias_act_f32); the kernels here are the
// HBM-bound glue: LayerNorm (+ channel mean for key-point scoring), key-point selection, and the
// three small attentions with their Cartesian relative-position bias MLP.
#include "pn_common.h"
#include <cfloat>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------
// LayerNorm over the channel axis, one wavefront per token row.
__global__ void layernorm_kernel(const float* __restrict__ x, size_t rows, int c, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, float* __restrict__ out,
                                 float* __restrict__ chan_mean) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * c;
  float s = 0.f;
  for (int k = lane; k < c; k += 64) s += xr[k];
  const float mean = pn::wave_sum(s) / (float)c;
  float v = 0.f;
  for (int k = lane; k < c; k += 64) {
    const float d = xr[k] - mean;
    v += d * d;
  }
  const float rstd = 1.f / sqrtf(pn::wave_sum(v) / (float)c + eps);
  float acc = 0.f;
  for (int k = lane; k < c; k += 64) {
    const float y = (xr[k] - mean) * rstd * gamma[k] + beta[k];
    out[row * c + k] = y;
    acc += y;
  }
  if (chan_mean) {
    acc = pn::wave_sum(acc);
    if (lane == 0) chan_mean[row] = acc / (float)c;
  }
}

// ---------------------------------------------------------------------------------------------
// Key points: per (batch, rolled column) the K rows with the largest score among
//   score[h] = s[h] if s[h] is a local maximum along range (window 3, borders excluded) else 0
// (set_transformer.py:134-147).  Ties are resolved towards the smaller row index.
// One wavefront per column.  Outputs: top_idx (B,K,W), kp (B, K*W, C) gathered rows of xn,
// kpos (B,K,W,2) gathered Cartesian positions.
__global__ void keypoints_kernel(const float* __restrict__ s, const float* __restrict__ xn, const float* __restrict__ pos,
                                 int B, int H, int W, int C, int K, int shift, int32_t* __restrict__ top_idx,
                                 float* __restrict__ kp, float* __restrict__ kpos) {
  extern __shared__ float sc[];  // H scores
  const int lane = threadIdx.x;
  const int b = blockIdx.x / W, wr = blockIdx.x % W;
  const int wp = (wr + shift) % W;
  for (int h = lane; h < H; h += 64) {
    const float v = s[((size_t)b * H + h) * W + wp];
    float lm = 0.f;
    if (h >= 1 && h <= H - 2) {
      const float a = s[((size_t)b * H + h - 1) * W + wp], c = s[((size_t)b * H + h + 1) * W + wp];
      lm = fmaxf(fmaxf(a, v), c);
    }
    sc[h] = (lm == v) ? v : 0.f * v;
  }
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    float best = -FLT_MAX;
    int bi = 0x7fffffff;
    for (int h = lane; h < H; h += 64) {
      const float v = sc[h];
      if (v > best || (v == best && h < bi)) {
        best = v;
        bi = h;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (lane == 0) {
      top_idx[((size_t)b * K + k) * W + wr] = bi;
      sc[bi] = -FLT_MAX;  // taken
      kpos[(((size_t)b * K + k) * W + wr) * 2 + 0] = pos[((size_t)bi * W + wp) * 2 + 0];
      kpos[(((size_t)b * K + k) * W + wr) * 2 + 1] = pos[((size_t)bi * W + wp) * 2 + 1];
    }
    __syncthreads();
    const float* src = xn + (((size_t)b * H + bi) * W + wp) * C;
    float* dst = kp + ((size_t)b * K * W + (size_t)k * W + wr) * C;
    for (int c = lane; c < C; c += 64) dst[c] = src[c];
  }
}

// relative-position bias: Conv1d(2->16) + BatchNorm1d(eval, folded) + ReLU + Conv1d(16->heads)
// pe layout: w1[16][2], scale[16], shift[16], w2[heads][16], b2[heads]
struct PosMlp {
  const float* pe;
  int heads;
  __device__ __forceinline__ void hidden(float dx, float dy, float (&h)[16]) const {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float t = (pe[2 * j] * dx + pe[2 * j + 1] * dy) * pe[32 + j] + pe[48 + j];
      h[j] = t > 0.f ? t : 0.f;
    }
  }
  __device__ __forceinline__ float out(const float (&h)[16], int head) const {
    const float* w2 = pe + 64 + head * 16;
    float o = pe[64 + heads * 16 + head];
#pragma unroll
    for (int j = 0; j < 16; ++j) o += w2[j] * h[j];
    return o;
  }
};

// ---------------------------------------------------------------------------------------------
// SectorAttention core (key points <- their azimuth column).  Block = (batch, rolled column),
// wave = head.  q comes from the (B, K*W, C) buffer read through the reference's raw
// reinterpretation as (B, C, K, W) (set_transformer.py:331-334); k|v is the (B,H,W,2C) projection
// of the normalised tokens.  out: (B, K*W, C).
template <int KT>
__global__ __launch_bounds__(256) void sector_kp_attn_kernel(const float* __restrict__ qraw, const float* __restrict__ kv,
                                                             const float* __restrict__ xpos, const float* __restrict__ kpos,
                                                             PosMlp pm, int B, int H, int W, int C, int K, int shift,
                                                             float scale, float* __restrict__ out) {
  extern __shared__ float lds[];
  const int heads = pm.heads, hd = C / heads;
  const int head = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x / W, wr = blockIdx.x % W, wp = (wr + shift) % W;
  float* qs = lds + head * (K * hd + K * H);  // q[K][hd], p[K][H]
  float* ps = qs + K * hd;
  if (head >= heads) return;
  const float* qb = qraw + (size_t)b * K * W * C;
  for (int i = lane; i < K * hd; i += 64) {
    const int k = i / hd, d = i - k * hd;
    qs[i] = qb[((size_t)(head * hd + d) * K + k) * W + wr] * scale;
  }
  __syncthreads();
  // logits: lanes over range rows
  for (int h = lane; h < H; h += 64) {
    const float* kr = kv + (((size_t)b * H + h) * W + wp) * (2 * C) + head * hd;
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    for (int d = 0; d < hd; d += 4) {
      const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + d);
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K)
          acc[k] += qs[k * hd + d] * kk[0] + qs[k * hd + d + 1] * kk[1] + qs[k * hd + d + 2] * kk[2] + qs[k * hd + d + 3] * kk[3];
    }
    const float px = xpos[((size_t)h * W + wp) * 2], py = xpos[((size_t)h * W + wp) * 2 + 1];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k >= K) continue;
      const float* kp2 = kpos + (((size_t)b * K + k) * W + wr) * 2;
      float hid[16];
      pm.hidden(kp2[0] - px, kp2[1] - py, hid);
      ps[k * H + h] = acc[k] + pm.out(hid, head);
    }
  }
  __syncthreads();
  // softmax over the rows, per key point
  for (int k = 0; k < K; ++k) {
    float m = -FLT_MAX;
    for (int h = lane; h < H; h += 64) m = fmaxf(m, ps[k * H + h]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int h = lane; h < H; h += 64) {
      const float e = expf(ps[k * H + h] - m);
      ps[k * H + h] = e;
      sum += e;
    }
    sum = pn::wave_sum(sum);
    const float inv = 1.f / sum;
    for (int h = lane; h < H; h += 64) ps[k * H + h] *= inv;
  }
  __syncthreads();
  // out[k][d] = sum_h p[k][h] * v[h][d]: lanes over d
  for (int d = lane; d < hd; d += 64) {
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    for (int h = 0; h < H; ++h) {
      const float vv = kv[(((size_t)b * H + h) * W + wp) * (2 * C) + C + head * hd + d];
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) acc[k] += ps[k * H + h] * vv;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) out[((size_t)b * K * W + (size_t)k * W + wr) * C + head * hd + d] = acc[k];
  }
}

// ---------------------------------------------------------------------------------------------
// RangeAttention core among key points: windows of K x win_w tokens.  qkv: (B, K*W, 3C) projection
// of the normalised key points (q | k | v).  Block = (batch, window), wave = head.
__global__ __launch_bounds__(256) void range_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ kpos,
                                                         PosMlp pm, int B, int W, int C, int K, int win_w, float scale,
                                                         float* __restrict__ out) {
  extern __shared__ float lds[];
  const int heads = pm.heads, hd = C / heads;
  const int head = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = W / win_w, n = K * win_w;
  const int b = blockIdx.x / nw, j = blockIdx.x % nw;
  if (head >= heads) return;
  float* qs = lds + head * (3 * n * hd + n * n);  // q, k, v [n][hd], p [n][n]
  float* ks = qs + n * hd;
  float* vs = ks + n * hd;
  float* ps = vs + n * hd;
  for (int i = lane; i < n * hd; i += 64) {
    const int t = i / hd, d = i - t * hd;
    const int k = t / win_w, ww = t - k * win_w;
    const float* row = qkv + ((size_t)b * K * W + (size_t)k * W + j * win_w + ww) * (3 * C) + head * hd + d;
    qs[i] = row[0] * scale;
    ks[i] = row[C];
    vs[i] = row[2 * C];
  }
  __syncthreads();
  for (int e = lane; e < n * n; e += 64) {
    const int qi = e / n, kj = e - qi * n;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc += qs[qi * hd + d] * ks[kj * hd + d];
    const int k1 = qi / win_w, w1 = qi - k1 * win_w, k2 = kj / win_w, w2 = kj - k2 * win_w;
    const float* p1 = kpos + (((size_t)b * K + k1) * W + j * win_w + w1) * 2;
    const float* p2 = kpos + (((size_t)b * K + k2) * W + j * win_w + w2) * 2;
    float hid[16];
    pm.hidden(p1[0] - p2[0], p1[1] - p2[1], hid);
    ps[e] = acc + pm.out(hid, head);
  }
  __syncthreads();
  for (int qi = lane; qi < n; qi += 64) {
    float m = -FLT_MAX;
    for (int kj = 0; kj < n; ++kj) m = fmaxf(m, ps[qi * n + kj]);
    float sum = 0.f;
    for (int kj = 0; kj < n; ++kj) {
      const float e = expf(ps[qi * n + kj] - m);
      ps[qi * n + kj] = e;
      sum += e;
    }
    const float inv = 1.f / sum;
    for (int kj = 0; kj < n; ++kj) ps[qi * n + kj] *= inv;
  }
  __syncthreads();
  for (int i = lane; i < n * hd; i += 64) {
    const int t = i / hd, d = i - t * hd;
    float acc = 0.f;
    for (int kj = 0; kj < n; ++kj) acc += ps[t * n + kj] * vs[kj * hd + d];
    const int k = t / win_w, ww = t - k * win_w;
    out[((size_t)b * K * W + (size_t)k * W + j * win_w + ww) * C + head * hd + d] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// SectorAttentionV2 core (azimuth column <- its key points).  q: (B,H,W,C) projection of the tokens;
// k|v: (B, K*W, 2C) projection of the key points, each half read through the raw (B,C,K,W)
// reinterpretation (set_transformer.py:417-425).  Block = (batch, rolled column), lanes over rows.
// out: (B,H,W,C) written at the PHYSICAL column (the roll-back of set_transformer.py:157-160).
template <int KT>
__global__ __launch_bounds__(256) void sector_col_attn_kernel(const float* __restrict__ q, const float* __restrict__ kvraw,
                                                              const float* __restrict__ xpos, const float* __restrict__ kpos,
                                                              PosMlp pm, int B, int H, int W, int C, int K, int shift,
                                                              float scale, float* __restrict__ out) {
  extern __shared__ float lds[];  // kk[K][C], vv[K][C]
  const int heads = pm.heads, hd = C / heads;
  const int b = blockIdx.x / W, wr = blockIdx.x % W, wp = (wr + shift) % W;
  float* kk = lds;
  float* vv = lds + K * C;
  const float* kvb = kvraw + (size_t)b * K * W * (2 * C);
  // raw view of the (K*W, 2C) buffer?  No: proj_k and proj_v outputs are separate (K*W, C) tensors in
  // the reference; here they are the two column halves of one (K*W, 2C) GEMM output, so element
  // [row][c] of proj_k is kvb[row*2C + c].  The reinterpretation (B, C, K, W) addresses the FLAT
  // (K*W*C) proj_k buffer: flat = ((c*K + k)*W + w)  ->  row = flat / C, col = flat % C.
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    const int k = i / C, c = i - k * C;
    const size_t flat = ((size_t)c * K + k) * W + wr;
    const size_t row = flat / C, col = flat - row * C;
    kk[i] = kvb[row * (2 * C) + col];
    vv[i] = kvb[row * (2 * C) + C + col];
  }
  __syncthreads();
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    const float* qr = q + (((size_t)b * H + h) * W + wp) * C;
    float* orow = out + (((size_t)b * H + h) * W + wp) * C;
    const float px = xpos[((size_t)h * W + wp) * 2], py = xpos[((size_t)h * W + wp) * 2 + 1];
    float hid[KT][16];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k >= K) continue;
      const float* kp2 = kpos + (((size_t)b * K + k) * W + wr) * 2;
      pm.hidden(px - kp2[0], py - kp2[1], hid[k]);
    }
    for (int head = 0; head < heads; ++head) {
      float lg[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) lg[k] = 0.f;
      for (int d = 0; d < hd; d += 4) {
        const f32x4 qq = *reinterpret_cast<const f32x4*>(qr + head * hd + d);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k >= K) continue;
          const float* kr = kk + k * C + head * hd + d;
          lg[k] += qq[0] * kr[0] + qq[1] * kr[1] + qq[2] * kr[2] + qq[3] * kr[3];
        }
      }
      float m = -FLT_MAX;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k >= K) continue;
        lg[k] = lg[k] * scale + pm.out(hid[k], head);
        m = fmaxf(m, lg[k]);
      }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k >= K) continue;
        lg[k] = expf(lg[k] - m);
        sum += lg[k];
      }
      const float inv = 1.f / sum;
      for (int d = 0; d < hd; d += 4) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k >= K) continue;
          const float p = lg[k] * inv;
          const float* vr = vv + k * C + head * hd + d;
          o[0] += p * vr[0]; o[1] += p * vr[1]; o[2] += p * vr[2]; o[3] += p * vr[3];
        }
        *reinterpret_cast<f32x4*>(orow + head * hd + d) = o;
      }
    }
  }
}

}  // namespace

extern "C" {

int pn_layernorm_f32(const float* x, size_t rows, int c, const float* gamma, const float* beta, float eps, float* out,
                     float* chan_mean, pn_stream_t stream) {
  PN_REQUIRE(x && gamma && beta && out && c >= 1, "layernorm: bad arguments");
  if (rows == 0) return PN_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, pn::S(stream), x, rows, c, gamma, beta,
                     eps, out, chan_mean);
  return pn::check_launch("layernorm_kernel");
}

int pn_setblock_keypoints(const float* chan_mean, const float* xn, const float* pos, int batch, int h, int w, int c, int k,
                          int shift, int32_t* top_idx, float* kp, float* kpos, pn_stream_t stream) {
  PN_REQUIRE(chan_mean && xn && pos && top_idx && kp && kpos, "keypoints: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 3 && w >= 1 && c >= 1 && k >= 1 && k <= h && k <= 8, "keypoints: bad sizes");
  hipLaunchKernelGGL(keypoints_kernel, dim3(batch * w), dim3(64), h * sizeof(float), pn::S(stream), chan_mean, xn, pos, batch,
                     h, w, c, k, shift, top_idx, kp, kpos);
  return pn::check_launch("keypoints_kernel");
}

int pn_setblock_sector_kp_attn(const float* q_raw, const float* kv, const float* xpos, const float* kpos, const float* pos_mlp,
                               int batch, int h, int w, int c, int heads, int k, int shift, float scale, float* out,
                               pn_stream_t stream) {
  PN_REQUIRE(q_raw && kv && xpos && kpos && pos_mlp && out, "sector_kp_attn: null pointer");
  PN_REQUIRE(heads >= 1 && heads <= 4 && c % heads == 0 && (c / heads) % 4 == 0 && k <= 8, "sector_kp_attn: bad sizes");
  const size_t smem = (size_t)heads * (k * (c / heads) + k * h) * sizeof(float);
  PN_REQUIRE(smem <= 64 * 1024, "sector_kp_attn: column too long for LDS");
  PosMlp pm{pos_mlp, heads};
  if (k <= 4)
    hipLaunchKernelGGL(sector_kp_attn_kernel<4>, dim3(batch * w), dim3(256), smem, pn::S(stream), q_raw, kv, xpos, kpos, pm, batch,
                       h, w, c, k, shift, scale, out);
  else
    hipLaunchKernelGGL(sector_kp_attn_kernel<8>, dim3(batch * w), dim3(256), smem, pn::S(stream), q_raw, kv, xpos, kpos, pm, batch,
                       h, w, c, k, shift, scale, out);
  return pn::check_launch("sector_kp_attn_kernel");
}

int pn_setblock_range_attn(const float* qkv, const float* kpos, const float* pos_mlp, int batch, int w, int c, int heads, int k,
                           int win_w, float scale, float* out, pn_stream_t stream) {
  PN_REQUIRE(qkv && kpos && pos_mlp && out, "range_attn: null pointer");
  PN_REQUIRE(heads >= 1 && heads <= 4 && c % heads == 0 && w % win_w == 0, "range_attn: bad sizes");
  const int n = k * win_w, hd = c / heads;
  const size_t smem = (size_t)heads * (3 * n * hd + n * n) * sizeof(float);
  PN_REQUIRE(smem <= 160 * 1024, "range_attn: window too large for LDS");
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&range_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  PosMlp pm{pos_mlp, heads};
  hipLaunchKernelGGL(range_attn_kernel, dim3(batch * (w / win_w)), dim3(256), smem, pn::S(stream), qkv, kpos, pm, batch, w, c, k,
                     win_w, scale, out);
  return pn::check_launch("range_attn_kernel");
}

int pn_setblock_sector_col_attn(const float* q, const float* kv_raw, const float* xpos, const float* kpos, const float* pos_mlp,
                                int batch, int h, int w, int c, int heads, int k, int shift, float scale, float* out,
                                pn_stream_t stream) {
  PN_REQUIRE(q && kv_raw && xpos && kpos && pos_mlp && out, "sector_col_attn: null pointer");
  PN_REQUIRE(heads >= 1 && c % heads == 0 && (c / heads) % 4 == 0 && k <= 8 && c % 4 == 0, "sector_col_attn: bad sizes");
  const size_t smem = (size_t)2 * k * c * sizeof(float);
  PosMlp pm{pos_mlp, heads};
  if (k <= 4)
    hipLaunchKernelGGL(sector_col_attn_kernel<4>, dim3(batch * w), dim3(256), smem, pn::S(stream), q, kv_raw, xpos, kpos, pm, batch,
                       h, w, c, k, shift, scale, out);
  else
    hipLaunchKernelGGL(sector_col_attn_kernel<8>, dim3(batch * w), dim3(256), smem, pn::S(stream), q, kv_raw, xpos, kpos, pm, batch,
                       h, w, c, k, shift, scale, out);
  return pn::check_launch("sector_col_attn_kernel");
}

}  // extern "C"
