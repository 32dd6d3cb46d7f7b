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

// token (b, h, w) of a (B, H, W) range-major map (the reference's order, set_transformer.py:118-131) or of the azimuth-major
// (B, W, H) map the dense BEV tensor arrives in (NHWC with theta outermost): with col_major every azimuth column -- the unit the two
// sector attentions and the key-point selection work on -- is ONE contiguous slab and no transpose surrounds the blocks
__device__ __forceinline__ size_t tok(int b, int h, int w, int H, int W, int cm) {
  return cm ? ((size_t)b * W + w) * H + h : ((size_t)b * H + h) * W + w;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the channel axis, one wavefront per token row.  NPL > 0: the row (c = 64 NPL values) stays in registers -- one pass
// over memory, NPL loads in flight per lane; element -> lane assignment and summation order are those of the generic loop (NPL = 0),
// so both give the same bits.
__device__ __forceinline__ unsigned short ln_bf16_bits(float f) {      // round to nearest even (the conversion of pn_f32_to_bf16)
  unsigned u = __builtin_bit_cast(unsigned, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <int NPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, size_t rows, int c, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                        float* __restrict__ chan_mean, unsigned short* __restrict__ out16) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * c;
  if constexpr (NPL > 0) {
    float v[NPL], g[NPL], be[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) v[j] = xr[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < NPL; ++j) { g[j] = gamma[lane + 64 * j]; be[j] = beta[lane + 64 * j]; }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) s += v[j];
    const float mean = pn::wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) { const float d = v[j] - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(pn::wave_sum(q) / (float)c + eps);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float y = (v[j] - mean) * rstd * g[j] + be[j];
      if (out) out[row * c + lane + 64 * j] = y;
      if (out16) out16[row * c + lane + 64 * j] = ln_bf16_bits(y);
      acc += y;
    }
    if (chan_mean) {
      acc = pn::wave_sum(acc);
      if (lane == 0) chan_mean[row] = acc / (float)c;
    }
  } else {
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
      if (out) out[row * c + k] = y;
      if (out16) out16[row * c + k] = ln_bf16_bits(y);
      acc += y;
    }
    if (chan_mean) {
      acc = pn::wave_sum(acc);
      if (lane == 0) chan_mean[row] = acc / (float)c;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Key points: per (batch, rolled column) the K rows with the largest score among
//   score[h] = s[h] if s[h] is a local maximum along range (window 3, borders excluded) else 0
// (set_transformer.py:134-147).  Ties are resolved towards the smaller row index.
// One wavefront per column.  Outputs: top_idx (B,K,W), kp (B, K*W, C) gathered rows of xn,
// kpos (B,K,W,2) gathered Cartesian positions.
__global__ void keypoints_kernel(const float* __restrict__ s, const float* __restrict__ xn, const float* __restrict__ pos,
                                 int B, int H, int W, int C, int K, int shift, int cm, int32_t* __restrict__ top_idx,
                                 float* __restrict__ kp, float* __restrict__ kpos) {
  extern __shared__ float sc[];  // H scores
  const int lane = threadIdx.x;
  const int b = blockIdx.x / W, wr = blockIdx.x % W;
  const int wp = (wr + shift) % W;
  for (int h = lane; h < H; h += 64) {
    const float v = s[tok(b, h, wp, H, W, cm)];
    float lm = 0.f;
    if (h >= 1 && h <= H - 2) {
      const float a = s[tok(b, h - 1, wp, H, W, cm)], c = s[tok(b, h + 1, wp, H, W, cm)];
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
    const float* src = xn + tok(b, bi, wp, H, W, cm) * C;
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

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
// sum over the 16 lanes of a row group (lanes 16 g .. 16 g + 15): every lane ends with the total
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// SectorAttention core (key points <- their azimuth column).  Block = (batch, rolled column).  q comes from the (B, K*W, C) buffer
// read through the reference's raw reinterpretation as (B, C, K, W) (set_transformer.py:331-334); k|v is the (tokens, 2C) projection
// of the normalised tokens.  out: (B, K*W, C).
// r3: SIXTEEN waves per column -- wave = (head, quarter of the rows) -- because the kernel is a chain of latencies, not of bytes: one
// block per column, and (r2) one wave per head walked the 144 rows in dependent steps (84 us for 75 MB).  The column is streamed in
// 16-byte pieces, four rows per wave instruction (lane = (row mod 4, 4-channel group): 256 contiguous bytes per row), all of a wave's
// loads of a phase in flight at once; partial dot products are reduced over the 16 lanes of a row; the relative-position bias of
// every (key point, row) is formed once up front, straight into the logit table; the four row quarters of P V are joined in LDS in
// a fixed order.
constexpr int SKP_CH = 4;        // row chunks per column
constexpr int SKP_UNR = 5;       // row groups (of 4) whose loads a wave keeps in flight
template <int KT>
__global__ __launch_bounds__(1024) void sector_kp_attn_kernel(const float* __restrict__ qraw, const float* __restrict__ kv,
                                                              const float* __restrict__ xpos, const float* __restrict__ kpos,
                                                              PosMlp pm, int B, int H, int W, int C, int K, int shift, int cm,
                                                              float scale, float* __restrict__ out) {
  extern __shared__ float lds[];
  const int heads = pm.heads, hd = C / heads;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int head = wv & 3, chunk = wv >> 2;
  const int b = blockIdx.x / W, wr = blockIdx.x % W, wp = (wr + shift) % W;
  float* qs = lds;                         // [heads][K][hd]
  float* ps = qs + heads * K * hd;         // [heads][K][H]
  float* part = ps + heads * K * H;        // [SKP_CH][heads][K][hd]
  const float* qb = qraw + (size_t)b * K * W * C;
  for (int i = tid; i < heads * K * hd; i += 1024) {
    const int hh = i / (K * hd), r = i - hh * (K * hd), k = r / hd, d = r - k * hd;
    qs[i] = qb[((size_t)(hh * hd + d) * K + k) * W + wr] * scale;
  }
  for (int i = tid; i < K * H; i += 1024) {
    const int k = i / H, h = i - k * H;
    const float px = xpos[((size_t)h * W + wp) * 2], py = xpos[((size_t)h * W + wp) * 2 + 1];
    const float* kp2 = kpos + (((size_t)b * K + k) * W + wr) * 2;
    float hid[16];
    pm.hidden(kp2[0] - px, kp2[1] - py, hid);
    for (int hh = 0; hh < heads; ++hh) ps[(hh * K + k) * H + h] = pm.out(hid, hh);
  }
  __syncthreads();
  const int rs = lane >> 4, d4 = lane & 15;
  const bool dok = d4 * 4 < hd, live = head < heads;
  const size_t tstride = cm ? 1 : (size_t)W;        // token step between consecutive rows of the column
  const size_t t0 = tok(b, 0, wp, H, W, cm);
  const int groups = (H + 3) >> 2, gpc = (groups + SKP_CH - 1) / SKP_CH;      // row groups, groups per chunk
  const int g0 = chunk * gpc;
  const float* kbase = kv + t0 * (2 * C) + (live ? head : 0) * hd + (dok ? d4 * 4 : 0);
  const size_t rstep = tstride * (size_t)(2 * C);
  float* psh = ps + (live ? head : 0) * K * H;
  {
    f32x4 qv[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) qv[k] = (k < K && dok && live) ? *reinterpret_cast<const f32x4*>(qs + (head * K + k) * hd + d4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int u0 = 0; u0 < gpc; u0 += SKP_UNR) {
      f32x4 kk[SKP_UNR];
#pragma unroll
      for (int u = 0; u < SKP_UNR; ++u) {
        const int h = min(4 * (g0 + u0 + u) + rs, H - 1);
        kk[u] = *reinterpret_cast<const f32x4*>(kbase + (size_t)h * rstep);
      }
#pragma unroll
      for (int u = 0; u < SKP_UNR; ++u) {
        const int g = g0 + u0 + u, h = 4 * g + rs;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k >= K) continue;
          const float sum = group16_sum(dok ? dot4(qv[k], kk[u]) : 0.f);
          if (d4 == 0 && h < H && u0 + u < gpc && live) psh[k * H + h] += sum;
        }
      }
    }
  }
  __syncthreads();
  // softmax over the rows: one wave per (head, key point)
  for (int pr = wv; pr < heads * K; pr += 16) {
    float* row = ps + pr * H;
    float m = -FLT_MAX;
    for (int h = lane; h < H; h += 64) m = fmaxf(m, row[h]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float sum = 0.f;
    for (int h = lane; h < H; h += 64) {
      const float e = expf(row[h] - m);
      row[h] = e;
      sum += e;
    }
    sum = pn::wave_sum(sum);
    const float inv = 1.f / sum;
    for (int h = lane; h < H; h += 64) row[h] *= inv;
  }
  __syncthreads();
  // partial out[k][d] over this wave's rows: lane (rs, d4) adds the rows h = rs (mod 4), then the four row residues are joined
  {
    f32x4 acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int u0 = 0; u0 < gpc; u0 += SKP_UNR) {
      f32x4 vv[SKP_UNR];
#pragma unroll
      for (int u = 0; u < SKP_UNR; ++u) {
        const int h = min(4 * (g0 + u0 + u) + rs, H - 1);
        vv[u] = *reinterpret_cast<const f32x4*>(kbase + C + (size_t)h * rstep);
      }
#pragma unroll
      for (int u = 0; u < SKP_UNR; ++u) {
        const int h = 4 * (g0 + u0 + u) + rs;
        const bool ok = h < H && u0 + u < gpc;
        const int hc = min(h, H - 1);
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (k < K) acc[k] += (ok ? psh[k * H + hc] : 0.f) * vv[u];
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k >= K) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[k][e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[k][e] = v;
      }
      if (rs == 0 && dok && live) *reinterpret_cast<f32x4*>(part + ((chunk * heads + head) * K + k) * hd + d4 * 4) = acc[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < heads * K * hd; i += 1024) {
    const int hh = i / (K * hd), r = i - hh * (K * hd), k = r / hd, d = r - k * hd;
    float v = part[i];
#pragma unroll
    for (int c = 1; c < SKP_CH; ++c) v += part[c * heads * K * hd + i];
    out[((size_t)b * K * W + (size_t)k * W + wr) * C + hh * hd + d] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// RangeAttention core among key points: windows of n = K x win_w tokens.  qkv: (B, K*W, 3C) projection
// of the normalised key points (q | k | v).
// r3: block = (batch, window, HEAD) (r2: one block per window with a wave per head -- 32 blocks for the Waymo map, and a K image whose
// row stride put every lane of a logit row on one LDS bank: 160 us).  q / k rows padded to hd + 1 floats (the lanes of a logit row
// walk k: bank = (kj + d) mod 32), the window's token positions staged once, softmax with eight threads per row, P V with float4 rows.
__global__ __launch_bounds__(256) void range_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ kpos,
                                                         PosMlp pm, int B, int W, int C, int K, int win_w, float scale,
                                                         float* __restrict__ out) {
  extern __shared__ float lds[];
  const int heads = pm.heads, hd = C / heads;
  const int tid = threadIdx.x;
  const int nw = W / win_w, n = K * win_w;
  const int head = blockIdx.x % heads, bj = blockIdx.x / heads;
  const int b = bj / nw, j = bj % nw;
  const int qld = hd + 1, pld = n + 1;
  float* qs = lds;                    // [n][hd + 1]
  float* ks = qs + n * qld;           // [n][hd + 1]
  float* vs = ks + n * qld;           // [n][hd]    (16-byte aligned: n * qld * 2 is a multiple of 4 when n is)
  float* ps = vs + n * hd;            // [n][n + 1]
  float* tp = ps + n * pld;           // [n][2] token positions
  auto trow = [&](int t) { const int k = t / win_w, ww = t - k * win_w; return (size_t)b * K * W + (size_t)k * W + j * win_w + ww; };
  for (int i = tid; i < n * hd; i += 256) {
    const int t = i / hd, d = i - t * hd;
    const float* row = qkv + trow(t) * (3 * C) + head * hd + d;
    qs[t * qld + d] = row[0] * scale;
    ks[t * qld + d] = row[C];
    vs[t * hd + d] = row[2 * C];
  }
  for (int i = tid; i < 2 * n; i += 256) {
    const int t = i >> 1, k = t / win_w, ww = t - k * win_w;
    tp[i] = kpos[(((size_t)b * K + k) * W + j * win_w + ww) * 2 + (i & 1)];
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += 256) {
    const int qi = e / n, kj = e - qi * n;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc += qs[qi * qld + d] * ks[kj * qld + d];
    float hid[16];
    pm.hidden(tp[2 * qi] - tp[2 * kj], tp[2 * qi + 1] - tp[2 * kj + 1], hid);
    ps[qi * pld + kj] = acc + pm.out(hid, head);
  }
  __syncthreads();
  // softmax: eight consecutive lanes per row
  for (int r = tid >> 3; r < n; r += 32) {
    const int sub = tid & 7;
    float m = -FLT_MAX;
    for (int kj = sub; kj < n; kj += 8) m = fmaxf(m, ps[r * pld + kj]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    m = fmaxf(m, __shfl_xor(m, 4, 64));
    float sum = 0.f;
    for (int kj = sub; kj < n; kj += 8) {
      const float e = expf(ps[r * pld + kj] - m);
      ps[r * pld + kj] = e;
      sum += e;
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    const float inv = 1.f / sum;
    for (int kj = sub; kj < n; kj += 8) ps[r * pld + kj] *= inv;
  }
  __syncthreads();
  const int hd4 = hd >> 2;
  for (int i = tid; i < n * hd4; i += 256) {
    const int t = i / hd4, q4 = i - t * hd4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kj = 0; kj < n; ++kj) acc += ps[t * pld + kj] * *reinterpret_cast<const f32x4*>(vs + kj * hd + q4 * 4);
    *reinterpret_cast<f32x4*>(out + trow(t) * C + head * hd + q4 * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// SectorAttentionV2 core (azimuth column <- its key points).  q: (tokens, C) projection of the tokens;
// k|v: (B, K*W, 2C) projection of the key points, each half read through the raw (B,C,K,W)
// reinterpretation (set_transformer.py:417-425).  Block = (batch, rolled column).
// out: (tokens, C) written at the PHYSICAL column (the roll-back of set_transformer.py:157-160).
// r3: four rows per wave instruction (lane = (row mod 4, 4-channel group), one head per step: 256 contiguous bytes per row in and out),
// 16-lane reductions for the four logits of a (row, head), the relative-position bias of every (row, key point, head) formed once up front.
template <int KT>
__global__ __launch_bounds__(1024) void sector_col_attn_kernel(const float* __restrict__ q, const float* __restrict__ kvraw,
                                                               const float* __restrict__ xpos, const float* __restrict__ kpos,
                                                               PosMlp pm, int B, int H, int W, int C, int K, int shift, int cm,
                                                               float scale, float* __restrict__ out) {
  extern __shared__ float lds[];  // kk[K][C], vv[K][C], bias[H][K][heads]
  const int heads = pm.heads, hd = C / heads;
  const int b = blockIdx.x / W, wr = blockIdx.x % W, wp = (wr + shift) % W;
  float* kk = lds;
  float* vv = lds + K * C;
  float* bt = vv + K * C;
  const float* kvb = kvraw + (size_t)b * K * W * (2 * C);
  // proj_k and proj_v outputs are separate (K*W, C) tensors in the reference; here they are the two column halves of one
  // (K*W, 2C) GEMM output, so element [row][c] of proj_k is kvb[row*2C + c].  The reinterpretation (B, C, K, W) addresses the FLAT
  // (K*W*C) proj_k buffer: flat = ((c*K + k)*W + w)  ->  row = flat / C, col = flat % C.
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    const int k = i / C, c = i - k * C;
    const unsigned flat = ((unsigned)c * K + k) * W + wr;
    const unsigned row = flat / (unsigned)C, col = flat - row * C;
    kk[i] = kvb[(size_t)row * (2 * C) + col];
    vv[i] = kvb[(size_t)row * (2 * C) + C + col];
  }
  for (int i = threadIdx.x; i < H * K; i += blockDim.x) {
    const int h = i / K, k = i - h * K;
    const float px = xpos[((size_t)h * W + wp) * 2], py = xpos[((size_t)h * W + wp) * 2 + 1];
    const float* kp2 = kpos + (((size_t)b * K + k) * W + wr) * 2;
    float hid[16];
    pm.hidden(px - kp2[0], py - kp2[1], hid);
    for (int head = 0; head < heads; ++head) bt[(h * K + k) * heads + head] = pm.out(hid, head);
  }
  __syncthreads();
  // sixteen waves per column (the kernel is latency-, not byte-bound: r2's one lane per row took 52 us for 75 MB): every wave takes the
  // row groups g = wv, wv + 16, ... (4 rows each: lane = (row mod 4, 4-channel group), 256 contiguous bytes per row and head) with the
  // q pieces of all heads (HT = 4 per pass) in flight before the first is used
  const int nwv = blockDim.x >> 6;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, rs = lane >> 4, d4 = lane & 15;
  const bool dok = d4 * 4 < hd;
  const size_t tstride = cm ? 1 : (size_t)W;
  const size_t t0 = tok(b, 0, wp, H, W, cm);
  constexpr int HT = 4;
  for (int hp = 0; hp < heads; hp += HT) {
    for (int h0 = 4 * wv; h0 < H; h0 += 4 * nwv) {
      const int h = h0 + rs;
      const int hc = min(h, H - 1);
      const bool ok = h < H && dok;
      const float* qr = q + (t0 + (size_t)hc * tstride) * C + (dok ? d4 * 4 : 0);
      float* orow = out + (t0 + (size_t)hc * tstride) * C;
      f32x4 qq[HT];
#pragma unroll
      for (int hh = 0; hh < HT; ++hh) qq[hh] = *reinterpret_cast<const f32x4*>(qr + min(hp + hh, heads - 1) * hd);
#pragma unroll
      for (int hh = 0; hh < HT; ++hh) {
        const int head = hp + hh;
        if (head >= heads) continue;
        const int co = head * hd + (dok ? d4 * 4 : 0);
        float lg[KT];
        float m = -FLT_MAX;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k >= K) continue;
          const float sum = group16_sum(dok ? dot4(qq[hh], *reinterpret_cast<const f32x4*>(kk + k * C + co)) : 0.f);
          lg[k] = sum * scale + bt[(hc * K + k) * heads + head];
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
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k >= K) continue;
          o += (lg[k] * inv) * *reinterpret_cast<const f32x4*>(vv + k * C + co);
        }
        if (ok) *reinterpret_cast<f32x4*>(orow + co) = o;
      }
    }
  }
}

}  // namespace

extern "C" {

static int layernorm_run(const float* x, size_t rows, int c, const float* gamma, const float* beta, float eps, float* out, float* chan_mean,
                         unsigned short* out16, pn_stream_t stream) {
  if (rows == 0) return PN_OK;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = pn::S(stream);
  switch (c % 64 == 0 ? c / 64 : 0) {      // register-resident rows for the channel counts of the model (64 .. 1024)
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, x, rows, c, gamma, beta, eps, out, chan_mean, out16); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, x, rows, c, gamma, beta, eps, out, chan_mean, out16); break;
    case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, x, rows, c, gamma, beta, eps, out, chan_mean, out16); break;
    case 8: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, st, x, rows, c, gamma, beta, eps, out, chan_mean, out16); break;
    case 16: hipLaunchKernelGGL(layernorm_kernel<16>, grid, block, 0, st, x, rows, c, gamma, beta, eps, out, chan_mean, out16); break;
    default: hipLaunchKernelGGL(layernorm_kernel<0>, grid, block, 0, st, x, rows, c, gamma, beta, eps, out, chan_mean, out16); break;
  }
  return pn::check_launch("layernorm_kernel");
}

int pn_layernorm_f32(const float* x, size_t rows, int c, const float* gamma, const float* beta, float eps, float* out,
                     float* chan_mean, pn_stream_t stream) {
  PN_REQUIRE(x && gamma && beta && out && c >= 1, "layernorm: bad arguments");
  return layernorm_run(x, rows, c, gamma, beta, eps, out, chan_mean, nullptr, stream);
}

// the same rows with a bf16 copy of the result for the bf16 GEMMs (pn_linear_bf16); out_f32 may be null when only the copy is consumed
int pn_layernorm_bf16out_f32(const float* x, size_t rows, int c, const float* gamma, const float* beta, float eps, float* out_f32, void* out_bf16,
                             float* chan_mean, pn_stream_t stream) {
  PN_REQUIRE(x && gamma && beta && out_bf16 && c >= 1, "layernorm_bf16out: bad arguments");
  return layernorm_run(x, rows, c, gamma, beta, eps, out_f32, chan_mean, static_cast<unsigned short*>(out_bf16), stream);
}

int pn_setblock_keypoints(const float* chan_mean, const float* xn, const float* pos, int batch, int h, int w, int c, int k,
                          int shift, int col_major, int32_t* top_idx, float* kp, float* kpos, pn_stream_t stream) {
  PN_REQUIRE(chan_mean && xn && pos && top_idx && kp && kpos, "keypoints: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 3 && w >= 1 && c >= 1 && k >= 1 && k <= h && k <= 8, "keypoints: bad sizes");
  hipLaunchKernelGGL(keypoints_kernel, dim3(batch * w), dim3(64), h * sizeof(float), pn::S(stream), chan_mean, xn, pos, batch,
                     h, w, c, k, shift, col_major != 0, top_idx, kp, kpos);
  return pn::check_launch("keypoints_kernel");
}

int pn_setblock_sector_kp_attn(const float* q_raw, const float* kv, const float* xpos, const float* kpos, const float* pos_mlp,
                               int batch, int h, int w, int c, int heads, int k, int shift, int col_major, float scale, float* out,
                               pn_stream_t stream) {
  PN_REQUIRE(q_raw && kv && xpos && kpos && pos_mlp && out, "sector_kp_attn: null pointer");
  PN_REQUIRE(heads >= 1 && heads <= 4 && c % heads == 0 && (c / heads) % 4 == 0 && c / heads <= 64 && k <= 8, "sector_kp_attn: bad sizes (head width <= 64)");
  PN_REQUIRE(((uintptr_t)kv & 15) == 0 && ((uintptr_t)out & 15) == 0 && c % 4 == 0, "sector_kp_attn: 16-byte aligned buffers");
  const size_t smem = ((size_t)heads * k * (c / heads) * (1 + SKP_CH) + (size_t)heads * k * h) * sizeof(float);
  PN_REQUIRE(smem <= 64 * 1024, "sector_kp_attn: column too long for LDS");
  PosMlp pm{pos_mlp, heads};
  if (k <= 4)
    hipLaunchKernelGGL(sector_kp_attn_kernel<4>, dim3(batch * w), dim3(1024), smem, pn::S(stream), q_raw, kv, xpos, kpos, pm, batch,
                       h, w, c, k, shift, col_major != 0, scale, out);
  else
    hipLaunchKernelGGL(sector_kp_attn_kernel<8>, dim3(batch * w), dim3(1024), smem, pn::S(stream), q_raw, kv, xpos, kpos, pm, batch,
                       h, w, c, k, shift, col_major != 0, scale, out);
  return pn::check_launch("sector_kp_attn_kernel");
}

int pn_setblock_range_attn(const float* qkv, const float* kpos, const float* pos_mlp, int batch, int w, int c, int heads, int k,
                           int win_w, float scale, float* out, pn_stream_t stream) {
  PN_REQUIRE(qkv && kpos && pos_mlp && out, "range_attn: null pointer");
  PN_REQUIRE(heads >= 1 && heads <= 4 && c % heads == 0 && w % win_w == 0 && (c / heads) % 4 == 0 && (k * win_w) % 4 == 0, "range_attn: bad sizes");
  PN_REQUIRE(((uintptr_t)out & 15) == 0 && c % 4 == 0, "range_attn: 16-byte aligned output");
  const int n = k * win_w, hd = c / heads;
  const size_t smem = ((size_t)2 * n * (hd + 1) + (size_t)n * hd + (size_t)n * (n + 1) + 2 * n) * sizeof(float);
  PN_REQUIRE(smem <= 160 * 1024, "range_attn: window too large for LDS");
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&range_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  }
  PosMlp pm{pos_mlp, heads};
  hipLaunchKernelGGL(range_attn_kernel, dim3(batch * (w / win_w) * heads), dim3(256), smem, pn::S(stream), qkv, kpos, pm, batch, w, c, k,
                     win_w, scale, out);
  return pn::check_launch("range_attn_kernel");
}

int pn_setblock_sector_col_attn(const float* q, const float* kv_raw, const float* xpos, const float* kpos, const float* pos_mlp,
                                int batch, int h, int w, int c, int heads, int k, int shift, int col_major, float scale, float* out,
                                pn_stream_t stream) {
  PN_REQUIRE(q && kv_raw && xpos && kpos && pos_mlp && out, "sector_col_attn: null pointer");
  PN_REQUIRE(heads >= 1 && c % heads == 0 && (c / heads) % 4 == 0 && c / heads <= 64 && k <= 8 && c % 4 == 0, "sector_col_attn: bad sizes (head width <= 64)");
  PN_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)out & 15) == 0, "sector_col_attn: 16-byte aligned buffers");
  const size_t smem = ((size_t)2 * k * c + (size_t)h * k * heads) * sizeof(float);
  PN_REQUIRE(smem <= 64 * 1024, "sector_col_attn: column too long for LDS");
  PosMlp pm{pos_mlp, heads};
  if (k <= 4)
    hipLaunchKernelGGL(sector_col_attn_kernel<4>, dim3(batch * w), dim3(1024), smem, pn::S(stream), q, kv_raw, xpos, kpos, pm, batch,
                       h, w, c, k, shift, col_major != 0, scale, out);
  else
    hipLaunchKernelGGL(sector_col_attn_kernel<8>, dim3(batch * w), dim3(1024), smem, pn::S(stream), q, kv_raw, xpos, kpos, pm, batch,
                       h, w, c, k, shift, col_major != 0, scale, out);
  return pn::check_launch("sector_col_attn_kernel");
}

}  // extern "C"
