// Window attention of the geometry-aware head.
// Reference (not executable there; this follows the repaired restatement in
// oracle/polar_oracle.py::swv_window_attention):
//   WindowAttention.forward        det3d/models/bbox_heads/swin_utils/sw2votev4_util.py:65-103
//   SwinTransformerBlock.forward   sw2votev4_util.py:127-188  (zero padding to window multiples, cyclic
//                                  shift, window partition / reverse, shift mask of BasicLayer :259-276)
// Per (window, head): q, k, v head slices of the 49 tokens (+ the vote embedding, a 3->16->C MLP of (pred_centers, vote_cls) added
// to all three),
//   s[i][j] = <q_i, k_j> / max(|q_i||k_j|, 1e-6) / max(tau, 0.01) + rpe(pos_i - pos_j) + mask(i, j)
// softmax over j, out_i = sum_j p_j v_j -- the products on the fp32 MFMA, see the kernel.
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

constexpr int WS = 7, NT = WS * WS, HD = 64;
#ifndef PN_SWV_EXP
#define PN_SWV_EXP 0   // diagnostic builds only: 1 no q/k/v loads, 2 no bias-table loads, 4 cheap logits (no division / exp), 8 no vote-embedding MFMAs
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// r3: the attention of one window on the fp32 MFMA.  Block = one window, wave = head (r2: one 64-thread block per (window, head), lane =
// query, everything on the VALU with 64-long dot products against LDS rows: 632 us per block of the Waymo map, 1.26 ms of the frame).
// The 49 tokens are padded to 64 rows (two 32-row MFMA tiles); every product is an exact-fp32 MFMA chain:
//   vote embedding   ve = vhid (64 x 16) W2^T (16 x 64), computed in BOTH orientations (lane = token / lane = channel: the same two
//                    operand fragments, swapped) so that it can be added to operands of either layout without a transpose
//   S^T = K Q^T      the TRANSPOSED logits: accumulator lane = query, registers = keys, so the softmax over the keys is a reduction
//                    over a lane's own registers plus ONE exchange with the lane holding the other half of the keys
//   out^T = V^T P^T  P^T is S^T's accumulator as it stands (B operand: lane = query, k = key); V^T comes from global memory with
//                    lane = channel; the accumulator (lane = query, registers = 4 consecutive channels x 4) is stored as float4
// k order inside every MFMA: step (register r) multiplies index rowmap(r, 0) from lane half 0 and rowmap(r, 1) from lane half 1, the
// same for both operands (rowmap(r, h) = (r & 3) + 8 (r >> 2) + 4 h, the accumulator's row map: what makes P^T usable in place).
// The relative-position MLP (2 -> 16 -> heads, per (query, key) pair) and the softmax stay on the VALU: ~3.5 k instructions per lane.
__device__ __forceinline__ int rowmap(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

//
// r5: the relative-position bias depends on the cell positions, the window frame and the rpe weights only -- not on the frame's features --
// and its hidden layer not on the head: TAB = true reads it from a table built once per (weights, map, shift) by swv_window_bias_kernel in
// THIS kernel's accumulator layout ([window][head][query tile][28 live key registers][64 lanes]: a register's 64 values are one 256-byte
// line), instead of ~64 VALU instructions per (query, key, head).  The table holds the values the on-the-fly form computes, bit for bit.
constexpr int kBiasRegs = 28;       // key registers per query tile: 16 of key tile 0 + the 12 of key tile 1 that are tokens (keys < 56)
__host__ __device__ constexpr size_t swv_bias_floats_per_window_head() { return 2 * kBiasRegs * 64; }

template <bool TAB>
__global__ __launch_bounds__(256) void swv_window_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ vote, int vote_ps,
                                                              const float* __restrict__ pos, SwvParams P, int H, int W, int C, int heads, int shift,
                                                              float* __restrict__ out, const float* __restrict__ bias_tab) {
  __shared__ int tok[64];
  __shared__ __attribute__((aligned(16))) float tinfo[64][4];      // px, py, region, key-valid (1 / 0)
  __shared__ __attribute__((aligned(16))) float vhid[64][20];      // hidden layer of the vote MLP per token (16 + pad)
  __shared__ float nkw[4][64];
  const int tid = threadIdx.x, lane = tid & 63, head = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
  const int nww = Wp / WS;
  const int wy = blockIdx.x / nww, wx = blockIdx.x - wy * nww;
  const int b = blockIdx.y;
  // ---- token bookkeeping: thread t < 64 (rows >= 49 are MFMA padding: never keys, never written)
  if (tid < 64) {
    const int t = tid;
    const bool row = t < NT;
    const int r = t / WS, c = t - r * WS;
    const int hs = wy * WS + r, wsx = wx * WS + c;                 // coordinates in the (shifted) window frame
    const int hp = (hs + shift) % Hp, wp = (wsx + shift) % Wp;     // padded-map coordinates of this token
    const bool valid = row && hp < H && wp < W;
    const int tk = valid ? (b * H + hp) * W + wp : -1;
    tok[t] = tk;
    const int ih = hs < Hp - WS ? 0 : (hs < Hp - shift ? 1 : 2), iw = wsx < Wp - WS ? 0 : (wsx < Wp - shift ? 1 : 2);
    tinfo[t][0] = valid ? pos[(hp * W + wp) * 2] : 0.f;
    tinfo[t][1] = valid ? pos[(hp * W + wp) * 2 + 1] : 0.f;
    tinfo[t][2] = (float)((row && shift > 0) ? ih * 3 + iw : 0);
    tinfo[t][3] = row ? 1.f : 0.f;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (valid) {
      const float* vp = vote + (size_t)tk * vote_ps;
      v0 = vp[0]; v1 = vp[1]; v2 = vp[2];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float hsum = P.vm_b1[m] + P.vm_w1[m * 3] * v0 + P.vm_w1[m * 3 + 1] * v1 + P.vm_w1[m * 3 + 2] * v2;
      vhid[t][m] = hsum > 0.f ? hsum : 0.f;
    }
  }
  __syncthreads();
  if (head >= heads) return;
  const int hc = head * HD;
  // ---- operand fragments of the vote MLP's second layer: k = m (16): step s multiplies m = 2 s + lh
  float w2f[2][8], vhf[2][8];       // [channel tile / token tile][step]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      w2f[tt][s8] = P.vm_w2[(hc + tt * 32 + li) * 16 + 2 * s8 + lh];
      vhf[tt][s8] = vhid[tt * 32 + li][2 * s8 + lh];
    }
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- K and Q operands: lane = token (tile tt), registers = channels rowmap(r, lh) of channel tile ct.  x + bias + ve.
  f32x16 kreg[2][2], qreg[2][2];
  float nq[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int tk = tok[tt * 32 + li];
    const float* row = qkv + (size_t)(tk >= 0 ? tk : 0) * 3 * C + hc;
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x16 ve = zero16;                // ve^T tile: rows = channels of tile ct, columns = tokens of tile tt
#pragma unroll
      for (int s8 = 0; s8 < ((PN_SWV_EXP & 8) ? 0 : 8); ++s8) ve = __builtin_amdgcn_mfma_f32_32x32x2f32(w2f[ct][s8], vhf[tt][s8], ve, 0, 0, 0);
      f32x16 q, k;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = ct * 32 + 8 * g + 4 * lh;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f}, kv = {0.f, 0.f, 0.f, 0.f};
        if (tk >= 0 && !(PN_SWV_EXP & 1)) {
          qv = *reinterpret_cast<const f32x4*>(row + ch);
          kv = *reinterpret_cast<const f32x4*>(row + C + ch);
        }
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(P.vm_b2 + hc + ch);
        f32x4 bq = {0.f, 0.f, 0.f, 0.f}, bk = {0.f, 0.f, 0.f, 0.f};
        if (P.qkv_bias) {
          bq = *reinterpret_cast<const f32x4*>(P.qkv_bias + hc + ch);
          bk = *reinterpret_cast<const f32x4*>(P.qkv_bias + C + hc + ch);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = ve[4 * g + j] + b2[j];
          const float qq = (tk >= 0 ? qv[j] : bq[j]) + e, kk = (tk >= 0 ? kv[j] : bk[j]) + e;
          q[4 * g + j] = qq;
          k[4 * g + j] = kk;
          sq = fmaf(qq, qq, sq);
          sk = fmaf(kk, kk, sk);
        }
      }
      qreg[tt][ct] = q;
      kreg[tt][ct] = k;
    }
    sq += __shfl_xor(sq, 32, 64);
    sk += __shfl_xor(sk, 32, 64);
    nq[tt] = sqrtf(sq);
    if (lh == 0) nkw[head][tt * 32 + li] = sqrtf(sk);
  }
  // ---- V^T operand: lane = channel (tile ct), registers = tokens rowmap(r, lh) of token tile tj.  v + bias + ve.
  f32x16 vreg[2][2];     // [token tile][channel tile]
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x16 ve = zero16;                // ve tile: rows = tokens of tile tj, columns = channels of tile ct
#pragma unroll
      for (int s8 = 0; s8 < ((PN_SWV_EXP & 8) ? 0 : 8); ++s8) ve = __builtin_amdgcn_mfma_f32_32x32x2f32(vhf[tj][s8], w2f[ct][s8], ve, 0, 0, 0);
      const int ch = hc + ct * 32 + li;
      const float b2 = P.vm_b2[ch], bv = P.qkv_bias ? P.qkv_bias[2 * C + ch] : 0.f;
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tk = tok[tj * 32 + rowmap(r, lh)];
        const float x = (tk >= 0 && !(PN_SWV_EXP & 1)) ? qkv[(size_t)tk * 3 * C + 2 * C + ch] : bv;
        v[r] = x + ve[r] + b2;
      }
      vreg[tj][ct] = v;
    }
  // relative-position MLP and temperature
  float rw1x[16], rw1y[16], rb1[16], rw2[16];
  float rb2 = 0.f;
  if constexpr (!TAB) {
#pragma unroll
    for (int m = 0; m < 16; ++m) { rw1x[m] = P.rp_w1[m * 2]; rw1y[m] = P.rp_w1[m * 2 + 1]; rb1[m] = P.rp_b1[m]; rw2[m] = P.rp_w2[head * 16 + m]; }
    rb2 = P.rp_b2[head];
  }
  const float* tab = TAB ? bias_tab + ((size_t)blockIdx.x * heads + head) * swv_bias_floats_per_window_head() + lane : nullptr;
  const float inv_tau = 1.f / fmaxf(P.tau[head], 0.01f);
  // ---- per query tile: S^T = K Q^T (lane = query), logits, softmax over the keys, out^T = V^T P^T
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int qi = ti * 32 + li;
    float rpt[kBiasRegs];        // TAB: this query tile's biases, requested before the products
    if constexpr (TAB) {
#pragma unroll
      for (int k = 0; k < kBiasRegs; ++k) rpt[k] = (PN_SWV_EXP & 2) ? 0.01f * k : tab[(ti * kBiasRegs + k) * 64];
    }
    f32x16 st[2] = {zero16, zero16};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[tj][ct][r], qreg[ti][ct][r], st[tj], 0, 0, 0);
    const f32x4 me = *reinterpret_cast<const f32x4*>(tinfo[qi]);
    const float nqi = nq[ti];
    float smax = -3.0e38f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tj * 32 + rowmap(r, lh);
        if (tj == 1 && (r >> 2) > 2) { st[tj][r] = -3.0e38f; continue; }     // keys >= 56: MFMA padding (compile-time skip)
        const f32x4 oj = *reinterpret_cast<const f32x4*>(tinfo[j]);
        float a = (PN_SWV_EXP & 4) ? st[tj][r] * inv_tau : st[tj][r] / fmaxf(nqi * nkw[head][j], 1e-6f) * inv_tau;
        float rp = rb2;
        if constexpr (TAB) {
          rp = rpt[tj * 16 + r];
        } else {
          const float dx = me[0] - oj[0], dy = me[1] - oj[1];
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            const float hdn = fmaf(rw1x[m], dx, fmaf(rw1y[m], dy, rb1[m]));
            rp = fmaf(rw2[m], hdn > 0.f ? hdn : 0.f, rp);
          }
        }
        a += rp;
        if (oj[2] != me[2]) a += -100.f;
        if (oj[3] == 0.f) a = -3.0e38f;            // rows 49 .. 55: not a token
        st[tj][r] = a;
        smax = fmaxf(smax, a);
      }
    smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = (tj == 1 && (r >> 2) > 2) ? 0.f : ((PN_SWV_EXP & 4) ? st[tj][r] - smax : expf(st[tj][r] - smax));
        st[tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[tj][r] *= inv;
    const int tk = tok[qi];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x16 o = zero16;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (tj == 1 && (r >> 2) > 2) continue;      // P is zero there
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[tj][ct][r], st[tj][r], o, 0, 0, 0);
        }
      if (tk >= 0) {
        float* op = out + (size_t)tk * C + hc + ct * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(op + 8 * g) = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
      }
    }
  }
}

// the bias table of swv_window_attn_kernel<true>: block = window of the (shifted, padded) frame, wave = head; the arithmetic of the
// on-the-fly form, term for term (positions of padded tokens are zero there too)
__global__ __launch_bounds__(256) void swv_window_bias_kernel(const float* __restrict__ pos, SwvParams P, int H, int W, int heads, int shift,
                                                              float* __restrict__ tab) {
  __shared__ float px[64], py[64];
  const int tid = threadIdx.x, lane = tid & 63, head = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int Hp = (H + WS - 1) / WS * WS, Wp = (W + WS - 1) / WS * WS;
  const int nww = Wp / WS;
  const int wy = blockIdx.x / nww, wx = blockIdx.x - wy * nww;
  if (tid < 64) {
    const int t = tid;
    const int r = t / WS, c = t - r * WS;
    const int hp = (wy * WS + r + shift) % Hp, wp = (wx * WS + c + shift) % Wp;
    const bool valid = t < NT && hp < H && wp < W;
    px[t] = valid ? pos[(hp * W + wp) * 2] : 0.f;
    py[t] = valid ? pos[(hp * W + wp) * 2 + 1] : 0.f;
  }
  __syncthreads();
  if (head >= heads) return;
  float rw1x[16], rw1y[16], rb1[16], rw2[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) { rw1x[m] = P.rp_w1[m * 2]; rw1y[m] = P.rp_w1[m * 2 + 1]; rb1[m] = P.rp_b1[m]; rw2[m] = P.rp_w2[head * 16 + m]; }
  const float rb2 = P.rp_b2[head];
  float* dst = tab + ((size_t)blockIdx.x * heads + head) * swv_bias_floats_per_window_head() + lane;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const float mx = px[ti * 32 + li], my = py[ti * 32 + li];
#pragma unroll 1
    for (int k = 0; k < kBiasRegs; ++k) {
      const int tj = k >> 4, r = k & 15;
      const int j = tj * 32 + rowmap(r, lh);
      const float dx = mx - px[j], dy = my - py[j];
      float rp = rb2;
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const float hdn = fmaf(rw1x[m], dx, fmaf(rw1y[m], dy, rb1[m]));
        rp = fmaf(rw2[m], hdn > 0.f ? hdn : 0.f, rp);
      }
      dst[(ti * kBiasRegs + k) * 64] = rp;
    }
  }
}

}  // namespace

extern "C" {

size_t pn_swv_window_bias_floats(int h, int w, int heads, int window) {
  if (window != WS || h < 1 || w < 1 || heads < 1 || heads > 4) return 0;
  return (size_t)((h + WS - 1) / WS) * ((w + WS - 1) / WS) * heads * swv_bias_floats_per_window_head();
}

int pn_swv_window_bias_table(const float* pos, const float* rpe_w1, const float* rpe_b1, const float* rpe_w2, const float* rpe_b2, int h, int w, int heads,
                             int window, int shift, float* table, pn_stream_t stream) {
  PN_REQUIRE(pos && rpe_w1 && rpe_b1 && rpe_w2 && rpe_b2 && table, "swv_window_bias_table: null pointer");
  PN_REQUIRE(window == WS && heads >= 1 && heads <= 4 && shift >= 0 && shift < WS && h >= 1 && w >= 1, "swv_window_bias_table: window 7, at most 4 heads");
  SwvParams p{nullptr, nullptr, nullptr, nullptr, nullptr, rpe_w1, rpe_b1, rpe_w2, rpe_b2, nullptr};
  const int nwh = (h + WS - 1) / WS, nww = (w + WS - 1) / WS;
  hipLaunchKernelGGL(swv_window_bias_kernel, dim3(nwh * nww), dim3(256), 0, pn::S(stream), pos, p, h, w, heads, shift, table);
  return pn::check_launch("swv_window_bias_kernel");
}

int pn_swv_window_attn(const float* qkv, const float* vote, int vote_pixel_stride, const float* pos, const float* qkv_bias,
                       const float* vote_w1, const float* vote_b1, const float* vote_w2, const float* vote_b2, const float* rpe_w1,
                       const float* rpe_b1, const float* rpe_w2, const float* rpe_b2, const float* tau, int batch, int h, int w, int c,
                       int heads, int window, int shift, const float* bias_table, float* out, pn_stream_t stream) {
  PN_REQUIRE(qkv && vote && pos && vote_w1 && vote_b1 && vote_w2 && vote_b2 && tau && out, "swv_window_attn: null pointer");
  PN_REQUIRE(bias_table || (rpe_w1 && rpe_b1 && rpe_w2 && rpe_b2), "swv_window_attn: the rpe weights or their bias table");
  PN_REQUIRE(window == WS && heads >= 1 && heads <= 4 && c == heads * HD, "swv_window_attn: built for window 7, head_dim 64 and at most 4 heads");
  PN_REQUIRE(shift >= 0 && shift < WS && batch >= 1 && h >= 1 && w >= 1 && vote_pixel_stride >= 3, "swv_window_attn: bad sizes");
  PN_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)vote_b2 & 15) == 0 && ((uintptr_t)qkv_bias & 15) == 0,
             "swv_window_attn: qkv, out, the vote bias and the qkv bias must be 16-byte aligned");
  SwvParams p{qkv_bias, vote_w1, vote_b1, vote_w2, vote_b2, rpe_w1, rpe_b1, rpe_w2, rpe_b2, tau};
  const int nwh = (h + WS - 1) / WS, nww = (w + WS - 1) / WS;
  if (bias_table)
    hipLaunchKernelGGL(swv_window_attn_kernel<true>, dim3(nwh * nww, batch), dim3(256), 0, pn::S(stream), qkv, vote, vote_pixel_stride, pos, p, h, w, c,
                       heads, shift, out, bias_table);
  else
    hipLaunchKernelGGL(swv_window_attn_kernel<false>, dim3(nwh * nww, batch), dim3(256), 0, pn::S(stream), qkv, vote, vote_pixel_stride, pos, p, h, w, c,
                       heads, shift, out, bias_table);
  return pn::check_launch("swv_window_attn_kernel");
}

}  // extern "C"
