// Backward of the BEV convolutions (training step).
//
//   data gradient    -- no new kernel: dX of a stride-1 convolution is a stride-1 convolution of dY
//                       with the taps mirrored and Cin/Cout swapped; dX of the stride-2 3x3
//                       convolutions (ZeroPad2d(1) + Conv2d(3, stride 2), rpn.py:126-134) is a
//                       4-phase 2x2-tap "sub-pixel" convolution of dY.  Both run on
//                       conv_mfma_kernel; this file only holds the weight packers for them.
//   weight gradient  -- dW[tap][ci][co] = sum_m X[pixel(m) + tap][ci] * dY[m][co]
//                       an MFMA GEMM whose reduction dimension is the output-pixel index m:
//                         rows    = input channels of one tap   (A = X^T, read from the NHWC map)
//                         columns = output channels             (B = dY)
//                         k       = 32 output pixels per step
//                       The pixel range is cut into `splits` slices (one block each) so that a few
//                       hundred blocks exist even when Cin x Cout x taps is a handful of tiles; the
//                       slices are summed in a fixed order by a second kernel (bitwise
//                       reproducible, no float atomics), which also writes torch's
//                       (Cout, Cin, KH, KW) layout.
//
// LDS images (per stage): A [32 pixels][BM + 8], B [32 pixels][BN + 8] floats.  MFMA operand
// fetch is one ds_read_b32 per k: lane (i = l & 31, h = l >> 5) reads row k = 8s + 4h + j,
// column i -- 32 consecutive floats per half, and 4 * (BM + 8) = 32 (mod 64) banks puts the two
// halves on disjoint banks.
#include "pn_common.h"
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// unsigned division by a launch-time constant without a divide or a branch (Granlund-Montgomery):
// q = (t + ((n - t) >> s1)) >> s2 with t = umulhi(m, n)
struct FastDiv {
  unsigned m, s1, s2;
};
inline FastDiv make_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  FastDiv f;
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, FastDiv f) {
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct WgradArgs {
  const float* in;
  const float* dy;
  float* part;
  int B, H, W, Cin, Cout, OH, OW;
  int KH, KW, stride, pad_h, pad_w;
  int in_ps, in_co, dy_ps, dy_co;
  int M;            // B * OH * OW
  int m_per_split;  // multiple of 32
  int ci_tiles, co_tiles, tiles_per_split, splits;
  int cin_pad, cout_pad;
  unsigned in_bytes, dy_bytes;
  FastDiv div_ohw, div_ow;
  // gather mode (sparse convolution, scn.py:97-192 under autograd): GEMM k index m = output site, tap t reads input row
  // nbr[m * taps + t] (-1: inactive); n_valid = device count of live output sites
  const int* nbr;
  const int* n_valid;
  // range-stratified convolution (STRAT): blockIdx.y = stratum, the pixel index runs over the stratum's OWsub columns of every row
  int OWsub;        // OW unless stratified
};

constexpr int WK = 32;  // pixels per K step


template <int TM, int TN, bool GATHER, bool STRAT = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int BM = TM * 64, BN = TN * 64;
  constexpr int LDA = BM + 8, LDB = BN + 8;
  constexpr int STAGE = WK * (LDA + LDB);
  constexpr int A_PER_T = BM / 32, B_PER_T = BN / 32;  // float4 per thread and step
  constexpr int A_ROWS = 256 / (BM / 4), B_ROWS = 256 / (BN / 4);  // pixels covered by one pass of the block
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  // Block order: all (tap, ci tile, co tile) blocks of one pixel slice are neighbours in dispatch order AND
  // on the same XCD (blocks are dealt round-robin over the 8 XCDs): they read the same dY rows and
  // overlapping X rows, which then come from that XCD's L2 instead of HBM (9 taps = 9x less HBM traffic).
  const int per_split = a.tiles_per_split;                 // taps * ci_tiles * co_tiles
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int split = (kq / per_split) * 8 + xcd;
  if (split >= a.splits) return;
  const int z = STRAT ? (int)blockIdx.y : 0;
  int t = kq - (kq / per_split) * per_split;
  const int cot = t % a.co_tiles; t /= a.co_tiles;
  const int cit = t % a.ci_tiles;
  const int tap = t / a.ci_tiles;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int ci0 = cit * BM, co0 = cot * BN;
  const int m_begin = split * a.m_per_split;
  int m_end = min(a.M, m_begin + a.m_per_split);
  if constexpr (GATHER) m_end = min(m_end, *a.n_valid);
  const int nsteps = m_end > m_begin ? (m_end - m_begin + WK - 1) / WK : 0;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);

  // this thread's pixels: A rows pa + A_ROWS*j, B rows pb + B_ROWS*j of every 32-pixel step
  const int ca = tid % (BM / 4), pa = tid / (BM / 4);
  const int cb = tid % (BN / 4), pb = tid / (BN / 4);
  const unsigned a_chan = (unsigned)(a.in_co + ci0 + ca * 4);
  const bool a_cok = ci0 + ca * 4 < a.Cin;
  const unsigned b_chan = (unsigned)(a.dy_co + co0 + cb * 4);
  const bool b_cok = co0 + cb * 4 < a.Cout;

  f32x4 ra[A_PER_T], rb[B_PER_T];
  int ld_m = m_begin;  // first pixel of the step being loaded
  auto load_global = [&]() {
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      // (b, oh, ow) of output pixel m by multiply-shift division: branch-free, no per-row state
      const unsigned m = (unsigned)(ld_m + pa + A_ROWS * j);
      if constexpr (GATHER) {
        const int idx = ((int)m < m_end) ? a.nbr[(size_t)m * (a.KH * a.KW) + tap] : -1;
        const unsigned vo = (a_cok && idx >= 0) ? ((unsigned)idx * (unsigned)a.in_ps + a_chan) * 4u : 0xffffffffu;
        ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, 0, 0));
        continue;
      }
      const unsigned b = fast_div(m, a.div_ohw);
      const unsigned rem = m - b * (unsigned)(a.OH * a.OWsub);
      const unsigned oh = fast_div(rem, a.div_ow);
      const unsigned ow = rem - oh * (unsigned)a.OWsub + (unsigned)(z * a.OWsub);
      const int ih = (int)oh * a.stride - a.pad_h + kh, iw = (int)ow * a.stride - a.pad_w + kw;
      const bool ok = a_cok && ((int)m < m_end) && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
      const unsigned pix = (b * (unsigned)a.H + (unsigned)ih) * (unsigned)a.W + (unsigned)iw;  // < 2^29 pixels (2 GiB map)
      const unsigned vo = ok ? (pix * (unsigned)a.in_ps + a_chan) * 4u : 0xffffffffu;
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) {
      const int m = ld_m + pb + B_ROWS * j;
      unsigned pix = (unsigned)m;
      if constexpr (STRAT) {   // pixel m of the stratum -> its place in the full-width map
        const unsigned b = fast_div(pix, a.div_ohw);
        const unsigned rem = pix - b * (unsigned)(a.OH * a.OWsub);
        const unsigned oh = fast_div(rem, a.div_ow);
        pix = (b * (unsigned)a.OH + oh) * (unsigned)a.OW + rem - oh * (unsigned)a.OWsub + (unsigned)(z * a.OWsub);
      }
      const unsigned vo = (b_cok && m < m_end) ? (pix * (unsigned)a.dy_ps + b_chan) * 4u : 0xffffffffu;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vo, 0, 0));
    }
    ld_m += WK;
  };
  auto store_lds = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + WK * LDA;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) *reinterpret_cast<f32x4*>(As + (pa + A_ROWS * j) * LDA + ca * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) *reinterpret_cast<f32x4*>(Bs + (pb + B_ROWS * j) * LDB + cb * 4) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) {
    load_global();
    store_lds(0);
    __syncthreads();
  }
  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < nsteps;
    const float* As = smem + buf * STAGE + wm * TM * 32 + li;
    const float* Bs = smem + buf * STAGE + WK * LDA + wn * TN * 32 + li;
    auto sub_step = [&](int s) {
      float fa[4][TM], fb[4][TN];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int row = 8 * s + 4 * lh + kk;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kk][i] = As[row * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kk][j] = Bs[row * LDB + j * 32];
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
    };
    // the address arithmetic and the issue of the next tile's loads sit in the shadow of the first
    // sub-step's MFMAs (the matrix pipe runs them while the wave issues VALU / VMEM instructions)
    sub_step(0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) load_global();
    __builtin_amdgcn_sched_barrier(0);
    sub_step(1);
    sub_step(2);
    sub_step(3);
    if (more) store_lds(buf ^ 1);
    __syncthreads();
  }

  // partial tile -> workspace [split][tap][ci][co]; D map: row (ci) = (r&3) + 8*(r>>2) + 4*lh, col (co) = li
  float* dst = a.part + ((size_t)((z * a.splits + split) * (a.KH * a.KW) + tap) * a.cin_pad + ci0 + wm * TM * 32) * a.cout_pad + co0 + wn * TN * 32 + li;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * a.cout_pad + j * 32] = acc[i][j][r];
}

// sum the pixel slices in slice order, write torch layout (Cout, Cin, KH, KW)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int taps, int cin, int cout, int cin_pad,
                                    int cout_pad, float* __restrict__ dw, int accumulate, int tap_major = 0) {
  const size_t total = (size_t)taps * cin * cout;
  const size_t slice = (size_t)taps * cin_pad * cout_pad;
  part += blockIdx.y * (size_t)splits * slice;   // range-stratified convolution: one weight set per blockIdx.y
  dw += blockIdx.y * total;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout);
    size_t r = i / cout;
    const int ci = (int)(r % cin);
    const int tap = (int)(r / cin);
    const float* p = part + ((size_t)tap * cin_pad + ci) * cout_pad + co;
    // four independent partial sums (fixed association): the loads of one round are in flight together
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      s0 += p[(size_t)k * slice]; s1 += p[(size_t)(k + 1) * slice]; s2 += p[(size_t)(k + 2) * slice]; s3 += p[(size_t)(k + 3) * slice];
    }
    for (; k < splits; ++k) s0 += p[(size_t)k * slice];
    const float s = (s0 + s1) + (s2 + s3);
    // torch (Cout, Cin, KH, KW) or spconv (Cout, taps, Cin)
    float* d = tap_major ? dw + ((size_t)co * taps + tap) * cin + ci : dw + ((size_t)co * cin + ci) * taps + tap;
    *d = accumulate ? *d + s : s;
  }
}

// per-channel sum over pixels (bias gradients): stage 1 = fixed pixel slices, stage 2 = slice order
__global__ void channel_sum_partial_kernel(const float* __restrict__ x, long long pixels, int ps, int co, int c, int slices,
                                           float* __restrict__ part) {
  const int ch = blockIdx.y * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const long long per = (pixels + slices - 1) / slices;
  const long long p0 = blockIdx.x * per, p1 = min(pixels, p0 + per);
  float s = 0.f;
  for (long long p = p0; p < p1; ++p) s += x[p * ps + co + ch];
  part[(size_t)blockIdx.x * c + ch] = s;
}

// r4: the same partials from 16-byte loads -- a thread keeps one group of four channels and one of the block's pixel lanes (two loads in
// flight per trip), the lanes join in LDS in lane order.  The scalar kernel above has one 64-lane wave per 64 channels walking its
// pixels one dependent load at a time: on the sparse encoder's 16-channel rows (150 k rows) a launch took up to 460 us, and the bias /
// scale gradients were 6.8 ms of the Waymo detector's 76 ms training iteration.  Needs c, the pixel stride and the channel offset to be
// multiples of 4 and c <= 1024.
__global__ __launch_bounds__(256) void channel_sum_partial_v4_kernel(const float* __restrict__ x, long long pixels, int ps, int co, int c, int slices,
                                                                     float* __restrict__ part) {
  __shared__ float red[1024];
  const int vpc = c / 4, ppb = 256 / vpc;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc;
  const long long per = (pixels + slices - 1) / slices;
  const long long p0 = blockIdx.x * per, p1 = min(pixels, p0 + per);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (pl < ppb) {
    const float* base = x + co + cv * 4;
    long long p = p0 + pl;
    for (; p + ppb < p1; p += 2 * ppb) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(base + p * ps);
      const f32x4 b = *reinterpret_cast<const f32x4*>(base + (p + ppb) * ps);
      s0 += a; s1 += b;
    }
    if (p < p1) s0 += *reinterpret_cast<const f32x4*>(base + p * ps);
    s0 += s1;
#pragma unroll
    for (int k = 0; k < 4; ++k) red[(cv * 4 + k) * ppb + pl] = s0[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += 256) {
    float t = 0.f;
    for (int j = 0; j < ppb; ++j) t += red[i * ppb + j];
    part[(size_t)blockIdx.x * c + i] = t;
  }
}

// one wave per channel, lane-strided then butterfly: fixed association order
__global__ __launch_bounds__(64) void channel_sum_final_kernel(const float* __restrict__ part, int slices, int c, float* __restrict__ out, int accumulate) {
  const int ch = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < slices; k += 64) s += part[(size_t)k * c + ch];
  s = pn::wave_sum(s);
  if (threadIdx.x == 0) out[ch] = accumulate ? out[ch] + s : s;
}

// ---- weight packers for the data-gradient convolutions ---------------------------------------
// stride 1: Wd[ci][co][kh][kw] = W[co][ci][KH-1-kh][KW-1-kw], packed like a forward weight with
// (cout, cin) := (Cin, Cout)
__global__ void pack_dgrad_s1_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw, int k_pad, int n_pad,
                                     float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3;
    r >>= 2;
    const int n = r % n_pad;  // = forward input channel
    r /= n_pad;
    const int k4 = r % (k_pad / 4);
    const int tap = (int)(r / (k_pad / 4));
    const int k = k4 * 4 + k1;  // = forward output channel
    float v = 0.f;
    if (n < cin && k < cout) {
      const int fh = kh - 1 - tap / kw, fw = kw - 1 - tap % kw;
      v = w[(((size_t)k * cin + n) * kh + fh) * kw + fw];
    }
    packed[i] = v;
  }
}

// 3x3 / stride 2 / pad 1: dX[2a+ph][2b+pw] = sum_{u,v in {0,1}} dY[a+u][b+v] * W[.][.][kh(ph,u)][kw(pw,v)]
//   parity 0: u = 0 -> k = 1, u = 1 -> none;   parity 1: u = 0 -> k = 2, u = 1 -> k = 0
// packed for the DECONV2 mode of conv_mfma_kernel with 2x2 taps: columns (d = 2*ph + pw, ci)
__global__ void pack_dgrad_s2_kernel(const float* __restrict__ w, int cout, int cin, int k_pad, int n_pad,
                                     float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3;
    r >>= 2;
    const int col = r % n_pad;
    r /= n_pad;
    const int k4 = r % (k_pad / 4);
    const int tap = (int)(r / (k_pad / 4));  // u*2 + v
    const int k = k4 * 4 + k1;               // forward output channel
    float v = 0.f;
    if (col < 4 * cin && k < cout) {
      const int d = col / cin, ci = col - d * cin;
      const int ph = d >> 1, pw = d & 1, u = tap >> 1, vv = tap & 1;
      const int fh = ph == 0 ? (u == 0 ? 1 : -1) : (u == 0 ? 2 : 0);
      const int fw = pw == 0 ? (vv == 0 ? 1 : -1) : (vv == 0 ? 2 : 0);
      if (fh >= 0 && fw >= 0) v = w[(((size_t)k * cin + ci) * 3 + fh) * 3 + fw];
    }
    packed[i] = v;
  }
}

// blocks per launch the pixel range is split for (512 block slots per launch: 2 blocks x 256 CUs; 1536 = three full rounds measured best at B=4, 94 TFLOP/s over the model layers)
static const int kWgradTargetBlocks = [] { const char* e = getenv("PN_WGRAD_BLOCKS"); return e ? atoi(e) : 1536; }();

static const int kWgradCapPartials = [] { const char* e = getenv("PN_WGRAD_CAP_PARTIALS"); return e ? atoi(e) : 4; }();   // divisor of the cap (4 measured best on both training steps), 0 = no cap

struct WgradPlan {
  int tm, tn, bm, bn, ci_tiles, co_tiles, cin_pad, cout_pad, splits, m_per_split, taps;
  long long M;
  int OH, OW, strata;
};

int plan_wgrad(const pn_conv_desc* d, WgradPlan& p) {
  PN_REQUIRE(d != nullptr, "wgrad: null descriptor");
  PN_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0, "wgrad: bad sizes");
  PN_REQUIRE(d->groups == 1 && !d->deconv2x2, "wgrad: plain and range-stratified convolutions only");
  PN_REQUIRE(d->kh >= 1 && d->kw >= 1 && d->stride >= 1, "wgrad: bad kernel params");
  p.OH = (d->in_h + 2 * d->pad_h + d->pad_h_end - d->kh) / d->stride + 1;
  p.OW = (d->in_w + 2 * d->pad_w + d->pad_w_end - d->kw) / d->stride + 1;
  PN_REQUIRE(p.OH > 0 && p.OW > 0, "wgrad: empty output");
  // range-stratified (center_head_parallel.py:27-59): `cout` filters PER stratum, dweight (strata * cout, cin, kh, kw); every stratum is
  // its own GEMM over the pixels of its column band (the halo columns of the input come from the neighbouring bands)
  p.strata = d->range_strata > 1 ? d->range_strata : 1;
  PN_REQUIRE(p.strata == 1 || (d->stride == 1 && p.OW % p.strata == 0 && d->cin <= 64 && d->cout <= 64),
             "wgrad: range strata need stride 1, a width divisible into the strata and at most 64 channels per stratum");
  p.M = (long long)d->batch * p.OH * (p.OW / p.strata);
  PN_REQUIRE(p.M < (1ll << 31) - 64, "wgrad: too many output pixels");
  p.tm = d->cin > 64 ? 2 : 1;
  p.tn = d->cout > 64 ? 2 : 1;
  p.bm = p.tm * 64; p.bn = p.tn * 64;
  p.ci_tiles = pn::cdiv(d->cin, p.bm); p.co_tiles = pn::cdiv(d->cout, p.bn);
  p.cin_pad = p.ci_tiles * p.bm; p.cout_pad = p.co_tiles * p.bn;
  p.taps = d->kh * d->kw;
  const long long tiles = (long long)p.taps * p.ci_tiles * p.co_tiles;
  long long s = std::max<long long>(1, kWgradTargetBlocks / tiles);  // never one block over a full round of the 512 block slots
  // r4: every slice writes a full (taps, cin, cout) partial and the reduction reads it back -- on a layer with few pixels and many
  // weights (128 -> 256 at 64 x 64: 85 slices x 1.2 MB against 50 MB of maps) that traffic was most of the launch pair.  The slice count
  // is capped where the partials reach a quarter of the maps' bytes, but not below one round of the block slots (13.3 -> 13.1 ms per training
  // iteration of the pillar model; the Waymo detector's large maps keep their three rounds: fewer cost it 4 ms)
  if (kWgradCapPartials) {
    const long long map_bytes = ((long long)d->batch * d->in_h * d->in_w * d->cin + p.M * d->cout) * 4;
    const long long w_bytes = (long long)p.taps * p.cin_pad * p.cout_pad * 4;
    const long long one_round = std::max<long long>(1, 512 / tiles);
    s = std::min(s, std::max(one_round, map_bytes / (kWgradCapPartials * w_bytes)));
  }
  s = std::min<long long>(s, std::max<long long>(1, p.M / 256));
  p.m_per_split = (int)(((p.M + s - 1) / s + WK - 1) / WK * WK);
  p.splits = (int)((p.M + p.m_per_split - 1) / p.m_per_split);
  return PN_OK;
}

template <int TM, int TN, bool GATHER = false, bool STRAT = false>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t st, int strata = 1) {
  constexpr size_t smem = 2 * (size_t)WK * (TM * 64 + 8 + TN * 64 + 8) * sizeof(float);
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<TM, TN, GATHER, STRAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int per_split = p.taps * p.ci_tiles * p.co_tiles;
  const int groups8 = pn::cdiv(p.splits, 8);  // splits are handed out in groups of 8, one per XCD
  hipLaunchKernelGGL((conv_wgrad_kernel<TM, TN, GATHER, STRAT>), dim3(groups8 * per_split * 8, strata), dim3(256), smem, st, a);
  return pn::check_launch("conv_wgrad_kernel");
}

// plan of the gathered (sparse) weight gradient: rows = output sites, `taps` neighbour columns
int plan_sparse_wgrad(int out_capacity, int taps, int cout, int cin, WgradPlan& p) {
  PN_REQUIRE(out_capacity >= 1 && taps >= 1 && taps <= 64 && cout >= 1 && cin >= 4 && cin % 4 == 0, "sparse_conv_wgrad: bad sizes (row width a multiple of 4)");
  p.OH = out_capacity; p.OW = 1; p.strata = 1;
  p.M = out_capacity;
  p.tm = cin > 64 ? 2 : 1;
  p.tn = cout > 64 ? 2 : 1;
  p.bm = p.tm * 64; p.bn = p.tn * 64;
  p.ci_tiles = pn::cdiv(cin, p.bm); p.co_tiles = pn::cdiv(cout, p.bn);
  p.cin_pad = p.ci_tiles * p.bm; p.cout_pad = p.co_tiles * p.bn;
  p.taps = taps;
  const long long tiles = (long long)p.taps * p.ci_tiles * p.co_tiles;
  long long s = std::max<long long>(1, kWgradTargetBlocks / tiles);
  s = std::min<long long>(s, std::max<long long>(1, p.M / 256));
  p.m_per_split = (int)(((p.M + s - 1) / s + WK - 1) / WK * WK);
  p.splits = (int)((p.M + p.m_per_split - 1) / p.m_per_split);
  return PN_OK;
}

constexpr int kSumSlices = 2048;
static const int kChannelSumV4 = [] { const char* e = getenv("PN_CHANNEL_SUM_V4"); return e ? atoi(e) : 1; }();

}  // namespace

extern "C" {

size_t pn_conv2d_wgrad_workspace_bytes(const pn_conv_desc* d) {
  WgradPlan p;
  if (plan_wgrad(d, p)) return 0;
  return (size_t)p.strata * p.splits * p.taps * p.cin_pad * p.cout_pad * sizeof(float);
}

int pn_conv2d_wgrad_f32(const pn_conv_desc* d, const float* in, const float* dout, float* dweight, int accumulate,
                        void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  WgradPlan p;
  if (int rc = plan_wgrad(d, p)) return rc;
  PN_REQUIRE(in && dout && dweight && workspace, "wgrad: null pointer");
  PN_REQUIRE(workspace_bytes >= pn_conv2d_wgrad_workspace_bytes(d), "wgrad: workspace too small");
  // 16-byte buffer loads only need dword alignment; channels past Cin / Cout inside a 4-channel
  // load land in rows / columns of the tile that are never written out, and the buffer range check
  // zero-fills the dwords past the end of the map
  PN_REQUIRE(d->in_pixel_stride >= d->in_channel_offset + d->cin && d->out_pixel_stride >= d->out_channel_offset + d->cout,
             "wgrad: channel slice does not fit the pixel stride");
  const unsigned long long in_bytes = (unsigned long long)d->batch * d->in_h * d->in_w * d->in_pixel_stride * 4ull;
  const unsigned long long dy_bytes = (unsigned long long)p.M * p.strata * d->out_pixel_stride * 4ull;
  PN_REQUIRE(in_bytes < (1ull << 31) && dy_bytes < (1ull << 31), "wgrad: maps larger than 2 GiB are not addressable by the buffer descriptor");
  WgradArgs a;
  a.in = in; a.dy = dout; a.part = static_cast<float*>(workspace);
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = d->cin; a.Cout = d->cout; a.OH = p.OH; a.OW = p.OW;
  a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.in_ps = d->in_pixel_stride; a.in_co = d->in_channel_offset; a.dy_ps = d->out_pixel_stride; a.dy_co = d->out_channel_offset;
  a.M = (int)p.M; a.m_per_split = p.m_per_split; a.ci_tiles = p.ci_tiles; a.cin_pad = p.cin_pad; a.cout_pad = p.cout_pad;
  a.co_tiles = p.co_tiles; a.tiles_per_split = p.taps * p.ci_tiles * p.co_tiles; a.splits = p.splits;
  a.in_bytes = (unsigned)in_bytes; a.dy_bytes = (unsigned)dy_bytes;
  a.OWsub = p.OW / p.strata;
  a.div_ohw = make_fastdiv((unsigned)(p.OH * a.OWsub)); a.div_ow = make_fastdiv((unsigned)a.OWsub);
  hipStream_t st = pn::S(stream);
  int rc;
  if (p.strata > 1) rc = launch_wgrad<1, 1, false, true>(a, p, st, p.strata);
  else if (p.tm == 2 && p.tn == 2) rc = launch_wgrad<2, 2>(a, p, st);
  else if (p.tm == 2) rc = launch_wgrad<2, 1>(a, p, st);
  else if (p.tn == 2) rc = launch_wgrad<1, 2>(a, p, st);
  else rc = launch_wgrad<1, 1>(a, p, st);
  if (rc) return rc;
  const size_t total = (size_t)p.taps * d->cin * d->cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256), p.strata), dim3(256), 0, st,
                     a.part, p.splits, p.taps, d->cin, d->cout, p.cin_pad, p.cout_pad, dweight, accumulate);
  return pn::check_launch("wgrad_reduce_kernel");
}

size_t pn_sparse_conv_wgrad_workspace_bytes(int out_capacity, int taps, int cout, int cin) {
  WgradPlan p;
  if (plan_sparse_wgrad(out_capacity, taps, cout, cin, p)) return 0;
  return (size_t)p.splits * p.taps * p.cin_pad * p.cout_pad * sizeof(float);
}

int pn_sparse_conv_wgrad_f32(const float* in, int in_rows, int cin, int cin_real, const float* dout, int cout, const int32_t* nbr, const int32_t* n_out,
                             int out_capacity, int taps, float* dw, int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  WgradPlan p;
  if (int rc = plan_sparse_wgrad(out_capacity, taps, cout, cin, p)) return rc;
  PN_REQUIRE(in && dout && nbr && n_out && dw && workspace, "sparse_conv_wgrad: null pointer");
  PN_REQUIRE(cin_real >= 1 && cin_real <= cin, "sparse_conv_wgrad: cin_real out of range");
  if (workspace_bytes < pn_sparse_conv_wgrad_workspace_bytes(out_capacity, taps, cout, cin)) return pn::fail(PN_ERR_WORKSPACE, "sparse_conv_wgrad: workspace too small");
  const unsigned long long dy_bytes = (unsigned long long)out_capacity * cout * 4ull;
  PN_REQUIRE(dy_bytes < (1ull << 31), "sparse_conv_wgrad: gradient matrix larger than 2 GiB");
  // the loader forms 32-bit byte offsets idx * cin * 4 into `in`: a feature matrix of 2 GiB or more would wrap / fall outside the descriptor
  PN_REQUIRE(in_rows > 0 && (unsigned long long)in_rows * cin * 4ull < (1ull << 31), "sparse_conv_wgrad: feature matrix (in_rows x cin) must be under 2 GiB");
  WgradArgs a{};
  a.in = in; a.dy = dout; a.part = static_cast<float*>(workspace);
  a.B = 1; a.H = out_capacity; a.W = 1; a.Cin = cin; a.Cout = cout; a.OH = out_capacity; a.OW = 1; a.OWsub = 1;
  a.KH = taps; a.KW = 1; a.stride = 1; a.pad_h = 0; a.pad_w = 0;
  a.in_ps = cin; a.in_co = 0; a.dy_ps = cout; a.dy_co = 0;
  a.M = out_capacity; a.m_per_split = p.m_per_split; a.ci_tiles = p.ci_tiles; a.cin_pad = p.cin_pad; a.cout_pad = p.cout_pad;
  a.co_tiles = p.co_tiles; a.tiles_per_split = p.taps * p.ci_tiles * p.co_tiles; a.splits = p.splits;
  a.in_bytes = (unsigned)((unsigned long long)in_rows * cin * 4ull);   // rows past the matrix read as zero (hardware bounds check)
  a.dy_bytes = (unsigned)dy_bytes;
  a.div_ohw = make_fastdiv((unsigned)out_capacity); a.div_ow = make_fastdiv(1u);
  a.nbr = nbr; a.n_valid = n_out;
  hipStream_t st = pn::S(stream);
  int rc;
  if (p.tm == 2 && p.tn == 2) rc = launch_wgrad<2, 2, true>(a, p, st);
  else if (p.tm == 2) rc = launch_wgrad<2, 1, true>(a, p, st);
  else if (p.tn == 2) rc = launch_wgrad<1, 2, true>(a, p, st);
  else rc = launch_wgrad<1, 1, true>(a, p, st);
  if (rc) return rc;
  const size_t total = (size_t)p.taps * cin_real * cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, st, a.part, p.splits, p.taps, cin_real, cout,
                     p.cin_pad, p.cout_pad, dw, accumulate, 1);
  return pn::check_launch("wgrad_reduce_kernel");
}

size_t pn_channel_sum_workspace_bytes(int c) { return (size_t)kSumSlices * c * sizeof(float); }

int pn_channel_sum_f32(const float* x, long long pixels, int pixel_stride, int channel_offset, int c, float* out,
                       int accumulate, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(x && out && workspace && pixels > 0 && c > 0 && pixel_stride >= c, "channel_sum: bad arguments");
  PN_REQUIRE(workspace_bytes >= pn_channel_sum_workspace_bytes(c), "channel_sum: workspace too small");
  int slices = (int)std::min<long long>(kSumSlices, pixels);
  float* part = static_cast<float*>(workspace);
  if (kChannelSumV4 && c % 4 == 0 && pixel_stride % 4 == 0 && channel_offset % 4 == 0 && c <= 1024 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int ppb = 256 / (c / 4);
    slices = (int)std::max<long long>(1, std::min<long long>(1024, pixels / (8 * ppb)));     // eight or more pixels per thread
    hipLaunchKernelGGL(channel_sum_partial_v4_kernel, dim3(slices), dim3(256), 0, pn::S(stream), x, pixels, pixel_stride, channel_offset, c, slices, part);
  } else {
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(slices, pn::cdiv(c, 64)), dim3(64), 0, pn::S(stream), x, pixels,
                       pixel_stride, channel_offset, c, slices, part);
  }
  if (int rc = pn::check_launch("channel_sum_partial_kernel")) return rc;
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(c), dim3(64), 0, pn::S(stream), part, slices, c, out, accumulate);
  return pn::check_launch("channel_sum_final_kernel");
}

int pn_pack_conv_dgrad_weight_f32(const float* w_oihw, int cout, int cin, int kh, int kw, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_dgrad: bad arguments");
  const int k_pad = pn::cdiv(cout, 32) * 32, n_pad = pn::cdiv(cin, 32) * 32;
  const size_t total = (size_t)kh * kw * k_pad * n_pad;  // == pn_conv_packed_weight_floats(cin, cout, kh, kw, 1)
  hipLaunchKernelGGL(pack_dgrad_s1_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                     pn::S(stream), w_oihw, cout, cin, kh, kw, k_pad, n_pad, packed, total);
  return pn::check_launch("pack_dgrad_s1_kernel");
}

size_t pn_conv_dgrad_s2_packed_weight_floats(int cout, int cin) {
  return (size_t)4 * (pn::cdiv(cout, 32) * 32) * (size_t)(pn::cdiv(4 * cin, 32) * 32);
}

int pn_pack_conv_dgrad_s2_weight_f32(const float* w_oihw, int cout, int cin, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && cout > 0 && cin > 0, "pack_dgrad_s2: bad arguments");
  const int k_pad = pn::cdiv(cout, 32) * 32, n_pad = pn::cdiv(4 * cin, 32) * 32;
  const size_t total = (size_t)4 * k_pad * n_pad;
  hipLaunchKernelGGL(pack_dgrad_s2_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                     pn::S(stream), w_oihw, cout, cin, k_pad, n_pad, packed, total);
  return pn::check_launch("pack_dgrad_s2_kernel");
}

}  // extern "C"
