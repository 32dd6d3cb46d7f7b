// bf16 BEV convolutions of the Waymo PARTNER config (BASELINE configs[3]: "bf16 BEV convs on MFMA") as an implicit GEMM on
// v_mfma_f32_16x16x32_bf16: out[pixel][n] = act(scale[n] * sum_{tap, c} in[pixel + tap][c] * w[n][tap][c] + shift[n]).
// Replaces, for the bf16 option, the cuDNN convolutions of RPN (det3d/models/necks/rpn.py:80-110, 124-142: Conv2d 3x3 stride 1 / 2 +
// BatchNorm + ReLU, ConvTranspose2d k = s = 2 / Conv2d 1x1 deblocks) and the 3x3 convolutions of E2ESWVoteHead
// (det3d/models/bbox_heads/e2e_swv_head.py:57-118).  r5: a kernel of its own -- until r4 the bf16 path was a template variant of the
// fp32 implicit-GEMM kernel (conv_mfma.hip) and reached 0.17 of the bf16 matrix peak.
//
// Shape of the kernel (gfx950):
//   * GEMM view: columns = output pixels of the whole batch (linear index, a tile may straddle rows and samples), rows = output
//     channels, K = (tap, input channel) in steps of 64 channels of one tap.  The MFMA's "A" operand is the WEIGHT tile and its "B"
//     operand the pixel tile, so an accumulator lane owns 4 consecutive output channels of one pixel: the NHWC store is 8 (bf16) or
//     16 (f32) contiguous bytes per lane.
//   * Both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 bytes per lane, no staging registers): one
//     instruction moves 8 rows x 128 bytes = whole cache lines of 8 pixels (or 8 weight rows).  The im2col gather is the per-lane
//     SOURCE address (pixel + tap offset); padding pixels take an offset beyond the buffer descriptor's range and arrive as zeros.
//   * LDS image of a stage: [rows][128 bytes], the 16-byte chunk c of row r stored at chunk c ^ (r & 7) (applied on the source
//     address, the LDS side of an LDS-DMA is lane-linear): the fragment reads (ds_read_b128, lane = (row l % 16, k block l / 16))
//     are bank-conflict free in all four lane groups of the instruction.
//   * Two stages, one barrier per K step: the loads of step s + 1 are issued right after the barrier that publishes step s and fly
//     during its MFMAs.  Two blocks per CU cover each other's barrier waits.
//   * Tiles are chosen per layer so that the launch is a whole number of rounds over the 256 CUs x 2 blocks (the Waymo maps are
//     2^k x 9 pixels: 144-pixel tiles), and blocks are numbered so that an XCD works on a contiguous run of pixel tiles (the halo
//     rows and the weights stay in its L2).
#include "pn_common.h"
#include <algorithm>

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct CbArgs {
  const void* in;
  const void* w;
  const float* scale;
  const float* shift;
  void* out;
  int B, H, W, OH, OW, cin, cout, kh, kw, stride, pad_h, pad_w;
  int in_ps, in_co, out_ps, out_co, act, deconv;      // pixel strides / channel offsets in ELEMENTS
  int M;                                              // GEMM columns: output pixels of the batch (deconv: input pixels)
  int N;                                              // GEMM rows: output channels (deconv: 4 x cout)
  int kchunks, ksteps;                                // cin / 64; taps x kchunks
  int Kw;                                             // elements per packed weight row = taps x cin
  int ptiles, ctiles;
  unsigned in_bytes, w_bytes;
  const float* res;                                   // GEMM use (pn_linear_bf16): f32 residual rows added after the activation, f32 output only
  int res_ps;
};

// exact-erf GELU as in linear.hip (erfc by a Chebyshev fit, |gelu - exact| < 3e-7 max(|x|, 1)): the f32 and the bf16 GEMMs share it
__device__ __forceinline__ float cb_gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.f));
  float p = fmaf(t, 0.17087277f, -0.82215223f);
  p = fmaf(t, p, 1.48851587f);
  p = fmaf(t, p, -1.13520398f);
  p = fmaf(t, p, 0.27886807f);
  p = fmaf(t, p, -0.18628806f);
  p = fmaf(t, p, 0.09678418f);
  p = fmaf(t, p, 0.37409196f);
  p = fmaf(t, p, 1.00002368f);
  p = fmaf(t, p, -1.26551223f);
  const float e = t * __builtin_amdgcn_exp2f(fmaf(-z, z, p) * 1.44269504088896341f);
  return 0.5f * x * (x >= 0.f ? 2.f - e : e);
}

__host__ __device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  unsigned u = __builtin_bit_cast(unsigned, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                    // round to nearest even
  return (unsigned short)(u >> 16);
}

// two f32 -> one register of two bf16, round to nearest even, on gfx950's v_cvt_pk_bf16_f32 (the bits of f32_to_bf16_bits for every finite value;
// the epilogue of a bf16 GEMM is vector work -- ten integer operations per value here were a third of the 256 -> 1024 GELU launch)
__device__ __forceinline__ unsigned cb_pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

#ifndef PN_CB_EXP
#define PN_CB_EXP 0      // diagnostic builds (tools/convbf16q.sh DEFS=-DPN_CB_EXP=k): 1 no pixel loads after step 0, 2 no weight loads after step 0,
#endif                   // 4 no MFMAs, 8 no fragment reads after step 0 -- wrong results, the time shows what a K step waits for
constexpr unsigned kOob = 0x80000000u;      // beyond every buffer this kernel is handed (sizes < 2 GiB are required)

// WP x WC waves; a wave owns (PT x 16) pixels x (CT x 16) output channels
// NS stages: 2 = the loads of step s + 1 fly during step s (two blocks per CU cover each other's waits); 4 = three steps ahead with counted
// vmcnt, for launches of at most one block per CU (the block then has the CU's LDS to itself)
#define CB_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void cb_wait_vmcnt(int n) {      // n is wave-uniform
  switch (n) {
    CB_WAIT_CASE(0) CB_WAIT_CASE(1) CB_WAIT_CASE(2) CB_WAIT_CASE(3) CB_WAIT_CASE(4) CB_WAIT_CASE(5) CB_WAIT_CASE(6) CB_WAIT_CASE(7) CB_WAIT_CASE(8)
    CB_WAIT_CASE(9) CB_WAIT_CASE(10) CB_WAIT_CASE(11) CB_WAIT_CASE(12) CB_WAIT_CASE(13) CB_WAIT_CASE(14) CB_WAIT_CASE(15) CB_WAIT_CASE(16)
    CB_WAIT_CASE(17) CB_WAIT_CASE(18) CB_WAIT_CASE(19) CB_WAIT_CASE(20) CB_WAIT_CASE(21) CB_WAIT_CASE(22) CB_WAIT_CASE(23) CB_WAIT_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int WP, int WC, int PT, int CT, bool F32OUT, int NS>
__global__ __launch_bounds__(WP* WC * 64) void conv_bf16_igemm_kernel(const CbArgs a) {
  constexpr int NW = WP * WC;
  constexpr int BP = WP * PT * 16, BC = WC * CT * 16;
  constexpr int PI = BP / 8, CI = BC / 8;                        // LDS-DMA instructions per stage: pixel rows, weight rows
  constexpr int NPI = (PI + NW - 1) / NW, NCI = (CI + NW - 1) / NW;
  constexpr int STAGE = (BP + BC) * 128;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wv / WC, wc = wv - wp * WC;

  // block -> tile: XCD x (= blockIdx % 8) takes a contiguous run of the work list, pixel tile major
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int ptile = work / a.ctiles, ctile = work - ptile * a.ctiles;
  const int m0 = ptile * BP, n0 = ctile * BC;

  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);

  // ---- loader: this lane's rows.  Instruction q of a stage covers rows 8 q .. 8 q + 7; lane -> (row 8 q + lane / 8, LDS chunk lane % 8),
  // which holds the row's global chunk (lane % 8) ^ (lane / 8)
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  int p_base[NPI], p_iy[NPI], p_ix[NPI];
#pragma unroll
  for (int j = 0; j < NPI; ++j) {
    const int q = wv + j * NW;
    const int m = m0 + 8 * q + lrow;
    const bool live = q < PI && m < a.M;
    const int hw = a.OH * a.OW;
    const int b = m / hw, rem = m - b * hw;
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.stride - a.pad_h, ix0 = ox * a.stride - a.pad_w;
    p_iy[j] = live ? iy0 : -0x40000000;          // every tap of a dead row fails the range test
    p_ix[j] = ix0;
    p_base[j] = (((b * a.H + iy0) * a.W + ix0) * a.in_ps + a.in_co) * 2 + lchunk * 16;
  }
  unsigned w_base[NCI];
#pragma unroll
  for (int j = 0; j < NCI; ++j) {
    const int q = wv + j * NW;
    const int n = n0 + 8 * q + lrow;
    w_base[j] = (q < CI && n < a.N) ? (unsigned)n * (unsigned)a.Kw * 2u + (unsigned)lchunk * 16u : kOob;
  }

  int s_dy = 0, s_dx = 0, s_kc = 0;      // tap and channel chunk of the step whose loads are issued next
  auto issue = [&](int step, int buf) {
    char* base = smem + buf * STAGE;
    const int tap_off = ((s_dy * a.W + s_dx) * a.in_ps + s_kc * 64) * 2;
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      const int q = wv + j * NW;
      if (q < PI && !((PN_CB_EXP & 1) && step > 0)) {
        const bool ok = (unsigned)(p_iy[j] + s_dy) < (unsigned)a.H && (unsigned)(p_ix[j] + s_dx) < (unsigned)a.W;
        const unsigned off = ok ? (unsigned)(p_base[j] + tap_off) : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(base + q * 1024), 16, off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NCI; ++j) {
      const int q = wv + j * NW;
      if (q < CI && !((PN_CB_EXP & 2) && step > 0)) {
        const unsigned off = w_base[j] == kOob ? kOob : w_base[j] + (unsigned)step * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(base + BP * 128 + q * 1024), 16, off, 0, 0, 0);
      }
    }
    if (++s_kc == a.kchunks) {
      s_kc = 0;
      if (++s_dx == a.kw) { s_dx = 0; ++s_dy; }
    }
  };

  // ---- fragment addresses: lane = (row l % 16 of the 16-row tile, k block l / 16); k half h of a 64-channel step
  const int frow = lane & 15, fkb = lane >> 4;
  int f_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) f_off[h] = frow * 128 + (((h * 4 + fkb) ^ (lane & 7)) << 4);
  const int p_frag = wp * PT * 16 * 128, c_frag = BP * 128 + wc * CT * 16 * 128;

  f32x4 acc[CT][PT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 wf[2][CT], pf[2][PT];
  constexpr int D = NS - 1;                    // prefetch distance in steps
  int nw_loads = 0;                            // LDS-DMA instructions this wave issues per stage
#pragma unroll
  for (int j = 0; j < NPI; ++j) nw_loads += (wv + j * NW < PI);
#pragma unroll
  for (int j = 0; j < NCI; ++j) nw_loads += (wv + j * NW < CI);
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < a.ksteps) issue(d, d);
  for (int s = 0; s < a.ksteps; ++s) {
    if constexpr (NS == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      cb_wait_vmcnt(min(D - 1, a.ksteps - 1 - s) * nw_loads);      // the stages requested after step s's may stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (s + D < a.ksteps) issue(s + D, (s + D) % NS);
    const char* st = smem + (s % NS) * STAGE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!((PN_CB_EXP & 8) && s > 0)) {
#pragma unroll
        for (int i = 0; i < CT; ++i) wf[h][i] = *reinterpret_cast<const bf16x8*>(st + c_frag + i * 2048 + f_off[h]);
#pragma unroll
        for (int j = 0; j < PT; ++j) pf[h][j] = *reinterpret_cast<const bf16x8*>(st + p_frag + j * 2048 + f_off[h]);
      }
      if (!(PN_CB_EXP & 4)) {
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][i], pf[h][j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < CT; ++i) acc[i][0][0] += (float)wf[h][i][0] + (float)pf[h][i % PT][1];
      }
    }
  }

  // ---- epilogue: lane = pixel l % 16 of the tile, channels 4 (l / 16) .. + 3 of the 16-channel tile
  const int cq = (lane >> 4) * 4;
  if (!a.deconv) {
    // through LDS (see the rows form below): [pixel][BC channels] rows, then 16 bytes per lane along the channel rows -- whole lines
    // instead of 8-byte pieces.  Output pixels of a tile are consecutive in memory (linear pixel index).
    constexpr int EB = F32OUT ? 4 : 2;
    constexpr int ROWB = BC * EB + 16;
    constexpr int PASSES = (BP * ROWB + 2 * STAGE - 1) / (2 * STAGE);          // 1, or 2 for f32 rows (sized for the two-stage form)
    constexpr int PTP = (PT + PASSES - 1) / PASSES;                            // pixel tiles of a wave per pass
    constexpr int SEGS = BC * EB / 16;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
      for (int jj = 0; jj < PTP; ++jj) {
        const int j = pass * PTP + jj;
        if (j >= PT) continue;
        const int pl = (wp * PTP + jj) * 16 + (lane & 15);                     // row of this pass's LDS image
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int nl = (wc * CT + i) * 16 + cq;
          const int n = n0 + nl;
          f32x4 v = acc[i][j];
          if (n < a.N) {
            if (a.scale) v = v * *reinterpret_cast<const f32x4*>(a.scale + n);
            if (a.shift) v = v + *reinterpret_cast<const f32x4*>(a.shift + n);
          }
          if (a.act == PN_ACT_RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
          } else if (a.act == PN_ACT_GELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = cb_gelu_erf(v[k]);
          }
          char* dst = smem + pl * ROWB + nl * EB;
          if constexpr (F32OUT) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
            const unsigned lo = cb_pack_bf16x2(v[0], v[1]);
            const unsigned hi = cb_pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = uint2{lo, hi};
          }
        }
      }
      __syncthreads();
      const int nseg = min(SEGS, (a.N - n0) * EB / 16);                        // (N is a multiple of 16: whole segments)
      for (int idx = tid; idx < WP * PTP * 16 * SEGS; idx += NW * 64) {
        const int pl = idx / SEGS, seg = idx - pl * SEGS;
        const int w_ = pl / (PTP * 16), rest = pl - w_ * (PTP * 16);          // wave row, (tile of the pass, pixel)
        const int jt = pass * PTP + rest / 16;
        const int m = m0 + (w_ * PT + jt) * 16 + (rest & 15);
        if (jt >= PT || m >= a.M || seg >= nseg) continue;
        uint4 val = *reinterpret_cast<const uint4*>(smem + pl * ROWB + seg * 16);
        if constexpr (F32OUT) {
          if (a.res) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(a.res + (size_t)m * (size_t)a.res_ps + n0 + seg * 4);
            val = __builtin_bit_cast(uint4, __builtin_bit_cast(f32x4, val) + r);
          }
        }
        char* o = static_cast<char*>(a.out) + ((size_t)m * (size_t)a.out_ps + a.out_co + n0) * EB + seg * 16;
        *reinterpret_cast<uint4*>(o) = val;
      }
      if (pass + 1 < PASSES) __syncthreads();
    }
    return;
  }
  // ConvTranspose2d(k = s = 2) as a 1x1 convolution with 4 x cout rows: quadrant q of input pixel (y, x) lands on (2 y + q / 2, 2 x + q % 2)
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int m = m0 + (wp * PT + j) * 16 + (lane & 15);
    if (m >= a.M) continue;
    const int hw = a.H * a.W;
    const int bb = m / hw, rem = m - bb * hw;
    const int yx_y = rem / a.W, yx_x = rem - yx_y * a.W;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int n = n0 + (wc * CT + i) * 16 + cq;
      if (n >= a.N) continue;
      const int quad = n / a.cout, co = n - quad * a.cout;
      const size_t opix = ((size_t)bb * 2 * a.H + 2 * yx_y + (quad >> 1)) * (size_t)(2 * a.W) + 2 * yx_x + (quad & 1);
      f32x4 v = acc[i][j];
      if (a.scale) v = v * *reinterpret_cast<const f32x4*>(a.scale + co);
      if (a.shift) v = v + *reinterpret_cast<const f32x4*>(a.shift + co);
      if (a.act == PN_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      const size_t o = opix * (size_t)a.out_ps + a.out_co + co;
      if constexpr (F32OUT) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + o) = v;
      } else {
        const unsigned lo = cb_pack_bf16x2(v[0], v[1]);
        const unsigned hi = cb_pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(static_cast<unsigned short*>(a.out) + o) = uint2{lo, hi};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 layers on maps whose rows tile 288 pixels (the Waymo BEV maps: 144 and 72 columns): the ROWS form.
// The implicit GEMM above fetches every input pixel once per tap -- nine times -- and at 2.4 GHz its K step waits for L2, not for the
// matrix pipe (loads alone: 57 GB/s per CU, the same time as the whole kernel).  Here a block owns TR whole output rows (TR x W = 288
// pixels, one block per CU, 12 waves) x 128 output channels and keeps, per 64-channel chunk, the (TR + 2) x (W + 2) input PATCH in LDS:
// a pixel is fetched once per chunk and read by all nine taps at shifted patch positions; only the 16 KB weight tile of a (chunk, tap)
// step is streamed through two alternating groups of stages (the next group requested while the current one multiplies, vmcnt(0) at the
// group boundary, raw barriers).  L2 -> LDS traffic per output drops 2.8 x.
// The patch image is [patch pixel][128 B] with chunk c of pixel r at c ^ (r & 7): 16 consecutive patch pixels from ANY start are
// conflict free for ds_read_b128 (a tile that straddles two output rows jumps by PS - W = 8 patch pixels: same residues, same banks).
struct CrArgs {
  const void* in;
  const void* w;
  const float* scale;
  const float* shift;
  void* out;
  int B, H, W, cin, cout, TR, PS, tiles_per_image;
  int in_ps, in_co, out_ps, out_co, act;
  int kchunks, ksteps, Kw, ctiles;
  int patch_rows;                 // (TR + 2) x PS
  int patch_bytes;                // rounded up to 1 KiB
  unsigned in_bytes, w_bytes;
};

constexpr int kRowsBP = 288, kRowsBC = 128, kRowsWaves = 12, kRowsStages = 4, kRowsMaxPI = 8;

template <bool F32OUT>
__global__ __launch_bounds__(kRowsWaves * 64) void conv_bf16_rows_kernel(const CrArgs a) {
  constexpr int WC = 2, PT = 3, CT = 4;
  constexpr int WSTAGE = kRowsBC * 128;
  extern __shared__ __attribute__((aligned(1024))) char smem[];      // [patch][4 weight stages]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wv / WC, wc = wv - wp * WC;

  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int ptile = work / a.ctiles, ctile = work - ptile * a.ctiles;
  const int b = ptile / a.tiles_per_image, y0 = (ptile - b * a.tiles_per_image) * a.TR;
  const int n0 = ctile * kRowsBC;

  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  char* const wst = smem + a.patch_bytes;

  // ---- patch loader: instruction q covers patch pixels 8 q .. 8 q + 7 (linear over (TR + 2) x PS); this wave takes q = wv, wv + 12, ...
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  const int npq = a.patch_bytes >> 10;
  unsigned p_off[kRowsMaxPI];
#pragma unroll
  for (int j = 0; j < kRowsMaxPI; ++j) {
    const int q = wv + j * kRowsWaves;
    const int r = 8 * q + lrow;
    const int prow = r / a.PS, pcol = r - prow * a.PS;
    const int y = y0 - 1 + prow, x = pcol - 1;
    const bool ok = q < npq && r < a.patch_rows && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    p_off[j] = ok ? (unsigned)((((b * a.H + y) * a.W + x) * a.in_ps + a.in_co) * 2 + lchunk * 16) : kOob;
  }
  // part 0 = patch rows 0 and 1 (instructions below qsplit = 2 PS / 8), part 1 = the rest, part 2 = everything
  const int qsplit = a.PS >> 2;
  auto issue_patch = [&](int chunk, int part) {
#pragma unroll
    for (int j = 0; j < kRowsMaxPI; ++j) {
      const int q = wv + j * kRowsWaves;
      if (q < npq && (part == 2 || (q < qsplit) == (part == 0))) {
        const unsigned off = p_off[j] == kOob ? kOob : p_off[j] + (unsigned)chunk * 128u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(smem + q * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  // ---- weight loader: waves 0 .. 7, two instructions (16 rows of 128 B) per stage each
  unsigned w_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 16 * wv + 8 * j + lrow;
    w_off[j] = (wv < 8 && n < a.cout) ? (unsigned)n * (unsigned)a.Kw * 2u + (unsigned)lchunk * 16u : kOob;
  }
  // K order of the packed rows is [tap][cin]: (chunk, tap) reads bytes (tap * cin + chunk * 64) * 2 of a row
  auto issue_w = [&](int chunk, int tap, int stage) {
    if (wv < 8) {
      const unsigned so = (unsigned)(tap * a.cin + chunk * 64) * 2u;
      char* dst = wst + stage * WSTAGE + wv * 2048;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(dst + j * 1024), 16, w_off[j] == kOob ? kOob : w_off[j] + so, 0, 0, 0);
    }
  };
  // a GROUP = the taps between two barriers: (0,1) (2,3) (4,5) (6,7) (8) of a chunk; group g's weight tiles sit in stages 2 (g & 1), + 1
  auto issue_group = [&](int g) {
    const int chunk = g / 5, pp = g - chunk * 5;
    issue_w(chunk, 2 * pp, 2 * (g & 1));
    if (pp < 4) issue_w(chunk, 2 * pp + 1, 2 * (g & 1) + 1);
  };

  // ---- fragments: weight rows as in the implicit GEMM; pixel tile j of this wave = block pixels (wp * 3 + j) * 16 + l % 16
  const int fkb = lane >> 4;
  int wf_off[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) wf_off[h] = (wc * CT * 16 + (lane & 15)) * 128 + (((h * 4 + fkb) ^ (lane & 7)) << 4);
  int pr0[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int p = (wp * PT + j) * 16 + (lane & 15);
    const int row = p / a.W, col = p - row * a.W;
    pr0[j] = row * a.PS + col;           // patch pixel of tap (0, 0)
  }

  f32x4 acc[CT][PT];
#pragma unroll
  for (int i = 0; i < CT; ++i)
#pragma unroll
    for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 wf[2][CT], pf[2][PT];
  auto tap_mfma = [&](const char* ws, int toff, bool first) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!((PN_CB_EXP & 8) && !first)) {
#pragma unroll
        for (int i = 0; i < CT; ++i) wf[h][i] = *reinterpret_cast<const bf16x8*>(ws + i * 2048 + wf_off[h]);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
          const int r = pr0[j] + toff;
          pf[h][j] = *reinterpret_cast<const bf16x8*>(smem + (r << 7) + (((h * 4 + fkb) ^ (r & 7)) << 4));
        }
      }
      if (!(PN_CB_EXP & 4)) {
        if (PN_CB_EXP & 128) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
          for (int j = 0; j < PT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][i], pf[h][j], acc[i][j], 0, 0, 0);
        if (PN_CB_EXP & 128) __builtin_amdgcn_s_setprio(0);
      } else {
#pragma unroll
        for (int i = 0; i < CT; ++i) acc[i][0][0] += (float)wf[h][i][0] + (float)pf[h][i % PT][1];
      }
    }
  };
  const int ngroups = (PN_CB_EXP & 64) ? 0 : 5 * a.kchunks;
  // Patch reload without a bubble (TR = 2): the taps of kernel row 2 (6, 7, 8) read patch rows 2 .. TR + 1 only, so rows 0 and 1 of the NEXT
  // chunk are requested when group (6,7) starts; the other rows when the next chunk starts -- its first taps (0, 1: kernel row 0) read
  // rows 0 .. TR - 1, which at TR = 2 are there already, and the rest has landed a group later (every barrier waits for vmcnt(0)).
  issue_patch(0, 2);
  issue_group(0);
  int pp = 0, chunk = 0;
  for (int g = 0; g < ngroups; ++g) {
    // group g's weight tiles (and patch rows) were requested one group ago: nothing younger is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(PN_CB_EXP & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!(PN_CB_EXP & 1)) {
      if (pp == 0 && chunk > 0) {
        issue_patch(chunk, 1);
        if (a.TR > 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (pp == 3 && chunk + 1 < a.kchunks) issue_patch(chunk + 1, 0);
    }
    if (g + 1 < ngroups && !(PN_CB_EXP & 2)) issue_group(g + 1);
    const char* ws = wst + 2 * (g & 1) * WSTAGE;
    const int t0 = 2 * pp;                                   // taps t0 (and t0 + 1): dy = t / 3, dx = t % 3
    const int dy0 = t0 / 3, dx0 = t0 - 3 * dy0;
    tap_mfma(ws, dy0 * a.PS + dx0, g == 0);
    if (pp < 4) {
      const int t1 = t0 + 1, dy1 = t1 / 3, dx1 = t1 - 3 * dy1;
      tap_mfma(ws + WSTAGE, dy1 * a.PS + dx1, false);
    }
    if (++pp == 5) { pp = 0; ++chunk; }
  }

  // ---- epilogue through LDS: a lane's accumulators are 4 channels of one pixel (8 bytes of bf16) -- stored straight to memory they are
  // partial lines, 7.8 us of a 33 us layer.  The tile goes to LDS as [pixel][128 channels] (row stride + 16 bytes) and leaves as 16 bytes
  // per lane, 16 lanes per pixel row: with out_ps == cout a wave writes 1 KiB of consecutive memory per instruction.
  constexpr int EB = F32OUT ? 4 : 2;                       // bytes per output element
  constexpr int ROWB = kRowsBC * EB + 16;                   // LDS row stride
  constexpr int PASSES = F32OUT ? 2 : 1;                    // f32: 144 pixels at a time (the tile would not fit)
  constexpr int PPX = kRowsBP / PASSES;
  const int cq = (lane >> 4) * 4;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done with the patch and the weight stages
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if (PASSES == 1 || (wp >= 3) == (pass == 1)) {
#pragma unroll
      for (int j = 0; j < PT; ++j) {
        const int p = (wp * PT + j) * 16 + (lane & 15) - pass * PPX;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
          const int nl = (wc * CT + i) * 16 + cq;            // channel inside the block's 128
          f32x4 v = acc[i][j];
          if (a.scale) v = v * *reinterpret_cast<const f32x4*>(a.scale + n0 + nl);
          if (a.shift) v = v + *reinterpret_cast<const f32x4*>(a.shift + n0 + nl);
          if (a.act == PN_ACT_RELU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          char* dst = smem + p * ROWB + nl * EB;
          if constexpr (F32OUT) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
            const unsigned lo = cb_pack_bf16x2(v[0], v[1]);
            const unsigned hi = cb_pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = uint2{lo, hi};
          }
        }
      }
    }
    __syncthreads();
    constexpr int SEGS = kRowsBC * EB / 16;                 // 16-byte segments per pixel row
    for (int idx = tid; idx < PPX * SEGS; idx += kRowsWaves * 64) {
      const int pl = idx / SEGS, seg = idx - pl * SEGS;
      const int p = pl + pass * PPX;
      const int row = p / a.W, col = p - row * a.W;
      const int y = y0 + row;
      if (y >= a.H || ((PN_CB_EXP & 32) && pl > 0)) continue;
      const size_t opix = ((size_t)b * a.H + y) * (size_t)a.W + col;
      const uint4 val = *reinterpret_cast<const uint4*>(smem + pl * ROWB + seg * 16);
      char* o = static_cast<char*>(a.out) + (opix * (size_t)a.out_ps + a.out_co + n0) * EB + seg * 16;
      *reinterpret_cast<uint4*>(o) = val;
    }
    if (pass + 1 < PASSES) __syncthreads();
  }
}

// the rows form takes: 3x3 / stride 1 / pad 1 on maps of 144 columns (TR = 2 rows of 144 = the 288-pixel tile; with more, shorter rows
// the next chunk's first taps would wait for most of the patch: measured 43 against 38 us on the 128 x 72 layers, which stay on the implicit
// GEMM), output channels a multiple of 128
bool rows_form_fits(const pn_conv_desc* d) {
  static const int off = [] { const char* e = getenv("PN_CONV_BF16_ROWS"); return e ? atoi(e) == 0 : 0; }();
  if (off || d->deconv2x2 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
  return d->in_w * 2 == kRowsBP && d->cout % kRowsBC == 0;
}

int launch_rows(const pn_conv_desc* d, CrArgs& a, bool f32out, hipStream_t st) {
  a.TR = kRowsBP / d->in_w;
  a.PS = d->in_w % 16 == 0 ? (d->in_w + 2 + 3) / 4 * 4 : d->in_w + 8;       // a multiple of 4: patch rows 0 and 1 end on an instruction boundary
  a.tiles_per_image = pn::cdiv(d->in_h, a.TR);
  a.ctiles = d->cout / kRowsBC;
  a.patch_rows = (a.TR + 2) * a.PS;
  a.patch_bytes = (a.patch_rows * 128 + 1023) / 1024 * 1024;
  const int lds = a.patch_bytes + kRowsStages * kRowsBC * 128;
  const unsigned grid = (unsigned)(d->batch * a.tiles_per_image * a.ctiles);
  static bool attr_f32[64] = {}, attr_b16[64] = {};
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  if (f32out) {
    auto kern = conv_bf16_rows_kernel<true>;
    if (pn::first_use_on_device(attr_f32)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (prof) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kRowsWaves * 64), lds, st, ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(kRowsWaves * 64), lds, st, a);
  } else {
    auto kern = conv_bf16_rows_kernel<false>;
    if (pn::first_use_on_device(attr_b16)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (prof) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kRowsWaves * 64), lds, st, ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(kRowsWaves * 64), lds, st, a);
  }
  return pn::check_launch("conv_bf16_rows_kernel");
}

// torch (Cout_total, Cin, KH, KW) f32 -> bf16 [rows][tap][cin], rows past cout_total zero
__global__ void pack_conv_weight_bf16_rows_kernel(const float* __restrict__ w, int cout, int cin, int taps, int rows, unsigned short* __restrict__ packed) {
  const size_t total = (size_t)rows * taps * cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cin);
    const int t = (int)((i / cin) % taps);
    const int n = (int)(i / ((size_t)cin * taps));
    packed[i] = n < cout ? f32_to_bf16_bits(w[((size_t)n * cin + c) * taps + t]) : (unsigned short)0;
  }
}

struct CbTile { int wp, wc, pt, ct; };

template <int WP, int WC, int PT, int CT, bool F32OUT, int NS>
int launch_tile_ns(CbArgs& a, hipStream_t st) {
  constexpr int BP = WP * PT * 16, BC = WC * CT * 16;
  const int lds = NS * (BP + BC) * 128;
  auto kern = conv_bf16_igemm_kernel<WP, WC, PT, CT, F32OUT, NS>;
  static bool attr[64] = {};
  if (pn::first_use_on_device(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  pn::ProfileSlot ps{};
  if (pn::take_profile_slot(ps)) hipExtLaunchKernelGGL(kern, dim3((unsigned)(a.ptiles * a.ctiles)), dim3(WP * WC * 64), lds, st, ps.start, ps.stop, 0, a);
  else hipLaunchKernelGGL(kern, dim3((unsigned)(a.ptiles * a.ctiles)), dim3(WP * WC * 64), lds, st, a);
  return pn::check_launch("conv_bf16_igemm_kernel");
}

template <int WP, int WC, int PT, int CT>
int launch_tile(CbArgs& a, bool f32out, hipStream_t st) {
  constexpr int BP = WP * PT * 16, BC = WC * CT * 16;
  a.ptiles = pn::cdiv(a.M, BP);
  a.ctiles = pn::cdiv(a.N, BC);
  // at most one block per CU: four stages, three steps of loads in flight (measured on the 128 x 72 layers: 39.3 against 38.1 us -- no gain,
  // the 6-wave block is MFMA-issue bound there; kept for launches whose blocks would otherwise wait on a single stage)
  const bool deep = a.ptiles * a.ctiles <= 256 && a.ksteps >= 4;
  if constexpr (4 * (BP + BC) * 128 <= 160 * 1024) {
    if (deep) return f32out ? launch_tile_ns<WP, WC, PT, CT, true, 4>(a, st) : launch_tile_ns<WP, WC, PT, CT, false, 4>(a, st);
  }
  return f32out ? launch_tile_ns<WP, WC, PT, CT, true, 2>(a, st) : launch_tile_ns<WP, WC, PT, CT, false, 2>(a, st);
}

// tile id: 0 = 144 px x 128 ch (6 waves), 1 = 144 x 64 (3 waves), 2 = 128 x 128 (4 waves), 3 = 128 x 64 (2 waves)
int pick_tile(const CbArgs& a) {
  static const int forced = [] { const char* e = getenv("PN_CONV_BF16_TILE"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced;
  // fewest idle block slots in the last round over 512 slots (256 CUs x 2 blocks), wide channel tiles first on ties
  int best = 0;
  double best_cost = 1e30;
  const int bp[4] = {144, 144, 128, 128}, bc[4] = {128, 64, 128, 64};
  for (int t = 0; t < 4; ++t) {
    if (bc[t] == 128 && a.N <= 64) continue;
    const long long tiles = (long long)pn::cdiv(a.M, bp[t]) * pn::cdiv(a.N, bc[t]);
    const long long rounds = (tiles + 511) / 512;
    // time ~ rounds x (tile work, with the narrow tiles' lower arithmetic intensity as a 10 % surcharge)
    const double cost = (double)rounds * bp[t] * bc[t] * (bc[t] == 64 ? 1.1 : 1.0);
    if (cost < best_cost) { best_cost = cost; best = t; }
  }
  return best;
}

}  // namespace

extern "C" {

size_t pn_conv_bf16_rows_packed_elems(int cout_total, int cin, int kh, int kw) {
  if (cout_total < 1 || cin < 1 || kh < 1 || kw < 1) return 0;
  return (size_t)((cout_total + 15) / 16 * 16) * kh * kw * cin;
}

int pn_pack_conv_weight_bf16_rows(const float* w_oihw, int cout_total, int cin, int kh, int kw, void* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && cout_total >= 1 && cin >= 1 && kh >= 1 && kw >= 1, "pack_conv_weight_bf16_rows: bad arguments");
  const int rows = (cout_total + 15) / 16 * 16;
  const size_t total = (size_t)rows * kh * kw * cin;
  hipLaunchKernelGGL(pack_conv_weight_bf16_rows_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w_oihw,
                     cout_total, cin, kh * kw, rows, static_cast<unsigned short*>(packed));
  return pn::check_launch("pack_conv_weight_bf16_rows_kernel");
}

int pn_conv2d_igemm_bf16_supported(const pn_conv_desc* d) {
  if (!d) return 0;
  if (d->groups != 1 || d->range_strata > 1 || d->accumulate || d->pad_h_end || d->pad_w_end || d->transpose_hw) return 0;
  if (d->cin < 64 || d->cin % 64 || d->cout < 16 || d->cout % 16) return 0;
  // (the output stores are 16 bytes wide also for bf16 outputs -- 8 elements --, so strides and offsets of 8; the slices must fit
  // their pixel strides: a wider slice would read / write into the next pixel, or past the buffer descriptor for the last one)
  if (d->in_pixel_stride % 8 || d->in_channel_offset % 8 || d->out_pixel_stride % 8 || d->out_channel_offset % 8) return 0;
  if (d->in_channel_offset < 0 || d->out_channel_offset < 0 || d->in_channel_offset + d->cin > d->in_pixel_stride ||
      d->out_channel_offset + d->cout > d->out_pixel_stride)
    return 0;
  if (!(d->act == PN_ACT_NONE || d->act == PN_ACT_RELU)) return 0;
  if (d->deconv2x2 && !(d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0)) return 0;
  if (d->stride < 1 || d->kh < 1 || d->kw < 1 || d->kh > 7 || d->kw > 7) return 0;
  // the buffer descriptors address the map and the packed weights with 32-bit offsets below 2 GiB (larger maps stay on pn_conv2d_nhwc_bf16)
  const unsigned long long in_bytes = (unsigned long long)d->batch * d->in_h * d->in_w * d->in_pixel_stride * 2ull;
  const unsigned long long n_rows = (unsigned long long)(d->deconv2x2 ? 4 * d->cout : d->cout);
  const unsigned long long w_bytes = (n_rows + 15) / 16 * 16 * (unsigned long long)d->kh * d->kw * d->cin * 2ull;
  if (d->batch < 1 || d->in_h < 1 || d->in_w < 1 || in_bytes >= (1ull << 31) || w_bytes >= (1ull << 31)) return 0;
  return 1;
}

int pn_conv2d_igemm_bf16(const pn_conv_desc* d, const void* in_bf16, const void* packed_rows_bf16, const float* scale, const float* shift, void* out,
                         int out_is_f32, pn_stream_t stream) {
  PN_REQUIRE(d && in_bf16 && packed_rows_bf16 && out, "conv_igemm_bf16: null pointer");
  PN_REQUIRE(pn_conv2d_igemm_bf16_supported(d), "conv_igemm_bf16: unsupported layer (cin a multiple of 64, cout of 16, one group, act none / ReLU)");
  PN_REQUIRE(((uintptr_t)in_bf16 & 15) == 0 && ((uintptr_t)packed_rows_bf16 & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)scale & 15) == 0 &&
                 ((uintptr_t)shift & 15) == 0,
             "conv_igemm_bf16: pointers must be 16-byte aligned");
  CbArgs a{};
  a.in = in_bf16; a.w = packed_rows_bf16; a.scale = scale; a.shift = shift; a.out = out;
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.cin = d->cin; a.cout = d->cout; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
  a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.OH = (d->in_h + 2 * d->pad_h - d->kh) / d->stride + 1;
  a.OW = (d->in_w + 2 * d->pad_w - d->kw) / d->stride + 1;
  PN_REQUIRE(a.OH >= 1 && a.OW >= 1, "conv_igemm_bf16: empty output");
  a.in_ps = d->in_pixel_stride; a.in_co = d->in_channel_offset; a.out_ps = d->out_pixel_stride; a.out_co = d->out_channel_offset;
  a.act = d->act; a.deconv = d->deconv2x2;
  a.M = d->batch * a.OH * a.OW;
  a.N = d->deconv2x2 ? 4 * d->cout : d->cout;
  a.kchunks = d->cin / 64;
  a.ksteps = d->kh * d->kw * a.kchunks;
  a.Kw = d->kh * d->kw * d->cin;
  const unsigned long long in_bytes = (unsigned long long)d->batch * d->in_h * d->in_w * d->in_pixel_stride * 2ull;
  const unsigned long long w_bytes = (unsigned long long)((a.N + 15) / 16 * 16) * a.Kw * 2ull;
  PN_REQUIRE(in_bytes < (1ull << 31) && w_bytes < (1ull << 31) && (unsigned long long)a.M * d->out_pixel_stride * 4ull < (1ull << 40),
             "conv_igemm_bf16: map too large for the buffer descriptor");
  a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
  hipStream_t st = pn::S(stream);
  if (rows_form_fits(d)) {
    CrArgs r{};
    r.in = in_bf16; r.w = packed_rows_bf16; r.scale = scale; r.shift = shift; r.out = out;
    r.B = d->batch; r.H = d->in_h; r.W = d->in_w; r.cin = d->cin; r.cout = d->cout;
    r.in_ps = a.in_ps; r.in_co = a.in_co; r.out_ps = a.out_ps; r.out_co = a.out_co; r.act = a.act;
    r.kchunks = a.kchunks; r.ksteps = a.ksteps; r.Kw = a.Kw; r.in_bytes = a.in_bytes; r.w_bytes = a.w_bytes;
    return launch_rows(d, r, out_is_f32 != 0, st);
  }
  switch (pick_tile(a)) {
    case 0: return launch_tile<3, 2, 3, 4>(a, out_is_f32 != 0, st);
    case 1: return launch_tile<3, 1, 3, 4>(a, out_is_f32 != 0, st);
    case 2: return launch_tile<2, 2, 4, 4>(a, out_is_f32 != 0, st);
    default: return launch_tile<2, 1, 4, 4>(a, out_is_f32 != 0, st);
  }
}

// nn.Linear on the same kernel (a 1x1 convolution over m "pixels"): out[m][:n] = act(x[m][:k] @ W^T + bias) (+ residual[m][:n], f32 output only).
// x: bf16 rows of ldx elements; W: pn_pack_conv_weight_bf16_rows(w (n, k), n, k, 1, 1); act none / ReLU / GELU (exact erf)
int pn_linear_bf16(const void* x_bf16, int m, int k, int ldx, const void* packed_rows_bf16, int n, const float* bias, int act, const float* residual,
                   int ldr, void* out, int ldo, int out_is_f32, pn_stream_t stream) {
  PN_REQUIRE(x_bf16 && packed_rows_bf16 && out && m >= 1, "linear_bf16: null pointer");
  PN_REQUIRE(k >= 64 && k % 64 == 0 && n >= 16 && n % 16 == 0 && ldx >= k && ldx % 8 == 0 && ldo >= n && ldo % 4 == 0,
             "linear_bf16: k a multiple of 64, n of 16, ldx of 8, ldo of 4");
  PN_REQUIRE(act == PN_ACT_NONE || act == PN_ACT_RELU || act == PN_ACT_GELU, "linear_bf16: activation none, ReLU or GELU");
  PN_REQUIRE(!residual || (out_is_f32 && ldr >= n && ldr % 4 == 0), "linear_bf16: a residual needs the f32 output and ldr a multiple of 4");
  PN_REQUIRE(((uintptr_t)x_bf16 & 15) == 0 && ((uintptr_t)packed_rows_bf16 & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
                 ((uintptr_t)residual & 15) == 0,
             "linear_bf16: pointers must be 16-byte aligned");
  CbArgs a{};
  a.in = x_bf16; a.w = packed_rows_bf16; a.scale = nullptr; a.shift = bias; a.out = out;
  a.B = 1; a.H = 1; a.W = m; a.OH = 1; a.OW = m; a.cin = k; a.cout = n; a.kh = a.kw = 1; a.stride = 1;
  a.in_ps = ldx; a.out_ps = ldo; a.act = act; a.M = m; a.N = n; a.kchunks = k / 64; a.ksteps = a.kchunks; a.Kw = k;
  a.res = residual; a.res_ps = ldr;
  const unsigned long long in_bytes = (unsigned long long)m * ldx * 2ull, w_bytes = (unsigned long long)((n + 15) / 16 * 16) * k * 2ull;
  PN_REQUIRE(in_bytes < (1ull << 31) && w_bytes < (1ull << 31), "linear_bf16: matrix too large for the buffer descriptor");
  a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
  hipStream_t st = pn::S(stream);
  switch (pick_tile(a)) {
    case 0: return launch_tile<3, 2, 3, 4>(a, out_is_f32 != 0, st);
    case 1: return launch_tile<3, 1, 3, 4>(a, out_is_f32 != 0, st);
    case 2: return launch_tile<2, 2, 4, 4>(a, out_is_f32 != 0, st);
    default: return launch_tile<2, 1, 4, 4>(a, out_is_f32 != 0, st);
  }
}

}  // extern "C"
