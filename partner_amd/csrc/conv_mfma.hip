// BEV 2-D convolutions as implicit GEMM on the gfx950 fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 64 FLOP/clk/SIMD).
//
// GEMM view (per group / range stratum z):
//   Out[m][n] = sum_k A[m][k] * Wp[k][n]
//   m = output pixel (b, oh, ow)  -- NHWC pixel order
//   n = output channel (or (di,dj,n) for the 2x2 transposed conv)
//   k = (kh, kw, cin)             -- cin fastest, 32 channels of one tap per K step
//
// Block = WM x WN waves; each wave owns TM x TN MFMA tiles of 32x32.  Per K step the block
// stages a [BM pixels][32 ch] activation tile (zero-filled outside the map: this is the
// ZeroPad2d / padding=1 of rpn.py:126-134) and a [32][BN] weight tile through LDS, register
// staged and double buffered (loads of step t+1 are in flight while step t is on the MFMA
// pipe; one barrier per step).
//
// LDS images
//   A: [BM][36] floats (rows padded by 4 floats = one ds_read_b128 width): lane (i = l&31,
//      h = l>>5) reads the float4 at row i, k = 8s+4h; 36*i mod 64 hits 16 distinct 4-bank
//      slots for 16 rows that are distinct mod 16 => conflict-free ds_read_b128.
//   B: [8][BN][4] floats -- the packed global layout, so staging is a straight copy and lane
//      (h, j) reads the float4 at k4 = 2s+h, column j: consecutive lanes, consecutive 16 B.
// k order inside an 8-wide k group: MFMA j (0..3) consumes k = 8s+j from lane half 0 and
// k = 8s+4+j from lane half 1, for A and B alike (the sum is order independent up to fp32
// rounding).
#include "pn_common.h"
#include "wino_planes.h"
#include <algorithm>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// element types of the kernel: DT_F32 (f32 in / f32 out), DT_BF16 (bf16 in / bf16 out, f32 accumulate) and
// DT_BF16_F32OUT (bf16 in / f32 out: last layer before an fp32 consumer).  In bytes the LDS images and the
// operand fetch are IDENTICAL for both input types -- a 16-byte fragment is 4 f32 (four 32x32x2 MFMAs) or
// 8 bf16 (one 32x32x16 MFMA) -- so one kernel body serves both; only the channel <-> byte arithmetic, the MFMA
// and the store differ.  A K step is 8 such chunks per row: 32 f32 or 64 bf16 channels.
enum { DT_F32 = 0, DT_BF16 = 1, DT_BF16_F32OUT = 2 };

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

enum { MODE_CONV = 0, MODE_DECONV2 = 1, MODE_STRAT = 2 };

struct ConvArgs {
  const float* in;
  const float* w;
  const float* scale;
  const float* shift;
  float* out;
  int B, H, W, Cin, Cout, OH, OW;
  int KH, KW, stride, pad_h, pad_w;
  int in_ps, in_co, out_ps, out_co;
  int act, mode;
  int M;           // GEMM rows per z
  int OWsub;       // output columns per z window (OW unless stratified)
  int cin_chunks;  // ceil(Cin/32)
  int cout_pad;    // packed column count (multiple of 32)
  int ncols;       // GEMM columns per z
  int nmt;         // m tiles
  int in_group_stride;   // input-channel offset per z (Cin for grouped conv, 0 otherwise)
  unsigned in_bytes;     // size of the input allocation seen through the buffer descriptor
  unsigned w_bytes;      // size of the packed weights
  int force_tile;        // >0: tile override (tuning / tests)
  const float* res;      // optional residual added after the activation (GEMM use)
  int res_ps;
  // gather mode (sparse convolution): GEMM row m = output site, tap t reads input row nbr[m*taps + t] (-1: none)
  const int* nbr;
  const int* n_valid;    // device count of valid output sites (rows beyond it are neither computed nor written)
  int res_pre_act;       // 1: out = act(conv*scale + shift + residual) (residual blocks); 0: act(...) + residual
  // r6 (pn_conv2d_nhwc_planes_f32): the output goes out as the F(4, 3) planes of conv_wchain.hip (wino_planes.h) instead of NHWC -- the
  // epilogue's LDS tiles hold whole map rows (BM % OW == 0), so a quad's neighbour pixels are in the block
  float* planes;
  unsigned plane_floats;
  // ---- statistics of the (affine-applied, pre-activation) output for the GroupNorm-family layer that follows
  // (st_part != null).  Every block writes ONE partial (sum, sum of squares) per column and tile (per-channel groups) or per
  // 32-row segment and wave column (all-channel groups); conv_stats_finalize_kernel (a separate small launch: folding them
  // inside this kernel costs registers, i.e. occupancy, in every variant) adds them in a fixed order and writes the affine
  // table the consumer applies while it loads its input tile.
  float* st_part;        // per-channel: [z][tile][cout_pad][2]; all-channel: [z][segment][WN][2]
  int st_S;              // range strata along the OUTPUT width inside one z (RSNorm on a plain conv: S; otherwise 1)
  int st_cg;             // channel groups of the norm: 1 (all columns) or ncols (per channel)
  const float* st_gamma; // [z or stratum][ncols]
  const float* st_beta;
  float st_eps;
  float* st_ab;          // out, optional: [B][st_ab_S][ncols][2] = (A, B): y = x*A + B
  int st_ab_S;           // strata count of the table (slot = stratum inside z, or z itself for the stratified conv)
  float* st_stat;        // out, optional: [B][st_ab_S][st_cg][2] = (mean, rstd)
  // ---- normalise-on-load (NORM_IN kernels): the input element (b, ih, iw, c) is replaced by relu(x*A + B) with
  // (A, B) = ni_ab[((b*ni_S + iw / (W / ni_S)) * ni_C + c)], zero padding stays zero
  const float* ni_ab;
  int ni_S, ni_C;
  int st_segs_z;         // 32-row segments per z (rounded up to whole tiles)
  int zdim;              // multi-job launches: number of z slices of this job
  int snt2;              // conv_small_n_multi_kernel: this job's blocks are 8 x 16 pixel tiles, two pixels per lane (small_n_tiled2_body)
};

constexpr int BK = 32;
constexpr int A_LD = BK + 4;
#ifndef PN_GATHER_EXP
#define PN_GATHER_EXP 0   // diagnostic build only (tools/micro/gather_ablate.hip): bit 0 neighbour = own row, 1 no input loads, 2 no weight loads, 3 no LDS stores
#endif

template <int WM, int WN, int TM, int TN, int DT, bool GATHER, bool NORM_IN>
__device__ __forceinline__ void conv_body(const ConvArgs& a, const int bid, const int n0, const int z) {
  constexpr int ES = DT == DT_F32 ? 4 : 2;   // bytes per input element
  constexpr int CPC = 16 / ES;               // channels per 16-byte chunk
  constexpr int BKC = 8 * CPC;               // channels per K step
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int A_FLOATS = BM * A_LD;
  constexpr int B_FLOATS = BK * BN;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int A_PER_T = (BM * 8) / NT;  // float4 per thread
  constexpr int B_PER_T = (8 * BN) / NT;
  static_assert((BM * 8) % NT == 0 && (8 * BN) % NT == 0, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of m tiles (neighbouring tiles share halo rows in that XCD's L2).
  // Gather mode: the grid is sized by the site CAPACITY, the live sites are the first *n_valid rows -- deal only the live
  // tiles over the XCDs (dealing all capacity tiles would put every live one on the first XCD or two).
  int mt;
  int m_valid = a.M;
  {
    int nmt = a.nmt;
    if constexpr (GATHER) {
      m_valid = min(a.M, *a.n_valid);
      nmt = (m_valid + BM - 1) / BM;
    }
    if constexpr (GATHER) {
      const int q = nmt >> 3, r = nmt & 7, x = bid & 7, idx = bid >> 3;
      if (idx >= (x < r ? q + 1 : q)) return;  // block uniform, before any barrier
      mt = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
    } else {
      mt = bid;   // the wrappers deal the tiles over the XCDs
    }
  }
  const int m0 = mt * BM;
  int* nbr_s = reinterpret_cast<int*>(smem + 2 * STAGE);  // gather mode: this block's [BM][taps] neighbour rows
  if constexpr (GATHER) {
    if (m0 >= m_valid) return;
    const int tps = a.KH * a.KW;
    for (int i = tid; i < BM * tps; i += NT) {
      const int m = m0 + i / tps;
      nbr_s[i] = m < m_valid ? a.nbr[(size_t)m0 * tps + i] : -1;
    }
    __syncthreads();
    // Taps for which none of this tile's sites has an active neighbour are skipped altogether: sites are stored in
    // key order, so a tile is a run of neighbouring cells and most of the 27 offsets point at empty space together.
    // tap_list[0 .. n) = the live taps in ascending order, tap_list[32] = n.
    int* tap_list = nbr_s + BM * 32;
    if (tid < tps) {
      int any = 0;
      for (int r = 0; r < BM; ++r) any |= nbr_s[r * tps + tid] >= 0;
      tap_list[33 + tid] = any;
    }
    __syncthreads();
    if (tid == 0) {
      int n = 0;
      for (int t = 0; t < tps; ++t)
        if (tap_list[33 + t]) tap_list[n++] = t;
      tap_list[32] = n;
    }
    __syncthreads();
  }

  // ---- global -> register staging through buffer loads --------------------------------------
  // Every per-step address is  descriptor base + per-thread voffset (fixed) + wave-uniform
  // soffset(tap, chunk): no per-step vector address arithmetic.  Taps that fall into the zero
  // padding (and rows / channels past the end) are redirected to voffset = ~0, which the
  // buffer range check turns into zeros -- no branches, no exec masking.
  // The descriptor base is moved back by the largest negative tap offset so that voffset >= 0.
  const int c4 = tid & 7;
  const int ohw = a.OH * a.OWsub;
  const int taps = a.KH * a.KW;
  const long long back = ((long long)a.pad_h * a.W + a.pad_w) * a.in_ps;  // floats
  unsigned a_off[A_PER_T];   // byte offset of (row, tap 0, chunk 0) relative to the shifted base
  unsigned a_mask[A_PER_T];  // bit t: tap t reads inside the map
  unsigned a_grp[NORM_IN ? A_PER_T : 1];
  float* ntab = smem + 2 * STAGE;   // NORM_IN: the (A, B) table [B * ni_S][ni_C][2] of the producing norm, staged once
  if constexpr (NORM_IN) {
    static_assert(!GATHER, "normalise-on-load is not combined with the gather mode");
    const int n = a.B * a.ni_S * a.ni_C * 2;
    for (int i = tid; i < n; i += NT) ntab[i] = a.ni_ab[i];
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < A_PER_T; ++j) {
    const int row = (tid >> 3) + (NT / 8) * j;
    const int m = m0 + row;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int b = mm / ohw;
    const int rem = mm - b * ohw;
    const int oh = rem / a.OWsub;
    const int ow = rem - oh * a.OWsub + (a.mode == MODE_STRAT ? z * a.OWsub : 0);
    const int ih0 = oh * a.stride - a.pad_h, iw0 = ow * a.stride - a.pad_w;
    const long long pix = ((long long)b * a.H + ih0) * a.W + iw0;  // may be negative by at most `back`/in_ps
    a_off[j] = GATHER ? (unsigned)((a.in_co + c4 * CPC) * ES)
                      : (unsigned)((pix * a.in_ps + back + a.in_co + z * a.in_group_stride + c4 * CPC) * ES);
    unsigned mk = 0;
    for (int kh = 0, t = 0; kh < a.KH; ++kh)
      for (int kw = 0; kw < a.KW; ++kw, ++t)
        if (ok && (unsigned)(ih0 + kh) < (unsigned)a.H && (unsigned)(iw0 + kw) < (unsigned)a.W) mk |= 1u << t;
    a_mask[j] = mk;
    if constexpr (NORM_IN) {   // statistics group (sample, range stratum) of the input pixel under each kw, 10 bits each
      const int wsub_in = a.W / a.ni_S;
      unsigned g = 0;
      for (int kw = 0; kw < a.KW; ++kw) {
        const int s_in = min(max((iw0 + kw) / wsub_in, 0), a.ni_S - 1);
        g |= (unsigned)(b * a.ni_S + s_in) << (10 * kw);
      }
      a_grp[j] = g;
    }
  }
  unsigned b_off[B_PER_T];
#pragma unroll
  for (int j = 0; j < B_PER_T; ++j) {
    const int idx = tid + NT * j;
    const int k4 = idx / BN, n = idx - k4 * BN;
    b_off[j] = (n0 + n < a.cout_pad) ? (unsigned)(((size_t)k4 * a.cout_pad + n0 + n) * 16) : 0xffffffffu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in)) - back * ES, 0, a.in_bytes + (unsigned)(back * ES), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.w)) + (size_t)z * taps * a.cin_chunks * 8 * a.cout_pad * 16, 0, a.w_bytes, 0x00020000);
  const int* tap_list = nbr_s + BM * 32;
  // Four-phase data gradient of a 3x3 stride-2 convolution (DECONV2 with 2x2 taps): output phase d = 2*ph + pw uses tap
  // (u, v) only if (ph || !u) && (pw || !v) -- 9 of the 16 (phase, tap) weight blocks are non-zero.  A column tile that lies
  // inside the phases [d_lo, d_hi] runs over the union of their live taps only (packed 4 bits per entry).
  unsigned tapsel = 0;
  int n_sel = 0;
  if (!GATHER && a.mode == MODE_DECONV2 && taps == 4) {
    const int d_lo = n0 / a.Cout, d_hi = min(3, (n0 + BN - 1) / a.Cout);
    unsigned live = 0;
    for (int d = d_lo; d <= d_hi; ++d)
      for (int t = 0; t < 4; ++t)
        if (((d >> 1) || !(t >> 1)) && ((d & 1) || !(t & 1))) live |= 1u << t;
    for (int t = 0; t < 4; ++t)
      if (live >> t & 1) tapsel |= (unsigned)t << (4 * n_sel++);
  }
  const int taps_loop = GATHER ? tap_list[32] : (n_sel ? n_sel : taps);   // taps the K loop runs over
  const int nsteps = taps_loop * a.cin_chunks;

  f32x4 ra[A_PER_T], rb[B_PER_T];

  // K order: channel chunk outermost, taps innermost -- the 9 taps of one 32-channel slab re-read
  // (nearly) the same lines, so the slab stays in this XCD's L2 while it is being used.
  // (tap, chunk) of the next tile to fetch, advanced incrementally: scalar adds only
  int ld_tap = 0, ld_kh = 0, ld_kw = 0, ld_chunk = 0;
  struct LdTag { int tap, kw, chunk; };   // what a register set holds (NORM_IN: needed when the set is stored to LDS)
  auto load_global_to = [&](bool live, f32x4 (&ra)[A_PER_T], f32x4 (&rb)[B_PER_T], LdTag& tag) {
    int tap = GATHER ? __builtin_amdgcn_readfirstlane(tap_list[ld_tap]) : ld_tap;
    int kh = ld_kh, kw = ld_kw;
    if (!GATHER && n_sel) {  // block-uniform: the live 2x2 taps of this column tile
      tap = (int)(tapsel >> (4 * ld_tap)) & 15;
      kh = tap >> 1;
      kw = tap & 1;
    }
    tag.tap = live ? tap : 31;   // 31: no tap (every mask bit clear)
    tag.kw = kw;
    tag.chunk = ld_chunk;
    const unsigned so_a = GATHER ? (unsigned)(ld_chunk * BKC * ES) : (unsigned)(((kh * a.W + kw) * a.in_ps + ld_chunk * BKC) * ES);
    const unsigned so_b = (unsigned)((tap * a.cin_chunks + ld_chunk) * 8) * (unsigned)a.cout_pad * 16u;
    // `live` == false (past the last K step): every lane is redirected out of range, the loads
    // return zeros without touching memory and the loop body stays branch-free
    const unsigned cok = (unsigned)(live && ld_chunk * BKC + c4 * CPC < a.Cin);
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      unsigned vo;
      if constexpr (GATHER) {
        int idx = nbr_s[((tid >> 3) + (NT / 8) * j) * taps + tap];   // input row of this (site, tap), -1: inactive
        if constexpr (PN_GATHER_EXP & 1) idx = min(m0 + (tid >> 3) + (NT / 8) * j, m_valid - 1);
        vo = (idx >= 0 && cok) ? (unsigned)idx * (unsigned)(a.in_ps * ES) + a_off[j] : 0xffffffffu;
        if constexpr (PN_GATHER_EXP & 2) vo = 0xffffffffu;
      } else {
        const unsigned sel = (a_mask[j] >> tap) & cok;             // 1: inside the map
        vo = a_off[j] | (0u - (1u - (sel & 1u)));                  // branch-free: ~0 when outside
      }
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, so_a, 0));
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (live && !(GATHER && (PN_GATHER_EXP & 4))) ? b_off[j] : 0xffffffffu, live ? so_b : 0u, 0));
    if (++ld_kw == a.KW) { ld_kw = 0; ++ld_kh; }
    if (++ld_tap == taps_loop) { ld_tap = 0; ld_kh = 0; ld_kw = 0; ++ld_chunk; }
  };
  auto store_lds_from = [&](int buf, const f32x4 (&ra)[A_PER_T], const f32x4 (&rb)[B_PER_T], const LdTag& tag) {
    float* As = smem + buf * STAGE;
    float* Bs = As + A_FLOATS;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const int row = (tid >> 3) + (NT / 8) * j;
      f32x4 v = ra[j];
      if constexpr (NORM_IN) {
        // the producing layer's GroupNorm + ReLU, applied on the way into LDS: y = max(x*A + B, 0); padding stays zero
        const int ch = tag.chunk * BKC + c4 * CPC;
        const bool sel = ((a_mask[j] >> tag.tap) & 1u) && ch < a.Cin;
        const int gi = (int)(a_grp[j] >> (10 * tag.kw)) & 1023;
        const float* tp = ntab + ((size_t)gi * a.ni_C + (sel ? ch + z * a.in_group_stride : 0)) * 2;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
        v[0] = sel ? fmaxf(fmaf(v[0], t0[0], t0[1]), 0.f) : 0.f;
        v[1] = sel ? fmaxf(fmaf(v[1], t0[2], t0[3]), 0.f) : 0.f;
        v[2] = sel ? fmaxf(fmaf(v[2], t1[0], t1[1]), 0.f) : 0.f;
        v[3] = sel ? fmaxf(fmaf(v[3], t1[2], t1[3]), 0.f) : 0.f;
      }
      *reinterpret_cast<f32x4*>(As + row * A_LD + c4 * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j) *reinterpret_cast<f32x4*>(Bs + (tid + NT * j) * 4) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- software pipeline -----------------------------------------------------------------
  // One barrier per K step; everything else sits in the shadow of the MFMAs:
  //   sub-step 0: read fragments of sub-step 1            | MFMAs of sub-step 0
  //   sub-step 1: read fragments of sub-step 2            | MFMAs of sub-step 1
  //               store the (landed) tile of step t+1 into the other LDS stage
  //   sub-step 2: read fragments of sub-step 3            | MFMAs of sub-step 2
  //               issue the buffer loads of step t+3 into the registers just stored (two register sets alternate:
  //               the tile of step t+2 is in flight in the other one)
  //   barrier     (all LDS stores of step t+1 landed, all reads of this stage issued)
  //   sub-step 3: read fragments of step t+1, sub-step 0  | MFMAs of sub-step 3
  // The stage written in step t was last read before the barrier of step t-1 (WAR safe); it is
  // first read after the barrier of step t (RAW safe).
  const int li = lane & 31, lh = lane >> 5;
  const int a_frag = (wm * TM * 32 + li) * A_LD + lh * 4;
  const int b_frag = A_FLOATS + (lh * BN + wn * TN * 32 + li) * 4;
  f32x4 af[2][TM], bf[2][TN];
  auto read_frags = [&](int buf, int sub, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
    const float* As = smem + buf * STAGE + a_frag + sub * 8;
    const float* Bs = smem + buf * STAGE + b_frag + sub * 2 * BN * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * A_LD);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 4);
  };
  auto mfma_sub = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
    if constexpr (DT == DT_F32) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
    } else {  // the 16-byte fragment is A[row][k = 8h .. 8h+7] / B[k = 8h .. 8h+7][col]: one 32x32x16 MFMA
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                             acc[i][j], 0, 0, 0);
    }
  };

  // Prefetch distance = two K steps (two register sets, used alternately): with one block per CU nothing else covers the
  // L2 / HBM latency of a tile, and a distance of one step (~1 us) left ~11 % of the loop waiting on vmcnt.
  // Prologue: the loads of steps 0, 1 and 2 are issued back to back (step 0 into a third set that is dead afterwards), so
  // the block pays one global-memory latency before its first MFMA.
  f32x4 ra2[A_PER_T], rb2[B_PER_T];
  LdTag tag1, tag2;
  {
    f32x4 ra0[A_PER_T], rb0[B_PER_T];
    LdTag tag0;
    load_global_to(true, ra0, rb0, tag0);
    load_global_to(nsteps > 1, ra, rb, tag1);
    load_global_to(nsteps > 2, ra2, rb2, tag2);
    store_lds_from(0, ra0, rb0, tag0);
  }
  __syncthreads();
  read_frags(0, 0, af[0], bf[0]);

  constexpr int NM = (DT == DT_F32 ? 4 : 1) * TM * TN;  // MFMAs per sub-step
  constexpr int NF = TM + TN;              // fragment reads per sub-step
  constexpr int NS = A_PER_T + B_PER_T;    // LDS stores == buffer loads per step
  auto kstep = [&](int t, int buf, f32x4 (&rx)[A_PER_T], f32x4 (&ry)[B_PER_T], LdTag& tag) {
    // sub-step 0: fragments of sub-step 1, then MFMAs
    read_frags(buf, 1, af[1], bf[1]);
    mfma_sub(af[0], bf[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_barrier(0);
    // sub-step 1: fragments of sub-step 2; the LDS stores of step t+1 are spread between the MFMAs
    read_frags(buf, 2, af[0], bf[0]);
    mfma_sub(af[1], bf[1]);
    store_lds_from(buf ^ 1, rx, ry, tag);  // the tile of step t+1 (past the last step: zeros into a stage nobody reads)
    __builtin_amdgcn_sched_group_barrier(0x100, NF, 1);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / NS > 0 ? NM / NS : 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 1);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 1);
    __builtin_amdgcn_sched_barrier(0);
    // sub-step 2: fragments of sub-step 3; the buffer loads of step t+3 (into the registers just stored) are spread between the MFMAs
    read_frags(buf, 3, af[1], bf[1]);
    mfma_sub(af[0], bf[0]);
    load_global_to(t + 3 < nsteps, rx, ry, tag);
    __builtin_amdgcn_sched_group_barrier(0x100, NF, 2);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / NS > 0 ? NM / NS : 1, 2);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 2);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 2);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // sub-step 3: first fragments of step t+1 (other stage), then MFMAs
    read_frags(buf ^ 1, 0, af[0], bf[0]);
    mfma_sub(af[1], bf[1]);
    __builtin_amdgcn_sched_group_barrier(0x100, NF, 3);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 3);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nsteps; t += 2) {
    kstep(t, 0, ra, rb, tag1);
    if (t + 1 < nsteps) kstep(t + 1, 1, ra2, rb2, tag2);
  }

  // ---- epilogue: per-channel affine + activation, NHWC store ----------------------------
  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // scale/shift of this lane's TN columns are fetched once (the output may alias nothing, but the
  // compiler cannot know that and would otherwise reload them after every store).
  // Fast path: transpose the wave's (TM*32) x (TN*32) tile through LDS (the staging buffers are
  // free now) so that every lane stores 16 contiguous bytes: 4x fewer store instructions than
  // the accumulator layout allows (the store tail is issue-bound, not bandwidth-bound).
  const bool vec_ok = (a.Cout % 4 == 0) && (a.out_ps % 4 == 0) && (a.out_co % 4 == 0) && (a.ncols % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  if (vec_ok) {
    constexpr int TR = TM * 32, TC = TN * 32, TLD = TC + 4;
    static_assert(WM * WN * TR * TLD <= 2 * STAGE, "epilogue tile does not fit the staging LDS");
    __syncthreads();  // every wave is done reading the last K-step tiles
    float* tile = smem + wave * (TR * TLD);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + j * 32 + li] = acc[i][j][r];
    if (a.planes) {
      // ---- planes epilogue: item = (channel quad of the block's columns, pixel quad of the block's rows), pixel quads fastest (a wave's plane
      // stores are runs of a plane row).  Pixel p, column c of the block tile lie in wave (p / TR, c / TC)'s LDS tile.
      __syncthreads();      // the tiles are read across waves here
      constexpr int QT = BM / 4, C4T = BN / 4;
      const int Wq = a.OW >> 2, c4n = a.Cout >> 2;
      for (int it = tid; it < QT * C4T; it += NT) {
        const int c4l = it / QT, q = it - c4l * QT;
        const int gn = n0 + c4l * 4, gm0 = m0 + 4 * q;
        if (gn >= a.ncols || gm0 >= m_valid) continue;
        const int rowi = gm0 / a.OW, xq = (gm0 - rowi * a.OW) >> 2;
        const int img = rowi / a.OH, r = rowi - img * a.OH;
        f32x4 vs = {1.f, 1.f, 1.f, 1.f}, vh = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) vs = *reinterpret_cast<const f32x4*>(a.scale + gn);
        if (a.shift) vh = *reinterpret_cast<const f32x4*>(a.shift + gn);
        const int cl = c4l * 4;
        const float* tcol = smem + (size_t)(cl / TC) * (TR * TLD) + (cl % TC);      // + wave row wm * WN * (TR * TLD) + row * TLD
        f32x4 d[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int p = 4 * q - 1 + i;                 // pixel of the block tile; out of the map row: zero padding
          const bool inside = (i > 0 || xq > 0) && (i < 5 || xq + 1 < Wq);
          const int pc = inside ? p : 4 * q;
          const f32x4 v = *reinterpret_cast<const f32x4*>(tcol + (size_t)(pc / TR) * (WN * TR * TLD) + (pc % TR) * TLD);
#pragma unroll
          for (int k = 0; k < 4; ++k) d[i][k] = inside ? pn::apply_act(fmaf(v[k], vs[k], vh[k]), a.act) : 0.f;
        }
        f32x4 vv[6];
        pn::wino4_input_transform4(d, vv);
        pn::wino4_store_planes(a.planes, vv, gn >> 2, c4n, a.plane_floats, img, r, xq, a.OH, Wq);
      }
      return;
    }
    // lanes: c4 = 16-byte column group, rr = row within a pass
    constexpr int CG = TC / 4;            // column groups per row (8 or 16)
    constexpr int RPP = 64 / CG;          // rows per pass
    const int cg = lane % CG, rr = lane / CG;
    const int gn = n0 + wn * TC + cg * 4;  // first GEMM column of this lane
    const bool cok = gn < a.ncols;
    int ch = gn, sidx = z * a.Cout + gn, d = 0;
    if (a.mode == MODE_DECONV2) {
      d = gn / a.Cout;
      ch = gn - d * a.Cout;
      sidx = ch;
    }
    f32x4 vs = {1.f, 1.f, 1.f, 1.f}, vh = {0.f, 0.f, 0.f, 0.f};
    if (cok && a.scale) vs = *reinterpret_cast<const f32x4*>(a.scale + sidx);
    if (cok && a.shift) vh = *reinterpret_cast<const f32x4*>(a.shift + sidx);
    const int coff = a.out_co + (a.mode == MODE_CONV ? z * a.Cout : 0) + ch;
#pragma unroll 4
    for (int p = 0; p < TR / RPP; ++p) {
      const int row = p * RPP + rr;
      const int gm = m0 + wm * TR + row;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * TLD + cg * 4);
      if (gm >= m_valid || !cok) continue;
      size_t pix;
      if (a.mode == MODE_CONV) {
        pix = (size_t)gm;
      } else {
        const int b = gm / ohw;
        const int rem = gm - b * ohw;
        const int oh = rem / a.OWsub;
        const int ow = rem - oh * a.OWsub + (a.mode == MODE_STRAT ? z * a.OWsub : 0);
        pix = a.mode == MODE_DECONV2 ? ((size_t)b * (2 * a.OH) + 2 * oh + (d >> 1)) * (2 * a.OW) + 2 * ow + (d & 1)
                                     : ((size_t)b * a.OH + oh) * a.OW + ow;
      }
      f32x4 o;
      if (a.res && a.res_pre_act) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + pix * a.res_ps + coff);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pn::apply_act(fmaf(v[k], vs[k], vh[k]) + rv[k], a.act);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pn::apply_act(fmaf(v[k], vs[k], vh[k]), a.act);
        if (a.res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + pix * a.res_ps + coff);
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] += rv[k];
        }
      }
      if constexpr (DT == DT_BF16) {
        using u16x4 = __attribute__((ext_vector_type(4))) unsigned short;
        u16x4 ob;
#pragma unroll
        for (int k = 0; k < 4; ++k) ob[k] = f32_to_bf16_rne(o[k]);
        *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(a.out) + pix * a.out_ps + coff) = ob;
      } else {
        *reinterpret_cast<f32x4*>(a.out + pix * a.out_ps + coff) = o;
      }
    }
    if (a.st_part) {
      // ---- output statistics, reduced inside the block before they leave it:
      //   per-channel groups: ONE (sum, sum of squares) per column for the whole tile (the host guarantees a tile never
      //                       straddles two samples)              -> part[z][tile][cout_pad]
      //   all-channel groups: one pair per 32-row segment and wave column (lanes reduced by the xor butterfly)
      //                                                            -> part[z][segment][WN]
      constexpr int LPC = 64 / TC > 0 ? 64 / TC : 1;   // lanes per column (TC = 32: two, 16 rows each; TC = 64: one, 32 rows)
      constexpr int RPL = 32 / LPC;
      __shared__ float s_red[WM][BN][2];
      const int col = lane % TC, part_h = lane / TC;
      const int gcol = n0 + wn * TC + col;
      float csc = 1.f, csh = 0.f;
      if (gcol < a.ncols) {
        if (a.scale) csc = a.scale[z * a.Cout + gcol];
        if (a.shift) csh = a.shift[z * a.Cout + gcol];
      }
      const bool per_channel = a.st_cg > 1;
      float2* part = reinterpret_cast<float2*>(a.st_part);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int r = 0; r < RPL; ++r) {
          const int row = i * 32 + part_h * RPL + r;
          const float v = fmaf(tile[row * TLD + col], csc, csh);
          const bool live = (m0 + wm * TR + row < m_valid) && gcol < a.ncols;
          s1 += live ? v : 0.f;
          s2 += live ? v * v : 0.f;
        }
        if (per_channel) {
          t1 += s1;
          t2 += s2;
        } else {
          s1 = pn::wave_sum(s1);
          s2 = pn::wave_sum(s2);
          if (lane == 0) {
            const int seg = (m0 + wm * TR) / 32 + i;
            part[((size_t)z * a.st_segs_z + seg) * WN + wn] = make_float2(s1, s2);
          }
        }
      }
      if (per_channel) {
        if constexpr (LPC == 2) {
          t1 += __shfl_xor(t1, 32, 64);
          t2 += __shfl_xor(t2, 32, 64);
        }
        if (part_h == 0) {
          s_red[wm][wn * TC + col][0] = t1;
          s_red[wm][wn * TC + col][1] = t2;
        }
        __syncthreads();
        if (tid < BN && tid < a.cout_pad) {
          float u1 = 0.f, u2 = 0.f;
#pragma unroll
          for (int k = 0; k < WM; ++k) {
            u1 += s_red[k][tid][0];
            u2 += s_red[k][tid][1];
          }
          part[((size_t)z * a.nmt + mt) * a.cout_pad + tid] = make_float2(u1, u2);
        }
      }
    }
    return;
  }
  float sc[TN], sh[TN];
  int col_off[TN];   // channel offset inside the output pixel, -1: column out of range
  int col_d[TN];     // deconv: which of the 2x2 positions
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int gn = n0 + wn * TN * 32 + j * 32 + li;
    const bool ok = gn < a.ncols;
    int ch = gn, sidx = z * a.Cout + gn, d = 0;
    if (a.mode == MODE_DECONV2) {
      d = gn / a.Cout;
      ch = gn - d * a.Cout;
      sidx = ch;
    }
    col_d[j] = d;
    col_off[j] = ok ? a.out_co + (a.mode == MODE_CONV ? z * a.Cout : 0) + ch : -1;
    sc[j] = (ok && a.scale) ? a.scale[sidx] : 1.f;
    sh[j] = (ok && a.shift) ? a.shift[sidx] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int gm = m0 + row;
      if (gm >= m_valid) continue;
      size_t pix;  // output pixel index (deconv: the top-left of the 2x2 cell)
      int b = 0, oh = 0, ow = 0;
      if (a.mode == MODE_CONV) {
        pix = (size_t)gm;
      } else {
        b = gm / ohw;
        const int rem = gm - b * ohw;
        oh = rem / a.OWsub;
        ow = rem - oh * a.OWsub + (a.mode == MODE_STRAT ? z * a.OWsub : 0);
        pix = ((size_t)b * a.OH + oh) * a.OW + ow;
      }
      float* orow = a.out + pix * a.out_ps;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (col_off[j] < 0) continue;
        float* dst = orow + col_off[j];
        size_t rpix = pix;
        if (a.mode == MODE_DECONV2) {
          rpix = ((size_t)b * (2 * a.OH) + 2 * oh + (col_d[j] >> 1)) * (2 * a.OW) + 2 * ow + (col_d[j] & 1);
          dst = a.out + rpix * a.out_ps + col_off[j];
        }
        float o;
        if (a.res && a.res_pre_act) {
          o = pn::apply_act(fmaf(acc[i][j][r], sc[j], sh[j]) + a.res[rpix * a.res_ps + col_off[j]], a.act);
        } else {
          o = pn::apply_act(fmaf(acc[i][j][r], sc[j], sh[j]), a.act);
          if (a.res) o += a.res[rpix * a.res_ps + col_off[j]];
        }
        if constexpr (DT == DT_BF16)
          reinterpret_cast<unsigned short*>(a.out)[dst - a.out] = f32_to_bf16_rne(o);
        else
          *dst = o;
      }
    }
  }
}


// single convolution: grid (m tiles, column tiles, z)
template <int WM, int WN, int TM, int TN, int DT = DT_F32, bool GATHER = false, bool NORM_IN = false>
__global__ __launch_bounds__(WM* WN * 64) void conv_mfma_kernel(ConvArgs a) {
  constexpr int BN = WN * TN * 32;
  int bid = blockIdx.x;
  if constexpr (!GATHER) {
    // XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of m tiles
    // (neighbouring tiles share halo rows in that XCD's L2)
    const int nmt = a.nmt, q = nmt >> 3, r = nmt & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  conv_body<WM, WN, TM, TN, DT, GATHER, NORM_IN>(a, bid, blockIdx.y * BN, blockIdx.z);
}

// several convolutions of one tile shape as ONE launch (the branches of a detection head: same map, different weights /
// input slices / output slices): the tiles of all jobs form one list, dealt over the XCDs in contiguous runs
constexpr int kMaxJobs = 8;
struct MultiArgs {
  int njobs, total;
  int first[kMaxJobs + 1];   // first tile of every job in the flattened list
  ConvArgs job[kMaxJobs];
};

template <int WM, int WN, int TM, int TN, bool NORM_IN>
__global__ __launch_bounds__(WM* WN * 64) void conv_multi_kernel(MultiArgs m_by_value) {
  constexpr int BN = WN * TN * 32;
  // The job table is read through the kernarg segment pointer, in the CONSTANT address space and at a block-uniform offset
  // (scalar loads into SGPRs, once): indexing the by-value parameter dynamically would make the compiler copy all of it to
  // scratch, and reading it through a generic pointer reloads every field inside the K loop.
  typedef const __attribute__((address_space(4))) int* kptr_t;
  const kptr_t base = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int njobs = base[offsetof(MultiArgs, njobs) / 4], nt = base[offsetof(MultiArgs, total) / 4];
  const int bid = blockIdx.x;
  const int q = nt >> 3, r = nt & 7, x = bid & 7, idx = bid >> 3;
  const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  int j = 0;
  for (int k = 1; k < njobs; ++k)
    if (t >= base[offsetof(MultiArgs, first) / 4 + k]) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = t - base[offsetof(MultiArgs, first) / 4 + j];
  static_assert(sizeof(ConvArgs) % 4 == 0 && offsetof(MultiArgs, job) % 4 == 0, "job table must be dword aligned");
  constexpr int JW = sizeof(ConvArgs) / 4;
  union { ConvArgs a; int w[JW]; } u;
  const kptr_t src = base + offsetof(MultiArgs, job) / 4 + j * JW;
#pragma unroll
  for (int i = 0; i < JW; ++i) u.w[i] = src[i];
  const ConvArgs& a = u.a;
  const int ntn = (a.ncols + BN - 1) / BN, per_z = a.nmt * ntn;
  const int z = local / per_z, rem = local - z * per_z;
  const int ny = rem / a.nmt, mt = rem - ny * a.nmt;
  conv_body<WM, WN, TM, TN, DT_F32, false, NORM_IN>(a, mt, ny * BN, z);
}


// ------------------------------------------------------------------------------------------
// Fold of the statistics partials a convolution's epilogue wrote (fixed order, fp64) -> affine table (A, B) / (mean, rstd).
// One block per (job, z, sample, 16-column chunk) for per-channel groups, per (job, z, sample, stratum) for all-channel ones.
struct FinArgs {
  int njobs, total;
  int first[kMaxJobs + 1];
  int bm, wn;
  ConvArgs job[kMaxJobs];
};

__device__ __forceinline__ void group_geometry(const ConvArgs& a, int& pix_b, int& wsub, int& spr) {
  pix_b = a.OH * a.OWsub;
  wsub = a.OWsub / a.st_S;
  spr = a.st_S > 1 ? wsub / 32 : 1;
}

// (sum, sum of squares) of the all-channel group (z, b, s) from part[z][segment][wn]: every thread of the block calls this;
// the result is the same on all of them (fixed association order)
template <int NT>
__device__ __forceinline__ void fold_allch_group(const ConvArgs& a, const float2* part, int wn_count, int z, int b, int s, double* red,
                                                 double& o1, double& o2) {
  int pix_b, wsub, spr;
  group_geometry(a, pix_b, wsub, spr);
  const int tid = threadIdx.x;
  const int count = (a.st_S == 1 ? (pix_b + 31) / 32 : a.OH * spr) * wn_count;
  double t1 = 0.0, t2 = 0.0;
  for (int e = tid; e < count; e += NT) {
    const int i = e / wn_count, w = e - i * wn_count;
    const int seg = a.st_S == 1 ? (b * pix_b) / 32 + i : ((b * a.OH + i / spr) * a.OWsub + s * wsub) / 32 + i % spr;
    const float2 v = part[((size_t)z * a.st_segs_z + seg) * wn_count + w];
    t1 += (double)v.x;
    t2 += (double)v.y;
  }
  t1 = pn::wave_sum(t1);   // xor butterfly: the same association order on every run
  t2 = pn::wave_sum(t2);
  __syncthreads();
  if ((tid & 63) == 0) {
    red[tid >> 6] = t1;
    red[NT / 64 + (tid >> 6)] = t2;
  }
  __syncthreads();
  o1 = 0.0;
  o2 = 0.0;
  for (int k = 0; k < NT / 64; ++k) {
    o1 += red[k];
    o2 += red[NT / 64 + k];
  }
}

__global__ __launch_bounds__(256) void conv_stats_finalize_kernel(FinArgs by_value) {
  constexpr int NT = 256;
  typedef const __attribute__((address_space(4))) int* kptr_t;
  const kptr_t base = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int njobs = base[offsetof(FinArgs, njobs) / 4];
  const int BM = base[offsetof(FinArgs, bm) / 4], WNc = base[offsetof(FinArgs, wn) / 4];
  const int t = blockIdx.x;
  int j = 0;
  for (int k = 1; k < njobs; ++k)
    if (t >= base[offsetof(FinArgs, first) / 4 + k]) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = t - base[offsetof(FinArgs, first) / 4 + j];
  constexpr int JW = sizeof(ConvArgs) / 4;
  union { ConvArgs a; int w[JW]; } u;
  const kptr_t src = base + offsetof(FinArgs, job) / 4 + j * JW;
#pragma unroll
  for (int i = 0; i < JW; ++i) u.w[i] = src[i];
  const ConvArgs& a = u.a;

  __shared__ double red[2 * NT];
  const int tid = threadIdx.x;
  const int cp = a.cout_pad, ncols = a.ncols;
  int pix_b, wsub, spr;
  group_geometry(a, pix_b, wsub, spr);
  const float2* part = reinterpret_cast<const float2*>(a.st_part);
  if (a.st_cg > 1) {
    // per-channel groups: part[z][tile][cp]
    const int chunks = (cp + 15) / 16;
    const int z = local / (a.B * chunks), rem = local - z * (a.B * chunks), b = rem / chunks, chunk = rem - b * chunks;
    const int c = chunk * 16 + (tid & 15), sl = tid >> 4;   // 16 slices of the sample's tiles
    const int tiles_b = a.B == 1 ? a.nmt : pix_b / BM;
    const float2* pb = part + ((size_t)z * a.nmt + (size_t)b * tiles_b) * cp + c;
    double t1 = 0.0, t2 = 0.0;
    if (c < cp) {
#pragma unroll 8
      for (int i = sl; i < tiles_b; i += 16) {
        const float2 v = pb[(size_t)i * cp];
        t1 += (double)v.x;
        t2 += (double)v.y;
      }
    }
    red[tid] = t1;
    red[NT + tid] = t2;
    __syncthreads();
    if (tid < 16 && c < ncols) {
      double u1 = 0.0, u2 = 0.0;
      for (int k = 0; k < 16; ++k) {
        u1 += red[k * 16 + tid];
        u2 += red[NT + k * 16 + tid];
      }
      const double n = (double)pix_b;
      const double mean = u1 / n;
      double var = u2 / n - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float rstd = (float)(1.0 / sqrt(var + (double)a.st_eps));
      const int slot = a.mode == MODE_STRAT ? z : 0;
      const size_t grp = (size_t)b * a.st_ab_S + slot;
      const float ga = a.st_gamma ? a.st_gamma[slot * ncols + c] : 1.f, be = a.st_beta ? a.st_beta[slot * ncols + c] : 0.f;
      const float A = ga * rstd;
      if (a.st_ab) reinterpret_cast<float2*>(a.st_ab)[grp * ncols + c] = make_float2(A, be - (float)mean * A);
      if (a.st_stat) reinterpret_cast<float2*>(a.st_stat)[grp * ncols + c] = make_float2((float)mean, rstd);
    }
    return;
  }
  // all-channel groups: part[z][segment][WN]
  const int S = a.st_S;
  const int z = local / (a.B * S), rem = local - z * (a.B * S), b = rem / S, s = rem - b * S;
  double u1, u2;
  fold_allch_group<NT>(a, part, WNc, z, b, s, red, u1, u2);
  const double n = (double)a.OH * wsub * ncols;
  const double mean = u1 / n;
  double var = u2 / n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float fmean = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)a.st_eps));
  const int slot = (a.mode == MODE_STRAT ? z : 0) + s;
  const size_t grp = (size_t)b * a.st_ab_S + slot;
  if (tid == 0 && a.st_stat) reinterpret_cast<float2*>(a.st_stat)[grp] = make_float2(fmean, rstd);
  if (a.st_ab)
    for (int c = tid; c < ncols; c += NT) {
      const float ga = a.st_gamma ? a.st_gamma[slot * ncols + c] : 1.f, be = a.st_beta ? a.st_beta[slot * ncols + c] : 0.f;
      const float A = ga * rstd;
      reinterpret_cast<float2*>(a.st_ab)[grp * ncols + c] = make_float2(A, be - fmean * A);
    }
}

// RSNorm / GroupNorm(1 group per stratum) + activation (+ calibrated copy) of a convolution output whose all-channel statistics
// partials were written by its epilogue: every block first folds the partials of ITS (sample, stratum) group (a few hundred
// entries), then normalises its rows -- no statistics pass over the map, no finalize launch.
struct ApplyArgs {
  ConvArgs conv;     // the producing convolution (geometry + st_part); its `out` is the input here
  int wn;            // wave columns of the producing tile
  int splits, rows_per_split;
  const float* gamma;  // [stratum][C]
  const float* beta;
  int act;
  float* out;
  int ops, oco;
  const float* mul;    // (H, W, C) maps, optional: out2 = out * mul + add
  const float* add;
  float* out2;
  int o2ps, o2co;
};

__global__ __launch_bounds__(256) void conv_stats_apply_kernel(ApplyArgs g) {
  constexpr int NT = 256;
  __shared__ double red[2 * (NT / 64)];
  const ConvArgs& a = g.conv;
  const int split = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  double u1, u2;
  fold_allch_group<NT>(a, reinterpret_cast<const float2*>(a.st_part), g.wn, 0, b, s, red, u1, u2);
  const int H = a.OH, W = a.OW, C = a.ncols;
  const int wsub = W / a.st_S;
  const double n = (double)H * wsub * C;
  const double dmean = u1 / n;
  double var = u2 / n - dmean * dmean;
  var = var < 0.0 ? 0.0 : var;
  const float mean = (float)dmean, rstd = (float)(1.0 / sqrt(var + (double)a.st_eps));
  const int vpc = C / 4;
  const int cv = threadIdx.x % vpc, pl = threadIdx.x / vpc, ppb = NT / vpc;
  float ga[4], be[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cv * 4 + k;
    ga[k] = g.gamma ? g.gamma[s * C + c] : 1.f;
    be[k] = g.beta ? g.beta[s * C + c] : 0.f;
  }
  const int y0 = split * g.rows_per_split, y1 = min(H, y0 + g.rows_per_split);
  const int npix = (y1 - y0) * wsub;
  for (int p = pl; p < npix; p += ppb) {
    const int y = y0 + p / wsub, x = s * wsub + p % wsub;
    const size_t pix = (size_t)(b * H + y) * W + x;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.out + pix * a.out_ps + a.out_co + cv * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = pn::apply_act((v[k] - mean) * rstd * ga[k] + be[k], g.act);
    *reinterpret_cast<f32x4*>(g.out + pix * g.ops + g.oco + cv * 4) = v;
    if (g.out2) {
      const size_t q = ((size_t)y * W + x) * C + cv * 4;
      const f32x4 m = *reinterpret_cast<const f32x4*>(g.mul + q);
      const f32x4 d = *reinterpret_cast<const f32x4*>(g.add + q);
      f32x4 w;
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = v[k] * m[k] + d[k];
      *reinterpret_cast<f32x4*>(g.out2 + pix * g.o2ps + g.o2co + cv * 4) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight packing: torch (Cout_total, Cin_g, KH, KW) -> [g][tap][cin_pad/4][cout_pad][4]
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int cout_g, int cin_g, int kh, int kw,
                                        int groups, int cin_pad, int cout_pad, float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3;
    r >>= 2;
    const int n = r % cout_pad;
    r /= cout_pad;
    const int k4 = r % (cin_pad / 4);
    r /= (cin_pad / 4);
    const int tap = r % (kh * kw);
    const int g = (int)(r / (kh * kw));
    const int c = k4 * 4 + k1;
    float v = 0.f;
    if (n < cout_g && c < cin_g) v = w[(((size_t)(g * cout_g + n) * cin_g + c) * kh + tap / kw) * kw + tap % kw];
    packed[i] = v;
  }
}

// ConvTranspose2d weight (Cin, Cout, 2, 2) -> 1x1 packed with columns (d = di*2+dj, n)
__global__ void pack_deconv_weight_kernel(const float* __restrict__ w, int cin, int cout, int cin_pad, int cout_pad,
                                          float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3;
    r >>= 2;
    const int col = r % cout_pad;
    const int k4 = (int)(r / cout_pad);
    const int c = k4 * 4 + k1;
    float v = 0.f;
    if (col < 4 * cout && c < cin) {
      const int d = col / cout, n = col - d * cout;
      v = w[((size_t)c * cout + n) * 4 + d];
    }
    packed[i] = v;
  }
}

__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               const float* bias, float eps, int c, float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float s = gamma[i] / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = beta[i] + ((bias ? bias[i] : 0.f) - mean[i]) * s;
}

// direct convolution, one thread per output element (load-time constant folding / cross-check)
__global__ void conv_direct_kernel(ConvArgs a, const float* __restrict__ w_oihw, int groups, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int n = r % (a.Cout * groups);
    r /= (a.Cout * groups);
    const int ow = r % a.OW;
    r /= a.OW;
    const int oh = r % a.OH;
    const int b = (int)(r / a.OH);
    const int g = n / a.Cout;
    float acc = 0.f;
    for (int kh = 0; kh < a.KH; ++kh) {
      const int ih = oh * a.stride - a.pad_h + kh;
      if ((unsigned)ih >= (unsigned)a.H) continue;
      for (int kw = 0; kw < a.KW; ++kw) {
        const int iw = ow * a.stride - a.pad_w + kw;
        if ((unsigned)iw >= (unsigned)a.W) continue;
        const float* ip = a.in + ((size_t)(b * a.H + ih) * a.W + iw) * a.in_ps + a.in_co + g * a.Cin;
        const float* wp = w_oihw + (size_t)n * a.Cin * a.KH * a.KW + kh * a.KW + kw;
        for (int c = 0; c < a.Cin; ++c) acc = fmaf(ip[c], wp[(size_t)c * a.KH * a.KW], acc);
      }
    }
    const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
    a.out[((size_t)(b * a.OH + oh) * a.OW + ow) * a.out_ps + a.out_co + n] = pn::apply_act(fmaf(acc, sc, sh), a.act);
  }
}

// ------------------------------------------------------------------------------------------
// Convolutions with very few output channels (the last layer of every head branch: 64 -> 1 / 2 / 3 / 10).  On the MFMA
// tile they are latency bound (a 64 x 32 tile does 16 MFMAs per K step but waits a full global round trip for each of its
// 18 steps: 19 us for 0.2 GFLOP).  Here one block = 64 output pixels x 4 waves; wave q owns a quarter of the input
// channels, lane = pixel, the weights of the wave's channels are wave-uniform (scalar loads feeding v_fmac), the four
// partial sums meet in LDS.  1024 waves for a 128 x 128 map, ~2300 FMAs each.
constexpr int kSmallN = 8;    // 10 .. 16 columns measured slower than the 64 x 32 MFMA tile inside a frame (32 vs 19 us)

template <int NOUT, int TAPS>
__global__ __launch_bounds__(256) void conv_small_n_kernel(ConvArgs a) {
  __shared__ float part[4][64][NOUT + 1];
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), z = blockIdx.y;
  const int m = blockIdx.x * 64 + lane;
  const bool live = m < a.M;
  const int mm = live ? m : 0;
  const int ohw = a.OH * a.OW;
  const int b = mm / ohw, rem = mm - b * ohw, oh = rem / a.OW, ow = rem - oh * a.OW;
  const int cq = ((a.Cin + 15) / 16) * 4;                 // channels per wave: whole quads, at most 16 (Cin <= 64)
  const int c0 = q * cq, c1 = min(a.Cin, c0 + cq);
  // every input value of the wave's pixel x channel-quarter is requested before the first one is used: with one
  // wave per SIMD there is nobody else to hide the latency behind
  f32x4 x[TAPS][4];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int kh = t / a.KW, kw = t - kh * a.KW;
    const int ih = oh * a.stride - a.pad_h + kh, iw = ow * a.stride - a.pad_w + kw;
    const bool in = live && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    const float* ip = a.in + ((size_t)(b * a.H + (in ? ih : 0)) * a.W + (in ? iw : 0)) * a.in_ps + a.in_co + z * a.in_group_stride;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = c0 + 4 * cc;
      x[t][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (in && c < c1) x[t][cc] = *reinterpret_cast<const f32x4*>(ip + c);
    }
  }
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
  // packed weights [g][tap][cin_pad/4][cout_pad][4]: the NOUT x 4 block of (n, k) for one channel quad is contiguous
  const int quads = a.cin_chunks * 8;
  const float* wz = a.w + ((size_t)z * TAPS * quads + (c0 >> 2)) * a.cout_pad * 4;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      if (c0 + 4 * cc < c1) {                                                   // wave-uniform
        const float* wp = wz + ((size_t)t * quads + cc) * a.cout_pad * 4;       // wave-uniform address -> scalar loads
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {   // columns past Cout are zero in the packed weights
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[n] = fmaf(x[t][cc][k], wp[n * 4 + k], acc[n]);
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) part[q][lane][n] = acc[n];
  __syncthreads();
  // thread (pixel p = tid & 63, output group g = tid >> 6): outputs n = g, g + 4, ...
  const int p = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int mo = blockIdx.x * 64 + p;
  if (mo >= a.M) return;
  for (int n = g; n < a.Cout; n += 4) {
    const float v = (part[0][p][n] + part[1][p][n]) + (part[2][p][n] + part[3][p][n]);
    const int sidx = z * a.Cout + n;
    const float sc = a.scale ? a.scale[sidx] : 1.f, sh = a.shift ? a.shift[sidx] : 0.f;
    a.out[(size_t)mo * a.out_ps + a.out_co + z * a.Cout + n] = pn::apply_act(fmaf(v, sc, sh), a.act);
  }
}


// ------------------------------------------------------------------------------------------
// The LAST convolutions of a detection head as one launch: a handful of output columns each (64 -> 1 .. 10), input read through
// the affine table of the GroupNorm that precedes it (relu(x*A + B), padding stays zero).  An MFMA tile is 3 % used at these
// widths (49 us for the five of CenterHeadSinglePos on 64 x 32 tiles); here block = 64 pixels x 4 waves of ONE job, wave q
// owns a quarter of the job's input channels, lane = pixel: all 9 x 4 float4 of the lane are requested before the first is used,
// the weights and (for per-channel norms) the affine pairs are wave-uniform scalar loads feeding v_fma, the four partial sums
// meet in LDS.  Jobs with range-stratified norm tables (ni_S > 1) are supported for 1x1 kernels (per-lane table rows).
template <int NOUT, int TAPS>
__device__ __forceinline__ void small_n_body(const ConvArgs& a, const int mtile, float (*part)[64][13], float* w_lds) {
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {  // stage the job's weights: packed global [tap][cin_pad/4][cout_pad][4] -> LDS [tap][16 quads][NOUT][4]
    const int quads = a.cin_chunks * 8;
    for (int i = threadIdx.x; i < TAPS * 16 * NOUT; i += 256) {
      const int n = i % NOUT, qd = (i / NOUT) % 16, t = i / (NOUT * 16);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (qd < quads && n < a.cout_pad) v = *reinterpret_cast<const f32x4*>(a.w + (((size_t)t * quads + qd) * a.cout_pad + n) * 4);
      *reinterpret_cast<f32x4*>(w_lds + (size_t)i * 4) = v;
    }
  }
  const int m = mtile * 64 + lane;
  const bool live = m < a.M;
  const int mm = live ? m : 0;
  const int ohw = a.OH * a.OW;
  const int b = mm / ohw, rem = mm - b * ohw, oh = rem / a.OW, ow = rem - oh * a.OW;
  const int cq = ((a.Cin + 15) / 16) * 4;                 // channels per wave: whole quads, at most 16 (Cin <= 64)
  const int c0 = q * cq, c1 = min(a.Cin, c0 + cq);
  f32x4 x[TAPS][4];
  unsigned inmask = 0;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int kh = t / a.KW, kw = t - kh * a.KW;
    const int ih = oh * a.stride - a.pad_h + kh, iw = ow * a.stride - a.pad_w + kw;
    const bool in = live && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    inmask |= (unsigned)in << t;
    const float* ip = a.in + ((size_t)(b * a.H + (in ? ih : 0)) * a.W + (in ? iw : 0)) * a.in_ps + a.in_co;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = c0 + 4 * cc;
      x[t][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (in && c < c1) x[t][cc] = *reinterpret_cast<const f32x4*>(ip + c);
    }
  }
  if (a.ni_ab) {
    if (a.ni_S == 1) {
      // per-channel norm: the (A, B) pairs of the wave's channels are wave-uniform -> scalar operands
      const float* tab = a.ni_ab + ((size_t)b * a.ni_C + c0) * 2;   // b is wave-uniform when a sample is a whole number of 64-pixel tiles
      const float* tb = (const float*)__builtin_assume_aligned(tab, 8);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if (c0 + 4 * cc < c1) {
          float A[4], Bv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            A[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tb[(cc * 4 + k) * 2])));
            Bv[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tb[(cc * 4 + k) * 2 + 1])));
          }
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            const bool in = (inmask >> t) & 1u;
#pragma unroll
            for (int k = 0; k < 4; ++k) x[t][cc][k] = in ? fmaxf(fmaf(x[t][cc][k], A[k], Bv[k]), 0.f) : 0.f;
          }
        }
      }
    } else {
      // range-stratified norm (TAPS == 1): the lane's table row depends on its input column
      const int wsub_in = a.W / a.ni_S;
      const int iw = ow * a.stride - a.pad_w;
      const int s_in = min(max(iw / wsub_in, 0), a.ni_S - 1);
      const float* tab = a.ni_ab + ((size_t)(b * a.ni_S + s_in) * a.ni_C + c0) * 2;
      const bool in = inmask & 1u;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        if (c0 + 4 * cc < c1) {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(tab + cc * 8), t1 = *reinterpret_cast<const f32x4*>(tab + cc * 8 + 4);
          x[0][cc][0] = in ? fmaxf(fmaf(x[0][cc][0], t0[0], t0[1]), 0.f) : 0.f;
          x[0][cc][1] = in ? fmaxf(fmaf(x[0][cc][1], t0[2], t0[3]), 0.f) : 0.f;
          x[0][cc][2] = in ? fmaxf(fmaf(x[0][cc][2], t1[0], t1[1]), 0.f) : 0.f;
          x[0][cc][3] = in ? fmaxf(fmaf(x[0][cc][3], t1[2], t1[3]), 0.f) : 0.f;
        }
      }
    }
  }
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
  __syncthreads();   // the staged weights are complete
  // weights of the job from LDS (staged by the block, layout [tap][quad][NOUT][4]): a wave-uniform ds_read_b128 is a broadcast
  // and pipelines 16 deep; the same values as wave-uniform SCALAR loads made every (tap, quad) group wait for the scalar
  // cache (the whole launch took 61 us instead of ~15)
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      if (c0 + 4 * cc < c1) {                                                   // wave-uniform
        const float* wp = w_lds + ((t * 16 + (c0 >> 2) + cc) * NOUT) * 4;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {   // columns past Cout are zero in the packed weights
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + n * 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[n] = fmaf(x[t][cc][k], wv[k], acc[n]);
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) part[q][lane][n] = acc[n];
  __syncthreads();
  // thread (pixel p = tid & 63, output group g = tid >> 6): outputs n = g, g + 4, ...
  const int p = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int mo = mtile * 64 + p;
  if (mo >= a.M) return;
  for (int n = g; n < a.Cout; n += 4) {
    const float v = (part[0][p][n] + part[1][p][n]) + (part[2][p][n] + part[3][p][n]);
    const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
    a.out[(size_t)mo * a.out_ps + a.out_co + n] = pn::apply_act(fmaf(v, sc, sh), a.act);
  }
}

// r3: the 3x3 / stride 1 / pad 1 jobs on a TILE of 4 rows x 16 columns (lane = pixel, wave q = a quarter of the input channels as above).
// r2 had every lane fetch its own nine 64-byte pieces: each input element crossed the L1 nine times, from 64 different lines per
// wave instruction (41.7 us for the five branches of CenterHeadSinglePos, 55 MB fetched for 21 MB of input).  Here wave q stages the
// 6 x 18 halo tile of its 16 channels through LDS once -- four lanes per pixel piece, 16 pieces per instruction, the producing
// GroupNorm's relu(x A + B) applied on the way in (once per element, not once per tap; out-of-map pixels stay zero) -- and the nine
// taps of a lane are ds_read_b128 at a pixel stride of 20 floats (16-byte slot 5 p mod 16: conflict-free over 16 consecutive pixels).
constexpr int SNT_W = 16, SNT_H = 4, SNT_HW = SNT_W + 2, SNT_PX = (SNT_H + 2) * SNT_HW, SNT_LD = 20;
template <int NOUT>
__device__ __forceinline__ void small_n_tiled_body(const ConvArgs& a, const int tile, float (*part)[64][13], float* w_lds, float* xt_all) {
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tx_n = a.OW / SNT_W, ty_n = a.OH / SNT_H;
  const int b = tile / (tx_n * ty_n), tr = tile - b * (tx_n * ty_n), ty = tr / tx_n, tx = tr - ty * tx_n;
  const int oh0 = ty * SNT_H, ow0 = tx * SNT_W;
  const int cq = ((a.Cin + 15) / 16) * 4;                 // channels per wave: whole quads, at most 16 (Cin <= 64)
  const int c0 = q * cq, c1 = min(a.Cin, c0 + cq);
  float* xt = xt_all + q * (SNT_PX * SNT_LD);
  // Both staging loops are branch-free (clamped addresses, values selected afterwards): every load of a thread is in flight before the
  // first is used -- under a branch the compiler serialises them, one L2 round trip per iteration.
  constexpr int XI = (SNT_PX * 4 + 63) / 64;      // tile items per lane: (halo pixel, channel quad of the wave's 16)
  f32x4 xv[XI], t0[XI], t1[XI];
  bool xin[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = lane + 64 * i;
    const int px = min(idx >> 2, SNT_PX - 1), cc = idx & 3;
    const int hr = px / SNT_HW, hc = px - hr * SNT_HW;
    const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
    const int c = c0 + 4 * cc;
    xin[i] = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && c < c1;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1), ccl = min(c, a.Cin - 4);
    xv[i] = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(b * a.H + ihc) * a.W + iwc) * a.in_ps + a.in_co + ccl);
    if (a.ni_ab) {      // wave-uniform: the producing norm's (A, B) pairs of the four channels
      const float* tab = a.ni_ab + ((size_t)b * a.ni_C + ccl) * 2;
      t0[i] = *reinterpret_cast<const f32x4*>(tab);
      t1[i] = *reinterpret_cast<const f32x4*>(tab + 4);
    }
  }
  {  // the job's weights: packed global [tap][cin_pad/4][cout_pad][4] -> LDS [tap][16 quads][NOUT][4]
    const int quads = a.cin_chunks * 8;
    constexpr int WI = (9 * 16 * NOUT + 255) / 256;
    f32x4 wv[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k) {
      const int i = min((int)threadIdx.x + 256 * k, 9 * 16 * NOUT - 1);
      const int n = i % NOUT, qd = (i / NOUT) % 16, t = i / (NOUT * 16);
      const bool ok = qd < quads && n < a.cout_pad;
      wv[k] = *reinterpret_cast<const f32x4*>(a.w + (((size_t)t * quads + min(qd, quads - 1)) * a.cout_pad + min(n, a.cout_pad - 1)) * 4);
      if (!ok) wv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < WI; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < 9 * 16 * NOUT) *reinterpret_cast<f32x4*>(w_lds + (size_t)i * 4) = wv[k];
    }
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = lane + 64 * i;
    f32x4 v = xv[i];
    if (a.ni_ab) {
      v[0] = fmaxf(fmaf(v[0], t0[i][0], t0[i][1]), 0.f);
      v[1] = fmaxf(fmaf(v[1], t0[i][2], t0[i][3]), 0.f);
      v[2] = fmaxf(fmaf(v[2], t1[i][0], t1[i][1]), 0.f);
      v[3] = fmaxf(fmaf(v[3], t1[i][2], t1[i][3]), 0.f);
    }
    if (!xin[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx < SNT_PX * 4) *reinterpret_cast<f32x4*>(xt + (idx >> 2) * SNT_LD + (idx & 3) * 4) = v;
  }
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
  __syncthreads();   // the staged weights and tiles are complete
  const int r = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kh = t / 3, kw = t - kh * 3;
    const float* xp = xt + ((r + kh) * SNT_HW + cl + kw) * SNT_LD;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      if (c0 + 4 * cc < c1) {                                                   // wave-uniform
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + cc * 4);
        const float* wp = w_lds + ((t * 16 + (c0 >> 2) + cc) * NOUT) * 4;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {   // columns past Cout are zero in the packed weights
#ifdef PN_SNT_EXP
          const f32x4 wv = {(float)n, (float)t, (float)cc, 1.f};      // diagnostic build: no weight reads
          (void)wp;
#else
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + n * 4);
#endif
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[n] = fmaf(xv[k], wv[k], acc[n]);
        }
      }
    }
  }
  __syncthreads();   // (the partial sums share the tiles' LDS: every wave is done reading)
#pragma unroll
  for (int n = 0; n < NOUT; ++n) part[q][lane][n] = acc[n];
  __syncthreads();
  // thread (pixel p = tid & 63, output group g = tid >> 6): outputs n = g, g + 4, ...
  const int p = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t mo = ((size_t)(b * a.OH + oh0 + (p >> 4)) * a.OW + ow0 + (p & 15));
  for (int n = g; n < a.Cout; n += 4) {
    const float v = (part[0][p][n] + part[1][p][n]) + (part[2][p][n] + part[3][p][n]);
    const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
    a.out[mo * a.out_ps + a.out_co + n] = pn::apply_act(fmaf(v, sc, sh), a.act);
  }
}

// r4: the same tile form with TWO pixels per lane -- an 8 x 16 tile, lane (r, c) owns rows r and r + 4.  The nine-tap loop of the form above
// issues one broadcast ds_read_b128 of weights per FOUR v_fma; the block's four waves share one LDS pipe (8 cycles per 64-lane b128 read),
// so the loop ran at the LDS's pace, not the VALU's (a build without the weight reads: 28 -> 15 us for the five branches).  With two pixels
// a weight read feeds eight v_fma.  The halo tile is 10 x 18 pixels x 16 channels per wave; instead of padding a pixel to 20 floats its four
// 16-byte slots are rotated by (pixel / 4) mod 4 -- 16 consecutive pixels x one slot then hit 16 different slot columns, conflict-free like
// the padded layout -- so four waves' tiles + the weights stay under 80 KB (two blocks per CU).  The partial sums reuse the tile's LDS.
// Same order of additions per output as the one-pixel form: same bits.
constexpr int SN2_H = 8, SN2_PX = (SN2_H + 2) * SNT_HW;
__device__ __forceinline__ int sn2_slot(int px, int cc) { return 4 * px + ((cc + (px >> 2)) & 3); }
template <int NOUT>
__device__ __forceinline__ void small_n_tiled2_body(const ConvArgs& a, const int tile, float* w_lds, float* xt_all) {
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tx_n = a.OW / SNT_W, ty_n = a.OH / SN2_H;
  const int b = tile / (tx_n * ty_n), tr = tile - b * (tx_n * ty_n), ty = tr / tx_n, tx = tr - ty * tx_n;
  const int oh0 = ty * SN2_H, ow0 = tx * SNT_W;
  const int cq = ((a.Cin + 15) / 16) * 4;                 // channels per wave: whole quads, at most 16 (Cin <= 64)
  const int c0 = q * cq, c1 = min(a.Cin, c0 + cq);
  float* xt = xt_all + q * (SN2_PX * 16);
  constexpr int XI = (SN2_PX * 4 + 63) / 64;      // tile items per lane: (halo pixel, channel quad of the wave's 16); the quad is lane & 3 for all
  f32x4 xv[XI];
  bool xin[XI];
  const int cc_l = lane & 3, c_l = c0 + 4 * cc_l, ccl = min(c_l, a.Cin - 4);
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int px = min((lane + 64 * i) >> 2, SN2_PX - 1);
    const int hr = px / SNT_HW, hc = px - hr * SNT_HW;
    const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
    xin[i] = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && c_l < c1;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);
    xv[i] = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(b * a.H + ihc) * a.W + iwc) * a.in_ps + a.in_co + ccl);
  }
  f32x4 t0 = {1.f, 0.f, 1.f, 0.f}, t1 = {1.f, 0.f, 1.f, 0.f};
  if (a.ni_ab) {      // the producing norm's (A, B) pairs of the lane's four channels
    const float* tab = a.ni_ab + ((size_t)b * a.ni_C + ccl) * 2;
    t0 = *reinterpret_cast<const f32x4*>(tab);
    t1 = *reinterpret_cast<const f32x4*>(tab + 4);
  }
  {  // the job's weights: packed global [tap][cin_pad/4][cout_pad][4] -> LDS [tap][16 quads][NOUT][4]
    const int quads = a.cin_chunks * 8;
    constexpr int WI = (9 * 16 * NOUT + 255) / 256;
    f32x4 wv[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k) {
      const int i = min((int)threadIdx.x + 256 * k, 9 * 16 * NOUT - 1);
      const int n = i % NOUT, qd = (i / NOUT) % 16, t = i / (NOUT * 16);
      const bool ok = qd < quads && n < a.cout_pad;
      wv[k] = *reinterpret_cast<const f32x4*>(a.w + (((size_t)t * quads + min(qd, quads - 1)) * a.cout_pad + min(n, a.cout_pad - 1)) * 4);
      if (!ok) wv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LDS [tap][quad][n / 2][k / 2][k & 1][n & 1]: a 16-byte read holds the weights of TWO outputs for two channels, the (w[n][k], w[n + 1][k])
    // pairs v_pk_fma_f32 takes as they lie
#pragma unroll
    for (int k = 0; k < WI; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < 9 * 16 * NOUT) {
        const int n = i % NOUT, tq = i / NOUT;
        float* dst = w_lds + ((size_t)tq * (NOUT / 2) + (n >> 1)) * 8 + (n & 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[(c >> 1) * 4 + (c & 1) * 2] = wv[k][c];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = lane + 64 * i;
    f32x4 v = xv[i];
    if (a.ni_ab) {
      v[0] = fmaxf(fmaf(v[0], t0[0], t0[1]), 0.f);
      v[1] = fmaxf(fmaf(v[1], t0[2], t0[3]), 0.f);
      v[2] = fmaxf(fmaf(v[2], t1[0], t1[1]), 0.f);
      v[3] = fmaxf(fmaf(v[3], t1[2], t1[3]), 0.f);
    }
    if (!xin[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx < SN2_PX * 4) *reinterpret_cast<f32x4*>(xt + sn2_slot(idx >> 2, cc_l) * 4) = v;
  }
  // outputs n, n + 1 of one pixel as the halves of a v_pk_fma_f32 (two IEEE fmas per instruction: the bits of two v_fma_f32 at half the VALU
  // issue): the weight pairs come out of LDS adjacent, the pixel's channel value is selected by op_sel -- no register moves to build operands
  // (the compiler's own pairing of the two PIXELS spent 1.2 v_mov per packed fma on it)
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  static_assert(NOUT % 2 == 0, "outputs are processed in pairs");
  f32x2 acc0[NOUT / 2], acc1[NOUT / 2];
#pragma unroll
  for (int n = 0; n < NOUT / 2; ++n) acc0[n] = acc1[n] = f32x2{0.f, 0.f};
  __syncthreads();   // the staged weights and tiles are complete
  const int r = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kh = t / 3, kw = t - kh * 3;
    const int p0 = (r + kh) * SNT_HW + cl + kw, p1 = p0 + 4 * SNT_HW;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      if (c0 + 4 * cc < c1) {                                                   // wave-uniform
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xt + sn2_slot(p0, cc) * 4);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xt + sn2_slot(p1, cc) * 4);
        const float* wp = w_lds + ((t * 16 + (c0 >> 2) + cc) * NOUT) * 4;
#pragma unroll
        for (int n = 0; n < NOUT / 2; ++n) {   // columns past Cout are zero in the packed weights
          const f32x4 wa = *reinterpret_cast<const f32x4*>(wp + n * 8), wb = *reinterpret_cast<const f32x4*>(wp + n * 8 + 4);
          const f32x2 w[4] = {f32x2{wa[0], wa[1]}, f32x2{wa[2], wa[3]}, f32x2{wb[0], wb[1]}, f32x2{wb[2], wb[3]}};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            acc0[n] = __builtin_elementwise_fma(f32x2{x0[k], x0[k]}, w[k], acc0[n]);
            acc1[n] = __builtin_elementwise_fma(f32x2{x1[k], x1[k]}, w[k], acc1[n]);
          }
        }
      }
    }
  }
  __syncthreads();   // every wave is done with its tile: the partial sums go into the same LDS
  float (*part)[128][13] = reinterpret_cast<float (*)[128][13]>(xt_all);
#pragma unroll
  for (int n = 0; n < NOUT; ++n) { part[q][lane][n] = acc0[n >> 1][n & 1]; part[q][64 + lane][n] = acc1[n >> 1][n & 1]; }
  __syncthreads();
  // thread (pixel p = tid & 127: rows 0 .. 3 then 4 .. 7, output group g = tid >> 7): outputs n = g, g + 2, ...
  const int p = threadIdx.x & 127, g = threadIdx.x >> 7;
  const size_t mo = ((size_t)(b * a.OH + oh0 + ((p & 63) >> 4) + 4 * (p >> 6)) * a.OW + ow0 + (p & 15));
  for (int n = g; n < a.Cout; n += 2) {
    const float v = (part[0][p][n] + part[1][p][n]) + (part[2][p][n] + part[3][p][n]);
    const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
    a.out[mo * a.out_ps + a.out_co + n] = pn::apply_act(fmaf(v, sc, sh), a.act);
  }
}

// r5: the same 8 x 16 tile on the MATRIX pipe.  The two-pixel form above is bound by the LDS pipe again -- a broadcast ds_read_b128 of weights
// costs the pipe as much as a read of 64 different addresses (8 cycles), 504 of them per wave and tile, two blocks per CU: 13 us of LDS time per
// pair of tiles -- and its weights (28 KB of LDS) keep a CU at two blocks.  v_mfma_f32_16x16x4_f32 takes the weights as a B fragment that lives in
// REGISTERS for the whole kernel (lane (n, j): w[tap][4 i + j][n], 36 registers for the wave's 16 channels x 9 taps, outputs padded to 16
// columns) and the pixels as the A operand straight from the tile (lane (m, j): channel 4 i + j of pixel m of an output row: one ds_read_b32,
// conflict-free by the slot rotation): 288 MFMAs per wave and tile instead of 1728 packed fmas + 504 LDS reads, no weights in LDS (three
// blocks per CU: the 768 tiles of the nuScenes head are ONE round).  An MFMA is a k-ordered fmaf chain (guide, "FP32-input MFMA") and the
// steps are issued in the order of the loop above (tap, channel quad, channel), so the outputs are bit-identical to it.
template <int KQ>      // channel quads per wave (cq / 4): compile time, so that the K loop is ONE basic block (a wave-uniform branch per step cut it into
                       // 120 blocks and the register allocator moved all eight accumulators at every boundary: 37 us instead of 15)
__device__ __forceinline__ void small_n_mfma_body(const ConvArgs& a, const int tile, float* xt_all) {
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tx_n = a.OW / SNT_W, ty_n = a.OH / SN2_H;
  const int b = tile / (tx_n * ty_n), tr = tile - b * (tx_n * ty_n), ty = tr / tx_n, tx = tr - ty * tx_n;
  const int oh0 = ty * SN2_H, ow0 = tx * SNT_W;
  const int cq = ((a.Cin + 15) / 16) * 4;                 // channels per wave: whole quads, at most 16 (Cin <= 64)
  const int c0 = q * cq, c1 = min(a.Cin, c0 + cq);
  float* xt = xt_all + q * (SN2_PX * 16);
  constexpr int XI = (SN2_PX * 4 + 63) / 64;
  f32x4 xv[XI];
  bool xin[XI];
  const int cc_l = lane & 3, c_l = c0 + 4 * cc_l, ccl = min(c_l, a.Cin - 4);
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int px = min((lane + 64 * i) >> 2, SN2_PX - 1);
    const int hr = px / SNT_HW, hc = px - hr * SNT_HW;
    const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
    xin[i] = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && c_l < c1;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);
    xv[i] = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(b * a.H + ihc) * a.W + iwc) * a.in_ps + a.in_co + ccl);
  }
  f32x4 t0 = {1.f, 0.f, 1.f, 0.f}, t1 = {1.f, 0.f, 1.f, 0.f};
  if (a.ni_ab) {      // the producing norm's (A, B) pairs of the lane's four channels
    const float* tab = a.ni_ab + ((size_t)b * a.ni_C + ccl) * 2;
    t0 = *reinterpret_cast<const f32x4*>(tab);
    t1 = *reinterpret_cast<const f32x4*>(tab + 4);
  }
  // B fragments: lane (n = lane & 15, j = lane >> 4) holds w[tap][channel c0 + 4 i + j][n] of the packed [tap][cin_pad / 4][cout_pad][4]
  const int mn = lane & 15, j = lane >> 4;
  const int quads = a.cin_chunks * 8;
  float wr[9][KQ];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
      const int quad = (c0 >> 2) + i;
      const bool ok = quad < quads && c0 + 4 * i < c1 && mn < a.cout_pad;
      const float v = a.w[(((size_t)t * quads + min(quad, quads - 1)) * a.cout_pad + min(mn, a.cout_pad - 1)) * 4 + j];
      wr[t][i] = ok ? v : 0.f;
    }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = lane + 64 * i;
    f32x4 v = xv[i];
    if (a.ni_ab) {
      v[0] = fmaxf(fmaf(v[0], t0[0], t0[1]), 0.f);
      v[1] = fmaxf(fmaf(v[1], t0[2], t0[3]), 0.f);
      v[2] = fmaxf(fmaf(v[2], t1[0], t1[1]), 0.f);
      v[3] = fmaxf(fmaf(v[3], t1[2], t1[3]), 0.f);
    }
    if (!xin[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx < SN2_PX * 4) *reinterpret_cast<f32x4*>(xt + sn2_slot(idx >> 2, cc_l) * 4) = v;
  }
  f32x4 acc[SN2_H];
#pragma unroll
  for (int g = 0; g < SN2_H; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // the tiles are complete
  // halo row rr, column offset kw: the pixel value is read ONCE and feeds the (up to) three output rows g = rr - kh it is a tap (kh, kw) of.
  // Output row g still sees its steps in the order (kh, kw, channel) -- rr ascending is kh ascending for a fixed g.
#pragma unroll
  for (int rr = 0; rr < SN2_H + 2; ++rr) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int p = rr * SNT_HW + mn + kw;
      const float* px = xt + 16 * p + j;
      const int r0 = p >> 2;
#pragma unroll
      for (int i = 0; i < KQ; ++i) {      // quads past the wave's last channel: zeros in the tile and in wr (the last wave of a Cin that is no multiple of 16)
        const float xa = px[((i + r0) & 3) * 4];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int g = rr - kh;
          if (g >= 0 && g < SN2_H) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wr[kh * 3 + kw][i], acc[g], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();   // every wave is done with its tile: the partial sums go into the same LDS
  float (*part)[128][13] = reinterpret_cast<float (*)[128][13]>(xt_all);
  if (mn < 13) {     // D: lane holds rows (pixels of the output row) 4 j + r, column (output) mn
#pragma unroll
    for (int g = 0; g < SN2_H; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[q][g * 16 + 4 * j + r][mn] = acc[g][r];
  }
  __syncthreads();
  // thread (pixel p = tid & 127 = 16 row + column, output group g = tid >> 7): outputs n = g, g + 2, ...
  const int p = threadIdx.x & 127, g2 = threadIdx.x >> 7;
  const size_t mo = ((size_t)(b * a.OH + oh0 + (p >> 4)) * a.OW + ow0 + (p & 15));
  for (int n = g2; n < a.Cout; n += 2) {
    const float v = (part[0][p][n] + part[1][p][n]) + (part[2][p][n] + part[3][p][n]);
    const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
    a.out[mo * a.out_ps + a.out_co + n] = pn::apply_act(fmaf(v, sc, sh), a.act);
  }
}

__host__ __device__ __forceinline__ bool small_n_tiled_ok(const ConvArgs& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad_h == 1 && a.pad_w == 1 && a.OW % SNT_W == 0 && a.OH % SNT_H == 0 && a.OH == a.H && a.OW == a.W &&
         (a.ni_ab == nullptr || a.ni_S == 1) && (a.in_ps % 4) == 0 && (a.in_co % 4) == 0;
}

// One to three output channels (every branch of a centre head but the heat map): even the 16-column MFMA tile above is 6 - 19 % used.  The
// G form uses the columns for (tap, output) instead: G[halo pixel][(t, n)] = sum_c x[pixel][c] w[t][c][n] is ONE 1 x 1 GEMM over the 10 x 18
// halo pixels (12 row tiles, 9 Cout <= 27 columns = NT <= 2 column tiles, K = Cin), a wave takes three row tiles with ALL channels, and
// out[p][n] = sum_t G[p + offset(t)][(t, n)] is a nine-term sum per output read back from LDS: 96 MFMAs per wave instead of 288.
// Summation order: channels first (a k-ordered fma chain per tap, the MFMA's), then the taps in order -- not the order of the forms above.
template <int NT>
__device__ __forceinline__ void small_n_gform_body(const ConvArgs& a, const int tile, float* xt_all) {
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tx_n = a.OW / SNT_W, ty_n = a.OH / SN2_H;
  const int b = tile / (tx_n * ty_n), tr = tile - b * (tx_n * ty_n), ty = tr / tx_n, tx = tr - ty * tx_n;
  const int oh0 = ty * SN2_H, ow0 = tx * SNT_W;
  // loader: wave q brings channels 16 q .. 16 q + 15 of every halo pixel (chunks past Cin: zeros), the image of chunk q is xt_all + q * SN2_PX * 16
  const int c0 = q * 16, c1 = min(a.Cin, c0 + 16);
  float* xt = xt_all + q * (SN2_PX * 16);
  constexpr int XI = (SN2_PX * 4 + 63) / 64;
  f32x4 xv[XI];
  bool xin[XI];
  const int cc_l = lane & 3, c_l = c0 + 4 * cc_l, ccl = max(0, min(c_l, a.Cin - 4));
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int px = min((lane + 64 * i) >> 2, SN2_PX - 1);
    const int hr = px / SNT_HW, hc = px - hr * SNT_HW;
    const int ih = oh0 - 1 + hr, iw = ow0 - 1 + hc;
    xin[i] = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && c_l < c1;
    const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);
    xv[i] = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(b * a.H + ihc) * a.W + iwc) * a.in_ps + a.in_co + ccl);
  }
  f32x4 t0 = {1.f, 0.f, 1.f, 0.f}, t1 = {1.f, 0.f, 1.f, 0.f};
  if (a.ni_ab) {
    const float* tab = a.ni_ab + ((size_t)b * a.ni_C + ccl) * 2;
    t0 = *reinterpret_cast<const f32x4*>(tab);
    t1 = *reinterpret_cast<const f32x4*>(tab + 4);
  }
  // B fragments of all 16 K steps: lane (col = lane & 15, j = lane >> 4), column 16 nt + col = (tap t, output n), channel 4 k4 + j
  const int mn = lane & 15, j = lane >> 4;
  const int quads = a.cin_chunks * 8, ncol = 9 * a.Cout;
  float wr[16][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = 16 * nt + mn, t = min(col / a.Cout, 8), n = col - (col / a.Cout) * a.Cout;
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
      const bool ok = col < ncol && k4 < quads && 4 * k4 < a.Cin;
      const float v = a.w[(((size_t)t * quads + min(k4, quads - 1)) * a.cout_pad + n) * 4 + j];
      wr[k4][nt] = ok ? v : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = lane + 64 * i;
    f32x4 v = xv[i];
    if (a.ni_ab) {
      v[0] = fmaxf(fmaf(v[0], t0[0], t0[1]), 0.f);
      v[1] = fmaxf(fmaf(v[1], t0[2], t0[3]), 0.f);
      v[2] = fmaxf(fmaf(v[2], t1[0], t1[1]), 0.f);
      v[3] = fmaxf(fmaf(v[3], t1[2], t1[3]), 0.f);
    }
    if (!xin[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (idx < SN2_PX * 4) *reinterpret_cast<f32x4*>(xt + sn2_slot(idx >> 2, cc_l) * 4) = v;
  }
  f32x4 acc[3][NT];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // the tiles are complete
  // wave q: halo pixels 48 q .. 48 q + 47 (rows past the 180th are clamped: computed, never read)
#pragma unroll
  for (int k4 = 0; k4 < 16; ++k4) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int px = min(48 * q + 16 * mt + mn, SN2_PX - 1);
      const float xa = xt_all[(k4 >> 2) * (SN2_PX * 16) + sn2_slot(px, k4 & 3) * 4 + j];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, wr[k4][nt], acc[mt][nt], 0, 0, 0);
    }
  }
  __syncthreads();   // every wave is done with the tiles: G goes into the same LDS
  constexpr int GLD = 16 * NT + 1;
  float* G = xt_all;      // [SN2_PX][GLD]
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int px = 48 * q + 16 * mt + 4 * j + r;
        if (px < SN2_PX) G[px * GLD + 16 * nt + mn] = acc[mt][nt][r];
      }
  __syncthreads();
  // thread -> (pixel p = 16 row + column, output n): the nine taps in order
  for (int o = threadIdx.x; o < 128 * a.Cout; o += 256) {
    const int n = o >> 7, p = o & 127, row = p >> 4, colp = p & 15;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) v += G[((row + t / 3) * SNT_HW + colp + t % 3) * GLD + t * a.Cout + n];
    const float sc = a.scale ? a.scale[n] : 1.f, sh = a.shift ? a.shift[n] : 0.f;
    a.out[((size_t)(b * a.OH + oh0 + row) * a.OW + ow0 + colp) * a.out_ps + a.out_co + n] = pn::apply_act(fmaf(v, sc, sh), a.act);
  }
}

// every job an 8 x 16-tile job (MultiArgs.job[].snt2) or a 1 x 1 job: the matrix-pipe body for the tiles, only the tiles in LDS
__global__ __launch_bounds__(256, 3) void conv_small_n_mfma_multi_kernel(MultiArgs m_by_value) {
  __shared__ __attribute__((aligned(16))) float xt_lds[4 * SN2_PX * 16];
  static_assert(4 * SN2_PX * 16 >= 4 * 128 * 13 && 4 * SN2_PX * 16 >= SN2_PX * 33, "the partial sums / the G image reuse the tiles' LDS");
  typedef const __attribute__((address_space(4))) int* kptr_t;
  const kptr_t base = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int njobs = base[offsetof(MultiArgs, njobs) / 4];
  const int t = blockIdx.x;
  int j = 0;
  for (int k = 1; k < njobs; ++k)
    if (t >= base[offsetof(MultiArgs, first) / 4 + k]) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = t - base[offsetof(MultiArgs, first) / 4 + j];
  constexpr int JW = sizeof(ConvArgs) / 4;
  union { ConvArgs a; int w[JW]; } u;
  const kptr_t src = base + offsetof(MultiArgs, job) / 4 + j * JW;
#pragma unroll
  for (int i = 0; i < JW; ++i) u.w[i] = src[i];
  const ConvArgs& a = u.a;
  if (a.snt2 && a.Cout <= 3) {
    if (a.Cout == 1) small_n_gform_body<1>(a, local, xt_lds);
    else small_n_gform_body<2>(a, local, xt_lds);
  } else if (a.snt2) {
    switch ((a.Cin + 15) / 16) {
      case 1: small_n_mfma_body<1>(a, local, xt_lds); break;
      case 2: small_n_mfma_body<2>(a, local, xt_lds); break;
      case 3: small_n_mfma_body<3>(a, local, xt_lds); break;
      default: small_n_mfma_body<4>(a, local, xt_lds); break;
    }
  } else {      // a 1 x 1 job of the same launch (the range-stratified branch's last layer): the vector body, its partial sums and weights in the tiles' LDS
    float (*part)[64][13] = reinterpret_cast<float (*)[64][13]>(xt_lds);
    float* w_lds = xt_lds + 4 * 64 * 13;
    static_assert(4 * 64 * 13 + 16 * 12 * 4 <= 4 * SN2_PX * 16, "the 1 x 1 body fits the tiles' LDS");
    if (a.ncols <= 4) small_n_body<4, 1>(a, local, part, w_lds);
    else small_n_body<12, 1>(a, local, part, w_lds);
  }
}

__global__ __launch_bounds__(256) void conv_small_n_multi_kernel(MultiArgs m_by_value) {
  __shared__ __attribute__((aligned(16))) float w_lds[9 * 16 * 12 * 4];
  constexpr int kXt = 4 * SNT_PX * SNT_LD > 4 * SN2_PX * 16 ? 4 * SNT_PX * SNT_LD : 4 * SN2_PX * 16;
  __shared__ __attribute__((aligned(16))) float xt_lds[kXt];
  float (*part)[64][13] = reinterpret_cast<float (*)[64][13]>(xt_lds);    // the partial sums reuse the tile's LDS (tiles + weights < 80 KB: two blocks per CU)
  typedef const __attribute__((address_space(4))) int* kptr_t;
  const kptr_t base = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int njobs = base[offsetof(MultiArgs, njobs) / 4];
  const int t = blockIdx.x;
  int j = 0;
  for (int k = 1; k < njobs; ++k)
    if (t >= base[offsetof(MultiArgs, first) / 4 + k]) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = t - base[offsetof(MultiArgs, first) / 4 + j];
  constexpr int JW = sizeof(ConvArgs) / 4;
  union { ConvArgs a; int w[JW]; } u;
  const kptr_t src = base + offsetof(MultiArgs, job) / 4 + j * JW;
#pragma unroll
  for (int i = 0; i < JW; ++i) u.w[i] = src[i];
  const ConvArgs& a = u.a;
  const int taps = a.KH * a.KW;
  if (taps == 1) {
    if (a.ncols <= 4) small_n_body<4, 1>(a, local, part, w_lds);
    else small_n_body<12, 1>(a, local, part, w_lds);
  } else if (a.snt2) {
    if (a.ncols <= 4) small_n_tiled2_body<4>(a, local, w_lds, xt_lds);
    else if (a.ncols <= 8) small_n_tiled2_body<8>(a, local, w_lds, xt_lds);
    else small_n_tiled2_body<12>(a, local, w_lds, xt_lds);
  } else if (small_n_tiled_ok(a)) {
    if (a.ncols <= 4) small_n_tiled_body<4>(a, local, part, w_lds, xt_lds);
    else if (a.ncols <= 8) small_n_tiled_body<8>(a, local, part, w_lds, xt_lds);
    else small_n_tiled_body<12>(a, local, part, w_lds, xt_lds);
  } else {
    if (a.ncols <= 4) small_n_body<4, 9>(a, local, part, w_lds);
    else if (a.ncols <= 8) small_n_body<8, 9>(a, local, part, w_lds);
    else small_n_body<12, 9>(a, local, part, w_lds);
  }
}

__global__ __launch_bounds__(256) void conv_tap_sum_kernel(const float* __restrict__ g, int ldg, int B, int H, int W, int cout, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, float* __restrict__ out, int out_ps, int out_co) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * H * W * cout) return;
  const int co = i % cout, pix = i / cout;
  const int x = pix % W, y = (pix / W) % H;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) s += g[(size_t)(pix + (t / 3 - 1) * W + (t % 3 - 1)) * ldg + t * cout + co];
  }
  const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
  out[(size_t)pix * out_ps + out_co + co] = pn::apply_act(fmaf(s, sc, sh), act);
}

template <int TAPS>
void launch_small_n(const ConvArgs& a, int zdim, hipStream_t st) {
  const dim3 grid(pn::cdiv(a.M, 64), zdim);
  pn::ProfileSlot ps;
  const bool prof = pn::take_profile_slot(ps);
  if (a.ncols <= 4) {
    if (prof) hipExtLaunchKernelGGL((conv_small_n_kernel<4, TAPS>), grid, dim3(256), 0, st, ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL((conv_small_n_kernel<4, TAPS>), grid, dim3(256), 0, st, a);
  } else {
    if (prof) hipExtLaunchKernelGGL((conv_small_n_kernel<8, TAPS>), grid, dim3(256), 0, st, ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL((conv_small_n_kernel<8, TAPS>), grid, dim3(256), 0, st, a);
  }
}

template <int WM, int WN, int TM, int TN, int DT = DT_F32, bool GATHER = false>
int launch_conv(const ConvArgs& a, int zdim, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  // gather mode keeps the block's [BM][taps <= 32] neighbour table behind the two staging buffers
  constexpr size_t smem = 2 * (size_t)(BM * A_LD + BK * BN) * sizeof(float) + (GATHER ? (size_t)(BM * 32 + 72) * sizeof(int) : 0);
  static bool attr_done[64] = {false};   // per device: the attribute belongs to the device's code object
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<WM, WN, TM, TN, DT, GATHER>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  ConvArgs b = a;
  b.nmt = pn::cdiv(a.M, BM);
  dim3 grid(b.nmt, pn::cdiv(a.ncols, BN), zdim);
  pn::ProfileSlot ps;
  if (pn::take_profile_slot(ps))
    hipExtLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN, DT, GATHER>), grid, dim3(WM * WN * 64), smem, st, ps.start, ps.stop, 0, b);
  else
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN, DT, GATHER>), grid, dim3(WM * WN * 64), smem, st, b);
  return pn::check_launch("conv_mfma_kernel");
}

// es = bytes per input element (4: f32, 2: bf16); a K step is 8 chunks of 16 bytes = 128 / es channels
int fill_args(const pn_conv_desc* d, ConvArgs& a, int& zdim, int es = 4) {
  PN_REQUIRE(d != nullptr, "conv: null descriptor");
  a = ConvArgs{};
  a.ni_S = 1;
  a.st_S = 1;
  PN_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->cin > 0 && d->cout > 0, "conv: bad sizes");
  PN_REQUIRE(d->groups >= 1 && d->kh >= 1 && d->kw >= 1 && d->stride >= 1, "conv: bad kernel params");
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = d->cin; a.Cout = d->cout;
  a.KH = d->kh; a.KW = d->kw; a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  PN_REQUIRE(d->pad_h_end >= 0 && d->pad_w_end >= 0, "conv: negative end padding");
  a.OH = (d->in_h + 2 * d->pad_h + d->pad_h_end - d->kh) / d->stride + 1;
  a.OW = (d->in_w + 2 * d->pad_w + d->pad_w_end - d->kw) / d->stride + 1;
  PN_REQUIRE(a.OH > 0 && a.OW > 0, "conv: empty output");
  a.in_ps = d->in_pixel_stride; a.in_co = d->in_channel_offset;
  a.out_ps = d->out_pixel_stride; a.out_co = d->out_channel_offset;
  a.act = d->act;
  a.mode = MODE_CONV; a.OWsub = a.OW; a.ncols = d->cout; a.in_group_stride = d->groups > 1 ? d->cin : 0;
  zdim = d->groups;
  if (d->deconv2x2) {
    // kh = kw = 1: ConvTranspose2d(k=2, s=2); kh = kw = 2 with one row/column of end padding: the
    // four-phase data gradient of a 3x3 stride-2 convolution (pn_pack_conv_dgrad_s2_weight_f32)
    PN_REQUIRE(d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 && d->groups == 1 &&
                   ((d->kh == 1 && d->kw == 1) || (d->kh == 2 && d->kw == 2 && d->pad_h_end == 1 && d->pad_w_end == 1)),
               "conv: deconv2x2 wants stride=1, pad=0, groups=1 and kh=kw=1 (or kh=kw=2 with end padding 1)");
    a.mode = MODE_DECONV2; a.ncols = 4 * d->cout;
  }
  if (d->range_strata > 1) {
    PN_REQUIRE(d->groups == 1 && !d->deconv2x2, "conv: range_strata needs groups=1 and no deconv");
    PN_REQUIRE(a.OW % d->range_strata == 0 && d->stride == 1, "conv: range axis not divisible into strata");
    a.mode = MODE_STRAT; a.OWsub = a.OW / d->range_strata; zdim = d->range_strata; a.in_group_stride = 0;
  }
  a.M = d->batch * a.OH * a.OWsub;
  a.cin_chunks = pn::cdiv(d->cin, 128 / es);
  a.cout_pad = pn::cdiv(a.ncols, 32) * 32;
  const unsigned long long in_bytes = (unsigned long long)d->batch * d->in_h * d->in_w * d->in_pixel_stride * (unsigned long long)es;
  PN_REQUIRE(in_bytes < (1ull << 31), "conv: input map larger than 2 GiB is not addressable by the buffer descriptor");
  PN_REQUIRE(d->kh * d->kw <= 32, "conv: at most 32 taps");
  a.in_bytes = (unsigned)in_bytes;
  a.w_bytes = (unsigned)((size_t)d->kh * d->kw * a.cin_chunks * 8 * a.cout_pad * 16);
  a.force_tile = 0;
  a.res = nullptr;
  a.res_ps = 0;
  a.nbr = nullptr;
  a.n_valid = nullptr;
  a.res_pre_act = 0;
  return PN_OK;
}

int dispatch_conv(const ConvArgs& a, int zdim, hipStream_t st) {
  // tile choice: fill the 256 CUs (8 waves per CU where possible) with the largest wave tile
  static const int forced = [] { const char* e = getenv("PN_CONV_TILE"); return e ? atoi(e) : 0; }();
  int tile = forced;
  if (tile == 0) {
    const long long t128 = (long long)pn::cdiv(a.M, 128) * pn::cdiv(a.ncols, 128) * zdim;
    const long long t64x128 = (long long)pn::cdiv(a.M, 64) * pn::cdiv(a.ncols, 128) * zdim;
    static const int min128 = [] { const char* e = getenv("PN_CONV_MIN128"); return e ? atoi(e) : 384; }();
    static const int min64x128 = [] { const char* e = getenv("PN_CONV_MIN64X128"); return e ? atoi(e) : 256; }();
    // mid-size maps (256 .. 1023 tiles of 64 x 128): eight waves of 32 x 32 per block measured 8 % faster than four of 32 x 64
    // on the 128 x 128 layers (103 vs 95 TFLOP/s), also on the stride-2 and 2x2 layers
    if (a.ncols > 64) tile = t128 >= min128 ? 1 : (t64x128 >= min64x128 ? 5 : 3);
    else if (a.ncols > 32) tile = 3;
    // <= 32 columns: with few blocks (<= 2 per CU) the 4-wave 64 x 64 block hides its load latency better than the 2-wave
    // 64 x 32 one although half of its columns are padding (17.7 vs 19.1 us for 64 -> 10, 15.9 vs 17.6 for the grouped 32 -> 32)
    else tile = (long long)pn::cdiv(a.M, 64) * zdim <= 512 ? 3 : 4;
  }
  switch (tile) {
    case 1: return launch_conv<2, 2, 2, 2>(a, zdim, st);  // 128 x 128, 4 waves of 64x64
    case 2: return launch_conv<2, 2, 1, 2>(a, zdim, st);  //  64 x 128, 4 waves of 32x64
    case 3: return launch_conv<2, 2, 1, 1>(a, zdim, st);  //  64 x  64, 4 waves of 32x32
    case 4: return launch_conv<2, 1, 1, 1>(a, zdim, st);  //  64 x  32, 2 waves of 32x32
    case 5: return launch_conv<2, 4, 1, 1>(a, zdim, st);  //  64 x 128, 8 waves of 32x32
    case 6: return launch_conv<2, 2, 2, 1>(a, zdim, st);  // 128 x  64, 4 waves of 64x32
    case 7: return launch_conv<4, 2, 1, 1>(a, zdim, st);  // 128 x  64, 8 waves of 32x32
    default: return pn::fail(PN_ERR_INVALID, "conv: unknown tile id %d", tile);
  }
}

template <int DT>
int dispatch_conv_bf16(const ConvArgs& a, int zdim, hipStream_t st) {
  static const int forced = [] { const char* e = getenv("PN_CONV_TILE_BF16"); return e ? atoi(e) : 0; }();
  int tile = forced;
  if (tile == 0) {
    const long long t128 = (long long)pn::cdiv(a.M, 128) * pn::cdiv(a.ncols, 128) * zdim;
    const long long t64x128 = (long long)pn::cdiv(a.M, 64) * pn::cdiv(a.ncols, 128) * zdim;
    if (a.ncols > 64) tile = t128 >= 384 ? 1 : (t64x128 >= 256 ? 2 : 3);
    else if (a.ncols > 32) tile = 3;
    else tile = 4;
  }
  switch (tile) {
    case 1: return launch_conv<2, 2, 2, 2, DT>(a, zdim, st);
    case 2: return launch_conv<2, 2, 1, 2, DT>(a, zdim, st);
    case 3: return launch_conv<2, 2, 1, 1, DT>(a, zdim, st);
    case 4: return launch_conv<2, 1, 1, 1, DT>(a, zdim, st);
    default: return pn::fail(PN_ERR_INVALID, "conv_bf16: unknown tile id %d", tile);
  }
}

// torch (Cout_total, Cin_g, KH, KW) f32 -> bf16 [g][tap][cin_pad/8][cout_pad][8]
__global__ void pack_conv_weight_bf16_kernel(const float* __restrict__ w, int cout_g, int cin_g, int kh, int kw, int groups, int cin_pad,
                                             int cout_pad, unsigned short* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 7;
    r >>= 3;
    const int n = r % cout_pad;
    r /= cout_pad;
    const int k8 = r % (cin_pad / 8);
    r /= (cin_pad / 8);
    const int tap = r % (kh * kw);
    const int g = (int)(r / (kh * kw));
    const int c = k8 * 8 + k1;
    float v = 0.f;
    if (n < cout_g && c < cin_g) v = w[(((size_t)(g * cout_g + n) * cin_g + c) * kh + tap / kw) * kw + tap % kw];
    packed[i] = f32_to_bf16_rne(v);
  }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = f32_to_bf16_rne(x[i]);
}

__global__ void bf16_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = __builtin_bit_cast(float, (unsigned)x[i] << 16);
}

template <int WM, int WN, int TM, int TN, bool NORM_IN>
int launch_multi(MultiArgs& m, size_t extra_smem, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t base = 2 * (size_t)(BM * A_LD + BK * BN) * sizeof(float);
  const size_t smem = base + extra_smem;
  PN_REQUIRE(smem <= 160 * 1024, "conv_multi: the normalisation table does not fit LDS");
  static size_t attr_smem[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || attr_smem[dev] < smem) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_multi_kernel<WM, WN, TM, TN, NORM_IN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (dev >= 0 && dev < 64) attr_smem[dev] = smem;
  }
  int total = 0;
  for (int j = 0; j < m.njobs; ++j) {
    ConvArgs& a = m.job[j];
    a.nmt = pn::cdiv(a.M, BM);
    const int tiles_z = a.nmt * pn::cdiv(a.ncols, BN);
    m.first[j] = total;
    total += tiles_z * a.zdim;
  }
  m.first[m.njobs] = total;
  m.total = total;
  pn::ProfileSlot ps;
  if (pn::take_profile_slot(ps))
    hipExtLaunchKernelGGL((conv_multi_kernel<WM, WN, TM, TN, NORM_IN>), dim3(total), dim3(WM * WN * 64), smem, st, ps.start, ps.stop, 0, m);
  else
    hipLaunchKernelGGL((conv_multi_kernel<WM, WN, TM, TN, NORM_IN>), dim3(total), dim3(WM * WN * 64), smem, st, m);
  return pn::check_launch("conv_multi_kernel");
}

int tile_bm(int tile) { return tile == 1 ? 128 : 64; }

}  // namespace

extern "C" {

/* ---- sparse convolution as a gathered GEMM: out[m] = act((sum_t W_t . in[nbr[m][t]]) * scale + shift) (+ residual[m]).
 * Same kernel, MODE "gather": the A tile loader takes the input row of every (output site, tap) from the block's neighbour table
 * (staged in LDS) instead of pixel arithmetic.  packed_w: pn_pack_conv_weight_f32 of the weight seen as (Cout, Cin, taps, 1). ---- */
}  // extern "C"

namespace {

// ---- the 16-channel level of the sparse encoder (conv_input and the two residual blocks of conv1, scn.py:112-123: 8 / 16 -> 16 channels
// on the finest grid) on the VALU.  On the gathered MFMA kernel a tap is one K step whose second half is zero padding, three quarters of
// the (site, tap) pairs do not exist (6.1 of 27 neighbours on a 64-beam sweep) and every step waits for a dependent gather: 160 us per
// layer at 100 k sites, which is 0.3 GFLOP and 30 MB.  Here four lanes share an output site: lane (site, q) fetches channel quad q of
// ALL 27 neighbour rows up front (one buffer load each, absent neighbours redirected out of range -> zeros, nothing waits on anything),
// then walks the taps that any of the wave's 16 sites has: the other three quads of the row come from the neighbouring lanes as DPP
// quad broadcasts folded into the FMAs, the tap's 16 x 4 weights of output quad q are read from LDS (staged once per block).
// Summation order: taps ascending, input channels ascending (the MFMA kernel adds the two k halves of a step pairwise: last-bit
// differences between the two routes).
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}

template <int CIN>
__global__ __launch_bounds__(256) void sparse_conv_c16_kernel(const float* __restrict__ in, unsigned in_bytes, const int32_t* __restrict__ nbr,
                                                              const int32_t* __restrict__ n_out, int cap, int taps, const float* __restrict__ packed_w,
                                                              int quads, int cout_pad, const float* __restrict__ scale, const float* __restrict__ shift,
                                                              int act, const float* __restrict__ residual, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float wl[27 * 256];      // [tap][ci 16][co 16]
  __shared__ int nb_s[64 * 27];                                     // this group's neighbour rows, [site][tap]
  const int tid = threadIdx.x, q = tid & 3, sl = tid >> 2;
  for (int i = tid; i < taps * 256; i += 256) {
    const int t = i >> 8, ci = (i >> 4) & 15, co = i & 15;
    wl[i] = ci < CIN ? packed_w[(((size_t)t * quads + (ci >> 2)) * cout_pad + co) * 4 + (ci & 3)] : 0.f;
  }
  const int n = min(*n_out, cap);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, in_bytes, 0x00020000);
  const f32x4 sc = scale ? *reinterpret_cast<const f32x4*>(scale + 4 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 sh = shift ? *reinterpret_cast<const f32x4*>(shift + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  // persistent blocks, groups dealt round-robin (contiguous runs per block measured 84 against 64 us: the live taps per group follow the
  // scene, so runs of neighbouring groups are unevenly expensive)
  for (int g = blockIdx.x; g * 64 < n; g += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < 64 * taps; i += 256) nb_s[i] = g * 64 + i / taps < n ? nbr[(size_t)g * 64 * taps + i] : -1;
    __syncthreads();
    const int site = g * 64 + sl;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (residual && site < n) r = *reinterpret_cast<const f32x4*>(residual + (size_t)site * 16 + 4 * q);
    f32x4 x[27];
    unsigned live = 0;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int idx = t < taps ? nb_s[sl * taps + t] : -1;
      live |= (__builtin_amdgcn_ballot_w64(idx >= 0) != 0ull ? 1u : 0u) << t;
      const unsigned vo = (idx >= 0 && 4 * q < CIN) ? (unsigned)idx * (unsigned)(CIN * 4) + 16u * q : 0xffffffffu;
      x[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0));
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if (live >> t & 1) {                                                  // wave-uniform
        const float* wt = wl + t * 256 + 4 * q;
#pragma unroll
        for (int c4 = 0; c4 < CIN / 4; ++c4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xv = c4 == 0 ? quad_bcast<0>(x[t][k]) : c4 == 1 ? quad_bcast<1>(x[t][k]) : c4 == 2 ? quad_bcast<2>(x[t][k]) : quad_bcast<3>(x[t][k]);
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wt + (4 * c4 + k) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, w4[j], acc[j]);
          }
        }
      }
    }
    if (site < n) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = pn::apply_act(fmaf(acc[j], sc[j], sh[j]) + r[j], act);
      *reinterpret_cast<f32x4*>(out + (size_t)site * 16 + 4 * q) = v;
    }
  }
}

}  // namespace

extern "C" {

int pn_sparse_conv_f32(const float* in, int in_rows, int cin, const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps,
                       const float* packed_w, int cout, const float* scale, const float* shift, int act, const float* residual, float* out,
                       pn_stream_t stream) {
  PN_REQUIRE(in && nbr && n_out && packed_w && out, "sparse_conv: null pointer");
  PN_REQUIRE(in_rows >= 1 && cin >= 4 && cin % 4 == 0 && cout >= 1 && out_capacity >= 1 && taps >= 1 && taps <= 32, "sparse_conv: bad sizes");
  PN_REQUIRE((unsigned long long)in_rows * cin * 4ull < (1ull << 31), "sparse_conv: input feature matrix larger than 2 GiB");
  ConvArgs a{};
  a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift; a.out = out;
  a.B = 1; a.H = in_rows; a.W = 1; a.Cin = cin; a.Cout = cout; a.OH = out_capacity; a.OW = 1;
  a.KH = taps; a.KW = 1; a.stride = 1; a.pad_h = 0; a.pad_w = 0;
  a.in_ps = cin; a.in_co = 0; a.out_ps = cout; a.out_co = 0; a.act = act; a.mode = MODE_CONV;
  a.M = out_capacity; a.OWsub = 1; a.cin_chunks = pn::cdiv(cin, BK); a.cout_pad = pn::cdiv(cout, 32) * 32; a.ncols = cout;
  a.in_group_stride = 0; a.in_bytes = (unsigned)((size_t)in_rows * cin * 4); a.w_bytes = (unsigned)((size_t)taps * a.cin_chunks * 8 * a.cout_pad * 16);
  a.res = residual; a.res_ps = cout; a.nbr = nbr; a.n_valid = n_out;
  a.res_pre_act = 1;  // SparseBasicBlock: relu(bn2(conv2(.)) + identity), scn.py:84-95
  hipStream_t st = pn::S(stream);
  static const int tile_exp = [] { const char* e = getenv("PN_SPARSE_TILE"); return e ? atoi(e) : 0; }();
  if (cout > 64) {
    if (tile_exp & 1) return launch_conv<2, 2, 1, 2, DT_F32, true>(a, 1, st);
    if (tile_exp & 4) return launch_conv<2, 2, 1, 1, DT_F32, true>(a, 1, st);
    return out_capacity >= 128 * 128 ? launch_conv<2, 2, 2, 2, DT_F32, true>(a, 1, st) : launch_conv<2, 2, 1, 2, DT_F32, true>(a, 1, st);
  }
  if (cout > 32) return (tile_exp & 2) ? launch_conv<2, 2, 2, 1, DT_F32, true>(a, 1, st) : launch_conv<2, 2, 1, 1, DT_F32, true>(a, 1, st);
  // 32 columns: four waves on 128 rows share the weight tile (64 x 32 tiles, two waves: 237 us for the 32 -> 32 layers of the bench frame,
  // 300 k sites; this form 207; two row tiles per wave 350)
  if (tile_exp & 16) return launch_conv<2, 1, 1, 1, DT_F32, true>(a, 1, st);
  return launch_conv<4, 1, 1, 1, DT_F32, true>(a, 1, st);
}

// The 16-channel level (cout == 16, cin 8 or 16) on the VALU kernel above.  Same arguments and result as pn_sparse_conv_f32 up to the
// summation order (taps ascending, channels ascending) -- a separate entry so that a caller names the form it runs.
int pn_sparse_conv_c16_f32(const float* in, int in_rows, int cin, const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps,
                           const float* packed_w, const float* scale, const float* shift, int act, const float* residual, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && nbr && n_out && packed_w && out, "sparse_conv_c16: null pointer");
  PN_REQUIRE((cin == 8 || cin == 16) && taps >= 1 && taps <= 27 && in_rows >= 1 && out_capacity >= 1, "sparse_conv_c16: cin must be 8 or 16, taps <= 27");
  PN_REQUIRE((size_t)in_rows * cin * 4 < 0xffffffffull, "sparse_conv_c16: input table too large for the buffer descriptor");
  PN_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)scale & 15) == 0 &&
                 ((uintptr_t)shift & 15) == 0, "sparse_conv_c16: pointers must be 16-byte aligned");
  PN_REQUIRE(act == PN_ACT_NONE || act == PN_ACT_RELU, "sparse_conv_c16: activation none or ReLU");
  hipStream_t st = pn::S(stream);
  pn::ProfileSlot ps;
  const bool prof = pn::take_profile_slot(ps);
  // persistent: two blocks per CU (233 registers), the 27 KB of weights staged once per block (one block per group: 75 us per layer at
  // 207 k sites, this form 64); fewer than 512 groups of capacity: one block each
  const int cap_groups = pn::cdiv(out_capacity, 64);
  static const int pgrid = [] { const char* e = getenv("PN_SPARSE_C16_GRID"); const int v = e ? atoi(e) : 512; return v >= 64 ? v : 512; }();
  const dim3 grid((unsigned)(cap_groups >= pgrid ? pgrid : cap_groups));
  const unsigned in_bytes = (unsigned)((size_t)in_rows * cin * 4);
  const int quads = pn::cdiv(cin, 32) * 8, cout_pad = 32;          // the layout pn_pack_conv_weight_f32 gives (16, cin, taps, 1)
  auto kern = cin == 8 ? &sparse_conv_c16_kernel<8> : &sparse_conv_c16_kernel<16>;
  if (prof) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, st, ps.start, ps.stop, 0, in, in_bytes, nbr, n_out, out_capacity, taps, packed_w, quads, cout_pad,
                                  scale, shift, act, residual, out);
  else hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, in, in_bytes, nbr, n_out, out_capacity, taps, packed_w, quads, cout_pad, scale, shift, act, residual, out);
  return pn::check_launch("sparse_conv_c16_kernel");
}

/* ---- bf16 variant (BASELINE configs[3]: "bf16 BEV convs on MFMA"): bf16 activations and weights, f32 accumulate on
 * v_mfma_f32_32x32x16_bf16, scale / shift / activation in f32, output bf16 (or f32 for the last layer) ---- */
size_t pn_conv_packed_weight_bf16_elems(int cout, int cin, int kh, int kw, int groups) {
  return (size_t)groups * kh * kw * (size_t)(pn::cdiv(cin, 64) * 64) * (size_t)(pn::cdiv(cout, 32) * 32);
}

int pn_pack_conv_weight_bf16(const float* w_oihw, int cout_total, int cin_per_group, int kh, int kw, int groups, void* packed,
                             pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && groups >= 1 && cout_total % groups == 0, "pack_conv_weight_bf16: bad arguments");
  const int cout_g = cout_total / groups;
  const int cin_pad = pn::cdiv(cin_per_group, 64) * 64, cout_pad = pn::cdiv(cout_g, 32) * 32;
  const size_t total = pn_conv_packed_weight_bf16_elems(cout_g, cin_per_group, kh, kw, groups);
  hipLaunchKernelGGL(pack_conv_weight_bf16_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream),
                     w_oihw, cout_g, cin_per_group, kh, kw, groups, cin_pad, cout_pad, static_cast<unsigned short*>(packed), total);
  return pn::check_launch("pack_conv_weight_bf16_kernel");
}

int pn_conv2d_nhwc_bf16(const pn_conv_desc* d, const void* in_bf16, const void* packed_w_bf16, const float* scale, const float* shift,
                        void* out, int out_is_f32, pn_stream_t stream) {
  ConvArgs a;
  int zdim = 1;
  if (int rc = fill_args(d, a, zdim, 2)) return rc;
  PN_REQUIRE(in_bf16 && packed_w_bf16 && out, "conv_bf16: null pointer");
  PN_REQUIRE(!d->deconv2x2 || (d->kh == 1 && d->kw == 1), "conv_bf16: only the ConvTranspose2d(k=2,s=2) form of deconv2x2");
  PN_REQUIRE(d->cin % 8 == 0 && d->in_pixel_stride % 8 == 0 && d->in_channel_offset % 8 == 0,
             "conv_bf16: cin, input pixel stride and channel offset must be multiples of 8");
  PN_REQUIRE(!d->accumulate, "conv_bf16: accumulate is not supported");
  PN_REQUIRE(((uintptr_t)in_bf16 & 15) == 0 && ((uintptr_t)packed_w_bf16 & 15) == 0, "conv_bf16: pointers must be 16-byte aligned");
  a.in = static_cast<const float*>(in_bf16); a.w = static_cast<const float*>(packed_w_bf16); a.scale = scale; a.shift = shift;
  a.out = static_cast<float*>(out);
  return out_is_f32 ? dispatch_conv_bf16<DT_BF16_F32OUT>(a, zdim, pn::S(stream)) : dispatch_conv_bf16<DT_BF16>(a, zdim, pn::S(stream));
}

int pn_f32_to_bf16(const float* x, void* y, size_t n, pn_stream_t stream) {
  PN_REQUIRE(x && y, "f32_to_bf16: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)std::min<size_t>(8192, (n + 255) / 256)), dim3(256), 0, pn::S(stream), x,
                     static_cast<unsigned short*>(y), n);
  return pn::check_launch("f32_to_bf16_kernel");
}

int pn_bf16_to_f32(const void* x, float* y, size_t n, pn_stream_t stream) {
  PN_REQUIRE(x && y, "bf16_to_f32: null pointer");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)std::min<size_t>(8192, (n + 255) / 256)), dim3(256), 0, pn::S(stream),
                     static_cast<const unsigned short*>(x), y, n);
  return pn::check_launch("bf16_to_f32_kernel");
}

size_t pn_conv_packed_weight_floats(int cout, int cin, int kh, int kw, int groups) {
  return (size_t)groups * kh * kw * (size_t)(pn::cdiv(cin, BK) * BK) * (size_t)(pn::cdiv(cout, 32) * 32);
}

int pn_pack_conv_weight_f32(const float* w_oihw, int cout_total, int cin_per_group, int kh, int kw, int groups,
                            float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && groups >= 1 && cout_total % groups == 0, "pack_conv_weight: bad arguments");
  const int cout_g = cout_total / groups;
  const int cin_pad = pn::cdiv(cin_per_group, BK) * BK, cout_pad = pn::cdiv(cout_g, 32) * 32;
  const size_t total = pn_conv_packed_weight_floats(cout_g, cin_per_group, kh, kw, groups);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                     pn::S(stream), w_oihw, cout_g, cin_per_group, kh, kw, groups, cin_pad, cout_pad, packed, total);
  return pn::check_launch("pack_conv_weight_kernel");
}

size_t pn_deconv2x2_packed_weight_floats(int cin, int cout) {
  return (size_t)(pn::cdiv(cin, BK) * BK) * (size_t)(pn::cdiv(4 * cout, 32) * 32);
}

int pn_pack_deconv2x2_weight_f32(const float* w_iohw, int cin, int cout, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_iohw && packed && cin > 0 && cout > 0, "pack_deconv_weight: bad arguments");
  const int cin_pad = pn::cdiv(cin, BK) * BK, cout_pad = pn::cdiv(4 * cout, 32) * 32;
  const size_t total = (size_t)cin_pad * cout_pad;
  hipLaunchKernelGGL(pack_deconv_weight_kernel, dim3(std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                     pn::S(stream), w_iohw, cin, cout, cin_pad, cout_pad, packed, total);
  return pn::check_launch("pack_deconv_weight_kernel");
}

int pn_fold_bn_f32(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                   float eps, int c, float* scale, float* shift, pn_stream_t stream) {
  PN_REQUIRE(gamma && beta && mean && var && scale && shift && c > 0, "fold_bn: bad arguments");
  hipLaunchKernelGGL(fold_bn_kernel, dim3(pn::cdiv(c, 256)), dim3(256), 0, pn::S(stream), gamma, beta, mean, var,
                     conv_bias, eps, c, scale, shift);
  return pn::check_launch("fold_bn_kernel");
}

int pn_conv2d_nhwc_f32(const pn_conv_desc* d, const float* in, const float* packed_w, const float* scale,
                       const float* shift, float* out, pn_stream_t stream) {
  ConvArgs a;
  int zdim = 1;
  if (int rc = fill_args(d, a, zdim)) return rc;
  PN_REQUIRE(in && packed_w && out, "conv: null pointer");
  PN_REQUIRE(d->cin % 4 == 0 && d->in_pixel_stride % 4 == 0 && d->in_channel_offset % 4 == 0,
             "conv: MFMA path needs cin, input pixel stride and channel offset to be multiples of 4 "
             "(use pn_conv2d_direct_nhwc_f32)");
  PN_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "conv: pointers must be 16-byte aligned");
  a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift; a.out = out;
  if (d->accumulate) {  // out += result: the residual input of the epilogue is the output itself
    a.res = out; a.res_ps = d->out_pixel_stride;
  }
  hipStream_t st = pn::S(stream);
  static const int small_n = [] { const char* e = getenv("PN_CONV_SMALL_N"); return e ? atoi(e) : 1; }();
  const int taps = a.KH * a.KW;
  if (small_n && a.mode == MODE_CONV && a.ncols <= kSmallN && a.Cin <= 64 && (taps == 9 || taps == 1) && a.res == nullptr &&
      a.M >= 4096) {
    // last layer of a head branch: too few columns for an MFMA tile to hide its load latency
    if (taps == 9) launch_small_n<9>(a, zdim, st);
    else launch_small_n<1>(a, zdim, st);
    return pn::check_launch("conv_small_n_kernel");
  }
  return dispatch_conv(a, zdim, st);
}

// r6: the same convolution with its output written as the F(4, 3) planes the chained layers read (pn_wino4_planes_floats(batch, oh, ow,
// cout) floats, not transposed) -- the stride-2 layer at the head of an RPN block feeds the block's chain without the NHWC map and the
// NHWC -> planes pass in between.  The epilogue transposes the accumulators through LDS anyway; with block tiles of whole map rows a quad's
// two neighbour pixels are in the same block.  Same values as pn_conv2d_nhwc_f32 + pn_wino4_planes_from_nhwc_f32, bit for bit.
int pn_conv2d_nhwc_planes_supported(const pn_conv_desc* d) {
  ConvArgs a;
  int zdim = 1;
  if (!d || fill_args(d, a, zdim)) return 0;
  if (a.mode != MODE_CONV || zdim != 1 || d->accumulate || (d->cout & 7) || (d->cin & 3) || (d->in_pixel_stride & 3) || (d->in_channel_offset & 3)) return 0;
  if ((a.OW & 3) || !((a.OW <= 64 && 64 % a.OW == 0) || a.OW == 128)) return 0;
  if (a.M % (a.OW == 128 ? 128 : 64)) return 0;
  return 1;
}

int pn_conv2d_nhwc_planes_f32(const pn_conv_desc* d, const float* in, const float* packed_w, const float* scale, const float* shift, float* planes,
                              pn_stream_t stream) {
  PN_REQUIRE(pn_conv2d_nhwc_planes_supported(d), "conv_planes: not covered (groups 1, cout % 8 == 0, output rows of 4 .. 64 pixels dividing 64, or of 128)");
  PN_REQUIRE(in && packed_w && planes && ((uintptr_t)in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0 && ((uintptr_t)planes & 15) == 0 &&
                 ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0, "conv_planes: null or misaligned pointer");
  ConvArgs a;
  int zdim = 1;
  if (int rc = fill_args(d, a, zdim)) return rc;
  a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift;
  a.out = planes; a.out_ps = d->cout; a.out_co = 0;      // (the vectorised epilogue's preconditions; nothing is stored NHWC)
  a.planes = planes;
  a.plane_floats = (unsigned)((size_t)d->batch * (a.OH + 2) * (a.OW / 4) * 4);
  hipStream_t st = pn::S(stream);
  if (a.OW == 128) {      // block tiles of 128 pixels: 128 x 128 where those fill the chip, else 128 x 64 on eight waves
    const long long t128 = (long long)pn::cdiv(a.M, 128) * pn::cdiv(a.ncols, 128);
    return t128 >= 384 ? launch_conv<2, 2, 2, 2>(a, zdim, st) : launch_conv<4, 2, 1, 1>(a, zdim, st);
  }
  return dispatch_conv(a, zdim, st);
}

size_t pn_conv_stat_partial_floats(const pn_conv_desc* d, int tile) {
  ConvArgs a;
  int zdim = 1;
  if (fill_args(d, a, zdim)) return 0;
  const int bm = tile_bm(tile);
  return (size_t)zdim * pn::cdiv(a.M, bm) * (bm / 32) * a.cout_pad * 2;
}

static int tile_wn(int tile) { return tile == 5 ? 4 : (tile == 4 ? 1 : 2); }

// pn_conv_job -> ConvArgs (validation included); `extra` = LDS bytes of the normalise-on-load table
static int job_to_args(const pn_conv_job& jb, int tile, ConvArgs& a, size_t& extra) {
  int zdim = 1;
  if (int rc = fill_args(&jb.desc, a, zdim)) return rc;
  const pn_conv_desc* d = &jb.desc;
  PN_REQUIRE(jb.in && jb.packed_w && jb.out, "conv_multi: null pointer");
  PN_REQUIRE(d->cin % 4 == 0 && d->in_pixel_stride % 4 == 0 && d->in_channel_offset % 4 == 0,
             "conv_multi: cin, input pixel stride and channel offset must be multiples of 4");
  PN_REQUIRE(((uintptr_t)jb.in & 15) == 0 && ((uintptr_t)jb.packed_w & 15) == 0, "conv_multi: pointers must be 16-byte aligned");
  PN_REQUIRE(!d->accumulate, "conv_multi: accumulate is not supported");
  a.in = jb.in; a.w = jb.packed_w; a.scale = jb.scale; a.shift = jb.shift; a.out = jb.out;
  a.zdim = zdim;
  const int bm = tile_bm(tile), bn = tile_wn(tile) * (tile == 1 ? 64 : 32);
  a.nmt = pn::cdiv(a.M, bm);
  a.st_segs_z = a.nmt * (bm / 32);
  if (jb.stat_partials) {
    PN_REQUIRE(a.mode != MODE_DECONV2, "conv_multi: no statistics for the transposed convolution");
    PN_REQUIRE((a.Cout % 4 == 0) && (a.out_ps % 4 == 0) && (a.out_co % 4 == 0) && (((uintptr_t)a.out & 15) == 0),
               "conv_multi: statistics need the vectorised epilogue (channel counts / offsets multiples of 4)");
    PN_REQUIRE(a.mode == MODE_STRAT || zdim == 1, "conv_multi: statistics of grouped convolutions are not supported");
    PN_REQUIRE(a.ncols <= bn, "conv_multi: statistics need the job's columns in one column tile");
    const int S = jb.stat_strata < 1 ? 1 : jb.stat_strata;
    PN_REQUIRE(S == 1 || (a.mode == MODE_CONV && a.OWsub % S == 0 && (a.OWsub / S) % 32 == 0),
               "conv_multi: stat_strata needs a plain convolution whose strata are multiples of 32 columns");
    PN_REQUIRE(a.B == 1 || (a.OH * a.OWsub) % (jb.stat_channel_groups > 1 ? bm : 32) == 0,
               "conv_multi: with batch > 1 a sample must be a whole number of tiles (per-channel statistics) / 32-row segments");
    PN_REQUIRE(jb.stat_channel_groups == 1 || jb.stat_channel_groups == a.ncols, "conv_multi: channel groups must be 1 or the column count");
    const int slots = a.mode == MODE_STRAT ? zdim : S;
    PN_REQUIRE(jb.stat_affine_strata >= slots, "conv_multi: stat_affine_strata too small");
    a.st_part = jb.stat_partials; a.st_S = S; a.st_cg = jb.stat_channel_groups;
    a.st_gamma = jb.stat_gamma; a.st_beta = jb.stat_beta; a.st_eps = jb.stat_eps;
    a.st_ab = jb.stat_affine; a.st_ab_S = jb.stat_affine_strata; a.st_stat = jb.stat_mean_rstd;
  }
  if (jb.norm_affine) {
    PN_REQUIRE(jb.norm_strata >= 1 && a.W % jb.norm_strata == 0 && a.KW <= 3 && a.KH * a.KW <= 16, "conv_multi: bad normalise-on-load geometry");
    PN_REQUIRE(jb.norm_channels >= a.Cin * (d->groups > 1 ? d->groups : 1) && a.B * jb.norm_strata < 1024, "conv_multi: normalisation table too small / too many groups");
    a.ni_ab = jb.norm_affine; a.ni_S = jb.norm_strata; a.ni_C = jb.norm_channels;
    extra = std::max(extra, (size_t)a.B * a.ni_S * a.ni_C * 2 * sizeof(float));
  }
  return PN_OK;
}

int pn_conv2d_multi_f32(const pn_conv_job* jobs, int njobs, int tile, pn_stream_t stream) {
  PN_REQUIRE(jobs && njobs >= 1 && njobs <= kMaxJobs, "conv_multi: 1 .. 8 jobs");
  PN_REQUIRE(tile == 1 || tile == 3 || tile == 4 || tile == 5, "conv_multi: tile must be 1 (128x128), 3 (64x64), 4 (64x32) or 5 (64x128)");
  MultiArgs m;
  memset(&m, 0, sizeof(m));
  m.njobs = njobs;
  bool norm_in = false;
  size_t extra = 0;
  for (int j = 0; j < njobs; ++j) {
    if (int rc = job_to_args(jobs[j], tile, m.job[j], extra)) return rc;
    norm_in = norm_in || m.job[j].ni_ab != nullptr;
  }
  if (norm_in)
    for (int j = 0; j < njobs; ++j) PN_REQUIRE(m.job[j].ni_ab, "conv_multi: normalise-on-load must be given for every job of the launch or for none");
  hipStream_t st = pn::S(stream);
  switch (tile) {
    case 1: PN_REQUIRE(!norm_in, "conv_multi: tile 1 has no normalise-on-load variant"); return launch_multi<2, 2, 2, 2, false>(m, 0, st);
    case 3: return norm_in ? launch_multi<2, 2, 1, 1, true>(m, extra, st) : launch_multi<2, 2, 1, 1, false>(m, 0, st);
    case 4: return norm_in ? launch_multi<2, 1, 1, 1, true>(m, extra, st) : launch_multi<2, 1, 1, 1, false>(m, 0, st);
    default: PN_REQUIRE(!norm_in, "conv_multi: tile 5 has no normalise-on-load variant"); return launch_multi<2, 4, 1, 1, false>(m, 0, st);
  }
}

// second half of a 3x3 / stride 1 / pad 1 convolution with one to three output channels computed as a GEMM over the pixels
// (g[pixel][tap * cout + co] = x[pixel] . w[co][:][tap], pn_linear_ksplit_f32 with the (9 cout, cin) tap matrix -- the input is read
// once) : out[b, y, x, co] = act(scale * sum_t g[b, y + kh - 1, x + kw - 1][t cout + co] + shift), taps in ascending order, pixels
// outside the map contribute nothing.  (The E2ESWVoteHead's 256 -> 1 heat-map / vote-class convolutions, e2e_swv_head.py:88-97, took
// 156 us each on a 32-column MFMA tile.)
int pn_conv3x3_tap_sum_f32(const float* g, int ldg, int batch, int h, int w, int cout, const float* scale, const float* shift, int act, float* out,
                           int out_pixel_stride, int out_channel_offset, pn_stream_t stream) {
  PN_REQUIRE(g && out && batch >= 1 && h >= 1 && w >= 1 && cout >= 1 && cout <= 3 && ldg >= 9 * cout, "conv3x3_tap_sum: bad arguments (cout 1 .. 3, ldg >= 9 cout)");
  PN_REQUIRE(out_pixel_stride >= out_channel_offset + cout && out_channel_offset >= 0, "conv3x3_tap_sum: channel slice does not fit the pixel stride");
  PN_REQUIRE((long long)batch * h * w * cout < (1ll << 31), "conv3x3_tap_sum: map too large");
  const int total = batch * h * w * cout;
  hipLaunchKernelGGL(conv_tap_sum_kernel, dim3(pn::cdiv(total, 256)), dim3(256), 0, pn::S(stream), g, ldg, batch, h, w, cout, scale, shift, act, out,
                     out_pixel_stride, out_channel_offset);
  return pn::check_launch("conv_tap_sum_kernel");
}

int pn_conv2d_small_n_multi_f32(const pn_conv_job* jobs, int njobs, pn_stream_t stream) {
  PN_REQUIRE(jobs && njobs >= 1 && njobs <= kMaxJobs, "conv_small_n_multi: 1 .. 8 jobs");
  MultiArgs m;
  memset(&m, 0, sizeof(m));
  m.njobs = njobs;
  int total = 0;
  // the 3 x 3 jobs first: their blocks are the long ones (the hardware starts blocks in index order, the short 1 x 1 blocks fill the tail)
  int order[kMaxJobs], no = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int j = 0; j < njobs; ++j)
      if ((jobs[j].desc.kh * jobs[j].desc.kw == 9) == (pass == 0)) order[no++] = j;
  for (int jo = 0; jo < njobs; ++jo) {
    const int j = jo;
    ConvArgs& a = m.job[j];
    size_t extra = 0;
    if (int rc = job_to_args(jobs[order[jo]], 3, a, extra)) return rc;
    const int taps = a.KH * a.KW;
    PN_REQUIRE(a.mode == MODE_CONV && a.zdim == 1 && a.ncols <= 12 && a.Cin <= 64 && (taps == 9 || taps == 1) && !jobs[order[jo]].stat_partials,
               "conv_small_n_multi: plain 1x1 / 3x3 convolutions with <= 64 input channels and <= 12 output columns, no statistics");
    PN_REQUIRE(!a.ni_ab || a.ni_S == 1 || taps == 1, "conv_small_n_multi: a range-stratified norm table needs a 1x1 kernel");
    PN_REQUIRE(!a.ni_ab || a.ni_S > 1 || a.B == 1 || (a.OH * a.OW) % 64 == 0, "conv_small_n_multi: with batch > 1 a sample must be a whole number of 64-pixel tiles");
    m.first[j] = total;
    static const int two = [] { const char* e = getenv("PN_SMALL_N_TWO"); return e ? atoi(e) : 1; }();
    a.snt2 = two && taps == 9 && small_n_tiled_ok(a) && a.OH % SN2_H == 0 && a.Cin % 4 == 0;
    total += a.snt2 ? a.M / 128 : pn::cdiv(a.M, 64);
  }
  m.first[njobs] = total;
  m.total = total;
  hipStream_t st = pn::S(stream);
  pn::ProfileSlot ps;
  // every job in 8 x 16 tiles: the matrix-pipe form (PN_SMALL_N_MFMA=0: the packed-fma form).  Jobs with Cout >= 4 give the packed-fma form's
  // bits (an MFMA is a k-ordered fma chain); jobs with Cout <= 3 take small_n_gform_body, which sums the channels first and the taps last:
  // another fp32 order, same parity tests -- the switch is NOT a bitwise A/B for the reg / height / rot branches
  static const int use_mfma = [] { const char* e = getenv("PN_SMALL_N_MFMA"); return e ? atoi(e) : 1; }();
  bool all_tiles = use_mfma != 0;
  for (int j = 0; j < njobs; ++j) all_tiles = all_tiles && (m.job[j].snt2 || m.job[j].KH * m.job[j].KW == 1);
  if (all_tiles) {
    if (pn::take_profile_slot(ps)) hipExtLaunchKernelGGL(conv_small_n_mfma_multi_kernel, dim3(total), dim3(256), 0, st, ps.start, ps.stop, 0, m);
    else hipLaunchKernelGGL(conv_small_n_mfma_multi_kernel, dim3(total), dim3(256), 0, st, m);
    return pn::check_launch("conv_small_n_mfma_multi_kernel");
  }
  if (pn::take_profile_slot(ps)) hipExtLaunchKernelGGL(conv_small_n_multi_kernel, dim3(total), dim3(256), 0, st, ps.start, ps.stop, 0, m);
  else hipLaunchKernelGGL(conv_small_n_multi_kernel, dim3(total), dim3(256), 0, st, m);
  return pn::check_launch("conv_small_n_multi_kernel");
}

int pn_conv_stats_finalize_f32(const pn_conv_job* jobs, int njobs, int tile, pn_stream_t stream) {
  PN_REQUIRE(jobs && njobs >= 1 && njobs <= kMaxJobs, "conv_stats_finalize: 1 .. 8 jobs");
  PN_REQUIRE(tile == 1 || tile == 3 || tile == 4 || tile == 5, "conv_stats_finalize: bad tile");
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.bm = tile_bm(tile);
  f.wn = tile_wn(tile);
  int n = 0, total = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!jobs[j].stat_partials) continue;
    PN_REQUIRE(jobs[j].stat_affine || jobs[j].stat_mean_rstd, "conv_stats_finalize: no output table");
    ConvArgs& a = f.job[n];
    size_t extra = 0;
    if (int rc = job_to_args(jobs[j], tile, a, extra)) return rc;
    f.first[n] = total;
    total += a.st_cg > 1 ? a.zdim * a.B * pn::cdiv(a.cout_pad, 16) : a.zdim * a.B * a.st_S;
    ++n;
  }
  PN_REQUIRE(n > 0, "conv_stats_finalize: no job carries statistics");
  f.njobs = n;
  f.first[n] = total;
  f.total = total;
  hipLaunchKernelGGL(conv_stats_finalize_kernel, dim3(total), dim3(256), 0, pn::S(stream), f);
  return pn::check_launch("conv_stats_finalize_kernel");
}

int pn_conv_stats_apply_f32(const pn_conv_job* producer, int tile, const float* gamma, const float* beta, int act, float* out,
                            int out_pixel_stride, int out_channel_offset, const float* mul, const float* add, float* out2,
                            int out2_pixel_stride, int out2_channel_offset, pn_stream_t stream) {
  PN_REQUIRE(producer && producer->stat_partials && out, "conv_stats_apply: null pointer");
  PN_REQUIRE(tile == 1 || tile == 3 || tile == 4 || tile == 5, "conv_stats_apply: bad tile");
  ApplyArgs g;
  memset(&g, 0, sizeof(g));
  size_t extra = 0;
  if (int rc = job_to_args(*producer, tile, g.conv, extra)) return rc;
  const ConvArgs& a = g.conv;
  PN_REQUIRE(a.mode == MODE_CONV && a.zdim == 1 && a.st_cg == 1, "conv_stats_apply: the producer must be a plain convolution with all-channel statistics");
  const int c = a.ncols;
  PN_REQUIRE(c % 4 == 0 && c <= 1024 && 1024 % c == 0, "conv_stats_apply: channel count must be a multiple of 4 dividing 1024");
  PN_REQUIRE(out_pixel_stride % 4 == 0 && out_channel_offset % 4 == 0, "conv_stats_apply: output stride / offset must be multiples of 4");
  PN_REQUIRE(out2 == nullptr || (mul && add && out2_pixel_stride % 4 == 0 && out2_channel_offset % 4 == 0), "conv_stats_apply: out2 needs mul, add and aligned strides");
  g.wn = tile_wn(tile);
  int splits = 1;
  while (splits < a.OH && (long long)a.B * a.st_S * splits < 512) splits *= 2;
  g.splits = splits;
  g.rows_per_split = pn::cdiv(a.OH, splits);
  g.gamma = gamma; g.beta = beta; g.act = act;
  g.out = out; g.ops = out_pixel_stride; g.oco = out_channel_offset;
  g.mul = mul; g.add = add; g.out2 = out2; g.o2ps = out2_pixel_stride; g.o2co = out2_channel_offset;
  hipLaunchKernelGGL(conv_stats_apply_kernel, dim3(splits, a.st_S, a.B), dim3(256), 0, pn::S(stream), g);
  return pn::check_launch("conv_stats_apply_kernel");
}

int pn_gemm_bias_act_f32(const float* x, int m, int k, int ldx, const float* packed_w, int n, const float* bias, int act,
                         const float* residual, int ldr, float* out, int ldo, pn_stream_t stream) {
  PN_REQUIRE(x && packed_w && out && m > 0 && k > 0 && n > 0, "gemm: bad arguments");
  PN_REQUIRE(k % 4 == 0 && ldx % 4 == 0 && ldx >= k && ldo >= n, "gemm: k and the row strides must be multiples of 4");
  PN_REQUIRE(residual == nullptr || ldr >= n, "gemm: bad residual stride");
  pn_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.batch = 1; d.in_h = m; d.in_w = 1; d.cin = k; d.cout = n; d.groups = 1; d.kh = d.kw = 1; d.stride = 1;
  d.in_pixel_stride = ldx; d.out_pixel_stride = ldo; d.act = act;
  ConvArgs a;
  int zdim = 1;
  if (int rc = fill_args(&d, a, zdim)) return rc;
  a.in = x; a.w = packed_w; a.scale = nullptr; a.shift = bias; a.out = out;
  a.res = residual; a.res_ps = ldr;
  return dispatch_conv(a, zdim, pn::S(stream));
}

int pn_conv2d_direct_nhwc_f32(const pn_conv_desc* d, const float* in, const float* w_oihw, const float* scale,
                              const float* shift, float* out, pn_stream_t stream) {
  ConvArgs a;
  int zdim = 1;
  if (int rc = fill_args(d, a, zdim)) return rc;
  PN_REQUIRE(in && w_oihw && out, "conv_direct: null pointer");
  PN_REQUIRE(!d->deconv2x2 && d->range_strata <= 1, "conv_direct: plain (grouped) convolutions only");
  a.in = in; a.w = nullptr; a.scale = scale; a.shift = shift; a.out = out;
  const size_t total = (size_t)a.B * a.OH * a.OW * a.Cout * d->groups;
  hipLaunchKernelGGL(conv_direct_kernel, dim3(std::min<size_t>(65535, (total + 255) / 256)), dim3(256), 0,
                     pn::S(stream), a, w_oihw, d->groups, total);
  return pn::check_launch("conv_direct_kernel");
}

}  // extern "C"
