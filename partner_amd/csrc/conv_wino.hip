// 3x3 / stride-1 / pad-1 convolution with a one-dimensional Winograd transform F(2, 3) along the image width, on the fp32 MFMA.
// Replaces (arithmetic) the same Conv2d + folded BatchNorm + ReLU layers of the RPN as conv_mfma.hip does
// (det3d/models/necks/rpn.py:124-142) -- those layers are MFMA-bound, and F(2, 3) needs 4 products for 2 outputs of a 3-tap
// filter: 6 instead of 9 MFMA-equivalents per output (1.5x fewer matrix cycles) at unchanged accumulator and staging budgets.
//
//   For a PAIR of horizontally adjacent outputs (x = 2p, 2p + 1) of one row and the four input pixels d0..d3 at x = 2p-1 .. 2p+2
//   of input row y + kh - 1:   m0 = (d0 - d2) g0     m1 = (d1 + d2) (g0 + g1 + g2) / 2     m2 = (d2 - d1) (g0 - g1 + g2) / 2     m3 = (d1 - d3) g2
//   out[2p] = sum_kh sum_ci (m0 + m1 + m2),   out[2p + 1] = sum_kh sum_ci (m1 - m2 - m3)        (g = the three kw taps of row kh).
//   So the layer is FOUR independent GEMMs (one per Winograd position q) with M = pairs, N = Cout, K = 3 * Cin, whose A operands are
//   sums / differences of two input pixels (formed in registers on the way into LDS: no transformed tensor is ever materialised)
//   and whose B operands are the transformed weights (packed once).  A block = 64 pairs (128 output pixels) x 64 columns; wave q
//   runs GEMM q on a 64 x 64 wave tile (the 2 x 2 MFMA tiling of the 128 x 128 direct tile); the output transform joins the four
//   waves' tiles through LDS in the epilogue, where the per-channel affine and the activation are applied as in the direct kernel.
//
// K step = 32 input channels of one kernel row kh: A stage [4 q][64 pairs][32 + 4], B stage [4 q][8 k4][64 cols][4] floats
// (69.6 KB; two stages).  MFMA operand layouts are those of conv_mfma.hip.
// Numerics: every product differs from the direct form by the rounding of one extra add on each operand (|error| ~ 1e-7 relative
// per term); the parity tests hold the same 1e-4 bound as the direct kernel.
#include "pn_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WBN = 64;                         // output columns per block tile
// LDS of a block: two stages of A [4 q][wp pairs][bkc + 4] and B [4 q][bkc / 4][64 cols][4] floats, reused by the epilogue's
// [4 q][wp][68] tile.  64 pairs x 32 channels: 139 KB (one block per CU); 64 pairs x 16 channels: 74 KB (two blocks per CU: one
// block's prologue / epilogue runs behind the other's MFMAs); 32 pairs x 32 channels: 102 KB.
constexpr size_t wino_smem(int wp, int bkc) {
  const size_t stage = 2 * (size_t)(4 * wp * (bkc + 4) + 4 * (bkc / 4) * WBN * 4) * sizeof(float);
  const size_t epi = (size_t)4 * wp * (WBN + 4) * sizeof(float);
  return stage > epi ? stage : epi;
}

struct WinoArgs {
  const float* in;
  const float* w;
  const float* scale;
  const float* shift;
  float* out;
  int B, H, W, Cin, Cout;
  int in_ps, in_co, out_ps, out_co;
  int act;
  int pairs_per_row, total_pairs, ptiles;
  int ncol;                // 64-column tiles
  int chunks, cout_pad;
  unsigned in_bytes, w_bytes;
};

// NW = 4: wave q = position q on a WP x 64 wave tile; NW = 8: waves q and q + 4 share position q and take 32 columns each
// (two waves per SIMD: the second wave fills the first one's barrier / staging bubbles).  WP = pairs per block tile: 64, or 32 for
// maps whose 64-pair tiles would not fill the chip (the 64 x 64 layers).  BKC = input channels per K step (32 or 16).
template <int NW, int WP, int BKC>
__global__ __launch_bounds__(NW * 64) void conv_wino_kernel(WinoArgs a) {
  constexpr int NT = NW * 64;
  constexpr int KQ = BKC / 4;              // 16-byte channel quads per row and K step
  constexpr int NH = 32 / BKC;             // K steps per 32-channel chunk of the packed weights
  constexpr int NSUB = BKC / 8;            // 8-channel MFMA sub-steps per K step
  constexpr int WA_LD = BKC + 4;
  constexpr int WA_FLOATS = 4 * WP * WA_LD;
  constexpr int WB_FLOATS = 4 * KQ * WBN * 4;
  constexpr int WSTAGE = WA_FLOATS + WB_FLOATS;
  constexpr int APASS = NT / KQ;           // pairs covered by one pass of the block's threads
  constexpr int PPT = WP / APASS > 0 ? WP / APASS : 1;   // pairs per thread in the loader
  constexpr bool A_ALL = WP >= APASS;      // false: only the first WP * KQ threads load activations
  constexpr int BPT = (4 * KQ * WBN) / NT; // weight float4 per thread
  constexpr int TN = NW == 4 ? 2 : 1;      // 32-column MFMA tiles per wave
  constexpr int TM = WP / 32;              // 32-pair MFMA tiles per wave
  static_assert(BPT >= 1 && (4 * KQ * WBN) % NT == 0, "weight tile / threads mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wq = wave & 3, wc = wave >> 2;   // position, column half (NW == 8)
  // PERSISTENT blocks: gridDim.x blocks (one per CU), dealt round-robin over the 8 XCDs by the hardware; every XCD owns a contiguous
  // run of pair tiles (neighbouring tiles share halo rows in that XCD's L2), its blocks walk the run's (pair tile, column tile) list
  // with the column tile fastest.  The next tile's first three K steps are requested BEFORE the previous tile's epilogue runs, so
  // the prologue's memory latency hides behind the output transform and stores.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int xq = a.ptiles >> 3, xr = a.ptiles & 7;
  const int px0 = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xtiles = (xcd < xr ? xq + 1 : xq) * a.ncol;
  int pt = 0, n0 = 0;
  const int pl = tid / KQ, c4 = tid % KQ;

  // ---- loader state: PPT pairs per thread, four input pixels each
  const long long back = ((long long)a.W + 1) * a.in_ps;   // floats: the descriptor base is moved back so that every voffset >= 0
  unsigned a_off[PPT];     // byte offset of pixel (b, oh - 1, 2 owp - 1), channel in_co + 4 c4, relative to the shifted base
  unsigned a_rmask[PPT];   // bit kh: input row oh + kh - 1 inside the map
  unsigned a_cmask[PPT];   // bit j: input column 2 owp - 1 + j inside the map
  unsigned b_off[BPT];
  int ld_kh = 0, ld_half = 0, ld_chunk = 0;   // K order: 32-channel chunk outermost, then its BKC-channel parts, kernel rows innermost
  auto setup_tile = [&](int tl) {
    pt = px0 + tl / a.ncol;
    n0 = (tl - (tl / a.ncol) * a.ncol) * WBN;
    ld_kh = 0; ld_half = 0; ld_chunk = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = pt * WP + pl + APASS * k;
      const bool ok = p < a.total_pairs && (A_ALL || tid < WP * KQ);
      const int pp = ok ? p : 0;
      const int rowi = pp / a.pairs_per_row, owp = pp - rowi * a.pairs_per_row;
      const int b = rowi / a.H, oh = rowi - b * a.H;
      const long long pix = ((long long)b * a.H + (oh - 1)) * a.W + (2 * owp - 1);
      a_off[k] = (unsigned)((pix * a.in_ps + back + a.in_co + c4 * 4) * 4);
      unsigned rm = 0, cm = 0;
      for (int kh = 0; kh < 3; ++kh)
        if (ok && (unsigned)(oh + kh - 1) < (unsigned)a.H) rm |= 1u << kh;
      for (int j = 0; j < 4; ++j)
        if ((unsigned)(2 * owp - 1 + j) < (unsigned)a.W) cm |= 1u << j;
      a_rmask[k] = rm;
      a_cmask[k] = cm;
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int idx = tid + NT * j;                 // (q, k4, col) of this K step's weight tile
      const int col = idx & (WBN - 1), k4 = (idx >> 6) % KQ, q = idx / (WBN * KQ);
      b_off[j] = (n0 + col < a.cout_pad) ? (unsigned)((((size_t)q * 8 + k4) * a.cout_pad + n0 + col) * 16) : 0xffffffffu;
    }
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.in)) - back * 4, 0,
                                                                         a.in_bytes + (unsigned)(back * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

  const int nsteps = 3 * NH * a.chunks;
  f32x4 ra[PPT][4], rb[BPT], ra2[PPT][4], rb2[BPT], ra0[PPT][4], rb0[BPT];
  auto load_global = [&](bool live, f32x4 (&ra)[PPT][4], f32x4 (&rb)[BPT]) {
    const int ch0 = ld_chunk * 32 + ld_half * BKC;
    const unsigned so_a = (unsigned)((ld_kh * a.W * a.in_ps + ch0) * 4);
    // packed weights [chunk][kh][q][k4 (8)][col][4]: the part's quads are k4 = half * KQ ..
    const unsigned so_b = ((unsigned)((ld_chunk * 3 + ld_kh) * 32) + (unsigned)(ld_half * KQ)) * (unsigned)a.cout_pad * 16u;
    const bool cok = live && ch0 + c4 * 4 < a.Cin;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const bool rok = cok && ((a_rmask[k] >> ld_kh) & 1u);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned vo = (rok && ((a_cmask[k] >> j) & 1u)) ? a_off[k] + (unsigned)(j * a.in_ps * 4) : 0xffffffffu;
        ra[k][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, so_a, 0));
      }
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, live ? b_off[j] : 0xffffffffu, live ? so_b : 0u, 0));
    if (++ld_kh == 3) {
      ld_kh = 0;
      if (++ld_half == NH) { ld_half = 0; ++ld_chunk; }
    }
  };
  auto store_lds = [&](int buf, const f32x4 (&ra)[PPT][4], const f32x4 (&rb)[BPT]) {
    float* As = smem + buf * WSTAGE;
    float* Bs = As + WA_FLOATS;
    if (A_ALL || tid < WP * KQ) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int pair = pl + APASS * k;
        const f32x4 d0 = ra[k][0], d1 = ra[k][1], d2 = ra[k][2], d3 = ra[k][3];
        *reinterpret_cast<f32x4*>(As + (0 * WP + pair) * WA_LD + c4 * 4) = d0 - d2;
        *reinterpret_cast<f32x4*>(As + (1 * WP + pair) * WA_LD + c4 * 4) = d1 + d2;
        *reinterpret_cast<f32x4*>(As + (2 * WP + pair) * WA_LD + c4 * 4) = d2 - d1;
        *reinterpret_cast<f32x4*>(As + (3 * WP + pair) * WA_LD + c4 * 4) = d1 - d3;
      }
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j) *reinterpret_cast<f32x4*>(Bs + (size_t)(tid + NT * j) * 4) = rb[j];
  };

  f32x16 acc[TM][TN];

  const int a_frag = (wq * WP + li) * WA_LD + lh * 4;
  const int b_frag = WA_FLOATS + ((wq * KQ + lh) * WBN + wc * 32 + li) * 4;
  f32x4 af[2][TM], bf[2][TN];
  auto read_frags = [&](int buf, int sub, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
    const float* As = smem + buf * WSTAGE + a_frag + sub * 8;
    const float* Bs = smem + buf * WSTAGE + b_frag + sub * 2 * WBN * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * WA_LD);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * 4);
  };
  auto mfma_sub = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
  };

  // ---- software pipeline (the schedule of conv_mfma.hip): one barrier per K step, everything else in the shadow of the MFMAs.
  // Every sub-step reads the next sub-step's fragments, then issues its MFMAs; the Winograd input transform + LDS stores of step
  // t+1 (into the other stage) ride on sub-step STORE_SUB, the buffer loads of step t+3 (into the registers just stored; two sets
  // alternate) on sub-step LOAD_SUB, the barrier sits before the last sub-step, which reads the first fragments of step t+1.
  constexpr int NM = 4 * TM * TN;          // MFMAs per sub-step
  constexpr int NF = TM + TN;              // fragment reads per sub-step
  constexpr int NS = PPT * 4 + BPT;        // LDS stores per step == buffer loads per step
  constexpr int STORE_SUB = NSUB == 4 ? 1 : 0, LOAD_SUB = NSUB == 4 ? 2 : 0;
  auto kstep = [&](int t, int buf, f32x4 (&rx)[PPT][4], f32x4 (&ry)[BPT]) {
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      f32x4 (&ca)[TM] = af[sub & 1];
      f32x4 (&cb)[TN] = bf[sub & 1];
      f32x4 (&na)[TM] = af[(sub + 1) & 1];
      f32x4 (&nb)[TN] = bf[(sub + 1) & 1];
      if (sub == NSUB - 1) {
        __syncthreads();
        read_frags(buf ^ 1, 0, na, nb);
      } else {
        read_frags(buf, sub + 1, na, nb);
      }
      mfma_sub(ca, cb);
      if (sub == STORE_SUB) store_lds(buf ^ 1, rx, ry);
      if (sub == LOAD_SUB) load_global(t + 3 < nsteps, rx, ry);
      __builtin_amdgcn_sched_group_barrier(0x100, NF, 0);
      if (sub == STORE_SUB) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, NM / (2 * NS) > 0 ? NM / (2 * NS) : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
      if (sub == LOAD_SUB) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, NM / (2 * NS) > 0 ? NM / (2 * NS) : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- epilogue of a finished tile: the four positions' tiles through LDS, output transform, affine + activation, two pixels per pair
  constexpr int TLD = WBN + 4;
  const bool vec_cols = (a.out_ps % 4 == 0) && (a.out_co % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  auto epilogue = [&](int ept, int en0) {
    float* T = smem;   // [4 q][WP pairs][TLD] (69.6 KB at 64 pairs): the staging buffers are free (the K loop ended with a barrier)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          T[(wq * WP + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + wc * 32 + j * 32 + li] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (WP * 16) / NT; ++k) {
      const int item = tid + NT * k;
      const int cq = item & 15, pair = item >> 4;
      const int p = ept * WP + pair;
      const int col = en0 + cq * 4;
      if (p >= a.total_pairs || col >= a.Cout) continue;
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(T + (0 * WP + pair) * TLD + cq * 4);
      const f32x4 m1 = *reinterpret_cast<const f32x4*>(T + (1 * WP + pair) * TLD + cq * 4);
      const f32x4 m2 = *reinterpret_cast<const f32x4*>(T + (2 * WP + pair) * TLD + cq * 4);
      const f32x4 m3 = *reinterpret_cast<const f32x4*>(T + (3 * WP + pair) * TLD + cq * 4);
      f32x4 y0 = (m0 + m1) + m2, y1 = (m1 - m2) - m3;
      const int rowi = p / a.pairs_per_row, owp = p - rowi * a.pairs_per_row;
      float* o = a.out + ((size_t)rowi * a.W + 2 * owp) * a.out_ps + a.out_co + col;
      const int nvalid = min(4, a.Cout - col);
      if (vec_cols && nvalid == 4) {
        f32x4 vs = {1.f, 1.f, 1.f, 1.f}, vh = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) vs = *reinterpret_cast<const f32x4*>(a.scale + col);
        if (a.shift) vh = *reinterpret_cast<const f32x4*>(a.shift + col);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          y0[c] = pn::apply_act(fmaf(y0[c], vs[c], vh[c]), a.act);
          y1[c] = pn::apply_act(fmaf(y1[c], vs[c], vh[c]), a.act);
        }
        *reinterpret_cast<f32x4*>(o) = y0;
        *reinterpret_cast<f32x4*>(o + a.out_ps) = y1;
      } else {
        for (int c = 0; c < nvalid; ++c) {
          const float sc = a.scale ? a.scale[col + c] : 1.f, sh = a.shift ? a.shift[col + c] : 0.f;
          o[c] = pn::apply_act(fmaf(y0[c], sc, sh), a.act);
          o[a.out_ps + c] = pn::apply_act(fmaf(y1[c], sc, sh), a.act);
        }
      }
    }
    __syncthreads();   // the tile in LDS has been consumed: the next tile's first stage may be stored
  };

  int prev_pt = -1, prev_n0 = 0;
  for (int tl = slot; tl < xtiles; tl += per_xcd) {
    setup_tile(tl);
    load_global(true, ra0, rb0);
    load_global(nsteps > 1, ra, rb);
    load_global(nsteps > 2, ra2, rb2);
    if (prev_pt >= 0) epilogue(prev_pt, prev_n0);   // runs while the three tiles just requested are in flight
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    store_lds(0, ra0, rb0);
    __syncthreads();
    read_frags(0, 0, af[0], bf[0]);
    for (int t = 0; t < nsteps; t += 2) {
      kstep(t, 0, ra, rb);
      if (t + 1 < nsteps) kstep(t + 1, 1, ra2, rb2);
    }
    __syncthreads();   // every wave is done with the last stage before the epilogue reuses the LDS
    prev_pt = pt;
    prev_n0 = n0;
  }
  if (prev_pt >= 0) epilogue(prev_pt, prev_n0);
}

// ---- variant for layers with >= 128 output columns: a block covers WP pairs x 128 columns with EIGHT 64-wide wave tiles (wave = position
// q x column half), the transformed INPUT goes through LDS as above, the transformed WEIGHTS do not: every wave's B operand is private
// to it (its position, its 64 columns), and the packed layout is the MFMA fragment layout, so each lane fetches its own 16-byte
// fragments straight from L2 one K step ahead.  LDS is the A image only (74 KB at 64 pairs), a barrier covers 64 MFMAs per wave as in
// the direct 128 x 128 tile, and the input transform is done once per 128 columns instead of once per 64.
constexpr int WBN2 = 128;
constexpr size_t wino_bd_smem(int wp) {
  const size_t stage = 2 * (size_t)(4 * wp * 36) * sizeof(float);
  const size_t epi = (size_t)4 * wp * (64 + 4) * sizeof(float);   // the epilogue handles the two column halves one after the other
  return stage > epi ? stage : epi;
}

template <int WP>
__global__ __launch_bounds__(512) void conv_wino_bd_kernel(WinoArgs a) {
  constexpr int NT = 512, WA_LD = 36, WA_FLOATS = 4 * WP * WA_LD;
  constexpr int TM = WP / 32;
  constexpr bool A_ALL = WP * 8 >= NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wq = wave & 3, wc = wave >> 2;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int xq = a.ptiles >> 3, xr = a.ptiles & 7;
  const int px0 = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xtiles = (xcd < xr ? xq + 1 : xq) * a.ncol;   // a.ncol = 128-column tiles here
  int pt = 0, n0 = 0;
  const int pl = tid >> 3, c4 = tid & 7;

  const long long back = ((long long)a.W + 1) * a.in_ps;
  unsigned a_off, a_rmask, a_cmask;
  unsigned b_base;   // byte offset of this lane's fragment for (k4 = lh, column j = 0) inside one (chunk, kh) block of the packed weights
  int ld_kh = 0, ld_chunk = 0;     // A loader position
  int lb_kh = 0, lb_chunk = 0;     // B loader position
  auto setup_tile = [&](int tl) {
    pt = px0 + tl / a.ncol;
    n0 = (tl - (tl / a.ncol) * a.ncol) * WBN2;
    ld_kh = ld_chunk = lb_kh = lb_chunk = 0;
    const int p = pt * WP + pl;
    const bool ok = p < a.total_pairs && (A_ALL || tid < WP * 8);
    const int pp = ok ? p : 0;
    const int rowi = pp / a.pairs_per_row, owp = pp - rowi * a.pairs_per_row;
    const int b = rowi / a.H, oh = rowi - b * a.H;
    const long long pix = ((long long)b * a.H + (oh - 1)) * a.W + (2 * owp - 1);
    a_off = (unsigned)((pix * a.in_ps + back + a.in_co + c4 * 4) * 4);
    unsigned rm = 0, cm = 0;
    for (int kh = 0; kh < 3; ++kh)
      if (ok && (unsigned)(oh + kh - 1) < (unsigned)a.H) rm |= 1u << kh;
    for (int j = 0; j < 4; ++j)
      if ((unsigned)(2 * owp - 1 + j) < (unsigned)a.W) cm |= 1u << j;
    a_rmask = rm;
    a_cmask = cm;
    const int col = n0 + wc * 64 + li;
    b_base = (unsigned)((((size_t)wq * 8 + lh) * a.cout_pad + col) * 16);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.in)) - back * 4, 0,
                                                                         a.in_bytes + (unsigned)(back * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int nsteps = 3 * a.chunks;

  f32x4 ra[4], ra2[4], ra0[4];
  auto load_a = [&](bool live, f32x4 (&r)[4]) {
    const unsigned so_a = (unsigned)((ld_kh * a.W * a.in_ps + ld_chunk * 32) * 4);
    const bool rok = live && ld_chunk * 32 + c4 * 4 < a.Cin && ((a_rmask >> ld_kh) & 1u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned vo = (rok && ((a_cmask >> j) & 1u)) ? a_off + (unsigned)(j * a.in_ps * 4) : 0xffffffffu;
      r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, so_a, 0));
    }
    if (++ld_kh == 3) { ld_kh = 0; ++ld_chunk; }
  };
  auto store_a = [&](int buf, const f32x4 (&r)[4]) {
    if (A_ALL || tid < WP * 8) {
      float* As = smem + buf * WA_FLOATS;
      *reinterpret_cast<f32x4*>(As + (0 * WP + pl) * WA_LD + c4 * 4) = r[0] - r[2];
      *reinterpret_cast<f32x4*>(As + (1 * WP + pl) * WA_LD + c4 * 4) = r[1] + r[2];
      *reinterpret_cast<f32x4*>(As + (2 * WP + pl) * WA_LD + c4 * 4) = r[2] - r[1];
      *reinterpret_cast<f32x4*>(As + (3 * WP + pl) * WA_LD + c4 * 4) = r[1] - r[3];
    }
  };
  // this lane's B fragments of one K step: [sub-step][32-column tile]
  f32x4 fb0[4][2], fb1[4][2];
  auto load_b = [&](bool live, f32x4 (&f)[4][2]) {
    const unsigned so_b = (unsigned)((lb_chunk * 3 + lb_kh) * 32) * (unsigned)a.cout_pad * 16u;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned vo = live ? b_base + (unsigned)(((size_t)2 * sub * a.cout_pad + j * 32) * 16) : 0xffffffffu;
        f[sub][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vo, live ? so_b : 0u, 0));
      }
    if (++lb_kh == 3) { lb_kh = 0; ++lb_chunk; }
  };

  f32x16 acc[TM][2];
  const int a_frag = (wq * WP + li) * WA_LD + lh * 4;
  f32x4 af[2][TM];
  auto read_a = [&](int buf, int sub, f32x4 (&fa)[TM]) {
    const float* As = smem + buf * WA_FLOATS + a_frag + sub * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * WA_LD);
  };
  auto mfma_sub = [&](const f32x4 (&fa)[TM], const f32x4 (&fbv)[2]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fbv[j][kk], acc[i][j], 0, 0, 0);
  };
  constexpr int NM = 4 * TM * 2;
  auto kstep = [&](int t, int buf, f32x4 (&rx)[4], f32x4 (&bc)[4][2], f32x4 (&bn)[4][2]) {
    // sub-step 0: A fragments of sub-step 1; the weight fragments of step t+1 are requested (a whole step ahead of their use)
    read_a(buf, 1, af[1]);
    load_b(t + 1 < nsteps, bn);
    mfma_sub(af[0], bc[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // sub-step 1: A fragments of sub-step 2; input transform + LDS stores of step t+1
    read_a(buf, 2, af[0]);
    mfma_sub(af[1], bc[1]);
    store_a(buf ^ 1, rx);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // sub-step 2: A fragments of sub-step 3; input loads of step t+3 into the registers just stored
    read_a(buf, 3, af[1]);
    mfma_sub(af[0], bc[2]);
    load_a(t + 3 < nsteps, rx);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_a(buf ^ 1, 0, af[0]);
    mfma_sub(af[1], bc[3]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  constexpr int TLD = 64 + 4;
  const bool vec_cols = (a.out_ps % 4 == 0) && (a.out_co % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  auto epilogue = [&](int ept, int en0) {
    float* T = smem;   // [4 q][WP][TLD]: one 64-column half at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wc == half) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(wq * WP + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + j * 32 + li] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < (WP * 16) / NT; ++k) {
        const int item = tid + NT * k;
        const int cq = item & 15, pair = item >> 4;
        const int p = ept * WP + pair;
        const int col = en0 + half * 64 + cq * 4;
        if (p >= a.total_pairs || col >= a.Cout) continue;
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(T + (0 * WP + pair) * TLD + cq * 4);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(T + (1 * WP + pair) * TLD + cq * 4);
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(T + (2 * WP + pair) * TLD + cq * 4);
        const f32x4 m3 = *reinterpret_cast<const f32x4*>(T + (3 * WP + pair) * TLD + cq * 4);
        f32x4 y0 = (m0 + m1) + m2, y1 = (m1 - m2) - m3;
        const int rowi = p / a.pairs_per_row, owp = p - rowi * a.pairs_per_row;
        float* o = a.out + ((size_t)rowi * a.W + 2 * owp) * a.out_ps + a.out_co + col;
        const int nvalid = min(4, a.Cout - col);
        if (vec_cols && nvalid == 4) {
          f32x4 vs = {1.f, 1.f, 1.f, 1.f}, vh = {0.f, 0.f, 0.f, 0.f};
          if (a.scale) vs = *reinterpret_cast<const f32x4*>(a.scale + col);
          if (a.shift) vh = *reinterpret_cast<const f32x4*>(a.shift + col);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            y0[c] = pn::apply_act(fmaf(y0[c], vs[c], vh[c]), a.act);
            y1[c] = pn::apply_act(fmaf(y1[c], vs[c], vh[c]), a.act);
          }
          *reinterpret_cast<f32x4*>(o) = y0;
          *reinterpret_cast<f32x4*>(o + a.out_ps) = y1;
        } else {
          for (int c = 0; c < nvalid; ++c) {
            const float sc = a.scale ? a.scale[col + c] : 1.f, sh = a.shift ? a.shift[col + c] : 0.f;
            o[c] = pn::apply_act(fmaf(y0[c], sc, sh), a.act);
            o[a.out_ps + c] = pn::apply_act(fmaf(y1[c], sc, sh), a.act);
          }
        }
      }
      __syncthreads();
    }
  };

  int prev_pt = -1, prev_n0 = 0;
  for (int tl = slot; tl < xtiles; tl += per_xcd) {
    setup_tile(tl);
    load_a(true, ra0);
    load_b(true, fb0);
    load_a(nsteps > 1, ra);
    load_a(nsteps > 2, ra2);
    if (prev_pt >= 0) epilogue(prev_pt, prev_n0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    store_a(0, ra0);
    __syncthreads();
    read_a(0, 0, af[0]);
    for (int t = 0; t < nsteps; t += 2) {
      kstep(t, 0, ra, fb0, fb1);
      if (t + 1 < nsteps) kstep(t + 1, 1, ra2, fb1, fb0);
    }
    __syncthreads();
    prev_pt = pt;
    prev_n0 = n0;
  }
  if (prev_pt >= 0) epilogue(prev_pt, prev_n0);
}

// torch (Cout, Cin, 3, 3) -> [chunk][kh][q][k4 (8)][cout_pad][4]: U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2
// dgrad: w is the FORWARD weight (cin, cout, 3, 3) of the layer whose data gradient this convolution is -- taps mirrored, channels swapped
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, int cout, int cin, int chunks, int cout_pad, float* __restrict__ packed, size_t total,
                                        int dgrad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3; r >>= 2;
    const int n = (int)(r % cout_pad); r /= cout_pad;
    const int k4 = r & 7; r >>= 3;
    const int q = r & 3; r >>= 2;
    const int kh = (int)(r % 3);
    const int chunk = (int)(r / 3);
    const int c = chunk * 32 + k4 * 4 + k1;
    float v = 0.f;
    if (n < cout && c < cin) {
      const float* g = dgrad ? w + (((size_t)c * cout + n) * 3 + (2 - kh)) * 3 : w + (((size_t)n * cin + c) * 3 + kh) * 3;
      const float g0 = dgrad ? g[2] : g[0], g1 = g[1], g2 = dgrad ? g[0] : g[2];
      v = q == 0 ? g0 : q == 1 ? (g0 + g1 + g2) * 0.5f : q == 2 ? (g0 - g1 + g2) * 0.5f : g2;
    }
    packed[i] = v;
  }
}

}  // namespace

extern "C" {

size_t pn_conv_wino_packed_weight_floats(int cout, int cin) {
  return (size_t)pn::cdiv(cin, 32) * 3 * 4 * 8 * (size_t)(pn::cdiv(cout, WBN) * WBN) * 4;
}

static int pack_wino(const float* w, int cout, int cin, float* packed, pn_stream_t stream, int dgrad) {
  PN_REQUIRE(w && packed && cout >= 1 && cin >= 1, "pack_conv_weight_wino: bad arguments");
  const int chunks = pn::cdiv(cin, 32), cout_pad = pn::cdiv(cout, WBN) * WBN;
  const size_t total = pn_conv_wino_packed_weight_floats(cout, cin);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w, cout, cin, chunks,
                     cout_pad, packed, total, dgrad);
  return pn::check_launch("pack_wino_weight_kernel");
}

int pn_pack_conv_weight_wino_f32(const float* w_oihw, int cout, int cin, float* packed, pn_stream_t stream) {
  return pack_wino(w_oihw, cout, cin, packed, stream, 0);
}

// the weights of the DATA-GRADIENT convolution straight from the forward layer's (Cout_fwd, Cin_fwd, 3, 3) tensor: the gradient conv has
// cout = Cin_fwd output and cin = Cout_fwd input channels, w'[n][c][kh][kw] = w[c][n][2 - kh][2 - kw]
int pn_pack_conv_dgrad_weight_wino_f32(const float* w_fwd_oihw, int cout_fwd, int cin_fwd, float* packed, pn_stream_t stream) {
  return pack_wino(w_fwd_oihw, cin_fwd, cout_fwd, packed, stream, 1);
}

int pn_conv2d_wino_nhwc_f32(const pn_conv_desc* d, const float* in, const float* packed_w, const float* scale, const float* shift, float* out,
                            pn_stream_t stream) {
  PN_REQUIRE(d && in && packed_w && out, "conv_wino: null pointer");
  PN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->groups == 1 && !d->deconv2x2 && d->range_strata <= 1 &&
                 !d->accumulate && d->pad_h_end == 0 && d->pad_w_end == 0,
             "conv_wino: plain 3x3 / stride 1 / pad 1 convolutions only");
  PN_REQUIRE(d->batch >= 1 && d->in_h >= 1 && d->in_w >= 2 && d->in_w % 2 == 0, "conv_wino: the map width must be even");
  PN_REQUIRE(d->cin >= 4 && d->cin % 4 == 0 && d->in_pixel_stride % 4 == 0 && d->in_channel_offset % 4 == 0 && d->cout >= 1,
             "conv_wino: cin, input pixel stride and channel offset must be multiples of 4");
  PN_REQUIRE(d->in_pixel_stride >= d->in_channel_offset + d->cin && d->out_pixel_stride >= d->out_channel_offset + d->cout,
             "conv_wino: channel slice does not fit the pixel stride");
  PN_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "conv_wino: pointers must be 16-byte aligned");
  const unsigned long long in_bytes = (unsigned long long)d->batch * d->in_h * d->in_w * d->in_pixel_stride * 4ull;
  PN_REQUIRE(in_bytes + ((unsigned long long)d->in_w + 1) * d->in_pixel_stride * 4ull < (1ull << 32), "conv_wino: input map too large for the buffer descriptor");
  WinoArgs a{};
  a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift; a.out = out;
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = d->cin; a.Cout = d->cout;
  a.in_ps = d->in_pixel_stride; a.in_co = d->in_channel_offset; a.out_ps = d->out_pixel_stride; a.out_co = d->out_channel_offset;
  a.act = d->act;
  a.pairs_per_row = d->in_w / 2;
  a.total_pairs = d->batch * d->in_h * a.pairs_per_row;
  a.chunks = pn::cdiv(d->cin, 32);
  a.cout_pad = pn::cdiv(d->cout, WBN) * WBN;
  a.in_bytes = (unsigned)in_bytes;
  a.w_bytes = (unsigned)(pn_conv_wino_packed_weight_floats(d->cout, d->cin) * 4);
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<8, 64, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_smem(64, 32));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<8, 32, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_smem(32, 32));
  }
  static const int force_wp = [] { const char* e = getenv("PN_WINO_PAIRS"); return e ? atoi(e) : 0; }();
  // 64-pair tiles when they fill the chip (>= 256 blocks), 32-pair tiles otherwise.  K steps of 32 channels: 16-channel steps fit two
  // blocks per CU (74 KB each) but measured slower on the 256 x 256 layers (8 waves of 64 x 32: 128 us, 4 waves of 64 x 64: 143 us,
  // against 114-117 us): half the MFMAs per barrier costs more than the second block's overlap gives.
  const int ncol = a.cout_pad / WBN;
  int wp = (long long)pn::cdiv(a.total_pairs, 64) * ncol >= 256 ? 64 : 32;
  if (force_wp == 32 || force_wp == 64) wp = force_wp;
  a.ptiles = pn::cdiv(a.total_pairs, wp);
  a.ncol = ncol;
  // one persistent block per CU (a multiple of 8: the XCD count), fewer when there are fewer tiles
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n / 8 * 8 : 256;
  }
  const int ncu = (dev >= 0 && dev < 64) ? cus[dev] : 256;
  const long long tiles = (long long)a.ptiles * ncol;
  const dim3 grid((unsigned)std::min<long long>(ncu, (tiles + 7) / 8 * 8));
  pn::ProfileSlot ps;
  const bool prof = pn::take_profile_slot(ps);
  hipStream_t st = pn::S(stream);
#define PN_WINO_LAUNCH(WPT)                                                                                                              \
  do {                                                                                                                                   \
    if (prof) hipExtLaunchKernelGGL((conv_wino_kernel<8, WPT, 32>), grid, dim3(512), wino_smem(WPT, 32), st, ps.start, ps.stop, 0, a);   \
    else hipLaunchKernelGGL((conv_wino_kernel<8, WPT, 32>), grid, dim3(512), wino_smem(WPT, 32), st, a);                                 \
  } while (0)
  static const int bd_on = [] { const char* e = getenv("PN_WINO_BDIRECT"); return e ? atoi(e) : 1; }();
  if (bd_on && a.cout_pad % WBN2 == 0) {
    // 128-column tiles with the weights fetched straight into the MFMA operands: when they fill the chip
    const int ncol2 = a.cout_pad / WBN2;
    // (the 32-pair form of this kernel measured slower than the LDS-weights kernel on the 128 x 128 layers: 33 vs 31 us)
    const int wp2 = (long long)pn::cdiv(a.total_pairs, 64) * ncol2 >= ncu ? 64 : 0;
    if (wp2) {
      static bool done_bd[64] = {false};
      if (pn::first_use_on_device(done_bd)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_bd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_bd_smem(64));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_bd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_bd_smem(32));
      }
      a.ptiles = pn::cdiv(a.total_pairs, wp2);
      a.ncol = ncol2;
      const long long tiles2 = (long long)a.ptiles * ncol2;
      const dim3 grid2((unsigned)std::min<long long>(ncu, (tiles2 + 7) / 8 * 8));
      if (wp2 == 64) {
        if (prof) hipExtLaunchKernelGGL((conv_wino_bd_kernel<64>), grid2, dim3(512), wino_bd_smem(64), st, ps.start, ps.stop, 0, a);
        else hipLaunchKernelGGL((conv_wino_bd_kernel<64>), grid2, dim3(512), wino_bd_smem(64), st, a);
      } else {
        if (prof) hipExtLaunchKernelGGL((conv_wino_bd_kernel<32>), grid2, dim3(512), wino_bd_smem(32), st, ps.start, ps.stop, 0, a);
        else hipLaunchKernelGGL((conv_wino_bd_kernel<32>), grid2, dim3(512), wino_bd_smem(32), st, a);
      }
      return pn::check_launch("conv_wino_bd_kernel");
    }
  }
  if (wp == 64) PN_WINO_LAUNCH(64);
  else PN_WINO_LAUNCH(32);
#undef PN_WINO_LAUNCH
  return pn::check_launch("conv_wino_kernel");
}

}  // extern "C"
