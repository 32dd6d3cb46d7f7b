// Chains of 3x3 / stride-1 / pad-1 convolutions (Conv2d + folded BatchNorm + ReLU, det3d/models/necks/rpn.py:124-142: the
// `layer_nums` same-shape layers of an RPN block) kept in the WINOGRAD DOMAIN between layers -- r4, the successor of
// conv_wino4_ks_kernel on the 128 x 128 and 64 x 64 maps of the nuScenes RPN.
//
// conv_wino4.hip's F(4, 3) takes an NHWC map, forms the six transformed inputs v0..v5 of every quad (four adjacent pixels of a row)
// in registers, moves them through LDS into the MFMA operand layout (one barrier per 24 MFMAs) and keeps all six positions in one
// wave.  On the small maps that form sits at 0.44 MFMA busy: ~110 transform VALU per 24 MFMAs (fp32 MFMA and VALU share the SIMD's
// fp32 lanes, tools/micro/mfma_valu_coexec.hip), twelve 16-byte loads per 24 MFMAs with no register reuse, one or two waves per SIMD.
//
// Here a layer's EPILOGUE writes what the NEXT layer multiplies: the output quad (after scale / shift / ReLU) is transformed at once
// (d0 and d5, the neighbours' edge pixels, come from the adjacent lanes) and stored as six planes V_p in the MFMA fragment layout.  The
// consumer then needs no transform, no LDS and no barrier in its K loop: every wave streams 16-byte fragments of ONE position p
// (weights U_p as the A operand, inputs V_p as the B operand) straight from L2 into v_mfma_f32_32x32x2_f32.  The six positions of a
// tile are six waves; K may additionally be split in two (KS) and the block may hold several column tiles (CT): 6 KS CT = 12 waves per
// block, THREE per SIMD, independent of each other until the join -- while one wave waits for its fragments the other two multiply.
//
//   V[p 6][cg C/8][h 2][b][y H+2][xq W/4][j 4]   channel c = 8 cg + 4 h + j; rows y = 0 and y = H + 1 of every image are zero (the
//   vertical padding: the three kernel rows are three row shifts of the same plane, no masks in the loop), written by whoever writes
//   the first / last image row.
//   Lane l = (h = l >> 5, i = l & 31) of a wave reads 16 bytes = four k-pairs of quad i: every half-wave 512 contiguous bytes.
//   U: conv_wino4.hip's packing unchanged ([chunk][kh][q 6][k4 8][cout_pad][4], k4 = 2 (cg & 3) + h).
//   D[cout][quad]: column (lane & 31) = quad, row = cout = (r & 3) + 8 (r >> 2) + 4 h -- registers 4 g .. 4 g + 3 of a lane are the four
//   channels j of (cg = g, h) of its quad: the layout of V, so the epilogue stores whole 16-byte fragments.
//
// Join: all waves leave their accumulators in LDS, the tile is then finished by (column tile, g) "virtual waves": sum the K slices,
// output transform, affine + activation, neighbour exchange by lane shuffles, input transform of the next layer, stores (V planes
// and / or NHWC).  Block tiles are whole rows (32 NB quads a multiple of W / 4), so no neighbour is in another block.
// Numerics: the same products and the same two transforms as conv_wino4.hip; only the summation order over K differs (per position,
// two halves).
#include "pn_common.h"
#include "wino_planes.h"
#include <algorithm>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct WChainArgs {
  const float* vin;
  const float* w;
  const float* scale;
  const float* shift;
  float* vout;     // next layer's planes (may be null)
  float* out;      // NHWC output (may be null)
  int B, H, W, Wq, Cin, Cout;
  int wq_log2;            // Wq is a power of two (it divides a 32- or 64-quad tile)
  int out_ps, out_co;      // out_ps here = floats between consecutive pixels ALONG the Winograd axis
  long long out_img, out_row;      // floats between images / between rows of the (possibly transposed) frame
  int act;
  int total_quads;
  int qtiles, ctiles;     // block tiles: quads / columns
  int cg_in, cg_out;      // Cin / 8, Cout / 8
  int cout_pad;
  unsigned plane_bytes;   // B (H + 2) Wq 16: one (p, cg, h) plane, the same for input and output (same map)
  unsigned vin_bytes, w_bytes;
  // ---- head branches (r4): statistics of the output for the GroupNorm that follows, and range-stratified weights
  float* st_part;         // nullable: [tile][row 2][Cout][2] (2-D kernel) per-channel (sum, sum of squares) of the tile's outputs
  int strata_rows;        // > 0: the layer is range-stratified along the frame's ROWS (center_head_parallel.py:27-59 on the transposed map): rows
                          // [s strata_rows, (s + 1) strata_rows) of every image use weight set s
  unsigned w_stratum_bytes;      // bytes between two weight sets
  int qt_begin;                  // 2-D kernel: the launch covers octet tiles [qt_begin, qt_begin + qtiles) (0 everywhere today)
#ifdef PN_WCHAIN_STAMP
  unsigned long long* stamps;   // diagnostic build only (tools/micro/wchain_check.hip): [block][wave][4] shader-clock stamps
#endif
};

#ifdef PN_WCHAIN_STAMP
unsigned long long* pn_wchain_stamp_buffer = nullptr;
#define WC_STAMP(k)                                                                                                      \
  do {                                                                                                                   \
    if (lane == 0) a.stamps[((size_t)blockIdx.x * 16 + w) * 4 + (k)] = __builtin_amdgcn_s_memtime();                       \
    if (lane == 0 && w == 0 && ((k) == 0 || (k) == 3)) a.stamps[((size_t)blockIdx.x * 16 + 12) * 4 + (k)] = wall_clock64(); /* 100 MHz */ \
  } while (0)
#else
#define WC_STAMP(k) do { } while (0)
#endif
#ifndef PN_WC3_EXP
#define PN_WC3_EXP 0      // diagnostic builds only (tools/micro/wchain3_ablate.hip), conv_wchain3_kernel: bit 0 no MFMAs, 1 no join, 2 no finish
#endif
#ifndef PN_WCHAIN_EXP
#define PN_WCHAIN_EXP 0   // diagnostic builds only (tools/micro/wchain_check.hip), 2-D kernel's K loop: bit 0 no height transform, 1 no plane loads, 2 no weight loads
#endif

// B^T d for four channels at once: wino_planes.h (the same expression tree as conv_wino4.hip's wino4_input_transform: bit-identical values)
__device__ __forceinline__ void wchain_input_transform(const f32x4 (&d)[6], f32x4 (&v)[6]) { pn::wino4_input_transform4(d, v); }

// Where element k (0 .. 63) of a tile lies, from the tile's block-uniform first row (image img0, row r0 of `rows` per image): tiles are
// whole rows of Wq = 1 << lg quads and never taller than an image, so no vector division
__device__ __forceinline__ void wchain_coords(int img0, int r0, int rows, int lg, int k, int& img, int& r, int& xq) {
  xq = k & ((1 << lg) - 1);
  r = r0 + (k >> lg);
  img = img0;
  if (r >= rows) {
    r -= rows;
    img += 1;
  }
}

// The tail of a tile for one lane: NT 32-quad tiles that follow each other along the map rows (NT 32 quads = whole rows), four output
// channels c0 .. c0 + 3 (tiles are always whole).  get_m(q, b) = the lane's four channels of position q of tile b (K slices summed); quad_of(b, ...) = where
// the lane's quad of tile b lies.  Output transform, affine + activation, the neighbours' edge pixels by lane shuffles, the next layer's
// input transform, stores: six plane fragments (+ the zero padding rows next to the first / last image row) and / or four NHWC pixels.
template <int NT, typename GetM, typename QuadOf>
__device__ __forceinline__ void wchain_finish(const WChainArgs& a, int c0, int li, int lh, float lo, GetM get_m, QuadOf quad_of, float* part = nullptr,
                                              int aff_off = 0) {
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + aff_off + c0);      // (aff_off: the tile's stratum x Cout of a stratified layer)
  if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + aff_off + c0);
  f32x4 y[NT][4];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    f32x4 m[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) m[q] = get_m(q, b);
    const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    const f32x4 y0 = (m[0] + s12) + s34;
    const f32x4 y1 = d12 + 2.f * d34;
    const f32x4 y2 = s12 + 4.f * s34;
    const f32x4 y3 = (d12 + 8.f * d34) + m[5];
    const f32x4 lo4 = {lo, lo, lo, lo};
    y[b][0] = __builtin_elementwise_max(__builtin_elementwise_fma(y0, sc, sh), lo4);
    y[b][1] = __builtin_elementwise_max(__builtin_elementwise_fma(y1, sc, sh), lo4);
    y[b][2] = __builtin_elementwise_max(__builtin_elementwise_fma(y2, sc, sh), lo4);
    y[b][3] = __builtin_elementwise_max(__builtin_elementwise_fma(y3, sc, sh), lo4);
  }
  if (part) {
    // per-channel (sum, sum of squares) of this virtual wave's outputs (NT x 32 quads x 4 pixels per channel), lanes folded by a fixed
    // xor butterfly inside each lane half (a half = four channels): part[c0 + j] = (sum, sumsq)
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        s1 += y[b][px];
        s2 += y[b][px] * y[b][px];
      }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += __shfl_xor(s1[j], o, 32);
        s2[j] += __shfl_xor(s2[j], o, 32);
      }
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<float2*>(part + (size_t)(c0 + j) * 2) = make_float2(s1[j], s2[j]);
    }
  }
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    int img, r, xq;
    quad_of(b, img, r, xq);
    if (a.vout) {
      f32x4 d[6];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float l = __shfl_up(y[b][3][j], 1, 32), rr = __shfl_down(y[b][0][j], 1, 32);
        if (b > 0) {
          const float t = __shfl(y[b > 0 ? b - 1 : 0][3][j], 31, 32);
          l = li == 0 ? t : l;
        }
        if (b + 1 < NT) {
          const float t = __shfl(y[b + 1 < NT ? b + 1 : b][0][j], 0, 32);
          rr = li == 31 ? t : rr;
        }
        d[0][j] = xq > 0 ? l : 0.f;
        d[5][j] = xq + 1 < a.Wq ? rr : 0.f;
      }
      d[1] = y[b][0]; d[2] = y[b][1]; d[3] = y[b][2]; d[4] = y[b][3];
      f32x4 vv[6];
      wchain_input_transform(d, vv);
      {
        float* o = a.vout + ((size_t)(c0 >> 3) * 2 + lh) * (a.plane_bytes >> 2) + ((size_t)(img * (a.H + 2) + r + 1) * a.Wq + xq) * 4;
        const size_t pstride = (size_t)a.cg_out * 2 * (a.plane_bytes >> 2);
#pragma unroll
        for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride) = vv[q];
        // the padding rows above the first and below the last image row belong to the tiles that hold those rows: the planes need
        // no separate clear (they may be uninitialised memory)
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if (r == 0) {
#pragma unroll
          for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride - (size_t)a.Wq * 4) = z4;
        }
        if (r == a.H - 1) {
#pragma unroll
          for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride + (size_t)a.Wq * 4) = z4;
        }
      }
    }
    if (a.out) {
      float* o = a.out + (size_t)img * a.out_img + (size_t)r * a.out_row + (size_t)(4 * xq) * a.out_ps + a.out_co + c0;
#pragma unroll
      for (int px = 0; px < 4; ++px) *reinterpret_cast<f32x4*>(o + (size_t)px * a.out_ps) = y[b][px];
    }
  }
}

// The prologue needs a dozen kernel arguments; left alone the compiler fetches them from the kernarg segment one group at a time, each
// group right before its first use, behind a wait of its own -- four or five dependent round trips (~1.5k cycles of a 35k-cycle kernel)
// before the first operand load is issued.  Naming them all as inputs of one empty asm statement makes it fetch them together.
#define WCHAIN_FETCH_ARGS(a)                                                                                                               \
  asm volatile("" ::"s"(a.vin), "s"(a.w), "s"(a.qtiles), "s"(a.ctiles), "s"(a.wq_log2), "s"(a.H), "s"(a.Wq), "s"(a.plane_bytes), "s"(a.cout_pad), \
               "s"(a.cg_in), "s"(a.vin_bytes), "s"(a.w_bytes))

template <int NA, int NB, int KS, int CT>
__global__ __launch_bounds__(64 * 6 * KS * CT) void conv_wchain_kernel(WChainArgs a) {
  constexpr int NW = 6 * KS * CT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  WCHAIN_FETCH_ARGS(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = w % 6, ks = (w / 6) % KS, ct = w / (6 * KS);
  const int li = lane & 31, lh = lane >> 5;

  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.vin), 0, a.vin_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned row16 = (unsigned)a.Wq * 16u;
  const int cg_per = a.cg_in / KS, cg0 = ks * cg_per, cg_last = cg0 + cg_per - 1;
  f32x4* J = reinterpret_cast<f32x4*>(smem);      // [wave][b][g][lane] x 4 registers
  const float lo = a.act == PN_ACT_RELU ? 0.f : -__builtin_inff();
  // persistent blocks: a block takes tiles bid, bid + grid, ...  The waves that finish no tile (the join's virtual waves are the first
  // 4 CT of the block) are already in the next tile's K loop while the others store the previous one.
  for (int bid = blockIdx.x; bid < a.qtiles * a.ctiles; bid += gridDim.x) {
  // tile -> (quad tile, column tile): the column tiles of a quad tile and runs of adjacent quad tiles share an XCD (its L2 then holds
  // the tile's planes once and the layer's weights once)
  int qt, ctile;
  {
    if ((a.qtiles & 7) == 0) {
      const int xcd = bid & 7, slot = bid >> 3;
      qt = xcd * (a.qtiles >> 3) + slot / a.ctiles;
      ctile = slot - (slot / a.ctiles) * a.ctiles;
    } else {
      qt = bid / a.ctiles;
      ctile = bid - qt * a.ctiles;
    }
  }
  WC_STAMP(0);
  const int n0 = (ctile * CT + ct) * 32 * NA;      // this wave's first output column
  const int q0 = qt * 32 * NB;

  // this lane's quads: byte offset of (image, row r - 1 + 1 = padded row r, xq) inside a plane, plus the lane half's plane
  const int row0 = q0 >> a.wq_log2, img0 = row0 / a.H, r0 = row0 - img0 * a.H;      // block-uniform (scalar) division
  unsigned voff[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int img, r, xq;
    wchain_coords(img0, r0, a.H, a.wq_log2, 32 * b + li, img, r, xq);
    voff[b] = (unsigned)(((img * (a.H + 2) + r) * a.Wq + xq) * 16) + (unsigned)lh * a.plane_bytes;
  }
  unsigned uoff[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) uoff[i] = (unsigned)(((size_t)lh * a.cout_pad + n0 + 32 * i + li) * 16);

  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.f;

  // operand ring: slot = kernel row kh; while (cg, kh) multiplies, (cg, kh + 1), (cg, kh + 2) and (cg + 1, kh) .. are in flight
  f32x4 u[3][NA], v[3][NB];
  auto load_step = [&](int cg, int kh) __attribute__((always_inline)) {
    const unsigned so_u = (unsigned)((((cg >> 2) * 3 + kh) * 6 + p) * 8 + (cg & 3) * 2) * cp16;
    const unsigned so_v = (unsigned)((p * a.cg_in + cg) * 2) * a.plane_bytes + (unsigned)kh * row16;
#pragma unroll
    for (int i = 0; i < NA; ++i) u[kh][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, uoff[i], so_u, 0));
#pragma unroll
    for (int b = 0; b < NB; ++b) v[kh][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff[b], so_v, 0));
  };
  // the issue order is pinned (sched_barrier): prologue and loop body then request the slots in the same order, and the compiler's
  // counted waits stay exact across the back edge -- vmcnt(2 (NA + NB)) before each step instead of vmcnt(0) at the loop head
  load_step(cg0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_step(cg0, 1);
  __builtin_amdgcn_sched_barrier(0);
  load_step(cg0, 2);
  __builtin_amdgcn_sched_barrier(0);
  WC_STAMP(1);
  for (int cg = cg0; cg <= cg_last; ++cg) {
    const int nx = cg < cg_last ? cg + 1 : cg_last;     // the last group's refill re-reads itself (stays inside the buffers)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[kh][i][j], v[kh][b][j], acc[i][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_step(nx, kh);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  WC_STAMP(2);
  // ---- join + epilogue, one pass per column sub-tile i of the waves
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    if (i > 0 || bid != (int)blockIdx.x) __syncthreads();      // the previous pass / tile has been read by its virtual waves
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = {acc[i][b][4 * g], acc[i][b][4 * g + 1], acc[i][b][4 * g + 2], acc[i][b][4 * g + 3]};
        J[((w * NB + b) * 4 + g) * 64 + lane] = t;
      }
    __syncthreads();
    for (int vw = w; vw < CT * 4; vw += NW) {
      const int ect = vw >> 2, g = vw & 3;
      const int c0 = ((ctile * CT + ect) * NA + i) * 32 + 8 * g + 4 * lh;     // four consecutive output channels
      wchain_finish<NB>(
          a, c0, li, lh, lo,
          [&](int q, int b) {
            f32x4 m = J[((((ect * KS) * 6 + q) * NB + b) * 4 + g) * 64 + lane];
#pragma unroll
            for (int k = 1; k < KS; ++k) m += J[((((ect * KS + k) * 6 + q) * NB + b) * 4 + g) * 64 + lane];
            return m;
          },
          [&](int b, int& img, int& r, int& xq) { wchain_coords(img0, r0, a.H, a.wq_log2, 32 * b + li, img, r, xq); });
    }
  }
  WC_STAMP(3);
  }
}

// ---- the same chain step with F(2, 3) along the map HEIGHT on top of F(4, 3) along the width: an OCTET (two rows x four pixels) from
// 4 x 6 = 24 products per input / output channel pair -- 3 per output against 4.5 -- with the planes format unchanged.  The height
// transform is linear in the rows and the planes hold every row already width-transformed, so the CONSUMER forms it while loading:
// a lane reads its octet's four input rows 2 t - 1 .. 2 t + 2 of position p (four fragments d0..d3, the padding rows give the zeros)
// and multiplies b0 = d0 - d2, b1 = d1 + d2, b2 = d2 - d1, b3 = d1 - d3 (four vector subtractions per 16 MFMAs) against the four
// height positions s of the weights U[s][p] = Gh g Gw^T; no sum over kernel rows is left, K = Cin.  A wave = one width position p of
// 32 columns x 32 octets with four accumulators (s); before the join it folds them to the two output rows (r0 = m0 + m1 + m2,
// r1 = m1 - m2 - m3), the tail is wchain_finish per output row.  Block = 6 KS CT QT waves: K halves, column tiles, and QT octet tiles
// that follow each other along a row pair (QT 32 octets = whole row pairs).
template <int KS, int CT, int QT>
__device__ __forceinline__ void wchain2_body(const WChainArgs& a, const int bid0, const int bid_step) {
  constexpr int NW = 6 * KS * CT * QT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  WCHAIN_FETCH_ARGS(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = w % 6, ks = (w / 6) % KS, ct = (w / (6 * KS)) % CT, qw = w / (6 * KS * CT);
  const int li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.vin), 0, a.vin_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned row16 = (unsigned)a.Wq * 16u;
  const int cg_per = a.cg_in / KS, cg0 = ks * cg_per, cg_last = cg0 + cg_per - 1;     // cg_per is even (Cin a multiple of 32)
  const int Hh = a.H >> 1;
  f32x4* J = reinterpret_cast<f32x4*>(smem);
  const float lo = a.act == PN_ACT_RELU ? 0.f : -__builtin_inff();
  // persistent blocks (see conv_wchain_kernel): tiles bid0, bid0 + step, ...
  for (int bid = bid0; bid < a.qtiles * a.ctiles; bid += bid_step) {
  int qt, ctile;       // the tile's octet-tile group / column-tile group (a.qtiles / a.ctiles of them)
  {
    if ((a.qtiles & 7) == 0) {
      const int xcd = bid & 7, slot = bid >> 3;
      qt = xcd * (a.qtiles >> 3) + slot / a.ctiles;
      ctile = slot - (slot / a.ctiles) * a.ctiles;
    } else {
      qt = bid / a.ctiles;
      ctile = bid - qt * a.ctiles;
    }
    qt += a.qt_begin;
  }
  WC_STAMP(0);
  const int n0 = (ctile * CT + ct) * 32;
  const int o0 = qt * 32 * QT;          // first octet of the block
  const int row0 = o0 >> a.wq_log2, img0 = row0 / Hh, t0 = row0 - img0 * Hh;      // block-uniform (scalar) division; rows = row pairs
  unsigned voff;
  {
    int img, t, xq;
    wchain_coords(img0, t0, Hh, a.wq_log2, 32 * qw + li, img, t, xq);
    // padded row index of image row 2 t - 1 is 2 t
    voff = (unsigned)(((img * (a.H + 2) + 2 * t) * a.Wq + xq) * 16) + (unsigned)lh * a.plane_bytes;
  }
  // range-stratified layer: the tile's rows (row pairs t0 ..) lie in ONE stratum (strata_rows is a multiple of the tile's rows)
  const unsigned uoff = (unsigned)(((size_t)lh * a.cout_pad + n0 + li) * 16) + (a.strata_rows > 0 ? (unsigned)((2 * t0) / a.strata_rows) * a.w_stratum_bytes : 0u);

  f32x16 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  f32x4 d[2][4], u[2][4];
  bool first = false;      // (diagnostic builds only)
  auto load_step = [&](int cg, int slot) __attribute__((always_inline)) {
    const unsigned so_v = (unsigned)((p * a.cg_in + cg) * 2) * a.plane_bytes;
    const unsigned so_u = (unsigned)((((cg >> 2) * 4) * 6 + p) * 8 + (cg & 3) * 2) * cp16;
    if (!(PN_WCHAIN_EXP & 2) || first) {
#pragma unroll
      for (int r = 0; r < 4; ++r) d[slot][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff, so_v + (unsigned)r * row16, 0));
    }
    if (!(PN_WCHAIN_EXP & 4) || first) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) u[slot][s2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, uoff, so_u + (unsigned)(s2 * 48) * cp16, 0));
    }
  };
  first = true;
  load_step(cg0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_step(cg0 + 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  first = false;
  WC_STAMP(1);
  for (int cg = cg0; cg <= cg_last; cg += 2) {
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int nx = cg + 2 + slot <= cg_last ? cg + 2 + slot : cg_last;      // the last refills re-read a live group (stay inside the buffers)
      f32x4 b[4];
      if constexpr (PN_WCHAIN_EXP & 1) {
        b[0] = d[slot][0]; b[1] = d[slot][1]; b[2] = d[slot][2]; b[3] = d[slot][3];
      } else {
        // (packed: eight v_pk_add_f32 instead of sixteen scalar ones -- the VALU work of this loop comes straight out of the MFMA time)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          f32x2 x0 = {d[slot][0][e], d[slot][0][e + 1]}, x1 = {d[slot][1][e], d[slot][1][e + 1]}, x2 = {d[slot][2][e], d[slot][2][e + 1]},
                x3 = {d[slot][3][e], d[slot][3][e + 1]};
          f32x2 r0, r1, r2, r3;
          asm volatile("v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                       "v_pk_add_f32 %1, %5, %6\n\t"
                       "v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                       "v_pk_add_f32 %3, %5, %7 neg_lo:[0,1] neg_hi:[0,1]"
                       : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                       : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
          b[0][e] = r0[0]; b[0][e + 1] = r0[1];
          b[1][e] = r1[0]; b[1][e + 1] = r1[1];
          b[2][e] = r2[0]; b[2][e + 1] = r2[1];
          b[3][e] = r3[0]; b[3][e + 1] = r3[1];
        }
      }
      if constexpr (PN_WCHAIN_EXP & 8) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[slot][s2][j], b[s2][j], acc[s2], 0, 0, 0);
      if constexpr (PN_WCHAIN_EXP & 8) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      load_step(nx, slot);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  WC_STAMP(2);
  // fold the four height positions to the two output rows, leave them in LDS: [wave][row][g][lane] x 4 registers
  if (bid != bid0) __syncthreads();      // the previous tile has been read by its virtual waves
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 r0, r1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 4 * g + k;
      r0[k] = (acc[0][r] + acc[1][r]) + acc[2][r];
      r1[k] = (acc[1][r] - acc[2][r]) - acc[3][r];
    }
    J[((w * 2 + 0) * 4 + g) * 64 + lane] = r0;
    J[((w * 2 + 1) * 4 + g) * 64 + lane] = r1;
  }
  __syncthreads();
  for (int vw = w; vw < CT * 8; vw += NW) {
    const int ect = vw >> 3, g = (vw >> 1) & 3, row = vw & 1;
    const int c0 = (ctile * CT + ect) * 32 + 8 * g + 4 * lh;
    wchain_finish<QT>(
        a, c0, li, lh, lo,
        [&](int q, int b) {
          f32x4 m = J[(((q + 6 * (KS * (ect + CT * b))) * 2 + row) * 4 + g) * 64 + lane];
#pragma unroll
          for (int k = 1; k < KS; ++k) m += J[(((q + 6 * (k + KS * (ect + CT * b))) * 2 + row) * 4 + g) * 64 + lane];
          return m;
        },
        [&](int b, int& img, int& r, int& xq) {
          int t;
          wchain_coords(img0, t0, Hh, a.wq_log2, 32 * b + li, img, t, xq);
          r = 2 * t + row;
        },
        a.st_part ? a.st_part + ((size_t)qt * 2 + row) * a.Cout * 2 : nullptr, a.strata_rows > 0 ? ((2 * t0) / a.strata_rows) * a.Cout : 0);
  }
  WC_STAMP(3);
  }
}

template <int KS, int CT, int QT>
__global__ __launch_bounds__(64 * 6 * KS * CT * QT) void conv_wchain2_kernel(WChainArgs a) {
  wchain2_body<KS, CT, QT>(a, blockIdx.x, gridDim.x);
}

// ---- F(4, 3) along the map HEIGHT as well (r5): a HEXADECET (four rows x four pixels) from 6 x 6 = 36 products per channel pair -- 2.25 per
// output against 3 -- with the planes format unchanged.  A wave that kept all six height positions would need 96 accumulator registers (two
// waves per SIMD: eight per CU where six width positions want twelve), so the height positions are split over TWO waves: wave (p, sh) holds
// positions 3 sh .. 3 sh + 2 of width position p (three accumulators), reads the five input rows its positions depend on (sh = 0: rows
// 4 t - 1 .. 4 t + 3, sh = 1: rows 4 t .. 4 t + 4 of position p), forms B^T d on the fly and folds its three positions to partial values of the
// four output rows (A^T is linear: the two halves add).  Join: the sh = 0 waves write the 6 x 4 row tiles, the sh = 1 waves add theirs
// (a fixed order), then wchain_finish per row.  Block = 12 waves = one tile of 32 hexadecets (whole row groups) x 32 columns, K = Cin
// unsplit; the join is 96 KB of LDS, which is why a 64-quad row (two tiles that the finish must see together) does not fit: frames of
// 32 quads and less.  Half the blocks of the 12-wave F(2,3) x F(4,3) form on the same map: chosen where other frames fill the chip.
template <int SH>
__device__ __forceinline__ void wchain3_kloop(const WChainArgs& a, const __amdgpu_buffer_rsrc_t rsrc_v, const __amdgpu_buffer_rsrc_t rsrc_w, const int p,
                                              const unsigned voff, const unsigned uoff, f32x16 (&acc)[3]) {
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned row16 = (unsigned)a.Wq * 16u;
  f32x4 d[2][5], u[2][3];
  bool first = false;      // (diagnostic builds only: PN_WCHAIN_EXP)
  auto load_step = [&](int cg, int slot) __attribute__((always_inline)) {
    const unsigned so_v = (unsigned)((p * a.cg_in + cg) * 2) * a.plane_bytes + (unsigned)SH * row16;
    const unsigned so_u = (unsigned)((((cg >> 2) * 6 + 3 * SH) * 6 + p) * 8 + (cg & 3) * 2) * cp16;
    if (!(PN_WCHAIN_EXP & 2) || first) {
#pragma unroll
      for (int r = 0; r < 5; ++r) d[slot][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff, so_v + (unsigned)r * row16, 0));
    }
    if (!(PN_WCHAIN_EXP & 4) || first) {
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) u[slot][s2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, uoff, so_u + (unsigned)(s2 * 48) * cp16, 0));
    }
  };
  const int cg_last = a.cg_in - 1;
  first = true;
  load_step(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_step(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  first = false;
  for (int cg = 0; cg <= cg_last; cg += 2) {
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int nx = cg + 2 + slot <= cg_last ? cg + 2 + slot : cg_last;      // the last refills re-read a live group (stay inside the buffers)
      f32x4 b[3];
      const f32x4 c4 = {4.f, 4.f, 4.f, 4.f}, m4 = {-4.f, -4.f, -4.f, -4.f}, m5 = {-5.f, -5.f, -5.f, -5.f}, c2 = {2.f, 2.f, 2.f, 2.f}, m2 = {-2.f, -2.f, -2.f, -2.f};
      if constexpr (PN_WCHAIN_EXP & 1) {
        b[0] = d[slot][0]; b[1] = d[slot][1]; b[2] = d[slot][2];
      } else if constexpr (SH == 0) {      // d[.][r] = input row 4 t - 1 + r: positions 0 .. 2 of B^T (wino4_input_transform4's tree)
        const f32x4 e = __builtin_elementwise_fma(m4, d[slot][2], d[slot][4]), o = __builtin_elementwise_fma(m4, d[slot][1], d[slot][3]);
        b[0] = __builtin_elementwise_fma(c4, d[slot][0], __builtin_elementwise_fma(m5, d[slot][2], d[slot][4]));
        b[1] = e + o;
        b[2] = e - o;
      } else {                      // d[.][r] = input row 4 t + r (row 1 + r of the six): positions 3 .. 5
        const f32x4 f = d[slot][3] - d[slot][1], t = d[slot][2] - d[slot][0];
        b[0] = __builtin_elementwise_fma(c2, t, f);
        b[1] = __builtin_elementwise_fma(m2, t, f);
        b[2] = __builtin_elementwise_fma(c4, d[slot][0], __builtin_elementwise_fma(m5, d[slot][2], d[slot][4]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          if (PN_WC3_EXP & 1) acc[s2][0] += u[slot][s2][j] * b[s2][j];      // (keeps the operands alive)
          else acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[slot][s2][j], b[s2][j], acc[s2], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      load_step(nx, slot);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// QT = 2: rows of 64 quads (the 256-pixel rows).  The join holds ONE 32-quad tile, so the two halves of a row group are computed one after
// the other; what the finish of one half needs from the other is one pixel per row and channel on either side of the cut -- the next layer's
// input transform of quad 31 takes the first pixel of quad 32 and vice versa.  Half 0 therefore leaves quad 31's four pixels and quad 30's
// last one in a 2.5 KB carry instead of storing quad 31's planes; half 1 takes its left neighbour from there and its first lane completes
// quad 31.  Same values as an undivided row.
template <int QT>
__global__ __launch_bounds__(64 * 12) void conv_wchain3_kernel(WChainArgs a) {
  constexpr int NW = 12;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  WCHAIN_FETCH_ARGS(a);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = w % 6, sh = w / 6;
  const int li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.vin), 0, a.vin_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int Hq = a.H >> 2;
  f32x4* J = reinterpret_cast<f32x4*>(smem);      // [p 6][row 4][g 4][lane 64]
  f32x4* carry = J + 6 * 4 * 4 * 64;              // QT = 2: [row 4][g 4][lh 2][5]: quad 31's pixels 0 .. 3 and quad 30's pixel 3
  const float lo = a.act == PN_ACT_RELU ? 0.f : -__builtin_inff();
  const int bid = blockIdx.x;
  int qt, ctile;
  if ((a.qtiles & 7) == 0) {      // an XCD takes a contiguous run of hexadecet tiles (neighbouring row groups share input rows)
    const int xcd = bid & 7, slot = bid >> 3;
    qt = xcd * (a.qtiles >> 3) + slot / a.ctiles;
    ctile = slot - (slot / a.ctiles) * a.ctiles;
  } else {
    qt = bid / a.ctiles;
    ctile = bid - qt * a.ctiles;
  }
  const int n0 = ctile * 32;
  const int o0 = qt * 32 * QT;          // first hexadecet of the block
  const int rg0 = o0 >> a.wq_log2, img0 = rg0 / Hq, t0 = rg0 - img0 * Hq;      // block-uniform; rows = row groups
  const unsigned uoff = (unsigned)(((size_t)lh * a.cout_pad + n0 + li) * 16);
#pragma unroll 1
  for (int sub = 0; sub < QT; ++sub) {
    int img, t, xq;
    wchain_coords(img0, t0, Hq, a.wq_log2, 32 * sub + li, img, t, xq);
    // padded row index of image row 4 t - 1 is 4 t
    const unsigned voff = (unsigned)(((img * (a.H + 2) + 4 * t) * a.Wq + xq) * 16) + (unsigned)lh * a.plane_bytes;
    f32x16 acc[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    if (sh == 0) wchain3_kloop<0>(a, rsrc_v, rsrc_w, p, voff, uoff, acc);
    else wchain3_kloop<1>(a, rsrc_v, rsrc_w, p, voff, uoff, acc);
    if (PN_WC3_EXP & 2) {      // no join: the accumulators only have to stay alive
      if (acc[0][0] + acc[1][3] + acc[2][7] == 1234.5f) J[lane] = f32x4{acc[0][1], acc[1][1], acc[2][1], 0.f};
    }
    if (sub > 0) __syncthreads();      // the previous half's join has been read
    // this half's share of the four output rows (A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1])
    if (sh == 0 && !(PN_WC3_EXP & 2)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 r0, r1, r2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 4 * g + k;
          r0[k] = (acc[0][r] + acc[1][r]) + acc[2][r];
          r1[k] = acc[1][r] - acc[2][r];
          r2[k] = acc[1][r] + acc[2][r];
        }
        J[((p * 4 + 0) * 4 + g) * 64 + lane] = r0;
        J[((p * 4 + 1) * 4 + g) * 64 + lane] = r1;
        J[((p * 4 + 2) * 4 + g) * 64 + lane] = r2;
        J[((p * 4 + 3) * 4 + g) * 64 + lane] = r1;
      }
    }
    __syncthreads();
    if (sh == 1 && !(PN_WC3_EXP & 2)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 r0, r1, r2, r3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 4 * g + k;
          const float s34 = acc[0][r] + acc[1][r], d34 = acc[0][r] - acc[1][r];
          r0[k] = s34;
          r1[k] = 2.f * d34;
          r2[k] = 4.f * s34;
          r3[k] = 8.f * d34 + acc[2][r];
        }
        J[((p * 4 + 0) * 4 + g) * 64 + lane] += r0;
        J[((p * 4 + 1) * 4 + g) * 64 + lane] += r1;
        J[((p * 4 + 2) * 4 + g) * 64 + lane] += r2;
        J[((p * 4 + 3) * 4 + g) * 64 + lane] += r3;
      }
    }
    __syncthreads();
    for (int vw = w; vw < ((PN_WC3_EXP & 4) ? 0 : 16); vw += NW) {
      const int g = vw >> 2, row = vw & 3;
      const int c0 = ctile * 32 + 8 * g + 4 * lh;
      if constexpr (QT == 1) {
        wchain_finish<1>(
            a, c0, li, lh, lo, [&](int q, int) { return J[((q * 4 + row) * 4 + g) * 64 + lane]; },
            [&](int, int& im, int& r, int& x) { im = img; r = 4 * t + row; x = xq; });
      } else {
        // wchain_finish for one half of a 64-quad row (see above); sub = 0: quads 0 .. 31, sub = 1: quads 32 .. 63
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, shf = {0.f, 0.f, 0.f, 0.f};
        if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + c0);
        if (a.shift) shf = *reinterpret_cast<const f32x4*>(a.shift + c0);
        f32x4 m[6], y[4];
#pragma unroll
        for (int q = 0; q < 6; ++q) m[q] = J[((q * 4 + row) * 4 + g) * 64 + lane];
        {
          const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
          const f32x4 lo4 = {lo, lo, lo, lo};
          y[0] = __builtin_elementwise_max(__builtin_elementwise_fma((m[0] + s12) + s34, sc, shf), lo4);
          y[1] = __builtin_elementwise_max(__builtin_elementwise_fma(d12 + 2.f * d34, sc, shf), lo4);
          y[2] = __builtin_elementwise_max(__builtin_elementwise_fma(s12 + 4.f * s34, sc, shf), lo4);
          y[3] = __builtin_elementwise_max(__builtin_elementwise_fma((d12 + 8.f * d34) + m[5], sc, shf), lo4);
        }
        const int r = 4 * t + row;
        if (a.out) {
          float* o = a.out + (size_t)img * a.out_img + (size_t)r * a.out_row + (size_t)(4 * xq) * a.out_ps + a.out_co + c0;
#pragma unroll
          for (int px = 0; px < 4; ++px) *reinterpret_cast<f32x4*>(o + (size_t)px * a.out_ps) = y[px];
        }
        if (a.vout) {
          f32x4* cr = carry + ((row * 4 + g) * 2 + lh) * 5;
          f32x4 l, rr;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            l[j] = __shfl_up(y[3][j], 1, 32);
            rr[j] = __shfl_down(y[0][j], 1, 32);
          }
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          auto store_planes = [&](const f32x4 (&d)[6], int xs) {
            f32x4 vv[6];
            wchain_input_transform(d, vv);
            float* o = a.vout + ((size_t)(c0 >> 3) * 2 + lh) * (a.plane_bytes >> 2) + ((size_t)(img * (a.H + 2) + r + 1) * a.Wq + xs) * 4;
            const size_t pstride = (size_t)a.cg_out * 2 * (a.plane_bytes >> 2);
#pragma unroll
            for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride) = vv[q];
            if (r == 0) {
#pragma unroll
              for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride - (size_t)a.Wq * 4) = z4;
            }
            if (r == a.H - 1) {
#pragma unroll
              for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride + (size_t)a.Wq * 4) = z4;
            }
          };
          if (sub == 0) {
            if (li == 31) {      // quad 31 waits for quad 32's first pixel: its pixels and its left neighbour into the carry
              cr[0] = y[0]; cr[1] = y[1]; cr[2] = y[2]; cr[3] = y[3]; cr[4] = l;
            } else {
              f32x4 d[6];
              d[0] = xq > 0 ? l : z4;
              d[1] = y[0]; d[2] = y[1]; d[3] = y[2]; d[4] = y[3];
              d[5] = rr;
              store_planes(d, xq);
            }
          } else {
            if (li == 0) {       // quad 31 (carry) with this lane's first pixel as its right neighbour, and quad 31's last pixel as this lane's left
              f32x4 d[6];
              d[0] = cr[4]; d[1] = cr[0]; d[2] = cr[1]; d[3] = cr[2]; d[4] = cr[3];
              d[5] = y[0];
              store_planes(d, xq - 1);
              l = cr[3];
            }
            f32x4 d[6];
            d[0] = l;
            d[1] = y[0]; d[2] = y[1]; d[3] = y[2]; d[4] = y[3];
            d[5] = xq + 1 < a.Wq ? rr : z4;
            store_planes(d, xq);
          }
        }
      }
    }
  }
}

// torch (Cout, Cin, 3, 3) -> [chunk][s 6][p 6][k4 8][cout_pad][4] = G g G^T in double, rounded once (G: F(4, 3), conv_wino4.hip's rows, for both axes)
__global__ void pack_wino44_weight_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3; r >>= 2;
    const int n = (int)(r % cout_pad); r /= cout_pad;
    const int k4 = r & 7; r >>= 3;
    const int q = (int)(r % 6); r /= 6;
    const int s = (int)(r % 6);
    const int chunk = (int)(r / 6);
    const int c = chunk * 32 + k4 * 4 + k1;
    double v = 0.0;
    if (n < cout && c < cin) {
      const float* g = w + ((size_t)n * cin + c) * 9;
      auto fold = [](int pos, double g0, double g1, double g2) {
        return pos == 0 ? g0 / 4.0 : pos == 1 ? -(g0 + g1 + g2) / 6.0 : pos == 2 ? -(g0 - g1 + g2) / 6.0 : pos == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0
               : pos == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0 : g2;
      };
      double row[3];      // the kernel rows folded by G[s]
      for (int kw = 0; kw < 3; ++kw) row[kw] = fold(s, g[kw], g[3 + kw], g[6 + kw]);
      v = fold(q, row[0], row[1], row[2]);
    }
    packed[i] = (float)v;
  }
}

// several layers of the same map and form as ONE launch (the head's branch groups: three launches of 128 - 384 short blocks each left the
// chip half empty between them): block -> (job, tile), one tile per block.  The job's arguments are copied out of the kernarg segment with
// scalar loads (a run-time index into a by-value array would go through scratch).
constexpr int kChainMultiJobs = 4;
struct WChainMulti {
  int njobs;
  int first[kChainMultiJobs + 1];
  WChainArgs job[kChainMultiJobs];
};

template <int KS, int CT, int QT>
__global__ __launch_bounds__(64 * 6 * KS * CT * QT) void conv_wchain2_multi_kernel(WChainMulti by_value) {
  typedef const __attribute__((address_space(4))) int* kptr_t;
  const kptr_t base = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int njobs = base[offsetof(WChainMulti, njobs) / 4];
  const int t = blockIdx.x;
  int j = 0;
  for (int k = 1; k < njobs; ++k)
    if (t >= base[offsetof(WChainMulti, first) / 4 + k]) j = k;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = t - base[offsetof(WChainMulti, first) / 4 + j];
  constexpr int JW = sizeof(WChainArgs) / 4;
  union { WChainArgs a; int w[JW]; } u;
  const kptr_t src = base + offsetof(WChainMulti, job) / 4 + j * JW;
#pragma unroll
  for (int i = 0; i < JW; ++i) u.w[i] = src[i];
  wchain2_body<KS, CT, QT>(u.a, local, 1 << 30);
}

// torch (Cout, Cin, 3, 3) -> [chunk][s 4][p 6][k4 8][cout_pad][4] = Gh g Gw^T in double, rounded once (Gw: conv_wino4.hip's F(4, 3) rows,
// Gh: F(2, 3): g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2 over the kernel rows)
__global__ void pack_wino24_weight_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3; r >>= 2;
    const int n = (int)(r % cout_pad); r /= cout_pad;
    const int k4 = r & 7; r >>= 3;
    const int q = (int)(r % 6); r /= 6;
    const int s = (int)(r & 3);
    const int chunk = (int)(r >> 2);
    const int c = chunk * 32 + k4 * 4 + k1;
    double v = 0.0;
    if (n < cout && c < cin) {
      const float* g = w + ((size_t)n * cin + c) * 9;
      double row[3];      // the kernel rows folded by Gh[s]
      for (int kw = 0; kw < 3; ++kw) {
        const double g0 = g[kw], g1 = g[3 + kw], g2 = g[6 + kw];
        row[kw] = s == 0 ? g0 : s == 1 ? (g0 + g1 + g2) * 0.5 : s == 2 ? (g0 - g1 + g2) * 0.5 : g2;
      }
      const double g0 = row[0], g1 = row[1], g2 = row[2];
      v = q == 0 ? g0 / 4.0 : q == 1 ? -(g0 + g1 + g2) / 6.0 : q == 2 ? -(g0 - g1 + g2) / 6.0 : q == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0
          : q == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0 : g2;
    }
    packed[i] = (float)v;
  }
}

// NHWC map -> the six planes (head of a chain).  One thread = one (quad, four channels): lanes along quads, so the plane stores are
// whole 16-byte fragments in quad order.
__global__ __launch_bounds__(256) void wchain_v_from_nhwc_kernel(const float* __restrict__ in, float* __restrict__ vout, int B, int H, int W, int Wq, int C,
                                                                 long long in_img, long long in_row, int in_ps, int in_co, int total_quads,
                                                                 unsigned plane_floats) {
  const int c4n = C >> 2;
  for (long long it = blockIdx.x * (long long)blockDim.x + threadIdx.x; it < (long long)total_quads * c4n; it += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(it / total_quads);
    const int Q = (int)(it - (long long)c4 * total_quads);
    const int rowi = Q / Wq, xq = Q - rowi * Wq;
    const int img = rowi / H, r = rowi - img * H;
    const float* px = in + (size_t)img * in_img + (size_t)r * in_row + (size_t)(4 * xq) * in_ps + in_co + c4 * 4;
    f32x4 d[6], v[6];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    d[0] = xq > 0 ? *reinterpret_cast<const f32x4*>(px - in_ps) : z;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[1 + k] = *reinterpret_cast<const f32x4*>(px + (size_t)k * in_ps);
    d[5] = xq + 1 < Wq ? *reinterpret_cast<const f32x4*>(px + (size_t)4 * in_ps) : z;
    wchain_input_transform(d, v);
    float* o = vout + (size_t)c4 * plane_floats + ((size_t)(img * (H + 2) + r + 1) * Wq + xq) * 4;      // plane (cg = c4 >> 1, h = c4 & 1)
    const size_t pstride = (size_t)c4n * plane_floats;
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride) = v[q];
    if (r == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride - (size_t)Wq * 4) = z;
    }
    if (r == H - 1) {
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(o + q * pstride + (size_t)Wq * 4) = z;
    }
  }
}

// ---- the head's GroupNorm-family statistics from the partials of pn_conv2d_wino24_chain_head_f32 -> affine tables (A, B): y = x A + B, the
// form conv_small_n_multi_kernel applies while it loads (conv_mfma.hip's conv_stats_finalize_kernel for the tiled kernels' partials).
// A job = a slice of `cmid` channels of one launch's output: per-channel groups (GroupNorm(C, C): statistics over the sample's pixels) or
// one group per range stratum over all the slice's channels (RangeStratified's GroupNorm(strata, strata C)).  Fixed order, fp64.
constexpr int kHeadFinJobs = 8;
struct HeadFinJob {
  const float* part;      // [tile][row 2][cout_total][2]
  int cout_total, c_off, cmid;
  int strata;             // 0 / 1: per-channel groups; > 1: all-channel groups per stratum of the frame's rows
  const float* gamma;     // [strata or 1][cmid]
  const float* beta;
  float eps;
  float* tab;             // [B][strata or 1][cmid][2]
};
struct HeadFinArgs {
  int njobs, B, rows, tile_rows;      // frame rows per sample; rows a partial tile covers
  long long row_pixels;               // pixels of one frame row
  HeadFinJob job[kHeadFinJobs];
};

__global__ __launch_bounds__(256) void wchain_head_finalize_kernel(HeadFinArgs f) {
  __shared__ double red[2][4];
  const int S0 = 8;      // blocks per (job, sample): strata (<= 8) or 1
  const int j = blockIdx.x / (f.B * S0), rem = blockIdx.x - j * (f.B * S0), b = rem / S0, s = rem - b * S0;
  if (j >= f.njobs) return;
  const HeadFinJob& jb = f.job[j];
  const int S = jb.strata > 1 ? jb.strata : 1;
  if (s >= S) return;
  const int tid = threadIdx.x;
  const int tiles_b = f.rows / f.tile_rows;      // tiles per sample
  const float2* part = reinterpret_cast<const float2*>(jb.part);
  if (S == 1) {
    // per-channel groups, <= 64 channels per pass: 256 / 64 slices of the sample's entries per channel (all their loads in flight), joined
    // through LDS in a fixed order
    __shared__ double sl1[4][64], sl2[4][64];
    for (int c0 = 0; c0 < jb.cmid; c0 += 64) {
      const int c = c0 + (tid & 63), slice = tid >> 6;
      double a1 = 0.0, a2 = 0.0;
      if (c < jb.cmid) {
#pragma unroll 8
        for (int e = slice; e < tiles_b * 2; e += 4) {
          const float2 v = part[((size_t)b * tiles_b * 2 + e) * jb.cout_total + jb.c_off + c];
          a1 += (double)v.x;
          a2 += (double)v.y;
        }
      }
      __syncthreads();
      sl1[slice][tid & 63] = a1;
      sl2[slice][tid & 63] = a2;
      __syncthreads();
      if (tid < 64 && c < jb.cmid) {
        const double t1 = ((sl1[0][tid] + sl1[1][tid]) + sl1[2][tid]) + sl1[3][tid], t2 = ((sl2[0][tid] + sl2[1][tid]) + sl2[2][tid]) + sl2[3][tid];
        const double n = (double)f.rows * (double)f.row_pixels;
        const double mean = t1 / n;
        double var = t2 / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float rstd = (float)(1.0 / sqrt(var + (double)jb.eps));
        const float ga = jb.gamma ? jb.gamma[c] : 1.f, be = jb.beta ? jb.beta[c] : 0.f;
        const float A = ga * rstd;
        reinterpret_cast<float2*>(jb.tab)[(size_t)b * jb.cmid + c] = make_float2(A, be - (float)mean * A);
      }
    }
    return;
  }
  // stratum s of sample b: tiles [s tiles_s, (s + 1) tiles_s) of the sample, both rows, all cmid channels
  const int tiles_s = tiles_b / S;
  const int count = tiles_s * 2 * jb.cmid;
  double t1 = 0.0, t2 = 0.0;
  for (int e = tid; e < count; e += 256) {
    const int ent = e / jb.cmid, c = e - ent * jb.cmid;
    const float2 v = part[((size_t)(b * tiles_b + s * tiles_s) * 2 + ent) * jb.cout_total + jb.c_off + c];
    t1 += (double)v.x;
    t2 += (double)v.y;
  }
  t1 = pn::wave_sum(t1);
  t2 = pn::wave_sum(t2);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = t1;
    red[1][tid >> 6] = t2;
  }
  __syncthreads();
  const double u1 = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], u2 = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
  const double n = (double)(f.rows / S) * (double)f.row_pixels * jb.cmid;
  const double mean = u1 / n;
  double var = u2 / n - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float fmean = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)jb.eps));
  for (int c = tid; c < jb.cmid; c += 256) {
    const float ga = jb.gamma ? jb.gamma[s * jb.cmid + c] : 1.f, be = jb.beta ? jb.beta[s * jb.cmid + c] : 0.f;
    const float A = ga * rstd;
    reinterpret_cast<float2*>(jb.tab)[((size_t)b * S + s) * jb.cmid + c] = make_float2(A, be - fmean * A);
  }
}

// one 12-wave block per CU (a multiple of 8: tiles bid and bid + grid then share an XCD); PN_WCHAIN_GRID overrides (0: one block per tile)
static int chain_grid_limit() {
  static const int force = [] { const char* e = getenv("PN_WCHAIN_GRID"); return e ? atoi(e) : -1; }();
  if (force == 0) return 1 << 30;
  if (force > 0) return force;
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n / 8 * 8 : 256;
  }
  return cus[dev];
}

struct ChainForm { int na, nb, ks, ct; };

// the form a layer takes: wave tile 32 NA columns x 32 NB quads, K split KS ways, CT column tiles per block (always 12 waves)
// (h, w: the frame the kernel works in -- the stored map, or its transpose when desc->transpose_hw)
static inline int frame_h(const pn_conv_desc* d) { return d->transpose_hw ? d->in_w : d->in_h; }
static inline int frame_w(const pn_conv_desc* d) { return d->transpose_hw ? d->in_h : d->in_w; }

static bool chain_form(const pn_conv_desc* d, ChainForm& f) {
  const int fh = frame_h(d), fw = frame_w(d);
  if (fw % 4) return false;
  const int wq = fw / 4;
  if (wq & (wq - 1)) return false;
  const long long quads = (long long)d->batch * fh * wq;
  static const int force = [] { const char* e = getenv("PN_WCHAIN_FORM"); return e ? atoi(e) : 0; }();
  // candidates, widest register tile first; a form fits when its block tile is whole rows and the grid covers the chip about once
  const ChainForm cands[] = {{2, 2, 1, 2}, {1, 2, 2, 1}, {1, 1, 2, 1}};
  int idx = 0;
  for (const ChainForm& c : cands) {
    ++idx;
    if (force && force != idx) continue;
    const int tq = 32 * c.nb, tc = 32 * c.na * c.ct;
    if (tq % wq != 0 || quads % tq != 0 || d->cout % tc != 0 || (d->cin / 8) % c.ks != 0 || tq / wq > fh) continue;
    const long long blocks = quads / tq * (d->cout / tc);
    if (!force && blocks < 192 && idx < 3) continue;      // a narrower form fills the chip better
    f = c;
    return true;
  }
  return false;
}

template <int NA, int NB, int KS, int CT>
static void launch_chain(const WChainArgs& a, hipStream_t st, bool prof, const pn::ProfileSlot& ps) {
  constexpr int NW = 6 * KS * CT;
  constexpr size_t smem = (size_t)NW * NB * 4 * 64 * 16;
  static bool done[64] = {false};
  if (pn::first_use_on_device(done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wchain_kernel<NA, NB, KS, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const dim3 grid((unsigned)std::min(a.qtiles * a.ctiles, chain_grid_limit()));
  if (prof) hipExtLaunchKernelGGL((conv_wchain_kernel<NA, NB, KS, CT>), grid, dim3(64 * NW), smem, st, ps.start, ps.stop, 0, a);
  else hipLaunchKernelGGL((conv_wchain_kernel<NA, NB, KS, CT>), grid, dim3(64 * NW), smem, st, a);
}

struct Chain2Form { int ks, ct, qt; };

// F(2,3) x F(4,3) form: octet tiles of 32 QT octets = whole row pairs; 12 waves per block
static bool chain2_form(const pn_conv_desc* d, Chain2Form& f, bool head = false) {
  const int fh = frame_h(d), fw = frame_w(d);
  if (fw % 4 || fh % 2) return false;
  const int wq = fw / 4;
  if (wq & (wq - 1)) return false;
  const long long octs = (long long)d->batch * (fh / 2) * wq;
  // beside other frames (pn_conv_desc.frames_in_flight > 1) the two-tile form without the K split comes first: half as many blocks, each with
  // twice the K loop and no K join -- alone on the chip a 128 x 128 map would leave half the CUs idle with it, with three frames in flight the
  // other frames take them (PN_WCHAIN2_PREFER: 0 never, 1 on the hint, 2 always)
  static const int prefer = [] { const char* e = getenv("PN_WCHAIN2_PREFER"); return e ? atoi(e) : 1; }();
  // (not for the head's multi-job launch, whose jobs share one form, and only while the wide form still has 128 blocks)
  const bool wide_first = !head && (prefer == 2 || (prefer == 1 && d->frames_in_flight > 1)) && (octs / 64) * (d->cout / 32) >= 128;
  const Chain2Form cands_a[] = {{2, 1, 1}, {1, 1, 2}}, cands_b[] = {{1, 1, 2}, {2, 1, 1}};
  const Chain2Form* cands = wide_first ? cands_b : cands_a;
  for (int ci = 0; ci < 2; ++ci) {
    const Chain2Form& c = cands[ci];
    const int tq = 32 * c.qt, tc = 32 * c.ct;
    if (tq % wq != 0 || octs % tq != 0 || d->cout % tc != 0 || (d->cin / 8) % (2 * c.ks) != 0 || tq / wq > fh / 2) continue;
    f = c;
    return true;
  }
  return false;
}

template <int KS, int CT, int QT>
static void launch_chain2(const WChainArgs& a, hipStream_t st, bool prof, const pn::ProfileSlot& ps) {
  constexpr int NW = 6 * KS * CT * QT;
  constexpr size_t smem = (size_t)NW * 2 * 4 * 64 * 16;
  static bool done[64] = {false};
  if (pn::first_use_on_device(done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wchain2_kernel<KS, CT, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const dim3 grid((unsigned)std::min(a.qtiles * a.ctiles, chain_grid_limit()));
  if (prof) hipExtLaunchKernelGGL((conv_wchain2_kernel<KS, CT, QT>), grid, dim3(64 * NW), smem, st, ps.start, ps.stop, 0, a);
  else hipLaunchKernelGGL((conv_wchain2_kernel<KS, CT, QT>), grid, dim3(64 * NW), smem, st, a);
}

template <int KS, int CT, int QT>
static void launch_chain2_multi(const WChainMulti& m, hipStream_t st, bool prof, const pn::ProfileSlot& ps) {
  constexpr int NW = 6 * KS * CT * QT;
  constexpr size_t smem = (size_t)NW * 2 * 4 * 64 * 16;
  static bool done[64] = {false};
  if (pn::first_use_on_device(done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wchain2_multi_kernel<KS, CT, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const dim3 grid((unsigned)m.first[m.njobs]);
  if (prof) hipExtLaunchKernelGGL((conv_wchain2_multi_kernel<KS, CT, QT>), grid, dim3(64 * NW), smem, st, ps.start, ps.stop, 0, m);
  else hipLaunchKernelGGL((conv_wchain2_multi_kernel<KS, CT, QT>), grid, dim3(64 * NW), smem, st, m);
}

}  // namespace

extern "C" {

size_t pn_wino4_planes_floats(int batch, int h, int w, int c) {
  if (batch < 1 || h < 1 || w < 4 || w % 4 || c < 8 || c % 8) return 0;
  return (size_t)6 * c * batch * (h + 2) * (w / 4);
}

static int chain_basic_ok(const pn_conv_desc* d, bool allow_strata);

int pn_conv_wino4_chain_supported(const pn_conv_desc* d) {
  if (!chain_basic_ok(d, false)) return 0;
  ChainForm f;
  return chain_form(d, f) ? 1 : 0;
}

static int chain_basic_ok(const pn_conv_desc* d, bool allow_strata) {
  if (!d) return 0;
  if (!(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->groups == 1 && !d->deconv2x2 &&
        (d->range_strata <= 1 || allow_strata) && !d->accumulate && d->pad_h_end == 0 && d->pad_w_end == 0))
    return 0;
  if (d->batch < 1 || frame_h(d) < 1 || frame_w(d) < 4 || d->cin % 32 || d->cout % 32 || d->cin < 32) return 0;
  if (!(d->act == PN_ACT_NONE || d->act == PN_ACT_RELU)) return 0;
  if ((unsigned long long)pn_wino4_planes_floats(d->batch, frame_h(d), frame_w(d), std::max(d->cin, d->cout)) * 4ull >= (1ull << 32)) return 0;
  return 1;
}

int pn_wino4_planes_from_nhwc_f32(const float* in, int batch, int h, int w, int c, int in_pixel_stride, int in_channel_offset, int transpose_hw,
                                  float* planes, pn_stream_t stream) {
  // h, w: the STORED map; transposed, the planes are those of its transpose (w rows of h pixels: the Winograd axis is the map's H axis)
  const int stored_w = w;
  if (transpose_hw) std::swap(h, w);
  PN_REQUIRE(in && planes && pn_wino4_planes_floats(batch, h, w, c) > 0, "wino4_planes_from_nhwc: bad arguments");
  PN_REQUIRE(in_pixel_stride % 4 == 0 && in_channel_offset % 4 == 0 && in_pixel_stride >= in_channel_offset + c && ((uintptr_t)in & 15) == 0 &&
                 ((uintptr_t)planes & 15) == 0,
             "wino4_planes_from_nhwc: strides / offsets must be multiples of 4 floats, pointers 16-byte aligned");
  const int wq = w / 4;
  const long long quads = (long long)batch * h * wq;
  PN_REQUIRE(quads < (1ll << 30), "wino4_planes_from_nhwc: map too large");
  const unsigned plane_floats = (unsigned)((size_t)batch * (h + 2) * wq * 4);
  const long long items = quads * (c / 4);
  const long long in_img = (long long)h * w * in_pixel_stride;
  const long long in_row = transpose_hw ? in_pixel_stride : (long long)stored_w * in_pixel_stride;
  const int in_px = transpose_hw ? stored_w * in_pixel_stride : in_pixel_stride;
  hipLaunchKernelGGL(wchain_v_from_nhwc_kernel, dim3((unsigned)std::min<long long>(8192, (items + 255) / 256)), dim3(256), 0, pn::S(stream), in, planes, batch, h,
                     w, wq, c, in_img, in_row, in_px, in_channel_offset, (int)quads, plane_floats);
  return pn::check_launch("wchain_v_from_nhwc_kernel");
}

int pn_conv2d_wino4_chain_f32(const pn_conv_desc* d, const float* planes_in, const float* packed_w, const float* scale, const float* shift,
                              float* planes_out, float* out_nhwc, pn_stream_t stream) {
  PN_REQUIRE(d && planes_in && packed_w && (planes_out || out_nhwc), "conv_wino4_chain: null pointer");
  PN_REQUIRE(pn_conv_wino4_chain_supported(d), "conv_wino4_chain: layer shape not supported (3x3 / stride 1 / pad 1, cin and cout multiples of 32, "
                                               "row-aligned tiles)");
  PN_REQUIRE(((uintptr_t)planes_in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0 && ((uintptr_t)planes_out & 15) == 0 && ((uintptr_t)out_nhwc & 15) == 0 &&
                 ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
             "conv_wino4_chain: pointers must be 16-byte aligned");
  if (out_nhwc)
    PN_REQUIRE(d->out_pixel_stride >= d->out_channel_offset + d->cout && d->out_pixel_stride % 4 == 0 && d->out_channel_offset % 4 == 0 &&
                   (unsigned long long)d->batch * d->in_h * d->in_w * d->out_pixel_stride < (1ull << 40),
               "conv_wino4_chain: output channel slice must fit the pixel stride, in multiples of 4 floats");
  ChainForm f;
  chain_form(d, f);
  WChainArgs a{};
  a.vin = planes_in; a.w = packed_w; a.scale = scale; a.shift = shift; a.vout = planes_out; a.out = out_nhwc;
  a.B = d->batch; a.H = frame_h(d); a.W = frame_w(d); a.Wq = a.W / 4; a.Cin = d->cin; a.Cout = d->cout;
  a.out_co = d->out_channel_offset;
  a.out_img = (long long)d->in_h * d->in_w * d->out_pixel_stride;
  a.out_ps = d->transpose_hw ? d->in_w * d->out_pixel_stride : d->out_pixel_stride;
  a.out_row = d->transpose_hw ? d->out_pixel_stride : (long long)d->in_w * d->out_pixel_stride;
  a.wq_log2 = __builtin_ctz((unsigned)a.Wq);
  a.act = d->act;
  a.total_quads = d->batch * a.H * a.Wq;
  a.qtiles = a.total_quads / (32 * f.nb);
  a.ctiles = d->cout / (32 * f.na * f.ct);
  a.cg_in = d->cin / 8; a.cg_out = d->cout / 8;
  a.cout_pad = pn::cdiv(d->cout, 128) * 128;
  a.plane_bytes = (unsigned)((size_t)d->batch * (a.H + 2) * a.Wq * 16);
  a.vin_bytes = (unsigned)(pn_wino4_planes_floats(d->batch, a.H, a.W, d->cin) * 4);
  a.w_bytes = (unsigned)(pn_conv_wino4_packed_weight_floats(d->cout, d->cin) * 4);
#ifdef PN_WCHAIN_STAMP
  a.stamps = pn_wchain_stamp_buffer;
#endif
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  hipStream_t st = pn::S(stream);
  if (f.na == 2 && f.nb == 2 && f.ks == 1 && f.ct == 2) launch_chain<2, 2, 1, 2>(a, st, prof, ps);
  else if (f.na == 1 && f.nb == 2 && f.ks == 2 && f.ct == 1) launch_chain<1, 2, 2, 1>(a, st, prof, ps);
  else launch_chain<1, 1, 2, 1>(a, st, prof, ps);
  return pn::check_launch("conv_wchain_kernel");
}

size_t pn_conv_wino24_packed_weight_floats(int cout, int cin) {
  return (size_t)pn::cdiv(cin, 32) * 4 * 6 * 8 * (size_t)(pn::cdiv(cout, 128) * 128) * 4;
}

int pn_pack_conv_weight_wino24_f32(const float* w_oihw, int cout, int cin, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && cout >= 1 && cin >= 1, "pack_conv_weight_wino24: bad arguments");
  const size_t total = pn_conv_wino24_packed_weight_floats(cout, cin);
  hipLaunchKernelGGL(pack_wino24_weight_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w_oihw, cout, cin,
                     pn::cdiv(cout, 128) * 128, packed, total);
  return pn::check_launch("pack_wino24_weight_kernel");
}

// range_strata > 1 (head entry only): RangeStratified convolution on the TRANSPOSED map -- the frame's rows are range positions, every tile
// (whole row pairs) lies in one stratum and takes that stratum's weight set
static int chain2_ok(const pn_conv_desc* d, bool head, Chain2Form& f) {
  if (!chain_basic_ok(d, head) || !chain2_form(d, f, head)) return 0;
  if (d->range_strata > 1) {
    if (!d->transpose_hw || frame_h(d) % d->range_strata) return 0;
    const int rows = frame_h(d) / d->range_strata, tile_rows = 2 * (32 * f.qt) / (frame_w(d) / 4);
    if (rows % tile_rows) return 0;
  }
  return 1;
}

int pn_conv_wino24_chain_supported(const pn_conv_desc* d) {
  Chain2Form f;
  return chain2_ok(d, false, f);
}

static int chain2_fill(const pn_conv_desc* d, const float* planes_in, const float* packed_w24, const float* scale, const float* shift, float* planes_out,
                       float* out_nhwc, float* stat_partials, bool head, WChainArgs& a, Chain2Form& f) {
  PN_REQUIRE(d && planes_in && packed_w24 && (planes_out || out_nhwc), "conv_wino24_chain: null pointer");
  PN_REQUIRE(chain2_ok(d, head, f), "conv_wino24_chain: layer shape not supported (3x3 / stride 1 / pad 1, even height, cin and cout multiples of 32, "
                                    "row-pair-aligned tiles; range strata: transposed map, whole tiles per stratum)");
  PN_REQUIRE(((uintptr_t)planes_in & 15) == 0 && ((uintptr_t)packed_w24 & 15) == 0 && ((uintptr_t)planes_out & 15) == 0 && ((uintptr_t)out_nhwc & 15) == 0 &&
                 ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0 && ((uintptr_t)stat_partials & 7) == 0,
             "conv_wino24_chain: pointers must be 16-byte aligned");
  PN_REQUIRE(!stat_partials || d->act == PN_ACT_NONE, "conv_wino24_chain: statistics are those of the stored output (no activation)");
  if (out_nhwc)
    PN_REQUIRE(d->out_pixel_stride >= d->out_channel_offset + d->cout && d->out_pixel_stride % 4 == 0 && d->out_channel_offset % 4 == 0,
               "conv_wino24_chain: output channel slice must fit the pixel stride, in multiples of 4 floats");
  a = WChainArgs{};
  a.vin = planes_in; a.w = packed_w24; a.scale = scale; a.shift = shift; a.vout = planes_out; a.out = out_nhwc;
  a.B = d->batch; a.H = frame_h(d); a.W = frame_w(d); a.Wq = a.W / 4; a.Cin = d->cin; a.Cout = d->cout;
  a.out_co = d->out_channel_offset;
  a.out_img = (long long)d->in_h * d->in_w * d->out_pixel_stride;
  a.out_ps = d->transpose_hw ? d->in_w * d->out_pixel_stride : d->out_pixel_stride;
  a.out_row = d->transpose_hw ? d->out_pixel_stride : (long long)d->in_w * d->out_pixel_stride;
  a.wq_log2 = __builtin_ctz((unsigned)a.Wq);
  a.act = d->act;
  a.total_quads = d->batch * a.H * a.Wq;
  a.qtiles = (a.total_quads / 2) / (32 * f.qt);
  a.ctiles = d->cout / (32 * f.ct);
  a.cg_in = d->cin / 8; a.cg_out = d->cout / 8;
  a.cout_pad = pn::cdiv(d->cout, 128) * 128;
  a.plane_bytes = (unsigned)((size_t)d->batch * (a.H + 2) * a.Wq * 16);
  a.vin_bytes = (unsigned)(pn_wino4_planes_floats(d->batch, a.H, a.W, d->cin) * 4);
  const int sets = d->range_strata > 1 ? d->range_strata : 1;
  a.w_stratum_bytes = (unsigned)(pn_conv_wino24_packed_weight_floats(d->cout, d->cin) * 4);
  a.w_bytes = a.w_stratum_bytes * (unsigned)sets;
  a.strata_rows = sets > 1 ? a.H / sets : 0;
  a.st_part = stat_partials;
#ifdef PN_WCHAIN_STAMP
  a.stamps = pn_wchain_stamp_buffer;
#endif
  return PN_OK;
}

static int chain2_run(const pn_conv_desc* d, const float* planes_in, const float* packed_w24, const float* scale, const float* shift, float* planes_out,
                      float* out_nhwc, float* stat_partials, bool head, pn_stream_t stream) {
  WChainArgs a;
  Chain2Form f;
  const int rc = chain2_fill(d, planes_in, packed_w24, scale, shift, planes_out, out_nhwc, stat_partials, head, a, f);
  if (rc != PN_OK) return rc;
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  hipStream_t st = pn::S(stream);
  if (f.ks == 2) launch_chain2<2, 1, 1>(a, st, prof, ps);
  else launch_chain2<1, 1, 2>(a, st, prof, ps);
  return pn::check_launch("conv_wchain2_kernel");
}

int pn_conv2d_wino24_chain_f32(const pn_conv_desc* d, const float* planes_in, const float* packed_w24, const float* scale, const float* shift,
                               float* planes_out, float* out_nhwc, pn_stream_t stream) {
  return chain2_run(d, planes_in, packed_w24, scale, shift, planes_out, out_nhwc, nullptr, false, stream);
}

// ---- F(4,3) x F(4,3) (conv_wchain3_kernel): frames of at most 32 quads per row (the join holds one 32-hexadecet tile), height a multiple of 4
size_t pn_conv_wino44_packed_weight_floats(int cout, int cin) {
  return (size_t)pn::cdiv(cin, 32) * 6 * 6 * 8 * (size_t)(pn::cdiv(cout, 128) * 128) * 4;
}

int pn_pack_conv_weight_wino44_f32(const float* w_oihw, int cout, int cin, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && cout >= 1 && cin >= 1, "pack_conv_weight_wino44: bad arguments");
  const size_t total = pn_conv_wino44_packed_weight_floats(cout, cin);
  hipLaunchKernelGGL(pack_wino44_weight_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w_oihw, cout, cin,
                     pn::cdiv(cout, 128) * 128, packed, total);
  return pn::check_launch("pack_wino44_weight_kernel");
}

int pn_conv_wino44_chain_supported(const pn_conv_desc* d) {
  if (!chain_basic_ok(d, false) || d->range_strata > 1) return 0;
  const int fh = frame_h(d), fw = frame_w(d);
  if (fw % 4 || fh % 4) return 0;
  const int wq = fw / 4;
  if ((wq & (wq - 1)) || wq > 64) return 0;
  const int tq = wq == 64 ? 64 : 32;                                      // hexadecets per block: whole row groups
  const long long hexes = (long long)d->batch * (fh / 4) * wq;
  if (tq % wq || hexes % tq || tq / wq > fh / 4 || (d->cin / 8) % 2) return 0;      // tiles no taller than an image, cg pairs
  return 1;
}

int pn_conv2d_wino44_chain_f32(const pn_conv_desc* d, const float* planes_in, const float* packed_w44, const float* scale, const float* shift,
                               float* planes_out, float* out_nhwc, pn_stream_t stream) {
  PN_REQUIRE(d && planes_in && packed_w44 && (planes_out || out_nhwc), "conv_wino44_chain: null pointer");
  PN_REQUIRE(pn_conv_wino44_chain_supported(d), "conv_wino44_chain: layer shape not supported (3x3 / stride 1 / pad 1, height a multiple of 4, at most 32 quads "
                                                "per row, cin and cout multiples of 32)");
  PN_REQUIRE(((uintptr_t)planes_in & 15) == 0 && ((uintptr_t)packed_w44 & 15) == 0 && ((uintptr_t)planes_out & 15) == 0 && ((uintptr_t)out_nhwc & 15) == 0 &&
                 ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
             "conv_wino44_chain: pointers must be 16-byte aligned");
  if (out_nhwc)
    PN_REQUIRE(d->out_pixel_stride >= d->out_channel_offset + d->cout && d->out_pixel_stride % 4 == 0 && d->out_channel_offset % 4 == 0,
               "conv_wino44_chain: output channel slice must fit the pixel stride, in multiples of 4 floats");
  WChainArgs a{};
  a.vin = planes_in; a.w = packed_w44; a.scale = scale; a.shift = shift; a.vout = planes_out; a.out = out_nhwc;
  a.B = d->batch; a.H = frame_h(d); a.W = frame_w(d); a.Wq = a.W / 4; a.Cin = d->cin; a.Cout = d->cout;
  a.out_co = d->out_channel_offset;
  a.out_img = (long long)d->in_h * d->in_w * d->out_pixel_stride;
  a.out_ps = d->transpose_hw ? d->in_w * d->out_pixel_stride : d->out_pixel_stride;
  a.out_row = d->transpose_hw ? d->out_pixel_stride : (long long)d->in_w * d->out_pixel_stride;
  a.wq_log2 = __builtin_ctz((unsigned)a.Wq);
  a.act = d->act;
  a.total_quads = d->batch * a.H * a.Wq;
  const int qt2 = a.Wq == 64 ? 2 : 1;
  a.qtiles = (a.total_quads / 4) / (32 * qt2);
  a.ctiles = d->cout / 32;
  a.cg_in = d->cin / 8; a.cg_out = d->cout / 8;
  a.cout_pad = pn::cdiv(d->cout, 128) * 128;
  a.plane_bytes = (unsigned)((size_t)d->batch * (a.H + 2) * a.Wq * 16);
  a.vin_bytes = (unsigned)(pn_wino4_planes_floats(d->batch, a.H, a.W, d->cin) * 4);
  a.w_bytes = (unsigned)(pn_conv_wino44_packed_weight_floats(d->cout, d->cin) * 4);
#ifdef PN_WCHAIN_STAMP
  a.stamps = pn_wchain_stamp_buffer;
#endif
  constexpr size_t smem = (size_t)6 * 4 * 4 * 64 * 16 + (size_t)4 * 4 * 2 * 5 * 16;      // the join + the carry of the two-half form
  static bool done[64] = {false};
  if (pn::first_use_on_device(done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wchain3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wchain3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  const dim3 grid((unsigned)(a.qtiles * a.ctiles));
  if (qt2 == 2) {
    if (prof) hipExtLaunchKernelGGL(conv_wchain3_kernel<2>, grid, dim3(64 * 12), smem, pn::S(stream), ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL(conv_wchain3_kernel<2>, grid, dim3(64 * 12), smem, pn::S(stream), a);
  } else {
    if (prof) hipExtLaunchKernelGGL(conv_wchain3_kernel<1>, grid, dim3(64 * 12), smem, pn::S(stream), ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL(conv_wchain3_kernel<1>, grid, dim3(64 * 12), smem, pn::S(stream), a);
  }
  return pn::check_launch("conv_wchain3_kernel");
}

// floats of the statistics partials of pn_conv2d_wino24_chain_head_f32: [tile][row 2][cout][2]
size_t pn_conv_wino24_chain_stat_floats(const pn_conv_desc* d) {
  Chain2Form f;
  if (!chain2_ok(d, true, f)) return 0;
  const long long octs = (long long)d->batch * (frame_h(d) / 2) * (frame_w(d) / 4);
  return (size_t)(octs / (32 * f.qt)) * 2 * d->cout * 2;
}

// rows of the frame (in the transposed frame: range positions) one tile of the partials covers: tile t = rows [t * rows_per_tile, ...) of
// the images laid end to end, both rows of a row pair kept apart ([tile][row][cout][2]: row r of pair k of the tile is index k = 0 only
// when the tile is one pair)
int pn_conv_wino24_chain_stat_tile_rows(const pn_conv_desc* d) {
  Chain2Form f;
  if (!chain2_ok(d, true, f)) return 0;
  return 2 * (32 * f.qt) / (frame_w(d) / 4);
}

int pn_conv2d_wino24_chain_head_f32(const pn_conv_desc* d, const float* planes_in, const float* packed_w24, const float* scale, const float* shift,
                                    float* planes_out, float* out_nhwc, float* stat_partials, pn_stream_t stream) {
  return chain2_run(d, planes_in, packed_w24, scale, shift, planes_out, out_nhwc, stat_partials, true, stream);
}

int pn_conv2d_wino24_chain_head_multi_f32(const pn_chain_head_job* jobs, int njobs, pn_stream_t stream) {
  PN_REQUIRE(jobs && njobs >= 1 && njobs <= kChainMultiJobs, "conv_wino24_chain_head_multi: 1 - 4 jobs");
  WChainMulti m{};
  m.njobs = njobs;
  Chain2Form f0{};
  for (int j = 0; j < njobs; ++j) {
    const pn_chain_head_job& q = jobs[j];
    Chain2Form f;
    const int rc = chain2_fill(q.desc, q.planes_in, q.packed_w24, q.scale, q.shift, q.planes_out, q.out_nhwc, q.stat_partials, true, m.job[j], f);
    if (rc != PN_OK) return rc;
    if (j == 0) f0 = f;
    PN_REQUIRE(f.ks == f0.ks && f.ct == f0.ct && f.qt == f0.qt, "conv_wino24_chain_head_multi: the jobs must share one kernel form (same map, same cin)");
    m.first[j + 1] = m.first[j] + m.job[j].qtiles * m.job[j].ctiles;
  }
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  hipStream_t st = pn::S(stream);
  // short K (the head's 64 input channels: four steps per K half), enough tiles for two blocks per CU and other frames in flight
  // (pn_conv_desc.frames_in_flight): six-wave blocks that keep the whole K -- half the waves, no K join; alone on the chip the launch is
  // 3 us slower (39.6 against 36.3 us), beside three other frames the whole job gains 1.5 % (PN_WCHAIN_MULTI_KS1=0 / 2: never / always)
  static const int ks1 = [] { const char* e = getenv("PN_WCHAIN_MULTI_KS1"); return e ? atoi(e) : 1; }();
  const bool others = ks1 == 2 || (ks1 == 1 && jobs[0].desc->frames_in_flight > 1);
  if (f0.ks == 2 && others && m.job[0].cg_in <= 8 && m.first[njobs] >= 2 * chain_grid_limit()) launch_chain2_multi<1, 1, 1>(m, st, prof, ps);
  else if (f0.ks == 2) launch_chain2_multi<2, 1, 1>(m, st, prof, ps);
  else launch_chain2_multi<1, 1, 2>(m, st, prof, ps);
  return pn::check_launch("conv_wchain2_multi_kernel");
}

int pn_wino24_chain_head_finalize_f32(const pn_head_stat_job* jobs, int njobs, int batch, int frame_rows, int frame_row_pixels, int tile_rows,
                                      pn_stream_t stream) {
  PN_REQUIRE(jobs && njobs >= 1 && njobs <= kHeadFinJobs && batch >= 1 && frame_rows >= 1 && tile_rows >= 1 && frame_rows % tile_rows == 0,
             "wino24_chain_head_finalize: bad arguments");
  HeadFinArgs f{};
  f.njobs = njobs; f.B = batch; f.rows = frame_rows; f.tile_rows = tile_rows; f.row_pixels = frame_row_pixels;
  for (int j = 0; j < njobs; ++j) {
    const pn_head_stat_job& q = jobs[j];
    PN_REQUIRE(q.partials && q.table && q.channels >= 1 && q.channel_offset >= 0 && q.channel_offset + q.channels <= q.cout_total,
               "wino24_chain_head_finalize: bad job");
    PN_REQUIRE(q.strata <= 8 && (q.strata <= 1 ? q.channels <= 256 : (frame_rows / tile_rows) % q.strata == 0),
               "wino24_chain_head_finalize: per-channel jobs take <= 256 channels; strata (<= 8) must divide the sample's tiles");
    f.job[j] = HeadFinJob{q.partials, q.cout_total, q.channel_offset, q.channels, q.strata, q.gamma, q.beta, q.eps, q.table};
  }
  hipLaunchKernelGGL(wchain_head_finalize_kernel, dim3((unsigned)(njobs * batch * 8)), dim3(256), 0, pn::S(stream), f);
  return pn::check_launch("wchain_head_finalize_kernel");
}

}  // extern "C"
