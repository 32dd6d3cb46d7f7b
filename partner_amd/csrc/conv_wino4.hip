// 3x3 / stride-1 / pad-1 convolution with the one-dimensional Winograd transform F(4, 3) along the image width, on the fp32 MFMA:
// the second step after conv_wino.hip's F(2, 3).  Same layers (Conv2d + folded BatchNorm + ReLU of the RPN,
// det3d/models/necks/rpn.py:124-142): a QUAD of horizontally adjacent outputs (x = 4t .. 4t + 3) of one row needs SIX products per
// kernel row and input channel instead of twelve -- 4.5 MFMA-equivalents per output against 9 (direct) and 6 (F(2, 3)).
//
//   d0..d5 = the input pixels x = 4t - 1 .. 4t + 4 of input row y + kh - 1, g0..g2 the three kw taps of kernel row kh:
//     v0 = 4 d0 - 5 d2 + d4            u0 = g0 / 4
//     v1 = -4 d1 - 4 d2 + d3 + d4      u1 = -(g0 + g1 + g2) / 6
//     v2 = 4 d1 - 4 d2 - d3 + d4       u2 = -(g0 - g1 + g2) / 6
//     v3 = -2 d1 - d2 + 2 d3 + d4      u3 = g0 / 24 + g1 / 12 + g2 / 6
//     v4 = 2 d1 - d2 - 2 d3 + d4       u4 = g0 / 24 - g1 / 12 + g2 / 6
//     v5 = 4 d1 - 5 d3 + d5            u5 = g2
//   m_q = sum_kh sum_ci v_q u_q;   out[4t] = m0 + m1 + m2 + m3 + m4,  out[4t+1] = m1 - m2 + 2 m3 - 2 m4,
//   out[4t+2] = m1 + m2 + 4 m3 + 4 m4,  out[4t+3] = m1 - m2 + 8 m3 - 8 m4 + m5       (interpolation points 0, +-1, +-2, inf).
//   SIX independent GEMMs (M = quads, N = Cout, K = 3 Cin).  Unlike the F(2, 3) kernel a wave owns ALL six positions of its
//   32 quads x 32 columns (6 x 16 accumulator registers), so the output transform is wave-local: no LDS join, the epilogue goes
//   from the accumulators to global memory.
//
// Block = FOUR waves (one per SIMD) = 32 quads (128 output pixels) x 128 columns, wave ct on columns 32 ct .. 32 ct + 31; the kernel
// fits 256 registers, so TWO blocks share a CU: independent blocks drift out of phase, each SIMD's second wave fills the first
// one's barrier / staging / epilogue bubbles without the two being tied by a barrier.  (Measured on the 256 x 256 x 128 -> 128
// layer, launches back to back: this form 80 us; eight waves per block on 64 quads with two wave rows 85-94 us depending on how
// much of the operand double-buffering survived the 256-register budget; four waves with 64 quads per wave and 512 registers 90 us;
// F(2, 3) 112-119 us; direct 175 us.)
//   The transformed input goes through LDS ([6 q][32 quads][32 + 4] per stage, formed in registers on the way in); the transformed
//   weights do not: the packed layout is the MFMA fragment layout and every lane fetches its own 16-byte fragments from L2 one
//   sub-step (24 MFMAs) ahead, as conv_wino_bd_kernel does.
// Numerics: the transforms carry small-integer factors (<= 8) and 1/24; measured on a 13-layer stack the error against fp64 is
// 1.4e-6 of the map's range (direct fp32: 0.8e-6, F(2, 3): 0.8e-6) -- far inside the 1e-4 parity bound the tests hold.
#include "pn_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int W4N = 128;        // output columns per block tile
constexpr int W4_LD = 36;       // floats per (position, quad) row of the A image: 32 channels + 4 (bank spread)

// two A stages + the block tile's 128 scale and 128 shift values (read by the epilogue from LDS: a vector-memory load there would
// have to wait -- the counter returns in order -- for every output store issued before it)
constexpr size_t wino4_smem(int quads) { return (2 * (size_t)(6 * quads * W4_LD) + 2 * W4N) * sizeof(float); }

// B^T d for four channels at once, in 13 fused multiply-adds / adds per channel (r2: 18 separate multiplies and adds -- the build runs
// with -ffp-contract=off -- and on this chip a VALU instruction is not hidden behind another wave's fp32 MFMAs: the two share the SIMD's
// fp32 lanes, tools/micro/mfma_valu_coexec.hip).  Vector-wide fma: the compiler emits v_pk_fma_f32 / v_pk_add_f32, two channels each.
//   v0 = 4 d0 - 5 d2 + d4,  v1 = e + o,  v2 = e - o  (e = d4 - 4 d2, o = d3 - 4 d1),  v3 = f + g,  v4 = f - g  (f = d4 - d2,
//   g = 2 (d3 - d1)),  v5 = 4 d1 - 5 d3 + d5
__device__ __forceinline__ void wino4_input_transform(const f32x4 (&d)[6], f32x4 (&v)[6]) {
  const f32x4 c4 = {4.f, 4.f, 4.f, 4.f}, m4 = {-4.f, -4.f, -4.f, -4.f}, m5 = {-5.f, -5.f, -5.f, -5.f}, c2 = {2.f, 2.f, 2.f, 2.f}, m2 = {-2.f, -2.f, -2.f, -2.f};
  const f32x4 e = __builtin_elementwise_fma(m4, d[2], d[4]), o = __builtin_elementwise_fma(m4, d[1], d[3]);
  const f32x4 f = d[4] - d[2], t = d[3] - d[1];
  v[0] = __builtin_elementwise_fma(c4, d[0], __builtin_elementwise_fma(m5, d[2], d[4]));
  v[1] = e + o;
  v[2] = e - o;
  v[3] = __builtin_elementwise_fma(c2, t, f);
  v[4] = __builtin_elementwise_fma(m2, t, f);
  v[5] = __builtin_elementwise_fma(c4, d[1], __builtin_elementwise_fma(m5, d[3], d[5]));
}

struct Wino4Args {
  const float* in;
  const float* w;
  const float* scale;
  const float* shift;
  float* out;
  int B, H, W, Cin, Cout;
  int in_ps, in_co, out_ps, out_co;
  int out_t;      // 1: the output map is stored transposed, [b][x][y][channels] (pn_conv_desc.transpose_hw)
  int act;
  int quads_per_row, total_quads, qtiles;
  int qt0;                 // first 32-quad tile of this launch (two-phase launches: the tail of a map goes to a second launch)
  int ncol;                // 128-column tiles
  int chunks, cout_pad;
  unsigned in_bytes, w_bytes;
#ifdef PN_WINO4_STAMP
  unsigned long long* stamps;   // diagnostic build only (tools/micro/wino4_stamps.hip): [block][8] shader-clock stamps of wave 0, first tile
#endif
};

#ifdef PN_WINO4_STAMP
unsigned long long* pn_wino4_stamp_buffer = nullptr;
#define W4_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (tid == 0 && tl == slot) a.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define W4_STAMP(k) do { } while (0)
#endif
#ifndef PN_WINO4_EXP
#define PN_WINO4_EXP 0   // diagnostic build only: bit 0 no input transform, 1 no per-step barrier, 2 no fragment reads, 3 no LDS stores, 4 no loads, 5 no weight loads, 6 no input loads
#endif

__global__ __launch_bounds__(256, 2) void conv_wino4_kernel(Wino4Args a) {
  constexpr int TM = 1;                       // 32-quad MFMA tiles per wave
  constexpr int WQ = 32;                      // quads per block tile
  constexpr int WA_FLOATS = 6 * WQ * W4_LD;
  constexpr int PPT = 1;                      // (quad, channel quad) items per thread in the loader: 32 x 8 items, 256 threads
  constexpr int PSTEP = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* affine = smem + 2 * WA_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, ct = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // persistent blocks, XCD-local runs of quad tiles (see conv_wino.hip)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int xq = a.qtiles >> 3, xr = a.qtiles & 7;
  const int px0 = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xtiles = (xcd < xr ? xq + 1 : xq) * a.ncol;
  int pt = 0, n0 = 0;
  const int pl = tid >> 3, c4 = tid & 7;

  const long long back = ((long long)a.W + 1) * a.in_ps;   // floats: the descriptor base is moved back so that every voffset >= 0
  // per item: byte offset of pixel (b, oh - 1, 4 oq - 1), channel in_co + 4 c4; the same with the first / the last of the six pixels
  // masked (0xffffffff = out of range -> the load returns 0) when it falls outside the row; bit kh of a_rmask: input row inside the map
  unsigned a_off[PPT], a_off0[PPT], a_off5[PPT], a_rmask[PPT];
  unsigned b_base = 0;     // byte offset of this lane's fragment (q = 0, sub-step 0) inside one (chunk, kh) block of the packed weights
  int ld_kh = 0, ld_chunk = 0;     // A loader position
  int lb_kh = 0, lb_chunk = 0;     // B loader position: the K step whose fragments are requested next
  auto setup_tile = [&](int tl) {
    pt = a.qt0 + px0 + tl / a.ncol;
    n0 = (tl - (tl / a.ncol) * a.ncol) * W4N;
    ld_kh = ld_chunk = lb_kh = lb_chunk = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = pt * WQ + pl + PSTEP * k;
      const bool ok = p < a.total_quads;
      const int pp = ok ? p : 0;
      const int rowi = pp / a.quads_per_row, oq = pp - rowi * a.quads_per_row;
      const int b = rowi / a.H, oh = rowi - b * a.H;
      const long long pix = ((long long)b * a.H + (oh - 1)) * a.W + (4 * oq - 1);
      const unsigned off = (unsigned)((pix * a.in_ps + back + a.in_co + c4 * 4) * 4);
      a_off[k] = off;
      a_off0[k] = oq > 0 ? off : 0xffffffffu;
      a_off5[k] = oq + 1 < a.quads_per_row ? off : 0xffffffffu;
      unsigned rm = 0;
      for (int kh = 0; kh < 3; ++kh)
        if (ok && (unsigned)(oh + kh - 1) < (unsigned)a.H) rm |= 1u << kh;
      a_rmask[k] = rm;
    }
    b_base = (unsigned)(((size_t)lh * a.cout_pad + n0 + ct * 32 + li) * 16);
    if (tid < W4N) {      // this tile's per-channel affine into LDS (read back before the tile's last barrier)
      const int c = n0 + tid;
      affine[tid] = (a.scale && c < a.Cout) ? a.scale[c] : 1.f;
      affine[W4N + tid] = (a.shift && c < a.Cout) ? a.shift[c] : 0.f;
    }
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.in)) - back * 4, 0,
                                                                         a.in_bytes + (unsigned)(back * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int nsteps = 3 * a.chunks;
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned ps4 = (unsigned)a.in_ps * 4u;

  f32x4 ra[PPT][6];      // the input pixels of the NEXT K step: stored (transformed) during this step, then reloaded for the step after
  auto load_a = [&](bool live) {
    const unsigned so_a = (unsigned)((ld_kh * a.W * a.in_ps + ld_chunk * 32) * 4);
    const bool cok = live & (ld_chunk * 32 + c4 * 4 < a.Cin);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const bool rok = cok & (((a_rmask[k] >> ld_kh) & 1u) != 0);
#pragma unroll
      for (int j = 0; j < 6; ++j) {   // the pixel step rides in the scalar offset (not part of the range check: the vector offset alone decides)
        const unsigned base = j == 0 ? a_off0[k] : j == 5 ? a_off5[k] : a_off[k];
        ra[k][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, rok ? base : 0xffffffffu, so_a + (unsigned)j * ps4, 0));
      }
    }
    if (++ld_kh == 3) { ld_kh = 0; ++ld_chunk; }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      float* As = smem + buf * WA_FLOATS + (pl + PSTEP * k) * W4_LD + c4 * 4;
      f32x4 v[6];
      wino4_input_transform(ra[k], v);
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(As + q * WQ * W4_LD) = v[q];
    }
  };
  // this lane's weight fragments of one sub-step: one per position.  packed [chunk][kh][q 6][k4 8][cout_pad][4]
  unsigned so_b = 0;        // scalar byte offset of the K step being requested
  auto b_step_offset = [&]() { return (unsigned)((lb_chunk * 3 + lb_kh) * 48) * cp16; };
  auto b_advance = [&]() {
    if (++lb_kh == 3) { lb_kh = 0; ++lb_chunk; }
    so_b = b_step_offset();
  };
  f32x16 acc[6][TM];
  const int a_frag = li * W4_LD + lh * 4;
  // operand fragments, double-buffered by sub-step: while sub-step s multiplies, the fragments of s + 1 are in flight -- the weights
  // from L2 (requested a whole sub-step = 24 TM MFMAs ahead, and BEFORE this sub-step's input loads: the vector-memory counter
  // returns in order, so a fragment requested after them could not be used before they have landed), the inputs from LDS
  f32x4 af[2][6][TM], bf[2][6];
  auto read_a = [&](int buf, int s, f32x4 (&f)[6][TM]) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i) f[q][i] = *reinterpret_cast<const f32x4*>(smem + buf * WA_FLOATS + a_frag + s * 8 + (q * WQ + 32 * i) * W4_LD);
  };
  auto load_b = [&](bool live, int s, f32x4 (&f)[6]) {
    const unsigned vo = live ? b_base : 0xffffffffu;      // (position, sub-step) in the scalar offset
#pragma unroll
    for (int q = 0; q < 6; ++q)
      f[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vo, so_b + (unsigned)(q * 8 + 2 * s) * cp16, 0));
  };
  // ---- one K step (32 channels of one kernel row) = four sub-steps of 24 TM MFMAs; the barrier sits before the last one.  The input
  // transform + LDS stores of step t+1 ride on sub-step 1, the input loads of step t+2 (into the registers just stored) on sub-step 2.
  constexpr int NM = 24 * TM;
  auto substep = [&](auto s_c, auto store_c, auto load_c, int t, int buf) __attribute__((always_inline)) {
    constexpr int s = decltype(s_c)::value;
    constexpr bool DO_STORE = decltype(store_c)::value, DO_LOAD = decltype(load_c)::value;
    constexpr bool last = s == 3;
    f32x4 (&ca)[6][TM] = af[s & 1];
    f32x4 (&cb)[6] = bf[s & 1];
    f32x4 (&na)[6][TM] = af[(s + 1) & 1];
    f32x4 (&nb)[6] = bf[(s + 1) & 1];
    if (last) {
      __syncthreads();
      b_advance();
      load_b(t + 1 < nsteps, 0, nb);
      read_a(buf ^ 1, 0, na);
    } else {
      load_b(true, s + 1, nb);
      read_a(buf, s + 1, na);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[q][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[q][i][kk], cb[q][kk], acc[q][i], 0, 0, 0);
    if (DO_STORE) store_a(buf ^ 1);
    if (DO_LOAD) load_a(t + 2 < nsteps);
    // schedule: the six weight loads and the 6 TM LDS reads under the first twelve MFMAs, then the transform + LDS stores and / or
    // the input loads spread over the rest
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
    }
    constexpr int REST = NM - 12;
    if (DO_STORE && DO_LOAD) {
#pragma unroll
      for (int k = 0; k < 6 * PPT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
#pragma unroll
      for (int k = 0; k < 6 * PPT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, (REST - 6 * PPT) / (6 * PPT) > 0 ? (REST - 6 * PPT) / (6 * PPT) : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    } else if (DO_STORE) {
#pragma unroll
      for (int k = 0; k < 6 * PPT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, REST / (6 * PPT), 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    } else if (DO_LOAD) {
#pragma unroll
      for (int k = 0; k < 6 * PPT; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, REST / (6 * PPT), 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using Yes = std::true_type;
  using No = std::false_type;
  // the transform + LDS stores of step t+1 ride on sub-step 1, the input loads of step t+2 (into the registers just stored) on sub-step 2
  auto kloop = [&]() __attribute__((always_inline)) {
    for (int t = 0; t < nsteps; ++t) {
      const int buf = t & 1;
      substep(I0{}, No{}, No{}, t, buf);
      substep(I1{}, Yes{}, No{}, t, buf);
      substep(I2{}, No{}, Yes{}, t, buf);
      substep(I3{}, No{}, No{}, t, buf);
    }
  };

  // ---- epilogue: output transform in registers, affine + activation, four pixels per quad
  auto epilogue = [&](int ept, int en0, float sc, float sh) {
    const int col = en0 + ct * 32 + li;
    if (col < a.Cout) {
      const float lo = a.act == PN_ACT_RELU ? 0.f : -__builtin_inff();     // act is NONE or RELU here (checked by the launcher)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = ept * WQ + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (p >= a.total_quads) continue;
          const float m0 = acc[0][i][r], m1 = acc[1][i][r], m2 = acc[2][i][r], m3 = acc[3][i][r], m4 = acc[4][i][r], m5 = acc[5][i][r];
          const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
          const float y0 = (m0 + s12) + s34;
          const float y1 = d12 + 2.f * d34;
          const float y2 = s12 + 4.f * s34;
          const float y3 = (d12 + 8.f * d34) + m5;
          const int rowi = p / a.quads_per_row, oq = p - rowi * a.quads_per_row;
          size_t px = (size_t)a.out_ps;
          float* o = a.out + ((size_t)rowi * a.W + 4 * oq) * a.out_ps + a.out_co + col;
          if (a.out_t) {
            const int img = rowi / a.H, y = rowi - img * a.H;
            px = (size_t)a.H * a.out_ps;
            o = a.out + ((size_t)img * a.W + 4 * oq) * px + (size_t)y * a.out_ps + a.out_co + col;
          }
          o[0] = fmaxf(fmaf(y0, sc, sh), lo);
          o[px] = fmaxf(fmaf(y1, sc, sh), lo);
          o[2 * px] = fmaxf(fmaf(y2, sc, sh), lo);
          o[3 * px] = fmaxf(fmaf(y3, sc, sh), lo);
        }
    }
  };

  int prev_pt = -1, prev_n0 = 0;
  float prev_sc = 1.f, prev_sh = 0.f;
  for (int tl = slot; tl < xtiles; tl += per_xcd) {
    setup_tile(tl);
    so_b = b_step_offset();
    load_a(true);
    load_b(true, 0, bf[0]);
    if (prev_pt >= 0) epilogue(prev_pt, prev_n0, prev_sc, prev_sh);   // runs while the loads just requested are in flight
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;
    store_a(0);
    load_a(nsteps > 1);           // step 1: stored during step 0
    __syncthreads();
    read_a(0, 0, af[0]);
    kloop();
    prev_sc = affine[ct * 32 + li];
    prev_sh = affine[W4N + ct * 32 + li];
    __syncthreads();   // every wave is done with the last stage before the next tile's first store reuses the LDS
    prev_pt = pt;
    prev_n0 = n0;
  }
  if (prev_pt >= 0) epilogue(prev_pt, prev_n0, prev_sc, prev_sh);
}

// ---- K-split form for maps whose 32-quad x 128-column tiles would not fill the chip (the 128 x 128 and 64 x 64 maps of the nuScenes
// RPN: 128 resp. 64 such tiles on 512 block slots).  Block tile = 32 quads x 32 columns, FOUR times as many blocks; the four waves of a
// block share the tile and split K: wave ks multiplies sub-step ks (8 of the 32 channels) of every K step, one barrier per K step.
// The four partial accumulator sets are joined through LDS all-to-all -- wave d receives rows 8 d .. 8 d + 7 (accumulator registers
// 4 d .. 4 d + 3 of all six positions) from the other three -- so every wave transforms and stores a quarter of the tile.
constexpr int W4K_N = 32;
constexpr size_t wino4_ks_smem() {
  const size_t stage = (2 * (size_t)(6 * 32 * W4_LD) + 2 * W4K_N) * sizeof(float);
  const size_t join = (size_t)4 * 3 * 6 * 64 * 4 * sizeof(float);      // [dst wave][src slot][q][lane][4 registers]
  return stage > join ? stage : join;
}

// Where a step's time goes (tools/micro/wino4_stamps.hip, 64 x 64 x 256 -> 256, one wave per SIMD): 2560 cycles per step for 1536 cycles
// of MFMA.  Without the six input loads 1730, without the six weight loads 1980, without both 1656: the costs ADD, so they are not
// latencies (a second register set that issues the input loads two steps ahead, and a third LDS stage that reads the next step's
// fragments under the MFMAs, both changed nothing and were dropped) -- a 32 x 32 tile moves 48 KB per step and CU through the vector
// L1 for 96 MFMAs (31 B per clock, the plain form 20), and a wave that issues in order stands behind its own loads.
__global__ __launch_bounds__(256, 2) void conv_wino4_ks_kernel(Wino4Args a) {
  constexpr int WQ = 32;
  constexpr int WA_FLOATS = 6 * WQ * W4_LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int xq = a.qtiles >> 3, xr = a.qtiles & 7;
  const int px0 = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int xtiles = (xcd < xr ? xq + 1 : xq) * a.ncol;      // a.ncol = 32-column tiles here
  int pt = 0, n0 = 0;
  const int pl = tid >> 3, c4 = tid & 7;

  const long long back = ((long long)a.W + 1) * a.in_ps;
  unsigned a_off = 0, a_off0 = 0, a_off5 = 0, a_rmask = 0;
  unsigned b_base = 0;
  int ld_kh = 0, ld_chunk = 0, lb_kh = 0, lb_chunk = 0;
  float sc = 1.f, sh = 0.f;
  auto setup_tile = [&](int tl) {
    pt = a.qt0 + px0 + tl / a.ncol;
    n0 = (tl - (tl / a.ncol) * a.ncol) * W4K_N;
    ld_kh = ld_chunk = lb_kh = lb_chunk = 0;
    const int p = pt * WQ + pl;
    const bool ok = p < a.total_quads;
    const int pp = ok ? p : 0;
    const int rowi = pp / a.quads_per_row, oq = pp - rowi * a.quads_per_row;
    const int b = rowi / a.H, oh = rowi - b * a.H;
    const long long pix = ((long long)b * a.H + (oh - 1)) * a.W + (4 * oq - 1);
    const unsigned off = (unsigned)((pix * a.in_ps + back + a.in_co + c4 * 4) * 4);
    a_off = off;
    a_off0 = oq > 0 ? off : 0xffffffffu;
    a_off5 = oq + 1 < a.quads_per_row ? off : 0xffffffffu;
    unsigned rm = 0;
    for (int kh = 0; kh < 3; ++kh)
      if (ok && (unsigned)(oh + kh - 1) < (unsigned)a.H) rm |= 1u << kh;
    a_rmask = rm;
    b_base = (unsigned)((((size_t)2 * ks + lh) * a.cout_pad + n0 + li) * 16);
    // the column's affine: requested here, a whole K loop before the epilogue uses it
    const int c = n0 + li;
    sc = (a.scale && c < a.Cout) ? a.scale[c] : 1.f;
    sh = (a.shift && c < a.Cout) ? a.shift[c] : 0.f;
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<float*>(a.in)) - back * 4, 0,
                                                                         a.in_bytes + (unsigned)(back * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int nsteps = 3 * a.chunks;
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned ps4 = (unsigned)a.in_ps * 4u;

  f32x4 ra[6];
  auto load_a = [&](bool live) {
    const unsigned so_a = (unsigned)((ld_kh * a.W * a.in_ps + ld_chunk * 32) * 4);
    const bool rok = live & (ld_chunk * 32 + c4 * 4 < a.Cin) & (((a_rmask >> ld_kh) & 1u) != 0);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const unsigned base = j == 0 ? a_off0 : j == 5 ? a_off5 : a_off;
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, rok ? base : 0xffffffffu, so_a + (unsigned)j * ps4, 0));
    }
    if (++ld_kh == 3) { ld_kh = 0; ++ld_chunk; }
  };
  auto store_a = [&](int stage) {
    float* As = smem + stage * WA_FLOATS + pl * W4_LD + c4 * 4;
    f32x4 v[6];
    if constexpr (PN_WINO4_EXP & 1) {
#pragma unroll
      for (int q = 0; q < 6; ++q) v[q] = ra[q];
    } else {
      wino4_input_transform(ra, v);
    }
    if constexpr (!(PN_WINO4_EXP & 8)) {
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(As + q * WQ * W4_LD) = v[q];
    } else {
      if (v[0][0] == 12345.f) *reinterpret_cast<f32x4*>(As) = v[1] + v[2] + v[3] + v[4] + v[5];
    }
  };
  unsigned so_b = 0;
  auto b_step_offset = [&]() { return (unsigned)((lb_chunk * 3 + lb_kh) * 48) * cp16; };
  f32x16 acc[6];
  const int a_frag = li * W4_LD + lh * 4 + ks * 8;
  f32x4 af[6], bf[2][6];
  auto read_a = [&](int stage, f32x4 (&f)[6]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) f[q] = *reinterpret_cast<const f32x4*>(smem + stage * WA_FLOATS + a_frag + q * WQ * W4_LD);
  };
  auto load_b = [&](bool live, f32x4 (&f)[6]) {      // this wave's sub-step of the K step at (lb_chunk, lb_kh), then advance
    const unsigned vo = live ? b_base : 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 6; ++q) f[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vo, so_b + (unsigned)(q * 8) * cp16, 0));
    if (++lb_kh == 3) { lb_kh = 0; ++lb_chunk; }
    so_b = b_step_offset();
  };
  // one K step: barrier (tile t is in LDS, tile t-1 has been read by everyone), this wave's six A fragments, 24 MFMAs; in their shadow
  // the weight fragments of step t+1, the transform + LDS stores of step t+1's tile and the input loads of step t+2
  auto kstep = [&](auto par_c, int t) __attribute__((always_inline)) {
    constexpr int buf = decltype(par_c)::value;
    if constexpr (!(PN_WINO4_EXP & 2)) __syncthreads();
    if constexpr (!(PN_WINO4_EXP & 4)) {
      read_a(buf, af);
    }
    if constexpr (!(PN_WINO4_EXP & (16 | 32))) load_b(t + 1 < nsteps, bf[buf ^ 1]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q][kk], bf[buf][q][kk], acc[q], 0, 0, 0);
    store_a(buf ^ 1);
    if constexpr (!(PN_WINO4_EXP & (16 | 64))) load_a(t + 2 < nsteps);
    // schedule: the six LDS reads first; every vector-memory load behind a pair of MFMAs (a wave issues in order: twelve loads in a
    // row wait for the CU's one texture-address path, shared with the three other waves that pass the barrier at the same moment, and
    // the MFMAs behind them wait too -- with the loads removed a step of the lone-wave layers takes 1656 cycles instead of 2730,
    // tools/micro/wino4_stamps.hip); the transform + LDS stores under the first half of the MFMAs, the input loads
    // under the second (a finer interleave -- one MFMA per five VALU -- measured slower: 34 against 31 us on the 64 x 64 x 256 layer)
#pragma unroll
    for (int k = 0; k < 6; ++k) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // join + epilogue of wave D (compile-time: the accumulator registers are indexed statically)
  auto finish = [&](auto d_c, int ept, int en0) __attribute__((always_inline)) {
    constexpr int D = decltype(d_c)::value;
    f32x4* J = reinterpret_cast<f32x4*>(smem) + lane;      // [dst wave][src slot][q][lane] x 4 registers: 128-bit LDS accesses
    // send: to every other wave its quarter of this wave's partial sums
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d == D) continue;
      const int sl = D < d ? D : D - 1;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const f32x4 v = {acc[q][4 * d], acc[q][4 * d + 1], acc[q][4 * d + 2], acc[q][4 * d + 3]};
        J[((d * 3 + sl) * 6 + q) * 64] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const f32x4 v = J[((D * 3 + sl) * 6 + q) * 64];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[q][4 * D + rr] += v[rr];
      }
    const int col = en0 + li;
    if (col < a.Cout) {
      const float lo = a.act == PN_ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * D + rr;
        const int p = ept * WQ + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p >= a.total_quads) continue;
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        const float y0 = (m0 + s12) + s34;
        const float y1 = d12 + 2.f * d34;
        const float y2 = s12 + 4.f * s34;
        const float y3 = (d12 + 8.f * d34) + m5;
        const int rowi = p / a.quads_per_row, oq = p - rowi * a.quads_per_row;
        size_t px = (size_t)a.out_ps;
        float* o = a.out + ((size_t)rowi * a.W + 4 * oq) * a.out_ps + a.out_co + col;
        if (a.out_t) {
          const int img = rowi / a.H, y = rowi - img * a.H;
          px = (size_t)a.H * a.out_ps;
          o = a.out + ((size_t)img * a.W + 4 * oq) * px + (size_t)y * a.out_ps + a.out_co + col;
        }
        o[0] = fmaxf(fmaf(y0, sc, sh), lo);
        o[px] = fmaxf(fmaf(y1, sc, sh), lo);
        o[2 * px] = fmaxf(fmaf(y2, sc, sh), lo);
        o[3 * px] = fmaxf(fmaf(y3, sc, sh), lo);
      }
    }
    __syncthreads();   // the join buffer overlaps the next tile's stages
  };

  for (int tl = slot; tl < xtiles; tl += per_xcd) {
    W4_STAMP(0);
    setup_tile(tl);
    so_b = b_step_offset();
    load_a(true);
    load_b(true, bf[0]);
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    W4_STAMP(1);
    store_a(0);
    load_a(nsteps > 1);
    W4_STAMP(2);
    for (int t = 0; t < nsteps; t += 2) {
      kstep(I0{}, t);
      if (t == 0) W4_STAMP(3);
      if (t + 1 < nsteps) kstep(I1{}, t + 1);
    }
    W4_STAMP(4);
    __syncthreads();   // every wave is done reading the last stage: the join may overwrite it
    W4_STAMP(5);
    if (ks == 0) finish(I0{}, pt, n0);
    else if (ks == 1) finish(I1{}, pt, n0);
    else if (ks == 2) finish(I2{}, pt, n0);
    else finish(I3{}, pt, n0);
    W4_STAMP(6);
  }
}

// torch (Cout, Cin, 3, 3) -> [chunk][kh][q 6][k4 8][cout_pad][4], transformed in double, rounded once
// dgrad: w is the FORWARD weight (cin, cout, 3, 3) of the layer whose data gradient this convolution is -- taps mirrored, channels swapped
__global__ void pack_wino4_weight_kernel(const float* __restrict__ w, int cout, int cin, int chunks, int cout_pad, float* __restrict__ packed, size_t total,
                                         int dgrad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int k1 = r & 3; r >>= 2;
    const int n = (int)(r % cout_pad); r /= cout_pad;
    const int k4 = r & 7; r >>= 3;
    const int q = (int)(r % 6); r /= 6;
    const int kh = (int)(r % 3);
    const int chunk = (int)(r / 3);
    const int c = chunk * 32 + k4 * 4 + k1;
    double v = 0.0;
    if (n < cout && c < cin) {
      const float* g = dgrad ? w + (((size_t)c * cout + n) * 3 + (2 - kh)) * 3 : w + (((size_t)n * cin + c) * 3 + kh) * 3;
      const double g0 = dgrad ? g[2] : g[0], g1 = g[1], g2 = dgrad ? g[0] : g[2];
      v = q == 0 ? g0 / 4.0 : q == 1 ? -(g0 + g1 + g2) / 6.0 : q == 2 ? -(g0 - g1 + g2) / 6.0 : q == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0
          : q == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0 : g2;
    }
    packed[i] = (float)v;
  }
}

}  // namespace

extern "C" {

size_t pn_conv_wino4_packed_weight_floats(int cout, int cin) {
  return (size_t)pn::cdiv(cin, 32) * 3 * 6 * 8 * (size_t)(pn::cdiv(cout, W4N) * W4N) * 4;
}

static int pack_wino4(const float* w, int cout, int cin, float* packed, pn_stream_t stream, int dgrad) {
  PN_REQUIRE(w && packed && cout >= 1 && cin >= 1, "pack_conv_weight_wino4: bad arguments");
  const int chunks = pn::cdiv(cin, 32), cout_pad = pn::cdiv(cout, W4N) * W4N;
  const size_t total = pn_conv_wino4_packed_weight_floats(cout, cin);
  hipLaunchKernelGGL(pack_wino4_weight_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w, cout, cin,
                     chunks, cout_pad, packed, total, dgrad);
  return pn::check_launch("pack_wino4_weight_kernel");
}

int pn_pack_conv_weight_wino4_f32(const float* w_oihw, int cout, int cin, float* packed, pn_stream_t stream) {
  return pack_wino4(w_oihw, cout, cin, packed, stream, 0);
}

// the weights of the DATA-GRADIENT convolution straight from the forward layer's (Cout_fwd, Cin_fwd, 3, 3) tensor (see conv_wino.hip)
int pn_pack_conv_dgrad_weight_wino4_f32(const float* w_fwd_oihw, int cout_fwd, int cin_fwd, float* packed, pn_stream_t stream) {
  return pack_wino4(w_fwd_oihw, cin_fwd, cout_fwd, packed, stream, 1);
}

int pn_conv_wino4_tiles(const pn_conv_desc* d) {
  if (!d || d->in_w % 4) return 0;
  const long long quads = (long long)d->batch * d->in_h * (d->in_w / 4);
  return (int)std::min<long long>(1 << 30, ((quads + 31) / 32) * pn::cdiv(d->cout, W4N));
}

// the form a launch takes: 1 = 32 quads x 128 columns per block (every wave on the whole K range), 2 = 32 quads x 32 columns per block
// with K split over the block's four waves (maps with fewer than ~3/4 of the 2-per-CU block slots in tiles of the first form)
static int wino4_form(long long tiles128, int ncu, int frames_in_flight) {
  static const int force = [] { const char* e = getenv("PN_WINO4_KSPLIT"); return e ? atoi(e) : -1; }();
  if (force == 0) return 1;
  if (force == 1) return 2;
  static const int min_tiles = [] { const char* e = getenv("PN_WINO4_PLAIN_MIN_TILES"); return e ? atoi(e) : 0; }();
  if (min_tiles > 0) return tiles128 >= min_tiles ? 1 : 2;
  // other frames run beside this launch (several engines on their own streams): they fill the CUs a plain-form launch of ~100-380 tiles
  // leaves idle, and the plain form issues less non-MFMA work per FLOP than the K-split form (four frames in flight, 128 x 128 layers in
  // the plain form: 1308 -> 1340 frames/s; one frame in flight the same choice costs 8 % latency, so it is taken on the hint only)
  const long long weight = frames_in_flight > 1 ? (frames_in_flight < 4 ? frames_in_flight : 4) : 1;
  return tiles128 * 4 * weight >= 3LL * 2 * ncu ? 1 : 2;
}

int pn_conv2d_wino4_nhwc_f32(const pn_conv_desc* d, const float* in, const float* packed_w, const float* scale, const float* shift, float* out,
                             pn_stream_t stream) {
  PN_REQUIRE(d && in && packed_w && out, "conv_wino4: null pointer");
  PN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->groups == 1 && !d->deconv2x2 && d->range_strata <= 1 &&
                 !d->accumulate && d->pad_h_end == 0 && d->pad_w_end == 0,
             "conv_wino4: plain 3x3 / stride 1 / pad 1 convolutions only");
  PN_REQUIRE(d->batch >= 1 && d->in_h >= 1 && d->in_w >= 4 && d->in_w % 4 == 0, "conv_wino4: the map width must be a multiple of 4");
  PN_REQUIRE(d->cin >= 4 && d->cin % 4 == 0 && d->in_pixel_stride % 4 == 0 && d->in_channel_offset % 4 == 0 && d->cout >= 1,
             "conv_wino4: cin, input pixel stride and channel offset must be multiples of 4");
  PN_REQUIRE(d->in_pixel_stride >= d->in_channel_offset + d->cin && d->out_pixel_stride >= d->out_channel_offset + d->cout,
             "conv_wino4: channel slice does not fit the pixel stride");
  PN_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "conv_wino4: pointers must be 16-byte aligned");
  PN_REQUIRE(d->act == PN_ACT_NONE || d->act == PN_ACT_RELU, "conv_wino4: activation none or ReLU");
  const unsigned long long in_bytes = (unsigned long long)d->batch * d->in_h * d->in_w * d->in_pixel_stride * 4ull;
  PN_REQUIRE(in_bytes + ((unsigned long long)d->in_w + 1) * d->in_pixel_stride * 4ull < (1ull << 32), "conv_wino4: input map too large for the buffer descriptor");
  Wino4Args a{};
  a.in = in; a.w = packed_w; a.scale = scale; a.shift = shift; a.out = out;
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = d->cin; a.Cout = d->cout;
  a.in_ps = d->in_pixel_stride; a.in_co = d->in_channel_offset; a.out_ps = d->out_pixel_stride; a.out_co = d->out_channel_offset;
  a.act = d->act;
  a.out_t = d->transpose_hw ? 1 : 0;
  a.quads_per_row = d->in_w / 4;
  a.total_quads = d->batch * d->in_h * a.quads_per_row;
  a.chunks = pn::cdiv(d->cin, 32);
  a.cout_pad = pn::cdiv(d->cout, W4N) * W4N;
  a.ncol = a.cout_pad / W4N;
  a.in_bytes = (unsigned)in_bytes;
  a.w_bytes = (unsigned)(pn_conv_wino4_packed_weight_floats(d->cout, d->cin) * 4);
#ifdef PN_WINO4_STAMP
  a.stamps = pn_wino4_stamp_buffer;
#endif
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino4_smem(32));
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n / 8 * 8 : 256;
  }
  const int ncu = (dev >= 0 && dev < 64) ? cus[dev] : 256;
  a.qtiles = pn::cdiv(a.total_quads, 32);
  a.qt0 = 0;
  hipStream_t st = pn::S(stream);
  // stop_only: the second launch of a two-phase layer carries the STOP event of the pair whose start rides on the first launch, so the
  // profiler's interval (and the FLOPs billed to it) covers both
  auto launch_ks = [&](Wino4Args k, bool prof, hipEvent_t stop_only) {
    static bool ks_done[64] = {false};
    if (pn::first_use_on_device(ks_done))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4_ks_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino4_ks_smem());
    k.ncol = pn::cdiv(k.Cout, W4K_N);
    const long long tiles = (long long)k.qtiles * k.ncol;
    const dim3 grid((unsigned)std::min<long long>(2 * ncu, (tiles + 7) / 8 * 8));
    pn::ProfileSlot ps;
    if (stop_only) hipExtLaunchKernelGGL(conv_wino4_ks_kernel, grid, dim3(256), wino4_ks_smem(), st, nullptr, stop_only, 0, k);
    else if (prof && pn::take_profile_slot(ps)) hipExtLaunchKernelGGL(conv_wino4_ks_kernel, grid, dim3(256), wino4_ks_smem(), st, ps.start, ps.stop, 0, k);
    else hipLaunchKernelGGL(conv_wino4_ks_kernel, grid, dim3(256), wino4_ks_smem(), st, k);
  };
  // layers whose column count is not a multiple of 128 (64-column layers) would waste the plain form's tile: K-split form
  if (a.Cout % W4N != 0 || wino4_form((long long)a.qtiles * a.ncol, ncu, d->frames_in_flight) == 2) {
    launch_ks(a, true, nullptr);
    return pn::check_launch("conv_wino4_ks_kernel");
  }
  // persistent blocks, two per CU (a multiple of 8: the XCD count), fewer when there are fewer tiles.
  // Two phases (r3): the plain form runs the WHOLE rounds of its 2-per-CU block slots; a last round that would be less than ~60 % full
  // (the Waymo maps: 576 tiles = 1.125 rounds, 1152 = 2.25) goes to a second launch in the K-split form, whose tiles are a quarter
  // as wide.  (The tail rows add their K slices in the K-split form's order: last-bit differences to the rows before them.)
  static const int two_phase = [] { const char* e = getenv("PN_WINO4_TWO_PHASE"); return e ? atoi(e) : 1; }();
  const long long slots = 2LL * ncu;
  long long tiles = (long long)a.qtiles * a.ncol;
  Wino4Args tail = a;
  tail.qtiles = 0;
  if (two_phase && tiles > slots && tiles % slots != 0 && (tiles % slots) * 10 < slots * 6 && slots % a.ncol == 0) {
    const int q1 = (int)((tiles / slots) * slots / a.ncol);      // quad tiles of the full rounds
    tail.qt0 = q1;
    tail.qtiles = a.qtiles - q1;
    a.qtiles = q1;
    tiles = (long long)a.qtiles * a.ncol;
  }
  pn::ProfileSlot ps;
  const bool prof = pn::take_profile_slot(ps);
  const dim3 grid((unsigned)std::min<long long>(slots, (tiles + 7) / 8 * 8));
  const bool two = tail.qtiles > 0;
  if (prof) hipExtLaunchKernelGGL(conv_wino4_kernel, grid, dim3(256), wino4_smem(32), st, ps.start, two ? nullptr : ps.stop, 0, a);
  else hipLaunchKernelGGL(conv_wino4_kernel, grid, dim3(256), wino4_smem(32), st, a);
  if (two) launch_ks(tail, false, prof ? ps.stop : nullptr);
  return pn::check_launch("conv_wino4_kernel");
}

}  // extern "C"
