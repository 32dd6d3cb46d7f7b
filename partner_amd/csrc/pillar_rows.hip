// The first 3x3 / stride-2 convolution of the BEV backbone on the sparse pillar canvas, ROW-BAND form (r6).
// Same layer as pillar_conv.hip (ZeroPad2d(1) + Conv2d(3, stride 2) + folded BatchNorm + ReLU of RPN block 0, det3d/models/necks/rpn.py:124-142,
// on DynamicPPScatter's canvas, det3d/models/readers/pillar_encoder.py:393-432) and the same terms -- only (pillar, tap) pairs are multiplied --
// but without pillar_conv.hip's round trip: that form writes one partial row per pair (32 MB at 28k pillars) and gathers them back per output
// pixel (pair_init + pair + pair_gemm + pair_reduce: 67 us, 114 MB moved for a 50 MB result).
//
// Here ONE block computes ONE output row (all Cout channels) and keeps it in LDS until it is finished:
//   * the frame's cells are sorted by key = (b T + t) R + r, i.e. canvas row by canvas row, and the fused frame index leaves
//     row_start[g] = number of pillars in rows < g (voxelize.hip: cell_scan_kernel, free).  Output row oy reads canvas rows 2 oy - 1 + kh,
//     kh = 0..2: three contiguous runs of the key list -- no search, no window scan, no atomics.  In a polar grid a row is one azimuth: every
//     row sees the whole range profile, so the rows carry the same work (one block per CU, one round).
//   * a pillar in column ix reaches output (ix + 1 - kw) / 2 through tap kw when ix + 1 - kw is even: even columns through kw = 1, odd columns
//     through kw = 0 and kw = 2.  Per canvas row the pillars are split by column parity into two lists; a list is cut into tiles of 16
//     pillars; a tile times a tap is one K = Cin chain of v_mfma_f32_16x16x4_f32 with D[cout][pillar]: the pillar's canvas row is the B
//     operand straight from L2 (16 bytes per lane and four MFMAs), the tap's weights the A operand (packed so that a lane's fragment is one
//     16-byte load), both taps of an odd tile share the B fragments.
//   * wave w owns output channels 16 w .. 16 w + 15 of the row for ALL taps: its accumulator columns go into the LDS row tile [OW][Cout]
//     with plain read-add-write (a tap maps a list's pillars to distinct outputs; the taps follow each other in the wave's program order; no
//     other wave touches these channels) -- no atomics, no barrier, a fixed summation order: bitwise reproducible.
//   * epilogue from the LDS tile: scale / shift / activation, then either the F(4, 3) planes the chained layers read (wino_planes.h: the
//     neighbouring quads' edge pixels are in the same tile) or NHWC.
// Numerics: the same products as the dense convolution, summed per tap in MFMA k-order and over the taps in the fixed order
// (kh, even list, odd list kw = 0, odd list kw = 2): agreement with a float64 convolution ~1e-6 of the map's range (tests/test_hip_wino.py).
#include "pn_common.h"
#include "wino_planes.h"
#include <algorithm>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#ifndef PN_ROWS_EXP
#define PN_ROWS_EXP 0   // diagnostic builds only (tools/rowsq.sh): 1 no (tile, tap) chains, 2 no epilogue, 4 no canvas / weight loads, 8 no MFMAs
#endif
constexpr int RW_LD = 132;       // floats per output pixel of the LDS row tile (128 channels + 4: rows two apart do not share banks)
constexpr int RW_MAXLIST = 256;  // pillars per (canvas row, column parity): at most R / 2 with R <= 512

struct RowsArgs {
  const float* canvas;
  const uint32_t* keys;
  const int32_t* row_start;
  const float* w;          // [tap 9][s Cin / 16][cout_pad][kq 4][m 4]
  const float* scale;
  const float* shift;
  float* planes;           // nullable
  float* out;              // nullable
  int B, T, R, OH, OW, Wq, in_ps, in_co, act, cout, cout_pad, out_ps, out_co, v_cap;
  unsigned plane_floats, canvas_bytes, w_bytes;
};

// One output row per block; NS = Cin / 16 (compile time: the K chain is fully unrolled)
template <int NS>
__global__ __launch_bounds__(512) void pillar_rows_kernel(RowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                                                   // [OW][RW_LD]
  int* lists = reinterpret_cast<int*>(smem + (size_t)a.OW * RW_LD);     // [3][2][RW_MAXLIST] column indices
  int* cnt = lists + 3 * 2 * RW_MAXLIST;                                // [3][2]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int go = blockIdx.x;                   // global output row b OH + oy
  const int b = go / a.OH, oy = go - b * a.OH;

  // ---- zero the row tile, build the six lists (waves 0..2: one canvas row each; their run bounds are requested before the fill)
  int run_lo = 0, run_hi = 0;
  if (wv < 3) {
    const int iy = 2 * oy - 1 + wv;
    if (iy >= 0 && iy < a.T) {
      run_lo = a.row_start[b * a.T + iy];
      run_hi = a.row_start[b * a.T + iy + 1];
    }
  }
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4* t4 = reinterpret_cast<f32x4*>(tile);
    for (int i = tid; i < a.OW * RW_LD / 4; i += 512) t4[i] = z;
  }
  if (wv < 3) {
    const int iy = 2 * oy - 1 + wv;
    int ne = 0, no = 0;
    if (iy >= 0 && iy < a.T) {
      const int gi = b * a.T + iy;
      const int lo = min(run_lo, a.v_cap), hi = min(run_hi, a.v_cap);
      const uint32_t key0 = (uint32_t)gi * (uint32_t)a.R;
      int* le = lists + (wv * 2 + 0) * RW_MAXLIST;
      int* lo_ = lists + (wv * 2 + 1) * RW_MAXLIST;
      for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        const bool live = i < hi;
        const int ix = live ? (int)(a.keys[i] - key0) : 0;
        const bool odd = live && (ix & 1), even = live && !(ix & 1);
        const unsigned long long me = __ballot(even), mo = __ballot(odd);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (even) le[ne + __popcll(me & below)] = ix;
        if (odd) lo_[no + __popcll(mo & below)] = ix;
        ne += __popcll(me);
        no += __popcll(mo);
      }
    }
    if (lane == 0) {
      cnt[wv * 2] = ne;
      cnt[wv * 2 + 1] = no;
    }
  }
  __syncthreads();

  // ---- the (tile, tap) chains of this wave's 16 output channels
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.canvas), 0, a.canvas_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const int j = lane & 15, kq = lane >> 4;
  const int c0 = wv * 16;
  const bool cols_live = c0 < a.cout;
  const unsigned w_lane = (unsigned)(((c0 + j) * 4 + kq) * 16);                 // byte offset of this lane's fragment inside one (tap, s) slab
  const unsigned w_slab = (unsigned)a.cout_pad * 64u;                            // bytes of one (tap, s) slab
  auto load_w = [&](int tap, f32x4 (&f)[NS]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      f[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (cols_live && !(PN_ROWS_EXP & 4)) ? w_lane : 0xffffffffu, (unsigned)(tap * NS + s) * w_slab, 0));
  };
  auto add_to_tile = [&](int ox, const f32x4 acc) __attribute__((always_inline)) {
    f32x4* p = reinterpret_cast<f32x4*>(tile + (size_t)ox * RW_LD + c0 + 4 * kq);
    *p = *p + acc;
  };
  // The three kernel rows one after the other (unrolled: the tap fragments live in registers).  Weight loads never wait in front of the
  // matrix work: w1 (even columns' tap) of the FIRST row is requested up front, w0 / w2 (odd columns' taps) at the start of a row -- they land
  // behind the even tiles' chains --, and w1 of the NEXT row as soon as the even tiles are done, behind the odd tiles' chains.
  f32x4 w0[NS], w1[NS], w2[NS];
  auto row_live = [&](int kh) { const int iy = 2 * oy - 1 + kh; return iy >= 0 && iy < a.T && !(PN_ROWS_EXP & 1); };
  if (row_live(0)) load_w(1, w1);
  else load_w(3 + 1, w1);
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    if (!row_live(kh)) continue;                                        // (block-uniform; only kh = 0 of the first output row)
    const int iy = 2 * oy - 1 + kh;
    const unsigned row_pix = (unsigned)(b * a.T + iy) * (unsigned)a.R;
    const int ne = cnt[kh * 2], no = cnt[kh * 2 + 1];
    const int te = (ne + 15) >> 4, nu = te + ((no + 15) >> 4);         // units: the even list's tiles, then the odd list's
    const int* le = lists + (kh * 2 + 0) * RW_MAXLIST;
    const int* lo_ = lists + (kh * 2 + 1) * RW_MAXLIST;
    // unit u -> (column of this lane's pillar, or -1)
    auto unit_ix = [&](int u) {
      const bool odd = u >= te;
      const int idx = (odd ? u - te : u) * 16 + j;
      return idx < (odd ? no : ne) ? (odd ? lo_[idx] : le[idx]) : -1;
    };
    auto request = [&](int ix, f32x4 (&xf)[NS]) __attribute__((always_inline)) {
      const unsigned voff = (ix >= 0 && !(PN_ROWS_EXP & 4)) ? ((row_pix + (unsigned)ix) * (unsigned)a.in_ps + (unsigned)a.in_co + 4u * (unsigned)kq) * 4u : 0xffffffffu;
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) xf[s2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_c, voff, (unsigned)s2 * 64u, 0));
    };
    // one unit: the next unit's canvas rows are requested BEFORE this unit's chains (a chain is 32 NS cycles of matrix work: the loads land
    // behind it); an even tile is one tap as two interleaved half chains (a single accumulator would wait 40 cycles per 32-cycle MFMA), an
    // odd tile its two taps interleaved on the same B fragments
    auto unit = [&](int u, int ix, const f32x4 (&xf)[NS], int ix_next, f32x4 (&xn)[NS]) __attribute__((always_inline)) {
      if (u + 1 < nu) request(ix_next, xn);
      f32x4 p = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
      if (u < te) {
#pragma unroll
        for (int s2 = 0; s2 < ((PN_ROWS_EXP & 8) ? 0 : NS); s2 += 2)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s2][m], xf[s2][m], p, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s2 + 1][m], xf[s2 + 1][m], q, 0, 0, 0);
          }
        if (ix >= 0) add_to_tile(ix >> 1, p + q);
        if (u + 1 == te && kh < 2) load_w((kh + 1) * 3 + 1, w1);       // the even tiles are done: the next row's w1 (block-uniform branch)
      } else {
#pragma unroll
        for (int s2 = 0; s2 < ((PN_ROWS_EXP & 8) ? 0 : NS); ++s2)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            p = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[s2][m], xf[s2][m], p, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s2][m], xf[s2][m], q, 0, 0, 0);
          }
        if (ix >= 0 && ((ix + 1) >> 1) < a.OW) add_to_tile((ix + 1) >> 1, p);      // kw = 0
        if (ix >= 0) add_to_tile((ix - 1) >> 1, q);                                    // kw = 2
      }
    };
    f32x4 xa[NS], xb[NS];
    int ixa = nu > 0 ? unit_ix(0) : -1, ixb = -1;
    if (nu > 0) request(ixa, xa);
    if (no > 0) {
      load_w(kh * 3 + 0, w0);
      load_w(kh * 3 + 2, w2);
    }
    if (te == 0 && kh < 2) load_w((kh + 1) * 3 + 1, w1);                // no even tile in this row: nothing reads w1 any more
#pragma unroll 1
    for (int u = 0; u < nu; u += 2) {
      ixb = u + 1 < nu ? unit_ix(u + 1) : -1;
      unit(u, ixa, xa, ixb, xb);
      if (u + 1 < nu) {
        ixa = u + 2 < nu ? unit_ix(u + 2) : -1;
        unit(u + 1, ixb, xb, ixa, xa);
      }
    }
  }
  __syncthreads();

  // ---- epilogue from the row tile
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const int c4n = a.cout >> 2;
  if (PN_ROWS_EXP & 2) return;
  if (a.planes) {
    // item = (channel quad c4, pixel quad xq), xq fastest: a wave's plane stores are whole runs of a plane row
    for (int it = tid; it < c4n * a.Wq; it += 512) {
      const int c4 = it / a.Wq, xq = it - c4 * a.Wq;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = z;
      if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + c4 * 4);
      if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + c4 * 4);
      f32x4 d[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int ox = 4 * xq - 1 + i;
        if (ox < 0 || ox >= a.OW) {
          d[i] = z;
        } else {
          const f32x4 s = *reinterpret_cast<const f32x4*>(tile + (size_t)ox * RW_LD + c4 * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c) d[i][c] = pn::apply_act(fmaf(s[c], sc[c], sh[c]), a.act);
        }
      }
      f32x4 vv[6];
      pn::wino4_input_transform4(d, vv);
      pn::wino4_store_planes(a.planes, vv, c4, c4n, a.plane_floats, b, oy, xq, a.OH, a.Wq);
    }
  }
  if (a.out) {
    // item = (pixel, channel quad), channels fastest: whole 16-byte runs of an NHWC pixel
    float* orow = a.out + ((size_t)go * a.OW) * a.out_ps + a.out_co;
    for (int it = tid; it < a.OW * c4n; it += 512) {
      const int ox = it / c4n, c4 = it - ox * c4n;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = z;
      if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + c4 * 4);
      if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + c4 * 4);
      const f32x4 s = *reinterpret_cast<const f32x4*>(tile + (size_t)ox * RW_LD + c4 * 4);
      f32x4 y;
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = pn::apply_act(fmaf(s[c], sc[c], sh[c]), a.act);
      *reinterpret_cast<f32x4*>(orow + (size_t)ox * a.out_ps + c4 * 4) = y;
    }
  }
}

// torch (Cout, Cin, 3, 3) -> [tap][s Cin / 16][cout_pad][kq 4][m 4]: element = w[cout][16 s + 4 kq + m][tap]
__global__ void pack_pillar_rows_weight_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i & 3), kq = (int)((i >> 2) & 3);
    size_t r = i >> 4;
    const int co = (int)(r % cout_pad);
    r /= cout_pad;
    const int ns = cin / 16;
    const int s = (int)(r % ns), tap = (int)(r / ns);
    const int ci = 16 * s + 4 * kq + m;
    packed[i] = co < cout ? w[((size_t)co * cin + ci) * 9 + tap] : 0.f;
  }
}

static int rows_cout_pad(int cout) { return pn::cdiv(cout, 16) * 16; }

}  // namespace

extern "C" {

// the row-band form covers: stride 2 on a (h, w) canvas with w % 8 == 0, w <= 512, output width a multiple of 64 quads' worth (ow / 4 in
// {16, 32, 64}), cin in {32, 64, 128}, cout <= 128 and a multiple of 4
int pn_pillar_conv_rows_supported(int batch, int h, int w, int cin, int cout, int stride) {
  if (stride != 2 || batch < 1 || h < 2 || w < 8 || (w & 7) || w > 2 * RW_MAXLIST) return 0;
  const int ow = (w - 1) / 2 + 1, wq = ow / 4;
  if (ow % 4 || !(wq == 16 || wq == 32 || wq == 64)) return 0;
  if (!(cin == 32 || cin == 64 || cin == 128) || cout < 4 || cout > 128 || (cout & 3)) return 0;
  return 1;
}

size_t pn_pillar_conv_rows_packed_weight_floats(int cout, int cin) { return (size_t)9 * cin * rows_cout_pad(cout); }

int pn_pack_pillar_conv_rows_weight_f32(const float* w_oihw, int cout, int cin, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_oihw && packed && cout >= 1 && cin >= 16 && cin % 16 == 0, "pack_pillar_conv_rows_weight: bad arguments");
  const size_t total = pn_pillar_conv_rows_packed_weight_floats(cout, cin);
  hipLaunchKernelGGL(pack_pillar_rows_weight_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w_oihw, cout, cin,
                     rows_cout_pad(cout), packed, total);
  return pn::check_launch("pack_pillar_rows_weight_kernel");
}

// canvas (batch, h, w, in_pixel_stride) NHWC with the frame's pillars in their cells; unq_keys: the frame's cell keys in ascending order
// ((b h + y) w + x); row_start [batch h + 1]: pillars in canvas rows before row g (pn_voxel_index_fused_rows_f32 leaves it); v_capacity bounds
// both.  Output: planes (pn_wino4_planes_floats(batch, oh, ow, cout) floats, not transposed) and / or the NHWC map `out`.
int pn_pillar_conv3x3_rows_f32(const float* canvas, int batch, int h, int w, int cin, int in_pixel_stride, int in_channel_offset, const uint32_t* unq_keys,
                               const int32_t* row_start, int v_capacity, const float* packed_rows_w, int cout, const float* scale, const float* shift, int act,
                               float* planes, float* out, int out_pixel_stride, int out_channel_offset, pn_stream_t stream) {
  PN_REQUIRE(canvas && unq_keys && row_start && packed_rows_w && (planes || out), "pillar_conv_rows: null pointer");
  PN_REQUIRE(pn_pillar_conv_rows_supported(batch, h, w, cin, cout, 2), "pillar_conv_rows: shape not covered (see pn_pillar_conv_rows_supported)");
  PN_REQUIRE(in_pixel_stride % 4 == 0 && in_channel_offset % 4 == 0 && in_channel_offset + cin <= in_pixel_stride && ((uintptr_t)canvas & 15) == 0,
             "pillar_conv_rows: the canvas slice must be 16-byte aligned");
  PN_REQUIRE(!out || (out_pixel_stride % 4 == 0 && out_channel_offset % 4 == 0 && out_channel_offset + cout <= out_pixel_stride && ((uintptr_t)out & 15) == 0),
             "pillar_conv_rows: the output slice must be 16-byte aligned");
  PN_REQUIRE(((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0 && ((uintptr_t)planes & 15) == 0, "pillar_conv_rows: 16-byte aligned tables");
  const unsigned long long cb = (unsigned long long)batch * h * w * in_pixel_stride * 4ull;
  PN_REQUIRE(cb < (1ull << 32), "pillar_conv_rows: canvases of 4 GiB or more are not addressable by the buffer descriptor");
  RowsArgs a{};
  a.canvas = canvas; a.keys = unq_keys; a.row_start = row_start; a.w = packed_rows_w; a.scale = scale; a.shift = shift; a.planes = planes; a.out = out;
  a.B = batch; a.T = h; a.R = w; a.OH = (h - 1) / 2 + 1; a.OW = (w - 1) / 2 + 1; a.Wq = a.OW / 4;
  a.in_ps = in_pixel_stride; a.in_co = in_channel_offset; a.act = act; a.cout = cout; a.cout_pad = rows_cout_pad(cout);
  a.out_ps = out_pixel_stride; a.out_co = out_channel_offset; a.v_cap = v_capacity;
  a.plane_floats = (unsigned)((size_t)batch * (a.OH + 2) * a.Wq * 4);
  a.canvas_bytes = (unsigned)cb;
  a.w_bytes = (unsigned)(pn_pillar_conv_rows_packed_weight_floats(cout, cin) * 4);
  const size_t smem = ((size_t)a.OW * RW_LD + 3 * 2 * RW_MAXLIST + 8) * sizeof(float);
  pn::ProfileSlot slot;
  const bool prof = pn::take_profile_slot(slot);
  hipStream_t st = pn::S(stream);
  const dim3 grid((unsigned)(batch * a.OH));
#define PN_ROWS_LAUNCH(NS)                                                                                                                  \
  do {                                                                                                                                      \
    static bool done[64] = {false};                                                                                                         \
    if (pn::first_use_on_device(done))                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pillar_rows_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    if (prof) hipExtLaunchKernelGGL(pillar_rows_kernel<NS>, grid, dim3(512), smem, st, slot.start, slot.stop, 0, a);                         \
    else hipLaunchKernelGGL(pillar_rows_kernel<NS>, grid, dim3(512), smem, st, a);                                                          \
  } while (0)
  if (cin == 128) PN_ROWS_LAUNCH(8);
  else if (cin == 64) PN_ROWS_LAUNCH(4);
  else PN_ROWS_LAUNCH(2);
#undef PN_ROWS_LAUNCH
  return pn::check_launch("pillar_rows_kernel");
}

}  // extern "C"
