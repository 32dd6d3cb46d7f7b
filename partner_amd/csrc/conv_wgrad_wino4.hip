// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions in the F(4, 3) width-Winograd domain (the backward counterpart of
// conv_wino4.hip; reference: the autograd of Conv2d in the RPN blocks, det3d/models/necks/rpn.py:124-142 under
// det3d/torchie/trainer/trainer.py:275-300 loss.backward()).
//
//   forward (per kernel row kh, input channel, output channel):  y[4t..4t+3] = A^T [ (G g) . (B^T d) ]     (d = 6 input pixels)
//   so   dL/dg = G^T [ (B^T d) . (A dy) ]   summed over all quads t, rows and samples:
//        V_q = (B^T d)_q  -- the forward's input transform           Z = A dy:  z0 = dy0,  z1 = dy0 + dy1 + dy2 + dy3,
//        z2 = dy0 - dy1 + dy2 - dy3,  z3 = dy0 + 2 dy1 + 4 dy2 + 8 dy3,  z4 = dy0 - 2 dy1 + 4 dy2 - 8 dy3,  z5 = dy3
//        dU[kh][q][ci][co] = sum_quads V_q[ci] Z_q[co]               -- SIX GEMMs per kernel row with K = quads (instead of twelve
//        products per quad for the three taps: half the MFMA work of the direct weight-gradient kernel)
//        dW[kh][0] = dU0/4 - dU1/6 - dU2/6 + dU3/24 + dU4/24,  dW[kh][1] = -dU1/6 + dU2/6 + dU3/12 - dU4/12,
//        dW[kh][2] = -dU1/6 - dU2/6 + dU3/6 + dU4/6 + dU5.
// wgrad_wino4_kernel: block = (slice of the quads, kernel row kh, position triple {0,1,2} | {3,4,5}, 128 x 128 (ci, co) tile); 8 waves,
// wave (m, n) accumulates ci rows 32 m .. 32 m + 31 x co columns 64 n .. 64 n + 63 for its three positions (6 accumulator tiles).
// A K step = 16 quads: every thread transforms one (quad, 4 input channels) and one (quad, 4 output channels) item in registers and
// stores the three V / three Z rows of its triple to LDS ([position][quad][160]: the row stride puts the two quads a wave reads at
// once 32 banks apart); the MFMA's contraction index is the quad, so both operands are 32-bit column reads of those rows.
// wgrad_wino4_reduce_kernel: slices summed in fixed order, G^T applied, dW written in torch layout (Cout, Cin, 3, 3).
#include "pn_common.h"
#include <algorithm>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_KQ = 16;        // quads per K step
constexpr int WG_LD = 160;       // floats per (position, quad) row in LDS: 128 channels + 32 (bank shift between the two rows of a read)
constexpr size_t WG_SMEM = 2 * 2 * (size_t)3 * WG_KQ * WG_LD * sizeof(float);   // two stages x (V, Z)

struct WgArgs {
  const float* x;
  const float* dy;
  float* part;              // [slice][kh 3][q 6][Cin][Cout]
  int B, H, W, Cin, Cout;
  int x_ps, x_co, dy_ps, dy_co;
  int qpr, total_quads, qps, nslices, co_tiles;
};

__global__ __launch_bounds__(512) void wgrad_wino4_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ST = 2 * 3 * WG_KQ * WG_LD;       // floats per stage: V then Z
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  const int slice = blockIdx.x;
  const int kh = blockIdx.y >> 1, half = blockIdx.y & 1;
  const int ci0 = (blockIdx.z / a.co_tiles) * 128, co0 = (blockIdx.z % a.co_tiles) * 128;
  const int q_begin = slice * a.qps, q_end = min(q_begin + a.qps, a.total_quads);
  const int nsteps = (q_end - q_begin + WG_KQ - 1) / WG_KQ;
  if (nsteps <= 0) return;
  const int ql = tid >> 5, c4 = tid & 31;        // loader item: quad of the step, 16-byte channel group

  f32x4 rx[6], rd[4];
  // this thread's quad of the step being loaded, as (sample, row, quad of the row): advanced by 16 quads per step without divisions
  int lg = q_begin + ql, ltq, lrow;
  {
    const int rowi = lg / a.qpr;
    ltq = lg - rowi * a.qpr;
    lrow = rowi;                 // b * H + oy
  }
  auto load_step = [&](int) {
    const bool ok = lg < q_end;
    const int tq = ltq;
    const int b = lrow / a.H, oy = lrow - b * a.H;
    const int iy = oy + kh - 1;
    const bool xok = ok && (unsigned)iy < (unsigned)a.H && ci0 + c4 * 4 < a.Cin;
    const float* xp = a.x + ((size_t)(b * a.H + (xok ? iy : 0)) * a.W + 4 * tq - 1) * a.x_ps + a.x_co + ci0 + c4 * 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bool cok = xok && (j > 0 || tq > 0) && (j < 5 || tq + 1 < a.qpr);
      rx[j] = cok ? *reinterpret_cast<const f32x4*>(xp + (size_t)j * a.x_ps) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool dok = ok && co0 + c4 * 4 < a.Cout;
    const float* dp = a.dy + ((size_t)(b * a.H + oy) * a.W + 4 * tq) * a.dy_ps + a.dy_co + co0 + c4 * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) rd[j] = dok ? *reinterpret_cast<const f32x4*>(dp + (size_t)j * a.dy_ps) : f32x4{0.f, 0.f, 0.f, 0.f};
    lg += WG_KQ;
    ltq += WG_KQ;
    while (ltq >= a.qpr) { ltq -= a.qpr; ++lrow; }
  };
  auto store_step = [&](int buf) {
    float* Vs = smem + buf * ST + ql * WG_LD + c4 * 4;
    float* Zs = Vs + 3 * WG_KQ * WG_LD;
    const f32x4 d0 = rx[0], d1 = rx[1], d2 = rx[2], d3 = rx[3], d4 = rx[4], d5 = rx[5];
    const f32x4 y0 = rd[0], y1 = rd[1], y2 = rd[2], y3 = rd[3];
    // fused multiply-adds on whole vectors (v_pk_fma_f32): the build runs with -ffp-contract=off, and on this chip a VALU instruction is
    // paid in MFMA time (r2: separate multiplies and adds, ~2.5x the instructions)
    const f32x4 c4v = {4.f, 4.f, 4.f, 4.f}, m4v = {-4.f, -4.f, -4.f, -4.f}, m5v = {-5.f, -5.f, -5.f, -5.f}, c2v = {2.f, 2.f, 2.f, 2.f}, c8v = {8.f, 8.f, 8.f, 8.f};
    if (half == 0) {
      const f32x4 e = __builtin_elementwise_fma(m4v, d2, d4), o = __builtin_elementwise_fma(m4v, d1, d3);
      *reinterpret_cast<f32x4*>(Vs + 0 * WG_KQ * WG_LD) = __builtin_elementwise_fma(c4v, d0, __builtin_elementwise_fma(m5v, d2, d4));
      *reinterpret_cast<f32x4*>(Vs + 1 * WG_KQ * WG_LD) = e + o;
      *reinterpret_cast<f32x4*>(Vs + 2 * WG_KQ * WG_LD) = e - o;
      const f32x4 se = y0 + y2, so = y1 + y3;
      *reinterpret_cast<f32x4*>(Zs + 0 * WG_KQ * WG_LD) = y0;
      *reinterpret_cast<f32x4*>(Zs + 1 * WG_KQ * WG_LD) = se + so;
      *reinterpret_cast<f32x4*>(Zs + 2 * WG_KQ * WG_LD) = se - so;
    } else {
      const f32x4 f = d4 - d2, t = d3 - d1;
      *reinterpret_cast<f32x4*>(Vs + 0 * WG_KQ * WG_LD) = __builtin_elementwise_fma(c2v, t, f);
      *reinterpret_cast<f32x4*>(Vs + 1 * WG_KQ * WG_LD) = __builtin_elementwise_fma(-c2v, t, f);
      *reinterpret_cast<f32x4*>(Vs + 2 * WG_KQ * WG_LD) = __builtin_elementwise_fma(c4v, d1, __builtin_elementwise_fma(m5v, d3, d5));
      const f32x4 se = __builtin_elementwise_fma(c4v, y2, y0), so = __builtin_elementwise_fma(c8v, y3, y1 + y1);
      *reinterpret_cast<f32x4*>(Zs + 0 * WG_KQ * WG_LD) = se + so;
      *reinterpret_cast<f32x4*>(Zs + 1 * WG_KQ * WG_LD) = se - so;
      *reinterpret_cast<f32x4*>(Zs + 2 * WG_KQ * WG_LD) = y3;
    }
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][j][r] = 0.f;

  load_step(0);
  store_step(0);
  __syncthreads();
  const int a_off = lh * WG_LD + 32 * wm + li;
  const int b_off = 3 * WG_KQ * WG_LD + lh * WG_LD + 64 * wn + li;
  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    if (t + 1 < nsteps) load_step(t + 1);                 // in flight during this step's MFMAs
    const float* S = smem + buf * ST;
    auto mfma_range = [&](int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
      for (int kp = k0; kp < k1; ++kp) {
        float av[3], bv[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          av[p] = S[a_off + (p * WG_KQ + 2 * kp) * WG_LD];
          bv[p][0] = S[b_off + (p * WG_KQ + 2 * kp) * WG_LD];
          bv[p][1] = S[b_off + (p * WG_KQ + 2 * kp) * WG_LD + 32];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p], bv[p][j], acc[p][j], 0, 0, 0);
      }
    };
    // the transform + LDS stores of the next step sit between the two halves of this step's MFMAs: issued behind the first 24, they run
    // while the matrix pipe works those off.  (PMC on the 256 x 256 layer: MFMA busy 0.62, no LDS bank conflicts; the loop is bound by
    // its barrier per 48 MFMAs -- with the loads switched off 398 of 490 us remain -- not by the staging VALU work or the divisions.)
    mfma_range(0, WG_KQ / 4);
    if (t + 1 < nsteps) store_step(buf ^ 1);
    mfma_range(WG_KQ / 4, WG_KQ / 2);
    __syncthreads();
  }
  // partial sums of this slice: [slice][kh][q][ci][co]
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int q = half * 3 + p;
    float* P = a.part + (((size_t)slice * 3 + kh) * 6 + q) * a.Cin * a.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = co0 + 64 * wn + 32 * j + li;
      if (co >= a.Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < a.Cin) P[(size_t)ci * a.Cout + co] = acc[p][j][r];
      }
    }
  }
}

__global__ void wgrad_wino4_reduce_kernel(const float* __restrict__ part, int nslices, int cin, int cout, float* __restrict__ dw, int accumulate) {
  const size_t per = (size_t)cin * cout;
  const size_t total = 3 * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kh = (int)(i / per);
    const size_t r = i - (size_t)kh * per;
    const int ci = (int)(r / cout), co = (int)(r - (size_t)ci * cout);
    float u[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nslices; ++s) {                  // fixed order: deterministic
      const float* P = part + (((size_t)s * 3 + kh) * 6) * per + r;
#pragma unroll
      for (int q = 0; q < 6; ++q) u[q] += P[(size_t)q * per];
    }
    const float g0 = u[0] * 0.25f - (u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 24.f);
    const float g1 = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
    const float g2 = (u[3] + u[4] - u[1] - u[2]) * (1.f / 6.f) + u[5];
    float* o = dw + (((size_t)co * cin + ci) * 3 + kh) * 3;
    if (accumulate) { o[0] += g0; o[1] += g1; o[2] += g2; }
    else { o[0] = g0; o[1] = g1; o[2] = g2; }
  }
}

struct Plan { int qps, nslices, ci_tiles, co_tiles; };
Plan plan(const pn_conv_desc* d) {
  Plan p;
  p.ci_tiles = pn::cdiv(d->cin, 128);
  p.co_tiles = pn::cdiv(d->cout, 128);
  const int total = d->batch * d->in_h * (d->in_w / 4);
  int n = std::max(1, 256 / (6 * p.ci_tiles * p.co_tiles));          // one round of blocks on 256 CUs
  n = std::min(n, std::max(1, total / (4 * WG_KQ)));                  // at least four K steps per slice
  p.qps = pn::cdiv(pn::cdiv(total, n), WG_KQ) * WG_KQ;
  p.nslices = pn::cdiv(total, p.qps);
  return p;
}

}  // namespace

extern "C" {

size_t pn_conv2d_wgrad_wino4_workspace_bytes(const pn_conv_desc* d) {
  if (!d || d->in_w % 4) return 0;
  const Plan p = plan(d);
  return (size_t)p.nslices * 18 * d->cin * d->cout * sizeof(float) + 256;
}

int pn_conv2d_wgrad_wino4_f32(const pn_conv_desc* d, const float* x, const float* dout, float* dw_oihw, int accumulate, void* workspace, size_t workspace_bytes,
                              pn_stream_t stream) {
  PN_REQUIRE(d && x && dout && dw_oihw && workspace, "wgrad_wino4: null pointer");
  PN_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_h == 1 && d->pad_w == 1 && d->groups == 1 && !d->deconv2x2 && d->range_strata <= 1,
             "wgrad_wino4: plain 3x3 / stride 1 / pad 1 convolutions only");
  PN_REQUIRE(d->batch >= 1 && d->in_h >= 1 && d->in_w >= 4 && d->in_w % 4 == 0, "wgrad_wino4: the map width must be a multiple of 4");
  PN_REQUIRE(d->cin % 4 == 0 && d->cout % 4 == 0 && d->in_pixel_stride % 4 == 0 && d->in_channel_offset % 4 == 0 && d->out_pixel_stride % 4 == 0 &&
                 d->out_channel_offset % 4 == 0,
             "wgrad_wino4: channel counts, pixel strides and channel offsets must be multiples of 4");
  PN_REQUIRE(d->in_pixel_stride >= d->in_channel_offset + d->cin && d->out_pixel_stride >= d->out_channel_offset + d->cout,
             "wgrad_wino4: channel slice does not fit the pixel stride");
  PN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)workspace & 15) == 0, "wgrad_wino4: pointers must be 16-byte aligned");
  if (workspace_bytes < pn_conv2d_wgrad_wino4_workspace_bytes(d)) return pn::fail(PN_ERR_WORKSPACE, "wgrad_wino4: workspace too small");
  const Plan p = plan(d);
  WgArgs a{};
  a.x = x; a.dy = dout; a.part = static_cast<float*>(workspace);
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w; a.Cin = d->cin; a.Cout = d->cout;
  a.x_ps = d->in_pixel_stride; a.x_co = d->in_channel_offset; a.dy_ps = d->out_pixel_stride; a.dy_co = d->out_channel_offset;
  a.qpr = d->in_w / 4; a.total_quads = d->batch * d->in_h * a.qpr; a.qps = p.qps; a.nslices = p.nslices; a.co_tiles = p.co_tiles;
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_wino4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_SMEM);
  hipStream_t st = pn::S(stream);
  hipLaunchKernelGGL(wgrad_wino4_kernel, dim3(p.nslices, 6, p.ci_tiles * p.co_tiles), dim3(512), WG_SMEM, st, a);
  const size_t total = (size_t)3 * d->cin * d->cout;
  hipLaunchKernelGGL(wgrad_wino4_reduce_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, st, a.part, p.nslices, d->cin, d->cout,
                     dw_oihw, accumulate);
  return pn::check_launch("wgrad_wino4 kernels");
}

}  // extern "C"
