// Token GEMMs (nn.Linear) of PARTNER's re-alignment attention and geometry-aware head on the gfx950 fp32 matrix pipe:
//   out[m][:N] = act(x[m][:K] @ W^T + bias) (+ residual[m][:N])
// Replaces every nn.Linear of SetBlock / SetAttention / Mlp (det3d/models/utils/set_transformer.py:37-53, 118-166, 216-259) and of
// the Swin stage of E2ESWVoteHead (det3d/models/bbox_heads/swin_utils/sw2votev4_util.py: qkv / proj / Mlp / PatchEmbed).
//
// Why not the convolution kernel (r2 ran these as 1x1 convolutions): the token matrices are [36 864 B x 256..1024] with K = 256 for
// most of the work, i.e. a 128 x 128 tile has only EIGHT K steps, and the 9.2 k .. 73 k rows meet 512 block slots in awkward
// ratios (576 tiles on 512 slots = two rounds, the second 12 % full).  This kernel is built for that regime:
//   * persistent blocks, XCD-local tile runs (N fastest: an M tile's x rows are fetched from HBM once per XCD, the <= 1 MB of
//     weights stay in every L2), and a tile shape picked per problem so that the last round is not mostly empty;
//   * the next tile's first two K steps of x are requested BEFORE the current tile's epilogue, which hides both the global
//     latency of the prologue and the store tail of the epilogue behind each other;
//   * the weight operand never touches LDS: the packed layout [K/4][Npad][4] IS the MFMA B-fragment layout, every lane fetches
//     its own 16 bytes from L2 one sub-step ahead; only x goes through LDS (register staged, [BM][32 + 4], double buffered);
//   * epilogue: accumulators -> wave-private LDS transpose -> 16-byte rows: bias, exact-erf GELU, residual and the store are
//     all float4 wide.
// Block = 4 waves as 2 x 2, wave tile (32 TM) x (32 TN), v_mfma_f32_32x32x2_f32: bit-exact fp32 FMA chains; k order inside a
// group of 8: MFMA j takes k = 8s + j from lane half 0 and k = 8s + 4 + j from lane half 1, for x and W alike.
#include "pn_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LK = 32;       // K step
constexpr int LA_LD = 36;    // floats per x row in LDS: 32 + 4 (conflict-free ds_read_b128 over 16 rows)
constexpr int LNPAD = 128;   // packed columns are padded to this

struct LinArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* out;
  int M, K, N, ldx, ldr, ldo, act;
  int npad, nsteps;
  int mtiles, ntiles;      // small-M form only
  int M1;                  // rows of the first phase
  unsigned x_bytes, w_bytes, r_bytes, o_bytes;
  // ---- LayerNorm folded around the GEMM (r6, pn_linear_ln_f32).  EX = 1: the input rows are LayerNorm(x) -- the caller passes the weight with
  // gamma multiplied in, its column sums and the bias with beta folded in; the epilogue applies rstd[m] (acc - mean[m] colsum[n]) from the
  // (sum, sum of squares) partials a producer left per row and 32-column group.  EX = 2: this launch IS such a producer: per row and
  // 32-column group of its own output (after activation and residual) it leaves (sum, sum of squares) in stat_out [M][N / 32][2].
  const float* ln_stat;    // EX = 1: [M][ln_parts][2]
  const float* ln_csum;    // EX = 1: [N]
  float* stat_out;         // EX = 2: [M][N / 32][2]
  int ln_parts;
  float ln_eps, ln_inv_k;
  unsigned ls_bytes, so_bytes;
#ifdef PN_LINEAR_STAMP
  unsigned long long* stamps;   // diagnostic build only (tools/micro/linear_stamps.hip): [block][tile < 4][8] shader-clock stamps of wave 0
#endif
};

#ifdef PN_LINEAR_STAMP
#define LIN_STAMP(k)                                                                                              \
  do {                                                                                                            \
    if (tid == 0 && tcount < 4) a.stamps[((size_t)blockIdx.x * 4 + tcount) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define LIN_STAMP(k) do { } while (0)
#endif

// exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)), in ~20 VALU instructions (the library erff is ~2x that, and on this chip VALU work is
// NOT hidden behind another wave's fp32 MFMAs: v_mfma_f32_32x32x2_f32 occupies the SIMD's fp32 lanes -- tools/micro/mfma_valu_coexec.hip
// measures the two as strictly additive -- so an epilogue instruction costs its full issue time).  erfc(z) = t exp(-z^2 + P(t)),
// t = 1 / (1 + z / 2), Chebyshev fit with fractional error < 1.2e-7 for all z >= 0 (Numerical Recipes, erfcc); 1 + erf(x / sqrt 2) is
// 2 - erfc for x >= 0 and erfc(|x| / sqrt 2) for x < 0, so nothing cancels: |gelu - exact| < 3e-7 max(|x|, 1).
__device__ __forceinline__ float gelu_erf(float x) {
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

// One PHASE of a launch: the rows [row_lo, row_hi) of the output in (64 TM) x (64 TN) tiles, persistent blocks, XCD-local tile runs.
template <int TM, int TN, bool GELU, int EX>
__device__ __forceinline__ void linear_phase(const LinArgs& a, const int row_lo, const int row_hi, float* smem) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int STAGE = BM * LA_LD;
  constexpr int APT = BM * 8 / 256;          // 16-byte chunks of x per thread and K step
  constexpr int TC = TN * 32, TLD = TC + 4;  // epilogue: one 32-row slab of the wave tile, row stride in floats
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1, li = lane & 31, lh = lane >> 5;

  // persistent blocks; XCD x owns a contiguous run of M tiles and walks it N-fastest
  const int mtiles = (row_hi - row_lo + BM - 1) / BM, ntiles = (a.N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int mq = mtiles >> 3, mr = mtiles & 7;
  const int mt0 = xcd < mr ? xcd * (mq + 1) : mr * (mq + 1) + (xcd - mr) * mq;
  const int xtiles = (xcd < mr ? mq + 1 : mq) * ntiles;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned np16 = (unsigned)a.npad * 16u;
  const int c4 = tid & 7, r0 = tid >> 3;       // loader: thread -> (row r0 + 32 j, 16-byte chunk c4)

  unsigned x_off[APT];     // byte offset of (row, chunk) or 0xffffffff past M
  unsigned b_off;          // byte offset of this lane's B fragment column block (sub-step 0 of K step 0)
  int m0 = 0, n0 = 0;
  auto setup = [&](int tl) {
    const int mt = mt0 + tl / ntiles;
    m0 = row_lo + mt * BM;
    n0 = (tl - (tl / ntiles) * ntiles) * BN;
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      const int row = m0 + r0 + 32 * j;
      x_off[j] = row < row_hi ? (unsigned)row * (unsigned)a.ldx * 4u + (unsigned)c4 * 16u : 0xffffffffu;
    }
    b_off = (unsigned)(lh * a.npad + n0 + wn * TC + li) * 16u;
  };

  f32x4 ra[2][APT];
  auto load_x = [&](int step, f32x4 (&r)[APT]) {
    const bool live = step < a.nsteps && step * LK + c4 * 4 < a.K;
    const unsigned so = (unsigned)step * LK * 4u;
#pragma unroll
    for (int j = 0; j < APT; ++j) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, live ? x_off[j] : 0xffffffffu, so, 0));
  };
  auto store_x = [&](int buf, const f32x4 (&r)[APT]) {
    float* As = smem + buf * STAGE + r0 * LA_LD + c4 * 4;
#pragma unroll
    for (int j = 0; j < APT; ++j) *reinterpret_cast<f32x4*>(As + 32 * j * LA_LD) = r[j];
  };
  // Operand fragments.  x: from LDS, one sub-step ahead.  W: straight from L2, a whole K STEP ahead (two sets of 4 sub-steps x TN
  // fragments, alternating by K step): the vector-memory counter returns in issue order, so a fragment requested AFTER the x loads of
  // step t+3 could not be used before those (HBM-latency) loads have landed -- the fragments of step t+1 are therefore requested in
  // sub-steps 0 / 1 of step t, BEFORE that step's x loads (sub-step 2), and what a wait on them drains is older than a K step.
  f32x4 af[2][TM], bf[2][4][TN];
  const int a_frag = (wm * TM * 32 + li) * LA_LD + lh * 4;
  auto read_a = [&](int buf, int s, f32x4 (&f)[TM]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f[i] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + a_frag + i * 32 * LA_LD + s * 8);
  };
  auto load_b = [&](int step, int s, f32x4 (&f)[TN]) {      // sub-step s of K step `step` (past the end: zeros)
    const unsigned vo = step < a.nsteps ? b_off : 0xffffffffu;
    const unsigned so = (unsigned)(step * 8 + 2 * s) * np16;
#pragma unroll
    for (int j = 0; j < TN; ++j) f[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vo, so + (unsigned)j * 512u, 0));
  };
  f32x16 acc[TM][TN];
  auto mfma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
  };
  constexpr int NM = 4 * TM * TN;   // MFMAs per sub-step
  // one K step (compile-time parity P: LDS stage and W fragment set): x of step t+1 (registers rx) goes to the other stage under
  // sub-step 1, x of step t+3 is requested under sub-step 2, the barrier sits before sub-step 3
  auto kstep = [&](auto par_c, int t, f32x4 (&rx)[APT]) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value;
    load_b(t + 1, 0, bf[P ^ 1][0]);
    load_b(t + 1, 1, bf[P ^ 1][1]);
    read_a(P, 1, af[1]);
    mfma(af[0], bf[P][0]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
    for (int k = 0; k < 2 * TN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_b(t + 1, 2, bf[P ^ 1][2]);
    load_b(t + 1, 3, bf[P ^ 1][3]);
    read_a(P, 2, af[0]);
    mfma(af[1], bf[P][1]);
    store_x(P ^ 1, rx);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 1);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
#pragma unroll
    for (int k = 0; k < 2 * TN; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
    }
#pragma unroll
    for (int k = 0; k < APT; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 1);
      __builtin_amdgcn_sched_group_barrier(0x008, (NM - 1 - 2 * TN) / APT > 0 ? (NM - 1 - 2 * TN) / APT : 1, 1);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_a(P, 3, af[1]);
    mfma(af[0], bf[P][2]);
    load_x(t + 3, rx);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 2);
#pragma unroll
    for (int k = 0; k < APT; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / APT > 0 ? NM / APT : 1, 2);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 2);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_a(P ^ 1, 0, af[0]);
    mfma(af[1], bf[P][3]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 3);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 3);
    __builtin_amdgcn_sched_barrier(0);
  };

  // epilogue of the tile at (em0, en0): 32-row slabs through this wave's private LDS region, then float4 rows.  The vector-memory
  // counter is shared by loads and stores and returns in order: a residual load issued after a store would have to wait for that
  // store's acknowledgement, so the residual of pass p+1 is requested BEFORE the store of pass p, and the bias (one float4 per lane
  // and tile: the lane's column group is the same in every pass) comes in with the tile's setup.  Everything is an unconditional
  // buffer access (rows past M / columns past N: offset 0xffffffff = dropped store, zero load): a load or store under a branch
  // makes the compiler fall back to s_waitcnt vmcnt(0) -- i.e. to one store round trip per pass (measured: 21 k cycles per tile).
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.r_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_ls = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ln_stat), 0, EX == 1 ? a.ls_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_so = __builtin_amdgcn_make_buffer_rsrc(a.stat_out, 0, EX == 2 ? a.so_bytes : 0u, 0x00020000);
  auto epilogue = [&](int em0, int en0, f32x4 bias4) {
    float* slab = smem + wv * (32 * TLD);
    constexpr int TC4 = TC / 4, NP = 32 * TC4 / 64, RPP = 64 / TC4;     // float4 per row, passes per slab, rows per pass
    const int q = lane % TC4, rrow = lane / TC4;
    const int n = en0 + wn * TC + q * 4;
    const bool nok = n < a.N;
    const bool relu = a.act == PN_ACT_RELU;
    f32x4 csum4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EX == 1) {
      if (nok) csum4 = *reinterpret_cast<const f32x4*>(a.ln_csum + n);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + j * 32 + li] = acc[i][j][r];
      const int mb = em0 + wm * TM * 32 + i * 32 + rrow;
      // EX = 1: lane L finalises the LayerNorm statistics of slab row L & 31 (the partials of the row's 32-column groups, summed in
      // group order); the passes below fetch their row's pair by lane shuffles
      float ln_mean = 0.f, ln_rstd = 1.f;
      if constexpr (EX == 1) {
        const int srow = em0 + wm * TM * 32 + i * 32 + li;
        const unsigned so = srow < row_hi ? (unsigned)srow * (unsigned)a.ln_parts * 8u : 0xffffffffu;
        float s1 = 0.f, s2 = 0.f;
        for (int g = 0; g < a.ln_parts; g += 2) {      // (ln_parts is even: K a multiple of 64)
          const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_ls, so, (unsigned)g * 8u, 0));
          s1 = (s1 + t[0]) + t[2];
          s2 = (s2 + t[1]) + t[3];
        }
        ln_mean = s1 * a.ln_inv_k;
        const float var = fmaxf(fmaf(-ln_mean, ln_mean, s2 * a.ln_inv_k), 0.f);
        ln_rstd = 1.0f / sqrtf(var + a.ln_eps);
      }
      // (no short-circuit: a && here becomes a divergent branch around the access)
      auto roff = [&](int p) { const bool ok = nok & (mb + p * RPP < row_hi); return ok ? ((unsigned)(mb + p * RPP) * (unsigned)a.ldr + (unsigned)n) * 4u : 0xffffffffu; };
      auto ooff = [&](int p) { const bool ok = nok & (mb + p * RPP < row_hi); return ok ? ((unsigned)(mb + p * RPP) * (unsigned)a.ldo + (unsigned)n) * 4u : 0xffffffffu; };
      f32x4 rcur = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, roff(0), 0, 0));
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f32x4 rnext = {0.f, 0.f, 0.f, 0.f};
        if (p + 1 < NP) rnext = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, roff(p + 1), 0, 0));
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + (p * RPP + rrow) * TLD + q * 4);
        if constexpr (EX == 1) {
          const float mu = __shfl(ln_mean, p * RPP + rrow, 64), rs = __shfl(ln_rstd, p * RPP + rrow, 64);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(-mu, csum4[e], v[e]) * rs;
        }
        v += bias4;
        if constexpr (GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);      // (a vector-wide v_pk_fma form of the polynomial measured the same)
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu ? fmaxf(v[e], 0.f) : v[e];      // NaN stays NaN when there is no activation
        }
        v += rcur;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_o, ooff(p), 0, 0);
        if constexpr (EX == 2) {
          // (sum, sum of squares) of this row's 32-column group: the group's eight lanes, folded by a fixed xor butterfly
          float s1 = (v[0] + v[1]) + (v[2] + v[3]);
          float s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          const bool ok = nok & (mb + p * RPP < row_hi) & ((q & 7) == 0);
          const unsigned off = ok ? ((unsigned)(mb + p * RPP) * (unsigned)(a.N >> 5) + (unsigned)(n >> 5)) * 8u : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, float2{s1, s2}), rsrc_so, off, 0, 0);
        }
        rcur = rnext;
      }
    }
  };

  int pm0 = -1, pn0 = 0;
  f32x4 pbias = {0.f, 0.f, 0.f, 0.f};
  int tcount = 0;
  (void)tcount;
  for (int tl = slot; tl < xtiles; tl += per_xcd, ++tcount) {
    LIN_STAMP(0);
    setup(tl);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    {
      const int n = n0 + wn * TC + (lane % (TC / 4)) * 4;
      if (a.bias && n < a.N) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) load_b(0, s4, bf[0][s4]);
    load_x(0, ra[0]);
    load_x(1, ra[1]);
    LIN_STAMP(1);
    if (pm0 >= 0) {
      epilogue(pm0, pn0, pbias);     // while this tile's first loads are in flight
      LIN_STAMP(2);
      __syncthreads();               // every wave's slab is read back before the stages are refilled
    }
    LIN_STAMP(3);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    store_x(0, ra[0]);
    load_x(2, ra[0]);
    __syncthreads();
    LIN_STAMP(4);
    read_a(0, 0, af[0]);
    for (int t = 0; t < a.nsteps; t += 2) {
      kstep(std::integral_constant<int, 0>{}, t, ra[1]);
      if (t + 1 < a.nsteps) kstep(std::integral_constant<int, 1>{}, t + 1, ra[0]);
    }
    LIN_STAMP(5);
    pm0 = m0;
    pn0 = n0;
    pbias = bias4;
  }
  if (pm0 >= 0) epilogue(pm0, pn0, pbias);
}

// A launch = the rows [0, M1) in (64 TM) x (64 TN) tiles -- a whole number of rounds of the persistent grid -- and, when the row count
// does not divide that way, the REST [M1, M) in smaller (64 TM2) x (64 TN2) tiles, so that the last round is not a handful of big
// tiles on an otherwise idle chip (73 728 x 256: 1152 big tiles on 512 slots were three rounds for 2.25 rounds of work).
template <int TM, int TN, int TM2, int TN2, int OCC, bool GELU, int EX>
__global__ __launch_bounds__(256, OCC) void linear_kernel(LinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  linear_phase<TM, TN, GELU, EX>(a, 0, a.M1, smem);
  if constexpr (TM2 > 0) {
    if (a.M1 < a.M) {
      __syncthreads();      // the first phase's last epilogue slabs are read back before the stages are refilled
      linear_phase<TM2, TN2, GELU, EX>(a, a.M1, a.M, smem);
    }
  }
}

// ---- small-M form (the key-point chains of the SetBlock: 1024 B rows, 2.5 % of the block's FLOPs but, launched as tiles, a third of
// its time: 64 tiles on 256 CUs, each a serial walk over K).  Block tile 32 x 32, the four waves split K: per 256-wide K chunk the
// block stages its 32 x rows through LDS with whole-row (1 KB) loads, wave ks multiplies k = 64 ks .. 64 ks + 63 of the chunk with
// its W fragments straight from L2 (all eight requested before the first is used), and the four partial tiles are joined through
// LDS in a fixed order (wave 0 + 1 + 2 + 3), every wave finishing a quarter of the rows.  The summation order differs from the
// tiled forms', so this form has its own entry point (pn_linear_ksplit_f32) and is never picked behind the caller's back.
constexpr int LS_CH = 256;              // K chunk
constexpr int LS_LD = LS_CH + 4;        // floats per x row in LDS (16-byte slots of 16 rows distinct: 65 i mod 16)
template <bool GELU>
__global__ __launch_bounds__(256, 2) void linear_small_kernel(LinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // two stages [32][LS_LD]; the join [4][16][64] reuses them
  const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int nt = blockIdx.x % a.ntiles, mt = blockIdx.x / a.ntiles;
  const int m0 = mt * 32, n0 = nt * 32;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned np16 = (unsigned)a.npad * 16u;
  const int nchunks = (a.K + LS_CH - 1) / LS_CH;
  // loader: wave w, step j -> row w + 4 j, lane -> 16 bytes at k = 4 lane of the chunk (one whole 1 KB row piece per wave instruction)
  unsigned xo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = m0 + ks + 4 * j;
    xo[j] = row < a.M ? (unsigned)row * (unsigned)a.ldx * 4u + (unsigned)lane * 16u : 0xffffffffu;
  }
  const unsigned bo = (unsigned)(lh * a.npad + n0 + li) * 16u;
  f32x4 rx[8], bf[8];
  auto load_x = [&](int c) {
    const bool live = c < nchunks && c * LS_CH + lane * 4 < a.K;
#pragma unroll
    for (int j = 0; j < 8; ++j) rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, live ? xo[j] : 0xffffffffu, (unsigned)c * LS_CH * 4u, 0));
  };
  // this wave's eight k groups of chunk c: k4 index = 64 c + 16 ks + 2 g + lh.  The packed buffer holds ceil(K / 32) * 8 such rows and the
  // scalar offset is NOT part of the descriptor's range check, so groups past them are redirected by the vector offset (they read zeros;
  // r3 read whatever followed the buffer for K not a multiple of 256 -- harmless only while x is zero there and the bytes are finite)
  const int krows = (a.K + LK - 1) / LK * 8;
  auto load_b = [&](int c) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int k4 = 64 * c + 16 * ks + 2 * g;
      bf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, k4 + lh < krows ? bo : 0xffffffffu, (unsigned)k4 * np16, 0));
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(smem + buf * 32 * LS_LD + (ks + 4 * j) * LS_LD + lane * 4) = rx[j];
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  load_b(0);
  load_x(0);
  store_x(0);
  load_x(1);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    f32x4 af[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) af[g] = *reinterpret_cast<const f32x4*>(smem + buf * 32 * LS_LD + li * LS_LD + 64 * ks + 8 * g + 4 * lh);
    f32x4 cb[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) cb[g] = bf[g];
    if (c + 1 < nchunks) {
      load_b(c + 1);
      store_x(buf ^ 1);       // chunk c + 1 (the other stage was last read in iteration c - 1, before that iteration's barrier)
      load_x(c + 2);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][kk], cb[g][kk], acc, 0, 0, 0);
    __syncthreads();
  }
  // join: every wave leaves its partial tile in LDS, wave d finishes accumulator registers 4 d .. 4 d + 3 (rows 8 d .. 8 d + 7)
  float* part = smem;      // [4][16][64]
#pragma unroll
  for (int r = 0; r < 16; ++r) part[(ks * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  const int n = n0 + li;
  const float b = (a.bias && n < a.N) ? a.bias[n] : 0.f;
  const bool relu = a.act == PN_ACT_RELU;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * ks + rr;
    float v = ((part[(0 * 16 + r) * 64 + lane] + part[(1 * 16 + r) * 64 + lane]) + part[(2 * 16 + r) * 64 + lane]) + part[(3 * 16 + r) * 64 + lane];
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    v += b;
    if constexpr (GELU) v = gelu_erf(v);
    else v = relu ? fmaxf(v, 0.f) : v;      // (a select, not fmaxf(v, -inf): that would turn a NaN accumulator into -inf)
    if (m < a.M && n < a.N) {
      if (a.res) v += a.res[(size_t)m * a.ldr + n];
      a.out[(size_t)m * a.ldo + n] = v;
    }
  }
}

// torch (N, K) -> [Kpad / 4][Npad][4], zero padded
__global__ void pack_linear_weight_kernel(const float* __restrict__ w, int n, int k, int npad, float* __restrict__ packed, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k1 = (int)(i & 3);
    const size_t r = i >> 2;
    const int col = (int)(r % npad);
    const int kk = (int)(r / npad) * 4 + k1;
    packed[i] = (col < n && kk < k) ? w[(size_t)col * k + kk] : 0.f;
  }
}

template <int TM, int TN, int TM2, int TN2, bool GELU, int EX>
int launch_linear_t(const LinArgs& a, int ncu, hipStream_t st, const pn::ProfileSlot* ps, int blocks_per_cu) {
  constexpr int OCC = 2;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  auto floats = [](int tm, int tn) { return std::max<size_t>(2 * (size_t)64 * tm * LA_LD, 4 * 32 * (size_t)(tn * 32 + 4)); };   // two x stages; the epilogue's slabs reuse them
  const size_t smem = std::max(floats(TM, TN), TM2 > 0 ? floats(TM2, TN2) : (size_t)0) * sizeof(float);
  static bool done[64] = {false};
  if (pn::first_use_on_device(done))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_kernel<TM, TN, TM2, TN2, OCC, GELU, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  long long tiles = (long long)pn::cdiv(a.M1, BM) * pn::cdiv(a.N, BN);
  if (TM2 > 0 && a.M1 < a.M) tiles = std::max(tiles, (long long)pn::cdiv(a.M - a.M1, 64 * std::max(TM2, 1)) * pn::cdiv(a.N, 64 * std::max(TN2, 1)));
  const dim3 grid((unsigned)std::min<long long>((long long)blocks_per_cu * ncu, (tiles + 7) / 8 * 8));
  if (ps) hipExtLaunchKernelGGL((linear_kernel<TM, TN, TM2, TN2, OCC, GELU, EX>), grid, dim3(256), smem, st, ps->start, ps->stop, 0, a);
  else hipLaunchKernelGGL((linear_kernel<TM, TN, TM2, TN2, OCC, GELU, EX>), grid, dim3(256), smem, st, a);
  return pn::check_launch("linear_kernel");
}

// the combinations the callers use: plain (any activation), LayerNorm-in (none / GELU), statistics-out (no activation)
template <int TM, int TN, int TM2, int TN2>
int launch_linear(const LinArgs& a, int ncu, hipStream_t st, const pn::ProfileSlot* ps, int blocks_per_cu = 2) {
  if (a.ln_stat)
    return a.act == PN_ACT_GELU ? launch_linear_t<TM, TN, TM2, TN2, true, 1>(a, ncu, st, ps, blocks_per_cu)
                                : launch_linear_t<TM, TN, TM2, TN2, false, 1>(a, ncu, st, ps, blocks_per_cu);
  if (a.stat_out) return launch_linear_t<TM, TN, TM2, TN2, false, 2>(a, ncu, st, ps, blocks_per_cu);
  return a.act == PN_ACT_GELU ? launch_linear_t<TM, TN, TM2, TN2, true, 0>(a, ncu, st, ps, blocks_per_cu)
                              : launch_linear_t<TM, TN, TM2, TN2, false, 0>(a, ncu, st, ps, blocks_per_cu);
}

}  // namespace

extern "C" {

size_t pn_linear_packed_weight_floats(int n, int k) {
  return (size_t)pn::cdiv(k, LK) * LK * (size_t)(pn::cdiv(n, LNPAD) * LNPAD);
}

int pn_pack_linear_weight_f32(const float* w_nk, int n, int k, float* packed, pn_stream_t stream) {
  PN_REQUIRE(w_nk && packed && n >= 1 && k >= 1, "pack_linear_weight: bad arguments");
  const size_t total = pn_linear_packed_weight_floats(n, k);
  hipLaunchKernelGGL(pack_linear_weight_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), w_nk, n, k,
                     pn::cdiv(n, LNPAD) * LNPAD, packed, total);
  return pn::check_launch("pack_linear_weight_kernel");
}

// The plan of a launch: form = 10 TM + TN of the main tiles (22 | 21 | 12 | 11), 1 = the K-split small-M form; rest = form of the second
// phase (0: none) and m1 = rows of the first.  Main tiles 128 x 128 whenever they fill at least one whole round of the 2-per-CU
// persistent grid: the rows of the whole rounds go to them, the remaining rows to the shape that needs the least time for them.
// PN_LINEAR_TILE / pn_linear_set_tile pin the main form (single phase).
struct LinPlan { int form, rest, m1; };
#ifdef PN_LINEAR_STAMP
unsigned long long* pn_linear_stamp_buffer = nullptr;    // set by the diagnostic tool before a launch
#endif
static int g_linear_tile = -1;     // -1: not read yet; 0: automatic
static double tile_cost(int tm, int tn, int k) {   // cycles of one round (two co-resident blocks, one tile each): MFMA time of both + the fixed part
  return 2.0 * (64.0 * tm * 64.0 * tn * k / 2048.0 * 64.0 / 4.0) * 1.06 + 5000.0 + (tm * tn < 4 ? 1500.0 : 0.0);
}
static int linear_tile_pin() {       // the pinned main form (0: automatic); reads PN_LINEAR_TILE once
  if (g_linear_tile < 0) { const char* e = getenv("PN_LINEAR_TILE"); g_linear_tile = e ? atoi(e) : 0; }
  return g_linear_tile;
}
static LinPlan linear_plan(int m, int n, int k, int ncu) {
  if (linear_tile_pin()) return {g_linear_tile, 0, m};
  // under half a round of big tiles: 64 x 64 tiles.  (The K-split form is never picked here: pn_linear_f32 keeps ONE summation order
  // -- the tiled forms all add in the order of the r2 convolution route -- and pn_linear_ksplit_f32 is the explicit other one.)
  if ((long long)pn::cdiv(m, 128) * pn::cdiv(n, 128) * 2 <= ncu) return {11, 0, m};
  const long long slots = 2LL * ncu;
  auto rounds_cost = [&](int rows, int tm, int tn) {
    const long long tiles = (long long)pn::cdiv(rows, 64 * tm) * pn::cdiv(n, 64 * tn);
    return (double)((tiles + slots - 1) / slots) * tile_cost(tm, tn, k);
  };
  const int nt = pn::cdiv(n, 128), mt = pn::cdiv(m, 128);
  const long long full = (long long)mt * nt / slots;             // whole rounds of 128 x 128 tiles
  LinPlan best{22, 0, m};
  double best_cost = rounds_cost(m, 2, 2);
  const int single[3][2] = {{2, 1}, {1, 2}, {1, 1}};
  for (auto& c : single) {
    const double v = rounds_cost(m, c[0], c[1]);
    if (v < best_cost) { best_cost = v; best = {c[0] * 10 + c[1], 0, m}; }
  }
  if (full >= 1) {
    const int m1 = (int)std::min<long long>(m, full * slots / nt * 128);
    if (m1 < m) {
      const int rest[2][2] = {{2, 1}, {1, 1}};
      for (auto& c : rest) {
        const double v = (double)full * tile_cost(2, 2, k) + rounds_cost(m - m1, c[0], c[1]);
        if (v < best_cost) { best_cost = v; best = {22, c[0] * 10 + c[1], m1}; }
      }
    }
  }
  return best;
}

int pn_linear_set_tile(int form) {
  PN_REQUIRE(form == 0 || form == 1 || form == 11 || form == 12 || form == 21 || form == 22, "linear_set_tile: 0 (automatic), 1, 11, 12, 21 or 22");
  g_linear_tile = form;
  return PN_OK;
}

enum { LIN_TILED = 0, LIN_KSPLIT = 1 };
static int linear_launch(const float* x, int m, int k, int ldx, const float* packed_w, int n, const float* bias, int act, const float* residual, int ldr,
                         float* out, int ldo, pn_stream_t stream, int mode, const float* ln_stats = nullptr, const float* ln_colsum = nullptr,
                         float ln_eps = 0.f, float* row_stats_out = nullptr) {
  const bool ksplit = mode == LIN_KSPLIT;
  PN_REQUIRE(!(ln_stats && row_stats_out), "linear_ln: a launch either consumes row statistics or produces them");
  PN_REQUIRE(!ln_stats || (ln_colsum && k % 64 == 0 && ((uintptr_t)ln_stats & 15) == 0 && ((uintptr_t)ln_colsum & 15) == 0 && !ksplit),
             "linear_ln: LayerNorm-in needs the column sums, k a multiple of 64 and 16-byte aligned tables");
  PN_REQUIRE(!row_stats_out || (n % 32 == 0 && act != PN_ACT_GELU && ((uintptr_t)row_stats_out & 7) == 0 && !ksplit),
             "linear_ln: statistics-out needs n a multiple of 32 and no GELU");
  PN_REQUIRE(x && packed_w && out && m > 0 && k > 0 && n > 0, "linear: bad arguments");
  PN_REQUIRE(k % 4 == 0 && ldx % 4 == 0 && ldx >= k && ldo >= n, "linear: k and the row strides must be multiples of 4");
  PN_REQUIRE(n % 4 == 0 && ldo % 4 == 0 && (residual == nullptr || (ldr >= n && ldr % 4 == 0)), "linear: n and the output / residual strides must be multiples of 4");
  PN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)packed_w & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)residual & 15) == 0 &&
                 ((uintptr_t)bias & 15) == 0, "linear: pointers must be 16-byte aligned");
  PN_REQUIRE(act == PN_ACT_NONE || act == PN_ACT_RELU || act == PN_ACT_GELU, "linear: activation none, ReLU or GELU");
  const unsigned long long xb = (unsigned long long)m * ldx * 4ull;
  const unsigned long long ob = (unsigned long long)m * ldo * 4ull, rb = residual ? (unsigned long long)m * ldr * 4ull : 0ull;
  PN_REQUIRE(xb < (1ull << 32) && ob < (1ull << 32) && rb < (1ull << 32), "linear: matrices of 4 GiB or more are not addressable by the buffer descriptors");
  LinArgs a{};
  a.x = x; a.w = packed_w; a.bias = bias; a.res = residual; a.out = out;
  a.M = m; a.K = k; a.N = n; a.ldx = ldx; a.ldr = ldr; a.ldo = ldo; a.act = act;
  a.npad = pn::cdiv(n, LNPAD) * LNPAD;
  a.nsteps = pn::cdiv(k, LK);
  a.x_bytes = (unsigned)xb;
  a.o_bytes = (unsigned)ob;
  a.r_bytes = (unsigned)rb;
#ifdef PN_LINEAR_STAMP
  a.stamps = pn_linear_stamp_buffer;
#endif
  a.w_bytes = (unsigned)(pn_linear_packed_weight_floats(n, k) * 4);
  a.ln_stat = ln_stats; a.ln_csum = ln_colsum; a.stat_out = row_stats_out;
  a.ln_parts = k / 32; a.ln_eps = ln_eps; a.ln_inv_k = 1.0f / (float)k;
  {
    const unsigned long long lsb = ln_stats ? (unsigned long long)m * (k / 32) * 8ull : 0ull, sob = row_stats_out ? (unsigned long long)m * (n / 32) * 8ull : 0ull;
    PN_REQUIRE(lsb < (1ull << 32) && sob < (1ull << 32), "linear_ln: statistics tables of 4 GiB or more");
    a.ls_bytes = (unsigned)lsb; a.so_bytes = (unsigned)sob;
  }
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && cus[dev] == 0) {
    int c = 0;
    cus[dev] = (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c >= 8) ? c / 8 * 8 : 256;
  }
  const int ncu = (dev >= 0 && dev < 64) ? cus[dev] : 256;
  pn::ProfileSlot slot;
  const bool prof = pn::take_profile_slot(slot);
  const pn::ProfileSlot* ps = prof ? &slot : nullptr;
  hipStream_t st = pn::S(stream);
  // (r3 fix: the pin used to be read inside linear_plan only, so the process's FIRST K-split call saw the "not read yet" marker as a pin
  // and ran a tiled form -- one call with the other summation order, enough to flip key points in the first frame of a process)
  const LinPlan plan = (ksplit && !linear_tile_pin()) ? LinPlan{1, 0, m} : linear_plan(m, n, k, ncu);
  a.M1 = plan.m1;
  PN_REQUIRE(plan.form != 1 || (!ln_stats && !row_stats_out), "linear_ln: not with the pinned K-split form");
  if (plan.form == 1) {
    a.mtiles = pn::cdiv(m, 32);
    a.ntiles = pn::cdiv(n, 32);
    const dim3 grid((unsigned)(a.mtiles * a.ntiles));
    const size_t smem = 2 * (size_t)32 * LS_LD * sizeof(float);
    static bool done[64] = {false};
    if (pn::first_use_on_device(done)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    }
    if (act == PN_ACT_GELU) {
      if (ps) hipExtLaunchKernelGGL(linear_small_kernel<true>, grid, dim3(256), smem, st, ps->start, ps->stop, 0, a);
      else hipLaunchKernelGGL(linear_small_kernel<true>, grid, dim3(256), smem, st, a);
    } else {
      if (ps) hipExtLaunchKernelGGL(linear_small_kernel<false>, grid, dim3(256), smem, st, ps->start, ps->stop, 0, a);
      else hipLaunchKernelGGL(linear_small_kernel<false>, grid, dim3(256), smem, st, a);
    }
    return pn::check_launch("linear_small_kernel");
  }
  switch (plan.form * 100 + plan.rest) {
    case 2221: return launch_linear<2, 2, 2, 1>(a, ncu, st, ps);
    case 2211: return launch_linear<2, 2, 1, 1>(a, ncu, st, ps);
    case 2100: return launch_linear<2, 1, 0, 0>(a, ncu, st, ps);
    case 1200: return launch_linear<1, 2, 0, 0>(a, ncu, st, ps);
    case 1100: return launch_linear<1, 1, 0, 0>(a, ncu, st, ps);
    default: return launch_linear<2, 2, 0, 0>(a, ncu, st, ps);
  }
}

int pn_linear_f32(const float* x, int m, int k, int ldx, const float* packed_w, int n, const float* bias, int act, const float* residual, int ldr,
                  float* out, int ldo, pn_stream_t stream) {
  return linear_launch(x, m, k, ldx, packed_w, n, bias, act, residual, ldr, out, ldo, stream, LIN_TILED);
}

int pn_linear_ksplit_f32(const float* x, int m, int k, int ldx, const float* packed_w, int n, const float* bias, int act, const float* residual,
                         int ldr, float* out, int ldo, pn_stream_t stream) {
  return linear_launch(x, m, k, ldx, packed_w, n, bias, act, residual, ldr, out, ldo, stream, LIN_KSPLIT);
}

int pn_linear_ln_f32(const float* x, int m, int k, int ldx, const float* packed_w, int n, const float* bias, int act, const float* residual, int ldr,
                     float* out, int ldo, const float* ln_stats, const float* ln_colsum, float ln_eps, float* row_stats_out, pn_stream_t stream) {
  return linear_launch(x, m, k, ldx, packed_w, n, bias, act, residual, ldr, out, ldo, stream, LIN_TILED, ln_stats, ln_colsum, ln_eps, row_stats_out);
}

}  // extern "C"
