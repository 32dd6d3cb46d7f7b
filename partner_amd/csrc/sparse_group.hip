// Sparse 3-D convolutions of the middle encoder (SubMConv3d / SparseConv3d of SpMiddleResNetFHD, det3d/models/backbones/scn.py:97-192) over
// GROUPS of 32 output sites with similar neighbourhoods -- r4, the successor of conv_mfma_kernel's gather mode on the 32 / 64 / 128-channel
// levels.
//
// The gather mode works on tiles of 128 consecutive sites (key order) and multiplies, for every tap any of the 128 sites has, ALL 128 rows:
// 189 GFLOP per sweep issued for 110 GFLOP of existing (site, tap) pairs (tools/sparse_tap_stats.py).  Two changes:
//   1. the unit is a WAVE with 32 sites: its own tap list (the union of its sites' neighbour masks), its own K loop, no LDS tile and no
//      barrier -- every lane gathers the 16-byte fragments of its site's neighbour row straight from L2 (a row is read 32 bytes per K
//      step by the two lane halves, the rest of its 128-byte lines is hit in L1 by the following steps), weights straight into the
//      operands as in conv_wchain.hip; three or four waves per SIMD run independently.
//   2. the 32 sites of a group are chosen by SORTING windows of 4096 sites (key order = spatially close) by their 27-bit neighbour
//      mask: sites on the same kind of surface share a group, and the union of a group's masks is close to each mask -- 138 GFLOP issued
//      with windows of 4096 (global sort: 130; 32 consecutive sites: 178).
// The result rows go back to their own places (perm), so the feature matrices keep the key order and every other kernel (neighbour
// tables, strided stages, the dense scatter) is untouched.  Deterministic: a site's sum runs over the group's taps in ascending order, each
// a K-ordered MFMA chain (absent neighbours add exact zeros); agrees with the gather mode to ~2e-6 of the output's range.
#include "pn_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int SG_WIN = 4096;      // sites per sorted window

// one block per window: masks -> bitonic sort of (mask, site) in LDS -> perm (site per slot, -1 past the live sites) and the union mask
// of every 32-slot group
__global__ __launch_bounds__(1024) void sparse_group_rows_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ n_valid, int cap, int taps,
                                                                 int32_t* __restrict__ perm, uint32_t* __restrict__ gmask) {
  __shared__ unsigned long long key[SG_WIN];
  const int n = min(*n_valid, cap);
  const int base = blockIdx.x * SG_WIN;
  if (base >= n) {      // a window without live sites: mark its slots empty (the convolution never reaches them, but keep the buffers defined)
    for (int i = threadIdx.x; i < SG_WIN && base + i < cap; i += 1024) perm[base + i] = -1;
    for (int i = threadIdx.x; i < SG_WIN / 32 && (base >> 5) + i < (cap + 31) / 32; i += 1024) gmask[(base >> 5) + i] = 0u;
    return;
  }
  for (int i = threadIdx.x; i < SG_WIN; i += 1024) {
    const int row = base + i;
    unsigned long long k = 1ull << 40;       // dead slots sort behind every live one
    if (row < n) {
      unsigned m = 0;
      const int32_t* p = nbr + (size_t)row * taps;
      for (int t = 0; t < taps; ++t) m |= (p[t] >= 0 ? 1u : 0u) << t;
      k = ((unsigned long long)m << 12) | (unsigned)i;
    }
    key[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= SG_WIN; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < SG_WIN / 2; i += 1024) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long x = key[lo], y = key[hi];
        if ((x > y) == up) {
          key[lo] = y;
          key[hi] = x;
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < SG_WIN; i += 1024) {
    const unsigned long long k = key[i];
    const bool live = k < (1ull << 40);
    if (base + i < cap) perm[base + i] = live ? base + (int)(k & 4095ull) : -1;
    unsigned m = live ? (unsigned)(k >> 12) : 0u;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m |= __shfl_xor(m, o, 32);
    if ((i & 31) == 0 && (base + i) / 32 < (cap + 31) / 32) gmask[(base + i) >> 5] = m;
  }
}

struct SpwArgs {
  const float* in;
  const int32_t* nbr;
  const int32_t* n_valid;
  const int32_t* perm;
  const uint32_t* gmask;
  const float* w;
  const float* scale;
  const float* shift;
  const float* res;
  float* out;
  int cap, taps, cin, cout;
  int cin_chunks, cout_pad;
  unsigned in_bytes, w_bytes;
  int act;
};

// wave = one group of 32 sites x 32 NC columns (blockIdx.y walks further column groups).  packed weights: pn_pack_conv_weight_f32's layout
// with (kh, kw) = (taps, 1): [tap][chunk][k4 8][cout_pad][4]
template <int NC>
__global__ __launch_bounds__(256) void sparse_conv_wave_kernel(SpwArgs a) {
  __shared__ int32_t s_src[4][32 * 28];      // per wave: [tap][site] neighbour rows of the group, then the group's own rows (slot 27)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int n = min(*a.n_valid, a.cap);
  // live groups: windows are sorted with their dead slots last, so the groups below ceil(n / 32) are exactly the ones with a live site;
  // blocks of 4 groups are dealt over the XCDs in contiguous runs (neighbouring groups gather neighbouring rows: one L2)
  const int nblk = ((n + 31) / 32 + 3) / 4;
  int mb;
  {
    const int q = nblk >> 3, r = nblk & 7, x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= (x < r ? q + 1 : q)) return;
    mb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int g = mb * 4 + wv;
  if (g * 32 >= n) return;      // (wave-uniform; no block barrier anywhere below)
  const unsigned gm = a.gmask[g];
  const int prow = g * 32 + li < a.cap ? a.perm[g * 32 + li] : -1;
  int32_t* src = s_src[wv];
  for (int t = lh; t < a.taps; t += 2) src[t * 32 + li] = prow >= 0 ? a.nbr[(size_t)prow * a.taps + t] : -1;
  if (lh == 0) src[27 * 32 + li] = prow;
  const int n0 = blockIdx.y * 32 * NC;
  unsigned uoff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) uoff[c] = (unsigned)(((size_t)lh * a.cout_pad + n0 + 32 * c + li) * 16);
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned row_bytes = (unsigned)a.cin * 4u;
  const int CG = a.cin >> 3;      // K steps (8 channels) per tap: even (cin a multiple of 16)

  f32x16 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  // the step sequence = (tap in the group's mask, ascending) x (8-channel group); the loads run two steps ahead of the MFMAs
  unsigned pm = gm;            // taps not yet requested
  int pt = 0, pcg = CG;        // the step requested next (pcg == CG: take the next tap first)
  unsigned pvo = 0xffffffffu;
  f32x4 fa[2], fb[2][NC];
  auto request = [&](int slot) __attribute__((always_inline)) {
    if (pcg == CG) {
      if (pm) {
        pt = __builtin_ctz(pm);
        pm &= pm - 1u;
        pcg = 0;
        const int s = src[pt * 32 + li];      // (written by this wave: the LDS executes a wave's accesses in order)
        pvo = s >= 0 ? (unsigned)s * row_bytes + (unsigned)lh * 16u : 0xffffffffu;
      } else {
        pvo = 0xffffffffu;      // past the last step: the loads return zeros and nobody multiplies them
        pcg = 0;
      }
    }
    const unsigned so_a = (unsigned)pcg * 32u;
    const unsigned so_w = (unsigned)(((pt * a.cin_chunks + (pcg >> 2)) * 8 + (pcg & 3) * 2)) * cp16;
    fa[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, pvo, so_a, 0));
#pragma unroll
    for (int c = 0; c < NC; ++c) fb[slot][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, uoff[c], so_w, 0));
    ++pcg;
  };
  const int nsteps = __builtin_popcount(gm) * CG;
  request(0);
  __builtin_amdgcn_sched_barrier(0);
  request(1);
  __builtin_amdgcn_sched_barrier(0);
  for (int s = 0; s < nsteps; s += 2) {
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][j], fb[slot][c][j], acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      request(slot);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: rows back to their own places
  const bool relu = a.act == PN_ACT_RELU;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = n0 + 32 * c + li;
    const bool cok = col < a.cout;
    const float sc = (cok && a.scale) ? a.scale[col] : 1.f;
    const float sh = (cok && a.shift) ? a.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = src[27 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      if (orow < 0 || !cok) continue;
      float v = fmaf(acc[c][r], sc, sh);
      if (a.res) v += a.res[(size_t)orow * a.cout + col];
      a.out[(size_t)orow * a.cout + col] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

}  // namespace

extern "C" {

int pn_sparse_group_rows(const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps, int32_t* perm, uint32_t* group_mask, pn_stream_t stream) {
  PN_REQUIRE(nbr && n_out && perm && group_mask && out_capacity >= 1 && taps >= 1 && taps <= 27, "sparse_group_rows: bad arguments");
  const int windows = pn::cdiv(out_capacity, SG_WIN);
  hipLaunchKernelGGL(sparse_group_rows_kernel, dim3((unsigned)windows), dim3(1024), 0, pn::S(stream), nbr, n_out, out_capacity, taps, perm, group_mask);
  return pn::check_launch("sparse_group_rows_kernel");
}

int pn_sparse_conv_grouped_f32(const float* in, int in_rows, int cin, const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps,
                               const int32_t* perm, const uint32_t* group_mask, const float* packed_w, int cout, const float* scale, const float* shift,
                               int act, const float* residual, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && nbr && n_out && perm && group_mask && packed_w && out, "sparse_conv_grouped: null pointer");
  PN_REQUIRE(in_rows >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 1 && out_capacity >= 1 && taps >= 1 && taps <= 27, "sparse_conv_grouped: cin must be a "
                                                                                                                     "multiple of 16, taps <= 27");
  PN_REQUIRE((unsigned long long)in_rows * cin * 4ull < (1ull << 32), "sparse_conv_grouped: input feature matrix too large for the buffer descriptor");
  PN_REQUIRE(act == PN_ACT_NONE || act == PN_ACT_RELU, "sparse_conv_grouped: activation none or ReLU");
  PN_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "sparse_conv_grouped: pointers must be 16-byte aligned");
  SpwArgs a{};
  a.in = in; a.nbr = nbr; a.n_valid = n_out; a.perm = perm; a.gmask = group_mask; a.w = packed_w; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
  a.cap = out_capacity; a.taps = taps; a.cin = cin; a.cout = cout;
  a.cin_chunks = pn::cdiv(cin, 32); a.cout_pad = pn::cdiv(cout, 32) * 32;
  a.in_bytes = (unsigned)((size_t)in_rows * cin * 4);
  a.w_bytes = (unsigned)((size_t)taps * a.cin_chunks * 8 * a.cout_pad * 16);
  a.act = act;
  hipStream_t st = pn::S(stream);
  const int blocks = (pn::cdiv(pn::cdiv(out_capacity, 32), 4) + 7) / 8 * 8;
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  const int ncol32 = a.cout_pad / 32;
  auto launch = [&](auto kern, int nc) {
    const dim3 grid((unsigned)blocks, (unsigned)pn::cdiv(ncol32, nc));
    if (prof) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, st, ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, a);
  };
  if (ncol32 >= 4) launch(&sparse_conv_wave_kernel<4>, 4);
  else if (ncol32 >= 2) launch(&sparse_conv_wave_kernel<2>, 2);
  else launch(&sparse_conv_wave_kernel<1>, 1);
  return pn::check_launch("sparse_conv_wave_kernel");
}

}  // extern "C"
