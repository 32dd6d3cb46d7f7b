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
#ifndef PN_SG_EXP
#define PN_SG_EXP 0      // diagnostic builds only (tools/micro/group_rows_check.hip): bit 0 no mask build, bit 1 no sort
#endif

// one block per window: masks -> bitonic sort of (mask, site) -> perm (site per slot, -1 past the live sites) and the union mask of every
// 32-slot group.  A thread keeps FOUR consecutive keys in registers: the exchange distances 1, 2 stay inside the thread, 4 .. 128 inside
// the wave (lane shuffles), only distances >= 256 go through LDS -- 10 of the 78 stages (the all-LDS version spent 63 us per launch, most
// of it in 78 block-wide barriers, on the 25 - 50 blocks a frame has).
__device__ __forceinline__ void sg_cmpx(unsigned long long& x, unsigned long long& y, bool up) {
  const unsigned long long lo = x < y ? x : y, hi = x < y ? y : x;
  x = up ? lo : hi;
  y = up ? hi : lo;
}

template <int WIN>
__global__ __launch_bounds__(WIN / 4) void sparse_group_rows_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ n_valid, int cap, int taps,
                                                                 int32_t* __restrict__ perm, uint32_t* __restrict__ gmask,
                                                                 const uint8_t* __restrict__ row_bits, int rows, int bits_per_row) {
  constexpr int NWV = WIN / 256;      // waves of the block (four keys per thread)
  __shared__ unsigned long long key[WIN];
  const int n = min(*n_valid, cap);
  const int base = blockIdx.x * WIN;
  // a window without live sites: nothing to do -- the convolutions walk the groups below ceil(n / 32) only, the slots behind stay undefined (the
  // capacity is several times the live count: filling them wrote 12 MB per call on the level with 3.2 M slots, from 750 blocks of 1024
  // threads that stood in the way of the convolutions on the other stream)
  if (base >= n) return;
  const int t = threadIdx.x;
  // the window's masks: its rows of the neighbour table are one contiguous block of taps-bit records -- read it coalesced, one ballot per 64
  // entries = 64 bits of the window's bit stream in LDS, a row's mask = bits [taps i, taps (i + 1)) of the stream.  (A thread walking its own
  // rows' 27 entries issued 108 scattered loads; OR-ing bits into per-row words with LDS atomics put ~27 lanes on one address: 42 of the
  // kernel's 63 us either way.)
  if (row_bits == nullptr) {
    const int live = min(n - base, WIN) * taps;
    const int words = (WIN * taps + 63) / 64;       // <= WIN * 32 / 64 = half of the array
    const int32_t* p = nbr + (size_t)base * taps;
    const int wave = t >> 6, lane = t & 63;
    constexpr int UN = 12;
    for (int w0 = wave; w0 < words && !(PN_SG_EXP & 1); w0 += NWV * UN) {
      int32_t v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int f = (w0 + NWV * u) * 64 + lane;
        v[u] = f < live ? p[f] : -1;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const unsigned long long bits = __ballot(v[u] >= 0);
        if (lane == 0 && w0 + NWV * u < words) key[w0 + NWV * u] = bits;
      }
    }
    if (t == 0) key[words] = 0ull;      // (a record that ends on the stream's last bit still reads the word behind it)
  }
  __syncthreads();
  unsigned long long e[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * t + r;
    unsigned m = 0;
    if (row_bits) {
      // r4: the neighbour kernel left one byte per (site, row of taps): 9 bytes per site instead of 27 table entries (the window's read of the
      // table, 442 KB through one CU, was 19 of the sort's 47 us)
      if (base + i < n) {
        const uint8_t* rb = row_bits + (size_t)(base + i) * rows;
        for (int q = 0; q < rows; ++q) m |= (unsigned)rb[q] << (q * bits_per_row);
      }
    } else {
      const unsigned o = (unsigned)(i * taps), sh = o & 63u;
      const unsigned long long lo = key[o >> 6], hi = key[(o >> 6) + 1];
      m = (unsigned)((lo >> sh) | (sh ? hi << (64u - sh) : 0ull)) & (taps >= 32 ? ~0u : (1u << taps) - 1u);
    }
    e[r] = base + i < n ? (((unsigned long long)m << 12) | (unsigned)i) : (1ull << 40);       // dead slots sort behind every live one
  }
  __syncthreads();      // (the LDS stages below reuse the array)
  for (int k = 2; k <= WIN && !(PN_SG_EXP & 2); k <<= 1) {
    const bool up = ((4 * t) & k) == 0;      // (k >= 4: the same for the thread's four keys; k = 2 handled below)
    for (int j = k >> 1; j >= 256; j >>= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) key[4 * t + r] = e[r];
      __syncthreads();
      const bool lower = ((4 * t) & j) == 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned long long o = key[(4 * t + r) ^ j];
        const bool take_min = lower == up;
        e[r] = take_min ? (e[r] < o ? e[r] : o) : (e[r] < o ? o : e[r]);
      }
      __syncthreads();
    }
    for (int j = min(k >> 1, 128); j >= 4; j >>= 1) {
      const bool lower = ((4 * t) & j) == 0;
      const bool take_min = lower == up;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned lo32 = __shfl_xor((unsigned)e[r], j >> 2, 64), hi32 = __shfl_xor((unsigned)(e[r] >> 32), j >> 2, 64);
        const unsigned long long o = ((unsigned long long)hi32 << 32) | lo32;
        e[r] = take_min ? (e[r] < o ? e[r] : o) : (e[r] < o ? o : e[r]);
      }
    }
    if (k == 2) {
      sg_cmpx(e[0], e[1], true);
      sg_cmpx(e[2], e[3], false);
    } else {
      if (k > 2) {
        sg_cmpx(e[0], e[2], up);
        sg_cmpx(e[1], e[3], up);
      }
      sg_cmpx(e[0], e[1], up);
      sg_cmpx(e[2], e[3], up);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * t + r;
    const unsigned long long k = e[r];
    const bool live = k < (1ull << 40);
    if (base + i < cap) perm[base + i] = live ? base + (int)(k & 4095ull) : -1;
  }
  // union mask of the 32-slot groups: 8 threads x 4 keys
  unsigned m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) m |= e[r] < (1ull << 40) ? (unsigned)(e[r] >> 12) : 0u;
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) m |= __shfl_xor(m, o, 8);
  if ((t & 7) == 0 && (base + 4 * t) / 32 < (cap + 31) / 32) gmask[(base + 4 * t) >> 5] = m;
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
  unsigned in_bytes, w_bytes, out_bytes;
  int act;
  const int32_t* bounds;               // nullable: [0..8] XCD cut points in groups, [9..17] in blocks of 4 groups (pn_sparse_group_balance)
  int wide4_groups, wide2_groups;      // a wave takes 4 / at least 2 column tiles from this many live groups on
};

#ifndef PN_SG_EXP
#define PN_SG_EXP 0      // diagnostic builds (tools/sparseq.sh): 1 no input gathers after the first, 2 no weight loads after the first, 4 no LDS image,
#endif                   // 8 no MFMAs, 16 no join / stores, 32 no neighbour-table prologue -- wrong results, the time shows what a unit waits for
// the taps of a mask in ascending order, one at a time
struct UnitCursor {
  unsigned m;            // this wave's taps after the current one
  int t;                 // current tap
  bool live;
  __device__ __forceinline__ void start(unsigned own) { m = own; live = true; t = 0; next(); }
  // branch-free (scalar selects): a branch inside the K loop splits it into basic blocks, and the register allocator then copies the
  // accumulator tiles in and out of the MFMA registers at every block boundary (59 v_mov + a drain of the matrix pipe per 8 MFMAs: r4 found
  // the block-per-group kernel at 0.59 of the peak for this reason alone)
  __device__ __forceinline__ void next() {
    live = m != 0u;
    t = live ? (int)__builtin_ctz(m | 0x80000000u) : t;
    m &= m - 1u;
  }
};

// wave = one group of 32 sites x 32 NC columns (blockIdx.y walks further column groups).  packed weights: pn_pack_conv_weight_f32's layout
// with (kh, kw) = (taps, 1): [tap][chunk][k4 8][cout_pad][4]
// Input rows reach the MFMA operands through a wave-private LDS image in WHOLE 128-byte lines: a chunk (32 channels of the 32 neighbour
// rows of one tap) is fetched by four loads of 8 rows x 128 bytes, written row-major (row stride CH + 4 floats) and read back as the four
// K steps' fragments.  (The first version had lane (site, half) fetch its own 16 bytes per K step: 32 different lines per load, each
// touched again by the next three steps -- with 16 waves per CU those lines no longer sit in the 32 KB L1, and the 64 -> 64 layers ran at
// 0.47 of the MFMA peak on 4x the L2 traffic.)  The LDS executes a wave's accesses in order, so the image needs no barrier and no
// second buffer: a chunk's four fragment reads are issued before the next chunk's stores.
template <int NC, int CH, bool ONE = false>      // ONE: the layer's input is one chunk wide (cin == CH): a unit = a tap, branch-free loop
__device__ __forceinline__ void sparse_conv_wave_body(const SpwArgs& a, int32_t* src, float* stage, int g, int n0, int lane) {
  constexpr int LPR = CH / 4;          // lanes per row of a staging load
  constexpr int RPI = 64 / LPR;        // rows per staging load
  constexpr int NI = 32 / RPI;         // staging loads per chunk
  constexpr int SPC = CH / 8;          // K steps (8 channels) per chunk
  constexpr int LD = CH + 4;           // floats per row of the image (rows 16 bytes apart modulo the bank row: conflict-free b128 accesses)
  const int li = lane & 31, lh = lane >> 5;
  const unsigned gm = a.gmask[g];
  const int prow = g * 32 + li < a.cap ? a.perm[g * 32 + li] : -1;
  {
    // the site's row of the neighbour table: every load requested before the first LDS store (one at a time the 14 loads of a lane were a
    // chain of memory latencies in front of the wave's first MFMA)
    int32_t nv[14];
    const int32_t* np = a.nbr + (size_t)max(prow, 0) * a.taps;
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int t = lh + 2 * k;
      nv[k] = (prow >= 0 && t < a.taps) ? np[min(t, a.taps - 1)] : -1;
    }
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int t = lh + 2 * k;
      if (t < 27) src[t * 32 + li] = nv[k];
    }
  }
  if (lh == 0) src[27 * 32 + li] = prow;
  const int srow = lane / LPR, scol = (lane % LPR) * 4;      // staging role: rows srow + RPI q, 16 bytes at float scol of the chunk
  unsigned uoff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) uoff[c] = (unsigned)(((size_t)lh * a.cout_pad + n0 + 32 * c + li) * 16);
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned row_bytes = (unsigned)a.cin * 4u;
  const int nch = a.cin / CH;          // chunks per tap
  const int CG = a.cin >> 3;           // K steps per tap

  // (r5: splitting a tile's K steps over four accumulators, as sparse_conv_group4_kernel does, changed nothing here: 117 us on the 32 -> 32
  // layers either way -- a unit is four K steps, the wave spends its time between them)
  f32x16 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  if constexpr (ONE) {
    // r4: the K loop as ONE basic block per tap.  The generic loop below keeps (tap, chunk) cursors with branches; the compiler cut it into
    // blocks of 4 - 8 MFMAs with ~40 scalar / vector instructions between them (0.43 of the MFMA peak on the 32 -> 32 layers).  Here the
    // cursors advance by scalar selects and the K step of every weight request is a compile-time constant.
    UnitCursor ca, cb;
    ca.start(gm);
    cb.start(gm);
    f32x4 ra[NI], fb[2][NC];
    bool first_a = true, first_b = true;      // (diagnostic builds only)
    auto request_a = [&]() __attribute__((always_inline)) {
      if ((PN_SG_EXP & 1) && !first_a) { ca.next(); return; }
      first_a = false;
#pragma unroll
      for (int q = 0; q < NI; ++q) {
        const int sv = src[ca.t * 32 + srow + RPI * q];      // (written by this wave: the LDS executes a wave's accesses in order)
        const int sidx = ca.live ? sv : -1;
        const unsigned vo = sidx >= 0 ? (unsigned)sidx * (unsigned)(CH * 4) + (unsigned)scol * 4u : 0xffffffffu;      // (cin == CH: a shift)
        ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, 0, 0));
      }
      ca.next();
    };
    auto request_b = [&](int slot, int kk) __attribute__((always_inline)) {
      if ((PN_SG_EXP & 2) && !first_b) return;
      const unsigned so_w = (unsigned)((cb.t * a.cin_chunks * 8 + kk * 2)) * cp16;
#pragma unroll
      for (int c = 0; c < NC; ++c) fb[slot][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, cb.live ? uoff[c] : 0xffffffffu, so_w, 0));
    };
    const int nunits = __builtin_popcount(gm);
    request_a();
    request_b(0, 0);
    request_b(1, 1);
    first_b = false;
    if (SPC == 2) cb.next();
    for (int u = 0; u < nunits; ++u) {
      f32x4 fa[SPC];
      if (PN_SG_EXP & 4) {
#pragma unroll
        for (int k = 0; k < SPC; ++k) fa[k] = ra[k % NI];
        request_a();
      } else {
#pragma unroll
        for (int q = 0; q < NI; ++q) *reinterpret_cast<f32x4*>(stage + (srow + RPI * q) * LD + scol) = ra[q];
        request_a();
#pragma unroll
        for (int k = 0; k < SPC; ++k) fa[k] = *reinterpret_cast<const f32x4*>(stage + li * LD + (2 * k + lh) * 4);
      }
#pragma unroll
      for (int k = 0; k < SPC; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            if (PN_SG_EXP & 8) acc[c][j] += fa[k][j] * fb[k & 1][c][j];
            else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k][j], fb[k & 1][c][j], acc[c], 0, 0, 0);
          }
        // the slot just used takes step k + 2: of this tap, or (past its last step) of the next one
        if (SPC == 4 && k == 2) cb.next();
        request_b(k & 1, (k + 2) % SPC);
        if (SPC == 2 && k == 1) cb.next();
      }
    }
  } else {
  // two request pointers walk the same sequence (taps of the group's mask ascending, channels ascending): the input rows one CHUNK ahead
  // of the MFMAs, the weight fragments two K STEPS ahead (r4: a whole chunk ahead changes nothing on the 32 -> 32 layers, 137 us either
  // way, and neither does serving every weight or every input row from a cache-resident set: the loop is not waiting for memory)
  unsigned am = gm;
  int at = 0, ach = nch;
  unsigned avo[NI];
  f32x4 ra[NI];
  auto request_a = [&]() __attribute__((always_inline)) {
    if (ach == nch) {
      ach = 0;
      if (am) {
        at = __builtin_ctz(am);
        am &= am - 1u;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
          const int sidx = src[at * 32 + srow + RPI * q];      // (written by this wave: the LDS executes a wave's accesses in order)
          avo[q] = sidx >= 0 ? (unsigned)sidx * row_bytes + (unsigned)scol * 4u : 0xffffffffu;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NI; ++q) avo[q] = 0xffffffffu;      // past the last chunk: zeros that nobody multiplies
      }
    }
#pragma unroll
    for (int q = 0; q < NI; ++q) ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, avo[q], (unsigned)ach * (CH * 4u), 0));
    ++ach;
  };
  unsigned bm = gm;
  int bt = 0, bcg = CG;
  f32x4 fb[2][NC];
  auto request_b = [&](int slot) __attribute__((always_inline)) {
    if (bcg == CG) {
      bcg = 0;
      if (bm) {
        bt = __builtin_ctz(bm);
        bm &= bm - 1u;
      }
    }
    const unsigned so_w = (unsigned)(((bt * a.cin_chunks + (bcg >> 2)) * 8 + (bcg & 3) * 2)) * cp16;
#pragma unroll
    for (int c = 0; c < NC; ++c) fb[slot][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, uoff[c], so_w, 0));
    ++bcg;
  };
  const int nunits = __builtin_popcount(gm) * nch;
  request_a();
  request_b(0);
  request_b(1);
  for (int u = 0; u < nunits; ++u) {
#pragma unroll
    for (int q = 0; q < NI; ++q) *reinterpret_cast<f32x4*>(stage + (srow + RPI * q) * LD + scol) = ra[q];
    request_a();
    f32x4 fa[SPC];
#pragma unroll
    for (int k = 0; k < SPC; ++k) fa[k] = *reinterpret_cast<const f32x4*>(stage + li * LD + (2 * k + lh) * 4);
#pragma unroll
    for (int k = 0; k < SPC; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k][j], fb[k & 1][c][j], acc[c], 0, 0, 0);
      request_b(k & 1);
    }
  }

  }
  if ((PN_SG_EXP & 16) && acc[0][0] != 12345.f) return;
  // ---- epilogue: rows back to their own places.  No branches: dead rows / columns are redirected out of the descriptors' range
  // (loads return 0, stores are dropped), the residuals of a column tile are all requested before the first store
  const bool relu = a.act == PN_ACT_RELU;
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.out_bytes : 0u, 0x00020000);
  unsigned ro[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = src[27 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
    ro[r] = orow >= 0 ? (unsigned)orow * (unsigned)a.cout * 4u : 0xffffffffu;
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int col = n0 + 32 * c + li;
    const bool cok = col < a.cout;
    const float sc = (cok && a.scale) ? a.scale[col] : 1.f;
    const float sh = (cok && a.shift) ? a.shift[col] : 0.f;
    float rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, (cok && ro[r] != 0xffffffffu) ? ro[r] + (unsigned)col * 4u : 0xffffffffu, 0, 0));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = fmaf(acc[c][r], sc, sh) + rv[r];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu ? fmaxf(v, 0.f) : v), rsrc_o,
                                            (cok && ro[r] != 0xffffffffu) ? ro[r] + (unsigned)col * 4u : 0xffffffffu, 0, 0);
    }
  }
}

// grid.y = the layer's 32-column tiles; how many of them a wave takes (NC) is decided here from the LIVE site count, which only the device
// knows: few groups (the 128-channel level: 24k sites = 750 groups for 1024 SIMDs) -> narrow waves, so that every SIMD has two or three
template <int NCMAX>
__global__ __launch_bounds__(256, 3) void sparse_conv_wave_kernel(SpwArgs a) {
  __shared__ int32_t s_src[4][32 * 28];      // per wave: [tap][site] neighbour rows of the group, then the group's own rows (slot 27)
  __shared__ __attribute__((aligned(16))) float s_stage[4][32 * 36];      // per wave: the current chunk of the 32 gathered rows
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = min(*a.n_valid, a.cap);
  const int groups = (n + 31) / 32;
  int nc = NCMAX;
  if (NCMAX >= 4 && groups < a.wide4_groups) nc = 2;
  if (NCMAX >= 2 && groups < a.wide2_groups) nc = 1;
  if (NCMAX >= 4 && (a.cin % 32) && nc == 4) nc = 2;      // (the 16-channel-input bodies exist for one and two column tiles)
  if ((int)blockIdx.y * nc >= NCMAX) return;
  // live groups: windows are sorted with their dead slots last, so the groups below ceil(n / 32) are exactly the ones with a live site;
  // blocks of 4 groups are dealt over the XCDs in contiguous runs (neighbouring groups gather neighbouring rows: one L2)
  const int nblk = (groups + 3) / 4;
  int mb;
  if (a.bounds) {      // XCD x walks the run [bounds[9 + x], bounds[10 + x]) of 4-group blocks: equal WORK (taps of the groups' masks) per XCD
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    mb = a.bounds[9 + x] + idx;
    if (mb >= a.bounds[10 + x]) return;
  } else {
    const int q = nblk >> 3, r = nblk & 7, x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= (x < r ? q + 1 : q)) return;
    mb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int g = mb * 4 + wv;
  if (g * 32 >= n) return;      // (wave-uniform; no block barrier anywhere below)
  const int n0 = blockIdx.y * 32 * nc;
  if (a.cin % 32) {      // 16 input channels (the first strided stage): 64-byte rows, chunks of 16 channels
    if (a.cin == 16) {
      if (NCMAX >= 2 && nc == 2) sparse_conv_wave_body<2, 16, true>(a, s_src[wv], s_stage[wv], g, n0, lane);
      else sparse_conv_wave_body<1, 16, true>(a, s_src[wv], s_stage[wv], g, n0, lane);
      return;
    }
    if (NCMAX >= 2 && nc == 2) sparse_conv_wave_body<2, 16>(a, s_src[wv], s_stage[wv], g, n0, lane);
    else sparse_conv_wave_body<1, 16>(a, s_src[wv], s_stage[wv], g, n0, lane);
    return;
  }
  if (a.cin == 32) {
    if (NCMAX >= 4 && nc == 4) sparse_conv_wave_body<4, 32, true>(a, s_src[wv], s_stage[wv], g, n0, lane);
    else if (NCMAX >= 2 && nc == 2) sparse_conv_wave_body<2, 32, true>(a, s_src[wv], s_stage[wv], g, n0, lane);
    else sparse_conv_wave_body<1, 32, true>(a, s_src[wv], s_stage[wv], g, n0, lane);
    return;
  }
  if (NCMAX >= 4 && nc == 4) sparse_conv_wave_body<4, 32>(a, s_src[wv], s_stage[wv], g, n0, lane);
  else if (NCMAX >= 2 && nc == 2) sparse_conv_wave_body<2, 32>(a, s_src[wv], s_stage[wv], g, n0, lane);
  else sparse_conv_wave_body<1, 32>(a, s_src[wv], s_stage[wv], g, n0, lane);
}

// ---- block = ONE group, its (tap, 32-channel chunk) units split over the block's four waves (K split), every wave with all
// NC column tiles; the four partial tiles are summed in a fixed order through LDS, wave w finishing a quarter.  For the 64- and
// 128-channel levels: a group is 20-80 us of MFMA work there, and with one WAVE per group the 1800-5800 waves of a level left the SIMDs
// with one or two waves each, unevenly (0.55 of the MFMA peak); blocks of a quarter of that work each, handed out as CUs free up, level it.
// A wave's share must not depend on WHICH sites share its group (the same site lands in other groups when the batch changes, and its sum
// must stay bit for bit the same: key-point selections downstream amplify last-bit differences): wave w takes channel chunk w & (nch - 1)
// of the taps whose NUMBER falls in its class -- every tap for four chunks (128 channels), t & 1 == w >> 1 for two (64 channels).  A site's
// four partial sums are then sums over its own neighbours in ascending tap order (absent ones add exact zeros), joined as ((0 + 1) + 2) + 3.

template <int NC>
__global__ __launch_bounds__(256, NC > 2 ? 3 : 4) void sparse_conv_group4_kernel(SpwArgs a) {
  constexpr int LD = 36;
  __shared__ int32_t s_src[32 * 28];
  constexpr int JC = NC > 2 ? 2 : NC;      // column tiles per join pass: 32 KB of LDS instead of 64 for NC = 4 (three blocks per CU instead of two)
  __shared__ __attribute__((aligned(16))) float s_buf[4 * 32 * 32 * JC];      // the waves' chunk images (4 x 32 x 36 floats) / afterwards the four partial tiles
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int n = min(*a.n_valid, a.cap);
  const int groups = (n + 31) / 32;
  int g;
  if (a.bounds) {      // XCD x walks the groups [bounds[x], bounds[x + 1]): contiguous (neighbouring groups gather neighbouring rows: one L2) AND
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;      // equal in work -- equal COUNTS left one XCD with 1.18 - 1.30 x the mean taps
    g = a.bounds[x] + idx;
    if (g >= a.bounds[x + 1]) return;
  } else {
    const int q = groups >> 3, r = groups & 7, x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= (x < r ? q + 1 : q)) return;
    g = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const unsigned gm = a.gmask[g];
  if (PN_SG_EXP & 32) {
    for (int i = tid; i < 28 * 32; i += 256) s_src[i] = (g * 32 + (i & 31)) % max(n, 1);
  } else {
    const int i = tid & 31, prow = g * 32 + i < a.cap ? a.perm[g * 32 + i] : -1;
    int32_t nv[4];
    const int32_t* np = a.nbr + (size_t)max(prow, 0) * a.taps;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = (tid >> 5) + 8 * k;
      nv[k] = (prow >= 0 && t < a.taps) ? np[min(t, a.taps - 1)] : -1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = (tid >> 5) + 8 * k;
      if (t < 27) s_src[t * 32 + i] = nv[k];
    }
    if (tid < 32) s_src[27 * 32 + i] = prow;
  }
  __syncthreads();
  float* stage = s_buf + wv * (32 * LD);
  const int srow = lane >> 3, scol = (lane & 7) * 4;
  const int n0 = blockIdx.y * 32 * NC;
  unsigned uoff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) uoff[c] = (unsigned)(((size_t)lh * a.cout_pad + n0 + 32 * c + li) * 16);
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);
  const unsigned cp16 = (unsigned)a.cout_pad * 16u;
  const unsigned row_bytes = (unsigned)a.cin * 4u;
  const int nch = a.cin >> 5;                                    // 4 or 2 chunks of 32 channels
  const int ch = wv & (nch - 1);
  const unsigned own = nch == 4 ? gm : gm & (0x55555555u << (wv >> 1));
  const int mine = __builtin_popcount(own);                      // units of this wave: (its taps) x (its chunk)

  // r5: FOUR independent accumulator tiles per wave.  tools/micro/mfma_dep_chain.hip: with three or four waves on a SIMD (this kernel's
  // occupancy) v_mfma_f32_32x32x2_f32 chains on one or two accumulator tiles per wave run at 103 - 122 TFLOP/s chip-wide, on four at 155 (with
  // one or two waves per SIMD the count does not matter) -- the two-tile form of this kernel was at exactly that: its bare MFMA loop, every
  // load / LDS access / join switched off, took 346 of the 128 -> 128 layer's 354 us.  NC = 2 therefore splits each tile's K steps by
  // parity over two accumulators (even steps of a unit into one, odd into the other, summed once before the join): the summation order
  // of a site is still a function of its own neighbours only (taps ascending, fixed step parity), so grouping and batch do not change a bit.
  constexpr int P = NC == 2 ? 2 : 1;
  f32x16 acc[NC][P];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.f;

  UnitCursor ca, cb;
  ca.start(own);
  cb.start(own);
  f32x4 ra[4], fb[2][NC];
  bool first_a = true, first_b = true;
  auto request_a = [&]() __attribute__((always_inline)) {
    if ((PN_SG_EXP & 1) && !first_a) { ca.next(); return; }
    first_a = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sv = s_src[ca.t * 32 + srow + 8 * q];
      const int sidx = ca.live ? sv : -1;
      const unsigned vo = sidx >= 0 ? (unsigned)sidx * row_bytes + (unsigned)scol * 4u : 0xffffffffu;
      ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, (unsigned)ch * 128u, 0));
    }
    ca.next();
  };
  // kk = the K step inside cb's unit: a compile-time constant at every call (the unit loop below is unrolled over its four steps)
  auto request_b = [&](int slot, int kk) __attribute__((always_inline)) {
    if ((PN_SG_EXP & 2) && !first_b) return;
    const unsigned so_w = (unsigned)(((cb.t * a.cin_chunks + ch) * 8 + kk * 2)) * cp16;
#pragma unroll
    for (int c = 0; c < NC; ++c) fb[slot][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, cb.live ? uoff[c] : 0xffffffffu, so_w, 0));
  };
  request_a();
  request_b(0, 0);
  request_b(1, 1);
  first_b = false;
  for (int u = 0; u < mine; ++u) {      // ONE basic block: no branch below (see UnitCursor)
    f32x4 fa[4];
    if (PN_SG_EXP & 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fa[k] = ra[k];
      request_a();
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(stage + (srow + 8 * q) * LD + scol) = ra[q];
      request_a();
#pragma unroll
      for (int k = 0; k < 4; ++k) fa[k] = *reinterpret_cast<const f32x4*>(stage + li * LD + (2 * k + lh) * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (PN_SG_EXP & 8) acc[c][j & (P - 1)][j] += fa[k][j] * fb[k & 1][c][j];
          else acc[c][j & (P - 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k][j], fb[k & 1][c][j], acc[c][j & (P - 1)], 0, 0, 0);
        }
      // the slot just used takes step k + 2: steps 2, 3 of this unit, then steps 0, 1 of the next one
      if (k == 1) {
        request_b(1, 3);
        cb.next();
      } else if (k == 0) {
        request_b(0, 2);
      } else {
        request_b(k & 1, k - 2);
      }
    }
  }
  // ---- join: [wave][c][r][lane] partial tiles, summed wave 0 + 1 + 2 + 3; wave w finishes registers 4 w .. 4 w + 3 (rows 8 w' .. ) of every tile;
  // JC column tiles per pass
  if constexpr (P == 2) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][0][r] += acc[c][1][r];
  }
  if ((PN_SG_EXP & 16) && acc[0][0][0] != 12345.f) return;
  const bool relu = a.act == PN_ACT_RELU;
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res), 0, a.res ? a.out_bytes : 0u, 0x00020000);
#pragma unroll
  for (int c0 = 0; c0 < NC; c0 += JC) {
    __syncthreads();      // every wave is done with its chunk image / the previous pass
#pragma unroll
    for (int c = 0; c < JC; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_buf[((wv * JC + c) * 16 + r) * 64 + lane] = acc[c0 + c][0][r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < JC; ++c) {
      const int col = n0 + 32 * (c0 + c) + li;
      const bool cok = col < a.cout;
      const float sc = (cok && a.scale) ? a.scale[col] : 1.f;
      const float sh = (cok && a.shift) ? a.shift[col] : 0.f;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * wv + rr;
        const int orow = s_src[27 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        const unsigned off = (cok && orow >= 0) ? ((unsigned)orow * (unsigned)a.cout + (unsigned)col) * 4u : 0xffffffffu;
        const float rv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, off, 0, 0));
        float v = ((s_buf[((0 * JC + c) * 16 + r) * 64 + lane] + s_buf[((1 * JC + c) * 16 + r) * 64 + lane]) + s_buf[((2 * JC + c) * 16 + r) * 64 + lane]) +
                  s_buf[((3 * JC + c) * 16 + r) * 64 + lane];
        v = fmaf(v, sc, sh) + rv;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, relu ? fmaxf(v, 0.f) : v), rsrc_o, off, 0, 0);
      }
    }
  }
}

// XCD cut points of equal work: prefix sums of the groups' tap counts (one block; <= ~100 k groups), cut where the prefix crosses k / 8 of the
// total.  bounds[0..8]: in groups; bounds[9..17]: in blocks of four groups (the wave-per-group kernel's blocks), cut on block boundaries.
__global__ __launch_bounds__(1024) void sparse_group_balance_kernel(const uint32_t* __restrict__ gmask, const int32_t* __restrict__ n_valid, int cap,
                                                                    int32_t* __restrict__ bounds) {
  __shared__ long long part[1024];
  __shared__ long long total_s;
  const int tid = threadIdx.x;
  const int n = min(*n_valid, cap), groups = (n + 31) / 32, nblk = (groups + 3) / 4;
  const int per = (nblk + 1023) / 1024;                       // 4-group blocks per thread (contiguous)
  const int b0 = min(nblk, tid * per), b1 = min(nblk, b0 + per);
  long long s = 0;
  for (int b = b0; b < b1; ++b)
    for (int g = 4 * b; g < min(groups, 4 * b + 4); ++g) s += __builtin_popcount(gmask[g]) + 1;      // + 1: a group's fixed cost (table, join)
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    long long run = 0;
    for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = run; run += v; }
    total_s = run;
    bounds[0] = 0; bounds[8] = groups; bounds[9] = 0; bounds[17] = nblk;
    for (int k = 1; k < 8; ++k) { bounds[k] = 0; bounds[9 + k] = 0; }      // a cut whose target is 0 stays here (tiny inputs)
  }
  __syncthreads();
  const long long total = total_s;
  long long run = part[tid];
  for (int b = b0; b < b1; ++b) {
    long long w = 0;
    for (int g = 4 * b; g < min(groups, 4 * b + 4); ++g) w += __builtin_popcount(gmask[g]) + 1;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const long long target = total * k / 8;
      if (run < target && run + w >= target) { bounds[9 + k] = b + 1; bounds[k] = min(groups, 4 * (b + 1)); }
    }
    run += w;
  }
  __syncthreads();
  if (tid == 0) {
    // a run is at most twice the mean (+ 4): the launchers size their grids for that (kBalanceSlack); the cuts stay ascending and cover everything
    const int maxb = 2 * ((nblk + 7) / 8) + 4;
    for (int k = 1; k < 8; ++k) {
      int c = bounds[9 + k];
      c = max(c, bounds[9 + k - 1]);
      c = min(c, bounds[9 + k - 1] + maxb);
      c = max(c, nblk - (8 - k) * maxb);
      c = min(max(c, 0), nblk);
      bounds[9 + k] = c;
      bounds[k] = min(groups, 4 * c);
    }
  }
}

}  // namespace

extern "C" {

int pn_sparse_group_balance(const uint32_t* group_mask, const int32_t* n_out, int out_capacity, int32_t* bounds, pn_stream_t stream) {
  PN_REQUIRE(group_mask && n_out && bounds && out_capacity >= 1, "sparse_group_balance: bad arguments");
  hipLaunchKernelGGL(sparse_group_balance_kernel, dim3(1), dim3(1024), 0, pn::S(stream), group_mask, n_out, out_capacity, bounds);
  return pn::check_launch("sparse_group_balance_kernel");
}

static int group_rows_run(const int32_t* nbr, const uint8_t* row_bits, int rows, int bits_per_row, const int32_t* n_out, int out_capacity, int taps, int32_t* perm,
                          uint32_t* group_mask, pn_stream_t stream) {
  PN_REQUIRE((nbr || row_bits) && n_out && perm && group_mask && out_capacity >= 1 && taps >= 1 && taps <= 27, "sparse_group_rows: bad arguments");
  // window = the span inside which sites may change places: 4096 groups best, 1024 finishes four times sooner per block (PN_SPARSE_WINDOW)
  static const int win = [] { const char* e = getenv("PN_SPARSE_WINDOW"); const int v = e ? atoi(e) : SG_WIN; return v == 1024 || v == 2048 ? v : SG_WIN; }();
  const int windows = pn::cdiv(out_capacity, win);
  if (win == 1024) hipLaunchKernelGGL(sparse_group_rows_kernel<1024>, dim3((unsigned)windows), dim3(256), 0, pn::S(stream), nbr, n_out, out_capacity, taps, perm, group_mask, row_bits, rows, bits_per_row);
  else if (win == 2048) hipLaunchKernelGGL(sparse_group_rows_kernel<2048>, dim3((unsigned)windows), dim3(512), 0, pn::S(stream), nbr, n_out, out_capacity, taps, perm, group_mask, row_bits, rows, bits_per_row);
  else hipLaunchKernelGGL(sparse_group_rows_kernel<SG_WIN>, dim3((unsigned)windows), dim3(SG_WIN / 4), 0, pn::S(stream), nbr, n_out, out_capacity, taps, perm, group_mask, row_bits, rows, bits_per_row);
  return pn::check_launch("sparse_group_rows_kernel");
}

int pn_sparse_group_rows(const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps, int32_t* perm, uint32_t* group_mask, pn_stream_t stream) {
  PN_REQUIRE(nbr, "sparse_group_rows: null pointer");
  return group_rows_run(nbr, nullptr, 0, 0, n_out, out_capacity, taps, perm, group_mask, stream);
}

// the same grouping from the row bytes of pn_sparse_neighbors_rows (rows = k0 k1 bytes per site, bits_per_row = k2): same perm and masks
int pn_sparse_group_rows_bits(const uint8_t* row_bits, int rows, int bits_per_row, const int32_t* n_out, int out_capacity, int32_t* perm, uint32_t* group_mask,
                              pn_stream_t stream) {
  PN_REQUIRE(row_bits && rows >= 1 && bits_per_row >= 1 && bits_per_row <= 8 && rows * bits_per_row <= 27, "sparse_group_rows_bits: bad arguments");
  return group_rows_run(nullptr, row_bits, rows, bits_per_row, n_out, out_capacity, rows * bits_per_row, perm, group_mask, stream);
}

int pn_sparse_conv_grouped_f32(const float* in, int in_rows, int cin, const int32_t* nbr, const int32_t* n_out, int out_capacity, int taps,
                               const int32_t* perm, const uint32_t* group_mask, const int32_t* xcd_bounds, const float* packed_w, int cout, const float* scale,
                               const float* shift, int act, const float* residual, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && nbr && n_out && perm && group_mask && packed_w && out, "sparse_conv_grouped: null pointer");
  PN_REQUIRE(in_rows >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 1 && out_capacity >= 1 && taps >= 1 && taps <= 27, "sparse_conv_grouped: cin must be a "
                                                                                                                     "multiple of 16, taps <= 27");
  PN_REQUIRE((unsigned long long)in_rows * cin * 4ull < (1ull << 32) && (unsigned long long)out_capacity * cout * 4ull < (1ull << 32),
             "sparse_conv_grouped: feature matrix too large for the buffer descriptor");
  PN_REQUIRE(act == PN_ACT_NONE || act == PN_ACT_RELU, "sparse_conv_grouped: activation none or ReLU");
  PN_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)packed_w & 15) == 0, "sparse_conv_grouped: pointers must be 16-byte aligned");
  SpwArgs a{};
  a.in = in; a.nbr = nbr; a.n_valid = n_out; a.perm = perm; a.gmask = group_mask; a.bounds = xcd_bounds; a.w = packed_w; a.scale = scale; a.shift = shift; a.res = residual; a.out = out;
  a.cap = out_capacity; a.taps = taps; a.cin = cin; a.cout = cout;
  a.cin_chunks = pn::cdiv(cin, 32); a.cout_pad = pn::cdiv(cout, 32) * 32;
  a.in_bytes = (unsigned)((size_t)in_rows * cin * 4);
  a.w_bytes = (unsigned)((size_t)taps * a.cin_chunks * 8 * a.cout_pad * 16);
  a.out_bytes = (unsigned)((size_t)out_capacity * cout * 4);
  a.act = act;
  static const int t4 = [] { const char* e = getenv("PN_SPARSE_WIDE4"); return e ? atoi(e) : 1536; }();
  static const int t2 = [] { const char* e = getenv("PN_SPARSE_WIDE2"); return e ? atoi(e) : 512; }();
  a.wide4_groups = t4; a.wide2_groups = t2;

  hipStream_t st = pn::S(stream);
  // grid: 8 XCDs x the longest run a balanced cut may give an XCD (sparse_group_balance_kernel: at most twice the mean + 4 blocks)
  const int nblk_cap = pn::cdiv(pn::cdiv(out_capacity, 32), 4);
  const int blocks = xcd_bounds ? 8 * (2 * ((nblk_cap + 7) / 8) + 4) : (nblk_cap + 7) / 8 * 8;
  pn::ProfileSlot ps{};
  const bool prof = pn::take_profile_slot(ps);
  const int ncol32 = a.cout_pad / 32;
  PN_REQUIRE(ncol32 == 1 || ncol32 == 2 || ncol32 == 4, "sparse_conv_grouped: 32, 64 or 128 output channels");
  auto launch = [&](auto kern) {
    const dim3 grid((unsigned)blocks, (unsigned)ncol32);
    if (prof) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, st, ps.start, ps.stop, 0, a);
    else hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, a);
  };
  static const int g4 = [] { const char* e = getenv("PN_SPARSE_GROUP4"); return e ? atoi(e) : 1; }();
  if (g4 && (cin == 64 || cin == 128) && ncol32 >= 2) {      // block per group, K split over its waves: the wave -> (channel chunk, tap subset)
    // map of the kernel exists for 2 or 4 chunks of 32 input channels only; every other width takes the wave kernel, which loops over chunks
    const int g_cap = pn::cdiv(out_capacity, 32);
    const dim3 grid((unsigned)(xcd_bounds ? 8 * (4 * (2 * ((nblk_cap + 7) / 8) + 4)) : (g_cap + 7) / 8 * 8), 1);
    // 128 columns: two blocks of 64 columns per group (the input rows are gathered twice; blocks half as long, four per CU instead of three:
    // 380 -> 365 us on the bench frame's 128 -> 128 layers).  Heavy-groups-first block orders were tried and lose: the contiguous run of
    // groups an XCD walks shares neighbour rows in its L2 (global order by tap count: +15 %)
    static const int split = [] { const char* e = getenv("PN_SPARSE_G4SPLIT"); return e ? atoi(e) : 1; }();
    if (ncol32 == 4 && split) {
      const dim3 grid2(grid.x, 2);
      if (prof) hipExtLaunchKernelGGL(sparse_conv_group4_kernel<2>, grid2, dim3(256), 0, st, ps.start, ps.stop, 0, a);
      else hipLaunchKernelGGL(sparse_conv_group4_kernel<2>, grid2, dim3(256), 0, st, a);
    } else if (ncol32 == 4) {
      if (prof) hipExtLaunchKernelGGL(sparse_conv_group4_kernel<4>, grid, dim3(256), 0, st, ps.start, ps.stop, 0, a);
      else hipLaunchKernelGGL(sparse_conv_group4_kernel<4>, grid, dim3(256), 0, st, a);
    } else {
      if (prof) hipExtLaunchKernelGGL(sparse_conv_group4_kernel<2>, grid, dim3(256), 0, st, ps.start, ps.stop, 0, a);
      else hipLaunchKernelGGL(sparse_conv_group4_kernel<2>, grid, dim3(256), 0, st, a);
    }
    return pn::check_launch("sparse_conv_group4_kernel");
  }
  if (ncol32 == 4) launch(&sparse_conv_wave_kernel<4>);
  else if (ncol32 == 2) launch(&sparse_conv_wave_kernel<2>);
  else launch(&sparse_conv_wave_kernel<1>);
  return pn::check_launch("sparse_conv_wave_kernel");
}

}  // extern "C"
