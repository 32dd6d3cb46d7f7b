// Polar voxelization on the device: point decoration, grid indices, unique voxel ranks.
//
// torch.unique(grid_ind, dim=0) returns rows in lexicographic (b,z,theta,r) order, i.e. in
// increasing linear key ((b*Z+z)*T+theta)*R+r.  The rank of a voxel is therefore the number
// of occupied cells with a smaller key = an exclusive prefix sum of popcounts over an
// occupancy bitmap.  No sort, no host synchronisation; the voxel count stays on the device.
#include "pn_common.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;  // words per thread
constexpr int kScanTile = kScanThreads * kScanItems;

// ---------------------------------------------------------------------------- V0
__global__ void cart_to_polar_kernel(const float* __restrict__ cart, int n, int f_in, float* __restrict__ polar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = cart + (size_t)i * f_in;
  float* o = polar + (size_t)i * (f_in + 2);
  const float x = p[0], y = p[1];
  // numpy: sqrt(x**2 + y**2) with every fp32 op rounded separately (no FMA contraction)
  // the fp64 square root rounded once more to fp32 is the correctly rounded fp32 root (53 >= 2*24+2)
  o[0] = (float)sqrt((double)__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
  o[1] = (float)atan2((double)y, (double)x);
  o[2] = p[2];
  o[3] = x;
  o[4] = y;
  for (int k = 3; k < f_in; ++k) o[k + 2] = p[k];
}

// ---------------------------------------------------------------------------- V1
struct GridParams {
  float lo[3], vs[3];
  int g[3];  // R, T, Z
};

__device__ __forceinline__ int cell_index(float p, float lo, float vs, int g) {
  // floor(clip((p - lo) / vs, 0, g-1)) with IEEE fp32 subtract and divide
  // IEEE fp32 quotient via fp64 (double rounding is innocuous for division at 53 >= 2*24+2 bits)
  float q = (float)((double)__fsub_rn(p, lo) / (double)vs);
  const float hi = (float)(g - 1);
  q = q < 0.f ? 0.f : q;  // also maps -0.0 and NaN>... (NaN compares false -> kept, floor(NaN) UB; inputs are NaN-free)
  q = q > hi ? hi : q;
  return (int)floorf(q);
}

__global__ void grid_index_kernel(const float* __restrict__ pts, int stride, int n_cap, const int32_t* __restrict__ offs,
                                  int batch, GridParams gp, int64_t* __restrict__ grid_ind, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(offs[batch], n_cap);
  if (i >= n) return;
  int b = 0;
  while (b + 1 < batch && i >= offs[b + 1]) ++b;
  const float* p = pts + (size_t)i * stride;
  const int r = cell_index(p[0], gp.lo[0], gp.vs[0], gp.g[0]);
  const int t = cell_index(p[1], gp.lo[1], gp.vs[1], gp.g[1]);
  const int z = cell_index(p[2], gp.lo[2], gp.vs[2], gp.g[2]);
  if (grid_ind) {
    int64_t* o = grid_ind + (size_t)i * 4;
    o[0] = b; o[1] = z; o[2] = t; o[3] = r;
  }
  if (keys) keys[i] = (uint32_t)((((size_t)b * gp.g[2] + z) * gp.g[1] + t) * gp.g[0] + r);
}

__global__ void keys_from_grid_ind_kernel(const int64_t* __restrict__ gi, int n, int R, int T, int Z, int batch,
                                          uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t* g = gi + (size_t)i * 4;
  // clamp defensively: an out-of-grid index must not write outside the bitmap
  const int64_t b = min(max(g[0], (int64_t)0), (int64_t)batch - 1), z = min(max(g[1], (int64_t)0), (int64_t)Z - 1);
  const int64_t t = min(max(g[2], (int64_t)0), (int64_t)T - 1), r = min(max(g[3], (int64_t)0), (int64_t)R - 1);
  keys[i] = (uint32_t)(((b * Z + z) * T + t) * R + r);
}

// ---------------------------------------------------------------------------- bitmap unique
__global__ void mark_kernel(const uint32_t* __restrict__ keys, int n_cap, const int32_t* __restrict__ n_dev,
                            uint32_t* __restrict__ bitmap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  if (i >= n) return;
  const uint32_t k = keys[i];
  if (k == 0xffffffffu) return;  // dropped point (hard voxelization: out of range)
  atomicOr(&bitmap[k >> 5], 1u << (k & 31));
}


// phase 1: per-tile totals.  MODE 0: popcount of bitmap words; MODE 1: values of an int array
// limited to the first *limit entries.
template <int MODE>
__device__ __forceinline__ uint32_t scan_item(const uint32_t* __restrict__ src, size_t idx, size_t n, uint32_t limit) {
  if (idx >= n) return 0;
  if (MODE == 0) return __popc(src[idx]);
  return idx < limit ? src[idx] : 0u;
}

template <int MODE>
__global__ void scan_tile_totals_kernel(const uint32_t* __restrict__ src, size_t n, const int32_t* __restrict__ limit_dev,
                                        uint32_t* __restrict__ tile_total) {
  const uint32_t limit = limit_dev ? (uint32_t)*limit_dev : 0xffffffffu;
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) s += scan_item<MODE>(src, base + k, n, limit);
  uint32_t tot;
  pn::block_exclusive_scan<kScanThreads>(s, &tot);
  if (threadIdx.x == 0) tile_total[blockIdx.x] = tot;
}

// phase 2: one block scans the tile totals in place (exclusive) and publishes the grand total
__global__ void scan_tile_offsets_kernel(uint32_t* __restrict__ tile_total, int ntiles, int32_t* __restrict__ grand_total,
                                         int32_t* __restrict__ grand_total2) {
  uint32_t carry = 0;
  for (int base = 0; base < ntiles; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < ntiles ? tile_total[i] : 0;
    uint32_t tot;
    const uint32_t ex = pn::block_exclusive_scan<kScanThreads>(v, &tot);
    if (i < ntiles) tile_total[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    if (grand_total) *grand_total = (int32_t)carry;
    if (grand_total2) *grand_total2 = (int32_t)carry;
  }
}

// phase 3 (bitmap): word ranks + one output row per occupied cell
__global__ void scan_emit_voxels_kernel(const uint32_t* __restrict__ bitmap, size_t nwords,
                                        const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ word_rank,
                                        uint32_t* __restrict__ unq_keys, int64_t* __restrict__ unq, int R, int T, int Z,
                                        int v_cap) {
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t words[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    words[k] = base + k < nwords ? bitmap[base + k] : 0u;
    s += __popc(words[k]);
  }
  uint32_t tot;
  uint32_t rank = tile_offset[blockIdx.x] + pn::block_exclusive_scan<kScanThreads>(s, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k >= nwords) break;
    word_rank[base + k] = rank;
    uint32_t wbits = words[k];
    while (wbits) {
      const int bit = __ffs(wbits) - 1;
      wbits &= wbits - 1;
      const uint32_t key = (uint32_t)((base + k) * 32 + bit);
      if ((int)rank < v_cap) {
        unq_keys[rank] = key;
        if (unq) {
          int64_t* o = unq + (size_t)rank * 4;
          uint32_t q = key;
          o[3] = q % R; q /= R;
          o[2] = q % T; q /= T;
          o[1] = q % Z; q /= Z;
          o[0] = q;
        }
      }
      ++rank;
    }
  }
}

// two zero fills in one launch (16-byte granules; both ranges 16-byte aligned, sizes multiples of 4)
__global__ void zero2_kernel(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < na) a[i] = 0u;
    else b[i - na] = 0u;
  }
}

__global__ void rank_points_kernel(const uint32_t* __restrict__ keys, int n_cap, const int32_t* __restrict__ n_dev,
                                   const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_rank,
                                   int32_t* __restrict__ inv, int32_t* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  if (i >= n) return;
  const uint32_t k = keys[i];
  if (k == 0xffffffffu) {
    inv[i] = -1;
    return;
  }
  const uint32_t w = k >> 5, bit = k & 31;
  const int32_t r = (int32_t)(word_rank[w] + __popc(bitmap[w] & ((1u << bit) - 1u)));
  inv[i] = r;
  atomicAdd(&cnt[r], 1);
}

// phase 3 (counts): voxel_start = exclusive scan of unq_cnt, voxel_start[V] = n
__global__ void scan_emit_starts_kernel(const uint32_t* __restrict__ cnt, size_t n_cap, const int32_t* __restrict__ v_dev,
                                        const uint32_t* __restrict__ tile_offset, int32_t* __restrict__ voxel_start) {
  const uint32_t V = (uint32_t)*v_dev;
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t vals[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    vals[k] = scan_item<1>(cnt, base + k, n_cap, V);
    s += vals[k];
  }
  uint32_t tot;
  uint32_t run = tile_offset[blockIdx.x] + pn::block_exclusive_scan<kScanThreads>(s, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k <= V && base + k <= n_cap) voxel_start[base + k] = (int32_t)run;
    run += vals[k];
  }
}

__global__ void bucket_fill_kernel(const int32_t* __restrict__ inv, int n_cap, const int32_t* __restrict__ n_dev,
                                   const int32_t* __restrict__ voxel_start, int32_t* __restrict__ cursor,
                                   int32_t* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
  if (i >= n) return;
  const int v = inv[i];
  order[voxel_start[v] + atomicAdd(&cursor[v], 1)] = i;
}


// ---------------------------------------------------------------------------- fused frame index (V0 + V1 + unique + bucketing)
// Three launches instead of thirteen for the per-frame index of the dynamic path (the stage is bound by the ~5 us a dependent
// launch costs inside a replayed graph, not by bandwidth):
//   1. cart -> polar, grid index, key, and pos = atomicAdd(cell_count[key], 1): the point's slot inside its cell
//   2. ONE single-pass scan (decoupled look-back) over the cell counts: an occupied cell's voxel rank = number of occupied
//      cells with a smaller key (= row order of torch.unique(dim=0)) and its first point slot = number of points in smaller
//      cells; emits unq_keys / voxel_start, leaves the start in cell_count[key]
//   3. order[cell_count[key] + pos] = point
// cell_count (one uint32 per grid cell) must be all zero on entry; pn_clear_frame_cells zeroes the frame's cells again (a
// sparse clear next to the canvas clear), so a persistent buffer never needs a dense fill.  Same outputs as
// pn_polar_grid_index_f32 + pn_unique_rank_bitmap + pn_bucket_points (order inside a voxel is unspecified in both).
__global__ void fused_polar_index_kernel(const float* __restrict__ cart, int n_cap, int f_in, const int32_t* __restrict__ offs, int batch,
                                         GridParams gp, float* __restrict__ polar, uint32_t* __restrict__ keys, int32_t* __restrict__ pos,
                                         uint32_t* __restrict__ cell_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(offs[batch], n_cap);
  if (i >= n) return;
  int b = 0;
  while (b + 1 < batch && i >= offs[b + 1]) ++b;
  const float* p = cart + (size_t)i * f_in;
  float* o = polar + (size_t)i * (f_in + 2);
  const float x = p[0], y = p[1], zc = p[2];
  const float rho = (float)sqrt((double)__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));   // as cart_to_polar_kernel
  const float phi = (float)atan2((double)y, (double)x);
  o[0] = rho; o[1] = phi; o[2] = zc; o[3] = x; o[4] = y;
  for (int k = 3; k < f_in; ++k) o[k + 2] = p[k];
  const int r = cell_index(rho, gp.lo[0], gp.vs[0], gp.g[0]);
  const int t = cell_index(phi, gp.lo[1], gp.vs[1], gp.g[1]);
  const int z = cell_index(zc, gp.lo[2], gp.vs[2], gp.g[2]);
  const uint32_t key = (uint32_t)((((size_t)b * gp.g[2] + z) * gp.g[1] + t) * gp.g[0] + r);
  keys[i] = key;
  pos[i] = (int32_t)atomicAdd(&cell_count[key], 1u);
}

// r6: the same from the RAW sweeps of a multi-sweep frame (BASELINE configs[4]): remove_close, the rigid transform and the time lag of
// pn_accumulate_sweeps_f32 (assign.hip; det3d/datasets/pipelines/loading.py:215-260) applied on the way, WITHOUT compacting the kept points first
// -- a removed point gets the key 0xffffffff (no cell has it: cells < 2^32), joins no cell, and order_fill_kernel skips it; its polar row is
// never read.  Three launches (count, offsets, scatter: 36 us at 300 k points) and the (n, 5) Cartesian copy are gone.  Same arithmetic per kept
// point as the two steps one after the other (double-precision transform rounded to float, then cart_to_polar_kernel's).
__global__ void fused_sweeps_index_kernel(const float* __restrict__ raw, int n_cap, int in_cols, const int32_t* __restrict__ sweep_offs, int sweeps,
                                          const double* __restrict__ mats, const float* __restrict__ lags, float radius, GridParams gp,
                                          float* __restrict__ polar, uint32_t* __restrict__ keys, int32_t* __restrict__ pos,
                                          uint32_t* __restrict__ cell_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(sweep_offs[sweeps], n_cap);
  if (i >= n) return;
  int s = 0;
  while (s + 1 < sweeps && i >= sweep_offs[s + 1]) ++s;
  const float* p = raw + (size_t)i * in_cols;
  float x = p[0], y = p[1], zc = p[2];
  if (s != 0) {      // the key frame is taken as it is
    if (fabsf(x) < radius && fabsf(y) < radius) {
      keys[i] = 0xffffffffu;
      pos[i] = -1;
      return;
    }
    const double* m = mats + (size_t)s * 16;
    const double xd = x, yd = y, zd = zc;
    x = (float)(m[0] * xd + m[1] * yd + m[2] * zd + m[3]);
    y = (float)(m[4] * xd + m[5] * yd + m[6] * zd + m[7]);
    zc = (float)(m[8] * xd + m[9] * yd + m[10] * zd + m[11]);
  }
  float* o = polar + (size_t)i * 7;
  const float rho = (float)sqrt((double)__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));   // as cart_to_polar_kernel
  const float phi = (float)atan2((double)y, (double)x);
  o[0] = rho; o[1] = phi; o[2] = zc; o[3] = x; o[4] = y; o[5] = p[3]; o[6] = lags[s];
  const int r = cell_index(rho, gp.lo[0], gp.vs[0], gp.g[0]);
  const int t = cell_index(phi, gp.lo[1], gp.vs[1], gp.g[1]);
  const int z = cell_index(zc, gp.lo[2], gp.vs[2], gp.g[2]);
  const uint32_t key = (uint32_t)(((size_t)z * gp.g[1] + t) * gp.g[0] + r);
  keys[i] = key;
  pos[i] = (int32_t)atomicAdd(&cell_count[key], 1u);
}

// tile state of the look-back scan: status (2 bits: 0 = nothing yet, 1 = tile aggregate, 2 = inclusive prefix) | occupied
// cells (30 bits) | points (32 bits).  Written / read with agent-scope relaxed atomics: the tiles of one launch sit on different
// XCDs whose L2s are not coherent with each other.
__device__ __forceinline__ unsigned long long pack_state(unsigned status, uint32_t nz, uint32_t pts) {
  return ((unsigned long long)status << 62) | ((unsigned long long)nz << 32) | pts;
}

__global__ __launch_bounds__(kScanThreads) void cell_scan_kernel(uint32_t* __restrict__ cell_count, size_t ncells, int ntiles,
                                                                 unsigned long long* __restrict__ tile_state, uint32_t* __restrict__ counters,
                                                                 uint32_t* __restrict__ unq_keys, int32_t* __restrict__ voxel_start,
                                                                 int32_t* __restrict__ num_voxels, int v_cap, int32_t* __restrict__ row_start, int row_cells) {
  __shared__ uint32_t s_tile, s_pre_nz, s_pre_pts;
  const int tid = threadIdx.x;
  if (tid == 0) s_tile = __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tiles in start order
  __syncthreads();
  const int tile = (int)s_tile;
  const size_t base = (size_t)tile * kScanTile + (size_t)tid * kScanItems;
  uint32_t c[kScanItems];
  uint32_t nz = 0, pts = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    c[k] = base + k < ncells ? cell_count[base + k] : 0u;
    nz += c[k] != 0u;
    pts += c[k];
  }
  uint32_t tot_nz, tot_pts;
  const uint32_t ex_nz = pn::block_exclusive_scan<kScanThreads>(nz, &tot_nz);
  const uint32_t ex_pts = pn::block_exclusive_scan<kScanThreads>(pts, &tot_pts);
  if (tid < 64) {   // wave 0: publish the aggregate, then look back 64 tiles at a time
    uint32_t pre_nz = 0, pre_pts = 0;
    if (tile > 0) {
      if (tid == 0) __hip_atomic_store(&tile_state[tile], pack_state(1u, tot_nz, tot_pts), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int hi = tile - 1;   // nearest predecessor not yet accounted for
      while (hi >= 0) {
        const int j = hi - tid;
        unsigned long long st = pack_state(2u, 0u, 0u);   // lanes past tile 0 behave like an (empty) inclusive prefix
        if (j >= 0) st = __hip_atomic_load(&tile_state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned status = (unsigned)(st >> 62);
        const unsigned long long none = __ballot(status == 0u), pref = __ballot(status == 2u);
        // usable run: lanes 0 .. first lane that is a prefix, provided no lane before it is still empty
        const int first_pref = pref ? __ffsll((long long)pref) - 1 : 64;
        const int first_none = none ? __ffsll((long long)none) - 1 : 64;
        if (first_none < first_pref) {   // a predecessor in the window has not published yet: wait for it
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        const bool use = tid <= first_pref;   // first_pref == 64: all 64 are aggregates
        uint32_t a_nz = use ? (uint32_t)(st >> 32) & 0x3fffffffu : 0u, a_pts = use ? (uint32_t)st : 0u;
        pre_nz += pn::wave_sum(a_nz);
        pre_pts += pn::wave_sum(a_pts);
        if (first_pref < 64) break;
        hi -= 64;
      }
    }
    if (tid == 0) {
      __hip_atomic_store(&tile_state[tile], pack_state(2u, pre_nz + tot_nz, pre_pts + tot_pts), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_pre_nz = pre_nz;
      s_pre_pts = pre_pts;
      if (tile == ntiles - 1) {   // grand totals
        const uint32_t V = min(pre_nz + tot_nz, (uint32_t)v_cap);
        *num_voxels = (int32_t)V;
        voxel_start[V] = (int32_t)(pre_pts + tot_pts);
        if (row_start) row_start[ncells / (size_t)row_cells] = (int32_t)V;
      }
    }
  }
  __syncthreads();
  uint32_t rank = s_pre_nz + ex_nz, start = s_pre_pts + ex_pts;
  // r6: the voxels are emitted in key order, i.e. grid row by grid row (a row = row_cells consecutive cells, a multiple of a thread's
  // kScanItems): row_start[g] = number of voxels in rows before g -- the row runs of unq_keys that pillar_rows.hip walks, for free
  if (row_start && base < ncells && base % (size_t)row_cells == 0) row_start[base / (size_t)row_cells] = (int32_t)min(rank, (uint32_t)v_cap);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (c[k] != 0u) {
      if ((int)rank < v_cap) {
        unq_keys[rank] = (uint32_t)(base + k);
        voxel_start[rank] = (int32_t)start;
      }
      cell_count[base + k] = start;   // the first point slot of the cell, read by order_fill_kernel
      ++rank;
      start += c[k];
    }
  }
  // self-cleaning: the block that finishes last resets the scan state for the next launch (a replayed graph has no memset node
  // to rely on); every other block is past its look-back by the time it takes its ticket
  // Every block's tile_state stores (write-through agent-scope stores made by this same thread 0) are DRAINED before its ticket
  // (s_waitcnt vmcnt(0): the store has been acknowledged), so the block that draws the last ticket zeroes the state after all of
  // them have landed -- a prefix store that lands late cannot survive the reset and reach the next replay.  (An acquire-release
  // ticket does the same with an L2 write-back per block: +3.6 us on the 64-block scan of a nuScenes sweep.)
  __syncthreads();
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_tile = __hip_atomic_fetch_add(&counters[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if ((int)s_tile == ntiles - 1) {
    for (int j = tid; j < ntiles; j += kScanThreads) tile_state[j] = 0ull;
    if (tid < 2) counters[tid] = 0u;
  }
}

__global__ void order_fill_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ pos, int n_cap, const int32_t* __restrict__ n_dev,
                                  const uint32_t* __restrict__ cell_count, int32_t* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(*n_dev, n_cap);
  if (i >= n) return;
  const uint32_t key = keys[i];
  if (key == 0xffffffffu) return;      // a point fused_sweeps_index_kernel removed
  order[cell_count[key] + (uint32_t)pos[i]] = i;
}

// sparse clear at the end of a frame: the canvas cells of the frame's voxels (16 bytes per thread, c / 4 threads per cell) and
// their cell_count entries
__global__ void clear_frame_cells_kernel(const uint32_t* __restrict__ ukeys, const int32_t* __restrict__ v_dev, int v_cap, int c4, int Z, int T,
                                         int R, float* __restrict__ canvas, uint32_t* __restrict__ cell_count) {
  const int V = min(*v_dev, v_cap);
  const int per = c4 > 0 ? c4 : 1;
  const size_t total = (size_t)V * per;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(i / per), k = (int)(i - (size_t)v * per);
    const uint32_t key0 = ukeys[v];
    if (canvas) {
      uint32_t key = key0;
      const int ri = key % R; key /= R;
      const int ti = key % T; key /= T;
      const int bi = key / Z;
      reinterpret_cast<float4*>(canvas)[(((size_t)bi * T + ti) * R + ri) * c4 + k] = float4{0.f, 0.f, 0.f, 0.f};
    }
    if (k == 0 && cell_count) cell_count[key0] = 0u;
  }
}

// ---------------------------------------------------------------------------- V2 hard voxelization
// Reference semantics (point_cloud_ops.py:7-72): points are visited in order; a point outside the
// grid is dropped; a voxel's id is its order of first appearance; at most max_voxels voxels are
// created (later new voxels are refused, existing ones keep accepting); a voxel keeps its first
// max_points points in point order.  Deterministic parallel formulation:
//   rank r of the cell in key order (bitmap scan)  ->  first[r] = min point index (atomicMin)
//   voxel id = number of "first" points with a smaller index (prefix sum over the points)
//   slot k of voxel r = k-th smallest point index of the cell: max_points rounds of atomicMin
__global__ void hard_keys_kernel(const float* __restrict__ pts, int n, int stride, GridParams gp, uint32_t* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + (size_t)i * stride;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q = floorf((float)((double)__fsub_rn(p[a], gp.lo[a]) / (double)gp.vs[a]));  // IEEE fp32 quotient
    ok = ok && q >= 0.f && q < (float)gp.g[a];
    c[a] = (int)q;
  }
  keys[i] = ok ? (uint32_t)(((size_t)c[2] * gp.g[1] + c[1]) * gp.g[0] + c[0]) : 0xffffffffu;
}

__global__ void hard_first_kernel(const int32_t* __restrict__ inv, int n, int32_t* __restrict__ first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || inv[i] < 0) return;
  atomicMin(&first[inv[i]], i);
}

__global__ void hard_flag_kernel(const int32_t* __restrict__ inv, int n, const int32_t* __restrict__ first,
                                 uint32_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (inv[i] >= 0 && first[inv[i]] == i) ? 1u : 0u;
}

// exclusive scan of flag[] over the points (third phase), written to vid_at[i]
__global__ void scan_emit_flags_kernel(const uint32_t* __restrict__ flag, size_t n, const uint32_t* __restrict__ tile_offset,
                                       int32_t* __restrict__ vid_at) {
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
  uint32_t vals[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    vals[k] = base + k < n ? flag[base + k] : 0u;
    s += vals[k];
  }
  uint32_t tot;
  uint32_t run = tile_offset[blockIdx.x] + pn::block_exclusive_scan<kScanThreads>(s, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) vid_at[base + k] = (int32_t)run;
    run += vals[k];
  }
}

// one selection round: sel_k[r] = smallest point index of cell r that is larger than sel_{k-1}[r]
__global__ void hard_select_kernel(const int32_t* __restrict__ inv, int n, const int32_t* __restrict__ prev,
                                   int32_t* __restrict__ cur) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = inv[i];
  if (r < 0) return;
  if (prev == nullptr || i > prev[r]) atomicMin(&cur[r], i);
}

__global__ void hard_fill_int_kernel(int32_t* __restrict__ p, size_t n, int32_t v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// gather: one thread per (cell rank, slot, feature)
__global__ void hard_gather_kernel(const float* __restrict__ pts, int stride, int f, const int32_t* __restrict__ v_dev,
                                   int v_cap, const int32_t* __restrict__ first, const int32_t* __restrict__ vid_at,
                                   const int32_t* __restrict__ sel, int max_points, int max_voxels,
                                   const uint32_t* __restrict__ ukeys, const int32_t* __restrict__ cnt, int R, int T,
                                   float* __restrict__ voxels, int32_t* __restrict__ coors, int32_t* __restrict__ num,
                                   int32_t* __restrict__ num_voxels) {
  const int V = min(*v_dev, v_cap);
  const size_t total = (size_t)V * max_points * f;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % f);
    const int slot = (int)((idx / f) % max_points);
    const int r = (int)(idx / ((size_t)f * max_points));
    const int vid = vid_at[first[r]];
    if (vid >= max_voxels) continue;
    const int src = sel[(size_t)slot * v_cap + r];
    if (src != 0x7fffffff) voxels[((size_t)vid * max_points + slot) * f + k] = pts[(size_t)src * stride + k];
    if (slot == 0 && k == 0) {
      uint32_t q = ukeys[r];
      coors[3 * vid + 2] = q % R; q /= R;
      coors[3 * vid + 1] = q % T; q /= T;
      coors[3 * vid + 0] = q;
      num[vid] = min(cnt[r], max_points);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *num_voxels = min(V, max_voxels);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct UniqueWs {
  size_t nwords, ntiles;
  size_t off_bitmap, off_rank, off_tiles, off_keys, total;
  UniqueWs(uint64_t cells, int n_cap) {
    nwords = (size_t)((cells + 31) / 32);
    ntiles = (nwords + kScanTile - 1) / kScanTile;
    off_bitmap = 0;
    off_rank = align256(off_bitmap + nwords * 4);
    off_tiles = align256(off_rank + nwords * 4);
    off_keys = align256(off_tiles + ntiles * 4);
    total = align256(off_keys + (size_t)n_cap * 4);
  }
};

// feats [batch][seg_rows][c], coors [batch][seg_rows][3], counts[batch] (clamped to seg_rows) -> rows of sample b at prefix(b) + r
__global__ __launch_bounds__(256) void concat_segments_kernel(const float* __restrict__ feats, const int32_t* __restrict__ coors,
                                                              const int32_t* __restrict__ counts, int batch, int seg_rows, int c,
                                                              float* __restrict__ feats_out, int32_t* __restrict__ coords4_out, int32_t* __restrict__ total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const int b = (int)(i / seg_rows), r = (int)(i - (long long)b * seg_rows);
  int prefix = 0, mine = 0, all = 0;
  for (int k = 0; k < batch; ++k) {
    const int n = min(max(counts[k], 0), seg_rows);
    if (k < b) prefix += n;
    if (k == b) mine = n;
    all += n;
  }
  if (i == 0) *total = all;
  if (b >= batch || r >= mine) return;
  const size_t src = (size_t)b * seg_rows + r, dst = (size_t)prefix + r;
  for (int k = 0; k < c; ++k) feats_out[dst * c + k] = feats[src * c + k];
  coords4_out[dst * 4] = b;
  coords4_out[dst * 4 + 1] = coors[src * 3];
  coords4_out[dst * 4 + 2] = coors[src * 3 + 1];
  coords4_out[dst * 4 + 3] = coors[src * 3 + 2];
}

}  // namespace

extern "C" {

int pn_cart_to_polar_f32(const float* cart, int n, int f_in, float* polar, pn_stream_t stream) {
  PN_REQUIRE(cart && polar && n >= 0 && f_in >= 3, "cart_to_polar: bad arguments");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(cart_to_polar_kernel, dim3(pn::cdiv(n, 256)), dim3(256), 0, pn::S(stream), cart, n, f_in, polar);
  return pn::check_launch("cart_to_polar_kernel");
}

int pn_polar_grid_index_f32(const float* points, int point_stride, int n_capacity, const int32_t* sample_offsets,
                            int batch, const float* range_lo, const float* voxel_size, const int32_t* grid,
                            int64_t* grid_ind, uint32_t* keys, pn_stream_t stream) {
  PN_REQUIRE(points && sample_offsets && range_lo && voxel_size && grid, "grid_index: null pointer");
  PN_REQUIRE(point_stride >= 3 && n_capacity >= 0 && batch >= 1, "grid_index: bad sizes");
  PN_REQUIRE((uint64_t)batch * grid[0] * grid[1] * grid[2] < (1ull << 32), "grid_index: more than 2^32 cells");
  if (n_capacity == 0) return PN_OK;
  GridParams gp;
  for (int k = 0; k < 3; ++k) {
    gp.lo[k] = range_lo[k];
    gp.vs[k] = voxel_size[k];
    gp.g[k] = grid[k];
  }
  hipLaunchKernelGGL(grid_index_kernel, dim3(pn::cdiv(n_capacity, 256)), dim3(256), 0, pn::S(stream), points,
                     point_stride, n_capacity, sample_offsets, batch, gp, grid_ind, keys);
  return pn::check_launch("grid_index_kernel");
}

int pn_keys_from_grid_ind(const int64_t* grid_ind, int n, const int32_t* grid, int batch, uint32_t* keys,
                          pn_stream_t stream) {
  PN_REQUIRE(grid_ind && grid && keys && n >= 0 && batch >= 1, "keys_from_grid_ind: bad arguments");
  PN_REQUIRE((uint64_t)batch * grid[0] * grid[1] * grid[2] < (1ull << 32), "keys_from_grid_ind: more than 2^32 cells");
  if (n == 0) return PN_OK;
  hipLaunchKernelGGL(keys_from_grid_ind_kernel, dim3(pn::cdiv(n, 256)), dim3(256), 0, pn::S(stream), grid_ind, n,
                     grid[0], grid[1], grid[2], batch, keys);
  return pn::check_launch("keys_from_grid_ind_kernel");
}

size_t pn_unique_workspace_bytes(uint64_t num_cells, int n_capacity) { return UniqueWs(num_cells, n_capacity).total; }

const uint32_t* pn_unique_keys_ptr(const void* workspace, uint64_t num_cells, int n_capacity) {
  return reinterpret_cast<const uint32_t*>(static_cast<const char*>(workspace) + UniqueWs(num_cells, n_capacity).off_keys);
}

int pn_unique_rank_bitmap(const uint32_t* keys, int n_capacity, const int32_t* n_dev, uint64_t num_cells,
                          const int32_t* grid, int64_t* unq, int32_t* unq_inv, int32_t* unq_cnt, int32_t* num_voxels,
                          void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(keys && grid && unq_inv && unq_cnt && num_voxels && workspace, "unique: null pointer");
  PN_REQUIRE(num_cells > 0 && num_cells < (1ull << 32) && n_capacity >= 0, "unique: bad sizes");
  UniqueWs ws(num_cells, n_capacity);
  if (workspace_bytes < ws.total) return pn::fail(PN_ERR_WORKSPACE, "unique: workspace %zu < %zu", workspace_bytes, ws.total);
  hipStream_t st = pn::S(stream);
  char* base = static_cast<char*>(workspace);
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(base + ws.off_bitmap);
  uint32_t* rank = reinterpret_cast<uint32_t*>(base + ws.off_rank);
  uint32_t* tiles = reinterpret_cast<uint32_t*>(base + ws.off_tiles);
  uint32_t* ukeys = reinterpret_cast<uint32_t*>(base + ws.off_keys);
  {  // bitmap and per-voxel counts cleared by one launch
    const size_t na = ws.nwords, nb = (size_t)(n_capacity > 0 ? n_capacity : 0);
    hipLaunchKernelGGL(zero2_kernel, dim3((unsigned)std::min<size_t>(2048, (na + nb + 255) / 256)), dim3(256), 0, st, bitmap, na,
                       reinterpret_cast<uint32_t*>(unq_cnt), nb);
  }
  const int pblocks = pn::cdiv(n_capacity > 0 ? n_capacity : 1, 256);
  if (n_capacity > 0) hipLaunchKernelGGL(mark_kernel, dim3(pblocks), dim3(256), 0, st, keys, n_capacity, n_dev, bitmap);
  // (a single-block scan -- one launch instead of three -- was measured SLOWER here: 8192 words + 28k emitted rows on one CU
  //  take longer than three dependent multi-block dispatches)
  hipLaunchKernelGGL(scan_tile_totals_kernel<0>, dim3((unsigned)ws.ntiles), dim3(kScanThreads), 0, st, bitmap, ws.nwords,
                     (const int32_t*)nullptr, tiles);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(kScanThreads), 0, st, tiles, (int)ws.ntiles, num_voxels,
                     (int32_t*)nullptr);
  hipLaunchKernelGGL(scan_emit_voxels_kernel, dim3((unsigned)ws.ntiles), dim3(kScanThreads), 0, st, bitmap, ws.nwords,
                     tiles, rank, ukeys, unq, grid[0], grid[1], grid[2], n_capacity);
  if (n_capacity > 0)
    hipLaunchKernelGGL(rank_points_kernel, dim3(pblocks), dim3(256), 0, st, keys, n_capacity, n_dev, bitmap, rank, unq_inv,
                       unq_cnt);
  return pn::check_launch("unique_rank_bitmap");
}

size_t pn_voxel_index_fused_state_bytes(uint64_t num_cells) {
  const size_t ntiles = (size_t)((num_cells + kScanTile - 1) / kScanTile);
  return align256(ntiles * 8) + 256;
}

static int voxel_index_fused(const float* cart, int n_capacity, int f_in, const int32_t* sample_offsets, int batch, const float* range_lo,
                             const float* voxel_size, const int32_t* grid, float* polar, uint32_t* keys, int32_t* pos, uint32_t* cell_count,
                             void* scan_state, size_t scan_state_bytes, uint32_t* unq_keys, int32_t* voxel_start, int32_t* order,
                             int32_t* num_voxels, int32_t* row_start, pn_stream_t stream) {
  PN_REQUIRE(sample_offsets && range_lo && voxel_size && grid && keys && pos && cell_count && scan_state && unq_keys && voxel_start && order &&
                 num_voxels && (n_capacity == 0 || (cart && polar)), "voxel_index_fused: null pointer");
  PN_REQUIRE(f_in >= 3 && n_capacity >= 0 && batch >= 1, "voxel_index_fused: bad sizes");
  const uint64_t cells = (uint64_t)batch * grid[0] * grid[1] * grid[2];
  PN_REQUIRE(cells > 0 && cells < (1ull << 32), "voxel_index_fused: more than 2^32 cells");
  PN_REQUIRE((uint64_t)n_capacity < (1ull << 30), "voxel_index_fused: too many points");
  if (scan_state_bytes < pn_voxel_index_fused_state_bytes(cells)) return pn::fail(PN_ERR_WORKSPACE, "voxel_index_fused: scan state too small");
  GridParams gp;
  for (int k = 0; k < 3; ++k) {
    gp.lo[k] = range_lo[k];
    gp.vs[k] = voxel_size[k];
    gp.g[k] = grid[k];
  }
  hipStream_t st = pn::S(stream);
  const int ntiles = (int)((cells + kScanTile - 1) / kScanTile);
  unsigned long long* tile_state = static_cast<unsigned long long*>(scan_state);
  uint32_t* counters = reinterpret_cast<uint32_t*>(static_cast<char*>(scan_state) + align256((size_t)ntiles * 8));
  const int pblocks = pn::cdiv(n_capacity > 0 ? n_capacity : 1, 256);
  if (n_capacity > 0)
    hipLaunchKernelGGL(fused_polar_index_kernel, dim3(pblocks), dim3(256), 0, st, cart, n_capacity, f_in, sample_offsets, batch, gp, polar, keys,
                       pos, cell_count);
  PN_REQUIRE(!row_start || grid[0] % kScanItems == 0, "voxel_index_fused_rows: the grid's first axis must be a multiple of 8 cells");
  hipLaunchKernelGGL(cell_scan_kernel, dim3(ntiles), dim3(kScanThreads), 0, st, cell_count, (size_t)cells, ntiles, tile_state, counters, unq_keys,
                     voxel_start, num_voxels, n_capacity, row_start, grid[0]);
  if (n_capacity > 0)
    hipLaunchKernelGGL(order_fill_kernel, dim3(pblocks), dim3(256), 0, st, keys, pos, n_capacity, sample_offsets + batch, cell_count, order);
  return pn::check_launch("voxel_index_fused");
}

int pn_voxel_index_fused_f32(const float* cart, int n_capacity, int f_in, const int32_t* sample_offsets, int batch, const float* range_lo,
                             const float* voxel_size, const int32_t* grid, float* polar, uint32_t* keys, int32_t* pos, uint32_t* cell_count,
                             void* scan_state, size_t scan_state_bytes, uint32_t* unq_keys, int32_t* voxel_start, int32_t* order,
                             int32_t* num_voxels, pn_stream_t stream) {
  return voxel_index_fused(cart, n_capacity, f_in, sample_offsets, batch, range_lo, voxel_size, grid, polar, keys, pos, cell_count, scan_state,
                           scan_state_bytes, unq_keys, voxel_start, order, num_voxels, nullptr, stream);
}

// the same, also leaving row_start [batch * grid[2] * grid[1] + 1]: the number of voxels in the grid rows (runs of grid[0] cells) before row g --
// unq_keys is sorted by key, so a row's voxels are the run [row_start[g], row_start[g + 1]).  grid[0] % 8 == 0.
int pn_voxel_index_fused_rows_f32(const float* cart, int n_capacity, int f_in, const int32_t* sample_offsets, int batch, const float* range_lo,
                                  const float* voxel_size, const int32_t* grid, float* polar, uint32_t* keys, int32_t* pos, uint32_t* cell_count,
                                  void* scan_state, size_t scan_state_bytes, uint32_t* unq_keys, int32_t* voxel_start, int32_t* order,
                                  int32_t* num_voxels, int32_t* row_start, pn_stream_t stream) {
  PN_REQUIRE(row_start, "voxel_index_fused_rows: null pointer");
  return voxel_index_fused(cart, n_capacity, f_in, sample_offsets, batch, range_lo, voxel_size, grid, polar, keys, pos, cell_count, scan_state,
                           scan_state_bytes, unq_keys, voxel_start, order, num_voxels, row_start, stream);
}

int pn_voxel_index_fused_sweeps_f32(const float* raw, int n_capacity, int raw_cols, const int32_t* sweep_offsets, int sweeps, const double* transforms,
                                    const float* time_lags, float min_distance, const float* range_lo, const float* voxel_size, const int32_t* grid,
                                    float* polar, uint32_t* keys, int32_t* pos, uint32_t* cell_count, void* scan_state, size_t scan_state_bytes,
                                    uint32_t* unq_keys, int32_t* voxel_start, int32_t* order, int32_t* num_voxels, int32_t* row_start,
                                    pn_stream_t stream) {
  PN_REQUIRE(raw && sweep_offsets && transforms && time_lags && range_lo && voxel_size && grid && polar && keys && pos && cell_count && scan_state &&
                 unq_keys && voxel_start && order && num_voxels, "voxel_index_fused_sweeps: null pointer");
  PN_REQUIRE(n_capacity >= 1 && raw_cols >= 4 && sweeps >= 1, "voxel_index_fused_sweeps: bad sizes");
  const uint64_t cells = (uint64_t)grid[0] * grid[1] * grid[2];
  PN_REQUIRE(cells > 0 && cells < 0xffffffffull, "voxel_index_fused_sweeps: more than 2^32 - 1 cells");
  PN_REQUIRE((uint64_t)n_capacity < (1ull << 30), "voxel_index_fused_sweeps: too many points");
  PN_REQUIRE(!row_start || grid[0] % kScanItems == 0, "voxel_index_fused_sweeps: the grid's first axis must be a multiple of 8 cells for row_start");
  if (scan_state_bytes < pn_voxel_index_fused_state_bytes(cells)) return pn::fail(PN_ERR_WORKSPACE, "voxel_index_fused_sweeps: scan state too small");
  GridParams gp;
  for (int k = 0; k < 3; ++k) {
    gp.lo[k] = range_lo[k];
    gp.vs[k] = voxel_size[k];
    gp.g[k] = grid[k];
  }
  hipStream_t st = pn::S(stream);
  const int ntiles = (int)((cells + kScanTile - 1) / kScanTile);
  unsigned long long* tile_state = static_cast<unsigned long long*>(scan_state);
  uint32_t* counters = reinterpret_cast<uint32_t*>(static_cast<char*>(scan_state) + align256((size_t)ntiles * 8));
  const int pblocks = pn::cdiv(n_capacity, 256);
  hipLaunchKernelGGL(fused_sweeps_index_kernel, dim3(pblocks), dim3(256), 0, st, raw, n_capacity, raw_cols, sweep_offsets, sweeps, transforms, time_lags,
                     min_distance, gp, polar, keys, pos, cell_count);
  hipLaunchKernelGGL(cell_scan_kernel, dim3(ntiles), dim3(kScanThreads), 0, st, cell_count, (size_t)cells, ntiles, tile_state, counters, unq_keys,
                     voxel_start, num_voxels, n_capacity, row_start, grid[0]);
  hipLaunchKernelGGL(order_fill_kernel, dim3(pblocks), dim3(256), 0, st, keys, pos, n_capacity, sweep_offsets + sweeps, cell_count, order);
  return pn::check_launch("voxel_index_fused_sweeps");
}

int pn_clear_frame_cells(const uint32_t* unq_keys, const int32_t* num_voxels, int v_capacity, const int32_t* grid, int c, float* canvas,
                         uint32_t* cell_count, pn_stream_t stream) {
  PN_REQUIRE(unq_keys && num_voxels && grid && (canvas || cell_count), "clear_frame_cells: bad arguments");
  PN_REQUIRE(!canvas || (c >= 4 && c % 4 == 0 && ((uintptr_t)canvas & 15) == 0), "clear_frame_cells: canvas needs c % 4 == 0 and 16-byte alignment");
  if (v_capacity == 0) return PN_OK;
  const int c4 = canvas ? c / 4 : 0;
  const size_t total = (size_t)v_capacity * (c4 > 0 ? c4 : 1);
  hipLaunchKernelGGL(clear_frame_cells_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, pn::S(stream), unq_keys,
                     num_voxels, v_capacity, c4, grid[2], grid[1], grid[0], canvas, cell_count);
  return pn::check_launch("clear_frame_cells_kernel");
}

size_t pn_bucket_workspace_bytes(int n_capacity) {
  const size_t ntiles = ((size_t)n_capacity + 1 + kScanTile - 1) / kScanTile;
  return align256(ntiles * 4) + align256((size_t)(n_capacity + 1) * 4);
}

int pn_bucket_points(const int32_t* unq_inv, const int32_t* unq_cnt, int n_capacity, const int32_t* n_dev,
                     const int32_t* num_voxels, int32_t* voxel_start, int32_t* order, void* workspace,
                     size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(unq_inv && unq_cnt && num_voxels && voxel_start && order && workspace, "bucket: null pointer");
  if (workspace_bytes < pn_bucket_workspace_bytes(n_capacity))
    return pn::fail(PN_ERR_WORKSPACE, "bucket: workspace %zu < %zu", workspace_bytes, pn_bucket_workspace_bytes(n_capacity));
  if (n_capacity == 0) return PN_OK;
  hipStream_t st = pn::S(stream);
  const size_t nitems = (size_t)n_capacity + 1;
  const size_t ntiles = (nitems + kScanTile - 1) / kScanTile;
  uint32_t* tiles = static_cast<uint32_t*>(workspace);
  int32_t* cursor = reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + align256(ntiles * 4));
  if (int rc = pn::zero_async(cursor, (size_t)n_capacity * 4, st)) return rc;
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(unq_cnt);
  hipLaunchKernelGGL(scan_tile_totals_kernel<1>, dim3((unsigned)ntiles), dim3(kScanThreads), 0, st, cnt, (size_t)n_capacity,
                     num_voxels, tiles);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(kScanThreads), 0, st, tiles, (int)ntiles, (int32_t*)nullptr,
                     (int32_t*)nullptr);
  hipLaunchKernelGGL(scan_emit_starts_kernel, dim3((unsigned)ntiles), dim3(kScanThreads), 0, st, cnt, (size_t)n_capacity,
                     num_voxels, tiles, voxel_start);
  hipLaunchKernelGGL(bucket_fill_kernel, dim3(pn::cdiv(n_capacity, 256)), dim3(256), 0, st, unq_inv, n_capacity, n_dev,
                     voxel_start, cursor, order);
  return pn::check_launch("bucket_points");
}

// Optional canonical order inside every voxel run: ascending point index.  The forward consumers are order independent
// (fixed-point means, maxima); the PFN BACKWARD sums per-point products in run order, so bit-reproducible training sorts
// the runs once per iteration.  One wave per run, rank sort (the point indices of a run are distinct).
__global__ __launch_bounds__(256) void sort_runs_kernel(const int32_t* __restrict__ voxel_start, const int32_t* __restrict__ v_dev,
                                                        int v_cap, const int32_t* __restrict__ in, int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int V = min(*v_dev, v_cap);
  for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < V; v += gridDim.x * 4) {
    const int s = voxel_start[v], n = voxel_start[v + 1] - s;
    for (int i = lane; i < n; i += 64) {
      const int32_t e = in[s + i];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += in[s + j] < e;
      out[s + rank] = e;
    }
  }
}

int pn_sort_voxel_runs(const int32_t* voxel_start, const int32_t* num_voxels, int voxel_capacity, const int32_t* order_in,
                       int32_t* order_out, pn_stream_t stream) {
  PN_REQUIRE(voxel_start && num_voxels && order_in && order_out && order_in != order_out, "sort_voxel_runs: bad pointers");
  if (voxel_capacity <= 0) return PN_OK;
  hipLaunchKernelGGL(sort_runs_kernel, dim3(std::min(2048, pn::cdiv(voxel_capacity, 4))), dim3(256), 0, pn::S(stream), voxel_start,
                     num_voxels, voxel_capacity, order_in, order_out);
  return pn::check_launch("sort_runs_kernel");
}


// Several samples' voxel lists -- each at its capacity, with its count on the device -- into ONE list with the batch index in front of the
// coordinates: what the collate of the reference does on the host (torch.cat of the per-sample voxels and F.pad of the coordinates,
// det3d/torchie/parallel/collate.py:126-128, 157-164) without the counts ever leaving the device.  Sample b's rows follow sample b-1's.
int pn_concat_voxel_segments_f32(const float* feats, const int32_t* coors, const int32_t* counts, int batch, int seg_rows, int c, float* feats_out,
                                 int32_t* coords4_out, int32_t* total, pn_stream_t stream) {
  PN_REQUIRE(feats && coors && counts && feats_out && coords4_out && total, "concat_voxel_segments: null pointer");
  PN_REQUIRE(batch >= 1 && batch <= 64 && seg_rows >= 1 && c >= 1 && (long long)batch * seg_rows < (1ll << 31), "concat_voxel_segments: bad sizes (1 .. 64 samples)");
  const long long rows = (long long)batch * seg_rows;
  hipLaunchKernelGGL(concat_segments_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, pn::S(stream), feats, coors, counts, batch, seg_rows, c,
                     feats_out, coords4_out, total);
  return pn::check_launch("concat_segments_kernel");
}

size_t pn_hard_voxelize_workspace_bytes(uint64_t num_cells, int n, int max_points) {
  // unique workspace + inv + cnt + first + flag + vid_at + tiles + sel[max_points][n] + nv
  const size_t ntiles = ((size_t)n + kScanTile - 1) / kScanTile;
  return UniqueWs(num_cells, n).total + 5 * align256((size_t)n * 4) + align256(ntiles * 4) + align256((size_t)n * 4) +
         align256((size_t)max_points * n * 4) + 256;
}

int pn_hard_voxelize_f32(const float* points, int n, int point_stride, int f, const float* voxel_size, const float* range,
                         int max_points, int max_voxels, float* voxels, int32_t* coors, int32_t* num_points,
                         int32_t* num_voxels, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(points && voxel_size && range && voxels && coors && num_points && num_voxels && workspace,
             "hard_voxelize: null pointer");
  PN_REQUIRE(n >= 0 && f >= 3 && point_stride >= f && max_points >= 1 && max_voxels >= 1, "hard_voxelize: bad sizes");
  GridParams gp;
  uint64_t cells = 1;
  for (int a = 0; a < 3; ++a) {
    gp.lo[a] = range[a];
    gp.vs[a] = voxel_size[a];
    gp.g[a] = (int)nearbyintf((range[3 + a] - range[a]) / voxel_size[a]);  // np.round(...) of the fp32 quotient
    PN_REQUIRE(gp.g[a] >= 1, "hard_voxelize: empty grid");
    cells *= (uint64_t)gp.g[a];
  }
  PN_REQUIRE(cells < (1ull << 32) - 1, "hard_voxelize: more than 2^32 cells");
  if (workspace_bytes < pn_hard_voxelize_workspace_bytes(cells, n, max_points))
    return pn::fail(PN_ERR_WORKSPACE, "hard_voxelize: workspace too small");
  hipStream_t st = pn::S(stream);
  if (int rc = pn::zero_async(voxels, (size_t)max_voxels * max_points * f * 4, st)) return rc;
  if (int rc = pn::zero_async(num_voxels, 4, st)) return rc;
  if (n == 0) return PN_OK;
  char* w = static_cast<char*>(workspace);
  UniqueWs uws(cells, n);
  void* uniq_ws = w; w += uws.total;
  auto take = [&](size_t bytes) { char* p = w; w += align256(bytes); return p; };
  uint32_t* keys = (uint32_t*)take((size_t)n * 4);
  int32_t* inv = (int32_t*)take((size_t)n * 4);
  int32_t* cnt = (int32_t*)take((size_t)n * 4);
  int32_t* first = (int32_t*)take((size_t)n * 4);
  uint32_t* flag = (uint32_t*)take((size_t)n * 4);
  int32_t* vid_at = (int32_t*)take((size_t)n * 4);
  const size_t ntiles = ((size_t)n + kScanTile - 1) / kScanTile;
  uint32_t* tiles = (uint32_t*)take(ntiles * 4);
  int32_t* sel = (int32_t*)take((size_t)max_points * n * 4);
  int32_t* nv = (int32_t*)take(4);
  const int pb = pn::cdiv(n, 256);
  hipLaunchKernelGGL(hard_keys_kernel, dim3(pb), dim3(256), 0, st, points, n, point_stride, gp, keys);
  int32_t grid3[3] = {gp.g[0], gp.g[1], gp.g[2]};
  if (int rc = pn_unique_rank_bitmap(keys, n, nullptr, cells, grid3, nullptr, inv, cnt, nv, uniq_ws, uws.total, stream)) return rc;
  const uint32_t* ukeys = pn_unique_keys_ptr(uniq_ws, cells, n);
  const unsigned fb = (unsigned)std::min<size_t>(2048, ((size_t)(max_points + 1) * n + 255) / 256);
  hipLaunchKernelGGL(hard_fill_int_kernel, dim3(fb), dim3(256), 0, st, first, (size_t)n, 0x7fffffff);
  hipLaunchKernelGGL(hard_fill_int_kernel, dim3(fb), dim3(256), 0, st, sel, (size_t)max_points * n, 0x7fffffff);
  hipLaunchKernelGGL(hard_first_kernel, dim3(pb), dim3(256), 0, st, inv, n, first);
  hipLaunchKernelGGL(hard_flag_kernel, dim3(pb), dim3(256), 0, st, inv, n, first, flag);
  hipLaunchKernelGGL(scan_tile_totals_kernel<1>, dim3((unsigned)ntiles), dim3(kScanThreads), 0, st, flag, (size_t)n,
                     (const int32_t*)nullptr, tiles);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(kScanThreads), 0, st, tiles, (int)ntiles, (int32_t*)nullptr,
                     (int32_t*)nullptr);
  hipLaunchKernelGGL(scan_emit_flags_kernel, dim3((unsigned)ntiles), dim3(kScanThreads), 0, st, flag, (size_t)n, tiles, vid_at);
  for (int k = 0; k < max_points; ++k)
    hipLaunchKernelGGL(hard_select_kernel, dim3(pb), dim3(256), 0, st, inv, n, k ? sel + (size_t)(k - 1) * n : (const int32_t*)nullptr,
                       sel + (size_t)k * n);
  const size_t total = (size_t)n * max_points * f;
  hipLaunchKernelGGL(hard_gather_kernel, dim3((unsigned)std::min<size_t>(8192, (total + 255) / 256)), dim3(256), 0, st, points,
                     point_stride, f, nv, n, first, vid_at, sel, max_points, max_voxels, ukeys, cnt, gp.g[0], gp.g[1], voxels,
                     coors, num_points, num_voxels);
  return pn::check_launch("hard_voxelize");
}

}  // extern "C"
