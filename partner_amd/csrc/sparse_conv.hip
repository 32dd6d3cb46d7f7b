// Index side of the sparse 3-D convolutions (SpMiddleResNetFHD, det3d/models/backbones/scn.py:97-192;
// the arithmetic lives in the third-party spconv package there -- parity unpinned, see oracle/polar_oracle.py).
//
// Active-site bookkeeping without hash tables: a level's active set is a BITMAP over its (B, D, H, W) cells plus the
// rank of the first set bit of every word (popcount scan, the same scheme as the voxel unique-rank):
//     index of site (b,z,y,x) = word_rank[key >> 5] + popcount(bitmap[key >> 5] & below(key & 31))
// so features are stored in key order, a neighbour lookup is two loads and a popcount, and the output sites of a
// strided convolution are found by marking bits and scanning -- deterministic, no atomics on values, no host sync
// (counts stay on the device; kernels size their grids by capacity and read the count).
//   mark (from coordinates / from the input sites of a strided convolution)  ->  3-phase scan  ->  keys in rank order
//   neighbour table  nbr[out site][tap] = input index or -1    (one table per indice_key, reused by every conv of it)
// The convolution itself is pn_sparse_conv_f32 (conv_mfma.hip, gather mode).
#include "pn_common.h"
#include <algorithm>

namespace {

constexpr int kT = 256, kItems = 8, kTile = kT * kItems;  // words per scan tile

struct Dims { int B, D, H, W; };
struct Geo { int k[3], s[3], p[3]; };

__device__ __forceinline__ uint32_t make_key(const Dims& d, int b, int z, int y, int x) { return (uint32_t)(((b * d.D + z) * d.H + y) * d.W + x); }

__device__ __forceinline__ void split_key(const Dims& d, uint32_t key, int& b, int& z, int& y, int& x) {
  x = key % d.W; key /= d.W;
  y = key % d.H; key /= d.H;
  z = key % d.D; b = key / d.D;
}

__device__ __forceinline__ int lookup(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_rank, uint32_t key) {
  const uint32_t w = key >> 5, bit = key & 31, bits = bitmap[w];
  if (!((bits >> bit) & 1u)) return -1;
  return (int)(word_rank[w] + __popc(bits & ((1u << bit) - 1u)));
}


__global__ void mark_coords_kernel(const int32_t* __restrict__ coords, int n_cap, const int32_t* __restrict__ n_dev, Dims d, uint32_t* __restrict__ bitmap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= min(*n_dev, n_cap)) return;
  const int32_t* c = coords + (size_t)i * 4;
  if ((unsigned)c[0] >= (unsigned)d.B || (unsigned)c[1] >= (unsigned)d.D || (unsigned)c[2] >= (unsigned)d.H || (unsigned)c[3] >= (unsigned)d.W) return;
  const uint32_t key = make_key(d, c[0], c[1], c[2], c[3]);
  atomicOr(&bitmap[key >> 5], 1u << (key & 31));
}

// output sites of a strided convolution: o = (p + pad - k) / stride for every tap k with an exact quotient inside the output grid
__global__ void mark_down_kernel(const uint32_t* __restrict__ in_keys, int n_cap, const int32_t* __restrict__ n_dev, Dims di, Dims dout, Geo g,
                                 uint32_t* __restrict__ bitmap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= min(*n_dev, n_cap)) return;
  int b, z, y, x;
  split_key(di, in_keys[i], b, z, y, x);
  for (int kz = 0; kz < g.k[0]; ++kz) {
    const int tz = z + g.p[0] - kz;
    if (tz < 0 || tz % g.s[0] || tz / g.s[0] >= dout.D) continue;
    for (int ky = 0; ky < g.k[1]; ++ky) {
      const int ty = y + g.p[1] - ky;
      if (ty < 0 || ty % g.s[1] || ty / g.s[1] >= dout.H) continue;
      for (int kx = 0; kx < g.k[2]; ++kx) {
        const int tx = x + g.p[2] - kx;
        if (tx < 0 || tx % g.s[2] || tx / g.s[2] >= dout.W) continue;
        const uint32_t key = make_key(dout, b, tz / g.s[0], ty / g.s[1], tx / g.s[2]);
        atomicOr(&bitmap[key >> 5], 1u << (key & 31));
      }
    }
  }
}

__global__ void tile_totals_kernel(const uint32_t* __restrict__ bitmap, size_t nwords, uint32_t* __restrict__ tile_total) {
  const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * kItems;
  uint32_t s = 0;
  if (base + kItems <= nwords) {      // the thread's eight words as two 16-byte loads (the index buffers are 256-byte aligned)
    const uint4 a = *reinterpret_cast<const uint4*>(bitmap + base), b = *reinterpret_cast<const uint4*>(bitmap + base + 4);
    s = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) s += base + k < nwords ? __popc(bitmap[base + k]) : 0;
  }
  uint32_t tot;
  pn::block_exclusive_scan<kT>(s, &tot);
  if (threadIdx.x == 0) tile_total[blockIdx.x] = tot;
}

__global__ void tile_offsets_kernel(uint32_t* __restrict__ tile_total, int ntiles, int32_t* __restrict__ grand_total, int cap) {
  // a thread takes a run of consecutive tiles, ONE block scan over the runs (a scan per 256 tiles was twelve dependent rounds on the finest
  // level's 2880 tiles: 9.6 us in front of the encoder's first convolution)
  const int per = (ntiles + kT - 1) / kT, i0 = threadIdx.x * per, i1 = min(ntiles, i0 + per);
  uint32_t run = 0;
  for (int i = i0; i < i1; ++i) run += tile_total[i];
  uint32_t tot;
  uint32_t ex = pn::block_exclusive_scan<kT>(run, &tot);
  for (int i = i0; i < i1; ++i) {
    const uint32_t v = tile_total[i];
    tile_total[i] = ex;
    ex += v;
  }
  if (threadIdx.x == 0) *grand_total = (int32_t)min(tot, (uint32_t)cap);
}

__global__ void emit_kernel(const uint32_t* __restrict__ bitmap, size_t nwords, const uint32_t* __restrict__ tile_offset, uint32_t* __restrict__ word_rank,
                            uint32_t* __restrict__ keys, int cap) {
  const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * kItems;
  uint32_t words[kItems], s = 0;
  const bool whole = base + kItems <= nwords;
  if (whole) {
    const uint4 a = *reinterpret_cast<const uint4*>(bitmap + base), b = *reinterpret_cast<const uint4*>(bitmap + base + 4);
    words[0] = a.x; words[1] = a.y; words[2] = a.z; words[3] = a.w; words[4] = b.x; words[5] = b.y; words[6] = b.z; words[7] = b.w;
#pragma unroll
    for (int k = 0; k < kItems; ++k) s += __popc(words[k]);
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      words[k] = base + k < nwords ? bitmap[base + k] : 0u;
      s += __popc(words[k]);
    }
  }
  uint32_t tot;
  uint32_t rank = tile_offset[blockIdx.x] + pn::block_exclusive_scan<kT>(s, &tot);
  if (whole) {      // the eight ranks as two 16-byte stores
    uint32_t r8[kItems], r = rank;
#pragma unroll
    for (int k = 0; k < kItems; ++k) { r8[k] = r; r += __popc(words[k]); }
    *reinterpret_cast<uint4*>(word_rank + base) = make_uint4(r8[0], r8[1], r8[2], r8[3]);
    *reinterpret_cast<uint4*>(word_rank + base + 4) = make_uint4(r8[4], r8[5], r8[6], r8[7]);
  }
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    if (base + k >= nwords) break;
    if (!whole) word_rank[base + k] = rank;
    uint32_t wb = words[k];
    while (wb) {
      const int bit = __ffs(wb) - 1;
      wb &= wb - 1;
      if ((int)rank < cap) keys[rank] = (uint32_t)((base + k) * 32 + bit);
      ++rank;
    }
  }
}

__global__ void rank_of_coords_kernel(const int32_t* __restrict__ coords, int n_cap, const int32_t* __restrict__ n_dev, Dims d,
                                      const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_rank, int32_t* __restrict__ rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cap) return;
  int r = -1;
  if (i < *n_dev) {
    const int32_t* c = coords + (size_t)i * 4;
    if ((unsigned)c[0] < (unsigned)d.B && (unsigned)c[1] < (unsigned)d.D && (unsigned)c[2] < (unsigned)d.H && (unsigned)c[3] < (unsigned)d.W)
      r = lookup(bitmap, word_rank, make_key(d, c[0], c[1], c[2], c[3]));
  }
  rank[i] = r;
}

// thread = (output site, kz, ky): the k[2] taps along x are neighbouring cells of one row of the input grid, i.e. (nearly always) bits of ONE
// bitmap word -- the word and its rank are loaded once per row instead of once per tap (r4: 5.6 M (site, tap) threads with two dependent
// scattered loads each took 90 us on the finest level, in front of the encoder's first convolution)
__global__ void neighbor_kernel(const uint32_t* __restrict__ out_keys, int out_cap, const int32_t* __restrict__ n_out, Dims dout,
                                const uint32_t* __restrict__ in_bitmap, const uint32_t* __restrict__ in_rank, Dims di, Geo g, int taps,
                                int32_t* __restrict__ nbr, uint8_t* __restrict__ row_bits) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int n = min(*n_out, out_cap);
  const int rows = g.k[0] * g.k[1];
  if (i >= (size_t)n * rows) return;
  const int site = (int)(i / rows), rt = (int)(i - (size_t)site * rows);
  const int ky = rt % g.k[1], kz = rt / g.k[1];
  int b, z, y, x;
  split_key(dout, out_keys[site], b, z, y, x);
  const int iz = z * g.s[0] - g.p[0] + kz, iy = y * g.s[1] - g.p[1] + ky, ix0 = x * g.s[2] - g.p[2];
  int32_t* o = nbr + (size_t)site * taps + rt * g.k[2];
  const bool row_ok = (unsigned)iz < (unsigned)di.D && (unsigned)iy < (unsigned)di.H;
  uint32_t w_cur = 0xffffffffu, bits = 0, base = 0, present = 0;
  for (int kx = 0; kx < g.k[2]; ++kx) {
    const int ix = ix0 + kx;
    int r = -1;
    if (row_ok && (unsigned)ix < (unsigned)di.W) {
      const uint32_t key = make_key(di, b, iz, iy, ix), w = key >> 5, bit = key & 31;
      if (w != w_cur) {
        w_cur = w;
        bits = in_bitmap[w];
        base = bits ? in_rank[w] : 0u;
      }
      if ((bits >> bit) & 1u) r = (int)(base + __popc(bits & ((1u << bit) - 1u)));
    }
    o[kx] = r;
    present |= (r >= 0 ? 1u : 0u) << kx;
  }
  // (nullable) which taps of this row exist: the grouping sort assembles a site's mask from its k0 k1 bytes instead of re-reading the table
  if (row_bits) row_bits[i] = (uint8_t)present;
}

__global__ void permute_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ rank, int n_cap, const int32_t* __restrict__ n_dev, int c,
                                    float* __restrict__ out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int n = min(*n_dev, n_cap);
  if (i >= (size_t)n * c) return;
  const int row = (int)(i / c), k = (int)(i - (size_t)row * c);
  const int r = rank[row];
  if (r >= 0) out[(size_t)r * c + k] = in[i];
}

// (site, c) -> out[b][y][x][c * D + z]   (== SparseConvTensor.dense() (N,C,D,H,W) viewed as (N, C*D, H, W), scn.py:176-179)
__global__ void to_dense_kernel(const float* __restrict__ feats, const uint32_t* __restrict__ keys, int cap, const int32_t* __restrict__ n_dev, Dims d, int c,
                                float* __restrict__ out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int n = min(*n_dev, cap);
  if (i >= (size_t)n * c) return;
  const int site = (int)(i / c), k = (int)(i - (size_t)site * c);
  int b, z, y, x;
  split_key(d, keys[site], b, z, y, x);
  out[(((size_t)b * d.H + y) * d.W + x) * ((size_t)c * d.D) + (size_t)k * d.D + z] = feats[i];
}

// the same dense map written from the OUTPUT side (r4): thread = (pixel, channel quad), the level's bitmap-rank index says which of the D
// cells above the pixel are active; every element of the map is written exactly once, 16-byte stores in the map's own order -- no zero
// fill in front and no scattered 4-byte writes (zero fill 15 us + scatter 30 us -> one pass, behind the encoder's last convolution)
__global__ void to_dense_lookup_kernel(const float* __restrict__ feats, const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_rank, Dims d,
                                       int c, float* __restrict__ out, size_t total) {
  const int c4n = c >> 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4n);
    const size_t pix = i / c4n;
    const int x = (int)(pix % d.W), y = (int)((pix / d.W) % d.H), b = (int)(pix / ((size_t)d.W * d.H));
    float* o = out + pix * ((size_t)c * d.D) + (size_t)(4 * q) * d.D;
    if (d.D == 2) {
      const int r0 = lookup(bitmap, word_rank, make_key(d, b, 0, y, x)), r1 = lookup(bitmap, word_rank, make_key(d, b, 1, y, x));
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 v0 = r0 >= 0 ? *reinterpret_cast<const float4*>(feats + (size_t)r0 * c + 4 * q) : z4;
      const float4 v1 = r1 >= 0 ? *reinterpret_cast<const float4*>(feats + (size_t)r1 * c + 4 * q) : z4;
      *reinterpret_cast<float4*>(o) = make_float4(v0.x, v1.x, v0.y, v1.y);
      *reinterpret_cast<float4*>(o + 4) = make_float4(v0.z, v1.z, v0.w, v1.w);
    } else {
      for (int z = 0; z < d.D; ++z) {
        const int r = lookup(bitmap, word_rank, make_key(d, b, z, y, x));
        for (int j = 0; j < 4; ++j) o[(size_t)j * d.D + z] = r >= 0 ? feats[(size_t)r * c + 4 * q + j] : 0.f;
      }
    }
  }
}

struct IndexBuf {
  uint32_t* bitmap; uint32_t* word_rank; uint32_t* tiles; size_t nwords; int ntiles; size_t bytes;
  IndexBuf(void* base, uint64_t cells) {
    nwords = (size_t)((cells + 31) / 32);
    ntiles = (int)((nwords + kTile - 1) / kTile);
    char* p = static_cast<char*>(base);
    bitmap = reinterpret_cast<uint32_t*>(p); p += (nwords * 4 + 255) / 256 * 256;
    word_rank = reinterpret_cast<uint32_t*>(p); p += (nwords * 4 + 255) / 256 * 256;
    tiles = reinterpret_cast<uint32_t*>(p); p += ((size_t)ntiles * 4 + 255) / 256 * 256;
    bytes = (size_t)(p - static_cast<char*>(base));
  }
};

int scan_and_emit(const IndexBuf& ib, uint32_t* keys, int cap, int32_t* count, hipStream_t st) {
  hipLaunchKernelGGL(tile_totals_kernel, dim3(ib.ntiles), dim3(kT), 0, st, ib.bitmap, ib.nwords, ib.tiles);
  hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(kT), 0, st, ib.tiles, ib.ntiles, count, cap);
  hipLaunchKernelGGL(emit_kernel, dim3(ib.ntiles), dim3(kT), 0, st, ib.bitmap, ib.nwords, ib.tiles, ib.word_rank, keys, cap);
  return pn::check_launch("sparse index scan");
}

inline Dims mk(const int32_t* d) { return Dims{d[0], d[1], d[2], d[3]}; }
inline Geo mkg(const int32_t* k, const int32_t* s, const int32_t* p) { return Geo{{k[0], k[1], k[2]}, {s[0], s[1], s[2]}, {p[0], p[1], p[2]}}; }
inline uint64_t cells_of(const int32_t* d) { return (uint64_t)d[0] * d[1] * d[2] * d[3]; }

}  // namespace

extern "C" {

size_t pn_sparse_index_bytes(uint64_t num_cells) { return IndexBuf(nullptr, num_cells).bytes; }

int pn_sparse_index_from_coords(const int32_t* coords, int n_capacity, const int32_t* n_dev, const int32_t* dims, void* index_buf, uint32_t* keys,
                                int32_t* count, int32_t* rank_of_input, pn_stream_t stream) {
  PN_REQUIRE(coords && n_dev && dims && index_buf && keys && count && rank_of_input && n_capacity >= 1, "sparse_index_from_coords: bad arguments");
  PN_REQUIRE(cells_of(dims) < (1ull << 32), "sparse_index: grid has more than 2^32 cells");
  IndexBuf ib(index_buf, cells_of(dims));
  hipStream_t st = pn::S(stream);
  if (int rc = pn::zero_async(ib.bitmap, ib.nwords * 4, st)) return rc;
  hipLaunchKernelGGL(mark_coords_kernel, dim3(pn::cdiv(n_capacity, 256)), dim3(256), 0, st, coords, n_capacity, n_dev, mk(dims), ib.bitmap);
  if (int rc = scan_and_emit(ib, keys, n_capacity, count, st)) return rc;
  hipLaunchKernelGGL(rank_of_coords_kernel, dim3(pn::cdiv(n_capacity, 256)), dim3(256), 0, st, coords, n_capacity, n_dev, mk(dims), ib.bitmap, ib.word_rank,
                     rank_of_input);
  return pn::check_launch("rank_of_coords_kernel");
}

int pn_sparse_index_downsample(const uint32_t* in_keys, int in_capacity, const int32_t* n_in, const int32_t* in_dims, const int32_t* kernel,
                               const int32_t* stride, const int32_t* pad, const int32_t* out_dims, void* out_index_buf, uint32_t* out_keys,
                               int out_capacity, int32_t* out_count, pn_stream_t stream) {
  PN_REQUIRE(in_keys && n_in && in_dims && kernel && stride && pad && out_dims && out_index_buf && out_keys && out_count, "sparse_index_downsample: null pointer");
  PN_REQUIRE(cells_of(out_dims) < (1ull << 32) && out_capacity >= 1, "sparse_index_downsample: bad sizes");
  IndexBuf ib(out_index_buf, cells_of(out_dims));
  hipStream_t st = pn::S(stream);
  if (int rc = pn::zero_async(ib.bitmap, ib.nwords * 4, st)) return rc;
  hipLaunchKernelGGL(mark_down_kernel, dim3(pn::cdiv(in_capacity, 256)), dim3(256), 0, st, in_keys, in_capacity, n_in, mk(in_dims), mk(out_dims),
                     mkg(kernel, stride, pad), ib.bitmap);
  return scan_and_emit(ib, out_keys, out_capacity, out_count, st);
}

static int neighbors_run(const uint32_t* out_keys, int out_capacity, const int32_t* n_out, const int32_t* out_dims, const void* in_index_buf,
                         const int32_t* in_dims, const int32_t* kernel, const int32_t* stride, const int32_t* pad, int32_t* nbr, uint8_t* row_bits,
                         pn_stream_t stream) {
  PN_REQUIRE(out_keys && n_out && out_dims && in_index_buf && in_dims && kernel && stride && pad && nbr, "sparse_neighbors: null pointer");
  const int taps = kernel[0] * kernel[1] * kernel[2];
  PN_REQUIRE(taps >= 1 && taps <= 32 && out_capacity >= 1, "sparse_neighbors: at most 32 taps");
  PN_REQUIRE(!row_bits || kernel[2] <= 8, "sparse_neighbors: row bits want at most 8 taps along x");
  IndexBuf ib(const_cast<void*>(in_index_buf), cells_of(in_dims));
  const size_t total = (size_t)out_capacity * kernel[0] * kernel[1];
  hipLaunchKernelGGL(neighbor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pn::S(stream), out_keys, out_capacity, n_out, mk(out_dims),
                     ib.bitmap, ib.word_rank, mk(in_dims), mkg(kernel, stride, pad), taps, nbr, row_bits);
  return pn::check_launch("neighbor_kernel");
}

int pn_sparse_neighbors(const uint32_t* out_keys, int out_capacity, const int32_t* n_out, const int32_t* out_dims, const void* in_index_buf,
                        const int32_t* in_dims, const int32_t* kernel, const int32_t* stride, const int32_t* pad, int32_t* nbr, pn_stream_t stream) {
  return neighbors_run(out_keys, out_capacity, n_out, out_dims, in_index_buf, in_dims, kernel, stride, pad, nbr, nullptr, stream);
}

// the same table plus, per (site, kz, ky), one byte whose bit kx says whether tap (kz, ky, kx) exists: [out_capacity][k0 k1] bytes, the input
// of pn_sparse_group_rows_bits (r4)
int pn_sparse_neighbors_rows(const uint32_t* out_keys, int out_capacity, const int32_t* n_out, const int32_t* out_dims, const void* in_index_buf,
                             const int32_t* in_dims, const int32_t* kernel, const int32_t* stride, const int32_t* pad, int32_t* nbr, uint8_t* row_bits,
                             pn_stream_t stream) {
  PN_REQUIRE(row_bits, "sparse_neighbors_rows: null pointer");
  return neighbors_run(out_keys, out_capacity, n_out, out_dims, in_index_buf, in_dims, kernel, stride, pad, nbr, row_bits, stream);
}

int pn_sparse_permute_rows(const float* in, const int32_t* rank, int n_capacity, const int32_t* n_dev, int c, float* out, pn_stream_t stream) {
  PN_REQUIRE(in && rank && n_dev && out && c >= 1 && n_capacity >= 1, "sparse_permute_rows: bad arguments");
  const size_t total = (size_t)n_capacity * c;
  hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pn::S(stream), in, rank, n_capacity, n_dev, c, out);
  return pn::check_launch("permute_rows_kernel");
}

int pn_sparse_to_dense_nhwc(const float* feats, const uint32_t* keys, int capacity, const int32_t* n_dev, const int32_t* dims, int c, float* out,
                            pn_stream_t stream) {
  PN_REQUIRE(feats && keys && n_dev && dims && out && c >= 1 && capacity >= 1, "sparse_to_dense: bad arguments");
  hipStream_t st = pn::S(stream);
  if (int rc = pn::zero_async(out, (size_t)cells_of(dims) * c * sizeof(float), st)) return rc;
  const size_t total = (size_t)capacity * c;
  hipLaunchKernelGGL(to_dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, feats, keys, capacity, n_dev, mk(dims), c, out);
  return pn::check_launch("to_dense_kernel");
}

// pn_sparse_to_dense_nhwc from the level's index instead of its key list (same result; c a multiple of 4)
int pn_sparse_to_dense_index_nhwc(const float* feats, const void* index_buf, const int32_t* dims, int c, float* out, pn_stream_t stream) {
  PN_REQUIRE(feats && index_buf && dims && out && c >= 4 && c % 4 == 0, "sparse_to_dense_index: bad arguments");
  PN_REQUIRE(((uintptr_t)feats & 15) == 0 && ((uintptr_t)out & 15) == 0, "sparse_to_dense_index: pointers must be 16-byte aligned");
  IndexBuf ib(const_cast<void*>(index_buf), cells_of(dims));
  const Dims d = mk(dims);
  const size_t total = (size_t)d.B * d.H * d.W * (c / 4);
  hipLaunchKernelGGL(to_dense_lookup_kernel, dim3((unsigned)std::min<size_t>(16384, (total + 255) / 256)), dim3(256), 0, pn::S(stream), feats, ib.bitmap, ib.word_rank, d,
                     c, out, total);
  return pn::check_launch("to_dense_lookup_kernel");
}

}  // extern "C"
