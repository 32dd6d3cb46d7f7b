// Per-voxel reductions and the dynamic pillar feature net.
//
// Points are bucketed by voxel rank (voxel_start / order from pn_bucket_points), so one
// wavefront owns a voxel's whole run of points: no float atomics anywhere.
//   * means: every addend is converted to 2^-24 fixed point (exact for |x| >= 1, half-ulp of
//     2^-24 below) and summed as int64 -> the sum does not depend on the order of the run,
//     results are bitwise reproducible and closer to the exact mean than an fp32 running sum.
//   * max:   order independent by nature.
// PFN mapping: lane = output channel.  The 16-channel decoration of the current point is
// wave-uniform; weights sit transposed in LDS ([k][n], consecutive lanes -> consecutive
// banks), the layer-0 activations of the current point go through a per-wave LDS row and are
// read back as broadcasts.
#include "pn_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr double kFix = 16777216.0;  // 2^24

__device__ __forceinline__ long long to_fix(float v) { return (long long)rintf(v * 16777216.0f); }

// ------------------------------------------------------------------------------- V3
__global__ void scatter_mean_kernel(const float* __restrict__ pts, int stride, int f, const int32_t* __restrict__ vstart,
                                    const int32_t* __restrict__ order, const int32_t* __restrict__ v_dev, int v_cap,
                                    float* __restrict__ mean) {
  const int V = min(*v_dev, v_cap);
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int v = wave; v < V; v += nwaves) {
    const int s = vstart[v], e = vstart[v + 1];
    for (int k0 = 0; k0 < f; k0 += 8) {
      long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = s + lane; i < e; i += 64) {
        const float* p = pts + (size_t)order[i] * stride + k0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < f) acc[k] += to_fix(p[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const long long t = pn::wave_sum(acc[k]);
        if (lane == k && k0 + k < f) mean[(size_t)v * f + k0 + k] = (float)((double)t / kFix / (double)(e - s));
      }
    }
  }
}

__global__ void hard_voxel_mean_kernel(const float* __restrict__ vox, const int32_t* __restrict__ num, int v, int p, int f,
                                       float* __restrict__ mean) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v * f) return;
  const int vi = i / f, k = i - vi * f;
  float s = 0.f;
  for (int j = 0; j < p; ++j) s += vox[((size_t)vi * p + j) * f + k];  // sequential fp32 sum == torch.sum over dim 1
  mean[i] = s / (float)num[vi];
}

// ------------------------------------------------------------------------------- V4 (+V5)
using f32x2 = __attribute__((ext_vector_type(2))) float;

struct PfnArgs {
  const float* pts;
  int stride;
  const int32_t* vstart;
  const int32_t* order;
  const int32_t* v_dev;
  int v_cap;
  const uint32_t* ukeys;
  int R, T, Z;
  const float* w0;
  int c0;
  const float* w1;
  int c1;
  float vx, vy, xoff, yoff;
  float* feat;
  float* canvas;
  uint32_t* cell_count;      // nullable (r6, the (32, 128) kernels): the frame index's per-cell counters, zeroed for this frame's voxels on the way
};

constexpr int kPfnWaves = 8;  // 8 waves share one LDS copy of the weights: 4 blocks (32 waves) per CU

__global__ __launch_bounds__(kPfnWaves * 64) void dynamic_pfn_kernel(PfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int c0 = a.c0, c1 = a.c1;
  float* w0t = lds;                 // [16][c0]
  float* w1t = w0t + 16 * c0;       // [2*c0][c1]
  float* rows = w1t + 2 * c0 * c1;  // per wave: [64] layer-0 activations / maxima
  for (int i = threadIdx.x; i < 16 * c0; i += blockDim.x) {
    const int k = i / c0, n = i - k * c0;
    w0t[i] = a.w0[n * 16 + k];
  }
  for (int i = threadIdx.x; i < 2 * c0 * c1; i += blockDim.x) {
    const int k = i / c1, n = i - k * c1;
    w1t[i] = a.w1[n * 2 * c0 + k];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  float* row = rows + wib * 64;
  const int V = min(*a.v_dev, a.v_cap);
  const int wave = blockIdx.x * kPfnWaves + wib;
  const int nwaves = gridDim.x * kPfnWaves;
  const bool two = c1 > 64;  // second output channel per lane

  // voxel metadata of the next iteration is fetched while the current voxel is processed
  int n_s = 0, n_e = 0;
  uint32_t n_key = 0;
  if (wave < V) { n_s = a.vstart[wave]; n_e = a.vstart[wave + 1]; n_key = a.ukeys[wave]; }
  for (int v = wave; v < V; v += nwaves) {
    const int s = n_s, e = n_e;
    uint32_t key = n_key;
    if (v + nwaves < V) { n_s = a.vstart[v + nwaves]; n_e = a.vstart[v + nwaves + 1]; n_key = a.ukeys[v + nwaves]; }
    const int ri = key % a.R; key /= a.R;
    const int ti = key % a.T; key /= a.T;
    const int bi = key / a.Z;
    // ---- voxel means of (x, y, z, rho, phi), fixed point ---------------------------------
    long long sx = 0, sy = 0, sz = 0, sr = 0, sp = 0;
    for (int i = s + lane; i < e; i += 64) {
      const float* p = a.pts + (size_t)a.order[i] * a.stride;
      sr += to_fix(p[0]); sp += to_fix(p[1]); sz += to_fix(p[2]); sx += to_fix(p[3]); sy += to_fix(p[4]);
    }
    const double inv_n = 1.0 / ((double)(e - s) * kFix);
    const float mx = (float)((double)pn::wave_sum(sx) * inv_n), my = (float)((double)pn::wave_sum(sy) * inv_n);
    const float mz = (float)((double)pn::wave_sum(sz) * inv_n), mr = (float)((double)pn::wave_sum(sr) * inv_n);
    const float mp = (float)((double)pn::wave_sum(sp) * inv_n);
    // pillar centre in polar and Cartesian coordinates (pillar_encoder.py:350-351,365)
    const float rc = __fadd_rn(__fmul_rn((float)ri, a.vx), a.xoff);
    const float pc = __fadd_rn(__fmul_rn((float)ti, a.vy), a.yoff);
    const float xc = __fmul_rn(rc, cosf(pc)), yc = __fmul_rn(rc, sinf(pc));

    auto layer0 = [&](const float* p) -> float {
      // 16-channel decoration of one point, then row `lane` of Linear(16 -> c0) + ReLU
      const float rho = p[0], phi = p[1], z = p[2], x = p[3], y = p[4];
      const float d[16] = {rho, phi, z, x, y, p[5], p[6], x - mx, y - my, z - mz, x - xc, y - yc,
                           rho - mr, phi - mp, rho - rc, phi - pc};
      float h = 0.f;
      if (lane < c0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) h = fmaf(w0t[k * c0 + lane], d[k], h);
      }
      return h > 0.f ? h : 0.f;
    };

    // ---- pass 1: per-voxel maximum of the layer-0 activations -----------------------------
    float m0 = 0.f;  // ReLU output is >= 0
    for (int i = s; i < e; ++i) m0 = fmaxf(m0, layer0(a.pts + (size_t)a.order[i] * a.stride));
    // voxel-constant half of layer 1: g[n] = sum_c W1[n][c0 + c] * m0[c]
    row[lane] = m0;
    float g0 = 0.f, g1 = 0.f;
    for (int c = 0; c < c0; ++c) {
      const float m = row[c];
      if (lane < c1) g0 = fmaf(w1t[(c0 + c) * c1 + lane], m, g0);
      if (two && lane + 64 < c1) g1 = fmaf(w1t[(c0 + c) * c1 + lane + 64], m, g1);
    }
    // ---- pass 2: layer 1 per point, running maximum ---------------------------------------
    float f0 = 0.f, f1 = 0.f;
    for (int i = s; i < e; ++i) {
      const float h = layer0(a.pts + (size_t)a.order[i] * a.stride);
      row[lane] = h;  // same wave writes and reads: in-order LDS, no barrier needed
      float y0 = g0, y1 = g1;
      for (int c = 0; c < c0; ++c) {
        const float hc = row[c];
        if (lane < c1) y0 = fmaf(w1t[c * c1 + lane], hc, y0);
        if (two && lane + 64 < c1) y1 = fmaf(w1t[c * c1 + lane + 64], hc, y1);
      }
      f0 = fmaxf(f0, y0);
      f1 = fmaxf(f1, y1);
    }
    if (a.feat) {
      if (lane < c1) a.feat[(size_t)v * c1 + lane] = f0;
      if (two && lane + 64 < c1) a.feat[(size_t)v * c1 + lane + 64] = f1;
    }
    if (a.canvas) {
      float* cv = a.canvas + (((size_t)bi * a.T + ti) * a.R + ri) * c1;
      if (lane < c1) cv[lane] = f0;
      if (two && lane + 64 < c1) cv[lane + 64] = f1;
    }
  }
}


// ---- specialisation for the reference's nuScenes reader (C0 = 32, C1 = 128) ------------------
// Weights live in registers (lane n holds rows n and n+64 of W1, lanes < 32 hold row n of W0), the
// layer-0 activation of the current point is broadcast with v_readlane (no LDS at all), the pillar
// (r, theta, b) comes packed from the unique kernel and cos/sin of the pillar azimuth from a table.
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

constexpr int kHeavyPillar = 64;  // pillars with more points than this take the block-per-pillar kernel
constexpr int kHeavyWaves = 4;   // = the waves of the batched path: both run as blocks of ONE launch

constexpr int kFwdBatch = 64;    // pillars per block and round
constexpr int kFwdPoints = 1024; // point rows staged in LDS per sub-batch (>= kHeavyPillar, so any non-heavy pillar fits)

// A wave walking alone through vstart -> order -> point pays three dependent memory latencies per pillar and per point
// (two waves per SIMD: little to switch to), which was all of this kernel's time.  The block therefore takes a batch of
// consecutive pillars -- their points are one consecutive range of order[] -- and stages run bounds, keys and the point
// rows in LDS with all 256 threads loading in parallel (three latencies per BATCH); each wave then works through its
// share of the pillars from LDS.  A batch whose points exceed the LDS rows is cut into sub-batches at pillar boundaries.
__device__ __forceinline__ void pfn_32_128_main(const PfnArgs& a, const float* __restrict__ cs_table, const int bid, const int nblk) {
  constexpr int NB = kFwdBatch, CAP = kFwdPoints;
  static_assert(CAP >= kHeavyPillar, "a non-heavy pillar must fit the staging rows");
  __shared__ int m_s[NB], m_e[NB];
  __shared__ uint32_t m_key[NB];
  __shared__ __attribute__((aligned(16))) float m_c[NB][12];  // per pillar: mx my mz mr mp rc pc xc yc, canvas cell (int bits)
  __shared__ __attribute__((aligned(16))) float p_l[CAP][8];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int V = min(*a.v_dev, a.v_cap);
  // r5: rows n and n + 64 of W1 as PAIRS: layer 1 runs on v_pk_fma_f32 (two IEEE fmas per lane and instruction -- the bits of two v_fma_f32).
  // Ablation at 300 k points (10 sweeps, 177 k pillars): 140 of the kernel's 192 us are this layer arithmetic, at 2/3 of the vector fp32 rate;
  // staging, per-pillar scalars and the stores together 52.  With the pairs: 171 us.  (Both layers as MFMA tiles over ALL point rows --
  // G = M0 W1b^T, Y = G[pillar] + H W1a^T, LDS-atomic maxima -- were built and measured: 183 us at 300 k, 52 against 33 us at 30 k points;
  // the per-batch chain of barriers and the bank-conflicting pillar-strided LDS accesses cost more than the matrix pipe saved.  Dropped.)
  float w0[16];
  f32x2 w1[64];
#pragma unroll
  for (int k = 0; k < 16; ++k) w0[k] = lane < 32 ? a.w0[lane * 16 + k] : 0.f;
#pragma unroll
  for (int k = 0; k < 64; ++k) w1[k] = f32x2{a.w1[lane * 64 + k], a.w1[(lane + 64) * 64 + k]};
  for (int v0 = bid * NB; v0 < V; v0 += nblk * NB) {
    const int nb = min(NB, V - v0);
    __syncthreads();  // the previous batch has been consumed
    if (tid < nb) {
      m_s[tid] = a.vstart[v0 + tid];
      m_e[tid] = a.vstart[v0 + tid + 1];
      m_key[tid] = a.ukeys[v0 + tid];
      if (a.cell_count) a.cell_count[a.ukeys[v0 + tid]] = 0u;      // (the index launches are done with the counter: the next frame finds it zero)
    }
    __syncthreads();
    int q0 = 0;
    while (q0 < nb) {  // block-uniform
      const int base = m_s[q0];
      if (m_e[q0] - base > kHeavyPillar) { ++q0; continue; }  // the heavy-pillar blocks of the same launch spreads those over a whole block
      int q1 = q0 + 1;
      while (q1 < nb && m_e[q1] - m_s[q1] <= kHeavyPillar && m_e[q1] - base <= CAP) ++q1;
      const int npts = m_e[q1 - 1] - base;
      for (int t = tid; t < npts; t += 256) {
        const float* src = a.pts + (size_t)a.order[base + t] * a.stride;
#pragma unroll
        for (int k = 0; k < 7; ++k) p_l[t][k] = src[k];
      }
      __syncthreads();
      // per-pillar scalars, one THREAD per pillar (the whole batch in one pass of wave instructions instead of one pass
      // per pillar): key decode, cell centre, exact fixed-point means
      if (q0 + tid < q1) {
        const int q = q0 + tid;
        uint32_t key = m_key[q];
        const int ri = key % a.R; key /= a.R;
        const int ti = key % a.T; key /= a.T;
        const int bi = key / a.Z;
        const int s = m_s[q] - base, e = m_e[q] - base;
        long long sx = 0, sy = 0, sz = 0, sr = 0, sp = 0;
        for (int i = s; i < e; ++i) {
          const float* p = p_l[i];
          sr += to_fix(p[0]); sp += to_fix(p[1]); sz += to_fix(p[2]); sx += to_fix(p[3]); sy += to_fix(p[4]);
        }
        const double inv_n = 1.0 / ((double)(e - s) * kFix);
        float* c = m_c[q];
        c[0] = (float)((double)sx * inv_n); c[1] = (float)((double)sy * inv_n); c[2] = (float)((double)sz * inv_n);
        c[3] = (float)((double)sr * inv_n); c[4] = (float)((double)sp * inv_n);
        const float rc = __fadd_rn(__fmul_rn((float)ri, a.vx), a.xoff);
        c[5] = rc;
        c[6] = __fadd_rn(__fmul_rn((float)ti, a.vy), a.yoff);
        c[7] = __fmul_rn(rc, cs_table[2 * ti]);
        c[8] = __fmul_rn(rc, cs_table[2 * ti + 1]);
        c[9] = __builtin_bit_cast(float, (bi * a.T + ti) * a.R + ri);
      }
      __syncthreads();
      for (int q = q0 + wib; q < q1; q += 4) {
        const int v = v0 + q;
        const int s = m_s[q] - base, e = m_e[q] - base;
        const float* c = m_c[q];
        const float mx = c[0], my = c[1], mz = c[2], mr = c[3], mp = c[4], rc = c[5], pc = c[6], xc = c[7], yc = c[8];
        const int cell = __builtin_bit_cast(int, c[9]);
        auto layer0 = [&](const float* p) -> float {
          const float rho = p[0], phi = p[1], z = p[2], x = p[3], y = p[4];
          const float d[16] = {rho, phi, z, x, y, p[5], p[6], x - mx, y - my, z - mz, x - xc, y - yc,
                               rho - mr, phi - mp, rho - rc, phi - pc};
          float h = 0.f;
#pragma unroll
          for (int k = 0; k < 16; ++k) h = fmaf(w0[k], d[k], h);
          return h > 0.f ? h : 0.f;  // lanes >= 32 carry zeros (their w0 is zero)
        };
        float m0 = 0.f;
        float h_first = 0.f;
        for (int i = s; i < e; ++i) {
          const float h = layer0(p_l[i]);
          if (i == s) h_first = h;
          m0 = fmaxf(m0, h);
        }
        f32x2 g = {0.f, 0.f};
        float f0 = 0.f, f1 = 0.f;
        if (e - s == 1) {  // one point (most pillars of a single sweep): maximum == the point, one broadcast serves both halves
          float hb[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) hb[k] = lane_bcast(h_first, k);
#pragma unroll
          for (int k = 0; k < 32; ++k) g = __builtin_elementwise_fma(w1[32 + k], f32x2{hb[k], hb[k]}, g);
#pragma unroll
          for (int k = 0; k < 32; ++k) g = __builtin_elementwise_fma(w1[k], f32x2{hb[k], hb[k]}, g);
          f0 = fmaxf(0.f, g[0]);
          f1 = fmaxf(0.f, g[1]);
        } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const float m = lane_bcast(m0, k);
          g = __builtin_elementwise_fma(w1[32 + k], f32x2{m, m}, g);
        }
        for (int i = s; i < e; ++i) {
          const float h = (i == s) ? h_first : layer0(p_l[i]);
          f32x2 y = g;
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            const float hc = lane_bcast(h, k);
            y = __builtin_elementwise_fma(w1[k], f32x2{hc, hc}, y);
          }
          f0 = fmaxf(f0, y[0]);
          f1 = fmaxf(f1, y[1]);
        }
        }
        if (a.feat) {
          a.feat[(size_t)v * 128 + lane] = f0;
          a.feat[(size_t)v * 128 + lane + 64] = f1;
        }
        if (a.canvas) {
          float* cv = a.canvas + (size_t)cell * 128;
          cv[lane] = f0;
          cv[lane + 64] = f1;
        }
      }
      q0 = q1;
      if (q0 < nb) __syncthreads();  // the staged rows are rewritten by the next sub-batch
    }
  }
}

// Pillars with more than kHeavyPillar points (a wall right in front of the sensor, a degenerate
// sweep): one 8-wave block per pillar, the three phases (means, layer-0 maxima, layer-1 maxima)
// meet in LDS.  Sums are exact int64 and maxima order independent, so the values are bit-identical
// to what the one-wave path would have produced.
template <int TH = kHeavyPillar>      // pillars with more than TH points (the tile path hands over at 16)
__device__ __forceinline__ void pfn_32_128_heavy(const PfnArgs& a, const float* __restrict__ cs_table, const int bid, const int nblk) {
  __shared__ long long part_sum[kHeavyWaves][5];
  __shared__ float part_max[kHeavyWaves][128];
  __shared__ int heavy_list[kHeavyWaves * 64];
  __shared__ int heavy_n;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int V = min(*a.v_dev, a.v_cap);
  float w0[16], w1a[64], w1b[64];
  bool loaded = false;
  // block b owns the pillars v == b (mod gridDim.x) -- neighbouring heavy pillars land on different
  // blocks; every thread tests one of them, the hits are compacted into an LDS list
  for (int base = 0; base < V; base += nblk * (kHeavyWaves * 64)) {
    if (threadIdx.x == 0) heavy_n = 0;
    __syncthreads();
    const int cand = base + bid + nblk * threadIdx.x;
    if (cand < V && a.vstart[cand + 1] - a.vstart[cand] > TH) heavy_list[atomicAdd(&heavy_n, 1)] = cand;
    __syncthreads();
    const int n_heavy = heavy_n;
  for (int hi = 0; hi < n_heavy; ++hi) {
    const int v = heavy_list[hi];
    const int s = a.vstart[v], e = a.vstart[v + 1];
    if (!loaded) {
#pragma unroll
      for (int k = 0; k < 16; ++k) w0[k] = lane < 32 ? a.w0[lane * 16 + k] : 0.f;
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        w1a[k] = a.w1[lane * 64 + k];
        w1b[k] = a.w1[(lane + 64) * 64 + k];
      }
      loaded = true;
    }
    uint32_t key = a.ukeys[v];
    const int ri = key % a.R; key /= a.R;
    const int ti = key % a.T; key /= a.T;
    const int bi = key / a.Z;
    // ---- phase 1: means ---------------------------------------------------------------------
    long long sx = 0, sy = 0, sz = 0, sr = 0, sp = 0;
    for (int i = s + (int)threadIdx.x; i < e; i += kHeavyWaves * 64) {
      const float* p = a.pts + (size_t)a.order[i] * a.stride;
      sr += to_fix(p[0]); sp += to_fix(p[1]); sz += to_fix(p[2]); sx += to_fix(p[3]); sy += to_fix(p[4]);
    }
    sx = pn::wave_sum(sx); sy = pn::wave_sum(sy); sz = pn::wave_sum(sz); sr = pn::wave_sum(sr); sp = pn::wave_sum(sp);
    if (lane == 0) { part_sum[wib][0] = sx; part_sum[wib][1] = sy; part_sum[wib][2] = sz; part_sum[wib][3] = sr; part_sum[wib][4] = sp; }
    __syncthreads();
    long long t[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < kHeavyWaves; ++w)
#pragma unroll
      for (int k = 0; k < 5; ++k) t[k] += part_sum[w][k];
    const double inv_n = 1.0 / ((double)(e - s) * kFix);
    const float mx = (float)((double)t[0] * inv_n), my = (float)((double)t[1] * inv_n), mz = (float)((double)t[2] * inv_n);
    const float mr = (float)((double)t[3] * inv_n), mp = (float)((double)t[4] * inv_n);
    const float rc = __fadd_rn(__fmul_rn((float)ri, a.vx), a.xoff);
    const float pc = __fadd_rn(__fmul_rn((float)ti, a.vy), a.yoff);
    const float xc = __fmul_rn(rc, cs_table[2 * ti]), yc = __fmul_rn(rc, cs_table[2 * ti + 1]);
    auto layer0 = [&](const float* p) -> float {
      const float rho = p[0], phi = p[1], z = p[2], x = p[3], y = p[4];
      const float d[16] = {rho, phi, z, x, y, p[5], p[6], x - mx, y - my, z - mz, x - xc, y - yc,
                           rho - mr, phi - mp, rho - rc, phi - pc};
      float h = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) h = fmaf(w0[k], d[k], h);
      return h > 0.f ? h : 0.f;
    };
    // ---- phase 2: layer-0 maxima, wave w takes points s+w, s+w+8, ... -------------------------
    float m0 = 0.f;
    for (int i = s + wib; i < e; i += kHeavyWaves) m0 = fmaxf(m0, layer0(a.pts + (size_t)a.order[i] * a.stride));
    part_max[wib][lane] = m0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kHeavyWaves; ++w) m0 = fmaxf(m0, part_max[w][lane]);
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float m = lane_bcast(m0, c);
      g0 = fmaf(w1a[32 + c], m, g0);
      g1 = fmaf(w1b[32 + c], m, g1);
    }
    __syncthreads();  // part_max is reused below
    // ---- phase 3: layer 1 ---------------------------------------------------------------------
    float f0 = 0.f, f1 = 0.f;
    for (int i = s + wib; i < e; i += kHeavyWaves) {
      const float h = layer0(a.pts + (size_t)a.order[i] * a.stride);
      float y0 = g0, y1 = g1;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float hc = lane_bcast(h, c);
        y0 = fmaf(w1a[c], hc, y0);
        y1 = fmaf(w1b[c], hc, y1);
      }
      f0 = fmaxf(f0, y0);
      f1 = fmaxf(f1, y1);
    }
    part_max[wib][lane] = f0;
    part_max[wib][lane + 64] = f1;
    __syncthreads();
    if (threadIdx.x < 128) {
      float f = 0.f;
#pragma unroll
      for (int w = 0; w < kHeavyWaves; ++w) f = fmaxf(f, part_max[w][threadIdx.x]);
      if (a.feat) a.feat[(size_t)v * 128 + threadIdx.x] = f;
      if (a.canvas) a.canvas[(((size_t)bi * a.T + ti) * a.R + ri) * 128 + threadIdx.x] = f;
    }
    __syncthreads();
  }
  }
}

// ---- r5 (late) / r6: the same arithmetic on the matrix pipe, WITHOUT giving up a bit.  v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (guide,
// "FP32-input MFMA"; conv_mfma.hip's small_n_mfma_body reproduces a vector kernel's bits with it), so layer 0 (16 features, k ascending) and
// layer 1 (the 32 maxima against W1's columns 32 .. 63, then the 32 activations against columns 0 .. 31 -- the order of the loop above) can run
// as chains inside MFMAs: H^T = W0 D^T, G^T = W1b M0, Y^T = G^T + W1a H.  Layer 0's accumulators ARE layer 1's B operand: W0's rows are fed in
// the order (16 nt + 4 r + g) for accumulator register r of lane group g, so that K step i = 4 nt + r of layer 1 finds channel 4 i + g in lane
// group g -- no LDS, no cross-lane traffic between the layers.
// r6: the sixteen columns of a tile are sixteen PILLARS, and a pillar's points are successive PASSES of its column (r5 had point rows as columns and
// a pillar's rows as neighbouring lanes: the two maxima were segmented lane scans, 17 of 124 us at 300 k points, and W1b M0 -- the same for every
// point of a pillar -- was multiplied once per point).  Now the maxima are elementwise v_max_i32 between passes, G costs 64 MFMAs per sixteen
// pillars, each pass 8 + 64.  The pillars of a batch are counting-sorted by their number of points (LDS atomics; a pillar's result does not depend on
// its column or its neighbours, so the order inside a count does not matter), a group of sixteen takes as many passes as its last pillar has
// points, and a shorter pillar repeats its last point (a maximum does not mind).  At 300 k points (1.7 per pillar): 3.1 k MFMAs per 256 pillars against
// 3.8 k, no scans.  W1a (64 registers) and W0 (8) stay in registers, W1b is read from a bank-conflict-free LDS copy once per group.  Pillars of more
// than 16 points take the block-per-pillar kernel.
constexpr int kTileRows = 16;
constexpr int kTileBatchMax = 256, kTileBatchMin = 64;      // pillars per block and round of the tile kernel
constexpr int kW1bPitch = 36;                               // 36 rho mod 64 are the sixteen multiples of 4: lanes (rho, jj) read 64 different banks
#ifndef PN_PFN_NB
#define PN_PFN_NB 0       // diagnostic builds: a fixed batch
#endif
#ifndef PN_PFN_EXP
#define PN_PFN_EXP 0      // diagnostic builds (-DPN_PFN_EXP=k): 1 no groups at all, 2 layer 1 cut to one K step, 16 rows staged in memory order, 32 no point loads, 64 no fixed-point sums, 128 no pillars (prologue only), 512 no azimuth table -- wrong results, times only
#endif
__device__ __forceinline__ void pfn_32_128_tiles(const PfnArgs& a, const float* __restrict__ cs_table, const int bid, const int nblk) {
  // the batch is up to 256 pillars (one per thread), sized from the frame: a 64-pillar batch of a dense multi-sweep frame is ~110 point rows behind
  // three barrier-separated latency phases (run bounds -> order -> point rows; the per-pillar scalars)
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  constexpr int NBMAX = kTileBatchMax, CAP = kFwdPoints;
  __shared__ int m_s[NBMAX], m_e[NBMAX];
  __shared__ uint32_t m_key[NBMAX];
  __shared__ __attribute__((aligned(16))) float m_c[NBMAX][12];  // per pillar: mx my mz mr mp rc pc xc yc, canvas cell (int bits)
  __shared__ __attribute__((aligned(16))) float p_l[CAP][8];
  __shared__ __attribute__((aligned(16))) float w1b_s[128 * kW1bPitch];      // W1's columns 32 .. 63 (the half that meets the pillar maxima)
  __shared__ unsigned char perm[NBMAX];                        // the sub-batch's pillars (index in the batch) in ascending order of their point count
  __shared__ int bins[kTileRows + 1], q_stop[4], g_next;
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int V = (PN_PFN_EXP & 128) ? 0 : min(*a.v_dev, a.v_cap);
  // every block the same number of rounds: share = pillars per block, cut into ceil(share / NBMAX) equal batches
  const int share = (V + nblk - 1) / nblk, rounds = max(1, (share + NBMAX - 1) / NBMAX);
  const int NB = PN_PFN_NB ? PN_PFN_NB : min(NBMAX, max(kTileBatchMin, (share + rounds - 1) / rounds));
  const int rho = lane & 15, jj = lane >> 4;
  // A fragments: lane (row rho of the 16-row tile, k = jj of the step)
  // (through LDS: a lane's values are 4-byte pieces of as many different lines -- fetched straight from memory by every wave that was 230 MB of
  // L2 traffic and 25 us at 30 k points.  The block copies W1 / W0 once, whole lines, W1's columns xor-ed with 4 (row mod 16) so that the
  // sixteen rows a read touches lie in sixteen different banks)
  float a0[2][4], a1h[8][8];
  {
    float* w1s = &p_l[0][0];      // 128 x 64 floats: the staging rows are not in use yet
    float* w0s = &m_c[0][0];      // 32 x 16
    for (int i = tid; i < 128 * 16; i += 256) {
      const int row = i >> 4, c4 = (i & 15) * 4;
      const f32x4 w = *reinterpret_cast<const f32x4*>(a.w1 + row * 64 + c4);
      *reinterpret_cast<f32x4*>(w1s + row * 64 + (c4 ^ (4 * (row & 15)))) = w;
      if (c4 >= 32) *reinterpret_cast<f32x4*>(w1b_s + row * kW1bPitch + (c4 - 32)) = w;
    }
    for (int i = tid; i < 32 * 4; i += 256) *reinterpret_cast<f32x4*>(w0s + i * 4) = *reinterpret_cast<const f32x4*>(a.w0 + i * 4);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) a0[nt][k4] = w0s[(16 * nt + 4 * (rho & 3) + (rho >> 2)) * 16 + 4 * k4 + jj];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int i = 0; i < 8; ++i) a1h[mt][i] = w1s[(16 * mt + rho) * 64 + ((4 * i + jj) ^ (4 * rho))];
  }      // (the batch loop opens with a barrier: every wave has its fragments before the rows are staged)
  const int p1_idx = jj == 3 ? 3 : 4 + jj;                                      // y p5 p6 x
  const int p2_idx = jj == 0 ? 4 : jj == 1 ? 2 : jj == 2 ? 3 : 4;              // y z x y
  const int c2_idx = jj == 0 ? 1 : jj == 1 ? 2 : jj == 2 ? 7 : 8;              // my mz xc yc
  const float* w1b_l = w1b_s + rho * kW1bPitch + jj;      // + 16 mt pitch + 4 i: row 16 mt + rho, column 32 + 4 i + jj
  for (int v0 = bid * NB; v0 < V; v0 += nblk * NB) {
    const int nb = min(NB, V - v0);
    __syncthreads();  // the previous batch has been consumed
    if (tid < nb) {
      m_s[tid] = a.vstart[v0 + tid];
      m_e[tid] = a.vstart[v0 + tid + 1];
      m_key[tid] = a.ukeys[v0 + tid];
      if (a.cell_count) a.cell_count[a.ukeys[v0 + tid]] = 0u;      // (the index launches are done with the counter: the next frame finds it zero)
    }
    __syncthreads();
    int q0 = 0;
    while (q0 < nb) {  // block-uniform
      const int base = m_s[q0];
      if (m_e[q0] - base > kTileRows) { ++q0; continue; }  // the block-per-pillar kernel of the same call takes those
      // the sub-batch ends in front of the first pillar that is large or no longer fits the staging rows: every thread tests one pillar, a ballot
      // per wave, the minimum over the four waves through LDS (r6: the serial walk -- three dependent LDS reads per pillar, by every thread -- was
      // 21 of the kernel's 124 us at 300 k points)
      {
        const bool stop = tid > q0 && tid < nb && (m_e[tid] - m_s[tid] > kTileRows || m_e[tid] - base > CAP);
        const unsigned long long sm = __ballot(stop);
        if (lane == 0) q_stop[wib] = sm ? 64 * wib + __builtin_ctzll(sm) : nb;
        if (tid <= kTileRows) bins[tid] = 0;
        if (tid == 64) g_next = 0;
      }
      __syncthreads();
      const int q1 = min(min(q_stop[0], q_stop[1]), min(q_stop[2], q_stop[3]));
      const int npts = m_e[q1 - 1] - base;
      // thread t <-> pillar q0 + t of the sub-batch: its place among the pillars of the same point count
      const int myq = q0 + tid;
      int my_cnt = 0, my_slot = 0;
      if (myq < q1) {
        my_cnt = m_e[myq] - m_s[myq];
        my_slot = atomicAdd(&bins[my_cnt], 1);
      }
      for (int t = tid; t < npts; t += 256) {
        const float* src = a.pts + (size_t)((PN_PFN_EXP & 16) ? base + t : a.order[base + t]) * a.stride;
        float sv[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) sv[k] = (PN_PFN_EXP & 32) ? (float)(t + k) : src[k];
        *reinterpret_cast<f32x4*>(&p_l[t][0]) = f32x4{sv[0], sv[1], sv[2], sv[3]};      // two 16-byte writes per row (seven 4-byte ones at a 32-byte pitch: 8-way conflicts)
        *reinterpret_cast<f32x4*>(&p_l[t][4]) = f32x4{sv[4], sv[5], sv[6], 0.f};
      }
      __syncthreads();
      // per-pillar scalars, one THREAD per pillar: key decode, cell centre, exact fixed-point means (as in pfn_32_128_main); the pillar's place
      if (myq < q1) {
        const int q = myq;
        uint32_t key = m_key[q];
        const int ri = key % a.R; key /= a.R;
        const int ti = key % a.T; key /= a.T;
        const int bi = key / a.Z;
        const int s = m_s[q] - base, e = m_e[q] - base;
        long long sx = 0, sy = 0, sz = 0, sr = 0, sp = 0;
        for (int i = s; i < e; ++i) {
          const float* p = p_l[i];
          if (!(PN_PFN_EXP & 64)) { sr += to_fix(p[0]); sp += to_fix(p[1]); sz += to_fix(p[2]); sx += to_fix(p[3]); sy += to_fix(p[4]); }
        }
        const double inv_n = 1.0 / ((double)(e - s) * kFix);
        float* c = m_c[q];
        c[0] = (float)((double)sx * inv_n); c[1] = (float)((double)sy * inv_n); c[2] = (float)((double)sz * inv_n);
        c[3] = (float)((double)sr * inv_n); c[4] = (float)((double)sp * inv_n);
        const float rc = __fadd_rn(__fmul_rn((float)ri, a.vx), a.xoff);
        c[5] = rc;
        c[6] = __fadd_rn(__fmul_rn((float)ti, a.vy), a.yoff);
        c[7] = (PN_PFN_EXP & 512) ? rc : __fmul_rn(rc, cs_table[2 * ti]);
        c[8] = (PN_PFN_EXP & 512) ? rc : __fmul_rn(rc, cs_table[2 * ti + 1]);
        c[9] = __builtin_bit_cast(float, (bi * a.T + ti) * a.R + ri);
        int place = my_slot;
#pragma unroll
        for (int n = 1; n < kTileRows; ++n) place += n < my_cnt ? bins[n] : 0;
        perm[place] = (unsigned char)q;
      }
      __syncthreads();
      const int np = q1 - q0;
      const int ngroups = (PN_PFN_EXP & 1) ? 0 : (np + 15) >> 4;
      // groups from the last (most passes) to the first, each wave taking the next one when it is free: the sorted order makes the costs uneven
      // (64 + 72 n MFMAs for n passes), a fixed round robin left the waves at 0.7 of their mean load
      for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(&g_next, 1);
        g = ngroups - 1 - __builtin_amdgcn_readfirstlane(g);
        if (g < 0) break;
        const int pi = 16 * g + rho;
        const bool live = pi < np;
        const int q = perm[live ? pi : 16 * g];
        const int q_last = perm[min(16 * g + 15, np - 1)];
        const int n_pass = __builtin_amdgcn_readfirstlane(m_e[q_last] - m_s[q_last]);      // ascending counts: the group's last pillar has the most points
        const int row0 = m_s[q] - base, last = m_e[q] - m_s[q] - 1;
        // layer 0 of pass j (the pillar's point min(j, last)): the 16 features of pfn_32_128_main's layer0 -- rho phi z x | y p5 p6 x-mx | y-my z-mz
        // x-xc y-yc | rho-mr phi-mp rho-rc phi-pc; this lane supplies feature 4 k4 + jj of its column to K step k4: one point component (a 4-byte
        // LDS read at a lane-constant offset) minus one pillar scalar (read once per group), no selects between them
        const float* mc = m_c[q];
        const float c1 = mc[0], c2 = mc[c2_idx], c3 = mc[3 + jj];      // (c1: mx, used by lane group 3 only)
        auto layer0 = [&](int j, f32x4 (&h)[2]) {
          const float* pr = p_l[row0 + min(j, last)];
          const float p0 = pr[jj], p1 = pr[p1_idx], p2 = pr[p2_idx], p3 = pr[jj & 1];
          const float b0 = p0, b1 = jj == 3 ? p1 - c1 : p1, b2 = p2 - c2, b3 = p3 - c3;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nt][0], b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nt][1], b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nt][2], b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nt][3], b3, acc, 0, 0, 0);
            h[nt] = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, acc), i32x4{0, 0, 0, 0}));      // max(0, .) on the bit patterns (below)
          }
        };
        // maxima of values >= 0 as signed-integer maxima of their bit patterns (the order of the patterns is the order of the floats; against 0 it
        // is max(0, .) as well: every negative float, -0.0 included, is a negative integer)
        f32x4 h0[2], m0[2];
        layer0(0, h0);
        m0[0] = h0[0]; m0[1] = h0[1];
        for (int j = 1; j < n_pass; ++j) {      // wave-uniform
          f32x4 h[2];
          layer0(j, h);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) m0[nt] = __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, m0[nt]), __builtin_bit_cast(i32x4, h[nt])));
        }
        // G = W1b M0: once per pillar (the pointer depends on the group so that the 64 values are READ here -- hoisted out of the loop they are 64 more
        // registers than the wave has)
        const float* w1b_g = w1b_l + (n_pass >> 5);
        f32x4 gv[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) gv[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < ((PN_PFN_EXP & 2) ? 1 : 8); ++i) {
          const float act = m0[i >> 2][i & 3];
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) gv[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1b_g[16 * mt * kW1bPitch + 4 * i], act, gv[mt], 0, 0, 0);
        }
        // Y = G + W1a H per pass; the running maximum starts at 0 (= the ReLU)
        i32x4 fv[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) fv[mt] = i32x4{0, 0, 0, 0};
        auto layer1 = [&](const f32x4 (&h)[2]) {      // (four row tiles at a time: 16 accumulators in flight beside G and the maxima)
#pragma unroll
          for (int mh = 0; mh < 8; mh += 4) {
            f32x4 yv[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) yv[mt] = gv[mh + mt];
#pragma unroll
            for (int i = 0; i < ((PN_PFN_EXP & 2) ? 1 : 8); ++i) {
              const float act = h[i >> 2][i & 3];
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) yv[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1h[mh + mt][i], act, yv[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) fv[mh + mt] = __builtin_elementwise_max(fv[mh + mt], __builtin_bit_cast(i32x4, yv[mt]));
          }
        };
        layer1(h0);
        for (int j = 1; j < n_pass; ++j) {
          f32x4 h[2];
          layer0(j, h);
          layer1(h);
        }
        if (live) {      // accumulator register r of lane group jj = channel 16 mt + 4 jj + r: 16 bytes per store
          const int v = v0 + q;
          const int cell = reinterpret_cast<const int*>(&m_c[q][0])[9];
          if (a.feat) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) *reinterpret_cast<i32x4*>(a.feat + (size_t)v * 128 + 16 * mt + 4 * jj) = fv[mt];
          }
          if (a.canvas) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) *reinterpret_cast<i32x4*>(a.canvas + (size_t)cell * 128 + 16 * mt + 4 * jj) = fv[mt];
          }
        }
      }
      q0 = q1;
      if (q0 < nb) __syncthreads();  // the staged rows are rewritten by the next sub-batch
    }
  }
}

// ONE launch for both populations of pillars: blocks [0, main_blocks) run the batched one-wave-per-pillar path, the remaining
// blocks look for pillars of more than kHeavyPillar points.  (Two launches cost ~5 us of dependent-dispatch latency inside a
// replayed graph, more than the heavy path's own work on a sweep that has no such pillar.  Both bodies are 4-wave blocks: an
// 8-wave heavy block made the merged kernel's register budget 128 and the batched path, whose weights alone are 144 registers,
// lost 6 us to it.)
static_assert(kHeavyWaves * 64 == 256, "both bodies of dynamic_pfn_32_128_kernel are 256-thread blocks");
__global__ __launch_bounds__(256) void dynamic_pfn_32_128_main_kernel(PfnArgs a, const float* __restrict__ cs_table) {
  pfn_32_128_main(a, cs_table, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dynamic_pfn_32_128_tile_kernel(PfnArgs a, const float* __restrict__ cs_table) {
  pfn_32_128_tiles(a, cs_table, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void dynamic_pfn_32_128_heavy16_kernel(PfnArgs a, const float* __restrict__ cs_table) {
  pfn_32_128_heavy<kTileRows>(a, cs_table, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void dynamic_pfn_32_128_heavy_kernel(PfnArgs a, const float* __restrict__ cs_table) {
  pfn_32_128_heavy(a, cs_table, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void dynamic_pfn_32_128_kernel(PfnArgs a, const float* __restrict__ cs_table, int main_blocks) {
  if ((int)blockIdx.x < main_blocks) {
    pfn_32_128_main(a, cs_table, blockIdx.x, main_blocks);
  } else {
    pfn_32_128_heavy(a, cs_table, blockIdx.x - main_blocks, gridDim.x - main_blocks);
  }
}


// ---- backward of the (32, 128) pillar feature net: weight gradients only (the points are data) -----
// Forward per pillar:  h0[p] = relu(W0 d16[p]);  m0 = max_p h0[p];  y1[p] = W1a h0[p] + W1b m0;
//                      out = max_p relu(y1[p])
// Backward: the gradient of out[n] flows to the first point p*[n] that attains the maximum (if it
// is positive), dm0 to the first point q*[c] attaining m0[c].  One wave per pillar like the forward
// kernel; lane n owns rows n and n+64 of dW1, lane c < 32 row c of dW0, accumulated in registers
// over all pillars of the wave and written once to a per-wave slab; a second kernel adds the slabs
// in wave order (deterministic).  The 128 -> 32 contraction dh0[c] = sum_n dy1[n] W1a[n][c] is a
// 5-step butterfly reduce-scatter (31 shuffles) instead of 32 full wave reductions.
__device__ __forceinline__ float bit_select(unsigned mask, float a, float b) {
  // mask ? a : b on the bit patterns (one v_bfi_b32).  A plain `c ? v[i + h] : v[i]` is turned by the
  // compiler into a dynamically indexed register array = a 32-way compare/select chain per access.
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a) & mask) | (__builtin_bit_cast(unsigned, b) & ~mask));
}

template <int N>
__device__ __forceinline__ float reduce_scatter(float (&v)[N], int lane) {
  // reduce-scatter of N (16 or 32) per-lane values over the 64 lanes: lane l returns sum_lanes v[l & (N-1)]
#pragma unroll
  for (int h = N / 2; h >= 1; h >>= 1) {
    const unsigned up = (lane & h) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < h; ++i) {
      const float lo = v[i], hi = v[i + h];
      const float keep = bit_select(up, hi, lo), send = bit_select(up, lo, hi);
      v[i] = keep + __shfl_xor(send, h, 64);
    }
  }
  float r = v[0];
#pragma unroll
  for (int m = N; m < 64; m <<= 1) r += __shfl_xor(r, m, 64);
  return r;
}

// One launch covers 64 rows of W1 per blockIdx.y ("pass"): lane n owns row 64*pass + n.  Keeping a
// single W1 row (+ its gradient row) per lane holds the kernel at ~200 VGPRs (two waves per SIMD, no
// scratch); the layer-0 part (dm0, dh0 -> dW0) is linear in the rows, so every pass contributes a
// partial dW0 and the reduce kernel adds them.
// slab per (pass, wave): dW1 rows [64*pass, 64*pass+64) x (2*C0), then the partial dW0 (C0, 16)
constexpr int kBwdBatch = 32;  // pillars staged per block and round
constexpr int kBwdPts = 4;     // points per pillar staged in LDS (pillars with more read global memory)

template <int C0, int C1>
__global__ __launch_bounds__(256) void dynamic_pfn_bwd_kernel(PfnArgs a, const float* __restrict__ cs_table,
                                                              const float* __restrict__ dfeat, const float* __restrict__ dcanvas,
                                                              float* __restrict__ slabs, int slab_waves, int skip_single) {
  constexpr int K1 = 2 * C0;             // row length of W1
  constexpr int SLAB = 64 * K1 + C0 * 16;
  constexpr int NB = kBwdBatch, KP = kBwdPts;
  // A wave working alone through vstart -> order -> point -> gradient row pays four dependent memory
  // latencies per pillar (one wave per SIMD: nothing else to switch to).  The block therefore stages a
  // batch of NB pillars in LDS with all 256 threads loading in parallel -- three latencies per batch --
  // and each wave then works through its share from LDS.
  __shared__ int m_s[NB], m_e[NB];
  __shared__ uint32_t m_key[NB];
  __shared__ float p_l[NB][KP][8];
  __shared__ float d_l[NB][64];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int wave = blockIdx.x * 4 + wib;
  const int row0 = blockIdx.y * 64, row = row0 + lane;
  const bool la = row < C1;
  const int V = min(*a.v_dev, a.v_cap);
  float w0[16], w1[K1];
#pragma unroll
  for (int k = 0; k < 16; ++k) w0[k] = lane < C0 ? a.w0[lane * 16 + k] : 0.f;
#pragma unroll
  for (int k = 0; k < K1; ++k) w1[k] = la ? a.w1[row * K1 + k] : 0.f;
  float dw1[K1], dw0[16];
#pragma unroll
  for (int k = 0; k < K1; ++k) dw1[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) dw0[k] = 0.f;

  // Batches of NB pillars.  Without pfn_bwd_single_kernel: consecutive pillars.  With it (skip_single): the block walks chunks of 256
  // pillars, compacts the ones with another point count into a queue IN PILLAR ORDER (ballots + the waves' counts: the same queue on
  // every run) and stages batches from the queue -- 7 % of the pillars on the nuScenes grid, where skipping inside consecutive batches
  // still paid the three staging latencies of every batch (188 us).
  __shared__ int m_v[NB];
  __shared__ int q_ids[256];
  __shared__ int q_cnt[4];
  const int outer_step = skip_single ? 256 : NB;
  for (int c0v = blockIdx.x * outer_step; c0v < V; c0v += gridDim.x * outer_step) {
    int qtotal = NB;
    if (skip_single) {
      __syncthreads();  // the previous chunk's queue has been consumed
      const int v = c0v + tid;
      const bool keep = v < V && (a.vstart[v + 1] - a.vstart[v]) != 1;
      const unsigned long long bal = __ballot(keep);
      if (lane == 0) q_cnt[wib] = __popcll(bal);
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wib; ++w) base += q_cnt[w];
      if (keep) q_ids[base + __popcll(bal & ((1ull << lane) - 1ull))] = v;
      qtotal = q_cnt[0] + q_cnt[1] + q_cnt[2] + q_cnt[3];
    }
    for (int b0 = 0; b0 < qtotal; b0 += NB) {
    __syncthreads();  // the previous batch has been consumed (and the queue is complete)
    if (tid < NB) {
      int v = skip_single ? (b0 + tid < qtotal ? q_ids[b0 + tid] : -1) : c0v + tid;
      if (v >= V) v = -1;
      m_v[tid] = v;
      m_s[tid] = v >= 0 ? a.vstart[v] : 0;
      m_e[tid] = v >= 0 ? a.vstart[v + 1] : 0;
      m_key[tid] = v >= 0 ? a.ukeys[v] : 0u;
    }
    __syncthreads();
    if (tid < NB * KP) {
      const int p = tid / KP, j = tid - p * KP;
      const int s = m_s[p], n = m_e[p] - s;
      if (j < n && n <= KP) {
        const float* src = a.pts + (size_t)a.order[s + j] * a.stride;
#pragma unroll
        for (int k = 0; k < 7; ++k) p_l[p][j][k] = src[k];
      }
    }
#pragma unroll
    for (int k = 0; k < NB * 64 / 256; ++k) {
      const int idx = tid + 256 * k, p = idx >> 6, r = idx & 63;
      float g = 0.f;
      if (m_v[p] >= 0 && row0 + r < C1) {
        if (dcanvas) {
          uint32_t key = m_key[p];
          const int ri = key % a.R; key /= a.R;
          const int ti = key % a.T; key /= a.T;
          const int bi = key / a.Z;
          g = dcanvas[(((size_t)bi * a.T + ti) * a.R + ri) * C1 + row0 + r];
        } else {
          g = dfeat[(size_t)m_v[p] * C1 + row0 + r];
        }
      }
      d_l[p][r] = g;
    }
    __syncthreads();

    for (int q = wib; q < NB && m_v[q] >= 0; q += 4) {
      const int s = m_s[q], e = m_e[q], n = e - s;
      uint32_t key = m_key[q];
      const int ri = key % a.R; key /= a.R;
      const int ti = key % a.T;
      const float rc = __fadd_rn(__fmul_rn((float)ri, a.vx), a.xoff);
      const float pc = __fadd_rn(__fmul_rn((float)ti, a.vy), a.yoff);
      const float xc = __fmul_rn(rc, cs_table[2 * ti]), yc = __fmul_rn(rc, cs_table[2 * ti + 1]);
      float mx, my, mz, mr, mp;
      float d[16];
      auto decorate = [&](const float* p) {
        const float rho = p[0], phi = p[1], z = p[2], x = p[3], y = p[4];
        const float t[16] = {rho, phi, z, x, y, p[5], p[6], x - mx, y - my, z - mz, x - xc, y - yc, rho - mr, phi - mp, rho - rc, phi - pc};
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = t[k];
      };
      auto layer0 = [&]() -> float {
        float h = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) h = fmaf(w0[k], d[k], h);
        return h > 0.f ? h : 0.f;
      };
      float u[C0];
      if (n <= KP) {
        // ---- everything from LDS; the layer-0 activations of the (at most KP) points stay in registers
        long long sx = 0, sy = 0, sz = 0, sr = 0, sp = 0;
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < n) {
            const float* p = p_l[q][j];
            sr += to_fix(p[0]); sp += to_fix(p[1]); sz += to_fix(p[2]); sx += to_fix(p[3]); sy += to_fix(p[4]);
          }
        const double inv_n = 1.0 / ((double)n * kFix);
        mx = (float)((double)sx * inv_n); my = (float)((double)sy * inv_n); mz = (float)((double)sz * inv_n);
        mr = (float)((double)sr * inv_n); mp = (float)((double)sp * inv_n);
        float h[KP], m0 = 0.f;
        int q0 = -1;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          h[j] = 0.f;
          if (j < n) {
            decorate(p_l[q][j]);
            h[j] = layer0();
            if (h[j] > m0) { m0 = h[j]; q0 = j; }
          }
        }
        float g0 = 0.f;
#pragma unroll
        for (int c = 0; c < C0; ++c) g0 = fmaf(w1[C0 + c], lane_bcast(m0, c), g0);
        float f0 = 0.f;
        int pa = -1;
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < n) {
            float y0 = g0;
#pragma unroll
            for (int c = 0; c < C0; ++c) y0 = fmaf(w1[c], lane_bcast(h[j], c), y0);
            if (y0 > f0) { f0 = y0; pa = j; }
          }
        const float da = (la && pa >= 0) ? d_l[q][lane] : 0.f;
#pragma unroll
        for (int c = 0; c < C0; ++c) {
          dw1[C0 + c] = fmaf(da, lane_bcast(m0, c), dw1[C0 + c]);
          u[c] = da * w1[C0 + c];
        }
        const float dm0 = reduce_scatter<C0>(u, lane);
#pragma unroll
        for (int j = 0; j < KP; ++j)
          if (j < n && __ballot((pa == j) || (q0 == j)) != 0ull) {
            const float ea = pa == j ? da : 0.f;
#pragma unroll
            for (int c = 0; c < C0; ++c) {
              dw1[c] = fmaf(ea, lane_bcast(h[j], c), dw1[c]);
              u[c] = ea * w1[c];
            }
            float dh = reduce_scatter<C0>(u, lane);
            if (q0 == j) dh += dm0;
            const float dy0 = (lane < C0 && h[j] > 0.f) ? dh : 0.f;
            decorate(p_l[q][j]);
#pragma unroll
            for (int k = 0; k < 16; ++k) dw0[k] = fmaf(dy0, d[k], dw0[k]);
          }
        continue;
      }
      // ---- general path: points from global memory, three passes
      long long sx = 0, sy = 0, sz = 0, sr = 0, sp = 0;
      for (int i = s + lane; i < e; i += 64) {
        const float* p = a.pts + (size_t)a.order[i] * a.stride;
        sr += to_fix(p[0]); sp += to_fix(p[1]); sz += to_fix(p[2]); sx += to_fix(p[3]); sy += to_fix(p[4]);
      }
      const double inv_n = 1.0 / ((double)n * kFix);
      mx = (float)((double)pn::wave_sum(sx) * inv_n); my = (float)((double)pn::wave_sum(sy) * inv_n);
      mz = (float)((double)pn::wave_sum(sz) * inv_n); mr = (float)((double)pn::wave_sum(sr) * inv_n);
      mp = (float)((double)pn::wave_sum(sp) * inv_n);
      float m0 = 0.f;
      int q0 = -1;
      for (int i = s; i < e; ++i) {
        decorate(a.pts + (size_t)a.order[i] * a.stride);
        const float h = layer0();
        if (h > m0) { m0 = h; q0 = i; }
      }
      float g0 = 0.f;
#pragma unroll
      for (int c = 0; c < C0; ++c) g0 = fmaf(w1[C0 + c], lane_bcast(m0, c), g0);
      float f0 = 0.f;
      int pa = -1;
      for (int i = s; i < e; ++i) {
        decorate(a.pts + (size_t)a.order[i] * a.stride);
        const float h = layer0();
        float y0 = g0;
#pragma unroll
        for (int c = 0; c < C0; ++c) y0 = fmaf(w1[c], lane_bcast(h, c), y0);
        if (y0 > f0) { f0 = y0; pa = i; }
      }
      const float da = (la && pa >= 0) ? d_l[q][lane] : 0.f;
#pragma unroll
      for (int c = 0; c < C0; ++c) {
        dw1[C0 + c] = fmaf(da, lane_bcast(m0, c), dw1[C0 + c]);
        u[c] = da * w1[C0 + c];
      }
      const float dm0 = reduce_scatter<C0>(u, lane);
      for (int i = s; i < e; ++i) {
        if (__ballot((pa == i) || (q0 == i)) == 0ull) continue;  // this point receives no gradient
        decorate(a.pts + (size_t)a.order[i] * a.stride);
        const float h = layer0();
        const float ea = pa == i ? da : 0.f;
#pragma unroll
        for (int c = 0; c < C0; ++c) {
          dw1[c] = fmaf(ea, lane_bcast(h, c), dw1[c]);
          u[c] = ea * w1[c];
        }
        float dh = reduce_scatter<C0>(u, lane);
        if (q0 == i) dh += dm0;
        const float dy0 = (lane < C0 && h > 0.f) ? dh : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dw0[k] = fmaf(dy0, d[k], dw0[k]);
      }
    }
    }   // batches of the chunk
  }
  float* slab = slabs + ((size_t)blockIdx.y * slab_waves + wave) * SLAB;
#pragma unroll
  for (int k = 0; k < K1; ++k) slab[lane * K1 + k] = dw1[k];
  if (lane < C0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) slab[64 * K1 + lane * 16 + k] = dw0[k];
  }
}

// ---- r4: the ONE-POINT pillars of the (32, 128) reader on the matrix cores ----------------------------------------------------------
// On the reference's 0.098 m x 0.0123 rad grid a 30k-point sweep fills 28k pillars: 93 % of them hold a single point, and the kernel
// above spends a wave, two 31-step cross-lane reduce-scatters and ~4 us on each (497 us for the 113 k pillars of a batch of 4, alone at
// the end of the iteration).  With one point the maxima are the point itself (m0 = h, argmax = the point wherever h > 0 / y > 0), so a
// tile of 32 such pillars is five small GEMMs on v_mfma_f32_32x32x2_f32 (p = pillar, c = layer-0 channel, row = layer-1 row):
//   H  [p][c]   = relu(D [p][16] W0^T)                         D = the decorated point (same arithmetic as the kernel above)
//   Y  [p][row] = H Wsum^T,  Wsum[row][c] = W1[row][c] + W1[row][32 + c]   (the point and the pillar maximum are the same vector)
//   DA [p][row] = Y > 0 ? dY[p][row] : 0
//   dW1[row][c] += DA^T H      (both halves of W1's row receive it)         dH[p][c] = DA Wsum,  DY0 = H > 0 ? dH : 0
//   dW0[c][k]  += DY0^T D
// One wave per tile, operands re-laid through the wave's own LDS strip (H, DA, DY0, D), Wsum staged once per block; the four waves of a block
// join their accumulators in a fixed order and write ONE slab pair in the layout of the kernel above, pfn_bwd_reduce_kernel adds them
// all.  Pillars with another point count are zero rows of a tile (dynamic_pfn_bwd_kernel skips the one-point ones in turn).
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kS1H = 33, kS1D = 17;
constexpr int kS1WaveFloats = 3 * 32 * kS1H + 32 * kS1D + 64;   // H, DA tile, DY0 (32 x 33 each), D (32 x 17), 32 row offsets (64-bit): with Wsum 76 KB per block, two blocks per CU
constexpr int kS1WsFloats = 128 * kS1H;
constexpr int kS1Waves = 8;     // two waves per SIMD: a tile's dependent index / point / dY loads hide behind the other wave's MFMAs
constexpr size_t kS1Smem = (size_t)(kS1WsFloats + kS1Waves * kS1WaveFloats) * sizeof(float);

__global__ __launch_bounds__(64 * kS1Waves) void pfn_bwd_single_kernel(PfnArgs a, const float* __restrict__ cs_table, const float* __restrict__ dfeat,
                                                             const float* __restrict__ dcanvas, float* __restrict__ slabs, int slab_waves, int wave_base) {
  constexpr int C0 = 32, C1 = 128, K1 = 64, SLAB = 64 * K1 + C0 * 16;
  extern __shared__ __attribute__((aligned(16))) float s1_lds[];
  float* Ws = s1_lds;
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6, li = lane & 31, lh = lane >> 5;
  float* Hs = s1_lds + kS1WsFloats + wib * kS1WaveFloats;
  float* DAs = Hs + 32 * kS1H;
  float* Y0s = DAs + 32 * kS1H;
  float* Ds = Y0s + 32 * kS1H;
  unsigned long long* Offs = reinterpret_cast<unsigned long long*>(Ds + 32 * kS1D);
  for (int i = tid; i < C1 * C0; i += 64 * kS1Waves) {
    const int row = i >> 5, c = i & 31;
    Ws[row * kS1H + c] = a.w1[row * K1 + c] + a.w1[row * K1 + C0 + c];
  }
  float w0r[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w0r[k] = a.w0[li * 16 + k];
  __syncthreads();
  f32x16 dw1acc[4];
  f32x16 dw0acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw0acc[r] = 0.f; dw1acc[0][r] = dw1acc[1][r] = dw1acc[2][r] = dw1acc[3][r] = 0.f; }
  const int V = min(*a.v_dev, a.v_cap);
  const int ntiles = (V + 31) / 32;
  const int wave = blockIdx.x * kS1Waves + wib, nwaves = gridDim.x * kS1Waves;
  auto prow = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * lh; };
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    const int v = tile * 32 + li;
    int s = 0, n = 0;
    uint32_t key = 0;
    if (v < V) { s = a.vstart[v]; n = a.vstart[v + 1] - s; key = a.ukeys[v]; }
    const bool one = n == 1;
    if (__ballot(one) == 0ull) continue;        // wave-uniform: no one-point pillar in the tile
    float d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = 0.f;
    uint32_t kk_ = key;
    const int ri = kk_ % a.R; kk_ /= a.R;
    const int ti = kk_ % a.T; kk_ /= a.T;
    const int bi = kk_ / a.Z;
    if (one) {
      const float* p = a.pts + (size_t)a.order[s] * a.stride;
      const float rho = p[0], phi = p[1], z = p[2], x = p[3], y = p[4], p5 = p[5], p6 = p[6];
      const float rc = __fadd_rn(__fmul_rn((float)ri, a.vx), a.xoff);
      const float pc = __fadd_rn(__fmul_rn((float)ti, a.vy), a.yoff);
      const float xc = __fmul_rn(rc, cs_table[2 * ti]), yc = __fmul_rn(rc, cs_table[2 * ti + 1]);
      const double inv_n = 1.0 / kFix;
      const float mx = (float)((double)to_fix(x) * inv_n), my = (float)((double)to_fix(y) * inv_n), mz = (float)((double)to_fix(z) * inv_n);
      const float mr = (float)((double)to_fix(rho) * inv_n), mp = (float)((double)to_fix(phi) * inv_n);
      const float t[16] = {rho, phi, z, x, y, p5, p6, x - mx, y - my, z - mz, x - xc, y - yc, rho - mr, phi - mp, rho - rc, phi - pc};
#pragma unroll
      for (int k = 0; k < 16; ++k) d[k] = t[k];
    }
    if (lh == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) Ds[li * kS1D + k] = d[k];
      Offs[li] = v >= V ? 0ull : dcanvas ? ((((unsigned long long)bi * a.T + ti) * a.R + ri) * C1) : ((unsigned long long)v * C1);
    }
    // H = relu(D W0^T)
    f32x16 H;
#pragma unroll
    for (int r = 0; r < 16; ++r) H[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) H = __builtin_amdgcn_mfma_f32_32x32x2f32(lh ? d[2 * kk + 1] : d[2 * kk], lh ? w0r[2 * kk + 1] : w0r[2 * kk], H, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) { H[r] = H[r] > 0.f ? H[r] : 0.f; Hs[prow(r) * kS1H + li] = H[r]; }
    // per 32-row tile nt of layer 1:  Y = H Wsum^T,  DA = Y > 0 ? dY : 0,  dW1[nt] += DA^T H,  dH += DA Wsum[nt rows]
    // DA leaves the MFMA as [p][row] with the row on the lane: that IS the A operand of DA^T H when K step (r, lane half) stands for
    // pillar prow(r) -- and H's accumulators hold exactly those pillars for the B operand -- so dW1 needs no LDS at all; only dH wants DA
    // with the pillar on the lane and goes through a 32 x 32 strip (per tile: 4 KB, not the 16 KB of the whole DA).
    float ha[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) ha[kk] = Hs[li * kS1H + 2 * kk + lh];
    const float* dsrc = dcanvas ? dcanvas : dfeat;
    f32x16 dH;
#pragma unroll
    for (int r = 0; r < 16; ++r) dH[r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float g[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = dsrc[Offs[prow(r)] + 32 * nt + li];   // unconditional (rows past V read row 0 and are masked by Y = 0): a branch here
                                                                                // cuts the MFMA run into basic blocks with accumulator copies between them
      f32x16 Y;
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[r] = 0.f;
      float wy[16];      // (operands of a whole MFMA run read first: one wait instead of a read - wait - MFMA chain at one wave per SIMD)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) wy[kk] = Ws[(32 * nt + li) * kS1H + 2 * kk + lh];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) Y = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[kk], wy[kk], Y, 0, 0, 0);
      float da[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { da[r] = Y[r] > 0.f ? g[r] : 0.f; DAs[prow(r) * kS1H + li] = da[r]; }
#pragma unroll
      for (int r = 0; r < 16; ++r) dw1acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(da[r], H[r], dw1acc[nt], 0, 0, 0);
      float ad[16], wd[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { ad[kk] = DAs[li * kS1H + 2 * kk + lh]; wd[kk] = Ws[(32 * nt + 2 * kk + lh) * kS1H + li]; }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) dH = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[kk], wd[kk], dH, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Y0s[prow(r) * kS1H + li] = H[r] > 0.f ? dH[r] : 0.f;
    // dW0 += DY0^T D
    float ay[16], bd[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { ay[kk] = Y0s[(2 * kk + lh) * kS1H + li]; bd[kk] = li < 16 ? Ds[(2 * kk + lh) * kS1D + li] : 0.f; }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) dw0acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[kk], bd[kk], dw0acc, 0, 0, 0);
  }
  // the block's four waves join (wave order), one 32-row tile of dW1 at a time through their strips, and write one slab per pass: rows x
  // (h half | m0 half, the same values), then dW0 with pass 0
  float* mine = s1_lds + kS1WsFloats + wib * kS1WaveFloats;
  const float* w0s = s1_lds + kS1WsFloats;
  const int bslot = wave_base + blockIdx.x;
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    __syncthreads();
    if (nt < 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[prow(r) * 32 + li] = dw1acc[nt < 4 ? nt : 0][r];
    } else if (li < 16) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[prow(r) * 16 + li] = dw0acc[r];
    }
    __syncthreads();
    const int count = nt < 4 ? 1024 : 512;
    for (int i = tid; i < count; i += 64 * kS1Waves) {
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < kS1Waves; wv += 4)     // wave order, four at a time
        t += (w0s[wv * kS1WaveFloats + i] + w0s[(wv + 1) * kS1WaveFloats + i]) + (w0s[(wv + 2) * kS1WaveFloats + i] + w0s[(wv + 3) * kS1WaveFloats + i]);
      if (nt < 4) {
        const int row = 32 * nt + (i >> 5), c = i & 31, pass = row >> 6, lrow = row & 63;
        float* slab = slabs + ((size_t)pass * slab_waves + bslot) * SLAB;
        slab[lrow * K1 + c] = t;
        slab[lrow * K1 + C0 + c] = t;
      } else {
        slabs[((size_t)0 * slab_waves + bslot) * SLAB + 64 * K1 + i] = t;
        slabs[((size_t)1 * slab_waves + bslot) * SLAB + 64 * K1 + i] = 0.f;
      }
    }
  }
}

// one output per 4 threads: thread part p adds the slabs of the waves w = p (mod 4) in order, the four partial
// sums are combined in a fixed order
__global__ __launch_bounds__(256) void pfn_bwd_reduce_kernel(const float* __restrict__ slabs, int nwaves, int passes, int c0, int c1,
                                                             float* __restrict__ dw0, float* __restrict__ dw1, int accumulate) {
  __shared__ float red[4][64];
  const int k1 = 2 * c0, slab = 64 * k1 + c0 * 16, w1_floats = c1 * k1;
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float t = 0.f;
  if (o < w1_floats + c0 * 16) {
    // eight slabs per trip, their loads in flight together, eight running sums joined in a fixed order (r4: one dependent load per trip
    // made the 38 MB of slabs a 124 us launch at the very end of the iteration, where nothing overlaps it)
    auto sum_pass = [&](int ps, int local) {
      const float* base = slabs + (size_t)ps * nwaves * slab + local;
      float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int w = part;
      for (; w + 28 < nwaves; w += 32) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base[(size_t)(w + 4 * j) * slab];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += v[j];
      }
      for (int j = 0; w < nwaves; w += 4, ++j) u[j & 7] += base[(size_t)w * slab];
      return ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    };
    if (o < w1_floats) {
      const int row = o / k1, pass = row >> 6, local = (row & 63) * k1 + (o - row * k1);
      t = sum_pass(pass, local);
    } else {
      const int local = 64 * k1 + (o - w1_floats);
      for (int ps = 0; ps < passes; ++ps) t += sum_pass(ps, local);
    }
  }
  red[part][threadIdx.x & 63] = t;
  __syncthreads();
  if (part == 0 && o < w1_floats + c0 * 16) {
    const int l = threadIdx.x;
    const float sum = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    float* dst = o < w1_floats ? dw1 + o : dw0 + (o - w1_floats);
    *dst = accumulate ? *dst + sum : sum;
  }
}

// cos / sin of every pillar-centre azimuth of the grid: table[2*t] = cos(t*vy + yoff), [2*t+1] = sin(...)
__global__ void center_table_kernel(int T, float vy, float yoff, float* __restrict__ table) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float pc = __fadd_rn(__fmul_rn((float)t, vy), yoff);
  table[2 * t] = cosf(pc);
  table[2 * t + 1] = sinf(pc);
}

__global__ void scatter_canvas_kernel(const float* __restrict__ feat, const int64_t* __restrict__ unq,
                                      const int32_t* __restrict__ v_dev, int v_cap, int c, int T, int R,
                                      float* __restrict__ canvas) {
  const int V = min(*v_dev, v_cap);
  const size_t total = (size_t)V * c;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(i / c), k = (int)(i - (size_t)v * c);
    const int64_t* u = unq + (size_t)v * 4;
    canvas[(((size_t)u[0] * T + u[2]) * R + u[3]) * c + k] = feat[i];
  }
}

// zero the canvas cells of the given voxel keys: 16 bytes per thread, c / 4 threads per cell
__global__ void clear_canvas_cells_kernel(const uint32_t* __restrict__ ukeys, const int32_t* __restrict__ v_dev, int v_cap, int c4,
                                          int Z, int T, int R, float* __restrict__ canvas) {
  const int V = min(*v_dev, v_cap);
  const size_t total = (size_t)V * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(i / c4), k = (int)(i - (size_t)v * c4);
    uint32_t key = ukeys[v];
    const int ri = key % R; key /= R;
    const int ti = key % T; key /= T;
    const int bi = key / Z;
    reinterpret_cast<float4*>(canvas)[(((size_t)bi * T + ti) * R + ri) * c4 + k] = float4{0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace

extern "C" {

int pn_clear_canvas_cells(const uint32_t* unq_keys, const int32_t* num_voxels, int v_capacity, const int32_t* grid, int c,
                          float* canvas, pn_stream_t stream) {
  PN_REQUIRE(unq_keys && num_voxels && grid && canvas && c >= 4 && c % 4 == 0, "clear_canvas_cells: bad arguments");
  PN_REQUIRE(((uintptr_t)canvas & 15) == 0, "clear_canvas_cells: canvas must be 16-byte aligned");
  if (v_capacity == 0) return PN_OK;
  const size_t total = (size_t)v_capacity * (c / 4);
  hipLaunchKernelGGL(clear_canvas_cells_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, pn::S(stream),
                     unq_keys, num_voxels, v_capacity, c / 4, grid[2], grid[1], grid[0], canvas);
  return pn::check_launch("clear_canvas_cells_kernel");
}

int pn_scatter_mean_f32(const float* points, int point_stride, int f, const int32_t* voxel_start, const int32_t* order,
                        const int32_t* num_voxels, int v_capacity, float* mean, pn_stream_t stream) {
  PN_REQUIRE(points && voxel_start && order && num_voxels && mean, "scatter_mean: null pointer");
  PN_REQUIRE(f >= 1 && point_stride >= f && v_capacity >= 0, "scatter_mean: bad sizes");
  if (v_capacity == 0) return PN_OK;
  const int blocks = std::min(2048, pn::cdiv(v_capacity, 4));
  hipLaunchKernelGGL(scatter_mean_kernel, dim3(blocks), dim3(256), 0, pn::S(stream), points, point_stride, f, voxel_start,
                     order, num_voxels, v_capacity, mean);
  return pn::check_launch("scatter_mean_kernel");
}

int pn_hard_voxel_mean_f32(const float* voxels, const int32_t* num_points, int v, int p, int f, float* mean,
                           pn_stream_t stream) {
  PN_REQUIRE(voxels && num_points && mean && v >= 0 && p >= 1 && f >= 1, "hard_voxel_mean: bad arguments");
  if (v == 0) return PN_OK;
  hipLaunchKernelGGL(hard_voxel_mean_kernel, dim3(pn::cdiv((long long)v * f, 256)), dim3(256), 0, pn::S(stream), voxels,
                     num_points, v, p, f, mean);
  return pn::check_launch("hard_voxel_mean_kernel");
}

int pn_dynamic_pfn_fwd(const float* points, int point_stride, const int32_t* voxel_start, const int32_t* order,
                       const int32_t* num_voxels, int v_capacity, const uint32_t* unq_keys, const int32_t* grid,
                       const float* w0, int c0, const float* w1, int c1, float vx, float vy, float x_offset,
                       float y_offset, float* features, float* canvas, pn_stream_t stream) {
  PN_REQUIRE(points && voxel_start && order && num_voxels && unq_keys && grid && w0 && w1, "dynamic_pfn: null pointer");
  PN_REQUIRE(point_stride >= 7, "dynamic_pfn: points need 7 features [rho,phi,z,x,y,i,t]");
  PN_REQUIRE(c0 >= 1 && c0 <= 64 && c1 >= 1 && c1 <= 128, "dynamic_pfn: supports C0 <= 64, C1 <= 128");
  PN_REQUIRE(features || canvas, "dynamic_pfn: no output requested");
  if (v_capacity == 0) return PN_OK;
  PfnArgs a{points, point_stride, voxel_start, order, num_voxels, v_capacity, unq_keys, grid[0], grid[1], grid[2],
            w0, c0, w1, c1, vx, vy, x_offset, y_offset, features, canvas};
  const size_t smem = (size_t)(16 * c0 + 2 * c0 * c1 + kPfnWaves * 64) * sizeof(float);
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dynamic_pfn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((16 * 64 + 2 * 64 * 128 + kPfnWaves * 64) * sizeof(float)));
  }
  const int blocks = std::max(1, std::min(1024, pn::cdiv(v_capacity, kPfnWaves * 2)));
  hipLaunchKernelGGL(dynamic_pfn_kernel, dim3(blocks), dim3(kPfnWaves * 64), smem, pn::S(stream), a);
  return pn::check_launch("dynamic_pfn_kernel");
}

size_t pn_pfn_center_table_floats(int t) { return (size_t)2 * t; }

int pn_pfn_center_table_f32(int t, float vy, float y_offset, float* table, pn_stream_t stream) {
  PN_REQUIRE(table && t >= 1, "pfn_center_table: bad arguments");
  hipLaunchKernelGGL(center_table_kernel, dim3(pn::cdiv(t, 256)), dim3(256), 0, pn::S(stream), t, vy, y_offset, table);
  return pn::check_launch("center_table_kernel");
}

// same contract as pn_dynamic_pfn_fwd plus the azimuth table of pn_pfn_center_table_f32; takes the
// register-resident fast path when (C0, C1) == (32, 128), otherwise falls back to the generic kernel
static int dynamic_pfn_fwd_table(const float* points, int point_stride, const int32_t* voxel_start, const int32_t* order,
                                 const int32_t* num_voxels, int v_capacity, const uint32_t* unq_keys, const int32_t* grid,
                                 const float* w0, int c0, const float* w1, int c1, float vx, float vy, float x_offset, float y_offset,
                                 const float* center_table, float* features, float* canvas, uint32_t* cell_count, pn_stream_t stream) {
  if (!(c0 == 32 && c1 == 128 && center_table))
    return pn_dynamic_pfn_fwd(points, point_stride, voxel_start, order, num_voxels, v_capacity, unq_keys, grid, w0, c0, w1, c1, vx, vy,
                              x_offset, y_offset, features, canvas, stream);
  PN_REQUIRE(points && voxel_start && order && num_voxels && unq_keys && grid && w0 && w1, "dynamic_pfn: null pointer");
  PN_REQUIRE(point_stride >= 7 && (features || canvas), "dynamic_pfn: bad arguments");
  if (v_capacity == 0) return PN_OK;
  PfnArgs a{points, point_stride, voxel_start, order, num_voxels, v_capacity, unq_keys, grid[0], grid[1], grid[2],
            w0, c0, w1, c1, vx, vy, x_offset, y_offset, features, canvas, cell_count};
  const int blocks = std::max(1, std::min(512, pn::cdiv(v_capacity, kFwdBatch)));  // persistent: 2 waves per SIMD, weights loaded once per wave
  // at most n / kHeavyPillar pillars can be heavy; the blocks find them by scanning voxel_start
  const int hblocks = std::max(1, std::min(256, pn::cdiv(v_capacity, kHeavyPillar)));
  // two launches by default: the merged kernel (PN_PFN_SPLIT=0) was measured 7 us SLOWER than main + heavy back to back
  // (45.3 vs 33.3 + 4.8 us inside a frame) although its register count and LDS are those of the batched path alone
  // r5 (late): point rows as 16-column MFMA tiles (same bits; PN_PFN_TILES=0: the wave-per-pillar kernel)
  static const int tiles = [] { const char* e = getenv("PN_PFN_TILES"); return e ? atoi(e) : 1; }();
  if (tiles) {
    const int hb16 = std::max(1, std::min(512, pn::cdiv(v_capacity, 64)));
    // (two launches: both bodies in one kernel measured 38 against 28 us at 30 k points and 207 against 137 at 300 k -- as for the r4 pair)
    hipLaunchKernelGGL(dynamic_pfn_32_128_tile_kernel, dim3(blocks), dim3(256), 0, pn::S(stream), a, center_table);
    hipLaunchKernelGGL(dynamic_pfn_32_128_heavy16_kernel, dim3(hb16), dim3(256), 0, pn::S(stream), a, center_table);
    return pn::check_launch("dynamic_pfn_32_128_tile_kernel");
  }
  static const int split = [] { const char* e = getenv("PN_PFN_SPLIT"); return e ? atoi(e) : 1; }();
  if (split) {
    hipLaunchKernelGGL(dynamic_pfn_32_128_main_kernel, dim3(blocks), dim3(256), 0, pn::S(stream), a, center_table);
    hipLaunchKernelGGL(dynamic_pfn_32_128_heavy_kernel, dim3(hblocks), dim3(256), 0, pn::S(stream), a, center_table);
    return pn::check_launch("dynamic_pfn_32_128_kernel (split)");
  }
  hipLaunchKernelGGL(dynamic_pfn_32_128_kernel, dim3(blocks + hblocks), dim3(kHeavyWaves * 64), 0, pn::S(stream), a, center_table, blocks);
  return pn::check_launch("dynamic_pfn_32_128_kernel");
}

// r6: the same launch also zeroes the frame index's per-cell counters (pn_voxel_index_fused_*'s cell_count) of the frame's voxels -- the index
// launches in front are done with them, and a frame engine whose canvas may stay dirty (PointPillars.forward_cart) then needs no
// pn_clear_frame_cells launch at all.  (32, 128) readers with an azimuth table only.
int pn_dynamic_pfn_fwd_table_clear(const float* points, int point_stride, const int32_t* voxel_start, const int32_t* order,
                                   const int32_t* num_voxels, int v_capacity, const uint32_t* unq_keys, const int32_t* grid,
                                   const float* w0, int c0, const float* w1, int c1, float vx, float vy, float x_offset, float y_offset,
                                   const float* center_table, float* features, float* canvas, uint32_t* cell_count, pn_stream_t stream) {
  PN_REQUIRE(c0 == 32 && c1 == 128 && center_table && cell_count, "dynamic_pfn_fwd_table_clear: the (32, 128) reader with its azimuth table and the cell counters");
  return dynamic_pfn_fwd_table(points, point_stride, voxel_start, order, num_voxels, v_capacity, unq_keys, grid, w0, c0, w1, c1, vx, vy, x_offset, y_offset,
                               center_table, features, canvas, cell_count, stream);
}

int pn_dynamic_pfn_fwd_table(const float* points, int point_stride, const int32_t* voxel_start, const int32_t* order,
                             const int32_t* num_voxels, int v_capacity, const uint32_t* unq_keys, const int32_t* grid,
                             const float* w0, int c0, const float* w1, int c1, float vx, float vy, float x_offset, float y_offset,
                             const float* center_table, float* features, float* canvas, pn_stream_t stream) {
  return dynamic_pfn_fwd_table(points, point_stride, voxel_start, order, num_voxels, v_capacity, unq_keys, grid, w0, c0, w1, c1, vx, vy, x_offset, y_offset,
                               center_table, features, canvas, nullptr, stream);
}

int pn_scatter_canvas_fwd(const float* features, const int64_t* unq, const int32_t* num_voxels, int v_capacity, int c,
                          int t, int r, float* canvas, pn_stream_t stream) {
  PN_REQUIRE(features && unq && num_voxels && canvas && c >= 1 && t >= 1 && r >= 1, "scatter_canvas: bad arguments");
  if (v_capacity == 0) return PN_OK;
  const size_t total = (size_t)v_capacity * c;
  hipLaunchKernelGGL(scatter_canvas_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                     pn::S(stream), features, unq, num_voxels, v_capacity, c, t, r, canvas);
  return pn::check_launch("scatter_canvas_kernel");
}

constexpr int kPfnBwdBlocks = 256;  // x 2 passes x 4 waves = two waves per SIMD for the (32, 128) reader
constexpr int kPfnBwdSlabMax = 64 * 64 + 32 * 16;
constexpr int kPfnBwdSingleBlocks = 256;   // pfn_bwd_single_kernel: one slab pair per block (512 blocks, two per CU: the same 88 us and more slabs to add)
static const int kPfnBwdSingle = [] { const char* e = getenv("PN_PFN_BWD_SINGLE"); return e ? atoi(e) : 1; }();

size_t pn_dynamic_pfn_bwd_workspace_bytes(void) { return (size_t)(kPfnBwdBlocks * 4 + kPfnBwdSingleBlocks) * 2 * kPfnBwdSlabMax * sizeof(float); }

int pn_dynamic_pfn_bwd(const float* points, int point_stride, const int32_t* voxel_start, const int32_t* order,
                       const int32_t* num_voxels, int v_capacity, const uint32_t* unq_keys, const int32_t* grid, const float* w0,
                       int c0, const float* w1, int c1, float vx, float vy, float x_offset, float y_offset, const float* center_table,
                       const float* d_features, const float* d_canvas, float* dw0, float* dw1, int accumulate, void* workspace,
                       size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(points && voxel_start && order && num_voxels && unq_keys && grid && w0 && w1 && center_table && dw0 && dw1 && workspace,
             "dynamic_pfn_bwd: null pointer");
  PN_REQUIRE((c0 == 32 && c1 == 128) || (c0 == 16 && c1 == 32),
             "dynamic_pfn_bwd: built for (C0, C1) = (32, 128) (nuScenes reader) and (16, 32) (reduced test model)");
  PN_REQUIRE(point_stride >= 7 && ((d_features != nullptr) != (d_canvas != nullptr)), "dynamic_pfn_bwd: give exactly one of d_features / d_canvas");
  PN_REQUIRE(workspace_bytes >= pn_dynamic_pfn_bwd_workspace_bytes(), "dynamic_pfn_bwd: workspace too small");
  PfnArgs a{points, point_stride, voxel_start, order, num_voxels, v_capacity, unq_keys, grid[0], grid[1], grid[2],
            w0, c0, w1, c1, vx, vy, x_offset, y_offset, nullptr, nullptr};
  float* slabs = static_cast<float*>(workspace);
  const int passes = pn::cdiv(c1, 64);
  const bool single = c0 == 32 && kPfnBwdSingle;
  const int slab_waves = kPfnBwdBlocks * 4 + (single ? kPfnBwdSingleBlocks : 0);
  if (single) {
    static bool done[64] = {false};
    if (pn::first_use_on_device(done))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pfn_bwd_single_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS1Smem);
    hipLaunchKernelGGL(pfn_bwd_single_kernel, dim3(kPfnBwdSingleBlocks), dim3(64 * kS1Waves), kS1Smem, pn::S(stream), a, center_table, d_features, d_canvas, slabs, slab_waves,
                       kPfnBwdBlocks * 4);
  }
  if (c0 == 32)
    hipLaunchKernelGGL((dynamic_pfn_bwd_kernel<32, 128>), dim3(kPfnBwdBlocks, passes), dim3(256), 0, pn::S(stream), a, center_table, d_features, d_canvas, slabs,
                       slab_waves, (int)single);
  else
    hipLaunchKernelGGL((dynamic_pfn_bwd_kernel<16, 32>), dim3(kPfnBwdBlocks, passes), dim3(256), 0, pn::S(stream), a, center_table, d_features, d_canvas, slabs,
                       slab_waves, 0);
  const int outs = c1 * 2 * c0 + c0 * 16;
  hipLaunchKernelGGL(pfn_bwd_reduce_kernel, dim3(pn::cdiv(outs, 64)), dim3(256), 0, pn::S(stream), slabs, slab_waves, passes, c0, c1,
                     dw0, dw1, accumulate);
  return pn::check_launch("dynamic_pfn_bwd");
}

}  // extern "C"

// =================================================================================================
// Static (hard-voxel) pillar feature net: PillarFeatureNet.forward + PFNLayer.forward_static
// (det3d/models/readers/pillar_encoder.py:74-169, 47-60), eval mode (BatchNorm1d folded into scale / shift).
// voxels (V, P, F): decoration [x, y, z, ..., x - mean(x,y,z), x - pillar centre (x, y)] (+ |xyz| when with_distance),
// padded slots zeroed (`features *= mask`), then up to two layers of Linear(no bias) -> BN -> ReLU -> max over the P
// slots.  As in the reference the PADDED slots take part in the maximum with relu(shift) (the BatchNorm shift of a zero
// row).  One wave per pillar, lane = output channel, weights transposed in LDS.
// =================================================================================================
namespace {

struct StaticPfnArgs {
  const float* vox; const int32_t* num; const int32_t* coors; const int32_t* v_dev;
  int v_cap, P, F, nin, with_dist;
  const float* w0; const float* s0; const float* h0; int c0;   // layer 0 weight (c0, nin), scale, shift
  const float* w1; const float* s1; const float* h1; int c1;   // optional layer 1 (c1, 2*c0); c1 = 0: single layer
  float vx, vy, xoff, yoff;
  float* out;  // (V, c_last)
};

constexpr int kSpWaves = 4, kSpMaxIn = 16, kSpMaxP = 32;

__global__ __launch_bounds__(kSpWaves * 64) void static_pfn_kernel(StaticPfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w0t = lds;                                   // [nin][c0]
  float* w1t = w0t + a.nin * a.c0;                    // [2*c0][c1]
  float* rows = w1t + (a.c1 ? 2 * a.c0 * a.c1 : 0);   // per wave: [P][64] layer-0 activations
  for (int i = threadIdx.x; i < a.nin * a.c0; i += blockDim.x) {
    const int k = i / a.c0, n = i - k * a.c0;
    w0t[i] = a.w0[n * a.nin + k];
  }
  if (a.c1)
    for (int i = threadIdx.x; i < 2 * a.c0 * a.c1; i += blockDim.x) {
      const int k = i / a.c1, n = i - k * a.c1;
      w1t[i] = a.w1[n * 2 * a.c0 + k];
    }
  __syncthreads();
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  float* act = rows + (size_t)wib * (a.P + 1) * 64;   // P activation rows + one row for the maxima
  const int V = min(*a.v_dev, a.v_cap);
  const float sc0 = lane < a.c0 ? a.s0[lane] : 0.f, sh0 = lane < a.c0 ? a.h0[lane] : 0.f;
  for (int v = blockIdx.x * kSpWaves + wib; v < V; v += gridDim.x * kSpWaves) {
    const int n = a.num[v];
    const float* vp = a.vox + (size_t)v * a.P * a.F;
    // mean of x, y, z over ALL P slots divided by num (the reference sums the zero padding too)
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int p = 0; p < a.P; ++p) { sx += vp[p * a.F]; sy += vp[p * a.F + 1]; sz += vp[p * a.F + 2]; }
    const float mx = sx / (float)n, my = sy / (float)n, mz = sz / (float)n;
    const float cx = (float)a.coors[(size_t)v * 4 + 3] * a.vx + a.xoff, cy = (float)a.coors[(size_t)v * 4 + 2] * a.vy + a.yoff;
    float m0 = -3.0e38f;
    for (int p = 0; p < a.P; ++p) {
      float d[kSpMaxIn];
      const bool live = p < n;
#pragma unroll
      for (int k = 0; k < kSpMaxIn; ++k) d[k] = 0.f;
      if (live) {
        const float* q = vp + p * a.F;
        for (int k = 0; k < a.F; ++k) d[k] = q[k];
        d[a.F] = q[0] - mx; d[a.F + 1] = q[1] - my; d[a.F + 2] = q[2] - mz;
        d[a.F + 3] = q[0] - cx; d[a.F + 4] = q[1] - cy;
        if (a.with_dist) d[a.F + 5] = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
      }
      float h = 0.f;
      if (lane < a.c0)
        for (int k = 0; k < a.nin; ++k) h = fmaf(w0t[k * a.c0 + lane], d[k], h);
      h = fmaf(h, sc0, sh0);
      h = h > 0.f ? h : 0.f;
      m0 = fmaxf(m0, h);
      act[p * 64 + lane] = h;
    }
    if (a.c1 == 0) {
      if (lane < a.c0) a.out[(size_t)v * a.c0 + lane] = m0;
      continue;
    }
    // layer 1 on [x_p, max]: the max half is the same for every slot
    float g0 = 0.f, g1 = 0.f;
    const bool two = a.c1 > 64;
    // broadcast m0 through LDS row P
    float* mrow = act + (size_t)a.P * 64;
    mrow[lane] = m0;
    for (int c = 0; c < a.c0; ++c) {
      const float m = mrow[c];
      if (lane < a.c1) g0 = fmaf(w1t[(a.c0 + c) * a.c1 + lane], m, g0);
      if (two && lane + 64 < a.c1) g1 = fmaf(w1t[(a.c0 + c) * a.c1 + lane + 64], m, g1);
    }
    const float s1a = lane < a.c1 ? a.s1[lane] : 0.f, h1a = lane < a.c1 ? a.h1[lane] : 0.f;
    const float s1b = (two && lane + 64 < a.c1) ? a.s1[lane + 64] : 0.f, h1b = (two && lane + 64 < a.c1) ? a.h1[lane + 64] : 0.f;
    float f0 = -3.0e38f, f1 = -3.0e38f;
    for (int p = 0; p < a.P; ++p) {
      float y0 = g0, y1 = g1;
      for (int c = 0; c < a.c0; ++c) {
        const float hc = act[p * 64 + c];
        if (lane < a.c1) y0 = fmaf(w1t[c * a.c1 + lane], hc, y0);
        if (two && lane + 64 < a.c1) y1 = fmaf(w1t[c * a.c1 + lane + 64], hc, y1);
      }
      y0 = fmaf(y0, s1a, h1a); y1 = fmaf(y1, s1b, h1b);
      f0 = fmaxf(f0, y0 > 0.f ? y0 : 0.f);
      f1 = fmaxf(f1, y1 > 0.f ? y1 : 0.f);
    }
    if (lane < a.c1) a.out[(size_t)v * a.c1 + lane] = f0;
    if (two && lane + 64 < a.c1) a.out[(size_t)v * a.c1 + lane + 64] = f1;
  }
}

}  // namespace

extern "C" {

int pn_static_pfn_fwd(const float* voxels, const int32_t* num_points, const int32_t* coors, const int32_t* num_voxels, int v_capacity, int p,
                      int f, int with_distance, const float* w0, const float* scale0, const float* shift0, int c0, const float* w1,
                      const float* scale1, const float* shift1, int c1, float vx, float vy, float x_offset, float y_offset, float* features,
                      pn_stream_t stream) {
  PN_REQUIRE(voxels && num_points && coors && num_voxels && w0 && scale0 && shift0 && features, "static_pfn: null pointer");
  const int nin = f + 5 + (with_distance ? 1 : 0);
  PN_REQUIRE(f >= 3 && nin <= kSpMaxIn && p >= 1 && p <= kSpMaxP, "static_pfn: at most 16 decorated input features and 32 points per pillar");
  PN_REQUIRE(c0 >= 1 && c0 <= 64 && (c1 == 0 || (w1 && scale1 && shift1 && c1 <= 128)), "static_pfn: supports C0 <= 64, C1 <= 128");
  if (v_capacity == 0) return PN_OK;
  StaticPfnArgs a{voxels, num_points, coors, num_voxels, v_capacity, p, f, nin, with_distance, w0, scale0, shift0, c0, w1, scale1, shift1, c1,
                  vx, vy, x_offset, y_offset, features};
  const size_t smem = (size_t)(nin * c0 + (c1 ? 2 * c0 * c1 : 0) + kSpWaves * (p + 1) * 64) * sizeof(float);
  static bool attr_done[64] = {false};
  if (pn::first_use_on_device(attr_done)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&static_pfn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  }
  PN_REQUIRE(smem <= 140 * 1024, "static_pfn: layer sizes exceed LDS");
  const int blocks = std::max(1, std::min(2048, pn::cdiv(v_capacity, kSpWaves)));
  hipLaunchKernelGGL(static_pfn_kernel, dim3(blocks), dim3(kSpWaves * 64), smem, pn::S(stream), a);
  return pn::check_launch("static_pfn_kernel");
}

}  // extern "C"
