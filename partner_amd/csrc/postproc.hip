// Decode + rotated NMS of the centre heads: head tensors -> boxes, on the device.
// Reference: decode / post_processing / predict   det3d/models/bbox_heads/center_head.py:350-402, 462-577, 404-460
//            rotate_nms_pcdet                      det3d/core/bbox/box_torch_ops.py:248-277
//            box_overlap / iou_bev / nms_kernel    det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-311
//            nms_gpu greedy reduce (on the HOST there, after a blocking D2H copy)   iou3d_nms.cpp:90-136
// Plain path only (no double flip, no stateful / per-class NMS, no panoptic, sector 0).
// Pipeline per sample, all on the caller's stream, no host round trip:
//   decode_kernel        every cell: sigmoid/max over classes, exp(dim), atan2(rot), polar cell centre -> Cartesian,
//                        score / range mask
//   select_sort_kernel   one block: order-preserving compaction of the valid cells into LDS (cap 8192), bitonic sort
//                        by (score desc, cell asc) -- torch.sort leaves ties unspecified --, first pre_max survive
//   nms_mask_kernel      64 x 64 IoU tiles -> suppression bit masks (same tiling as the reference)
//   nms_reduce_kernel    one wave walks the sorted boxes, lane l owns word l of the "removed" bit set
//   gather_kernel        kept[:post_max] -> box3d_lidar / scores / label_preds
// Rotated IoU: overlap polygon = edge crossings + contained corners (1e-2 margin), ordered by angle about their mean,
// fan-summed cross products -- the oracle's box_nms.c is the same arithmetic on the CPU.
#include "pn_common.h"
#include <type_traits>
#include "box_geom.h"
#include <algorithm>

namespace {

constexpr int kCap = 8192;   // candidates per sample that enter the sort
constexpr int kSortThreads = 1024;

using pn_geom::iou_bev;
using pn_geom::pt;

struct DecodeArgs {
  const float* hm; int hm_ps, ncls;
  const float* reg; int reg_ps;
  const float* hei; int hei_ps;
  const float* dim; int dim_ps;
  const float* rot; int rot_ps;
  const float* vel; int vel_ps;
  int B, H, W, cylinder, rectify, nb;
  float sx, sy, x0, y0, thr;
  float lo[3], hi[3];
  float* boxes;   // (B, cells, nb)
  float* score;   // (B, cells): score of valid cells, -1 otherwise
  int* label;     // (B, cells)
  // geometry-aware head (E2ESWVoteHead.decode, e2e_swv_head.py:313-366): Cartesian centre = reg + offset_grid, score rectified by
  // the IoU branch, rot = atan2(rot[1], rot[0]), rectified heading wrapped into (-pi, pi]
  int swv;
  int activated;   // double-flip path: hm already holds (averaged) probabilities and dim (averaged) sizes -- no sigmoid / exp here
  // stateful NMS across azimuth sectors (center_head.py:486-501): the sector's candidates are rotated into the sweep's frame BEFORE the
  // NMS and the previous sectors' detections are appended as `extra` further candidates per sample (rows cells .. cells + extra - 1)
  int extra;
  int pre_rot;
  float rot_c, rot_s, rot_angle;
  const float* iou; int iou_ps; int iou_factor;
  const float* grid;   // offset_grid, planar (2, H, W)
};

__global__ void decode_kernel(DecodeArgs a) {
  const int cells = a.H * a.W;
  const size_t total = (size_t)a.B * cells;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cell = (int)(i % cells);
    const int yy = cell / a.W, xx = cell - yy * a.W;
    const float* ph = a.hm + i * a.hm_ps;
    float best = -1.f;
    int lab = 0;
    float q = 1.f;
    if (a.swv && a.iou) {
      const float u = fminf(fmaxf((a.iou[i * a.iou_ps] + 1.f) * 0.5f, 0.f), 1.f);
      q = a.iou_factor == 1 ? u : powf(u, (float)a.iou_factor);
    }
    for (int c = 0; c < a.ncls; ++c) {
      const float s = (a.activated ? ph[c] : 1.f / (1.f + expf(-ph[c]))) * q;
      if (s > best) { best = s; lab = c; }   // first maximum, as torch.max
    }
    const float* pr = a.reg + i * a.reg_ps;
    float x, y, r = atan2f(a.rot[i * a.rot_ps], a.rot[i * a.rot_ps + 1]), azs = 0.f;
    if (a.swv) {
      x = pr[0] + a.grid[cell];
      y = pr[1] + a.grid[cells + cell];
      r = atan2f(a.rot[i * a.rot_ps + 1], a.rot[i * a.rot_ps]);
      if (a.rectify) {
        r += atan2f(y, x);
        if (r > 3.14159265358979323846f) r -= 6.28318530717958647692f;
        else if (r < -3.14159265358979323846f) r += 6.28318530717958647692f;
      }
    } else if (a.cylinder) {
      const float rho = (float)xx * a.sx + a.x0, az = (float)yy * a.sy + a.y0;
      x = rho * cosf(az) + pr[0];
      y = rho * sinf(az) + pr[1];
      if (a.rectify) { azs = atan2f(y, x); r += azs; }
    } else {
      x = ((float)xx + pr[0]) * a.sx + a.x0;
      y = ((float)yy + pr[1]) * a.sy + a.y0;
    }
    const float z = a.hei[i * a.hei_ps];
    const size_t oi = (size_t)(i / cells) * (cells + a.extra) + cell;   // row of this cell in the candidate arrays
    float* o = a.boxes + oi * a.nb;
    o[0] = x; o[1] = y; o[2] = z;
    const float* pd = a.dim + i * a.dim_ps;
    if (a.activated) { o[3] = pd[0]; o[4] = pd[1]; o[5] = pd[2]; }
    else { o[3] = expf(pd[0]); o[4] = expf(pd[1]); o[5] = expf(pd[2]); }
    if (a.vel) {
      float vx = a.vel[i * a.vel_ps], vy = a.vel[i * a.vel_ps + 1];
      if (a.cylinder && a.rectify) {
        const float vr = sqrtf(vx * vx + vy * vy), va = atan2f(vy, vx) + azs;
        vx = vr * cosf(va); vy = vr * sinf(va);
      }
      o[6] = vx; o[7] = vy;
    }
    o[a.nb - 1] = r;
    const bool ok = best > a.thr && x >= a.lo[0] && y >= a.lo[1] && z >= a.lo[2] && x <= a.hi[0] && y <= a.hi[1] && z <= a.hi[2];
    if (a.pre_rot) {   // after the range mask, as the reference: [x y] @ [[c, s], [-s, c]], heading -= angle, velocity like the centre
      o[0] = __fadd_rn(__fmul_rn(x, a.rot_c), __fmul_rn(y, -a.rot_s));
      o[1] = __fadd_rn(__fmul_rn(x, a.rot_s), __fmul_rn(y, a.rot_c));
      o[a.nb - 1] = r - a.rot_angle;
      if (a.vel) {
        const float vx = o[6], vy = o[7];
        o[6] = __fadd_rn(__fmul_rn(vx, a.rot_c), __fmul_rn(vy, -a.rot_s));
        o[7] = __fadd_rn(__fmul_rn(vx, a.rot_s), __fmul_rn(vy, a.rot_c));
      }
    }
    a.score[oi] = ok ? best : -1.f;
    a.label[oi] = lab;
  }
}

// rows cells .. cells + extra - 1 of every sample's candidate arrays = the detections of the previous sectors (score -1 past the count)
__global__ void append_prev_kernel(const float* __restrict__ pb, const float* __restrict__ ps, const int64_t* __restrict__ pl, const int32_t* __restrict__ pc,
                                   int batch, int cells, int extra, int nb, float* __restrict__ boxes, float* __restrict__ score, int* __restrict__ label) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * extra) return;
  const int b = i / extra, k = i - b * extra;
  const size_t row = (size_t)b * (cells + extra) + cells + k;
  const bool live = k < pc[b];
  score[row] = live ? ps[i] : -1.f;
  label[row] = live ? (int)pl[i] : 0;
  for (int c = 0; c < nb; ++c) boxes[row * nb + c] = live ? pb[(size_t)i * nb + c] : 0.f;
}

// double-flip test-time augmentation (CenterHead.double_flip_decode, center_head.py:289-346): the batch holds groups of four
// clouds [original, y -> -y, x -> -x, both]; the maps of copies 1..3 are flipped back along H / W / both, regression offsets,
// heading (sin, cos) and velocity get the signs of the mirrored frame, and the four are averaged -- probabilities (sigmoid) for the
// heat map and sizes (exp) for dim, so the decode that follows must not apply them again.
struct FlipArgs {
  const float* hm; int hm_ps, ncls;
  const float* reg; int reg_ps;
  const float* hei; int hei_ps;
  const float* dim; int dim_ps;
  const float* rot; int rot_ps;
  const float* vel; int vel_ps;
  int B, H, W;   // B = merged samples (input batch = 4 B)
  float *o_hm, *o_reg, *o_hei, *o_dim, *o_rot, *o_vel;
};
__global__ void double_flip_merge_kernel(FlipArgs a) {
  const int cells = a.H * a.W;
  const size_t total = (size_t)a.B * cells;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / cells), cell = (int)(i % cells);
    const int y = cell / a.W, x = cell - y * a.W;
    size_t src[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = (k & 1) ? a.H - 1 - y : y, xx = (k & 2) ? a.W - 1 - x : x;
      src[k] = ((size_t)(4 * b + k) * a.H + yy) * a.W + xx;
    }
    // torch.mean over the copy axis: running sum in copy order, then / 4
    for (int c = 0; c < a.ncls; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += 1.f / (1.f + expf(-a.hm[src[k] * a.hm_ps + c]));
      a.o_hm[i * a.ncls + c] = s * 0.25f;
    }
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += expf(a.dim[src[k] * a.dim_ps + c]);
      a.o_dim[i * 3 + c] = s * 0.25f;
    }
    {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a.hei[src[k] * a.hei_ps];
      a.o_hei[i] = s * 0.25f;
    }
    float rx = 0.f, ry = 0.f, rs = 0.f, rc = 0.f, vx = 0.f, vy = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* pr = a.reg + src[k] * a.reg_ps;
      rx += (k & 2) ? 1.f - pr[0] : pr[0];          // x -> -x: reg_x -> 1 - reg_x
      ry += (k & 1) ? 1.f - pr[1] : pr[1];          // y -> -y: reg_y -> 1 - reg_y
      const float* pt = a.rot + src[k] * a.rot_ps;  // [sin, cos]
      rs += (k & 2) ? -pt[0] : pt[0];
      rc += (k & 1) ? -pt[1] : pt[1];
      if (a.vel) {
        const float* pv = a.vel + src[k] * a.vel_ps;
        vx += (k & 2) ? -pv[0] : pv[0];
        vy += (k & 1) ? -pv[1] : pv[1];
      }
    }
    a.o_reg[i * 2] = rx * 0.25f; a.o_reg[i * 2 + 1] = ry * 0.25f;
    a.o_rot[i * 2] = rs * 0.25f; a.o_rot[i * 2 + 1] = rc * 0.25f;
    if (a.vel) { a.o_vel[i * 2] = vx * 0.25f; a.o_vel[i * 2 + 1] = vy * 0.25f; }
  }
}

#ifdef PN_SORT_STAMP
unsigned long long* pn_sort_stamps = nullptr;      // diagnostic build only (tools/micro/select_sort_check.hip): shader-clock stamps of the phases
#define SORT_STAMP(k) do { if (threadIdx.x == 0 && pn_sort_stamps_dev) pn_sort_stamps_dev[k] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long* pn_sort_stamps_dev = nullptr;
#else
#define SORT_STAMP(k) do { } while (0)
#endif

// one block per sample
constexpr int kHistBins = 4096;   // score bits [30:19]: exponent + 4 mantissa bits
constexpr size_t kSortLds = (size_t)kCap * sizeof(unsigned long long) + (size_t)kHistBins * sizeof(int);

__global__ __launch_bounds__(kSortThreads) void select_sort_kernel(const float* __restrict__ score, const float* __restrict__ boxes, int cells, int nb,
                                                                   int pre_max, int* __restrict__ sel_cell, float* __restrict__ nms_boxes,
                                                                   int* __restrict__ n_sel, const int* __restrict__ label,
                                                                   int* __restrict__ sel_label) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sort_lds[];
  unsigned long long* keys = sort_lds;                       // [kCap]
  int* hist = reinterpret_cast<int*>(sort_lds + kCap);       // [kHistBins]
  __shared__ int wave_cnt[kSortThreads / 64];
  __shared__ int chunk_tot[kHistBins / 64];
  __shared__ int n_valid, cut_bin, n_ge, n_kept, sub_cut;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* sc = score + (size_t)b * cells;
  SORT_STAMP(0);
  if (tid == 0) { n_valid = 0; n_kept = 0; }
  for (int i = tid; i < kHistBins; i += kSortThreads) hist[i] = 0;
  __syncthreads();
  // key: score bits (positive floats order like unsigned) then ~cell, so that a descending sort yields score descending,
  // cell ascending.  Only the first pre_max of that order are used, so:
  // ---- pass 1: histogram of the score bits [30:19] of the valid cells (integer LDS atomics: order independent)
  constexpr int kBatch = 16;   // scores per thread requested back to back (one memory latency per 16k cells instead of 16)
  {
    int mine = 0;
    for (int c0 = 0; c0 < cells; c0 += kSortThreads * kBatch) {
      float sv[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int c = c0 + k * kSortThreads + tid;
        sv[k] = c < cells ? sc[c] : -1.f;
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k)
        if (sv[k] >= 0.f) { atomicAdd(&hist[(__builtin_bit_cast(unsigned, sv[k]) >> 19) & (kHistBins - 1)], 1); ++mine; }
    }
    mine = pn::wave_sum(mine);
    if (lane == 0 && mine) atomicAdd(&n_valid, mine);
  }
  __syncthreads();
  SORT_STAMP(1);
  // ---- the bin in which the `need`-th best falls -> (bin, number of entries at or above it); bin 0 / total when fewer than `need` entries
  //      exist.  Two levels: every wave totals four 64-bin chunks, then one wave scans the 64 chunk totals from the top and the bins of
  //      the chunk that holds the `need`-th (r4; one wave walking the histogram 64 bins per step took 9 us on a score map whose top bins
  //      are empty -- 32 dependent steps before the first occupied bin).  Called by every thread of the block.
  auto find_cut = [&](int need, int total, int* bin_out, int* ge_out) {
    for (int c = wv; c < kHistBins / 64; c += kSortThreads / 64) {
      int v = hist[64 * c + lane];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) chunk_tot[c] = v;
    }
    __syncthreads();
    if (wv == 0) {
      int found = 0, ge = total;
      if (total > need) {
        auto scan_from_top = [&](int v, int base, int& pos, int& upto) {      // lane 0 = the highest entry; -> first lane whose inclusive sum reaches need
          int inc = v;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
          }
          const unsigned long long hit = __ballot(base + inc >= need);
          pos = hit ? __ffsll((long long)hit) - 1 : 63;
          upto = base + __shfl(inc, pos, 64);
        };
        int lc, upto_c;
        scan_from_top(chunk_tot[kHistBins / 64 - 1 - lane], 0, lc, upto_c);
        const int chunk = kHistBins / 64 - 1 - lc;
        const int above = upto_c - chunk_tot[chunk];                          // entries in the chunks above
        int lb, upto_b;
        scan_from_top(hist[64 * chunk + 63 - lane], above, lb, upto_b);
        found = 64 * chunk + 63 - lb;
        ge = upto_b;
      }
      if (lane == 0) { *bin_out = found; *ge_out = ge; }
    }
    __syncthreads();
  };
  find_cut(pre_max, n_valid, &cut_bin, &n_ge);
  if (tid == 0) sub_cut = 0;
  __syncthreads();
  SORT_STAMP(2);
  const unsigned cut = (unsigned)cut_bin;
  // More candidates at or above the cut bin than ONE power of two above pre_max (r4; before: than the sort buffer holds): refine INSIDE the
  // cut bin with a second histogram on the next 12 bits.  The sort then runs on ~pre_max keys instead of every key of the cut bin (a
  // 2048- or 4096-key bitonic sort was 31 - 62 us of this kernel; the extra pass over the L2-resident scores is 3 us), and on a
  // near-constant score map (an untrained head puts a whole map into one bin of the top 12 score bits) the candidates are still the
  // best by score and not the first in cell order.
  int refine_above = 1;
  while (refine_above < pre_max) refine_above <<= 1;
  refine_above = min(refine_above, kCap);
  if (n_ge > refine_above) {
    const int n_above = n_ge - hist[cut];    // entries in the bins above the cut: fewer than pre_max by construction
    __syncthreads();
    for (int i = tid; i < kHistBins; i += kSortThreads) hist[i] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < cells; c0 += kSortThreads * kBatch) {
      float sv[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int c = c0 + k * kSortThreads + tid;
        sv[k] = c < cells ? sc[c] : -1.f;
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const unsigned u = __builtin_bit_cast(unsigned, sv[k]);
        if (sv[k] >= 0.f && ((u >> 19) & (kHistBins - 1)) == cut) atomicAdd(&hist[(u >> 7) & (kHistBins - 1)], 1);
      }
    }
    __syncthreads();
    const int n_ge_before = n_ge;
    find_cut(pre_max - n_above, n_ge_before - n_above, &sub_cut, &n_ge);
    if (tid == 0) n_ge += n_above;
    __syncthreads();
  }
  SORT_STAMP(3);
  const unsigned scut = (unsigned)sub_cut;
  auto at_or_above = [&](float s) {
    const unsigned u = __builtin_bit_cast(unsigned, s), bin = (u >> 19) & (kHistBins - 1);
    return s >= 0.f && (bin > cut || (bin == cut && ((u >> 7) & (kHistBins - 1)) >= scut));
  };
  // ---- pass 2: the candidates at or above the cut (a superset of the top pre_max, usually a few hundred more) -> keys
  if (n_ge <= kCap) {
    // the sort below orders them by the full key, so the order in which they land here is irrelevant: one LDS atomic per wave
    for (int c0 = 0; c0 < cells; c0 += kSortThreads * kBatch) {
      float sv[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int c = c0 + k * kSortThreads + tid;
        sv[k] = c < cells ? sc[c] : -1.f;
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int c = c0 + k * kSortThreads + tid;
        const float s = sv[k];
        const bool v = at_or_above(s);
        const unsigned long long bal = __ballot(v);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&n_kept, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (v) keys[base + __popcll(bal & ((1ull << lane) - 1ull))] =
            ((unsigned long long)__builtin_bit_cast(unsigned, s) << 32) | (unsigned)(0xffffffffu - (unsigned)c);
      }
    }
  } else {
    // still too many: more than kCap scores agree in their top 24 bits.  First everything strictly above the refined cut, then
    // the refined cut bin itself in cell order until the sort buffer is full (ordered compaction, block scans).
    for (int phase = 0; phase < 2; ++phase)
      for (int c0 = 0; c0 < cells; c0 += kSortThreads) {
        const int c = c0 + tid;
        const float s = c < cells ? sc[c] : -1.f;
        const unsigned u = __builtin_bit_cast(unsigned, s), bin = (u >> 19) & (kHistBins - 1), sub = (u >> 7) & (kHistBins - 1);
        const bool above = bin > cut || (bin == cut && sub > scut), equal = bin == cut && sub == scut;
        const bool v = s >= 0.f && (phase == 0 ? above : equal);
        const unsigned long long bal = __ballot(v);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = n_kept;
        for (int k = 0; k < wv; ++k) off += wave_cnt[k];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (v && pos < kCap) keys[pos] = ((unsigned long long)__builtin_bit_cast(unsigned, s) << 32) | (unsigned)(0xffffffffu - (unsigned)c);
        __syncthreads();
        if (tid == 0) {
          int t = 0;
          for (int k = 0; k < kSortThreads / 64; ++k) t += wave_cnt[k];
          n_kept += t;
        }
        __syncthreads();
      }
  }
  __syncthreads();
  SORT_STAMP(4);
  const int n = min(n_kept, kCap);
  int npad = 1;
  while (npad < n) npad <<= 1;
  for (int i = n + tid; i < npad; i += kSortThreads) keys[i] = 0ull;
  __syncthreads();
  // ---- bitonic sort, descending.  Up to 2048 keys (the usual case since r4's refinement: ~pre_max candidates) stay in REGISTERS, one or two
  //      per thread: exchange distances inside a wave are lane shuffles -- no LDS write -> read round trip per pass, 45 of the 55 passes of a
  //      1024-key sort --, only distances >= 64 E go through LDS.  Same network, same result.
  auto reg_sort = [&](auto e_tag) {
    constexpr int E = decltype(e_tag)::value;
    const bool active = E * tid < npad;
    unsigned long long e[E];
#pragma unroll
    for (int r = 0; r < E; ++r) e[r] = active ? keys[E * tid + r] : 0ull;
    auto take = [](unsigned long long x, unsigned long long o, bool mx) { return mx ? (x > o ? x : o) : (x < o ? x : o); };
    for (int k = 2; k <= npad; k <<= 1) {
      for (int j = k >> 1; j >= 64 * E; j >>= 1) {
        __syncthreads();      // (the previous LDS stage's reads)
        if (active) {
#pragma unroll
          for (int r = 0; r < E; ++r) keys[E * tid + r] = e[r];
        }
        __syncthreads();
        if (active) {
#pragma unroll
          for (int r = 0; r < E; ++r) {
            const int i = E * tid + r;
            e[r] = take(e[r], keys[i ^ j], ((i & j) == 0) == ((i & k) == 0));
          }
        }
      }
      for (int j = min(k >> 1, 32 * E); j >= E; j >>= 1) {
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int i = E * tid + r;
          const unsigned lo32 = __shfl_xor((unsigned)e[r], j / E, 64), hi32 = __shfl_xor((unsigned)(e[r] >> 32), j / E, 64);
          e[r] = take(e[r], ((unsigned long long)hi32 << 32) | lo32, ((i & j) == 0) == ((i & k) == 0));
        }
      }
      if constexpr (E == 2) {
        const bool desc = ((2 * tid) & k) == 0;
        const unsigned long long hi = e[0] > e[1] ? e[0] : e[1], lo = e[0] > e[1] ? e[1] : e[0];
        e[0] = desc ? hi : lo;
        e[1] = desc ? lo : hi;
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int r = 0; r < E; ++r) keys[E * tid + r] = e[r];
    }
  };
  if (npad <= kSortThreads) reg_sort(std::integral_constant<int, 1>{});
  else if (npad <= 2 * kSortThreads) reg_sort(std::integral_constant<int, 2>{});
  else
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += kSortThreads) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = keys[i], c = keys[p];
          const bool desc = (i & k) == 0;
          if (desc ? a < c : a > c) { keys[i] = c; keys[p] = a; }
        }
      }
      // a pass with j < 64 stays inside 64-key chunks, and a chunk belongs to one wave (i = tid + 1024 * r): between two such
      // passes the wave's own in-order LDS traffic is all the ordering needed -- 51 of the 66 passes of a 2048-key sort
      const bool cur_local = j < 64, next_local = j > 1 ? (j >> 1) < 64 : k < 64;
      if (!(cur_local && next_local)) __syncthreads();
    }
  __syncthreads();
  SORT_STAMP(5);
  const int m = min(n, pre_max);
  if (tid == 0) n_sel[b] = m;
  for (int i = tid; i < m; i += kSortThreads) {
    const int cell = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
    sel_cell[(size_t)b * pre_max + i] = cell;
    sel_label[(size_t)b * pre_max + i] = label[(size_t)b * cells + cell];
    const float* src = boxes + ((size_t)b * cells + cell) * nb;
    float* d = nms_boxes + ((size_t)b * pre_max + i) * 7;
    // rotate_nms_pcdet's convention: [x, y, z, dims[1], dims[0], dims[2], -rot - pi/2]
    d[0] = src[0]; d[1] = src[1]; d[2] = src[2]; d[3] = src[4]; d[4] = src[3]; d[5] = src[5];
    d[6] = -src[nb - 1] - 1.57079632679489661923f;
  }
  SORT_STAMP(6);
  if (tid == 0) { SORT_STAMP(7); }
#ifdef PN_SORT_STAMP
  if (tid == 0 && pn_sort_stamps_dev) { pn_sort_stamps_dev[8] = (unsigned long long)n_valid; pn_sort_stamps_dev[9] = (unsigned long long)n_ge; pn_sort_stamps_dev[10] = (unsigned long long)n; pn_sort_stamps_dev[11] = (unsigned long long)npad; }
#endif
}

// cheap necessary condition for a non-empty overlap (the test at the top of iou_bev)
__device__ __forceinline__ bool may_overlap(const float* a, const float* b) {
  const float dx = a[0] - b[0], dy = a[1] - b[1];
  const float r = 0.5f * (sqrtf(a[3] * a[3] + a[4] * a[4]) + sqrtf(b[3] * b[3] + b[4] * b[4])) + 0.05f;
  return dx * dx + dy * dy <= r * r;
}

// grid (col blocks, row blocks, B), 64 threads: bit j of mask[row][col block] <=> IoU(row, 64*col block + j) > thresh, j after row.
// Two phases so that the expensive polygon intersection is not paid by a whole wave whenever ONE of its rows has a near
// neighbour: (1) every row marks its candidate columns with the cheap distance test, (2) the tile's candidate pairs are
// laid out as one list in LDS and dealt evenly over the 64 lanes; hits are OR-ed into the row masks (integer atomics).
// per_class: boxes of different classes never suppress each other (detectron2's batched_nms_rotated moves every class to its
// own region of the plane before one NMS; the same decision without the coordinate offsets).
constexpr int kMaskThreads = 256;
__global__ __launch_bounds__(kMaskThreads) void nms_mask_kernel(const float* __restrict__ nms_boxes, const int* __restrict__ n_sel, int pre_max, int col_blocks,
                                                                float thresh, unsigned long long* __restrict__ mask, const int* __restrict__ sel_label,
                                                                int per_class) {
  // r4: four waves per tile.  Phase 1 (row = lane of wave 0 .. 3: each wave tests a quarter of the columns) and the pair list are cheap; the
  // polygon intersections of phase 2 -- up to 4096 candidate pairs on a tile of crowded boxes, ~2 k clocks each and divergent -- are dealt over
  // 256 lanes instead of 64 (one wave per tile left 136 waves on the chip for 63 us at 1000 boxes).
  __shared__ float cb[64 * 7], rbx[64 * 7];
  __shared__ int clab[64];
  __shared__ unsigned long long cand[64], hit[64];
  __shared__ unsigned short pairs[64 * 64];
  __shared__ int total_s;
  const int b = blockIdx.z, rb = blockIdx.y, cbk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = n_sel[b];
  if (cbk < rb || rb * 64 >= n || cbk * 64 >= n) return;
  const float* bx = nms_boxes + (size_t)b * pre_max * 7;
  const int col_size = min(64, n - cbk * 64), row_size = min(64, n - rb * 64);
  for (int i = tid; i < col_size * 7; i += kMaskThreads) cb[i] = bx[(size_t)cbk * 64 * 7 + i];
  for (int i = tid; i < row_size * 7; i += kMaskThreads) rbx[i] = bx[(size_t)rb * 64 * 7 + i];
  if (tid < 64) {
    clab[tid] = (per_class && tid < col_size) ? sel_label[(size_t)b * pre_max + cbk * 64 + tid] : 0;
    hit[tid] = 0ull;
    cand[tid] = 0ull;
  }
  const int my_label = (per_class && lane < row_size) ? sel_label[(size_t)b * pre_max + rb * 64 + lane] : 0;
  __syncthreads();
  {
    unsigned long long c = 0ull;
    if (lane < row_size) {
      const int i0 = max(16 * wv, rb == cbk ? lane + 1 : 0), i1 = min(16 * wv + 16, col_size);
      for (int i = i0; i < i1; ++i)
        if (clab[i] == my_label && may_overlap(rbx + lane * 7, cb + i * 7)) c |= 1ull << i;
    }
    if (c) atomicOr(&cand[lane], c);
  }
  __syncthreads();
  if (wv == 0) {
    // exclusive prefix of the per-row candidate counts -> each row writes its (row, col) pairs into the list
    unsigned long long c = cand[lane];
    int cnt = __popcll(c), inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) total_s = inc;
    int pos = inc - cnt;
    while (c) {
      const int i = __ffsll((long long)c) - 1;
      c &= c - 1;
      pairs[pos++] = (unsigned short)(lane << 6 | i);
    }
  }
  __syncthreads();
  const int total = total_s;
  for (int p = tid; p < total; p += kMaskThreads) {
    const int r = pairs[p] >> 6, i = pairs[p] & 63;
    if (iou_bev(rbx + r * 7, cb + i * 7) > thresh) atomicOr(&hit[r], 1ull << i);
  }
  __syncthreads();
  if (tid < row_size) mask[((size_t)b * pre_max + rb * 64 + tid) * col_blocks + cbk] = hit[tid];
}

// grid B, one wave: lane l owns word l of the removed set (pre_max <= 4096)
__global__ __launch_bounds__(64) void nms_reduce_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ n_sel, int pre_max,
                                                        int col_blocks, int post_max, int* __restrict__ keep, int* __restrict__ n_keep) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = n_sel[b];
  const unsigned long long* mk = mask + (size_t)b * pre_max * col_blocks;
  unsigned long long removed = 0ull;
  int m = 0;
  unsigned long long next = (n > 0 && lane < col_blocks && lane >= 0) ? mk[lane] : 0ull;  // row 0
  for (int i = 0; i < n && m < post_max; ++i) {
    const unsigned long long row = next;
    if (i + 1 < n) next = (lane < col_blocks && lane >= (i + 1) / 64) ? mk[(size_t)(i + 1) * col_blocks + lane] : 0ull;  // prefetch
    const unsigned long long word = __shfl(removed, i >> 6, 64);
    if (!((word >> (i & 63)) & 1ull)) {
      if (lane == 0) keep[(size_t)b * post_max + m] = i;
      ++m;
      if (lane >= (i >> 6)) removed |= row;  // words before the row's own block were never written (upper triangle only)
    }
  }
  if (lane == 0) n_keep[b] = m;
}

// r4: the same greedy walk with the sample's mask rows staged in LDS first (pre_max * col_blocks * 8 bytes <= 128 KB: 1000 boxes) and the
// suppressed boxes skipped by bit scans: one LDS row read per KEPT box (<= post_max of them) instead of one global row per candidate -- the
// walk over 1000 candidates of crowded boxes took 26 us, a quarter of the post-processing.  Identical keep list.
constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void nms_reduce_lds_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ n_sel, int pre_max,
                                                                        int col_blocks, int post_max, int* __restrict__ keep, int* __restrict__ n_keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long rows[];      // [n][col_blocks]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n = n_sel[b];
  const unsigned long long* mk = mask + (size_t)b * pre_max * col_blocks;
  // (words left of a row's own block were never written by nms_mask_kernel: they are staged as they are and masked out below)
  // (every load of a thread in flight before the first lands: one at a time the staging alone took 30 us)
  {
    const int total = n * col_blocks;
    constexpr int UN = 8;
    for (int i0 = tid; i0 < total; i0 += kReduceThreads * UN) {
      unsigned long long v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = i0 + u * kReduceThreads < total ? mk[i0 + u * kReduceThreads] : 0ull;
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (i0 + u * kReduceThreads < total) rows[i0 + u * kReduceThreads] = v[u];
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  unsigned long long removed = 0ull;
  int m = 0;
  const int nblk = (n + 63) >> 6;
  for (int blk = 0; blk < nblk && m < post_max; ++blk) {
    const int left = n - 64 * blk;
    const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    unsigned long long done = 0ull;      // bits of this block already decided
    while (m < post_max) {
      const unsigned long long alive = ~__shfl(removed, blk, 64) & valid & ~done;
      if (!alive) break;
      const int bit = __ffsll((long long)alive) - 1, i = 64 * blk + bit;
      if (lane == 0) keep[(size_t)b * post_max + m] = i;
      ++m;
      if (lane >= blk && lane < col_blocks) removed |= rows[(size_t)i * col_blocks + lane];
      done |= bit == 63 ? ~0ull : ((2ull << bit) - 1ull);
    }
  }
  if (lane == 0) n_keep[b] = m;
}

__global__ void gather_kernel(const int* __restrict__ keep, const int* __restrict__ n_keep, const int* __restrict__ sel_cell,
                              const float* __restrict__ boxes, const float* __restrict__ score, const int* __restrict__ label, int cells, int nb,
                              int pre_max, int post_max, float* __restrict__ out_boxes, float* __restrict__ out_scores,
                              int64_t* __restrict__ out_labels, int* __restrict__ out_cells) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_keep[b]) return;
  const int cell = sel_cell[(size_t)b * pre_max + keep[(size_t)b * post_max + i]];
  const size_t src = (size_t)b * cells + cell, dst = (size_t)b * post_max + i;
  for (int k = 0; k < nb; ++k) out_boxes[dst * nb + k] = boxes[src * nb + k];
  out_scores[dst] = score[src];
  out_labels[dst] = label[src];
  out_cells[dst] = cell;
}

struct Ws {
  float* boxes; float* score; int* label; int* sel_cell; int* sel_label; float* nms_boxes; int* n_sel; unsigned long long* mask; int* keep;
  size_t bytes;
};

Ws carve(void* base, int batch, int cells, int nb, int pre_max, int post_max) {
  Ws w;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t n) { char* r = p; p += (n + 255) / 256 * 256; return r; };
  const int cbk = (pre_max + 63) / 64;
  w.boxes = (float*)take((size_t)batch * cells * nb * 4);
  w.score = (float*)take((size_t)batch * cells * 4);
  w.label = (int*)take((size_t)batch * cells * 4);
  w.sel_cell = (int*)take((size_t)batch * pre_max * 4);
  w.sel_label = (int*)take((size_t)batch * pre_max * 4);
  w.nms_boxes = (float*)take((size_t)batch * pre_max * 7 * 4);
  w.n_sel = (int*)take((size_t)batch * 4);
  w.mask = (unsigned long long*)take((size_t)batch * pre_max * cbk * 8);
  w.keep = (int*)take((size_t)batch * post_max * 4);
  w.bytes = (size_t)(p - static_cast<char*>(base));
  return w;
}

// decode -> select / sort -> IoU masks -> greedy reduce -> gather, shared by the CenterHead and the E2ESWVoteHead entry points
struct PrevDets { const float* boxes; const float* scores; const int64_t* labels; const int32_t* count; };

int run_decode_nms(DecodeArgs a, const float* post_center_range, float nms_iou_threshold, int per_class_nms, int pre_max, int post_max,
                   float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_cells, int32_t* out_count, void* workspace,
                   size_t workspace_bytes, pn_stream_t stream, PrevDets prev = PrevDets{nullptr, nullptr, nullptr, nullptr}) {
  const int nb = a.nb, grid_cells = a.H * a.W, cells = grid_cells + a.extra, batch = a.B;
  Ws ws = carve(workspace, batch, cells, nb, pre_max, post_max);
  PN_REQUIRE(workspace_bytes >= ws.bytes, "decode_nms: workspace too small");
  hipStream_t st = pn::S(stream);
  for (int k = 0; k < 3; ++k) { a.lo[k] = post_center_range[k]; a.hi[k] = post_center_range[3 + k]; }
  a.boxes = ws.boxes; a.score = ws.score; a.label = ws.label;
  const size_t total = (size_t)batch * grid_cells;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)std::min<size_t>(2048, (total + 255) / 256)), dim3(256), 0, st, a);
  if (a.extra > 0)
    hipLaunchKernelGGL(append_prev_kernel, dim3(pn::cdiv(batch * a.extra, 256)), dim3(256), 0, st, prev.boxes, prev.scores, prev.labels, prev.count, batch,
                       grid_cells, a.extra, nb, ws.boxes, ws.score, ws.label);
  static bool sort_attr[64] = {false};
  if (pn::first_use_on_device(sort_attr)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&select_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSortLds);
  }
  hipLaunchKernelGGL(select_sort_kernel, dim3(batch), dim3(kSortThreads), kSortLds, st, ws.score, ws.boxes, cells, nb, pre_max, ws.sel_cell,
                     ws.nms_boxes, ws.n_sel, ws.label, ws.sel_label);
  const int cbk = (pre_max + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cbk, cbk, batch), dim3(kMaskThreads), 0, st, ws.nms_boxes, ws.n_sel, pre_max, cbk, nms_iou_threshold, ws.mask,
                     ws.sel_label, per_class_nms != 0);
  const size_t rows_bytes = (size_t)pre_max * cbk * sizeof(unsigned long long);
  if (rows_bytes <= 128 * 1024) {
    static bool red_attr[64] = {false};
    if (pn::first_use_on_device(red_attr))
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_reduce_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(nms_reduce_lds_kernel, dim3(batch), dim3(kReduceThreads), rows_bytes, st, ws.mask, ws.n_sel, pre_max, cbk, post_max, ws.keep, out_count);
  } else {
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(batch), dim3(64), 0, st, ws.mask, ws.n_sel, pre_max, cbk, post_max, ws.keep, out_count);
  }
  hipLaunchKernelGGL(gather_kernel, dim3((post_max + 127) / 128, batch), dim3(128), 0, st, ws.keep, out_count, ws.sel_cell, ws.boxes, ws.score,
                     ws.label, cells, nb, pre_max, post_max, out_boxes, out_scores, out_labels, out_cells);
  return pn::check_launch("decode_nms");
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// boxes of a sector back into the sweep's frame (center_head.py:533-545): [x y] @ [[c, s], [-s, c]] with c = cos(angle), s = sin(angle)
// rounded to fp32 as torch.tensor(..., dtype=float) does, heading -= angle, velocity rotated like the centre
__global__ void rotate_boxes_kernel(float* __restrict__ boxes, const int32_t* __restrict__ counts, int cap, int nb, float c, float s, float angle, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * cap) return;
  const int b = i / cap, k = i - b * cap;
  if (k >= counts[b]) return;
  float* bx = boxes + (size_t)i * nb;
  const float x = bx[0], y = bx[1];
  bx[0] = __fadd_rn(__fmul_rn(x, c), __fmul_rn(y, -s));
  bx[1] = __fadd_rn(__fmul_rn(x, s), __fmul_rn(y, c));
  bx[nb - 1] = bx[nb - 1] - angle;
  if (nb > 7) {
    const float vx = bx[6], vy = bx[7];
    bx[6] = __fadd_rn(__fmul_rn(vx, c), __fmul_rn(vy, -s));
    bx[7] = __fadd_rn(__fmul_rn(vx, s), __fmul_rn(vy, c));
  }
}
}  // namespace

extern "C" {

int pn_rotate_boxes_f32(float* boxes, const int32_t* counts, int batch, int capacity, int box_dims, double angle, pn_stream_t stream) {
  PN_REQUIRE(boxes && counts && batch >= 1 && capacity >= 1 && box_dims >= 7, "rotate_boxes: bad arguments");
  hipLaunchKernelGGL(rotate_boxes_kernel, dim3(pn::cdiv(batch * capacity, 256)), dim3(256), 0, pn::S(stream), boxes, counts, capacity, box_dims,
                     (float)cos(angle), (float)sin(angle), (float)angle, batch);
  return pn::check_launch("rotate_boxes_kernel");
}

size_t pn_center_decode_nms_workspace_bytes(int batch, int cells, int box_dims, int pre_max, int post_max) {
  return carve(nullptr, batch, cells, box_dims, pre_max, post_max).bytes;
}

int pn_center_decode_nms_f32(const float* hm, int hm_pixel_stride, int classes, const float* reg, int reg_pixel_stride, const float* height,
                             int height_pixel_stride, const float* dim, int dim_pixel_stride, const float* rot, int rot_pixel_stride,
                             const float* vel, int vel_pixel_stride, int batch, int h, int w, int cylinder, float step_x, float step_y,
                             float origin_x, float origin_y, int rectify, float score_threshold, const float* post_center_range,
                             float nms_iou_threshold, int per_class_nms, int pre_max, int post_max, float* out_boxes, float* out_scores,
                             int64_t* out_labels, int32_t* out_cells, int32_t* out_count, void* workspace, size_t workspace_bytes,
                             pn_stream_t stream) {
  PN_REQUIRE(hm && reg && height && dim && rot && post_center_range && out_boxes && out_scores && out_labels && out_cells && out_count && workspace,
             "center_decode_nms: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && classes >= 1, "center_decode_nms: bad sizes");
  PN_REQUIRE(pre_max >= 1 && pre_max <= 4096 && post_max >= 1, "center_decode_nms: nms_pre_max_size must be in [1, 4096]");
  DecodeArgs a{};
  a.hm = hm; a.hm_ps = hm_pixel_stride; a.ncls = classes; a.reg = reg; a.reg_ps = reg_pixel_stride; a.hei = height; a.hei_ps = height_pixel_stride;
  a.dim = dim; a.dim_ps = dim_pixel_stride; a.rot = rot; a.rot_ps = rot_pixel_stride; a.vel = vel; a.vel_ps = vel_pixel_stride;
  a.B = batch; a.H = h; a.W = w; a.cylinder = cylinder; a.rectify = rectify; a.nb = vel ? 9 : 7;
  a.sx = step_x; a.sy = step_y; a.x0 = origin_x; a.y0 = origin_y; a.thr = score_threshold;
  return run_decode_nms(a, post_center_range, nms_iou_threshold, per_class_nms, pre_max, post_max, out_boxes, out_scores, out_labels, out_cells,
                        out_count, workspace, workspace_bytes, stream);
}

int pn_center_decode_nms_merged_f32(const float* hm_prob, int hm_pixel_stride, int classes, const float* reg, int reg_pixel_stride,
                                    const float* height, int height_pixel_stride, const float* dim_size, int dim_pixel_stride, const float* rot,
                                    int rot_pixel_stride, const float* vel, int vel_pixel_stride, int batch, int h, int w, int cylinder,
                                    float step_x, float step_y, float origin_x, float origin_y, int rectify, float score_threshold,
                                    const float* post_center_range, float nms_iou_threshold, int per_class_nms, int pre_max, int post_max,
                                    float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_cells, int32_t* out_count,
                                    void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(hm_prob && reg && height && dim_size && rot && post_center_range && out_boxes && out_scores && out_labels && out_cells && out_count &&
                 workspace, "center_decode_nms_merged: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && classes >= 1, "center_decode_nms_merged: bad sizes");
  PN_REQUIRE(pre_max >= 1 && pre_max <= 4096 && post_max >= 1, "center_decode_nms_merged: nms_pre_max_size must be in [1, 4096]");
  DecodeArgs a{};
  a.hm = hm_prob; a.hm_ps = hm_pixel_stride; a.ncls = classes; a.reg = reg; a.reg_ps = reg_pixel_stride; a.hei = height; a.hei_ps = height_pixel_stride;
  a.dim = dim_size; a.dim_ps = dim_pixel_stride; a.rot = rot; a.rot_ps = rot_pixel_stride; a.vel = vel; a.vel_ps = vel_pixel_stride;
  a.B = batch; a.H = h; a.W = w; a.cylinder = cylinder; a.rectify = rectify; a.nb = vel ? 9 : 7;
  a.sx = step_x; a.sy = step_y; a.x0 = origin_x; a.y0 = origin_y; a.thr = score_threshold;
  a.activated = 1;
  return run_decode_nms(a, post_center_range, nms_iou_threshold, per_class_nms, pre_max, post_max, out_boxes, out_scores, out_labels, out_cells,
                        out_count, workspace, workspace_bytes, stream);
}

int pn_center_decode_nms_stateful_f32(const float* hm, int hm_pixel_stride, int classes, const float* reg, int reg_pixel_stride, const float* height,
                                      int height_pixel_stride, const float* dim, int dim_pixel_stride, const float* rot, int rot_pixel_stride,
                                      const float* vel, int vel_pixel_stride, int batch, int h, int w, int cylinder, float step_x, float step_y,
                                      float origin_x, float origin_y, int rectify, float score_threshold, const float* post_center_range,
                                      float nms_iou_threshold, int per_class_nms, int pre_max, int post_max, double sector_angle,
                                      const float* prev_boxes, const float* prev_scores, const int64_t* prev_labels, const int32_t* prev_count,
                                      int prev_capacity, float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_cells,
                                      int32_t* out_count, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(hm && reg && height && dim && rot && post_center_range && out_boxes && out_scores && out_labels && out_cells && out_count && workspace,
             "center_decode_nms_stateful: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && classes >= 1 && prev_capacity >= 0, "center_decode_nms_stateful: bad sizes");
  PN_REQUIRE(prev_capacity == 0 || (prev_boxes && prev_scores && prev_labels && prev_count), "center_decode_nms_stateful: previous detections missing");
  PN_REQUIRE(pre_max >= 1 && pre_max <= 4096 && post_max >= 1, "center_decode_nms_stateful: nms_pre_max_size must be in [1, 4096]");
  DecodeArgs a{};
  a.hm = hm; a.hm_ps = hm_pixel_stride; a.ncls = classes; a.reg = reg; a.reg_ps = reg_pixel_stride; a.hei = height; a.hei_ps = height_pixel_stride;
  a.dim = dim; a.dim_ps = dim_pixel_stride; a.rot = rot; a.rot_ps = rot_pixel_stride; a.vel = vel; a.vel_ps = vel_pixel_stride;
  a.B = batch; a.H = h; a.W = w; a.cylinder = cylinder; a.rectify = rectify; a.nb = vel ? 9 : 7;
  a.sx = step_x; a.sy = step_y; a.x0 = origin_x; a.y0 = origin_y; a.thr = score_threshold;
  a.extra = prev_capacity;
  a.pre_rot = sector_angle != 0.0;
  a.rot_c = (float)cos(sector_angle); a.rot_s = (float)sin(sector_angle); a.rot_angle = (float)sector_angle;
  return run_decode_nms(a, post_center_range, nms_iou_threshold, per_class_nms, pre_max, post_max, out_boxes, out_scores, out_labels, out_cells,
                        out_count, workspace, workspace_bytes, stream, PrevDets{prev_boxes, prev_scores, prev_labels, prev_count});
}

int pn_double_flip_merge_f32(const float* hm, int hm_pixel_stride, int classes, const float* reg, int reg_pixel_stride, const float* height,
                             int height_pixel_stride, const float* dim, int dim_pixel_stride, const float* rot, int rot_pixel_stride,
                             const float* vel, int vel_pixel_stride, int merged_batch, int h, int w, float* out_hm, float* out_reg,
                             float* out_height, float* out_dim, float* out_rot, float* out_vel, pn_stream_t stream) {
  PN_REQUIRE(hm && reg && height && dim && rot && out_hm && out_reg && out_height && out_dim && out_rot && (!vel || out_vel),
             "double_flip_merge: null pointer");
  PN_REQUIRE(merged_batch >= 1 && h >= 1 && w >= 1 && classes >= 1, "double_flip_merge: bad sizes");
  FlipArgs a{};
  a.hm = hm; a.hm_ps = hm_pixel_stride; a.ncls = classes; a.reg = reg; a.reg_ps = reg_pixel_stride; a.hei = height; a.hei_ps = height_pixel_stride;
  a.dim = dim; a.dim_ps = dim_pixel_stride; a.rot = rot; a.rot_ps = rot_pixel_stride; a.vel = vel; a.vel_ps = vel_pixel_stride;
  a.B = merged_batch; a.H = h; a.W = w;
  a.o_hm = out_hm; a.o_reg = out_reg; a.o_hei = out_height; a.o_dim = out_dim; a.o_rot = out_rot; a.o_vel = out_vel;
  const size_t total = (size_t)merged_batch * h * w;
  hipLaunchKernelGGL(double_flip_merge_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, pn::S(stream), a);
  return pn::check_launch("double_flip_merge_kernel");
}

int pn_swv_decode_nms_f32(const float* hm, int hm_pixel_stride, int classes, const float* reg, int reg_pixel_stride, const float* height,
                          int height_pixel_stride, const float* dim, int dim_pixel_stride, const float* rot, int rot_pixel_stride,
                          const float* iou, int iou_pixel_stride, int iou_factor, const float* offset_grid, int batch, int h, int w,
                          int rectify, float score_threshold, const float* post_center_range, float nms_iou_threshold, int per_class_nms,
                          int pre_max, int post_max, float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_cells,
                          int32_t* out_count, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(hm && reg && height && dim && rot && offset_grid && post_center_range && out_boxes && out_scores && out_labels && out_cells &&
                 out_count && workspace, "swv_decode_nms: null pointer");
  PN_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && classes >= 1 && iou_factor >= 0, "swv_decode_nms: bad sizes");
  PN_REQUIRE(pre_max >= 1 && pre_max <= 4096 && post_max >= 1, "swv_decode_nms: nms_pre_max_size must be in [1, 4096]");
  DecodeArgs a{};
  a.hm = hm; a.hm_ps = hm_pixel_stride; a.ncls = classes; a.reg = reg; a.reg_ps = reg_pixel_stride; a.hei = height; a.hei_ps = height_pixel_stride;
  a.dim = dim; a.dim_ps = dim_pixel_stride; a.rot = rot; a.rot_ps = rot_pixel_stride; a.vel = nullptr; a.vel_ps = 0;
  a.B = batch; a.H = h; a.W = w; a.cylinder = 1; a.rectify = rectify; a.nb = 7; a.thr = score_threshold;
  a.swv = 1; a.iou = iou; a.iou_ps = iou_pixel_stride; a.iou_factor = iou_factor; a.grid = offset_grid;
  return run_decode_nms(a, post_center_range, nms_iou_threshold, per_class_nms, pre_max, post_max, out_boxes, out_scores, out_labels, out_cells,
                        out_count, workspace, workspace_bytes, stream);
}

}  // extern "C"
