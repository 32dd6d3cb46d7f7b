// CenterPoint target assignment on the polar grid, on the device: ground-truth boxes -> heat map,
// ind / mask / cat / anno_box of the training step, so that the loss kernels are fed without a CPU dataloader pass.
// Reference: AssignLabel.assign_heatmap_polar     det3d/datasets/pipelines/preprocess.py:253-342
//            gaussian_radius, draw_umich_gaussian  det3d/core/utils/center_utils.py:18-64
//            center_to_corner_box2d               det3d/core/bbox/box_np_ops.py:265-285, 55-85, 207-220
// The arithmetic keeps the reference's dtypes (float32 boxes / voxel_size / pc_range; the real-world cell centre in
// float64) so that the integer outputs agree; float32 divisions and square roots are done in double and rounded once
// (exact for float32 operands).  Two launches: one thread per object (geometry, ind/mask/cat/anno_box, draw record),
// then one block per object splatting its Gaussian with an integer atomicMax on the float bits (all values are >= 0,
// the maximum is order independent => deterministic).
#include "pn_common.h"

namespace {

struct AssignArgs {
  const float* boxes; const int32_t* classes; const int32_t* num_gt;
  int B, max_gt, cols, max_objs, ncls, R, A, osf, min_radius, rectify;
  float vs0, vs1, pc0, pc1, overlap;
  float* hm; int64_t* ind; uint8_t* mask; int64_t* cat; float* anno;
  int4* draw;  // (B, max_objs): x, y, radius, class (-1: nothing to draw)
};

__device__ __forceinline__ float fdiv(float a, float b) { return (float)((double)a / (double)b); }
__device__ __forceinline__ float fsqrt(float a) { return (float)sqrt((double)a); }

__device__ float gaussian_radius_f32(float h, float w, float mo) {
  const float b1 = h + w;
  const float c1 = fdiv(w * h * (float)(1.0 - (double)mo), (float)(1.0 + (double)mo));
  const float r1 = fdiv(b1 + fsqrt(b1 * b1 - 4.f * c1), 2.f);
  const float b2 = 2.f * (h + w);
  const float c2 = (float)(1.0 - (double)mo) * w * h;
  const float r2 = fdiv(b2 + fsqrt(b2 * b2 - 16.f * c2), 2.f);
  const float a3 = (float)(4.0 * (double)mo);
  const float b3 = (float)(-2.0 * (double)mo) * (h + w);
  const float c3 = (float)((double)mo - 1.0) * w * h;
  const float r3 = fdiv(b3 + fsqrt(b3 * b3 - 4.f * a3 * c3), 2.f);
  return fminf(r1, fminf(r2, r3));
}

__global__ void assign_objects_kernel(AssignArgs a) {
  const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.max_objs) return;
  int4 rec = make_int4(0, 0, 0, -1);
  const int n = min(min(a.num_gt[b], a.max_gt), a.max_objs);
  if (k < n) {
    const float* bx = a.boxes + ((size_t)b * a.max_gt + k) * a.cols;
    // footprint corners rotated by COLUMN 6 of the box, as the reference does (preprocess.py:266)
    const float s = sinf(bx[6]), c = cosf(bx[6]);
    const float ux[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, uy[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    float rmin = 3.0e38f, rmax = -3.0e38f, amin = 3.0e38f, amax = -3.0e38f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float lx = bx[3] * ux[q], ly = bx[4] * uy[q];
      const float cx = lx * c + ly * s + bx[0], cy = lx * (-s) + ly * c + bx[1];
      const float rho = fsqrt(cx * cx + cy * cy), az = atan2f(cy, cx);
      rmin = fminf(rmin, rho); rmax = fmaxf(rmax, rho); amin = fminf(amin, az); amax = fmaxf(amax, az);
    }
    const float dr = fdiv(fdiv(rmax - rmin, a.vs0), (float)a.osf), da = fdiv(fdiv(amax - amin, a.vs1), (float)a.osf);
    if (dr > 0.f && da > 0.f) {
      const float r = fsqrt(bx[0] * bx[0] + bx[1] * bx[1]), az = atan2f(bx[1], bx[0]);
      const int radius = max(a.min_radius, (int)gaussian_radius_f32(dr, da, a.overlap) - (r > 30.f ? 1 : 0));
      const float ctr = fdiv(fdiv(r - a.pc0, a.vs0), (float)a.osf), cta = fdiv(fdiv(az - a.pc1, a.vs1), (float)a.osf);
      const int ir = (int)ctr;
      const int ia = min(max((int)cta, 0), a.A - 1);
      if (ir >= 0 && ir < a.R) {
        const int cls = a.classes[(size_t)b * a.max_gt + k] - 1;
        rec = make_int4((int)ctr, (int)cta, radius, cls);     // the splat is centred on the UNCLIPPED cell
        const double r_real = (double)ir * a.osf * (double)a.vs0 + (double)a.pc0;
        const double a_real = (double)ia * a.osf * (double)a.vs1 + (double)a.pc1;
        const double xc = r_real * cos(a_real), yc = r_real * sin(a_real);
        float vx = bx[6], vy = bx[7], rot = bx[a.cols - 1];
        if (a.rectify) {
          rot = rot - az;
          const float vr = fsqrt(vx * vx + vy * vy), va = atan2f(vy, vx) - az;
          vx = vr * cosf(va); vy = vr * sinf(va);
        }
        const size_t o = (size_t)b * a.max_objs + k;
        a.cat[o] = cls; a.ind[o] = (int64_t)ia * a.R + ir; a.mask[o] = 1;
        float* an = a.anno + o * 10;
        an[0] = (float)((double)bx[0] - xc); an[1] = (float)((double)bx[1] - yc); an[2] = bx[2];
        an[3] = logf(bx[3]); an[4] = logf(bx[4]); an[5] = logf(bx[5]); an[6] = vx; an[7] = vy; an[8] = sinf(rot); an[9] = cosf(rot);
      }
    }
  }
  a.draw[(size_t)b * a.max_objs + k] = rec;
}

// grid (max_objs, B): element-wise maximum of the heat map with the object's Gaussian window (center_utils.py:46-64)
__global__ void draw_gaussian_kernel(AssignArgs a) {
  const int b = blockIdx.y;
  const int4 rec = a.draw[(size_t)b * a.max_objs + blockIdx.x];
  if (rec.w < 0 || rec.w >= a.ncls) return;
  const int x = rec.x, y = rec.y, radius = rec.z;
  const int left = min(x, radius), right = min(a.R - x, radius + 1), top = min(y, radius), bottom = min(a.A - y, radius + 1);
  const int wdt = right + left, hgt = bottom + top;
  if (wdt <= 0 || hgt <= 0) return;
  const double sigma = (double)(2 * radius + 1) / 6.0, inv = 1.0 / (2.0 * sigma * sigma);
  unsigned* plane = reinterpret_cast<unsigned*>(a.hm + ((size_t)b * a.ncls + rec.w) * a.A * a.R);
  for (int i = threadIdx.x; i < wdt * hgt; i += blockDim.x) {
    const int dy = i / wdt - top, dx = i % wdt - left;
    const int yy = y + dy, xx = x + dx;
    if ((unsigned)yy >= (unsigned)a.A || (unsigned)xx >= (unsigned)a.R) continue;
    const float g = (float)exp(-(double)(dx * dx + dy * dy) * inv);
    atomicMax(plane + (size_t)yy * a.R + xx, __builtin_bit_cast(unsigned, g));
  }
}

}  // namespace

extern "C" {

size_t pn_assign_heatmap_workspace_bytes(int batch, int max_objs) { return (size_t)batch * max_objs * sizeof(int4); }

int pn_assign_heatmap_polar_f32(const float* gt_boxes, const int32_t* gt_classes, const int32_t* num_gt, int batch, int max_gt, int box_cols,
                                int max_objs, int classes, int feature_r, int feature_a, float voxel_size_r, float voxel_size_a, float range_r0,
                                float range_a0, int out_size_factor, float gaussian_overlap, int min_radius, int rectify, float* hm, int64_t* ind,
                                uint8_t* mask, int64_t* cat, float* anno_box, void* workspace, size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(gt_boxes && gt_classes && num_gt && hm && ind && mask && cat && anno_box && workspace, "assign_heatmap: null pointer");
  PN_REQUIRE(batch >= 1 && max_gt >= 1 && box_cols >= 9 && max_objs >= 1 && classes >= 1 && feature_r >= 1 && feature_a >= 1 && out_size_factor >= 1,
             "assign_heatmap: bad sizes (boxes are [x,y,z,l,w,h,vx,vy,...,rot] with at least 9 columns)");
  PN_REQUIRE(workspace_bytes >= pn_assign_heatmap_workspace_bytes(batch, max_objs), "assign_heatmap: workspace too small");
  hipStream_t st = pn::S(stream);
  if (int rc = pn::zero_async(hm, (size_t)batch * classes * feature_a * feature_r * 4, st)) return rc;
  if (int rc = pn::zero_async(ind, (size_t)batch * max_objs * 8, st)) return rc;
  if (int rc = pn::zero_async(cat, (size_t)batch * max_objs * 8, st)) return rc;
  if (int rc = pn::zero_async(mask, (size_t)batch * max_objs, st)) return rc;
  if (int rc = pn::zero_async(anno_box, (size_t)batch * max_objs * 10 * 4, st)) return rc;
  AssignArgs a{gt_boxes, gt_classes, num_gt, batch, max_gt, box_cols, max_objs, classes, feature_r, feature_a, out_size_factor, min_radius, rectify,
               voxel_size_r, voxel_size_a, range_r0, range_a0, gaussian_overlap, hm, ind, mask, cat, anno_box, static_cast<int4*>(workspace)};
  hipLaunchKernelGGL(assign_objects_kernel, dim3(pn::cdiv(max_objs, 128), batch), dim3(128), 0, st, a);
  hipLaunchKernelGGL(draw_gaussian_kernel, dim3(max_objs, batch), dim3(256), 0, st, a);
  return pn::check_launch("assign_heatmap_polar");
}

}  // extern "C"

// =================================================================================================
// Multi-sweep accumulation on the device (the device half; BASELINE configs[4] "10-sweep
// accumulation"): raw sweeps (N, 5) f32 [x, y, z, intensity, ring] concatenated key frame first -> (N', 5)
// [x, y, z, intensity, time lag] in the key frame.
// Reference: read_file / remove_close / read_sweep   det3d/datasets/pipelines/loading.py:42-84
//            LoadPointCloudFromFile.get_points       loading.py:216-332 (concatenation order: key frame, then the sweeps)
// Points of the past sweeps with |x| < 1 and |y| < 1 (in their own frame) are dropped, the rest are moved with the
// sweep's 4x4 transform (float64 matrix, float64 product, rounded to float32 -- as numpy does) and tagged with the
// sweep's time lag.  Order-preserving compaction: flag -> block scan (fixed order) -> scatter; the count stays on
// the device.
// =================================================================================================
namespace {

constexpr int kAccT = 256, kAccItems = 8;

struct AccArgs {
  const float* raw; int n; int in_cols; const int32_t* offsets; int sweeps; const double* mats; const float* lags; float radius;
  float* out; int32_t* count; uint32_t* tile; int ntiles;
};

__device__ __forceinline__ int sweep_of(const AccArgs& a, int i) {
  int s = 0;
  while (s + 1 < a.sweeps && i >= a.offsets[s + 1]) ++s;   // <= 10 sweeps: linear search on a cached array
  return s;
}

__device__ __forceinline__ bool acc_keep(const AccArgs& a, int i, int s) {
  if (i >= a.offsets[a.sweeps]) return false;   // rows of a capacity-sized buffer past the last sweep
  if (s == 0) return true;   // the key frame is taken as it is (loading.py:228-232)
  const float* p = a.raw + (size_t)i * a.in_cols;
  return !(fabsf(p[0]) < a.radius && fabsf(p[1]) < a.radius);
}


__global__ void acc_count_kernel(AccArgs a) {
  const int base = (blockIdx.x * kAccT + threadIdx.x) * kAccItems;
  uint32_t c = 0;
  for (int k = 0; k < kAccItems; ++k) {
    const int i = base + k;
    if (i < a.n && acc_keep(a, i, sweep_of(a, i))) ++c;
  }
  uint32_t tot;
  pn::block_exclusive_scan<kAccT>(c, &tot);
  if (threadIdx.x == 0) a.tile[blockIdx.x] = tot;
}

__global__ void acc_offsets_kernel(AccArgs a) {
  uint32_t carry = 0;
  for (int b0 = 0; b0 < a.ntiles; b0 += kAccT) {
    const int i = b0 + threadIdx.x;
    const uint32_t v = i < a.ntiles ? a.tile[i] : 0;
    uint32_t tot;
    const uint32_t ex = pn::block_exclusive_scan<kAccT>(v, &tot);
    if (i < a.ntiles) a.tile[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *a.count = (int32_t)carry;
}

__global__ void acc_scatter_kernel(AccArgs a) {
  const int base = (blockIdx.x * kAccT + threadIdx.x) * kAccItems;
  bool keep[kAccItems];
  int sw[kAccItems];
  uint32_t c = 0;
  for (int k = 0; k < kAccItems; ++k) {
    const int i = base + k;
    sw[k] = i < a.n ? sweep_of(a, i) : 0;
    keep[k] = i < a.n && acc_keep(a, i, sw[k]);
    c += keep[k];
  }
  uint32_t tot;
  uint32_t pos = a.tile[blockIdx.x] + pn::block_exclusive_scan<kAccT>(c, &tot);
  for (int k = 0; k < kAccItems; ++k) {
    if (!keep[k]) continue;
    const float* p = a.raw + (size_t)(base + k) * a.in_cols;
    float* o = a.out + (size_t)pos * 5;
    if (sw[k] == 0) {
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    } else {
      const double* m = a.mats + (size_t)sw[k] * 16;
      const double x = p[0], y = p[1], z = p[2];
      o[0] = (float)(m[0] * x + m[1] * y + m[2] * z + m[3]);
      o[1] = (float)(m[4] * x + m[5] * y + m[6] * z + m[7]);
      o[2] = (float)(m[8] * x + m[9] * y + m[10] * z + m[11]);
    }
    o[3] = p[3];
    o[4] = a.lags[sw[k]];
    ++pos;
  }
}

}  // namespace

extern "C" {

size_t pn_accumulate_sweeps_workspace_bytes(int n) { return (size_t)pn::cdiv(n, kAccT * kAccItems) * sizeof(uint32_t) + 256; }

int pn_accumulate_sweeps_f32(const float* raw, int n, int in_cols, const int32_t* sweep_offsets, int sweeps, const double* transforms,
                             const float* time_lags, float min_distance, float* out, int32_t* out_count, void* workspace,
                             size_t workspace_bytes, pn_stream_t stream) {
  PN_REQUIRE(raw && sweep_offsets && transforms && time_lags && out && out_count && workspace, "accumulate_sweeps: null pointer");
  PN_REQUIRE(n >= 1 && in_cols >= 4 && sweeps >= 1, "accumulate_sweeps: bad sizes");
  PN_REQUIRE(workspace_bytes >= pn_accumulate_sweeps_workspace_bytes(n), "accumulate_sweeps: workspace too small");
  AccArgs a{raw, n, in_cols, sweep_offsets, sweeps, transforms, time_lags, min_distance, out, out_count, static_cast<uint32_t*>(workspace),
            pn::cdiv(n, kAccT * kAccItems)};
  hipStream_t st = pn::S(stream);
  hipLaunchKernelGGL(acc_count_kernel, dim3(a.ntiles), dim3(kAccT), 0, st, a);
  hipLaunchKernelGGL(acc_offsets_kernel, dim3(1), dim3(kAccT), 0, st, a);
  hipLaunchKernelGGL(acc_scatter_kernel, dim3(a.ntiles), dim3(kAccT), 0, st, a);
  return pn::check_launch("accumulate_sweeps");
}

}  // extern "C"
