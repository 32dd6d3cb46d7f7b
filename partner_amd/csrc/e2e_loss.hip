// Training side of the geometry-aware head: vote-map targets, matcher cost, set criterion.
// Reference (the restatement these kernels are checked against is oracle/e2e_loss_oracle.py, pinned to the reference's own
// GroundTruthProcessor / CenterCoder / TimeMatcher / SetCriterion outputs by tests/golden/e2e_loss.npz):
//   GroundTruthProcessor.process / draw_votemap      det3d/models/bbox_heads/e2e_modules.py:31-148
//   draw_center_to_votemap, gaussian_radius          det3d/models/e2e_utils/centernet_utils.py:5-32, 68-88
//   CenterCoder.encode / get_delta / decode_torch    det3d/models/e2e_utils/box_coder_utils.py:107-138, 172-244
//   TimeMatcher cost                                 det3d/models/e2e_utils/matcher.py:78-93, 136-147
//   SetCriterion                                     det3d/models/e2e_utils/set_crit.py:68-206
//   focal / smooth-L1 / IoU losses                   det3d/models/e2e_utils/loss_utils.py:447-535, 583-594
//   boxes_iou3d_gpu                                  det3d/ops/iou3d_nms/iou3d_nms_utils.py:38-72
// The assignment itself (a sequential shortest-augmenting-path algorithm; scipy's linear_sum_assignment on the HOST in the
// reference, matcher.py:149) is pn_lsap_f32 below: host code, as in the reference.
// Every reduction has a fixed association order (fp64 block partials, folded by one block): bit-reproducible losses / gradients.
#include "pn_common.h"
#include "box_geom.h"
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace {

constexpr int kT = 256;

// ------------------------------------------------------------------------------------------------ ground-truth split
// one wave per sample: drop the all-zero padding tail (e2e_modules.py:50-57), keep the rows of the task's classes in class order
__global__ __launch_bounds__(64) void gt_compact_kernel(const float* __restrict__ gbox, int max_boxes, int cols, const int32_t* __restrict__ class_ids,
                                                        int n_classes, float* __restrict__ out_boxes, int32_t* __restrict__ out_cls,
                                                        int32_t* __restrict__ out_cnt) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* rows = gbox + (size_t)b * max_boxes * cols;
  // count = index of the last row whose (box part) sum is non-zero, at least 0: rows [0, count] are looked at
  int last = 0;
  for (int i = lane; i < max_boxes; i += 64) {   // the box part after e2e_swv_head.py:207: columns [0 .. 5, cols - 2]
    float s = 0.f;
    for (int c = 0; c < 6; ++c) s += rows[(size_t)i * cols + c];
    s += rows[(size_t)i * cols + cols - 2];
    if (s != 0.f) last = max(last, i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
  const int n_rows = last + 1;
  int n_out = 0;
  for (int k = 0; k < n_classes; ++k) {
    const int want = class_ids[k];
    for (int base = 0; base < n_rows; base += 64) {
      const int i = base + lane;
      const bool hit = i < n_rows && (int)rows[(size_t)i * cols + cols - 1] == want;
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int dst = n_out + __popcll(m & ((1ull << lane) - 1ull));
        float* o = out_boxes + ((size_t)b * max_boxes + dst) * 7;
        for (int c = 0; c < 6; ++c) o[c] = rows[(size_t)i * cols + c];
        o[6] = rows[(size_t)i * cols + cols - 2];   // heading (velocity columns, if any, sit between the size and the heading)
        out_cls[(size_t)b * max_boxes + dst] = k;
      }
      n_out += __popcll(m);
    }
  }
  if (lane == 0) out_cnt[b] = n_out;
}

// ------------------------------------------------------------------------------------------------ vote map
struct VoteObj {
  int valid, rho_i, phi_i, r_rho, r_phi, cls;
  float cx, cy, crho, cphi;
  double inv2s2;   // 1 / (2 sigma^2), sigma = max(2 r_rho + 1, 2 r_phi + 1) / 6
};

struct VoteArgs {
  const float* boxes;      // (B, max_boxes, 7)
  const int32_t* cls;      // (B, max_boxes)
  const int32_t* cnt;      // (B)
  int B, max_boxes, n_classes, H, W, stride, num_max_objs;
  float vs_rho, vs_phi, min_rho, min_phi;
  float ov_minus, ov_plus, ov;   // (1 - overlap), (1 + overlap), overlap: formed in double as python does, then rounded to fp32
  VoteObj* objs;           // (B, max_boxes)
  float* votemap;          // (B, H, W, 4 + n_classes)
  unsigned long long* part;  // per block: cells with votemap[..., 0] != 0
  int32_t* vote_count;
};

__device__ __forceinline__ float gaussian_radius_f(float h, float w, float om, float op, float ov) {
  // centernet_utils.py:5-32 in fp32, every operation rounded as torch does on 0-dim fp32 tensors (python scalars enter as fp32)
  const float b1 = h + w;
  const float c1 = w * h * om / op;
  const float r1 = (b1 + sqrtf(b1 * b1 - 4 * c1)) / 2;
  const float b2 = 2 * (h + w);
  const float c2 = om * w * h;
  const float r2 = (b2 + sqrtf(b2 * b2 - 16 * c2)) / 2;
  const float a3x4 = (float)(4.0 * (4.0 * (double)ov));   // 4 * a3 with a3 = 4 * min_overlap formed in python floats
  const float b3 = (float)(-2.0 * (double)ov) * (h + w);
  const float c3 = -om * w * h;
  const float r3 = (b3 + sqrtf(b3 * b3 - a3x4 * c3)) / 2;
  return fminf(fminf(r1, r2), r3);
}

// one thread per (sample, object): the scalars of draw_votemap's loop body (e2e_modules.py:98-146)
__global__ void vote_prep_kernel(VoteArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * a.max_boxes) return;
  const int b = idx / a.max_boxes, k = idx - b * a.max_boxes;
  VoteObj o;
  o.valid = 0;
  o.rho_i = o.phi_i = o.r_rho = o.r_phi = o.cls = 0;
  o.cx = o.cy = o.crho = o.cphi = 0.f;
  o.inv2s2 = 0.0;
  if (k < min(a.cnt[b], a.num_max_objs)) {
    const float* bx = a.boxes + (size_t)idx * 7;
    const float x = bx[0], y = bx[1], dx = bx[3], dy = bx[4];
    const float s = sinf(bx[6]), c = cosf(bx[6]);
    // center_to_corner_box2d: corners = dims * ([0,0],[0,1],[1,1],[1,0] - 0.5), rotated by [[c, -s], [s, c]]^T, + centre
    const float nx[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, ny[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    float rmax = -1e30f, rmin = 1e30f, pmax = -1e30f, pmin = 1e30f, ph[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float px = dx * nx[j], py = dy * ny[j];
      const float qx = __fadd_rn(__fadd_rn(__fmul_rn(px, c), __fmul_rn(py, s)), x);       // einsum("aij,jka->aik"): x' = px*c + py*s
      const float qy = __fadd_rn(__fadd_rn(__fmul_rn(px, -s), __fmul_rn(py, c)), y);      //                         y' = -px*s + py*c
      const float rho = (float)sqrt((double)__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)));
      const float phi = (float)atan2((double)qy, (double)qx);
      ph[j] = phi;
      rmax = fmaxf(rmax, rho); rmin = fminf(rmin, rho); pmax = fmaxf(pmax, phi); pmin = fminf(pmin, phi);
    }
    const float fstride = (float)a.stride;
    float drho = (rmax - rmin) / a.vs_rho / fstride;
    float dphi = (pmax - pmin) / a.vs_phi / fstride;
    const float crho = (float)sqrt((double)__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
    const float cphi = (float)atan2((double)y, (double)x);
    const int rho_i = (int)((crho - a.min_rho) / a.vs_rho / fstride);      // .int(): truncation
    const int phi_i = (int)((cphi - a.min_phi) / a.vs_phi / fstride);
    bool ok = drho > 0.f && dphi > 0.f && rho_i >= 0 && rho_i < a.W && phi_i >= 0 && phi_i < a.H;
    if (ok && dphi > (float)a.H / 4.f) {   // the box straddles the +-pi seam (e2e_modules.py:136-143)
      float trunc;
      if (cphi > 0.f) {
        float m = 1e30f;
        for (int j = 0; j < 4; ++j)
          if (ph[j] > 0.f) m = fminf(m, ph[j]);
        trunc = (float)M_PI - m;
      } else {
        float m = -1e30f;
        for (int j = 0; j < 4; ++j)
          if (ph[j] <= 0.f) m = fmaxf(m, ph[j]);
        trunc = m + (float)M_PI;
      }
      dphi = trunc / a.vs_phi / fstride;
    }
    if (ok) {
      o.valid = 1;
      o.rho_i = rho_i; o.phi_i = phi_i;
      o.r_rho = (int)gaussian_radius_f(drho, drho, a.ov_minus, a.ov_plus, a.ov);
      o.r_phi = (int)gaussian_radius_f(dphi, dphi, a.ov_minus, a.ov_plus, a.ov);
      o.cls = a.cls[idx];
      o.cx = x; o.cy = y; o.crho = crho; o.cphi = cphi;
      const double sigma = (double)max(2 * o.r_rho + 1, 2 * o.r_phi + 1) / 6.0;
      o.inv2s2 = 1.0 / (2.0 * sigma * sigma);
    }
  }
  a.objs[idx] = o;
}

// one thread per cell: the LAST object whose window covers the cell gives the centre, every covering object contributes its
// Gaussian to the class maximum (the reference draws the objects one after the other)
__global__ __launch_bounds__(kT) void vote_raster_kernel(VoteArgs a) {
  extern __shared__ VoteObj s_obj[];
  __shared__ unsigned s_cnt[kT / 64];
  const int b = blockIdx.y;
  const int n = min(min(a.cnt[b], a.num_max_objs), a.max_boxes);
  for (int i = threadIdx.x; i < n; i += kT) s_obj[i] = a.objs[(size_t)b * a.max_boxes + i];
  __syncthreads();
  const int cell = blockIdx.x * kT + threadIdx.x;
  const int ch = 4 + a.n_classes;
  bool nonzero = false;
  if (cell < a.H * a.W) {
    const int py = cell / a.W, px = cell - py * a.W;
    float ctr[4] = {0.f, 0.f, 0.f, 0.f};
    float* out = a.votemap + ((size_t)b * a.H * a.W + cell) * ch;
    for (int c = 0; c < a.n_classes; ++c) out[4 + c] = 0.f;
    for (int k = 0; k < n; ++k) {
      const VoteObj& o = s_obj[k];
      if (!o.valid) continue;
      const int dy = py - o.phi_i, dx = px - o.rho_i;
      if (abs(dy) > o.r_phi || abs(dx) > o.r_rho) continue;    // window clipped to the map by construction (the cell is inside it)
      ctr[0] = o.cx; ctr[1] = o.cy; ctr[2] = o.crho; ctr[3] = o.cphi;
      double g = exp(-(double)(dx * dx + dy * dy) * o.inv2s2);
      if (g < 2.220446049250313e-16) g = 0.0;                  // gaussian2D: h[h < eps * max] = 0 (max = 1 at the centre)
      out[4 + o.cls] = fmaxf(out[4 + o.cls], (float)g);
    }
    out[0] = ctr[0]; out[1] = ctr[1]; out[2] = ctr[2]; out[3] = ctr[3];
    nonzero = ctr[0] != 0.f;
  }
  const unsigned long long m = __ballot(nonzero);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (unsigned)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kT / 64; ++w) t += s_cnt[w];
    a.part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kT) void count_fold_kernel(const unsigned long long* __restrict__ part, int n, int32_t* __restrict__ out) {
  __shared__ unsigned long long red[kT];
  unsigned long long t = 0;
  for (int i = threadIdx.x; i < n; i += kT) t += part[i];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (int32_t)red[0];
}

// ------------------------------------------------------------------------------------------------ head tensors
struct Preds {
  const float* hm; int hm_ps; int ncls;
  const float* reg; int reg_ps;
  const float* hei; int hei_ps;
  const float* dim; int dim_ps;
  const float* rot; int rot_ps;
  const float* iou; int iou_ps;          // nullable
  const float* ctr; int ctr_ps;          // pred_centers
  const float* vcls; int vcls_ps;        // pred_vote_cls
  const float* grid;                     // offset_grid planar (2, H, W)
  int B, H, W;
};

// pred_boxes row of a query: [x + gx, y + gy, z, log dims (3), cos, sin]  (get_proper_xy + the concatenation at e2e_swv_head.py:212-222)
__device__ __forceinline__ void load_pred_box(const Preds& p, int b, int q, float* o) {
  const int HW = p.H * p.W;
  const size_t pix = (size_t)b * HW + q;
  o[0] = p.reg[pix * p.reg_ps] + p.grid[q];
  o[1] = p.reg[pix * p.reg_ps + 1] + p.grid[HW + q];
  o[2] = p.hei[pix * p.hei_ps];
  o[3] = p.dim[pix * p.dim_ps]; o[4] = p.dim[pix * p.dim_ps + 1]; o[5] = p.dim[pix * p.dim_ps + 2];
  o[6] = p.rot[pix * p.rot_ps]; o[7] = p.rot[pix * p.rot_ps + 1];
}

__device__ __forceinline__ void encode_gt(const float* g, float* e) {
  e[0] = g[0]; e[1] = g[1]; e[2] = g[2];
  e[3] = logf(fmaxf(g[3], 1e-5f)); e[4] = logf(fmaxf(g[4], 1e-5f)); e[5] = logf(fmaxf(g[5], 1e-5f));
  e[6] = cosf(g[6]); e[7] = sinf(g[6]);
}

// ------------------------------------------------------------------------------------------------ matcher cost
// cost[b][g][q] = -(sigmoid(hm[b, q, cls_g]) ** w_ce) * (exp(-sum_c |cw_c (pred_c - enc_c)|) ** w_bbox)   (matcher.py:78-93, 136-147)
__global__ __launch_bounds__(kT) void match_cost_kernel(Preds p, const float* __restrict__ gt_boxes, const int32_t* __restrict__ gt_cls,
                                                        const int32_t* __restrict__ gt_cnt, int max_boxes, float w_ce, float w_bbox,
                                                        const float* __restrict__ cw, float* __restrict__ cost, int rows) {
  extern __shared__ float s_enc[];   // [n][9]: encoded box + class
  const int b = blockIdx.y;
  const int n = min(gt_cnt[b], rows);
  for (int i = threadIdx.x; i < n; i += kT) {
    float e[8];
    encode_gt(gt_boxes + ((size_t)b * max_boxes + i) * 7, e);
    for (int c = 0; c < 8; ++c) s_enc[i * 9 + c] = e[c] * cw[c];
    s_enc[i * 9 + 8] = __int_as_float(gt_cls[(size_t)b * max_boxes + i]);
  }
  __syncthreads();
  const int q = blockIdx.x * kT + threadIdx.x;
  const int Q = p.H * p.W;
  if (q >= Q) return;
  float pb[8];
  load_pred_box(p, b, q, pb);
  for (int c = 0; c < 8; ++c) pb[c] *= cw[c];
  for (int g = 0; g < n; ++g) {
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) d += fabsf(pb[c] - s_enc[g * 9 + c]);
    const int cls = __float_as_int(s_enc[g * 9 + 8]);
    const float pr = 1.f / (1.f + expf(-p.hm[((size_t)b * Q + q) * p.hm_ps + cls]));
    cost[((size_t)b * rows + g) * Q + q] = -1.f * (powf(pr, w_ce) * powf(expf(-d), w_bbox));
  }
}

// ------------------------------------------------------------------------------------------------ criterion
struct CritArgs {
  Preds p;
  const float* votemap;          // (B, H, W, 4 + C)
  const int32_t* vote_count;     // device scalar
  const int32_t* m_b;            // matches: sample, query, gt row
  const int32_t* m_q;
  const int32_t* m_g;
  int n_match;
  const float* gt_boxes; const int32_t* gt_cls; int max_boxes;
  float num_boxes;
  float w_ce, w_bbox, w_vote, w_vote_cls, w_iou;
  float sigma, gamma, alpha;
  float cw[8];
  int32_t* pos_label;            // (B, Q): 0 = background, class + 1 at the matched queries (scratch)
  double* part;                  // block partials: [which][block]
  int nblk_hm, nblk_vc, nblk_vote;
  float* out;                    // [det, ce, bbox, vote, vote_cls, iou, loc_elem x 8]
  float* d_hm; float* d_boxes; float* d_ctr; float* d_vcls; float* d_iou;   // NHWC gradients (nullable all together)
};

__device__ __forceinline__ float smooth_l1_sigma(float x, float sigma, float& grad) {
  const float s2 = sigma * sigma, ax = fabsf(x);
  if (ax < 1.f / s2) {
    grad = s2 * x;
    return 0.5f * (sigma * x) * (sigma * x);
  }
  grad = x > 0.f ? 1.f : -1.f;
  return ax - 0.5f / s2;
}

__device__ __forceinline__ float focal_term(float x, float t, float gamma, float alpha, float& grad) {
  // E2ESigmoidFocalClassificationLoss (loss_utils.py:478-503) and its derivative in the logit
  const float pr = 1.f / (1.f + expf(-x));
  const float aw = t * alpha + (1.f - t) * (1.f - alpha);
  const float pt = t * (1.f - pr) + (1.f - t) * pr;
  const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
  const float ptg = powf(pt, gamma);
  const float dpt = (1.f - 2.f * t) * pr * (1.f - pr);
  grad = aw * (gamma * powf(pt, gamma - 1.f) * dpt * bce + ptg * (pr - t));
  return aw * ptg * bce;
}

__device__ __forceinline__ void block_store_partial(double v, double* dst) {
  __shared__ double red[kT / 64];
  v = pn::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kT / 64; ++w) t += red[w];
    *dst = t;
  }
}

__global__ void pos_label_kernel(CritArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_match) return;
  const int Q = a.p.H * a.p.W;
  a.pos_label[(size_t)a.m_b[i] * Q + a.m_q[i]] = a.gt_cls[(size_t)a.m_b[i] * a.max_boxes + a.m_g[i]] + 1;
}

// classification loss over every (query, class): target = one-hot of the matched queries   (set_crit.py:168-176)
__global__ __launch_bounds__(kT) void crit_ce_kernel(CritArgs a) {
  const size_t total = (size_t)a.p.B * a.p.H * a.p.W * a.p.ncls;
  const float gs = a.w_ce / a.num_boxes;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < total; i += (size_t)gridDim.x * kT) {
    const size_t pix = i / a.p.ncls;
    const int c = (int)(i - pix * a.p.ncls);
    const float t = a.pos_label[pix] == c + 1 ? 1.f : 0.f;
    float g;
    acc += (double)focal_term(a.p.hm[pix * a.p.hm_ps + c], t, a.gamma, a.alpha, g);
    if (a.d_hm) a.d_hm[pix * a.p.ncls + c] = g * gs;
  }
  block_store_partial(acc, a.part + blockIdx.x);
}

// vote classification: dense targets from the vote map's class channels, normalised by the number of vote cells  (set_crit.py:162-166)
__global__ __launch_bounds__(kT) void crit_vote_cls_kernel(CritArgs a) {
  const size_t total = (size_t)a.p.B * a.p.H * a.p.W * a.p.ncls;
  const float vn = fmaxf((float)*a.vote_count, 1.f);
  const float gs = a.w_vote_cls / vn;
  const int ch = 4 + a.p.ncls;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < total; i += (size_t)gridDim.x * kT) {
    const size_t pix = i / a.p.ncls;
    const int c = (int)(i - pix * a.p.ncls);
    float g;
    acc += (double)focal_term(a.p.vcls[pix * a.p.vcls_ps + c], a.votemap[pix * ch + 4 + c], a.gamma, a.alpha, g);
    if (a.d_vcls) a.d_vcls[pix * a.p.ncls + c] = g * gs;
  }
  block_store_partial(acc, a.part + a.nblk_hm + blockIdx.x);
}

// vote regression at the cells whose vote-map centre is set  (set_crit.py:129-146, 152-160)
__global__ __launch_bounds__(kT) void crit_vote_kernel(CritArgs a) {
  const int Q = a.p.H * a.p.W;
  const size_t total = (size_t)a.p.B * Q;
  const float vn = fmaxf((float)*a.vote_count, 1.f);
  const float gs = a.w_vote / vn;
  const int ch = 4 + a.p.ncls;
  double acc = 0.0;
  for (size_t pix = (size_t)blockIdx.x * kT + threadIdx.x; pix < total; pix += (size_t)gridDim.x * kT) {
    const int q = (int)(pix % Q);
    const float* vm = a.votemap + pix * ch;
    float g0 = 0.f, g1 = 0.f;
    if (vm[0] != 0.f) {
      const float d0 = (a.p.ctr[pix * a.p.ctr_ps] + a.p.grid[q]) - vm[0];
      const float d1 = (a.p.ctr[pix * a.p.ctr_ps + 1] + a.p.grid[Q + q]) - vm[1];
      acc += (double)smooth_l1_sigma(d0, a.sigma, g0) + (double)smooth_l1_sigma(d1, a.sigma, g1);
    }
    if (a.d_ctr) {
      a.d_ctr[pix * 2] = g0 * gs;
      a.d_ctr[pix * 2 + 1] = g1 * gs;
    }
  }
  block_store_partial(acc, a.part + a.nblk_hm + a.nblk_vc + blockIdx.x);
}

// matched pairs: box regression + IoU branch, folded by ONE block (<= a few thousand pairs), then the grand total
__global__ __launch_bounds__(kT) void crit_matched_final_kernel(CritArgs a) {
  __shared__ double red[kT][10];
  const int Q = a.p.H * a.p.W;
  double elem[8] = {0, 0, 0, 0, 0, 0, 0, 0}, iou_acc = 0.0;
  for (int i = threadIdx.x; i < a.n_match; i += kT) {
    const int b = a.m_b[i], q = a.m_q[i];
    const float* g = a.gt_boxes + ((size_t)b * a.max_boxes + a.m_g[i]) * 7;
    float pb[8], e[8];
    load_pred_box(a.p, b, q, pb);
    encode_gt(g, e);
    const size_t pix = (size_t)b * Q + q;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float gr;
      elem[c] += (double)smooth_l1_sigma((e[c] - pb[c]) * a.cw[c], a.sigma, gr);
      if (a.d_boxes) a.d_boxes[pix * 8 + c] = -a.cw[c] * gr * (a.w_bbox / a.num_boxes);   // delta = gt - pred
    }
    if (a.p.iou) {
      // target: 2 * IoU3D(decoded prediction, gt) - 1 (no gradient through the target), smooth L1 with beta = 1
      float da[7], db[7];
      da[0] = pb[0]; da[1] = pb[1]; da[2] = pb[2]; da[3] = expf(pb[4]); da[4] = expf(pb[3]); da[5] = expf(pb[5]);
      da[6] = -atan2f(pb[7], pb[6]) - (float)M_PI / 2;
      db[0] = g[0]; db[1] = g[1]; db[2] = g[2]; db[3] = g[4]; db[4] = g[3]; db[5] = g[5]; db[6] = -g[6] - (float)M_PI / 2;
      const float bev = pn_geom::overlap_bev(da, db);
      const float hmax = fminf(da[2] + da[5] / 2, db[2] + db[5] / 2), hmin = fmaxf(da[2] - da[5] / 2, db[2] - db[5] / 2);
      const float ov = bev * fmaxf(hmax - hmin, 0.f);
      float iou = ov / fmaxf(da[3] * da[4] * da[5] + db[3] * db[4] * db[5] - ov, 1e-6f);
      if (iou != iou) iou = 0.f;
      const float tgt = 2.f * iou - 1.f;
      const float d = a.p.iou[pix * a.p.iou_ps] - tgt, ad = fabsf(d);
      iou_acc += ad < 1.f ? 0.5 * (double)d * d : (double)ad - 0.5;
      if (a.d_iou) a.d_iou[pix] = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * (a.w_iou / a.num_boxes);
    }
  }
  for (int c = 0; c < 8; ++c) red[threadIdx.x][c] = elem[c];
  red[threadIdx.x][8] = iou_acc;
  // slot 9: this thread's share of the dense partials (ce, vote_cls, vote), three separate folds below
  __syncthreads();
  __shared__ double tot[12];
  if (threadIdx.x < 9) {
    double t = 0.0;
    for (int k = 0; k < kT; ++k) t += red[k][threadIdx.x];
    tot[threadIdx.x] = t;
  } else if (threadIdx.x < 12) {
    const int which = threadIdx.x - 9;
    const double* src = a.part + (which == 0 ? 0 : (which == 1 ? a.nblk_hm : a.nblk_hm + a.nblk_vc));
    const int n = which == 0 ? a.nblk_hm : (which == 1 ? a.nblk_vc : a.nblk_vote);
    double t = 0.0;
    for (int k = 0; k < n; ++k) t += src[k];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double vn = fmax((double)*a.vote_count, 1.0), nb = (double)a.num_boxes;
    double bbox = 0.0;
    for (int c = 0; c < 8; ++c) {
      bbox += tot[c];
      a.out[6 + c] = (float)(tot[c] / nb);
    }
    const double l_ce = tot[9] / nb, l_bbox = bbox / nb, l_vc = tot[10] / vn, l_vote = tot[11] / vn, l_iou = a.p.iou ? tot[8] / nb : 0.0;
    a.out[1] = (float)l_ce; a.out[2] = (float)l_bbox; a.out[3] = (float)l_vote; a.out[4] = (float)l_vc; a.out[5] = (float)l_iou;
    a.out[0] = (float)(a.w_ce * l_ce + a.w_bbox * l_bbox + a.w_vote * l_vote + a.w_vote_cls * l_vc + (a.p.iou ? a.w_iou * l_iou : 0.0));
  }
}

int fill_preds(Preds& p, const float* hm, int hm_ps, int ncls, const float* reg, int reg_ps, const float* hei, int hei_ps, const float* dim,
               int dim_ps, const float* rot, int rot_ps, const float* iou, int iou_ps, const float* ctr, int ctr_ps, const float* vcls,
               int vcls_ps, const float* grid, int B, int H, int W) {
  PN_REQUIRE(hm && reg && hei && dim && rot && grid && B >= 1 && H >= 1 && W >= 1 && ncls >= 1, "swv criterion: bad head tensors");
  p.hm = hm; p.hm_ps = hm_ps; p.ncls = ncls; p.reg = reg; p.reg_ps = reg_ps; p.hei = hei; p.hei_ps = hei_ps; p.dim = dim; p.dim_ps = dim_ps;
  p.rot = rot; p.rot_ps = rot_ps; p.iou = iou; p.iou_ps = iou_ps; p.ctr = ctr; p.ctr_ps = ctr_ps; p.vcls = vcls; p.vcls_ps = vcls_ps;
  p.grid = grid; p.B = B; p.H = H; p.W = W;
  return PN_OK;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int pn_swv_gt_compact(const float* global_box, int batch, int max_boxes, int cols, const int32_t* class_ids, int n_classes, float* gt_boxes,
                      int32_t* gt_classes, int32_t* gt_counts, pn_stream_t stream) {
  PN_REQUIRE(global_box && class_ids && gt_boxes && gt_classes && gt_counts, "swv_gt_compact: null pointer");
  PN_REQUIRE(batch >= 1 && max_boxes >= 1 && cols >= 8 && n_classes >= 1, "swv_gt_compact: rows are [x, y, z, dx, dy, dz, heading, class]");
  hipLaunchKernelGGL(gt_compact_kernel, dim3(batch), dim3(64), 0, pn::S(stream), global_box, max_boxes, cols, class_ids, n_classes, gt_boxes,
                     gt_classes, gt_counts);
  return pn::check_launch("gt_compact_kernel");
}

size_t pn_swv_votemap_workspace_bytes(int batch, int max_boxes, int h, int w) {
  return al256((size_t)batch * max_boxes * sizeof(VoteObj)) + al256((size_t)batch * pn::cdiv(h * w, kT) * 8);
}

int pn_swv_draw_votemap_f32(const float* gt_boxes, const int32_t* gt_classes, const int32_t* gt_counts, int batch, int max_boxes, int n_classes,
                            const float* max_space, const float* min_space, const int32_t* grid, int stride, int num_max_objs,
                            double gaussian_overlap_d, float* votemap, int32_t* vote_count, void* workspace, size_t workspace_bytes,
                            pn_stream_t stream) {
  PN_REQUIRE(gt_boxes && gt_classes && gt_counts && max_space && min_space && grid && votemap && vote_count && workspace, "swv_draw_votemap: null pointer");
  PN_REQUIRE(batch >= 1 && max_boxes >= 1 && n_classes >= 1 && stride >= 1, "swv_draw_votemap: bad sizes");
  VoteArgs a;
  a.boxes = gt_boxes; a.cls = gt_classes; a.cnt = gt_counts; a.B = batch; a.max_boxes = max_boxes; a.n_classes = n_classes;
  a.W = grid[0] / stride; a.H = grid[1] / stride;      // feature_map_size = grid_size[::-1] / stride: [z, phi, rho]
  a.stride = stride; a.num_max_objs = num_max_objs;
  // voxel sizes as the reference forms them: float64 (max - min) / grid, rounded to fp32 when they meet the fp32 box tensors
  a.vs_rho = (float)(((double)max_space[0] - (double)min_space[0]) / grid[0]);
  a.vs_phi = (float)(((double)max_space[1] - (double)min_space[1]) / grid[1]);
  a.min_rho = min_space[0]; a.min_phi = min_space[1];
  a.ov_minus = (float)(1.0 - (double)gaussian_overlap_d); a.ov_plus = (float)(1.0 + (double)gaussian_overlap_d); a.ov = (float)gaussian_overlap_d;
  PN_REQUIRE(a.H >= 1 && a.W >= 1, "swv_draw_votemap: empty map");
  if (workspace_bytes < pn_swv_votemap_workspace_bytes(batch, max_boxes, a.H, a.W)) return pn::fail(PN_ERR_WORKSPACE, "swv_draw_votemap: workspace too small");
  a.objs = static_cast<VoteObj*>(workspace);
  a.part = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + al256((size_t)batch * max_boxes * sizeof(VoteObj)));
  a.votemap = votemap; a.vote_count = vote_count;
  hipStream_t st = pn::S(stream);
  const size_t smem = (size_t)max_boxes * sizeof(VoteObj);
  PN_REQUIRE(smem <= 64 * 1024, "swv_draw_votemap: more than ~1300 boxes per sample");
  hipLaunchKernelGGL(vote_prep_kernel, dim3(pn::cdiv(batch * max_boxes, 256)), dim3(256), 0, st, a);
  const int nb = pn::cdiv(a.H * a.W, kT);
  hipLaunchKernelGGL(vote_raster_kernel, dim3(nb, batch), dim3(kT), smem, st, a);
  hipLaunchKernelGGL(count_fold_kernel, dim3(1), dim3(kT), 0, st, a.part, nb * batch, vote_count);
  return pn::check_launch("swv_draw_votemap");
}

int pn_swv_match_cost_f32(const float* hm, int hm_ps, int ncls, const float* reg, int reg_ps, const float* height, int height_ps,
                          const float* dim, int dim_ps, const float* rot, int rot_ps, const float* offset_grid, int batch, int h, int w,
                          const float* gt_boxes, const int32_t* gt_classes, const int32_t* gt_counts, int max_boxes, int rows, float w_ce,
                          float w_bbox, const float* code_weights, float* cost, pn_stream_t stream) {
  Preds p;
  if (int rc = fill_preds(p, hm, hm_ps, ncls, reg, reg_ps, height, height_ps, dim, dim_ps, rot, rot_ps, nullptr, 0, nullptr, 0, nullptr, 0,
                          offset_grid, batch, h, w)) return rc;
  PN_REQUIRE(gt_boxes && gt_classes && gt_counts && code_weights && cost && rows >= 1 && rows <= max_boxes, "swv_match_cost: bad arguments");
  const size_t smem = (size_t)rows * 9 * sizeof(float);
  PN_REQUIRE(smem <= 64 * 1024, "swv_match_cost: too many ground-truth rows");
  hipLaunchKernelGGL(match_cost_kernel, dim3(pn::cdiv(h * w, kT), batch), dim3(kT), smem, pn::S(stream), p, gt_boxes, gt_classes, gt_counts,
                     max_boxes, w_ce, w_bbox, code_weights, cost, rows);
  return pn::check_launch("match_cost_kernel");
}

/* Rectangular linear sum assignment on the HOST (scipy.optimize.linear_sum_assignment in the reference, matcher.py:149):
 * shortest augmenting paths with dual variables (Crouse, "On implementing 2D rectangular assignment algorithms", 2016).
 * cost: (nr, nc) row major, nr <= nc; col_of_row[nr] receives the column assigned to every row. */
int pn_lsap_f32(const float* cost, int nr, int nc, int32_t* col_of_row) {
  PN_REQUIRE(cost && col_of_row && nr >= 0 && nc >= nr, "lsap: needs nr <= nc");
  if (nr == 0) return PN_OK;
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> u(nr, 0.0), v(nc, 0.0), shortest(nc);
  std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  for (int cur = 0; cur < nr; ++cur) {
    // augmenting path from row `cur`
    double min_val = 0.0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;   // scipy scans the columns in this (reversed) order
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(shortest.begin(), shortest.end(), inf);
    int sink = -1, i = cur;
    while (sink == -1) {
      int index = -1;
      double lowest = inf;
      SR[i] = 1;
      const float* row = cost + (size_t)i * nc;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = remaining[it];
        const double r = min_val + (double)row[j] - u[i] - v[j];
        if (r < shortest[j]) {
          path[j] = i;
          shortest[j] = r;
        }
        // prefer a column that is still unassigned among equal distances (as scipy does)
        if (shortest[j] < lowest || (shortest[j] == lowest && row4col[j] == -1)) {
          lowest = shortest[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == inf) return pn::fail(PN_ERR_INVALID, "lsap: infeasible cost matrix");
      const int j = remaining[index];
      if (row4col[j] == -1) sink = j;
      else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--num_remaining];
    }
    u[cur] += min_val;
    for (int r = 0; r < nr; ++r)
      if (SR[r] && r != cur) u[r] += min_val - shortest[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= min_val - shortest[j];
    int j = sink;
    while (true) {
      const int r = path[j];
      row4col[j] = r;
      std::swap(col4row[r], j);
      if (r == cur) break;
    }
  }
  for (int r = 0; r < nr; ++r) col_of_row[r] = col4row[r];
  return PN_OK;
}

size_t pn_swv_criterion_workspace_bytes(int batch, int h, int w) {
  return al256((size_t)batch * h * w * 4) + al256((size_t)3 * 1024 * 8);
}

int pn_swv_set_criterion_f32(const float* hm, int hm_ps, int ncls, const float* reg, int reg_ps, const float* height, int height_ps,
                             const float* dim, int dim_ps, const float* rot, int rot_ps, const float* iou, int iou_ps, const float* pred_centers,
                             int centers_ps, const float* pred_vote_cls, int vote_cls_ps, const float* offset_grid, int batch, int h, int w,
                             const float* votemap, const int32_t* vote_count, const int32_t* match_sample, const int32_t* match_query,
                             const int32_t* match_gt, int n_match, const float* gt_boxes, const int32_t* gt_classes, int max_boxes,
                             float num_boxes, const float* loss_weights, float sigma, float gamma, float alpha, const float* code_weights,
                             float* out, float* d_hm, float* d_boxes, float* d_centers, float* d_vote_cls, float* d_iou, void* workspace,
                             size_t workspace_bytes, pn_stream_t stream) {
  CritArgs a;
  if (int rc = fill_preds(a.p, hm, hm_ps, ncls, reg, reg_ps, height, height_ps, dim, dim_ps, rot, rot_ps, iou, iou_ps, pred_centers, centers_ps,
                          pred_vote_cls, vote_cls_ps, offset_grid, batch, h, w)) return rc;
  PN_REQUIRE(pred_centers && pred_vote_cls && votemap && vote_count && gt_boxes && gt_classes && loss_weights && code_weights && out && workspace,
             "swv_set_criterion: null pointer");
  PN_REQUIRE(n_match == 0 || (match_sample && match_query && match_gt), "swv_set_criterion: matches missing");
  PN_REQUIRE(num_boxes > 0.f, "swv_set_criterion: num_boxes must be positive (clamped to >= 1 by the caller)");
  const bool grads = d_hm || d_boxes || d_centers || d_vote_cls || d_iou;
  PN_REQUIRE(!grads || (d_hm && d_boxes && d_centers && d_vote_cls && (d_iou || !iou)), "swv_set_criterion: either every gradient buffer or none");
  if (workspace_bytes < pn_swv_criterion_workspace_bytes(batch, h, w)) return pn::fail(PN_ERR_WORKSPACE, "swv_set_criterion: workspace too small");
  a.votemap = votemap; a.vote_count = vote_count; a.m_b = match_sample; a.m_q = match_query; a.m_g = match_gt; a.n_match = n_match;
  a.gt_boxes = gt_boxes; a.gt_cls = gt_classes; a.max_boxes = max_boxes; a.num_boxes = num_boxes;
  a.w_ce = loss_weights[0]; a.w_bbox = loss_weights[1]; a.w_vote = loss_weights[2]; a.w_vote_cls = loss_weights[3]; a.w_iou = loss_weights[4];
  a.sigma = sigma; a.gamma = gamma; a.alpha = alpha;
  for (int c = 0; c < 8; ++c) a.cw[c] = code_weights[c];
  a.pos_label = static_cast<int32_t*>(workspace);
  a.part = reinterpret_cast<double*>(static_cast<char*>(workspace) + al256((size_t)batch * h * w * 4));
  const size_t cells = (size_t)batch * h * w;
  a.nblk_hm = (int)std::min<size_t>(1024, (cells * ncls + kT - 1) / kT);
  a.nblk_vc = a.nblk_hm;
  a.nblk_vote = (int)std::min<size_t>(1024, (cells + kT - 1) / kT);
  a.out = out; a.d_hm = d_hm; a.d_boxes = d_boxes; a.d_ctr = d_centers; a.d_vcls = d_vote_cls; a.d_iou = d_iou;
  hipStream_t st = pn::S(stream);
  if (int rc = pn::zero_async(a.pos_label, cells * 4, st)) return rc;
  if (d_boxes)
    if (int rc = pn::zero_async(d_boxes, cells * 8 * 4, st)) return rc;
  if (d_iou)
    if (int rc = pn::zero_async(d_iou, cells * 4, st)) return rc;
  if (n_match > 0) hipLaunchKernelGGL(pos_label_kernel, dim3(pn::cdiv(n_match, 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(crit_ce_kernel, dim3(a.nblk_hm), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(crit_vote_cls_kernel, dim3(a.nblk_vc), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(crit_vote_kernel, dim3(a.nblk_vote), dim3(kT), 0, st, a);
  hipLaunchKernelGGL(crit_matched_final_kernel, dim3(1), dim3(kT), 0, st, a);
  return pn::check_launch("swv_set_criterion");
}

}  // extern "C"
