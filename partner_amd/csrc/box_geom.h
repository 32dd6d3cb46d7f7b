// Rotated-rectangle geometry in the BEV plane shared by the NMS (postproc.hip) and the IoU target of the geometry-aware head's
// loss (e2e_loss.hip).  Arithmetic of box_overlap / iou_bev, det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-311: overlap polygon =
// edge crossings + contained corners (1e-2 margin), ordered by angle about their mean, fan-summed cross products.
// Boxes are [x, y, z, dx, dy, dz, angle] in the kernel's own convention (rotate_nms_pcdet's / to_pcdet's output).
#pragma once
#include <hip/hip_runtime.h>

namespace pn_geom {

constexpr float kEps = 1e-8f;

struct pt { float x, y; };

__device__ __forceinline__ float cross3(pt p1, pt p2, pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

__device__ __forceinline__ bool bbox_overlap(pt p1, pt p2, pt q1, pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) && fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) &&
         fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ inline bool seg_intersection(pt p1, pt p0, pt q1, pt q0, pt& ans) {
  if (!bbox_overlap(p0, p1, q0, q1)) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ inline bool inside(const float* box, pt p) {
  const float c = cosf(-box[6]), s = sinf(-box[6]);
  const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
  const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
  return fabsf(rx) < box[3] / 2 + 1e-2f && fabsf(ry) < box[4] / 2 + 1e-2f;
}

__device__ inline void corners(const float* b, pt* c) {
  const float hx = b[3] / 2, hy = b[4] / 2, co = cosf(b[6]), si = sinf(b[6]);
  const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float px = b[0] + lx[k], py = b[1] + ly[k];
    c[k].x = (px - b[0]) * co + (py - b[1]) * (-si) + b[0];
    c[k].y = (px - b[0]) * si + (py - b[1]) * co + b[1];
  }
  c[4] = c[0];
}

// area of the intersection of the two rectangles' footprints
__device__ inline float overlap_bev(const float* a, const float* b) {
  // Boxes whose centres are further apart than the sum of their half diagonals (+ the 1e-2 containment margin) share no
  // point: the overlap polygon below would come out empty (area 0), so skip its ~1.5 kFLOP.
  {
    const float dx = a[0] - b[0], dy = a[1] - b[1];
    const float r = 0.5f * (sqrtf(a[3] * a[3] + a[4] * a[4]) + sqrtf(b[3] * b[3] + b[4] * b[4])) + 0.05f;
    if (dx * dx + dy * dy > r * r) return 0.f;
  }
  pt ca[5], cb[5], poly[16], ctr = {0.f, 0.f};
  corners(a, ca);
  corners(b, cb);
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], poly[cnt])) {
        ctr.x += poly[cnt].x; ctr.y += poly[cnt].y; ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (inside(a, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; poly[cnt++] = cb[k]; }
    if (inside(b, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; poly[cnt++] = ca[k]; }
  }
  ctr.x /= cnt; ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(poly[i].y - ctr.y, poly[i].x - ctr.x) > atan2f(poly[i + 1].y - ctr.y, poly[i + 1].x - ctr.x)) {
        const pt t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const float ax = poly[k].x - poly[0].x, ay = poly[k].y - poly[0].y;
    const float bx = poly[k + 1].x - poly[0].x, by = poly[k + 1].y - poly[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

__device__ inline float iou_bev(const float* a, const float* b) {
  const float so = overlap_bev(a, b);
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  return so / fmaxf(sa + sb - so, kEps);
}

}  // namespace pn_geom
