/* TEST INFRASTRUCTURE (oracle): plain-C restatement of the reference's BEV rotated IoU and greedy NMS.
 *
 * Follows the algorithm of
 *   box_overlap / iou_bev / nms_kernel     det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-311
 *   nms_gpu (host-side greedy reduce)      det3d/ops/iou3d_nms/src/iou3d_nms.cpp:90-136
 *   rotate_nms_pcdet                       det3d/core/bbox/box_torch_ops.py:248-277
 * PARITY UNPINNED: the reference implementation is CUDA only (its CPU twin iou3d_cpu.cpp includes
 * <cuda.h>, absent here), so no reference output could be captured; the restatement is checked against
 * an independent float64 convex-clipping computation of the same quantity (tests/test_oracle_nms.py).
 *
 * Box: [x, y, z, dx, dy, dz, heading]; BEV rectangle = axis-aligned (dx, dy) rectangle rotated by
 * `heading` about its centre.  Overlap polygon = {edge intersections} U {corners of one box inside the
 * other (margin 1e-2)}; its vertices are ordered by angle about their mean and the area is the fan sum
 * of cross products.  IoU = overlap / max(area_a + area_b - overlap, 1e-8). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define OV_EPS 1e-8f

typedef struct { float x, y; } pt;

static float cross3(pt p1, pt p2, pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

static int bbox_overlap(pt p1, pt p2, pt q1, pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

/* segment p0-p1 against q0-q1; proper crossings only */
static int seg_intersection(pt p1, pt p0, pt q1, pt q0, pt *ans) {
  if (!bbox_overlap(p0, p1, q0, q1)) return 0;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > OV_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static int inside(const float *box, pt p) {
  const float c = cosf(-box[6]), s = sinf(-box[6]);
  const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
  const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
  return fabsf(rx) < box[3] / 2 + 1e-2f && fabsf(ry) < box[4] / 2 + 1e-2f;
}

static void corners(const float *b, pt *c) {
  const float hx = b[3] / 2, hy = b[4] / 2, co = cosf(b[6]), si = sinf(b[6]);
  const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
  for (int k = 0; k < 4; ++k) {
    /* rotate (centre + local) about the centre */
    const float px = b[0] + lx[k], py = b[1] + ly[k];
    c[k].x = (px - b[0]) * co + (py - b[1]) * (-si) + b[0];
    c[k].y = (px - b[0]) * si + (py - b[1]) * co + b[1];
  }
  c[4] = c[0];
}

float ov_box_overlap(const float *a, const float *b) {
  pt ca[5], cb[5], poly[16], ctr = {0.f, 0.f};
  corners(a, ca);
  corners(b, cb);
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &poly[cnt])) {
        ctr.x += poly[cnt].x; ctr.y += poly[cnt].y; ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (inside(a, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; poly[cnt++] = cb[k]; }
    if (inside(b, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; poly[cnt++] = ca[k]; }
  }
  ctr.x /= cnt; ctr.y /= cnt;  /* cnt == 0: NaN centre, the loops below do nothing, area 0 */
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

float ov_iou_bev(const float *a, const float *b) {
  const float sa = a[3] * a[4], sb = b[3] * b[4], so = ov_box_overlap(a, b);
  return so / fmaxf(sa + sb - so, OV_EPS);
}

void ov_iou_bev_matrix(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = ov_iou_bev(a + 7 * i, b + 7 * j);
}

/* greedy NMS over boxes already sorted by descending score; keep[] receives indices, returns the count */
int ov_nms_sorted(const float *boxes, int n, float thresh, int64_t *keep) {
  unsigned char *dead = (unsigned char *)calloc((size_t)n + 1, 1);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (dead[i]) continue;
    keep[m++] = i;
    for (int j = i + 1; j < n; ++j)
      if (!dead[j] && ov_iou_bev(boxes + 7 * i, boxes + 7 * j) > thresh) dead[j] = 1;
  }
  free(dead);
  return m;
}
