/*
 * CPU ORACLE (plain C) -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 * Integer / index stages of the polar-voxel path restated in C; checked by
 * tests/test_oracle_c.py against the same golden vectors as oracle/polar_oracle.py
 * (captured from the reference by tests/golden/make_golden.py).
 *
 *   ov_grid_index    det3d/datasets/pipelines/voxelization.py:165-168 (+ collate.py:157-164)
 *   ov_unique        torch.unique(dim=0) at det3d/models/readers/pillar_encoder.py:398
 *   ov_scatter_mean  torch_scatter.scatter_mean at det3d/models/readers/voxel_encoder.py:43
 *   ov_hard_voxelize det3d/ops/point_cloud/point_cloud_ops.py:7-72 (reverse-index kernel)
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; -ffp-contract=off so fp32 ops are rounded one by one)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* points (n x stride) fp32 polar; out (n x 4) int64 [b,z,theta,r]; grid = {R,T,Z} */
void ov_grid_index(const float *pts, int n, int stride, int b, const float *lo, const float *vs, const int *grid,
                   int64_t *out) {
  for (int i = 0; i < n; ++i) {
    int64_t c[3];
    for (int a = 0; a < 3; ++a) {
      volatile float d = pts[(size_t)i * stride + a] - lo[a]; /* fp32 subtract */
      volatile float q = d / vs[a];                            /* fp32 divide   */
      double qq = (double)q;
      if (qq < 0.0) qq = 0.0;
      if (qq > (double)(grid[a] - 1)) qq = (double)(grid[a] - 1);
      c[a] = (int64_t)floor(qq);
    }
    out[4 * (size_t)i + 0] = b;
    out[4 * (size_t)i + 1] = c[2];
    out[4 * (size_t)i + 2] = c[1];
    out[4 * (size_t)i + 3] = c[0];
  }
}

typedef struct { uint64_t key; int idx; } kv_t;
static int cmp_kv(const void *a, const void *b) {
  const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* lexicographic unique of (n x 4) rows; returns V; unq (cap n x 4), inv (n), cnt (cap n) */
int ov_unique(const int64_t *gi, int n, const int *grid, int64_t *unq, int64_t *inv, int64_t *cnt) {
  const uint64_t R = grid[0], T = grid[1], Z = grid[2];
  kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    const int64_t *g = gi + 4 * (size_t)i;
    kv[i].key = (((uint64_t)g[0] * Z + (uint64_t)g[1]) * T + (uint64_t)g[2]) * R + (uint64_t)g[3];
    kv[i].idx = i;
  }
  qsort(kv, (size_t)n, sizeof(kv_t), cmp_kv);
  int v = -1;
  for (int i = 0; i < n; ++i) {
    if (i == 0 || kv[i].key != kv[i - 1].key) {
      ++v;
      uint64_t q = kv[i].key;
      unq[4 * (size_t)v + 3] = (int64_t)(q % R); q /= R;
      unq[4 * (size_t)v + 2] = (int64_t)(q % T); q /= T;
      unq[4 * (size_t)v + 1] = (int64_t)(q % Z); q /= Z;
      unq[4 * (size_t)v + 0] = (int64_t)q;
      cnt[v] = 0;
    }
    inv[kv[i].idx] = v;
    cnt[v] += 1;
  }
  free(kv);
  return v + 1;
}

/* per-voxel mean, fp32 running sum in point order (the order index_add_ uses on the CPU) */
void ov_scatter_mean(const float *x, int n, int f, const int64_t *inv, int v, float *mean) {
  int64_t *c = (int64_t *)calloc((size_t)(v > 0 ? v : 1), sizeof(int64_t));
  memset(mean, 0, sizeof(float) * (size_t)v * f);
  for (int i = 0; i < n; ++i) {
    float *m = mean + (size_t)inv[i] * f;
    for (int k = 0; k < f; ++k) m[k] += x[(size_t)i * f + k];
    c[inv[i]] += 1;
  }
  for (int j = 0; j < v; ++j)
    for (int k = 0; k < f; ++k) mean[(size_t)j * f + k] /= (float)(c[j] > 0 ? c[j] : 1);
  free(c);
}

/* hard voxelization: first-come voxel ids, <=max_points per voxel, <=max_voxels voxels, out-of-range dropped.
 * voxels (max_voxels x max_points x f) zero-filled by the caller; coors (max_voxels x 3) [z,theta,r]; returns V */
int ov_hard_voxelize(const float *pts, int n, int f, const float *vs, const float *range, int max_points, int max_voxels,
                     float *voxels, int32_t *coors, int32_t *num) {
  int grid[3];
  for (int a = 0; a < 3; ++a) grid[a] = (int)nearbyintf((range[3 + a] - range[a]) / vs[a]);
  const size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  int32_t *map = (int32_t *)malloc(sizeof(int32_t) * cells);
  memset(map, 0xff, sizeof(int32_t) * cells);
  int nv = 0;
  for (int i = 0; i < n; ++i) {
    int c[3], ok = 1;
    for (int a = 0; a < 3; ++a) {
      volatile float d = pts[(size_t)i * f + a] - range[a];
      volatile float q = d / vs[a];
      const float fl = floorf(q);
      if (fl < 0.f || fl >= (float)grid[a]) { ok = 0; break; }
      c[a] = (int)fl;
    }
    if (!ok) continue;
    const size_t cell = ((size_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
    int id = map[cell];
    if (id < 0) {
      if (nv >= max_voxels) continue;
      id = nv++;
      map[cell] = id;
      coors[3 * id + 0] = c[2]; coors[3 * id + 1] = c[1]; coors[3 * id + 2] = c[0];
      num[id] = 0;
    }
    if (num[id] < max_points) {
      memcpy(voxels + ((size_t)id * max_points + num[id]) * f, pts + (size_t)i * f, sizeof(float) * f);
      num[id] += 1;
    }
  }
  free(map);
  return nv;
}
