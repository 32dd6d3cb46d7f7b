/*
 * partner_hip.h -- C ABI of libpartner_hip.so (gfx950 / MI355X).
 *
 * The reference (fudan-zvg/PARTNER, a det3d fork) has no FFI for this path: its plugin
 * boundary is the Python registry API (det3d/utils/registry.py:28-78,
 * det3d/models/builder.py:17-53) and the GPU work is library calls (torch.unique,
 * torch_scatter, cuDNN).  This header is the boundary a det3d maintainer would bind with
 * ctypes: every entry point below names the reference function whose
 * arithmetic it replaces (file:line under the reference root).
 *
 * Conventions
 *   - extern "C"; plain pointers and sizes; no torch / C++ types.
 *   - every data pointer is a DEVICE pointer owned by the caller; the library never
 *     allocates, frees or synchronises (hipGraph-capturable).  Scratch memory is passed in
 *     as `workspace`; its size comes from the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void*.
 *   - return value: 0 = ok, <0 = error (PN_ERR_*); the message is kept per host thread and
 *     read with pn_last_error().  The library never exits the process.
 *   - activations are NHWC fp32 ("channels-last"): element (b,y,x,c) lives at
 *     ((b*H+y)*W+x)*pixel_stride + channel_offset + c.  BEV axes: y = theta (azimuth),
 *     x = r (range), exactly the (B,C,theta,r) logical layout of the reference tensors.
 *   - voxel indices are [b, z, theta, r] (reference order, collate.py:157-164).
 */
#ifndef PARTNER_HIP_H
#define PARTNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_OK 0
#define PN_ERR_INVALID (-1)   /* bad argument / unsupported shape              */
#define PN_ERR_LAUNCH (-2)    /* HIP reported a launch error                   */
#define PN_ERR_WORKSPACE (-3) /* workspace too small                           */

#define PN_ACT_NONE 0
#define PN_ACT_RELU 1
#define PN_ACT_TANH 2
#define PN_ACT_GELU 3

typedef void *pn_stream_t;

int pn_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
int pn_last_error(char *buf, size_t buf_len);
/* number of HIP devices visible, -1 if the runtime is unusable */
int pn_device_count(void);
/* Opaque per-device handle: the only object the library keeps for a caller.  Creation queries the device and FAILS LOUDLY when it
 * is not a gfx950 part (the library holds gfx950 code objects only); pn_handle_info reports what the roofline figures are priced
 * against (compute units, LDS per CU, HBM bytes, architecture string).  The compute entry points do not need a handle: they take
 * device pointers and a stream and keep no state between calls. */
typedef struct pn_handle_s *pn_handle_t;
int pn_handle_create(int device, pn_handle_t *out);
int pn_handle_destroy(pn_handle_t handle);
int pn_handle_info(pn_handle_t handle, int *device, int *compute_units, int *lds_bytes_per_cu, unsigned long long *hbm_bytes,
                   char *arch, size_t arch_len);
/* PCI bus id of the handle's device ("0000:05:00.0"), NUL terminated: identifies the physical GPU behind a rank in the
 * multi-GPU bench line (one process per GPU, as tools/train.py:103-107 launches the reference). */
int pn_handle_pci_bus_id(pn_handle_t handle, char *buf, size_t buf_len);

/* ---------------------------------------------------------------------------------------
 * V0  cart -> polar point decoration.
 * Replaces transform_points(pc,'cylinder')  det3d/datasets/pipelines/utils.py:34-47
 * cart: (n, f_in>=3) [x,y,z,rest...]  ->  polar: (n, f_in+2) [rho,phi,z,x,y,rest...]
 * rho = sqrt(x*x+y*y) in IEEE fp32 (bit-exact); phi = atan2 evaluated in fp64 and rounded
 * once to fp32 (differs from glibc atan2f by at most 1 ulp).
 */
int pn_cart_to_polar_f32(const float *cart, int n, int f_in, float *polar, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * V1  dynamic voxelization indices.
 * Replaces Voxelization.voxelize_dynamic  det3d/datasets/pipelines/voxelization.py:165-168
 *          + batch index prepend          det3d/torchie/parallel/collate.py:157-164
 * grid_ind[i] = [b, floor(clip((p[i,2]-lo[2])/vs[2],0,Z-1)), ...theta..., ...r...]  (int64, N x 4)
 * fp32 subtract, fp32 IEEE divide, clamp, floor -- bit-exact with numpy.
 * sample_offsets: device int32[batch+1], prefix offsets of each sample's points
 *                 (n = sample_offsets[batch] <= n_capacity).
 * range_lo[3], voxel_size[3]: host floats (r,theta,z order).  grid[3] = {R,T,Z}.
 * Either output may be NULL: grid_ind (int64 N x 4) and/or keys (uint32 linear key
 * ((b*Z+z)*T+theta)*R+r).
 */
int pn_polar_grid_index_f32(const float *points, int point_stride, int n_capacity,
                            const int32_t *sample_offsets, int batch, const float *range_lo,
                            const float *voxel_size, const int32_t *grid, int64_t *grid_ind,
                            uint32_t *keys, pn_stream_t stream);

/* keys from caller-supplied int64 grid_ind (N x 4), for the reference API where grid_ind is an input */
int pn_keys_from_grid_ind(const int64_t *grid_ind, int n, const int32_t *grid, int batch,
                          uint32_t *keys, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Unique voxels in lexicographic (b,z,theta,r) order == rank in linear-key order.
 * Replaces torch.unique(grid_ind, return_inverse, return_counts, dim=0)
 *          det3d/models/readers/pillar_encoder.py:398 ; voxel_encoder.py:42
 * Occupancy bitmap + popcount prefix scan; no sort, no host sync.
 *   keys       uint32[n]            (from pn_polar_grid_index_f32 / pn_keys_from_grid_ind)
 *   n_dev      optional device int32: actual n (<= n_capacity); NULL -> n_capacity
 *   num_cells  batch*Z*T*R  (< 2^32)
 * outputs (capacity n_capacity rows each; bit-exact vs torch.unique):
 *   unq        int64[V x 4]  (may be NULL)     unq_inv  int32[n]
 *   unq_cnt    int32[V]                        num_voxels  device int32[1] = V
 */
size_t pn_unique_workspace_bytes(uint64_t num_cells, int n_capacity);
int pn_unique_rank_bitmap(const uint32_t *keys, int n_capacity, const int32_t *n_dev,
                          uint64_t num_cells, const int32_t *grid, int64_t *unq, int32_t *unq_inv,
                          int32_t *unq_cnt, int32_t *num_voxels, void *workspace,
                          size_t workspace_bytes, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Bucket the points by voxel rank: voxel_start[v] = exclusive scan of unq_cnt (V+1 entries,
 * capacity n_capacity+1) and order[] = point indices grouped by voxel (order inside a voxel
 * is unspecified; every forward consumer below reduces with order-independent arithmetic).
 */
size_t pn_bucket_workspace_bytes(int n_capacity);
int pn_bucket_points(const int32_t *unq_inv, const int32_t *unq_cnt, int n_capacity,
                     const int32_t *n_dev, const int32_t *num_voxels, int32_t *voxel_start,
                     int32_t *order, void *workspace, size_t workspace_bytes, pn_stream_t stream);
/* ---------------------------------------------------------------------------------------
 * The per-frame index of the dynamic path in THREE launches (V0 + V1 + unique + bucketing): cart -> polar rows and grid
 * index as pn_cart_to_polar_f32 / pn_polar_grid_index_f32, voxel ranks in the row order of torch.unique(grid_ind, dim=0)
 * (det3d/models/readers/pillar_encoder.py:398) by ONE single-pass scan over per-cell point counts, points bucketed by voxel.
 * Same unq_keys / voxel_start / num_voxels / point sets per voxel as pn_unique_rank_bitmap + pn_bucket_points.
 *   cell_count: one uint32 per grid cell (batch*Z*T*R), ALL ZERO on entry; left holding the voxels' first point slots --
 *     pn_clear_frame_cells zeroes those entries again (sparse), so a persistent buffer never needs a dense fill.
 *   scan_state: pn_voxel_index_fused_state_bytes bytes, all zero on entry, left all zero.
 *   polar (n, f_in+2), keys (n) uint32, pos (n) int32 (slot of the point inside its cell), unq_keys (n), voxel_start (n+1),
 *   order (n), num_voxels (1): outputs; rows / entries past the counts are not written.
 */
size_t pn_voxel_index_fused_state_bytes(uint64_t num_cells);
int pn_voxel_index_fused_f32(const float *cart, int n_capacity, int f_in, const int32_t *sample_offsets,
                             int batch, const float *range_lo, const float *voxel_size,
                             const int32_t *grid, float *polar, uint32_t *keys, int32_t *pos,
                             uint32_t *cell_count, void *scan_state, size_t scan_state_bytes,
                             uint32_t *unq_keys, int32_t *voxel_start, int32_t *order,
                             int32_t *num_voxels, pn_stream_t stream);
/* The same index, also leaving row_start [batch * grid[2] * grid[1] + 1] (r6): the number of voxels in the grid rows (runs of grid[0]
 * cells) before row g.  unq_keys is in key order, so the voxels of row g are the run [row_start[g], row_start[g + 1]) of it: what
 * pn_pillar_conv3x3_rows_f32 walks (one canvas row of the pillar map per run).  Written by the scan itself; grid[0] % 8 == 0. */
int pn_voxel_index_fused_rows_f32(const float *cart, int n_capacity, int f_in, const int32_t *sample_offsets,
                                  int batch, const float *range_lo, const float *voxel_size,
                                  const int32_t *grid, float *polar, uint32_t *keys, int32_t *pos,
                                  uint32_t *cell_count, void *scan_state, size_t scan_state_bytes,
                                  uint32_t *unq_keys, int32_t *voxel_start, int32_t *order,
                                  int32_t *num_voxels, int32_t *row_start, pn_stream_t stream);
/* The frame index of a MULTI-SWEEP frame straight from its raw sweeps (r6; BASELINE configs[4]; one sample): the accumulation of
 * pn_accumulate_sweeps_f32 (det3d/datasets/pipelines/loading.py:215-260: remove_close on the non-key sweeps, rigid transform, time lag) and
 * pn_voxel_index_fused[_rows]_f32 in the same three launches.  raw (n_capacity, raw_cols >= 4) concatenated sweeps, key frame first;
 * sweep_offsets (sweeps + 1) on the device; transforms (sweeps, 4, 4) float64; time_lags (sweeps).  The kept points are NOT compacted: point i of
 * the index is row i of raw, a removed point has keys[i] = 0xffffffff, pos[i] = -1 and appears in no voxel; polar (n_capacity, 7) rows
 * [rho, phi, z, x, y, intensity, dt] of removed points are not written.  row_start: nullable (as pn_voxel_index_fused_rows_f32).  Per kept
 * point the same arithmetic as the two calls one after the other. */
int pn_voxel_index_fused_sweeps_f32(const float *raw, int n_capacity, int raw_cols, const int32_t *sweep_offsets,
                                    int sweeps, const double *transforms, const float *time_lags,
                                    float min_distance, const float *range_lo, const float *voxel_size,
                                    const int32_t *grid, float *polar, uint32_t *keys, int32_t *pos,
                                    uint32_t *cell_count, void *scan_state, size_t scan_state_bytes,
                                    uint32_t *unq_keys, int32_t *voxel_start, int32_t *order,
                                    int32_t *num_voxels, int32_t *row_start, pn_stream_t stream);
/* end of frame: zero the canvas cells (nullable) and the cell_count entries (nullable) of the frame's voxels */
int pn_clear_frame_cells(const uint32_t *unq_keys, const int32_t *num_voxels, int v_capacity,
                         const int32_t *grid, int c, float *canvas, uint32_t *cell_count,
                         pn_stream_t stream);

/* order_out = order_in with every voxel run sorted by ascending point index (out of place).  Only
 * the PFN backward (pn_dynamic_pfn_bwd) depends on the order inside a run -- through floating-point
 * summation order -- so a training step that must be bit-reproducible sorts the runs first. */
int pn_sort_voxel_runs(const int32_t *voxel_start, const int32_t *num_voxels, int voxel_capacity,
                       const int32_t *order_in, int32_t *order_out, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * V2  hard voxelization (first-come voxel ids, <= max_points points per voxel in point order,
 * <= max_voxels voxels, points outside the grid dropped).
 * Replaces points_to_voxel / _points_to_voxel_reverse_kernel
 *          det3d/ops/point_cloud/point_cloud_ops.py:146-224, 7-72 (via VoxelGenerator.generate,
 *          det3d/core/input/voxel_generator.py:19-32)
 * points (n x f, row stride point_stride; first three features are the grid coordinates),
 * voxel_size[3], range[6] host floats (fp32 arithmetic as in the reference).  Outputs, bit-exact:
 *   voxels (max_voxels x max_points x f) zero filled, coors int32 (max_voxels x 3) [z,theta,r],
 *   num_points int32 (max_voxels), num_voxels device int32[1].
 * Deterministic: no sort, no order-dependent atomics (see voxelize.hip).
 */
size_t pn_hard_voxelize_workspace_bytes(uint64_t num_cells, int n, int max_points);
int pn_hard_voxelize_f32(const float *points, int n, int point_stride, int f, const float *voxel_size,
                         const float *range, int max_points, int max_voxels, float *voxels,
                         int32_t *coors, int32_t *num_points, int32_t *num_voxels, void *workspace,
                         size_t workspace_bytes, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * V3  per-voxel mean of point features.
 * Replaces torch_scatter.scatter_mean(features, unq_inv) in DynamicVoxelEncoderV1.forward
 *          det3d/models/readers/voxel_encoder.py:38-45
 * Sums are taken in 2^-24 fixed point (order independent => bitwise reproducible).
 * points (n x f, row stride point_stride) -> mean (V x f).
 */
int pn_scatter_mean_f32(const float *points, int point_stride, int f, const int32_t *voxel_start,
                        const int32_t *order, const int32_t *num_voxels, int v_capacity,
                        float *mean, pn_stream_t stream);

/* VoxelFeatureExtractorV3.forward  det3d/models/readers/voxel_encoder.py:15-22
 * voxels (V x P x F) , num_points int32[V] -> (V x F) = sum_P / num_points */
int pn_hard_voxel_mean_f32(const float *voxels, const int32_t *num_points, int v, int p, int f,
                           float *mean, pn_stream_t stream);

/* The collate of several samples' hard voxels (torch.cat of the per-sample lists, F.pad(coordinates, ((0, 0), (1, 0)), value = b),
 * det3d/torchie/parallel/collate.py:126-128, 157-164) on the device, counts included: feats [batch][seg_rows][c] and coors [batch][seg_rows][3]
 * hold every sample's list at its capacity, counts[batch] the live rows of each; the live rows are packed in sample order into
 * feats_out / coords4_out ([b, z, y, x] rows), *total = their sum.  Rows past *total are left untouched. */
int pn_concat_voxel_segments_f32(const float *feats, const int32_t *coors, const int32_t *counts, int batch,
                                 int seg_rows, int c, float *feats_out, int32_t *coords4_out,
                                 int32_t *total, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * V4 (+V5)  DynamicPFNet forward, cylinder grid, full decoration (16 channels):
 *   [points(7) | xyz-mean_voxel(3) | x-xc, y-yc | (rho,phi)-mean_voxel(2) | rho-rc, phi-pc]
 *   -> Linear(16->C0, no bias) ReLU segmax, concat -> Linear(2*C0->C1, no bias) ReLU segmax
 * Replaces DynamicPFNet.forward / feature_deco / PFNLayer.forward_dynamic / get_cluster /
 *          polar2cart   det3d/models/readers/pillar_encoder.py:393-406,338-391,63-71,228-249
 * and, when `canvas` != NULL, DynamicPPScatter.forward  pillar_encoder.py:418-432
 *   w0: (C0 x 16) row-major (torch Linear weight), w1: (C1 x 2*C0).  C0 <= 64, C1 <= 128,
 *   both multiples of 4.
 *   vx, vy, x_offset, y_offset: as computed by DynamicPFNet.__init__ (pillar_encoder.py:330-334)
 *   unq_keys: uint32[V] linear key per voxel rank (workspace product of pn_unique_rank_bitmap,
 *             see pn_unique_keys_ptr) -- gives (b,theta,r) of each voxel.
 *   features: (v_capacity x C1) or NULL.   canvas: NHWC (batch, T, R, C1) or NULL; the caller
 *   zero-fills the canvas (pn_fill_zero) -- only occupied cells are written here.
 */
int pn_dynamic_pfn_fwd(const float *points, int point_stride, const int32_t *voxel_start,
                       const int32_t *order, const int32_t *num_voxels, int v_capacity,
                       const uint32_t *unq_keys, const int32_t *grid, const float *w0, int c0,
                       const float *w1, int c1, float vx, float vy, float x_offset, float y_offset,
                       float *features, float *canvas, pn_stream_t stream);

/* cos/sin table of the pillar-centre azimuths (2*T floats) and the table-driven variant of the call
 * above (register-resident weights for the reference's C0 = 32, C1 = 128 reader) */
size_t pn_pfn_center_table_floats(int t);
int pn_pfn_center_table_f32(int t, float vy, float y_offset, float *table, pn_stream_t stream);
int pn_dynamic_pfn_fwd_table(const float *points, int point_stride, const int32_t *voxel_start,
                             const int32_t *order, const int32_t *num_voxels, int v_capacity,
                             const uint32_t *unq_keys, const int32_t *grid, const float *w0, int c0,
                             const float *w1, int c1, float vx, float vy, float x_offset,
                             float y_offset, const float *center_table, float *features, float *canvas,
                             pn_stream_t stream);
/* The same launch, which also zeroes cell_count[key] (the per-cell counters of pn_voxel_index_fused_*) for the frame's voxels (r6): the index
 * launches in front are done with them, so a frame engine that leaves its canvas dirty needs no pn_clear_frame_cells launch.  (c0, c1) = (32, 128). */
int pn_dynamic_pfn_fwd_table_clear(const float *points, int point_stride, const int32_t *voxel_start,
                                   const int32_t *order, const int32_t *num_voxels, int v_capacity,
                                   const uint32_t *unq_keys, const int32_t *grid, const float *w0, int c0,
                                   const float *w1, int c1, float vx, float vy, float x_offset,
                                   float y_offset, const float *center_table, float *features,
                                   float *canvas, uint32_t *cell_count, pn_stream_t stream);

/* pointer to the uint32 key-per-voxel array inside a pn_unique_rank_bitmap workspace */
const uint32_t *pn_unique_keys_ptr(const void *workspace, uint64_t num_cells, int n_capacity);

/* Static (hard-voxel) pillar feature net, eval mode: PillarFeatureNet.forward + PFNLayer.forward_static
 * (det3d/models/readers/pillar_encoder.py:74-169, 47-60).  voxels (V,P,F), num_points (V), coors (V,4) int32 [b,z,y,x];
 * decoration = [features, xyz - mean, xy - pillar centre (, |xyz|)], padded slots zeroed; per layer
 * Linear(no bias) -> BatchNorm1d (folded: scale, shift) -> ReLU -> max over the P slots (padded slots included, as in
 * the reference).  c1 = 0: single layer (features (V,c0)); else two layers with the [x, max] concatenation
 * (features (V,c1)).  The pillar count is read from the device. */
int pn_static_pfn_fwd(const float *voxels, const int32_t *num_points, const int32_t *coors,
                      const int32_t *num_voxels, int v_capacity, int p, int f, int with_distance,
                      const float *w0, const float *scale0, const float *shift0, int c0, const float *w1,
                      const float *scale1, const float *shift1, int c1, float vx, float vy,
                      float x_offset, float y_offset, float *features, pn_stream_t stream);

/* Backward of the pillar feature net, (C0, C1) = (32, 128) or (16, 32): gradients of the two Linear weights
 * (autograd through pillar_encoder.py:393-406 / 63-71; the points are data, no gradient).
 * The incoming gradient is either d_features (V,128) or the dense canvas gradient d_canvas
 * (B,T,R,C1) NHWC, of which only the occupied cells are read (= backward of DynamicPPScatter,
 * pillar_encoder.py:418-432, fused).  dw0 (32,16), dw1 (128,64) in torch layout.  The maximum's
 * gradient goes to the first point attaining it (torch_scatter.scatter_max semantics).
 * Deterministic: per-wave partial sums, added in wave order. */
size_t pn_dynamic_pfn_bwd_workspace_bytes(void);
int pn_dynamic_pfn_bwd(const float *points, int point_stride, const int32_t *voxel_start,
                       const int32_t *order, const int32_t *num_voxels, int v_capacity,
                       const uint32_t *unq_keys, const int32_t *grid, const float *w0, int c0,
                       const float *w1, int c1, float vx, float vy, float x_offset, float y_offset,
                       const float *center_table, const float *d_features, const float *d_canvas,
                       float *dw0, float *dw1, int accumulate, void *workspace,
                       size_t workspace_bytes, pn_stream_t stream);

/* V5 alone: DynamicPPScatter.forward (pillar_encoder.py:418-432) / PointPillarsScatter
 * features (V x C), unq int64 (V x 4) -> canvas NHWC (batch,T,R,C); caller zero-fills. */
int pn_scatter_canvas_fwd(const float *features, const int64_t *unq, const int32_t *num_voxels,
                          int v_capacity, int c, int t, int r, float *canvas, pn_stream_t stream);

int pn_fill_zero(void *ptr, size_t bytes, pn_stream_t stream);
/* Sparse clear: zero the canvas cells of the voxels in unq_keys (the cells pn_dynamic_pfn_fwd wrote).
 * A frame engine that owns a persistent canvas (zero between frames) calls this after the first
 * convolution has consumed the canvas instead of re-filling the whole map (134 MB for the
 * nuScenes grid) before every frame.  grid = {R, T, Z} as for the PFN. */
int pn_clear_canvas_cells(const uint32_t *unq_keys, const int32_t *num_voxels, int v_capacity,
                          const int32_t *grid, int c, float *canvas, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * B1 / H*  2-D convolution on the BEV map as an implicit GEMM on fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), fused per-output-channel affine (folded BatchNorm or bias) and
 * activation:  out = act(conv(in, w) * scale[n] + shift[n])
 * Replaces nn.Conv2d(+ZeroPad2d)+BatchNorm2d(eval)+ReLU triples of RPN._make_layer and the
 * deblocks  det3d/models/necks/rpn.py:80-142, and the head convolutions
 *          det3d/models/bbox_heads/center_head_parallel.py:120-177, center_head.py:65-109
 *
 * Weights are used in a packed layout produced once by pn_pack_conv_weight_f32 from the
 * torch layout (Cout, Cin/groups, KH, KW).
 */
typedef struct pn_conv_desc {
  int32_t batch, in_h, in_w;
  int32_t cin;          /* input channels per group  */
  int32_t cout;         /* output channels per group */
  int32_t groups;
  int32_t kh, kw, stride, pad_h, pad_w;
  int32_t in_pixel_stride, in_channel_offset;   /* NHWC slice addressing (floats) */
  int32_t out_pixel_stride, out_channel_offset;
  int32_t act;          /* PN_ACT_*  */
  int32_t deconv2x2;    /* 1: weights are a ConvTranspose2d(k=2,s=2) packed by
                           pn_pack_deconv2x2_weight_f32; kh=kw=1; output is (2H x 2W) */
  int32_t range_strata; /* >1: RangeStratified convolution (center_head_parallel.py:27-59):
                           the x (range) axis is cut into `range_strata` equal windows, window s
                           uses weight group s (packed with groups = range_strata) and
                           scale/shift[s*cout + n]; every window reads the same `cin` input
                           channels, halo columns come from the neighbouring windows (zeros at
                           the map border).  groups must be 1.  0/1: ordinary convolution. */
  int32_t pad_h_end, pad_w_end; /* extra zero rows / columns after the map, on top of pad_h / pad_w
                                   (asymmetric nn.ZeroPad2d); 0 = symmetric padding */
  int32_t accumulate;           /* 1: out += result (after scale/shift/act); used to sum the data
                                   gradients of branches that share an input */
  int32_t frames_in_flight;     /* hint, 0 / 1 = none: this launch belongs to one of N independent frames that run at the
                                   same time on other streams (several hipGraph engines).  A kernel may then take a form
                                   that does not fill the chip on its own but wastes less work (pn_conv2d_wino4_nhwc_f32:
                                   the plain F(4,3) form from ~96 tiles on instead of the K-split form).  Results differ
                                   from the unhinted launch only in the summation order of the form. */
  int32_t transpose_hw;         /* chained F(4,3) entries only (pn_conv*_wino4*_chain*, r4): 1 = work on the TRANSPOSE of the stored map -- the
                                   Winograd axis is the map's H axis (maps whose width / 4 is not a power of two but whose height / 4 is:
                                   the Waymo BEV maps, 256 x 144).  in_h / in_w and the strides keep describing the stored NHWC map; the
                                   packed weights must be those of the transposed kernel (w[.][.][kw][kh]). */
} pn_conv_desc;

size_t pn_conv_packed_weight_floats(int cout, int cin, int kh, int kw, int groups);
int pn_pack_conv_weight_f32(const float *w_oihw, int cout_total, int cin_per_group, int kh, int kw,
                            int groups, float *packed, pn_stream_t stream);
/* ConvTranspose2d weight (Cin, Cout, 2, 2) -> packed 1x1 weight with 4*Cout outputs */
size_t pn_deconv2x2_packed_weight_floats(int cin, int cout);
int pn_pack_deconv2x2_weight_f32(const float *w_iohw, int cin, int cout, float *packed,
                                 pn_stream_t stream);
/* scale/shift may be NULL (=> 1 / 0); they have groups*cout entries (cout for deconv) */
int pn_conv2d_nhwc_f32(const pn_conv_desc *desc, const float *in, const float *packed_w,
                       const float *scale, const float *shift, float *out, pn_stream_t stream);

/* bf16 variant of the same kernel (BASELINE configs[3], "bf16 BEV convs on MFMA"): bf16 NHWC activations
 * and bf16 packed weights, f32 accumulation on v_mfma_f32_32x32x16_bf16, scale / shift / activation in f32,
 * output bf16 (out_is_f32 = 0) or f32 (1, e.g. the last layer before an fp32 consumer).  The descriptor is the
 * same (strides / offsets in elements); cin, the input pixel stride and channel offset must be multiples of 8.
 * A ConvTranspose2d(k=2,s=2) weight (Cin,Cout,2,2) is packed as the 1x1 convolution weight
 * (4*Cout, Cin, 1, 1) with row (2*di+dj)*Cout + n and run with deconv2x2 = 1.  Round-to-nearest-even
 * conversions: pn_f32_to_bf16 / pn_bf16_to_f32. */
size_t pn_conv_packed_weight_bf16_elems(int cout, int cin, int kh, int kw, int groups);
int pn_pack_conv_weight_bf16(const float *w_oihw, int cout_total, int cin_per_group, int kh, int kw,
                             int groups, void *packed, pn_stream_t stream);
int pn_conv2d_nhwc_bf16(const pn_conv_desc *desc, const void *in_bf16, const void *packed_w_bf16,
                        const float *scale, const float *shift, void *out, int out_is_f32,
                        pn_stream_t stream);

/* r5: the bf16 BEV convolutions as a kernel of their own (csrc/conv_bf16.hip; replaces for the bf16 option the cuDNN convolutions of
 * det3d/models/necks/rpn.py:80-110,124-142 and of det3d/models/bbox_heads/e2e_swv_head.py:57-118): implicit GEMM on
 * v_mfma_f32_16x16x32_bf16, both operands global -> LDS by LDS-DMA, LDS-staged whole-line stores; 3x3 / stride 1 layers on maps whose rows
 * tile 288 pixels keep the input patch of a 64-channel chunk in LDS and read it for all nine taps (the "rows" form).
 * Layers it takes (pn_conv2d_igemm_bf16_supported): one group, no range strata, cin a multiple of 64, cout of 16, activation none / ReLU,
 * pixel strides / channel offsets multiples of 8 (in) and 4 (out); ConvTranspose2d(k = s = 2) as deconv2x2 with kh = kw = 1 and the weight
 * rows ordered (2 di + dj) * cout + n.  Weights: pn_pack_conv_weight_bf16_rows -> bf16 [rows padded to 16][kh * kw][cin].
 * Same descriptor, operands, scale / shift / act and output conventions as pn_conv2d_nhwc_bf16. */
size_t pn_conv_bf16_rows_packed_elems(int cout_total, int cin, int kh, int kw);
int pn_pack_conv_weight_bf16_rows(const float *w_oihw, int cout_total, int cin, int kh, int kw, void *packed, pn_stream_t stream);
int pn_conv2d_igemm_bf16_supported(const pn_conv_desc *desc);
int pn_conv2d_igemm_bf16(const pn_conv_desc *desc, const void *in_bf16, const void *packed_rows_bf16, const float *scale,
                         const float *shift, void *out, int out_is_f32, pn_stream_t stream);
/* nn.Linear on the same kernel (bf16 option of the SetBlock / Swin token GEMMs, det3d/models/utils/set_transformer.py:37-53,118-166 and
 * swin_utils/sw2votev4_util.py qkv / proj / Mlp): out[m][:n] = act(x[m][:k] @ W^T + bias) (+ residual[m][:n]).  x: bf16 rows of ldx
 * elements; W: pn_pack_conv_weight_bf16_rows(w (n, k), n, k, 1, 1); k a multiple of 64, n of 16; act PN_ACT_NONE | RELU | GELU (exact
 * erf); residual: f32, only with the f32 output.  out: bf16 or f32 rows of ldo elements. */
int pn_linear_bf16(const void *x_bf16, int m, int k, int ldx, const void *packed_rows_bf16, int n, const float *bias, int act,
                   const float *residual, int ldr, void *out, int ldo, int out_is_f32, pn_stream_t stream);
int pn_f32_to_bf16(const float *x, void *y, size_t n, pn_stream_t stream);
int pn_bf16_to_f32(const void *x, float *y, size_t n, pn_stream_t stream);

/* plain direct convolution (any channel count, no MFMA): used for constant folding of the
 * position-conditioned calibration (center_head_parallel.py:243-266) and as an on-device
 * cross-check of the MFMA kernel.  Weights in torch layout (Cout, Cin/groups, KH, KW). */
int pn_conv2d_direct_nhwc_f32(const pn_conv_desc *desc, const float *in, const float *w_oihw,
                              const float *scale, const float *shift, float *out,
                              pn_stream_t stream);

/* fold eval-mode BatchNorm (and an optional conv bias) into scale/shift:
 * scale = gamma/sqrt(var+eps), shift = beta + (bias-mean)*scale     (rpn.py:128-140, eps 1e-3) */
int pn_fold_bn_f32(const float *gamma, const float *beta, const float *mean, const float *var,
                   const float *conv_bias, float eps, int c, float *scale, float *shift,
                   pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Backward of the convolutions (training step: loss.backward() through rpn.py:124-159 and
 * center_head_parallel.py:120-196; torch autograd's conv2d backward in the reference).
 *
 * Weight gradient: dW = sum over output pixels of in[pixel + tap] (x) dout[pixel], written in
 * torch layout (Cout, Cin, KH, KW); `desc` is the FORWARD descriptor (out_pixel_stride /
 * out_channel_offset address `dout`; act is ignored).  accumulate != 0 adds to dweight.
 * Deterministic: pixel slices are reduced in a fixed order, no float atomics.
 * A ConvTranspose2d(k=2,s=2) weight gradient is the weight gradient of the 2x2 / stride-2
 * convolution that maps its OUTPUT gradient to its INPUT (swap the roles of the two tensors).
 * desc->range_strata > 1 (r4; RangeStratified, center_head_parallel.py:27-59: stride 1, width a multiple of the strata, cin and
 * cout <= 64 per stratum): `dout` is the (B, H, W, cout) gradient, dweight (strata * cout, cin, KH, KW); stratum s sums the
 * pixels of its own column band (the input's halo columns come from the neighbouring bands), one GEMM per stratum in one launch.
 */
size_t pn_conv2d_wgrad_workspace_bytes(const pn_conv_desc *desc);
int pn_conv2d_wgrad_f32(const pn_conv_desc *desc, const float *in, const float *dout, float *dweight,
                        int accumulate, void *workspace, size_t workspace_bytes,
                        pn_stream_t stream);
/* per-channel sum over pixels (bias gradient = sum of dout): out[c] (+)= sum_p x[p*stride+off+c] */
size_t pn_channel_sum_workspace_bytes(int c);
int pn_channel_sum_f32(const float *x, long long pixels, int pixel_stride, int channel_offset, int c,
                       float *out, int accumulate, void *workspace, size_t workspace_bytes,
                       pn_stream_t stream);
/* Data gradient = pn_conv2d_nhwc_f32 on dout with re-packed weights:
 *   stride 1:  pack with pn_pack_conv_dgrad_weight_f32 (taps mirrored, Cin/Cout swapped;
 *              pn_conv_packed_weight_floats(cin, cout, kh, kw, 1) floats), run a kh x kw
 *              convolution with pad' = k-1-pad, cin' = Cout, cout' = Cin.
 *   3x3 stride 2 pad 1 (ZeroPad2d(1)+Conv2d(3,2), rpn.py:126-134): pack with
 *              pn_pack_conv_dgrad_s2_weight_f32, run with deconv2x2 = 1, kh = kw = 2,
 *              pad_h_end = pad_w_end = 1, cin' = Cout, cout' = Cin on the (OH, OW) map of dout;
 *              the output is the (2*OH, 2*OW) input gradient.
 *   2x2 stride 2 pad 0: the forward weight (Cout, Cin, 2, 2) IS a ConvTranspose2d weight for the
 *              data gradient: pn_pack_deconv2x2_weight_f32(w, Cout, Cin).
 */
int pn_pack_conv_dgrad_weight_f32(const float *w_oihw, int cout, int cin, int kh, int kw,
                                  float *packed, pn_stream_t stream);
size_t pn_conv_dgrad_s2_packed_weight_floats(int cout, int cin);
int pn_pack_conv_dgrad_s2_weight_f32(const float *w_oihw, int cout, int cin, float *packed,
                                     pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * GroupNorm family on NHWC maps: statistics over {channel block} x {all theta} x {range
 * stratum}; covers
 *   RSNorm(num_heads, num_groups, C)      det3d/models/utils/norm.py:58-75
 *   nn.GroupNorm(G, C)                    center_head_parallel.py:148,160,174
 *   the GroupNorm inside RangeStratified  center_head_parallel.py:36-41
 * channel_groups: number of contiguous channel blocks; range_strata: number of equal
 * slices of the x (range) axis.  gamma/beta are indexed [stratum*C + c] (the stacked-channel
 * order the reference builds with torch.cat) and have range_strata*C entries.
 * out = act(norm(x)*gamma+beta); if mul/add != NULL (1,H,W,C maps) a second output
 * out2 = out*mul + add is written (feature undistortion, center_head_parallel.py:268).
 */
size_t pn_groupnorm_workspace_bytes(int batch, int channel_groups, int range_strata);
int pn_groupnorm_strat_fwd(const float *x, int batch, int h, int w, int c, int pixel_stride,
                           int channel_offset, int channel_groups, int range_strata,
                           const float *gamma, const float *beta, float eps, int act, float *out,
                           int out_pixel_stride, int out_channel_offset, const float *mul,
                           const float *add, float *out2, void *workspace, size_t workspace_bytes,
                           pn_stream_t stream);
/* r4 (training): the same call that also leaves mean_rstd [batch][range_strata][channel_groups][2] with the caller (nullable), for
 * pn_groupnorm_strat_bwd_stat -- the backward then starts from the forward's statistics instead of two more passes over x. */
int pn_groupnorm_strat_fwd_stat(const float *x, int batch, int h, int w, int c, int pixel_stride,
                                int channel_offset, int channel_groups, int range_strata,
                                const float *gamma, const float *beta, float eps, int act, float *out,
                                int out_pixel_stride, int out_channel_offset, const float *mul,
                                const float *add, float *out2, float *mean_rstd_out, void *workspace,
                                size_t workspace_bytes, pn_stream_t stream);
/* pn_groupnorm_strat_fwd whose result leaves as the F(4, 3) planes of conv_wchain.hip instead of the map (planes2, nullable: the planes of
 * out * mul + add); transpose_hw: the planes of the transposed map.  Each buffer: pn_wino4_planes_floats(batch, h, w, c) floats (transposed:
 * (batch, w, h, c)).  norm.py:58-75 + the input transform of the convolutions that follow. */
int pn_groupnorm_strat_planes_f32(const float *x, int batch, int h, int w, int c, int pixel_stride, int channel_offset, int channel_groups,
                                  int range_strata, const float *gamma, const float *beta, float eps, int act, const float *mul,
                                  const float *add, int transpose_hw, float *planes, float *planes2, void *workspace,
                                  size_t workspace_bytes, pn_stream_t stream);
/* the normalisation pass of pn_groupnorm_strat_fwd alone: mean_rstd [batch][range_strata][channel_groups][2] comes from the
 * producing convolution's epilogue (pn_conv2d_multi_f32 stat_mean_rstd); out2 = out*mul + add goes to its own
 * pixel stride / channel offset (e.g. the second half of a concatenated map) */
int pn_groupnorm_apply_f32(const float *x, int batch, int h, int w, int c, int pixel_stride,
                           int channel_offset, int channel_groups, int range_strata,
                           const float *mean_rstd, const float *gamma, const float *beta, int act,
                           float *out, int out_pixel_stride, int out_channel_offset, const float *mul,
                           const float *add, float *out2, int out2_pixel_stride,
                           int out2_channel_offset, pn_stream_t stream);

/* Backward of pn_groupnorm_strat_fwd (autograd through RSNorm / GroupNorm + ReLU and the
 * calibration x*W(pos)+b(pos), center_head_parallel.py:148-176,268).  dout: gradient of `out`;
 * dout2 (nullable, pixel stride c): gradient of `out2`, then mul is required and dmul / dadd
 * ((1,H,W,c) maps) receive the gradients of the calibration maps.  dgamma / dbeta have
 * range_strata*c entries (nullable).  dx may alias dout.  act: PN_ACT_NONE or PN_ACT_RELU.
 * The statistics are recomputed from x.  accumulate != 0 adds to dgamma/dbeta/dmul/dadd. */
size_t pn_groupnorm_bwd_workspace_bytes(int batch, int c, int channel_groups, int range_strata);
int pn_groupnorm_strat_bwd(const float *x, const float *dout, const float *dout2, const float *mul,
                           int batch, int h, int w, int c, int pixel_stride, int channel_offset,
                           int dout_pixel_stride, int dout_channel_offset, int channel_groups,
                           int range_strata, const float *gamma, const float *beta, float eps,
                           int act, float *dx, int dx_pixel_stride, int dx_channel_offset,
                           float *dgamma, float *dbeta, float *dmul, float *dadd, int accumulate,
                           void *workspace, size_t workspace_bytes, pn_stream_t stream);
/* mean_rstd: the statistics pn_groupnorm_strat_fwd_stat kept (NULL: recomputed, as pn_groupnorm_strat_bwd) */
int pn_groupnorm_strat_bwd_stat(const float *x, const float *dout, const float *dout2, const float *mul,
                                int batch, int h, int w, int c, int pixel_stride, int channel_offset,
                                int dout_pixel_stride, int dout_channel_offset, int channel_groups,
                                int range_strata, const float *gamma, const float *beta, float eps,
                                int act, float *dx, int dx_pixel_stride, int dx_channel_offset,
                                float *dgamma, float *dbeta, float *dmul, float *dadd, int accumulate,
                                const float *mean_rstd, void *workspace, size_t workspace_bytes,
                                pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm2d in training mode + the activation that follows it (rpn.py:128-140 under
 * model.train(); torch.nn.functional.batch_norm(training=True) in the reference), on an NHWC
 * channel slice of `pixels` = B*H*W pixels.  Batch statistics are biased; running_var receives the
 * unbiased one, running = (1-momentum)*running + momentum*batch (torch semantics; NULL = no
 * update).  saved_stat (2*c floats: mean, rstd per channel) is what the backward needs.
 * Backward: g = dout * act'(y);  dgamma = sum g*xhat;  dbeta = sum g;
 *           dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)).   dx may alias dout.
 * Deterministic (slice partials in fp64, summed in slice order).
 */
size_t pn_batchnorm_workspace_bytes(int c);
int pn_batchnorm_train_fwd(const float *x, long long pixels, int c, int pixel_stride,
                           int channel_offset, const float *gamma, const float *beta, float eps,
                           float momentum, int act, float *running_mean, float *running_var,
                           float *out, int out_pixel_stride, int out_channel_offset,
                           float *saved_stat, void *workspace, size_t workspace_bytes,
                           pn_stream_t stream);
int pn_batchnorm_bwd(const float *x, const float *dout, long long pixels, int c, int pixel_stride,
                     int channel_offset, int dout_pixel_stride, int dout_channel_offset,
                     const float *gamma, const float *beta, int act, const float *saved_stat,
                     float *dx, int dx_pixel_stride, int dx_channel_offset, float *dgamma,
                     float *dbeta, int accumulate, void *workspace, size_t workspace_bytes,
                     pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Dense linear layer on the same MFMA kernel (a 1x1 convolution over a token "image"):
 *   out[m][:n] = act(x[m][:k] @ W^T + bias) (+ residual[m][:n])
 * Replaces nn.Linear / Mlp / proj of the attention block  det3d/models/utils/set_transformer.py:37-53
 * packed_w: pn_pack_conv_weight_f32 of the (n, k) torch weight seen as (n, k, 1, 1).
 */
int pn_gemm_bias_act_f32(const float *x, int m, int k, int ldx, const float *packed_w, int n,
                         const float *bias, int act, const float *residual, int ldr, float *out,
                         int ldo, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The token GEMM of r3 (csrc/linear.hip): the same contract as pn_gemm_bias_act_f32 on a kernel built for the token matrices of
 * the attention block and the geometry-aware head (K = 256..1024, 1k..74k rows): persistent XCD-local tiles, the weight operand
 * straight from L2 in MFMA fragment layout, tile shape picked per problem, a K-split form for the 1k-row key-point chains.
 * Replaces nn.Linear of SetBlock / SetAttention / Mlp  det3d/models/utils/set_transformer.py:37-53,118-166,216-259 and of the Swin
 * stage  det3d/models/bbox_heads/swin_utils/sw2votev4_util.py (qkv / proj / Mlp / PatchEmbed.proj).
 *   packed_w: pn_pack_linear_weight_f32 of the (n, k) torch weight -> [ceil(k/32)*8][ceil(n/128)*128][4] floats, zero padded
 *   n, ldo, ldr multiples of 4; x / out / residual / bias 16-byte aligned; act: PN_ACT_NONE | RELU | GELU (exact erf)
 */
size_t pn_linear_packed_weight_floats(int n, int k);
int pn_pack_linear_weight_f32(const float *w_nk, int n, int k, float *packed, pn_stream_t stream);
int pn_linear_f32(const float *x, int m, int k, int ldx, const float *packed_w, int n, const float *bias, int act,
                  const float *residual, int ldr, float *out, int ldo, pn_stream_t stream);
/* The same contract on the K-split form (32 x 32 tiles, the block's four waves split K, partial tiles joined in LDS in a fixed order):
 * for matrices of a few thousand rows -- the key-point chains of the SetBlock, set_transformer.py:307-354 on 4 x 256 key points per
 * sample -- where tiles would leave most of the chip idle on a serial walk over K.  Its fp32 summation order differs from
 * pn_linear_f32's (which adds in the order of the r2 convolution route, bit for bit), hence the separate entry: a caller's results
 * never depend on a form picked behind its back. */
int pn_linear_ksplit_f32(const float *x, int m, int k, int ldx, const float *packed_w, int n, const float *bias, int act,
                         const float *residual, int ldr, float *out, int ldo, pn_stream_t stream);
/* ---------------------------------------------------------------------------------------
 * Gradient exchange of the DDP training step for hosts that do not go through torch.distributed (r6; SURVEY 8(b) "optional"): a thin wrapper
 * over RCCL.  Replaces what the reference's trainer gets from torch.nn.parallel.DistributedDataParallel (det3d/torchie/apis/train.py:325-336:
 * one process per GPU, NCCL backend; coalesced all-reduce det3d/core/utils/dist_utils.py:8-57) and the parameter broadcast of its construction.
 * One communicator per process on the CURRENT device; collectives are enqueued on the caller's HIP stream, nothing synchronises the host.
 * RCCL is bound at run time (dlopen on the first call): the library loads without it and these entry points then fail with PN_ERR_INVALID.
 *   pn_comm_unique_id: rank 0 draws the 128-byte id (pn_comm_unique_id_bytes) and hands it to the other ranks by the host's side channel
 *   pn_allreduce_f32: out[i] = sum (op 0) / max (op 1) over the ranks of in[i]; in == out allowed.  pn_broadcast_f32: rank root's buffer to all. */
size_t pn_comm_unique_id_bytes(void);
int pn_comm_unique_id(void *id_out);
int pn_comm_create(const void *id, int rank, int world, void **comm_out);
int pn_comm_destroy(void *comm);
int pn_allreduce_f32(void *comm, const float *in, float *out, size_t count, int op, pn_stream_t stream);
int pn_broadcast_f32(void *comm, float *buf, size_t count, int root, pn_stream_t stream);
/* pn_linear_f32 with a LayerNorm folded around it (r6): the norm -> Linear pairs of the attention block and of the Swin stage
 * (set_transformer.py:160-165 `x + mlp(norm2(x))`, sw2votev4_util.py:127-188 `attn(norm1(x))`, `mlp(norm2(x))`) without the normalised
 * tokens ever being written.  LayerNorm(x) W^T + b = rstd (x (W gamma)^T - mean colsum) + (b + W beta), so the normalisation is an affine
 * map of the GEMM's own accumulators once (mean, rstd) of every input row are known:
 *   producer  (row_stats_out != NULL, n % 32 == 0, no GELU): besides its output the launch leaves, per output row and 32-column group,
 *             (sum, sum of squares) of the values it stores (after activation and residual) in row_stats_out [m][n / 32][2];
 *   consumer  (ln_stats != NULL, k % 64 == 0): x is the producer's output; ln_stats [m][k / 32][2] its statistics table; packed_w the
 *             packed (W gamma) (columns of W scaled by the LayerNorm weight), ln_colsum [n] its row sums over k, bias = b + W beta;
 *             the epilogue forms mean / rstd = 1 / sqrt(var + ln_eps) per row (var = E[x^2] - mean^2, the partials added in group order)
 *             and applies them before activation and residual.
 * A launch is one or the other (or neither: then it is pn_linear_f32).  Tiled forms only. */
int pn_linear_ln_f32(const float *x, int m, int k, int ldx, const float *packed_w, int n, const float *bias, int act,
                     const float *residual, int ldr, float *out, int ldo, const float *ln_stats, const float *ln_colsum, float ln_eps,
                     float *row_stats_out, pn_stream_t stream);
/* tuning hook (tools/linear_bench.py): pins the tile form of every following pn_linear_f32 of the process -- 22 / 21 / 12 / 11 =
 * (64 TM) x (64 TN) block tiles, 1 = the K-split form, 0 = automatic (default; also PN_LINEAR_TILE in the environment) */
int pn_linear_set_tile(int form);

/* ---------------------------------------------------------------------------------------
 * A1  global representation re-alignment (SetBlock / SetAttention), non-GEMM parts.
 * Tokens are (B, H, W, C) fp32 in physical column order; `shift` (0 or win_w/2) is the azimuth
 * roll of the odd blocks, applied as an index mapping.  pos: (H, W, 2) Cartesian cell centres
 * (det3d/models/detectors/voxelnet.py:10-25).  pos_mlp: the folded relative-position MLP
 * [w1(16x2) | bn_scale(16) | bn_shift(16) | w2(heads x 16) | b2(heads)]
 * (Conv1d(2,16)+BatchNorm1d(eval)+ReLU+Conv1d(16,heads), set_transformer.py:96-100).
 */
/* nn.LayerNorm over C; chan_mean (rows) optional = mean_C(out)  (set_transformer.py:121,135) */
int pn_layernorm_f32(const float *x, size_t rows, int c, const float *gamma, const float *beta,
                     float eps, float *out, float *chan_mean, pn_stream_t stream);
/* the same rows with a bf16 copy (round to nearest even) for pn_linear_bf16; out_f32 may be NULL when only the copy is consumed */
int pn_layernorm_bf16out_f32(const float *x, size_t rows, int c, const float *gamma, const float *beta, float eps, float *out_f32,
                             void *out_bf16, float *chan_mean, pn_stream_t stream);
/* Token order of the three column-wise entries below: col_major = 0: range-major (B, H, W) tokens, the reference's order
 * (set_transformer.py:118-131); col_major = 1: azimuth-major (B, W, H) tokens -- the order the dense BEV map arrives in (NHWC, theta
 * outermost), every azimuth column one contiguous slab, no transpose around the blocks (voxelnet.py:211,219 permute instead). */
/* key-point selection: top-k local maxima of chan_mean along range per azimuth column
 * (set_transformer.py:134-147) -> top_idx int32 (B,k,W), kp (B, k*W, C), kpos (B,k,W,2) */
int pn_setblock_keypoints(const float *chan_mean, const float *xn, const float *pos, int batch, int h,
                          int w, int c, int k, int shift, int col_major, int32_t *top_idx, float *kp,
                          float *kpos, pn_stream_t stream);
/* SectorAttention core: key points attend to their column (set_transformer.py:307-354).
 * q_raw: proj_q(kp) (B, k*W, C) read through the reference's raw (B,C,k,W) view; kv: (tokens, 2C) */
int pn_setblock_sector_kp_attn(const float *q_raw, const float *kv, const float *xpos,
                               const float *kpos, const float *pos_mlp, int batch, int h, int w, int c,
                               int heads, int k, int shift, int col_major, float scale, float *out,
                               pn_stream_t stream);
/* RangeAttention core among key points, windows k x win_w (set_transformer.py:216-259);
 * qkv: (B, k*W, 3C) -> out (B, k*W, C) */
int pn_setblock_range_attn(const float *qkv, const float *kpos, const float *pos_mlp, int batch, int w,
                           int c, int heads, int k, int win_w, float scale, float *out,
                           pn_stream_t stream);
/* SectorAttentionV2 core: every column token attends to its k key points
 * (set_transformer.py:392-440); q: (tokens, C), kv_raw: (B, k*W, 2C) -> out (tokens, C) */
int pn_setblock_sector_col_attn(const float *q, const float *kv_raw, const float *xpos,
                                const float *kpos, const float *pos_mlp, int batch, int h, int w, int c,
                                int heads, int k, int shift, int col_major, float scale, float *out,
                                pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * H3  window attention of the geometry-aware head E2ESWVoteHead
 * (WindowAttention / SwinTransformerBlock, det3d/models/bbox_heads/swin_utils/sw2votev4_util.py:65-103,
 * 127-188; that file cannot run in the reference -- semantics = the repaired restatement in
 * oracle/polar_oracle.py, parity unpinned by the reference).
 * qkv: (B,H,W,3C) token map = Linear(C,3C) (bias included) of the LayerNorm-ed tokens; qkv_bias (3C,
 * nullable) is what the zero-padded tokens outside the map see; vote: (B,H,W,>=3) = (pred_centers, vote_cls);
 * pos: (H,W,2) Cartesian cell centres; vote MLP 3->16->C, rpe MLP 2->16->heads (1x1 conv weights),
 * tau (heads).  window = 7, head_dim = 64; shift = 0 or window/2.  out: (B,H,W,C) attention output
 * before the projection.  Zero padding to window multiples, cyclic shift, window partition, the shift
 * mask and their inverses are index arithmetic inside the kernel.
 * bias_table (nullable): the relative-position bias rpe(pos_i - pos_j) of every (window, head, query,
 * key), pn_swv_window_bias_floats(h, w, heads, window) floats written by pn_swv_window_bias_table
 * for the same pos / rpe weights / shift -- it depends on the model and the map only, so a caller
 * builds it once per set of weights; the rpe pointers may then be null.  Null: computed per call.
 * Both forms give the same bits. */
int pn_swv_window_attn(const float *qkv, const float *vote, int vote_pixel_stride, const float *pos,
                       const float *qkv_bias, const float *vote_w1, const float *vote_b1,
                       const float *vote_w2, const float *vote_b2, const float *rpe_w1,
                       const float *rpe_b1, const float *rpe_w2, const float *rpe_b2, const float *tau,
                       int batch, int h, int w, int c, int heads, int window, int shift,
                       const float *bias_table, float *out, pn_stream_t stream);
size_t pn_swv_window_bias_floats(int h, int w, int heads, int window);
int pn_swv_window_bias_table(const float *pos, const float *rpe_w1, const float *rpe_b1,
                             const float *rpe_w2, const float *rpe_b2, int h, int w, int heads,
                             int window, int shift, float *table, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * L1  CenterPoint loss, forward value.
 * Replaces CenterHead.loss / _sigmoid      det3d/models/bbox_heads/center_head.py:244-288
 *          FastFocalLoss, RegLoss           det3d/models/losses/centernet_loss.py:26-54, 6-24
 * hm_logits: NHWC head output (pre-sigmoid), hm_target: (B,classes,H,W) as in the reference example;
 * box sources: up to 5 NHWC head outputs concatenated in the reference order (reg,height,dim[,vel],rot);
 * ind/cat int64 (B,max_objs), mask uint8, anno_box (B,max_objs,anno_dim), anno_sel[box_dims] = the
 * anno_box column of every predicted box dimension (identity, or the vel-free selection
 * [0,1,2,3,4,5,-2,-1] of center_head.py:265).
 * out[4 + box_dims] = [det_loss, hm_loss, loc_loss, num_positive, loc_loss_elem...]
 */
size_t pn_center_loss_workspace_bytes(void);
int pn_center_loss_fwd(const float *hm_logits, int hm_pixel_stride, const float *hm_target, int batch,
                       int classes, int h, int w, const float *const *box_ptrs,
                       const int *box_pixel_strides, const int *box_channels, int n_box_src,
                       const int64_t *ind, const uint8_t *mask, const int64_t *cat,
                       const float *anno_box, int anno_dim, const int *anno_sel, int max_objs,
                       int box_dims, const float *code_weights, float weight, float *out,
                       void *workspace, size_t workspace_bytes, pn_stream_t stream);

/* Backward of pn_center_loss_fwd (loss.backward() of center_head.py:248-288): gradients with
 * respect to the hm logits and the box head outputs.  fwd_out is the `out` vector of the forward
 * call (num_positive is read from it on the device); grad_scale multiplies everything (1/world
 * for a mean over ranks, or 1).  d_hm (B,H,W,d_hm_pixel_stride) and d_box_ptrs[i]
 * (B,H,W,d_box_pixel_strides[i]) are fully overwritten (zeros where no gradient flows, including
 * pad channels), so they can feed the MFMA data/weight-gradient kernels directly.
 * Objects that share a cell are folded in object order: bitwise reproducible. */
int pn_center_loss_bwd(const float *hm_logits, int hm_pixel_stride, const float *hm_target, int batch,
                       int classes, int h, int w, const float *const *box_ptrs,
                       const int *box_pixel_strides, const int *box_channels, int n_box_src,
                       const int64_t *ind, const uint8_t *mask, const int64_t *cat,
                       const float *anno_box, int anno_dim, const int *anno_sel, int max_objs,
                       int box_dims, const float *code_weights, float weight, const float *fwd_out,
                       float grad_scale, float *d_hm, int d_hm_pixel_stride, float *const *d_box_ptrs,
                       const int *d_box_pixel_strides, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * T1  optimizer side of the training step, over the FLAT fp32 parameter / gradient buffers.
 * pn_grad_norm_f32: total L2 norm of the gradients into a device scalar (clip_grad_norm_,
 *   det3d/torchie/trainer/hooks/optimizer.py:10-13); deterministic fp64 two-stage sum.
 * pn_adam_step_f32: g *= min(1, max_norm/(total_norm+1e-6)) (skipped if total_norm == NULL);
 *   p *= 1 - weight_decay*lr (OptimWrapper.step, fastai_optim.py:155-171);  torch.optim.Adam update
 *   with betas (beta1, beta2), bias correction for the 1-based `step`.  lr and beta1 are the
 *   OneCycle values of this step (learning_schedules_fastai.py:77-95), computed by the host.
 */
size_t pn_grad_norm_workspace_bytes(void);
int pn_grad_norm_f32(const float *grads, size_t n, float *total_norm, void *workspace,
                     size_t workspace_bytes, pn_stream_t stream);
int pn_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, size_t n,
                     int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                     const float *total_norm, float max_norm, pn_stream_t stream);
/* element-wise helpers of the backward pass: dx = dy*(1-y^2);  out = a+b;  RangeStratified
 * gradient (B,H,W,c) -> (B,H,W,strata*c) with the stratum's block filled and zeros elsewhere, so
 * that its weight / data gradients are those of an ordinary convolution with strata*c outputs
 * (center_head_parallel.py:45-59). */
int pn_tanh_bwd_f32(const float *y, const float *dy, float *dx, size_t n, pn_stream_t stream);
/* dx = dy where the ReLU OUTPUT y is positive, else 0 (plain CenterHead: conv + ReLU chains, center_head.py:65-109) */
int pn_relu_bwd_f32(const float *y, const float *dy, float *dx, size_t n, pn_stream_t stream);
int pn_add_f32(const float *a, const float *b, float *out, size_t n, pn_stream_t stream);
int pn_strat_expand_f32(const float *dy, int batch, int h, int w, int c, int strata, float *out,
                        pn_stream_t stream);
/* r4: data gradient of the RangeStratified 3x3 convolution without the expansion.  With z (B,H,W,3c) = the stratified 3x1
 * convolution of dy whose weight set s holds, for width tap kx, rows [kx*c, (kx+1)*c) = W_s[:, :, 2-ky, kx]^T (pn_conv2d_nhwc_f32,
 * range_strata, kh 3, kw 1 -- the weights follow the stratum of the dy pixel, which is what the gradient needs),
 * dx[y, x] (+)= z[y, x+1, 0:c] + z[y, x, c:2c] + z[y, x-1, 2c:3c]: this call.  1/strata of the expansion's multiply-adds. */
int pn_strat_dgrad_combine_f32(const float *z, int batch, int h, int w, int c, float *dx, int dx_pixel_stride, int dx_channel_offset,
                               int accumulate, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * next-2  decode + rotated NMS: head tensors -> boxes, entirely on the device and on the caller's stream.
 * Replaces CenterHead.decode / post_processing (plain path)   det3d/models/bbox_heads/center_head.py:350-402, 462-577
 *          rotate_nms_pcdet                                   det3d/core/bbox/box_torch_ops.py:248-277
 *          nms_gpu: IoU bit masks + greedy reduce             det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-311,
 *                                                             iou3d_nms.cpp:90-136 (host loop after a blocking copy there)
 * Inputs are the raw NHWC head outputs (pixel strides in floats); vel may be NULL (7-dim boxes, else 9:
 * [x,y,z,dims(3),vel(2),rot]).  cylinder != 0: cell (y, x) has centre rho = x*step_x + origin_x,
 * azimuth = y*step_y + origin_y (step = out_size_factor * voxel_size); otherwise Cartesian cells.
 * post_center_range: 6 HOST floats.  Ties in the score order are broken by cell index.  At most 8192 cells
 * above the score threshold enter the sort (first in cell order).  nms_pre_max_size <= 4096.
 * Outputs per sample: out_count[b] boxes in out_boxes[b][:count] (post_max rows allocated), scores, labels
 * (int64), cells (flat y*W+x).  The rotated-IoU arithmetic has no runnable reference here (CUDA only):
 * parity is against oracle/box_nms.c, unpinned by the reference.
 * per_class_nms != 0: test_cfg.per_class_nms of the reference's nuScenes configs (center_head.py:514-518,
 * detectron2 layers.batched_nms_rotated, third party, absent here): one greedy pass in score order in which only
 * boxes of the SAME class suppress each other; the reference passes every candidate, here the first pre_max
 * (<= 4096) of the score order enter -- pass pre_max = 4096 for that mode. */
size_t pn_center_decode_nms_workspace_bytes(int batch, int cells, int box_dims, int pre_max,
                                            int post_max);
int pn_center_decode_nms_f32(const float *hm, int hm_pixel_stride, int classes, const float *reg,
                             int reg_pixel_stride, const float *height, int height_pixel_stride,
                             const float *dim, int dim_pixel_stride, const float *rot,
                             int rot_pixel_stride, const float *vel, int vel_pixel_stride, int batch,
                             int h, int w, int cylinder, float step_x, float step_y, float origin_x,
                             float origin_y, int rectify, float score_threshold,
                             const float *post_center_range, float nms_iou_threshold,
                             int per_class_nms, int pre_max, int post_max, float *out_boxes, float *out_scores, int64_t *out_labels,
                             int32_t *out_cells, int32_t *out_count, void *workspace,
                             size_t workspace_bytes, pn_stream_t stream);

/* Stateful NMS across azimuth sectors (test_cfg.stateful_nms, center_head.py:486-501, 507-509): the sector's thresholded candidates
 * are rotated by sector_angle into the sweep's frame BEFORE the NMS, the detections of the previous sectors (prev_* : (batch,
 * prev_capacity, box_dims) boxes / (batch, prev_capacity) scores, int64 labels / (batch) int32 counts; prev_capacity may be 0)
 * compete in the same NMS, and post_max is the caller's nms_post_max_size * (sec_id + 1).  out_cells >= h*w marks a carried-over
 * detection (row h*w + k of the previous list).  Workspace: pn_center_decode_nms_workspace_bytes(batch, h*w + prev_capacity, ...). */
int pn_center_decode_nms_stateful_f32(const float *hm, int hm_pixel_stride, int classes, const float *reg,
                                      int reg_pixel_stride, const float *height, int height_pixel_stride,
                                      const float *dim, int dim_pixel_stride, const float *rot, int rot_pixel_stride,
                                      const float *vel, int vel_pixel_stride, int batch, int h, int w, int cylinder,
                                      float step_x, float step_y, float origin_x, float origin_y, int rectify,
                                      float score_threshold, const float *post_center_range, float nms_iou_threshold,
                                      int per_class_nms, int pre_max, int post_max, double sector_angle,
                                      const float *prev_boxes, const float *prev_scores, const int64_t *prev_labels,
                                      const int32_t *prev_count, int prev_capacity, float *out_boxes, float *out_scores,
                                      int64_t *out_labels, int32_t *out_cells, int32_t *out_count, void *workspace,
                                      size_t workspace_bytes, pn_stream_t stream);
/* Double-flip test-time augmentation (CenterHead.double_flip_decode, center_head.py:289-346): the input batch is 4 * merged_batch
 * samples in groups [original, y -> -y, x -> -x, both]; copies 1..3 are flipped back along H / W / both, reg / rot (sin, cos) /
 * vel take the mirrored frame's signs, and the four are averaged: out_hm = mean sigmoid(hm) (probabilities), out_dim = mean
 * exp(dim) (sizes), the others plain means.  Outputs are contiguous (merged_batch, h, w, c).  The decode that follows is
 * pn_center_decode_nms_merged_f32 = pn_center_decode_nms_f32 without the sigmoid / exp (center_head.py:350-353 `if not
 * double_flip`). */
int pn_double_flip_merge_f32(const float *hm, int hm_pixel_stride, int classes, const float *reg, int reg_pixel_stride,
                             const float *height, int height_pixel_stride, const float *dim, int dim_pixel_stride,
                             const float *rot, int rot_pixel_stride, const float *vel, int vel_pixel_stride,
                             int merged_batch, int h, int w, float *out_hm, float *out_reg, float *out_height,
                             float *out_dim, float *out_rot, float *out_vel, pn_stream_t stream);
int pn_center_decode_nms_merged_f32(const float *hm_prob, int hm_pixel_stride, int classes, const float *reg,
                                    int reg_pixel_stride, const float *height, int height_pixel_stride,
                                    const float *dim_size, int dim_pixel_stride, const float *rot, int rot_pixel_stride,
                                    const float *vel, int vel_pixel_stride, int batch, int h, int w, int cylinder,
                                    float step_x, float step_y, float origin_x, float origin_y, int rectify,
                                    float score_threshold, const float *post_center_range, float nms_iou_threshold,
                                    int per_class_nms, int pre_max, int post_max, float *out_boxes, float *out_scores,
                                    int64_t *out_labels, int32_t *out_cells, int32_t *out_count, void *workspace,
                                    size_t workspace_bytes, pn_stream_t stream);

/* Same pipeline for the geometry-aware head: E2ESWVoteHead.decode / post_processing (e2e_swv_head.py:313-366, 368-470; code that
 * does not run in the reference, restated from its text): score = sigmoid(hm) * clamp((iou + 1) / 2, 0, 1)^iou_factor (iou may be
 * NULL), centre = reg + offset_grid (planar (2, H, W) Cartesian cell centres), rot = atan2(rot[1], rot[0]), and with rectify the
 * heading + atan2(y, x) wrapped into (-pi, pi].  7-dim boxes.  Workspace: pn_center_decode_nms_workspace_bytes(batch, h*w, 7, ..). */
int pn_swv_decode_nms_f32(const float *hm, int hm_pixel_stride, int classes, const float *reg,
                          int reg_pixel_stride, const float *height, int height_pixel_stride,
                          const float *dim, int dim_pixel_stride, const float *rot, int rot_pixel_stride,
                          const float *iou, int iou_pixel_stride, int iou_factor,
                          const float *offset_grid, int batch, int h, int w, int rectify,
                          float score_threshold, const float *post_center_range,
                          float nms_iou_threshold, int per_class_nms, int pre_max, int post_max,
                          float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_cells,
                          int32_t *out_count, void *workspace, size_t workspace_bytes,
                          pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Segmentation head SingleConvHead (det3d/models/seg_heads/seg_head.py:53-83, 176-195), the `seg` super-task of the
 * reference's nuScenes polar config; secondary to the detection hot path.  The head's 1x1 convolution over
 * cat[canvas, bilinear_up(x2)] is evaluated as conv(canvas) + bilinear_up(conv(x2)) (pn_conv2d_nhwc_f32 twice):
 *   pn_bilinear_upsample_add_f32   out (B,H,W,C) += bilinear(low (B,h,w,C)), F.interpolate(align_corners=False) weights
 *   pn_seg_point_labels            labels[i] = 1 + argmax_c seg[y_i, x_i, c] for one sample; grid_ind rows [z, y, x] (int64)
 */
int pn_bilinear_upsample_add_f32(const float *low, int batch, int h, int w, int c, int out_h, int out_w,
                                 float *out, pn_stream_t stream);
int pn_seg_point_labels(const float *seg_sample, int h, int w, int classes, const int64_t *grid_ind,
                        int n, int64_t *labels, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * next-1  sparse 3-D convolutions of the middle encoder SpMiddleResNetFHD
 * (det3d/models/backbones/scn.py:17-192; the arithmetic is the third-party spconv package there: SubMConv3d /
 * SparseConv3d / SparseConvTensor.dense -- parity unpinned, restated in oracle/polar_oracle.py).
 * A level's active set = bitmap over its (B,D,H,W) cells + per-word ranks ("index", pn_sparse_index_bytes);
 * features live in key order, key = ((b*D+z)*H+y)*W+x.  All counts stay on the device.
 *   pn_sparse_index_from_coords   coords (n,4) int32 [b,z,y,x] -> index, keys (rank order), count, and the rank of
 *                                 every input row (pn_sparse_permute_rows then puts the features in key order)
 *   pn_sparse_index_downsample    active output sites of a strided SparseConv3d (kernel/stride/pad per axis z,y,x)
 *   pn_sparse_neighbors           nbr[out site][tap] = input index or -1; tap = (kz*KH + ky)*KW + kx; one table
 *                                 serves every convolution that shares an indice_key
 *   pn_sparse_conv_f32            out = act((sum_t W_t in[nbr[.][t]]) * scale + shift + residual), on the MFMA
 *                                 kernel in gather mode; packed_w = pn_pack_conv_weight_f32 of (Cout, Cin, taps, 1)
 *   pn_sparse_conv_c16_f32        the same for the 16-channel level (conv_input and conv1's SparseBasicBlocks, scn.py:112-123: cin 8 or
 *                                 16, cout 16) on a VALU kernel: four lanes per site, every neighbour row fetched up front; same packed
 *                                 weights; summation order taps ascending, channels ascending (last-bit differences to the MFMA form,
 *                                 hence a separate entry: the caller names the form)
 *   pn_sparse_to_dense_nhwc       (B, H, W, C*D) with channel c*D + z  ==  .dense().view(N, C*D, H, W) in NHWC
 */
size_t pn_sparse_index_bytes(uint64_t num_cells);
int pn_sparse_index_from_coords(const int32_t *coords, int n_capacity, const int32_t *n_dev,
                                const int32_t *dims, void *index_buf, uint32_t *keys, int32_t *count,
                                int32_t *rank_of_input, pn_stream_t stream);
int pn_sparse_index_downsample(const uint32_t *in_keys, int in_capacity, const int32_t *n_in,
                               const int32_t *in_dims, const int32_t *kernel, const int32_t *stride,
                               const int32_t *pad, const int32_t *out_dims, void *out_index_buf,
                               uint32_t *out_keys, int out_capacity, int32_t *out_count,
                               pn_stream_t stream);
int pn_sparse_neighbors(const uint32_t *out_keys, int out_capacity, const int32_t *n_out,
                        const int32_t *out_dims, const void *in_index_buf, const int32_t *in_dims,
                        const int32_t *kernel, const int32_t *stride, const int32_t *pad, int32_t *nbr,
                        pn_stream_t stream);
int pn_sparse_permute_rows(const float *in, const int32_t *rank, int n_capacity, const int32_t *n_dev,
                           int c, float *out, pn_stream_t stream);
int pn_sparse_conv_f32(const float *in, int in_rows, int cin, const int32_t *nbr, const int32_t *n_out,
                       int out_capacity, int taps, const float *packed_w, int cout, const float *scale,
                       const float *shift, int act, const float *residual, float *out,
                       pn_stream_t stream);
int pn_sparse_conv_c16_f32(const float *in, int in_rows, int cin, const int32_t *nbr, const int32_t *n_out,
                           int out_capacity, int taps, const float *packed_w, const float *scale,
                           const float *shift, int act, const float *residual, float *out,
                           pn_stream_t stream);
/* The same convolution over GROUPS of 32 output sites with similar neighbourhoods (r4, the 32 / 64 / 128-channel levels and the strided
 * stages): pn_sparse_group_rows sorts every window of 4096 sites of a neighbour table by its neighbour mask and returns perm[out_capacity]
 * (the site of every slot, -1 past the live sites) and group_mask[ceil(out_capacity / 32)] (the union of the masks of slots 32 g .. 32 g + 31);
 * one table's groups serve every convolution that shares it (indice_key).  pn_sparse_conv_grouped_f32 then runs one wave per group over
 * the group's taps only (no tile-wide tap union: 0.73 x the MFMA work of pn_sparse_conv_f32 on a 64-beam sweep) and writes the rows to their
 * own places: same arguments and layout as pn_sparse_conv_f32, cin a multiple of 16; results equal up to the summation order (taps
 * ascending, channels ascending within a tap; ~2e-6 of the output's range).  scn.py:25-48,112-181. */
int pn_sparse_group_rows(const int32_t *nbr, const int32_t *n_out, int out_capacity, int taps, int32_t *perm, uint32_t *group_mask,
                         pn_stream_t stream);
/* r4: the neighbour kernel can leave one byte per (site, kz, ky) -- bit kx set when tap (kz, ky, kx) exists; [out_capacity][k0 k1] bytes --
 * and the grouping sort assembles a site's mask from those k0 k1 bytes instead of re-reading the table (442 KB per window through one CU).
 * Same table, same perm and group masks as pn_sparse_neighbors + pn_sparse_group_rows.  Slots of windows without a live site are left
 * undefined by both forms (the convolutions walk the groups below ceil(n / 32) only). */
int pn_sparse_neighbors_rows(const uint32_t *out_keys, int out_capacity, const int32_t *n_out, const int32_t *out_dims, const void *in_index_buf,
                             const int32_t *in_dims, const int32_t *kernel, const int32_t *stride, const int32_t *pad, int32_t *nbr,
                             uint8_t *row_bits, pn_stream_t stream);
int pn_sparse_group_rows_bits(const uint8_t *row_bits, int rows, int bits_per_row, const int32_t *n_out, int out_capacity, int32_t *perm,
                              uint32_t *group_mask, pn_stream_t stream);
/* r5: xcd_bounds (nullable, 18 int32 written by pn_sparse_group_balance from the same masks): cut points of the XCDs' contiguous runs of
 * groups chosen for equal WORK (taps of the groups' masks) instead of equal counts -- with equal counts one XCD carried 1.18 - 1.30 x the mean
 * on the bench frame and every layer waited for it.  Results do not depend on it. */
int pn_sparse_group_balance(const uint32_t *group_mask, const int32_t *n_out, int out_capacity, int32_t *xcd_bounds, pn_stream_t stream);
int pn_sparse_conv_grouped_f32(const float *in, int in_rows, int cin, const int32_t *nbr, const int32_t *n_out, int out_capacity, int taps,
                               const int32_t *perm, const uint32_t *group_mask, const int32_t *xcd_bounds, const float *packed_w, int cout,
                               const float *scale, const float *shift, int act, const float *residual, float *out, pn_stream_t stream);
int pn_sparse_to_dense_nhwc(const float *feats, const uint32_t *keys, int capacity,
                            const int32_t *n_dev, const int32_t *dims, int c, float *out,
                            pn_stream_t stream);
/* r4: the same map written from the output side -- per pixel the level's index (pn_sparse_index_*) says which cells are active; every
 * element is stored once in the map's own order, no zero fill and no scatter (c a multiple of 4).  scn.py:176-179. */
int pn_sparse_to_dense_index_nhwc(const float *feats, const void *index_buf, const int32_t *dims, int c, float *out, pn_stream_t stream);

/* Backward side of the sparse convolutions (scn.py:97-192 under autograd; spconv's own backward in the reference).
 *   pn_sparse_neighbors_transpose  inv[in row][tap] = the output row that reads it through that tap, or -1 (at most one
 *                                  exists): the data gradient is then pn_sparse_conv_f32 over `inv` with the (Cin, Cout)
 *                                  transposed weights -- a gather, no atomics
 *   pn_sparse_conv_wgrad_f32       dw[co][tap][ci] (+)= sum_i dout[i][co] * in[nbr[i][tap]][ci]  (spconv weight layout
 *                                  (Cout, kD, kH, kW, Cin)); `in_rows` x `cin` = the feature matrix `in` (cin a multiple of 4;
 *                                  in_rows * cin * 4 < 2 GiB: row offsets are 32-bit), cin_real <= cin the channels kept; the MFMA weight-gradient kernel of pn_conv2d_wgrad_f32 with the neighbour
 *                                  table in its loader, row slices reduced in slice order (no atomics)
 *   pn_sparse_from_dense_nhwc      gradient of pn_sparse_to_dense_nhwc: gathers (B, H, W, C*D) back to the active rows
 *   pn_add_relu_f32                out = max(a + b, 0)  (SparseBasicBlock's residual join, scn.py:84-95)
 */
int pn_sparse_neighbors_transpose(const int32_t *nbr, const int32_t *n_out, int out_capacity, int taps, int in_rows,
                                  int32_t *inv, pn_stream_t stream);
size_t pn_sparse_conv_wgrad_workspace_bytes(int out_capacity, int taps, int cout, int cin);
int pn_sparse_conv_wgrad_f32(const float *in, int in_rows, int cin, int cin_real, const float *dout, int cout, const int32_t *nbr,
                             const int32_t *n_out, int out_capacity, int taps, float *dw, int accumulate,
                             void *workspace, size_t workspace_bytes, pn_stream_t stream);
int pn_sparse_from_dense_nhwc(const float *dense, const uint32_t *keys, int capacity, const int32_t *n_dev,
                              const int32_t *dims, int c, float *feats, pn_stream_t stream);
int pn_add_relu_f32(const float *a, const float *b, float *out, size_t n, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * next-3  CenterPoint target assignment on the polar grid, on the device.
 * Replaces AssignLabel.assign_heatmap_polar   det3d/datasets/pipelines/preprocess.py:253-342
 *          gaussian_radius / draw_umich_gaussian det3d/core/utils/center_utils.py:18-64
 * gt_boxes (B, max_gt, box_cols >= 9) f32 [x,y,z,l,w,h,vx,vy,...,rot], gt_classes (B, max_gt) 1-based within the task,
 * num_gt (B) device counts.  Outputs as the reference's example dict: hm (B, classes, feature_a, feature_r),
 * ind / cat int64 and mask uint8 (B, max_objs), anno_box (B, max_objs, 10) = [dx, dy, z, log l, log w, log h, vx, vy,
 * sin rot, cos rot]; all fully overwritten.  As in the reference the footprint is rotated by COLUMN 6 of the box. */
size_t pn_assign_heatmap_workspace_bytes(int batch, int max_objs);
int pn_assign_heatmap_polar_f32(const float *gt_boxes, const int32_t *gt_classes, const int32_t *num_gt,
                                int batch, int max_gt, int box_cols, int max_objs, int classes,
                                int feature_r, int feature_a, float voxel_size_r, float voxel_size_a,
                                float range_r0, float range_a0, int out_size_factor,
                                float gaussian_overlap, int min_radius, int rectify, float *hm,
                                int64_t *ind, uint8_t *mask, int64_t *cat, float *anno_box,
                                void *workspace, size_t workspace_bytes, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * next-4 (device half)  multi-sweep accumulation for streaming inference (BASELINE configs[4]).
 * Replaces read_sweep / remove_close + the concatenation of LoadPointCloudFromFile
 *          det3d/datasets/pipelines/loading.py:62-84, 216-332
 * raw: the sweeps' points (n, in_cols >= 4) f32 [x,y,z,intensity,...] concatenated, key frame first; sweep_offsets
 * (sweeps+1) int32; transforms (sweeps,4,4) float64 row-major (entry 0 unused); time_lags (sweeps) f32.
 * Past sweeps lose the points with |x| < min_distance and |y| < min_distance (own frame), are moved into the key
 * frame and tagged with their time lag.  out (n,5) [x,y,z,intensity,dt], order preserved; out_count on the device. */
size_t pn_accumulate_sweeps_workspace_bytes(int n);
int pn_accumulate_sweeps_f32(const float *raw, int n, int in_cols, const int32_t *sweep_offsets,
                             int sweeps, const double *transforms, const float *time_lags,
                             float min_distance, float *out, int32_t *out_count, void *workspace,
                             size_t workspace_bytes, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Training side of the geometry-aware head E2ESWVoteHead (Waymo PARTNER config): targets, matcher cost, set criterion.
 * Replaces GroundTruthProcessor.process / draw_votemap (det3d/models/bbox_heads/e2e_modules.py:31-148, centernet_utils.py:68-88),
 * TimeMatcher's cost (det3d/models/e2e_utils/matcher.py:78-93, 136-147; the assignment itself is scipy on the host there,
 * pn_lsap_f32 here) and SetCriterion (det3d/models/e2e_utils/set_crit.py:68-206 with loss_utils.py:447-535, 583-594 and
 * iou3d_nms_utils.py:38-72).  Head tensors are NHWC views given as pointer + pixel stride; offset_grid is planar (2, H, W).
 */
/* example['global_box'] (batch, max_boxes, cols >= 8) rows [x, y, z, dx, dy, dz, heading, class], all-zero rows as padding ->
 * per sample the rows of the task's classes (class_ids, in that order): gt_boxes (batch, max_boxes, 7), gt_classes = position in
 * class_ids, gt_counts (batch) */
int pn_swv_gt_compact(const float *global_box, int batch, int max_boxes, int cols, const int32_t *class_ids,
                      int n_classes, float *gt_boxes, int32_t *gt_classes, int32_t *gt_counts,
                      pn_stream_t stream);
size_t pn_swv_votemap_workspace_bytes(int batch, int max_boxes, int h, int w);
/* votemap (batch, H = grid[1]/stride, W = grid[0]/stride, 4 + n_classes): [cx, cy, c_rho, c_phi] of the last object whose window
 * covers the cell + per class the maximum of the objects' Gaussians; vote_count (1): cells with a centre (votemap[..., 0] != 0) */
int pn_swv_draw_votemap_f32(const float *gt_boxes, const int32_t *gt_classes, const int32_t *gt_counts, int batch,
                            int max_boxes, int n_classes, const float *max_space, const float *min_space,
                            const int32_t *grid, int stride, int num_max_objs, double gaussian_overlap,
                            float *votemap, int32_t *vote_count, void *workspace, size_t workspace_bytes,
                            pn_stream_t stream);
/* cost (batch, rows, H*W): row g of sample b = -(sigmoid(hm[q, cls_g])^w_ce * exp(-L1(code_weights*(pred_q - encode(gt_g))))^w_bbox)
 * for g < gt_counts[b] (rows >= the largest count; other rows are not written) */
int pn_swv_match_cost_f32(const float *hm, int hm_ps, int ncls, const float *reg, int reg_ps, const float *height,
                          int height_ps, const float *dim, int dim_ps, const float *rot, int rot_ps,
                          const float *offset_grid, int batch, int h, int w, const float *gt_boxes,
                          const int32_t *gt_classes, const int32_t *gt_counts, int max_boxes, int rows, float w_ce,
                          float w_bbox, const float *code_weights, float *cost, pn_stream_t stream);
/* HOST function on host memory: rectangular linear sum assignment, cost (nr, nc) row major with nr <= nc; col_of_row[nr] */
int pn_lsap_f32(const float *cost, int nr, int nc, int32_t *col_of_row);
size_t pn_swv_criterion_workspace_bytes(int batch, int h, int w);
/* out[14] = [det_loss, loss_ce, loss_bbox, loss_vote, loss_vote_cls, loss_iou, loc_loss_elem x 8] (device);
 * loss_weights = [ce, bbox, vote, vote_cls, iou]; iou nullable (no IoU branch).  Gradients of det_loss w.r.t. the head tensors,
 * dense NHWC (all or none): d_hm (B,H,W,ncls), d_boxes (B,H,W,8: reg 2, height 1, dim 3, rot 2), d_centers (B,H,W,2),
 * d_vote_cls (B,H,W,ncls), d_iou (B,H,W,1). */
int pn_swv_set_criterion_f32(const float *hm, int hm_ps, int ncls, const float *reg, int reg_ps, const float *height,
                             int height_ps, const float *dim, int dim_ps, const float *rot, int rot_ps,
                             const float *iou, int iou_ps, const float *pred_centers, int centers_ps,
                             const float *pred_vote_cls, int vote_cls_ps, const float *offset_grid, int batch, int h,
                             int w, const float *votemap, const int32_t *vote_count, const int32_t *match_sample,
                             const int32_t *match_query, const int32_t *match_gt, int n_match, const float *gt_boxes,
                             const int32_t *gt_classes, int max_boxes, float num_boxes, const float *loss_weights,
                             float sigma, float gamma, float alpha, const float *code_weights, float *out, float *d_hm,
                             float *d_boxes, float *d_centers, float *d_vote_cls, float *d_iou, void *workspace,
                             size_t workspace_bytes, pn_stream_t stream);

/* boxes of sector `sec_id` back into the sweep's frame after the per-sector NMS (center_head.py:533-545): centre (and velocity)
 * rotated by +angle, heading -= angle; boxes (batch, capacity, box_dims) with counts (batch) valid rows each */
int pn_rotate_boxes_f32(float *boxes, const int32_t *counts, int batch, int capacity, int box_dims, double angle,
                        pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Sector streaming (a sweep processed as azimuth sectors, one after the other).
 */
size_t pn_split_polar_sectors_workspace_bytes(int n_capacity, int nsectors, int batch);
/* Voxelization.voxelize_streaming_polar (det3d/datasets/pipelines/voxelization.py:305-393): polar points (n, f >= 5)
 * [rho, phi, z, x, y, ...] of `batch` samples -> the same rows grouped by (sector, sample) in their original order
 * (out_offsets: nsectors*batch + 1 entries), phi shifted into the first sector's range, x / y recomputed from (rho, phi), and the
 * [b, z, theta, r] grid indices / linear keys against the SECTOR grid (theta cells = grid[1] / nsectors), both nullable.
 * pc_range (6) / voxel_size (3) / grid (3) are those of the full sweep. */
int pn_split_polar_sectors_f32(const float *points, int n_capacity, int f, const int32_t *sample_offsets, int batch,
                               int nsectors, const float *pc_range, const float *voxel_size, const int32_t *grid,
                               float *out_points, int64_t *grid_ind, uint32_t *keys, int32_t *out_offsets,
                               void *workspace, size_t workspace_bytes, pn_stream_t stream);
/* The row concatenations of the context-padding convolutions (ConvContext / ConvBDCP, det3d/models/necks/rpn_context.py:10-44,
 * 98-158: torch.cat / F.pad along the azimuth axis of NCHW maps = whole-row pieces of NHWC maps): output sample k (of n_out <= 64,
 * out_rows rows of w pixels x c channels) = pieces[3k], pieces[3k+1], pieces[3k+2] stacked; a piece is `rows` rows starting at
 * `src` (already offset to its sample / first row / channel, pixel stride in floats) or zeros when src is NULL. */
typedef struct {
  const float *src;
  int32_t pixel_stride;
  int32_t rows;
} pn_row_piece;
int pn_assemble_rows_f32(const pn_row_piece *pieces, int n_out, int out_rows, int w, int c, float *out,
                         int out_pixel_stride, int out_channel_offset, pn_stream_t stream);

/* layout helpers at the API boundary */
int pn_nchw_to_nhwc_f32(const float *in, int b, int c, int h, int w, float *out, pn_stream_t stream);
int pn_nhwc_to_nchw_f32(const float *in, int b, int c, int h, int w, int pixel_stride,
                        int channel_offset, float *out, pn_stream_t stream);
/* (B,H,W,C) -> (B,W,H,C): the (theta,r) <-> (r,theta) token order change around the attention
 * blocks, x.permute(0,1,3,2) at det3d/models/detectors/voxelnet.py:211,219 */
int pn_transpose_hw_f32(const float *in, int b, int h, int w, int c, float *out, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Several convolutions of one tile shape as ONE launch, with the GroupNorm-family layer between two convolutions folded
 * into them: the producing convolution's epilogue emits per-tile statistics partials, a small second launch folds them
 * into an affine table (deterministic: fixed-order fold, no float atomics); the consuming convolution applies
 * relu(x*A + B) while it loads its input tile, so the normalised map is never written.
 * Replaces the head of CenterHeadSingle / CenterHeadSinglePos.forward: shared_conv + RSNorm, the five first-stage
 * branches (RangeStratified 'reg', grouped 'rot_vel', 'height', 'dim', 'hm') and their last convolutions
 * (det3d/models/bbox_heads/center_head_parallel.py:120-196, 262-284; det3d/models/utils/norm.py:58-75).
 *   desc / in / packed_w / scale / shift / out: as pn_conv2d_nhwc_f32 (range_strata > 1 and groups > 1 become z slices).
 *   stat_*: statistics of the affine-applied, pre-activation output (desc.act must be PN_ACT_NONE for them to be the
 *     norm's input): stat_partials (pn_conv_stat_partial_floats floats of scratch), stat_strata (RSNorm on a plain convolution: strata along the output width, each a multiple
 *     of 32 columns; 1 otherwise -- the stratified convolution's strata are its z slices), stat_channel_groups (1: one group
 *     over all columns; cout: per channel), stat_gamma / stat_beta ([stratum][cout]), stat_eps;
 *     outputs (written by pn_conv_stats_finalize_f32): stat_affine [batch][stat_affine_strata][cout][2] = (A, B) with y = x*A + B  and / or
 *              stat_mean_rstd [batch][stat_affine_strata][groups][2] (the layout pn_groupnorm_apply_f32 reads).
 *   norm_*: normalise-on-load: input element (b, ih, iw, c) is read as relu(x*A + B), (A, B) =
 *     norm_affine[((b*norm_strata + iw / (in_w / norm_strata)) * norm_channels + c)]; zero padding stays zero.
 *     Either every job of a launch has it or none.
 *   tile: 1 = 128x128, 3 = 64x64, 4 = 64x32, 5 = 64x128 (pixels x columns per block); 3 and 4 support norm_*.
 */
typedef struct {
  pn_conv_desc desc;
  const float *in;
  const float *packed_w;
  const float *scale;
  const float *shift;
  float *out;
  float *stat_partials;
  int32_t stat_strata;
  int32_t stat_channel_groups;
  const float *stat_gamma;
  const float *stat_beta;
  float stat_eps;
  int32_t stat_affine_strata;
  float *stat_affine;
  float *stat_mean_rstd;
  const float *norm_affine;
  int32_t norm_strata;
  int32_t norm_channels;
} pn_conv_job;
/* A plain 3x3 / stride 1 / pad 1 / groups 1 layer (map width even; BatchNorm folded into scale / shift, activation in the epilogue)
 * with a one-dimensional Winograd transform F(2, 3) along the width: four GEMMs over pairs of adjacent output pixels, 6 instead of 9
 * MFMA-equivalents per output.  The transformed operands are formed in registers (inputs) / at pack time (weights:
 * pn_pack_conv_weight_wino_f32 from torch layout (Cout, Cin, 3, 3)); results agree with pn_conv2d_nhwc_f32 to ~1e-6 relative.
 * rpn.py:124-142 (the stride-1 convolutions of every block). */
size_t pn_conv_wino_packed_weight_floats(int cout, int cin);
int pn_pack_conv_weight_wino_f32(const float *w_oihw, int cout, int cin, float *packed, pn_stream_t stream);
/* the same layout for the DATA GRADIENT of that layer (grad_input of Conv2d, rpn.py:124-142 under autograd): a 3x3 / stride 1 / pad 1
 * convolution of dout with w'[n][c][kh][kw] = w[c][n][2-kh][2-kw]; packed straight from the forward (Cout, Cin, 3, 3) tensor, no
 * flipped / transposed copy in between.  The packed size is pn_conv_wino_packed_weight_floats(cin_fwd, cout_fwd). */
int pn_pack_conv_dgrad_weight_wino_f32(const float *w_fwd_oihw, int cout_fwd, int cin_fwd, float *packed, pn_stream_t stream);
int pn_conv2d_wino_nhwc_f32(const pn_conv_desc *desc, const float *in, const float *packed_w, const float *scale,
                            const float *shift, float *out, pn_stream_t stream);
/* The same layers with F(4, 3) along the width (map width a multiple of 4; activation none or ReLU): six GEMMs over quads of
 * adjacent output pixels, 4.5 MFMA-equivalents per output; a wave owns all six positions of its 32 quads x 32 columns, so the output
 * transform is wave-local.  Block tile 32 quads x 128 columns, two blocks per CU; pn_conv_wino4_tiles = the number of block tiles
 * of a launch (the caller takes this kernel when they fill the chip).  Weights: pn_pack_conv_weight_wino4_f32 from torch layout
 * (transformed in double, rounded once); results agree with pn_conv2d_nhwc_f32 to ~4e-6 of the map's range.  rpn.py:124-142. */
size_t pn_conv_wino4_packed_weight_floats(int cout, int cin);
int pn_pack_conv_weight_wino4_f32(const float *w_oihw, int cout, int cin, float *packed, pn_stream_t stream);
/* F(4,3) weights of the layer's DATA GRADIENT from the forward tensor (see pn_pack_conv_dgrad_weight_wino_f32) */
int pn_pack_conv_dgrad_weight_wino4_f32(const float *w_fwd_oihw, int cout_fwd, int cin_fwd, float *packed, pn_stream_t stream);
int pn_conv_wino4_tiles(const pn_conv_desc *desc);
int pn_conv2d_wino4_nhwc_f32(const pn_conv_desc *desc, const float *in, const float *packed_w, const float *scale,
                             const float *shift, float *out, pn_stream_t stream);
/* CHAINS of such layers kept in the F(4, 3) domain between layers (r4; the `layer_nums` same-shape Conv2d + BatchNorm + ReLU layers of an
 * RPN block, rpn.py:124-142, replacing pn_conv2d_wino4_nhwc_f32 launch by launch): a layer reads its input as six PLANES
 *   V[p 6][cg C/8][h 2][b][y H+2][xq W/4][j 4]     (channel 8 cg + 4 h + j; rows y = 0 and y = H + 1 of every image are zero padding)
 * = the transformed quads in the MFMA fragment layout, and its epilogue writes the next layer's planes (output transform, scale / shift /
 * activation, input transform) and / or the NHWC map.  pn_wino4_planes_floats = size of such a buffer (0: shape not representable);
 * pn_wino4_planes_from_nhwc_f32 forms the planes of an NHWC channel slice (head of a chain; transpose_hw: of the transposed map, then pn_wino4_planes_floats
 * takes (w, h)); the writers also write the padding rows, so
 * the buffers need no initialisation.  Weights: pn_pack_conv_weight_wino4_f32 unchanged.  pn_conv_wino4_chain_supported: 3x3 / stride 1 /
 * pad 1, cin and cout multiples of 32, activation none or ReLU, whole map rows per 32- or 64-quad tile (W / 4 divides 64).  planes_out or
 * out_nhwc may be NULL (not both); desc->out_* address out_nhwc, desc->in_pixel_stride / in_channel_offset are not used.  Results agree
 * with pn_conv2d_wino4_nhwc_f32 up to the summation order over K. */
size_t pn_wino4_planes_floats(int batch, int h, int w, int c);
int pn_wino4_planes_from_nhwc_f32(const float *in, int batch, int h, int w, int c, int in_pixel_stride, int in_channel_offset,
                                  int transpose_hw, float *planes, pn_stream_t stream);
int pn_conv_wino4_chain_supported(const pn_conv_desc *desc);
int pn_conv2d_wino4_chain_f32(const pn_conv_desc *desc, const float *planes_in, const float *packed_w, const float *scale,
                              const float *shift, float *planes_out, float *out_nhwc, pn_stream_t stream);
/* The same chain step with F(2, 3) along the map height on top of F(4, 3) along the width (even map height): 24 products per OCTET
 * (two rows x four pixels) and channel pair, 3 per output against 4.5.  Planes in / out unchanged -- the consumer forms the height
 * transform from four rows of a plane while it loads them -- only the weights differ: pn_pack_conv_weight_wino24_f32 from torch layout
 * ([chunk][s 4][p 6][k4 8][cout_pad][4] = Gh g Gw^T in double, rounded once).  Agrees with the direct kernel to ~5e-6 of the map's range. */
size_t pn_conv_wino24_packed_weight_floats(int cout, int cin);
int pn_pack_conv_weight_wino24_f32(const float *w_oihw, int cout, int cin, float *packed, pn_stream_t stream);
int pn_conv_wino24_chain_supported(const pn_conv_desc *desc);
int pn_conv2d_wino24_chain_f32(const pn_conv_desc *desc, const float *planes_in, const float *packed_w24, const float *scale,
                               const float *shift, float *planes_out, float *out_nhwc, pn_stream_t stream);

/* r5: the same chain step with F(4,3) along the map height as well (csrc/conv_wchain.hip, conv_wchain3_kernel): 2.25 MFMA-equivalents per
 * output, tiles of four rows x 128 pixels x 32 channels, the six height positions split over two waves.  Frames of at most 64 quads per row
 * (256 pixels along the Winograd axis: two 128-pixel halves per block, one after the other) and a height that is a multiple of 4; half as
 * many blocks as the F(2,3) x F(4,3) form on the same map -- the host side picks it where other frames fill the chip
 * (pn_conv_desc.frames_in_flight > 1).  Planes format unchanged (reads and writes the planes
 * of pn_wino4_planes_floats); weights from pn_pack_conv_weight_wino44_f32.  Replaces the same reference layers as pn_conv2d_wino4_chain_f32
 * (det3d/models/necks/rpn.py:124-142). */
size_t pn_conv_wino44_packed_weight_floats(int cout, int cin);
int pn_pack_conv_weight_wino44_f32(const float *w_oihw, int cout, int cin, float *packed, pn_stream_t stream);
int pn_conv_wino44_chain_supported(const pn_conv_desc *desc);
int pn_conv2d_wino44_chain_f32(const pn_conv_desc *desc, const float *planes_in, const float *packed_w44, const float *scale,
                               const float *shift, float *planes_out, float *out_nhwc, pn_stream_t stream);
/* The centre head's branch convolutions in the same domain (r4; center_head_parallel.py:120-196 behind its RSNorm: Conv2d / RangeStratified
 * 64 -> 64 + GroupNorm + ReLU + last convolution).  pn_groupnorm_strat_planes_f32 (below, with the GroupNorm entries) writes the normalised
 * shared map as planes; pn_conv2d_wino24_chain_head_f32 = pn_conv2d_wino24_chain_f32 plus
 *   stat_partials   (nullable) per-channel (sum, sum of squares) of the output per tile and row: [tile][2][cout][2] floats
 *                   (pn_conv_wino24_chain_stat_floats; a tile covers pn_conv_wino24_chain_stat_tile_rows frame rows; activation must be none)
 *   desc->range_strata > 1: RangeStratified weights (center_head_parallel.py:27-59) on the TRANSPOSED map (transpose_hw = 1: the frame's rows
 *                   are the range positions): range_strata weight sets packed one after the other, rows [s R / S, (s + 1) R / S) take set s
 * pn_wino24_chain_head_finalize_f32 folds the partials of up to 8 channel slices into the affine tables (A, B) (y = x A + B) that
 * pn_conv2d_small_n_multi_f32 applies on load: per-channel groups (strata <= 1) or one group per stratum over the slice's channels. */
typedef struct pn_head_stat_job {
  const float *partials;      /* of the launch the slice belongs to */
  int32_t cout_total;         /* output channels of that launch */
  int32_t channel_offset;     /* first channel of the slice */
  int32_t channels;
  int32_t strata;             /* 0 / 1: GroupNorm(C, C); > 1: GroupNorm(strata, strata * C) over range strata */
  const float *gamma;         /* [max(strata, 1)][channels], NULL = 1 */
  const float *beta;
  float eps;
  float *table;               /* out: [batch][max(strata, 1)][channels][2] */
} pn_head_stat_job;
size_t pn_conv_wino24_chain_stat_floats(const pn_conv_desc *desc);
int pn_conv_wino24_chain_stat_tile_rows(const pn_conv_desc *desc);
int pn_conv2d_wino24_chain_head_f32(const pn_conv_desc *desc, const float *planes_in, const float *packed_w24, const float *scale,
                                    const float *shift, float *planes_out, float *out_nhwc, float *stat_partials, pn_stream_t stream);
int pn_wino24_chain_head_finalize_f32(const pn_head_stat_job *jobs, int njobs, int batch, int frame_rows, int frame_row_pixels, int tile_rows,
                                      pn_stream_t stream);
/* Up to 4 pn_conv2d_wino24_chain_head_f32 layers of one map and cin as ONE launch (the head's branch groups are 128 - 384 short blocks
 * each; apart they leave the chip half empty between launches).  Results are those of the separate calls, bit for bit. */
typedef struct pn_chain_head_job {
  const pn_conv_desc *desc;
  const float *planes_in, *packed_w24, *scale, *shift;
  float *planes_out, *out_nhwc, *stat_partials;
} pn_chain_head_job;
int pn_conv2d_wino24_chain_head_multi_f32(const pn_chain_head_job *jobs, int njobs, pn_stream_t stream);
/* Weight gradient of a plain 3x3 / stride 1 / pad 1 convolution in the F(4, 3) domain (map width a multiple of 4): six GEMMs per kernel
 * row over the quads, dW = G^T [ (B^T d) (A dy)^T ], half the MFMA work of pn_conv2d_wgrad_f32; desc as for pn_conv2d_wgrad_f32 (in_* = the
 * layer's input x, out_* = dout); slices summed in fixed order (deterministic).  Autograd of the RPN's Conv2d layers, rpn.py:124-142 under
 * trainer.py:275-300. */
size_t pn_conv2d_wgrad_wino4_workspace_bytes(const pn_conv_desc *desc);
int pn_conv2d_wgrad_wino4_f32(const pn_conv_desc *desc, const float *x, const float *dout, float *dw_oihw, int accumulate,
                              void *workspace, size_t workspace_bytes, pn_stream_t stream);
/* The FIRST 3x3 convolution of the backbone (ZeroPad2d(1) + Conv2d(3, stride 1 | 2) + folded BatchNorm + activation, rpn.py:124-142) on
 * the pillar canvas of DynamicPPScatter (pillar_encoder.py:393-432), exploiting its sparsity: only (pillar, tap) pairs are multiplied
 * -- per tap one gathered MFMA GEMM over the pairs whose parity reaches an output, then a fixed-order reduction over the nine taps with
 * the affine + activation (outputs no pillar reaches get act(shift)).  unq_keys / num_voxels: the frame's sorted cell keys
 * ((b * H + y) * W + x) and their count on the device (pn_unique_voxels / the fused frame index); the caller promises that every
 * non-zero pixel of the canvas is among them.  cin 32, 64 or 128.  Weights: pn_pack_pillar_conv_weight_f32 from torch
 * layout (Cout, Cin, 3, 3).  Deterministic; agrees with pn_conv2d_nhwc_f32 to ~1e-6 of the map's range. */
size_t pn_pillar_conv_packed_weight_floats(int cout, int cin);
int pn_pack_pillar_conv_weight_f32(const float *w_oihw, int cout, int cin, float *packed, pn_stream_t stream);
size_t pn_pillar_conv_workspace_bytes(int v_capacity, int batch, int oh, int ow, int cout);
int pn_pillar_conv3x3_f32(const float *canvas, int batch, int h, int w, int cin, int in_pixel_stride, int in_channel_offset,
                          const uint32_t *unq_keys, const int32_t *num_voxels, int v_capacity, int stride,
                          const float *packed_w, int cout, const float *scale, const float *shift, int act, float *out,
                          int out_pixel_stride, int out_channel_offset, void *workspace, size_t workspace_bytes,
                          pn_stream_t stream);
/* The same layer with its output written as the F(4, 3) planes the chained layers read (pn_wino4_planes_floats(batch, oh, ow, cout)
 * floats, not transposed; r4): the reduction over the taps forms them directly, the NHWC map is never stored.  Bit-identical to
 * pn_pillar_conv3x3_f32 + pn_wino4_planes_from_nhwc_f32.  pn_pillar_conv_planes_supported: ow % 4 == 0, cout % 8 == 0, whole waves of
 * quads (batch * oh * ow / 4 a multiple of 64). */
int pn_pillar_conv_planes_supported(int batch, int oh, int ow, int cout);
int pn_pillar_conv3x3_planes_f32(const float *canvas, int batch, int h, int w, int cin, int in_pixel_stride, int in_channel_offset,
                                 const uint32_t *unq_keys, const int32_t *num_voxels, int v_capacity, int stride,
                                 const float *packed_w, int cout, const float *scale, const float *shift, int act, float *planes,
                                 void *workspace, size_t workspace_bytes, pn_stream_t stream);
/* The same layer in its ROW-BAND form (r6, csrc/pillar_rows.hip; stride 2): one block per OUTPUT ROW keeps the row's Cout channels in LDS,
 * walks the three canvas rows it reads as runs of the sorted key list (row_start of pn_voxel_index_fused_rows_f32), multiplies every pillar
 * by the taps its column parity reaches (v_mfma_f32_16x16x4_f32 chains, canvas rows and weights straight from L2) and finishes the row from
 * LDS -- no pair lists, no per-pair partial rows in memory (pn_pillar_conv3x3_f32 moves 64 MB of them per frame), one launch instead of four.
 * Same terms as the dense convolution, fixed summation order (bitwise reproducible), not the bits of pn_pillar_conv3x3_f32.
 *   packed_rows_w: pn_pack_pillar_conv_rows_weight_f32 ([tap][cin / 16][cout padded to 16][4][4] floats)
 *   planes and / or out: the F(4, 3) planes (pn_wino4_planes_floats(batch, oh, ow, cout) floats, not transposed) and / or the NHWC map
 *   pn_pillar_conv_rows_supported: stride 2, w % 8 == 0, w <= 512, ow / 4 in {16, 32, 64}, cin in {32, 64, 128}, cout <= 128, cout % 4 == 0 */
int pn_pillar_conv_rows_supported(int batch, int h, int w, int cin, int cout, int stride);
size_t pn_pillar_conv_rows_packed_weight_floats(int cout, int cin);
int pn_pack_pillar_conv_rows_weight_f32(const float *w_oihw, int cout, int cin, float *packed, pn_stream_t stream);
int pn_pillar_conv3x3_rows_f32(const float *canvas, int batch, int h, int w, int cin, int in_pixel_stride, int in_channel_offset,
                               const uint32_t *unq_keys, const int32_t *row_start, int v_capacity, const float *packed_rows_w, int cout,
                               const float *scale, const float *shift, int act, float *planes, float *out, int out_pixel_stride,
                               int out_channel_offset, pn_stream_t stream);
/* pn_conv2d_nhwc_f32 with its output written as the F(4, 3) planes the chained layers read (r6): the stride-2 convolution at the head of an
 * RPN block (det3d/models/necks/rpn.py:124-142) feeds the block's chain (pn_conv2d_wino*_chain_f32) without the NHWC map and the
 * pn_wino4_planes_from_nhwc_f32 pass in between.  planes: pn_wino4_planes_floats(batch, oh, ow, cout) floats, not transposed; bit for bit the
 * planes of the NHWC result.  Supported: groups 1, no deconvolution / strata / accumulate, cout % 8 == 0, output rows of 128 pixels or of a
 * width dividing 64, whole block tiles of rows. */
int pn_conv2d_nhwc_planes_supported(const pn_conv_desc *d);
int pn_conv2d_nhwc_planes_f32(const pn_conv_desc *d, const float *in, const float *packed_w, const float *scale, const float *shift,
                              float *planes, pn_stream_t stream);
/* The same convolution in TRAINING: the pair tables are built once per iteration (pn_pillar_pairs_build; pn_pillar_pairs_bytes of
 * caller-owned memory that lives from the forward to the backward) and shared by
 *   forward          pn_pillar_conv3x3_tables_f32        (oh, ow = the output map; workspace 9 * cap * cout floats, cap = v_capacity
 *                                                         rounded up to 128)
 *   data gradient    pn_pillar_conv3x3_dgrad_f32         d(pillar features) [v_capacity][cin], rows in the order of unq_keys -- what
 *                                                         pn_dynamic_pfn_bwd takes as d_features; packed_wt = pn_pack_pillar_conv_weight_f32
 *                                                         of the weight with its first two axes swapped; workspace 9 * cap * cin floats
 *   weight gradient  pn_pillar_conv3x3_wgrad_f32         dW (Cout, Cin, 3, 3) = sum over the pairs of dout[out]^T x[in], blocks summed in
 *                                                         fixed order (deterministic); cin, cout <= 128
 * Autograd of Conv2d in RPN block 0, rpn.py:124-142 under trainer.py:275-300. */
size_t pn_pillar_pairs_bytes(int v_capacity, int batch, int oh, int ow);
int pn_pillar_pairs_build(const uint32_t *unq_keys, const int32_t *num_voxels, int v_capacity, int batch, int h, int w,
                          int stride, void *pair_tables, size_t table_bytes, pn_stream_t stream);
int pn_pillar_conv3x3_tables_f32(const float *canvas, int batch, int oh, int ow, int cin, int in_pixel_stride,
                                 int in_channel_offset, const void *pair_tables, int v_capacity, const float *packed_w,
                                 int cout, const float *scale, const float *shift, int act, float *out,
                                 int out_pixel_stride, int out_channel_offset, void *workspace, size_t workspace_bytes,
                                 pn_stream_t stream);
int pn_pillar_conv3x3_dgrad_f32(const float *dout, int batch, int oh, int ow, int cout, int dout_pixel_stride,
                                int dout_channel_offset, const void *pair_tables, const int32_t *num_voxels,
                                int v_capacity, const float *packed_wt, int cin, float *dfeat, void *workspace,
                                size_t workspace_bytes, pn_stream_t stream);
size_t pn_pillar_conv_wgrad_workspace_bytes(int v_capacity, int cin, int cout);
int pn_pillar_conv3x3_wgrad_f32(const float *canvas, int in_pixel_stride, int in_channel_offset, int cin, const float *dout,
                                int dout_pixel_stride, int dout_channel_offset, int cout, const void *pair_tables,
                                int v_capacity, int batch, int oh, int ow, float *dw_oihw, int accumulate, void *workspace,
                                size_t workspace_bytes, pn_stream_t stream);
size_t pn_conv_stat_partial_floats(const pn_conv_desc *desc, int tile);
int pn_conv2d_multi_f32(const pn_conv_job *jobs, int njobs, int tile, pn_stream_t stream);
/* the same job list on the VALU kernel for convolutions with very few output columns (1x1 / 3x3, <= 64 input channels,
 * <= 12 columns: the last convolutions of the head branches, center_head_parallel.py:140-175), norm_* supported (a
 * range-stratified table only for 1x1 kernels); one launch for all jobs */
int pn_conv2d_small_n_multi_f32(const pn_conv_job *jobs, int njobs, pn_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolutions with one to three output channels over many input channels (E2ESWVoteHead's 256 -> 1 heat-map
 * and vote-class layers, e2e_swv_head.py:88-97) as GEMM + tap sum: g = pn_linear_ksplit_f32(x, W9) with W9[(t * cout + co)][ci] =
 * w[co][ci][t / 3][t % 3] (rows padded to a multiple of 4) reads the input once; this entry adds the nine shifted columns:
 * out[b,y,x,co] = act(scale[co] * sum_t g[b, y + t/3 - 1, x + t%3 - 1][t * cout + co] + shift[co]), taps ascending, zero padding. */
int pn_conv3x3_tap_sum_f32(const float *g, int ldg, int batch, int h, int w, int cout, const float *scale,
                           const float *shift, int act, float *out, int out_pixel_stride,
                           int out_channel_offset, pn_stream_t stream);
/* folds the partials the jobs' epilogues wrote (same job array and tile as the pn_conv2d_multi_f32 call; jobs without
 * stat_partials are skipped) into stat_affine / stat_mean_rstd: one small launch, fixed association order */
int pn_conv_stats_finalize_f32(const pn_conv_job *jobs, int njobs, int tile, pn_stream_t stream);
/* RSNorm + activation of `producer->out` (a plain convolution launched with all-channel statistics, stat_strata = the norm's
 * range strata) WITHOUT a finalize launch: every block folds the partials of its own (sample, stratum) group first.
 * gamma / beta [stratum][cout]; out2 = out*mul + add with (H, W, cout) maps (CenterHeadSinglePos calibration), optional. */
int pn_conv_stats_apply_f32(const pn_conv_job *producer, int tile, const float *gamma, const float *beta, int act,
                            float *out, int out_pixel_stride, int out_channel_offset, const float *mul,
                            const float *add, float *out2, int out2_pixel_stride, int out2_channel_offset,
                            pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Differentiable primitives for the TRAINING forms of the attention blocks (det3d/models/utils/set_transformer.py:118-166,
 * 216-259, 307-354, 392-440 under torch autograd; the inference kernels above keep no intermediates).
 * pn_contract_f32: C[g0,g1,g2, m0,m1, n0,n1] (+)= alpha * sum_{k0,k1} A[g.., m0,m1, k0,k1] * B[g.., n0,n1, k0,k1]; strides in
 * elements, stride_a = {g0,g1,g2,m0,m1,k0,k1}, stride_b = {g0,g1,g2,n0,n1,k0,k1}, stride_c = {g0,g1,g2,m0,m1,n0,n1},
 * dims = {G0,G1,G2,M0,M1,N0,N1,K0,K1}.  The head / window / raw-view permutations of the reference are strides here.
 */
int pn_contract_f32(const float *a, const int64_t *stride_a, const float *b, const int64_t *stride_b, float *c,
                    const int64_t *stride_c, const int32_t *dims, float alpha, int accumulate, pn_stream_t stream);
/* softmax over the middle axis of a contiguous (outer, n, inner) array and its backward (dx = y * (dy - sum(y * dy))) */
int pn_softmax_f32(const float *x, float *y, long long outer, int n, int inner, pn_stream_t stream);
int pn_softmax_bwd_f32(const float *y, const float *dy, float *dx, long long outer, int n, int inner, pn_stream_t stream);
/* backward of pn_layernorm_f32 over rows of c <= 1024 values: dx, and dgamma / dbeta (+= when accumulate) */
size_t pn_layernorm_bwd_workspace_bytes(long long rows, int c);
int pn_layernorm_bwd_f32(const float *x, const float *dy, const float *gamma, float eps, long long rows, int c, float *dx,
                         float *dgamma, float *dbeta, int accumulate, void *workspace, size_t workspace_bytes,
                         pn_stream_t stream);
/* exact (erf) GELU and its backward */
int pn_gelu_f32(const float *x, float *y, size_t n, pn_stream_t stream);
int pn_gelu_bwd_f32(const float *x, const float *dy, float *dx, size_t n, pn_stream_t stream);
/* rel[g0,g1, m0,m1, n0,n1][0..1] = a[g.., m0,m1][0..1] - b[g.., n0,n1][0..1], columns 2..cols-1 zero, output contiguous:
 * the Cartesian offsets the relative-position MLPs take (set_transformer.py:230-236, 321-326).  Strides in floats,
 * stride_a = {g0,g1,m0,m1}, stride_b = {g0,g1,n0,n1}, dims = {G0,G1,M0,M1,N0,N1}. */
int pn_pair_diff_f32(const float *a, const int64_t *stride_a, const float *b, const int64_t *stride_b,
                     const int32_t *dims, int cols, float *rel, pn_stream_t stream);
/* nn.Dropout (row_len = 1) / DropPath (row_len = elements of one sample): y = x * mask, mask in {0, 1/(1-p)} from a
 * counter-based generator keyed by (seed, element / row_len); the mask is kept for the backward (dx = dy * mask) */
int pn_dropout_f32(const float *x, size_t n, size_t row_len, float p, uint64_t seed, float *y, float *mask,
                   pn_stream_t stream);
int pn_mul_f32(const float *a, const float *b, float *y, size_t n, pn_stream_t stream);
/* dst[b, index[b,k,w], w, :] += src[b, k, w, :]: backward of the key-point row gather (set_transformer.py:144-147) */
int pn_scatter_rows_f32(const float *src, const int32_t *index, int batch, int k, int h, int w, int c, float *dst,
                        pn_stream_t stream);
/* torch.roll(x, shift, dims=2) of a (batch, h, w, c) map (set_transformer.py:121-124, 160-163) */
int pn_roll_w_f32(const float *x, int batch, int h, int w, int c, int shift, float *y, pn_stream_t stream);
/* shifted-window plumbing (sw2votev4_util.py:140-176): zero-pad (batch, h, w, c) to (batch, hp, wp, c) and roll by
 * (-shift, -shift); pn_crop_roll_f32 is its adjoint (roll back by +shift, crop) and the forward's way back to the map */
int pn_pad_roll_f32(const float *x, int batch, int h, int w, int hp, int wp, int c, int shift, float *y, pn_stream_t stream);
int pn_crop_roll_f32(const float *y, int batch, int h, int w, int hp, int wp, int c, int shift, float *x, pn_stream_t stream);
/* y[r, ch] = x[r, ch] * scale[ch] over n = rows*c elements; y = 1/max(x, lo) and its backward (the clamped per-head
 * temperature tau of the cosine attention, sw2votev4_util.py:84) */
int pn_scale_channels_f32(const float *x, const float *scale, size_t n, int c, float *y, pn_stream_t stream);
int pn_recip_clamp_f32(const float *x, float lo, int n, float *y, pn_stream_t stream);
int pn_recip_clamp_bwd_f32(const float *x, const float *dy, float lo, int n, float *dx, pn_stream_t stream);
/* F.normalize over rows of c values (cosine attention, swin_transformer_v2.py:156-158) and its backward; inv_norm[rows] */
int pn_l2_normalize_f32(const float *x, long long rows, int c, float eps, float *y, float *inv_norm, pn_stream_t stream);
int pn_l2_normalize_bwd_f32(const float *y, const float *dy, const float *inv_norm, long long rows, int c, float *dx,
                            pn_stream_t stream);

/* next-4  ego-motion warp of the previous sweep's feature maps for the bidirectional context padding
 * (PolarStreamBDCP.forward_one_sweep `feature_only`, det3d/models/detectors/polarstream.py:318-372 with get_grids :223-238 and
 * get_center :239-247): the previous sweep's map, given as its nsectors sector maps stacked sector-major in the batch axis
 * (nsectors * batch, h / nsectors, w, c) NHWC (h = azimuth rows of the whole sweep, w = range columns; nsectors = 1: a plain
 * (batch, h, w, c) map) -> out (batch, h, w, c): the whole-sweep map resampled at the rotated cell positions (the torch.cat of
 * polarstream.py:343-349 is folded into the read).  rot2x2: (batch, 2, 2) row-major `transform_matrix[:2, :2]`
 * (voxelization.py:447); bilinear, zero padding, align_corners = False (torch.nn.functional.grid_sample defaults). */
int pn_polar_warp_f32(const float *in, const float *rot2x2, int batch, int nsectors, int h, int w, int c, float range_lo,
                      float range_hi, float azimuth_lo, float azimuth_hi, float *out, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * next-4  global augmentation of one training sample, in place on the device: points (n, point_stride >= 3) [x, y, z, ...]
 * and boxes (m, box_cols = 7 | 9) [x, y, z, w, l, h, (vx, vy,) heading].  Order and arithmetic of
 * prep.random_flip_both / global_rotation / global_scaling_v2 / global_translate_
 * (det3d/core/sampler/preprocess.py:803-832, 771-788, 835-839, 940-962, called from
 * det3d/datasets/pipelines/preprocess.py:107-117); the random draws are the caller's (partner_amd/augment.py draws
 * them with np.random in the reference's order).  flip_y: y -> -y (first flip), flip_x: x -> -x (second flip);
 * rot_sin / rot_cos / rot_angle in float32 as the reference builds its matrix; translate: 3 doubles or NULL.
 */
int pn_global_augment_f32(float *points, int n, int point_stride, float *boxes, int m, int box_cols, int flip_y,
                          int flip_x, int do_rotation, float rot_sin, float rot_cos, float rot_angle, float scale,
                          const double *translate, pn_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * timing helper: HIP events on `stream`, used by bench.py for the roofline object.
 */
typedef void *pn_event_t;
int pn_event_create(pn_event_t *ev);
int pn_event_destroy(pn_event_t ev);
int pn_event_record(pn_event_t ev, pn_stream_t stream);
int pn_event_elapsed_ms(pn_event_t start, pn_event_t stop, float *ms); /* synchronises on stop */
/* Arms the calling host thread: the NEXT convolution / GEMM launch of this thread (pn_conv2d_nhwc_f32,
 * pn_conv2d_nhwc_bf16, pn_conv2d_multi_f32, pn_gemm_bias_act_f32, pn_linear_f32, pn_linear_ksplit_f32, pn_sparse_conv_f32) attaches `start` / `stop` to the
 * kernel dispatch itself (hipExtLaunchKernelGGL), so that pn_event_elapsed_ms(start, stop) is that kernel's execution
 * time -- what a rocprofv3 kernel trace reports -- without the launch gaps a pair of hipEventRecord calls around an eager
 * launch includes.  One shot; not usable while the stream is being captured into a hipGraph. */
int pn_profile_next_launch(pn_event_t start, pn_event_t stop);

#ifdef __cplusplus
}
#endif
#endif /* PARTNER_HIP_H */
