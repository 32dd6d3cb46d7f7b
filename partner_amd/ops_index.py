"""Frame index and scatter stage over the C ABI (V0 .. V5 of SURVEY 8a, hard voxelization, target assignment, sector split, sweep
accumulation): device memory from PyTorch, kernels from libpartner_hip."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple  # noqa: F401

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc  # noqa: F401
from .ops_common import *  # noqa: F401,F403
from .ops_common import _f32, _workspace  # noqa: F401
from .routes import R, S  # noqa: F401


# ------------------------------------------------------------------------------ V0 / V1
def cart_to_polar(cart: torch.Tensor) -> torch.Tensor:
    hip.require_device(cart)
    cart = cart.contiguous()
    n, f = cart.shape
    out = torch.empty((n, f + 2), dtype=torch.float32, device=cart.device)
    hip.call("pn_cart_to_polar_f32", cart.data_ptr(), n, f, out.data_ptr(), hip.stream())
    return out


@dataclass
class GridSpec:
    """polar grid of a voxel generator: lo = range[:3], voxel size, grid = (R, T, Z)"""
    lo: Tuple[float, float, float]
    vs: Tuple[float, float, float]
    grid: Tuple[int, int, int]

    @staticmethod
    def from_range(pc_range: Sequence[float], voxel_size: Sequence[float]) -> "GridSpec":
        import numpy as np

        r = np.asarray(pc_range, dtype=np.float32)
        v = np.asarray(voxel_size, dtype=np.float32)
        g = np.round((r[3:] - r[:3]) / v).astype(np.int64)  # VoxelGenerator.__init__ (voxel_generator.py:6-17)
        return GridSpec(tuple(float(x) for x in r[:3]), tuple(float(x) for x in v), tuple(int(x) for x in g))

    def c_arrays(self):
        return (C.c_float * 3)(*self.lo), (C.c_float * 3)(*self.vs), (C.c_int32 * 3)(*self.grid)

    def num_cells(self, batch: int) -> int:
        return batch * self.grid[0] * self.grid[1] * self.grid[2]


def grid_index(points: torch.Tensor, sample_offsets: torch.Tensor, batch: int, spec: GridSpec, want_grid_ind=True,
               want_keys=True):
    """points (N,F>=3) polar fp32, sample_offsets int32 (batch+1) on device."""
    hip.require_device(points, sample_offsets)
    assert points.dtype == torch.float32 and points.is_contiguous() and sample_offsets.dtype == torch.int32
    n = points.shape[0]
    gi = torch.empty((n, 4), dtype=torch.int64, device=points.device) if want_grid_ind else None
    keys = torch.empty((n,), dtype=torch.int32, device=points.device) if want_keys else None
    lo, vs, g = spec.c_arrays()
    hip.call("pn_polar_grid_index_f32", points.data_ptr(), points.shape[1], n, sample_offsets.data_ptr(), batch, lo, vs, g,
             hip.ptr(gi), hip.ptr(keys), hip.stream())
    return gi, keys


def keys_from_grid_ind(grid_ind: torch.Tensor, spec: GridSpec, batch: int) -> torch.Tensor:
    hip.require_device(grid_ind)
    assert grid_ind.dtype == torch.int64 and grid_ind.is_contiguous()
    n = grid_ind.shape[0]
    keys = torch.empty((n,), dtype=torch.int32, device=grid_ind.device)
    _, _, g = spec.c_arrays()
    hip.call("pn_keys_from_grid_ind", grid_ind.data_ptr(), n, g, batch, keys.data_ptr(), hip.stream())
    return keys


# ------------------------------------------------------------------------------ unique / bucket
@dataclass
class VoxelIndex:
    """device-side result of the bitmap unique + bucketing (no host sync needed to use it)"""
    n_cap: int
    num_cells: int
    spec: GridSpec
    batch: int
    unq: Optional[torch.Tensor]       # (n_cap,4) int64, first V rows valid
    unq_inv: Optional[torch.Tensor]   # (n_cap,) int32 (None on the fused frame-index path)
    unq_cnt: Optional[torch.Tensor]   # (n_cap,) int32, first V valid (None on the fused frame-index path)
    num_voxels: torch.Tensor          # (1,) int32 on device
    voxel_start: torch.Tensor         # (n_cap+1,) int32
    order: torch.Tensor               # (n_cap,) int32
    workspace: torch.Tensor           # keeps unq_keys alive
    unq_keys_ptr: int

    def count(self) -> int:
        """V on the host (synchronises)"""
        return int(self.num_voxels.item())


def build_voxel_index(keys: torch.Tensor, spec: GridSpec, batch: int, n_dev: Optional[torch.Tensor] = None,
                      want_unq=True, sorted_runs=False) -> VoxelIndex:
    """``sorted_runs``: points of a voxel in ascending index order (bit-reproducible PFN backward); otherwise the
    order inside a voxel is unspecified, which no forward kernel depends on."""
    hip.require_device(keys)
    lib = hip.load()
    dev = keys.device
    n = keys.shape[0]
    cells = spec.num_cells(batch)
    ws_bytes = lib.pn_unique_workspace_bytes(cells, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    unq = torch.empty((max(n, 1), 4), dtype=torch.int64, device=dev) if want_unq else None
    inv = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    cnt = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    nv = torch.empty((1,), dtype=torch.int32, device=dev)          # always written by the rank scan
    _, _, g = spec.c_arrays()
    st = hip.stream()
    hip.call("pn_unique_rank_bitmap", keys.data_ptr(), n, hip.ptr(n_dev), cells, g, hip.ptr(unq), inv.data_ptr(),
             cnt.data_ptr(), nv.data_ptr(), ws.data_ptr(), ws_bytes, st)
    bws_bytes = lib.pn_bucket_workspace_bytes(n)
    bws = torch.empty(max(bws_bytes, 1), dtype=torch.uint8, device=dev)
    # entries [0, V] are written by the bucket scan; entries past V are never read (every consumer is bounded by num_voxels)
    vstart = torch.empty((n + 1,), dtype=torch.int32, device=dev) if n > 0 else torch.zeros((1,), dtype=torch.int32, device=dev)
    order = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    hip.call("pn_bucket_points", inv.data_ptr(), cnt.data_ptr(), n, hip.ptr(n_dev), nv.data_ptr(), vstart.data_ptr(),
             order.data_ptr(), bws.data_ptr(), bws_bytes, st)
    if sorted_runs and n > 0:
        raw, order = order, torch.empty_like(order)
        hip.call("pn_sort_voxel_runs", vstart.data_ptr(), nv.data_ptr(), n, raw.data_ptr(), order.data_ptr(), st)
    kp = lib.pn_unique_keys_ptr(ws.data_ptr(), cells, n)
    return VoxelIndex(n, cells, spec, batch, unq, inv, cnt, nv, vstart, order, ws, kp)


class FrameIndexState:
    """persistent scratch of the fused frame index (``fused_voxel_index``): one uint32 per grid cell and the scan state, all zero
    between frames (the frame's cells are cleared again by ``clear_frame_cells``) -- owned by whoever replays frames (an engine),
    one per stream in flight"""

    MAX_CELLS = 1 << 24   # 64 MB of counters; larger grids (the Waymo 3-D grid) take the bitmap path

    def __init__(self, spec: GridSpec, batch: int, device):
        lib = hip.load()
        self.cells = spec.num_cells(batch)
        self.spec, self.batch = spec, batch
        self.cell_count = torch.zeros((self.cells,), dtype=torch.int32, device=device)
        self.scan_state = torch.zeros((int(lib.pn_voxel_index_fused_state_bytes(self.cells)),), dtype=torch.uint8, device=device)

    @staticmethod
    def supported(spec: GridSpec, batch: int) -> bool:
        return spec.num_cells(batch) <= FrameIndexState.MAX_CELLS




def fused_voxel_index(cart: torch.Tensor, sample_offsets: torch.Tensor, batch: int, spec: GridSpec, state: Optional[FrameIndexState] = None):
    """cart (N, F>=3) Cartesian points -> (polar (N, F+2), VoxelIndex) in three launches (V0 + V1 + unique + bucketing).
    ``state``: persistent zeroed scratch (see FrameIndexState); without it a fresh zero-filled one is used (two extra fills)."""
    hip.require_device(cart, sample_offsets)
    assert cart.dtype == torch.float32 and cart.is_contiguous() and sample_offsets.dtype == torch.int32
    if state is None:
        state = FrameIndexState(spec, batch, cart.device)
    assert state.cells == spec.num_cells(batch)
    n, f = cart.shape
    dev = cart.device
    i32 = dict(dtype=torch.int32, device=dev)
    polar = torch.empty((n, f + 2), dtype=torch.float32, device=dev)
    keys = torch.empty((max(n, 1),), **i32)
    pos = torch.empty((max(n, 1),), **i32)
    ukeys = torch.empty((max(n, 1),), **i32)
    vstart = torch.empty((n + 1,), **i32)
    order = torch.empty((max(n, 1),), **i32)
    nv = torch.empty((1,), **i32)
    lo, vs, g = spec.c_arrays()
    # r6: a pillar grid (one cell along z) whose rows are whole groups of eight cells also gets row_start -- the runs of unq_keys per canvas row
    # that the row-band first convolution walks (PillarConvLayer, csrc/pillar_rows.hip); the scan writes it on the way
    row_start = None
    if spec.grid[2] == 1 and spec.grid[0] % 8 == 0 and R.pillar_rows:
        row_start = torch.empty((batch * spec.grid[1] + 1,), **i32)
        hip.call("pn_voxel_index_fused_rows_f32", cart.data_ptr(), n, f, sample_offsets.data_ptr(), batch, lo, vs, g, polar.data_ptr(), keys.data_ptr(),
                 pos.data_ptr(), state.cell_count.data_ptr(), state.scan_state.data_ptr(), state.scan_state.numel(), ukeys.data_ptr(), vstart.data_ptr(),
                 order.data_ptr(), nv.data_ptr(), row_start.data_ptr(), hip.stream())
    else:
        hip.call("pn_voxel_index_fused_f32", cart.data_ptr(), n, f, sample_offsets.data_ptr(), batch, lo, vs, g, polar.data_ptr(), keys.data_ptr(),
                 pos.data_ptr(), state.cell_count.data_ptr(), state.scan_state.data_ptr(), state.scan_state.numel(), ukeys.data_ptr(), vstart.data_ptr(),
                 order.data_ptr(), nv.data_ptr(), hip.stream())
    vi = VoxelIndex(n, state.cells, spec, batch, None, None, None, nv, vstart, order, ukeys, ukeys.data_ptr())
    vi.keys, vi.state, vi.row_start = keys, state, row_start
    return polar, vi


def fused_voxel_index_sweeps(raw: torch.Tensor, sweep_offsets: torch.Tensor, transforms: torch.Tensor, time_lags: torch.Tensor, spec: GridSpec,
                             state: FrameIndexState, min_distance: float = 1.0):
    """the RAW sweeps of one multi-sweep frame (as ``accumulate_sweeps`` takes them) -> (polar (n, 7), VoxelIndex): accumulation and frame
    index in the three launches of ``fused_voxel_index`` (pn_voxel_index_fused_sweeps_f32, r6).  The kept points are not compacted: point i
    is row i of ``raw``; removed points are in no voxel and their polar rows are unwritten."""
    hip.require_device(raw, sweep_offsets, transforms, time_lags)
    assert raw.is_contiguous() and raw.dtype == torch.float32 and transforms.dtype == torch.float64 and transforms.is_contiguous()
    assert sweep_offsets.dtype == torch.int32 and state.cells == spec.num_cells(1)
    n, cols = raw.shape
    dev = raw.device
    i32 = dict(dtype=torch.int32, device=dev)
    polar = torch.empty((n, 7), dtype=torch.float32, device=dev)
    keys, pos, ukeys, order = (torch.empty((n,), **i32) for _ in range(4))
    vstart = torch.empty((n + 1,), **i32)
    nv = torch.empty((1,), **i32)
    lo, vs, g = spec.c_arrays()
    row_start = torch.empty((spec.grid[1] + 1,), **i32) if (spec.grid[2] == 1 and spec.grid[0] % 8 == 0 and R.pillar_rows) else None
    hip.call("pn_voxel_index_fused_sweeps_f32", raw.data_ptr(), n, cols, sweep_offsets.data_ptr(), transforms.shape[0], transforms.data_ptr(),
             time_lags.data_ptr(), float(min_distance), lo, vs, g, polar.data_ptr(), keys.data_ptr(), pos.data_ptr(), state.cell_count.data_ptr(),
             state.scan_state.data_ptr(), state.scan_state.numel(), ukeys.data_ptr(), vstart.data_ptr(), order.data_ptr(), nv.data_ptr(),
             hip.ptr(row_start), hip.stream())
    vi = VoxelIndex(n, state.cells, spec, 1, None, None, None, nv, vstart, order, ukeys, ukeys.data_ptr())
    vi.keys, vi.state, vi.row_start = keys, state, row_start
    return polar, vi


def clear_frame_cells(canvas: Optional[torch.Tensor], vi: VoxelIndex, state: Optional[FrameIndexState] = None, v_cap: Optional[int] = None) -> None:
    """sparse clear at the end of a frame: the canvas cells of the frame's voxels and their ``cell_count`` entries"""
    if canvas is None and state is None:
        return
    _, _, g = vi.spec.c_arrays()
    hip.call("pn_clear_frame_cells", vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap if v_cap is None else v_cap, g,
             0 if canvas is None else canvas.shape[-1], hip.ptr(canvas), None if state is None else state.cell_count.data_ptr(), hip.stream())


def scatter_mean(points: torch.Tensor, vi: VoxelIndex, v_cap: Optional[int] = None) -> torch.Tensor:
    hip.require_device(points)
    v_cap = vi.n_cap if v_cap is None else v_cap
    f = points.shape[1]
    out = torch.empty((max(v_cap, 1), f), dtype=torch.float32, device=points.device)
    hip.call("pn_scatter_mean_f32", points.data_ptr(), points.stride(0), f, vi.voxel_start.data_ptr(), vi.order.data_ptr(),
             vi.num_voxels.data_ptr(), v_cap, out.data_ptr(), hip.stream())
    return out


def hard_voxel_mean(voxels: torch.Tensor, num_points: torch.Tensor) -> torch.Tensor:
    hip.require_device(voxels, num_points)
    voxels = voxels.contiguous()
    v, p, f = voxels.shape
    out = torch.empty((v, f), dtype=torch.float32, device=voxels.device)
    hip.call("pn_hard_voxel_mean_f32", voxels.data_ptr(), num_points.to(torch.int32).contiguous().data_ptr(), v, p, f,
             out.data_ptr(), hip.stream())
    return out


_CENTER_TABLES = {}


def pfn_center_table(t: int, vy: float, y_offset: float, device) -> torch.Tensor:
    key = (t, float(vy), float(y_offset), str(device))
    tab = _CENTER_TABLES.get(key)
    if tab is None:
        tab = torch.empty((2 * t,), dtype=torch.float32, device=device)
        hip.call("pn_pfn_center_table_f32", t, float(vy), float(y_offset), tab.data_ptr(), hip.stream())
        _CENTER_TABLES[key] = tab
    return tab


def dynamic_pfn(points: torch.Tensor, vi: VoxelIndex, w0: torch.Tensor, w1: torch.Tensor, vx: float, vy: float,
                x_offset: float, y_offset: float, features: Optional[torch.Tensor], canvas: Optional[torch.Tensor],
                v_cap: Optional[int] = None, clear_index: Optional[FrameIndexState] = None) -> bool:
    """``clear_index`` (r6): the frame-index state whose per-cell counters the launch zeroes for the frame's voxels on the way (the (32, 128) reader's
    kernels; -> True when it did, False when the caller still has to clear them with ``clear_frame_cells``)"""
    hip.require_device(points, w0, w1)
    assert w0.is_contiguous() and w1.is_contiguous() and points.is_contiguous()
    c0, c1 = w0.shape[0], w1.shape[0]
    assert w0.shape[1] == 16 and w1.shape[1] == 2 * c0
    _, _, g = vi.spec.c_arrays()
    tab = pfn_center_table(vi.spec.grid[1], vy, y_offset, points.device)
    if clear_index is not None and (c0, c1) == (32, 128) and R.pfn_clears_index:
        hip.call("pn_dynamic_pfn_fwd_table_clear", points.data_ptr(), points.stride(0), vi.voxel_start.data_ptr(), vi.order.data_ptr(),
                 vi.num_voxels.data_ptr(), vi.n_cap if v_cap is None else v_cap, vi.unq_keys_ptr, g, w0.data_ptr(), c0,
                 w1.data_ptr(), c1, float(vx), float(vy), float(x_offset), float(y_offset), tab.data_ptr(), hip.ptr(features),
                 hip.ptr(canvas), clear_index.cell_count.data_ptr(), hip.stream())
        return True
    hip.call("pn_dynamic_pfn_fwd_table", points.data_ptr(), points.stride(0), vi.voxel_start.data_ptr(), vi.order.data_ptr(),
             vi.num_voxels.data_ptr(), vi.n_cap if v_cap is None else v_cap, vi.unq_keys_ptr, g, w0.data_ptr(), c0,
             w1.data_ptr(), c1, float(vx), float(vy), float(x_offset), float(y_offset), tab.data_ptr(), hip.ptr(features),
             hip.ptr(canvas), hip.stream())
    return False


def clear_canvas_cells(canvas: torch.Tensor, vi: VoxelIndex, v_cap: Optional[int] = None) -> None:
    """zero the cells of ``vi``'s voxels in a persistent NHWC canvas (sparse clear after the canvas has been consumed)"""
    hip.require_device(canvas)
    assert canvas.is_contiguous() and canvas.dtype == torch.float32
    _, _, g = vi.spec.c_arrays()
    hip.call("pn_clear_canvas_cells", vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap if v_cap is None else v_cap, g,
             canvas.shape[-1], canvas.data_ptr(), hip.stream())


def scatter_canvas(features: torch.Tensor, unq: torch.Tensor, batch: int, t: int, r: int,
                   num_voxels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> zero-filled NHWC canvas (batch, T, R, C) with features written at unq[:, (0,2,3)]"""
    hip.require_device(features, unq)
    features = features.contiguous()
    unq = unq.contiguous()
    v, c = features.shape
    canvas = torch.empty((batch, t, r, c), dtype=torch.float32, device=features.device)
    st = hip.stream()
    hip.call("pn_fill_zero", canvas.data_ptr(), canvas.numel() * 4, st)
    if num_voxels is None:
        num_voxels = torch.full((1,), v, dtype=torch.int32, device=features.device)
    hip.call("pn_scatter_canvas_fwd", features.data_ptr(), unq.data_ptr(), num_voxels.data_ptr(), v, c, t, r,
             canvas.data_ptr(), st)
    return canvas

# ------------------------------------------------------------------------------ V2 hard voxelization
def hard_voxelize(points: torch.Tensor, voxel_size, pc_range, max_points: int, max_voxels: int):
    """-> voxels (max_voxels, max_points, F), coors int32 (max_voxels, 3) [z,theta,r], num_points int32
    (max_voxels,), num_voxels (1,) int32 on the device; rows >= num_voxels are zero / undefined."""
    import numpy as np

    hip.require_device(points)
    lib = hip.load()
    assert points.dtype == torch.float32 and points.is_contiguous()
    n, f = points.shape
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(pc_range, dtype=np.float32)
    grid = np.round((rg[3:] - rg[:3]) / vs).astype(np.int64)
    cells = int(grid[0]) * int(grid[1]) * int(grid[2])
    dev = points.device
    voxels = torch.empty((max_voxels, max_points, f), dtype=torch.float32, device=dev)
    coors = torch.zeros((max_voxels, 3), dtype=torch.int32, device=dev)
    num = torch.zeros((max_voxels,), dtype=torch.int32, device=dev)
    nv = torch.empty((1,), dtype=torch.int32, device=dev)
    ws_bytes = lib.pn_hard_voxelize_workspace_bytes(cells, n, max_points)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    hip.call("pn_hard_voxelize_f32", points.data_ptr(), n, points.stride(0), f, (C.c_float * 3)(*vs.tolist()),
             (C.c_float * 6)(*rg.tolist()), int(max_points), int(max_voxels), voxels.data_ptr(), coors.data_ptr(),
             num.data_ptr(), nv.data_ptr(), ws.data_ptr(), ws_bytes, hip.stream())
    return voxels, coors, num, nv

# ------------------------------------------------------------------------------ next-3 target assignment
class CenterLossTargets:
    """device copies of one task's targets (example['hm'|'ind'|'mask'|'cat'|'anno_box'][t])"""

    def __init__(self, hm, ind, mask, cat, anno_box, device):
        self.hm = hm.to(device).float().contiguous()
        self.ind = ind.to(device).long().contiguous()
        self.mask = mask.to(device).to(torch.uint8).contiguous()
        self.cat = cat.to(device).long().contiguous()
        self.anno = anno_box.to(device).float().contiguous()


def assign_heatmap_polar(gt_boxes: torch.Tensor, gt_classes: torch.Tensor, num_gt: torch.Tensor, classes: int, max_objs: int,
                         feature_map_size, voxel_size, pc_range, out_size_factor: int, gaussian_overlap=0.1, min_radius=2,
                         rectify=False) -> CenterLossTargets:
    """gt_boxes (B, max_gt, 9) f32, gt_classes (B, max_gt) int32 (1-based), num_gt (B) int32, all on the device
    -> CenterLossTargets (hm, ind, mask, cat, anno_box) ready for center_loss / PolarPillarTrainStep.step"""
    hip.require_device(gt_boxes, gt_classes, num_gt)
    lib = hip.load()
    assert gt_boxes.dtype == torch.float32 and gt_classes.dtype == torch.int32 and num_gt.dtype == torch.int32
    assert gt_boxes.is_contiguous() and gt_classes.is_contiguous()
    b, max_gt, cols = gt_boxes.shape
    dev = gt_boxes.device
    fr, fa = int(feature_map_size[0]), int(feature_map_size[1])
    t = CenterLossTargets.__new__(CenterLossTargets)
    t.hm = torch.empty((b, classes, fa, fr), dtype=torch.float32, device=dev)
    t.ind = torch.empty((b, max_objs), dtype=torch.int64, device=dev)
    t.mask = torch.empty((b, max_objs), dtype=torch.uint8, device=dev)
    t.cat = torch.empty((b, max_objs), dtype=torch.int64, device=dev)
    t.anno = torch.empty((b, max_objs, 10), dtype=torch.float32, device=dev)
    nbytes = lib.pn_assign_heatmap_workspace_bytes(b, max_objs)
    ws = _workspace(nbytes, dev)
    hip.call("pn_assign_heatmap_polar_f32", gt_boxes.data_ptr(), gt_classes.data_ptr(), num_gt.data_ptr(), b, max_gt, cols, max_objs, classes,
             fr, fa, float(voxel_size[0]), float(voxel_size[1]), float(pc_range[0]), float(pc_range[1]), int(out_size_factor),
             float(gaussian_overlap), int(min_radius), int(bool(rectify)), t.hm.data_ptr(), t.ind.data_ptr(), t.mask.data_ptr(), t.cat.data_ptr(),
             t.anno.data_ptr(), ws.data_ptr(), nbytes, hip.stream())
    return t


# ------------------------------------------------------------------------------ next-4 sector streaming
def split_polar_sectors(points: torch.Tensor, sample_offsets: torch.Tensor, batch: int, nsectors: int, pc_range, voxel_size,
                        want_grid_ind=True, want_keys=False):
    """Voxelization.voxelize_streaming_polar (voxelization.py:305-393) on the device: polar points (N, F >= 5) of ``batch`` samples
    -> (points grouped by (sector, sample) in their original order, with phi shifted into the first sector and x / y recomputed;
    part offsets (nsectors * batch + 1,) int32 on the device; grid_ind (N, 4) int64 [b, z, theta, r] against the sector grid; keys)"""
    import numpy as np
    hip.require_device(points, sample_offsets)
    lib = hip.load()
    assert points.dtype == torch.float32 and points.is_contiguous() and sample_offsets.dtype == torch.int32
    n, f = points.shape
    rg, vs = np.asarray(pc_range, dtype=np.float32), np.asarray(voxel_size, dtype=np.float32)
    grid = np.round((rg[3:] - rg[:3]) / vs).astype(np.int64)
    dev = points.device
    out = torch.empty_like(points)
    offs = torch.empty((nsectors * batch + 1,), dtype=torch.int32, device=dev)
    gi = torch.empty((max(n, 1), 4), dtype=torch.int64, device=dev) if want_grid_ind else None
    keys = torch.empty((max(n, 1),), dtype=torch.int32, device=dev) if want_keys else None
    nbytes = lib.pn_split_polar_sectors_workspace_bytes(n, nsectors, batch)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    hip.call("pn_split_polar_sectors_f32", points.data_ptr(), n, f, sample_offsets.data_ptr(), batch, nsectors, (C.c_float * 6)(*rg.tolist()),
             (C.c_float * 3)(*vs.tolist()), (C.c_int32 * 3)(*[int(g) for g in grid]), out.data_ptr(), hip.ptr(gi), hip.ptr(keys), offs.data_ptr(),
             ws.data_ptr(), nbytes, hip.stream())
    return out, offs, gi, keys


def assemble_rows(samples, w: int, c: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``samples``: per output sample a list of up to three row pieces, each ``(None, rows)`` (zeros) or ``(tensor, sample, row0, rows)``
    taken from an NHWC map (B, H, W, C') with C' >= c -- the torch.cat / F.pad along the azimuth axis of rpn_context.py's
    convolutions.  -> NHWC (len(samples), sum(rows), w, c)"""
    n = len(samples)
    rows_out = sum(p[-1] for p in samples[0])
    arr = (hip.RowPiece * (3 * n))()
    dev = None
    for k, pieces in enumerate(samples):
        assert len(pieces) <= 3 and sum(p[-1] for p in pieces) == rows_out
        for j in range(3):
            e = arr[3 * k + j]
            if j >= len(pieces) or pieces[j][0] is None:
                e.src, e.pixel_stride, e.rows = None, 0, (pieces[j][-1] if j < len(pieces) else 0)
                continue
            t, smp, r0, rows = pieces[j]
            hip.require_device(t)
            assert t.dim() == 4 and t.is_contiguous() and t.shape[2] == w and t.shape[3] >= c and 0 <= r0 and r0 + rows <= t.shape[1]
            dev = t.device
            e.src = t.data_ptr() + 4 * ((smp * t.shape[1] + r0) * t.shape[2] * t.shape[3])
            e.pixel_stride, e.rows = t.shape[3], rows
    if out is None:
        out = torch.empty((n, rows_out, w, c), dtype=torch.float32, device=dev)
    hip.call("pn_assemble_rows_f32", arr, n, rows_out, w, c, out.data_ptr(), out.shape[3], 0, hip.stream())
    return out


# ------------------------------------------------------------------------------ next-4 sweep accumulation
def accumulate_sweeps(raw: torch.Tensor, sweep_offsets: torch.Tensor, transforms: torch.Tensor, time_lags: torch.Tensor, min_distance=1.0,
                      count: Optional[torch.Tensor] = None):
    """raw (n, >=4) f32 concatenated sweeps (key frame first), sweep_offsets (S+1) int32, transforms (S,4,4) float64, time_lags (S) f32,
    all on the device -> (out (n,5) f32 [x,y,z,intensity,dt] of which the first count rows are valid, count (1,) int32 on the device)"""
    hip.require_device(raw, sweep_offsets, transforms, time_lags)
    lib = hip.load()
    assert raw.is_contiguous() and raw.dtype == torch.float32 and transforms.dtype == torch.float64 and transforms.is_contiguous()
    n, cols = raw.shape
    out = torch.empty((n, 5), dtype=torch.float32, device=raw.device)
    if count is None:
        count = torch.empty(1, dtype=torch.int32, device=raw.device)
    nbytes = lib.pn_accumulate_sweeps_workspace_bytes(n)
    ws = _workspace(nbytes, raw.device)
    hip.call("pn_accumulate_sweeps_f32", raw.data_ptr(), n, cols, sweep_offsets.data_ptr(), transforms.shape[0], transforms.data_ptr(),
             time_lags.data_ptr(), float(min_distance), out.data_ptr(), count.data_ptr(), ws.data_ptr(), nbytes, hip.stream())
    return out, count
