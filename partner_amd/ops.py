"""Tensor-level wrappers over the C ABI (device memory from PyTorch, kernels from libpartner_hip).

Activations are NHWC fp32 tensors of shape (B, H, W, C); ``as_nchw`` / ``to_nhwc`` convert at the
det3d API boundary (a channels-last NCHW view is free).  Nothing here touches the oracle and
nothing falls back to PyTorch arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc  # noqa: F401


def _f32(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


# ------------------------------------------------------------------------------ layout
def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) logical tensor -> contiguous (B,H,W,C)."""
    hip.require_device(x)
    assert x.dim() == 4 and x.dtype == torch.float32
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous():
        return xp
    x = x.contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
    hip.call("pn_nchw_to_nhwc_f32", x.data_ptr(), b, c, h, w, out.data_ptr(), hip.stream())
    return out


def as_nchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    """(B,H,W,C) -> logical (B,C,H,W) view (channels-last strides, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def pixel_stride(t: torch.Tensor) -> int:
    """pixel stride (floats) of a logical (B, c, H, W) tensor that is a channels-last view (possibly a channel slice of a wider NHWC
    map); raises if it is not one.  A size-1 channel dimension may carry any stride."""
    b, c, h, w = t.shape
    ok = (c == 1 or t.stride(1) == 1) and t.stride(2) == w * t.stride(3) and (b == 1 or t.stride(0) == h * t.stride(2)) and t.stride(3) >= c
    if not ok:
        raise hip.PartnerHipError("head tensors must be channels-last (NHWC-backed) views")
    return t.stride(3)


def nhwc_slice_to_nchw(x_nhwc: torch.Tensor, c0: int, c: int) -> torch.Tensor:
    """contiguous NCHW copy of channels [c0, c0+c) of an NHWC tensor"""
    b, h, w, ct = x_nhwc.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=x_nhwc.device)
    hip.call("pn_nhwc_to_nchw_f32", x_nhwc.data_ptr(), b, c, h, w, ct, c0, out.data_ptr(), hip.stream())
    return out


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


_PILLAR_ROWS_ON = os.environ.get("PN_PILLAR_ROWS", "1") != "0"      # 0: the first convolution on (pillar, tap) pair lists (r2 - r5) everywhere


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
    if spec.grid[2] == 1 and spec.grid[0] % 8 == 0 and _PILLAR_ROWS_ON:
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


def clear_frame_cells(canvas: Optional[torch.Tensor], vi: VoxelIndex, state: Optional[FrameIndexState] = None, v_cap: Optional[int] = None) -> None:
    """sparse clear at the end of a frame: the canvas cells of the frame's voxels and their ``cell_count`` entries"""
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
                v_cap: Optional[int] = None) -> None:
    hip.require_device(points, w0, w1)
    assert w0.is_contiguous() and w1.is_contiguous() and points.is_contiguous()
    c0, c1 = w0.shape[0], w1.shape[0]
    assert w0.shape[1] == 16 and w1.shape[1] == 2 * c0
    _, _, g = vi.spec.c_arrays()
    tab = pfn_center_table(vi.spec.grid[1], vy, y_offset, points.device)
    hip.call("pn_dynamic_pfn_fwd_table", points.data_ptr(), points.stride(0), vi.voxel_start.data_ptr(), vi.order.data_ptr(),
             vi.num_voxels.data_ptr(), vi.n_cap if v_cap is None else v_cap, vi.unq_keys_ptr, g, w0.data_ptr(), c0,
             w1.data_ptr(), c1, float(vx), float(vy), float(x_offset), float(y_offset), tab.data_ptr(), hip.ptr(features),
             hip.ptr(canvas), hip.stream())


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


# ------------------------------------------------------------------------------ convolution
_FRAMES_IN_FLIGHT = 1


class frames_in_flight:
    """``with ops.frames_in_flight(n):`` -- the convolutions launched (or captured into a hipGraph) inside the block carry the hint that n
    independent frames run at the same time on other streams (pn_conv_desc.frames_in_flight); engine.FrameEngine wraps its capture in it"""

    def __init__(self, n: int):
        self.n, self.prev = max(1, int(n)), 1

    def __enter__(self):
        global _FRAMES_IN_FLIGHT
        self.prev, _FRAMES_IN_FLIGHT = _FRAMES_IN_FLIGHT, self.n
        return self

    def __exit__(self, *exc):
        global _FRAMES_IN_FLIGHT
        _FRAMES_IN_FLIGHT = self.prev
        return False


class ConvProfiler:
    """Execution time of every MFMA-conv launch (bench.py roofline): the event pair is attached to the kernel dispatch
    itself (``pn_profile_next_launch`` -> hipExtLaunchKernelGGL), so the elapsed time is the kernel's own duration -- the
    figure a rocprofv3 kernel trace reports -- and not the launch gap of an eager stream."""

    def __init__(self):
        self.lib = hip.load()
        self.pairs = []   # (start, stop, flops, tag)
        self.free = []

    def _event(self):
        if self.free:
            return self.free.pop()
        ev = C.c_void_p()
        hip.call("pn_event_create", C.byref(ev))
        return ev

    def begin(self, stream):
        a, b = self._event(), self._event()
        hip.call("pn_profile_next_launch", a, b)
        return (a, b)

    def end(self, evs, flops, stream, tag=None, issued=None, dense=None):
        """flops: what the launch is billed (dense-convolution count; the sparse first layer: the pairs it multiplies);
        issued: FLOPs that reach the MFMA (default: by the Winograd form named in the tag); dense: the direct dense algorithm's
        count (default = flops)"""
        if issued is None:
            issued = flops * (2.0 / 3.0 if tag and "F(2,3)" in tag else 0.5 if tag and "F(4,3)" in tag else 1.0)
        self.pairs.append((evs[0], evs[1], flops, tag, issued, flops if dense is None else dense))

    def collect(self, by_tag=False, full=False):
        """-> (total billed FLOPs, total milliseconds, launches[, {tag: (flops, ms, launches)}]); synchronises.
        full=True: the per-tag tuples are (flops, ms, launches, issued FLOPs, dense-algorithm FLOPs)"""
        flops, ms = 0.0, 0.0
        tags = {}
        out = C.c_float()
        for a, b, f, tag, iss, den in self.pairs:
            hip.call("pn_event_elapsed_ms", a, b, C.byref(out))
            ms += out.value
            flops += f
            t = tags.setdefault(tag, [0.0, 0.0, 0, 0.0, 0.0])
            t[0] += f
            t[1] += out.value
            t[2] += 1
            t[3] += iss
            t[4] += den
            self.free += [a, b]
        n = len(self.pairs)
        self.pairs = []
        if not by_tag:
            return (flops, ms, n)
        return (flops, ms, n, {k: (tuple(v) if full else tuple(v[:3])) for k, v in tags.items()})


_PROFILER: Optional[ConvProfiler] = None


def enable_conv_profiling() -> ConvProfiler:
    global _PROFILER
    _PROFILER = ConvProfiler()
    return _PROFILER


def disable_conv_profiling() -> None:
    global _PROFILER
    _PROFILER = None


_WINO_ON = os.environ.get("PN_CONV_WINO", "1") != "0"
_WINO_MIN_TILES = int(os.environ.get("PN_CONV_WINO_MIN_TILES", "256"))
# F(4, 3) (conv_wino4.hip): PN_CONV_WINO4=0 keeps F(2, 3); taken from this many 32-quad x 32-column tiles on (the kernel's K-split form
# runs one block per such tile, the plain form one per 32 quads x 128 columns when those fill the chip).  Measured against F(2, 3):
# 256 x 256 x 128 -> 128: 79 us / 112; 128 x 128 x 128 -> 128 (512 tiles): 27 / 31; 64 x 64 x 256 -> 256 (256 tiles): 31 / 37;
# Waymo RPN 256 x 144 x 128 -> 128: 61 / 105, 128 x 72 x 256 -> 256: 71 / 104
_WINO4_ON = os.environ.get("PN_CONV_WINO4", "1") != "0"
_TAPSUM_ON = os.environ.get("PN_CONV_TAPSUM", "1") != "0"   # 3x3 layers with <= 3 output channels over >= 128 input channels as GEMM + tap sum
_WINO4_MIN_TILES = int(os.environ.get("PN_CONV_WINO4_MIN_TILES", "256"))
_WINO4_DGRAD = os.environ.get("PN_CONV_WINO4_DGRAD", "1") != "0"     # F(4, 3) for the training data gradients (the forward stays on F(2, 3), see train.py)
_CHAIN2D_ON = os.environ.get("PN_CONV_CHAIN2D", "1") != "0"           # chained layers: F(2, 3) along the height on top of F(4, 3) along the width


class ConvLayer:
    """One packed convolution (+ per-channel affine + activation) on NHWC maps.

    weight: torch layout (Cout, Cin/groups, KH, KW), or (Cin, Cout, 2, 2) when ``deconv2x2``.
    scale / shift: per-output-channel affine (folded BatchNorm, or bias as shift)."""

    def __init__(self, weight: torch.Tensor, stride=1, pad=0, groups=1, scale=None, shift=None, act=ACT_NONE,
                 deconv2x2=False, range_strata=0, dtype="f32", wino4=True):
        hip.require_device(weight)
        lib = hip.load()
        w = weight.detach().contiguous().float()
        dev = w.device
        st = hip.stream()
        self.deconv2x2, self.range_strata, self.groups = bool(deconv2x2), int(range_strata), int(groups)
        self.stride, self.act = int(stride), int(act)
        self.pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
        assert dtype in ("f32", "bf16")
        self.dtype = dtype
        if dtype == "bf16":
            # bf16 activations / weights, f32 accumulate (pn_conv2d_nhwc_bf16)
            if deconv2x2:
                cin, cout = w.shape[0], w.shape[1]
                assert tuple(w.shape[2:]) == (2, 2)
                self.cin, self.cout, self.kh, self.kw = cin, cout, 1, 1
                wc = w.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1).contiguous()  # row (2*di+dj)*Cout + n
                self.packed = torch.empty(lib.pn_conv_packed_weight_bf16_elems(4 * cout, cin, 1, 1, 1), dtype=torch.bfloat16, device=dev)
                hip.call("pn_pack_conv_weight_bf16", wc.data_ptr(), 4 * cout, cin, 1, 1, 1, self.packed.data_ptr(), st)
                self._pack_bf16_rows(lib, wc, 4 * cout, cin, 1, 1, st)
            else:
                pack_groups = self.range_strata if self.range_strata > 1 else self.groups
                cout_t, cin_g, kh, kw = w.shape
                self.cin, self.cout, self.kh, self.kw = cin_g, cout_t // pack_groups, kh, kw
                self.packed = torch.empty(lib.pn_conv_packed_weight_bf16_elems(self.cout, cin_g, kh, kw, pack_groups), dtype=torch.bfloat16,
                                          device=dev)
                hip.call("pn_pack_conv_weight_bf16", w.data_ptr(), cout_t, cin_g, kh, kw, pack_groups, self.packed.data_ptr(), st)
                if pack_groups == 1:
                    self._pack_bf16_rows(lib, w, cout_t, cin_g, kh, kw, st)
        elif deconv2x2:
            cin, cout = w.shape[0], w.shape[1]
            assert tuple(w.shape[2:]) == (2, 2)
            self.cin, self.cout, self.kh, self.kw = cin, cout, 1, 1
            self.packed = _f32(lib.pn_deconv2x2_packed_weight_floats(cin, cout), dev)
            hip.call("pn_pack_deconv2x2_weight_f32", w.data_ptr(), cin, cout, self.packed.data_ptr(), st)
        else:
            pack_groups = self.range_strata if self.range_strata > 1 else self.groups
            cout_t, cin_g, kh, kw = w.shape
            self.cin, self.cout, self.kh, self.kw = cin_g, cout_t // pack_groups, kh, kw
            self.packed = _f32(lib.pn_conv_packed_weight_floats(self.cout, cin_g, kh, kw, pack_groups), dev)
            hip.call("pn_pack_conv_weight_f32", w.data_ptr(), cout_t, cin_g, kh, kw, pack_groups, self.packed.data_ptr(), st)
        self.scale = None if scale is None else scale.detach().contiguous().float()
        self.shift = None if shift is None else shift.detach().contiguous().float()
        self.out_channels = self.cout * (self.groups if not deconv2x2 else 1)
        self._pack_cin = self.cin
        # plain 3x3 / stride 1 / pad 1 layers also keep the width-Winograd F(2, 3) weights (conv_wino.hip: 6 instead of 9 MFMA
        # equivalents per output); used when the map is large enough to fill the chip with its 64-pair x 64-column tiles
        self.wino_packed = None
        if (dtype == "f32" and not deconv2x2 and self.range_strata <= 1 and self.groups == 1 and (self.kh, self.kw) == (3, 3)
                and self.stride == 1 and self.pad == (1, 1) and self.cin % 4 == 0 and _WINO_ON):
            self.wino_packed = _f32(lib.pn_conv_wino_packed_weight_floats(self.cout, self.cin), dev)
            hip.call("pn_pack_conv_weight_wino_f32", w.data_ptr(), self.cout, self.cin, self.wino_packed.data_ptr(), st)
        # ... and the F(4, 3) weights (4.5 MFMA equivalents per output) when the kernel's 32-column wave tiles fit the layer
        self.wino4_packed = None
        if self.wino_packed is not None and _WINO4_ON and wino4 and self.cout % 32 == 0 and self.act in (ACT_NONE, ACT_RELU):
            self.wino4_packed = _f32(lib.pn_conv_wino4_packed_weight_floats(self.cout, self.cin), dev)
            hip.call("pn_pack_conv_weight_wino4_f32", w.data_ptr(), self.cout, self.cin, self.wino4_packed.data_ptr(), st)
        # ... and the F(2, 3) x F(4, 3) weights of the chained form (conv_wchain.hip, ops.conv_chain: 3 MFMA equivalents per output)
        self.wino24_packed = None
        self._w_ref = None               # the layer's weight, for the packs of the TRANSPOSED kernel a chain on a transposed map takes (lazy)
        self._chain_t: dict = {}
        if self.wino4_packed is not None and _CHAIN2D_ON and self.cin % 32 == 0:
            self.wino24_packed = _f32(lib.pn_conv_wino24_packed_weight_floats(self.cout, self.cin), dev)
            hip.call("pn_pack_conv_weight_wino24_f32", w.data_ptr(), self.cout, self.cin, self.wino24_packed.data_ptr(), st)
            self._w_ref = w

        # ... and 3x3 layers with one to three output channels over many input channels (the geometry-aware head's 256 -> 1 heat-map and
        # vote-class convolutions): a GEMM over the pixels against the (9 cout, cin) tap matrix + a nine-term shifted sum
        # (pn_conv3x3_tap_sum_f32) -- the input is read once; on a 32-column MFMA tile these layers ran at 1.4 TFLOP/s
        self.tap_packed = None
        if (dtype == "f32" and not deconv2x2 and self.range_strata <= 1 and self.groups == 1 and (self.kh, self.kw) == (3, 3) and self.stride == 1
                and self.pad == (1, 1) and self.cout <= 3 and self.cin >= 128 and self.cin % 4 == 0 and _TAPSUM_ON and _LINEAR_ON):
            self.tap_n = (9 * self.cout + 3) // 4 * 4
            self.tap_packed = _f32(lib.pn_linear_packed_weight_floats(self.tap_n, self.cin), dev)
            self._pack_taps(w)

    def _pack_bf16_rows(self, lib, w: torch.Tensor, rows: int, cin: int, kh: int, kw: int, st) -> None:
        """the [row][tap][cin] weights of the bf16 implicit-GEMM kernel (csrc/conv_bf16.hip, r5): the layers of the Waymo BEV maps -- cin a
        multiple of 64, cout of 16 -- run there, every other bf16 layer on the general kernel (pn_conv2d_nhwc_bf16)"""
        self.packed_rows = None
        if cin % 64 == 0 and self.cout % 16 == 0 and self.act in (ACT_NONE, ACT_RELU):
            self.packed_rows = torch.empty(lib.pn_conv_bf16_rows_packed_elems(rows, cin, kh, kw), dtype=torch.bfloat16, device=w.device)
            hip.call("pn_pack_conv_weight_bf16_rows", w.data_ptr(), rows, cin, kh, kw, self.packed_rows.data_ptr(), st)

    def planes_desc(self, b: int, h: int, w: int, ct: int, in_channel_offset: int = 0):
        """the descriptor of ``to_planes`` on a (b, h, w, ct) map, or None where the direct kernel's planes epilogue does not apply"""
        if self.dtype != "f32" or self.deconv2x2 or self.range_strata > 1 or self.groups != 1 or not _CONV_PLANES_ON:
            return None
        d = ConvDesc(b, h, w, self.cin, self.cout, 1, self.kh, self.kw, self.stride, self.pad[0], self.pad[1], ct, in_channel_offset, self.cout, 0,
                     self.act, 0, 0, 0, 0, 0)
        return d if hip.load().pn_conv2d_nhwc_planes_supported(C.byref(d)) else None

    def to_planes(self, x: torch.Tensor, planes: torch.Tensor, in_channel_offset: int = 0) -> None:
        """this convolution (direct implicit-GEMM kernel) with its output written as the F(4, 3) planes of ``conv_chain`` (csrc/conv_mfma.hip,
        r6): the stride-2 layer at the head of an RPN block feeds the block's chain without the NHWC map in between"""
        hip.require_device(x, planes)
        b, h, w, ct = x.shape
        d = self.planes_desc(b, h, w, ct, in_channel_offset)
        assert d is not None, "ConvLayer.to_planes: check planes_desc first"
        self._ensure("direct")
        oh, ow = self.out_hw(h, w)
        assert planes.numel() >= hip.load().pn_wino4_planes_floats(b, oh, ow, self.cout)
        st = hip.stream()
        prof = _PROFILER
        if prof is not None:
            ev = prof.begin(st)
        hip.call("pn_conv2d_nhwc_planes_f32", C.byref(d), x.data_ptr(), self.packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                 planes.data_ptr(), st)
        if prof is not None:
            prof.end(ev, 2.0 * b * oh * ow * self.cout * self.cin * self.kh * self.kw, st, tag=f"{oh}x{ow} {self.cin}->{self.cout} k{self.kh} -> planes")

    def chain_weights(self, two_d: bool, transposed: bool) -> torch.Tensor:
        """packed weights of the chained F(4,3) / F(2,3)xF(4,3) forms; ``transposed``: of the kernel with kh and kw swapped (the chain then runs
        on the transposed map), packed on first use.  The transposed layouts belong to the inference path (the head's
        branch chain): they are not tracked by ``prepack_used`` and a stale one is repacked lazily on the caller's stream"""
        key = "wino44" if two_d == "wino44" else ("wino24" if two_d else "wino4")
        if not transposed:
            if key == "wino44" and getattr(self, "wino44_packed", None) is None:      # F(4,3) x F(4,3) (r5): packed on first use
                lib, w = hip.load(), getattr(self, "_stale_w", None)
                w = self._w_ref if w is None else w
                self.wino44_packed = _f32(lib.pn_conv_wino44_packed_weight_floats(self.cout, self._pack_cin), w.device)
                hip.call("pn_pack_conv_weight_wino44_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino44_packed.data_ptr(), hip.stream())
                self.__dict__.setdefault("_used", set()).add("wino44")
                (getattr(self, "_stale", None) or set()).discard("wino44")
                return self.wino44_packed
            self._ensure(key)
            return {"wino44": getattr(self, "wino44_packed", None), "wino24": self.wino24_packed, "wino4": self.wino4_packed}[key]
        stale = getattr(self, "_stale", None) or ()
        if key not in self._chain_t or (key + "_t") in stale:
            lib, w = hip.load(), getattr(self, "_stale_w", None)
            w = self._w_ref if w is None else w
            wt = w.detach().float().transpose(2, 3).contiguous()
            fam = key
            buf = self._chain_t.get(key)
            if buf is None:
                buf = _f32(getattr(lib, f"pn_conv_{fam}_packed_weight_floats")(self.cout, self._pack_cin), w.device)
            hip.call(f"pn_pack_conv_weight_{fam}_f32", wt.data_ptr(), self.cout, self._pack_cin, buf.data_ptr(), hip.stream())
            self._chain_t[key] = buf
            if not isinstance(getattr(self, "_chain_t_src", None), dict):
                self._chain_t_src = {}
            self._chain_t_src[key] = wt   # (the launch is asynchronous: the transposed copy stays referenced, one per layout)
            if stale:
                stale.discard(key + "_t")
        return self._chain_t[key]

    def _pack_taps(self, w: torch.Tensor) -> None:
        w9 = torch.zeros((self.tap_n, self.cin), dtype=torch.float32, device=w.device)
        w9[:9 * self.cout] = w.reshape(self.cout, -1, 9)[:, :self.cin].permute(2, 0, 1).reshape(9 * self.cout, -1)   # row t * cout + co
        hip.call("pn_pack_linear_weight_f32", w9.data_ptr(), self.tap_n, self.cin, self.tap_packed.data_ptr(), hip.stream())

    def repack(self, weight: torch.Tensor, shift: Optional[torch.Tensor] = None, token=None) -> None:
        """refresh the packed copies from an updated weight of the same shape (training: once per step).  LAZY: a layout (direct,
        F(2, 3), F(4, 3)) is packed when the next call takes it -- a layer keeps up to three and uses one per map size, and the
        tiny pack launches were 0.8 ms of an 18.8 ms training iteration.  ``weight`` must stay valid (and unchanged) until then: the
        training steps hand in views of their flat parameter buffer, which the optimizer rewrites only after the backward.
        ``token``: a call with the token of the previous call is a no-op (the training steps refresh every layer once at the start of an
        iteration, ``prepack_used``, and pass the iteration's token from the layers' forward / backward)"""
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        w = weight.detach()
        assert w.is_contiguous() and w.dtype == torch.float32
        self._stale_w = w
        self._stale = {"direct"} | ({"wino"} if self.wino_packed is not None else set()) | ({"wino4"} if self.wino4_packed is not None else set())
        if self.tap_packed is not None:
            self._stale.add("tap")
        if getattr(self, "wino24_packed", None) is not None:
            self._stale.add("wino24")
        if getattr(self, "wino44_packed", None) is not None:
            self._stale.add("wino44")
        for k in getattr(self, "_chain_t", {}):
            self._stale.add(k + "_t")
        if shift is not None:
            self.shift = shift

    def prepack_used(self) -> None:
        """pack, on the current stream, the layouts the layer's calls have taken so far (after ``repack``)"""
        for layout in sorted(getattr(self, "_used", ())):
            self._ensure(layout)

    def _ensure(self, layout: str) -> None:
        if not layout.endswith("_t"):
            self.__dict__.setdefault("_used", set()).add(layout)
        stale = getattr(self, "_stale", None)
        if not stale or layout not in stale:
            return
        stale.discard(layout)
        w, st = self._stale_w, hip.stream()
        if layout == "direct":
            if self.deconv2x2:
                hip.call("pn_pack_deconv2x2_weight_f32", w.data_ptr(), self._pack_cin, self.cout, self.packed.data_ptr(), st)
            else:
                pack_groups = self.range_strata if self.range_strata > 1 else self.groups
                hip.call("pn_pack_conv_weight_f32", w.data_ptr(), w.shape[0], self._pack_cin, self.kh, self.kw, pack_groups,
                         self.packed.data_ptr(), st)
        elif layout == "tap":
            self._pack_taps(w)
        elif layout == "wino":
            hip.call("pn_pack_conv_weight_wino_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino_packed.data_ptr(), st)
        elif layout == "wino24":
            hip.call("pn_pack_conv_weight_wino24_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino24_packed.data_ptr(), st)
        elif layout == "wino44":
            hip.call("pn_pack_conv_weight_wino44_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino44_packed.data_ptr(), st)
        else:
            hip.call("pn_pack_conv_weight_wino4_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino4_packed.data_ptr(), st)

    def _use_wino(self, b: int, h: int, w: int, accumulate: bool) -> bool:
        if self.wino_packed is None or accumulate or w % 2 or self.cin != self._pack_cin:
            return False
        tiles = ((b * h * (w // 2) + 31) // 32) * ((self.cout + 63) // 64)   # 32-pair x 64-column tiles (the kernel takes 64-pair ones when they fill the chip)
        return tiles >= _WINO_MIN_TILES

    def _use_wino4(self, b: int, h: int, w: int, accumulate: bool) -> bool:   # (the tile-count gate below is about leaving the direct kernel, not about the form)
        if self.wino4_packed is None or accumulate or w % 4 or self.cin != self._pack_cin or h * w > getattr(self, "wino4_max_pixels", 1 << 62):
            return False
        return ((b * h * (w // 4) + 31) // 32) * (self.cout // 32) >= _WINO4_MIN_TILES

    def pad_input_channels(self, cin_padded: int) -> "ConvLayer":
        """declare that the input map carries zero pad channels up to a multiple of 4 (e.g. the 5-channel
        position encoding stored with 8): the packed rows past the real Cin are already zero"""
        assert cin_padded >= self.cin and cin_padded % 4 == 0 and (cin_padded + 31) // 32 == (self.cin + 31) // 32
        self.cin = cin_padded
        return self

    def out_hw(self, h: int, w: int) -> Tuple[int, int]:
        if self.deconv2x2:
            return 2 * h, 2 * w
        return ((h + 2 * self.pad[0] - self.kh) // self.stride + 1, (w + 2 * self.pad[1] - self.kw) // self.stride + 1)

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, out_channel_offset=0, in_channel_offset=0,
                 in_channels: Optional[int] = None, accumulate=False, out_f32=False, out_transposed=False) -> torch.Tensor:
        """x: NHWC (B,H,W,Ct).  Reads channels [in_channel_offset, +cin*groups); writes channels
        [out_channel_offset, +out_channels) of ``out`` (allocated if None).  bf16 layers take / return
        torch.bfloat16 maps (``out_f32``: f32 output).  ``out_transposed`` (layers on the F(4,3) kernel only): ``out`` is (B, W, H, C), the
        map stored transposed (pn_conv_desc.transpose_hw)."""
        hip.require_device(x)
        in_dt = torch.bfloat16 if self.dtype == "bf16" else torch.float32
        assert x.dim() == 4 and x.is_contiguous() and x.dtype == in_dt
        b, h, w, ct = x.shape
        oh, ow = self.out_hw(h, w)
        if out is None:
            out = torch.empty((b, oh, ow, self.out_channels), dtype=torch.float32 if (out_f32 or self.dtype == "f32") else torch.bfloat16,
                              device=x.device)
        if self.dtype == "bf16":
            assert out.shape[:3] == (b, oh, ow) and out.is_contiguous()
            d = ConvDesc(b, h, w, self.cin, self.cout, self.groups, self.kh, self.kw, self.stride, self.pad[0], self.pad[1],
                         ct, in_channel_offset, out.shape[3], out_channel_offset, self.act, int(self.deconv2x2), self.range_strata)
            st = hip.stream()
            prof = _PROFILER
            if prof is not None:
                ev = prof.begin(st)
            igemm = getattr(self, "packed_rows", None) is not None and hip.load().pn_conv2d_igemm_bf16_supported(C.byref(d)) == 1
            if igemm:
                hip.call("pn_conv2d_igemm_bf16", C.byref(d), x.data_ptr(), self.packed_rows.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                         out.data_ptr(), int(out.dtype == torch.float32), st)
            else:
                hip.call("pn_conv2d_nhwc_bf16", C.byref(d), x.data_ptr(), self.packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                         out.data_ptr(), int(out.dtype == torch.float32), st)
            if prof is not None:
                macs = b * h * w * 4 * self.cout * self.cin if self.deconv2x2 else b * oh * ow * self.groups * self.cout * self.cin * self.kh * self.kw
                prof.end(ev, 2.0 * macs, st, tag=f"{oh}x{ow} {self.cin * self.groups}->{self.out_channels} k{self.kh} bf16")
            return out
        assert out.shape[:3] == ((b, ow, oh) if out_transposed else (b, oh, ow)) and out.is_contiguous()
        d = ConvDesc(b, h, w, self.cin, self.cout, self.groups, self.kh, self.kw, self.stride, self.pad[0], self.pad[1],
                     ct, in_channel_offset, out.shape[3], out_channel_offset, self.act, int(self.deconv2x2),
                     self.range_strata, 0, 0, int(accumulate))
        d.frames_in_flight = _FRAMES_IN_FLIGHT
        d.transpose_hw = int(bool(out_transposed))
        st = hip.stream()
        prof = _PROFILER
        if prof is not None:
            ev = prof.begin(st)
        use_tap = self.tap_packed is not None and not accumulate and self.cin == self._pack_cin and in_channel_offset % 4 == 0 and ct % 4 == 0
        use_wino4 = not use_tap and self._use_wino4(b, h, w, accumulate)
        use_wino = not use_tap and not use_wino4 and self._use_wino(b, h, w, accumulate)
        assert use_wino4 or not out_transposed, "ConvLayer: a transposed output needs the F(4,3) kernel (check _use_wino4 first)"
        self._ensure("tap" if use_tap else "wino4" if use_wino4 else "wino" if use_wino else "direct")
        if use_tap:
            m = b * h * w
            g = torch.empty((m, self.tap_n), dtype=torch.float32, device=x.device)
            hip.call("pn_linear_ksplit_f32", x.data_ptr() + 4 * in_channel_offset, m, self.cin, ct, self.tap_packed.data_ptr(), self.tap_n, None, ACT_NONE,
                     None, self.tap_n, g.data_ptr(), self.tap_n, st)
            hip.call("pn_conv3x3_tap_sum_f32", g.data_ptr(), self.tap_n, b, h, w, self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act,
                     out.data_ptr(), out.shape[3], out_channel_offset, st)
            if prof is not None:
                prof.end(ev, 2.0 * m * self.cout * self.cin * 9, st, tag=f"{oh}x{ow} {self.cin}->{self.out_channels} k3 gemm+taps", issued=2.0 * m * self.cin * 32)
            return out
        if use_wino4:
            hip.call("pn_conv2d_wino4_nhwc_f32", C.byref(d), x.data_ptr(), self.wino4_packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                     out.data_ptr(), st)
        elif use_wino:
            hip.call("pn_conv2d_wino_nhwc_f32", C.byref(d), x.data_ptr(), self.wino_packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                     out.data_ptr(), st)
        else:
            hip.call("pn_conv2d_nhwc_f32", C.byref(d), x.data_ptr(), self.packed.data_ptr(), hip.ptr(self.scale),
                     hip.ptr(self.shift), out.data_ptr(), st)
        if prof is not None:
            # algorithmic FLOPs = 2 * output pixels * Cout * Cin * KH * KW (per group), dense-conv count
            if self.deconv2x2:
                macs = b * h * w * 4 * self.cout * self.cin
            else:
                z = self.groups
                macs = b * oh * ow * z * self.cout * self.cin * self.kh * self.kw
            prof.end(ev, 2.0 * macs, st, tag=f"{oh}x{ow} {self.cin * self.groups}->{self.out_channels} k{self.kh}{'t' if self.deconv2x2 else ''}{'s' if self.range_strata > 1 else ''}{' F(2,3)' if use_wino else ' F(4,3)' if use_wino4 else ''}")
        return out


# Runs of same-map 3x3 / stride-1 layers kept in the F(4, 3) domain between layers (csrc/conv_wchain.hip): PN_CONV_CHAIN=0 runs them one
# pn_conv2d_wino4_nhwc_f32 launch each, as r3 did
_CHAIN_ON = os.environ.get("PN_CONV_CHAIN", "1") != "0"
_CONV_PLANES_ON = os.environ.get("PN_CONV_PLANES", "1") != "0"     # 0: a block's first (stride-2) layer writes NHWC and a separate pass forms the chain's planes (r4, r5)


def _chain_desc(layer: "ConvLayer", b: int, h: int, w: int, out_ps: int = 0, out_co: int = 0, transposed: bool = False):
    d = ConvDesc(b, h, w, layer.cin, layer.cout, 1, 3, 3, 1, 1, 1, layer.cin, 0, out_ps or layer.cout, out_co, layer.act, 0, 0, 0, 0, 0)
    d.frames_in_flight = _FRAMES_IN_FLIGHT
    d.transpose_hw = int(transposed)
    return d


def _chain_orientation(layers, b: int, h: int, w: int):
    """None, or whether the chain works on the transposed map (False: the Winograd axis is W; True: it is H -- maps like the Waymo BEV's
    256 x 144, whose W / 4 = 36 is not a power of two)"""
    if not _CHAIN_ON or not layers:
        return None
    lib = hip.load()
    for k, l in enumerate(layers):
        if l.dtype != "f32" or l.wino4_packed is None or l.cin != l._pack_cin or (k and l.cin != layers[k - 1].cout):
            return None
    for transposed in (False, True):
        if transposed and any(getattr(l, "_w_ref", None) is None for l in layers):
            break
        if all(lib.pn_conv_wino4_chain_supported(C.byref(_chain_desc(l, b, h, w, transposed=transposed))) for l in layers):
            return transposed
    return None


def conv_chain_orientation(layers, b: int, h: int, w: int):
    """None: the layers cannot run as a chain on a (b, h, w) map; False / True: they can, on the map as stored / transposed"""
    return _chain_orientation(layers, b, h, w)


def conv_chain_supported(layers, b: int, h: int, w: int) -> bool:
    """can ``layers`` (consecutive ConvLayers, each feeding the next) run as one Winograd-domain chain on a (b, h, w) map?"""
    return _chain_orientation(layers, b, h, w) is not None


def conv_chain(layers, x: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, out_channel_offset=0, in_channel_offset=0, planes_from=None,
               shape=None, device=None) -> torch.Tensor:
    """x NHWC (B, H, W, Ct) -> the NHWC output of the last layer.  One launch forms the six F(4, 3) planes of x, then every layer reads
    planes and writes planes (two buffers, alternating); the last one writes the map.  Same arithmetic as the layers one by one
    (ConvLayer.__call__ on pn_conv2d_wino4_nhwc_f32) up to the summation order over the input channels.
    ``planes_from(buffer)`` (with ``shape`` = (B, H, W), ``device``, x None): the producer writes the NOT transposed planes of the first
    layer's input itself (PillarConvLayer: the map never exists in NHWC)."""
    if planes_from is None:
        hip.require_device(x)
        assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
        b, h, w, ct = x.shape
        dev = x.device
    else:
        (b, h, w), dev = shape, device
    lib, st = hip.load(), hip.stream()
    tr = _chain_orientation(layers, b, h, w)
    assert tr is not None, "conv_chain: check conv_chain_supported first"
    assert planes_from is None or not tr, "conv_chain: a planes producer writes the map's own orientation"
    cmax = max([layers[0].cin] + [l.cout for l in layers[:-1]])
    n = lib.pn_wino4_planes_floats(b, w if tr else h, h if tr else w, cmax)
    bufs = [torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev) if len(layers) > 1 else None]
    if planes_from is None:
        hip.call("pn_wino4_planes_from_nhwc_f32", x.data_ptr(), b, h, w, layers[0].cin, ct, in_channel_offset, int(tr), bufs[0].data_ptr(), st)
    else:
        planes_from(bufs[0])
    last = layers[-1]
    if out is None:
        out = torch.empty((b, h, w, last.cout), dtype=torch.float32, device=dev)
    assert out.shape[:3] == (b, h, w) and out.is_contiguous()
    prof = _PROFILER
    for k, l in enumerate(layers):
        is_last = k == len(layers) - 1
        d = _chain_desc(l, b, h, w, out.shape[3], out_channel_offset, transposed=tr) if is_last else _chain_desc(l, b, h, w, transposed=tr)
        two_d = l.wino24_packed is not None and _chain_two_d(lib, d)
        if two_d and _chain_44(lib, d):
            two_d = "wino44"
            _CHAIN44_LAUNCHES[0] += 1
        wts = l.chain_weights(two_d, tr)
        if prof is not None:
            ev = prof.begin(st)
        hip.call("pn_conv2d_wino44_chain_f32" if two_d == "wino44" else ("pn_conv2d_wino24_chain_f32" if two_d else "pn_conv2d_wino4_chain_f32"), C.byref(d), bufs[k & 1].data_ptr(),
                 wts.data_ptr(), hip.ptr(l.scale), hip.ptr(l.shift),
                 None if is_last else bufs[(k + 1) & 1].data_ptr(), out.data_ptr() if is_last else None, st)
        if prof is not None:
            flops = 2.0 * b * h * w * l.cout * l.cin * 9
            tag = "F(4,3)xF(4,3) chain" if two_d == "wino44" else ("F(2,3)xF(4,3) chain" if two_d else "F(4,3) chain")
            prof.end(ev, flops, st, tag=f"{h}x{w} {l.cin}->{l.cout} k3 {tag}", issued=flops / 4.0 if two_d == "wino44" else (flops / 3.0 if two_d else None))
    return out


def _chain_two_d(lib, d) -> bool:
    """the chained layer's form: F(2,3) x F(4,3) where the kernel has a 12-wave form for the map; on maps whose 2-D tiles cover less than
    3/4 of the CUs (64 x 64 x 256: 128 blocks) only when other frames run beside this one (they take the free CUs, and the 2-D form
    issues 1.5x fewer MFMAs); alone on the chip the 1-D form's 256 blocks finish sooner"""
    if not _CHAIN2D_ON or not lib.pn_conv_wino24_chain_supported(C.byref(d)):
        return False
    fh, fw = (d.in_w, d.in_h) if d.transpose_hw else (d.in_h, d.in_w)
    octs, wq = d.batch * (fh // 2) * (fw // 4), fw // 4
    blocks = (octs // (64 if wq > 32 else 32)) * (d.cout // 32)
    return blocks >= 192 or d.frames_in_flight > 1


_CHAIN44_ON = os.environ.get("PN_CONV_CHAIN44", "1") != "0"            # chained layers on 128-pixel rows: F(4, 3) along the height as well where it pays


class chain44:
    """``with ops.chain44(False):`` -- the chained layers launched (or captured) inside the block take F(2,3)xF(4,3) where the
    frames-in-flight hint alone would pick F(4,3)xF(4,3) (``True``: the default rule of ``_chain_44``).  Which form is faster with other
    frames in flight differs from box to box (r5: +2 .. 3 % on the builder's boxes, -4.5 % on the driver's), so engine.FramePipeline
    captures both and keeps the one it MEASURES faster; PN_CONV_CHAIN44=0 still forces the form off process-wide."""

    def __init__(self, on: bool):
        self.on, self.prev = bool(on), True

    def __enter__(self):
        global _CHAIN44_ROUTE
        self.prev, _CHAIN44_ROUTE = _CHAIN44_ROUTE, self.on
        return self

    def __exit__(self, *exc):
        global _CHAIN44_ROUTE
        _CHAIN44_ROUTE = self.prev
        return False


_CHAIN44_ROUTE = True
_CHAIN44_LAUNCHES = [0]


def chain44_launches_seen() -> int:
    """how many chained-layer launches took the F(4,3)xF(4,3) form so far in this process (engine.FramePipeline: is there a choice to measure?)"""
    return _CHAIN44_LAUNCHES[0]


def _chain_44(lib, d) -> bool:
    """F(4,3) x F(4,3) (conv_wchain3_kernel, r5: 2.25 MFMA equivalents per output; one block per four rows x 128 pixels x 32 channels, a 256-pixel
    row as two such halves one after the other): WITH OTHER FRAMES IN FLIGHT, where its blocks are whole rounds of the 256 CUs (256 x 256 x 128:
    256 blocks, 55 against 62 us per layer; a batch of four 128 x 128 maps 55 against 72) or at least half a round (one 128 x 128 map, 128 blocks:
    26 against 30 us for the form the hint picks otherwise; alone on the chip the 256-block K-split form finishes in 21).  Not for a frame
    alone on the chip: the 256 x 256 launch itself is 4 - 9 us shorter there too, but on two of the four boxes it was measured on every OTHER
    matrix kernel of the frame then ran 4 - 5 % longer (853.8 against 843.1 us of kernel time per frame; on the other boxes 796 against 812) --
    the one-frame latency moved by -20 .. +17 us with the box, the in-flight rate rose on all of them (+2 .. 3 %)"""
    if not (_CHAIN44_ON and _CHAIN44_ROUTE) or d.frames_in_flight <= 1 or not lib.pn_conv_wino44_chain_supported(C.byref(d)):
        return False
    fh, fw = (d.in_w, d.in_h) if d.transpose_hw else (d.in_h, d.in_w)
    tq = 64 if fw // 4 == 64 else 32
    blocks = (d.batch * (fh // 4) * (fw // 4) // tq) * (d.cout // 32)
    return blocks % 256 == 0 or 128 <= blocks <= 256


_PILLAR_CONV_ON = os.environ.get("PN_PILLAR_CONV", "1") != "0"
_PILLAR_PLANES_ON = os.environ.get("PN_PILLAR_PLANES", "1") != "0"      # the pillar layer writes the chain's planes itself (no NHWC map in between)
# taken when the pillar capacity bounds the (pillar, tap) pairs to this fraction of the dense (output, tap) pairs
_PILLAR_CONV_MAX_FILL = float(os.environ.get("PN_PILLAR_CONV_MAX_FILL", "0.35"))
_PILLAR_ROWS_MAX_FILL = float(os.environ.get("PN_PILLAR_ROWS_MAX_FILL", "1.25"))


class PillarConvLayer:
    """The backbone's first 3x3 convolution on the SPARSE pillar canvas (csrc/pillar_conv.hip; rpn.py:124-142 on the canvas of
    pillar_encoder.py:393-432): (pillar, tap) pairs -> one gathered MFMA GEMM per tap -> fixed-order reduction over the taps with the
    folded BatchNorm + activation.  ``__call__(canvas, vi)``: ``vi`` = the frame's VoxelIndex; every non-zero pixel of ``canvas``
    must be one of its cells."""

    def __init__(self, weight: torch.Tensor, stride: int, scale=None, shift=None, act=ACT_NONE):
        hip.require_device(weight)
        lib = hip.load()
        w = weight.detach().contiguous().float()
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        assert tuple(w.shape[2:]) == (3, 3) and stride in (1, 2) and self.cin in (32, 64, 128) and self.cout % 4 == 0
        self.stride, self.act = int(stride), int(act)
        self.packed = _f32(lib.pn_pillar_conv_packed_weight_floats(self.cout, self.cin), w.device)
        hip.call("pn_pack_pillar_conv_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), hip.stream())
        self.scale = None if scale is None else scale.detach().contiguous().float()
        self.shift = None if shift is None else shift.detach().contiguous().float()
        self.packed_rows = None
        if self.stride == 2 and self.cin % 16 == 0 and self.cout <= 128:      # the row-band form (csrc/pillar_rows.hip) where the frame index leaves row_start
            self.packed_rows = _f32(lib.pn_pillar_conv_rows_packed_weight_floats(self.cout, self.cin), w.device)
            hip.call("pn_pack_pillar_conv_rows_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed_rows.data_ptr(), hip.stream())

    def rows_form(self, vi: "VoxelIndex", b: int, h: int, w: int) -> bool:
        """does this frame take the row-band kernel?  (the fused frame index left row_start and the shape is covered)"""
        return (self.packed_rows is not None and getattr(vi, "row_start", None) is not None and _PILLAR_ROWS_ON
                and bool(hip.load().pn_pillar_conv_rows_supported(b, h, w, self.cin, self.cout, self.stride)))

    @staticmethod
    def supports(conv_weight: torch.Tensor, stride: int, groups: int) -> bool:
        co, ci, kh, kw = conv_weight.shape
        return _PILLAR_CONV_ON and (kh, kw) == (3, 3) and stride in (1, 2) and groups == 1 and ci in (32, 64, 128) and co % 4 == 0

    def worth_it(self, vi: "VoxelIndex", b: int, h: int, w: int) -> bool:
        """the pillar capacity (known on the host: no sync) bounds the pairs: 9 / stride^2 per pillar.  The pair-list form pays up to 0.35 of
        the dense (output, tap) pairs; the row-band form (no partial rows in memory, the planes straight from LDS) still wins on the 300k-point
        frames of BASELINE configs[4] -- 180k pillars, 2/3 of the cells: p50 1.208 -> 1.186 ms -- so it is taken up to a capacity bound of 1.25
        (the capacity counts points, not pillars; a completely full canvas would lose about a third on this one layer)."""
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        limit = max(_PILLAR_CONV_MAX_FILL, _PILLAR_ROWS_MAX_FILL) if self.rows_form(vi, b, h, w) else _PILLAR_CONV_MAX_FILL
        return vi.n_cap * 9.0 / (self.stride * self.stride) <= limit * 9.0 * b * oh * ow

    def planes_supported(self, b: int, h: int, w: int) -> bool:
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        return _PILLAR_PLANES_ON and bool(hip.load().pn_pillar_conv_planes_supported(b, oh, ow, self.cout))

    def __call__(self, canvas: torch.Tensor, vi: "VoxelIndex", out: Optional[torch.Tensor] = None, planes: Optional[torch.Tensor] = None):
        """-> the NHWC output, or (``planes`` given: a buffer of pn_wino4_planes_floats(b, oh, ow, cout) floats) None with the output written as
        the F(4, 3) planes of ops.conv_chain"""
        hip.require_device(canvas)
        lib = hip.load()
        assert canvas.dim() == 4 and canvas.is_contiguous() and canvas.dtype == torch.float32 and canvas.shape[3] >= self.cin
        b, h, w, ct = canvas.shape
        assert (w, h) == (vi.spec.grid[0], vi.spec.grid[1]) and b == vi.batch and vi.spec.grid[2] == 1, "the voxel index does not describe this canvas"
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        if out is None and planes is None:
            out = torch.empty((b, oh, ow, self.cout), dtype=torch.float32, device=canvas.device)
        if self.rows_form(vi, b, h, w):
            st = hip.stream()
            prof = _PROFILER
            if prof is not None:
                ev = prof.begin(st)
            if planes is not None:
                assert planes.numel() >= lib.pn_wino4_planes_floats(b, oh, ow, self.cout)
            hip.call("pn_pillar_conv3x3_rows_f32", canvas.data_ptr(), b, h, w, self.cin, ct, 0, vi.unq_keys_ptr, vi.row_start.data_ptr(), vi.n_cap,
                     self.packed_rows.data_ptr(), self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act, hip.ptr(planes),
                     None if planes is not None else out.data_ptr(), 0 if planes is not None else out.shape[3], 0, st)
            if prof is not None:
                # FLOPs actually multiplied: the frame's (pillar, tap) pairs, counted on the host from the key list (profiling runs only)
                v = vi.count()
                k = vi.workspace[:v].to(torch.int64) & 0xffffffff
                ix, iy = k % w, (k // w) % h
                tx = torch.where(ix % 2 == 0, 1, 1 + ((ix + 1) // 2 < ow).long())
                ty = torch.where(iy % 2 == 0, 1, 1 + ((iy + 1) // 2 < oh).long())
                pairs = int((tx * ty).sum())
                prof.end(ev, 2.0 * pairs * self.cout * self.cin, st, tag=f"{oh}x{ow} {self.cin}->{self.cout} k3 pillars (row bands)",
                         dense=2.0 * b * oh * ow * 9 * self.cout * self.cin)
            return out
        nbytes = lib.pn_pillar_conv_workspace_bytes(vi.n_cap, b, oh, ow, self.cout)
        ws = _workspace(nbytes, canvas.device)
        st = hip.stream()
        prof = _PROFILER
        if prof is not None:
            ev = prof.begin(st)
        if planes is not None:
            assert planes.numel() >= lib.pn_wino4_planes_floats(b, oh, ow, self.cout)
            hip.call("pn_pillar_conv3x3_planes_f32", canvas.data_ptr(), b, h, w, self.cin, ct, 0, vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap,
                     self.stride, self.packed.data_ptr(), self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act, planes.data_ptr(),
                     ws.data_ptr(), nbytes, st)
        else:
            hip.call("pn_pillar_conv3x3_f32", canvas.data_ptr(), b, h, w, self.cin, ct, 0, vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap, self.stride,
                     self.packed.data_ptr(), self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act, out.data_ptr(), out.shape[3], 0,
                     ws.data_ptr(), nbytes, st)
        if prof is not None:
            # FLOPs actually multiplied: the (pillar, tap) pairs of THIS frame (the nine counters head the workspace; reading them
            # synchronises -- profiling runs only); the events bracket the pair, GEMM and reduce kernels
            pairs = int(ws[:36].view(torch.int32).sum().item())
            prof.end(ev, 2.0 * pairs * self.cout * self.cin, st, tag=f"{oh}x{ow} {self.cin}->{self.cout} k3 pillars",
                     dense=2.0 * b * oh * ow * 9 * self.cout * self.cin)
        return out

    # ---- training: pair tables built once per iteration, shared by forward, data gradient and weight gradient
    def repack(self, weight: torch.Tensor, token=None) -> None:
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        w = weight.detach().contiguous().float()
        hip.call("pn_pack_pillar_conv_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), hip.stream())
        if getattr(self, "packed_t", None) is None:
            self.packed_t = _f32(hip.load().pn_pillar_conv_packed_weight_floats(self.cin, self.cout), w.device)
        wt = w.permute(1, 0, 2, 3).contiguous()          # (Cin, Cout, 3, 3): the data gradient multiplies by W_tap^T, same tap
        hip.call("pn_pack_pillar_conv_weight_f32", wt.data_ptr(), self.cin, self.cout, self.packed_t.data_ptr(), hip.stream())

    def build_tables(self, vi: "VoxelIndex", b: int, h: int, w: int) -> torch.Tensor:
        lib = hip.load()
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        nbytes = lib.pn_pillar_pairs_bytes(vi.n_cap, b, oh, ow)
        tables = torch.empty(nbytes, dtype=torch.uint8, device=vi.num_voxels.device)
        hip.call("pn_pillar_pairs_build", vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap, b, h, w, self.stride, tables.data_ptr(), nbytes, hip.stream())
        return tables

    def forward_tables(self, canvas: torch.Tensor, vi: "VoxelIndex", tables: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, h, w, ct = canvas.shape
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        if out is None:
            out = torch.empty((b, oh, ow, self.cout), dtype=torch.float32, device=canvas.device)
        cap = (vi.n_cap + 127) // 128 * 128
        ws = _workspace(9 * cap * self.cout * 4, canvas.device)
        hip.call("pn_pillar_conv3x3_tables_f32", canvas.data_ptr(), b, oh, ow, self.cin, ct, 0, tables.data_ptr(), vi.n_cap, self.packed.data_ptr(), self.cout,
                 hip.ptr(self.scale), hip.ptr(self.shift), self.act, out.data_ptr(), out.shape[3], 0, ws.data_ptr(), ws.numel(), hip.stream())
        return out

    def dgrad_features(self, dout: torch.Tensor, vi: "VoxelIndex", tables: torch.Tensor) -> torch.Tensor:
        """d(pillar features) (n_cap, Cin), rows in the order of the index's cells (what ``dynamic_pfn_bwd`` takes as d_features)"""
        hip.require_device(dout)
        assert dout.is_contiguous() and dout.shape[3] >= self.cout
        b, oh, ow, ct = dout.shape
        dfeat = torch.empty((max(vi.n_cap, 1), self.cin), dtype=torch.float32, device=dout.device)
        cap = (vi.n_cap + 127) // 128 * 128
        ws = _workspace(9 * cap * self.cin * 4, dout.device)
        hip.call("pn_pillar_conv3x3_dgrad_f32", dout.data_ptr(), b, oh, ow, self.cout, ct, 0, tables.data_ptr(), vi.num_voxels.data_ptr(), vi.n_cap,
                 self.packed_t.data_ptr(), self.cin, dfeat.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream())
        return dfeat

    def wgrad(self, canvas: torch.Tensor, dout: torch.Tensor, vi: "VoxelIndex", tables: torch.Tensor, out: Optional[torch.Tensor] = None,
              accumulate=False) -> torch.Tensor:
        lib = hip.load()
        b, oh, ow, ct = dout.shape
        if out is None:
            out = torch.empty((self.cout, self.cin, 3, 3), dtype=torch.float32, device=dout.device)
        nbytes = lib.pn_pillar_conv_wgrad_workspace_bytes(vi.n_cap, self.cin, self.cout)
        ws = _workspace(nbytes, dout.device)
        hip.call("pn_pillar_conv3x3_wgrad_f32", canvas.data_ptr(), canvas.shape[3], 0, self.cin, dout.data_ptr(), ct, 0, self.cout, tables.data_ptr(), vi.n_cap,
                 b, oh, ow, out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
        return out


class ConvJob:
    """One convolution of a multi-job launch (``conv_multi``): a packed ``ConvLayer`` applied to a channel slice of ``x``,
    writing a channel slice of ``out``; optionally emitting the statistics of the GroupNorm-family layer that follows it
    (``stats=dict(strata, channel_groups, gamma, beta, eps, affine=, affine_strata=, mean_rstd=)``) and / or reading its input
    through the affine table of the norm that precedes it (``norm=(table, strata, channels)``: relu(x*A + B) on load)."""

    def __init__(self, layer: "ConvLayer", x: torch.Tensor, out: torch.Tensor, in_channel_offset=0, out_channel_offset=0, stats=None,
                 norm=None):
        assert layer.dtype == "f32" and not layer.deconv2x2
        hip.require_device(x, out)
        assert x.dim() == 4 and x.is_contiguous() and out.is_contiguous() and x.dtype == torch.float32
        self.layer, self.x, self.out, self.in_co, self.out_co, self.stats, self.norm = layer, x, out, in_channel_offset, out_channel_offset, stats, norm
        b, h, w, ct = x.shape
        oh, ow = layer.out_hw(h, w)
        assert out.shape[:3] == (b, oh, ow)
        self.desc = ConvDesc(b, h, w, layer.cin, layer.cout, layer.groups, layer.kh, layer.kw, layer.stride, layer.pad[0], layer.pad[1],
                             ct, in_channel_offset, out.shape[3], out_channel_offset, layer.act, 0, layer.range_strata, 0, 0, 0)
        self.macs = b * oh * ow * layer.groups * layer.cout * layer.cin * layer.kh * layer.kw

    def partial_floats(self, tile: int) -> int:
        return int(hip.load().pn_conv_stat_partial_floats(C.byref(self.desc), tile))



def _job_array(jobs: Sequence[ConvJob]):
    arr = (hip.ConvJob * len(jobs))()
    for k, jb in enumerate(jobs):
        c = arr[k]
        c.desc = jb.desc
        jb.layer._ensure("direct")
        c.in_, c.packed_w, c.out = jb.x.data_ptr(), jb.layer.packed.data_ptr(), jb.out.data_ptr()
        c.scale, c.shift = hip.ptr(jb.layer.scale), hip.ptr(jb.layer.shift)
        if jb.stats is not None:
            st = jb.stats
            c.stat_partials = st["partials"].data_ptr()
            c.stat_strata, c.stat_channel_groups = int(st.get("strata", 1)), int(st["channel_groups"])
            c.stat_gamma, c.stat_beta, c.stat_eps = hip.ptr(st.get("gamma")), hip.ptr(st.get("beta")), float(st["eps"])
            c.stat_affine_strata = int(st["affine_strata"])
            c.stat_affine, c.stat_mean_rstd = hip.ptr(st.get("affine")), hip.ptr(st.get("mean_rstd"))
        if jb.norm is not None:
            tab, strata, channels = jb.norm
            c.norm_affine, c.norm_strata, c.norm_channels = tab.data_ptr(), int(strata), int(channels)
    return arr


def conv_multi(jobs: Sequence[ConvJob], tile: int) -> None:
    """run the jobs as ONE launch of the MFMA kernel (tile: 1 = 128x128, 3 = 64x64, 4 = 64x32, 5 = 64x128).
    Statistics jobs need ``stats['partials']`` (float scratch of ``job.partial_floats(tile)``); ``conv_stats_finalize`` /
    ``conv_stats_apply`` turn the partials into the norm's affine table / apply the norm."""
    arr = _job_array(jobs)
    st = hip.stream()
    prof = _PROFILER
    if prof is not None:
        ev = prof.begin(st)
    hip.call("pn_conv2d_multi_f32", arr, len(jobs), int(tile), st)
    if prof is not None:
        j0 = jobs[0]
        prof.end(ev, 2.0 * sum(j.macs for j in jobs), st,
                 tag=f"multi x{len(jobs)} {j0.out.shape[1]}x{j0.out.shape[2]} {sum(j.layer.cin * j.layer.groups for j in jobs)}->{sum(j.layer.out_channels for j in jobs)} k{j0.layer.kh}")


def conv_small_n_multi(jobs: Sequence[ConvJob]) -> None:
    """the jobs (1x1 / 3x3, <= 64 input channels, <= 12 output columns each) as ONE launch of the VALU kernel"""
    st = hip.stream()
    prof = _PROFILER
    if prof is not None:
        ev = prof.begin(st)
    hip.call("pn_conv2d_small_n_multi_f32", _job_array(jobs), len(jobs), st)
    if prof is not None:
        j0 = jobs[0]
        prof.end(ev, 2.0 * sum(j.macs for j in jobs), st,
                 tag=f"small-n x{len(jobs)} {j0.out.shape[1]}x{j0.out.shape[2]} {sum(j.layer.cin * j.layer.groups for j in jobs)}->{sum(j.layer.out_channels for j in jobs)}")


def conv_stats_finalize(jobs: Sequence[ConvJob], tile: int) -> None:
    """fold the statistics partials of the jobs (same list / tile as the ``conv_multi`` call) into their affine tables"""
    hip.call("pn_conv_stats_finalize_f32", _job_array(jobs), len(jobs), int(tile), hip.stream())


def conv_stats_apply(producer: ConvJob, tile: int, gamma, beta, act, out: torch.Tensor, out_channel_offset=0, mul=None, add=None,
                     out2: Optional[torch.Tensor] = None, out2_channel_offset=0) -> None:
    """RSNorm + activation (+ calibrated copy) of ``producer.out`` from the partials its epilogue wrote; no finalize launch"""
    hip.require_device(out)
    hip.call("pn_conv_stats_apply_f32", _job_array([producer]), int(tile), hip.ptr(gamma), hip.ptr(beta), int(act), out.data_ptr(), out.shape[3],
             out_channel_offset, hip.ptr(mul), hip.ptr(add), hip.ptr(out2), 0 if out2 is None else out2.shape[3], out2_channel_offset, hip.stream())


def groupnorm_apply(x: torch.Tensor, channel_groups: int, range_strata: int, mean_rstd: torch.Tensor, gamma, beta, act, out: torch.Tensor,
                    out_channel_offset=0, mul=None, add=None, out2: Optional[torch.Tensor] = None, out2_channel_offset=0,
                    channels: Optional[int] = None, channel_offset=0) -> None:
    """normalisation pass with the statistics already on the device (from a convolution's epilogue)"""
    hip.require_device(x, out, mean_rstd)
    b, h, w, ct = x.shape
    c = ct - channel_offset if channels is None else channels
    hip.call("pn_groupnorm_apply_f32", x.data_ptr(), b, h, w, c, ct, channel_offset, channel_groups, range_strata, mean_rstd.data_ptr(),
             hip.ptr(gamma), hip.ptr(beta), int(act), out.data_ptr(), out.shape[3], out_channel_offset, hip.ptr(mul), hip.ptr(add), hip.ptr(out2),
             0 if out2 is None else out2.shape[3], out2_channel_offset, hip.stream())


def conv2d_direct(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride=1, pad=0, groups=1,
                  act=ACT_NONE) -> torch.Tensor:
    """plain direct convolution (any channel counts); NHWC in / NHWC out"""
    hip.require_device(x, weight)
    w = weight.detach().contiguous().float()
    cout_t, cin_g, kh, kw = w.shape
    b, h, wd, ct = x.shape
    oh, ow = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    out = torch.empty((b, oh, ow, cout_t), dtype=torch.float32, device=x.device)
    d = ConvDesc(b, h, wd, cin_g, cout_t // groups, groups, kh, kw, stride, pad, pad, ct, 0, cout_t, 0, act, 0, 0)
    sh = None if bias is None else bias.detach().contiguous().float()
    hip.call("pn_conv2d_direct_nhwc_f32", C.byref(d), x.data_ptr(), w.data_ptr(), None, hip.ptr(sh), out.data_ptr(),
             hip.stream())
    return out


def fold_bn(gamma, beta, mean, var, eps: float, conv_bias=None):
    hip.require_device(gamma)
    c = gamma.numel()
    scale, shift = _f32(c, gamma.device), _f32(c, gamma.device)
    args = [t.detach().contiguous().float() for t in (gamma, beta, mean, var)]
    cb = None if conv_bias is None else conv_bias.detach().contiguous().float()
    hip.call("pn_fold_bn_f32", *(t.data_ptr() for t in args), hip.ptr(cb), float(eps), c, scale.data_ptr(),
             shift.data_ptr(), hip.stream())
    return scale, shift


# ------------------------------------------------------------------------------ norms
def groupnorm_strat(x: torch.Tensor, channel_groups: int, range_strata: int, gamma: torch.Tensor, beta: torch.Tensor,
                    eps=1e-5, act=ACT_NONE, out: Optional[torch.Tensor] = None, mul: Optional[torch.Tensor] = None,
                    add: Optional[torch.Tensor] = None, stat_out: Optional[torch.Tensor] = None):
    """x NHWC (B,H,W,C).  gamma/beta have range_strata*C entries in stacked order [stratum][channel].
    Returns out, or (out, out*mul+add) when mul/add ((H,W,C) maps) are given.  ``stat_out`` (B * strata * groups * 2 floats): keeps the
    (mean, rstd) pairs for ``groupnorm_strat_bwd(..., stat=)``."""
    hip.require_device(x)
    lib = hip.load()
    assert x.is_contiguous()
    b, h, w, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    out2 = torch.empty_like(x) if mul is not None else None
    ws_bytes = lib.pn_groupnorm_workspace_bytes(b, channel_groups, range_strata)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    assert stat_out is None or (stat_out.numel() >= 2 * b * range_strata * channel_groups and stat_out.dtype == torch.float32)
    hip.call("pn_groupnorm_strat_fwd_stat", x.data_ptr(), b, h, w, c, c, 0, channel_groups, range_strata, hip.ptr(gamma),
             hip.ptr(beta), float(eps), int(act), out.data_ptr(), c, 0, hip.ptr(mul), hip.ptr(add), hip.ptr(out2), hip.ptr(stat_out),
             ws.data_ptr(), ws_bytes, hip.stream())
    return out if out2 is None else (out, out2)


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


# ------------------------------------------------------------------------------ GEMM / attention glue (A1)
ACT_GELU = hip.ACT_GELU


_LINEAR_ON = os.environ.get("PN_LINEAR", "1") != "0"     # 0: the r2 route (1x1 convolution on conv_mfma_kernel)
_LN_FOLD_ON = os.environ.get("PN_LN_FOLD", "1") != "0"   # 0: every LayerNorm in front of a token GEMM as its own pass (r3 - r5)


class GemmLayer:
    """packed nn.Linear: y = act(x @ W^T + b) (+ residual) on the token-GEMM kernel (csrc/linear.hip: pn_linear_f32);
    ``PN_LINEAR=0`` keeps the r2 route through the MFMA convolution kernel (a 1x1 convolution)"""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, ksplit: bool = False):
        """``ksplit``: the layer runs on a few thousand rows at most (the key-point chains of the SetBlock): pn_linear_ksplit_f32, the
        K-split form with its own fp32 summation order, for every call of this layer"""
        hip.require_device(weight)
        lib = hip.load()
        w = weight.detach().contiguous().float()
        self.n, self.k = w.shape
        self.linear = _LINEAR_ON and self.n % 4 == 0 and self.k % 4 == 0
        self.entry = "pn_linear_ksplit_f32" if (ksplit and self.linear) else "pn_linear_f32" if self.linear else "pn_gemm_bias_act_f32"
        if self.linear:
            self.packed = _f32(lib.pn_linear_packed_weight_floats(self.n, self.k), w.device)
            hip.call("pn_pack_linear_weight_f32", w.data_ptr(), self.n, self.k, self.packed.data_ptr(), hip.stream())
        else:
            self.packed = _f32(lib.pn_conv_packed_weight_floats(self.n, self.k, 1, 1, 1), w.device)
            hip.call("pn_pack_conv_weight_f32", w.data_ptr(), self.n, self.k, 1, 1, 1, self.packed.data_ptr(), hip.stream())
        self.bias = None if bias is None else bias.detach().contiguous().float()
        self._w_f32 = w                  # source of the bf16 pack (made on the first bf16 call)
        self.ln = None

    @property
    def bf16_ok(self) -> bool:
        """can this layer run on the bf16 matrix pipe (pn_linear_bf16: k a multiple of 64, n of 16)?  Callers keep a GEMM in f32 where not."""
        return self.k % 64 == 0 and self.n % 16 == 0

    def prepack_bf16(self) -> None:
        """pack the bf16 weights now (set_compute_dtype('bf16')), not on the first call -- which may be inside a hipGraph capture"""
        if self.bf16_ok and getattr(self, "packed_bf16", None) is None:
            self.packed_bf16 = torch.empty(hip.load().pn_conv_bf16_rows_packed_elems(self.n, self.k, 1, 1), dtype=torch.bfloat16, device=self._w_f32.device)
            hip.call("pn_pack_conv_weight_bf16_rows", self._w_f32.data_ptr(), self.n, self.k, 1, 1, self.packed_bf16.data_ptr(), hip.stream())

    @property
    def stats_ok(self) -> bool:
        """can this layer leave the row statistics a LayerNorm-folding consumer needs (``__call__(..., stats_out=True)``)?"""
        return self.linear and self.entry == "pn_linear_f32" and self.n % 32 == 0 and _LN_FOLD_ON

    def fold_layernorm(self, norm) -> bool:
        """Fold ``norm`` (an nn.LayerNorm over this layer's k inputs) into the layer: LayerNorm(x) W^T + b = rstd (x (W gamma)^T - mean colsum)
        + (b + W beta).  After this ``__call__(x, ln_stats=table)`` takes the UN-normalised rows and the statistics table their producer left
        (pn_linear_ln_f32); plain calls keep the plain weights.  False (nothing changed) where the fold does not apply."""
        if not (self.linear and self.entry == "pn_linear_f32" and self.k % 64 == 0 and _LN_FOLD_ON):
            return False
        lib = hip.load()
        w64 = self._w_f32.double()
        g, b = norm.weight.detach().double().to(w64.device), norm.bias.detach().double().to(w64.device)
        wg = (w64 * g[None, :])
        wg32 = wg.float().contiguous()
        packed = _f32(lib.pn_linear_packed_weight_floats(self.n, self.k), wg32.device)
        hip.call("pn_pack_linear_weight_f32", wg32.data_ptr(), self.n, self.k, packed.data_ptr(), hip.stream())
        b0 = self.bias.double() if self.bias is not None else torch.zeros(self.n, dtype=torch.float64, device=w64.device)
        # colsum over the ROUNDED folded weights: what the MFMA multiplies, so mean * colsum cancels the accumulated mean term exactly in exact arithmetic
        self.ln = dict(packed=packed, colsum=wg32.double().sum(1).float().contiguous(), bias=(b0 + w64 @ b).float().contiguous(), eps=float(norm.eps))
        return True

    def __call__(self, x: torch.Tensor, act=ACT_NONE, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 out_bf16: bool = False, ln_stats: Optional[torch.Tensor] = None, stats_out: bool = False):
        """x: (m, k) f32 -- or bf16: the layer then runs on the bf16 matrix pipe (pn_linear_bf16, csrc/conv_bf16.hip; weights packed as bf16
        on first use, f32 accumulation, bias / activation / residual in f32) and returns f32, or bf16 with ``out_bf16`` (the input of
        another bf16 layer)"""
        assert x.dim() == 2 and x.is_contiguous() and x.shape[1] == self.k
        m = x.shape[0]
        st = hip.stream()
        prof = _PROFILER
        if x.dtype == torch.bfloat16:
            assert self.bf16_ok, "bf16 GEMM: k a multiple of 64, n of 16 (check GemmLayer.bf16_ok and keep the layer in f32 otherwise)"
            self.prepack_bf16()
            if out is None:
                out = torch.empty((m, self.n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
            if prof is not None:
                ev = prof.begin(st)
            hip.call("pn_linear_bf16", x.data_ptr(), m, self.k, self.k, self.packed_bf16.data_ptr(), self.n, hip.ptr(self.bias), int(act),
                     hip.ptr(residual), self.n, out.data_ptr(), self.n, int(out.dtype == torch.float32), st)
            if prof is not None:
                prof.end(ev, 2.0 * m * self.n * self.k, st, tag=f"gemm {m}x{self.k}->{self.n} bf16")
            return out
        assert not out_bf16, "a bf16 output needs a bf16 input"
        if out is None:
            out = torch.empty((m, self.n), dtype=torch.float32, device=x.device)
        if prof is not None:
            ev = prof.begin(st)
        stats = None
        if ln_stats is not None or stats_out:
            # LayerNorm folded around the GEMM (pn_linear_ln_f32): consumer of a statistics table (``ln_stats``: x is the UN-normalised rows;
            # needs fold_layernorm) or producer of one (``stats_out``: -> (out, table [m][n / 32][2]))
            assert not (ln_stats is not None and stats_out)
            if ln_stats is not None:
                assert self.ln is not None and tuple(ln_stats.shape) == (m, self.k // 32, 2) and ln_stats.is_contiguous()
                hip.call("pn_linear_ln_f32", x.data_ptr(), m, self.k, self.k, self.ln["packed"].data_ptr(), self.n, self.ln["bias"].data_ptr(), int(act),
                         hip.ptr(residual), self.n, out.data_ptr(), self.n, ln_stats.data_ptr(), self.ln["colsum"].data_ptr(), self.ln["eps"], None, st)
            else:
                assert self.stats_ok
                stats = torch.empty((m, self.n // 32, 2), dtype=torch.float32, device=x.device)
                hip.call("pn_linear_ln_f32", x.data_ptr(), m, self.k, self.k, self.packed.data_ptr(), self.n, hip.ptr(self.bias), int(act),
                         hip.ptr(residual), self.n, out.data_ptr(), self.n, None, None, 0.0, stats.data_ptr(), st)
        else:
            hip.call(self.entry, x.data_ptr(), m, self.k, self.k, self.packed.data_ptr(), self.n,
                     hip.ptr(self.bias), int(act), hip.ptr(residual), self.n, out.data_ptr(), self.n, st)
        if prof is not None:
            prof.end(ev, 2.0 * m * self.n * self.k, st, tag=f"gemm {m}x{self.k}->{self.n}")
        return (out, stats) if stats_out else out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, want_chan_mean=False, bf16_copy=False, f32_out=True):
    """-> out [, chan_mean] [, bf16 copy]; ``bf16_copy``: also the result rounded to bf16 (input of the bf16 GEMMs); with ``f32_out`` False
    only the copy is written (and returned in place of ``out``)"""
    hip.require_device(x)
    assert x.dim() == 2 and x.is_contiguous()
    rows, c = x.shape
    out = torch.empty_like(x) if f32_out else None
    cm = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_chan_mean else None
    if bf16_copy:
        o16 = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
        hip.call("pn_layernorm_bf16out_f32", x.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(), float(eps), hip.ptr(out), o16.data_ptr(),
                 hip.ptr(cm), hip.stream())
        res = ((out,) if f32_out else ()) + ((cm,) if want_chan_mean else ()) + (o16,)
        return res if len(res) > 1 else res[0]
    assert f32_out
    hip.call("pn_layernorm_f32", x.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(), float(eps), out.data_ptr(),
             hip.ptr(cm), hip.stream())
    return (out, cm) if want_chan_mean else out


# ------------------------------------------------------------------------------ conv backward (T1)
_WGRAD_STREAM = os.environ.get("PN_TRAIN_WGRAD_STREAM", "1") != "0"


_CONCURRENT: dict = {}


def concurrent_stream(device=None) -> "torch.cuda.Stream":
    """A stream whose work really overlaps with the CURRENT stream's.  The HIP runtime multiplexes streams onto a handful of hardware queues
    (four by default) in creation order; two streams that land on the same queue run one after the other.  Which stream collides with
    which depends on how many streams the process made before -- measured with the training step: default + second stream 13.9 ms per
    iteration, 15.3 ms (the one-stream time) when exactly six other streams had been created earlier, and with GPU_MAX_HW_QUEUES=8 the
    collision just moves (tools/hwq.py).  So the choice is measured: candidates are probed with two spin kernels of ~0.3 ms, one on the
    current stream and one on the candidate, and the first candidate that finishes the pair in about the time of one is kept (cached per
    current stream).  Inside a hipGraph capture nothing is probed: a graph's branches are scheduled by the graph, not by these streams."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    main = torch.cuda.current_stream(dev)
    key = (str(dev), main.cuda_stream)
    got = _CONCURRENT.get(key)
    if got is not None:
        return got
    if torch.cuda.is_current_stream_capturing():
        any_key = (str(dev), "capture")
        if any_key not in _CONCURRENT:
            _CONCURRENT[any_key] = next((v for k, v in _CONCURRENT.items() if k[0] == str(dev)), None) or torch.cuda.Stream(device=dev)
        return _CONCURRENT[any_key]
    cycles = 600000

    def pair_ms(cand):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if cand is not None:
            cand.wait_stream(main)
            with torch.cuda.stream(cand):
                torch.cuda._sleep(cycles)
        torch.cuda._sleep(cycles)
        if cand is not None:
            main.wait_stream(cand)
        e1.record(main)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    pair_ms(None)
    one = min(pair_ms(None) for _ in range(2))
    best, best_t = None, float("inf")
    for _ in range(12):
        cand = torch.cuda.Stream(device=dev)
        t = min(pair_ms(cand) for _ in range(2))
        if t < best_t:
            best, best_t = cand, t
        if t < 1.4 * one:
            break
    _CONCURRENT[key] = best
    return best


def probe_streams(device=None) -> None:
    """Explicit form of the probing ``concurrent_stream`` does on first use: ~60 device-wide synchronisations and up to twelve
    stream creations.  Serving / training loops call it once up front, on the stream they will run on and BEFORE any hipGraph capture
    starts on another thread (a device-wide synchronisation invalidates a capture in progress); afterwards ``concurrent_stream`` is a
    dictionary lookup.  The frames-in-flight hint (``frames_in_flight``) is a process global: engines are captured from one thread."""
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        concurrent_stream(device)


def concurrent_streams(k: int, device=None, candidates: int = 16):
    """k streams that overlap with EACH OTHER (several hipGraph engines replaying at once: engines whose streams share a hardware queue run
    their frames one after the other).  Greedy: a candidate joins the set when a spin kernel on it and one on every member finish in about
    the time of one; if the runtime has fewer independent queues than k, the best candidates found fill the set."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    main = torch.cuda.current_stream(dev)
    cycles = 600000

    def pair_ms(a, b):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for st in (a, b):
            if st is not None:
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
        for st in (a, b):
            if st is not None:
                main.wait_stream(st)
        e1.record(main)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    first = torch.cuda.Stream(device=dev)
    pair_ms(first, None)
    one = min(pair_ms(first, None) for _ in range(2))
    chosen, spare = [first], []
    for _ in range(candidates):
        if len(chosen) >= k:
            break
        cand = torch.cuda.Stream(device=dev)
        worst = max(min(pair_ms(cand, m) for _ in range(2)) for m in chosen)
        if worst < 1.4 * one:
            chosen.append(cand)
        else:
            spare.append((worst, cand))
    spare.sort(key=lambda wc: wc[0])
    while len(chosen) < k and spare:
        chosen.append(spare.pop(0)[1])
    while len(chosen) < k:
        chosen.append(torch.cuda.Stream(device=dev))
    return chosen[:k]


class SideStream:
    """Weight gradients off the critical path of backward: dW of a layer is needed by nobody before the gradient exchange / the optimizer,
    while the chain  d(out) -> BatchNorm backward -> data gradient -> previous layer  is serial.  ``run`` queues a layer's weight-gradient
    launches (kernel + slice reduction + bias sums) on a second HIP stream behind the work queued so far; the data gradient goes on on the
    main stream and the two overlap -- the 64 x 64 / 128 x 128 layers do not fill the chip on their own.  ``join`` makes the main stream
    wait (before a gradient bucket is handed to the exchange, and at the end of backward).  Same kernels, same results: nothing here
    depends on the order two independent kernels finish in.  The second stream is picked on first use so that it really overlaps with the
    caller's stream (``concurrent_stream``).  ``PN_TRAIN_WGRAD_STREAM=0`` keeps everything on one stream."""

    def __init__(self, device):
        self.on = _WGRAD_STREAM and torch.device(device).type == "cuda" and torch.cuda.is_available()
        self.device = device
        self._dirty: list = []        # every side stream that has run something since the last join
        self.keep: list = []

    @property
    def stream(self):
        """the second stream for the CURRENT stream (None when switched off)"""
        if not self.on:
            return None
        return concurrent_stream(self.device)

    @stream.setter
    def stream(self, value):
        if value is None:
            self.on = False

    def run(self, fn, *reads, after=None):
        """``after``: an event of the main stream the launches wait for instead of everything queued on it so far"""
        side = self.stream
        if side is None:
            fn()
            return
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        # the buffers the side stream reads stay referenced until the join: freed earlier, the caching allocator would hand them to the
        # main stream again while the side stream still reads them.  (Tensor.record_stream does the same bookkeeping inside the allocator,
        # but with hundreds of large cross-stream blocks per iteration it kept the allocator from reusing memory: the PARTNER detector's
        # training iteration went from 103 to 184 ms.)
        self.keep.extend(t for t in reads if t is not None)
        if not any(side is x for x in self._dirty):
            self._dirty.append(side)

    @property
    def dirty(self) -> bool:
        return bool(self._dirty)

    def join(self):
        """the current stream waits for EVERY side stream used since the last join (the instance is shared per device and the side stream
        depends on the caller's current stream: run() from two different streams between joins leaves two of them dirty), then the
        read buffers are released"""
        if self._dirty:
            cur = torch.cuda.current_stream()
            for side in self._dirty:
                cur.wait_stream(side)
            self._dirty.clear()
            self.keep.clear()


_SHARED_SIDE: dict = {}


def shared_side_stream(device) -> SideStream:
    """ONE side stream per device for the tapes (a tape lives for one iteration; a HIP stream made per iteration would also get a fresh
    pool in the caching allocator, i.e. a hipMalloc for every buffer it ever allocates)"""
    key = str(torch.device(device))
    if key not in _SHARED_SIDE:
        _SHARED_SIDE[key] = SideStream(device)
    return _SHARED_SIDE[key]


def conv_wgrad(x: torch.Tensor, dout: torch.Tensor, kh: int, kw: int, stride=1, pad=0, cin: Optional[int] = None,
               in_channel_offset=0, cout: Optional[int] = None, dout_channel_offset=0, out: Optional[torch.Tensor] = None,
               accumulate=False, range_strata=0) -> torch.Tensor:
    """dW (Cout, Cin, KH, KW) of a convolution x -> y given dout = dL/dy; NHWC maps.  ``range_strata`` > 1: the RangeStratified
    convolution (one weight set per band of W / strata columns) -> dW (strata * Cout, Cin, KH, KW)."""
    hip.require_device(x, dout)
    lib = hip.load()
    assert x.dim() == 4 and dout.dim() == 4 and x.is_contiguous() and dout.is_contiguous()
    b, h, w, ct = x.shape
    cin = ct - in_channel_offset if cin is None else cin
    cout = dout.shape[3] - dout_channel_offset if cout is None else cout
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    d = ConvDesc(b, h, w, cin, cout, 1, kh, kw, stride, ph, pw, ct, in_channel_offset, dout.shape[3], dout_channel_offset, 0, 0, int(range_strata))
    oh, ow = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    assert dout.shape[:3] == (b, oh, ow), (dout.shape, (b, oh, ow))
    if out is None:
        out = torch.empty((max(1, int(range_strata)) * cout, cin, kh, kw), dtype=torch.float32, device=x.device)
    if (range_strata <= 1 and _WGRAD_WINO4 and (kh, kw, stride, ph, pw) == (3, 3, 1, 1, 1) and w % 4 == 0 and cin % 4 == 0 and cout % 4 == 0 and in_channel_offset % 4 == 0
            and dout_channel_offset % 4 == 0 and ct % 4 == 0 and dout.shape[3] % 4 == 0 and b * h * (w // 4) >= _WGRAD_WINO4_MIN_QUADS
            and cin * cout >= 0.75 * (-(-cin // 128) * 128) * (-(-cout // 128) * 128)):      # its 128 x 128 (ci, co) tiles mostly full
        # F(4, 3) weight gradient (conv_wgrad_wino4.hip): half the MFMA work on the maps large enough to fill the chip with its slices
        nbytes = lib.pn_conv2d_wgrad_wino4_workspace_bytes(C.byref(d))
        ws = _workspace(nbytes, x.device)
        hip.call("pn_conv2d_wgrad_wino4_f32", C.byref(d), x.data_ptr(), dout.data_ptr(), out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
        return out
    nbytes = lib.pn_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = _workspace(nbytes, x.device)
    hip.call("pn_conv2d_wgrad_f32", C.byref(d), x.data_ptr(), dout.data_ptr(), out.data_ptr(), int(accumulate), ws.data_ptr(),
             nbytes, hip.stream())
    return out


_WS = {}
_WGRAD_WINO4 = os.environ.get("PN_CONV_WGRAD_WINO4", "1") != "0"
_WGRAD_WINO4_MIN_QUADS = int(os.environ.get("PN_CONV_WGRAD_WINO4_MIN_QUADS", "4096"))


def _workspace(nbytes: int, dev) -> torch.Tensor:
    """grow-only scratch buffer per (device, stream): the backward kernels need their workspace only
    until the launch that consumes it has been queued on the same stream"""
    if torch.cuda.is_current_stream_capturing():
        # every hipGraph capture runs on torch's shared capture stream: a cached buffer keyed by the stream would be shared by all
        # captured engines (and owned by the first graph's pool) -- a race once the engines replay concurrently.  Inside a capture
        # the scratch is a plain allocation of that graph's private pool.
        return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    key = (str(dev), hip.stream())
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = t
    return t


def channel_sum(x: torch.Tensor, c: Optional[int] = None, channel_offset=0, out: Optional[torch.Tensor] = None,
                accumulate=False) -> torch.Tensor:
    """sum over all pixels of an NHWC map, per channel (bias gradient)"""
    hip.require_device(x)
    lib = hip.load()
    ct = x.shape[-1]
    c = ct - channel_offset if c is None else c
    pixels = x.numel() // ct
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib.pn_channel_sum_workspace_bytes(c)
    ws = _workspace(nbytes, x.device)
    hip.call("pn_channel_sum_f32", x.data_ptr(), pixels, ct, channel_offset, c, out.data_ptr(), int(accumulate), ws.data_ptr(),
             nbytes, hip.stream())
    return out


class ConvDgrad:
    """Data gradient of ``Conv2d(weight, stride, pad)`` as a convolution of dout on the MFMA kernel.

    Supported geometries (all the reference's BEV path uses): stride 1 (any k, pad);
    3x3 / stride 2 / pad 1; 2x2 / stride 2 / pad 0.  ``repack(weight)`` refreshes the packed copy
    after an optimizer step without reallocating."""

    def __init__(self, weight: torch.Tensor, stride=1, pad=0):
        hip.require_device(weight)
        lib = hip.load()
        cout, cin, kh, kw = weight.shape
        self.cout, self.cin, self.kh, self.kw, self.stride, self.pad = cout, cin, kh, kw, int(stride), int(pad)
        dev = weight.device
        if self.stride == 1:
            self.kind = "s1"
            self.packed = _f32(lib.pn_conv_packed_weight_floats(cin, cout, kh, kw, 1), dev)
        elif self.stride == 2 and (kh, kw, self.pad) == (3, 3, 1):
            self.kind = "s2k3"
            self.packed = _f32(lib.pn_conv_dgrad_s2_packed_weight_floats(cout, cin), dev)
        elif self.stride == 2 and (kh, kw, self.pad) == (2, 2, 0):
            self.kind = "s2k2"
            self.packed = _f32(lib.pn_deconv2x2_packed_weight_floats(cout, cin), dev)
        else:
            raise hip.PartnerHipError(f"ConvDgrad: unsupported geometry k={kh}x{kw} stride={stride} pad={pad}")
        # the data gradient of a 3x3 / stride-1 / pad-1 convolution is itself one (taps mirrored, channels swapped): it takes the
        # width-Winograd kernel on large maps, like the forward layer (conv_wino.hip)
        self.wino_packed = self.wino4_packed = None
        if self.kind == "s1" and (kh, kw, self.pad) == (3, 3, 1) and cout % 4 == 0 and _WINO_ON:
            self.wino_packed = _f32(lib.pn_conv_wino_packed_weight_floats(cin, cout), dev)
            if _WINO4_ON and _WINO4_DGRAD and cin % 32 == 0:
                self.wino4_packed = _f32(lib.pn_conv_wino4_packed_weight_floats(cin, cout), dev)
        self.repack(weight)

    def repack(self, weight: torch.Tensor, token=None) -> None:
        """lazy, as ConvLayer.repack: the layout a call takes is packed on first use"""
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        self._stale_w = weight.detach().contiguous().float()
        self._stale = {"direct"} | ({"wino"} if self.wino_packed is not None else set()) | ({"wino4"} if self.wino4_packed is not None else set())

    def prepack_used(self) -> None:
        for layout in sorted(getattr(self, "_used", ())):
            self._ensure(layout)

    def _ensure(self, layout: str) -> None:
        self.__dict__.setdefault("_used", set()).add(layout)
        if layout not in self._stale:
            return
        self._stale.discard(layout)
        w, st = self._stale_w, hip.stream()
        if layout == "direct":
            if self.kind == "s1":
                hip.call("pn_pack_conv_dgrad_weight_f32", w.data_ptr(), self.cout, self.cin, self.kh, self.kw, self.packed.data_ptr(), st)
            elif self.kind == "s2k3":
                hip.call("pn_pack_conv_dgrad_s2_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), st)
            else:
                hip.call("pn_pack_deconv2x2_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), st)
            return
        # the gradient convolution's weight is the forward one with the taps mirrored and the channels swapped: the pack kernels read it so
        # (r2 made a flipped + transposed copy first: two more launches per layer and iteration)
        if layout == "wino":
            hip.call("pn_pack_conv_dgrad_weight_wino_f32", w.data_ptr(), self.cout, self.cin, self.wino_packed.data_ptr(), st)
        else:
            hip.call("pn_pack_conv_dgrad_weight_wino4_f32", w.data_ptr(), self.cout, self.cin, self.wino4_packed.data_ptr(), st)

    def __call__(self, dout: torch.Tensor, out: Optional[torch.Tensor] = None, dout_channel_offset=0,
                 out_channel_offset=0, accumulate=False) -> torch.Tensor:
        """dout: NHWC (B,OH,OW,Ct) -> dx (B,H,W,Cin); H = OH*stride (the reference's maps are even-sized)"""
        hip.require_device(dout)
        assert dout.dim() == 4 and dout.is_contiguous()
        b, oh, ow, ct = dout.shape
        # the MFMA loader fetches 4 channels at a time: a Cout that is not a multiple of 4 needs dout
        # stored with zero-filled pad channels (packed weight rows past Cout are zero as well)
        cin_eff = (self.cout + 3) // 4 * 4
        if dout_channel_offset + cin_eff > ct:
            raise hip.PartnerHipError(f"ConvDgrad: dout needs {cin_eff - self.cout} zero pad channel(s) after its {self.cout} channels")
        if self.kind == "s1":
            h, w = oh + self.kh - 1 - 2 * self.pad, ow + self.kw - 1 - 2 * self.pad
            pd = self.kh - 1 - self.pad
            d = ConvDesc(b, oh, ow, cin_eff, self.cin, 1, self.kh, self.kw, 1, pd, self.kw - 1 - self.pad, ct, dout_channel_offset,
                         0, out_channel_offset, 0, 0, 0)
        elif self.kind == "s2k3":
            h, w = 2 * oh, 2 * ow
            d = ConvDesc(b, oh, ow, cin_eff, self.cin, 1, 2, 2, 1, 0, 0, ct, dout_channel_offset, 0, out_channel_offset, 0, 1, 0, 1, 1)
        else:
            h, w = 2 * oh, 2 * ow
            d = ConvDesc(b, oh, ow, cin_eff, self.cin, 1, 1, 1, 1, 0, 0, ct, dout_channel_offset, 0, out_channel_offset, 0, 1, 0)
        if out is None:
            out = torch.empty((b, h, w, self.cin), dtype=torch.float32, device=dout.device)
        assert out.shape[:3] == (b, h, w) and out.is_contiguous()
        d.out_pixel_stride = out.shape[3]
        d.accumulate = int(accumulate)
        if (self.wino4_packed is not None and not accumulate and ow % 4 == 0 and cin_eff == self.cout
                and ((b * oh * (ow // 4) + 31) // 32) * (self.cin // 32) >= _WINO4_MIN_TILES):
            self._ensure("wino4")
            hip.call("pn_conv2d_wino4_nhwc_f32", C.byref(d), dout.data_ptr(), self.wino4_packed.data_ptr(), None, None, out.data_ptr(), hip.stream())
            return out
        if (self.wino_packed is not None and not accumulate and ow % 2 == 0 and cin_eff == self.cout
                and ((b * oh * (ow // 2) + 31) // 32) * ((self.cin + 63) // 64) >= _WINO_MIN_TILES):
            self._ensure("wino")
            hip.call("pn_conv2d_wino_nhwc_f32", C.byref(d), dout.data_ptr(), self.wino_packed.data_ptr(), None, None, out.data_ptr(), hip.stream())
            return out
        self._ensure("direct")
        hip.call("pn_conv2d_nhwc_f32", C.byref(d), dout.data_ptr(), self.packed.data_ptr(), None, None, out.data_ptr(), hip.stream())
        return out


def batchnorm_train(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, momentum: float,
                    running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], act=ACT_RELU,
                    out: Optional[torch.Tensor] = None, c: Optional[int] = None, channel_offset=0, out_channel_offset=0):
    """training-mode BatchNorm2d + activation on an NHWC map -> (out, saved_stat)"""
    hip.require_device(x)
    lib = hip.load()
    ct = x.shape[-1]
    c = ct - channel_offset if c is None else c
    pixels = x.numel() // ct
    if out is None:
        out = torch.empty(x.shape[:-1] + (c,), dtype=torch.float32, device=x.device)
    stat = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    nbytes = lib.pn_batchnorm_workspace_bytes(c)
    ws = _workspace(nbytes, x.device)
    hip.call("pn_batchnorm_train_fwd", x.data_ptr(), pixels, c, ct, channel_offset, hip.ptr(gamma), hip.ptr(beta), float(eps),
             float(momentum), int(act), hip.ptr(running_mean), hip.ptr(running_var), out.data_ptr(), out.shape[-1],
             out_channel_offset, stat.data_ptr(), ws.data_ptr(), nbytes, hip.stream())
    return out, stat


def batchnorm_bwd(x: torch.Tensor, dout: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, stat: torch.Tensor, act=ACT_RELU,
                  dx: Optional[torch.Tensor] = None, dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  accumulate=False, c: Optional[int] = None, channel_offset=0, dout_channel_offset=0, dx_channel_offset=0):
    """backward of batchnorm_train (+ its activation) -> (dx, dgamma, dbeta); dx may be dout (in place)"""
    hip.require_device(x, dout)
    lib = hip.load()
    ct = x.shape[-1]
    c = ct - channel_offset if c is None else c
    pixels = x.numel() // ct
    if dx is None:
        dx = torch.empty(x.shape[:-1] + (c,), dtype=torch.float32, device=x.device)
    if dgamma is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib.pn_batchnorm_workspace_bytes(c)
    ws = _workspace(nbytes, x.device)
    hip.call("pn_batchnorm_bwd", x.data_ptr(), dout.data_ptr(), pixels, c, ct, channel_offset, dout.shape[-1], dout_channel_offset,
             hip.ptr(gamma), hip.ptr(beta), int(act), stat.data_ptr(), dx.data_ptr(), dx.shape[-1], dx_channel_offset,
             dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
    return dx, dgamma, dbeta


def groupnorm_strat_bwd(x: torch.Tensor, dout: torch.Tensor, channel_groups: int, range_strata: int, gamma: torch.Tensor,
                        beta: torch.Tensor, eps=1e-5, act=ACT_NONE, dout2: Optional[torch.Tensor] = None,
                        mul: Optional[torch.Tensor] = None, dx: Optional[torch.Tensor] = None, dgamma=None, dbeta=None,
                        dmul=None, dadd=None, accumulate=False, stat: Optional[torch.Tensor] = None):
    """backward of groupnorm_strat -> (dx, dgamma, dbeta[, dmul, dadd]); dx may be dout.  ``stat``: the forward's ``stat_out`` (else the
    statistics are recomputed from x)"""
    hip.require_device(x, dout)
    lib = hip.load()
    assert x.is_contiguous() and dout.is_contiguous()
    b, h, w, c = x.shape
    dev = x.device
    if dx is None:
        dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = _f32(range_strata * c, dev)
    if dbeta is None:
        dbeta = _f32(range_strata * c, dev)
    if dout2 is not None:
        assert mul is not None and dout2.is_contiguous()
        if dmul is None:
            dmul = torch.empty((h, w, c), dtype=torch.float32, device=dev)
        if dadd is None:
            dadd = torch.empty((h, w, c), dtype=torch.float32, device=dev)
    nbytes = lib.pn_groupnorm_bwd_workspace_bytes(b, c, channel_groups, range_strata)
    ws = _workspace(nbytes, dev)
    hip.call("pn_groupnorm_strat_bwd_stat", x.data_ptr(), dout.data_ptr(), hip.ptr(dout2), hip.ptr(mul), b, h, w, c, c, 0, dout.shape[-1], 0,
             channel_groups, range_strata, hip.ptr(gamma), hip.ptr(beta), float(eps), int(act), dx.data_ptr(), dx.shape[-1], 0,
             dgamma.data_ptr(), dbeta.data_ptr(), hip.ptr(dmul), hip.ptr(dadd), int(accumulate), hip.ptr(stat), ws.data_ptr(), nbytes, hip.stream())
    if dout2 is not None:
        return dx, dgamma, dbeta, dmul, dadd
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------ L1 loss, NHWC level
class CenterLossTargets:
    """device copies of one task's targets (example['hm'|'ind'|'mask'|'cat'|'anno_box'][t])"""

    def __init__(self, hm, ind, mask, cat, anno_box, device):
        self.hm = hm.to(device).float().contiguous()
        self.ind = ind.to(device).long().contiguous()
        self.mask = mask.to(device).to(torch.uint8).contiguous()
        self.cat = cat.to(device).long().contiguous()
        self.anno = anno_box.to(device).float().contiguous()


_CODE_WEIGHTS: dict = {}


def _loss_common(hm: torch.Tensor, ncls: int, boxes, tg: CenterLossTargets, code_weights, with_vel: bool):
    b, h, w, _ = hm.shape
    ndim = sum(n for _, n in boxes)
    ad = tg.anno.shape[-1]
    sel = list(range(ndim)) if with_vel else [0, 1, 2, 3, 4, 5, ad - 2, ad - 1]
    # cached on the device: a fresh torch.tensor(..., device=) is a pageable host-to-device copy, which makes the host wait for the stream
    # -- twice per training iteration, in the middle of it (forward loss, backward loss): the GPU then idles while the host catches up
    key = (tuple(float(v) for v in list(code_weights)[:ndim]), str(hm.device))
    cw = _CODE_WEIGHTS.get(key)
    if cw is None:
        cw = _CODE_WEIGHTS[key] = torch.tensor(list(key[0]), dtype=torch.float32, device=hm.device)
    n = len(boxes)
    # pixel strides come from the tensors' strides, so channel-slice views of wider NHWC maps work
    args = (hm.data_ptr(), hm.stride(2), tg.hm.data_ptr(), b, ncls, h, w, (C.c_void_p * n)(*[t.data_ptr() for t, _ in boxes]),
            (C.c_int * n)(*[t.stride(2) for t, _ in boxes]), (C.c_int * n)(*[c for _, c in boxes]), n, tg.ind.data_ptr(),
            tg.mask.data_ptr(), tg.cat.data_ptr(), tg.anno.data_ptr(), ad, (C.c_int * ndim)(*sel), tg.ind.shape[1], ndim, cw.data_ptr())
    return args, ndim, cw


def center_loss(hm: torch.Tensor, ncls: int, boxes, tg: CenterLossTargets, code_weights, weight: float, with_vel=True):
    """hm: NHWC logits (B,H,W,>=ncls); boxes: [(NHWC tensor, channels)] in the reference order
    (reg, height, dim[, vel], rot).  -> out[4+ndim] = [det, hm, loc, num_pos, elem...] (device)"""
    hip.require_device(hm)
    lib = hip.load()
    args, ndim, cw = _loss_common(hm, ncls, boxes, tg, code_weights, with_vel)
    out = torch.empty((4 + ndim,), dtype=torch.float32, device=hm.device)
    wsb = lib.pn_center_loss_workspace_bytes()
    ws = _workspace(wsb, hm.device)
    hip.call("pn_center_loss_fwd", *args, float(weight), out.data_ptr(), ws.data_ptr(), wsb, hip.stream())
    return out


def center_loss_bwd(hm: torch.Tensor, ncls: int, boxes, tg: CenterLossTargets, code_weights, weight: float, fwd_out: torch.Tensor,
                    grad_scale=1.0, with_vel=True, d_hm: Optional[torch.Tensor] = None, d_boxes=None):
    """-> (d_hm, [d_box...]) NHWC, channel counts padded to multiples of 4 (pad channels zero)"""
    hip.require_device(hm)
    args, ndim, cw = _loss_common(hm, ncls, boxes, tg, code_weights, with_vel)
    b, h, w, _ = hm.shape
    pad4 = lambda c: (c + 3) // 4 * 4  # noqa: E731
    if d_hm is None:
        d_hm = torch.empty((b, h, w, pad4(ncls)), dtype=torch.float32, device=hm.device)
    if d_boxes is None:
        d_boxes = [torch.empty((b, h, w, pad4(c)), dtype=torch.float32, device=hm.device) for _, c in boxes]
    n = len(boxes)
    hip.call("pn_center_loss_bwd", *args, float(weight), fwd_out.data_ptr(), float(grad_scale), d_hm.data_ptr(), d_hm.shape[3],
             (C.c_void_p * n)(*[t.data_ptr() for t in d_boxes]), (C.c_int * n)(*[t.shape[3] for t in d_boxes]), hip.stream())
    return d_hm, d_boxes


def dynamic_pfn_bwd(points: torch.Tensor, vi: VoxelIndex, w0: torch.Tensor, w1: torch.Tensor, vx: float, vy: float,
                    x_offset: float, y_offset: float, d_features: Optional[torch.Tensor] = None,
                    d_canvas: Optional[torch.Tensor] = None, dw0: Optional[torch.Tensor] = None,
                    dw1: Optional[torch.Tensor] = None, accumulate=False, center_table: Optional[torch.Tensor] = None):
    """weight gradients of the (32,128) DynamicPFNet given d_features (V,128) or the canvas gradient (B,T,R,128)"""
    hip.require_device(points, w0, w1)
    lib = hip.load()
    dev = points.device
    if center_table is None:
        center_table = pfn_center_table(vi.spec.grid[1], vy, y_offset, dev)
    if dw0 is None:
        dw0 = torch.empty_like(w0)
    if dw1 is None:
        dw1 = torch.empty_like(w1)
    _, _, g = vi.spec.c_arrays()
    nbytes = lib.pn_dynamic_pfn_bwd_workspace_bytes()
    ws = _workspace(nbytes, dev)
    hip.call("pn_dynamic_pfn_bwd", points.data_ptr(), points.shape[1], vi.voxel_start.data_ptr(), vi.order.data_ptr(),
             vi.num_voxels.data_ptr(), vi.n_cap, vi.unq_keys_ptr, g, w0.data_ptr(), w0.shape[0], w1.data_ptr(), w1.shape[0],
             float(vx), float(vy), float(x_offset), float(y_offset), center_table.data_ptr(), hip.ptr(d_features), hip.ptr(d_canvas),
             dw0.data_ptr(), dw1.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
    return dw0, dw1


# ------------------------------------------------------------------------------ T1 small kernels
def grad_norm(flat_grads: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """L2 norm of a flat fp32 buffer -> device scalar (1,)"""
    hip.require_device(flat_grads)
    lib = hip.load()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=flat_grads.device)
    nbytes = lib.pn_grad_norm_workspace_bytes()
    ws = _workspace(nbytes, flat_grads.device)
    hip.call("pn_grad_norm_f32", flat_grads.data_ptr(), flat_grads.numel(), out.data_ptr(), ws.data_ptr(), nbytes, hip.stream())
    return out


def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int, lr: float,
              beta1: float, beta2=0.99, eps=1e-8, weight_decay=0.01, total_norm: Optional[torch.Tensor] = None, max_norm=35.0) -> None:
    """fused clip + decoupled weight decay + Adam over flat buffers (in place)"""
    hip.require_device(params, grads, exp_avg, exp_avg_sq)
    assert params.numel() == grads.numel() == exp_avg.numel() == exp_avg_sq.numel()
    hip.call("pn_adam_step_f32", params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), params.numel(), int(step),
             float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), hip.ptr(total_norm), float(max_norm), hip.stream())


def tanh_bwd(y: torch.Tensor, dy: torch.Tensor, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
    hip.require_device(y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.numel() == dy.numel()
    if dx is None:
        dx = torch.empty_like(dy)
    hip.call("pn_tanh_bwd_f32", y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), hip.stream())
    return dx


def relu_bwd(y: torch.Tensor, dy: torch.Tensor, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = dy * (y > 0), y = the ReLU output; dx may be dy"""
    hip.require_device(y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.numel() == dy.numel()
    if dx is None:
        dx = torch.empty_like(dy)
    hip.call("pn_relu_bwd_f32", y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), hip.stream())
    return dx


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    hip.require_device(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    if out is None:
        out = torch.empty_like(a)
    hip.call("pn_add_f32", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), hip.stream())
    return out


def strat_expand(dy: torch.Tensor, strata: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,H,W,C) -> (B,H,W,strata*C): block of the pixel's range stratum = dy, zeros elsewhere"""
    hip.require_device(dy)
    assert dy.is_contiguous()
    b, h, w, c = dy.shape
    if out is None:
        out = torch.empty((b, h, w, strata * c), dtype=torch.float32, device=dy.device)
    hip.call("pn_strat_expand_f32", dy.data_ptr(), b, h, w, c, strata, out.data_ptr(), hip.stream())
    return out


class StratConvDgrad:
    """Data gradient of the RangeStratified 3x3 convolution (weight (strata * Cout, Cin, 3, 3), center_head_parallel.py:27-59) at the
    convolution's own multiply-add count.  The weight set a tap takes follows the stratum of the dy pixel it reads, so per width tap kx
    the column convolution z_kx[y, x] = sum_ky W_s(x)[:, :, ky, kx]^T dy[y - ky + 1, x] is a STRATIFIED 3x1 convolution of dy (the forward
    kernel, 3 * Cin outputs per stratum), and dx[y, x] = z_0[y, x + 1] + z_1[y, x] + z_2[y, x - 1] (pn_strat_dgrad_combine_f32).  r3 expanded
    dy to strata * Cout channels and ran an ordinary gradient convolution over mostly zeros (8 x the work on the reference's head)."""

    def __init__(self, weight: torch.Tensor, strata: int):
        hip.require_device(weight)
        ct, cin, kh, kw = weight.shape
        assert (kh, kw) == (3, 3) and ct % strata == 0 and cin % 4 == 0
        self.strata, self.cin, self.cout = int(strata), cin, ct // strata
        # w'[s, kx * Cin + ci, co, ky', 0] = w[s * Cout + co, ci, 2 - ky', kx]: one gather of the flat weight
        idx = torch.arange(weight.numel(), device=weight.device).view(strata, self.cout, cin, 3, 3)
        self._idx = idx.flip(3).permute(0, 4, 2, 1, 3).reshape(-1).contiguous()
        self._shape = (strata * 3 * cin, self.cout, 3, 1)
        self._w = weight.detach().reshape(-1)[self._idx].view(self._shape)
        self.layer = ConvLayer(self._w, stride=1, pad=(1, 0), range_strata=self.strata)

    def repack(self, weight: torch.Tensor, token=None) -> None:
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        torch.index_select(weight.detach().reshape(-1), 0, self._idx, out=self._w.view(-1))
        self.layer.repack(self._w)

    def prepack_used(self) -> None:
        self.layer.prepack_used()

    def __call__(self, dy: torch.Tensor, out: Optional[torch.Tensor] = None, out_channel_offset=0, accumulate=False) -> torch.Tensor:
        hip.require_device(dy)
        b, h, w, c = dy.shape
        assert c == self.cout and dy.is_contiguous()
        z = self.layer(dy)
        if out is None:
            out = torch.empty((b, h, w, self.cin), dtype=torch.float32, device=dy.device)
        assert out.shape[:3] == (b, h, w) and out.is_contiguous()
        hip.call("pn_strat_dgrad_combine_f32", z.data_ptr(), b, h, w, self.cin, out.data_ptr(), out.shape[3], out_channel_offset, int(accumulate),
                 hip.stream())
        return out


def strat_channel_sum(dy: torch.Tensor, strata: int, out: torch.Tensor) -> torch.Tensor:
    """bias gradient of the RangeStratified convolution: sums of dy (B,H,W,C) over the pixels of every column band -> out (strata * C)"""
    b, h, w, c = dy.shape
    cols = channel_sum(dy.view(1, b * h, 1, w * c))       # per (column, channel) over the rows, fixed order
    torch.sum(cols.view(strata, w // strata, c), dim=1, out=out.view(strata, c))
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """f32 -> bf16 (round to nearest even) on the HIP kernel; same shape"""
    hip.require_device(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    hip.call("pn_f32_to_bf16", x.data_ptr(), y.data_ptr(), x.numel(), hip.stream())
    return y


def to_f32(x: torch.Tensor) -> torch.Tensor:
    hip.require_device(x)
    assert x.is_contiguous() and x.dtype == torch.bfloat16
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    hip.call("pn_bf16_to_f32", x.data_ptr(), y.data_ptr(), x.numel(), hip.stream())
    return y


# ------------------------------------------------------------------------------ next-3 target assignment
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
