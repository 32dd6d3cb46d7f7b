"""Readers and BEV scatter modules (registered under the reference's names).

Reference: det3d/models/readers/pillar_encoder.py, voxel_encoder.py.
"""
from __future__ import annotations

import torch
from torch import nn

from . import hip, ops
from .builder import BACKBONES, READERS
from .nn_utils import PlanCache, build_norm_layer, eval_only


class PFNLayer(nn.Module):
    """Linear(no bias) + BN1d parameters of one pillar-feature layer (pillar_encoder.py:19-71).
    The BN1d is constructed but NOT applied on the dynamic path (forward_dynamic :63-71) -- it is
    kept so that state_dicts match."""

    def __init__(self, in_channels, out_channels, norm_cfg=None, last_layer=False):
        super().__init__()
        self.name = "PFNLayer"
        self.last_vfe = last_layer
        self.units = out_channels if last_layer else out_channels // 2
        self.norm_cfg = norm_cfg or dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        self.norm = build_norm_layer(self.norm_cfg, self.units)[1]


def _batch_size(data, grid_ind) -> int:
    b = data.get("batch_size") if isinstance(data, dict) else None
    if b is None:
        b = int(grid_ind[:, 0].max().item()) + 1
    return int(b)


@READERS.register_module
class DynamicPFNet(nn.Module):
    """Pillar feature net with dynamic voxelization (pillar_encoder.py:263-406)."""

    def __init__(self, num_input_features=4, num_filters=(64,), voxel_shape="cuboid", xyz_cluster=False,
                 raz_cluster=False, xy_center=False, ra_center=False, voxel_size=(0.2, 0.2, 4),
                 pc_range=(0, -40, -3, 70.4, 40, 1), norm_cfg=None):
        super().__init__()
        self.name = "DynamicPFNet"
        assert len(num_filters) > 0
        self.num_input = num_input_features
        self.voxel_shape = voxel_shape
        self.xyz_cluster, self.raz_cluster, self.xy_center, self.ra_center = xyz_cluster, raz_cluster, xy_center, ra_center
        nin = num_input_features + (3 if xyz_cluster else 0) + (2 if xy_center else 0) + (2 if ra_center else 0)
        if raz_cluster:
            nin += 2 if xyz_cluster else 3
        filters = [nin] + list(num_filters)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(filters[i], filters[i + 1], norm_cfg=norm_cfg, last_layer=(i == len(filters) - 2))
            for i in range(len(filters) - 1)])
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]
        self.y_offset = self.vy / 2 + pc_range[1]
        self.voxel_size, self.pc_range = list(voxel_size), list(pc_range)
        self.times = []

    # -- what the fused HIP kernel covers ---------------------------------------------------
    def _check_supported(self):
        full = self.xyz_cluster and self.raz_cluster and self.xy_center and self.ra_center
        if not (full and self.voxel_shape != "cuboid" and self.num_input == 7 and len(self.pfn_layers) == 2):
            raise NotImplementedError(
                "DynamicPFNet HIP kernel: only the cylinder grid with the full 16-channel decoration "
                "(xyz_cluster, raz_cluster, xy_center, ra_center) and two PFN layers is implemented")

    @property
    def out_channels(self) -> int:
        return self.pfn_layers[-1].units

    def encode(self, points: torch.Tensor, vi: ops.VoxelIndex, features, canvas, clear_index=None) -> bool:
        """device-only entry used by the detector fast path (no host sync).  ``clear_index``: the frame-index state whose counters the launch may
        zero on the way (-> True when it did: ops.dynamic_pfn)"""
        self._check_supported()
        return ops.dynamic_pfn(points, vi, self.pfn_layers[0].linear.weight.detach(), self.pfn_layers[1].linear.weight.detach(),
                               self.vx, self.vy, self.x_offset, self.y_offset, features, canvas, clear_index=clear_index)

    def forward(self, data):
        """data: dict(points (N,7) f32, grid_ind (N,4) int64 [b,z,theta,r], [batch_size]) ->
        (features (V,C) f32, unq (V,4) int64) exactly as pillar_encoder.py:393-406."""
        points, grid_ind = data["points"], data["grid_ind"]
        hip.require_device(points, grid_ind)
        batch = _batch_size(data, grid_ind)
        spec = ops.GridSpec.from_range(self.pc_range, self.voxel_size)
        keys = ops.keys_from_grid_ind(grid_ind.to(torch.int64).contiguous(), spec, batch)
        vi = ops.build_voxel_index(keys, spec, batch)
        feats = torch.empty((max(vi.n_cap, 1), self.out_channels), dtype=torch.float32, device=points.device)
        self.encode(points.contiguous(), vi, feats, None)
        v = vi.count()
        return feats[:v], vi.unq[:v]


@READERS.register_module
class DynamicVoxelEncoderV1(nn.Module):
    """unique + per-voxel mean of the point features (voxel_encoder.py:26-45)."""

    def __init__(self, num_input_features=7, out_channels=16, name="DynamicVoxelEncoderV1", voxel_size=None, pc_range=None,
                 grid_size=None):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features
        self.point_density = False
        self.voxel_size, self.pc_range, self.grid_size = voxel_size, pc_range, grid_size

    def forward(self, data):
        points, grid_ind = data["points"], data["grid_ind"]
        hip.require_device(points, grid_ind)
        batch = _batch_size(data, grid_ind)
        if data.get("grid_size") is not None:
            g = [int(v) for v in data["grid_size"]]
            spec = ops.GridSpec((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (g[0], g[1], g[2]))
        elif self.grid_size is not None:
            g = [int(v) for v in self.grid_size]
            spec = ops.GridSpec((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (g[0], g[1], g[2]))
        else:
            spec = ops.GridSpec.from_range(self.pc_range, self.voxel_size)
        keys = ops.keys_from_grid_ind(grid_ind.to(torch.int64).contiguous(), spec, batch)
        vi = ops.build_voxel_index(keys, spec, batch)
        mean = ops.scatter_mean(points.contiguous(), vi)
        v = vi.count()
        return mean[:v], vi.unq[:v]


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    """mean of the (<= P) points of each hard voxel (voxel_encoder.py:7-22)."""

    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        assert self.num_input_features == features.shape[-1]
        return ops.hard_voxel_mean(features, num_voxels)


@READERS.register_module
class PillarFeatureNet(nn.Module):
    """Static (hard-voxel) pillar feature net (pillar_encoder.py:74-169).  Registered so that
    configs naming it build; its forward is outside this round's hot path."""

    def __init__(self, num_input_features=4, num_filters=(64,), with_distance=False, voxel_size=(0.2, 0.2, 4),
                 pc_range=(0, -40, -3, 70.4, 40, 1), norm_cfg=None):
        super().__init__()
        self.name = "PillarFeatureNet"
        nin = num_input_features + 5 + (1 if with_distance else 0)
        filters = [nin] + list(num_filters)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(filters[i], filters[i + 1], norm_cfg=norm_cfg, last_layer=(i == len(filters) - 2))
            for i in range(len(filters) - 1)])
        self.num_input, self._with_distance = num_input_features, bool(with_distance)
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset, self.y_offset = self.vx / 2 + pc_range[0], self.vy / 2 + pc_range[1]
        self._plan = PlanCache()

    def _build_plan(self):
        if len(self.pfn_layers) > 2:
            raise NotImplementedError("PillarFeatureNet HIP kernel: one or two PFN layers")
        plan = []
        for l in self.pfn_layers:
            scale, shift = ops.fold_bn(l.norm.weight, l.norm.bias, l.norm.running_mean, l.norm.running_var, l.norm.eps)
            plan.append((l.linear.weight.detach().float().contiguous(), scale, shift))
        return plan

    def forward(self, features, num_voxels, coors):
        """features (V,P,F) f32, num_voxels (V,) int, coors (V,4) int [b,z,y,x] -> (V, C) as pillar_encoder.py:131-169 (eval mode)"""
        hip.require_device(features, num_voxels, coors)
        eval_only(self, "PillarFeatureNet")
        plan = self._plan.get(self, self._build_plan)
        v, p, f = features.shape
        assert f == self.num_input
        w0, s0, h0 = plan[0]
        w1, s1, h1 = plan[1] if len(plan) > 1 else (None, None, None)
        c0, c1 = w0.shape[0], (0 if w1 is None else w1.shape[0])
        out = torch.empty((v, c1 or c0), dtype=torch.float32, device=features.device)
        nv = torch.full((1,), v, dtype=torch.int32, device=features.device)
        hip.call("pn_static_pfn_fwd", features.contiguous().data_ptr(), num_voxels.to(torch.int32).contiguous().data_ptr(),
                 coors.to(torch.int32).contiguous().data_ptr(), nv.data_ptr(), v, p, f, int(self._with_distance), w0.data_ptr(), s0.data_ptr(),
                 h0.data_ptr(), c0, hip.ptr(w1), hip.ptr(s1), hip.ptr(h1), c1, float(self.vx), float(self.vy), float(self.x_offset),
                 float(self.y_offset), out.data_ptr(), hip.stream())
        return out


@BACKBONES.register_module
class DynamicPPScatter(nn.Module):
    """voxel features -> dense BEV canvas (pillar_encoder.py:409-432).  Returns a logical
    (B, C, ny=theta, nx=r) tensor stored channels-last."""

    def __init__(self, **kwargs):
        super().__init__()
        self.name = "DynamicPPScatter"

    def forward(self, voxel_features, unq, batch_size, grid_size):
        nx, ny = int(grid_size[0]), int(grid_size[1])
        canvas = ops.scatter_canvas(voxel_features, unq.to(torch.int64), int(batch_size), ny, nx)
        return ops.as_nchw(canvas)


@BACKBONES.register_module
class PointPillarsScatter(nn.Module):
    """static twin of DynamicPPScatter: coords (V,4) [b,z,y,x] (pillar_encoder.py:173-225)."""

    def __init__(self, num_input_features=64, norm_cfg=None, name="PointPillarsScatter", **kwargs):
        super().__init__()
        self.name = name
        self.nchannels = num_input_features

    def forward(self, voxel_features, coords, batch_size, input_shape):
        nx, ny = int(input_shape[0]), int(input_shape[1])
        canvas = ops.scatter_canvas(voxel_features, coords.to(torch.int64), int(batch_size), ny, nx)
        return ops.as_nchw(canvas)
