"""Device-side counterpart of the reference's VoxelGenerator
(det3d/core/input/voxel_generator.py:5-48): same constructor, same properties, ``generate``
returns (voxels, coordinates, num_points_per_voxel) -- computed by the HIP hard-voxelization
kernels instead of the numba loop of det3d/ops/point_cloud/point_cloud_ops.py."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        self._point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self._voxel_size = np.array(voxel_size, dtype=np.float32)
        self._grid_size = np.round((self._point_cloud_range[3:] - self._point_cloud_range[:3]) / self._voxel_size).astype(np.int64)
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels

    def generate(self, points: torch.Tensor, max_voxels: int = -1):
        """points: (N, F) fp32 on the device.  -> voxels (V,P,F), coordinates (V,3) int32 [z,theta,r],
        num_points_per_voxel (V,) int32   (V read back from the device)."""
        mv = self._max_voxels if max_voxels == -1 else max_voxels
        voxels, coors, num, nv = ops.hard_voxelize(points.contiguous(), self._voxel_size, self._point_cloud_range,
                                                   self._max_num_points, mv)
        v = int(nv.item())
        return voxels[:v], coors[:v], num[:v]

    voxel_size = property(lambda self: self._voxel_size)
    max_num_points_per_voxel = property(lambda self: self._max_num_points)
    point_cloud_range = property(lambda self: self._point_cloud_range)
    grid_size = property(lambda self: self._grid_size)
