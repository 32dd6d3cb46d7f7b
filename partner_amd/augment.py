"""Global augmentation of a training sample on the device (the part of ``Preprocess`` that touches every
point): flips, rotation about z, scaling, translation of the cloud and of the ground-truth boxes.

Reference: ``Preprocess.__call__`` det3d/datasets/pipelines/preprocess.py:107-117 calling ``prep.random_flip_both``,
``global_rotation``, ``global_scaling_v2``, ``global_translate_`` (det3d/core/sampler/preprocess.py:803-832, 771-788, 835-839,
940-962).  The random numbers are drawn here, on the host, from ``np.random`` in exactly the reference's order, so that a
seeded pipeline produces the reference's augmentation; the arithmetic on the N points and M boxes is one HIP launch each
(``pn_global_augment_f32``).  Ground-truth database sampling (``db_sampler``) is dataset plumbing and stays on the host."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip


@dataclass
class AugmentDraw:
    flip_y: bool          # first flip of random_flip_both (y -> -y)
    flip_x: bool          # second flip (x -> -x)
    rotation: float       # radians
    scale: float
    translate: Optional[np.ndarray]   # (3,) float64 or None (all standard deviations zero)


class GlobalAugment:
    """``global_rot_noise`` ([lo, hi] or a half-width), ``global_scale_noise`` (min, max), ``global_translate_std`` (scalar or 3)
    -- the fields of the reference's train_preprocessor config."""

    def __init__(self, global_rot_noise=(-0.78539816, 0.78539816), global_scale_noise=(0.95, 1.05), global_translate_std=0.0, flip_probability=0.5):
        self.rot = list(global_rot_noise) if isinstance(global_rot_noise, (list, tuple)) else [-global_rot_noise, global_rot_noise]
        self.scale = tuple(global_scale_noise)
        std = global_translate_std
        self.trans_std = np.array([std, std, std], dtype=np.float64) if not isinstance(std, (list, tuple, np.ndarray)) else np.asarray(std, np.float64)
        self.p = float(flip_probability)

    def draw(self) -> AugmentDraw:
        """np.random calls in the reference's order: two flip choices, rotation, scale, (three normals)"""
        p = self.p
        flip_y = bool(np.random.choice([False, True], replace=False, p=[1 - p, p]))
        flip_x = bool(np.random.choice([False, True], replace=False, p=[1 - p, p]))
        rot = float(np.random.uniform(self.rot[0], self.rot[1]))
        sc = float(np.random.uniform(self.scale[0], self.scale[1]))
        tr = None
        if not all(e == 0 for e in self.trans_std):
            # the reference draws x with std[0], y with std[1] and z with std[0] again (preprocess.py:951-957)
            tr = np.array([np.random.normal(0, self.trans_std[0], 1), np.random.normal(0, self.trans_std[1], 1),
                           np.random.normal(0, self.trans_std[0], 1)]).T.reshape(3).astype(np.float64)
        return AugmentDraw(flip_y, flip_x, rot, sc, tr)

    @staticmethod
    def apply(points: torch.Tensor, boxes: Optional[torch.Tensor], d: AugmentDraw) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """in place on device tensors: points (N, F >= 3) f32, boxes (M, 7 | 9) f32 [x, y, z, w, l, h, (vx, vy,) heading]"""
        hip.require_device(points)
        assert points.dtype == torch.float32 and points.is_contiguous() and points.dim() == 2
        m = 0
        if boxes is not None and boxes.numel():
            hip.require_device(boxes)
            assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.shape[1] in (7, 9)
            m = boxes.shape[0]
        ang = np.float32(d.rotation)
        tr = None if d.translate is None else (C.c_double * 3)(*[float(v) for v in d.translate])
        # sin / cos in double, then rounded into the float32 matrix -- as np.array([...], dtype=points.dtype) does
        hip.call("pn_global_augment_f32", points.data_ptr(), points.shape[0], points.shape[1], hip.ptr(boxes) if m else None, m,
                 boxes.shape[1] if m else 7, int(d.flip_y), int(d.flip_x), 1, float(np.float32(np.sin(d.rotation))), float(np.float32(np.cos(d.rotation))),
                 float(ang), float(np.float32(d.scale)), tr, hip.stream())
        return points, boxes

    def __call__(self, points: torch.Tensor, boxes: Optional[torch.Tensor]):
        return self.apply(points, boxes, self.draw())
