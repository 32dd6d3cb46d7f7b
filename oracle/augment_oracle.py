"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  numpy restatement of the reference's global augmentation with the random
draws passed in: prep.random_flip_both / global_rotation / global_scaling_v2 / global_translate_
(det3d/core/sampler/preprocess.py:803-832, 771-788, 835-839, 940-962; rotation_points_single_angle box_np_ops.py:182-204).
Pinned by tests/golden/augment.npz (outputs of the reference functions themselves under seeded np.random)."""
import numpy as np


def _rot_z(pts3: np.ndarray, angle: float) -> np.ndarray:
    s, c = np.sin(angle), np.cos(angle)
    m = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=pts3.dtype)
    return pts3 @ m


def global_augment(points: np.ndarray, boxes: np.ndarray, flip_y: bool, flip_x: bool, rotation: float, scale: float, translate=None):
    """points (N, F) f32, boxes (M, 7 | 9) f32 -> augmented copies"""
    p, b = points.copy(), boxes.copy()
    if flip_y:
        b[:, 1] = -b[:, 1]
        b[:, -1] = -b[:, -1] + np.pi
        p[:, 1] = -p[:, 1]
        if b.shape[1] > 7:
            b[:, 7] = -b[:, 7]
    if flip_x:
        b[:, 0] = -b[:, 0]
        p[:, 0] = -p[:, 0]
        b[:, -1] = -b[:, -1] + 2 * np.pi
        if b.shape[1] > 7:
            b[:, 6] = -b[:, 6]
    p[:, :3] = _rot_z(p[:, :3], rotation)
    b[:, :3] = _rot_z(b[:, :3], rotation)
    if b.shape[1] > 7:
        b[:, 6:8] = _rot_z(np.hstack([b[:, 6:8], np.zeros((b.shape[0], 1))]), rotation)[:, :2]
    b[:, -1] += rotation
    p[:, :3] *= scale
    b[:, :-1] *= scale
    if translate is not None:
        t = np.asarray(translate, np.float64).reshape(1, 3)
        p[:, :3] += t
        b[:, :3] += t
    return p, b
