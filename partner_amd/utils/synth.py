"""Deterministic synthetic sweeps and parameter fills shared by goldens, tests and bench.

Nothing here depends on torch's RNG: every value comes from ``numpy.random.default_rng``
seeded from integers / CRC32 of names, so the exact same inputs and weights can be
re-created in the golden generator (this container, next to the reference import), in the
CPU tests and on the GPU box without shipping large fixtures.

Synthetic sweep recipe: SURVEY.md section 8(d) "Synthetic inputs".
Polar decoration: restatement of the reference's ``transform_points(pc, 'cylinder')``
(/root/reference/det3d/datasets/pipelines/utils.py:34-47).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

# nuScenes polar-pillar grid of the reference config
# (/root/reference/configs/nusc/pp/polarstream_det_n_seg_1_sector.py:10-19)
NUSC_RANGE = (0.3, -3.1488, -5.0, 50.476, 3.1488, 3.0)
NUSC_VOXEL = (0.098, 0.0123, 8.0)
# secondary synthetic grid named by BASELINE.json (0.3125 m x 0.05 rad -> 160 x 126 x 1)
COARSE_RANGE = (0.0, -3.15, -5.0, 50.0, 3.15, 3.0)
COARSE_VOXEL = (0.3125, 0.05, 8.0)
# Waymo PARTNER grid (/root/reference/configs/waymo/voxelnet/waymo_partner_36epoch.py:10-21)
WAYMO_RANGE = (0.3, -3.14368, -2.0, 75.18, 3.14368, 4.0)
WAYMO_VOXEL = (0.065, 0.00307, 0.15)


def synth_sweep_cart(n_points: int, seed: int = 0, rho_max: float = 50.0,
                     n_sweeps: int = 1) -> np.ndarray:
    """(N,5) float32 [x, y, z, intensity, dt] synthetic lidar sweep."""
    rng = np.random.default_rng(seed)
    rho = rng.uniform(1.0, rho_max, n_points)
    phi = rng.uniform(-np.pi, np.pi, n_points)
    z = rng.uniform(-3.0, 1.0, n_points)
    inten = rng.uniform(0.0, 1.0, n_points)
    if n_sweeps > 1:
        dt = 0.05 * rng.integers(0, n_sweeps, n_points).astype(np.float64)
    else:
        dt = np.zeros(n_points)
    pts = np.stack([rho * np.cos(phi), rho * np.sin(phi), z, inten, dt], axis=1)
    return pts.astype(np.float32)


def cart_to_polar_host(pc: np.ndarray) -> np.ndarray:
    """(N,5+) [x,y,z,rest] -> (N,7+) [rho,phi,z,x,y,rest] in the dtype of ``pc``.

    Same arithmetic as the reference's cylinder branch of ``transform_points``.
    """
    rho = np.sqrt(pc[:, 0] ** 2 + pc[:, 1] ** 2)
    phi = np.arctan2(pc[:, 1], pc[:, 0])
    return np.hstack((rho[:, None], phi[:, None], pc[:, 2:3], pc[:, :2], pc[:, 3:]))


def synth_sweep_polar(n_points: int, seed: int = 0, **kw) -> np.ndarray:
    return np.ascontiguousarray(cart_to_polar_host(synth_sweep_cart(n_points, seed, **kw)))


def synth_sweep_beams_cart(n_points: int, seed: int = 0, rho_max: float = 74.0, beams: int = 64, sensor_height: float = 1.8,
                           elev_deg=(-17.6, 2.4), n_objects: int = 60) -> np.ndarray:
    """(N,5) float32 [x, y, z, intensity, dt]: a spinning multi-beam lidar over a ground plane with box-shaped
    obstacles (SURVEY.md 8d "realistic variant").  Unlike the uniform sweep, the returns lie on 2-D surfaces, so the
    active set of a sparse 3-D backbone stays sparse through the strided stages (uniform points dilate to a dense grid)."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(rng.choice(np.linspace(elev_deg[0], elev_deg[1], beams), n_points))
    az = rng.uniform(-np.pi, np.pi, n_points)
    ground = np.where(elev < -1e-3, sensor_height / np.tan(-np.minimum(elev, -1e-3)), np.inf)   # range of the ground hit
    # obstacles: vertical cylinders of radius 1-3 m at random places; a ray stops at the first one in its azimuth window
    obj_r, obj_az = rng.uniform(5.0, rho_max, n_objects), rng.uniform(-np.pi, np.pi, n_objects)
    obj_w = rng.uniform(1.0, 3.0, n_objects) / obj_r                                              # half width in radians
    hit = np.full(n_points, np.inf)
    for r_, a_, w_ in zip(obj_r, obj_az, obj_w):
        d = np.abs(np.angle(np.exp(1j * (az - a_))))
        hit = np.where(d < w_, np.minimum(hit, r_), hit)
    rho = np.minimum(np.minimum(ground, hit), rho_max * rng.uniform(0.6, 1.0, n_points))         # far returns drop out at random ranges
    rho = np.clip(rho + rng.normal(0.0, 0.02, n_points), 0.5, rho_max)
    z = np.where(rho >= ground - 0.05, -sensor_height, rho * np.tan(elev)) + rng.normal(0.0, 0.02, n_points)
    pts = np.stack([rho * np.cos(az), rho * np.sin(az), z, rng.uniform(0.0, 1.0, n_points), np.zeros(n_points)], axis=1)
    return pts.astype(np.float32)


def synth_sweep_beams_polar(n_points: int, seed: int = 0, **kw) -> np.ndarray:
    return np.ascontiguousarray(cart_to_polar_host(synth_sweep_beams_cart(n_points, seed, **kw)))


def synth_gt_boxes(n: int, seed: int, rho_max: float = 48.0):
    """(n, 9) float32 [x, y, z, l, w, h, vx, vy, rot] + classes (n,) int 1..10: boxes of mixed size all over the range,
    a few outside the feature map and a few overlapping"""
    r = np.random.default_rng(seed)
    rho, az = r.uniform(0.5, rho_max + 6.0, n), r.uniform(-np.pi, np.pi, n)
    b = np.zeros((n, 9), np.float32)
    b[:, 0], b[:, 1] = rho * np.cos(az), rho * np.sin(az)
    b[:, 2] = r.uniform(-3, 1, n)
    b[:, 3:6] = np.exp(r.uniform(np.log(0.4), np.log(11.0), (n, 3)))
    b[:, 6:8] = r.standard_normal((n, 2)) * 3
    b[:, 8] = r.uniform(-np.pi, np.pi, n)
    b[1, :2] = b[0, :2] + 0.3            # two objects in (almost) the same cell
    return b, r.integers(1, 11, n).astype(np.int64)


def synth_vehicle_boxes(batch: int, max_boxes: int, seed: int, rho_max: float = 72.0):
    """example['global_box'] of the Waymo PARTNER head: (batch, max_boxes, 8) f32 rows [x, y, z, dx, dy, dz, heading, class = 1],
    zero rows as padding; a few boxes on the +-pi azimuth seam, one right at the sensor, one long truck"""
    r = np.random.default_rng(seed)
    out = np.zeros((batch, max_boxes, 8), np.float32)
    for b in range(batch):
        n = int(r.integers(max_boxes // 3, max_boxes - 2)) if b else max_boxes - 5
        rho, az = r.uniform(3.0, rho_max, n), r.uniform(-np.pi, np.pi, n)
        az[:3] = np.array([np.pi - 0.004, -np.pi + 0.006, np.pi - 0.02])[:min(3, n)]      # across / next to the seam
        rho[3 % n] = 1.2
        out[b, :n, 0], out[b, :n, 1] = rho * np.cos(az), rho * np.sin(az)
        out[b, :n, 2] = r.uniform(-1.0, 2.0, n)
        out[b, :n, 3] = r.uniform(3.5, 5.5, n)
        out[b, :n, 4] = r.uniform(1.6, 2.3, n)
        out[b, :n, 5] = r.uniform(1.4, 2.2, n)
        out[b, 4 % n, 3] = 14.0
        out[b, :n, 6] = r.uniform(-np.pi, np.pi, n)
        out[b, :n, 7] = 1.0
    return out


def synth_swv_preds(batch: int, h: int, w: int, seed: int, boxes: np.ndarray = None, offset_grid: np.ndarray = None):
    """head tensors of the geometry-aware head, logical (B, c, H, W) f32: hm (1), reg (2), height (1), dim (3, log), rot (2: cos, sin),
    iou (1), pred_centers (2), pred_vote_cls (1).  With ``boxes`` / ``offset_grid`` given, the cells next to ground-truth centres
    get predictions close to the truth (a trained head), so that the matching is decided by more than noise."""
    r = np.random.default_rng(seed)
    f = np.float32
    p = dict(hm=(r.standard_normal((batch, 1, h, w)) * 1.2 - 3.0).astype(f), reg=(r.standard_normal((batch, 2, h, w)) * 0.6).astype(f),
             height=(r.standard_normal((batch, 1, h, w)) * 0.5 + 0.5).astype(f), dim=(r.standard_normal((batch, 3, h, w)) * 0.2 + np.log(2.5)).astype(f),
             rot=r.standard_normal((batch, 2, h, w)).astype(f), iou=(r.standard_normal((batch, 1, h, w)) * 0.5).astype(f),
             pred_centers=(r.standard_normal((batch, 2, h, w)) * 0.8).astype(f), pred_vote_cls=(r.standard_normal((batch, 1, h, w)) - 2.0).astype(f))
    if boxes is not None and offset_grid is not None:
        g = np.asarray(offset_grid, f).reshape(2, h * w)
        for b in range(batch):
            for row in boxes[b]:
                if row[3] == 0:
                    continue
                d = (g[0] - row[0]) ** 2 + (g[1] - row[1]) ** 2
                for cell in np.argsort(d)[:3]:
                    y, x = divmod(int(cell), w)
                    p["hm"][b, 0, y, x] = f(r.normal(1.5, 0.7))
                    p["reg"][b, :, y, x] = (row[:2] - g[:, cell]) + r.normal(0, 0.15, 2)
                    p["height"][b, 0, y, x] = row[2] + r.normal(0, 0.1)
                    p["dim"][b, :, y, x] = np.log(row[3:6]) + r.normal(0, 0.05, 3)
                    p["rot"][b, :, y, x] = [np.cos(row[6]) + r.normal(0, 0.05), np.sin(row[6]) + r.normal(0, 0.05)]
    return p


def synth_raw_sweeps(n_sweeps: int, n_points: int, seed: int = 0):
    """raw nuScenes-style sweeps for the accumulation tests: list of (n,5) f32 [x,y,z,intensity,ring] (a good share of the points
    near the sensor so that remove_close matters), (n_sweeps,4,4) float64 rigid transforms (entry 0 = identity), time lags f32"""
    r = np.random.default_rng(seed)
    clouds, mats, lags = [], np.tile(np.eye(4), (n_sweeps, 1, 1)), np.zeros(n_sweeps, np.float32)
    for s in range(n_sweeps):
        c = synth_sweep_cart(n_points + 37 * s, seed=seed * 100 + s)
        near = r.random(len(c)) < 0.15
        c[near, :2] = r.uniform(-1.5, 1.5, (int(near.sum()), 2)).astype(np.float32)
        c[:, 4] = r.integers(0, 32, len(c)).astype(np.float32)          # ring index column of the .bin layout
        clouds.append(c)
        if s > 0:
            yaw, t = r.uniform(-0.2, 0.2), r.uniform(-3, 3, 3)
            mats[s, :3, :3] = [[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]
            mats[s, :3, 3] = t
            lags[s] = 0.05 * s
    return clouds, mats, lags


def _rng_for(name: str, base_seed: int) -> np.random.Generator:
    return np.random.default_rng([base_seed, zlib.crc32(name.encode())])


def seeded_normal(name: str, shape, base_seed: int = 0, scale: float = 1.0) -> np.ndarray:
    """deterministic N(0, scale) float32 array keyed by a name: test inputs that fixtures need not store"""
    return (scale * _rng_for(name, base_seed).standard_normal(tuple(shape))).astype(np.float32)


class Shape:
    """stand-in with a ``.shape`` for ``fill_state_dict`` (weights of a module that is not built here)"""

    def __init__(self, *shape):
        self.shape = tuple(shape)


def fill_state_dict(state: "OrderedDict[str, object]", base_seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic, name-keyed values for every entry of a ``state_dict``.

    * conv / linear weights (ndim >= 2): N(0, sqrt(2/fan_in))   (keeps activations O(1))
    * norm weights (1-D ``*.weight``):    U(0.5, 1.5)
    * biases (1-D ``*.bias``):            N(0, 0.1)
    * ``running_mean``:                   N(0, 0.1);  ``running_var``: U(0.5, 1.5)
    * ``num_batches_tracked``:            0
    Returned arrays have the dtype/shape of the originals.
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, ref in state.items():
        shape = tuple(ref.shape)
        rng = _rng_for(name, base_seed)
        if name.endswith("num_batches_tracked"):
            val = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_var"):
            val = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif name.endswith("running_mean"):
            val = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            val = (np.sqrt(2.0 / fan_in) * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("weight"):
            val = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        else:
            val = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        out[name] = val
    return out


def load_filled(module, base_seed: int = 0) -> None:
    """Overwrite every parameter/buffer of a torch module with ``fill_state_dict`` values."""
    import torch

    sd = module.state_dict()
    filled = fill_state_dict(sd, base_seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in filled.items()}, strict=True)
