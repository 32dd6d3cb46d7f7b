"""CPU restatement of sector streaming: TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench's cpu_baseline).

  voxelize_streaming_polar      det3d/datasets/pipelines/voxelization.py:305-393
  ConvContext / RPNTECP         det3d/models/necks/rpn_context.py:10-95   (trailing-edge padding)
  ConvBDCP / RPNBDCP            det3d/models/necks/rpn_context.py:98-215  (bidirectional padding)
  sector rotation of the boxes  det3d/models/bbox_heads/center_head.py:533-545

Pinned to the reference by tests/golden/stream.npz (tests/golden/make_golden.py::gen_stream runs the reference's Voxelization,
RPNTECP and RPNBDCP): tests/test_oracle_stream.py.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

SD = Dict[str, Tensor]


def voxelize_streaming_polar(points: np.ndarray, pc_range, voxel_size, nsectors: int):
    """points (N, F >= 5) polar rows [rho, phi, z, x, y, ...] of ONE sample -> per sector (points, grid_ind (n, 3) [z, theta, r])
    (voxelization.py:305-393, the detection part)"""
    rg, vs = np.asarray(pc_range, np.float32), np.asarray(voxel_size, np.float32)
    grid = np.round((rg[3:] - rg[:3]) / vs).astype(np.int64)
    min_az, max_az = rg[1], rg[4]
    interval = (max_az - min_az) / nsectors
    cur_grid = grid.copy()
    cur_grid[1] //= nsectors
    out = []
    for i in range(nsectors):
        lo, hi = min_az + i * interval, min_az + (i + 1) * interval
        if i == 0:
            idx = np.where(points[:, 1] < hi)[0]
        elif i == nsectors - 1:
            idx = np.where(points[:, 1] >= lo)[0]
        else:
            idx = np.where((points[:, 1] >= lo) & (points[:, 1] < hi))[0]
        p = points[idx].copy()
        p[:, 1] -= lo - rg[1]
        p[:, 3] = p[:, 0] * np.cos(p[:, 1])
        p[:, 4] = p[:, 0] * np.sin(p[:, 1])
        gi = np.floor(np.clip((p[:, :3] - rg[:3]) / vs, a_min=0, a_max=cur_grid - 1)).astype(np.int64)[:, ::-1]
        out.append((p, gi.copy()))
    return out, cur_grid


def _bn_relu(sd: SD, p: str, x: Tensor, eps=1e-3) -> Tensor:
    return F.relu(F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, eps))


def _deblocks(sd: SD, prefix: str, i: int, x: Tensor, us_strides, up_start: int):
    j = i - up_start
    if j < 0:
        return None
    p = f"{prefix}deblocks.{j}."
    s = us_strides[j]
    if s > 1:
        y = F.conv_transpose2d(x, sd[p + "0.weight"], stride=int(s))
    else:
        k = int(np.round(1 / s))
        y = F.conv2d(x, sd[p + "0.weight"], stride=k)
    return _bn_relu(sd, p + "1.", y)


def rpn_tecp(sd: SD, prefix: str, x: Tensor, layer_nums, ds_strides, us_strides, prev_context: List[Tensor] = ()):
    """RPNTECP.forward (rpn_context.py:75-95, ConvContext.forward :30-44); eval-mode BatchNorm (eps 1e-3)"""
    prev, cur, ups = list(prev_context), [], []
    up_start = len(layer_nums) - len(us_strides)
    for i, n in enumerate(layer_nums):
        for k in range(n + 1):
            p = f"{prefix}blocks.{i}.{k}.block."
            pad = 1
            cur.append(x[:, :, -pad:, :])
            if not prev:
                xp = F.pad(x, (1, 1, 1, 1))
            else:
                xp = F.pad(torch.cat([prev.pop(0), x], 2), (1, 1, 0, pad))
            x = _bn_relu(sd, p + "1.", F.conv2d(xp, sd[p + "0.weight"], stride=ds_strides[i] if k == 0 else 1))
        u = _deblocks(sd, prefix, i, x, us_strides, up_start)
        if u is not None:
            ups.append(u)
    return (torch.cat(ups, 1) if ups else x), cur


def rpn_bdcp(sd: SD, prefix: str, x: Tensor, layer_nums, ds_strides, us_strides, prev_sweep: List[Tensor] = (), prev_context: List[Tensor] = (),
             sec_id=0, nsectors=1, mode="feature_only", cfg_nsectors=1):
    """RPNBDCP.forward (rpn_context.py:191-215, ConvBDCP.forward :112-158), restated literally"""
    prev, cur, ups, layer_id = list(prev_context), [], [], 0
    up_start = len(layer_nums) - len(us_strides)
    pad = 1
    for i, n in enumerate(layer_nums):
        for k in range(n + 1):
            p = f"{prefix}blocks.{i}.{k}.block."
            cur.append(x)
            if mode == "feature_only":
                if nsectors == 1:
                    xp = F.pad(x, (0, 0, pad, pad), mode="circular")
                else:
                    t = x.reshape([nsectors, x.shape[0] // nsectors, x.shape[1], x.shape[2], x.shape[3]])
                    tmp = torch.cat((t[:-1, :, :, -pad:, :], t[1:]), -2)
                    t = torch.cat((tmp, F.pad(t[-1:], (0, 0, 0, pad))), 0)
                    tmp = torch.cat((t[:-1], t[1:, :, :, :pad:, :]), -2)
                    t = torch.cat((F.pad(t[:1], (0, 0, pad, 0)), tmp), 0)
                    xp = t.reshape((-1, t.shape[-3], t.shape[-2], t.shape[-1]))
                xp = F.pad(xp, (pad, pad, 0, 0))
            else:
                ps = prev_sweep[layer_id]
                layer_id = (layer_id + 1) % len(prev_sweep)
                full_az, az = ps.shape[-2], x.shape[-2]
                ns = full_az // az
                if ns == 1:
                    xp = F.pad(x, (0, 0, pad, pad), mode="circular")
                elif sec_id == 0:
                    if cfg_nsectors == ns:
                        xp = torch.cat([ps[:, :, -pad:, :], x, ps[:, :, (sec_id + 1) * az:((sec_id + 1) * az + pad), :]], 2)
                    else:
                        xp = torch.cat([F.pad(x, (0, 0, pad, 0)), ps[:, :, (sec_id + 1) * az:((sec_id + 1) * az + pad), :]], 2)
                elif sec_id == ns - 1:
                    pc = prev.pop(0)
                    if cfg_nsectors == ns:
                        xp = torch.cat([pc[:, :, -pad:, :], x, ps[:, :, :pad, :]], 2)
                    else:
                        xp = torch.cat([pc[:, :, -pad:, :], F.pad(x, (0, 0, 0, pad))], 2)
                else:
                    pc = prev.pop(0)
                    xp = torch.cat([pc[:, :, -pad:, :], x, ps[:, :, (sec_id + 1) * az:((sec_id + 1) * az + pad), :]], 2)
                xp = F.pad(xp, (pad, pad, 0, 0))
            x = _bn_relu(sd, p + "1.", F.conv2d(xp, sd[p + "0.weight"], stride=ds_strides[i] if k == 0 else 1))
        u = _deblocks(sd, prefix, i, x, us_strides, up_start)
        if u is not None:
            ups.append(u)
    return (torch.cat(ups, 1) if ups else x), cur


def rotate_sector_boxes(boxes: np.ndarray, angle: float) -> np.ndarray:
    """center_head.py:533-545: boxes (n, 7 | 9) of a sector back into the sweep's frame"""
    b = torch.from_numpy(np.asarray(boxes, np.float32).copy())
    rot_sin, rot_cos = np.sin(-angle), np.cos(angle)
    m = torch.tensor([[rot_cos, -rot_sin], [rot_sin, rot_cos]], dtype=torch.float)
    b[:, :2] = b[:, :2] @ m
    b[:, -1] -= angle
    if b.shape[1] > 7:
        b[:, 6:8] = b[:, 6:8] @ m
    return b.numpy()


def warp_prev_sweep(cur_sweep: List[Tensor], transform: Tensor, nsectors: int, pc_range) -> List[Tensor]:
    """the `feature_only` tail of PolarStreamBDCP.forward_one_sweep (polarstream.py:318-372, get_grids :223-238, get_center :239-247):
    per-layer inputs of the stacked sectors (nsectors * bs, C, h, W) -> whole-sweep maps (bs, C, nsectors * h, W) resampled at the
    rotated cell positions with torch's own grid_sample (bilinear, zeros, align_corners=False).  transform: (bs, 2, 2)."""
    out = []
    center = [(pc_range[3] + pc_range[0]) / 2, (pc_range[4] + pc_range[1]) / 2]
    half_a, half_r = (pc_range[4] - pc_range[1]) / 2, (pc_range[3] - pc_range[0]) / 2
    bs = transform.shape[0]
    for x in cur_sweep:
        if nsectors > 1:
            x = x.reshape((nsectors, bs, -1, x.shape[-2], x.shape[-1]))
            x = torch.cat([x[j] for j in range(nsectors)], -2)
        H, W = x.shape[-2], x.shape[-1]
        ga, gr = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        ga = (pc_range[4] - pc_range[1]) / H * ga + pc_range[1]
        gr = (pc_range[3] - pc_range[0]) / W * gr + pc_range[0]
        grid = torch.stack([gr * torch.cos(ga), gr * torch.sin(ga)], -1)
        grid = torch.einsum("bjk,mnk->bmnj", transform.float(), grid)
        rho = torch.norm(grid, dim=-1, keepdim=True)
        az = torch.atan2(grid[:, :, :, 1], grid[:, :, :, 0]).unsqueeze(-1)
        rho = (rho - center[0]) / half_r
        az = (az - center[1]) / half_a
        out.append(F.grid_sample(x, torch.cat([rho, az], -1), align_corners=False))
    return out

