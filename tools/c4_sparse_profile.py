#!/usr/bin/env python3
"""Sparse encoder of the Waymo PARTNER config on one 64-beam 180k-point sweep, a few iterations: meant to be run under
rocprofv3 --kernel-trace --stats (per-kernel breakdown of SpMiddleResNetFHD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import partner_amd as P
from partner_amd.voxel_generator import VoxelGenerator
from partner_amd.utils import synth

dev = torch.device("cuda:0")
cfg4 = P.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "waymo", "polar_partner_c4.py"))
m4 = P.build_detector(cfg4.model, train_cfg=cfg4.train_cfg, test_cfg=None)
geo = {k: getattr(m4.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
synth.load_filled(m4, 31)
for k, v in geo.items():
    getattr(m4.bbox_head, k).data.copy_(v)
m4 = m4.to(dev).eval()
vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
sw = torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=0)).to(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(iters + 3):
    if i == 3:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    voxels, coors, num = vg.generate(sw)
    coords4 = torch.cat([torch.zeros((coors.shape[0], 1), dtype=coors.dtype, device=dev), coors], 1)
    out = m4.backbone.forward_nhwc(m4.reader(voxels, num), coords4, 1, [1152, 2048, 40])
e1.record()
torch.cuda.synchronize()
print(f"sparse encoder: {e0.elapsed_time(e1) / iters:.3f} ms per sweep, {voxels.shape[0]} voxels")
