#!/usr/bin/env python3
"""Waymo PARTNER config end to end (one 180k-point 64-beam sweep per iteration: voxelize -> VFE -> sparse encoder -> 2 x SetBlock -> RPN
-> E2ESWVoteHead), a few iterations: run under rocprofv3 --kernel-trace --stats.  argv: [f32 | bf16] [iterations] [sweeps per step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import partner_amd as P
from partner_amd.voxel_generator import VoxelGenerator
from partner_amd.utils import synth

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg4 = P.Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "waymo", "polar_partner_c4.py"))
m4 = P.build_detector(cfg4.model, train_cfg=cfg4.train_cfg, test_cfg=None)
geo = {k: getattr(m4.bbox_head, k).clone() for k in ("offset_grid", "xy_offset")}
synth.load_filled(m4, 31)
for k, v in geo.items():
    getattr(m4.bbox_head, k).data.copy_(v)
m4 = m4.to(dev).eval()
if mode == "bf16":
    m4.set_compute_dtype("bf16")
vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
sws = [torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=b)).to(dev) for b in range(batch)]


def frame4():
    vs, cs, ns, nv = [], [], [], []
    for b, sw in enumerate(sws):
        voxels, coors, num = vg.generate(sw)
        vs.append(voxels); ns.append(num); nv.append(int(voxels.shape[0]))
        cs.append(torch.cat([torch.full((coors.shape[0], 1), b, dtype=coors.dtype, device=dev), coors], 1))
    ex = dict(voxels=torch.cat(vs), coordinates=torch.cat(cs), num_points=torch.cat(ns), num_voxels=nv, shape=[np.array([1152, 2048, 40])] * batch)
    return m4(ex, return_loss=False)


for i in range(iters + 3):
    if i == 3:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    frame4()
e1.record()
torch.cuda.synchronize()
print(f"C4 end to end {mode}, {batch} sweep(s) per step: {e0.elapsed_time(e1) / iters:.3f} ms per step ({iters} iterations after 3 warm-ups)")
