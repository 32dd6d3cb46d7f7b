#!/usr/bin/env python3
"""The sparse encoder of the Waymo PARTNER config (bs 2) captured as ONE hipGraph and replayed: run under
rocprofv3 --kernel-trace (tools/c4_sparse_graph_timeline.sh) to see the replay's kernels in time order with their queues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd.utils import legs, synth
from partner_amd.voxel_generator import VoxelGenerator

dev = torch.device("cuda:0")
m, cfg = legs.build_waymo_partner(dev)
vg = VoxelGenerator(synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, 150000)
vs, cs, ns = [], [], []
for b in range(2):
    voxels, coors, num = vg.generate(torch.from_numpy(synth.synth_sweep_beams_polar(180000, seed=b)).to(dev))[:3]
    vs.append(voxels); ns.append(num)
    cs.append(torch.cat([torch.full((coors.shape[0], 1), b, dtype=coors.dtype, device=dev), coors], 1))
voxels, coords, num = torch.cat(vs), torch.cat(cs), torch.cat(ns)
feats = m.reader(voxels, num)
run = lambda: m.backbone.forward_nhwc(feats, coords, 2, [1152, 2048, 40])
for _ in range(3):
    run()
torch.cuda.synchronize()
st = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    run()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        out = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(6):
    if i == 1:
        e0.record()
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"sparse encoder, one graph replay: {e0.elapsed_time(e1) / 5:.3f} ms")
