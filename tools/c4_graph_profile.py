#!/usr/bin/env python3
"""Waymo PARTNER config as ONE hipGraph replay per bs = 2 frame (FrameEngine over VoxelNetV3.forward_points: Cartesian points -> head
tensors, counts never leave the device) -- the regime of the bench line's c4.one_graph_bs2.  Run under rocprofv3 --kernel-trace --stats
(tools/prof_any.sh): every kernel of the replayed graph shows up once per replay.  argv: [f32 | bf16] [replays] [sweeps per replay]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd.engine import FrameEngine
from partner_amd.utils import legs, synth

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m, _ = legs.build_waymo_partner(dev)
if mode == "bf16":
    m.set_compute_dtype("bf16")
cart = torch.cat([torch.from_numpy(synth.synth_sweep_beams_cart(180000, seed=b)).to(dev) for b in range(batch)])
eng = FrameEngine(m, batch, 180000).capture()
for _ in range(3):
    eng.run(cart)
torch.cuda.synchronize()
lat = []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(cart)
    torch.cuda.synchronize()
    lat.append(1e3 * (time.perf_counter() - t0))
lat.sort()
print(f"C4 one graph {mode} bs {batch}: p50 {lat[len(lat) // 2]:.3f} ms over {reps} replays")
