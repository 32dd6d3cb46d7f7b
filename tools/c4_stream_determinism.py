#!/usr/bin/env python3
"""The Waymo PARTNER detector at bs = 2 with its dense stages per sample on two streams (VoxelNetV3.dense_stages_nhwc): N eager steps and N hipGraph
replays of the same frame must give the same bits every time (a race between the two sample streams would show up as a changing digest).
  python tools/c4_stream_determinism.py [N]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd import hip
from partner_amd.engine import FrameEngine
from partner_amd.utils import legs, synth

hip.load()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
m, _ = legs.build_waymo_partner(dev)
cart = torch.cat([torch.from_numpy(synth.synth_sweep_beams_cart(180000, seed=b)).to(dev) for b in range(2)])


def digest(out):
    h = hashlib.sha256()
    for k in sorted(out):
        h.update(out[k].contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


eng = FrameEngine(m, 2, 180000)
eng.cart.copy_(cart)
seen = set()
for i in range(N):
    seen.add(digest(eng._step()))
print("eager steps:", N, "distinct digests:", len(seen))
eng.capture()
seen_g = set()
for i in range(N):
    seen_g.add(digest(eng.run(cart)))
print("graph replays:", N, "distinct digests:", len(seen_g), "| eager == graph:", seen == seen_g)
