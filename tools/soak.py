#!/usr/bin/env python3
"""Memory / time stability of the long-running paths: N training iterations of the pillar model and of the PARTNER detector, N hipGraph
replays of the Waymo frame engine; prints allocated / reserved memory and the iteration time at intervals.   python tools/soak.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import partner_amd as P
from partner_amd import hip, ops
from partner_amd.utils import legs, synth
hip.load(); dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def report(tag, i, t0, n):
    torch.cuda.synchronize()
    print(f"{tag} iter {i:4d}: {1e3 * (time.perf_counter() - t0) / n:8.2f} ms/it (last block)  allocated {torch.cuda.memory_allocated() / 2**30:6.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB", flush=True)


# ---- pillar training step (the bench's own leg: 4 sweeps of 30k points per iteration)
m = P.build_detector(bench.c2_model_cfg()); synth.load_filled(m, 0); m = m.to(dev)
leg = bench.TrainLeg(m, dev, 0, 4, 30000, 10 * N)
t0 = time.perf_counter()
for i in range(N):
    leg.step(i)
    if (i + 1) % (N // 4) == 0:
        report("pillar-train", i + 1, t0, N // 4); t0 = time.perf_counter()
del leg, m
torch.cuda.empty_cache()
# ---- PARTNER detector training step (bs = 2)
from tools.train_partner_profile import build_model, build_example
from partner_amd.train_partner import PartnerTrainStep
mp = build_model(dev); ex = build_example(dev, 2)
ps = PartnerTrainStep(mp, total_steps=10 * N)
t0 = time.perf_counter()
NP = max(N // 4, 8)
for i in range(NP):
    ps.step(ex)
    if (i + 1) % (NP // 4) == 0:
        report("partner-train", i + 1, t0, NP // 4); t0 = time.perf_counter()
del ps, mp, ex
torch.cuda.empty_cache()
# ---- Waymo frame engine, bs = 2
from partner_amd.engine import FrameEngine
m4, _ = legs.build_waymo_partner(dev)
cart = torch.cat([torch.from_numpy(synth.synth_sweep_beams_cart(180000, seed=b)).to(dev) for b in range(2)])
eng = FrameEngine(m4, 2, 180000).capture()
t0 = time.perf_counter()
for i in range(N):
    eng.run(cart)
    if (i + 1) % (N // 4) == 0:
        report("waymo-graph-bs2", i + 1, t0, N // 4); t0 = time.perf_counter()
