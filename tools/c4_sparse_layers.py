#!/usr/bin/env python3
"""Per-layer table of the sparse encoder of the Waymo PARTNER config at bs 2 (BASELINE configs[3]): live sites, existing (site, tap) pairs,
the rows the grouped kernels issue (32 x the union mask of every live group), kernel time (events around each launch, eager) and the
fractions of the fp32 MFMA peak.  python tools/c4_sparse_layers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from partner_amd.utils import legs, synth
from partner_amd.voxel_generator import VoxelGenerator
from partner_amd.sparse_backbone import SpMiddleResNetFHD

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
rec, orig = [], SpMiddleResNetFHD._conv


def spy(f, n_rows, nbr, count, cap, layer, act, residual=None, groups=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    out = orig(f, n_rows, nbr, count, cap, layer, act, residual, groups)
    e1.record()
    torch.cuda.synchronize()
    rec.append((nbr, count, layer, groups, e0.elapsed_time(e1) * 1e3))
    return out


SpMiddleResNetFHD._conv = staticmethod(spy)
run()
SpMiddleResNetFHD._conv = staticmethod(orig)
PEAK = 157.3
print(f"{'layer':>14s} {'sites':>8s} {'pairs/site':>10s} {'issued/useful':>13s} {'us':>8s} {'TF/s issued':>11s} {'frac':>6s} {'frac useful':>11s}")
tot_t = tot_i = tot_u = 0.0
for nbr, count, layer, groups, us in rec:
    n = int(count.item())
    f = 2.0 * layer["cin"] * layer["cout"]
    pairs = float((nbr[:n] >= 0).sum())
    issued = pairs
    if groups is not None and layer["cout"] >= 32:
        gm = groups[1][:(n + 31) // 32].to(torch.int64) & 0xffffffff
        issued = float(sum(((gm >> t) & 1) for t in range(layer["taps"])).sum()) * 32
    tot_t += us; tot_i += issued * f; tot_u += pairs * f
    print(f"{layer['cin']:4d}->{layer['cout']:4d} t{layer['taps']:2d} {n:8d} {pairs / max(n, 1):10.2f} {issued / max(pairs, 1):13.3f} {us:8.1f} {issued * f / us / 1e6:11.1f} "
          f"{issued * f / us / 1e6 / PEAK:6.3f} {pairs * f / us / 1e6 / PEAK:11.3f}")
print(f"total {tot_t:.0f} us, issued {tot_i / 1e9:.1f} GF ({tot_i / tot_t / 1e6 / PEAK:.3f} of peak), useful {tot_u / 1e9:.1f} GF ({tot_u / tot_t / 1e6 / PEAK:.3f})")
