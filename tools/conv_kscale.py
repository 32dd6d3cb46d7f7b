#!/usr/bin/env python3
"""K-scaling probe: time per K step of the conv kernel = slope of time vs Cin (tuning tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd import ops
dev = torch.device("cuda:0")
H = int(os.environ.get("HW", "256"))
B = int(os.environ.get("BATCH", "1"))
res = {}
for cin in (128, 512, 1024):
    x = torch.randn((B, H, H, cin), device=dev)
    w = torch.randn((128, cin, 3, 3), device=dev) * 0.02
    layer = ops.ConvLayer(w, stride=1, pad=1, act=1)
    out = layer(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        layer(x, out=out)
    e1.record()
    torch.cuda.synchronize()
    res[cin] = e0.elapsed_time(e1) * 100  # us per launch
    print(f"cin={cin:5d} steps={9*cin//32:4d}  {res[cin]:8.1f} us  {2*B*H*H*128*cin*9/res[cin]*1e-6:6.1f} TF")
slope = (res[1024] - res[128]) / (9 * (1024 - 128) / 32)
print(f"per-step {slope:.3f} us ; fixed {res[128] - 36*slope:.1f} us ; asymptotic {2*B*H*H*128*32/slope*1e-6:.1f} TF")
