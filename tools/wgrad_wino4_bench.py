#!/usr/bin/env python3
"""F(4,3) weight gradient (csrc/conv_wgrad_wino4.hip) against the direct weight-gradient kernel and float64 autograd on the training
shapes (bs = 4): errors and event-timed launches.  python tools/wgrad_wino4_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd import ops, hip
dev = torch.device("cuda:0")
for (b, h, w, cin, cout) in [(1, 8, 16, 8, 12), (2, 33, 64, 36, 20), (4, 256, 256, 128, 128), (4, 128, 128, 128, 128), (4, 64, 64, 256, 256), (4, 128, 128, 384, 64)]:
    torch.manual_seed(0)
    x = torch.randn((b, h, w, cin), device=dev)
    dy = torch.randn((b, h, w, cout), device=dev)
    ops.R.conv_wgrad_wino4_min_quads = 0
    ops.R.conv_wgrad_wino4 = True
    g4 = ops.conv_wgrad(x, dy, 3, 3, 1, 1)
    ops.R.conv_wgrad_wino4 = False
    gd = ops.conv_wgrad(x, dy, 3, 3, 1, 1)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(False)
    wt = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(xr, wt, padding=1)
    (y * dy.permute(0, 3, 1, 2).double()).sum().backward()
    ref = wt.grad
    sc = float(ref.abs().max())
    e4, ed = float((g4.double() - ref).abs().max()) / sc, float((gd.double() - ref).abs().max()) / sc
    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    ops.R.conv_wgrad_wino4 = True; t4 = t(lambda: ops.conv_wgrad(x, dy, 3, 3, 1, 1))
    ops.R.conv_wgrad_wino4 = False; td = t(lambda: ops.conv_wgrad(x, dy, 3, 3, 1, 1))
    print(f"{b}x{h}x{w} {cin}->{cout}: err F(4,3) {e4:.2e} direct {ed:.2e} | F(4,3) {t4:.1f} us  direct {td:.1f} us  x{td / t4:.2f}")
