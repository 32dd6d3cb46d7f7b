#!/usr/bin/env python3
"""One stride-1 3x3 layer at batch 4 (the training iteration's shapes): the F(4,3) kernel against NHWC -> planes + one 2-D chain launch.
r4 result: 96.6 vs 103 us at 128^2 x 128, 107 vs 77 at 64^2 x 256, 337 vs 382 at 256^2 x 128 -- at batch 4 the plain F(4,3) form has tiles
enough and the conversion eats the chain's gain, so training keeps F(4,3).   python tools/chain_single_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from partner_amd import ops, hip
dev = torch.device("cuda:0")
hip.load()
for (b, h, w, cin, cout) in [(4, 128, 128, 128, 128), (4, 64, 64, 256, 256), (4, 256, 256, 128, 128)]:
    wt = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    layer = ops.ConvLayer(wt, pad=1, act=ops.ACT_NONE)
    x = torch.relu(torch.randn(b, h, w, cin, device=dev))
    y0 = layer(x)
    ok = ops.conv_chain_supported([layer], b, h, w)
    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_w4 = t(lambda: layer(x))
    if ok:
        y1 = ops.conv_chain([layer], x)
        err = float((y1 - y0).abs().max() / y0.abs().max())
        t_ch = t(lambda: ops.conv_chain([layer], x))
        print(f"bs{b} {h}x{w} {cin}->{cout}: F(4,3) {t_w4:.1f} us, planes + 2-D chain {t_ch:.1f} us, max diff {err:.2e}")
    else:
        print(f"bs{b} {h}x{w}: chain not supported; F(4,3) {t_w4:.1f} us")
