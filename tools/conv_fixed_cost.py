#!/usr/bin/env python3
"""Launches the 128-column 3x3 convolution at 9 / 18 / 36 / 72 K steps (Cin = 32 .. 256) on an HW x HW map; run under
rocprofv3 --kernel-trace: the durations against the step count give the per-step time and the fixed cost of a launch.
  HW=256 rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/conv_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd import ops
dev = torch.device("cuda:0")
H = int(os.environ.get("HW", "256"))
for cin in (32, 64, 128, 256):
    x = torch.randn((1, H, H, cin), device=dev)
    w = torch.randn((128, cin, 3, 3), device=dev) * 0.02
    layer = ops.ConvLayer(w, stride=1, pad=1, act=1)
    out = layer(x)
    for _ in range(10):
        layer(x, out=out)
    torch.cuda.synchronize()
