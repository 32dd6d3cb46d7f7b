#!/usr/bin/env python3
"""E2ESWVoteHead and 2 x SetBlock of the Waymo PARTNER config, a few iterations each: run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import partner_amd as P
from partner_amd.attention import SetBlock, waymo_bev_pos
from partner_amd.utils import synth

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "head"
iters = 10
if what == "head":
    tasks = [dict(num_class=1, class_names=["VEHICLE"])]
    head = P.build_bbox_head(dict(
        type="E2ESWVoteHead", in_channels=512, tasks=tasks, dataset="waymo", weight=2, code_weights=[1.0] * 8, out_size_factor=8,
        common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)}, voxel_shape="cylinder",
        CODER_CONFIG={"code_size": 7, "encode_angle_by_sincos": True},
        GT_PROCESSOR_CONFIG={"max_volumn_space": [75.18, 3.14368, 4.0], "min_volumn_space": [0.3, -3.14368, -2.0], "grid_size": np.array([1152, 2048, 40])},
        HEAD_CONFIG={"kernel_size": 3, "sw_head_version": "votev4", "window_size": 7, "sl_depth": [2], "code_size": 7, "encode_angle_by_sincos": True,
                     "iou_loss": True, "init_bias": -2.19, "num_classes": 1}))
    geo = {k: getattr(head, k).clone() for k in ("offset_grid", "xy_offset")}
    synth.load_filled(head, 4)
    for k, v in geo.items():
        getattr(head, k).data.copy_(v)
    head = head.to(dev).eval()
    xh = torch.randn((1, 256, 144, 512), device=dev)
    fn = lambda: head.forward_nhwc(xh)
else:
    pos = waymo_bev_pos()
    blks = []
    for i in range(2):
        b = SetBlock(in_dim=256, embed_dim_scale=1, num_heads=4, reso=(144, 256), mlp_ratio=4.0, qkv_bias=True, H_sp=144, W_sp=1, H=4, W=8,
                     pos=pos, shift=(i == 1)); synth.load_filled(b, 70 + i); blks.append(b.to(dev).eval())
    x = torch.randn((1, 144 * 256, 256), device=dev)
    def fn():
        y = x
        for b in blks:
            y = b(y)
        return y
for i in range(iters + 3):
    if i == 3:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn()
e1.record()
torch.cuda.synchronize()
print(f"{what}: {e0.elapsed_time(e1) / iters:.3f} ms")
