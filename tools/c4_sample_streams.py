#!/usr/bin/env python3
"""Dense stages of the Waymo PARTNER step at bs = 2 (SetBlocks, RPN, head): ONE launch sequence over the batch against the two samples on two
streams (branches of the same hipGraph).  The 2-D chain launches of the batch are 576 tiles on 256 CUs = 2.25 rounds paid as 3; two
per-sample sequences of 288-tile launches fill each other's tails."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partner_amd import hip, ops
from partner_amd.utils import legs

hip.load()
dev = torch.device("cuda:0")
m, cfg = legs.build_waymo_partner(dev)
torch.manual_seed(0)
x_sp = torch.randn((2, 256, 144, 256), device=dev) * 0.5
x_at = m.realign_nhwc(x_sp)
x_rpn = m.neck.forward_nhwc(x_at)
ops.probe_streams()
side = ops.concurrent_stream()


def per_sample(fn, x):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    y0 = fn(x[0:1])
    with torch.cuda.stream(side):
        y1 = fn(x[1:2])
    main.wait_stream(side)
    return y0, y1


def check(a, b):
    if isinstance(a, dict):
        return all(torch.equal(a[k][0:1], b[0][k]) and torch.equal(a[k][1:2], b[1][k]) for k in a)
    return torch.equal(a[0:1], b[0]) and torch.equal(a[1:2], b[1])


for name, fn, x in (("setblocks_x2", m.realign_nhwc, x_sp), ("rpn", m.neck.forward_nhwc, x_at), ("head", m.bbox_head.forward_nhwc, x_rpn),
                    ("rpn + head", lambda t: m.bbox_head.forward_nhwc(m.neck.forward_nhwc(t)), x_at),
                    ("setblocks + rpn + head", lambda t: m.bbox_head.forward_nhwc(m.neck.forward_nhwc(m.realign_nhwc(t))), x_sp)):
    same = check(fn(x), per_sample(fn, x))
    res = {"batch": [], "per-sample streams": [], "per-sample streams, in-flight hint": []}

    def hinted():
        with ops.frames_in_flight(2):
            return per_sample(fn, x)
    for rep in range(3):
        res["batch"].append(legs.graph_time_ms(lambda: fn(x), 10, 3)[0])
        res["per-sample streams"].append(legs.graph_time_ms(lambda: per_sample(fn, x), 10, 3)[0])
        res["per-sample streams, in-flight hint"].append(legs.graph_time_ms(hinted, 10, 3)[0])
    print(f"{name}: " + " | ".join(f"{k} {sorted(v)[1]:.3f} ms" for k, v in res.items()) + f" | same bits: {same}")
if "--diff" in sys.argv:
    a, b = m.bbox_head.forward_nhwc(x_rpn), per_sample(m.bbox_head.forward_nhwc, x_rpn)
    for k in a:
        for s in range(2):
            d = (a[k][s:s + 1] - b[s][k]).abs().max().item()
            print(k, s, d, a[k].abs().max().item())
    a1 = m.bbox_head.forward_nhwc(x_rpn[0:1].contiguous())
    for k in a:
        print("single vs batch", k, (a[k][0:1] - a1[k]).abs().max().item())

