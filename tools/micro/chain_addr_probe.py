#!/usr/bin/env python3
"""Does the 256 x 256 x 128 -> 128 chained layer's time depend on WHERE its two planes buffers sit?  The eager roofline pass of bench.py is
bimodal from process to process (dominant kernel 62 vs 67 us); this runs the same three-layer chain with the planes buffers carved out of
one big allocation at different offsets.   python tools/micro/chain_addr_probe.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from partner_amd import ops, hip
dev = torch.device("cuda:0")
lib = hip.load()
b, h, w, c = 1, 256, 256, 128
layers = [ops.ConvLayer(torch.randn(c, c, 3, 3, device=dev) * 0.05, pad=1, act=ops.ACT_RELU) for _ in range(3)]
n = lib.pn_wino4_planes_floats(b, h, w, c)
x = torch.relu(torch.randn(b, h, w, c, device=dev))
big = torch.empty(2 * n + (64 << 20) // 4, dtype=torch.float32, device=dev)
out = torch.empty((b, h, w, c), dtype=torch.float32, device=dev)
st = hip.stream()
print("planes floats", n, "bytes", 4 * n, "base", hex(big.data_ptr()))


def run(off0, off1, reps=30):
    b0 = big[off0 // 4: off0 // 4 + n]
    b1 = big[(4 * n + off1) // 4: (4 * n + off1) // 4 + n]
    bufs = [b0, b1]
    hip.call("pn_wino4_planes_from_nhwc_f32", x.data_ptr(), b, h, w, c, c, 0, 0, b0.data_ptr(), st)

    def chain():
        for k, l in enumerate(layers):
            last = k == 2
            d = ops._chain_desc(l, b, h, w, c, 0, transposed=False) if last else ops._chain_desc(l, b, h, w, transposed=False)
            wts = l.chain_weights(True, False)
            hip.call("pn_conv2d_wino24_chain_f32", C.byref(d), bufs[k & 1].data_ptr(), wts.data_ptr(), hip.ptr(l.scale), hip.ptr(l.shift),
                     None if last else bufs[(k + 1) & 1].data_ptr(), out.data_ptr() if last else None, st)
    for _ in range(10):
        chain()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        chain()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 3 * 1e3


for off0, off1 in [(0, 0), (256, 0), (4096, 0), (0, 4096), (65536, 0), (0, 65536), (1 << 20, 0), (0, 1 << 20), (2 << 20, 2 << 20), (4096, 8192), (1 << 21, 0), (0, 1 << 21),
                   (3 << 19, 5 << 18), (0, 0)]:
    print(f"offsets {off0:>9} {off1:>9}: {run(off0, off1):7.2f} us per layer")
