"""A small reverse-mode tape over the HIP primitives, for the TRAINING forms of the attention blocks.

The BEV convolution path has hand-written fused backward passes (train.py).  The SetBlock (set_transformer.py:37-493) and
the shifted-window stage of E2ESWVoteHead are long compositions of linear layers, LayerNorm, GELU, softmax and small
per-window matrix products, so their backward is assembled from per-primitive backward kernels by this tape instead of one
monolithic kernel: every forward primitive records a closure that turns the gradient of its output into gradients of its
inputs.  All arithmetic is in csrc/ (MFMA GEMM / wgrad / dgrad kernels for the linear layers, csrc/autodiff.hip for the rest);
this file only wires pointers.  In the reference the same graph is torch autograd's (det3d/torchie/trainer/trainer.py:275-300
calls loss.backward())."""
from __future__ import annotations

import os

import ctypes as C
from typing import Callable, List, Optional, Sequence

import torch

from . import hip, ops
from .routes import R


class Node:
    __slots__ = ("v", "g", "bw", "owned", "name", "needs_grad")

    def __init__(self, v: torch.Tensor, bw: Optional[Callable] = None, name: Optional[str] = None, needs_grad=True):
        self.v, self.bw, self.name, self.needs_grad = v, bw, name, needs_grad
        self.g: Optional[torch.Tensor] = None
        self.owned = False


class Tape:
    def __init__(self):
        self.nodes: List[Node] = []
        self.params: List[Node] = []
        self._views = []   # (leaf, reshaped alias) pairs
        self._slices = []  # (leaf, row-slice alias, lo, hi)

    def param(self, v: torch.Tensor, name: str) -> Node:
        n = Node(v, None, name)
        self.params.append(n)
        return n

    def reshaped(self, leaf: Node, shape) -> Node:
        """alias of a leaf under another shape (a Conv1d(k=1) weight used as a matrix); its gradient is routed back to the leaf"""
        alias = Node(leaf.v.view(shape), None, leaf.name)
        self._views.append((leaf, alias))
        return alias

    def sliced(self, leaf: Node, lo: int, hi: int) -> Node:
        """alias of rows [lo, hi) of a leaf (the q / k / v thirds of a fused qkv weight); gradient routed into the leaf's rows"""
        alias = Node(leaf.v[lo:hi], None, leaf.name)
        self._slices.append((leaf, alias, lo, hi))
        return alias

    def const(self, v: torch.Tensor) -> Node:
        return Node(v, None, None, needs_grad=False)

    def input(self, v: torch.Tensor) -> Node:
        return Node(v, None, "input")

    def new(self, v: torch.Tensor, bw: Callable) -> Node:
        n = Node(v, bw)
        self.nodes.append(n)
        return n

    def backward(self, root: Optional[Node] = None, grad: Optional[torch.Tensor] = None) -> None:
        """reverse sweep; ``root`` / ``grad`` seed one output (further outputs may have been seeded with ``accumulate``)"""
        if root is not None:
            accumulate(root, grad)
        for n in reversed(self.nodes):
            if n.g is not None:
                n.bw(n.g)
                n.g = None   # interior gradients are dead after use
        side_join(self.nodes[-1].v.device if self.nodes else None)      # every parameter gradient is complete before it is routed / read
        for leaf, alias in self._views:
            if alias.g is not None:
                accumulate(leaf, alias.g.view(leaf.v.shape), own=alias.owned)
                alias.g = None
        touched = {}
        for leaf, alias, lo, hi in self._slices:
            if alias.g is None:
                continue
            if id(leaf) not in touched:
                touched[id(leaf)] = (leaf, torch.zeros_like(leaf.v))
            touched[id(leaf)][1][lo:hi].copy_(alias.g)
            alias.g = None
        for leaf, full in touched.values():
            accumulate(leaf, full, own=True)


def side_run(fn, *reads) -> None:
    """queue a layer's weight-gradient launches (and the accumulation into the parameter's gradient) on the device's second stream: weight
    gradients run beside the data-gradient chain (ops.SideStream).  A module-level function on purpose: a backward closure that
    held its Tape would close a reference cycle (tape -> node -> closure -> tape) and keep a whole iteration's activations alive until the
    cyclic collector runs -- 12.7 GB more per iteration when it was tried."""
    dev = next((r.device for r in reads if r is not None), None)
    ops.shared_side_stream(dev if dev is not None else "cpu").run(fn, *reads)


def side_join(device) -> None:
    if device is not None:
        ops.shared_side_stream(device).join()


def accumulate(node: Node, g: torch.Tensor, own=False) -> None:
    """node.g += g; ``own``: the caller hands the buffer over (it may be updated in place later)"""
    if not node.needs_grad:
        return
    if node.g is None:
        node.g, node.owned = g, own
    elif node.owned:
        ops.add(node.g, g, out=node.g)
    else:
        node.g, node.owned = ops.add(node.g, g), True


def _i64(vals: Sequence[int]):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


def _i32(vals: Sequence[int]):
    return (C.c_int32 * len(vals))(*[int(v) for v in vals])


# ------------------------------------------------------------------------------------------------ primitives
def linear(t: Tape, x: Node, w: Node, b: Optional[Node], k_pad: Optional[int] = None, relu=False) -> Node:
    """y = [relu](x @ w^T + b); x (M, K) contiguous.  ``k_pad``: x already carries k_pad >= K columns (zeros past K), the weight
    is padded to match (the MFMA loader reads 4 input channels at a time)."""
    wv = w.v
    n_out, k = wv.shape
    if k_pad is not None and k_pad != k:
        wp = torch.zeros((n_out, k_pad), dtype=torch.float32, device=wv.device)
        wp[:, :k].copy_(wv)
        wv = wp
    kk = wv.shape[1]
    m = x.v.shape[0]
    y = ops.GemmLayer(wv, None if b is None else b.v)(x.v, act=ops.ACT_RELU if relu else ops.ACT_NONE)

    def bw(dy):
        if relu:
            dy = ops.relu_bwd(y, dy)
        dy4 = dy.view(1, m, 1, n_out)
        if x.needs_grad:
            dx = ops.ConvDgrad(wv.view(n_out, kk, 1, 1), 1, 0)(dy4).view(m, kk)
            accumulate(x, dx, own=True)
        def weight_grads():
            dw = ops.conv_wgrad(x.v.view(1, m, 1, kk), dy4, 1, 1).view(n_out, kk)
            accumulate(w, dw[:, :k].contiguous() if kk != k else dw, own=True)
            if b is not None:
                accumulate(b, ops.channel_sum(dy4), own=True)

        side_run(weight_grads, dy, x.v)

    return t.new(y, bw)


def layernorm(t: Tape, x: Node, gamma: Node, beta: Node, eps: float, want_chan_mean=False):
    out = ops.layernorm(x.v, gamma.v, beta.v, eps, want_chan_mean=want_chan_mean)
    y, cm = out if want_chan_mean else (out, None)
    rows, c = x.v.shape

    def bw(dy):
        lib = hip.load()
        dx = torch.empty_like(x.v)
        dg = torch.empty(c, dtype=torch.float32, device=dy.device)
        db = torch.empty(c, dtype=torch.float32, device=dy.device)
        nbytes = lib.pn_layernorm_bwd_workspace_bytes(rows, c)
        ws = ops._workspace(nbytes, dy.device)
        hip.call("pn_layernorm_bwd_f32", x.v.data_ptr(), dy.data_ptr(), gamma.v.data_ptr(), float(eps), rows, c, dx.data_ptr(), dg.data_ptr(),
                 db.data_ptr(), 0, ws.data_ptr(), nbytes, hip.stream())
        accumulate(x, dx, own=True)
        accumulate(gamma, dg, own=True)
        accumulate(beta, db, own=True)

    n = t.new(y, bw)
    return (n, cm) if want_chan_mean else n


def gelu(t: Tape, x: Node) -> Node:
    y = torch.empty_like(x.v)
    hip.call("pn_gelu_f32", x.v.data_ptr(), y.data_ptr(), x.v.numel(), hip.stream())

    def bw(dy):
        dx = torch.empty_like(dy)
        hip.call("pn_gelu_bwd_f32", x.v.data_ptr(), dy.data_ptr(), dx.data_ptr(), dy.numel(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def add(t: Tape, a: Node, b: Node) -> Node:
    y = ops.add(a.v, b.v)

    def bw(dy):
        accumulate(a, dy)
        accumulate(b, dy)

    return t.new(y, bw)


def view(t: Tape, x: Node, shape) -> Node:
    """reshape of a contiguous value (no copy); the gradient is viewed back"""
    shp = x.v.shape

    def bw(dy):
        accumulate(x, dy.view(shp))

    return t.new(x.v.view(shape), bw)


def add_broadcast(t: Tape, x: Node, b: Node, batch: int) -> Node:
    """y[i] = x[i] + b for the ``batch`` equal leading slices of x (a per-window bias shared by every sample)"""
    y = torch.empty_like(x.v)
    xs, ys = x.v.view(batch, -1), y.view(batch, -1)
    bf = b.v.view(-1)
    for i in range(batch):
        ops.add(xs[i], bf, out=ys[i])

    def bw(dy):
        accumulate(x, dy)
        d = dy.view(batch, -1)
        db = d[0].clone() if batch == 1 else ops.add(d[0], d[1])
        for i in range(2, batch):
            ops.add(db, d[i], out=db)
        accumulate(b, db.view(b.v.shape), own=True)

    return t.new(y, bw)


def dropout(t: Tape, x: Node, p: float, seed: int, row_len=1) -> Node:
    """nn.Dropout (row_len 1) / DropPath (row_len = one sample); p == 0 is the identity and records nothing"""
    if p <= 0.0:
        return x
    y, mask = torch.empty_like(x.v), torch.empty_like(x.v)
    hip.call("pn_dropout_f32", x.v.data_ptr(), x.v.numel(), int(row_len), float(p), int(seed) & (2**64 - 1), y.data_ptr(), mask.data_ptr(),
             hip.stream())

    def bw(dy):
        dx = torch.empty_like(dy)
        hip.call("pn_mul_f32", dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def _contract_raw(a, sa, b, sb, c, sc, dims, alpha, acc=False):
    hip.call("pn_contract_f32", a.data_ptr(), _i64(sa), b.data_ptr(), _i64(sb), c.data_ptr(), _i64(sc), _i32(dims), float(alpha), int(acc),
             hip.stream())


def contract(t: Tape, a: Node, sa, b: Node, sb, out_shape, sc, dims, alpha=1.0) -> Node:
    """C[g, m, n] = alpha * sum_k A[g, m, k] * B[g, n, k]; see pn_contract_f32 for the stride / dims vectors (7 / 7 / 7 / 9 ints).
    The stride maps of A and B must be bijections onto their buffers (true of the head / window permutations used here)."""
    sa, sb, sc, dims = list(sa), list(sb), list(sc), list(dims)
    y = torch.empty(out_shape, dtype=torch.float32, device=a.v.device)
    _contract_raw(a.v, sa, b.v, sb, y, sc, dims, alpha)
    g = dims[:3]

    def bw(dy):
        if a.needs_grad:   # dA[g, m, k] = alpha * sum_n dC[g, m, n] * B[g, n, k]
            da = torch.zeros_like(a.v)
            _contract_raw(dy, sc[:3] + sc[3:5] + sc[5:7], b.v, sb[:3] + sb[5:7] + sb[3:5], da, sa[:3] + sa[3:5] + sa[5:7],
                          g + dims[3:5] + dims[7:9] + dims[5:7], alpha)
            accumulate(a, da, own=True)
        if b.needs_grad:   # dB[g, n, k] = alpha * sum_m dC[g, m, n] * A[g, m, k]
            db = torch.zeros_like(b.v)
            _contract_raw(dy, sc[:3] + sc[5:7] + sc[3:5], a.v, sa[:3] + sa[5:7] + sa[3:5], db, sb[:3] + sb[3:5] + sb[5:7],
                          g + dims[5:7] + dims[7:9] + dims[3:5], alpha)
            accumulate(b, db, own=True)

    return t.new(y, bw)


def softmax(t: Tape, x: Node, outer: int, n: int, inner: int) -> Node:
    y = torch.empty_like(x.v)
    hip.call("pn_softmax_f32", x.v.data_ptr(), y.data_ptr(), outer, n, inner, hip.stream())

    def bw(dy):
        dx = torch.empty_like(dy)
        hip.call("pn_softmax_bwd_f32", y.data_ptr(), dy.data_ptr(), dx.data_ptr(), outer, n, inner, hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def roll_w(t: Tape, x: Node, b: int, h: int, w: int, c: int, shift: int) -> Node:
    """torch.roll(x.view(b, h, w, c), shift, dims=2)"""
    if shift % w == 0:
        return x
    y = torch.empty_like(x.v)
    hip.call("pn_roll_w_f32", x.v.data_ptr(), b, h, w, c, int(shift), y.data_ptr(), hip.stream())

    def bw(dy):
        dx = torch.empty_like(dy)
        hip.call("pn_roll_w_f32", dy.data_ptr(), b, h, w, c, -int(shift), dx.data_ptr(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def roll_w_raw(x: torch.Tensor, b: int, h: int, w: int, c: int, shift: int) -> torch.Tensor:
    if shift % w == 0:
        return x
    y = torch.empty_like(x)
    hip.call("pn_roll_w_f32", x.data_ptr(), b, h, w, c, int(shift), y.data_ptr(), hip.stream())
    return y


def pair_diff(a: torch.Tensor, sa, b: torch.Tensor, sb, dims, cols=4) -> torch.Tensor:
    total = 1
    for d in dims:
        total *= d
    rel = torch.empty((total, cols), dtype=torch.float32, device=a.device)
    hip.call("pn_pair_diff_f32", a.data_ptr(), _i64(sa), b.data_ptr(), _i64(sb), _i32(dims), cols, rel.data_ptr(), hip.stream())
    return rel


def batchnorm_rows(t: Tape, x: Node, bn: torch.nn.BatchNorm1d, gamma: Node, beta: Node, training: bool) -> Node:
    """BatchNorm1d + ReLU over rows (rows, C) (the Conv1d/BN1d/ReLU of the relative-position MLPs, set_transformer.py:66-71):
    batch statistics (and a running-stat update) in training mode, the running statistics otherwise"""
    rows, c = x.v.shape
    if training:
        y, stat = ops.batchnorm_train(x.v.view(1, rows, 1, c), gamma.v, beta.v, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                      act=ops.ACT_RELU)

        def bw(dy):
            dx, dg, db = ops.batchnorm_bwd(x.v.view(1, rows, 1, c), dy.view(1, rows, 1, c), gamma.v, beta.v, stat, act=ops.ACT_RELU)
            accumulate(x, dx.view(rows, c), own=True)
            accumulate(gamma, dg, own=True)
            accumulate(beta, db, own=True)

        return t.new(y.view(rows, c), bw)
    raise NotImplementedError("batchnorm_rows: frozen-statistics BatchNorm1d inside a training step is not used by the reference")


def gather_keypoints(t: Tape, xn: Node, cm: torch.Tensor, pos: torch.Tensor, b: int, h: int, w: int, c: int, k: int):
    """top-k local maxima rows per azimuth column (set_transformer.py:134-147): -> (kp node (b*k*w, c), kpos (b,k,w,2), top (b,k,w));
    the row choice is piecewise constant, gradients flow through the gathered rows only"""
    dev = xn.v.device
    top = torch.empty((b, k, w), dtype=torch.int32, device=dev)
    kp = torch.empty((b * k * w, c), dtype=torch.float32, device=dev)
    kpos = torch.empty((b, k, w, 2), dtype=torch.float32, device=dev)
    hip.call("pn_setblock_keypoints", cm.data_ptr(), xn.v.data_ptr(), pos.data_ptr(), b, h, w, c, k, 0, 0, top.data_ptr(), kp.data_ptr(),
             kpos.data_ptr(), hip.stream())

    def bw(dy):
        dx = torch.zeros_like(xn.v)
        hip.call("pn_scatter_rows_f32", dy.data_ptr(), top.data_ptr(), b, k, h, w, c, dx.data_ptr(), hip.stream())
        accumulate(xn, dx, own=True)

    return t.new(kp, bw), kpos, top


def l2_normalize(t: Tape, x: Node, eps=1e-12) -> Node:
    rows, c = x.v.shape
    y = torch.empty_like(x.v)
    inv = torch.empty(rows, dtype=torch.float32, device=x.v.device)
    hip.call("pn_l2_normalize_f32", x.v.data_ptr(), rows, c, float(eps), y.data_ptr(), inv.data_ptr(), hip.stream())

    def bw(dy):
        dx = torch.empty_like(dy)
        hip.call("pn_l2_normalize_bwd_f32", y.data_ptr(), dy.data_ptr(), inv.data_ptr(), rows, c, dx.data_ptr(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


# ------------------------------------------------------------------------------------------------ dense NHWC maps
def _pad4(dy: torch.Tensor) -> torch.Tensor:
    """zero pad channels up to a multiple of 4 (the MFMA loader of the data-gradient GEMM reads 4 at a time)"""
    c = dy.shape[-1]
    if c % 4 == 0:
        return dy
    out = torch.zeros(dy.shape[:-1] + ((c + 3) // 4 * 4,), dtype=torch.float32, device=dy.device)
    out[..., :c].copy_(dy)
    return out




def conv2d(t: Tape, x: Node, w: Node, b: Optional[Node], stride=1, pad=0, relu=False) -> Node:
    """Conv2d(+bias)(+ReLU) on an NHWC map; w (Cout, Cin, k, k).  x may carry zero pad channels past Cin."""
    wv = w.v
    cout, cin, k, _ = wv.shape
    layer = ops.ConvLayer(wv, stride=stride, pad=pad, shift=None if b is None else b.v, act=ops.ACT_RELU if relu else ops.ACT_NONE, wino4=R.train_wino4_max_pixels > 0)
    layer.wino4_max_pixels = R.train_wino4_max_pixels       # F(4, 3) forward on the smaller maps only, as in train.py
    ct = x.v.shape[3]
    if ct != cin:
        layer.pad_input_channels(ct)
    y = layer(x.v)

    def bw(dy):
        if relu:
            dy = ops.relu_bwd(y, dy)
        dy4 = _pad4(dy)
        def weight_grads():
            dw = ops.conv_wgrad(x.v, dy4, k, k, stride, pad, cin=cin, cout=cout)
            accumulate(w, dw, own=True)
            if b is not None:
                accumulate(b, ops.channel_sum(dy4, c=cout), own=True)

        side_run(weight_grads, dy4, x.v)
        if x.needs_grad:
            dx = ops.ConvDgrad(wv, stride, pad)(dy4)
            if ct != cin:
                full = torch.zeros_like(x.v)
                full[..., :cin].copy_(dx)
                dx = full
            accumulate(x, dx, own=True)

    return t.new(y, bw)


def batchnorm2d(t: Tape, x: Node, bn: torch.nn.Module, gamma: Node, beta: Node, relu=True) -> Node:
    """training-mode BatchNorm2d (+ReLU) on an NHWC map (batch statistics; running statistics updated)"""
    act = ops.ACT_RELU if relu else ops.ACT_NONE
    y, stat = ops.batchnorm_train(x.v, gamma.v, beta.v, bn.eps, bn.momentum, bn.running_mean, bn.running_var, act=act)

    def bw(dy):
        dx, dg, db = ops.batchnorm_bwd(x.v, dy, gamma.v, beta.v, stat, act=act)
        accumulate(x, dx, own=True)
        accumulate(gamma, dg, own=True)
        accumulate(beta, db, own=True)

    return t.new(y, bw)


def concat_channels(t: Tape, parts: Sequence[Node], width: int) -> Node:
    """channel concatenation of NHWC maps into a ``width``-channel map (zero filled past the parts)"""
    shp = parts[0].v.shape[:-1]
    y = torch.zeros(shp + (width,), dtype=torch.float32, device=parts[0].v.device)
    offs, o = [], 0
    for p in parts:
        c = p.v.shape[-1]
        y[..., o:o + c].copy_(p.v)
        offs.append((o, c))
        o += c

    def bw(dy):
        for p, (o, c) in zip(parts, offs):
            accumulate(p, dy[..., o:o + c].contiguous(), own=True)

    return t.new(y, bw)


def pad_roll(t: Tape, x: Node, b: int, h: int, w: int, hp: int, wp: int, c: int, shift: int) -> Node:
    """(b*h*w, c) map -> zero-padded to hp x wp and rolled by (-shift, -shift): (b*hp*wp, c)"""
    y = torch.empty((b * hp * wp, c), dtype=torch.float32, device=x.v.device)
    hip.call("pn_pad_roll_f32", x.v.data_ptr(), b, h, w, hp, wp, c, shift, y.data_ptr(), hip.stream())

    def bw(dy):
        dx = torch.empty_like(x.v)
        hip.call("pn_crop_roll_f32", dy.data_ptr(), b, h, w, hp, wp, c, shift, dx.data_ptr(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def pad_roll_raw(x: torch.Tensor, b: int, h: int, w: int, hp: int, wp: int, c: int, shift: int) -> torch.Tensor:
    y = torch.empty((b * hp * wp, c), dtype=torch.float32, device=x.device)
    hip.call("pn_pad_roll_f32", x.data_ptr(), b, h, w, hp, wp, c, shift, y.data_ptr(), hip.stream())
    return y


def crop_roll(t: Tape, y: Node, b: int, h: int, w: int, hp: int, wp: int, c: int, shift: int) -> Node:
    """inverse plumbing: roll back by (+shift, +shift) and crop to h x w"""
    x = torch.empty((b * h * w, c), dtype=torch.float32, device=y.v.device)
    hip.call("pn_crop_roll_f32", y.v.data_ptr(), b, h, w, hp, wp, c, shift, x.data_ptr(), hip.stream())

    def bw(dx):
        dy = torch.empty_like(y.v)
        hip.call("pn_pad_roll_f32", dx.data_ptr(), b, h, w, hp, wp, c, shift, dy.data_ptr(), hip.stream())
        accumulate(y, dy, own=True)

    return t.new(x, bw)


def scale_channels(t: Tape, x: Node, s: Node) -> Node:
    """y[r, c] = x[r, c] * s[c]"""
    c = s.v.numel()
    y = torch.empty_like(x.v)
    hip.call("pn_scale_channels_f32", x.v.data_ptr(), s.v.data_ptr(), x.v.numel(), c, y.data_ptr(), hip.stream())

    def bw(dy):
        if x.needs_grad:
            dx = torch.empty_like(dy)
            hip.call("pn_scale_channels_f32", dy.data_ptr(), s.v.data_ptr(), dy.numel(), c, dx.data_ptr(), hip.stream())
            accumulate(x, dx, own=True)
        prod = torch.empty_like(dy)
        hip.call("pn_mul_f32", dy.data_ptr(), x.v.data_ptr(), prod.data_ptr(), dy.numel(), hip.stream())
        accumulate(s, ops.channel_sum(prod.view(1, -1, 1, c)).view(s.v.shape), own=True)

    return t.new(y, bw)


def recip_clamp(t: Tape, x: Node, lo: float) -> Node:
    """y = 1 / max(x, lo) on a small parameter vector"""
    n = x.v.numel()
    y = torch.empty_like(x.v)
    hip.call("pn_recip_clamp_f32", x.v.data_ptr(), float(lo), n, y.data_ptr(), hip.stream())

    def bw(dy):
        dx = torch.empty_like(x.v)
        hip.call("pn_recip_clamp_bwd_f32", x.v.data_ptr(), dy.contiguous().data_ptr(), float(lo), n, dx.data_ptr(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def transpose_hw(t: Tape, x: Node, b: int, h: int, w: int, c: int) -> Node:
    """(b, h, w, c) -> (b, w, h, c) (the reference's permute between the azimuth-major BEV map and the range-major token order,
    voxelnet.py:210-221)"""
    y = torch.empty((b, w, h, c), dtype=torch.float32, device=x.v.device)
    hip.call("pn_transpose_hw_f32", x.v.data_ptr(), b, h, w, c, y.data_ptr(), hip.stream())

    def bw(dy):
        dx = torch.empty(x.v.shape, dtype=torch.float32, device=dy.device)
        hip.call("pn_transpose_hw_f32", dy.data_ptr(), b, w, h, c, dx.data_ptr(), hip.stream())
        accumulate(x, dx, own=True)

    return t.new(y, bw)


def conv_transpose2d(t: Tape, x: Node, w: Node) -> Node:
    """ConvTranspose2d(kernel = stride, no bias) of the RPN deblocks (rpn.py:80-110): weight (Cin, Cout, k, k), k in {1, 2}"""
    wv = w.v
    cin, cout, k, _ = wv.shape
    if k == 1:   # a 1x1 convolution with the transposed weight
        wt = wv.permute(1, 0, 2, 3).contiguous()
        y = ops.ConvLayer(wt, stride=1, pad=0)(x.v)

        def bw1(dy):
            side_run(lambda: accumulate(w, ops.conv_wgrad(x.v, dy, 1, 1, 1, 0).permute(1, 0, 2, 3).contiguous(), own=True), dy, x.v)   # (Cout, Cin, 1, 1) -> weight layout
            if x.needs_grad:
                accumulate(x, ops.ConvDgrad(wt, 1, 0)(dy), own=True)

        return t.new(y, bw1)
    if k != 2:
        raise hip.PartnerHipError("conv_transpose2d: only kernel = stride in {1, 2} has a HIP kernel")
    y = ops.ConvLayer(wv, deconv2x2=True)(x.v)

    def bw2(dy):
        # the transposed convolution's weight gradient is the gradient of the stride-2 convolution dy -> x with the same tensor
        side_run(lambda: accumulate(w, ops.conv_wgrad(dy, x.v, 2, 2, 2, 0), own=True), dy, x.v)                          # (Cin, Cout, 2, 2)
        if x.needs_grad:
            accumulate(x, ops.ConvLayer(wv, stride=2, pad=0)(dy), own=True)

    return t.new(y, bw2)
