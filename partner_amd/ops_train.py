"""Training-side operators over the C ABI (T1 / L1 of SURVEY 8a): weight and data gradients, BatchNorm / GroupNorm-family backward, the
centre loss, the PFN backward, gradient norm + fused clip / weight decay / Adam."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple  # noqa: F401

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc  # noqa: F401
from .ops_common import *  # noqa: F401,F403
from .ops_common import _f32, _workspace  # noqa: F401
from .routes import R, S  # noqa: F401
from .ops_conv import ConvLayer, ConvJob  # noqa: F401
from .ops_index import CenterLossTargets, VoxelIndex, pfn_center_table  # noqa: F401

# ------------------------------------------------------------------------------ conv backward (T1)
def conv_wgrad(x: torch.Tensor, dout: torch.Tensor, kh: int, kw: int, stride=1, pad=0, cin: Optional[int] = None,
               in_channel_offset=0, cout: Optional[int] = None, dout_channel_offset=0, out: Optional[torch.Tensor] = None,
               accumulate=False, range_strata=0) -> torch.Tensor:
    """dW (Cout, Cin, KH, KW) of a convolution x -> y given dout = dL/dy; NHWC maps.  ``range_strata`` > 1: the RangeStratified
    convolution (one weight set per band of W / strata columns) -> dW (strata * Cout, Cin, KH, KW)."""
    hip.require_device(x, dout)
    lib = hip.load()
    assert x.dim() == 4 and dout.dim() == 4 and x.is_contiguous() and dout.is_contiguous()
    b, h, w, ct = x.shape
    cin = ct - in_channel_offset if cin is None else cin
    cout = dout.shape[3] - dout_channel_offset if cout is None else cout
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    d = ConvDesc(b, h, w, cin, cout, 1, kh, kw, stride, ph, pw, ct, in_channel_offset, dout.shape[3], dout_channel_offset, 0, 0, int(range_strata))
    oh, ow = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
    assert dout.shape[:3] == (b, oh, ow), (dout.shape, (b, oh, ow))
    if out is None:
        out = torch.empty((max(1, int(range_strata)) * cout, cin, kh, kw), dtype=torch.float32, device=x.device)
    if (range_strata <= 1 and R.conv_wgrad_wino4 and (kh, kw, stride, ph, pw) == (3, 3, 1, 1, 1) and w % 4 == 0 and cin % 4 == 0 and cout % 4 == 0 and in_channel_offset % 4 == 0
            and dout_channel_offset % 4 == 0 and ct % 4 == 0 and dout.shape[3] % 4 == 0 and b * h * (w // 4) >= R.conv_wgrad_wino4_min_quads
            and cin * cout >= 0.75 * (-(-cin // 128) * 128) * (-(-cout // 128) * 128)):      # its 128 x 128 (ci, co) tiles mostly full
        # F(4, 3) weight gradient (conv_wgrad_wino4.hip): half the MFMA work on the maps large enough to fill the chip with its slices
        nbytes = lib.pn_conv2d_wgrad_wino4_workspace_bytes(C.byref(d))
        ws = _workspace(nbytes, x.device)
        hip.call("pn_conv2d_wgrad_wino4_f32", C.byref(d), x.data_ptr(), dout.data_ptr(), out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
        return out
    nbytes = lib.pn_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = _workspace(nbytes, x.device)
    hip.call("pn_conv2d_wgrad_f32", C.byref(d), x.data_ptr(), dout.data_ptr(), out.data_ptr(), int(accumulate), ws.data_ptr(),
             nbytes, hip.stream())
    return out






def channel_sum(x: torch.Tensor, c: Optional[int] = None, channel_offset=0, out: Optional[torch.Tensor] = None,
                accumulate=False) -> torch.Tensor:
    """sum over all pixels of an NHWC map, per channel (bias gradient)"""
    hip.require_device(x)
    lib = hip.load()
    ct = x.shape[-1]
    c = ct - channel_offset if c is None else c
    pixels = x.numel() // ct
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib.pn_channel_sum_workspace_bytes(c)
    ws = _workspace(nbytes, x.device)
    hip.call("pn_channel_sum_f32", x.data_ptr(), pixels, ct, channel_offset, c, out.data_ptr(), int(accumulate), ws.data_ptr(),
             nbytes, hip.stream())
    return out


class ConvDgrad:
    """Data gradient of ``Conv2d(weight, stride, pad)`` as a convolution of dout on the MFMA kernel.

    Supported geometries (all the reference's BEV path uses): stride 1 (any k, pad);
    3x3 / stride 2 / pad 1; 2x2 / stride 2 / pad 0.  ``repack(weight)`` refreshes the packed copy
    after an optimizer step without reallocating."""

    def __init__(self, weight: torch.Tensor, stride=1, pad=0):
        hip.require_device(weight)
        lib = hip.load()
        cout, cin, kh, kw = weight.shape
        self.cout, self.cin, self.kh, self.kw, self.stride, self.pad = cout, cin, kh, kw, int(stride), int(pad)
        dev = weight.device
        if self.stride == 1:
            self.kind = "s1"
            self.packed = _f32(lib.pn_conv_packed_weight_floats(cin, cout, kh, kw, 1), dev)
        elif self.stride == 2 and (kh, kw, self.pad) == (3, 3, 1):
            self.kind = "s2k3"
            self.packed = _f32(lib.pn_conv_dgrad_s2_packed_weight_floats(cout, cin), dev)
        elif self.stride == 2 and (kh, kw, self.pad) == (2, 2, 0):
            self.kind = "s2k2"
            self.packed = _f32(lib.pn_deconv2x2_packed_weight_floats(cout, cin), dev)
        else:
            raise hip.PartnerHipError(f"ConvDgrad: unsupported geometry k={kh}x{kw} stride={stride} pad={pad}")
        # the data gradient of a 3x3 / stride-1 / pad-1 convolution is itself one (taps mirrored, channels swapped): it takes the
        # width-Winograd kernel on large maps, like the forward layer (conv_wino.hip)
        self.wino_packed = self.wino4_packed = None
        if self.kind == "s1" and (kh, kw, self.pad) == (3, 3, 1) and cout % 4 == 0 and R.conv_wino:
            self.wino_packed = _f32(lib.pn_conv_wino_packed_weight_floats(cin, cout), dev)
            if R.conv_wino4 and R.conv_wino4_dgrad and cin % 32 == 0:
                self.wino4_packed = _f32(lib.pn_conv_wino4_packed_weight_floats(cin, cout), dev)
        self.repack(weight)

    def repack(self, weight: torch.Tensor, token=None) -> None:
        """lazy, as ConvLayer.repack: the layout a call takes is packed on first use"""
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        self._stale_w = weight.detach().contiguous().float()
        self._stale = {"direct"} | ({"wino"} if self.wino_packed is not None else set()) | ({"wino4"} if self.wino4_packed is not None else set())

    def prepack_used(self) -> None:
        for layout in sorted(getattr(self, "_used", ())):
            self._ensure(layout)

    def _ensure(self, layout: str) -> None:
        self.__dict__.setdefault("_used", set()).add(layout)
        if layout not in self._stale:
            return
        self._stale.discard(layout)
        w, st = self._stale_w, hip.stream()
        if layout == "direct":
            if self.kind == "s1":
                hip.call("pn_pack_conv_dgrad_weight_f32", w.data_ptr(), self.cout, self.cin, self.kh, self.kw, self.packed.data_ptr(), st)
            elif self.kind == "s2k3":
                hip.call("pn_pack_conv_dgrad_s2_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), st)
            else:
                hip.call("pn_pack_deconv2x2_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), st)
            return
        # the gradient convolution's weight is the forward one with the taps mirrored and the channels swapped: the pack kernels read it so
        # (r2 made a flipped + transposed copy first: two more launches per layer and iteration)
        if layout == "wino":
            hip.call("pn_pack_conv_dgrad_weight_wino_f32", w.data_ptr(), self.cout, self.cin, self.wino_packed.data_ptr(), st)
        else:
            hip.call("pn_pack_conv_dgrad_weight_wino4_f32", w.data_ptr(), self.cout, self.cin, self.wino4_packed.data_ptr(), st)

    def __call__(self, dout: torch.Tensor, out: Optional[torch.Tensor] = None, dout_channel_offset=0,
                 out_channel_offset=0, accumulate=False) -> torch.Tensor:
        """dout: NHWC (B,OH,OW,Ct) -> dx (B,H,W,Cin); H = OH*stride (the reference's maps are even-sized)"""
        hip.require_device(dout)
        assert dout.dim() == 4 and dout.is_contiguous()
        b, oh, ow, ct = dout.shape
        # the MFMA loader fetches 4 channels at a time: a Cout that is not a multiple of 4 needs dout
        # stored with zero-filled pad channels (packed weight rows past Cout are zero as well)
        cin_eff = (self.cout + 3) // 4 * 4
        if dout_channel_offset + cin_eff > ct:
            raise hip.PartnerHipError(f"ConvDgrad: dout needs {cin_eff - self.cout} zero pad channel(s) after its {self.cout} channels")
        if self.kind == "s1":
            h, w = oh + self.kh - 1 - 2 * self.pad, ow + self.kw - 1 - 2 * self.pad
            pd = self.kh - 1 - self.pad
            d = ConvDesc(b, oh, ow, cin_eff, self.cin, 1, self.kh, self.kw, 1, pd, self.kw - 1 - self.pad, ct, dout_channel_offset,
                         0, out_channel_offset, 0, 0, 0)
        elif self.kind == "s2k3":
            h, w = 2 * oh, 2 * ow
            d = ConvDesc(b, oh, ow, cin_eff, self.cin, 1, 2, 2, 1, 0, 0, ct, dout_channel_offset, 0, out_channel_offset, 0, 1, 0, 1, 1)
        else:
            h, w = 2 * oh, 2 * ow
            d = ConvDesc(b, oh, ow, cin_eff, self.cin, 1, 1, 1, 1, 0, 0, ct, dout_channel_offset, 0, out_channel_offset, 0, 1, 0)
        if out is None:
            out = torch.empty((b, h, w, self.cin), dtype=torch.float32, device=dout.device)
        assert out.shape[:3] == (b, h, w) and out.is_contiguous()
        d.out_pixel_stride = out.shape[3]
        d.accumulate = int(accumulate)
        if (self.wino4_packed is not None and not accumulate and ow % 4 == 0 and cin_eff == self.cout
                and ((b * oh * (ow // 4) + 31) // 32) * (self.cin // 32) >= R.conv_wino4_min_tiles):
            self._ensure("wino4")
            hip.call("pn_conv2d_wino4_nhwc_f32", C.byref(d), dout.data_ptr(), self.wino4_packed.data_ptr(), None, None, out.data_ptr(), hip.stream())
            return out
        if (self.wino_packed is not None and not accumulate and ow % 2 == 0 and cin_eff == self.cout
                and ((b * oh * (ow // 2) + 31) // 32) * ((self.cin + 63) // 64) >= R.conv_wino_min_tiles):
            self._ensure("wino")
            hip.call("pn_conv2d_wino_nhwc_f32", C.byref(d), dout.data_ptr(), self.wino_packed.data_ptr(), None, None, out.data_ptr(), hip.stream())
            return out
        self._ensure("direct")
        hip.call("pn_conv2d_nhwc_f32", C.byref(d), dout.data_ptr(), self.packed.data_ptr(), None, None, out.data_ptr(), hip.stream())
        return out


def batchnorm_train(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, momentum: float,
                    running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor], act=ACT_RELU,
                    out: Optional[torch.Tensor] = None, c: Optional[int] = None, channel_offset=0, out_channel_offset=0):
    """training-mode BatchNorm2d + activation on an NHWC map -> (out, saved_stat)"""
    hip.require_device(x)
    lib = hip.load()
    ct = x.shape[-1]
    c = ct - channel_offset if c is None else c
    pixels = x.numel() // ct
    if out is None:
        out = torch.empty(x.shape[:-1] + (c,), dtype=torch.float32, device=x.device)
    stat = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    nbytes = lib.pn_batchnorm_workspace_bytes(c)
    ws = _workspace(nbytes, x.device)
    hip.call("pn_batchnorm_train_fwd", x.data_ptr(), pixels, c, ct, channel_offset, hip.ptr(gamma), hip.ptr(beta), float(eps),
             float(momentum), int(act), hip.ptr(running_mean), hip.ptr(running_var), out.data_ptr(), out.shape[-1],
             out_channel_offset, stat.data_ptr(), ws.data_ptr(), nbytes, hip.stream())
    return out, stat


def batchnorm_bwd(x: torch.Tensor, dout: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, stat: torch.Tensor, act=ACT_RELU,
                  dx: Optional[torch.Tensor] = None, dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                  accumulate=False, c: Optional[int] = None, channel_offset=0, dout_channel_offset=0, dx_channel_offset=0):
    """backward of batchnorm_train (+ its activation) -> (dx, dgamma, dbeta); dx may be dout (in place)"""
    hip.require_device(x, dout)
    lib = hip.load()
    ct = x.shape[-1]
    c = ct - channel_offset if c is None else c
    pixels = x.numel() // ct
    if dx is None:
        dx = torch.empty(x.shape[:-1] + (c,), dtype=torch.float32, device=x.device)
    if dgamma is None:
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    nbytes = lib.pn_batchnorm_workspace_bytes(c)
    ws = _workspace(nbytes, x.device)
    hip.call("pn_batchnorm_bwd", x.data_ptr(), dout.data_ptr(), pixels, c, ct, channel_offset, dout.shape[-1], dout_channel_offset,
             hip.ptr(gamma), hip.ptr(beta), int(act), stat.data_ptr(), dx.data_ptr(), dx.shape[-1], dx_channel_offset,
             dgamma.data_ptr(), dbeta.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
    return dx, dgamma, dbeta


def groupnorm_strat_bwd(x: torch.Tensor, dout: torch.Tensor, channel_groups: int, range_strata: int, gamma: torch.Tensor,
                        beta: torch.Tensor, eps=1e-5, act=ACT_NONE, dout2: Optional[torch.Tensor] = None,
                        mul: Optional[torch.Tensor] = None, dx: Optional[torch.Tensor] = None, dgamma=None, dbeta=None,
                        dmul=None, dadd=None, accumulate=False, stat: Optional[torch.Tensor] = None):
    """backward of groupnorm_strat -> (dx, dgamma, dbeta[, dmul, dadd]); dx may be dout.  ``stat``: the forward's ``stat_out`` (else the
    statistics are recomputed from x)"""
    hip.require_device(x, dout)
    lib = hip.load()
    assert x.is_contiguous() and dout.is_contiguous()
    b, h, w, c = x.shape
    dev = x.device
    if dx is None:
        dx = torch.empty_like(x)
    if dgamma is None:
        dgamma = _f32(range_strata * c, dev)
    if dbeta is None:
        dbeta = _f32(range_strata * c, dev)
    if dout2 is not None:
        assert mul is not None and dout2.is_contiguous()
        if dmul is None:
            dmul = torch.empty((h, w, c), dtype=torch.float32, device=dev)
        if dadd is None:
            dadd = torch.empty((h, w, c), dtype=torch.float32, device=dev)
    nbytes = lib.pn_groupnorm_bwd_workspace_bytes(b, c, channel_groups, range_strata)
    ws = _workspace(nbytes, dev)
    hip.call("pn_groupnorm_strat_bwd_stat", x.data_ptr(), dout.data_ptr(), hip.ptr(dout2), hip.ptr(mul), b, h, w, c, c, 0, dout.shape[-1], 0,
             channel_groups, range_strata, hip.ptr(gamma), hip.ptr(beta), float(eps), int(act), dx.data_ptr(), dx.shape[-1], 0,
             dgamma.data_ptr(), dbeta.data_ptr(), hip.ptr(dmul), hip.ptr(dadd), int(accumulate), hip.ptr(stat), ws.data_ptr(), nbytes, hip.stream())
    if dout2 is not None:
        return dx, dgamma, dbeta, dmul, dadd
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------ L1 loss, NHWC level
_CODE_WEIGHTS: dict = {}


def _loss_common(hm: torch.Tensor, ncls: int, boxes, tg: CenterLossTargets, code_weights, with_vel: bool):
    b, h, w, _ = hm.shape
    ndim = sum(n for _, n in boxes)
    ad = tg.anno.shape[-1]
    sel = list(range(ndim)) if with_vel else [0, 1, 2, 3, 4, 5, ad - 2, ad - 1]
    # cached on the device: a fresh torch.tensor(..., device=) is a pageable host-to-device copy, which makes the host wait for the stream
    # -- twice per training iteration, in the middle of it (forward loss, backward loss): the GPU then idles while the host catches up
    key = (tuple(float(v) for v in list(code_weights)[:ndim]), str(hm.device))
    cw = _CODE_WEIGHTS.get(key)
    if cw is None:
        cw = _CODE_WEIGHTS[key] = torch.tensor(list(key[0]), dtype=torch.float32, device=hm.device)
    n = len(boxes)
    # pixel strides come from the tensors' strides, so channel-slice views of wider NHWC maps work
    args = (hm.data_ptr(), hm.stride(2), tg.hm.data_ptr(), b, ncls, h, w, (C.c_void_p * n)(*[t.data_ptr() for t, _ in boxes]),
            (C.c_int * n)(*[t.stride(2) for t, _ in boxes]), (C.c_int * n)(*[c for _, c in boxes]), n, tg.ind.data_ptr(),
            tg.mask.data_ptr(), tg.cat.data_ptr(), tg.anno.data_ptr(), ad, (C.c_int * ndim)(*sel), tg.ind.shape[1], ndim, cw.data_ptr())
    return args, ndim, cw


def center_loss(hm: torch.Tensor, ncls: int, boxes, tg: CenterLossTargets, code_weights, weight: float, with_vel=True):
    """hm: NHWC logits (B,H,W,>=ncls); boxes: [(NHWC tensor, channels)] in the reference order
    (reg, height, dim[, vel], rot).  -> out[4+ndim] = [det, hm, loc, num_pos, elem...] (device)"""
    hip.require_device(hm)
    lib = hip.load()
    args, ndim, cw = _loss_common(hm, ncls, boxes, tg, code_weights, with_vel)
    out = torch.empty((4 + ndim,), dtype=torch.float32, device=hm.device)
    wsb = lib.pn_center_loss_workspace_bytes()
    ws = _workspace(wsb, hm.device)
    hip.call("pn_center_loss_fwd", *args, float(weight), out.data_ptr(), ws.data_ptr(), wsb, hip.stream())
    return out


def center_loss_bwd(hm: torch.Tensor, ncls: int, boxes, tg: CenterLossTargets, code_weights, weight: float, fwd_out: torch.Tensor,
                    grad_scale=1.0, with_vel=True, d_hm: Optional[torch.Tensor] = None, d_boxes=None):
    """-> (d_hm, [d_box...]) NHWC, channel counts padded to multiples of 4 (pad channels zero)"""
    hip.require_device(hm)
    args, ndim, cw = _loss_common(hm, ncls, boxes, tg, code_weights, with_vel)
    b, h, w, _ = hm.shape
    pad4 = lambda c: (c + 3) // 4 * 4  # noqa: E731
    if d_hm is None:
        d_hm = torch.empty((b, h, w, pad4(ncls)), dtype=torch.float32, device=hm.device)
    if d_boxes is None:
        d_boxes = [torch.empty((b, h, w, pad4(c)), dtype=torch.float32, device=hm.device) for _, c in boxes]
    n = len(boxes)
    hip.call("pn_center_loss_bwd", *args, float(weight), fwd_out.data_ptr(), float(grad_scale), d_hm.data_ptr(), d_hm.shape[3],
             (C.c_void_p * n)(*[t.data_ptr() for t in d_boxes]), (C.c_int * n)(*[t.shape[3] for t in d_boxes]), hip.stream())
    return d_hm, d_boxes


def dynamic_pfn_bwd(points: torch.Tensor, vi: VoxelIndex, w0: torch.Tensor, w1: torch.Tensor, vx: float, vy: float,
                    x_offset: float, y_offset: float, d_features: Optional[torch.Tensor] = None,
                    d_canvas: Optional[torch.Tensor] = None, dw0: Optional[torch.Tensor] = None,
                    dw1: Optional[torch.Tensor] = None, accumulate=False, center_table: Optional[torch.Tensor] = None):
    """weight gradients of the (32,128) DynamicPFNet given d_features (V,128) or the canvas gradient (B,T,R,128)"""
    hip.require_device(points, w0, w1)
    lib = hip.load()
    dev = points.device
    if center_table is None:
        center_table = pfn_center_table(vi.spec.grid[1], vy, y_offset, dev)
    if dw0 is None:
        dw0 = torch.empty_like(w0)
    if dw1 is None:
        dw1 = torch.empty_like(w1)
    _, _, g = vi.spec.c_arrays()
    nbytes = lib.pn_dynamic_pfn_bwd_workspace_bytes()
    ws = _workspace(nbytes, dev)
    hip.call("pn_dynamic_pfn_bwd", points.data_ptr(), points.shape[1], vi.voxel_start.data_ptr(), vi.order.data_ptr(),
             vi.num_voxels.data_ptr(), vi.n_cap, vi.unq_keys_ptr, g, w0.data_ptr(), w0.shape[0], w1.data_ptr(), w1.shape[0],
             float(vx), float(vy), float(x_offset), float(y_offset), center_table.data_ptr(), hip.ptr(d_features), hip.ptr(d_canvas),
             dw0.data_ptr(), dw1.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
    return dw0, dw1


# ------------------------------------------------------------------------------ T1 small kernels
def grad_norm(flat_grads: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """L2 norm of a flat fp32 buffer -> device scalar (1,)"""
    hip.require_device(flat_grads)
    lib = hip.load()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=flat_grads.device)
    nbytes = lib.pn_grad_norm_workspace_bytes()
    ws = _workspace(nbytes, flat_grads.device)
    hip.call("pn_grad_norm_f32", flat_grads.data_ptr(), flat_grads.numel(), out.data_ptr(), ws.data_ptr(), nbytes, hip.stream())
    return out


def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int, lr: float,
              beta1: float, beta2=0.99, eps=1e-8, weight_decay=0.01, total_norm: Optional[torch.Tensor] = None, max_norm=35.0) -> None:
    """fused clip + decoupled weight decay + Adam over flat buffers (in place)"""
    hip.require_device(params, grads, exp_avg, exp_avg_sq)
    assert params.numel() == grads.numel() == exp_avg.numel() == exp_avg_sq.numel()
    hip.call("pn_adam_step_f32", params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), params.numel(), int(step),
             float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), hip.ptr(total_norm), float(max_norm), hip.stream())


def tanh_bwd(y: torch.Tensor, dy: torch.Tensor, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
    hip.require_device(y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.numel() == dy.numel()
    if dx is None:
        dx = torch.empty_like(dy)
    hip.call("pn_tanh_bwd_f32", y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), hip.stream())
    return dx


def relu_bwd(y: torch.Tensor, dy: torch.Tensor, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = dy * (y > 0), y = the ReLU output; dx may be dy"""
    hip.require_device(y, dy)
    assert y.is_contiguous() and dy.is_contiguous() and y.numel() == dy.numel()
    if dx is None:
        dx = torch.empty_like(dy)
    hip.call("pn_relu_bwd_f32", y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), hip.stream())
    return dx


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    hip.require_device(a, b)
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    if out is None:
        out = torch.empty_like(a)
    hip.call("pn_add_f32", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), hip.stream())
    return out


def strat_expand(dy: torch.Tensor, strata: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,H,W,C) -> (B,H,W,strata*C): block of the pixel's range stratum = dy, zeros elsewhere"""
    hip.require_device(dy)
    assert dy.is_contiguous()
    b, h, w, c = dy.shape
    if out is None:
        out = torch.empty((b, h, w, strata * c), dtype=torch.float32, device=dy.device)
    hip.call("pn_strat_expand_f32", dy.data_ptr(), b, h, w, c, strata, out.data_ptr(), hip.stream())
    return out


class StratConvDgrad:
    """Data gradient of the RangeStratified 3x3 convolution (weight (strata * Cout, Cin, 3, 3), center_head_parallel.py:27-59) at the
    convolution's own multiply-add count.  The weight set a tap takes follows the stratum of the dy pixel it reads, so per width tap kx
    the column convolution z_kx[y, x] = sum_ky W_s(x)[:, :, ky, kx]^T dy[y - ky + 1, x] is a STRATIFIED 3x1 convolution of dy (the forward
    kernel, 3 * Cin outputs per stratum), and dx[y, x] = z_0[y, x + 1] + z_1[y, x] + z_2[y, x - 1] (pn_strat_dgrad_combine_f32).  r3 expanded
    dy to strata * Cout channels and ran an ordinary gradient convolution over mostly zeros (8 x the work on the reference's head)."""

    def __init__(self, weight: torch.Tensor, strata: int):
        hip.require_device(weight)
        ct, cin, kh, kw = weight.shape
        assert (kh, kw) == (3, 3) and ct % strata == 0 and cin % 4 == 0
        self.strata, self.cin, self.cout = int(strata), cin, ct // strata
        # w'[s, kx * Cin + ci, co, ky', 0] = w[s * Cout + co, ci, 2 - ky', kx]: one gather of the flat weight
        idx = torch.arange(weight.numel(), device=weight.device).view(strata, self.cout, cin, 3, 3)
        self._idx = idx.flip(3).permute(0, 4, 2, 1, 3).reshape(-1).contiguous()
        self._shape = (strata * 3 * cin, self.cout, 3, 1)
        self._w = weight.detach().reshape(-1)[self._idx].view(self._shape)
        self.layer = ConvLayer(self._w, stride=1, pad=(1, 0), range_strata=self.strata)

    def repack(self, weight: torch.Tensor, token=None) -> None:
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        torch.index_select(weight.detach().reshape(-1), 0, self._idx, out=self._w.view(-1))
        self.layer.repack(self._w)

    def prepack_used(self) -> None:
        self.layer.prepack_used()

    def __call__(self, dy: torch.Tensor, out: Optional[torch.Tensor] = None, out_channel_offset=0, accumulate=False) -> torch.Tensor:
        hip.require_device(dy)
        b, h, w, c = dy.shape
        assert c == self.cout and dy.is_contiguous()
        z = self.layer(dy)
        if out is None:
            out = torch.empty((b, h, w, self.cin), dtype=torch.float32, device=dy.device)
        assert out.shape[:3] == (b, h, w) and out.is_contiguous()
        hip.call("pn_strat_dgrad_combine_f32", z.data_ptr(), b, h, w, self.cin, out.data_ptr(), out.shape[3], out_channel_offset, int(accumulate),
                 hip.stream())
        return out


def strat_channel_sum(dy: torch.Tensor, strata: int, out: torch.Tensor) -> torch.Tensor:
    """bias gradient of the RangeStratified convolution: sums of dy (B,H,W,C) over the pixels of every column band -> out (strata * C)"""
    b, h, w, c = dy.shape
    cols = channel_sum(dy.view(1, b * h, 1, w * c))       # per (column, channel) over the rows, fixed order
    torch.sum(cols.view(strata, w // strata, c), dim=1, out=out.view(strata, c))
    return out
