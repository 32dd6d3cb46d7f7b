"""RPN variants with context padding for sector streaming, registered under the reference's names.

Reference: det3d/models/necks/rpn_context.py:10-44 (ConvContext), 46-95 (RPNTECP: trailing-edge padding), 98-158 (ConvBDCP),
160-215 (RPNBDCP: bidirectional padding).  A sweep is processed as azimuth sectors; a 3x3 convolution at a sector border needs rows
of the neighbouring sector: the trailing edge comes from the sector processed just before (``prev_context``), the leading edge --
bidirectional padding only -- from the PREVIOUS sweep's features (``prev_sweep``) or, with every sector of a sweep stacked in
the batch (``mode='feature_only'``), from the neighbouring entries of the stack.  With ``nsectors == 1`` the bidirectional neck
is the plain RPN with CIRCULAR padding along the azimuth.

Execution: the row pieces are gathered into one padded NHWC map by ``pn_assemble_rows_f32`` (whole-row copies: the azimuth axis
is the H axis of the NHWC maps) and the convolution runs on the MFMA kernel with padding 0 along the azimuth and 1 along the
range.  ``mode='feature_only'`` with several sectors reproduces the reference's indexing literally (rpn_context.py:117-124),
including the fact that its entry 0 is built from sector 1.
"""
from __future__ import annotations

import torch
from torch import nn

from . import hip, ops
from .builder import NECKS
from .necks import RPN
from .nn_utils import Sequential, build_norm_layer, eval_only


class ConvContext(nn.Module):
    """parameters of one context-padded convolution (rpn_context.py:10-29): ``block`` = Conv2d(no padding) + norm + ReLU"""

    def __init__(self, inplanes, outplanes, kernel, stride, padding, bias, norm_cfg):
        super().__init__()
        self.block = Sequential(nn.Conv2d(inplanes, outplanes, kernel, stride=stride, bias=bias), build_norm_layer(norm_cfg, outplanes)[1], nn.ReLU())
        self.padding = padding


class ConvBDCP(ConvContext):
    def __init__(self, inplanes, outplanes, kernel, stride, padding, bias, norm_cfg, nsectors):
        super().__init__(inplanes, outplanes, kernel, stride, padding, bias, norm_cfg)
        self.nsectors = nsectors


@NECKS.register_module
class RPNTECP(RPN):
    """RPN with trailing-edge padding (rpn_context.py:46-95)"""

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        blk = Sequential(ConvContext(inplanes, planes, 3, stride, 1, False, self._norm_cfg))
        for _ in range(num_blocks):
            blk.add(ConvContext(planes, planes, 3, 1, 1, False, self._norm_cfg))
        return blk, planes

    # ---------------------------------------------------------------------------------------
    def _build_plan(self, dtype="f32"):
        if dtype != "f32":
            raise NotImplementedError("the context-padding necks run in f32")
        plan = super_plan = dict(blocks=[], deblocks=[])
        for blk in self.blocks:
            layers = []
            for cc in blk._modules.values():
                conv, bn = cc.block[0], cc.block[1]
                scale, shift = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv.bias)
                mk = lambda pad: ops.ConvLayer(conv.weight, stride=conv.stride[0], pad=pad, scale=scale, shift=shift, act=ops.ACT_RELU)  # noqa: E731
                layers.append(dict(zero=mk((1, 1)), ctx=mk((0, 1)), padding=cc.padding, mod=cc))
            plan["blocks"].append(layers)
        # deblocks as in the plain RPN
        base = RPN._build_plan(_DeblockOnly(self), "f32")
        plan["deblocks"] = base["deblocks"]
        return super_plan

    def _deblocks(self, plan, i, x, out, off):
        j = i - self._upsample_start_idx
        if j < 0:
            return out, off
        de = plan["deblocks"][j]
        if out is None:
            oh, ow = de.out_hw(x.shape[1], x.shape[2])
            out = torch.empty((x.shape[0], oh, ow, sum(self._num_upsample_filters)), dtype=torch.float32, device=x.device)
        de(x, out=out, out_channel_offset=off)
        return out, off + self._num_upsample_filters[j]

    def _conv(self, layer, x, top, bottom):
        """x NHWC; top / bottom: None (zero rows) or a piece list per sample [(tensor, sample, row0, rows)]"""
        p = layer["padding"]
        if top is None and bottom is None:
            return layer["zero"](x)
        b, h, w, c = x.shape
        samples = []
        for k in range(b):
            samples.append([(None, p) if top is None else top[k], (x, k, 0, h), (None, p) if bottom is None else bottom[k]])
        return layer["ctx"](ops.assemble_rows(samples, w, c))

    def forward_nhwc(self, x, prev_context=(), sec_id=0):
        """x NHWC (B, az, r, C); prev_context: the list this method returned for the sector before (NHWC row blocks).
        -> (NHWC output, cur_context)"""
        eval_only(self, type(self).__name__)
        plan = self._plan.get(self, self._build_plan)
        prev = list(prev_context)
        cur, out, off = [], None, 0
        for i, layers in enumerate(plan["blocks"]):
            for layer in layers:
                p = layer["padding"]
                cur.append(x[:, x.shape[1] - p:].contiguous())          # the trailing rows of this layer's input
                if prev:
                    ctx = prev.pop(0)
                    x = self._conv(layer, x, [(ctx, k, 0, p) for k in range(x.shape[0])], None)
                else:
                    x = self._conv(layer, x, None, None)
            out, off = self._deblocks(plan, i, x, out, off)
        return (out if out is not None else x), cur

    def forward(self, x, prev_context=[], sec_id=0):
        """logical (B, C, az, r) in / out and context, as rpn_context.py:75-95"""
        hip.require_device(x)
        y, cur = self.forward_nhwc(ops.to_nhwc(x), [ops.to_nhwc(c) for c in prev_context], sec_id)
        return ops.as_nchw(y), [ops.as_nchw(c) for c in cur]


class _DeblockOnly:
    """view of a neck exposing only what RPN._build_plan needs for the deblocks"""

    def __init__(self, neck):
        self.blocks, self.deblocks, self._fused = [], neck.deblocks, RPN._fused


@NECKS.register_module
class RPNBDCP(RPNTECP):
    """RPN with bidirectional padding (rpn_context.py:160-215)"""

    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters, num_input_features, norm_cfg=None,
                 name="rpn", logger=None, **kwargs):
        self.nsectors = kwargs.get("nsectors", 1)
        super().__init__(layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters, num_input_features, norm_cfg, name,
                         logger, **kwargs)

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        blk = Sequential(ConvBDCP(inplanes, planes, 3, stride, 1, False, self._norm_cfg, self.nsectors))
        for _ in range(num_blocks):
            blk.add(ConvBDCP(planes, planes, 3, 1, 1, False, self._norm_cfg, self.nsectors))
        return blk, planes

    def _pad_feature_only(self, layer, x, nsectors):
        p = layer["padding"]
        n, h, w, c = x.shape
        if nsectors == 1:   # circular along the azimuth
            return layer["ctx"](ops.assemble_rows([[(x, k, h - p, p), (x, k, 0, h), (x, k, 0, p)] for k in range(n)], w, c))
        # stacked sectors, index = sector * B + b (rpn_context.py:117-124, restated literally):
        #   e_j = [tail(s_j); s_{j+1}] (j < S-1),  e_{S-1} = [s_{S-1}; zeros];  out_0 = [zeros; e_0],  out_k = [e_{k-1}; head_p(e_k)]
        bsz = n // nsectors
        samples = []
        for s in range(nsectors):
            for b in range(bsz):
                at = lambda sec: sec * bsz + b  # noqa: E731
                if s == 0:
                    samples.append([(None, p), (x, at(0), h - p, p), (x, at(1), 0, h)])
                else:
                    head = (x, at(s), h - p, p) if s < nsectors - 1 else (x, at(s), 0, p)
                    # e_{s-1} = [tail(s_{s-1}); s_s]
                    samples.append([(x, at(s - 1), h - p, p), (x, at(s), 0, h), head])
        return layer["ctx"](ops.assemble_rows(samples, w, c))

    def _pad_streaming(self, layer, x, prev_sweep, prev, sec_id):
        p = layer["padding"]
        n, h, w, c = x.shape
        full = prev_sweep.shape[1]
        nsec = full // h
        if nsec == 1:
            return layer["ctx"](ops.assemble_rows([[(x, k, h - p, p), (x, k, 0, h), (x, k, 0, p)] for k in range(n)], w, c)), prev
        lead = lambda k: (prev_sweep, k, (sec_id + 1) * h, p)  # noqa: E731
        if sec_id == 0:
            top = (lambda k: (prev_sweep, k, full - p, p)) if layer["mod"].nsectors == nsec else (lambda k: (None, p))
            samples = [[top(k), (x, k, 0, h), lead(k)] for k in range(n)]
        elif sec_id == nsec - 1:
            ctx = prev.pop(0)
            bottom = (lambda k: (prev_sweep, k, 0, p)) if layer["mod"].nsectors == nsec else (lambda k: (None, p))
            samples = [[(ctx, k, ctx.shape[1] - p, p), (x, k, 0, h), bottom(k)] for k in range(n)]
        else:
            ctx = prev.pop(0)
            samples = [[(ctx, k, ctx.shape[1] - p, p), (x, k, 0, h), lead(k)] for k in range(n)]
        return layer["ctx"](ops.assemble_rows(samples, w, c)), prev

    def forward_nhwc(self, x, prev_sweep=(), prev_context=(), sec_id=0, nsectors=1, mode="feature_only"):
        eval_only(self, "RPNBDCP")
        plan = self._plan.get(self, self._build_plan)
        prev = list(prev_context)
        cur, out, off, layer_id = [], None, 0, 0
        for i, layers in enumerate(plan["blocks"]):
            for layer in layers:
                cur.append(x)                                        # the whole input of the layer (rpn_context.py:114)
                if mode == "feature_only":
                    x = self._pad_feature_only(layer, x, nsectors)
                else:
                    ps = prev_sweep[layer_id]
                    layer_id = (layer_id + 1) % len(prev_sweep)
                    x, prev = self._pad_streaming(layer, x, ps, prev, sec_id)
            out, off = self._deblocks(plan, i, x, out, off)
        return (out if out is not None else x), cur

    def forward(self, x, prev_sweep=[], prev_context=[], sec_id=0, nsectors=1, mode="feature_only"):
        hip.require_device(x)
        y, cur = self.forward_nhwc(ops.to_nhwc(x), [ops.to_nhwc(t) for t in prev_sweep], [ops.to_nhwc(t) for t in prev_context], sec_id, nsectors, mode)
        return ops.as_nchw(y), [ops.as_nchw(c) for c in cur]
