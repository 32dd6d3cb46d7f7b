"""RPN: the dense 2-D BEV backbone (det3d/models/necks/rpn.py:23-159), executed as a chain of
fused conv + folded-BatchNorm + ReLU launches of the fp32-MFMA implicit-GEMM kernel.  Every
deblock writes straight into its channel slice of the concatenated output (no torch.cat)."""
from __future__ import annotations

import logging

import numpy as np
import torch
from torch import nn

from . import hip, ops
from .builder import NECKS
from .nn_utils import PlanCache, Sequential, build_norm_layer, eval_only


@NECKS.register_module
class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        self._layer_strides = ds_layer_strides
        self._num_filters = ds_num_filters
        self._layer_nums = layer_nums
        self._upsample_strides = us_layer_strides
        self._num_upsample_filters = us_num_filters
        self._num_input_features = num_input_features
        self._norm_cfg = norm_cfg or dict(type="BN", eps=1e-3, momentum=0.01)
        assert len(ds_layer_strides) == len(layer_nums) == len(ds_num_filters)
        assert len(us_num_filters) == len(us_layer_strides)
        self._upsample_start_idx = len(layer_nums) - len(us_layer_strides)
        ratios = [us_layer_strides[i] / np.prod(ds_layer_strides[: i + self._upsample_start_idx + 1])
                  for i in range(len(us_layer_strides))]
        assert all(r == ratios[0] for r in ratios), "all deblocks must end at the same resolution"

        cin = [num_input_features, *ds_num_filters[:-1]]
        blocks, deblocks = [], []
        for i, n in enumerate(layer_nums):
            blk, _ = self._make_layer(cin[i], ds_num_filters[i], n, stride=ds_layer_strides[i])
            blocks.append(blk)
            j = i - self._upsample_start_idx
            if j >= 0:
                us, cout = us_layer_strides[j], us_num_filters[j]
                if us > 1:
                    up = nn.ConvTranspose2d(ds_num_filters[i], cout, int(us), stride=int(us), bias=False)
                else:
                    k = int(np.round(1 / us))
                    up = nn.Conv2d(ds_num_filters[i], cout, k, stride=k, bias=False)
                deblocks.append(Sequential(up, build_norm_layer(self._norm_cfg, cout)[1], nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        self._plan = PlanCache()
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        """rpn.py:124-142: ZeroPad2d(1) + Conv3x3(stride) + norm + ReLU, then num_blocks x (Conv3x3(pad 1) + norm + ReLU)"""
        blk = Sequential(nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False),
                         build_norm_layer(self._norm_cfg, planes)[1], nn.ReLU())
        for _ in range(num_blocks):
            blk.add(nn.Conv2d(planes, planes, 3, padding=1, bias=False))
            blk.add(build_norm_layer(self._norm_cfg, planes)[1])
            blk.add(nn.ReLU())
        return blk, planes

    @property
    def downsample_factor(self):
        f = np.prod(self._layer_strides)
        if len(self._upsample_strides) > 0:
            f /= self._upsample_strides[-1]
        return f

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    # ---------------------------------------------------------------------------------------
    @staticmethod
    def _fused(conv, bn, stride, pad, deconv=False, dtype="f32"):
        scale, shift = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, conv.bias)
        return ops.ConvLayer(conv.weight, stride=stride, pad=pad, scale=scale, shift=shift, act=ops.ACT_RELU,
                             deconv2x2=deconv, dtype=dtype)

    def set_compute_dtype(self, dtype: str) -> "RPN":
        """"f32" (default, the reference's arithmetic) or "bf16": bf16 activations / weights with f32 accumulation on
        the bf16 MFMA (BASELINE configs[3]); the concatenated output stays f32 for the head."""
        assert dtype in ("f32", "bf16")
        self.compute_dtype = dtype
        return self

    def _build_plan(self, dtype="f32"):
        plan = dict(blocks=[], deblocks=[], pillar0=None)
        if dtype == "f32" and len(self.blocks):
            c0, b0 = list(self.blocks[0]._modules.values())[1:3]
            if ops.PillarConvLayer.supports(c0.weight, c0.stride[0], c0.groups):
                scale, shift = ops.fold_bn(b0.weight, b0.bias, b0.running_mean, b0.running_var, b0.eps, c0.bias)
                plan["pillar0"] = ops.PillarConvLayer(c0.weight, c0.stride[0], scale=scale, shift=shift, act=ops.ACT_RELU)
        for blk in self.blocks:
            mods = list(blk._modules.values())
            layers = [self._fused(mods[1], mods[2], mods[1].stride[0], 1, dtype=dtype)]  # ZeroPad2d(1) + conv == pad 1
            for k in range(4, len(mods), 3):
                layers.append(self._fused(mods[k], mods[k + 1], 1, 1, dtype=dtype))
            plan["blocks"].append(layers)
        for de in self.deblocks:
            up, bn = de[0], de[1]
            if isinstance(up, nn.ConvTranspose2d):
                if up.stride[0] == 1 and up.kernel_size[0] == 1:  # ConvTranspose2d(k=1, s=1) == 1x1 convolution with the transposed weight
                    w1 = up.weight.detach().permute(1, 0, 2, 3).contiguous()
                    scale, shift = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, up.bias)
                    plan["deblocks"].append(ops.ConvLayer(w1, stride=1, pad=0, scale=scale, shift=shift, act=ops.ACT_RELU, dtype=dtype))
                    continue
                if up.stride[0] != 2:
                    raise NotImplementedError("RPN deblock: only ConvTranspose2d(k=2, s=2) / (k=1, s=1) have a HIP kernel")
                plan["deblocks"].append(self._fused(up, bn, 1, 0, deconv=True, dtype=dtype))
            else:
                plan["deblocks"].append(self._fused(up, bn, up.stride[0], 0, dtype=dtype))
        return plan

    def forward_nhwc(self, x: torch.Tensor, return_blocks=False, pillars=None):
        """x: NHWC (B,H,W,C) f32 -> NHWC (B,H',W',sum(us_filters)) f32.  ``pillars``: the frame's ops.VoxelIndex when x is the
        pillar canvas (its cells are the only non-zero pixels): the first convolution then multiplies (pillar, tap) pairs only"""
        eval_only(self, "RPN")
        dtype = getattr(self, "compute_dtype", "f32")
        if dtype == "bf16":
            if not hasattr(self, "_plan_bf16"):
                self._plan_bf16 = PlanCache()
            plan = self._plan_bf16.get(self, lambda: self._build_plan("bf16"))
            x = ops.to_bf16(x)
        else:
            plan = self._plan.get(self, self._build_plan)
        out, off, block_outs = None, 0, []
        self.canvas_read_by_pillars_only = False      # set below when the first layer took the row-band pillar form (it reads the frame's cells only)
        for i, layers in enumerate(plan["blocks"]):
            for k, layer in enumerate(layers):
                p0 = plan.get("pillar0") if (i == 0 and k == 0 and pillars is not None and dtype == "f32") else None
                if p0 is not None and p0.worth_it(pillars, x.shape[0], x.shape[1], x.shape[2]):
                    b0, oh, ow = x.shape[0], (x.shape[1] - 1) // p0.stride + 1, (x.shape[2] - 1) // p0.stride + 1
                    rest = layers[1:]
                    self.canvas_read_by_pillars_only = p0.rows_form(pillars, b0, x.shape[1], x.shape[2])
                    if (rest and all(l.stride == 1 for l in rest) and p0.planes_supported(b0, x.shape[1], x.shape[2])
                            and ops.conv_chain_orientation(rest, b0, oh, ow) is False):
                        # the pillar layer's tap reduction writes the chain's planes itself: its 33 MB map never exists in NHWC
                        canvas = x
                        x = ops.conv_chain(rest, None, shape=(b0, oh, ow), device=x.device, planes_from=lambda buf: p0(canvas, pillars, planes=buf))
                        break
                    x = p0(x, pillars)
                elif (k == 0 and dtype == "f32" and layer.stride > 1 and len(layers) > 1 and all(l.stride == 1 for l in layers[1:])
                      and layer.planes_desc(*x.shape) is not None
                      and ops.conv_chain_orientation(layers[1:], x.shape[0], *layer.out_hw(x.shape[1], x.shape[2])) is False):
                    # r6: the block's stride-2 layer writes the chain's planes from its own epilogue (csrc/conv_mfma.hip): no NHWC map, no
                    # NHWC -> planes pass between it and the block's stride-1 layers
                    src, b0 = x, x.shape[0]
                    oh, ow = layer.out_hw(x.shape[1], x.shape[2])
                    x = ops.conv_chain(layers[1:], None, shape=(b0, oh, ow), device=x.device, planes_from=lambda buf, l=layer, s=src: l.to_planes(s, buf))
                    break
                elif k <= 1 and dtype == "f32" and layer.stride == 1 and ops.conv_chain_supported(layers[k:], x.shape[0], x.shape[1], x.shape[2]):
                    # the block's same-shape layers stay in the F(4, 3) domain (conv_wchain.hip); a stride-1 first layer (Waymo block 0) joins them
                    x = ops.conv_chain(layers[k:], x)
                    break
                else:
                    x = layer(x)
            block_outs.append(x)
            j = i - self._upsample_start_idx
            if j >= 0:
                de = plan["deblocks"][j]
                if out is None:
                    oh, ow = de.out_hw(x.shape[1], x.shape[2])
                    odt = torch.bfloat16 if (dtype == "bf16" and getattr(self, "bf16_output", False) and not return_blocks) else torch.float32
                    out = torch.empty((x.shape[0], oh, ow, sum(self._num_upsample_filters)), dtype=odt, device=x.device)
                de(x, out=out, out_channel_offset=off)
                off += self._num_upsample_filters[j]
        res = out if out is not None else x
        return (res, block_outs) if return_blocks else res

    def forward(self, x):
        """logical (B,C,H,W) in / out, as rpn.py:150-159"""
        eval_only(self, "RPN")
        hip.require_device(x)
        return ops.as_nchw(self.forward_nhwc(ops.to_nhwc(x)))
