"""Training step of the nuScenes polar-pillar model on the HIP kernels.

Host mirror of the reference's training iteration
  Trainer.train / batch_processor_inline   det3d/torchie/trainer/trainer.py:446-501, 414-444
  parse_second_losses                      det3d/torchie/trainer/trainer.py:116-137
  DistOptimizerHook / allreduce_grads      det3d/core/utils/dist_utils.py:51-57, 31-42
  OptimizerHook.clip_grads                 det3d/torchie/trainer/hooks/optimizer.py:10-13
  OptimWrapper.step + OneCycle             det3d/solver/fastai_optim.py:155-171,
                                           det3d/solver/learning_schedules_fastai.py:77-95
without autograd: forward in training mode (batch-statistics BatchNorm), the CenterPoint loss, an
explicit backward through every layer of  DynamicPFNet -> DynamicPPScatter -> RPN ->
CenterHeadSinglePos  and one fused clip + decoupled-weight-decay + Adam update over a flat fp32
parameter buffer.  All arithmetic runs in kernels of libpartner_hip; PyTorch allocates tensors and
(for world_size > 1) all-reduces the flat gradient buffer through torch.distributed (RCCL).

The module's parameters are re-bound to views of the flat buffer, so ``state_dict()`` keeps the
reference's keys and the inference path (after ``PlanCache`` invalidation) sees the trained weights.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import hip, ops
from .routes import R
from .heads import CenterHeadSingle, CenterHeadSinglePos, RangeStratified
from .nn_utils import RSNorm
from .readers import DynamicPFNet

RELU, NONE, TANH = ops.ACT_RELU, ops.ACT_NONE, ops.ACT_TANH


def one_cycle(step: int, total_step: int, lr_max: float, moms: Sequence[float], div_factor: float, pct_start: float):
    """(lr, beta1) the OneCycle scheduler sets before optimizer step ``step`` (0-based);
    learning_schedules_fastai.py:52-95 (the last phase whose start has been reached wins)."""
    a1 = int(total_step * pct_start)
    low = lr_max / div_factor

    def cos(start, end, pct):
        return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)

    if step >= a1:
        pct = (step - a1) / (total_step - a1)
        return cos(lr_max, low / 1e4, pct), cos(moms[1], moms[0], pct)
    pct = step / a1
    return cos(low, lr_max, pct), cos(moms[0], moms[1], pct)


_SideStream = ops.SideStream


class ParamStore:
    """All trainable parameters of a module in ONE flat fp32 buffer (+ flat grad / Adam moments);
    every parameter starts on a 16-byte boundary.  The flat gradient buffer is what gets all-reduced
    (one collective per step, dist_utils.py:8-28 coalesces the same way)."""

    def __init__(self, model: nn.Module, device, order_key=None):
        """``order_key(name) -> sortable``: position of a parameter in the flat buffer (default: named_parameters order);
        the training step orders the buffer by the time a gradient is complete in backward, so that a gradient bucket is one
        contiguous range"""
        self.names: List[str] = []
        self.offsets: Dict[str, Tuple[int, torch.Size]] = {}
        off = 0
        named = list(model.named_parameters())
        if order_key is not None:
            named.sort(key=lambda kv: order_key(kv[0]))   # stable: ties keep the module order
        for name, p in named:
            self.names.append(name)
            self.offsets[name] = (off, p.shape)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=device)
        self.side = _SideStream(device)
        self.fresh = None            # token of the iteration whose packed weights were refreshed ahead (PolarPillarTrainStep._prepack)
        self.convs: List[object] = []   # the step's convolution wrappers (prepack_fwd / prepack_bwd), in construction order
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for name, p in model.named_parameters():
            o, shape = self.offsets[name]
            view = self.flat_p[o:o + p.numel()].view(shape)
            view.copy_(p.detach().to(device=device, dtype=torch.float32))
            p.data = view
            self.p[name] = view
            self.g[name] = self.flat_g[o:o + p.numel()].view(shape)


def invalidate_inference_plans(model: nn.Module) -> None:
    """the kernels update the parameters behind PyTorch's version counters: drop the packed-weight plans of the
    inference path (f32 and bf16) so that the next eval forward re-packs from the trained weights"""
    for m in model.modules():
        for attr in ("_plan", "_plan_bf16"):
            pc = getattr(m, attr, None)
            if pc is not None and hasattr(pc, "plan"):
                pc.plan = None


def optimizer_state(ps: ParamStore, it: int, sched: dict) -> Dict[str, object]:
    """Adam moments of the flat buffer + where every parameter lives in it (names, offsets, sizes): a checkpoint is independent
    of the ORDER of the buffer (the reference keeps optimizer state per parameter, det3d/torchie/trainer/trainer.py:342-372)"""
    return dict(iter=it, exp_avg=ps.flat_m.clone(), exp_avg_sq=ps.flat_v.clone(), names=list(ps.names),
                offsets=[int(ps.offsets[n][0]) for n in ps.names], numels=[int(ps.offsets[n][1].numel()) for n in ps.names],
                schedule=dict(sched))


def load_optimizer_state(ps: ParamStore, state: Dict[str, object]) -> int:
    """restore the moments BY NAME: a checkpoint written with another layout of the flat buffer (e.g. before the buffer was
    ordered by backward completion) loads as long as it holds the same parameters; returns the iteration count"""
    names = list(state["names"])
    if sorted(names) != sorted(ps.names):
        raise ValueError("optimizer state belongs to a model with different parameters")
    if names == list(ps.names) and state["exp_avg"].numel() == ps.total and "offsets" not in state:
        ps.flat_m.copy_(state["exp_avg"])       # same order, legacy checkpoint without the offset table
        ps.flat_v.copy_(state["exp_avg_sq"])
        return int(state["iter"])
    if "offsets" in state:
        offs, nums = list(state["offsets"]), list(state["numels"])
    else:   # legacy checkpoint (names only): parameters were laid out in the saved order, each on a 4-float boundary
        nums = [int(ps.offsets[n][1].numel()) for n in names]
        offs, o = [], 0
        for k in nums:
            offs.append(o)
            o += (k + 3) // 4 * 4
    m_src, v_src = state["exp_avg"], state["exp_avg_sq"]
    for n, o, k in zip(names, offs, nums):
        dst, shape = ps.offsets[n]
        if shape.numel() != k:
            raise ValueError(f"optimizer state: parameter {n} has {k} elements in the checkpoint, {shape.numel()} in the model")
        ps.flat_m[dst:dst + k].copy_(m_src[o:o + k])
        ps.flat_v[dst:dst + k].copy_(v_src[o:o + k])
    return int(state["iter"])


# F(4, 3) in the training FORWARD only on maps of at most this many pixels (the 128 x 128 and 64 x 64 layers): with it on the 256 x 256
# layers too the full-size gradient test misses its 2e-2 bound on the 95th percentile (2.2e-2: the forward's rounding is amplified
# through the batch statistics); the data gradients take F(4, 3) everywhere (ops.ConvDgrad)
# forward of the 3x3 / stride-1 layers on maps up to this size on the chained F(2,3) x F(4,3) kernel (NHWC -> planes + one launch: 77 against
# 91 us on the 64 x 64 x 256 layers at batch 4, -0.05 ms per iteration; the gradient statistics of the full-size test do not move; the
# same kernel as the data gradient measured no gain beside the weight gradients and is not used)


class _Conv:
    """plain convolution (optional bias, optional fused activation) with explicit backward"""

    def __init__(self, ps: ParamStore, wname: str, bname: Optional[str], stride=1, pad=0, act=NONE, transposed=False,
                 cin_pad: Optional[int] = None):
        self.ps, self.wname, self.bname = ps, wname, bname
        w = ps.p[wname]
        self.stride, self.pad, self.act, self.transposed = stride, pad, act, transposed
        self.k = w.shape[2]
        if transposed:  # ConvTranspose2d(k=2, s=2): weight (Cin, Cout, 2, 2)
            self.layer = ops.ConvLayer(w, deconv2x2=True, act=act)
            # its data gradient is the stride-2 convolution with the same tensor read as (Cout_conv, Cin_conv, 2, 2)
            self.dgrad = ops.ConvLayer(w, stride=2, pad=0)
        else:
            self.layer = ops.ConvLayer(w, stride=stride, pad=pad, shift=None if bname is None else ps.p[bname], act=act, wino4=R.train_wino4_max_pixels > 0)
            self.layer.wino4_max_pixels = R.train_wino4_max_pixels
            if cin_pad is not None:
                self.layer.pad_input_channels(cin_pad)
            self.dgrad = ops.ConvDgrad(w, stride, pad)
        self.cin_real = w.shape[1] if not transposed else w.shape[0]
        self.x = None
        self.in_co = 0
        ps.convs.append(self)

    def prepack_fwd(self, token):
        self.layer.repack(self.ps.p[self.wname], None if self.bname is None else self.ps.p[self.bname], token=token)
        self.layer.prepack_used()

    def prepack_bwd(self, token):
        self.dgrad.repack(self.ps.p[self.wname], token=token)
        self.dgrad.prepack_used()

    def _chain_ok(self, t, layer) -> bool:
        return (R.train_chain_max_pixels > 0 and not self.transposed and self.k == 3 and self.stride == 1 and self.pad == 1
                and t.shape[1] * t.shape[2] <= R.train_chain_max_pixels and t.shape[3] == layer.cin
                and ops.conv_chain_supported([layer], t.shape[0], t.shape[1], t.shape[2]))

    def fwd(self, x, out=None, out_co=0, in_co=0):
        w = self.ps.p[self.wname]
        self.layer.repack(w, None if self.bname is None else self.ps.p[self.bname], token=self.ps.fresh)
        self.x, self.in_co = x, in_co
        if out is None and in_co == 0 and self._chain_ok(x, self.layer):
            return ops.conv_chain([self.layer], x)
        return self.layer(x, out=out, out_channel_offset=out_co, in_channel_offset=in_co)

    def bwd(self, dout, cout: Optional[int] = None, need_dx=True, dx=None, dx_co=0, accumulate=False):
        """dout: NHWC gradient of the (pre-activation) output, own tensor (channel offset 0)"""
        w = self.ps.p[self.wname]
        gw = self.ps.g[self.wname]
        if self.transposed:
            self.ps.side.run(lambda: ops.conv_wgrad(dout, self.x, 2, 2, 2, 0, cout=w.shape[0], dout_channel_offset=self.in_co, out=gw), dout, self.x)
            if need_dx:
                self.dgrad.repack(w, token=self.ps.fresh)
                return self.dgrad(dout, out=dx, out_channel_offset=dx_co, accumulate=accumulate)
            return None
        cout = w.shape[0] if cout is None else cout

        def weight_grads():
            ops.conv_wgrad(self.x, dout, self.k, self.k, self.stride, self.pad, cin=self.cin_real, in_channel_offset=self.in_co,
                           cout=cout, out=gw)
            if self.bname is not None:
                ops.channel_sum(dout, c=cout, out=self.ps.g[self.bname])

        self.ps.side.run(weight_grads, dout, self.x)
        if need_dx:
            self.dgrad.repack(w, token=self.ps.fresh)
            return self.dgrad(dout, out=dx, out_channel_offset=dx_co, accumulate=accumulate)
        return None


class _PillarConv:
    """the first convolution of the RPN on the sparse pillar canvas (ops.PillarConvLayer, csrc/pillar_conv.hip): forward over the
    (pillar, tap) pairs, data gradient straight to d(pillar features), weight gradient over the pairs -- the dense forms spend
    2.2 ms of a bs = 4 iteration on a map that is 89 % zeros.  ``vi`` (the iteration's VoxelIndex) is set before ``fwd``."""

    def __init__(self, ps: ParamStore, wname: str, stride: int):
        self.ps, self.wname = ps, wname
        self.layer = ops.PillarConvLayer(ps.p[wname], stride)
        self.vi = self.x = self.tables = None
        ps.convs.append(self)

    def prepack_fwd(self, token):
        self.layer.repack(self.ps.p[self.wname], token=token)

    def prepack_bwd(self, token):
        pass

    def fwd(self, x, out=None, out_co=0, in_co=0):
        assert out is None and in_co == 0
        self.layer.repack(self.ps.p[self.wname], token=self.ps.fresh)
        self.x = x
        self.tables = self.layer.build_tables(self.vi, x.shape[0], x.shape[1], x.shape[2])
        return self.layer.forward_tables(x, self.vi, self.tables)

    def bwd(self, dout, cout=None, need_dx=True, dx=None, dx_co=0, accumulate=False):
        """-> d(pillar features) (n_cap, Cin) instead of a dense canvas gradient (dynamic_pfn_bwd takes either)"""
        self.ps.side.run(lambda: self.layer.wgrad(self.x, dout, self.vi, self.tables, out=self.ps.g[self.wname]), dout, self.x, self.tables)
        return self.layer.dgrad_features(dout, self.vi, self.tables) if need_dx else None


class _ConvBNReLU:
    """Conv2d / ConvTranspose2d (no bias) + BatchNorm2d (batch statistics) + ReLU  (rpn.py:124-142, 80-110)"""

    def __init__(self, ps: ParamStore, prefix: str, conv_idx: int, bn: nn.BatchNorm2d, conv: nn.Module):
        transposed = isinstance(conv, nn.ConvTranspose2d)
        pad = 1 if conv.kernel_size[0] == 3 else 0  # nn.ZeroPad2d(1) + Conv2d(3, padding=0) == padding 1
        self.conv = _Conv(ps, f"{prefix}{conv_idx}.weight", None, conv.stride[0], pad, NONE, transposed)
        self.gname, self.bname = f"{prefix}{conv_idx + 1}.weight", f"{prefix}{conv_idx + 1}.bias"
        self.ps, self.bn = ps, bn
        self.y = self.stat = None

    def fwd(self, x, out=None, out_co=0):
        self.y = self.conv.fwd(x)
        res, self.stat = ops.batchnorm_train(self.y, self.ps.p[self.gname], self.ps.p[self.bname], self.bn.eps, self.bn.momentum,
                                             self.bn.running_mean, self.bn.running_var, act=RELU, out=out, out_channel_offset=out_co)
        return res

    def bwd(self, dout, dout_co=0, need_dx=True, dx=None, accumulate=False):
        c = self.y.shape[3]
        inplace = dout.shape[3] == c and dout_co == 0
        dy = dout if inplace else torch.empty_like(self.y)
        ops.batchnorm_bwd(self.y, dout, self.ps.p[self.gname], self.ps.p[self.bname], self.stat, act=RELU, dx=dy,
                          dgamma=self.ps.g[self.gname], dbeta=self.ps.g[self.bname], dout_channel_offset=dout_co)
        return self.conv.bwd(dy, need_dx=need_dx, dx=dx, accumulate=accumulate)


class _ConvReLU:
    """Conv2d(bias) + ReLU of the plain CenterHead (center_head.py:65-109, 166-242)"""

    def __init__(self, ps: ParamStore, wname: str, bname: str, pad: int):
        self.conv = _Conv(ps, wname, bname, 1, pad, act=RELU)
        self.z = None

    def fwd(self, x):
        self.z = self.conv.fwd(x)   # ReLU fused into the epilogue; its output is also the backward mask
        return self.z

    def bwd(self, dout, dx=None, accumulate=False):
        dy = ops.relu_bwd(self.z, dout, dx=dout)
        return self.conv.bwd(dy, dx=dx, accumulate=accumulate)


class _ConvGNReLU:
    """Conv2d(bias) + GroupNorm-family + ReLU (+ calibration second output) of the merged heads"""

    def __init__(self, ps, wname, bname, gname, bename, cgroups, strata, eps, groups=1):
        self.ps = ps
        self.groups, self.cgroups, self.strata, self.eps = groups, cgroups, strata, eps
        self.wname, self.bname, self.gname, self.bename = wname, bname, gname, bename
        w = ps.p[wname]
        self.cout = w.shape[0]
        if groups == 1:
            self.convs = [_Conv(ps, wname, bname, 1, 1)]
        else:  # grouped convolution: forward in one launch, backward per group on weight / channel slices
            self.layer = ops.ConvLayer(w, stride=1, pad=1, groups=groups, shift=ps.p[bname])
            cg = self.cout // groups
            self.dgrads = [ops.ConvDgrad(w[g * cg:(g + 1) * cg], 1, 1) for g in range(groups)]
        self.x = self.y = None

    def fwd(self, x, mul=None, add=None):
        self.x = x
        if self.groups == 1:
            self.y = self.convs[0].fwd(x)
        else:
            self.layer.repack(self.ps.p[self.wname], self.ps.p[self.bname])
            self.y = self.layer(x)
        self.mul = mul
        self.stat = torch.empty(2 * self.y.shape[0] * self.strata * self.cgroups, dtype=torch.float32, device=self.y.device)
        return ops.groupnorm_strat(self.y, self.cgroups, self.strata, self.ps.p[self.gname], self.ps.p[self.bename], self.eps,
                                   act=RELU, mul=mul, add=add, stat_out=self.stat)

    def bwd(self, dout, dout2=None, dx=None, accumulate=False, need_dx=True):
        res = ops.groupnorm_strat_bwd(self.y, dout, self.cgroups, self.strata, self.ps.p[self.gname], self.ps.p[self.bename], self.eps,
                                      RELU, dout2=dout2, mul=self.mul if dout2 is not None else None, dx=dout,
                                      dgamma=self.ps.g[self.gname], dbeta=self.ps.g[self.bename], stat=self.stat)
        dy = res[0]
        extra = res[3:] if dout2 is not None else ()
        if self.groups == 1:
            dxr = self.convs[0].bwd(dy, need_dx=need_dx, dx=dx, accumulate=accumulate)
            return (dxr,) + tuple(extra)
        w, gw = self.ps.p[self.wname], self.ps.g[self.wname]
        cg, cin_g = self.cout // self.groups, w.shape[1]
        def weight_grads():
            ops.channel_sum(dy, out=self.ps.g[self.bname])
            for g in range(self.groups):
                ops.conv_wgrad(self.x, dy, 3, 3, 1, 1, cin=cin_g, in_channel_offset=g * cin_g, cout=cg, dout_channel_offset=g * cg,
                               out=gw[g * cg:(g + 1) * cg])

        self.ps.side.run(weight_grads, dy, self.x)
        if dx is None:
            dx = torch.empty_like(self.x)
        for g in range(self.groups):
            self.dgrads[g].repack(w[g * cg:(g + 1) * cg])
            self.dgrads[g](dy, out=dx, dout_channel_offset=g * cg, out_channel_offset=g * cin_g, accumulate=accumulate)
        return (dx,) + tuple(extra)


class _StratConvGNReLU:
    """RangeStratified: per-range-stratum 3x3 convolution + GroupNorm(strata) + ReLU (center_head_parallel.py:27-59)"""

    def __init__(self, ps, prefix: str, m: RangeStratified):
        self.ps, self.strata, self.nheads = ps, m.ngroups * m.nheads, m.nheads
        self.wname, self.bname = prefix + "conv.0.weight", prefix + "conv.0.bias"
        self.gname, self.bename = prefix + "conv.1.weight", prefix + "conv.1.bias"
        self.eps = m.conv[1].eps
        w = ps.p[self.wname]
        self.layer = ops.ConvLayer(w, stride=1, pad=1, range_strata=self.strata, shift=ps.p[self.bname])
        # r4: gradients at the convolution's own multiply-add count (ops.StratConvDgrad, pn_conv2d_wgrad_f32 with range_strata); r3
        # expanded dy to strata * C channels and ran ordinary gradient convolutions over the zeros: 1.0 ms of a 14 ms iteration
        self.masked = not R.train_strat_expand
        self.dgrad = ops.StratConvDgrad(w, self.strata) if self.masked else ops.ConvDgrad(w, 1, 1)
        self.x = self.y = None
        ps.convs.append(self)

    def prepack_fwd(self, token):
        self.layer.repack(self.ps.p[self.wname], self.ps.p[self.bname], token=token)
        self.layer.prepack_used()

    def prepack_bwd(self, token):
        self.dgrad.repack(self.ps.p[self.wname], token=token)
        self.dgrad.prepack_used()

    def fwd(self, x):
        self.x = x
        self.layer.repack(self.ps.p[self.wname], self.ps.p[self.bname], token=self.ps.fresh)
        self.y = self.layer(x)
        self.stat = torch.empty(2 * self.y.shape[0] * self.strata * self.nheads, dtype=torch.float32, device=self.y.device)
        return ops.groupnorm_strat(self.y, self.nheads, self.strata, self.ps.p[self.gname], self.ps.p[self.bename], self.eps, act=RELU,
                                   stat_out=self.stat)

    def bwd(self, dout, dx, accumulate):
        dy, _, _ = ops.groupnorm_strat_bwd(self.y, dout, self.nheads, self.strata, self.ps.p[self.gname], self.ps.p[self.bename], self.eps,
                                           RELU, dx=dout, dgamma=self.ps.g[self.gname], dbeta=self.ps.g[self.bename], stat=self.stat)
        w = self.ps.p[self.wname]
        if self.masked:
            def weight_grads():
                ops.conv_wgrad(self.x, dy, 3, 3, 1, 1, out=self.ps.g[self.wname], range_strata=self.strata)
                ops.strat_channel_sum(dy, self.strata, self.ps.g[self.bname])

            self.ps.side.run(weight_grads, dy, self.x)
            self.dgrad.repack(w, token=self.ps.fresh)
            return self.dgrad(dy, out=dx, accumulate=accumulate)
        full = ops.strat_expand(dy, self.strata)  # zeros outside the pixel's stratum: an ordinary conv gradient (the r3 form)

        def weight_grads():
            ops.conv_wgrad(self.x, full, 3, 3, 1, 1, out=self.ps.g[self.wname])
            ops.channel_sum(full, out=self.ps.g[self.bname])

        self.ps.side.run(weight_grads, full, self.x)
        self.dgrad.repack(w, token=self.ps.fresh)
        return self.dgrad(full, out=dx, accumulate=accumulate)


class PolarPillarTrainStep:
    """One training iteration of PointPillars(DynamicPFNet, DynamicPPScatter, RPN, CenterHeadSinglePos).

    ``step(points, sample_offsets, batch, targets)`` runs forward (train mode), loss, backward, gradient
    all-reduce (if torch.distributed is initialised with world_size > 1), clip + decoupled wd + Adam with
    the OneCycle schedule, and returns the loss vector [det, hm, loc, num_pos, elem...] (device)."""

    def __init__(self, model: nn.Module, total_steps: int, lr_max=0.005, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4,
                 weight_decay=0.01, max_norm=35.0, beta2=0.99, eps=1e-8):
        self.model = model
        reader, neck, head = model.reader, model.neck, model.bbox_head
        if not isinstance(reader, DynamicPFNet):
            raise NotImplementedError("training step: the reader must be a DynamicPFNet")
        self.plain_head = type(head).__name__ == "CenterHead"
        if not (isinstance(head, CenterHeadSingle) or self.plain_head):
            raise NotImplementedError("training step: the head must be CenterHead, CenterHeadSingle or CenterHeadSinglePos")
        reader._check_supported()
        dev = next(model.parameters()).device
        hip.require_device(next(model.parameters()))
        self.dev = dev
        # flat-buffer order = reverse of the order in which backward completes the gradients: reader | block 0 (+ its deblock)
        # | block 1 ... | head, so that the buckets of the gradient exchange (head first) are contiguous ranges
        up0 = neck._upsample_start_idx

        def order_key(name: str):
            parts = name.split(".")
            if parts[0] == "reader":
                return 0
            if parts[0] == "neck" and parts[1] == "blocks":
                return 1 + 2 * int(parts[2])
            if parts[0] == "neck" and parts[1] == "deblocks":
                return 2 + 2 * (int(parts[2]) + up0)
            return 1000 if parts[0] == "bbox_head" else 999

        self.ps = ps = ParamStore(model, dev, order_key)
        # gradient buckets [lo, hi) in the order backward completes them: the head, the last RPN block (+ deblock), the rest
        first_head = min((ps.offsets[n][0] for n in ps.names if n.startswith("bbox_head.")), default=ps.total)
        last_blk = len(neck.blocks) - 1
        first_last = min((ps.offsets[n][0] for n in ps.names if order_key(n) >= 1 + 2 * last_blk), default=first_head)
        self.buckets = [b for b in ((first_head, ps.total), (first_last, first_head), (0, first_last)) if b[1] > b[0]]
        self.sched = dict(total=total_steps, lr_max=lr_max, moms=tuple(moms), div=div_factor, pct=pct_start)
        self.wd, self.max_norm, self.beta2, self.eps = weight_decay, max_norm, beta2, eps
        self.iter = 0
        self._exchange = None
        self.reader, self.neck, self.head = reader, neck, head
        self.spec = ops.GridSpec.from_range(reader.pc_range, reader.voxel_size)
        self.w0, self.w1 = "reader.pfn_layers.0.linear.weight", "reader.pfn_layers.1.linear.weight"
        # ---- RPN
        self.blocks: List[List[_ConvBNReLU]] = []
        for i, blk in enumerate(neck.blocks):
            mods = list(blk._modules.values())
            layers = [_ConvBNReLU(ps, f"neck.blocks.{i}.", 1, mods[2], mods[1])]
            if i == 0 and R.train_pillar_conv and ops.PillarConvLayer.supports(mods[1].weight, mods[1].stride[0], mods[1].groups) and mods[1].out_channels <= 128 \
                    and mods[1].out_channels in (32, 64, 128):
                layers[0].conv = _PillarConv(ps, f"neck.blocks.{i}.1.weight", mods[1].stride[0])
            for k in range(4, len(mods), 3):
                layers.append(_ConvBNReLU(ps, f"neck.blocks.{i}.", k, mods[k + 1], mods[k]))
            self.blocks.append(layers)
        self.deblocks = [_ConvBNReLU(ps, f"neck.deblocks.{j}.", 0, de[1], de[0]) for j, de in enumerate(neck.deblocks)]
        self.up_start = neck._upsample_start_idx
        self.up_filters = list(neck._num_upsample_filters)
        # ---- head
        hp = "bbox_head."
        self.code_weights, self.loss_weight = list(head.code_weights), float(head.weight)
        self.ncls = sum(head.num_classes)
        self.has_pos = False
        if self.plain_head:
            self.shared = _ConvReLU(ps, hp + "shared_conv.0.weight", hp + "shared_conv.0.bias", 1)
            # one set of branches per task (center_head.py:166-242: `for task in self.tasks`); r6: any number of tasks -- the branch key is
            # (task, name), every task has its own loss (center_head.py:250: one term per task, summed by the trainer) and its own targets
            self.branches = {}
            self.task_ncls = [int(n) for n in head.num_classes]
            for t, task in enumerate(head.tasks):
                for name in task.heads:
                    convs = [(i, mod) for i, mod in enumerate(getattr(task, name)._modules.values()) if isinstance(mod, nn.Conv2d)]
                    bp = f"{hp}tasks.{t}.{name}."
                    hidden = [_ConvReLU(ps, f"{bp}{i}.weight", f"{bp}{i}.bias", mod.padding[0]) for i, mod in convs[:-1]]
                    i, mod = convs[-1]
                    self.branches[(t, name)] = ("plain", hidden, _Conv(ps, f"{bp}{i}.weight", f"{bp}{i}.bias", 1, mod.padding[0]), 1)
            return
        rs = head.shared_conv[1]
        assert isinstance(rs, RSNorm)
        self.shared = _ConvGNReLU(ps, hp + "shared_conv.0.weight", hp + "shared_conv.0.bias", hp + "shared_conv.1.groupnorm.weight",
                                  hp + "shared_conv.1.groupnorm.bias", rs.num_heads, rs.num_groups, rs.groupnorm.eps)
        self.branches = {}
        for name in head.heads:
            fc = getattr(head, name)
            mods = list(fc._modules.values())
            bp = f"{hp}{name}."
            if isinstance(mods[0], RangeStratified):
                first = _StratConvGNReLU(ps, bp + "0.", mods[0])
                last = _Conv(ps, bp + "1.weight", bp + "1.bias", 1, 0)
                self.branches[name] = ("strat", first, last, 1)
            else:
                assert len(mods) == 4, "training step: heads with one hidden conv (num_conv = 2)"
                groups = mods[0].groups
                first = _ConvGNReLU(ps, bp + "0.weight", bp + "0.bias", bp + "1.weight", bp + "1.bias", mods[1].num_groups, 1, mods[1].eps,
                                    groups=groups)
                w = ps.p[bp + "3.weight"]
                if groups == 1:
                    last = _Conv(ps, bp + "3.weight", bp + "3.bias", 1, 1)
                else:
                    last = ops.ConvLayer(w, stride=1, pad=1, groups=groups, shift=ps.p[bp + "3.bias"])
                self.branches[name] = ("conv", first, last, groups)
        self.has_pos = isinstance(head, CenterHeadSinglePos)
        if self.has_pos:
            pos = ops.to_nhwc(head.pos_encoding.to(dev).float())               # (1, A, R, 5)
            self.pos8 = torch.zeros(pos.shape[:3] + (8,), dtype=torch.float32, device=dev)
            self.pos8[..., :5] = pos
            self.cal = {}
            for kind, last_act in (("calibration_weight", TANH), ("calibration_bias", NONE)):
                cp = f"{hp}{kind}."
                self.cal[kind] = (_Conv(ps, cp + "0.weight", cp + "0.bias", 1, 1, act=TANH, cin_pad=8),
                                  _Conv(ps, cp + "2.weight", cp + "2.bias", 1, 0, act=last_act), last_act)
        self.code_weights, self.loss_weight = list(head.code_weights), float(head.weight)
        self.ncls = sum(head.num_classes)

    # ------------------------------------------------------------------------------------------
    def _forward(self, points, sample_offsets, batch, grid_ind=None):
        ps, spec = self.ps, self.spec
        if grid_ind is None:
            _, keys = ops.grid_index(points, sample_offsets, batch, spec, want_grid_ind=False)
            n_dev = sample_offsets[batch:]
        else:
            keys = ops.keys_from_grid_ind(grid_ind.to(torch.int64).contiguous(), spec, batch)
            n_dev = None
        self.vi = ops.build_voxel_index(keys, spec, batch, n_dev=n_dev, want_unq=False, sorted_runs=True)
        self.points = points
        canvas = torch.empty((batch, spec.grid[1], spec.grid[0], self.reader.out_channels), dtype=torch.float32, device=self.dev)
        first = self.blocks[0][0].conv
        if isinstance(first, _PillarConv):
            first.vi = self.vi          # only the pillars' cells of the canvas are ever read: no 134 MB-per-sample zero fill
        else:
            hip.call("pn_fill_zero", canvas.data_ptr(), canvas.numel() * 4, hip.stream())
        r = self.reader
        ops.dynamic_pfn(points, self.vi, ps.p[self.w0], ps.p[self.w1], r.vx, r.vy, r.x_offset, r.y_offset, None, canvas)
        self._prepack()     # issued behind the scatter stage's launches, runs beside them
        x = canvas
        out, off = None, 0
        self.block_out = []
        for i, layers in enumerate(self.blocks):
            if i < 2:
                self._packs_ready(i)
            for layer in layers:
                x = layer.fwd(x)
            self.block_out.append(x)
            j = i - self.up_start
            if j >= 0:
                de = self.deblocks[j]
                if out is None:
                    oh, ow = de.conv.layer.out_hw(x.shape[1], x.shape[2])
                    out = torch.empty((batch, oh, ow, sum(self.up_filters)), dtype=torch.float32, device=self.dev)
                de.fwd(x, out=out, out_co=off)
                off += self.up_filters[j]
        self._packs_ready(1)    # models with a single block reach the head without having met event 1 in the loop
        self.x2 = out
        # ---- head
        mul = add = None
        if self.has_pos:
            self.cal_mid = {}
            maps = {}
            for kind, (c0, c1, _) in self.cal.items():
                mid = c0.fwd(self.pos8)
                self.cal_mid[kind] = mid
                maps[kind] = c1.fwd(mid)
            self.cal_w, self.cal_b = maps["calibration_weight"], maps["calibration_bias"]
            mul, add = self.cal_w[0], self.cal_b[0]
        if self.plain_head:
            xs = x_hm = self.shared.fwd(out)
        elif mul is not None:
            xs, x_hm = self.shared.fwd(out, mul=mul, add=add)
        else:
            xs = x_hm = self.shared.fwd(out)
        self.xs, self.x_hm = xs, x_hm
        preds = {}
        self.branch_mid = {}
        self.preds_tasks = None
        for name, (kind, first, last, groups) in self.branches.items():
            if kind == "plain":
                t, nm = name
                if self.preds_tasks is None:
                    self.preds_tasks = [dict() for _ in self.task_ncls]
                z = xs
                for layer in first:
                    z = layer.fwd(z)
                self.preds_tasks[t][nm] = last.fwd(z)
                continue
            z = first.fwd(x_hm if name == "hm" else xs)
            self.branch_mid[name] = z
            if kind == "conv" and groups > 1:
                last.repack(ps.p[f"bbox_head.{name}.3.weight"], ps.p[f"bbox_head.{name}.3.bias"])
                y = last(z)
            else:
                y = last.fwd(z)
            if "_" in name:
                names = name.split("_")
                dim = y.shape[3] // len(names)
                for k, nm in enumerate(names):
                    preds[nm] = y[..., k * dim:(k + 1) * dim]
            else:
                preds[name] = y
        if self.preds_tasks is not None:
            preds = self.preds_tasks[0]
        self.preds = preds
        return preds

    def _loss_sources(self, preds=None):
        preds = self.preds if preds is None else preds
        order = ["reg", "height", "dim"] + (["vel"] if "vel" in preds else []) + ["rot"]
        return order, [(preds[k], preds[k].shape[3]) for k in order]

    def _task_list(self, targets):
        """-> [(prediction dict, class count, targets)] per task; a single-task step takes its targets bare, several tasks a list"""
        if getattr(self, "preds_tasks", None) is None or len(self.preds_tasks) == 1:
            tg = targets[0] if isinstance(targets, (list, tuple)) else targets
            return [(self.preds, self.ncls, tg)]
        assert isinstance(targets, (list, tuple)) and len(targets) == len(self.preds_tasks), "one CenterLossTargets per task"
        return [(p, n, tg) for p, n, tg in zip(self.preds_tasks, self.task_ncls, targets)]

    # ------------------------------------------------------------------------------------------
    def _backward(self, targets: ops.CenterLossTargets, loss_out, grad_scale: float):
        ps = self.ps
        tasks = self._task_list(targets)
        loss_out = loss_out if isinstance(loss_out, (list, tuple)) else [loss_out]
        multi = len(tasks) > 1
        d_pred = {}
        for t, ((preds, ncls, tg), lo) in enumerate(zip(tasks, loss_out)):
            order, boxes = self._loss_sources(preds)
            d_hm, d_boxes = ops.center_loss_bwd(preds["hm"], ncls, boxes, tg, self.code_weights, self.loss_weight, lo,
                                                grad_scale=grad_scale, with_vel="vel" in preds)
            for nm, d in list(zip(order, d_boxes)) + [("hm", d_hm)]:
                d_pred[(t, nm) if (multi or self.plain_head) else nm] = d
        d_xs = torch.empty_like(self.xs)
        d_xhm = None
        first_into_xs = True
        for name, (kind, first, last, groups) in self.branches.items():
            if kind == "plain":
                dz = last.bwd(d_pred[name], cout=ps.p[last.wname].shape[0], dx=None if first else d_xs, accumulate=(not first) and not first_into_xs)
                for li in range(len(first) - 1, -1, -1):
                    into_xs = li == 0
                    dz = first[li].bwd(dz, dx=d_xs if into_xs else None, accumulate=into_xs and not first_into_xs)
                first_into_xs = False
                continue
            z = self.branch_mid[name]
            if kind == "conv" and groups > 1:
                # grouped final convolution: one output tensor per merged head (e.g. rot | vel)
                names = name.split("_")
                w, gw, gb = ps.p[f"bbox_head.{name}.3.weight"], ps.g[f"bbox_head.{name}.3.weight"], ps.g[f"bbox_head.{name}.3.bias"]
                co, ci = w.shape[0] // groups, w.shape[1]
                dz = torch.empty_like(z)
                for g_, nm in enumerate(names):
                    dy = d_pred[nm]
                    def weight_grads(dy=dy, g_=g_):
                        ops.conv_wgrad(z, dy, 3, 3, 1, 1, cin=ci, in_channel_offset=g_ * ci, cout=co, out=gw[g_ * co:(g_ + 1) * co])
                        ops.channel_sum(dy, c=co, out=gb[g_ * co:(g_ + 1) * co])

                    ps.side.run(weight_grads, dy, z)
                    ops.ConvDgrad(w[g_ * co:(g_ + 1) * co], 1, 1)(dy, out=dz, out_channel_offset=g_ * ci)
            else:
                cout = ps.p[last.wname].shape[0]
                dz = last.bwd(d_pred[name], cout=cout)
            if name == "hm" and self.x_hm is not self.xs:
                d_xhm = first.bwd(dz)[0]
            elif kind == "strat":
                first.bwd(dz, dx=d_xs, accumulate=not first_into_xs)
                first_into_xs = False
            else:
                first.bwd(dz, dx=d_xs, accumulate=not first_into_xs)
                first_into_xs = False
        # shared conv + RSNorm (+ calibration)
        if self.plain_head:
            d_x2 = self.shared.bwd(d_xs)
        else:
            res = self.shared.bwd(d_xs, dout2=d_xhm)
            d_x2 = res[0]
        if self.has_pos:
            d_calw, d_calb = res[1], res[2]
            for kind, dmap, ymap in (("calibration_weight", d_calw, self.cal_w), ("calibration_bias", d_calb, self.cal_b)):
                c0, c1, last_act = self.cal[kind]
                d = dmap[None].contiguous()
                if last_act == TANH:
                    d = ops.tanh_bwd(ymap, d)
                dmid = c1.bwd(d)
                dmid = ops.tanh_bwd(self.cal_mid[kind], dmid, dx=dmid)
                c0.bwd(dmid, need_dx=False)
        self._bucket_ready(0)   # every head gradient is queued: its exchange overlaps the backward of the RPN
        # RPN
        nblk = len(self.blocks)
        d_block = None
        for i in range(nblk - 1, -1, -1):
            j = i - self.up_start
            if j >= 0:
                off = sum(self.up_filters[:j])
                if d_block is None:
                    d_block = self.deblocks[j].bwd(d_x2, dout_co=off)
                else:
                    self.deblocks[j].bwd(d_x2, dout_co=off, dx=d_block, accumulate=True)
            d = d_block
            for layer in reversed(self.blocks[i]):
                d = layer.bwd(d)
            d_block = d
            if i == nblk - 1 and len(self.buckets) > 2:
                self._bucket_ready(1)
        r = self.reader
        sparse = isinstance(self.blocks[0][0].conv, _PillarConv)      # then d_block is d(pillar features) (n_cap, C), not a canvas gradient
        ops.dynamic_pfn_bwd(self.points, self.vi, ps.p[self.w0], ps.p[self.w1], r.vx, r.vy, r.x_offset, r.y_offset,
                            d_features=d_block if sparse else None, d_canvas=None if sparse else d_block, dw0=ps.g[self.w0], dw1=ps.g[self.w1])
        ps.side.join()

    def _bucket_ready(self, k: int) -> None:
        """start the SUM all-reduce of gradient bucket k (asynchronous: RCCL's stream waits for the kernels queued so far and
        runs next to the rest of backward; the reference's DDP reducer does the same per bucket, det3d/torchie/apis/train.py:330-336)"""
        if self._exchange is not None:
            self.ps.side.join()      # the bucket's weight gradients are computed on the side stream
            self._exchange.ready(k)

    # ------------------------------------------------------------------------------------------
    def _prepack(self):
        """Refresh the packed weight copies of the iteration AHEAD of their use, on the side stream: the forward is a serial chain on the
        main stream with the side stream idle, and every layer's pack launch (2 - 10 us, ~65 per iteration) sat in that chain in front of
        its convolution.  Three events: the first RPN stage's forward layouts (the main stream meets them after the PFN), the other forward
        layouts, the data-gradient layouts.  A layout is refreshed here once a call has taken it (the first iteration packs lazily as
        before); ``ps.fresh`` is the iteration's token, the wrappers' own repack calls see it and do nothing."""
        ps = self.ps
        side = ps.side.stream if R.train_prepack and getattr(self, "_prepack_armed", False) else None   # armed by forward_backward only
        self._prepack_armed = False
        self._pack_events = None
        if side is None:
            ps.fresh = None
            return
        tok = ps.fresh = object()
        # group 0 = every layout the main stream takes before it waits for event 1: block 0 and the deblock fed by block 0
        first = {id(layer.conv) for layer in self.blocks[0]}
        if self.up_start == 0 and self.deblocks:
            first.add(id(self.deblocks[0].conv))
        groups = ([c for c in ps.convs if id(c) in first], [c for c in ps.convs if id(c) not in first])
        if getattr(self, "_pack_ev", None) is None:
            self._pack_ev = [torch.cuda.Event() for _ in range(3)]

        def packs():
            for k, group in enumerate(groups):
                for c in group:
                    c.prepack_fwd(tok)
                self._pack_ev[k].record()
            for c in ps.convs:
                c.prepack_bwd(tok)
            self._pack_ev[2].record()

        ps.side.run(packs, after=self._iter_start)
        self._pack_events = self._pack_ev

    def _packs_ready(self, k: int) -> None:
        if getattr(self, "_pack_events", None) is not None:
            torch.cuda.current_stream().wait_event(self._pack_events[k])

    def forward_backward(self, points, sample_offsets, batch, targets: ops.CenterLossTargets, grid_ind=None, grad_scale=1.0):
        """forward + loss + backward; gradients land in ``self.ps.flat_g`` (overwritten, not accumulated)"""
        if getattr(self, "_iter_start", None) is None and self.dev.type == "cuda":
            self._iter_start = torch.cuda.Event()
        if getattr(self, "_iter_start", None) is not None:
            self._iter_start.record()     # the previous iteration's optimizer step is queued before it: the packs wait for this, not for the PFN
            self._prepack_armed = True
        try:
            self._forward(points, sample_offsets, batch, grid_ind)
            losses = []
            for preds, ncls, tg in self._task_list(targets):
                order, boxes = self._loss_sources(preds)
                losses.append(ops.center_loss(preds["hm"], ncls, boxes, tg, self.code_weights, self.loss_weight, with_vel="vel" in preds))
            self._packs_ready(2)
            self._backward(targets, losses, grad_scale)
            loss = losses[0] if len(losses) == 1 else torch.stack(losses)      # several tasks: one row per task (the trainer sums the det_loss terms)
        finally:
            self.ps.fresh = None
            self._pack_events = None
        return loss

    def optimizer_step(self):
        s = self.sched
        lr, beta1 = one_cycle(self.iter, s["total"], s["lr_max"], s["moms"], s["div"], s["pct"])
        total_norm = ops.grad_norm(self.ps.flat_g)
        ops.adam_step(self.ps.flat_p, self.ps.flat_g, self.ps.flat_m, self.ps.flat_v, self.iter + 1, lr, beta1, self.beta2, self.eps,
                      self.wd, total_norm=total_norm, max_norm=self.max_norm)
        self.iter += 1
        self.invalidate_inference_plans()
        return total_norm

    # ---- checkpoint / resume (the reference saves model + optimizer state per epoch, det3d/torchie/trainer/trainer.py:342-372)
    def state_dict(self) -> Dict[str, object]:
        """optimizer-side state; the parameters themselves are in ``model.state_dict()`` (views of the flat buffer)"""
        return optimizer_state(self.ps, self.iter, self.sched)

    def load_state_dict(self, state: Dict[str, object]) -> None:
        self.iter = load_optimizer_state(self.ps, state)
        self.sched.update(state.get("schedule", {}))
        self.invalidate_inference_plans()

    def invalidate_inference_plans(self):
        invalidate_inference_plans(self.model)

    def step(self, points, sample_offsets, batch, targets: ops.CenterLossTargets, grid_ind=None):
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi and self.ps.side.on and dist.get_backend() == "gloo":
            # gloo is the transport of the same-GPU dry runs (several ranks on ONE device, tests / bench.py --backend gloo): with the ranks'
            # extra streams the device's hardware queues are oversubscribed across processes and every collective waits for a time slice
            # (1585 against 42 ms per iteration, two ranks on one GPU) -- one stream per rank there
            self.ps.side.on = False
        return self._step(points, sample_offsets, batch, targets, grid_ind)

    def _step(self, points, sample_offsets, batch, targets: ops.CenterLossTargets, grid_ind=None):
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        from .dist_utils import GradExchange
        # exchange_enabled = False: a timing-only mode of bench.py (the same iteration without the collectives; ranks then diverge)
        # exchange_at_world_1 = True: the collectives run over a ONE-rank group too (dist_utils.init(single_rank_group=True)): the RCCL path
        # of a one-GPU box; the result is the no-exchange step's, bit for bit
        force = bool(getattr(self, "exchange_at_world_1", False)) and world == 1 and dist.is_available() and dist.is_initialized()
        self._exchange = GradExchange(self.ps.flat_g, self.buckets, force=force) if (world > 1 or force) and getattr(self, "exchange_enabled", True) else None
        # DDP's broadcast_buffers=True (the reference's setting, dist_utils.BufferSync): rank 0's BatchNorm running statistics to every rank
        # at the start of the forward -- one broadcast of one flat tensor, queued in front of the iteration
        if (world > 1 or force) and getattr(self, "broadcast_buffers", True) and getattr(self, "exchange_enabled", True):
            if getattr(self, "_bufsync", None) is None:
                from .dist_utils import BufferSync
                self._bufsync = BufferSync(self.model, self.dev)
            self._bufsync.sync(force=force)
        # the reference averages the gradients over ranks (dist_utils.py:17-28): fold 1/world into the loss gradient
        try:
            loss = self.forward_backward(points, sample_offsets, batch, targets, grid_ind, grad_scale=1.0 / world)
            if self._exchange is not None:
                self._exchange.finish()   # the last bucket (reader + first blocks), then the stream waits for all of them
        finally:
            self._exchange = None
        self.optimizer_step()
        return loss

    def sync_initial_params(self):
        """rank 0's parameters (and BatchNorm running statistics) to every rank, once before the first step"""
        from .dist_utils import broadcast_flat_params
        force = bool(getattr(self, "exchange_at_world_1", False))
        broadcast_flat_params(self.ps.flat_p, force=force)
        for b in self.model.buffers():
            if b.dtype.is_floating_point:
                broadcast_flat_params(b, force=force)
