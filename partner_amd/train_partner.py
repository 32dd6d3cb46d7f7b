"""One training iteration of the PARTNER detector itself (configs/waymo/voxelnet/waymo_partner_36epoch.py: VoxelNetV3 =
mean VFE -> SpMiddleResNetFHD -> 2 x SetBlock -> RPN -> E2ESWVoteHead), the counterpart of train.PolarPillarTrainStep for
the nuScenes polar-pillar model.

Reference call stack of the iteration: det3d/torchie/trainer/trainer.py:275-300 (batch_processor -> model(example,
return_loss=True) -> loss.backward() -> optimizer hooks), det3d/models/detectors/voxelnet.py:239-301 (forward),
det3d/torchie/apis/train.py:198-215 / det3d/solver/fastai_optim.py:155-171 (OneCycle Adam with decoupled weight decay),
det3d/torchie/trainer/hooks/optimizer.py:10-13 (clip_grad_norm_ 35).

The whole forward is recorded on the reverse-mode tape of autodiff.py (sparse_train / attention_train / swv_head_train plus the
RPN written here with the same primitives); the set criterion (csrc/e2e_loss.hip) returns the loss and the gradients of the
head tensors, the tape carries them back to every parameter, and the flat-buffer optimizer of train.py finishes the step.
All arithmetic is HIP kernels; BatchNorm layers use batch statistics and update their running buffers; dropout / drop-path of
the SetBlocks follow the rates the detector was built with (the counter-based generator of pn_dropout_f32)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import autodiff as ad
from . import hip, ops
from .attention_train import set_block_train
from .sparse_train import sp_middle_resnet_fhd_train
from .swv_head_train import e2e_swv_head_train
from .train import ParamStore, invalidate_inference_plans, one_cycle


def rpn_train(t: ad.Tape, neck: nn.Module, x: ad.Node, prefix="neck.") -> ad.Node:
    """RPN.forward (det3d/models/necks/rpn.py:144-159) in training mode on the tape: NHWC in, concatenated deblock outputs out"""
    P: Dict[str, ad.Node] = {name: t.param(p.data, prefix + name) for name, p in neck.named_parameters()}
    ups = []
    for i, blk in enumerate(neck.blocks):
        mods = list(blk._modules.values())
        bp = f"blocks.{i}."
        x = ad.conv2d(t, x, P[bp + "1.weight"], None, stride=mods[1].stride[0], pad=1)   # ZeroPad2d(1) + Conv2d(3, padding 0)
        x = ad.batchnorm2d(t, x, mods[2], P[bp + "2.weight"], P[bp + "2.bias"], relu=True)
        for k in range(4, len(mods), 3):
            x = ad.conv2d(t, x, P[f"{bp}{k}.weight"], None, stride=1, pad=1)
            x = ad.batchnorm2d(t, x, mods[k + 1], P[f"{bp}{k + 1}.weight"], P[f"{bp}{k + 1}.bias"], relu=True)
        j = i - neck._upsample_start_idx
        if j >= 0:
            de = neck.deblocks[j]
            dp = f"deblocks.{j}."
            if isinstance(de[0], nn.ConvTranspose2d):
                u = ad.conv_transpose2d(t, x, P[dp + "0.weight"])
            else:
                u = ad.conv2d(t, x, P[dp + "0.weight"], None, stride=de[0].stride[0], pad=0)
            ups.append(ad.batchnorm2d(t, u, de[1], P[dp + "1.weight"], P[dp + "1.bias"], relu=True))
    if not ups:
        return x
    return ups[0] if len(ups) == 1 else ad.concat_channels(t, ups, sum(u.v.shape[3] for u in ups))


class PartnerTrainStep:
    """``step(example)``: forward (training mode), set criterion, backward, gradient all-reduce (when torch.distributed is
    initialised with world_size > 1), clip + decoupled weight decay + Adam under the OneCycle schedule.  ``example`` carries the
    hard-voxel keys of the reference's collate (voxels (V,P,F), coordinates (V,4) [b,z,y,x], num_points (V,), num_voxels (B,),
    shape [[x,y,z]]) and ``global_box`` (B, M, 7[+2]+1).  Returns the reference's loss dict."""

    def __init__(self, model: nn.Module, total_steps: int, lr_max=0.003, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4, weight_decay=0.01,
                 max_norm=35.0, beta2=0.99, eps=1e-8, drop=None, attn_drop=None, drop_path=None, seed=0):
        for attr in ("reader", "backbone", "attns", "neck", "bbox_head"):
            if not hasattr(model, attr):
                raise NotImplementedError(f"PartnerTrainStep: the detector has no `{attr}` (expected VoxelNetV3)")
        self.model = model
        p0 = next(model.parameters())
        hip.require_device(p0)
        self.dev = p0.device
        self.ps = ParamStore(model, self.dev)
        self.sched = dict(total=total_steps, lr_max=lr_max, moms=tuple(moms), div=div_factor, pct=pct_start)
        self.wd, self.max_norm, self.beta2, self.eps = weight_decay, max_norm, beta2, eps
        # VoxelNetV3 builds its SetBlocks with drop = attn_drop = drop_path = 0.1 (voxelnet.py:192-199)
        self.drop = 0.1 if drop is None else float(drop)
        self.attn_drop = 0.1 if attn_drop is None else float(attn_drop)
        self.drop_path = 0.1 if drop_path is None else float(drop_path)
        self.seed = int(seed)
        self.iter = 0
        self.last_tape: Optional[ad.Tape] = None

    # ------------------------------------------------------------------------------------------
    def forward_backward(self, example, grad_scale=1.0):
        """forward + loss + backward; the gradients land in ``self.ps.flat_g`` (overwritten).  -> the loss dict of the head"""
        m = self.model
        hip.require_device(example["voxels"], example["coordinates"])
        batch = len(example["num_voxels"])
        shape = [int(v) for v in example["shape"][0]]
        t = ad.Tape()
        feats = m.reader(example["voxels"], example["num_points"])
        x = sp_middle_resnet_fhd_train(t, m.backbone, feats, example["coordinates"], batch, shape, prefix="backbone.")
        b, th, r, c = x.v.shape
        tok = ad.view(t, ad.transpose_hw(t, x, b, th, r, c), (b * r * th, c))       # range-major tokens
        for i, blk in enumerate(m.attns):
            if tuple(blk.patches_resolution) != (r, th):
                raise ValueError(f"SetBlock resolution {tuple(blk.patches_resolution)} does not match the encoder's BEV map {(r, th)}")
            tok = set_block_train(t, blk, tok, b, prefix=f"attns.{i}.", drop=self.drop, attn_drop=self.attn_drop, drop_path=self.drop_path,
                                  seed=self.seed * 7919 + self.iter * 16 + i)
        x = ad.transpose_hw(t, ad.view(t, tok, (b, r, th, c)), b, r, th, c)
        if getattr(m, "with_neck", m.neck is not None):
            x = rpn_train(t, m.neck, x)
        out = e2e_swv_head_train(t, m.bbox_head, x, prefix="bbox_head.")
        bx = out["boxes"].v
        nchw = lambda v: v.permute(0, 3, 1, 2)   # noqa: E731  (logical views, as the head returns them)
        pd = dict(pred_centers=nchw(out["pred_centers"].v), pred_vote_cls=nchw(out["pred_vote_cls"].v), hm=nchw(out["hm"].v), reg=nchw(bx[..., 0:2]),
                  height=nchw(bx[..., 2:3]), dim=nchw(bx[..., 3:6]), rot=nchw(bx[..., 6:8]))
        if "iou" in out:
            pd["iou"] = nchw(out["iou"].v)
        losses = m.bbox_head.loss(example, {"det_preds": [pd]})
        g = m.bbox_head.last_loss["grads"]
        seeds = [("hm", "d_hm"), ("boxes", "d_boxes"), ("pred_centers", "d_centers"), ("pred_vote_cls", "d_vote_cls"), ("iou", "d_iou")]
        for name, gname in seeds:
            if name in out and g.get(gname) is not None:
                gv = g[gname]
                if grad_scale != 1.0:
                    gv = self._scaled(gv, grad_scale)
                ad.accumulate(out[name], gv)
        t.backward()
        # parameter gradients into the flat buffer (parameters the graph never touches keep a zero gradient, as their .grad
        # stays None -> skipped by the reference's optimizer)
        hip.call("pn_fill_zero", self.ps.flat_g.data_ptr(), self.ps.flat_g.numel() * 4, hip.stream())
        for leaf in t.params:
            if leaf.g is not None:
                ops.add(self.ps.g[leaf.name].view(-1), leaf.g.contiguous().view(-1), out=self.ps.g[leaf.name].view(-1))
        self.last_tape = t
        return losses

    def _scaled(self, g: torch.Tensor, s: float) -> torch.Tensor:
        sc = torch.full((g.shape[-1],), float(s), dtype=torch.float32, device=g.device)
        y = torch.empty_like(g)
        hip.call("pn_scale_channels_f32", g.data_ptr(), sc.data_ptr(), g.numel(), g.shape[-1], y.data_ptr(), hip.stream())
        return y

    def optimizer_step(self):
        s = self.sched
        lr, beta1 = one_cycle(self.iter, s["total"], s["lr_max"], s["moms"], s["div"], s["pct"])
        total_norm = ops.grad_norm(self.ps.flat_g)
        ops.adam_step(self.ps.flat_p, self.ps.flat_g, self.ps.flat_m, self.ps.flat_v, self.iter + 1, lr, beta1, self.beta2, self.eps, self.wd,
                      total_norm=total_norm, max_norm=self.max_norm)
        self.iter += 1
        invalidate_inference_plans(self.model)   # the kernels updated the parameters behind the plans' packed copies (f32 and bf16)
        return total_norm

    def sync_initial_params(self):
        """rank 0's parameters (and floating-point buffers: BatchNorm running statistics) to every rank, once before the first step
        -- what DistributedDataParallel's constructor does in the reference (det3d/torchie/apis/train.py:330-336)"""
        from .dist_utils import broadcast_flat_params
        broadcast_flat_params(self.ps.flat_p)
        for b in self.model.buffers():
            if b.dtype.is_floating_point:
                broadcast_flat_params(b)

    def step(self, example):
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        losses = self.forward_backward(example, grad_scale=1.0 / world)
        if world > 1:
            from .dist_utils import GradExchange
            ex = GradExchange(self.ps.flat_g, [(0, self.ps.total)])
            ex.ready(0)
            ex.finish()
        self.optimizer_step()
        return losses
