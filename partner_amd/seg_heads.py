"""Segmentation head of the reference's nuScenes polar config (`super_tasks = ['det', 'seg']`): SingleConvHead
(det3d/models/seg_heads/seg_head.py:53-83, 176-195) and its loss object SegLoss (det3d/models/losses/seg_loss.py:8-22).
Secondary to the detection hot path: forward and per-point prediction run on the HIP kernels (eval); the loss (Lovasz softmax +
cross entropy) is not built."""
from __future__ import annotations

import torch
from torch import nn

from . import builder, hip, ops
from .builder import LOSSES, SEG_HEAD
from .nn_utils import PlanCache, eval_only


@LOSSES.register_module
class SegLoss(nn.Module):
    """parameter-free; kept so that the config's ``loss=dict(type='SegLoss', ignore=-1)`` builds"""

    def __init__(self, ignore=0):
        super().__init__()
        self.ignore = ignore

    def forward(self, outputs, labels):
        raise NotImplementedError("SegLoss (Lovasz softmax + cross entropy, seg_loss.py:19-22) is not built: segmentation training is out of scope")


@SEG_HEAD.register_module
class SingleConvHead(nn.Module):
    """seg_preds = Conv2d(in_channels, num_classes, kernel)(cat[x1, bilinear_up(x2 -> size of x1)]), x1 = BEV canvas, x2 = RPN output.

    With the config's kernel = 1 the convolution commutes with the interpolation, so the head runs as
    conv(x1; W[:, :C1]) + bias + bilinear_up(conv(x2; W[:, C1:])) and the (C1 + C2)-channel map at canvas resolution is never built."""

    def __init__(self, kernel=1, num_classes=16, in_channels=448, weight=1, loss=None):
        super().__init__()
        self.num_classes = num_classes
        self.conv = nn.Conv2d(in_channels, num_classes, kernel, padding=kernel // 2)
        self.weight = weight
        self.loss_func = builder.build_loss(loss) if loss is not None else None
        self._plan = PlanCache()

    def _build_plan(self, c1: int):
        if self.conv.kernel_size != (1, 1):
            raise NotImplementedError("SingleConvHead: only kernel=1 (the reference config) has a HIP path")
        w = self.conv.weight
        n = self.num_classes
        npad = (n + 3) // 4 * 4                      # the up-sampling kernel moves 16-byte channel groups
        wa = torch.zeros((npad, c1, 1, 1), dtype=w.dtype, device=w.device)
        wb = torch.zeros((npad, w.shape[1] - c1, 1, 1), dtype=w.dtype, device=w.device)
        bias = torch.zeros((npad,), dtype=w.dtype, device=w.device)
        wa[:n], wb[:n], bias[:n] = w[:, :c1].detach(), w[:, c1:].detach(), self.conv.bias.detach()
        return dict(c1=c1, npad=npad, a=ops.ConvLayer(wa, shift=bias, act=ops.ACT_NONE), b=ops.ConvLayer(wb, act=ops.ACT_NONE))

    def forward_nhwc(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        """x1 (B,H,W,C1), x2 (B,h,w,C2) NHWC -> logits (B,H,W,num_classes [padded to a multiple of 4]) NHWC"""
        hip.require_device(x1, x2)
        eval_only(self, "SingleConvHead")
        plan = self._plan.get(self, lambda: self._build_plan(x1.shape[3]))
        assert plan["c1"] == x1.shape[3] and x1.shape[3] + x2.shape[3] == self.conv.in_channels, "channel split does not match in_channels"
        out = plan["a"](x1)
        low = plan["b"](x2)
        b, h, w, c = low.shape
        hip.call("pn_bilinear_upsample_add_f32", low.data_ptr(), b, h, w, c, out.shape[1], out.shape[2], out.data_ptr(), hip.stream())
        return out

    def forward(self, x1, x2):
        """logical (B,C,H,W) tensors, as seg_head.py:75-83 -> {'seg_preds': (B, num_classes, H, W)}"""
        out = self.forward_nhwc(ops.to_nhwc(x1), ops.to_nhwc(x2))
        return {"seg_preds": out.permute(0, 3, 1, 2)[:, :self.num_classes]}

    def loss(self, example, preds_dicts, **kwargs):
        raise NotImplementedError("SingleConvHead.loss: segmentation training is out of scope (SegLoss is not built)")

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """per-point semantic labels (seg_head.py:176-195): 1 + argmax over the classes at every point's BEV cell
        (``example['valid_grid_ind'][i]``: (n_i, 3) [z, y, x]).  -> list of {token: labels (n_i,) int64}"""
        seg = preds_dicts["seg_preds"]
        hip.require_device(seg)
        assert seg.dim() == 4 and seg.stride(1) == 1, "seg_preds must be the channels-last view produced by forward"
        bsz, ncls, h, w = seg.shape
        cstride = seg.stride(3)                                  # padded channel count of the NHWC buffer
        out = []
        for i in range(bsz):
            gi = example["valid_grid_ind"][i]
            gi = (gi if torch.is_tensor(gi) else torch.as_tensor(gi)).to(seg.device).to(torch.int64).contiguous()
            labels = torch.empty((gi.shape[0],), dtype=torch.int64, device=seg.device)
            # the kernel reads `classes` logits with the buffer's pixel stride: pass the sample's base pointer and the real class count
            base = seg[i].permute(1, 2, 0)                       # (H, W, ncls) view, pixel stride = cstride
            assert base.stride(2) == 1 and base.stride(1) == cstride
            if cstride != ncls:                                  # padded buffer: compact once (small: H*W*ncls floats)
                base = base.contiguous()
            hip.call("pn_seg_point_labels", base.data_ptr(), h, w, ncls, gi.data_ptr(), gi.shape[0], labels.data_ptr(), hip.stream())
            meta = example["metadata"][i]
            token = meta["token"] if isinstance(meta, dict) and "token" in meta else i
            out.append({token: labels})
        return out
