"""Centre heads (registered under the reference's names).

Reference: det3d/models/bbox_heads/center_head.py:65-109,166-242 (CenterHead / SepHead),
center_head_parallel.py:27-59,70-196,199-284 (RangeStratified, CenterHeadSingle,
CenterHeadSinglePos), det3d/models/utils/norm.py:58-75 (RSNorm).

Forward = HIP launches only: fp32-MFMA convolutions with fused bias/activation, the stratified
GroupNorm kernel, and -- for CenterHeadSinglePos -- the position-conditioned calibration folded
to two constant (H,W,C) maps at plan-build time (they do not depend on the input).
"""
from __future__ import annotations

import copy
import ctypes as C
import logging
import os
from typing import Dict, List

import torch
from torch import nn

from . import hip, ops
from .routes import R
from .builder import BBOX_HEADS
from .nn_utils import PlanCache, RSNorm, Sequential, eval_only



class RangeStratified(nn.Module):
    """parameters of the range-stratified 3x3 convolution + GroupNorm (center_head_parallel.py:27-59)"""

    def __init__(self, kernel, nheads, ngroups, inchannels, outchannels, act="ReLU"):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(inchannels * ngroups * nheads, outchannels * ngroups * nheads, kernel, groups=ngroups * nheads),
            nn.GroupNorm(ngroups * nheads, outchannels * ngroups * nheads),
            nn.ReLU(inplace=True))
        self.kernel, self.nheads, self.ngroups = tuple(kernel), nheads, ngroups
        self.inchannels, self.outchannels = inchannels, outchannels


class SepHead(nn.Module):
    """per-task separate heads of the plain CenterHead (center_head.py:65-109)"""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=False, init_bias=-2.19, **kwargs):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            fc = Sequential()
            for _ in range(num_conv - 1):
                fc.add(nn.Conv2d(in_channels, head_conv, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True))
                fc.add(nn.ReLU())
            fc.add(nn.Conv2d(head_conv, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True))
            if "hm" in head:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                        nn.init.constant_(m.bias, 0)
            setattr(self, head, fc)


def _dense_weight(conv: nn.Conv2d) -> torch.Tensor:
    """weight of a grouped convolution as the block-diagonal dense (Cout, Cin, KH, KW) tensor (groups == 1: the weight itself)"""
    w = conv.weight.detach()
    g = conv.groups
    if g == 1:
        return w
    cout, cin_g = w.shape[0], w.shape[1]
    dense = torch.zeros((cout, cin_g * g) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
    cg = cout // g
    for k in range(g):
        dense[k * cg:(k + 1) * cg, k * cin_g:(k + 1) * cin_g] = w[k * cg:(k + 1) * cg]
    return dense


def _conv_bias(conv: nn.Conv2d, act, **kw) -> ops.ConvLayer:
    return ops.ConvLayer(conv.weight, stride=conv.stride[0], pad=conv.padding[0], groups=conv.groups, shift=conv.bias,
                         act=act, **kw)


@BBOX_HEADS.register_module
class CenterHead(nn.Module):
    """Plain CenterPoint head: shared 3x3 conv + ReLU, then per task / per head
    (conv3x3 + ReLU) x (n-1) + conv3x3."""

    def __init__(self, in_channels=[128, ], tasks=[], dataset="nuscenes", weight=0.25, code_weights=[], common_heads=dict(),
                 logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2, dcn_head=False, voxel_shape="cuboid"):
        super().__init__()
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.code_weights, self.weight, self.dataset, self.voxel_shape = code_weights, weight, dataset, voxel_shape
        self.in_channels, self.num_classes = in_channels, num_classes
        self.box_n_dim = 9 if "vel" in common_heads else 7
        self.use_direction_classifier = False
        self.logger = logger or logging.getLogger("CenterHead")
        self.logger.info(f"num_classes: {num_classes}")
        if dcn_head:
            raise NotImplementedError("dcn_head=True (deformable convolution head) is outside the hot path (SURVEY.md 2.1)")
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, kernel_size=3, padding=1, bias=True),
                                         nn.ReLU(inplace=True))
        self.tasks = nn.ModuleList()
        for ncls in num_classes:
            heads = copy.deepcopy(common_heads)
            heads.update(dict(hm=(ncls, num_hm_conv)))
            self.tasks.append(SepHead(share_conv_channel, heads, bn=True, init_bias=init_bias, final_kernel=3))
        self._plan = PlanCache()
        self.logger.info("Finish CenterHead Initialization")

    def _build_plan(self):
        plan = dict(shared=_conv_bias(self.shared_conv[0], ops.ACT_RELU), tasks=[])
        for task in self.tasks:
            heads = {}
            for name in task.heads:
                convs = [m for m in getattr(task, name)._modules.values() if isinstance(m, nn.Conv2d)]
                heads[name] = [_conv_bias(c, ops.ACT_RELU if k < len(convs) - 1 else ops.ACT_NONE) for k, c in enumerate(convs)]
            plan["tasks"].append(heads)
        return plan

    def forward(self, x, *kwargs):
        hip.require_device(x)
        eval_only(self, type(self).__name__)
        plan = self._plan.get(self, self._build_plan)
        xs = plan["shared"](ops.to_nhwc(x))
        rets: List[Dict[str, torch.Tensor]] = []
        for heads in plan["tasks"]:
            d = {}
            for name, layers in heads.items():
                y = xs
                for layer in layers:
                    y = layer(y)
                d[name] = ops.as_nchw(y)
            rets.append(d)
        return {"det_preds": rets}

    def loss(self, example, preds_dicts, **kwargs):
        """Forward value of the CenterPoint loss (center_head.py:248-288) on the HIP kernel.  Returns the
        reference's dict of per-task lists.  (No autograd graph: the backward kernels of the training step
        are not built yet)"""
        import ctypes as C
        from collections import defaultdict
        lib = hip.load()
        rets = defaultdict(list)
        for t, pd in enumerate(preds_dicts["det_preds"]):
            hm = pd["hm"]
            hip.require_device(hm)
            b, ncls, h, w = hm.shape
            order = ["reg", "height", "dim"] + (["vel"] if "vel" in pd else []) + ["rot"]
            srcs = [pd[k] for k in order]
            for s_ in [hm] + srcs:
                assert s_.stride(1) == 1 and s_.stride(2) == w * s_.stride(3), "head tensors must be channels-last views"
            ndim = sum(s_.shape[1] for s_ in srcs)
            dev = hm.device
            tgt = example["hm"][t].to(dev).float().contiguous()
            ind = example["ind"][t].to(dev).long().contiguous()
            mask = example["mask"][t].to(dev).to(torch.uint8).contiguous()
            cat = example["cat"][t].to(dev).long().contiguous()
            anno = example["anno_box"][t].to(dev).float().contiguous()
            ad = anno.shape[-1]
            sel = list(range(ndim)) if "vel" in pd else [0, 1, 2, 3, 4, 5, ad - 2, ad - 1]
            cw = torch.tensor(list(self.code_weights)[:ndim], dtype=torch.float32, device=dev)
            out = torch.empty((4 + ndim,), dtype=torch.float32, device=dev)
            wsb = lib.pn_center_loss_workspace_bytes()
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            hip.call("pn_center_loss_fwd", hm.data_ptr(), hm.stride(3), tgt.data_ptr(), b, ncls, h, w,
                     (C.c_void_p * len(srcs))(*[s_.data_ptr() for s_ in srcs]), (C.c_int * len(srcs))(*[s_.stride(3) for s_ in srcs]),
                     (C.c_int * len(srcs))(*[s_.shape[1] for s_ in srcs]), len(srcs), ind.data_ptr(), mask.data_ptr(), cat.data_ptr(),
                     anno.data_ptr(), ad, (C.c_int * ndim)(*sel), ind.shape[1], ndim, cw.data_ptr(), float(self.weight),
                     out.data_ptr(), ws.data_ptr(), wsb, hip.stream())
            rets["det_loss"].append(out[0])
            rets["hm_loss"].append(out[1].detach().cpu())
            rets["loc_loss"].append(out[2])
            rets["loc_loss_elem"].append(out[4:].detach().cpu())
            rets["num_positive"].append(out[3])
        return rets

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """decode + rotated NMS on the device (center_head.py:404-460, 350-402, 462-577): the multi-class rotate_nms_pcdet path and
        the per-class one (test_cfg.per_class_nms, batched_nms_rotated); test_cfg.double_flip merges groups of four flipped copies first
        (center_head.py:289-346, 425-427), test_cfg.stateful_nms lets the previous sectors' detections (kwargs prev_dets / sec_id) compete
        (center_head.py:466, 486-509); panoptic fusion raises NotImplementedError (segmentation is out of scope).  Returns the reference's list (one dict per sample) of
        'box3d_lidar' (n, 9|7), 'scores', 'label_preds', 'metadata'."""
        import ctypes as C
        lib = hip.load()
        get = (lambda k, d=None: test_cfg.get(k, d)) if hasattr(test_cfg, "get") else (lambda k, d=None: getattr(test_cfg, k, d))
        if get("panoptic", False):
            raise NotImplementedError("predict: test_cfg.panoptic (instance ids for the segmentation super-task) is not built")
        double_flip = bool(get("double_flip", False))   # center_head.py:412, 425-427
        stateful = bool(get("stateful_nms", False))     # center_head.py:466, 486-501, 507-509
        if stateful and (double_flip or kwargs.get("device_only", False)):
            raise NotImplementedError("predict: stateful NMS is not combined with double flip / device_only outputs")
        per_class = bool(get("per_class_nms", False))   # batched_nms_rotated of the nuScenes configs (center_head.py:514-518)
        if kwargs.get("device_only", False) and len(preds_dicts["det_preds"]) != 1:
            raise NotImplementedError("predict(device_only=True) supports a single task")
        prev_dets = kwargs.get("prev_dets") if stateful else None
        sec_id = int(kwargs.get("sec_id", 0))
        nms = get("nms")
        nget = (lambda k: nms[k]) if isinstance(nms, dict) else (lambda k: getattr(nms, k))
        pre_max, post_max, iou_thr = int(nget("nms_pre_max_size")), int(nget("nms_post_max_size")), float(nget("nms_iou_threshold"))
        if per_class:
            pre_max = 4096   # the reference passes every candidate to the per-class NMS; the device path takes the best 4096
        if pre_max > 4096:
            raise ValueError(f"predict: nms_pre_max_size = {pre_max} exceeds the device NMS capacity of 4096 candidates per sample "
                             "(INTEGRATION.md, limits)")
        pcr = list(get("post_center_limit_range"))
        assert len(pcr) == 6, "predict: post_center_limit_range must have 6 entries"
        osf, vs, pr = get("out_size_factor"), get("voxel_size"), get("pc_range")
        cyl = int(self.voxel_shape == "cylinder")
        if not cyl:
            pr = [float(v) for v in example["pc_range"][0]][:2] if "pc_range" in example else pr  # ref_pc_range of the reference
        rets = []
        for task_id, pd in enumerate(preds_dicts["det_preds"]):
            hm = pd["hm"]
            hip.require_device(hm)
            b, ncls, h, w = hm.shape
            names = ["hm", "reg", "height", "dim", "rot"] + (["vel"] if "vel" in pd else [])
            for k in names:
                t = pd[k]
                assert t.stride(1) == 1 and t.stride(2) == w * t.stride(3), "head tensors must be channels-last views"
            nb = 9 if "vel" in pd else 7
            dev = hm.device
            vel = pd.get("vel")
            decode_fn = "pn_center_decode_nms_f32"
            if double_flip:
                # the batch holds groups of four clouds [original, y-flip, x-flip, both]: flip the maps back, fix the signs, average
                # (double_flip_decode, center_head.py:289-346); the merged hm / dim are probabilities / sizes already
                if b % 4 != 0:
                    raise ValueError(f"predict(double_flip): the batch ({b}) must hold groups of four flipped copies")
                b //= 4
                f32 = dict(dtype=torch.float32, device=dev)
                mg = dict(hm=torch.empty((b, h, w, ncls), **f32), reg=torch.empty((b, h, w, 2), **f32), height=torch.empty((b, h, w, 1), **f32),
                          dim=torch.empty((b, h, w, 3), **f32), rot=torch.empty((b, h, w, 2), **f32))
                if vel is not None:
                    mg["vel"] = torch.empty((b, h, w, 2), **f32)
                hip.call("pn_double_flip_merge_f32", hm.data_ptr(), hm.stride(3), ncls, pd["reg"].data_ptr(), pd["reg"].stride(3),
                         pd["height"].data_ptr(), pd["height"].stride(3), pd["dim"].data_ptr(), pd["dim"].stride(3), pd["rot"].data_ptr(),
                         pd["rot"].stride(3), hip.ptr(vel), 0 if vel is None else vel.stride(3), b, h, w, mg["hm"].data_ptr(), mg["reg"].data_ptr(),
                         mg["height"].data_ptr(), mg["dim"].data_ptr(), mg["rot"].data_ptr(), hip.ptr(mg.get("vel")), hip.stream())
                pd = {k: v.permute(0, 3, 1, 2) for k, v in mg.items()}
                hm, vel = pd["hm"], pd.get("vel")
                decode_fn = "pn_center_decode_nms_merged_f32"
            out_boxes = torch.empty((b, post_max, nb), dtype=torch.float32, device=dev)
            out_scores = torch.empty((b, post_max), dtype=torch.float32, device=dev)
            out_labels = torch.empty((b, post_max), dtype=torch.int64, device=dev)
            out_cells = torch.empty((b, post_max), dtype=torch.int32, device=dev)
            out_count = torch.empty((b,), dtype=torch.int32, device=dev)
            wsb = lib.pn_center_decode_nms_workspace_bytes(b, h * w, nb, pre_max, post_max)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            if stateful:
                # the previous sectors' detections compete in this sector's NMS; this sector's candidates are rotated into the sweep's
                # frame first; the output may hold nms_post_max_size * (sec_id + 1) boxes
                interval = float(get("interval")) if sec_id > 0 else 0.0
                angle = (interval * sec_id if cyl else 2 * 3.141592653589793 / interval * sec_id) if sec_id > 0 else 0.0
                prev = None if prev_dets is None else prev_dets[task_id]
                pcap = 0 if prev is None else max([int(d["scores"].numel()) for d in prev] + [0])
                pmax = post_max * (sec_id + 1)
                f32 = dict(dtype=torch.float32, device=dev)
                pb = torch.zeros((b, max(pcap, 1), nb), **f32)
                ps = torch.zeros((b, max(pcap, 1)), **f32)
                pl = torch.zeros((b, max(pcap, 1)), dtype=torch.int64, device=dev)
                pc = torch.zeros((b,), dtype=torch.int32, device=dev)
                if pcap:
                    for i, d in enumerate(prev):
                        n = int(d["scores"].numel())
                        pb[i, :n].copy_(d["box3d_lidar"])
                        ps[i, :n].copy_(d["scores"])
                        pl[i, :n].copy_(d["label_preds"])
                        pc[i] = n
                out_boxes = torch.empty((b, pmax, nb), **f32)
                out_scores = torch.empty((b, pmax), **f32)
                out_labels = torch.empty((b, pmax), dtype=torch.int64, device=dev)
                out_cells = torch.empty((b, pmax), dtype=torch.int32, device=dev)
                wsb = lib.pn_center_decode_nms_workspace_bytes(b, h * w + pcap, nb, pre_max, pmax)
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                hip.call("pn_center_decode_nms_stateful_f32", hm.data_ptr(), hm.stride(3), ncls, pd["reg"].data_ptr(), pd["reg"].stride(3),
                         pd["height"].data_ptr(), pd["height"].stride(3), pd["dim"].data_ptr(), pd["dim"].stride(3), pd["rot"].data_ptr(),
                         pd["rot"].stride(3), hip.ptr(vel), 0 if vel is None else vel.stride(3), b, h, w, cyl, float(osf) * float(vs[0]),
                         float(osf) * float(vs[1]), float(pr[0]), float(pr[1]), int(bool(get("rectify", False))), float(get("score_threshold")),
                         (C.c_float * 6)(*[float(v) for v in pcr]), iou_thr, int(per_class), pre_max, pmax, float(angle), pb.data_ptr(), ps.data_ptr(),
                         pl.data_ptr(), pc.data_ptr(), pcap, out_boxes.data_ptr(), out_scores.data_ptr(), out_labels.data_ptr(), out_cells.data_ptr(),
                         out_count.data_ptr(), ws.data_ptr(), wsb, hip.stream())
                counts = out_count.cpu().tolist()
                rets.append([dict(box3d_lidar=out_boxes[i, :n], scores=out_scores[i, :n], label_preds=out_labels[i, :n], cells=out_cells[i, :n])
                             for i, n in enumerate(counts)])
                continue
            hip.call(decode_fn, hm.data_ptr(), hm.stride(3), ncls, pd["reg"].data_ptr(), pd["reg"].stride(3),
                     pd["height"].data_ptr(), pd["height"].stride(3), pd["dim"].data_ptr(), pd["dim"].stride(3), pd["rot"].data_ptr(),
                     pd["rot"].stride(3), hip.ptr(vel), 0 if vel is None else vel.stride(3), b, h, w, cyl, float(osf) * float(vs[0]),
                     float(osf) * float(vs[1]), float(pr[0]), float(pr[1]), int(bool(get("rectify", False))), float(get("score_threshold")),
                     (C.c_float * 6)(*[float(v) for v in pcr]), iou_thr, int(per_class), pre_max, post_max, out_boxes.data_ptr(), out_scores.data_ptr(),
                     out_labels.data_ptr(), out_cells.data_ptr(), out_count.data_ptr(), ws.data_ptr(), wsb, hip.stream())
            if sec_id > 0:
                # sector streaming: the sector's boxes back into the sweep's frame (center_head.py:533-545)
                interval = float(get("interval"))
                angle = interval * sec_id if cyl else 2 * 3.141592653589793 / interval * sec_id
                hip.call("pn_rotate_boxes_f32", out_boxes.data_ptr(), out_count.data_ptr(), b, post_max, nb, float(angle), hip.stream())
            if kwargs.get("device_only", False):
                # fixed-size outputs + device counts: nothing leaves the stream (hipGraph capturable); one task only
                return dict(box3d_lidar=out_boxes, scores=out_scores, label_preds=out_labels, cells=out_cells, count=out_count)
            counts = out_count.cpu().tolist()  # the one host sync of the call: the API returns exact-size tensors
            rets.append([dict(box3d_lidar=out_boxes[i, :n], scores=out_scores[i, :n], label_preds=out_labels[i, :n], cells=out_cells[i, :n])
                         for i, n in enumerate(counts)])
        if stateful:
            return rets   # per task, per sample, unmerged: the next sector's prev_dets (center_head.py:439-440)
        metas = example.get("metadata", [None] * len(rets[0])) if isinstance(example, dict) else [None] * len(rets[0])
        if double_flip:
            metas = list(metas)[::4] if len(metas) >= 4 * len(rets[0]) else metas   # meta_list[:4 * batch:4]
        ret_list = []
        for i in range(len(rets[0])):
            flag, labels = 0, []
            for j, ncls in enumerate(self.num_classes[:len(rets)]):
                labels.append(rets[j][i]["label_preds"] + flag)
                flag += ncls
            ret_list.append(dict(box3d_lidar=torch.cat([r[i]["box3d_lidar"] for r in rets]), scores=torch.cat([r[i]["scores"] for r in rets]),
                                 label_preds=torch.cat(labels), cells=torch.cat([r[i]["cells"] for r in rets]), metadata=metas[i]))
        return ret_list


@BBOX_HEADS.register_module
class CenterHeadSingle(CenterHead):
    """Single-group merged heads with RSNorm / RangeStratified / GroupNorm (center_head_parallel.py:70-196)."""

    def __init__(self, in_channels=[128, ], tasks=[], dataset="nuscenes", weight=0.25, code_weights=[], common_heads=dict(),
                 logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2, dcn_head=False, voxel_shape="cuboid",
                 act="ReLU"):
        super().__init__(in_channels, tasks, dataset, weight, code_weights, common_heads, logger, init_bias,
                         share_conv_channel, num_hm_conv, dcn_head, voxel_shape)
        if act != "ReLU":
            raise NotImplementedError("only act='ReLU' is supported (the reference marks 'Mish' as not working)")
        self.num_heads = len(self.num_classes)
        self.common_heads = common_heads
        self.heads = copy.deepcopy(common_heads)
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, kernel_size=3, padding=1, bias=True),
                                         RSNorm(1, 4, share_conv_channel), nn.ReLU(inplace=True))
        self.tasks = None
        head_conv, k = 64, 3
        for head, (classes, num_conv) in common_heads.items():
            fc = Sequential()
            if "reg" in head:
                fc.add(RangeStratified((3, 3), 1, 8, share_conv_channel, head_conv, act))
                fc.add(nn.Conv2d(head_conv, classes, kernel_size=1, bias=True))
            else:
                n = len(head.split("_")) if "_" in head else 1
                for _ in range(num_conv - 1):
                    fc.add(nn.Conv2d(share_conv_channel, head_conv, kernel_size=k, stride=1, padding=k // 2, bias=True, groups=n))
                    fc.add(nn.GroupNorm(head_conv, head_conv))
                    fc.add(nn.ReLU(inplace=True))
                fc.add(nn.Conv2d(head_conv, classes * n, kernel_size=k, stride=1, padding=k // 2, bias=True, groups=n))
            setattr(self, head, fc)
        self.hm = Sequential()
        self.heads.update(dict(hm=(sum(self.num_classes), num_hm_conv)))
        for _ in range(num_hm_conv - 1):
            self.hm.add(nn.Conv2d(share_conv_channel, head_conv, kernel_size=k, stride=1, padding=k // 2, bias=True))
            self.hm.add(nn.GroupNorm(head_conv, head_conv))
            self.hm.add(nn.ReLU(inplace=True))
        self.hm.add(nn.Conv2d(head_conv, sum(self.num_classes), kernel_size=k, stride=1, padding=k // 2, bias=True))
        self.logger.info("Finish CenterHead Initialization")

    # ---------------------------------------------------------------------------------------
    def _branch_plan(self, fc: Sequential):
        """-> list of steps: ('conv', ConvLayer) | ('gn', channel_groups, strata, gamma, beta, eps)"""
        steps = []
        for m in fc._modules.values():
            if isinstance(m, RangeStratified):
                conv, gn = m.conv[0], m.conv[1]
                steps.append(("conv", ops.ConvLayer(conv.weight, stride=1, pad=1, range_strata=m.ngroups * m.nheads,
                                                    shift=conv.bias, act=ops.ACT_NONE)))
                steps.append(("gn", m.nheads, m.ngroups, gn.weight.detach(), gn.bias.detach(), gn.eps))
            elif isinstance(m, nn.Conv2d):
                steps.append(("conv", _conv_bias(m, ops.ACT_NONE)))
            elif isinstance(m, nn.GroupNorm):
                steps.append(("gn", m.num_groups, 1, m.weight.detach(), m.bias.detach(), m.eps))
        return steps

    def _build_plan(self):
        rs = self.shared_conv[1]
        plan = dict(shared=_conv_bias(self.shared_conv[0], ops.ACT_NONE),
                    shared_gn=(rs.num_heads, rs.num_groups, rs.groupnorm.weight.detach(), rs.groupnorm.bias.detach(),
                               rs.groupnorm.eps),
                    branches={name: self._branch_plan(getattr(self, name)) for name in self.heads})
        plan["merged"] = self._merge_branches(plan["branches"])
        plan["fused"] = self._build_fused()
        return plan

    # ---- fused execution: 5 launches instead of ~26 (convolutions + 15 GroupNorm passes) ----------
    def _build_fused(self):
        """Layers of the fused path, or None when the head does not have the shape the fused kernels cover: every branch
        = [3x3 conv (or RangeStratified) -> GroupNorm-family -> ReLU -> last conv], 64 hidden channels.
        Launch 1: shared conv, its epilogue emits the RSNorm statistics partials.  Launch 2: RSNorm + ReLU (+ calibration) ->
        [xs | x_hm], folding the partials in its prologue.  Launch 3: ALL first-stage branch convolutions as one multi-job launch,
        each emitting its GroupNorm statistics partials; 3b: one small launch folds them into affine tables.  Launch 4: ALL last
        convolutions as one multi-job launch that applies GroupNorm + ReLU while loading its input."""
        rs = self.shared_conv[1]
        if not isinstance(rs, RSNorm) or rs.num_heads != 1:
            return None
        c_sh = self.shared_conv[0].out_channels
        if c_sh % 32 != 0 or c_sh > 128:
            return None
        first, last, raws = [], [], []
        for name in self.heads:
            mods = list(getattr(self, name)._modules.values())
            if isinstance(mods[0], RangeStratified):
                if len(mods) != 2 or mods[0].nheads != 1:
                    return None
                conv, gn, fin = mods[0].conv[0], mods[0].conv[1], mods[1]
                cmid = mods[0].outchannels
                lay = ops.ConvLayer(conv.weight, stride=1, pad=1, range_strata=mods[0].ngroups, shift=conv.bias, act=ops.ACT_NONE)
                st = dict(strata=1, channel_groups=1, gamma=gn.weight.detach().float().contiguous(), beta=gn.bias.detach().float().contiguous(), eps=gn.eps,
                          affine_strata=mods[0].ngroups)
            else:
                if len(mods) != 4 or not (isinstance(mods[0], nn.Conv2d) and isinstance(mods[1], nn.GroupNorm) and isinstance(mods[3], nn.Conv2d)):
                    return None
                conv, gn, fin = mods[0], mods[1], mods[3]
                cmid = conv.out_channels
                if gn.num_groups != gn.num_channels or conv.kernel_size != (3, 3) or conv.padding != (1, 1):
                    return None
                lay = ops.ConvLayer(_dense_weight(conv), stride=1, pad=1, shift=conv.bias, act=ops.ACT_NONE)
                st = dict(strata=1, channel_groups=cmid, gamma=gn.weight.detach().float().contiguous(), beta=gn.bias.detach().float().contiguous(), eps=gn.eps,
                          affine_strata=1)
            if cmid % 32 != 0 or cmid > 128 or fin.kernel_size[0] != fin.kernel_size[1]:
                return None
            first.append((name, lay, st, cmid))
            raws.append(dict(w=(conv.weight if isinstance(mods[0], RangeStratified) else _dense_weight(conv)).detach().float().contiguous(),
                             bias=None if conv.bias is None else conv.bias.detach().float().contiguous(),
                             strata=mods[0].ngroups if isinstance(mods[0], RangeStratified) else 0))
            last.append((name, ops.ConvLayer(_dense_weight(fin), stride=1, pad=fin.padding[0], shift=fin.bias, act=ops.ACT_NONE), fin.out_channels))
        return dict(shared=_conv_bias(self.shared_conv[0], ops.ACT_NONE), first=first, last=last, c_sh=c_sh, first_raw=raws, chain={},
                    rs=(rs.num_groups, rs.groupnorm.weight.detach().float().contiguous(), rs.groupnorm.bias.detach().float().contiguous(), rs.groupnorm.eps))

    def _fused_ok(self, fz, b, h, w) -> bool:
        if fz is None:
            return False
        s = fz["rs"][0]
        if w % s != 0 or (w // s) % 32 != 0 or b * max(s, 8) >= 1024:
            return False
        for name, lay, st, cmid in fz["first"]:
            if lay.range_strata > 1 and (w % lay.range_strata != 0 or (b > 1 and (h * (w // lay.range_strata)) % 32 != 0)):
                return False
        return b == 1 or (h * w) % 64 == 0

    # ---- r4: the first-stage branches in the Winograd domain (csrc/conv_wchain.hip, pn_conv2d_wino24_chain_head_f32) ----------------------
    def _chain_head_plan(self, fz, b, h, w, dev, calibrated: bool):
        """The branch convolutions 64 -> 64 as chained F(2,3)xF(4,3) launches on the TRANSPOSED map (the Winograd axis is the azimuth, the
        frame's rows are the range positions -- a RangeStratified branch then takes one weight set per row stratum): the RSNorm pass writes
        planes instead of the map, the launches write the raw branch maps and the GroupNorm statistics partials, one small launch folds
        those into the affine tables the last convolutions apply on load.  None when a shape does not fit (the tiled multi-job launch runs)."""
        key = (b, h, w, calibrated)
        if key in fz["chain"]:
            return fz["chain"][key]
        plan = None
        lib = hip.load()
        c_sh = fz["c_sh"]
        ok = R.conv_chain and R.head_chain and c_sh % 32 == 0 and h % 4 == 0 and w % 2 == 0 and lib.pn_wino4_planes_floats(b, w, h, c_sh) > 0
        if ok:
            # launches: every stratified branch alone; the heat-map branch alone when it reads the calibrated copy; the rest together
            groups, rest = [], []
            for (name, lay, st, cmid), raw in zip(fz["first"], fz["first_raw"]):
                if raw["strata"] > 1:
                    groups.append(dict(src="xs", members=[(name, st, cmid, raw)], strata=raw["strata"]))
                elif name == "hm" and calibrated:
                    groups.append(dict(src="hm", members=[(name, st, cmid, raw)], strata=0))
                else:
                    rest.append((name, st, cmid, raw))
            if rest:
                groups.insert(0, dict(src="xs", members=rest, strata=0))
            cm_tot, off = sum(cm for *_, cm in fz["first"]), 0
            offsets, tile_rows = {}, None
            # fixed limits of the entry points this plan feeds (conv_wchain.hip kChainMultiJobs = 4 launches' jobs; norm.hip kHeadFinJobs = 8
            # branches, 8 strata, 256 channels per branch): a head beyond them takes the tiled multi-job launch, as the docstring says
            if (len(groups) > 4 or len(fz["first"]) > 8 or any(g["strata"] > 8 for g in groups)
                    or any(cm > 256 for *_, cm in fz["first"])):
                ok, groups = False, []
            for g in groups:
                cout = sum(m[2] for m in g["members"])
                S = g["strata"]
                d = ops.ConvDesc(b, h, w, c_sh, cout, 1, 3, 3, 1, 1, 1, c_sh, 0, cm_tot, off, ops.ACT_NONE, 0, S, 0, 0, 0)
                d.transpose_hw = 1
                nstat = lib.pn_conv_wino24_chain_stat_floats(C.byref(d))
                if nstat == 0 or cout % 32:
                    ok = False
                    break
                tr = lib.pn_conv_wino24_chain_stat_tile_rows(C.byref(d))
                tile_rows = tr if tile_rows is None else tile_rows
                ok = ok and tr == tile_rows
                nw = lib.pn_conv_wino24_packed_weight_floats(cout, c_sh)
                sets = max(S, 1)
                packed = torch.empty(nw * sets, dtype=torch.float32, device=dev)
                keep = []
                if S > 1:      # one member: weight (S * cmid, c_sh, 3, 3), bias (S * cmid)
                    wr = g["members"][0][3]["w"]
                    cm = g["members"][0][2]
                    for k in range(S):
                        wt = wr[k * cm:(k + 1) * cm].transpose(2, 3).contiguous()
                        keep.append(wt)
                        hip.call("pn_pack_conv_weight_wino24_f32", wt.data_ptr(), cout, c_sh, packed.data_ptr() + 4 * nw * k, hip.stream())
                    bias = g["members"][0][3]["bias"]
                else:
                    wt = torch.cat([m[3]["w"] for m in g["members"]], 0).transpose(2, 3).contiguous()
                    keep.append(wt)
                    hip.call("pn_pack_conv_weight_wino24_f32", wt.data_ptr(), cout, c_sh, packed.data_ptr(), hip.stream())
                    bs = [m[3]["bias"] if m[3]["bias"] is not None else torch.zeros(m[2], dtype=torch.float32, device=dev) for m in g["members"]]
                    bias = torch.cat(bs).contiguous()
                g.update(desc=d, packed=packed, bias=bias, cout=cout, nstat=int(nstat), off=off, keep=keep)
                o2 = 0
                for name, st, cmid, raw in g["members"]:
                    offsets[name] = (off + o2, o2, g)
                    o2 += cmid
                off += cout
            if ok:
                torch.cuda.current_stream().synchronize()      # (the transposed weight copies are temporaries of the packing launches)
                for g in groups:
                    g.pop("keep")
                plan = dict(groups=groups, offsets=offsets, cm_tot=cm_tot, tile_rows=int(tile_rows), planes=int(lib.pn_wino4_planes_floats(b, w, h, c_sh)))
        fz["chain"][key] = plan
        return plan

    def _first_stage_chained(self, fz, cp, raw, mul, add):
        """raw: the shared convolution's output stored TRANSPOSED (b, w, h, c); mul / add likewise (w, h, c)
        -> (mid, {name: (table, strata, cmid, channel offset in mid)})"""
        lib = hip.load()
        b, w, h, c_sh = raw.shape
        dev, st = raw.device, hip.stream()
        f32 = dict(dtype=torch.float32, device=dev)
        s_rs, g_rs, b_rs, eps_rs = fz["rs"]
        planes_xs = torch.empty(cp["planes"], **f32)
        planes_hm = torch.empty(cp["planes"], **f32) if mul is not None else None
        nbytes = lib.pn_groupnorm_workspace_bytes(b, 1, s_rs)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        hip.call("pn_groupnorm_strat_planes_f32", raw.data_ptr(), b, h, w, c_sh, c_sh, 0, 1, s_rs, hip.ptr(g_rs), hip.ptr(b_rs), float(eps_rs), ops.ACT_RELU,
                 hip.ptr(mul), hip.ptr(add), 2, planes_xs.data_ptr(), hip.ptr(planes_hm), ws.data_ptr(), nbytes, st)
        mid = torch.empty((b, h, w, cp["cm_tot"]), **f32)
        prof = ops.S.profiler
        parts = [torch.empty(g["nstat"], **f32) for g in cp["groups"]]
        cjobs = (hip.ChainHeadJob * len(parts))()
        for cj, g, part in zip(cjobs, cp["groups"], parts):
            src = planes_hm if (g["src"] == "hm" and planes_hm is not None) else planes_xs
            g["desc"].frames_in_flight = ops.S.frames_in_flight
            cj.desc = C.pointer(g["desc"])
            cj.planes_in, cj.packed_w24, cj.scale, cj.shift = src.data_ptr(), g["packed"].data_ptr(), None, hip.ptr(g["bias"])
            cj.planes_out, cj.out_nhwc, cj.stat_partials = None, mid.data_ptr(), part.data_ptr()
        if prof is not None:
            ev = prof.begin(st)
        # one launch for every branch group (apart: 128 - 384 short blocks each, the chip half empty between them)
        hip.call("pn_conv2d_wino24_chain_head_multi_f32", cjobs, len(parts), st)
        if prof is not None:
            flops = 2.0 * b * h * w * cp["cm_tot"] * c_sh * 9
            prof.end(ev, flops, st, tag=f"{h}x{w} {c_sh}->{cp['cm_tot']} k3 F(2,3)xF(4,3) head", issued=flops / 3.0)
        jobs = (hip.HeadStatJob * len(fz["first"]))()
        tabs = {}
        keep = []
        for k, (name, lay, stt, cmid) in enumerate(fz["first"]):
            moff, goff, g = cp["offsets"][name]
            gi = cp["groups"].index(g)
            S = g["strata"] if g["strata"] > 1 else 1
            tab = torch.empty((b, S, cmid, 2), **f32)
            j = jobs[k]
            j.partials, j.cout_total, j.channel_offset, j.channels, j.strata = parts[gi].data_ptr(), g["cout"], goff, cmid, g["strata"]
            j.gamma, j.beta, j.eps, j.table = hip.ptr(stt["gamma"]), hip.ptr(stt["beta"]), float(stt["eps"]), tab.data_ptr()
            tabs[name] = (tab, S, cmid, moff)
            keep.append(tab)
        hip.call("pn_wino24_chain_head_finalize_f32", jobs, len(fz["first"]), b, w, h, cp["tile_rows"], st)
        return mid, tabs

    def _forward_fused(self, plan, x: torch.Tensor):
        fz = plan["fused"]
        b, h, w, _ = x.shape
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        c_sh = fz["c_sh"]
        mul, add = self._calibration(plan, x[..., :c_sh])
        s_rs, g_rs, b_rs, eps_rs = fz["rs"]
        raw = torch.empty((b, h, w, c_sh), **f32)
        wino_shared = fz["shared"]._use_wino4(b, h, w, False) and not getattr(self, "force_stats_epilogue", False)
        cp = self._chain_head_plan(fz, b, h, w, dev, mul is not None) if (wino_shared and not getattr(self, "force_tiled_branches", False)) else None
        if cp is not None:
            # r4: shared convolution (F(4,3)) stored transposed -- the head's chain runs on the transposed map, and a transposed source lets
            # the RSNorm pass read and write along contiguous memory --, RSNorm + ReLU straight into planes, the branch convolutions chained
            # in the Winograd domain
            raw = torch.empty((b, w, h, c_sh), **f32)
            fz["shared"](x, out=raw, out_transposed=True)
            if mul is not None and "cal_t" not in cp:
                cp["cal_t"] = (mul.transpose(0, 1).contiguous(), add.transpose(0, 1).contiguous())
            mul, add = cp["cal_t"] if mul is not None else (None, None)
            mid, tabs_by_name = self._first_stage_chained(fz, cp, raw, mul, add)
            return self._last_stage(fz, mid, [tabs_by_name[name] for name, *_ in fz["first"]], f32)
        j0 = ops.ConvJob(fz["shared"], x, raw)
        xcat = torch.empty((b, h, w, c_sh * (2 if mul is not None else 1)), **f32)
        cm_tot = sum(cm for *_, cm in fz["first"])
        mid = torch.empty((b, h, w, cm_tot), **f32)
        srcs = {name: (xcat, c_sh if (name == "hm" and mul is not None) else 0) for name, *_ in fz["first"]}   # branch input: (map, channel offset)
        if wino_shared:
            # the shared convolution on the F(4, 3) kernel (44 us against 73 for the direct multi-job launch with the statistics
            # epilogue, 384 -> 64 at 128 x 128) + the stand-alone RSNorm pass (statistics, finalize, apply: the kernels of 4.3)
            fz["shared"](x, out=raw)
            if mul is not None:
                xs, x_hm = ops.groupnorm_strat(raw, 1, s_rs, g_rs, b_rs, eps_rs, act=ops.ACT_RELU, mul=mul, add=add)
            else:
                xs = x_hm = ops.groupnorm_strat(raw, 1, s_rs, g_rs, b_rs, eps_rs, act=ops.ACT_RELU)
            srcs = {name: ((x_hm if name == "hm" else xs), 0) for name, *_ in fz["first"]}
        else:
            # launch 1: shared convolution, RSNorm statistics partials in its epilogue
            j0.stats = dict(strata=s_rs, channel_groups=1, gamma=None, beta=None, eps=eps_rs, affine_strata=s_rs,
                            partials=torch.empty((j0.partial_floats(3),), **f32))
            ops.conv_multi([j0], 3)
            # launch 2: RSNorm + ReLU -> xs, and the position-calibrated copy for the heat-map branch (each block folds the partials of
            # its own stratum first: no statistics pass, no finalize launch)
            ops.conv_stats_apply(j0, 3, g_rs, b_rs, ops.ACT_RELU, xcat, 0, mul=mul, add=add, out2=xcat if mul is not None else None,
                                 out2_channel_offset=c_sh)
        jobs1, off = [], 0
        for name, lay, st, cmid in fz["first"]:
            jobs1.append(ops.ConvJob(lay, srcs[name][0], mid, in_channel_offset=srcs[name][1], out_channel_offset=off))
            off += cmid
        # launch 3: every first-stage branch, GroupNorm statistics partials in the epilogue; launch 3b folds them
        tabs, off = [], 0
        for jb, (name, lay, st, cmid) in zip(jobs1, fz["first"]):
            tab = torch.empty((b, st["affine_strata"], cmid, 2), **f32)
            jb.stats = dict(st, affine=tab, partials=torch.empty((jb.partial_floats(3),), **f32))
            tabs.append((tab, st["affine_strata"], cmid, off))
            off += cmid
        ops.conv_multi(jobs1, 3)
        ops.conv_stats_finalize(jobs1, 3)
        return self._last_stage(fz, mid, tabs, f32)

    def _last_stage(self, fz, mid, tabs, f32):
        """launch 4: every last convolution, GroupNorm + ReLU applied while the input tile is loaded.  tabs: per branch (affine table, strata,
        channels, channel offset in mid)"""
        b, h, w, _ = mid.shape
        widths = [(co + 3) // 4 * 4 for *_, co in fz["last"]]
        out = torch.empty((b, h, w, sum(widths)), **f32)
        jobs, off, outs = [], 0, {}
        for (name, lay, co), (tab, strata, cmid, in_off), wd in zip(fz["last"], tabs, widths):
            jobs.append(ops.ConvJob(lay, mid, out, in_channel_offset=in_off, out_channel_offset=off, norm=(tab, strata, cmid)))
            outs[name] = out[..., off:off + co]
            off += wd
        small = all(j.layer.cin <= 64 and j.layer.out_channels <= 12 and (j.layer.kh, j.layer.kw) in ((1, 1), (3, 3)) and j.layer.stride == 1
                    and (strata == 1 or j.layer.kh == 1) for j, (_, strata, _, _) in zip(jobs, tabs))
        if small and not getattr(self, "force_mfma_last", False):
            ops.conv_small_n_multi(jobs)     # few output columns: 16-column MFMA tiles over (tap, output) columns (r5; conv_mfma.hip small_n_gform_body)
        else:
            ops.conv_multi(jobs, 4)
        return outs

    def _merge_branches(self, branches):
        """Branches conv3x3(64->64) + per-channel GroupNorm + ReLU + conv that read the same shared map (everything but
        'hm' and the stratified 'reg') are run pairwise as ONE 64->128 convolution + ONE 128-channel GroupNorm; the
        final convolutions then read their half of the 128-channel map.  Same arithmetic per output channel (the
        normalisation is per channel), half the launches and a full 128-column MFMA tile instead of two of 64."""
        cands = []
        for name, steps in branches.items():
            if name == "hm" or len(steps) != 3 or steps[0][0] != "conv" or steps[1][0] != "gn" or steps[2][0] != "conv":
                continue
            fc = [m for m in getattr(self, name)._modules.values()]
            conv, gn = fc[0], fc[1]
            if not (isinstance(conv, nn.Conv2d) and conv.groups == 1 and isinstance(gn, nn.GroupNorm) and gn.num_groups == gn.num_channels):
                continue
            cands.append((name, conv, gn, steps[2][1]))
        merged = []
        for i in range(0, len(cands) - 1, 2):
            (na, ca, ga, fa), (nb, cb, gb, fb) = cands[i], cands[i + 1]
            if ca.weight.shape != cb.weight.shape or ga.eps != gb.eps:
                continue
            conv = ops.ConvLayer(torch.cat([ca.weight, cb.weight], 0), stride=1, pad=ca.padding[0],
                                 shift=torch.cat([ca.bias, cb.bias], 0).detach(), act=ops.ACT_NONE)
            gn = (ga.num_groups + gb.num_groups, 1, torch.cat([ga.weight, gb.weight]).detach(), torch.cat([ga.bias, gb.bias]).detach(), ga.eps)
            merged.append(dict(names=(na, nb), conv=conv, gn=gn, finals=(fa, fb), split=ca.weight.shape[0]))
        return merged

    @staticmethod
    def _run_branch(steps, x):
        for st in steps:
            if st[0] == "conv":
                x = st[1](x)
            else:  # GroupNorm is always followed by ReLU in these heads
                x = ops.groupnorm_strat(x, st[1], st[2], st[3], st[4], st[5], act=ops.ACT_RELU)
        return x

    def _calibration(self, plan, like: torch.Tensor):
        return None, None

    def forward(self, x, **kwargs):
        hip.require_device(x)
        eval_only(self, type(self).__name__)
        plan = self._plan.get(self, self._build_plan)
        xh = ops.to_nhwc(x)
        if self._fused_ok(plan["fused"], xh.shape[0], xh.shape[1], xh.shape[2]) and not getattr(self, "force_unfused", False):
            outs = self._forward_fused(plan, xh)
            ret = {}
            for name in plan["branches"]:
                y = ops.as_nchw(outs[name])
                if "_" in name:
                    names = name.split("_")
                    dim = y.shape[1] // len(names)
                    for j, nm in enumerate(names):
                        ret[nm] = y[:, j * dim:(j + 1) * dim]
                else:
                    ret[name] = y
            return {"det_preds": [ret]}
        raw = plan["shared"](xh)
        cg, st, ga, be, eps = plan["shared_gn"]
        mul, add = self._calibration(plan, raw)
        if mul is not None:
            xs, x_hm = ops.groupnorm_strat(raw, cg, st, ga, be, eps, act=ops.ACT_RELU, mul=mul, add=add)
        else:
            xs = x_hm = ops.groupnorm_strat(raw, cg, st, ga, be, eps, act=ops.ACT_RELU)
        ret = {}
        outs = {}
        for mg in plan["merged"]:
            cg, st, ga, be, eps = mg["gn"]
            h = ops.groupnorm_strat(mg["conv"](xs), cg, st, ga, be, eps, act=ops.ACT_RELU)
            for j, (name, final) in enumerate(zip(mg["names"], mg["finals"])):
                outs[name] = final(h, in_channel_offset=j * mg["split"])
        for name, steps in plan["branches"].items():
            y = ops.as_nchw(outs[name] if name in outs else self._run_branch(steps, x_hm if name == "hm" else xs))
            if "_" in name:
                names = name.split("_")
                dim = y.shape[1] // len(names)
                for j, nm in enumerate(names):
                    ret[nm] = y[:, j * dim:(j + 1) * dim]
            elif "heightdim" in name:
                ret["height"], ret["dim"] = y[:, :1], y[:, 1:]
            else:
                ret[name] = y
        return {"det_preds": [ret]}


def polar_pos_encoding(voxel_generator, out_size_factor) -> torch.Tensor:
    """(1,5,A,R) = [r cos(a), r sin(a), r, cos(a), sin(a)] at the BEV cell corners of the head map
    (center_head_parallel.py:229-241).  Input independent; built once on the host."""
    pr, vs, ns = list(voxel_generator["range"]), voxel_generator["voxel_size"], voxel_generator["nsectors"]
    az_hi = pr[1] + (pr[4] - pr[1]) / ns
    r_size = round((pr[3] - pr[0]) / vs[0] / out_size_factor)
    a_size = round((az_hi - pr[1]) / vs[1] / out_size_factor)
    ga, gr = torch.meshgrid(torch.arange(a_size), torch.arange(r_size), indexing="ij")
    ga = ga * out_size_factor * vs[1] + pr[1]
    gr = gr * out_size_factor * vs[0] + pr[0]
    c, s = torch.cos(ga), torch.sin(ga)
    return torch.stack([gr * c, gr * s, gr, c, s])[None]


@BBOX_HEADS.register_module
class CenterHeadSinglePos(CenterHeadSingle):
    """CenterHeadSingle + position-conditioned feature undistortion of the heat-map branch:
    hm = head(x * W(pos) + b(pos))  (center_head_parallel.py:199-284)."""

    def __init__(self, in_channels=[128, ], tasks=[], dataset="nuscenes", weight=0.25, code_weights=[], common_heads=dict(),
                 logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2, dcn_head=False, voxel_shape="cuboid",
                 voxel_generator=None, out_size_factor=4):
        super().__init__(in_channels, tasks, dataset, weight, code_weights, common_heads, logger, init_bias,
                         share_conv_channel, num_hm_conv, dcn_head, voxel_shape)
        head_conv = 64
        with torch.no_grad():
            self.pos_encoding = polar_pos_encoding(voxel_generator, out_size_factor)  # plain attribute, as in the reference
        self.calibration_weight = Sequential(nn.Conv2d(5, head_conv, kernel_size=3, padding=1), nn.Tanh(),
                                             nn.Conv2d(head_conv, head_conv, kernel_size=1), nn.Tanh())
        self.calibration_bias = Sequential(nn.Conv2d(5, head_conv, kernel_size=3, padding=1), nn.Tanh(),
                                           nn.Conv2d(head_conv, head_conv, kernel_size=1))

    def _build_plan(self):
        plan = super()._build_plan()
        dev = self.calibration_weight[0].weight.device
        hip.require_device(self.calibration_weight[0].weight)
        pos = ops.to_nhwc(self.pos_encoding.to(dev))

        def fold(seq, last_act):
            h = ops.conv2d_direct(pos, seq[0].weight, seq[0].bias, 1, 1, 1, act=ops.ACT_TANH)  # Cin = 5: direct kernel
            return ops.ConvLayer(seq[2].weight, shift=seq[2].bias, act=last_act)(h)[0].contiguous()  # (A,R,C)

        plan["cal_mul"] = fold(self.calibration_weight, ops.ACT_TANH)
        plan["cal_add"] = fold(self.calibration_bias, ops.ACT_NONE)
        return plan

    def _calibration(self, plan, like):
        if tuple(plan["cal_mul"].shape[:2]) != tuple(like.shape[1:3]):
            raise ValueError(f"head input map {tuple(like.shape[1:3])} does not match the position encoding "
                             f"{tuple(plan['cal_mul'].shape[:2])} of the configured voxel grid")
        return plan["cal_mul"], plan["cal_add"]
