"""Detectors (orchestration) under the reference's names.

Reference: det3d/models/detectors/single_stage.py:25-50, point_pillars.py:40-110,
voxelnet.py:47-131,171-301.

``PointPillars.forward`` keeps the reference's ``example`` contract.  In eval mode the dynamic
branch runs as one device-side chain with no host synchronisation: linear keys -> bitmap
unique-rank -> bucketing -> fused PFN writing the NHWC canvas -> RPN -> head.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import builder, hip, ops
from .builder import DETECTORS
from .nn_utils import eval_only
from .readers import DynamicPFNet
from .routes import R, S


@DETECTORS.register_module
class SingleStageDetector(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, seg_head=None, part_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None, nsectors=1):
        super().__init__()
        self.reader = builder.build_reader(reader)
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self.bbox_head = builder.build_bbox_head(bbox_head) if bbox_head is not None else None
        # the `seg` super-task of the nuScenes polar config (SingleConvHead): forward + per-point prediction, eval only
        self.seg_head = builder.build_seg_head(seg_head) if seg_head is not None else None
        if part_head is not None:
            raise NotImplementedError("part heads are outside the hot path (SURVEY.md 2.1); build with part_head=None")
        self.part_head = None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained)

    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None

    def init_weights(self, pretrained=None):
        if pretrained is None:
            return
        sd = torch.load(pretrained, map_location="cpu")
        sd = sd.get("state_dict", sd)
        self.load_state_dict({k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}, strict=False)


@DETECTORS.register_module
class PointPillars(SingleStageDetector):
    def __init__(self, reader, backbone, neck, bbox_head, seg_head=None, part_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__(reader, backbone, neck, bbox_head, seg_head, part_head, train_cfg, test_cfg, pretrained)

    # -- stage API of the reference (point_pillars.py:40-53) ---------------------------------
    def extract_feat_dynamic(self, data):
        feats, unq = self.reader(data)
        x1 = self.backbone(feats, unq, data["batch_size"], data["grid_size"])
        return x1, self.neck(x1)

    # -- fused device-side chain --------------------------------------------------------------
    def encode_canvas(self, points: torch.Tensor, keys: torch.Tensor, spec: ops.GridSpec, batch: int,
                      n_dev: Optional[torch.Tensor] = None, canvas: Optional[torch.Tensor] = None, return_index=False):
        """points (N,7) + linear voxel keys -> NHWC canvas (B, T, R, C); no host sync.
        ``canvas``: a persistent all-zero map to write into (see ``forward_points``); otherwise a fresh zero-filled one."""
        if not isinstance(self.reader, DynamicPFNet):
            raise NotImplementedError("fused encode path needs a DynamicPFNet reader")
        vi = ops.build_voxel_index(keys, spec, batch, n_dev=n_dev, want_unq=False)
        shape = (batch, spec.grid[1], spec.grid[0], self.reader.out_channels)
        if canvas is None:
            canvas = torch.empty(shape, dtype=torch.float32, device=points.device)
            hip.call("pn_fill_zero", canvas.data_ptr(), canvas.numel() * 4, hip.stream())
        else:
            hip.require_device(canvas)
            assert tuple(canvas.shape) == shape and canvas.is_contiguous() and canvas.dtype == torch.float32
        self.reader.encode(points, vi, None, canvas)
        return (canvas, vi) if return_index else canvas

    def _neck_on_canvas(self, canvas: torch.Tensor, vi) -> torch.Tensor:
        """the backbone on a freshly encoded canvas: plain RPNs take the voxel index with it (sparse first convolution)"""
        from .necks import RPN
        if type(self.neck) is RPN:
            return self.neck.forward_nhwc(canvas, pillars=vi)
        return self.neck.forward_nhwc(canvas)

    def new_canvas(self, batch: int, spec: Optional[ops.GridSpec] = None, device=None) -> torch.Tensor:
        """a zeroed persistent canvas for ``forward_points(..., canvas=)``"""
        spec = spec or ops.GridSpec.from_range(self.reader.pc_range, self.reader.voxel_size)
        device = device or next(self.parameters()).device
        return torch.zeros((batch, spec.grid[1], spec.grid[0], self.reader.out_channels), dtype=torch.float32, device=device)

    def forward_points(self, points: torch.Tensor, sample_offsets: torch.Tensor, batch: int,
                       spec: Optional[ops.GridSpec] = None, canvas: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Hot path from polar-decorated points: grid indices are computed on the device (V1).
        points (N,7) [rho,phi,z,x,y,i,t]; sample_offsets int32 (batch+1).  -> head tensors.
        ``canvas``: a persistent map owned by the caller (``new_canvas``), all zero on entry; the frame's pillars are
        written into it and their cells are cleared again once the backbone has consumed it (a sparse clear of ~28k
        cells instead of a 134 MB fill per frame) -- all zero on exit."""
        eval_only(self, "PointPillars")
        spec = spec or ops.GridSpec.from_range(self.reader.pc_range, self.reader.voxel_size)
        _, keys = ops.grid_index(points, sample_offsets, batch, spec, want_grid_ind=False)
        if canvas is None:
            cv, vi = self.encode_canvas(points, keys, spec, batch, n_dev=sample_offsets[batch:], return_index=True)
            x2 = self._neck_on_canvas(cv, vi)
        else:
            cv, vi = self.encode_canvas(points, keys, spec, batch, n_dev=sample_offsets[batch:], canvas=canvas, return_index=True)
            x2 = self._neck_on_canvas(cv, vi)
            ops.clear_canvas_cells(cv, vi)
        return self.bbox_head(ops.as_nchw(x2))["det_preds"][0]

    def new_index_state(self, batch: int, spec: Optional[ops.GridSpec] = None, device=None):
        """persistent scratch of the fused frame index for ``forward_cart(..., index_state=)``; None when the grid is too large"""
        spec = spec or ops.GridSpec.from_range(self.reader.pc_range, self.reader.voxel_size)
        device = device or next(self.parameters()).device
        return ops.FrameIndexState(spec, batch, device) if ops.FrameIndexState.supported(spec, batch) else None

    def encode_cart(self, cart: torch.Tensor, sample_offsets: torch.Tensor, batch: int, spec: ops.GridSpec,
                    canvas: Optional[torch.Tensor] = None, index_state=None):
        """V0..V5 from Cartesian points: fused frame index (3 launches) + fused PFN writing the canvas cells (1 launch).
        -> (canvas, VoxelIndex, polar points)"""
        if not isinstance(self.reader, DynamicPFNet):
            raise NotImplementedError("fused encode path needs a DynamicPFNet reader")
        shape = (batch, spec.grid[1], spec.grid[0], self.reader.out_channels)
        if index_state is None and not ops.FrameIndexState.supported(spec, batch):
            polar = ops.cart_to_polar(cart)
            _, keys = ops.grid_index(polar, sample_offsets, batch, spec, want_grid_ind=False)
            cv, vi = self.encode_canvas(polar, keys, spec, batch, n_dev=sample_offsets[batch:], canvas=canvas, return_index=True)
            return cv, vi, polar
        polar, vi = ops.fused_voxel_index(cart, sample_offsets, batch, spec, index_state)
        if canvas is None:
            canvas = torch.empty(shape, dtype=torch.float32, device=cart.device)
            hip.call("pn_fill_zero", canvas.data_ptr(), canvas.numel() * 4, hip.stream())
        else:
            hip.require_device(canvas)
            assert tuple(canvas.shape) == shape and canvas.is_contiguous() and canvas.dtype == torch.float32
        vi.index_cleared = self.reader.encode(polar, vi, None, canvas, clear_index=index_state)      # (r6: the reader's launch zeroes the index counters)
        return canvas, vi, polar

    def forward_sweeps(self, raw: torch.Tensor, sweep_offsets: torch.Tensor, transforms: torch.Tensor, time_lags: torch.Tensor, spec: ops.GridSpec,
                       canvas: torch.Tensor, index_state, canvas_may_stay_dirty: bool = False, min_distance: float = 1.0) -> Dict[str, torch.Tensor]:
        """``forward_cart`` for ONE multi-sweep frame handed over as its raw sweeps (BASELINE configs[4]): the accumulation
        (``ops.accumulate_sweeps``: remove_close, rigid transforms, time lags) happens inside the frame index's first launch
        (``ops.fused_voxel_index_sweeps``, r6) -- no compaction, no Cartesian copy, three launches less.  Same head tensors, bit for bit
        (the reader's sums and maxima do not depend on the order or the numbering of a pillar's points)."""
        eval_only(self, "PointPillars")
        if not isinstance(self.reader, DynamicPFNet):
            raise NotImplementedError("fused encode path needs a DynamicPFNet reader")
        hip.require_device(canvas)
        assert tuple(canvas.shape) == (1, spec.grid[1], spec.grid[0], self.reader.out_channels) and canvas.is_contiguous()
        polar, vi = ops.fused_voxel_index_sweeps(raw, sweep_offsets, transforms, time_lags, spec, index_state, min_distance)
        cleared = self.reader.encode(polar, vi, None, canvas, clear_index=index_state)
        x2 = self._neck_on_canvas(canvas, vi)
        leave = canvas_may_stay_dirty and self.seg_head is None and getattr(self.neck, "canvas_read_by_pillars_only", False)
        ops.clear_frame_cells(None if leave else canvas, vi, None if cleared else index_state)
        return self.bbox_head(ops.as_nchw(x2))["det_preds"][0]

    def scatter_stage(self, cart: torch.Tensor, sample_offsets: torch.Tensor, batch: int, spec: ops.GridSpec, canvas: torch.Tensor,
                      index_state=None):
        """V0..V5 alone, exactly as a frame of ``forward_cart(canvas=, index_state=)`` runs them (fused frame index, fused PFN
        writing the persistent canvas, sparse clear): what bench.py's ``roofline_scatter`` times.
        -> the VoxelIndex (voxel count on the device)"""
        cv, vi, _ = self.encode_cart(cart, sample_offsets, batch, spec, canvas=canvas, index_state=index_state)
        ops.clear_frame_cells(cv, vi, getattr(vi, "state", None) if (index_state is not None and not getattr(vi, "index_cleared", False)) else None)
        return vi

    def forward_cart(self, cart: torch.Tensor, sample_offsets: torch.Tensor, batch: int, spec: Optional[ops.GridSpec] = None,
                     canvas: Optional[torch.Tensor] = None, index_state=None, canvas_may_stay_dirty: bool = False) -> Dict[str, torch.Tensor]:
        """The hot path from CARTESIAN points (N, 5) [x, y, z, intensity, dt] resident on the device: V0 .. H2, head tensors out.
        ``canvas`` / ``index_state``: persistent buffers owned by the caller (``new_canvas`` / ``new_index_state``), all zero on
        entry and on exit (the frame's cells are cleared once the backbone's first layer has consumed the canvas).
        ``canvas_may_stay_dirty`` (r6; the frame engines, whose canvas nobody else sees): where the first convolution took the row-band pillar
        form -- it reads exactly the cells of the frame's key list and nothing else does -- the old frame's feature rows are left in the
        canvas (only the index counters are cleared): 14.5 MB of zero stores less per 30k-point frame, 92 MB per 300k-point frame.  Such a
        canvas must never reach a dense consumer."""
        eval_only(self, "PointPillars")
        spec = spec or ops.GridSpec.from_range(self.reader.pc_range, self.reader.voxel_size)
        cv, vi, _ = self.encode_cart(cart, sample_offsets, batch, spec, canvas=canvas, index_state=index_state)
        x2 = self._neck_on_canvas(cv, vi)
        if canvas is not None or index_state is not None:
            leave = (canvas_may_stay_dirty and canvas is not None and index_state is not None and self.seg_head is None
                     and getattr(self.neck, "canvas_read_by_pillars_only", False))
            ops.clear_frame_cells(None if leave else (cv if canvas is not None else None), vi, None if getattr(vi, "index_cleared", False) else index_state)
        return self.bbox_head(ops.as_nchw(x2))["det_preds"][0]

    def extract_preds(self, example) -> Dict[str, object]:
        """reference ``example`` dict (dynamic branch keys) -> {'det_preds': [...]}"""
        eval_only(self, "PointPillars")
        if "voxels" in example:   # hard-voxel (static) branch, point_pillars.py:27-38 / 60-76
            voxels, coords = example["voxels"], example["coordinates"]
            hip.require_device(voxels, coords)
            feats = self.reader(voxels, example["num_points"], coords)
            x1 = self.backbone(feats, coords, len(example["num_voxels"]), [int(v) for v in example["shape"][0]])
            x1h = ops.to_nhwc(x1)
            x2 = self.neck.forward_nhwc(x1h) if self.with_neck else x1h
            return self._heads(x1h, x2)
        points, grid_ind = example["points"], example["grid_ind"]
        hip.require_device(points, grid_ind)
        batch = len(example["num_points"])
        g = [int(v) for v in example["grid_size"][0]]
        spec = ops.GridSpec.from_range(self.reader.pc_range, self.reader.voxel_size)
        if list(spec.grid) != g:
            raise ValueError(f"example grid_size {g} does not match the reader's voxel grid {list(spec.grid)}")
        keys = ops.keys_from_grid_ind(grid_ind.to(torch.int64).contiguous(), spec, batch)
        canvas, vi = self.encode_canvas(points.contiguous(), keys, spec, batch, return_index=True)
        x2 = self._neck_on_canvas(canvas, vi)
        return self._heads(canvas, x2)

    def _heads(self, x1_nhwc: torch.Tensor, x2_nhwc: torch.Tensor) -> Dict[str, object]:
        """point_pillars.py:91-96: the detection head on the RPN output, the segmentation head (when the config has one) on the
        canvas + the RPN output"""
        preds = {}
        if self.bbox_head is not None:
            preds.update(self.bbox_head(ops.as_nchw(x2_nhwc)))
        if self.seg_head is not None:
            seg = self.seg_head.forward_nhwc(x1_nhwc, x2_nhwc)
            preds["seg_preds"] = seg.permute(0, 3, 1, 2)[:, :self.seg_head.num_classes]
        return preds

    def forward(self, example, return_loss=True, **kwargs):
        preds = self.extract_preds(example)
        if return_loss:
            if self.seg_head is not None:
                raise NotImplementedError("training with the segmentation head is out of scope (SegLoss is not built)")
            return self.bbox_head.loss(example, preds)
        if kwargs.get("raw_preds", False) or self.test_cfg is None:
            return preds
        ret = {}
        if self.bbox_head is not None:
            ret["det"] = self.bbox_head.predict(example, preds, self.test_cfg, **kwargs)
        if self.seg_head is not None:
            ret["seg"] = self.seg_head.predict(example, preds, self.test_cfg)
        return ret


@DETECTORS.register_module
class PolarStream(PointPillars):
    """PolarStream (det3d/models/detectors/polarstream.py:7-180): a sweep given as a LIST of azimuth-sector examples is processed
    sector by sector, the neck (RPNTECP / RPNBDCP) pads every sector with the context rows the previous sector left behind, and the
    per-sector detections are rotated back into the sweep's frame and concatenated (single_stage.py:83-165).  A single example (dict)
    is the full-sweep case.  Eval mode; detection super-task; stateful NMS across sectors (test_cfg.stateful_nms, the default of the
    reference's 4-sector configs) is supported, panoptic fusion is not."""

    def forward_one_sector(self, example, return_loss=True, **kwargs):
        eval_only(self, "PolarStream")
        points, grid_ind = example["points"], example["grid_ind"]
        hip.require_device(points, grid_ind)
        batch = len(example["num_points"])
        g = [int(v) for v in example["grid_size"][0]]
        # the sector grid: the reader's range with the azimuth axis cut to the sector (voxelization.py:318-323)
        spec = ops.GridSpec(tuple(float(v) for v in self.reader.pc_range[:3]), tuple(float(v) for v in self.reader.voxel_size), (g[0], g[1], g[2]))
        keys = ops.keys_from_grid_ind(grid_ind.to(torch.int64).contiguous(), spec, batch)
        canvas = self.encode_canvas(points.contiguous(), keys, spec, batch)
        nxt = []
        if hasattr(self.neck, "_pad_feature_only") or not hasattr(self.neck, "forward_nhwc"):
            raise NotImplementedError("PolarStream drives the trailing-edge neck (RPNTECP) or the plain RPN; RPNBDCP is driven by PolarStreamBDCP's two-sweep loop")
        from .necks_context import RPNTECP
        if isinstance(self.neck, RPNTECP):      # the trailing-edge neck takes (and returns) the context rows
            x2, nxt = self.neck.forward_nhwc(canvas, kwargs.get("prev_context", []), kwargs.get("sec_id", 0))
        else:                                   # plain RPN: no context, its own signature (return_blocks, pillars)
            x2 = self.neck.forward_nhwc(canvas)
        preds = self.bbox_head(ops.as_nchw(x2))
        ret = {}
        if return_loss:
            ret.update(self.bbox_head.loss(example, preds))
        elif kwargs.get("raw_preds", False) or self.test_cfg is None:
            ret.update(preds)
        else:
            ret["det"] = self.bbox_head.predict(example, preds, self.test_cfg, sec_id=kwargs.get("sec_id", 0), prev_dets=kwargs.get("prev_dets"))
        if len(nxt):
            ret["next_context"] = nxt
        return ret

    def forward(self, example, return_loss=True, **kwargs):
        if isinstance(example, dict):
            return self.forward_one_sector(example, return_loss, **kwargs)
        get = (lambda k, d=None: self.test_cfg.get(k, d)) if hasattr(self.test_cfg, "get") else (lambda k, d=None: getattr(self.test_cfg, k, d))
        if self.test_cfg is not None and get("panoptic", False):
            raise NotImplementedError("PolarStream: panoptic fusion across sectors is not built")
        stateful = self.test_cfg is not None and bool(get("stateful_nms", False))
        rets, prev = [], []
        for i, ex in enumerate(example):
            kw = dict(kwargs, prev_context=prev, sec_id=i)
            if stateful and i > 0 and "det" in rets[-1]:
                kw["prev_dets"] = rets[-1]["det"]     # polarstream.py:91-92
            r = self.forward_one_sector(ex, return_loss, **kw)
            prev = r.pop("next_context", []) if i < len(example) - 1 else []
            r.pop("next_context", None)
            rets.append(r)
        out = self.merge_sectors(rets, len(example[-1]["num_points"]), stateful)
        if stateful and "det" in out:
            for det, meta in zip(out["det"], example[-1].get("metadata", [None] * len(out["det"]))):
                det["metadata"] = meta
        return out

    def merge_sectors(self, sectors, batch_size, stateful=False):
        """single_stage.py:83-165 for the keys this build produces: losses are lists concatenated over sectors, detections are
        concatenated per sample; with stateful NMS the LAST sector's per-task lists already hold the whole sweep (merge_dets
        :137-153: tasks concatenated, labels offset by the preceding tasks' class counts)"""
        out = {}
        for k in sectors[0]:
            vals = [s[k] for s in sectors]
            if "loss" in k:
                out[k] = [v for lst in vals for v in lst]
            elif k == "det" and stateful:
                tasks = vals[-1]
                merged = []
                for i in range(len(tasks[0])):
                    flag, labels = 0, []
                    for j, ncls in enumerate(self.bbox_head.num_classes[:len(tasks)]):
                        labels.append(tasks[j][i]["label_preds"] + flag)
                        flag += ncls
                    merged.append(dict(box3d_lidar=torch.cat([t[i]["box3d_lidar"] for t in tasks]), scores=torch.cat([t[i]["scores"] for t in tasks]),
                                       label_preds=torch.cat(labels)))
                out[k] = merged
            elif k == "det":
                merged = []
                for i in range(len(vals[0])):
                    d = {}
                    for f in vals[0][0]:
                        d[f] = vals[0][i][f] if f == "metadata" else torch.cat([v[i][f] for v in vals])
                    merged.append(d)
                out[k] = merged
            elif k == "det_preds":
                out[k] = vals
        return out


@DETECTORS.register_module
class PolarStreamBDCP(PolarStream):
    """PolarStream with bidirectional context padding (det3d/models/detectors/polarstream.py:180-470): ``forward`` takes TWO sweeps.
    The previous sweep runs in ``feature_only`` mode (all its sectors stacked in the batch axis, sector-major); the per-layer inputs
    of the neck (RPNBDCP) are glued back into whole-sweep maps and warped into the current sweep's frame by the ego rotation
    (``transform_matrix`` (bs, 2, 2); `pn_polar_warp_f32`).  The current sweep is then streamed sector by sector: trailing-edge context
    from the sector before, leading-edge context from the warped previous sweep.  Eval mode, detection super-task."""

    def __init__(self, reader, backbone, neck, bbox_head, seg_head=None, part_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 nsectors=1):
        neck = dict(neck)
        neck["nsectors"] = nsectors          # polarstream.py:205
        super().__init__(reader, backbone, neck, bbox_head, seg_head, part_head, train_cfg, test_cfg, pretrained)
        self.nsectors = int(nsectors)

    def _canvas(self, example):
        points, grid_ind = example["points"], example["grid_ind"]
        hip.require_device(points, grid_ind)
        batch = len(example["num_points"])
        g = [int(v) for v in example["grid_size"][0]]
        spec = ops.GridSpec(tuple(float(v) for v in self.reader.pc_range[:3]), tuple(float(v) for v in self.reader.voxel_size), (g[0], g[1], g[2]))
        keys = ops.keys_from_grid_ind(grid_ind.to(torch.int64).contiguous(), spec, batch)
        return self.encode_canvas(points.contiguous(), keys, spec, batch), batch

    def _get(self, k, d=None):
        return self.test_cfg.get(k, d) if hasattr(self.test_cfg, "get") else getattr(self.test_cfg, k, d)

    def warp_prev_sweep(self, cur_sweep, transform, bs):
        """per-layer inputs of the stacked sectors (nsectors * bs, h, w, c) -> whole-sweep maps (bs, nsectors * h, w, c) warped by the
        (bs, 2, 2) rotation (polarstream.py:340-358)"""
        pr = self._get("pc_range")
        rot = transform.to(cur_sweep[0].device).float().contiguous()
        out = []
        for x in cur_sweep:
            n, h, w, c = x.shape
            assert n == self.nsectors * bs and x.is_contiguous()
            warped = torch.empty((bs, self.nsectors * h, w, c), dtype=torch.float32, device=x.device)
            hip.call("pn_polar_warp_f32", x.data_ptr(), rot.data_ptr(), bs, self.nsectors, self.nsectors * h, w, c, float(pr[0]), float(pr[3]), float(pr[1]),
                     float(pr[4]), warped.data_ptr(), hip.stream())
            out.append(warped)
        return out

    def forward_one_sweep(self, example, mode="feature_only", return_loss=False, **kwargs):
        eval_only(self, "PolarStreamBDCP")
        if return_loss:
            raise NotImplementedError("PolarStreamBDCP: training (the two-sweep loss path) is not built")
        canvas, batch = self._canvas(example)
        nsec = self.nsectors
        bs = batch // nsec
        if mode == "feature_only":
            _, cur = self.neck.forward_nhwc(canvas, nsectors=nsec, mode="feature_only")
            return self.warp_prev_sweep(cur, torch.as_tensor(example["transform_matrix"])[:bs], bs)
        prev_sweep = kwargs["prev_sweep"]
        if nsec == 1:
            x2, _ = self.neck.forward_nhwc(canvas, prev_sweep=prev_sweep, nsectors=1, mode=mode)
        else:
            outs, ctx = [], []
            for j in range(nsec):
                y, ctx = self.neck.forward_nhwc(canvas[j * bs:(j + 1) * bs].contiguous(), prev_sweep=prev_sweep, prev_context=ctx, sec_id=j, nsectors=nsec,
                                                mode=mode)
                outs.append(y)
            x2 = torch.cat(outs, 0)
        preds = self.bbox_head(ops.as_nchw(x2))
        if kwargs.get("raw_preds", False) or self.test_cfg is None:
            return [dict(det_preds=[{k: v[i * bs:(i + 1) * bs] for k, v in t.items()} for t in preds["det_preds"]]) for i in range(nsec)]
        stateful = bool(self._get("stateful_nms", False))
        rets, prev_dets = [], None
        metas = example.get("metadata", [None] * batch)
        for i in range(nsec):
            pr = {"det_preds": [{k: v[i * bs:(i + 1) * bs] for k, v in t.items()} for t in preds["det_preds"]]}
            ex = dict(metadata=metas[i * bs:(i + 1) * bs])
            if "pc_range" in example:
                ex["pc_range"] = example["pc_range"][i * bs:(i + 1) * bs]
            det = self.bbox_head.predict(ex, pr, self.test_cfg, sec_id=i, prev_dets=prev_dets)
            rets.append({"det": det})
            prev_dets = det if stateful and i < nsec - 1 else None
        return rets

    def forward(self, example, return_loss=True, **kwargs):
        """example: [previous sweep, current sweep] (polarstream.py:266-290)"""
        if isinstance(example, dict) or len(example) != 2:
            raise ValueError("PolarStreamBDCP.forward takes a list of two sweeps [previous, current]")
        prev_sweep = self.forward_one_sweep(example[0], "feature_only", False, **kwargs)
        rets = self.forward_one_sweep(example[1], "train" if return_loss else "eval", return_loss, prev_sweep=prev_sweep, **kwargs)
        ex = example[1]
        bs = len(ex["num_points"]) // self.nsectors
        stateful = self.test_cfg is not None and bool(self._get("stateful_nms", False))
        out = self.merge_sectors(rets, bs, stateful)
        if stateful and "det" in out:
            for det, meta in zip(out["det"], ex.get("metadata", [None] * bs)[:bs]):
                det["metadata"] = meta
        return out


@DETECTORS.register_module
class VoxelNet(SingleStageDetector):
    """VoxelNet (voxelnet.py:27-131): reader -> sparse 3-D middle encoder -> RPN -> head, the detector of the reference's
    CenterPoint-style voxel configs (VoxelNetV3 is this plus the re-alignment attention).  Eval mode, on the HIP kernels."""

    def __init__(self, reader, backbone, neck, bbox_head, seg_head=None, part_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__(reader, backbone, neck, bbox_head, seg_head, part_head, train_cfg, test_cfg, pretrained)

    def extract_feat_hard(self, data):
        """voxelnet.py:47-58; returns the NHWC neck output"""
        feats = self.reader(data["features"], data["num_voxels"])
        x = self.backbone.forward_nhwc(feats, data["coors"], data["batch_size"], data["input_shape"])
        return self.neck.forward_nhwc(x) if self.with_neck else x

    def extract_feat_dynamic(self, data):
        """voxelnet.py:60-70: dynamic voxel encoder (mean of the points of a voxel) -> sparse backbone on unq"""
        feats, unq = self.reader(data)
        x = self.backbone.forward_nhwc(feats, unq.to(torch.int32), data["batch_size"], [int(v) for v in data["grid_size"]])
        return self.neck.forward_nhwc(x) if self.with_neck else x

    def forward(self, example, return_loss=True, **kwargs):
        eval_only(self, "VoxelNet")
        if "voxels" in example:
            hip.require_device(example["voxels"], example["coordinates"])
            data = dict(features=example["voxels"], num_voxels=example["num_points"], coors=example["coordinates"],
                        batch_size=len(example["num_voxels"]), input_shape=[int(v) for v in example["shape"][0]])
            x = self.extract_feat_hard(data)
        else:
            hip.require_device(example["points"], example["grid_ind"])
            data = dict(points=example["points"], grid_ind=example["grid_ind"], num_points=example["num_points"],
                        batch_size=len(example["num_points"]), voxel_size=example["voxel_size"][0], pc_range=example["pc_range"][0],
                        grid_size=example["grid_size"][0])
            x = self.extract_feat_dynamic(data)
        preds = self.bbox_head(ops.as_nchw(x))
        if return_loss:
            return self.bbox_head.loss(example, preds)
        if kwargs.get("raw_preds", False) or self.test_cfg is None:
            return preds
        return {"det": self.bbox_head.predict(example, preds, self.test_cfg, **kwargs)}


@DETECTORS.register_module
class VoxelNetV3(SingleStageDetector):
    """PARTNER detector (voxelnet.py:171-301): hard voxels -> mean VFE -> sparse 3-D backbone -> two SetBlocks
    (global representation re-alignment) -> RPN -> head, end to end on the HIP kernels (eval mode)."""

    def __init__(self, reader, backbone, neck, bbox_head, seg_head=None, part_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__(reader, backbone, neck, bbox_head, seg_head, part_head, train_cfg, test_cfg, pretrained)
        from .attention import SetBlock, waymo_bev_pos
        self.bev_pos = waymo_bev_pos()
        self.attns = nn.ModuleList([
            SetBlock(in_dim=256, embed_dim_scale=1, num_heads=4, reso=(144, 256), mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                     H_sp=144, W_sp=1, H=4, W=8, drop=0.1, attn_drop=0.1, drop_path=0.1, norm_layer=nn.LayerNorm,
                     pos=self.bev_pos, shift=(i % 2 == 1)) for i in range(2)])

    def set_compute_dtype(self, dtype: str) -> "VoxelNetV3":
        """"f32" (default: the reference's arithmetic, the parity path) or "bf16" (BASELINE configs[3]): the dense BEV stages on the bf16
        matrix pipe -- the RPN's and the head's convolutions (csrc/conv_bf16.hip) and the token GEMMs of the SetBlocks and of the head's
        Swin stage (pn_linear_bf16) -- with f32 accumulation; voxelization, the sparse encoder, LayerNorm / attention cores / GroupNorm,
        residual streams and every output stay f32."""
        assert dtype in ("f32", "bf16")
        # Only the LAST SetBlock: a block's output feeds the next block's key-point selection (top-4 local maxima per azimuth column, a discrete
        # choice among near-ties), and a bf16-sized perturbation of block 0's output flips enough of block 1's key points to move the head
        # tensors by 7 % of their range (measured, seeded weights) -- block 0 therefore stays f32; inside the last block the selection reads the
        # f32 LayerNorm output before any bf16 product.
        for i, attn in enumerate(self.attns):
            attn.set_compute_dtype(dtype if i == len(self.attns) - 1 else "f32")
        if self.with_neck:
            self.neck.set_compute_dtype(dtype)
        self.bbox_head.set_compute_dtype(dtype)
        if self.with_neck:
            # r6: the RPN hands its output map over in bf16 (its deblocks' epilogues round once) where the head reads it in bf16 anyway: the
            # f32 map (151 MB at bs 2) and the head's f32 -> bf16 pass over it are gone.  Same bits as rounding the f32 map.
            self.neck.bf16_output = bool(dtype == "bf16" and getattr(self.bbox_head, "takes_bf16_input", lambda: False)())
        return self

    def realign(self, x: torch.Tensor) -> torch.Tensor:
        """x: logical (B, C=256, theta=256, r=144) dense BEV map -> same shape (voxelnet.py:210-221)"""
        hip.require_device(x)
        return ops.as_nchw(self.realign_nhwc(ops.to_nhwc(x)))

    def realign_nhwc(self, xh: torch.Tensor) -> torch.Tensor:
        """NHWC (B, theta, r, C) -> same, through the two SetBlocks.  The reference permutes the map to range-major tokens and back
        (voxelnet.py:211,219); the blocks here take the tokens in the map's own azimuth-major order (``SetBlock.forward_cols``), so
        nothing is transposed -- element for element the same result (``test_setblock_column_major_equals_range_major``)."""
        b, t, r, c = xh.shape
        y = xh.contiguous().view(b, t * r, c)
        for attn in self.attns:
            y = attn.forward_cols(y)
        return y.view(b, t, r, c)

    def dense_stages_nhwc(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """the dense BEV map of the sparse backbone (B, theta, r, C) -> the head's NHWC tensors: 2 x SetBlock -> RPN -> head.
        r6: a batch of several samples runs them PER SAMPLE on two streams (branches of the same hipGraph when captured): the 2-D chain
        launches of a bs = 2 map are 576 tiles on 256 CUs -- 2.25 rounds paid as 3 --, two per-sample sequences of 288-tile launches fill
        each other's tails (tools/c4_sample_streams.py: SetBlocks + RPN + head 7.36 -> 6.99 ms).  Every kernel of these stages works on one
        sample's rows at a time, so a sample's tensors are those of the same sample alone (``PN_SAMPLE_STREAMS=0``: one sequence over the batch)."""
        def one(xs):
            y = self.realign_nhwc(xs)
            if self.with_neck:
                y = self.neck.forward_nhwc(y)
            out = self.bbox_head.forward_nhwc(y)
            out.pop("_feat", None)
            return out
        b = x.shape[0]
        if b < 2 or not R.sample_streams or not x.is_cuda:
            return one(x)
        main = torch.cuda.current_stream(x.device)
        side = ops.concurrent_stream(x.device)
        side.wait_stream(main)
        outs = [None] * b
        built = S.lazy_builds
        for i in range(0, b, 2):
            outs[i] = one(x[i:i + 1])
        if S.lazy_builds != built:      # plans / weight layouts were built on the way (the first call): they are queued on THIS stream
            side.wait_stream(main)
        with torch.cuda.stream(side):
            for i in range(1, b, 2):
                outs[i] = one(x[i:i + 1])
        main.wait_stream(side)
        for i in range(1, b, 2):
            for v in outs[i].values():
                v.record_stream(main)
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}

    def extract_feat_hard(self, data):
        """voxelnet.py:202-227: mean VFE -> sparse 3-D backbone -> 2 x SetBlock -> RPN; returns the NHWC neck output"""
        feats = self.reader(data["features"], data["num_voxels"])
        x = self.backbone.forward_nhwc(feats, data["coors"], data["batch_size"], data["input_shape"])
        x = self.realign_nhwc(x)
        if self.with_neck:
            x = self.neck.forward_nhwc(x)
        return x

    def forward_points(self, points: torch.Tensor, voxel_generator=None, sample_offsets=None):
        """Fused path from polar-decorated points (N, F) on the device: hard voxelization (device, count stays on the device) -> mean VFE
        -> sparse backbone -> 2 x SetBlock -> RPN -> head.  No host synchronisation: capturable in a hipGraph (``engine.FrameEngine``).
        ``voxel_generator``: dict(range, voxel_size, max_points_in_voxel, max_voxel_num); default = the one the config hands to the head.
        ``sample_offsets``: python ints [0, n_0, n_0 + n_1, ...] for a batch of several sweeps (r3): every sample is voxelized on its own
        and the lists are joined on the device (pn_concat_voxel_segments_f32 = the reference's collate, counts included)."""
        eval_only(self, "VoxelNetV3")
        vg = voxel_generator or getattr(self.bbox_head, "voxel_generator_cfg", None)
        if vg is None:
            raise ValueError("VoxelNetV3.forward_points needs the voxel_generator section of the config")
        mv = vg["max_voxel_num"]
        mv = int(mv[0] if isinstance(mv, (list, tuple)) else mv)
        rg, vs = vg["range"], vg["voxel_size"]
        grid = [int(round((rg[3 + a] - rg[a]) / vs[a])) for a in range(3)]
        offs = [0, int(points.shape[0])] if sample_offsets is None else [int(v) for v in sample_offsets]
        batch = len(offs) - 1
        if batch == 1:
            voxels, coors, num, nv = ops.hard_voxelize(points, vg["voxel_size"], vg["range"], int(vg["max_points_in_voxel"]), mv)
            feats = self.reader(voxels, num)
            coords4 = torch.cat([torch.zeros((mv, 1), dtype=torch.int32, device=points.device), coors], 1)
        else:
            fs, cs, nvs = [], [], []
            for b in range(batch):
                voxels, coors, num, nv = ops.hard_voxelize(points[offs[b]:offs[b + 1]], vg["voxel_size"], vg["range"], int(vg["max_points_in_voxel"]), mv)
                fs.append(self.reader(voxels, num))
                cs.append(coors)
                nvs.append(nv)
            seg_f, seg_c, counts = torch.stack(fs).contiguous(), torch.stack(cs).contiguous(), torch.cat(nvs).contiguous()
            feats = torch.zeros((batch * mv, seg_f.shape[2]), dtype=torch.float32, device=points.device)
            coords4 = torch.zeros((batch * mv, 4), dtype=torch.int32, device=points.device)
            nv = torch.empty((1,), dtype=torch.int32, device=points.device)
            hip.call("pn_concat_voxel_segments_f32", seg_f.data_ptr(), seg_c.data_ptr(), counts.data_ptr(), batch, mv, int(seg_f.shape[2]),
                     feats.data_ptr(), coords4.data_ptr(), nv.data_ptr(), hip.stream())
        out = self.dense_stages_nhwc(self.backbone.forward_nhwc(feats, coords4, batch, grid, n_voxels=nv))
        return {k: v.permute(0, 3, 1, 2) for k, v in out.items()}

    def extract_feat_dynamic(self, data):
        """voxelnet.py:228-237: dynamic voxel encoder -> sparse backbone on unq -> RPN.  As in the reference this branch does NOT pass through
        the re-alignment attention (only ``extract_feat_hard`` does); returns the NHWC neck output"""
        feats, unq = self.reader(data)
        x = self.backbone.forward_nhwc(feats, unq.to(torch.int32), data["batch_size"], [int(v) for v in data["grid_size"]])
        return self.neck.forward_nhwc(x) if self.with_neck else x

    def forward(self, example, return_loss=True, **kwargs):
        """voxelnet.py:239-301.  Hard-voxel branch (the Waymo PARTNER config): example keys voxels (V,P,F), coordinates (V,4) [b,z,y,x],
        num_points (V,), num_voxels (B,), shape [[x,y,z]]; dynamic branch (r6): points, grid_ind, num_points, voxel_size, pc_range, grid_size"""
        if "voxels" not in example:
            hip.require_device(example["points"], example["grid_ind"])
            data = dict(points=example["points"], grid_ind=example["grid_ind"], num_points=example["num_points"],
                        batch_size=len(example["num_points"]), voxel_size=example["voxel_size"][0], pc_range=example["pc_range"][0],
                        grid_size=example["grid_size"][0])
            x = self.extract_feat_dynamic(data)
        else:
            hip.require_device(example["voxels"], example["coordinates"])
            data = dict(features=example["voxels"], num_voxels=example["num_points"], coors=example["coordinates"],
                        batch_size=len(example["num_voxels"]), input_shape=[int(v) for v in example["shape"][0]])
            x = None
        if x is None and hasattr(self.bbox_head, "forward_nhwc"):
            feats = self.reader(data["features"], data["num_voxels"])
            head_out = self.dense_stages_nhwc(self.backbone.forward_nhwc(feats, data["coors"], data["batch_size"], data["input_shape"]))
        else:
            x = self.extract_feat_hard(data) if x is None else x
            head_out = self.bbox_head.forward_nhwc(x) if hasattr(self.bbox_head, "forward_nhwc") else None
        if head_out is not None:
            head_out.pop("_feat", None)
            preds = {"det_preds": [{k: v.permute(0, 3, 1, 2) for k, v in head_out.items()}]}
        else:
            preds = self.bbox_head(ops.as_nchw(x))
        if return_loss:
            return self.bbox_head.loss(example, preds)
        if kwargs.get("raw_preds", False) or self.test_cfg is None:
            return preds
        return {"det": self.bbox_head.predict(example, preds, self.test_cfg, **kwargs)}
