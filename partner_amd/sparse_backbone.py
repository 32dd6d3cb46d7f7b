"""SpMiddleResNetFHD: the sparse 3-D middle encoder of the Waymo PARTNER config.

Reference: det3d/models/backbones/scn.py:17-192 (SparseBasicBlock, SpMiddleResNetFHD); the convolution
arithmetic there is the third-party ``spconv`` package (SubMConv3d / SparseConv3d / SparseConvTensor), which is
not part of the reference tree -- parity is against ``oracle/polar_oracle.py::sp_middle_resnet_fhd`` (dense
restatement with activity masks), unpinned by the reference.

Module / parameter names follow the reference (``conv_input.0.weight``, ``conv2.3.conv1.weight`` ...), weights in
spconv 2.x layout (Cout, kD, kH, kW, Cin).  Forward only, eval mode, on libpartner_hip:
bitmap-rank active-site index per resolution level, one neighbour table per ``indice_key``, every convolution a
gathered GEMM on the MFMA kernel with BatchNorm folded and ReLU / residual fused.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List

import numpy as np
import torch
from torch import nn

from . import hip, ops
from .routes import R
from .builder import BACKBONES
from .nn_utils import PlanCache, build_norm_layer, eval_only



class SubMConv3d(nn.Module):
    """parameters of spconv.SubMConv3d (submanifold convolution: the active set does not change)"""

    def __init__(self, cin, cout, kernel_size, bias=True, indice_key=None):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        self.kernel_size, self.stride, self.padding, self.indice_key = k, (1, 1, 1), tuple(v // 2 for v in k), indice_key
        self.weight = nn.Parameter(torch.empty(cout, *k, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)


class SparseConv3d(nn.Module):
    """parameters of spconv.SparseConv3d (strided: output sites = receptive fields that hold an active input)"""

    def __init__(self, cin, cout, kernel_size, stride, padding=0, bias=True):
        super().__init__()
        t3 = lambda v: (v,) * 3 if isinstance(v, int) else tuple(v)  # noqa: E731
        self.kernel_size, self.stride, self.padding = t3(kernel_size), t3(stride), t3(padding)
        self.weight = nn.Parameter(torch.empty(cout, *self.kernel_size, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)


class SparseBasicBlock(nn.Module):
    """scn.py:52-95"""

    def __init__(self, inplanes, planes, norm_cfg=None, indice_key=None):
        super().__init__()
        norm_cfg = norm_cfg or dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.conv1 = SubMConv3d(inplanes, planes, 3, bias=True, indice_key=indice_key)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, 3, bias=True, indice_key=indice_key)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]


@BACKBONES.register_module
class SpMiddleResNetFHD(nn.Module):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", **kwargs):
        super().__init__()
        self.name = name
        self.extra_sp_shape = list(kwargs.get("extra_sp_shape", [1, 0, 0]))
        norm_cfg = norm_cfg or dict(type="BN1d", eps=1e-3, momentum=0.01)
        bn = lambda c: build_norm_layer(norm_cfg, c)[1]  # noqa: E731
        blk = lambda c, key: SparseBasicBlock(c, c, norm_cfg=norm_cfg, indice_key=key)  # noqa: E731
        self.conv_input = nn.Sequential(SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"), bn(16), nn.ReLU(inplace=True))
        self.conv1 = nn.Sequential(blk(16, "res0"), blk(16, "res0"))
        self.conv2 = nn.Sequential(SparseConv3d(16, 32, 3, 2, padding=1, bias=False), bn(32), nn.ReLU(inplace=True), blk(32, "res1"), blk(32, "res1"))
        self.conv3 = nn.Sequential(SparseConv3d(32, 64, 3, 2, padding=1, bias=False), bn(64), nn.ReLU(inplace=True), blk(64, "res2"), blk(64, "res2"))
        pad = 1 if self.extra_sp_shape[0] == 0 else 0
        self.conv4 = nn.Sequential(SparseConv3d(64, 128, 3, 2, padding=[pad, 1, 1], bias=False), bn(128), nn.ReLU(inplace=True), blk(128, "res3"),
                                   blk(128, "res3"))
        self.extra_conv = nn.Sequential(SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), bn(128), nn.ReLU())
        self._plan = PlanCache()

    # ---------------------------------------------------------------------------------------
    @staticmethod
    def _pack(conv, bnorm, cin_pad=None):
        """(packed weight, scale, shift): BatchNorm1d (eval) and the conv bias folded into the epilogue"""
        w = conv.weight.detach().float()                      # (Cout, kd, kh, kw, Cin)
        cout, cin = w.shape[0], w.shape[4]
        taps = w.shape[1] * w.shape[2] * w.shape[3]
        w2 = w.reshape(cout, taps, cin).permute(0, 2, 1).contiguous()   # (Cout, Cin, taps)
        if cin_pad is not None and cin_pad != cin:
            w2 = torch.cat([w2, torch.zeros((cout, cin_pad - cin, taps), dtype=w2.dtype, device=w2.device)], 1).contiguous()
            cin = cin_pad
        lib = hip.load()
        packed = torch.empty(lib.pn_conv_packed_weight_floats(cout, cin, taps, 1, 1), dtype=torch.float32, device=w.device)
        hip.call("pn_pack_conv_weight_f32", w2.data_ptr(), cout, cin, taps, 1, 1, packed.data_ptr(), hip.stream())
        scale, shift = ops.fold_bn(bnorm.weight, bnorm.bias, bnorm.running_mean, bnorm.running_var, bnorm.eps, conv.bias)
        return dict(packed=packed, scale=scale, shift=shift, cin=cin, cout=cout, taps=taps, geo=(conv.kernel_size, conv.stride, conv.padding))

    def _build_plan(self):
        cin0 = self.conv_input[0].weight.shape[4]
        plan = dict(cin0=(cin0 + 3) // 4 * 4, input=self._pack(self.conv_input[0], self.conv_input[1], (cin0 + 3) // 4 * 4), stages=[])

        def blocks(seq):
            return [(self._pack(b.conv1, b.bn1), self._pack(b.conv2, b.bn2)) for b in seq if isinstance(b, SparseBasicBlock)]

        plan["stages"].append(dict(down=None, blocks=blocks(self.conv1)))
        for seq in (self.conv2, self.conv3, self.conv4):
            plan["stages"].append(dict(down=self._pack(seq[0], seq[1]), blocks=blocks(seq)))
        plan["extra"] = self._pack(self.extra_conv[0], self.extra_conv[1])
        return plan

    @staticmethod
    def _out_dims(dims, geo):
        k, s, p = geo
        return [dims[0]] + [(dims[1 + a] + 2 * p[a] - k[a]) // s[a] + 1 for a in range(3)]

    @staticmethod
    def _conv(feats, n_rows, nbr, count, cap, layer, act, residual=None, groups=None):
        out = torch.empty((cap, layer["cout"]), dtype=torch.float32, device=feats.device)
        if groups is not None and layer["cin"] % 16 == 0 and layer["cout"] in (32, 64, 128) and layer["taps"] <= 27:
            # 32 / 64 / 128-channel levels and the strided stages: one wave per group of 32 similar sites, the group's taps only (sparse_group.hip)
            hip.call("pn_sparse_conv_grouped_f32", feats.data_ptr(), n_rows, layer["cin"], nbr.data_ptr(), count.data_ptr(), cap, layer["taps"],
                     groups[0].data_ptr(), groups[1].data_ptr(), groups[2].data_ptr() if len(groups) > 2 else None, layer["packed"].data_ptr(),
                     layer["cout"], layer["scale"].data_ptr(),
                     layer["shift"].data_ptr(), int(act), hip.ptr(residual), out.data_ptr(), hip.stream())
            return out
        if R.sparse_c16 and layer["cout"] == 16 and layer["cin"] in (8, 16) and layer["taps"] <= 27:
            # conv_input / conv1 (scn.py:112-123): the 16-channel level has its own kernel (inference; the training tape keeps one form)
            hip.call("pn_sparse_conv_c16_f32", feats.data_ptr(), n_rows, layer["cin"], nbr.data_ptr(), count.data_ptr(), cap, layer["taps"],
                     layer["packed"].data_ptr(), layer["scale"].data_ptr(), layer["shift"].data_ptr(), int(act), hip.ptr(residual), out.data_ptr(),
                     hip.stream())
            return out
        hip.call("pn_sparse_conv_f32", feats.data_ptr(), n_rows, layer["cin"], nbr.data_ptr(), count.data_ptr(), cap, layer["taps"],
                 layer["packed"].data_ptr(), layer["cout"], layer["scale"].data_ptr(), layer["shift"].data_ptr(), int(act), hip.ptr(residual),
                 out.data_ptr(), hip.stream())
        return out

    @staticmethod
    def _i3(v):
        return (C.c_int32 * 3)(*[int(x) for x in v])

    def _neighbors(self, keys, cap, count, out_dims, in_index, in_dims, geo, out=None, row_bits=None):
        taps = geo[0][0] * geo[0][1] * geo[0][2]
        nbr = out if out is not None else torch.empty((cap, taps), dtype=torch.int32, device=keys.device)
        if row_bits is not None:      # (+ one byte per (site, row of taps) for the grouping sort)
            hip.call("pn_sparse_neighbors_rows", keys.data_ptr(), cap, count.data_ptr(), (C.c_int32 * 4)(*out_dims), in_index.data_ptr(),
                     (C.c_int32 * 4)(*in_dims), self._i3(geo[0]), self._i3(geo[1]), self._i3(geo[2]), nbr.data_ptr(), row_bits.data_ptr(), hip.stream())
            return nbr
        hip.call("pn_sparse_neighbors", keys.data_ptr(), cap, count.data_ptr(), (C.c_int32 * 4)(*out_dims), in_index.data_ptr(),
                 (C.c_int32 * 4)(*in_dims), self._i3(geo[0]), self._i3(geo[1]), self._i3(geo[2]), nbr.data_ptr(), hip.stream())
        return nbr

    def forward_nhwc(self, voxel_features: torch.Tensor, coors: torch.Tensor, batch_size: int, input_shape, n_voxels: torch.Tensor = None):
        """voxel_features (V,C) f32, coors (V,4) int [b,z,y,x], input_shape [x,y,z] -> NHWC (B, H', W', 128*D') dense BEV map.
        ``n_voxels``: optional device int32 count (rows beyond it are ignored); no host synchronisation inside."""
        eval_only(self, "SpMiddleResNetFHD")
        hip.require_device(voxel_features, coors)
        lib = hip.load()
        plan = self._plan.get(self, self._build_plan)
        dev = voxel_features.device
        st = hip.stream()
        V, cin = voxel_features.shape
        D, H, W = (int(v) + e for v, e in zip(list(input_shape)[::-1], self.extra_sp_shape))
        dims = [int(batch_size), D, H, W]
        coors = coors.to(torch.int32).contiguous()
        if n_voxels is None:
            n_voxels = torch.full((1,), V, dtype=torch.int32, device=dev)

        def new_index(d):
            cells = d[0] * d[1] * d[2] * d[3]
            return torch.empty(lib.pn_sparse_index_bytes(cells), dtype=torch.uint8, device=dev)

        # ---- the STRUCTURE of every level (active sets of the strided stages, neighbour tables) depends on the voxel coordinates only, not on
        # the features: it is built on a second stream beside the convolutions of the earlier levels (r3; 0.8 ms of small latency-bound
        # launches left the critical path of the frame).  Every buffer is allocated here, on the calling stream, before the fork; a level's
        # convolutions wait for that level's event.  Inside a hipGraph capture the fork / joins become graph dependencies.
        subm_geo = ((3, 3, 3), (1, 1, 1), (1, 1, 1))
        main = torch.cuda.current_stream()
        side = ops.concurrent_stream(dev) if R.sparse_struct_stream else None     # a stream that really overlaps with this one (hardware-queue mapping)

        def tbl(rows, geo):
            return torch.empty((rows, geo[0][0] * geo[0][1] * geo[0][2]), dtype=torch.int32, device=dev)

        def grp(rows):      # (perm, group masks) of a neighbour table: the sites sorted by neighbourhood, pn_sparse_group_rows
            if not R.sparse_grouped:
                return None
            # (perm, group masks, XCD cut points of equal work: pn_sparse_group_balance)
            return (torch.empty(rows, dtype=torch.int32, device=dev), torch.empty((rows + 31) // 32, dtype=torch.int32, device=dev),
                    torch.zeros(18, dtype=torch.int32, device=dev))

        def bits(rows, geo, g):      # the row bytes the neighbour kernel leaves for the sort (only where a sort follows)
            return torch.empty((rows, geo[0][0] * geo[0][1]), dtype=torch.uint8, device=dev) if (g is not None and R.sparse_row_bits and geo[0][2] <= 8) else None

        def group(table, g, level, rb=None, geo=None):
            if g is None:
                return
            if rb is not None:
                hip.call("pn_sparse_group_rows_bits", rb.data_ptr(), rb.shape[1], geo[0][2], level["count"].data_ptr(), level["cap"], g[0].data_ptr(),
                         g[1].data_ptr(), hip.stream())
            else:
                hip.call("pn_sparse_group_rows", table.data_ptr(), level["count"].data_ptr(), level["cap"], table.shape[1], g[0].data_ptr(), g[1].data_ptr(),
                         hip.stream())
            hip.call("pn_sparse_group_balance", g[1].data_ptr(), level["count"].data_ptr(), level["cap"], g[2].data_ptr(), hip.stream())

        levels = []          # per level: dict(index, keys, count, cap, dims, nbr, down=(dnbr) or None)
        cap = V
        lv = dict(index=new_index(dims), keys=torch.empty(cap, dtype=torch.int32, device=dev), count=torch.empty(1, dtype=torch.int32, device=dev),
                  cap=cap, dims=dims, dnbr=None)
        lv["nbr"] = tbl(cap, subm_geo)
        lv["grp"], lv["dgrp"] = None, None      # level 0 is the 16-channel level (its own kernel)
        rank = torch.empty(cap, dtype=torch.int32, device=dev)
        levels.append(lv)
        geos = [stage["down"]["geo"] for stage in plan["stages"] if stage["down"] is not None] + [plan["extra"]["geo"]]
        for gi, geo in enumerate(geos):
            prev = levels[-1]
            odims = self._out_dims(prev["dims"], geo)
            ocap = int(min(odims[0] * odims[1] * odims[2] * odims[3], 8 * prev["cap"]))
            nl = dict(index=new_index(odims), keys=torch.empty(ocap, dtype=torch.int32, device=dev), count=torch.empty(1, dtype=torch.int32, device=dev),
                      cap=ocap, dims=odims, geo=geo, dnbr=tbl(ocap, geo))
            nl["nbr"] = tbl(ocap, subm_geo) if gi + 1 < len(geos) else None      # the last level (extra_conv) has no submanifold layers
            nl["dgrp"] = grp(ocap)
            nl["grp"] = grp(ocap) if nl["nbr"] is not None else None
            nl["dbits"], nl["bits"] = bits(ocap, geo, nl["dgrp"]), bits(ocap, subm_geo, nl["grp"])
            levels.append(nl)

        def build_structure():
            st = hip.stream()
            l0 = levels[0]
            hip.call("pn_sparse_index_from_coords", coors.data_ptr(), V, n_voxels.data_ptr(), (C.c_int32 * 4)(*dims), l0["index"].data_ptr(),
                     l0["keys"].data_ptr(), l0["count"].data_ptr(), rank.data_ptr(), st)
            l0["indexed"] = self._mark(side)      # the rank of every voxel is known: the features can go into key order beside the neighbour table
            self._neighbors(l0["keys"], l0["cap"], l0["count"], l0["dims"], l0["index"], l0["dims"], subm_geo, out=l0["nbr"])
            l0["ready"] = self._mark(side)
            for prev, cur in zip(levels[:-1], levels[1:]):
                geo = cur["geo"]
                hip.call("pn_sparse_index_downsample", prev["keys"].data_ptr(), prev["cap"], prev["count"].data_ptr(), (C.c_int32 * 4)(*prev["dims"]),
                         self._i3(geo[0]), self._i3(geo[1]), self._i3(geo[2]), (C.c_int32 * 4)(*cur["dims"]), cur["index"].data_ptr(),
                         cur["keys"].data_ptr(), cur["cap"], cur["count"].data_ptr(), st)
                self._neighbors(cur["keys"], cur["cap"], cur["count"], cur["dims"], prev["index"], prev["dims"], geo, out=cur["dnbr"], row_bits=cur["dbits"])
                group(cur["dnbr"], cur["dgrp"], cur, cur["dbits"], geo)
                if cur["nbr"] is not None:
                    self._neighbors(cur["keys"], cur["cap"], cur["count"], cur["dims"], cur["index"], cur["dims"], subm_geo, out=cur["nbr"],
                                    row_bits=cur["bits"])
                    group(cur["nbr"], cur["grp"], cur, cur["bits"], subm_geo)
                cur["ready"] = self._mark(side)

        if side is None:
            build_structure()
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                build_structure()

        def wait(level):
            ev = level.get("ready")
            if ev is not None:
                main.wait_event(ev)

        # ---- features: into key order (zero padded to a multiple of 4 channels), then the convolutions level by level
        st = hip.stream()
        c0 = plan["cin0"]
        src = voxel_features.contiguous().float()
        if c0 != cin:
            src = torch.cat([src, torch.zeros((V, c0 - cin), dtype=torch.float32, device=dev)], 1).contiguous()
        feats = torch.zeros((cap, c0), dtype=torch.float32, device=dev)
        cur = levels[0]
        if cur.get("indexed") is not None:
            main.wait_event(cur["indexed"])
        else:
            wait(cur)
        hip.call("pn_sparse_permute_rows", src.data_ptr(), rank.data_ptr(), V, n_voxels.data_ptr(), c0, feats.data_ptr(), st)
        wait(cur)
        x = self._conv(feats, cur["cap"], cur["nbr"], cur["count"], cur["cap"], plan["input"], ops.ACT_RELU)
        li = 0
        for stage in plan["stages"]:
            if stage["down"] is not None:
                li += 1
                nxt = levels[li]
                wait(nxt)
                x = self._conv(x, cur["cap"], nxt["dnbr"], nxt["count"], nxt["cap"], stage["down"], ops.ACT_RELU, groups=nxt["dgrp"])
                cur = nxt
            for c1, c2 in stage["blocks"]:
                y = self._conv(x, cur["cap"], cur["nbr"], cur["count"], cur["cap"], c1, ops.ACT_RELU, groups=cur["grp"])
                x = self._conv(y, cur["cap"], cur["nbr"], cur["count"], cur["cap"], c2, ops.ACT_RELU, residual=x, groups=cur["grp"])
        last = levels[li + 1]
        wait(last)
        x = self._conv(x, cur["cap"], last["dnbr"], last["count"], last["cap"], plan["extra"], ops.ACT_RELU, groups=last["dgrp"])
        cch = plan["extra"]["cout"]
        odims = last["dims"]
        out = torch.empty((odims[0], odims[2], odims[3], cch * odims[1]), dtype=torch.float32, device=dev)
        if cch % 4 == 0:      # written from the output side through the level's index: one coalesced pass, no zero fill (r4)
            hip.call("pn_sparse_to_dense_index_nhwc", x.data_ptr(), last["index"].data_ptr(), (C.c_int32 * 4)(*odims), cch, out.data_ptr(), st)
        else:
            hip.call("pn_sparse_to_dense_nhwc", x.data_ptr(), last["keys"].data_ptr(), last["cap"], last["count"].data_ptr(), (C.c_int32 * 4)(*odims), cch,
                     out.data_ptr(), st)
        return out

    @staticmethod
    def _mark(side):
        if side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(side)
        return ev

    def forward(self, voxel_features, coors, batch_size, input_shape):
        """(ret (B, C*D, H, W) logical NCHW view, multi_scale_voxel_features) as scn.py:157-192; the multi-scale sparse tensors are
        only used by the reference's segmentation branch and are not materialised"""
        return ops.as_nchw(self.forward_nhwc(voxel_features, coors, batch_size, input_shape)), None
