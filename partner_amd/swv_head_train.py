"""Training form of E2ESWVoteHead's forward (det3d/models/bbox_heads/e2e_swv_head.py:150-173 and the shifted-window stage
det3d/models/bbox_heads/swin_utils/sw2votev4_util.py:42-419, under autograd in the reference), composed on the tape of
autodiff.py; the set criterion that consumes its outputs and returns their gradients is swv_head.E2ESWVoteHead.loss
(csrc/e2e_loss.hip).  The reference class cannot be imported: PARITY UNPINNED; the checker is fp64 autograd over
``oracle/polar_oracle.py::e2e_swv_head(train=True)``.

Window attention as strides: tokens are zero-padded to multiples of the window, rolled by (-shift, -shift) (pn_pad_roll_f32) and
kept as ONE (B, Hp, Wp, C) array; a window is then the index map (b*nWh + wi | wj | head ; row r, col c ; d) that
pn_contract_f32 takes as strides, so neither window_partition nor window_reverse copies anything.  Scores are stored
(window, query, key, head): the per-head temperature is a channel scale, the relative-position MLP output (one row of `heads`
values per (query, key) pair, shared by all samples) and the shift mask are added as a broadcast bias, softmax runs over the key
axis.  Cosine attention normalises q and k per head (F.normalize); the reference divides by max(|q||k|, 1e-6) instead, which is
the same value unless a norm underflows."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import autodiff as ad
from . import hip
from .swv_head import E2ESWVoteHead


_SHIFT_MASKS: dict = {}


def _shift_mask(hp: int, wp: int, ws: int, shift: int, heads: int, dev) -> torch.Tensor:
    """(nW * N * N, heads) additive mask of BasicLayer.forward (sw2votev4_util.py:259-276): 0 inside a region, -100 across.  A constant of
    the geometry: built once per (map, window, shift, heads, device) -- a numpy build + host-to-device copy per iteration is a host sync"""
    key = (hp, wp, ws, shift, heads, str(dev))
    if key not in _SHIFT_MASKS:
        _SHIFT_MASKS[key] = _build_shift_mask(hp, wp, ws, shift, heads, dev)
    return _SHIFT_MASKS[key]


def _build_shift_mask(hp: int, wp: int, ws: int, shift: int, heads: int, dev) -> torch.Tensor:
    img = np.zeros((hp, wp), np.int32)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    win = img.reshape(hp // ws, ws, wp // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    am = np.where(win[:, :, None] != win[:, None, :], -100.0, 0.0).astype(np.float32)   # [window][query][key]
    return torch.from_numpy(np.repeat(am.reshape(-1, 1), heads, 1)).to(dev).contiguous()


def e2e_swv_head_train(t: ad.Tape, head: E2ESWVoteHead, x: ad.Node, prefix="bbox_head.") -> Dict[str, ad.Node]:
    """x: node holding the NHWC neck output (B, H, W, Cin) -> nodes pred_centers (B,H,W,2), pred_vote_cls (B,H,W,1), hm (B,H,W,ncls),
    boxes (B,H,W,code_size) [reg | height | dim | rot], iou (B,H,W,1) when the head has one"""
    hip.require_device(x.v)
    B, H, W, cin = x.v.shape
    dev = x.v.device
    P: Dict[str, ad.Node] = {name: t.param(p.data, prefix + name) for name, p in head.named_parameters()}
    L = head.layer
    C, heads, ws = L.embed_dim, L.num_heads, head.window_size
    d = C // heads
    N = ws * ws
    n = B * H * W
    hp, wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    nwh, nww = hp // ws, wp // ws
    pos = head.offset_grid[0].permute(1, 2, 0).contiguous().float().to(dev)   # (H, W, 2)
    if tuple(pos.shape[:2]) != (H, W):
        raise ValueError(f"head input map {(H, W)} does not match the configured offset grid {tuple(pos.shape[:2])}")

    def conv(name, inp, relu=False):
        m = head.get_submodule(name)
        return ad.conv2d(t, inp, P[name + ".weight"], P.get(name + ".bias"), stride=1, pad=m.padding[0], relu=relu)

    def bn(name, inp):
        return ad.batchnorm2d(t, inp, head.get_submodule(name), P[name + ".weight"], P[name + ".bias"], relu=True)

    def lin(name, inp, relu=False, k_pad=None, as_matrix=False):
        w = P[name + ".weight"]
        if as_matrix:   # Conv1d / Conv2d with kernel 1 used as a matrix
            w = t.reshaped(w, w.v.shape[:2])
        return ad.linear(t, inp, w, P.get(name + ".bias"), k_pad=k_pad, relu=relu)

    def ln(name, inp):
        return ad.layernorm(t, inp, P[name + ".weight"], P[name + ".bias"], head.get_submodule(name).eps)

    # ---- vote branch (e2e_swv_head.py:152-156)
    centers = conv("vote_head.2", conv("vote_head.0", x, relu=True))
    vote_cls = conv("vote_cls_head.3", bn("vote_cls_head.1", conv("vote_cls_head.0", x)))
    vote4 = ad.view(t, ad.concat_channels(t, [centers, vote_cls], 4), (n, 4))

    # ---- shifted-window stage
    tok = lin("layer.patch_embed.proj", ad.view(t, x, (n, cin)), as_matrix=True)
    tok = ln("layer.patch_embed.norm", tok)
    tok_map = dict(g=[ws * wp * C, ws * C, d], row=[wp * C, C], d=[1, 0])
    s_str = [nww * N * N * heads, N * N * heads, 1, ws * N * heads, N * heads, ws * heads, heads]
    g_dims = [B * nwh, nww, heads]
    for i, blk in enumerate(L.layers[0].blocks):
        bp = f"layer.layers.0.blocks.{i}."
        shift = int(blk.shift_size)
        y = ad.pad_roll(t, ln(bp + "norm1", tok), B, H, W, hp, wp, C, shift)
        vp = ad.pad_roll(t, vote4, B, H, W, hp, wp, 4, shift)
        posp = ad.pad_roll_raw(pos, 1, H, W, hp, wp, 2, shift)                 # (hp*wp, 2), shared by the samples
        ve = lin(bp + "attn.vote_mlp.2", lin(bp + "attn.vote_mlp.0", vp, relu=True, k_pad=4, as_matrix=True), as_matrix=True)
        qkv_w, qkv_b = P[bp + "attn.qkv.weight"], P.get(bp + "attn.qkv.bias")
        parts = []
        for k in range(3):
            wk = t.sliced(qkv_w, k * C, (k + 1) * C)
            bk = None if qkv_b is None else t.sliced(qkv_b, k * C, (k + 1) * C)
            parts.append(ad.add(t, ad.linear(t, y, wk, bk), ve))
        q, k_, v = parts
        rows = B * hp * wp
        qn = ad.view(t, ad.l2_normalize(t, ad.view(t, q, (rows * heads, d))), (rows, C))
        kn = ad.view(t, ad.l2_normalize(t, ad.view(t, k_, (rows * heads, d))), (rows, C))
        tm = tok_map["g"] + tok_map["row"] + tok_map["d"]
        s = ad.contract(t, qn, tm, kn, tm, (B * nwh * nww * N * N, heads), s_str, g_dims + [ws, ws, ws, ws, d, 1])
        s = ad.scale_channels(t, s, ad.recip_clamp(t, t.reshaped(P[bp + "attn.tau"], (heads,)), 0.01))
        # relative-position bias from pairwise Cartesian offsets inside each window (sw2votev4_util.py:86-92)
        pstr = [ws * wp * 2, ws * 2, wp * 2, 2]
        rel = ad.pair_diff(posp, pstr, posp, pstr, [nwh, nww, ws, ws, ws, ws])
        bias = lin(bp + "attn.rpe.2", lin(bp + "attn.rpe.0", t.const(rel), relu=True, k_pad=4, as_matrix=True), as_matrix=True)
        if shift > 0:
            bias = ad.add(t, bias, t.const(_shift_mask(hp, wp, ws, shift, heads, dev)))
        s = ad.add_broadcast(t, s, bias, B)
        p = ad.softmax(t, s, B * nwh * nww * N, N, heads)
        o = ad.contract(t, p, s_str, v, tok_map["g"] + tok_map["d"] + tok_map["row"], (rows, C), tm, g_dims + [ws, ws, d, 1, ws, ws])
        o = lin(bp + "attn.proj", ad.crop_roll(t, o, B, H, W, hp, wp, C, shift))
        tok = ad.add(t, tok, o)
        z = lin(bp + "mlp.fc2", ad.gelu(t, lin(bp + "mlp.fc1", ln(bp + "norm2", tok))))
        tok = ad.add(t, tok, z)
    feat = ad.view(t, ln("layer.norm0", tok), (B, H, W, C))

    # ---- prediction branches (e2e_swv_head.py:158-171)
    h = feat
    for i in range(2):
        h = bn(f"cls_head.{i}.1", conv(f"cls_head.{i}.0", h))
    out = dict(pred_centers=centers, pred_vote_cls=vote_cls, hm=conv("cls_head.2", h),
               boxes=conv("bbox_head.2", conv("bbox_head.0", feat, relu=True)))
    if head.iou_loss:
        out["iou"] = conv("iou_head.2", conv("iou_head.0", feat, relu=True))
    return out
