"""Training form of SetBlock (det3d/models/utils/set_transformer.py:118-166 SetAttention.forward, 216-259 RangeAttention,
307-354 SectorAttention, 392-440 SectorAttentionV2, under autograd in the reference): the same arithmetic as the fused
inference kernels of csrc/attention.hip, composed from differentiable primitives on the tape of autodiff.py so that every
parameter of the block receives its gradient.

Layout notes (B samples, H range rows, W azimuth columns, C channels, K key points per column, heads x hd = C):
  * tokens are (B, H, W, C) range-major, as in the reference; odd blocks work in the frame rolled by -W_win/2 along azimuth
  * the reference's permutes (`_cols`, the window split, and the RAW (B, C, K, W) reinterpretation of the key-point buffer at
    set_transformer.py:331-334 / 417-425) are index maps; they are handed to pn_contract_f32 as strides, no copies
  * score maps are stored (group, query, key, head) so that the relative-position bias (one row of `heads` values per
    (query, key) pair from the Conv1d MLP) is added elementwise and the softmax runs over the key axis with inner stride heads
  * BatchNorm1d of the position MLPs uses batch statistics in training (torch semantics of module.train())
Dropout (drop / attn_drop) and DropPath draw from the counter-based generator of pn_dropout_f32 (different stream from
torch's Philox, same distribution); with all three rates 0 the block is deterministic and is what the parity tests compare."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import autodiff as ad
from . import hip, ops
from .attention import SetBlock


class _Params:
    """tape leaves for the parameters of a module tree, by dotted name"""

    def __init__(self, tape: ad.Tape, module: torch.nn.Module, prefix: str):
        self.nodes: Dict[str, ad.Node] = {}
        for name, p in module.named_parameters():
            self.nodes[name] = tape.param(p.data, prefix + name)

    def __call__(self, name: str) -> ad.Node:
        return self.nodes[name]

    def maybe(self, name: str) -> Optional[ad.Node]:
        return self.nodes.get(name)


def _lin(t, P, prefix, x, k_pad=None):
    return ad.linear(t, x, P(prefix + ".weight"), P.maybe(prefix + ".bias"), k_pad=k_pad)


def _conv1d(t, P, prefix, x, k_pad=None):
    """Conv1d(kernel 1) over rows: weight (out, in, 1) viewed (out, in)"""
    w = P(prefix + ".weight")
    return ad.linear(t, x, t.reshaped(w, w.v.shape[:2]), P.maybe(prefix + ".bias"), k_pad=k_pad)


def _mlp(t, P, prefix, x, drop, seeds):
    h = ad.gelu(t, _lin(t, P, prefix + ".fc1", x))
    h = ad.dropout(t, h, drop, seeds())
    return ad.dropout(t, _lin(t, P, prefix + ".fc2", h), drop, seeds())


def _pos_bias(t, P, mod, prefix, rel: torch.Tensor, training: bool):
    """rel (pairs, 4) [dx, dy, 0, 0] -> (pairs, heads): Conv1d(2->16) + BatchNorm1d + ReLU + Conv1d(16->heads)"""
    x = t.const(rel)
    y = _conv1d(t, P, prefix + ".0", x, k_pad=rel.shape[1])
    y = ad.batchnorm_rows(t, y, mod[1], P(prefix + ".1.weight"), P(prefix + ".1.bias"), training)
    return _conv1d(t, P, prefix + ".3", y)


class _Seeds:
    def __init__(self, seed: int):
        self.seed, self.n = int(seed), 0

    def __call__(self) -> int:
        self.n += 1
        return self.seed * 1000003 + self.n


def set_block_train(t: ad.Tape, block: SetBlock, x: ad.Node, batch: int, prefix="", drop=0.0, attn_drop=0.0, drop_path=0.0, seed=0,
                    bn_training=True) -> ad.Node:
    """x: node holding (B*H*W, C) tokens -> node of the same shape; records the backward on ``t``.
    ``block.last_top_idx`` receives the selected key-point rows (B, K, W) as in the inference forward."""
    a = block.attns
    H, W = block.patches_resolution
    B, L = batch, H * W
    C = block.in_dim
    heads, K, win_w = a.num_heads, a.H, a.W
    hd, scale, sh = C // heads, float(a.scale), a.shift_size
    nw, n = W // win_w, K * win_w
    assert x.v.shape == (B * L, C) and x.v.is_contiguous()
    hip.require_device(x.v)
    P = _Params(t, a, prefix + "attns.")
    seeds = _Seeds(seed)
    sample = L * C
    dev = x.v.device
    pos = block.__dict__.get("_pos_dev")          # the cells' Cartesian positions: a constant, copied to the device once
    if pos is None or pos.device != dev or pos.shape[:2] != (H, W):
        pos = block.pos_cart.reshape(H, W, 2).to(dev).float().contiguous()
        block.__dict__["_pos_dev"] = pos

    xn, cm = ad.layernorm(t, x, P("norm1.weight"), P("norm1.bias"), a.norm1.eps, want_chan_mean=True)
    if sh:   # work in the rolled frame (set_transformer.py:121-124); rolled back before the output projection
        xn = ad.roll_w(t, xn, B, H, W, C, -sh)
        cm = ad.roll_w_raw(cm, B, H, W, 1, -sh)
        pos = ad.roll_w_raw(pos, 1, H, W, 2, -sh)
    kp, kpos, top = ad.gather_keypoints(t, xn, cm, pos, B, H, W, C, K)
    block.last_top_idx = top

    # strides (in floats) of the index maps
    col = dict(g=[L * C, C, hd], row=[W * C, 0], d=[1, 0])                       # (B, H*W, C) tokens as per-column windows, rows = range
    raw = dict(g=[K * W * C, 1, hd * K * W], row=[W, 0], d=[K * W, 0])           # raw (B, C, K, W) view of a (B, K*W, C) buffer, rows = key point
    kpt = dict(g=[K * W * C, C, hd], row=[W * C, 0], d=[1, 0])                   # (B, K, W, C) key-point buffer per column, rows = key point
    win = dict(g=[K * W * C, win_w * C, hd], row=[W * C, C], d=[1, 0])           # (B, K, W, C) as K x win_w windows, rows = (k, ww)

    def attention(q, qmap, k, v, kvmap, g_dims, m_dims, n_dims, bias, out_map, out_rows):
        """softmax(scale * q k^T + bias) v with the maps above; score layout (g0, g1, m, n, heads)"""
        M, N = m_dims[0] * m_dims[1], n_dims[0] * n_dims[1]
        g0, g1 = g_dims
        s_str = [g1 * M * N * heads, M * N * heads, 1, m_dims[1] * N * heads, N * heads, n_dims[1] * heads, heads]
        s = ad.contract(t, q, qmap["g"] + qmap["row"] + qmap["d"], k, kvmap["g"] + kvmap["row"] + kvmap["d"], (g0 * g1 * M * N, heads), s_str,
                        [g0, g1, heads] + m_dims + n_dims + [hd, 1], alpha=scale)
        s = ad.add(t, s, bias)
        p = ad.softmax(t, s, g0 * g1 * M, N, heads)
        p = ad.dropout(t, p, attn_drop, seeds())
        # out[g, m, d] = sum_n p[g, m, n] v[g, n, d]: the contraction index is the key row
        return ad.contract(t, p, s_str[:3] + s_str[3:5] + s_str[5:7], v, kvmap["g"] + kvmap["d"] + kvmap["row"], (out_rows, C),
                           out_map["g"] + out_map["row"] + out_map["d"], [g0, g1, heads] + m_dims + [hd, 1] + n_dims)

    # ---- sector attention 1: key points <- their range column (set_transformer.py:307-354)
    s1m = a.sector_attn1
    rel = ad.pair_diff(kpos, [K * W * 2, 2, W * 2, 0], pos, [0, 2, W * 2, 0], [B, W, K, 1, H, 1])
    bias = _pos_bias(t, P, s1m.pos_embedding_cart, "sector_attn1.pos_embedding_cart", rel, bn_training)
    o = attention(_lin(t, P, "sector_attn1.proj_q", kp), raw, _lin(t, P, "sector_attn1.proj_k", xn), _lin(t, P, "sector_attn1.proj_v", xn), col,
                  [B, W], [K, 1], [H, 1], bias, kpt, B * K * W)
    o = _lin(t, P, "sector_attn1.proj", o)   # (proj_drop is constructed by the reference but never applied)
    s1 = ad.add(t, kp, ad.dropout(t, o, drop_path, seeds(), row_len=K * W * C))
    m1 = _mlp(t, P, "sector_attn1.mlp", ad.layernorm(t, s1, P("sector_attn1.norm2.weight"), P("sector_attn1.norm2.bias"), s1m.norm2.eps), drop, seeds)
    s1 = ad.add(t, s1, ad.dropout(t, m1, drop_path, seeds(), row_len=K * W * C))

    # ---- range attention among the key points of win_w neighbouring columns (set_transformer.py:216-259)
    ram = a.range_attn
    sn = ad.layernorm(t, s1, P("range_attn.norm1.weight"), P("range_attn.norm1.bias"), ram.norm1.eps)
    wstr = [K * W * 2, win_w * 2, W * 2, 2]
    rel = ad.pair_diff(kpos, wstr, kpos, wstr, [B, nw, K, win_w, K, win_w])
    bias = _pos_bias(t, P, ram.pos_embedding_cart, "range_attn.pos_embedding_cart", rel, bn_training)
    o = attention(_lin(t, P, "range_attn.proj_q", sn), win, _lin(t, P, "range_attn.proj_k", sn), _lin(t, P, "range_attn.proj_v", sn), win,
                  [B, nw], [K, win_w], [K, win_w], bias, win, B * K * W)
    o = _lin(t, P, "range_attn.proj", o)
    s2 = ad.add(t, s1, ad.dropout(t, o, drop_path, seeds(), row_len=K * W * C))
    m2 = _mlp(t, P, "range_attn.mlp", ad.layernorm(t, s2, P("range_attn.norm2.weight"), P("range_attn.norm2.bias"), ram.norm2.eps), drop, seeds)
    s2 = ad.add(t, s2, ad.dropout(t, m2, drop_path, seeds(), row_len=K * W * C))

    # ---- sector attention 2: every cell of the column <- the refined key points (set_transformer.py:392-440)
    rel = ad.pair_diff(pos, [0, 2, W * 2, 0], kpos, [K * W * 2, 2, W * 2, 0], [B, W, H, 1, K, 1])
    bias = _pos_bias(t, P, a.sector_attn2.pos_embedding_cart, "sector_attn2.pos_embedding_cart", rel, bn_training)
    o = attention(_lin(t, P, "sector_attn2.proj_q", xn), col, _lin(t, P, "sector_attn2.proj_k", s2), _lin(t, P, "sector_attn2.proj_v", s2), raw,
                  [B, W], [H, 1], [K, 1], bias, col, B * L)
    if sh:
        o = ad.roll_w(t, o, B, H, W, C, sh)
    o = _lin(t, P, "proj", o)
    y = ad.add(t, x, ad.dropout(t, o, drop_path, seeds(), row_len=sample))
    m3 = _mlp(t, P, "mlp", ad.layernorm(t, y, P("norm2.weight"), P("norm2.bias"), a.norm2.eps), drop, seeds)
    return ad.add(t, y, ad.dropout(t, m3, drop_path, seeds(), row_len=sample))
