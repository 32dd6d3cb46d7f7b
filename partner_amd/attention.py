"""PARTNER's global representation re-alignment (SetBlock) under the reference's names and
parameter tree (det3d/models/utils/set_transformer.py:37-493); forward = HIP kernels only:
MFMA GEMMs for every linear layer (bias / GELU / residual fused in the epilogue) plus the
LayerNorm, key-point selection and the three attention cores of csrc/attention.hip.

Only the configuration VoxelNetV3 builds is implemented (voxelnet.py:192-199): embed_dim_scale=1,
H_sp = full range column, W_sp = 1; dropout / drop-path are identities in eval mode."""
from __future__ import annotations

import torch
from torch import nn

from . import hip, ops
from .nn_utils import PlanCache, eval_only


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


def _pos_embedding(num_heads):
    return nn.Sequential(nn.Conv1d(2, 16, kernel_size=1), nn.BatchNorm1d(16), nn.ReLU(inplace=True),
                         nn.Conv1d(16, num_heads, kernel_size=1))


class RangeAttention(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.proj_q, self.proj_k, self.proj_v = (nn.Linear(dim, dim, bias=qkv_bias) for _ in range(3))
        self.proj = nn.Linear(dim, dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim, act_layer)
        self.norm2 = norm_layer(dim)
        self.pos_embedding_cart = _pos_embedding(num_heads)


class SectorAttention(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm, act_layer=nn.GELU):
        super().__init__()
        self.proj_q, self.proj_k, self.proj_v = (nn.Linear(dim, dim, bias=qkv_bias) for _ in range(3))
        self.proj = nn.Linear(dim, dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim, act_layer)
        self.norm2 = norm_layer(dim)
        self.pos_embedding_cart = _pos_embedding(num_heads)


class SectorAttentionV2(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.proj_q, self.proj_k, self.proj_v = (nn.Linear(dim, dim, bias=qkv_bias) for _ in range(3))
        self.pos_embedding_cart = _pos_embedding(num_heads)


class SetAttention(nn.Module):
    def __init__(self, dim, resolution, H_sp=144, W_sp=1, H=4, W=8, num_heads=8, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 norm_layer=nn.LayerNorm, act_layer=nn.GELU, shift=True):
        super().__init__()
        self.dim, self.resolution, self.num_heads = dim, tuple(resolution), num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.H_sp, self.W_sp, self.H, self.W = H_sp, W_sp, H, W
        self.shift_size = W // 2 if shift else 0
        self.norm1 = norm_layer(dim)
        self.proj = nn.Linear(dim, dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), dim, act_layer)
        self.norm2 = norm_layer(dim)
        self.pos_embedding_cart = _pos_embedding(num_heads)  # present (and unused) in the reference too
        self.range_attn = RangeAttention(dim, num_heads, mlp_ratio, qkv_bias, norm_layer, act_layer)
        self.sector_attn1 = SectorAttention(dim, num_heads, mlp_ratio, qkv_bias, norm_layer, act_layer)
        self.sector_attn2 = SectorAttentionV2(dim, num_heads, qkv_bias)


def _fold_pos_mlp(seq: nn.Sequential) -> torch.Tensor:
    """[w1(16x2) | bn_scale | bn_shift | w2(heads x 16) | b2] with BatchNorm1d(eval) and the first bias folded"""
    c1, bn, c2 = seq[0], seq[1], seq[3]
    scale, shift = ops.fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, c1.bias)
    return torch.cat([c1.weight.detach().reshape(-1), scale, shift, c2.weight.detach().reshape(-1), c2.bias.detach()]).contiguous()


def _cat_linear(*lins):
    w = torch.cat([l.weight.detach() for l in lins], 0)
    b = torch.cat([l.bias.detach() for l in lins], 0) if lins[0].bias is not None else None
    return ops.GemmLayer(w, b)


class SetBlock(nn.Module):
    def __init__(self, in_dim, embed_dim_scale, reso, num_heads, H_sp=4, W_sp=4, H=4, W=8, mlp_ratio=4.0, qkv_bias=False,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, pos=None,
                 shift=True):
        super().__init__()
        if embed_dim_scale != 1:
            raise NotImplementedError("SetBlock: only embed_dim_scale=1 (the VoxelNetV3 configuration) is implemented")
        if H_sp != reso[0] or W_sp != 1:
            raise NotImplementedError("SetBlock: sector windows must be whole range columns (H_sp = reso[0], W_sp = 1)")
        self.in_dim, self.embed_dim, self.num_heads = in_dim, in_dim, num_heads
        self.patches_resolution = tuple(reso)
        self.pos_cart = pos[..., :2]  # plain attribute, as in the reference
        self.attns = SetAttention(in_dim, reso, H_sp=H_sp, W_sp=W_sp, H=H, W=W, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                  qkv_bias=qkv_bias, qk_scale=qk_scale, norm_layer=norm_layer, act_layer=act_layer, shift=shift)
        self._plan = PlanCache()
        self.compute_dtype = "f32"

    def set_compute_dtype(self, dtype: str) -> "SetBlock":
        """"f32" (default, the reference's arithmetic) or "bf16": the five token GEMMs over all H x W tokens (key / value and query
        projections, output projection, the MLP) on the bf16 matrix pipe (pn_linear_bf16: bf16 operands, f32 accumulation; LayerNorm,
        attention cores, the key-point chain, bias / GELU / residuals stay f32).  The option of BASELINE configs[3]; not the parity path."""
        assert dtype in ("f32", "bf16")
        self.compute_dtype = dtype
        return self

    def _build_plan(self):
        a = self.attns
        G = ops.GemmLayer
        S = lambda w, b: ops.GemmLayer(w, b, ksplit=True)   # noqa: E731 -- layers of the key-point chain: K x W = 1024 rows per sample
        dev = a.norm1.weight.device

        def mlp(m, g):
            return g(m.fc1.weight, m.fc1.bias), g(m.fc2.weight, m.fc2.bias)

        def cat(*lins, ksplit=False):
            w = torch.cat([l.weight.detach() for l in lins], 0)
            b = torch.cat([l.bias.detach() for l in lins], 0) if lins[0].bias is not None else None
            return ops.GemmLayer(w, b, ksplit=ksplit)

        s1, ra, s2 = a.sector_attn1, a.range_attn, a.sector_attn2
        proj, mlp_g = G(a.proj.weight, a.proj.bias), mlp(a.mlp, G)
        # r6: norm2 is folded into the MLP's first GEMM (GemmLayer.fold_layernorm): the output projection leaves the row statistics of
        # x + proj(.) in its epilogue and fc1 normalises its own accumulators -- the normalised tokens are never written or read
        ln2_folded = proj.stats_ok and mlp_g[0].fold_layernorm(a.norm2)
        return dict(
            ln2_folded=ln2_folded,
            pos=self.pos_cart.reshape(self.patches_resolution[0], self.patches_resolution[1], 2).to(dev).float().contiguous(),
            s1_q=S(s1.proj_q.weight, s1.proj_q.bias), s1_kv=cat(s1.proj_k, s1.proj_v), s1_proj=S(s1.proj.weight, s1.proj.bias),
            s1_mlp=mlp(s1.mlp, S), s1_pe=_fold_pos_mlp(s1.pos_embedding_cart),
            ra_qkv=cat(ra.proj_q, ra.proj_k, ra.proj_v, ksplit=True), ra_proj=S(ra.proj.weight, ra.proj.bias), ra_mlp=mlp(ra.mlp, S),
            ra_pe=_fold_pos_mlp(ra.pos_embedding_cart),
            s2_q=G(s2.proj_q.weight, s2.proj_q.bias), s2_kv=cat(s2.proj_k, s2.proj_v, ksplit=True), s2_pe=_fold_pos_mlp(s2.pos_embedding_cart),
            proj=proj, mlp=mlp_g)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, H*W, C) tokens (range-major, the reference's order) -> same shape"""
        return self._run(x, col_major=False)

    def forward_cols(self, x: torch.Tensor) -> torch.Tensor:
        """x: (B, W*H, C) tokens in AZIMUTH-major order -- the dense BEV map as it stands in NHWC memory, (B, theta, r, C) -- -> same
        shape and order.  Every op of the block is per token, per azimuth column or per key-point window, so the token order in
        memory is free: with the columns contiguous the two sector attentions stream their slab and the permutes of
        voxelnet.py:211,219 disappear.  Element for element the result equals ``forward`` on the transposed tokens."""
        return self._run(x, col_major=True)

    def _run(self, x: torch.Tensor, col_major: bool) -> torch.Tensor:
        eval_only(self, "SetBlock")
        hip.require_device(x)
        a = self.attns
        p = self._plan.get(self, self._build_plan)
        H, W = self.patches_resolution
        B, L, C = x.shape
        assert L == H * W, "flatten img_tokens has wrong size"
        K, heads, sh, st = a.H, a.num_heads, a.shift_size, hip.stream()
        cm = int(col_major)
        ln = lambda t, n: ops.layernorm(t, n.weight.detach(), n.bias.detach(), n.eps)  # noqa: E731
        x2 = x.contiguous().view(B * L, C).float()
        # (every GEMM of the bf16 route must be packable for the bf16 pipe -- k % 64, n % 16 --, else the block stays in f32)
        b16 = self.compute_dtype == "bf16" and C % 64 == 0 and all(g.bf16_ok for g in (p["s1_kv"], p["s2_q"], p["proj"], p["mlp"][0], p["mlp"][1]))
        if b16:
            xn, cmean, xn_in = ops.layernorm(x2, a.norm1.weight.detach(), a.norm1.bias.detach(), a.norm1.eps, want_chan_mean=True, bf16_copy=True)
        else:
            xn, cmean = ops.layernorm(x2, a.norm1.weight.detach(), a.norm1.bias.detach(), a.norm1.eps, want_chan_mean=True)
            xn_in = xn
        dev = x.device
        top = torch.empty((B, K, W), dtype=torch.int32, device=dev)
        kp = torch.empty((B * K * W, C), dtype=torch.float32, device=dev)
        kpos = torch.empty((B, K, W, 2), dtype=torch.float32, device=dev)
        hip.call("pn_setblock_keypoints", cmean.data_ptr(), xn.data_ptr(), p["pos"].data_ptr(), B, H, W, C, K, sh, cm, top.data_ptr(),
                 kp.data_ptr(), kpos.data_ptr(), st)
        self.last_top_idx = top
        # sector attention 1: key points <- column
        q1, kv1 = p["s1_q"](kp), p["s1_kv"](xn_in)
        o1 = torch.empty_like(kp)
        hip.call("pn_setblock_sector_kp_attn", q1.data_ptr(), kv1.data_ptr(), p["pos"].data_ptr(), kpos.data_ptr(),
                 p["s1_pe"].data_ptr(), B, H, W, C, heads, K, sh, cm, float(a.scale), o1.data_ptr(), st)
        s1 = p["s1_proj"](o1, residual=kp)
        s1 = p["s1_mlp"][1](p["s1_mlp"][0](ln(s1, a.sector_attn1.norm2), act=ops.ACT_GELU), residual=s1)
        # range attention among key points
        qkv = p["ra_qkv"](ln(s1, a.range_attn.norm1))
        o2 = torch.empty_like(kp)
        hip.call("pn_setblock_range_attn", qkv.data_ptr(), kpos.data_ptr(), p["ra_pe"].data_ptr(), B, W, C, heads, K, a.W,
                 float(a.scale), o2.data_ptr(), st)
        s2 = p["ra_proj"](o2, residual=s1)
        s2 = p["ra_mlp"][1](p["ra_mlp"][0](ln(s2, a.range_attn.norm2), act=ops.ACT_GELU), residual=s2)
        # sector attention 2: column <- key points
        # (measured and dropped, r3: the query projection on a second stream beside the key-point chain -- one block per CU so that the
        # chain's kernels still get dispatched -- 1.442 against 1.423 ms for the two blocks in one hipGraph: nothing to win)
        q3, kv3 = p["s2_q"](xn_in), p["s2_kv"](s2)
        o3 = torch.empty_like(x2)
        hip.call("pn_setblock_sector_col_attn", q3.data_ptr(), kv3.data_ptr(), p["pos"].data_ptr(), kpos.data_ptr(),
                 p["s2_pe"].data_ptr(), B, H, W, C, heads, K, sh, cm, float(a.scale), o3.data_ptr(), st)
        if b16:
            y = p["proj"](ops.to_bf16(o3), residual=x2)
            z16 = ops.layernorm(y, a.norm2.weight.detach(), a.norm2.bias.detach(), a.norm2.eps, bf16_copy=True, f32_out=False)
            y = p["mlp"][1](p["mlp"][0](z16, act=ops.ACT_GELU, out_bf16=True), residual=y)
            return y.view(B, L, C)
        if p["ln2_folded"]:
            y, ystats = p["proj"](o3, residual=x2, stats_out=True)
            y = p["mlp"][1](p["mlp"][0](y, act=ops.ACT_GELU, ln_stats=ystats), residual=y)
            return y.view(B, L, C)
        y = p["proj"](o3, residual=x2)
        y = p["mlp"][1](p["mlp"][0](ln(y, a.norm2), act=ops.ACT_GELU), residual=y)
        return y.view(B, L, C)


def waymo_bev_pos(x_size=144, y_size=256, pc_range=(0.3, -3.14368, -2.0, 75.18, 3.14368, 4.0), voxel_size=(0.065, 0.00307, 0.15),
                  scale=8) -> torch.Tensor:
    """(1, x_size, y_size, 4) = [x, y, r, phi] at BEV cell centres (det3d/models/detectors/voxelnet.py:10-25)"""
    ri = torch.linspace(0, x_size - 1, x_size)[:, None].expand(x_size, y_size) + 0.5
    ti = torch.linspace(0, y_size - 1, y_size)[None, :].expand(x_size, y_size) + 0.5
    r = ri * voxel_size[0] * scale + pc_range[0]
    phi = ti * voxel_size[1] * scale + pc_range[1]
    return torch.stack([r * torch.cos(phi), r * torch.sin(phi), r, phi], dim=2)[None]
