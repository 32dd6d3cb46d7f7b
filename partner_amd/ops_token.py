"""Token GEMMs (nn.Linear) and LayerNorm of the attention block and the geometry-aware head over the C ABI."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple  # noqa: F401

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_TANH, ConvDesc  # noqa: F401
from .ops_common import *  # noqa: F401,F403
from .ops_common import _f32, _workspace  # noqa: F401
from .routes import R, S  # noqa: F401


# ------------------------------------------------------------------------------ GEMM / attention glue (A1)
ACT_GELU = hip.ACT_GELU




class GemmLayer:
    """packed nn.Linear: y = act(x @ W^T + b) (+ residual) on the token-GEMM kernel (csrc/linear.hip: pn_linear_f32);
    ``PN_LINEAR=0`` keeps the r2 route through the MFMA convolution kernel (a 1x1 convolution)"""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, ksplit: bool = False):
        """``ksplit``: the layer runs on a few thousand rows at most (the key-point chains of the SetBlock): pn_linear_ksplit_f32, the
        K-split form with its own fp32 summation order, for every call of this layer"""
        hip.require_device(weight)
        lib = hip.load()
        w = weight.detach().contiguous().float()
        self.n, self.k = w.shape
        self.linear = R.linear and self.n % 4 == 0 and self.k % 4 == 0
        self.entry = "pn_linear_ksplit_f32" if (ksplit and self.linear) else "pn_linear_f32" if self.linear else "pn_gemm_bias_act_f32"
        if self.linear:
            self.packed = _f32(lib.pn_linear_packed_weight_floats(self.n, self.k), w.device)
            hip.call("pn_pack_linear_weight_f32", w.data_ptr(), self.n, self.k, self.packed.data_ptr(), hip.stream())
        else:
            self.packed = _f32(lib.pn_conv_packed_weight_floats(self.n, self.k, 1, 1, 1), w.device)
            hip.call("pn_pack_conv_weight_f32", w.data_ptr(), self.n, self.k, 1, 1, 1, self.packed.data_ptr(), hip.stream())
        self.bias = None if bias is None else bias.detach().contiguous().float()
        self._w_f32 = w                  # source of the bf16 pack (made on the first bf16 call)
        self.ln = None

    @property
    def bf16_ok(self) -> bool:
        """can this layer run on the bf16 matrix pipe (pn_linear_bf16: k a multiple of 64, n of 16)?  Callers keep a GEMM in f32 where not."""
        return self.k % 64 == 0 and self.n % 16 == 0

    def prepack_bf16(self) -> None:
        """pack the bf16 weights now (set_compute_dtype('bf16')), not on the first call -- which may be inside a hipGraph capture"""
        if self.bf16_ok and getattr(self, "packed_bf16", None) is None:
            self.packed_bf16 = torch.empty(hip.load().pn_conv_bf16_rows_packed_elems(self.n, self.k, 1, 1), dtype=torch.bfloat16, device=self._w_f32.device)
            hip.call("pn_pack_conv_weight_bf16_rows", self._w_f32.data_ptr(), self.n, self.k, 1, 1, self.packed_bf16.data_ptr(), hip.stream())

    @property
    def stats_ok(self) -> bool:
        """can this layer leave the row statistics a LayerNorm-folding consumer needs (``__call__(..., stats_out=True)``)?"""
        return self.linear and self.entry == "pn_linear_f32" and self.n % 32 == 0 and R.ln_fold

    def fold_layernorm(self, norm) -> bool:
        """Fold ``norm`` (an nn.LayerNorm over this layer's k inputs) into the layer: LayerNorm(x) W^T + b = rstd (x (W gamma)^T - mean colsum)
        + (b + W beta).  After this ``__call__(x, ln_stats=table)`` takes the UN-normalised rows and the statistics table their producer left
        (pn_linear_ln_f32); plain calls keep the plain weights.  False (nothing changed) where the fold does not apply."""
        if not (self.linear and self.entry == "pn_linear_f32" and self.k % 64 == 0 and R.ln_fold):
            return False
        lib = hip.load()
        w64 = self._w_f32.double()
        g, b = norm.weight.detach().double().to(w64.device), norm.bias.detach().double().to(w64.device)
        wg = (w64 * g[None, :])
        wg32 = wg.float().contiguous()
        packed = _f32(lib.pn_linear_packed_weight_floats(self.n, self.k), wg32.device)
        hip.call("pn_pack_linear_weight_f32", wg32.data_ptr(), self.n, self.k, packed.data_ptr(), hip.stream())
        b0 = self.bias.double() if self.bias is not None else torch.zeros(self.n, dtype=torch.float64, device=w64.device)
        # colsum over the ROUNDED folded weights: what the MFMA multiplies, so mean * colsum cancels the accumulated mean term exactly in exact arithmetic
        self.ln = dict(packed=packed, colsum=wg32.double().sum(1).float().contiguous(), bias=(b0 + w64 @ b).float().contiguous(), eps=float(norm.eps))
        return True

    def __call__(self, x: torch.Tensor, act=ACT_NONE, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 out_bf16: bool = False, ln_stats: Optional[torch.Tensor] = None, stats_out: bool = False):
        """x: (m, k) f32 -- or bf16: the layer then runs on the bf16 matrix pipe (pn_linear_bf16, csrc/conv_bf16.hip; weights packed as bf16
        on first use, f32 accumulation, bias / activation / residual in f32) and returns f32, or bf16 with ``out_bf16`` (the input of
        another bf16 layer)"""
        assert x.dim() == 2 and x.is_contiguous() and x.shape[1] == self.k
        m = x.shape[0]
        st = hip.stream()
        prof = S.profiler
        if x.dtype == torch.bfloat16:
            assert self.bf16_ok, "bf16 GEMM: k a multiple of 64, n of 16 (check GemmLayer.bf16_ok and keep the layer in f32 otherwise)"
            self.prepack_bf16()
            if out is None:
                out = torch.empty((m, self.n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
            if prof is not None:
                ev = prof.begin(st)
            hip.call("pn_linear_bf16", x.data_ptr(), m, self.k, self.k, self.packed_bf16.data_ptr(), self.n, hip.ptr(self.bias), int(act),
                     hip.ptr(residual), self.n, out.data_ptr(), self.n, int(out.dtype == torch.float32), st)
            if prof is not None:
                prof.end(ev, 2.0 * m * self.n * self.k, st, tag=f"gemm {m}x{self.k}->{self.n} bf16")
            return out
        assert not out_bf16, "a bf16 output needs a bf16 input"
        if out is None:
            out = torch.empty((m, self.n), dtype=torch.float32, device=x.device)
        if prof is not None:
            ev = prof.begin(st)
        stats = None
        if ln_stats is not None or stats_out:
            # LayerNorm folded around the GEMM (pn_linear_ln_f32): consumer of a statistics table (``ln_stats``: x is the UN-normalised rows;
            # needs fold_layernorm) or producer of one (``stats_out``: -> (out, table [m][n / 32][2]))
            assert not (ln_stats is not None and stats_out)
            if ln_stats is not None:
                assert self.ln is not None and tuple(ln_stats.shape) == (m, self.k // 32, 2) and ln_stats.is_contiguous()
                hip.call("pn_linear_ln_f32", x.data_ptr(), m, self.k, self.k, self.ln["packed"].data_ptr(), self.n, self.ln["bias"].data_ptr(), int(act),
                         hip.ptr(residual), self.n, out.data_ptr(), self.n, ln_stats.data_ptr(), self.ln["colsum"].data_ptr(), self.ln["eps"], None, st)
            else:
                assert self.stats_ok
                stats = torch.empty((m, self.n // 32, 2), dtype=torch.float32, device=x.device)
                hip.call("pn_linear_ln_f32", x.data_ptr(), m, self.k, self.k, self.packed.data_ptr(), self.n, hip.ptr(self.bias), int(act),
                         hip.ptr(residual), self.n, out.data_ptr(), self.n, None, None, 0.0, stats.data_ptr(), st)
        else:
            hip.call(self.entry, x.data_ptr(), m, self.k, self.k, self.packed.data_ptr(), self.n,
                     hip.ptr(self.bias), int(act), hip.ptr(residual), self.n, out.data_ptr(), self.n, st)
        if prof is not None:
            prof.end(ev, 2.0 * m * self.n * self.k, st, tag=f"gemm {m}x{self.k}->{self.n}")
        return (out, stats) if stats_out else out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, want_chan_mean=False, bf16_copy=False, f32_out=True):
    """-> out [, chan_mean] [, bf16 copy]; ``bf16_copy``: also the result rounded to bf16 (input of the bf16 GEMMs); with ``f32_out`` False
    only the copy is written (and returned in place of ``out``)"""
    hip.require_device(x)
    assert x.dim() == 2 and x.is_contiguous()
    rows, c = x.shape
    out = torch.empty_like(x) if f32_out else None
    cm = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_chan_mean else None
    if bf16_copy:
        o16 = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
        hip.call("pn_layernorm_bf16out_f32", x.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(), float(eps), hip.ptr(out), o16.data_ptr(),
                 hip.ptr(cm), hip.stream())
        res = ((out,) if f32_out else ()) + ((cm,) if want_chan_mean else ()) + (o16,)
        return res if len(res) > 1 else res[0]
    assert f32_out
    hip.call("pn_layernorm_f32", x.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(), float(eps), out.data_ptr(),
             hip.ptr(cm), hip.stream())
    return (out, cm) if want_chan_mean else out
