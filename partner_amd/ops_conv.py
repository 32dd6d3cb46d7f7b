"""Convolutions of the BEV backbone and the heads over the C ABI: ConvLayer (direct / F(2,3) / F(4,3) / chained forms), the first
convolution on the pillar canvas, multi-job launches with GroupNorm-family statistics, the conv profiler of bench.py's roofline pass."""
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
from .ops_index import VoxelIndex  # noqa: F401

# ------------------------------------------------------------------------------ convolution


class frames_in_flight:
    """``with ops.frames_in_flight(n):`` -- the convolutions launched (or captured into a hipGraph) inside the block carry the hint that n
    independent frames run at the same time on other streams (pn_conv_desc.frames_in_flight); engine.FrameEngine wraps its capture in it"""

    def __init__(self, n: int):
        self.n, self.prev = max(1, int(n)), 1

    def __enter__(self):
        self.prev, S.frames_in_flight = S.frames_in_flight, self.n
        return self

    def __exit__(self, *exc):
        S.frames_in_flight = self.prev
        return False


class ConvProfiler:
    """Execution time of every MFMA-conv launch (bench.py roofline): the event pair is attached to the kernel dispatch
    itself (``pn_profile_next_launch`` -> hipExtLaunchKernelGGL), so the elapsed time is the kernel's own duration -- the
    figure a rocprofv3 kernel trace reports -- and not the launch gap of an eager stream."""

    def __init__(self):
        self.lib = hip.load()
        self.pairs = []   # (start, stop, flops, tag)
        self.free = []

    def _event(self):
        if self.free:
            return self.free.pop()
        ev = C.c_void_p()
        hip.call("pn_event_create", C.byref(ev))
        return ev

    def begin(self, stream):
        a, b = self._event(), self._event()
        hip.call("pn_profile_next_launch", a, b)
        return (a, b)

    def end(self, evs, flops, stream, tag=None, issued=None, dense=None):
        """flops: what the launch is billed (dense-convolution count; the sparse first layer: the pairs it multiplies);
        issued: FLOPs that reach the MFMA (default: by the Winograd form named in the tag); dense: the direct dense algorithm's
        count (default = flops)"""
        if issued is None:
            issued = flops * (2.0 / 3.0 if tag and "F(2,3)" in tag else 0.5 if tag and "F(4,3)" in tag else 1.0)
        self.pairs.append((evs[0], evs[1], flops, tag, issued, flops if dense is None else dense))

    def collect(self, by_tag=False, full=False):
        """-> (total billed FLOPs, total milliseconds, launches[, {tag: (flops, ms, launches)}]); synchronises.
        full=True: the per-tag tuples are (flops, ms, launches, issued FLOPs, dense-algorithm FLOPs)"""
        flops, ms = 0.0, 0.0
        tags = {}
        out = C.c_float()
        for a, b, f, tag, iss, den in self.pairs:
            hip.call("pn_event_elapsed_ms", a, b, C.byref(out))
            ms += out.value
            flops += f
            t = tags.setdefault(tag, [0.0, 0.0, 0, 0.0, 0.0])
            t[0] += f
            t[1] += out.value
            t[2] += 1
            t[3] += iss
            t[4] += den
            self.free += [a, b]
        n = len(self.pairs)
        self.pairs = []
        if not by_tag:
            return (flops, ms, n)
        return (flops, ms, n, {k: (tuple(v) if full else tuple(v[:3])) for k, v in tags.items()})




def enable_conv_profiling() -> ConvProfiler:
    S.profiler = ConvProfiler()
    return S.profiler


def disable_conv_profiling() -> None:
    S.profiler = None


# F(4, 3) (conv_wino4.hip): PN_CONV_WINO4=0 keeps F(2, 3); taken from this many 32-quad x 32-column tiles on (the kernel's K-split form
# runs one block per such tile, the plain form one per 32 quads x 128 columns when those fill the chip).  Measured against F(2, 3):
# 256 x 256 x 128 -> 128: 79 us / 112; 128 x 128 x 128 -> 128 (512 tiles): 27 / 31; 64 x 64 x 256 -> 256 (256 tiles): 31 / 37;
# Waymo RPN 256 x 144 x 128 -> 128: 61 / 105, 128 x 72 x 256 -> 256: 71 / 104


class ConvLayer:
    """One packed convolution (+ per-channel affine + activation) on NHWC maps.

    weight: torch layout (Cout, Cin/groups, KH, KW), or (Cin, Cout, 2, 2) when ``deconv2x2``.
    scale / shift: per-output-channel affine (folded BatchNorm, or bias as shift)."""

    def __init__(self, weight: torch.Tensor, stride=1, pad=0, groups=1, scale=None, shift=None, act=ACT_NONE,
                 deconv2x2=False, range_strata=0, dtype="f32", wino4=True):
        hip.require_device(weight)
        lib = hip.load()
        w = weight.detach().contiguous().float()
        dev = w.device
        st = hip.stream()
        self.deconv2x2, self.range_strata, self.groups = bool(deconv2x2), int(range_strata), int(groups)
        self.stride, self.act = int(stride), int(act)
        self.pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
        assert dtype in ("f32", "bf16")
        self.dtype = dtype
        if dtype == "bf16":
            # bf16 activations / weights, f32 accumulate (pn_conv2d_nhwc_bf16)
            if deconv2x2:
                cin, cout = w.shape[0], w.shape[1]
                assert tuple(w.shape[2:]) == (2, 2)
                self.cin, self.cout, self.kh, self.kw = cin, cout, 1, 1
                wc = w.permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1).contiguous()  # row (2*di+dj)*Cout + n
                self.packed = torch.empty(lib.pn_conv_packed_weight_bf16_elems(4 * cout, cin, 1, 1, 1), dtype=torch.bfloat16, device=dev)
                hip.call("pn_pack_conv_weight_bf16", wc.data_ptr(), 4 * cout, cin, 1, 1, 1, self.packed.data_ptr(), st)
                self._pack_bf16_rows(lib, wc, 4 * cout, cin, 1, 1, st)
            else:
                pack_groups = self.range_strata if self.range_strata > 1 else self.groups
                cout_t, cin_g, kh, kw = w.shape
                self.cin, self.cout, self.kh, self.kw = cin_g, cout_t // pack_groups, kh, kw
                self.packed = None
                self._bf16_general = (w, cout_t, cin_g, kh, kw, pack_groups)      # (a reference to the caller's f32 weight, no copy)
                if pack_groups == 1:
                    self._pack_bf16_rows(lib, w, cout_t, cin_g, kh, kw, st)
                if getattr(self, "packed_rows", None) is None:
                    self._ensure_bf16_general()      # no rows form for this shape: the general layout is the one that runs
                # (with a rows form the general layout is packed only if a call ever needs it -- one packed copy per layer, not two)
        elif deconv2x2:
            cin, cout = w.shape[0], w.shape[1]
            assert tuple(w.shape[2:]) == (2, 2)
            self.cin, self.cout, self.kh, self.kw = cin, cout, 1, 1
            self.packed = _f32(lib.pn_deconv2x2_packed_weight_floats(cin, cout), dev)
            hip.call("pn_pack_deconv2x2_weight_f32", w.data_ptr(), cin, cout, self.packed.data_ptr(), st)
        else:
            pack_groups = self.range_strata if self.range_strata > 1 else self.groups
            cout_t, cin_g, kh, kw = w.shape
            self.cin, self.cout, self.kh, self.kw = cin_g, cout_t // pack_groups, kh, kw
            self.packed = _f32(lib.pn_conv_packed_weight_floats(self.cout, cin_g, kh, kw, pack_groups), dev)
            hip.call("pn_pack_conv_weight_f32", w.data_ptr(), cout_t, cin_g, kh, kw, pack_groups, self.packed.data_ptr(), st)
        self.scale = None if scale is None else scale.detach().contiguous().float()
        self.shift = None if shift is None else shift.detach().contiguous().float()
        self.out_channels = self.cout * (self.groups if not deconv2x2 else 1)
        self._pack_cin = self.cin
        # plain 3x3 / stride 1 / pad 1 layers also keep the width-Winograd F(2, 3) weights (conv_wino.hip: 6 instead of 9 MFMA
        # equivalents per output); used when the map is large enough to fill the chip with its 64-pair x 64-column tiles
        self.wino_packed = None
        if (dtype == "f32" and not deconv2x2 and self.range_strata <= 1 and self.groups == 1 and (self.kh, self.kw) == (3, 3)
                and self.stride == 1 and self.pad == (1, 1) and self.cin % 4 == 0 and R.conv_wino):
            self.wino_packed = _f32(lib.pn_conv_wino_packed_weight_floats(self.cout, self.cin), dev)
            hip.call("pn_pack_conv_weight_wino_f32", w.data_ptr(), self.cout, self.cin, self.wino_packed.data_ptr(), st)
        # ... and the F(4, 3) weights (4.5 MFMA equivalents per output) when the kernel's 32-column wave tiles fit the layer
        self.wino4_packed = None
        if self.wino_packed is not None and R.conv_wino4 and wino4 and self.cout % 32 == 0 and self.act in (ACT_NONE, ACT_RELU):
            self.wino4_packed = _f32(lib.pn_conv_wino4_packed_weight_floats(self.cout, self.cin), dev)
            hip.call("pn_pack_conv_weight_wino4_f32", w.data_ptr(), self.cout, self.cin, self.wino4_packed.data_ptr(), st)
        # ... and the F(2, 3) x F(4, 3) weights of the chained form (conv_wchain.hip, ops.conv_chain: 3 MFMA equivalents per output)
        self.wino24_packed = None
        self._w_ref = None               # the layer's weight, for the packs of the TRANSPOSED kernel a chain on a transposed map takes (lazy)
        self._chain_t: dict = {}
        if self.wino4_packed is not None and R.conv_chain2d and self.cin % 32 == 0:
            self.wino24_packed = _f32(lib.pn_conv_wino24_packed_weight_floats(self.cout, self.cin), dev)
            hip.call("pn_pack_conv_weight_wino24_f32", w.data_ptr(), self.cout, self.cin, self.wino24_packed.data_ptr(), st)
            self._w_ref = w

        # ... and 3x3 layers with one to three output channels over many input channels (the geometry-aware head's 256 -> 1 heat-map and
        # vote-class convolutions): a GEMM over the pixels against the (9 cout, cin) tap matrix + a nine-term shifted sum
        # (pn_conv3x3_tap_sum_f32) -- the input is read once; on a 32-column MFMA tile these layers ran at 1.4 TFLOP/s
        self.tap_packed = None
        if (dtype == "f32" and not deconv2x2 and self.range_strata <= 1 and self.groups == 1 and (self.kh, self.kw) == (3, 3) and self.stride == 1
                and self.pad == (1, 1) and self.cout <= 3 and self.cin >= 128 and self.cin % 4 == 0 and R.conv_tapsum and R.linear):
            self.tap_n = (9 * self.cout + 3) // 4 * 4
            self.tap_packed = _f32(lib.pn_linear_packed_weight_floats(self.tap_n, self.cin), dev)
            self._pack_taps(w)

    def _ensure_bf16_general(self) -> None:
        if self.packed is None:
            w, cout_t, cin_g, kh, kw, pack_groups = self._bf16_general
            self.packed = torch.empty(hip.load().pn_conv_packed_weight_bf16_elems(self.cout, cin_g, kh, kw, pack_groups), dtype=torch.bfloat16,
                                      device=w.device)
            hip.call("pn_pack_conv_weight_bf16", w.data_ptr(), cout_t, cin_g, kh, kw, pack_groups, self.packed.data_ptr(), hip.stream())

    def _pack_bf16_rows(self, lib, w: torch.Tensor, rows: int, cin: int, kh: int, kw: int, st) -> None:
        """the [row][tap][cin] weights of the bf16 implicit-GEMM kernel (csrc/conv_bf16.hip, r5): the layers of the Waymo BEV maps -- cin a
        multiple of 64, cout of 16 -- run there, every other bf16 layer on the general kernel (pn_conv2d_nhwc_bf16)"""
        self.packed_rows = None
        if cin % 64 == 0 and self.cout % 16 == 0 and self.act in (ACT_NONE, ACT_RELU):
            self.packed_rows = torch.empty(lib.pn_conv_bf16_rows_packed_elems(rows, cin, kh, kw), dtype=torch.bfloat16, device=w.device)
            hip.call("pn_pack_conv_weight_bf16_rows", w.data_ptr(), rows, cin, kh, kw, self.packed_rows.data_ptr(), st)

    def planes_desc(self, b: int, h: int, w: int, ct: int, in_channel_offset: int = 0):
        """the descriptor of ``to_planes`` on a (b, h, w, ct) map, or None where the direct kernel's planes epilogue does not apply"""
        if self.dtype != "f32" or self.deconv2x2 or self.range_strata > 1 or self.groups != 1 or not R.conv_planes:
            return None
        d = ConvDesc(b, h, w, self.cin, self.cout, 1, self.kh, self.kw, self.stride, self.pad[0], self.pad[1], ct, in_channel_offset, self.cout, 0,
                     self.act, 0, 0, 0, 0, 0)
        return d if hip.load().pn_conv2d_nhwc_planes_supported(C.byref(d)) else None

    def to_planes(self, x: torch.Tensor, planes: torch.Tensor, in_channel_offset: int = 0) -> None:
        """this convolution (direct implicit-GEMM kernel) with its output written as the F(4, 3) planes of ``conv_chain`` (csrc/conv_mfma.hip,
        r6): the stride-2 layer at the head of an RPN block feeds the block's chain without the NHWC map in between"""
        hip.require_device(x, planes)
        b, h, w, ct = x.shape
        d = self.planes_desc(b, h, w, ct, in_channel_offset)
        assert d is not None, "ConvLayer.to_planes: check planes_desc first"
        self._ensure("direct")
        oh, ow = self.out_hw(h, w)
        assert planes.numel() >= hip.load().pn_wino4_planes_floats(b, oh, ow, self.cout)
        st = hip.stream()
        prof = S.profiler
        if prof is not None:
            ev = prof.begin(st)
        hip.call("pn_conv2d_nhwc_planes_f32", C.byref(d), x.data_ptr(), self.packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                 planes.data_ptr(), st)
        if prof is not None:
            prof.end(ev, 2.0 * b * oh * ow * self.cout * self.cin * self.kh * self.kw, st, tag=f"{oh}x{ow} {self.cin}->{self.cout} k{self.kh} -> planes")

    def chain_weights(self, two_d: bool, transposed: bool) -> torch.Tensor:
        """packed weights of the chained F(4,3) / F(2,3)xF(4,3) forms; ``transposed``: of the kernel with kh and kw swapped (the chain then runs
        on the transposed map), packed on first use.  The transposed layouts belong to the inference path (the head's
        branch chain): they are not tracked by ``prepack_used`` and a stale one is repacked lazily on the caller's stream"""
        key = "wino44" if two_d == "wino44" else ("wino24" if two_d else "wino4")
        if not transposed:
            if key == "wino44" and getattr(self, "wino44_packed", None) is None:      # F(4,3) x F(4,3) (r5): packed on first use
                lib, w = hip.load(), getattr(self, "_stale_w", None)
                w = self._w_ref if w is None else w
                self.wino44_packed = _f32(lib.pn_conv_wino44_packed_weight_floats(self.cout, self._pack_cin), w.device)
                hip.call("pn_pack_conv_weight_wino44_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino44_packed.data_ptr(), hip.stream())
                self.__dict__.setdefault("_used", set()).add("wino44")
                (getattr(self, "_stale", None) or set()).discard("wino44")
                return self.wino44_packed
            self._ensure(key)
            return {"wino44": getattr(self, "wino44_packed", None), "wino24": self.wino24_packed, "wino4": self.wino4_packed}[key]
        stale = getattr(self, "_stale", None) or ()
        if key not in self._chain_t or (key + "_t") in stale:
            lib, w = hip.load(), getattr(self, "_stale_w", None)
            w = self._w_ref if w is None else w
            wt = w.detach().float().transpose(2, 3).contiguous()
            fam = key
            buf = self._chain_t.get(key)
            if buf is None:
                buf = _f32(getattr(lib, f"pn_conv_{fam}_packed_weight_floats")(self.cout, self._pack_cin), w.device)
            hip.call(f"pn_pack_conv_weight_{fam}_f32", wt.data_ptr(), self.cout, self._pack_cin, buf.data_ptr(), hip.stream())
            self._chain_t[key] = buf
            if not isinstance(getattr(self, "_chain_t_src", None), dict):
                self._chain_t_src = {}
            self._chain_t_src[key] = wt   # (the launch is asynchronous: the transposed copy stays referenced, one per layout)
            if stale:
                stale.discard(key + "_t")
        return self._chain_t[key]

    def _pack_taps(self, w: torch.Tensor) -> None:
        w9 = torch.zeros((self.tap_n, self.cin), dtype=torch.float32, device=w.device)
        w9[:9 * self.cout] = w.reshape(self.cout, -1, 9)[:, :self.cin].permute(2, 0, 1).reshape(9 * self.cout, -1)   # row t * cout + co
        hip.call("pn_pack_linear_weight_f32", w9.data_ptr(), self.tap_n, self.cin, self.tap_packed.data_ptr(), hip.stream())

    def repack(self, weight: torch.Tensor, shift: Optional[torch.Tensor] = None, token=None) -> None:
        """refresh the packed copies from an updated weight of the same shape (training: once per step).  LAZY: a layout (direct,
        F(2, 3), F(4, 3)) is packed when the next call takes it -- a layer keeps up to three and uses one per map size, and the
        tiny pack launches were 0.8 ms of an 18.8 ms training iteration.  ``weight`` must stay valid (and unchanged) until then: the
        training steps hand in views of their flat parameter buffer, which the optimizer rewrites only after the backward.
        ``token``: a call with the token of the previous call is a no-op (the training steps refresh every layer once at the start of an
        iteration, ``prepack_used``, and pass the iteration's token from the layers' forward / backward)"""
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        w = weight.detach()
        assert w.is_contiguous() and w.dtype == torch.float32
        self._stale_w = w
        self._stale = {"direct"} | ({"wino"} if self.wino_packed is not None else set()) | ({"wino4"} if self.wino4_packed is not None else set())
        if self.tap_packed is not None:
            self._stale.add("tap")
        if getattr(self, "wino24_packed", None) is not None:
            self._stale.add("wino24")
        if getattr(self, "wino44_packed", None) is not None:
            self._stale.add("wino44")
        for k in getattr(self, "_chain_t", {}):
            self._stale.add(k + "_t")
        if shift is not None:
            self.shift = shift

    def prepack_used(self) -> None:
        """pack, on the current stream, the layouts the layer's calls have taken so far (after ``repack``)"""
        for layout in sorted(getattr(self, "_used", ())):
            self._ensure(layout)

    def _ensure(self, layout: str) -> None:
        if not layout.endswith("_t"):
            self.__dict__.setdefault("_used", set()).add(layout)
        stale = getattr(self, "_stale", None)
        if not stale or layout not in stale:
            return
        stale.discard(layout)
        w, st = self._stale_w, hip.stream()
        if layout == "direct":
            if self.deconv2x2:
                hip.call("pn_pack_deconv2x2_weight_f32", w.data_ptr(), self._pack_cin, self.cout, self.packed.data_ptr(), st)
            else:
                pack_groups = self.range_strata if self.range_strata > 1 else self.groups
                hip.call("pn_pack_conv_weight_f32", w.data_ptr(), w.shape[0], self._pack_cin, self.kh, self.kw, pack_groups,
                         self.packed.data_ptr(), st)
        elif layout == "tap":
            self._pack_taps(w)
        elif layout == "wino":
            hip.call("pn_pack_conv_weight_wino_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino_packed.data_ptr(), st)
        elif layout == "wino24":
            hip.call("pn_pack_conv_weight_wino24_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino24_packed.data_ptr(), st)
        elif layout == "wino44":
            hip.call("pn_pack_conv_weight_wino44_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino44_packed.data_ptr(), st)
        else:
            hip.call("pn_pack_conv_weight_wino4_f32", w.data_ptr(), self.cout, self._pack_cin, self.wino4_packed.data_ptr(), st)

    def _use_wino(self, b: int, h: int, w: int, accumulate: bool) -> bool:
        if self.wino_packed is None or accumulate or w % 2 or self.cin != self._pack_cin:
            return False
        tiles = ((b * h * (w // 2) + 31) // 32) * ((self.cout + 63) // 64)   # 32-pair x 64-column tiles (the kernel takes 64-pair ones when they fill the chip)
        return tiles >= R.conv_wino_min_tiles

    def _use_wino4(self, b: int, h: int, w: int, accumulate: bool) -> bool:   # (the tile-count gate below is about leaving the direct kernel, not about the form)
        if self.wino4_packed is None or accumulate or w % 4 or self.cin != self._pack_cin or h * w > getattr(self, "wino4_max_pixels", 1 << 62):
            return False
        return ((b * h * (w // 4) + 31) // 32) * (self.cout // 32) >= R.conv_wino4_min_tiles

    def pad_input_channels(self, cin_padded: int) -> "ConvLayer":
        """declare that the input map carries zero pad channels up to a multiple of 4 (e.g. the 5-channel
        position encoding stored with 8): the packed rows past the real Cin are already zero"""
        assert cin_padded >= self.cin and cin_padded % 4 == 0 and (cin_padded + 31) // 32 == (self.cin + 31) // 32
        self.cin = cin_padded
        return self

    def out_hw(self, h: int, w: int) -> Tuple[int, int]:
        if self.deconv2x2:
            return 2 * h, 2 * w
        return ((h + 2 * self.pad[0] - self.kh) // self.stride + 1, (w + 2 * self.pad[1] - self.kw) // self.stride + 1)

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, out_channel_offset=0, in_channel_offset=0,
                 in_channels: Optional[int] = None, accumulate=False, out_f32=False, out_transposed=False) -> torch.Tensor:
        """x: NHWC (B,H,W,Ct).  Reads channels [in_channel_offset, +cin*groups); writes channels
        [out_channel_offset, +out_channels) of ``out`` (allocated if None).  bf16 layers take / return
        torch.bfloat16 maps (``out_f32``: f32 output).  ``out_transposed`` (layers on the F(4,3) kernel only): ``out`` is (B, W, H, C), the
        map stored transposed (pn_conv_desc.transpose_hw)."""
        hip.require_device(x)
        in_dt = torch.bfloat16 if self.dtype == "bf16" else torch.float32
        assert x.dim() == 4 and x.is_contiguous() and x.dtype == in_dt
        b, h, w, ct = x.shape
        oh, ow = self.out_hw(h, w)
        if out is None:
            out = torch.empty((b, oh, ow, self.out_channels), dtype=torch.float32 if (out_f32 or self.dtype == "f32") else torch.bfloat16,
                              device=x.device)
        if self.dtype == "bf16":
            assert out.shape[:3] == (b, oh, ow) and out.is_contiguous()
            d = ConvDesc(b, h, w, self.cin, self.cout, self.groups, self.kh, self.kw, self.stride, self.pad[0], self.pad[1],
                         ct, in_channel_offset, out.shape[3], out_channel_offset, self.act, int(self.deconv2x2), self.range_strata)
            st = hip.stream()
            prof = S.profiler
            if prof is not None:
                ev = prof.begin(st)
            igemm = getattr(self, "packed_rows", None) is not None and hip.load().pn_conv2d_igemm_bf16_supported(C.byref(d)) == 1
            if igemm:
                hip.call("pn_conv2d_igemm_bf16", C.byref(d), x.data_ptr(), self.packed_rows.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                         out.data_ptr(), int(out.dtype == torch.float32), st)
            else:
                if self.packed is None:
                    self._ensure_bf16_general()
                hip.call("pn_conv2d_nhwc_bf16", C.byref(d), x.data_ptr(), self.packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                         out.data_ptr(), int(out.dtype == torch.float32), st)
            if prof is not None:
                macs = b * h * w * 4 * self.cout * self.cin if self.deconv2x2 else b * oh * ow * self.groups * self.cout * self.cin * self.kh * self.kw
                prof.end(ev, 2.0 * macs, st, tag=f"{oh}x{ow} {self.cin * self.groups}->{self.out_channels} k{self.kh} bf16")
            return out
        assert out.shape[:3] == ((b, ow, oh) if out_transposed else (b, oh, ow)) and out.is_contiguous()
        d = ConvDesc(b, h, w, self.cin, self.cout, self.groups, self.kh, self.kw, self.stride, self.pad[0], self.pad[1],
                     ct, in_channel_offset, out.shape[3], out_channel_offset, self.act, int(self.deconv2x2),
                     self.range_strata, 0, 0, int(accumulate))
        d.frames_in_flight = S.frames_in_flight
        d.transpose_hw = int(bool(out_transposed))
        st = hip.stream()
        prof = S.profiler
        if prof is not None:
            ev = prof.begin(st)
        use_tap = self.tap_packed is not None and not accumulate and self.cin == self._pack_cin and in_channel_offset % 4 == 0 and ct % 4 == 0
        use_wino4 = not use_tap and self._use_wino4(b, h, w, accumulate)
        use_wino = not use_tap and not use_wino4 and self._use_wino(b, h, w, accumulate)
        assert use_wino4 or not out_transposed, "ConvLayer: a transposed output needs the F(4,3) kernel (check _use_wino4 first)"
        self._ensure("tap" if use_tap else "wino4" if use_wino4 else "wino" if use_wino else "direct")
        if use_tap:
            m = b * h * w
            g = torch.empty((m, self.tap_n), dtype=torch.float32, device=x.device)
            hip.call("pn_linear_ksplit_f32", x.data_ptr() + 4 * in_channel_offset, m, self.cin, ct, self.tap_packed.data_ptr(), self.tap_n, None, ACT_NONE,
                     None, self.tap_n, g.data_ptr(), self.tap_n, st)
            hip.call("pn_conv3x3_tap_sum_f32", g.data_ptr(), self.tap_n, b, h, w, self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act,
                     out.data_ptr(), out.shape[3], out_channel_offset, st)
            if prof is not None:
                prof.end(ev, 2.0 * m * self.cout * self.cin * 9, st, tag=f"{oh}x{ow} {self.cin}->{self.out_channels} k3 gemm+taps", issued=2.0 * m * self.cin * 32)
            return out
        if use_wino4:
            hip.call("pn_conv2d_wino4_nhwc_f32", C.byref(d), x.data_ptr(), self.wino4_packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                     out.data_ptr(), st)
        elif use_wino:
            hip.call("pn_conv2d_wino_nhwc_f32", C.byref(d), x.data_ptr(), self.wino_packed.data_ptr(), hip.ptr(self.scale), hip.ptr(self.shift),
                     out.data_ptr(), st)
        else:
            hip.call("pn_conv2d_nhwc_f32", C.byref(d), x.data_ptr(), self.packed.data_ptr(), hip.ptr(self.scale),
                     hip.ptr(self.shift), out.data_ptr(), st)
        if prof is not None:
            # algorithmic FLOPs = 2 * output pixels * Cout * Cin * KH * KW (per group), dense-conv count
            if self.deconv2x2:
                macs = b * h * w * 4 * self.cout * self.cin
            else:
                z = self.groups
                macs = b * oh * ow * z * self.cout * self.cin * self.kh * self.kw
            prof.end(ev, 2.0 * macs, st, tag=f"{oh}x{ow} {self.cin * self.groups}->{self.out_channels} k{self.kh}{'t' if self.deconv2x2 else ''}{'s' if self.range_strata > 1 else ''}{' F(2,3)' if use_wino else ' F(4,3)' if use_wino4 else ''}")
        return out


# Runs of same-map 3x3 / stride-1 layers kept in the F(4, 3) domain between layers (csrc/conv_wchain.hip): PN_CONV_CHAIN=0 runs them one
# pn_conv2d_wino4_nhwc_f32 launch each, as r3 did


def _chain_desc(layer: "ConvLayer", b: int, h: int, w: int, out_ps: int = 0, out_co: int = 0, transposed: bool = False):
    d = ConvDesc(b, h, w, layer.cin, layer.cout, 1, 3, 3, 1, 1, 1, layer.cin, 0, out_ps or layer.cout, out_co, layer.act, 0, 0, 0, 0, 0)
    d.frames_in_flight = S.frames_in_flight
    d.transpose_hw = int(transposed)
    return d


def _chain_orientation(layers, b: int, h: int, w: int):
    """None, or whether the chain works on the transposed map (False: the Winograd axis is W; True: it is H -- maps like the Waymo BEV's
    256 x 144, whose W / 4 = 36 is not a power of two)"""
    if not R.conv_chain or not layers:
        return None
    lib = hip.load()
    for k, l in enumerate(layers):
        if l.dtype != "f32" or l.wino4_packed is None or l.cin != l._pack_cin or (k and l.cin != layers[k - 1].cout):
            return None
    for transposed in (False, True):
        if transposed and any(getattr(l, "_w_ref", None) is None for l in layers):
            break
        if all(lib.pn_conv_wino4_chain_supported(C.byref(_chain_desc(l, b, h, w, transposed=transposed))) for l in layers):
            return transposed
    return None


def conv_chain_orientation(layers, b: int, h: int, w: int):
    """None: the layers cannot run as a chain on a (b, h, w) map; False / True: they can, on the map as stored / transposed"""
    return _chain_orientation(layers, b, h, w)


def conv_chain_supported(layers, b: int, h: int, w: int) -> bool:
    """can ``layers`` (consecutive ConvLayers, each feeding the next) run as one Winograd-domain chain on a (b, h, w) map?"""
    return _chain_orientation(layers, b, h, w) is not None


def chain_planes(x: torch.Tensor, layers, in_channel_offset: int = 0) -> torch.Tensor:
    """the F(4, 3) planes of NHWC ``x`` in the orientation ``conv_chain(layers, ...)`` works in, for SEVERAL chains that read the same map
    (``conv_chain(..., planes=)``: E2ESWVoteHead's class and box / IoU branches) -- one NHWC -> planes pass instead of one per chain"""
    hip.require_device(x)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    b, h, w, ct = x.shape
    tr = _chain_orientation(layers, b, h, w)
    assert tr is not None, "chain_planes: check conv_chain_supported first"
    lib = hip.load()
    buf = torch.empty(lib.pn_wino4_planes_floats(b, w if tr else h, h if tr else w, layers[0].cin), dtype=torch.float32, device=x.device)
    hip.call("pn_wino4_planes_from_nhwc_f32", x.data_ptr(), b, h, w, layers[0].cin, ct, in_channel_offset, int(tr), buf.data_ptr(), hip.stream())
    return buf


def conv_chain(layers, x: Optional[torch.Tensor], out: Optional[torch.Tensor] = None, out_channel_offset=0, in_channel_offset=0, planes_from=None,
               shape=None, device=None, planes: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x NHWC (B, H, W, Ct) -> the NHWC output of the last layer.  One launch forms the six F(4, 3) planes of x, then every layer reads
    planes and writes planes (two buffers, alternating); the last one writes the map.  Same arithmetic as the layers one by one
    (ConvLayer.__call__ on pn_conv2d_wino4_nhwc_f32) up to the summation order over the input channels.
    ``planes_from(buffer)`` (with ``shape`` = (B, H, W), ``device``, x None): the producer writes the NOT transposed planes of the first
    layer's input itself (PillarConvLayer: the map never exists in NHWC).  ``planes`` (with ``shape``, ``device``, x None): the first layer's
    input as ``chain_planes`` built it; it is only read (other chains share it)."""
    if planes_from is None and planes is None:
        hip.require_device(x)
        assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
        b, h, w, ct = x.shape
        dev = x.device
    else:
        (b, h, w), dev = shape, device
    lib, st = hip.load(), hip.stream()
    tr = _chain_orientation(layers, b, h, w)
    assert tr is not None, "conv_chain: check conv_chain_supported first"
    assert planes_from is None or not tr, "conv_chain: a planes producer writes the map's own orientation"
    if planes is not None:      # shared input planes: layer 0 reads them, the layers behind alternate between two buffers of their own
        nmid = len(layers) - 1
        n = lib.pn_wino4_planes_floats(b, w if tr else h, h if tr else w, max([8] + [l.cout for l in layers[:-1]]))
        own = [torch.empty(n, dtype=torch.float32, device=dev) if nmid > k else None for k in range(2)]
        bufs = None
    else:
        cmax = max([layers[0].cin] + [l.cout for l in layers[:-1]])
        n = lib.pn_wino4_planes_floats(b, w if tr else h, h if tr else w, cmax)
        bufs = [torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev) if len(layers) > 1 else None]
        if planes_from is None:
            hip.call("pn_wino4_planes_from_nhwc_f32", x.data_ptr(), b, h, w, layers[0].cin, ct, in_channel_offset, int(tr), bufs[0].data_ptr(), st)
        else:
            planes_from(bufs[0])

    def src_dst(k):
        if bufs is not None:
            return bufs[k & 1], bufs[(k + 1) & 1]
        return (planes if k == 0 else own[(k - 1) & 1]), own[k & 1]
    last = layers[-1]
    if out is None:
        out = torch.empty((b, h, w, last.cout), dtype=torch.float32, device=dev)
    assert out.shape[:3] == (b, h, w) and out.is_contiguous()
    prof = S.profiler
    for k, l in enumerate(layers):
        is_last = k == len(layers) - 1
        d = _chain_desc(l, b, h, w, out.shape[3], out_channel_offset, transposed=tr) if is_last else _chain_desc(l, b, h, w, transposed=tr)
        two_d = l.wino24_packed is not None and _chain_two_d(lib, d)
        if two_d and _chain_44(lib, d):
            two_d = "wino44"
            S.chain44_launches += 1
        wts = l.chain_weights(two_d, tr)
        if prof is not None:
            ev = prof.begin(st)
        src, dst = src_dst(k)
        hip.call("pn_conv2d_wino44_chain_f32" if two_d == "wino44" else ("pn_conv2d_wino24_chain_f32" if two_d else "pn_conv2d_wino4_chain_f32"), C.byref(d), src.data_ptr(),
                 wts.data_ptr(), hip.ptr(l.scale), hip.ptr(l.shift),
                 None if is_last else dst.data_ptr(), out.data_ptr() if is_last else None, st)
        if prof is not None:
            flops = 2.0 * b * h * w * l.cout * l.cin * 9
            tag = "F(4,3)xF(4,3) chain" if two_d == "wino44" else ("F(2,3)xF(4,3) chain" if two_d else "F(4,3) chain")
            prof.end(ev, flops, st, tag=f"{h}x{w} {l.cin}->{l.cout} k3 {tag}", issued=flops / 4.0 if two_d == "wino44" else (flops / 3.0 if two_d else None))
    return out


def _chain_two_d(lib, d) -> bool:
    """the chained layer's form: F(2,3) x F(4,3) where the kernel has a 12-wave form for the map; on maps whose 2-D tiles cover less than
    3/4 of the CUs (64 x 64 x 256: 128 blocks) only when other frames run beside this one (they take the free CUs, and the 2-D form
    issues 1.5x fewer MFMAs); alone on the chip the 1-D form's 256 blocks finish sooner"""
    if not R.conv_chain2d or not lib.pn_conv_wino24_chain_supported(C.byref(d)):
        return False
    fh, fw = (d.in_w, d.in_h) if d.transpose_hw else (d.in_h, d.in_w)
    octs, wq = d.batch * (fh // 2) * (fw // 4), fw // 4
    blocks = (octs // (64 if wq > 32 else 32)) * (d.cout // 32)
    return blocks >= 192 or d.frames_in_flight > 1




class chain44:
    """``with ops.chain44(False):`` -- the chained layers launched (or captured) inside the block take F(2,3)xF(4,3) where the
    frames-in-flight hint alone would pick F(4,3)xF(4,3) (``True``: the default rule of ``_chain_44``).  Which form is faster with other
    frames in flight differs from box to box (r5: +2 .. 3 % on the builder's boxes, -4.5 % on the driver's), so engine.FramePipeline
    captures both and keeps the one it MEASURES faster; PN_CONV_CHAIN44=0 still forces the form off process-wide."""

    def __init__(self, on: bool):
        self.on, self.prev = bool(on), True

    def __enter__(self):
        self.prev, S.chain44_route = S.chain44_route, self.on
        return self

    def __exit__(self, *exc):
        S.chain44_route = self.prev
        return False




def chain44_launches_seen() -> int:
    """how many chained-layer launches took the F(4,3)xF(4,3) form so far in this process (engine.FramePipeline: is there a choice to measure?)"""
    return S.chain44_launches


def _chain_44(lib, d) -> bool:
    """F(4,3) x F(4,3) (conv_wchain3_kernel, r5: 2.25 MFMA equivalents per output; one block per four rows x 128 pixels x 32 channels, a 256-pixel
    row as two such halves one after the other): WITH OTHER FRAMES IN FLIGHT, where its blocks are whole rounds of the 256 CUs (256 x 256 x 128:
    256 blocks, 55 against 62 us per layer; a batch of four 128 x 128 maps 55 against 72) or at least half a round (one 128 x 128 map, 128 blocks:
    26 against 30 us for the form the hint picks otherwise; alone on the chip the 256-block K-split form finishes in 21).  Not for a frame
    alone on the chip: the 256 x 256 launch itself is 4 - 9 us shorter there too, but on two of the four boxes it was measured on every OTHER
    matrix kernel of the frame then ran 4 - 5 % longer (853.8 against 843.1 us of kernel time per frame; on the other boxes 796 against 812) --
    the one-frame latency moved by -20 .. +17 us with the box, the in-flight rate rose on all of them (+2 .. 3 %)"""
    if not (R.conv_chain44 and S.chain44_route) or d.frames_in_flight <= 1 or not lib.pn_conv_wino44_chain_supported(C.byref(d)):
        return False
    fh, fw = (d.in_w, d.in_h) if d.transpose_hw else (d.in_h, d.in_w)
    tq = 64 if fw // 4 == 64 else 32
    blocks = (d.batch * (fh // 4) * (fw // 4) // tq) * (d.cout // 32)
    return blocks % 256 == 0 or 128 <= blocks <= 256


# taken when the pillar capacity bounds the (pillar, tap) pairs to this fraction of the dense (output, tap) pairs


class PillarConvLayer:
    """The backbone's first 3x3 convolution on the SPARSE pillar canvas (csrc/pillar_conv.hip; rpn.py:124-142 on the canvas of
    pillar_encoder.py:393-432): (pillar, tap) pairs -> one gathered MFMA GEMM per tap -> fixed-order reduction over the taps with the
    folded BatchNorm + activation.  ``__call__(canvas, vi)``: ``vi`` = the frame's VoxelIndex; every non-zero pixel of ``canvas``
    must be one of its cells."""

    def __init__(self, weight: torch.Tensor, stride: int, scale=None, shift=None, act=ACT_NONE):
        hip.require_device(weight)
        lib = hip.load()
        w = weight.detach().contiguous().float()
        self.cout, self.cin = int(w.shape[0]), int(w.shape[1])
        assert tuple(w.shape[2:]) == (3, 3) and stride in (1, 2) and self.cin in (32, 64, 128) and self.cout % 4 == 0
        self.stride, self.act = int(stride), int(act)
        self.packed = _f32(lib.pn_pillar_conv_packed_weight_floats(self.cout, self.cin), w.device)
        hip.call("pn_pack_pillar_conv_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), hip.stream())
        self.scale = None if scale is None else scale.detach().contiguous().float()
        self.shift = None if shift is None else shift.detach().contiguous().float()
        self.packed_rows = None
        if self.stride == 2 and self.cin % 16 == 0 and self.cout <= 128:      # the row-band form (csrc/pillar_rows.hip) where the frame index leaves row_start
            self.packed_rows = _f32(lib.pn_pillar_conv_rows_packed_weight_floats(self.cout, self.cin), w.device)
            hip.call("pn_pack_pillar_conv_rows_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed_rows.data_ptr(), hip.stream())

    def rows_form(self, vi: "VoxelIndex", b: int, h: int, w: int) -> bool:
        """does this frame take the row-band kernel?  (the fused frame index left row_start and the shape is covered)"""
        return (self.packed_rows is not None and getattr(vi, "row_start", None) is not None and R.pillar_rows
                and bool(hip.load().pn_pillar_conv_rows_supported(b, h, w, self.cin, self.cout, self.stride)))

    @staticmethod
    def supports(conv_weight: torch.Tensor, stride: int, groups: int) -> bool:
        co, ci, kh, kw = conv_weight.shape
        return R.pillar_conv and (kh, kw) == (3, 3) and stride in (1, 2) and groups == 1 and ci in (32, 64, 128) and co % 4 == 0

    def worth_it(self, vi: "VoxelIndex", b: int, h: int, w: int) -> bool:
        """the pillar capacity (known on the host: no sync) bounds the pairs: 9 / stride^2 per pillar.  The pair-list form pays up to 0.35 of
        the dense (output, tap) pairs; the row-band form (no partial rows in memory, the planes straight from LDS) still wins on the 300k-point
        frames of BASELINE configs[4] -- 180k pillars, 2/3 of the cells: p50 1.208 -> 1.186 ms -- so it is taken up to a capacity bound of 1.25
        (the capacity counts points, not pillars; a completely full canvas would lose about a third on this one layer)."""
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        limit = max(R.pillar_conv_max_fill, R.pillar_rows_max_fill) if self.rows_form(vi, b, h, w) else R.pillar_conv_max_fill
        return vi.n_cap * 9.0 / (self.stride * self.stride) <= limit * 9.0 * b * oh * ow

    def planes_supported(self, b: int, h: int, w: int) -> bool:
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        return R.pillar_planes and bool(hip.load().pn_pillar_conv_planes_supported(b, oh, ow, self.cout))

    def __call__(self, canvas: torch.Tensor, vi: "VoxelIndex", out: Optional[torch.Tensor] = None, planes: Optional[torch.Tensor] = None):
        """-> the NHWC output, or (``planes`` given: a buffer of pn_wino4_planes_floats(b, oh, ow, cout) floats) None with the output written as
        the F(4, 3) planes of ops.conv_chain"""
        hip.require_device(canvas)
        lib = hip.load()
        assert canvas.dim() == 4 and canvas.is_contiguous() and canvas.dtype == torch.float32 and canvas.shape[3] >= self.cin
        b, h, w, ct = canvas.shape
        assert (w, h) == (vi.spec.grid[0], vi.spec.grid[1]) and b == vi.batch and vi.spec.grid[2] == 1, "the voxel index does not describe this canvas"
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        if out is None and planes is None:
            out = torch.empty((b, oh, ow, self.cout), dtype=torch.float32, device=canvas.device)
        if self.rows_form(vi, b, h, w):
            st = hip.stream()
            prof = S.profiler
            if prof is not None:
                ev = prof.begin(st)
            if planes is not None:
                assert planes.numel() >= lib.pn_wino4_planes_floats(b, oh, ow, self.cout)
            hip.call("pn_pillar_conv3x3_rows_f32", canvas.data_ptr(), b, h, w, self.cin, ct, 0, vi.unq_keys_ptr, vi.row_start.data_ptr(), vi.n_cap,
                     self.packed_rows.data_ptr(), self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act, hip.ptr(planes),
                     None if planes is not None else out.data_ptr(), 0 if planes is not None else out.shape[3], 0, st)
            if prof is not None:
                # FLOPs actually multiplied: the frame's (pillar, tap) pairs, counted on the host from the key list (profiling runs only)
                v = vi.count()
                k = vi.workspace[:v].to(torch.int64) & 0xffffffff
                ix, iy = k % w, (k // w) % h
                tx = torch.where(ix % 2 == 0, 1, 1 + ((ix + 1) // 2 < ow).long())
                ty = torch.where(iy % 2 == 0, 1, 1 + ((iy + 1) // 2 < oh).long())
                pairs = int((tx * ty).sum())
                prof.end(ev, 2.0 * pairs * self.cout * self.cin, st, tag=f"{oh}x{ow} {self.cin}->{self.cout} k3 pillars (row bands)",
                         dense=2.0 * b * oh * ow * 9 * self.cout * self.cin)
            return out
        nbytes = lib.pn_pillar_conv_workspace_bytes(vi.n_cap, b, oh, ow, self.cout)
        ws = _workspace(nbytes, canvas.device)
        st = hip.stream()
        prof = S.profiler
        if prof is not None:
            ev = prof.begin(st)
        if planes is not None:
            assert planes.numel() >= lib.pn_wino4_planes_floats(b, oh, ow, self.cout)
            hip.call("pn_pillar_conv3x3_planes_f32", canvas.data_ptr(), b, h, w, self.cin, ct, 0, vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap,
                     self.stride, self.packed.data_ptr(), self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act, planes.data_ptr(),
                     ws.data_ptr(), nbytes, st)
        else:
            hip.call("pn_pillar_conv3x3_f32", canvas.data_ptr(), b, h, w, self.cin, ct, 0, vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap, self.stride,
                     self.packed.data_ptr(), self.cout, hip.ptr(self.scale), hip.ptr(self.shift), self.act, out.data_ptr(), out.shape[3], 0,
                     ws.data_ptr(), nbytes, st)
        if prof is not None:
            # FLOPs actually multiplied: the (pillar, tap) pairs of THIS frame (the nine counters head the workspace; reading them
            # synchronises -- profiling runs only); the events bracket the pair, GEMM and reduce kernels
            pairs = int(ws[:36].view(torch.int32).sum().item())
            prof.end(ev, 2.0 * pairs * self.cout * self.cin, st, tag=f"{oh}x{ow} {self.cin}->{self.cout} k3 pillars",
                     dense=2.0 * b * oh * ow * 9 * self.cout * self.cin)
        return out

    # ---- training: pair tables built once per iteration, shared by forward, data gradient and weight gradient
    def repack(self, weight: torch.Tensor, token=None) -> None:
        if token is not None and getattr(self, "_token", None) is token:
            return
        self._token = token
        w = weight.detach().contiguous().float()
        hip.call("pn_pack_pillar_conv_weight_f32", w.data_ptr(), self.cout, self.cin, self.packed.data_ptr(), hip.stream())
        if getattr(self, "packed_t", None) is None:
            self.packed_t = _f32(hip.load().pn_pillar_conv_packed_weight_floats(self.cin, self.cout), w.device)
        wt = w.permute(1, 0, 2, 3).contiguous()          # (Cin, Cout, 3, 3): the data gradient multiplies by W_tap^T, same tap
        hip.call("pn_pack_pillar_conv_weight_f32", wt.data_ptr(), self.cin, self.cout, self.packed_t.data_ptr(), hip.stream())

    def build_tables(self, vi: "VoxelIndex", b: int, h: int, w: int) -> torch.Tensor:
        lib = hip.load()
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        nbytes = lib.pn_pillar_pairs_bytes(vi.n_cap, b, oh, ow)
        tables = torch.empty(nbytes, dtype=torch.uint8, device=vi.num_voxels.device)
        hip.call("pn_pillar_pairs_build", vi.unq_keys_ptr, vi.num_voxels.data_ptr(), vi.n_cap, b, h, w, self.stride, tables.data_ptr(), nbytes, hip.stream())
        return tables

    def forward_tables(self, canvas: torch.Tensor, vi: "VoxelIndex", tables: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, h, w, ct = canvas.shape
        oh, ow = (h - 1) // self.stride + 1, (w - 1) // self.stride + 1
        if out is None:
            out = torch.empty((b, oh, ow, self.cout), dtype=torch.float32, device=canvas.device)
        cap = (vi.n_cap + 127) // 128 * 128
        ws = _workspace(9 * cap * self.cout * 4, canvas.device)
        hip.call("pn_pillar_conv3x3_tables_f32", canvas.data_ptr(), b, oh, ow, self.cin, ct, 0, tables.data_ptr(), vi.n_cap, self.packed.data_ptr(), self.cout,
                 hip.ptr(self.scale), hip.ptr(self.shift), self.act, out.data_ptr(), out.shape[3], 0, ws.data_ptr(), ws.numel(), hip.stream())
        return out

    def dgrad_features(self, dout: torch.Tensor, vi: "VoxelIndex", tables: torch.Tensor) -> torch.Tensor:
        """d(pillar features) (n_cap, Cin), rows in the order of the index's cells (what ``dynamic_pfn_bwd`` takes as d_features)"""
        hip.require_device(dout)
        assert dout.is_contiguous() and dout.shape[3] >= self.cout
        b, oh, ow, ct = dout.shape
        dfeat = torch.empty((max(vi.n_cap, 1), self.cin), dtype=torch.float32, device=dout.device)
        cap = (vi.n_cap + 127) // 128 * 128
        ws = _workspace(9 * cap * self.cin * 4, dout.device)
        hip.call("pn_pillar_conv3x3_dgrad_f32", dout.data_ptr(), b, oh, ow, self.cout, ct, 0, tables.data_ptr(), vi.num_voxels.data_ptr(), vi.n_cap,
                 self.packed_t.data_ptr(), self.cin, dfeat.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream())
        return dfeat

    def wgrad(self, canvas: torch.Tensor, dout: torch.Tensor, vi: "VoxelIndex", tables: torch.Tensor, out: Optional[torch.Tensor] = None,
              accumulate=False) -> torch.Tensor:
        lib = hip.load()
        b, oh, ow, ct = dout.shape
        if out is None:
            out = torch.empty((self.cout, self.cin, 3, 3), dtype=torch.float32, device=dout.device)
        nbytes = lib.pn_pillar_conv_wgrad_workspace_bytes(vi.n_cap, self.cin, self.cout)
        ws = _workspace(nbytes, dout.device)
        hip.call("pn_pillar_conv3x3_wgrad_f32", canvas.data_ptr(), canvas.shape[3], 0, self.cin, dout.data_ptr(), ct, 0, self.cout, tables.data_ptr(), vi.n_cap,
                 b, oh, ow, out.data_ptr(), int(accumulate), ws.data_ptr(), nbytes, hip.stream())
        return out


class ConvJob:
    """One convolution of a multi-job launch (``conv_multi``): a packed ``ConvLayer`` applied to a channel slice of ``x``,
    writing a channel slice of ``out``; optionally emitting the statistics of the GroupNorm-family layer that follows it
    (``stats=dict(strata, channel_groups, gamma, beta, eps, affine=, affine_strata=, mean_rstd=)``) and / or reading its input
    through the affine table of the norm that precedes it (``norm=(table, strata, channels)``: relu(x*A + B) on load)."""

    def __init__(self, layer: "ConvLayer", x: torch.Tensor, out: torch.Tensor, in_channel_offset=0, out_channel_offset=0, stats=None,
                 norm=None):
        assert layer.dtype == "f32" and not layer.deconv2x2
        hip.require_device(x, out)
        assert x.dim() == 4 and x.is_contiguous() and out.is_contiguous() and x.dtype == torch.float32
        self.layer, self.x, self.out, self.in_co, self.out_co, self.stats, self.norm = layer, x, out, in_channel_offset, out_channel_offset, stats, norm
        b, h, w, ct = x.shape
        oh, ow = layer.out_hw(h, w)
        assert out.shape[:3] == (b, oh, ow)
        self.desc = ConvDesc(b, h, w, layer.cin, layer.cout, layer.groups, layer.kh, layer.kw, layer.stride, layer.pad[0], layer.pad[1],
                             ct, in_channel_offset, out.shape[3], out_channel_offset, layer.act, 0, layer.range_strata, 0, 0, 0)
        self.macs = b * oh * ow * layer.groups * layer.cout * layer.cin * layer.kh * layer.kw

    def partial_floats(self, tile: int) -> int:
        return int(hip.load().pn_conv_stat_partial_floats(C.byref(self.desc), tile))



def _job_array(jobs: Sequence[ConvJob]):
    arr = (hip.ConvJob * len(jobs))()
    for k, jb in enumerate(jobs):
        c = arr[k]
        c.desc = jb.desc
        jb.layer._ensure("direct")
        c.in_, c.packed_w, c.out = jb.x.data_ptr(), jb.layer.packed.data_ptr(), jb.out.data_ptr()
        c.scale, c.shift = hip.ptr(jb.layer.scale), hip.ptr(jb.layer.shift)
        if jb.stats is not None:
            st = jb.stats
            c.stat_partials = st["partials"].data_ptr()
            c.stat_strata, c.stat_channel_groups = int(st.get("strata", 1)), int(st["channel_groups"])
            c.stat_gamma, c.stat_beta, c.stat_eps = hip.ptr(st.get("gamma")), hip.ptr(st.get("beta")), float(st["eps"])
            c.stat_affine_strata = int(st["affine_strata"])
            c.stat_affine, c.stat_mean_rstd = hip.ptr(st.get("affine")), hip.ptr(st.get("mean_rstd"))
        if jb.norm is not None:
            tab, strata, channels = jb.norm
            c.norm_affine, c.norm_strata, c.norm_channels = tab.data_ptr(), int(strata), int(channels)
    return arr


def conv_multi(jobs: Sequence[ConvJob], tile: int) -> None:
    """run the jobs as ONE launch of the MFMA kernel (tile: 1 = 128x128, 3 = 64x64, 4 = 64x32, 5 = 64x128).
    Statistics jobs need ``stats['partials']`` (float scratch of ``job.partial_floats(tile)``); ``conv_stats_finalize`` /
    ``conv_stats_apply`` turn the partials into the norm's affine table / apply the norm."""
    arr = _job_array(jobs)
    st = hip.stream()
    prof = S.profiler
    if prof is not None:
        ev = prof.begin(st)
    hip.call("pn_conv2d_multi_f32", arr, len(jobs), int(tile), st)
    if prof is not None:
        j0 = jobs[0]
        prof.end(ev, 2.0 * sum(j.macs for j in jobs), st,
                 tag=f"multi x{len(jobs)} {j0.out.shape[1]}x{j0.out.shape[2]} {sum(j.layer.cin * j.layer.groups for j in jobs)}->{sum(j.layer.out_channels for j in jobs)} k{j0.layer.kh}")


def conv_small_n_multi(jobs: Sequence[ConvJob]) -> None:
    """the jobs (1x1 / 3x3, <= 64 input channels, <= 12 output columns each) as ONE launch of the VALU kernel"""
    st = hip.stream()
    prof = S.profiler
    if prof is not None:
        ev = prof.begin(st)
    hip.call("pn_conv2d_small_n_multi_f32", _job_array(jobs), len(jobs), st)
    if prof is not None:
        j0 = jobs[0]
        prof.end(ev, 2.0 * sum(j.macs for j in jobs), st,
                 tag=f"small-n x{len(jobs)} {j0.out.shape[1]}x{j0.out.shape[2]} {sum(j.layer.cin * j.layer.groups for j in jobs)}->{sum(j.layer.out_channels for j in jobs)}")


def conv_stats_finalize(jobs: Sequence[ConvJob], tile: int) -> None:
    """fold the statistics partials of the jobs (same list / tile as the ``conv_multi`` call) into their affine tables"""
    hip.call("pn_conv_stats_finalize_f32", _job_array(jobs), len(jobs), int(tile), hip.stream())


def conv_stats_apply(producer: ConvJob, tile: int, gamma, beta, act, out: torch.Tensor, out_channel_offset=0, mul=None, add=None,
                     out2: Optional[torch.Tensor] = None, out2_channel_offset=0) -> None:
    """RSNorm + activation (+ calibrated copy) of ``producer.out`` from the partials its epilogue wrote; no finalize launch"""
    hip.require_device(out)
    hip.call("pn_conv_stats_apply_f32", _job_array([producer]), int(tile), hip.ptr(gamma), hip.ptr(beta), int(act), out.data_ptr(), out.shape[3],
             out_channel_offset, hip.ptr(mul), hip.ptr(add), hip.ptr(out2), 0 if out2 is None else out2.shape[3], out2_channel_offset, hip.stream())


def groupnorm_apply(x: torch.Tensor, channel_groups: int, range_strata: int, mean_rstd: torch.Tensor, gamma, beta, act, out: torch.Tensor,
                    out_channel_offset=0, mul=None, add=None, out2: Optional[torch.Tensor] = None, out2_channel_offset=0,
                    channels: Optional[int] = None, channel_offset=0) -> None:
    """normalisation pass with the statistics already on the device (from a convolution's epilogue)"""
    hip.require_device(x, out, mean_rstd)
    b, h, w, ct = x.shape
    c = ct - channel_offset if channels is None else channels
    hip.call("pn_groupnorm_apply_f32", x.data_ptr(), b, h, w, c, ct, channel_offset, channel_groups, range_strata, mean_rstd.data_ptr(),
             hip.ptr(gamma), hip.ptr(beta), int(act), out.data_ptr(), out.shape[3], out_channel_offset, hip.ptr(mul), hip.ptr(add), hip.ptr(out2),
             0 if out2 is None else out2.shape[3], out2_channel_offset, hip.stream())


def conv2d_direct(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride=1, pad=0, groups=1,
                  act=ACT_NONE) -> torch.Tensor:
    """plain direct convolution (any channel counts); NHWC in / NHWC out"""
    hip.require_device(x, weight)
    w = weight.detach().contiguous().float()
    cout_t, cin_g, kh, kw = w.shape
    b, h, wd, ct = x.shape
    oh, ow = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    out = torch.empty((b, oh, ow, cout_t), dtype=torch.float32, device=x.device)
    d = ConvDesc(b, h, wd, cin_g, cout_t // groups, groups, kh, kw, stride, pad, pad, ct, 0, cout_t, 0, act, 0, 0)
    sh = None if bias is None else bias.detach().contiguous().float()
    hip.call("pn_conv2d_direct_nhwc_f32", C.byref(d), x.data_ptr(), w.data_ptr(), None, hip.ptr(sh), out.data_ptr(),
             hip.stream())
    return out


def fold_bn(gamma, beta, mean, var, eps: float, conv_bias=None):
    hip.require_device(gamma)
    c = gamma.numel()
    scale, shift = _f32(c, gamma.device), _f32(c, gamma.device)
    args = [t.detach().contiguous().float() for t in (gamma, beta, mean, var)]
    cb = None if conv_bias is None else conv_bias.detach().contiguous().float()
    hip.call("pn_fold_bn_f32", *(t.data_ptr() for t in args), hip.ptr(cb), float(eps), c, scale.data_ptr(),
             shift.data_ptr(), hip.stream())
    return scale, shift


# ------------------------------------------------------------------------------ norms
def groupnorm_strat(x: torch.Tensor, channel_groups: int, range_strata: int, gamma: torch.Tensor, beta: torch.Tensor,
                    eps=1e-5, act=ACT_NONE, out: Optional[torch.Tensor] = None, mul: Optional[torch.Tensor] = None,
                    add: Optional[torch.Tensor] = None, stat_out: Optional[torch.Tensor] = None):
    """x NHWC (B,H,W,C).  gamma/beta have range_strata*C entries in stacked order [stratum][channel].
    Returns out, or (out, out*mul+add) when mul/add ((H,W,C) maps) are given.  ``stat_out`` (B * strata * groups * 2 floats): keeps the
    (mean, rstd) pairs for ``groupnorm_strat_bwd(..., stat=)``."""
    hip.require_device(x)
    lib = hip.load()
    assert x.is_contiguous()
    b, h, w, c = x.shape
    if out is None:
        out = torch.empty_like(x)
    out2 = torch.empty_like(x) if mul is not None else None
    ws_bytes = lib.pn_groupnorm_workspace_bytes(b, channel_groups, range_strata)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    assert stat_out is None or (stat_out.numel() >= 2 * b * range_strata * channel_groups and stat_out.dtype == torch.float32)
    hip.call("pn_groupnorm_strat_fwd_stat", x.data_ptr(), b, h, w, c, c, 0, channel_groups, range_strata, hip.ptr(gamma),
             hip.ptr(beta), float(eps), int(act), out.data_ptr(), c, 0, hip.ptr(mul), hip.ptr(add), hip.ptr(out2), hip.ptr(stat_out),
             ws.data_ptr(), ws_bytes, hip.stream())
    return out if out2 is None else (out, out2)
