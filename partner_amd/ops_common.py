"""Shared helpers of the stage modules (ops_index / ops_conv / ops_token / ops_train; ``ops`` re-exports everything): NHWC layout at the det3d
boundary, scratch buffers, bf16 conversion, and the HIP streams that really overlap (the runtime multiplexes streams onto a few hardware
queues).  Nothing here touches the oracle and nothing falls back to PyTorch arithmetic."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import hip
from .routes import R, S  # noqa: F401


def _f32(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


# ------------------------------------------------------------------------------ layout
def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) logical tensor -> contiguous (B,H,W,C)."""
    hip.require_device(x)
    assert x.dim() == 4 and x.dtype == torch.float32
    xp = x.permute(0, 2, 3, 1)
    if xp.is_contiguous():
        return xp
    x = x.contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
    hip.call("pn_nchw_to_nhwc_f32", x.data_ptr(), b, c, h, w, out.data_ptr(), hip.stream())
    return out


def as_nchw(x_nhwc: torch.Tensor) -> torch.Tensor:
    """(B,H,W,C) -> logical (B,C,H,W) view (channels-last strides, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


def pixel_stride(t: torch.Tensor) -> int:
    """pixel stride (floats) of a logical (B, c, H, W) tensor that is a channels-last view (possibly a channel slice of a wider NHWC
    map); raises if it is not one.  A size-1 channel dimension may carry any stride."""
    b, c, h, w = t.shape
    ok = (c == 1 or t.stride(1) == 1) and t.stride(2) == w * t.stride(3) and (b == 1 or t.stride(0) == h * t.stride(2)) and t.stride(3) >= c
    if not ok:
        raise hip.PartnerHipError("head tensors must be channels-last (NHWC-backed) views")
    return t.stride(3)


def nhwc_slice_to_nchw(x_nhwc: torch.Tensor, c0: int, c: int) -> torch.Tensor:
    """contiguous NCHW copy of channels [c0, c0+c) of an NHWC tensor"""
    b, h, w, ct = x_nhwc.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=x_nhwc.device)
    hip.call("pn_nhwc_to_nchw_f32", x_nhwc.data_ptr(), b, c, h, w, ct, c0, out.data_ptr(), hip.stream())
    return out

_WS = {}


def _workspace(nbytes: int, dev) -> torch.Tensor:
    """grow-only scratch buffer per (device, stream): the backward kernels need their workspace only
    until the launch that consumes it has been queued on the same stream"""
    if torch.cuda.is_current_stream_capturing():
        # every hipGraph capture runs on torch's shared capture stream: a cached buffer keyed by the stream would be shared by all
        # captured engines (and owned by the first graph's pool) -- a race once the engines replay concurrently.  Inside a capture
        # the scratch is a plain allocation of that graph's private pool.
        return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    key = (str(dev), hip.stream())
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = t
    return t


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """f32 -> bf16 (round to nearest even) on the HIP kernel; same shape"""
    hip.require_device(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    hip.call("pn_f32_to_bf16", x.data_ptr(), y.data_ptr(), x.numel(), hip.stream())
    return y


def to_f32(x: torch.Tensor) -> torch.Tensor:
    hip.require_device(x)
    assert x.is_contiguous() and x.dtype == torch.bfloat16
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    hip.call("pn_bf16_to_f32", x.data_ptr(), y.data_ptr(), x.numel(), hip.stream())
    return y


# ------------------------------------------------------------------------------ streams


_CONCURRENT: dict = {}


def concurrent_stream(device=None) -> "torch.cuda.Stream":
    """A stream whose work really overlaps with the CURRENT stream's.  The HIP runtime multiplexes streams onto a handful of hardware queues
    (four by default) in creation order; two streams that land on the same queue run one after the other.  Which stream collides with
    which depends on how many streams the process made before -- measured with the training step: default + second stream 13.9 ms per
    iteration, 15.3 ms (the one-stream time) when exactly six other streams had been created earlier, and with GPU_MAX_HW_QUEUES=8 the
    collision just moves (tools/hwq.py).  So the choice is measured: candidates are probed with two spin kernels of ~0.3 ms, one on the
    current stream and one on the candidate, and the first candidate that finishes the pair in about the time of one is kept (cached per
    current stream).  Inside a hipGraph capture nothing is probed: a graph's branches are scheduled by the graph, not by these streams."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    main = torch.cuda.current_stream(dev)
    key = (str(dev), main.cuda_stream)
    got = _CONCURRENT.get(key)
    if got is not None:
        return got
    if torch.cuda.is_current_stream_capturing():
        any_key = (str(dev), "capture")
        if any_key not in _CONCURRENT:
            _CONCURRENT[any_key] = next((v for k, v in _CONCURRENT.items() if k[0] == str(dev)), None) or torch.cuda.Stream(device=dev)
        return _CONCURRENT[any_key]
    cycles = 600000

    def pair_ms(cand):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if cand is not None:
            cand.wait_stream(main)
            with torch.cuda.stream(cand):
                torch.cuda._sleep(cycles)
        torch.cuda._sleep(cycles)
        if cand is not None:
            main.wait_stream(cand)
        e1.record(main)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    pair_ms(None)
    one = min(pair_ms(None) for _ in range(2))
    best, best_t = None, float("inf")
    for _ in range(12):
        cand = torch.cuda.Stream(device=dev)
        t = min(pair_ms(cand) for _ in range(2))
        if t < best_t:
            best, best_t = cand, t
        if t < 1.4 * one:
            break
    _CONCURRENT[key] = best
    return best


def probe_streams(device=None) -> None:
    """Explicit form of the probing ``concurrent_stream`` does on first use: ~60 device-wide synchronisations and up to twelve
    stream creations.  Serving / training loops call it once up front, on the stream they will run on and BEFORE any hipGraph capture
    starts on another thread (a device-wide synchronisation invalidates a capture in progress); afterwards ``concurrent_stream`` is a
    dictionary lookup.  The frames-in-flight hint (``frames_in_flight``) is a process global: engines are captured from one thread."""
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        concurrent_stream(device)


def concurrent_streams(k: int, device=None, candidates: int = 16):
    """k streams that overlap with EACH OTHER (several hipGraph engines replaying at once: engines whose streams share a hardware queue run
    their frames one after the other).  Greedy: a candidate joins the set when a spin kernel on it and one on every member finish in about
    the time of one; if the runtime has fewer independent queues than k, the best candidates found fill the set."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    main = torch.cuda.current_stream(dev)
    cycles = 600000

    def pair_ms(a, b):
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for st in (a, b):
            if st is not None:
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
        for st in (a, b):
            if st is not None:
                main.wait_stream(st)
        e1.record(main)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    first = torch.cuda.Stream(device=dev)
    pair_ms(first, None)
    one = min(pair_ms(first, None) for _ in range(2))
    chosen, spare = [first], []
    for _ in range(candidates):
        if len(chosen) >= k:
            break
        cand = torch.cuda.Stream(device=dev)
        worst = max(min(pair_ms(cand, m) for _ in range(2)) for m in chosen)
        if worst < 1.4 * one:
            chosen.append(cand)
        else:
            spare.append((worst, cand))
    spare.sort(key=lambda wc: wc[0])
    while len(chosen) < k and spare:
        chosen.append(spare.pop(0)[1])
    while len(chosen) < k:
        chosen.append(torch.cuda.Stream(device=dev))
    return chosen[:k]


class SideStream:
    """Weight gradients off the critical path of backward: dW of a layer is needed by nobody before the gradient exchange / the optimizer,
    while the chain  d(out) -> BatchNorm backward -> data gradient -> previous layer  is serial.  ``run`` queues a layer's weight-gradient
    launches (kernel + slice reduction + bias sums) on a second HIP stream behind the work queued so far; the data gradient goes on on the
    main stream and the two overlap -- the 64 x 64 / 128 x 128 layers do not fill the chip on their own.  ``join`` makes the main stream
    wait (before a gradient bucket is handed to the exchange, and at the end of backward).  Same kernels, same results: nothing here
    depends on the order two independent kernels finish in.  The second stream is picked on first use so that it really overlaps with the
    caller's stream (``concurrent_stream``).  ``PN_TRAIN_WGRAD_STREAM=0`` keeps everything on one stream."""

    def __init__(self, device):
        self.on = R.train_wgrad_stream and torch.device(device).type == "cuda" and torch.cuda.is_available()
        self.device = device
        self._dirty: list = []        # every side stream that has run something since the last join
        self.keep: list = []

    @property
    def stream(self):
        """the second stream for the CURRENT stream (None when switched off)"""
        if not self.on:
            return None
        return concurrent_stream(self.device)

    @stream.setter
    def stream(self, value):
        if value is None:
            self.on = False

    def run(self, fn, *reads, after=None):
        """``after``: an event of the main stream the launches wait for instead of everything queued on it so far"""
        side = self.stream
        if side is None:
            fn()
            return
        if after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        # the buffers the side stream reads stay referenced until the join: freed earlier, the caching allocator would hand them to the
        # main stream again while the side stream still reads them.  (Tensor.record_stream does the same bookkeeping inside the allocator,
        # but with hundreds of large cross-stream blocks per iteration it kept the allocator from reusing memory: the PARTNER detector's
        # training iteration went from 103 to 184 ms.)
        self.keep.extend(t for t in reads if t is not None)
        if not any(side is x for x in self._dirty):
            self._dirty.append(side)

    @property
    def dirty(self) -> bool:
        return bool(self._dirty)

    def join(self):
        """the current stream waits for EVERY side stream used since the last join (the instance is shared per device and the side stream
        depends on the caller's current stream: run() from two different streams between joins leaves two of them dirty), then the
        read buffers are released"""
        if self._dirty:
            cur = torch.cuda.current_stream()
            for side in self._dirty:
                cur.wait_stream(side)
            self._dirty.clear()
            self.keep.clear()


_SHARED_SIDE: dict = {}


def shared_side_stream(device) -> SideStream:
    """ONE side stream per device for the tapes (a tape lives for one iteration; a HIP stream made per iteration would also get a fresh
    pool in the caching allocator, i.e. a hipMalloc for every buffer it ever allocates)"""
    key = str(torch.device(device))
    if key not in _SHARED_SIDE:
        _SHARED_SIDE[key] = SideStream(device)
    return _SHARED_SIDE[key]
