"""Per-frame hipGraph engine: the whole hot path of one step (cart->polar, voxelize, PFN, RPN,
head) is captured once into a HIP graph and replayed per frame -- no Python / launch overhead
between the ~90 kernels of a frame.  Every kernel of the path is capturable by construction
(no host sync, no allocation inside the C ABI, voxel count stays on the device)."""
from __future__ import annotations

from typing import Dict

import torch

from . import hip, ops


class FrameEngine:
    def __init__(self, model, batch: int, points_per_sweep: int, spec: ops.GridSpec = None, point_features: int = 5):
        hip.load()
        self.model = model.eval()
        dev = next(model.parameters()).device
        hip.require_device(next(model.parameters()))
        self.batch, self.n = batch, points_per_sweep
        self.spec = spec or ops.GridSpec.from_range(model.reader.pc_range, model.reader.voxel_size)
        # static input buffer; pre-filled with a spread-out synthetic sweep so that the capture warm-up does not
        # run the degenerate "every point in one pillar" case
        from .utils import synth
        import numpy as np
        init = np.concatenate([synth.synth_sweep_cart(points_per_sweep, seed=977 + b) for b in range(batch)], 0)
        if point_features != init.shape[1]:
            init = np.concatenate([init, np.zeros((init.shape[0], point_features - init.shape[1]), np.float32)], 1) \
                if point_features > init.shape[1] else init[:, :point_features]
        self.cart = torch.from_numpy(np.ascontiguousarray(init)).to(dev)
        self.offsets = torch.tensor([points_per_sweep * b for b in range(batch + 1)], dtype=torch.int32, device=dev)
        self.graph = None
        self.outputs: Dict[str, torch.Tensor] = {}

    def _step(self):
        polar = ops.cart_to_polar(self.cart)
        return self.model.forward_points(polar, self.offsets, self.batch, self.spec)

    def capture(self, warmup: int = 3, stream: "torch.cuda.Stream" = None) -> "FrameEngine":
        """capture the frame into a HIP graph; `stream` (optional) is the stream the graph will be
        replayed on (several engines on different streams overlap their frames on the GPU)"""
        self.stream = stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # builds plans / packs weights / sets kernel attributes outside the capture
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = self._step()
        return self

    def run(self, cart: torch.Tensor) -> Dict[str, torch.Tensor]:
        """cart: (batch*points, 5) on the device.  Returns the head tensors (static buffers,
        overwritten by the next call)."""
        if self.graph is None:
            self.capture()
        if getattr(self, "stream", None) is not None:
            with torch.cuda.stream(self.stream):
                self.cart.copy_(cart, non_blocking=True)
                self.graph.replay()
        else:
            self.cart.copy_(cart, non_blocking=True)
            self.graph.replay()
        return self.outputs
