"""Frame-level data parallelism helpers (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).

The inference hot path shards by frame with no data-path collective (frames are independent,
like the reference's DistributedSampler, det3d/datasets/loader/sampler.py:74-96); the only
exchanges are the barrier around the timed region and a MAX-reduce of the elapsed time.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device: Optional[torch.device] = None, timeout_s: Optional[float] = None, single_rank_group: bool = False) -> bool:
    """initialise the default process group from the torchrun environment; False when world size is 1.
    ``timeout_s`` bounds every collective wait (a rank that faults leaves the others with an error, not a hang).
    ``single_rank_group``: create the group at world size 1 too -- the one-GPU box's way to run the RCCL code path itself
    (communicator creation with ``device_id``, asynchronous collectives on device slices, stream-level waits); the exchange
    helpers below then take ``force=True``."""
    _, _, world = env_rank_world()
    if world <= 1 and not single_rank_group:
        return False
    if world <= 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if timeout_s is not None:
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return True


def frame_shard(frame_ids: List[int], rank: int, world: int) -> List[int]:
    """frames of this rank: frame i goes to rank i mod world (round-robin, no padding)"""
    return [f for k, f in enumerate(frame_ids) if k % world == rank]


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ---- training step: the one exchange per iteration ------------------------------------
def _group_live(force: bool) -> bool:
    """a collective is issued when the group has more than one rank, or on request at world size 1 (``force``)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)


def allreduce_flat_grads(flat_g: torch.Tensor, force: bool = False) -> None:
    """SUM all-reduce of the flat fp32 gradient buffer, in place (one collective per step; the
    reference coalesces per-parameter grads into buckets, det3d/core/utils/dist_utils.py:8-28).
    The 1/world of the reference's average is folded into the loss gradient by the caller, so after
    this call every rank holds the mean gradient."""
    if _group_live(force):
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)


class GradExchange:
    """Bucketed gradient exchange overlapped with backward (the reference's DDP reducer buckets in reverse layer
    order, det3d/torchie/apis/train.py:330-336).  ``buckets`` are contiguous [lo, hi) ranges of the flat gradient buffer in the
    order backward completes them; ``ready(k)`` starts the asynchronous SUM all-reduce of bucket k (RCCL's stream waits for the
    kernels queued so far and runs next to the rest of backward), ``finish()`` starts whatever has not been started and makes
    the compute stream wait for all of them.  World size 1: no-ops, unless ``force`` (the collectives then run over the
    one-rank group and leave the buffer as it was -- the path itself is what is exercised)."""

    def __init__(self, flat_g: torch.Tensor, buckets, force: bool = False):
        self.flat_g, self.buckets = flat_g, list(buckets)
        self.active = _group_live(force)
        self.issued = [False] * len(self.buckets)
        self.pending = []

    def ready(self, k: int) -> None:
        if self.active and k < len(self.buckets) and not self.issued[k]:
            lo, hi = self.buckets[k]
            self.issued[k] = True
            self.pending.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> None:
        for k in range(len(self.buckets)):
            self.ready(k)
        for w in self.pending:
            w.wait()   # stream-level wait on GPU backends; the host is not blocked
        self.pending = []


def broadcast_flat_params(flat_p: torch.Tensor, src: int = 0, force: bool = False) -> None:
    """rank `src` -> all, once before training (what DistributedDataParallel's constructor does,
    det3d/torchie/apis/train.py:330-336)"""
    if _group_live(force):
        dist.broadcast(flat_p, src=src)


class BufferSync:
    """DistributedDataParallel's ``broadcast_buffers=True`` (its default, which det3d/torchie/apis/train.py:330-336 keeps): at the start of every
    forward rank 0's BUFFERS -- the BatchNorm running statistics -- replace every other rank's, so the ranks never drift apart in what an
    evaluation between epochs would see.  (The reference converts to SyncBN only when apex is installed; without it -- "No APEX!" -- this
    broadcast is all the coupling the BatchNorm layers of the ranks have, and it is what this class reproduces.)
    The floating-point buffers of ``model`` are re-pointed, once, at views of ONE flat tensor (as ParamStore does with the parameters), so a
    sync is a single broadcast with no packing; integer buffers (``num_batches_tracked``) advance in lock step and are left alone."""

    def __init__(self, model: torch.nn.Module, device=None):
        bufs = [(n, b) for n, b in model.named_buffers() if b is not None and b.dtype == torch.float32 and b.numel() > 0]
        device = device if device is not None else (bufs[0][1].device if bufs else torch.device("cpu"))
        self.names = [n for n, _ in bufs]
        self.total = sum(b.numel() for _, b in bufs)
        self.flat = torch.empty(self.total, dtype=torch.float32, device=device)
        off = 0
        for _, b in bufs:
            n = b.numel()
            view = self.flat[off:off + n].view(b.shape)
            view.copy_(b.detach().to(device))
            b.data = view
            off += n

    def sync(self, src: int = 0, force: bool = False) -> None:
        """rank ``src``'s buffers to every rank (one broadcast); a no-op without a process group of more than one rank (unless ``force``)"""
        if self.total and _group_live(force):
            dist.broadcast(self.flat, src=src)
