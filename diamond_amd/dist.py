"""Data parallelism for the imagined-rollout path: one process per GPU, imagination envs
sharded on the batch axis (no data-path collective), ONE all-reduce(mean) of the actor-critic
gradients per optimiser step -- the only collective on the hot path (reference: DDP hooks fired
by loss.backward(), trainer.py:366 / utils.py:105-106).

MI355X/xGMI design choice: all gradients live in one contiguous fp32 bucket (3.2 M floats =
12.9 MB at 64x64), reduced with a single RCCL call -- xGMI is point-to-point, so one large
message uses all 7 links at once instead of 28 latency-bound per-tensor rings.  Parameters'
`.grad` are views into the bucket, so there is no gather/scatter copy around the collective.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist
from torch import Tensor, nn


class GradAllReducer:
    def __init__(self, params: Sequence[nn.Parameter], group=None) -> None:
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.bucket = torch.zeros(total, device=ref.device, dtype=ref.dtype)
        # measurement hook (bench.py): with `timing` on, every all_reduce_mean is bracketed by two events on the current
        # stream (the collective runs on RCCL's own stream, but a blocking dist.all_reduce makes the current stream wait
        # for it, so the pair spans it); `elapsed_ms()` reads them after a synchronisation
        self.timing = False
        self._events: List = []
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.bucket[off:off + n].view_as(p)  # autograd accumulates in place into the view
            off += n

    def _ensure_views(self) -> None:
        """If something replaced p.grad (e.g. zero_grad(set_to_none=True)), copy back into the bucket."""
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.bucket[off:off + n].view_as(p)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n

    @torch.no_grad()
    def all_reduce_mean(self) -> Tensor:
        self._ensure_views()
        # (also at world size 1: the same RCCL call on the same bucket, so that a 1-GPU run of the distributed path
        #  exercises everything an N-GPU run does)
        if dist.is_available() and dist.is_initialized():
            timed = self.timing and self.bucket.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group)
            if dist.get_world_size(self.group) > 1:
                self.bucket.div_(dist.get_world_size(self.group))
            if timed:
                e1.record()
                self._events.append((e0, e1))
        return self.bucket

    def elapsed_ms(self) -> List[float]:
        """Device time of every timed all_reduce_mean since the last call (synchronise first); clears the list."""
        out = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return out


def broadcast_parameters(module: nn.Module, src: int = 0, group=None) -> None:
    """Startup parameter/buffer broadcast (what the DDP constructor does, utils.py:106): ONE flat message per dtype
    (xGMI is point-to-point: a single large broadcast instead of hundreds of latency-bound small ones), copied back
    with `copy_` -- c10d collectives write through the storage without bumping `Tensor._version`, and the packed
    kernel-layout copies (engine.PackCache / FilmTable) are cached per parameter version."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        tensors = list(module.parameters()) + list(module.buffers())
        for dtype in sorted({t.dtype for t in tensors}, key=str):
            group_t = [t for t in tensors if t.dtype == dtype]
            flat = torch.cat([t.detach().reshape(-1) for t in group_t])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in group_t:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))  # in-place on the tensor itself: bumps _version
                off += n


@torch.no_grad()
def parameter_checksum(module: nn.Module) -> float:
    """Order-dependent fp64 checksum of all parameters: equal on every rank iff the replicas hold the same values
    (bench.py asserts it after the timed steps)."""
    total = torch.zeros((), dtype=torch.float64, device=next(module.parameters()).device)
    for i, p in enumerate(module.parameters()):
        total += p.detach().double().sum() * (1.0 + (i % 7) * 0.125)
    return float(total)
