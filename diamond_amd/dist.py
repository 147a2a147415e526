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
    """early: parameters whose gradients are FINAL before the backward pass is over (the actor-critic's LSTM and heads: autograd
    walks the window from its last step to its first and, inside a step, from the heads through the LSTM to the encoder -- so the
    LSTM / head gradients are complete when step 0's LSTM node has run, while step 0's encoder backward is still to come).  They sit
    at the front of the bucket, and as soon as autograd has accumulated the last of them (post-accumulate-grad hooks: each fires
    once per backward, after the parameter's last use) their slice is all-reduced on a SIDE stream, under the rest of the backward
    pass -- what DDP's bucket hooks do for the reference (utils.py:105-106), specialised to the two moments this path has.  At
    64x64 the slice is 12.6 of 12.9 MB and the encoder's last backward ~1.4 ms; at 256x256 (configs[4]) it is 134 of 138.7 MB
    (`lstm.weight_ih` is 2048 x 16384) against an encoder backward of tens of milliseconds.  all_reduce_mean() then reduces the rest
    and waits for the early slice.  (One backward pass per all_reduce_mean(): with gradient accumulation over several passes
    construct it without `early`.)"""

    def __init__(self, params: Sequence[nn.Parameter], group=None, early: Sequence[nn.Parameter] = ()) -> None:
        early_ids = {id(p) for p in early if p.requires_grad}
        ps = [p for p in params if p.requires_grad]
        self.params: List[nn.Parameter] = [p for p in ps if id(p) in early_ids] + [p for p in ps if id(p) not in early_ids]
        self.group = group
        self.num_early = sum(1 for p in self.params if id(p) in early_ids)
        self.early_numel = sum(p.numel() for p in self.params[:self.num_early])
        self._early_left = self.num_early
        self._early_work = None   # (async work handle, side stream | None) of the early slice's all-reduce in flight
        self._side = None
        self.early_launches = 0   # (tests: the early slice went out from inside backward())
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.bucket = torch.zeros(total, device=ref.device, dtype=ref.dtype)
        self._hooks = [p.register_post_accumulate_grad_hook(self._early_ready) for p in self.params[:self.num_early]]
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

    def close(self) -> None:
        """Remove the early slice's hooks from the parameters (a reducer that is replaced by another one)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

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

    def _early_ready(self, p: Tensor) -> None:
        """post-accumulate-grad hook of an early parameter: the last one of a backward pass sends the early slice off"""
        self._early_left -= 1
        if self._early_left > 0:
            return
        self._early_left = self.num_early
        if not (dist.is_available() and dist.is_initialized()) or self._early_work is not None:
            return
        # the gradients must have been accumulated INTO the bucket's views (a replaced .grad is copied back by _ensure_views
        # in all_reduce_mean: then this backward's early slice goes with the rest)
        off = 0
        for q in self.params[:self.num_early]:
            if q.grad is None or q.grad.data_ptr() != self.bucket.data_ptr() + off * self.bucket.element_size():
                return
            off += q.numel()
        sl = self.bucket[:self.early_numel]
        if self.bucket.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.bucket.device)
            self._side.wait_stream(torch.cuda.current_stream())  # the accumulations are on the current stream
            with torch.cuda.stream(self._side):
                work = dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._early_work = work
        self.early_launches += 1

    @torch.no_grad()
    def all_reduce_mean(self) -> Tensor:
        early_done, self._early_work = self._early_work, None
        if early_done is None:
            self._ensure_views()
        # (also at world size 1: the same RCCL call on the same bucket, so that a 1-GPU run of the distributed path
        #  exercises everything an N-GPU run does)
        if dist.is_available() and dist.is_initialized():
            timed = self.timing and self.bucket.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rest = self.bucket if early_done is None else self.bucket[self.early_numel:]
            if rest.numel():
                dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.group)
            if early_done is not None:
                early_done.wait()  # (RCCL: the current stream waits for the collective's stream; gloo: the host does)
                if self._side is not None:
                    torch.cuda.current_stream().wait_stream(self._side)
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
