"""The world-model training step as ONE hipGraph (reference trainer.py:349-388: `loss, metrics = model(batch);
loss.backward(); clip_grad_norm_; opt.step(); opt.zero_grad()`).

At the reference's batch of 32 the step is launch-bound: ~600 kernel launches of 5-30 us each, issued by a Python
interpreter that needs longer per launch than the GPU needs per kernel.  Every dmd_* entry point is asynchronous on
torch's current stream, allocates nothing and synchronises nothing (include/diamond_hip.h), and the step itself is free
of host synchronisations (Denoiser.forward masks the loss arithmetically instead of gathering), so forward, backward,
clipping and the optimizer update record into one graph that is replayed per step on static input buffers:

    step = GraphedTrainStep(agent.denoiser, opt, max_grad_norm, example_batch)     # opt: capturable=True (+ fused=True)
    for batch in loader:
        loss, metrics = step(batch)          # == model(batch) ... opt.step(); opt.zero_grad() of the eager loop

Nothing of an earlier eager step's autograd graph may still be referenced when the step is constructed (e.g. a kept `loss`
tensor): its AccumulateGrad nodes belong to the stream they were created on, and autograd would synchronise the capturing
stream with it (torch warns "AccumulateGrad node's stream does not match"; the capture then aborts).

The optimizer: `torch.optim.AdamW(..., capturable=True, fused=True)`.  The capturable FOREACH form divides every tensor by its
0-dim bias corrections with one broadcast kernel each -- 2 x 236 launches of ~4 us, 1.5 ms of an 11.9 ms step
(profiles/r04_train_kernel_stats_foreach_adamw.csv); torch's fused form is one multi-tensor kernel: 10.35 ms.  It does not
bump `Tensor._version`; engine's optimizer post-step hook marks the weight caches stale instead.

What is captured is exactly the eager step's launch sequence.  Weight packing (engine.PackCache, blocks.FilmTable) is keyed
on parameter versions / that hook: the warm-up steps' optimizer updates make every copy stale, so every pack / transpose kernel is recorded
too: the step ends with an explicit refresh of every cache behind the optimizer update (one dmd_pack_jobs launch per cache,
in place), so after a replay the packed copies equal the parameters -- also for readers that never look a copy up again (the
sampler's captured imagination graphs read the packed buffers directly).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple

import torch
from torch import Tensor, nn


def _weight_caches(model: nn.Module):
    from . import engine as E

    for m in model.modules():
        if isinstance(getattr(m, "_cache", None), E.PackCache):
            yield m._cache
        film = getattr(m, "_film", None)
        if film is not None and hasattr(film, "refresh"):
            yield film


def _mark_weight_caches_stale(model: nn.Module) -> None:
    """Every packed copy of the model's parameters is stale (buffers and job tables are kept: engine.PackCache.invalidate)."""
    for c in _weight_caches(model):
        c.invalidate()


def _refresh_weight_caches(model: nn.Module) -> None:
    """Rebuild, in place and on the current stream, every packed copy that is stale (capturable: no allocation of a table,
    no upload -- the job tables exist after the warm-up)."""
    for c in _weight_caches(model):
        c.refresh()


class GraphedTrainStep:
    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, max_grad_norm: Optional[float], example_batch: Any,
                 warmup_steps: int = 3, fields: Tuple[str, ...] = ("obs", "act", "mask_padding")) -> None:
        assert torch.cuda.is_available(), "GraphedTrainStep needs the GPU"
        for group in optimizer.param_groups:
            assert group.get("capturable", False), \
                "construct the optimizer with capturable=True (its step counter must live on the device to be replayed)"
        self.model, self.optimizer, self.max_grad_norm, self.fields = model, optimizer, max_grad_norm, fields
        self.static = {k: getattr(example_batch, k).detach().clone() for k in fields}
        self._batch = type("StaticBatch", (), {})()
        for k, v in self.static.items():
            setattr(self._batch, k, v)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # eager warm-up: kernel attributes, optimizer state, parameter versions bumped
            for _ in range(max(1, warmup_steps)):
                self._eager()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # The packed copies are rebuilt at the END of the captured step, behind the optimizer update (not at its start): a
        # replay then leaves them equal to the parameters it leaves, and anything that reads them without a lookup -- the
        # sampler's captured imagination graphs -- is never a step behind.  So they have to be fresh going in:
        _mark_weight_caches_stale(model)
        _refresh_weight_caches(model)
        optimizer.zero_grad(set_to_none=True)  # the gradients of the captured step come from the graph's own pool
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads' HIP calls (the RCCL watchdog polls its events) must not invalidate this capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss, self.metrics = self._eager(zero=False)
        # (gradients stay allocated: the captured backward writes, not accumulates, into them at every replay)

    def _eager(self, zero: bool = True):
        loss, metrics = self.model(self._batch)
        loss.backward()
        if self.max_grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        _refresh_weight_caches(self.model)  # one dmd_pack_jobs launch per cache, on the updated weights
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        return loss.detach(), {k: (v.detach() if isinstance(v, Tensor) else v) for k, v in metrics.items()}

    def __call__(self, batch: Any) -> Tuple[Tensor, Dict[str, Any]]:
        for k, buf in self.static.items():
            src = getattr(batch, k)
            assert src.shape == buf.shape and src.dtype == buf.dtype, \
                f"batch.{k}: {tuple(src.shape)} {src.dtype}, captured with {tuple(buf.shape)} {buf.dtype} (static shapes)"
            buf.copy_(src, non_blocking=True)
        self.graph.replay()
        self._replays = getattr(self, "_replays", 0) + 1
        if self._replays % 64 == 0:  # the always-on audit of the packed copies (engine.WeightAudit): no lookup runs inside a replay
            for c in _weight_caches(self.model):
                for a in c.audits():
                    a.run()
        # The replayed optimizer update changed every parameter without bumping its `_version`; the packed copies were
        # rebuilt from the new values by the replay itself, in place: the stamps of capture time still describe them, and
        # graphs captured elsewhere from the same weights (DiffusionSampler.sample_ring_graphed) read the new values through
        # the same pointers.
        return self.loss, self.metrics
