"""Deterministic synthetic weights and inputs (no network: no checkpoints, no datasets).

The reference zero-initialises conv2 / conv_out / attention out_proj / actor+critic heads
(models/blocks.py:59-60,139; inner_model.py:42; actor_critic.py:50-53), which makes a
default-init model output a spatial constant; parity fixtures therefore overwrite EVERY
tensor of the state dict with a name-keyed pseudo-random fill.  The fill depends only on
(seed, tensor name, shape), so the reference Agent (build container, fixture generation)
and this repo's Agent (GPU box, tests) get bit-identical weights as long as their
state-dict keys and shapes agree -- which tests/test_boundary.py checks.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import torch
from torch import Tensor


def _std_for(name: str, shape) -> Tuple[float, float]:
    """(mean, std) of the fill for a tensor."""
    if len(shape) >= 2:
        if "emb" in name:  # embeddings + the Fourier-feature buffer: unit normal like the reference
            return 0.0, 1.0
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return 0.0, fan_in ** -0.5
    if name.endswith("norm.weight"):  # GroupNorm gamma
        return 1.0, 0.1
    return 0.0, 0.05  # biases, GroupNorm beta


@torch.no_grad()
def fill_state_dict_(sd: Dict[str, Tensor], seed: int = 0) -> None:
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            continue
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        mean, std = _std_for(name, tuple(t.shape))
        v = torch.randn(tuple(t.shape), generator=g, dtype=torch.float32) * std + mean
        t.copy_(v.to(t.dtype))


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite all parameters and floating-point buffers of `module` in place."""
    fill_state_dict_(module.state_dict(), seed)


def synthetic_frames(g: torch.Generator, *shape) -> Tensor:
    """uint8-derived frames in [-1, 1] (what Episode.load produces, data/episode.py:36-41)."""
    u8 = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    return u8.float().div(255).mul(2).sub(1)


def synthetic_actions(g: torch.Generator, num_actions: int, *shape) -> Tensor:
    return torch.randint(0, num_actions, shape, generator=g, dtype=torch.long)


def initial_condition_batches(seed: int, batch: int, num_actions: int, t: int = 4, c: int = 3,
                              h: int = 64, w: int = 64) -> Iterator[Tuple[Tensor, Tensor]]:
    """Endless stream of (obs (B,T,C,H,W) in [-1,1], act (B,T) int64) initial conditions."""
    g = torch.Generator().manual_seed(seed)
    while True:
        yield synthetic_frames(g, batch, t, c, h, w), synthetic_actions(g, num_actions, batch, t)


def rew_end_train_batch(g, b=3, t=6):
    """Synthetic (B, T) segment for RewEndModel.forward: rewards in {-2..2}, one episode end (sample 1, step 3) with its
    `final_observation`, and a padded tail (sample 2).  Shared by tests/golden/make_golden.py and the parity test (tensors on CPU)."""
    obs = synthetic_frames(g, b, t, 3, 64, 64)
    act = synthetic_actions(g, 4, b, t)
    rew = torch.randint(-2, 3, (b, t), generator=g).float()
    end = torch.zeros(b, t, dtype=torch.long)
    end[1, 3] = 1
    mask = torch.ones(b, t, dtype=torch.bool)
    mask[1, 4:] = False  # steps after the end of the episode are padding
    mask[2, 4:] = False
    info = [{} for _ in range(b)]
    info[1]["final_observation"] = synthetic_frames(g, 1, 3, 64, 64)[0]
    return dict(obs=obs, act=act, rew=rew, end=end, trunc=torch.zeros(b, t, dtype=torch.long), mask_padding=mask, info=info,
                segment_ids=None)
